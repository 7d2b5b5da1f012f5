#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_shim_session.py tests/test_gpu_reference_cases.py -x -q -m gpu 2>&1 | tail -12

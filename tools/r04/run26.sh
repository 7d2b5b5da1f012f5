#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_jit.py tests/test_gpu_typed.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5

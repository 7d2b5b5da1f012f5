#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_jit.py tests/test_gpu_typed.py tests/test_gpu_parity.py tests/test_gpu_reference_cases.py -x -q 2>&1 | grep -E "AssertionError|assert |Error|passed|failed" | head -10

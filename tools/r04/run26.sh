#!/bin/bash
python tools/host_time_probe.py C3 2>/dev/null | tail -1
python tools/host_time_probe.py C2 2>/dev/null | tail -1
timeout 1500 python -m pytest tests/test_gpu_threads.py tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_cluster_merge.py tests/test_gpu_shim_session.py tests/test_gpu_host_fuzz.py -x -q 2>&1 | tail -2

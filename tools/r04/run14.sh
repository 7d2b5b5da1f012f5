#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export VH_PLACE_TRIALS=1 VH_ABLATE_NO_PHASE2=1
python tools/c3_stream_probe.py 2>&1 | tail -1 | cut -c1-150
python tools/c3_stream_probe.py extra4 2>&1 | tail -1 | cut -c1-150
VH_JIT_ABLATE=1 python tools/c3_stream_probe.py 2>&1 | tail -1 | cut -c1-150
VH_JIT_ABLATE=1 python tools/c3_stream_probe.py extra4 2>&1 | tail -1 | cut -c1-150
VH_JIT_ABLATE=1 VH_BLOCKS_PER_CU=6 python tools/c3_stream_probe.py extra4 2>&1 | tail -1 | cut -c1-150
VH_JIT_ABLATE=3 python tools/c3_stream_probe.py extra4 2>&1 | tail -1 | cut -c1-150

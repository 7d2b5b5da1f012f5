#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py > gpurun_out/r03/bench_full.json 2> gpurun_out/r03/bench_full.err; tail -c 1500 gpurun_out/r03/bench_full.json; tail -3 gpurun_out/r03/bench_full.err
echo; echo "== hostprof"; python tools/hostprof.py 2>&1 | tail -1
echo "== overhead"; python tools/overhead.py 2>&1 | tail -4

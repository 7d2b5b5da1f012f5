#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export VH_PLACE_TRIALS=1
VH_TIMES=1 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>&1 | grep "vh times" | tail -3
python bench.py --steps 50 --warmup 5 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
(cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04/kt_c3 -o c3 -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu --no-check --no-reference-layout > /root/repo/gpurun_out/r04/kt_c3.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find gpurun_out/r04/kt_c3 -name "*_results.db" | head -1) gpurun_out/r04/c3_kernel_stats.csv; head -12 gpurun_out/r04/c3_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r04/kt_c3
VH_TIMES=1 python bench.py --workload C2 --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>&1 | grep "vh times" | tail -2
python bench.py --workload C2 --steps 50 --warmup 5 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"

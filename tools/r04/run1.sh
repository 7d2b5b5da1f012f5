#!/bin/bash
# round 4, call 1: where C5 and C3 stand at the round's first head, and two cheap questions (tail effect of the aggregation grid; small pools)
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tools/c5_probe.py C5 > $O/c5_base.json 2> $O/c5_base.err; tail -c 600 $O/c5_base.json
python tools/c5_probe.py C5t > $O/c5t_base.json 2>> $O/c5_base.err; tail -c 400 $O/c5t_base.json
for b in 2 8 16; do VH_HP_BPP=$b python tools/c5_probe.py C5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bpp $b', d['kernel_ms'], d['wall_ms'])"; done
for l in 0.5 0.85; do VH_HP_LOAD_G=$l VH_HP_LOAD_S=$l python tools/c5_probe.py C5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('load $l', d['kernel_ms'], d['wall_ms'])"; done
(cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$O/kt_c5 -o c5 -- python /root/repo/tools/c5_probe.py C5 125 3 > /root/repo/$O/kt_c5.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find $O/kt_c5 -name "*_results.db" | head -1) $O/c5_base_kernel_stats.csv; head -12 $O/c5_base_kernel_stats.csv | cut -c1-150
rm -rf $O/kt_c5
for i in 1 2 3; do VH_PLACE_TRIALS=1 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 notrials run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done
VH_PLACE_TRIALS=1 python tools/mall_probe.py 100 50 25 2>&1 | tail -5

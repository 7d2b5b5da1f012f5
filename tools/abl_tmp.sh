cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
Q="--no-cpu --no-check --no-reference-layout --no-cpu-parallel"
for i in 1 2 3 4; do
for B in 3 4 5; do
  export VH_BLOCKS_PER_CU=$B
  python bench.py $Q --steps 20 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bpc $B', $i, round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done
done
for B in 3 4; do
  export VH_BLOCKS_PER_CU=$B
  python bench.py $Q --steps 20 --warmup 3 --segments 125 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eighth bpc $B', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done

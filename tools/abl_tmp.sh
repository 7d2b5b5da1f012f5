mkdir -p gpurun_out/pring
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_typed.py tests/test_gpu_jit.py tests/test_gpu_pack.py tests/test_gpu_narrow.py -q -x ) > gpurun_out/pring/tests.log 2>&1; tail -5 gpurun_out/pring/tests.log
Q="--no-cpu --no-check --no-reference-layout --no-cpu-parallel"
for V in ring wave; do
  unset VH_TEST_HOOKS VH_NO_PART_RING
  if [ $V = wave ]; then export VH_TEST_HOOKS=1 VH_NO_PART_RING=1; fi
  for i in 1 2 3; do python bench.py $Q --steps 20 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', $i, round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['roofline']['kernel'][:90])"; done
  python bench.py $Q --steps 20 --warmup 3 --segments 125 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V eighth', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done

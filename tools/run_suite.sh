mkdir -p gpurun_out/verify
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 2000 python -m pytest tests -q -m gpu -x ) > gpurun_out/verify/gpu_default.log 2>&1; tail -4 gpurun_out/verify/gpu_default.log

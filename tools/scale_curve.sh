#!/bin/bash
# The strong-scaling curve of BASELINE.json's metric on one node (C4: C3's 1 B rows sharded over N GPUs, RCCL reduce of the partial tables):
# bench.py at N = 1, 2, 4, 8 back to back, one JSON line each, N capped at the GPUs present. usage: bash tools/scale_curve.sh [out.jsonl] [steps]
OUT=${1:-gpurun_out/scale_curve.jsonl}
STEPS=${2:-20}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && break
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup 3 --no-cpu >> "$OUT" 2> "$OUT.err$N"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps $STEPS --warmup 3 >> "$OUT" 2> "$OUT.err$N"
  fi
  tail -1 "$OUT" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=%d  %.1f G rows/s  %.3f ms/step  kernel %.3f ms  %s' % (d['n_gpus'], d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['parallelism'][:60]))"
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
if rows:
    base = rows[0]["value"]
    print("N  rows/s        speedup  efficiency")
    for d in rows: print("%d  %.4g  %.2fx    %.0f %%" % (d["n_gpus"], d["value"], d["value"] / base, 100 * d["value"] / base / d["n_gpus"]))
PY

#!/bin/bash
# usage: tools/r03_exp.sh <tag> <bench args...> -- reads env settings, one per line, from stdin ("-" = none) and runs the bench once per line
TAG=$1; shift
OUT=gpurun_out/r03/$TAG; mkdir -p $OUT
i=0
while IFS= read -r LINE; do
  i=$((i+1))
  [ "$LINE" = "-" ] && LINE=""
  env $LINE timeout 600 python bench.py --no-cpu --no-check "$@" > $OUT/$i.json 2> $OUT/$i.err
  python - "$OUT/$i.json" "$LINE" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-60s step %.3f ms  kernel %.3f ms  %s" % (sys.argv[2] or "(default)", d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel']))
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
done

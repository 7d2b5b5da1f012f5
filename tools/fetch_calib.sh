#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (tools/experiments/fetch_calib.hip). usage (GPU box): bash tools/fetch_calib.sh <out.json>
OUT=${1:-gpurun_out/fetch_calibration.json}
export TMPDIR=/tmp; REPO=$PWD; D=$REPO/gpurun_out/calib; rm -rf $D; mkdir -p $D
[ -x tools/experiments/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/experiments/fetch_calib tools/experiments/fetch_calib.hip
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C -d $D/$C -o c -- $REPO/tools/experiments/fetch_calib > $D/$C.log 2>&1)
done
python - "$D" "$OUT" <<'PY'
import glob, json, os, sqlite3, sys
d, out = sys.argv[1], sys.argv[2]
known = json.loads([l for l in open(os.path.join(d, "FETCH_SIZE.log")) if l.startswith("{")][-1])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(d, c, "**", "*_results.db"), recursive=True):
        for k, v in sqlite3.connect(f).execute("select kernel_name, value from counters_collection where counter_name = ?", (c,)):
            name = (k[:k.index(">(") + 1] if ">(" in k else k.split("(")[0]).replace("void ", "")
            res.setdefault(name, {})[c + "_KiB"] = v
rows = {}
def put(name, kernel_sub, counter, known_bytes):
    for k, v in res.items():
        if kernel_sub in k and counter + "_KiB" in v:
            rows[name] = {"kernel": k, "counter": counter, "counter_bytes": v[counter + "_KiB"] * 1024.0, "known_bytes": known_bytes, "ratio": v[counter + "_KiB"] * 1024.0 / known_bytes}
put("read 4 B/lane", "calib_read<unsigned int>", "FETCH_SIZE", known["read4"])
put("read 8 B/lane", "calib_read<unsigned int __vector(2)>", "FETCH_SIZE", known["read8"])
put("read 16 B/lane", "calib_read<unsigned int __vector(4)>", "FETCH_SIZE", known["read16"])
for k in list(res):
    if "calib_read<" in k and "ext_vector" not in k and "unsigned int>" not in k: pass
put("gather 32 B records (vs 128-byte lines touched)", "calib_gather32", "FETCH_SIZE", known["gather32_lines"])
# in-order gathers (a scan's survivors): against the bytes of the WHOLE array — 1.0 = every 64-byte half asked for is counted in full
put("gather 4 B, every other row, in order (vs array bytes)", "calib_gather_inorder<unsigned int>", "FETCH_SIZE", known["gather4_every2_array"])
put("gather 8 B, one row in 20, in order (vs array bytes)", "calib_gather_inorder<unsigned int __vector(2)>", "FETCH_SIZE", known["gather8_every20_array"])
put("write 16 B scattered", "calib_write16", "WRITE_SIZE", known["write16p"])
put("write whole 128 B lines", "calib_write128", "WRITE_SIZE", known["write128"])
json.dump({"rows": rows, "kernels_seen": sorted(res)}, open(out, "w"), indent=1)
for n, r in rows.items(): print("%-50s counter %.3e  known %.3e  ratio %.3f" % (n, r["counter_bytes"], r["known_bytes"], r["ratio"]))
print("kernels:", sorted(res))
PY
rm -rf $D

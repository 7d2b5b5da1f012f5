"""Worker of bench.py's courtesy CPU baseline #2 ("the same loop run segment-parallel on all physical
cores ... labelled not reference behaviour"). Test/bench infrastructure, like everything under oracle/.

Each worker builds its own few segments of the workload (same generator), then runs the emitted C++ twin over them
again and again inside a common wall-clock window and reports the rows it got through.
usage: python -m oracle.parallel_worker <workload> <segments> <first_segment> <start_epoch> <seconds>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    name, nseg, first, start, seconds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
    from oracle import cpu_twin
    from tests.parity import build_oracle_table
    from viyadb_amd import synth
    w = synth.WORKLOADS[name]()
    ot = build_oracle_table(w, nseg, w.segment_rows, row_base=first * w.segment_rows, csr=True)
    tw = cpu_twin.Twin(ot, w.query)
    now = getattr(w, "now", None)
    tw.run(now=now)
    missed = max(0.0, time.time() - start)          # ready after the window opened: this worker's share of it is short
    while time.time() < start:
        time.sleep(0.001)
    rows = 0
    busy = 0.0
    while time.time() < start + seconds:
        tw.run(now=now)
        rows += nseg * w.segment_rows
        busy += tw.last_seconds
    print(json.dumps({"rows": rows, "busy": busy, "late": max(0.0, time.time() - (start + seconds)), "missed": missed}))


if __name__ == "__main__":
    main()

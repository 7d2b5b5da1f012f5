#!/usr/bin/env python3
"""Is HBM uniformly fast? After a 92 GB filler (the C3 table's footprint), allocate N buffers of 2.6 GB and time, per buffer:
a streaming fill, a streaming read, and 50 M scattered 16-byte writes (the tuple append's pattern). Prints address + GB/s."""
import json, sys, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
filler_gb = int(sys.argv[2]) if len(sys.argv) > 2 else 92
dev = "cuda"
filler = [torch.empty(4 << 30, dtype=torch.uint8, device=dev) for _ in range(filler_gb // 4)]
for f in filler[:2]: f.zero_()
size = 2608465600 // 16 * 16
idx = (torch.randperm(size // 16, device=dev)[:50_000_000]).to(torch.int64)
val = torch.ones((50_000_000, 2), dtype=torch.int64, device=dev)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
bufs = []
for k in range(n):
    b = torch.empty(size, dtype=torch.uint8, device=dev)
    bufs.append(b)
    v = b.view(torch.int64).view(-1, 2)
    fill = timed(lambda: b.zero_())
    read = timed(lambda: v.sum())
    scat = timed(lambda: v.index_copy_(0, idx, val))
    print(json.dumps({"buf": k, "ptr": hex(b.data_ptr()), "fill_GBs": round(size / fill / 1e6, 1), "read_GBs": round(size / read / 1e6, 1),
                      "scatter_ms": round(scat, 3)}), flush=True)

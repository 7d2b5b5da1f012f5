#!/usr/bin/env python3
"""Does a PLAIN stream over a buffer depend on where the buffer lies? Sixteen 4 GB buffers (hipMalloc through torch, 1-3 GB spacers between
them), each read ten times by one reduction kernel; then pairs of them read together (a + b elementwise: two streams at fixed distance).
profiles/r06/NOTES.md, "Placement"."""
import json
import torch

torch.cuda.init()
N = 1 << 30          # int32 elements: 4 GB
bufs, spacers = [], []
for i in range(16):
    bufs.append(torch.ones(N, dtype=torch.int32, device="cuda"))
    spacers.append(torch.empty((1 + i % 3) << 30, dtype=torch.uint8, device="cuda"))


def timed(fn, n=10):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


single = [round(timed(lambda b=b: b.sum()), 4) for b in bufs]
print(json.dumps({"one stream over 4 GB, ms": single, "addresses": [hex(b.data_ptr()) for b in bufs]}))
out = torch.empty(N, dtype=torch.int32, device="cuda")
pairs = [round(timed(lambda i=i: torch.add(bufs[i], bufs[(i + 5) % 16], out=out)), 4) for i in range(16)]
print(json.dumps({"two streams read + one written (a[i] + a[i+5]), ms": pairs}))

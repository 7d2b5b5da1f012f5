"""Sharded-query scenarios (tests/test_gpu_distributed.py): per-rank data that makes the ranks' LOCAL views differ — disjoint
value ranges of the group columns, different selectivities, one shard with far more groups than the other, an empty shard —
so that only a plan agreed between the ranks gives identically indexed tables and matching collectives.
`build(name, ranks)` -> (oracle Table holding the segments of those ranks, query JSON, plan flags, groups_hint)."""
import numpy as np

from oracle import viya_oracle as vo

SEG = 40_000


def _table(dims, mets):
    return vo.Table({"name": "t", "segment_size": SEG, "dimensions": dims, "metrics": mets})


def _rng(name, rank, seg):
    return np.random.default_rng(abs(hash((name, rank, seg))) % (2 ** 32)) if False else np.random.default_rng([sum(map(ord, name)), rank, seg])


def build(name, ranks, world=2, grown=False):
    U = [{"name": "a", "type": "uint"}, {"name": "b", "type": "uint"}, {"name": "f", "type": "uint"}]
    M = [{"name": "v", "type": "long_sum"}, {"name": "count", "type": "count"}, {"name": "umin", "type": "uint_min"},
         {"name": "lmax", "type": "ulong_max"}, {"name": "d", "type": "double_sum"}]
    flags, hint = 0, 0
    q = {"type": "aggregate", "table": "t", "dimensions": ["a", "b"], "metrics": ["v", "count", "umin", "lmax"],
         "filter": {"op": "lt", "column": "f", "value": "50"}}
    t = _table(U, M)
    segs_per_rank = 2
    for rank in ranks:
        for s in range(segs_per_rank + (1 if grown and rank == 1 else 0)):
            r = _rng(name, rank, s)
            n = SEG - 17 * (rank + 1)
            a = r.integers(0, 60, n)
            b = r.integers(0, 50, n)
            f = r.integers(0, 100, n)
            if name == "disjoint_ranges":            # rank r only holds a in [1000 r, 1000 r + 60), b in [7 r, 7 r + 50)
                a = a + 1000 * rank
                b = b + 7 * rank
                if s >= segs_per_rank:               # the segment that appears between two queries: outside every range agreed so far
                    a = a + 700
                    b = b + 90
            elif name == "part_vs_global":           # rank 0: everything passes; rank 1: 1 % passes. 200 x 200 groups: too big for LDS
                a = r.integers(0, 200, n)
                b = r.integers(0, 200, n)
                f = r.integers(0, 50, n) if rank == 0 else np.where(r.random(n) < 0.01, 10, 90)
            elif name == "one_rank_overflows":       # hash table sized for a handful of groups: only rank 0 has many
                if rank != 0:
                    a = r.integers(0, 2, n)
                    b = r.integers(0, 2, n)
                else:
                    a = r.integers(0, 900, n)     # 45 000 groups against the 4096 slots a hint of 4 groups buys
            elif name == "empty_shard":
                if rank == 1:
                    n = 0
                    a, b, f = a[:0], b[:0], f[:0]
            elif name == "sparse_keys":              # float / wide keys: hash organisation
                pass
            elif name == "uniform":
                pass
            else:
                raise KeyError(name)
            n = len(a)
            dims = [a.astype(np.uint32), b.astype(np.uint32), f.astype(np.uint32)]
            mets = [r.integers(-10 ** 6, 10 ** 6, n).astype(np.int64), r.integers(1, 4, n).astype(np.uint32),
                    r.integers(0, 2 ** 32 - 1, n, dtype=np.uint64).astype(np.uint32), r.integers(0, 2 ** 63, n, dtype=np.uint64) * np.uint64(2) + np.uint64(rank),
                    r.integers(-1000, 1000, n) / 8.0]
            t.add_segment_arrays(dims, mets, None, n)
    if name == "one_rank_overflows":
        flags, hint = 1, 4                           # VH_PLAN_FORCE_HASH, tiny hint
    if name == "sparse_keys":
        q = dict(q, dimensions=["a", "lmaxkey"]) if False else q
        flags = 1
    return t, q, flags, hint

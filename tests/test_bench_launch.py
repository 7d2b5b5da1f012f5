"""`python bench.py --gpus N` from a plain shell must start: with no WORLD_SIZE / RANK in the environment it re-executes
itself as N ranks under torch.distributed.run on a free port of 127.0.0.1. No GPU here, so the ranks stop after the rendezvous
(--rendezvous-only); the same command without that flag is a `-m gpu` test (tests/test_gpu_distributed.py: two ranks on one GPU, the
callback transport, the "rccl" object says "gloo-fallback" and the line is marked degraded)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env["VH_BENCH_BACKEND"] = "gloo"
    return env


def test_plain_shell_gpus_2_launches_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--segments", "4", "--segment-rows", "20000", "--rendezvous-only"],
                       env=_plain_env(), capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    d = json.loads(lines[0])
    assert d["launched"] and d["n_ranks"] == 2
    assert sorted(x["rank"] for x in d["ranks"]) == [0, 1] and len({x["pid"] for x in d["ranks"]}) == 2
    assert "torch.distributed.run" in r.stderr             # the launcher says what it ran


def test_a_world_size_that_disagrees_with_gpus_is_refused():
    env = dict(_plain_env(), WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "must agree" in r.stderr

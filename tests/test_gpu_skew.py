"""Skewed shapes of the bench's queries (tools/skew_probe.py measures them at full size): a Zipf-like group key, a table
loaded in the order of a predicate column (the reference's own scenario: test/index.cc:44-75), a hot composite key under the hashed
partitioning. Each against the oracle on identical generated rows (the numpy twin of the generator modes is compared column by column
first), through the planner's own choice and through the partitioning organisations forced, compiled and pre-built kernels."""
import pytest

from tests.parity import check_workload
from viyadb_amd import capi, synth

pytestmark = pytest.mark.gpu

FORCE_PART = 64


@pytest.fixture(scope="module", autouse=True)
def _init():
    from viyadb_amd import executor
    executor.init(0)


@pytest.mark.parametrize("flags", [0, FORCE_PART, FORCE_PART | capi.PLAN_FORCE_JIT, FORCE_PART | capi.PLAN_NO_JIT, 1])
def test_zipf_group_key(flags):
    w = synth.c3z(segment_rows=200_000)
    res, st = check_workload(w, nseg=6, flags=flags)
    hot = res.keys[0] == 0                      # d0 = 0 holds a tenth of the rows: its 100 groups a tenth of the survivors
    assert 0.07 < res.states[1][hot].sum() / res.states[1].sum() < 0.13


@pytest.mark.parametrize("flags", [0, FORCE_PART | capi.PLAN_FORCE_JIT, FORCE_PART | capi.PLAN_NO_JIT])
def test_table_loaded_in_predicate_order(flags):
    w = synth.c3s(segment_rows=100_000, total_segments=20)        # 2 000 rows per value of d3: 50 values per segment, `d3 < 447` = the first 9 segments
    res, st = check_workload(w, nseg=20, flags=flags)
    assert res.scanned_segments == 9 and res.scanned_recs == 20 * 100_000        # 11 segments skipped by min / max, all counted as scanned rows (scan.cc:44-51)


HP = 1 | (1 << 18) | (1 << 20)      # hash organisation, compiled scan, hashed partitioning


@pytest.mark.parametrize("wl", ["C5h"])
@pytest.mark.parametrize("flags", [0, HP])
def test_hot_composite_key_with_distinct_counts(wl, flags):
    w = synth.WORKLOADS[wl](segment_rows=60_000)
    res, st = check_workload(w, nseg=4, flags=flags | capi.PLAN_CARD32)
    top = int(res.states[1].argmax())
    assert res.states[1][top] > 0.15 * res.states[1].sum()        # the hot (t, u): a tenth of the rows, a fifth of the survivors (u < 500 000 keeps half of the rest)

"""Fragment competition on the GPU (`adh_fragcomp`, csrc/adh_fragcomp.hip) against the CPU oracle
(`_compete_for_fragments`, alphadia/fragcomp/fragcomp.py:51-143): survivors must be identical - the rule is
greedy and order dependent, the GPU resolves it from per-PSM neighbour bitmaps instead of nested loops."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from alphadia_amd import runtime

    return runtime.get_context(0)


def _table(rng, n_win, per, rt_span, pool_size, max_frag=13, jitter=5e-6, gaps=0):
    """PSMs in processing order; `pool_size` distinct fragment masses: a small pool = many conflicts."""
    n = n_win * (per + gaps)
    rt = rng.uniform(0, rt_span, n).astype(np.float32)
    nfrag = rng.integers(0, max_frag, n)
    stop = np.cumsum(nfrag)
    start = stop - nfrag
    pool = np.sort(rng.uniform(200, 1800, pool_size)).astype(np.float32)
    mz = (rng.choice(pool, int(stop[-1])) * (1 + rng.normal(0, jitter, int(stop[-1])))).astype(np.float32)
    ws = np.arange(n_win) * (per + gaps)
    return ws, ws + per, rt, start, stop, mz


@pytest.mark.parametrize("seed, per, rt_span, pool", [
    (1, 900, 60, 400),      # few conflicts
    (2, 700, 20, 40),       # many conflicts, chains of waiting PSMs
    (3, 300, 2, 14),        # everything within the tolerance of everything: dense rows, long chains
    (4, 1500, 600, 25),     # sparse in RT, heavy in fragments
    (5, 33, 5, 10),         # rows shorter than one 16-bit word and just above
])
def test_random_tables_vs_oracle(ctx, oracle_lib, seed, per, rt_span, pool):
    rng = np.random.default_rng(seed)
    ws, we, rt, start, stop, mz = _table(rng, 7, per, rt_span, pool)
    got = ctx.fragcomp(ws, we, rt, start, stop, mz, 3, 15)
    exp = oracle_lib.fragcomp(ws, we, rt, start, stop, mz, 3, 15, n_threads=8)
    assert np.array_equal(got, exp) and 0 < exp.sum() < len(rt)
    st = ctx.fragcomp_stats()
    assert not st["serial"] and st["pairs"] > 0 and st["kernel_ms"] > 0


def test_serial_kernel_and_bitmap_path_agree(ctx, oracle_lib):
    rng = np.random.default_rng(11)
    ws, we, rt, start, stop, mz = _table(rng, 5, 400, 15, 30)
    a = ctx.fragcomp(ws, we, rt, start, stop, mz, 3, 15)
    os.environ["ADH_FRAGCOMP_SERIAL"] = "1"
    try:
        b = ctx.fragcomp(ws, we, rt, start, stop, mz, 3, 15)
        assert ctx.fragcomp_stats()["serial"]
    finally:
        del os.environ["ADH_FRAGCOMP_SERIAL"]
    assert np.array_equal(a, b)
    assert np.array_equal(a, oracle_lib.fragcomp(ws, we, rt, start, stop, mz, 3, 15, n_threads=4))


def test_rows_outside_windows_invalid_rows_and_odd_values(ctx, oracle_lib):
    rng = np.random.default_rng(21)
    # gaps: 20 rows after every window belong to no window and must come back untouched
    ws, we, rt, start, stop, mz = _table(rng, 6, 250, 12, 20, gaps=20)
    n = len(rt)
    rt[rng.integers(0, n, 40)] = np.nan
    rt[rng.integers(0, n, 10)] = np.inf
    rt[rng.integers(0, n, 10)] = -np.inf
    rt[rng.integers(0, n, 10)] = -0.0
    mz[rng.integers(0, len(mz), 200)] = 0.0      # unobserved fragments: 0 / 0 and x / 0 as numpy has them
    valid0 = rng.random(n) > 0.2                  # a fifth of the PSMs enter invalid
    got = ctx.fragcomp(ws, we, rt, start, stop, mz, 3, 15, valid=valid0)
    exp = oracle_lib.fragcomp(ws, we, rt, start, stop, mz, 3, 15, n_threads=4, valid=valid0)
    assert np.array_equal(got, exp)
    outside = np.ones(n, dtype=bool)
    for a, b in zip(ws, we):
        outside[a:b] = False
    assert outside.sum() == 120 and np.array_equal(got[outside], valid0[outside])
    assert not got[~valid0].any()


@pytest.mark.parametrize("rt_tol, ppm_tol", [(0.0, 15.0), (3.0, 0.0), (-1.0, 15.0), (3.0, 1e13), (np.inf, 15.0),
                                              (3.0, np.nan), (0.5, 3.0)])
def test_tolerance_corner_values(ctx, oracle_lib, rt_tol, ppm_tol):
    rng = np.random.default_rng(31)
    ws, we, rt, start, stop, mz = _table(rng, 3, 200, 8, 12)
    got = ctx.fragcomp(ws, we, rt, start, stop, mz, rt_tol, ppm_tol)
    exp = oracle_lib.fragcomp(ws, we, rt, start, stop, mz, rt_tol, ppm_tol, n_threads=4)
    assert np.array_equal(got, exp)


def test_long_fragment_lists(ctx, oracle_lib):
    """More fragments per PSM than the 32 the edge kernel keeps in LDS (transfer-library tables)."""
    rng = np.random.default_rng(41)
    ws, we, rt, start, stop, mz = _table(rng, 4, 150, 10, 300, max_frag=90)
    got = ctx.fragcomp(ws, we, rt, start, stop, mz, 3, 15)
    exp = oracle_lib.fragcomp(ws, we, rt, start, stop, mz, 3, 15, n_threads=8)
    assert np.array_equal(got, exp) and 0 < exp.sum() < len(rt)


def test_overlapping_windows_are_rejected(ctx):
    from alphadia_amd.runtime import HipBackendError

    rt = np.zeros(10, dtype=np.float32)
    z = np.zeros(10, dtype=np.int64)
    with pytest.raises(HipBackendError, match="overlap"):
        ctx.fragcomp(np.array([0, 4]), np.array([5, 10]), rt, z, z, np.zeros(1, np.float32), 3, 15)


def test_search_sized_table(ctx, oracle_lib):
    """100 000 PSMs in 60 windows over two hours, a tenth of them sharing fragments with a neighbour:
    the shape fdr.py:146-163 hands over (bench.py's fragment-competition leg at a tenth of its size)."""
    import synthetic as syn

    t = syn.make_competition_table(100_000, seed=7)
    got = ctx.fragcomp(t["window_start"], t["window_stop"], t["rt"], t["frag_start"], t["frag_stop"], t["mz"], 3, 15)
    exp = oracle_lib.fragcomp(t["window_start"], t["window_stop"], t["rt"], t["frag_start"], t["frag_stop"], t["mz"],
                              3, 15, n_threads=8)
    assert np.array_equal(got, exp)
    removed = len(exp) - int(exp.sum())
    assert 0.02 * len(exp) < removed < 0.2 * len(exp)
    st = ctx.fragcomp_stats()
    assert st["rounds"] <= 8 and not st["serial"]

"""timsTOF frame-major -> TOF-major transposition.  Integer work: bit-exact."""
import numpy as np
import pytest

import helpers as H


def numpy_transpose(tof_indices, push_indptr, n_tof, values):
    """CPU restatement of `_transpose` (alphadia/raw_data/bruker.py:155-280): a stable counting
    sort of the events by TOF index."""
    counts = np.diff(push_indptr)
    push_of = np.repeat(np.arange(len(counts), dtype=np.uint32), counts)
    order = np.argsort(tof_indices, kind="stable")
    tof_indptr = np.zeros(n_tof + 1, dtype=np.int64)
    np.cumsum(np.bincount(tof_indices, minlength=n_tof), out=tof_indptr[1:])
    return push_of[order], tof_indptr, values[order]


def test_numpy_restatement_matches_reference_golden():
    z = np.load(H.golden_path("transpose.npz"))
    p, ip, v = numpy_transpose(z["tof_indices"], z["push_indptr"], int(z["n_tof"]), z["values"])
    assert np.array_equal(p, z["out_push_indices"]) and p.dtype == z["out_push_indices"].dtype
    assert np.array_equal(ip, z["out_tof_indptr"]) and ip.dtype == z["out_tof_indptr"].dtype
    assert np.array_equal(v, z["out_values"]) and v.dtype == z["out_values"].dtype


@pytest.mark.gpu
def test_hip_transpose_matches_reference_golden_and_numpy():
    from alphadia_amd import runtime

    ctx = runtime.get_context(0)
    z = np.load(H.golden_path("transpose.npz"))
    p, ip, v = ctx.transpose_timstof(z["tof_indices"], z["push_indptr"], int(z["n_tof"]), z["values"])
    assert np.array_equal(p, z["out_push_indices"])
    assert np.array_equal(ip, z["out_tof_indptr"])
    assert np.array_equal(v, z["out_values"])
    # a larger run: 2e5 pushes, 2e7 events, 4e5 TOF bins
    rng = np.random.default_rng(5)
    n_push, n_tof = 200_000, 400_000
    counts = rng.poisson(100, n_push)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(ptr[-1])
    tof = rng.integers(0, n_tof, n).astype(np.uint32)
    val = rng.integers(1, 60000, n).astype(np.uint16)
    p, ip, v = ctx.transpose_timstof(tof, ptr, n_tof, val)
    ep, eip, ev = numpy_transpose(tof, ptr, n_tof, val)
    assert np.array_equal(p, ep) and np.array_equal(ip, eip) and np.array_equal(v, ev)
    # empty input and empty pushes
    p, ip, v = ctx.transpose_timstof(np.zeros(0, np.uint32), np.zeros(4, np.int64), 5, np.zeros(0, np.uint16))
    assert p.size == 0 and np.array_equal(ip, np.zeros(6, np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("slab_events", [1_000, 77_777, 1_999_999])
def test_hip_transpose_in_slabs_equals_one_sort(monkeypatch, slab_events):
    """Runs of 2^31 events and more are transposed slab by slab (whole pushes, < 2^31 events each: per-slab
    counts per TOF bin, then every slab's sorted runs scattered behind the earlier slabs' inside each bin).
    ADH_TRANSPOSE_SLAB_EVENTS shrinks the slabs so that a small run takes that path: same arrays as one sort."""
    from alphadia_amd import runtime

    ctx = runtime.get_context(0)
    rng = np.random.default_rng(9)
    n_push, n_tof = 30_000, 5_000
    counts = rng.poisson(60, n_push)
    counts[rng.integers(0, n_push, 500)] = 0  # empty pushes
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(ptr[-1])
    tof = rng.integers(0, n_tof, n).astype(np.uint32)
    tof[rng.integers(0, n, 1000)] = 17  # a crowded bin
    val = rng.integers(1, 60000, n).astype(np.uint16)
    ep, eip, ev = numpy_transpose(tof, ptr, n_tof, val)
    monkeypatch.setenv("ADH_TRANSPOSE_SLAB_EVENTS", str(slab_events))
    p, ip, v = ctx.transpose_timstof(tof, ptr, n_tof, val)
    assert np.array_equal(ip, eip) and np.array_equal(p, ep) and np.array_equal(v, ev)
    bad = tof.copy()
    bad[n // 2] = n_tof  # outside the table
    with pytest.raises(runtime.HipBackendError, match="n_tof"):
        ctx.transpose_timstof(bad, ptr, n_tof, val)

"""CPU: host-side mirror of the reference interface (no kernels involved)."""

import re
import subprocess

import numpy as np
import pandas as pd
import pytest

import helpers as H
from alphadia_amd import _abi
from alphadia_amd.scoring import (
    DEFAULT_FEATURE_COLUMNS,
    FRAGMENT_DF_COLUMNS,
    CandidateScoringConfig,
    OutputPsmDF,
    assemble_candidates,
    calculate_score_groups,
    collect_candidates,
    collect_fragments,
    merge_missing_columns,
    multiplex_candidates,
)


def test_score_groups_kat():
    # tests/unit_tests/search/scoring/test_scoring_utils.py:63-120
    base = dict(
        precursor_idx=np.arange(10),
        elution_group_idx=np.array([0, 0, 0, 0, 0, 1, 1, 1, 1, 1]),
        channel=np.array([0, 1, 2, 3, 0, 0, 1, 2, 3, 0]),
        decoy=np.array([0, 0, 0, 0, 1, 0, 0, 0, 0, 1]),
    )
    assert np.allclose(calculate_score_groups(pd.DataFrame(base))["score_group_idx"], np.arange(10))
    g = calculate_score_groups(pd.DataFrame(base), group_channels=True)
    assert np.allclose(g["score_group_idx"], [0, 0, 0, 0, 1, 2, 2, 2, 2, 3])
    with_rank = dict(base, rank=np.array([0, 1, 2, 3, 4, 0, 1, 2, 3, 4]))
    g = calculate_score_groups(pd.DataFrame(with_rank), group_channels=True)
    assert np.allclose(g["score_group_idx"], np.arange(10))
    df = pd.DataFrame(
        dict(
            precursor_idx=np.arange(10),
            elution_group_idx=np.array([0, 0, 0, 0, 1, 1, 1, 1, 0, 0]),
            channel=np.array([0, 0, 1, 1, 0, 0, 1, 1, 0, 0]),
            decoy=np.array([0, 0, 0, 0, 0, 0, 0, 0, 1, 1]),
            rank=np.array([0, 1, 0, 1, 0, 1, 0, 1, 0, 1]),
        )
    )
    g = calculate_score_groups(df, group_channels=True)
    assert np.allclose(g["score_group_idx"], [0, 0, 1, 1, 2, 3, 4, 4, 5, 5])


def test_multiplex_candidates_kat():
    # tests/unit_tests/search/scoring/test_scoring_utils.py:15-60
    cand = pd.DataFrame(
        {
            "elution_group_idx": [0, 0, 1], "precursor_idx": [0, 1, 3], "proba": [0.1, 0.4, 0.3],
            "rank": [0, 0, 0], "frame_start": [0, 0, 0], "frame_center": [0, 0, 0],
            "frame_stop": [0, 0, 0], "scan_start": [0, 0, 0], "scan_stop": [0, 0, 0],
            "scan_center": [0, 0, 0],
        }
    )
    prec = pd.DataFrame(
        {
            "precursor_idx": [0, 1, 2, 3, 4, 5], "elution_group_idx": [0, 0, 0, 1, 1, 1],
            "decoy": [0] * 6, "channel": [0, 4, 8, 0, 4, 8], "flat_frag_start_idx": [0] * 6,
            "flat_frag_stop_idx": [0] * 6, "charge": [2] * 6, "rt_library": [0] * 6,
            "mobility_library": [0] * 6, "mz_library": [0] * 6, "proteins": ["A"] * 6,
            "genes": ["A"] * 6,
        }
    )
    assert len(multiplex_candidates(cand, prec, channels=[])) == 0
    m = multiplex_candidates(cand, prec, channels=[0, 4, 8])
    assert m["precursor_idx"].tolist() == [0, 1, 2, 3, 4, 5]
    assert np.allclose(m["proba"], [0.1, 0.1, 0.1, 0.3, 0.3, 0.3])


def test_merge_missing_columns_kat():
    left = pd.DataFrame([{"idx": 1, "col_1": 0, "col_2": 0}])
    right = pd.DataFrame([{"idx": 1, "col_3": 0, "col_4": 0}])
    with pytest.raises(ValueError):
        merge_missing_columns(left, right, ["col_3"], on="idx_doesnt_exist")
    with pytest.raises(ValueError):
        merge_missing_columns(left, right, ["col_5"], on="idx")
    df = merge_missing_columns(left, right, ["col_3"], on="idx")
    assert list(df.columns) == ["idx", "col_1", "col_2", "col_3"]
    # both code paths (lookup / DataFrame.merge fallback) return a fresh RangeIndex, like DataFrame.merge
    left = pd.DataFrame({"idx": [3, 1, 2], "a": [30, 10, 20]}, index=[7, 5, 9])
    right = pd.DataFrame({"idx": [1, 2, 3], "b": [1.5, 2.5, 3.5]})
    fast = merge_missing_columns(left, right, ["b"], on="idx")
    slow = left.merge(right[["idx", "b"]], on="idx", how="left")
    pd.testing.assert_frame_equal(fast, slow)
    assert list(fast.index) == [0, 1, 2] and list(left.index) == [7, 5, 9]
    dup = pd.DataFrame({"idx": [1, 1, 2, 3], "b": [1.0, 1.1, 2.0, 3.0]})  # duplicate keys: the merge path
    assert list(merge_missing_columns(left, dup, ["b"], on="idx").index) == [0, 1, 2, 3]


def test_assemble_candidates_order_matches_reference():
    g = H.load_scoring_golden("handler_default")
    shuffled = g.candidates_df.sample(frac=1.0, random_state=5).reset_index(drop=True)
    soa = assemble_candidates(shuffled, g.library.precursor_df, "mz_library")
    assert np.array_equal(soa["precursor_idx"], g.z["order_precursor_idx"])
    assert np.array_equal(soa["rank"], g.z["order_rank"])
    assert soa["isotope_intensity"].shape == (len(shuffled), 4)
    assert soa["isotope_intensity"].dtype == np.float32 and soa["frame_start"].dtype == np.int64


@pytest.mark.parametrize("name, grouped", [("handler_default", False), ("multiplex", True)])
def test_assemble_candidates_fast_paths_equal_the_general_one(name, grouped):
    """A candidate table already in score-group order skips the sort, a library whose precursor_idx is its row
    number skips the search: both must give the columns the general path gives (a shuffled table, a library
    with a gap in its index)."""
    g = H.load_scoring_golden(name)
    pdf = g.library.precursor_df.sort_values("precursor_idx").reset_index(drop=True)
    assert np.array_equal(pdf["precursor_idx"].values, np.arange(len(pdf)))  # the dense case
    kw = dict(score_grouped=grouped, reference_channel=0 if grouped else -1)
    ref = assemble_candidates(g.candidates_df.sample(frac=1.0, random_state=9).reset_index(drop=True), pdf, "mz_library", **kw)
    in_order = g.candidates_df.iloc[np.lexsort((g.candidates_df["precursor_idx"].values, g.candidates_df["rank"].values,
                                                pdf["decoy"].values[g.candidates_df["precursor_idx"].values],
                                                g.candidates_df["elution_group_idx"].values))].reset_index(drop=True)
    fast = assemble_candidates(in_order, pdf, "mz_library", **kw)
    assert np.array_equal(fast["order"], np.arange(len(in_order)))
    used = np.unique(g.candidates_df["precursor_idx"].values)
    spare = np.setdiff1d(pdf["precursor_idx"].values, used)
    gap = pdf[pdf["precursor_idx"].values != spare[0]].reset_index(drop=True) if len(spare) else pdf.iloc[::-1]
    general = assemble_candidates(in_order, gap, "mz_library", **kw)
    for k in ref:
        if k in ("order", "prec_row"):
            continue
        assert np.array_equal(ref[k], fast[k]), k
        assert np.array_equal(ref[k], general[k]), k
        assert ref[k].dtype == fast[k].dtype == general[k].dtype, k
    assert np.array_equal(pdf["precursor_idx"].values[fast["prec_row"]], fast["precursor_idx"])
    assert np.array_equal(gap["precursor_idx"].values[general["prec_row"]], general["precursor_idx"])


def test_assemble_candidates_errors_and_reference_channel():
    g = H.load_scoring_golden("handler_default")
    bad = g.candidates_df.copy()
    bad.loc[0, "precursor_idx"] = 10_000_000
    with pytest.raises(ValueError):
        assemble_candidates(bad, g.library.precursor_df, "mz_library")
    dup = pd.concat([g.candidates_df.iloc[:1], g.candidates_df.iloc[:1]])
    with pytest.raises(ValueError, match="unique within a score group"):
        assemble_candidates(dup, g.library.precursor_df, "mz_library", score_grouped=True)
    # reference channel 4 never occurs -> every group is skipped (score_group.py:50-64)
    soa = assemble_candidates(g.candidates_df, g.library.precursor_df, "mz_library", reference_channel=4)
    assert (soa["flags"] == _abi.FLAG_SKIP).all()
    soa = assemble_candidates(g.candidates_df, g.library.precursor_df, "mz_library", reference_channel=0)
    assert (soa["flags"] == 0).all()


def test_collect_frames_follow_reference_column_contract():
    """Feed the reference's own OutputPsmDF (golden) through our collectors: the column
    names/order and derived columns must equal what the reference produced."""
    g = H.load_scoring_golden("handler_default")
    psm = OutputPsmDF({k: v.copy() for k, v in g.expected.items()})
    psm.valid = g.expected["valid"].astype(bool)
    fdf = collect_candidates(
        g.candidates_df, psm, g.library.precursor_df, "rt_library", "mobility_library", "mz_library"
    )
    assert list(fdf.columns[:46]) == DEFAULT_FEATURE_COLUMNS
    assert sorted(fdf.columns) == sorted(g.z["features_df_columns"].tolist())
    assert np.array_equal(fdf["precursor_idx"].values, g.z["features_df_precursor_idx"])
    assert np.array_equal(fdf["rank"].values, g.z["features_df_rank"])
    assert np.allclose(fdf["delta_rt"].values, g.z["features_df_delta_rt"], equal_nan=True)
    assert (fdf["n_K"] == 1).all() and (fdf["n_P"] == 2).all() and (fdf["n_R"] == 0).all()
    frdf = collect_fragments(psm, g.library.precursor_df)
    # the reference appends the merged columns in the order of a Python set (scoring/utils.py:236-240)
    assert list(frdf.columns[:14]) == FRAGMENT_DF_COLUMNS
    assert sorted(frdf.columns) == sorted(g.z["fragments_df_columns"].tolist())
    assert len(frdf) == int(g.z["fragments_df_n"])
    assert np.array_equal(frdf["precursor_idx"].values, g.z["fragments_df_precursor_idx"])
    assert np.array_equal(frdf["mz_observed"].values, g.z["fragments_df_mz_observed"])


def test_config_update_and_validate():
    cfg = CandidateScoringConfig()
    # defaults of config.py:73-85
    assert (cfg.top_k_fragments, cfg.top_k_isotopes, cfg.quant_window) == (12, 4, 3)
    assert cfg.quant_all is False and cfg.experimental_xic is False and cfg.reference_channel == -1
    cfg.update({"top_k_fragments": 9999, "precursor_mz_tolerance": 7.5, "quant_all": 1})
    assert cfg.top_k_fragments == 9999 and cfg.precursor_mz_tolerance == 7 and cfg.quant_all is True
    with pytest.raises(ValueError):
        cfg.update({"does_not_exist": 1})
    with pytest.raises(ValueError):
        cfg.update({"top_k_fragments": "many"})
    cfg.fragment_mz_tolerance = 500
    with pytest.raises(AssertionError):
        cfg.validate()
    j = CandidateScoringConfig().to_jitclass()
    assert j.precursor_mz_tolerance.dtype == np.float32 and j.top_k_fragments.dtype == np.uint32


def test_candidate_hash_kat():
    # tests/unit_tests/fragcomp/test_fragcomp.py:103-113
    from alphadia_amd.fragcomp import candidate_hash

    h = candidate_hash(np.array([1, 2, 1000000]), np.array([0, 1, 2]))
    assert all(h == np.array([1, 4294967298, 8590934592])) and h.dtype == np.uint64


def test_abi_header_symbols_are_exported():
    """The C-ABI library loads and exports every function include/alphadia_hip.h declares."""
    import os

    from alphadia_amd import runtime

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "alphadia_hip.h")).read()
    declared = set(re.findall(r"\b(adh_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 12
    nm = subprocess.run(["nm", "-D", "--defined-only", runtime.LIB_PATH], capture_output=True, text=True)
    exported = {line.split()[-1] for line in nm.stdout.splitlines() if line.strip()}
    assert declared <= exported, declared - exported
    assert declared == set(runtime.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(runtime.lib, name) is not None


def test_abi_struct_sizes():
    import ctypes as C

    assert C.sizeof(_abi.ScoringConfig) == 80  # 11 x 4 bytes, padding, sigma[2] + delta_mu[2] of the quadrupole (doubles)
    assert C.sizeof(_abi.Output) == 8 + 8 + 20 * 8  # n, top_k(+pad), 20 pointers
    assert C.sizeof(_abi.Candidates) == 8 + 14 * 8 + 8
    assert C.sizeof(_abi.AlphaRaw) == 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8


def test_gather_based_collection_equals_the_merges():
    """collect_candidates / collect_fragments with the row maps of assemble_candidates give the
    same frames as the reference's hash joins (scoring.py:394-580), also for a shuffled library."""
    from alphadia_amd.scoring import (OutputPsmDF, assemble_candidates, collect_candidates, collect_fragments)

    g = H.load_scoring_golden("multiplex")
    rng = np.random.default_rng(3)
    pdf = g.library.precursor_df.sample(frac=1.0, random_state=5).reset_index(drop=True)  # unsorted on purpose
    pdf["sequence"] = rng.choice(np.array(["PEPTIDEK", "KRPKRP", "AAAA", "PRK"], dtype=object), len(pdf))
    cand = g.candidates_df.copy()
    cand["score"] = rng.random(len(cand)).astype(np.float32)
    soa = assemble_candidates(cand, pdf, "mz_library", score_grouped=True, reference_channel=0)
    out = OutputPsmDF(dict(g.expected))
    slow_f = collect_candidates(cand, out, pdf, "rt_library", "mobility_library", "mz_library")
    seq = pdf["sequence"]
    fast_f = collect_candidates(cand, out, pdf, "rt_library", "mobility_library", "mz_library",
                                row_maps=(soa["order"], soa["prec_row"]),
                                sequence_counts=tuple(seq.str.count(a).values for a in ("K", "R", "P")))
    assert list(slow_f.columns) == list(fast_f.columns)
    assert (slow_f.dtypes == fast_f.dtypes).all()
    pd.testing.assert_frame_equal(slow_f, fast_f)
    slow_r = collect_fragments(out, pdf)
    fast_r = collect_fragments(out, pdf, prec_rows=soa["prec_row"])
    pd.testing.assert_frame_equal(slow_r, fast_r)
    assert len(fast_f) > 100 and len(fast_r) > 500


def _helpers_golden():
    z = np.load(H.golden_path("host_helpers.npz"))
    lib = pd.DataFrame({k[4:]: z[k] for k in z.files if k.startswith("lib_")})
    cand = pd.DataFrame({k[5:]: z[k] for k in z.files if k.startswith("cand_")})
    return z, lib, cand


@pytest.mark.parametrize("name, kw", [("default", {}), ("with_decoys", {"remove_decoys": False}),
                                      ("two_channels", {"channels": [0, 8]})])
def test_multiplex_candidates_matches_reference(name, kw):
    """Golden from the reference's ``multiplex_candidates`` (scoring/utils.py:114-200): shuffled
    library, several candidates per elution group, ties in proba, groups without candidates."""
    z, lib, cand = _helpers_golden()
    got = multiplex_candidates(cand, lib, **kw)
    cols = z[f"mult_{name}_columns"].tolist()
    assert list(got.columns) == cols
    assert len(got) == len(z[f"mult_{name}_precursor_idx"]) > 100
    for c in cols:
        assert np.array_equal(got[c].values, z[f"mult_{name}_{c}"]), c


@pytest.mark.parametrize("name, grouped", [("plain", False), ("grouped", True)])
def test_score_groups_match_reference(name, grouped):
    """``calculate_score_groups`` and the grouping inside ``assemble_candidates`` against the
    reference's own output (scoring/utils.py:269-410)."""
    z, lib, cand = _helpers_golden()
    got = calculate_score_groups(cand, group_channels=grouped)
    for c in ("precursor_idx", "rank", "score_group_idx"):
        assert np.array_equal(got[c].values, z[f"groups_{name}_{c}"]), c


def test_log_table_header_matches_its_generator():
    """adh_log_table.h (table of the float64 log of float32 arguments in the selection kernel) is generated:
    regenerate the 128 entries and compare with the committed header, and check the algorithm built on the
    table against an 80-bit log on a sample (<= 1 ulp of float64, identical after rounding to float32)."""
    import os
    import re
    import struct
    from fractions import Fraction

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "alphadia_amd", "csrc", "adh_log_table.h")).read()
    rows = re.findall(r"\{(\S+), (\S+)\},", text)
    assert len(rows) == 128
    ld = np.longdouble
    inv, tab = np.empty(128), np.empty(128)
    for i, (a, b) in enumerate(rows):
        inv[i], tab[i] = float.fromhex(a), float.fromhex(b)
        want_inv = float(ld(1) / (ld(1) + ld(i) / ld(128)))
        assert inv[i] == want_inv
        assert tab[i] == (float(-np.log(ld(want_inv))) if i else 0.0)

    def fma(a, b, c):  # correctly rounded
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    def log_f32(x32):
        bits = struct.unpack("<I", struct.pack("<f", x32))[0]
        e, mant = (bits >> 23) - 127, bits & 0x7FFFFF
        i = mant >> 16
        m = struct.unpack("<f", struct.pack("<I", mant | 0x3F800000))[0]
        r = fma(float(m), inv[i], -1.0)
        q = 1.0 / 9.0
        for c in (-1 / 8, 1 / 7, -1 / 6, 1 / 5, -1 / 4, 1 / 3, -1 / 2):
            q = fma(q, r, c)
        p = fma(r * r, q, r)
        return e * 6.93147180369123816490e-01 + ((tab[i] + p) + e * 1.90821492927058770002e-10)

    rng = np.random.default_rng(5)
    xs = np.concatenate([np.float32(1) + rng.random(400, dtype=np.float32) * np.float32(1e-3),
                         np.float32(1) + rng.random(400, dtype=np.float32),
                         rng.random(400, dtype=np.float32) * np.float32(60000) + np.float32(1),
                         np.array([np.nextafter(np.float32(1), np.float32(2))], dtype=np.float32)])
    for x in xs:
        mine, ref = log_f32(float(x)), float(np.log(ld(float(x))))
        assert abs(mine - ref) <= np.spacing(abs(ref)), x
        assert np.float32(mine) == np.float32(np.log(np.float64(x))), x


def test_threaded_object_gather_matches_numpy():
    """runtime.take_objects (adh_host_take_objects under the GIL) == NumPy's gather, reference counts included."""
    import sys

    from alphadia_amd import runtime

    rng = np.random.default_rng(3)
    src = np.array([f"PEPTIDE{i}" for i in range(5000)] + [None, ""], dtype=object)
    idx = rng.integers(0, len(src), size=600_000)
    probe = src[17]
    before = sys.getrefcount(probe)
    got = runtime.take_objects(src, idx, threads=8)
    if sys.implementation.name == "cpython" and sys.version_info[:2] <= (3, 11):
        assert runtime._OBJ_TAKE is True  # the threaded path ran (not NumPy's fallback)
    assert got.dtype == object and (got == src[idx]).all()
    assert all(a is b for a, b in zip(got[:2000], src[idx[:2000]]))
    assert sys.getrefcount(probe) - before == int((idx == 17).sum())
    del got
    assert sys.getrefcount(probe) == before
    # small inputs and non-object arrays take NumPy's path
    assert np.array_equal(runtime.take_objects(np.arange(10), np.array([3, 1])), np.array([3, 1]))


def test_assemble_rejects_an_index_a_library_with_duplicates_lacks():
    """a library whose precursor_idx column repeats a value ([0, 1, 1, 3]) passes the end-point test of the
    identity fast path; a candidate with the absent index 2 must still be refused, not mapped to a wrong row."""
    from alphadia_amd.scoring import assemble_candidates

    lib = pd.DataFrame({"precursor_idx": np.array([0, 1, 1, 3], np.uint32), "elution_group_idx": np.arange(4, dtype=np.uint32),
                        "decoy": np.zeros(4, np.uint8), "channel": np.zeros(4, np.uint8), "charge": np.full(4, 2, np.uint8),
                        "flat_frag_start_idx": np.arange(4, dtype=np.uint32) * 5,
                        "flat_frag_stop_idx": np.arange(1, 5, dtype=np.uint32) * 5,
                        "mz_library": np.array([500, 600, 700, 800], np.float32), "i_0": np.ones(4, np.float32)})
    lib = lib.iloc[[3, 0, 1, 2]].reset_index(drop=True)  # unsorted: the argsort branch runs
    cand = pd.DataFrame({"precursor_idx": np.array([2], np.uint32), "rank": np.zeros(1, np.uint8),
                         "elution_group_idx": np.zeros(1, np.uint32), "scan_start": [0], "scan_stop": [1], "scan_center": [0],
                         "frame_start": [0], "frame_stop": [20], "frame_center": [10]})
    with pytest.raises(ValueError, match="missing from precursors_flat"):
        assemble_candidates(cand, lib, "mz_library")
    ok = cand.assign(precursor_idx=np.array([3], np.uint32))
    soa = assemble_candidates(ok, lib, "mz_library")
    assert float(soa["precursor_mz"][0]) == 800.0


def test_the_median_network_of_the_fused_kernel_sorts_every_zero_one_input():
    """`ADH_SORT12_NETWORK` (adh_features_fast.hip) replaces the 16-input bitonic sort in the per-cycle median of the
    fused kernel (scoring_utils.py:120-152 takes np.median of <= 12 fragments): by the zero-one principle a
    compare-exchange network that sorts all 2^12 zero-one inputs sorts everything."""
    import os

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphadia_amd", "csrc",
                            "adh_features_fast.hip")).read()
    body = src[src.index("#define ADH_SORT12_NETWORK(CE)"):]
    body = body[:body.index("\n\n")]
    pairs = [(int(a), int(b)) for a, b in re.findall(r"CE\((\d+), (\d+)\)", body)]
    assert len(pairs) == 39 and all(0 <= a < b < 12 for a, b in pairs)
    bits = ((np.arange(1 << 12)[:, None] >> np.arange(12)[None, :]) & 1).astype(np.int8)
    for a, b in pairs:
        lo, hi = np.minimum(bits[:, a], bits[:, b]), np.maximum(bits[:, a], bits[:, b])
        bits[:, a], bits[:, b] = lo, hi
    assert (np.diff(bits, axis=1) >= 0).all()


def test_host_threads_follow_the_cgroup_quota(tmp_path, monkeypatch):
    """the host team of a scoring call is cut from the cgroup CPU quota (cpu.max), not from the
    hardware thread count, and LOCAL_WORLD_SIZE ranks share that quota.  adh_host_threads needs no GPU."""
    import os

    from alphadia_amd import runtime

    monkeypatch.delenv("ADH_HOST_THREADS", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    fake = tmp_path / "cpu.max"
    monkeypatch.setenv("ADH_CGROUP_CPU_MAX", str(fake))
    visible = len(os.sched_getaffinity(0))
    big = 1 << 40

    fake.write_text("max 100000\n")
    t, budget = runtime.host_threads(big)
    assert budget == visible and t == min(16, visible)

    # a quota of 16 cores on a box that shows 256 hardware threads (the pool's GPU boxes): eight ranks -> two threads
    fake.write_text("1600000 100000\n")
    t1, budget = runtime.host_threads(big)
    assert budget == min(16, visible) and t1 == min(16, visible)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    t8, _ = runtime.host_threads(big)
    assert t8 == max(min(16, visible) // 8, 1)
    total_threads_on_the_node = 8 * t8
    assert total_threads_on_the_node <= max(budget, 8)

    # fractional and tiny quotas: at least one thread
    fake.write_text("50000 100000\n")
    assert runtime.host_threads(big) == (1, 1)
    # small tables: a thread per 16 384 rows at least
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    fake.write_text("max 100000\n")
    assert runtime.host_threads(20000)[0] == 1
    assert runtime.host_threads(3 * 16384)[0] == min(3, visible)
    # the override
    monkeypatch.setenv("ADH_HOST_THREADS", "5")
    assert runtime.host_threads(big)[0] == 5

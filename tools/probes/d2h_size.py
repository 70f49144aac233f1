"""Scratch: copy-out rate of one process scoring tables of different sizes one after the other (ADH_DEBUG_TIMING=2 on
stderr): does a small table stay slow after a large one, and the other way round?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import synthetic as syn
from alphadia_amd import runtime  # noqa: E402
from alphadia_amd.distributed import slice_soa  # noqa: E402
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled  # noqa: E402

case = syn.make_case(int(os.environ.get("PRECURSORS", 1_000_000)), int(os.environ.get("CYCLES", 4800)), config_id=2, per_precursor=3, threads=os.cpu_count())
cfg = CandidateScoringConfig()
cfg.update(dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1, precursor_mz_tolerance=10,
                fragment_mz_tolerance=15, exclude_shared_ions=True, quant_window=3, quant_all=True,
                experimental_xic=True, top_k_fragments=12))
cfgj = cfg.to_jitclass()
ctx = runtime.get_context(0)
soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library", pool=ctx.pinned)
if os.environ.get("PREALLOC_EARLY"):  # the pool's output buffers exist before the run is staged
    from alphadia_amd import _abi as _abi_early

    for name, (shape, dt) in _abi_early.output_shapes(int(os.environ["PREALLOC_EARLY"]), 12).items():
        ctx.pinned.empty("out:" + name, shape, dt)
ctx.stage_run(case.dia)
ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"))
n = len(soa["precursor_idx"])
if os.environ.get("RESTORE_HOST_MB"):  # one large page-locked allocation, freed again, after the run is staged
    import ctypes as C

    p = C.c_void_p()
    assert runtime.lib.adh_host_alloc(C.byref(p), C.c_uint64(int(os.environ["RESTORE_HOST_MB"]) << 20)) == 0
    assert runtime.lib.adh_host_free(p) == 0
    print(f"==== {os.environ['RESTORE_HOST_MB']} MB of page-locked memory allocated and freed", file=sys.stderr, flush=True)
from alphadia_amd import _abi  # noqa: E402
import numpy as np  # noqa: E402

if os.environ.get("PREALLOC_ROWS"):  # the pool's output buffers sized for a larger table before anything is scored
    big = int(os.environ["PREALLOC_ROWS"])
    for name, (shape, dt) in _abi.output_shapes(big, 12).items():
        ctx.pinned.empty("out:" + name, shape, dt)
    print(f"==== pool pre-sized for {big} rows", file=sys.stderr, flush=True)
# ROWS: comma-separated groups "rows[:ENV=VALUE[:ENV=VALUE...]]" - the variables are set for that group only
for group in os.environ.get("ROWS", "1500000,3000000,1500000,600000,3000000").split(","):
    parts = group.split(":")
    rows = int(parts[0])
    extra = dict(p.split("=", 1) for p in parts[1:])
    for k in [k for k in os.environ if k.startswith("ADH_DEBUG_STOP")]:
        del os.environ[k]
    os.environ.update(extra)
    packed = pack_assembled(slice_soa(soa, 0, min(rows, n)))
    if os.environ.get("SLEEP"):
        import time

        time.sleep(float(os.environ["SLEEP"]))
    print(f"==== {rows} rows", file=sys.stderr, flush=True)
    import glob

    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_pcie")) + sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_fclk")) \
            + sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_socclk")) + sorted(glob.glob("/sys/class/drm/card*/device/current_link_speed")):
        try:
            txt = open(f).read().strip().replace("\n", " | ")
        except OSError as exc:
            txt = str(exc)
        print(f"   {f.split('/')[4]} {os.path.basename(f)}: {txt}", file=sys.stderr, flush=True)
    for _ in range(int(os.environ.get("CALLS", 3))):
        ctx.score_host(packed, cfgj, with_stats=bool(os.environ.get("WITH_STATS")), reuse_buffers=True)

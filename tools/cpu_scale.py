"""Thread scaling of the CPU oracle (the `cpu_baseline` of bench.py) on this host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # synthetic data generators
import synthetic as syn  # noqa: E402
from alphadia_amd.distributed import slice_soa  # noqa: E402
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled  # noqa: E402
from oracle import oracle  # noqa: E402



def numa_interleave(a):
    """mbind(MPOL_INTERLEAVE | MPOL_MF_MOVE) over the pages of a numpy array; returns the syscall result."""
    import ctypes

    libc = ctypes.CDLL(None, use_errno=True)
    nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")])
    if nodes < 2:
        return "single node"
    addr = a.ctypes.data
    start = (addr + 4095) & ~4095
    length = ((addr + a.nbytes) & ~4095) - start
    mask = ctypes.c_ulong((1 << nodes) - 1)
    rc = libc.syscall(237, ctypes.c_void_p(start), ctypes.c_ulong(length), 3, ctypes.byref(mask), ctypes.c_ulong(nodes + 1), 2)
    return rc if rc == 0 else f"errno {ctypes.get_errno()}"


case = syn.make_case(int(os.environ.get("N_PREC", 100000)), 4800, config_id=2, per_precursor=3, threads=os.cpu_count())
cfg = CandidateScoringConfig()
cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True, experimental_xic=True))
soa = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
cols = fragment_columns(case.library.fragment_df, "mz_library")
n = len(soa["precursor_idx"])
if os.environ.get("INTERLEAVE"):
    t = time.time()
    print("interleave:", numa_interleave(case.dia.mz_values), numa_interleave(case.dia.intensity_values),
          f"{time.time() - t:.2f}s", flush=True)
for th in [int(x) for x in os.environ.get("THREADS", "1,8,32,64,128,256").split(",")]:
    if th > (os.cpu_count() or 1):
        continue
    m = min(n, 3000 * th)
    p = pack_assembled(slice_soa(soa, 0, m))
    oracle.score(case.dia, cols, p, cfg.to_jitclass(), n_threads=th)
    t = time.time()
    oracle.score(case.dia, cols, p, cfg.to_jitclass(), n_threads=th, reuse=oracle.score.last_buffers)
    dt = time.time() - t
    print(f"{th:4d} threads: {m / dt:12,.0f} candidates/s", flush=True)

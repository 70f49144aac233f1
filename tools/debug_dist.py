import sys, os; sys.path.insert(0,'.')
import numpy as np, torch, torch.distributed as dist
from alphadia_amd import runtime, synthetic as syn
from alphadia_amd.distributed import DeviceTables, shard_bounds, slice_soa, all_gather_tables
from alphadia_amd.scoring import CandidateScoringConfig, assemble_candidates, fragment_columns, pack_assembled
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
case = syn.make_case(40000, 1200, config_id=2, per_precursor=3, threads=64)
print('[debug] mz checksum', float(case.dia.mz_values[::1000].astype(np.float64).sum()), float(case.dia.intensity_values[::1000].astype(np.float64).sum()), int(case.candidates_df['frame_start'].sum()), flush=True)
cfg = CandidateScoringConfig(); cfg.update(dict(top_k_isotopes=3, precursor_mz_tolerance=10, fragment_mz_tolerance=15, quant_all=True, experimental_xic=True))
cfgj = cfg.to_jitclass()
soa_all = assemble_candidates(case.candidates_df, case.library.precursor_df, "mz_library")
ctx = runtime.get_context(0)
ctx.stage_run(case.dia); ctx.stage_fragments(*fragment_columns(case.library.fragment_df, "mz_library"))
a, b = shard_bounds(soa_all["score_group_idx"], rank, world)
soa = slice_soa(soa_all, a, b)
ctx.upload_candidates(pack_assembled(soa))
tables = DeviceTables(-(-len(soa_all["precursor_idx"]) // world), 12, dev, with_stats=True)
out = tables.as_output(b - a)
st = torch.cuda.current_stream().cuda_stream  # 0 = default stream
for it in range(3):
    tables.zero_()
    ctx.score_uploaded(cfgj, out, st)
    v0 = -1
    g = all_gather_tables(tables.buffer, world)
    torch.cuda.synchronize()
    v1 = tables.to_host()["valid"][: b - a].mean()
    vg = [tables.to_host(g[r])["valid"].mean() for r in range(world)]
    print(rank, it, "before", v0, "after", v1, "gathered", vg, flush=True)
dist.destroy_process_group()

"""Which HIP / HSA / RCCL copies end up in a process that imports torch AND opens a communicator
through libalphadia_hip.so (torch wheels bundle their own ROCm libraries)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1] if len(sys.argv) > 1 else "ours_first"
if order == "torch_first":
    import torch  # noqa: F401
from alphadia_amd import runtime

ctx = runtime.get_context(0)
if order == "ours_first":
    import torch  # noqa: F401
    torch.nn.Linear(4, 4)
try:
    ctx.comm_init(0, 1, max_rows_per_rank=1000)
    print(order, "comm ok", ctx.comm_info() if hasattr(ctx, "comm_info") else "")
except Exception as e:  # noqa: BLE001
    print(order, "comm FAILED:", str(e)[:200])
seen = set()
for line in open("/proc/self/maps"):
    p = line.split()[-1]
    if any(k in p for k in ("rccl", "amdhip", "hsa-runtime", "alphadia_hip")) and p not in seen:
        seen.add(p)
        print("  ", p)

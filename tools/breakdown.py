"""Per-stage timings (HIP events inside libmipme + around the C-ABI calls) of eager steps for a water box of a given size.
    python tools/breakdown.py <n_side> <n_mesh>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
import bench
from torchpme_amd import workloads, ops, _lib
n_side, n_mesh = int(sys.argv[1]), int(sys.argv[2])
w = workloads.water_box(n_side=n_side, n_mesh=n_mesh)
f = bench.Frame(w, torch.device("cuda", 0))
for _ in range(5): f.step()
ops.PROFILE = {}
_lib.profile_enable(True)
for _ in range(10): f.step()
torch.cuda.synchronize()
print(w.n_atoms, "atoms", w.n_pairs, "pairs", n_mesh, "^3")
print("stages us:", {k: round(ms / calls * 1000, 1) for k, (calls, ms) in sorted(_lib.profile_report().items(), key=lambda kv: -kv[1][1])})
print("calls  us:", {k: round(sum(a.elapsed_time(b) for a, b in v) / len(v) * 1000, 1) for k, v in ops.PROFILE.items()})

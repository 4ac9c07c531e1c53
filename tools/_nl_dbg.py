import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torchpme_amd as tpa
from torchpme_amd import workloads
w = workloads.water_box()
pos = torch.tensor(w.positions, device="cuda", dtype=torch.float32); cell = torch.tensor(w.cell, device="cuda", dtype=torch.float32)
nl = tpa.NeighborStream(pos, cell, w.cutoff, row_capacity=400)
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
print(os.environ.get("MIPME_NL_DEBUG"), "update us", timed(nl.update))

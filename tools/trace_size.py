"""One system size of tools/size_sweep.py, a few replays (for a rocprofv3 kernel trace):  python tools/trace_size.py 56 256"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
import bench  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

n_side, n_mesh = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
w = workloads.water_box(n_side=n_side, n_mesh=n_mesh)
f = bench.Frame(w, dev)
g = tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos, f.pairs, f.shifts)
for _ in range(30):
    g()
torch.cuda.synchronize()
print(w.n_atoms, w.n_pairs, float(g.energy))

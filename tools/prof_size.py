"""Kernel trace of the graph-replayed step at one size of tools/size_sweep.py; run under rocprofv3.
    python tools/prof_size.py n_side n_mesh [replays]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
import bench  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

n_side, n_mesh = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
w = workloads.water_box(n_side=n_side, n_mesh=n_mesh)
f = bench.Frame(w, torch.device("cuda", 0))
g = tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos, f.pairs, f.shifts)
for _ in range(n):
    g.graph.replay()
torch.cuda.synchronize()
print(w.n_atoms, w.n_pairs, float(g.energy))

"""Kernel traces of the less-travelled paths (run under rocprofv3 --kernel-trace --stats): which = stress | charges | ewald |
coldlist.    python tools/prof_paths.py which [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
f = bench.Frame(workloads.water_box(), torch.device("cuda", 0))
if which == "stress":  # graph replay with the cell gradient
    g = tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos.detach(), f.pairs, f.shifts, cell_gradient=True)
    for _ in range(n):
        g()
elif which == "charges":  # eager, gradient w.r.t. the charges as well (charge models)
    for _ in range(n):
        f.pos.grad = None
        q = f.q.clone().requires_grad_(True)
        d = tpa.pair_distances(f.pos, f.pairs, f.cell, f.shifts)
        V = f.calc(q, f.cell, f.pos, f.pairs, d)
        (q * V).sum().backward()
elif which == "coldlist":
    for _ in range(n // 4):
        f.step_cold_list("list")
elif which == "ewald":
    w = workloads.water_box(n_side=8)
    fe = bench.Frame(w, torch.device("cuda", 0))
    calc = tpa.EwaldCalculator(tpa.CoulombPotential(smearing=w.smearing), lr_wavelength=2.0).to(torch.float32)
    for _ in range(n):
        fe.pos.grad = None
        d = tpa.pair_distances(fe.pos, fe.pairs, fe.cell, fe.shifts)
        V = calc(fe.q, fe.cell, fe.pos, fe.pairs, d)
        (fe.q * V).sum().backward()
torch.cuda.synchronize()
print(which, "done")

"""Repeatability of the device neighbour stream: N rebuilds of the same positions must give identical rows; also the spread of
fp32 energies / forces over repeated graph replays with refreshes in between."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cut = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
w = workloads.water_box(n_side=n_side, n_mesh=32, cutoff=cut)
dev = torch.device("cuda")
for dt in (torch.float32, torch.float64):
    pos = torch.tensor(w.positions, device=dev, dtype=dt)
    cell = torch.tensor(w.cell, device=dev, dtype=dt)
    q = torch.tensor(w.charges, device=dev, dtype=dt)
    nl = tpa.NeighborStream(pos, cell, w.cutoff)
    torch.cuda.synchronize()
    N, cap = nl.n_atoms, nl.row_capacity
    rp0 = nl.row_ptr.clone()
    idx = torch.arange(N * cap, device=dev).view(N, cap)
    lens = (rp0[: 3 * N].view(N, 3)[:, 2] - rp0[: 3 * N].view(N, 3)[:, 0]).long()
    valid = idx - idx[:, :1] < lens[:, None]
    w0 = torch.where(valid, nl.words[: N * cap].view(N, cap), torch.zeros((), dtype=torch.int32, device=dev))
    bad = 0
    for it in range(300):
        nl.update()
        w1 = torch.where(valid, nl.words[: N * cap].view(N, cap), torch.zeros((), dtype=torch.int32, device=dev))
        if not torch.equal(nl.row_ptr, rp0) or not torch.equal(w1, w0):
            bad += 1
    print(dt, "rebuilds that differ:", bad, "of 300; entries", int(lens.sum()), "expected", 2 * w.n_pairs)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=5)
    step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff)
    Es, Fs = [], []
    for it in range(100):
        if it % 3 == 0:
            step.refresh()
        E, F = step()
        Es.append(E.item())
        Fs.append(F.clone())
    Fs = torch.stack(Fs)
    print("  E spread", max(Es) - min(Es), "rel", (max(Es) - min(Es)) / abs(Es[0]), " F spread rel-L2",
          float((Fs - Fs[0]).norm(dim=(1, 2)).max() / Fs[0].norm()))

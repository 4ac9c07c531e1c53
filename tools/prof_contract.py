"""Kernel trace of the energy step with the whole autograd contract (binned step, cfg3); run under rocprofv3.
    MODE=F | Fq | Fqc (default)   LIVE=0|1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
from torchpme_amd import workloads
w = workloads.water_box()
dev = torch.device("cuda"); dt = torch.float32
pos, cell, q = (torch.tensor(x, device=dev, dtype=dt) for x in (w.positions, w.cell, w.charges))
calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
mode = os.environ.get("MODE", "Fqc")
kw = dict(charge_gradient="q" in mode, cell_gradient="c" in mode)
if os.environ.get("LIVE", "0") == "1":
    step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff, **kw)
else:
    step = tpa.GraphedEnergyForces(calc, q, cell, pos, torch.tensor(w.pairs, device=dev), torch.tensor(w.shifts, device=dev, dtype=dt), **kw)
for _ in range(400): step.graph.replay()
torch.cuda.synchronize()
print(mode, float(step.energy))

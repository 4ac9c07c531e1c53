"""ms per replay of the graph-captured energy step at cfg3 (HIP events around 2 000 replays, three repeats):
    MODE=F | Fq | Fqc (default)   LIVE=0|1   [MIPME_LIB=<variant library>]      -- the A/B companion of tools/prof_contract.py"""
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
for _ in range(200): step.graph.replay()
torch.cuda.synchronize()
out = []
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(2000): step.graph.replay()
    b.record()
    torch.cuda.synchronize()
    out.append(round(a.elapsed_time(b) / 2000, 5))
print(mode, "live" if os.environ.get("LIVE", "0") == "1" else "binned", out, "ms  E =", float(step.energy))

"""Kernel trace of the live-bin step (GraphedEnergyForces(neighbors=cutoff)) vs the list-based step at cfg3; run under rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
from torchpme_amd import workloads
w = workloads.water_box()
dev = torch.device("cuda"); dt = torch.float32
pos = torch.tensor(w.positions, device=dev, dtype=dt); cell = torch.tensor(w.cell, device=dev, dtype=dt); q = torch.tensor(w.charges, device=dev, dtype=dt)
calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
live = os.environ.get("LIVE", "1") == "1"
step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff, live_bins=live)
for _ in range(300): step.graph.replay()
torch.cuda.synchronize()
for _ in range(20): step.refresh_graph.replay()
torch.cuda.synchronize()
print("live" if step._live is not None else "not live", float(step.energy))

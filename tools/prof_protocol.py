"""The reference's TuningTimings protocol (tuning/tuner.py:337-373) on the cfg3 box: wall time per call, and (PROFILE=1) a cProfile
of the host side.  Run under rocprofv3 for the kernel list."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
from torchpme_amd import workloads
w = workloads.water_box()
dev = torch.device("cuda"); dt = torch.float32
pos0, cell0, q0 = (torch.tensor(x, device=dev, dtype=dt) for x in (w.positions, w.cell, w.charges))
pairs = torch.tensor(w.pairs, device=dev); shifts = torch.tensor(w.shifts, device=dev, dtype=dt)
calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
d_fixed = tpa.pair_distances(pos0, pairs, cell0, shifts).detach().clone()
def protocol():
    positions, cell, charges = pos0.clone(), cell0.clone(), q0.clone()
    for t in (positions, cell, charges):
        t.requires_grad_(True)
    result = calc.forward(positions=positions, charges=charges, cell=cell, neighbor_indices=pairs, neighbor_distances=d_fixed)
    value = result.sum()
    value.backward(retain_graph=True)
    return positions.grad, cell.grad, charges.grad
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(20): protocol()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): protocol()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"protocol: host-side {1e3 * t_host / n:.4f} ms/call, with final sync {1e3 * t_all / n:.4f} ms/call")
if os.environ.get("PROFILE") == "1":
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n): protocol()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)

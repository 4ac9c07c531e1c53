"""GPU timing probe: energy + forces (+ stress: cell gradient) eager steps on the cfg3 water box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
import bench
from torchpme_amd import workloads, ops
f = bench.Frame(workloads.water_box(), torch.device("cuda", 0))
def step(with_cell):
    f.pos.grad = None
    cell = f.cell.clone().requires_grad_(with_cell)
    d = tpa.pair_distances(f.pos, f.pairs, cell, f.shifts)
    V = f.calc(f.q, cell, f.pos, f.pairs, d)
    E = tpa.weighted_sum(V, f.q)
    E.backward()
    return cell.grad
for with_cell in (False, True):
    for _ in range(10): step(with_cell)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): g = step(with_cell)
    torch.cuda.synchronize()
    print("cell gradient" if with_cell else "forces only  ", "%.3f ms/step (eager)" % ((time.perf_counter() - t0) * 10))
ops.PROFILE = {}
from torchpme_amd import _lib
_lib.profile_enable(True)
for _ in range(20): step(True)
torch.cuda.synchronize()
print({k: round(ms / calls * 1000, 1) for k, (calls, ms) in _lib.profile_report().items()})
print({k: round(sum(a.elapsed_time(b) for a, b in v) / len(v) * 1000, 1) for k, v in ops.PROFILE.items()})

_lib.profile_enable(False)
ops.PROFILE = None
# the same step replayed as a HIP graph (GraphedEnergyForces), without and with the cell gradient
for with_cell in (False, True):
    g = tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos.detach(), f.pairs, f.shifts, cell_gradient=with_cell)
    for _ in range(20): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): g()
    torch.cuda.synchronize()
    print("graph replay,", "energy + forces + dE/dcell" if with_cell else "energy + forces          ",
          "%.4f ms/step" % ((time.perf_counter() - t0) / 300 * 1e3))

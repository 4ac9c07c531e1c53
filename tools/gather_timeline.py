"""Workgroup timeline of the gather launches (live-bin step and binned step): library built with
``bash tools/build_variant.sh gtimeline -DMIPME_WG_TIMELINE=2`` (the gather stamps are a build of their own: the spread launch shares the
buffer); run with MIPME_LIB=<that .so>.  This is how the three
slow-downs of the first live gather were found (profiles/r03_experiments.txt): scratch arrays, vector loads from the kernarg
segment, and an LDS-promoted private array that made every wave read the AQL dispatch packet in host memory."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
from torchpme_amd import workloads, _lib
w = workloads.water_box()
dev = torch.device("cuda"); dt = torch.float32
pos = torch.tensor(w.positions, device=dev, dtype=dt); cell = torch.tensor(w.cell, device=dev, dtype=dt); q = torch.tensor(w.charges, device=dev, dtype=dt)
calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff)
for _ in range(5): step._live.step()
torch.cuda.synchronize()
lib = _lib.load()
n = 512
buf = np.zeros(n * 4, dtype=np.int64)
lib.mipme_debug_wg_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.mipme_debug_wg_timeline(buf.ctypes.data, n * 4) == 0
t = buf.reshape(n, 4)
start, end = t[:, 0] * 0.01, t[:, 1] * 0.01
t0 = start.min(); start -= t0; end -= t0
d = end - start
print("gather WGs: start %.2f..%.2f end %.2f..%.2f lifetime mean %.2f min %.2f max %.2f" % (start.min(), start.max(), end.min(), end.max(), d.mean(), d.min(), d.max()))
o = np.argsort(d)[::-1][:10]
print("longest:", [(int(k), round(float(start[k]), 2), round(float(d[k]), 2)) for k in o])
print("deciles of lifetime", np.round(np.quantile(d, np.linspace(0, 1, 11)), 2))
print("deciles of start", np.round(np.quantile(start, np.linspace(0, 1, 11)), 2))
ph = np.zeros(512 * 8, dtype=np.int64)
lib.mipme_debug_wg_phase.argtypes = [C.c_void_p, C.c_int]
assert lib.mipme_debug_wg_phase(ph.ctypes.data, 512 * 8) == 0
ph = ph.reshape(512, 8)
T0 = t[:, 0]
it2 = ph[:, 6] >= 2
late = np.arange(512) >= 256
p1 = (ph[:, 1] - T0) * 0.01
for nm, m in (("iters1", ~it2), ("iters2", it2), ("early", ~late), ("late", late), ("early&iters1", ~late & ~it2), ("early&iters2", ~late & it2), ("late&iters1", late & ~it2), ("late&iters2", late & it2)):
    print(nm, int(m.sum()), "lifetime mean %.2f" % d[m].mean(), "first load mean %.2f min %.2f max %.2f" % (p1[m].mean(), p1[m].min(), p1[m].max()))
for k in o[:4].tolist() + np.argsort(d)[:3].tolist():
    print("wg", k, "iters", ph[k, 6], ph[k, 7], "phases (us after start):", [round(float((ph[k, j] - T0[k]) * 0.01), 2) for j in range(6)], "end", round(float(d[k]), 2))


# the binned (list-based) step: its gather stamps last
pairs = torch.tensor(w.pairs, device=dev); shifts = torch.tensor(w.shifts, device=dev, dtype=dt)
old = tpa.GraphedEnergyForces(calc, q, cell, pos, pairs, shifts)
for _ in range(3): old()
torch.cuda.synchronize()
assert lib.mipme_debug_wg_timeline(buf.ctypes.data, n * 4) == 0
t = buf.reshape(n, 4)
start, end = t[:, 0] * 0.01, t[:, 1] * 0.01
t0 = start.min(); start -= t0; end -= t0
d = end - start
print("OLD gather WGs: start %.2f..%.2f end %.2f..%.2f lifetime mean %.2f min %.2f max %.2f" % (start.min(), start.max(), end.min(), end.max(), d.mean(), d.min(), d.max()))
print("deciles of lifetime", np.round(np.quantile(d, np.linspace(0, 1, 11)), 2))

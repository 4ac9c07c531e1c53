"""Neighbour-list refresh on the device (cfg3 by default): time of the two library calls, of the captured refresh graph, of
the first step after a refresh, and the amortised MD step at refresh intervals 10 / 20.  Usage: python tools/time_refresh.py [n_side]"""
import ctypes as C
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import _lib, workloads  # noqa: E402

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 22
skin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
w = workloads.water_box(n_side=n_side)
dev = torch.device("cuda")
dt = torch.float32
pos = torch.tensor(w.positions, device=dev, dtype=dt)
cell = torch.tensor(w.cell, device=dev, dtype=dt)
q = torch.tensor(w.charges, device=dev, dtype=dt)
calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


nl = tpa.NeighborStream(pos, cell, w.cutoff + skin)
nl.check(synchronize=True)
print(f"N={w.n_atoms} half pairs={w.n_pairs} entries={nl.n_entries} row_capacity={nl.row_capacity} longest={nl.longest_row} "
      f"n_cells={list(nl._desc.n_cells)} reach={list(nl._desc.reach)}")
lib = _lib.load()
st = _lib.current_stream(dev)
d = nl._desc
t_bin = timed(lambda: lib.mipme_nl_bin(st, 0, C.byref(d), nl.n_atoms, pos.data_ptr(), nl._ws.data_ptr()))
t_all = timed(lambda: nl.update())
print(f"eager: bin {t_bin*1e3:.1f} us, bin + stream + report {t_all*1e3:.1f} us")
t0 = time.perf_counter()
p, s, dd = tpa.neighbor_list_device(pos, cell, w.cutoff)
torch.cuda.synchronize()
print(f"reference-format list (pairs, shifts, distances; {len(p)} pairs): {(time.perf_counter()-t0)*1e3:.2f} ms")
t0 = time.perf_counter()
p, s, dd = tpa.neighbor_list_device(pos, cell, w.cutoff)
torch.cuda.synchronize()
print(f"  second call: {(time.perf_counter()-t0)*1e3:.2f} ms")

step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff + skin)
step_list = tpa.GraphedEnergyForces(calc, q, cell, pos, torch.tensor(w.pairs, device=dev), torch.tensor(w.shifts, device=dev, dtype=dt))
E1, F1 = step()
E2, F2 = step_list()
print(f"E stream {E1.item():.6f}  E list {E2.item():.6f}  max|dF| {float((F1-F2).abs().max()):.3e} (|F|max {float(F2.abs().max()):.3f})")
t_step = timed(lambda: step.graph.replay(), 200, 20)
t_step_list = timed(lambda: step_list.graph.replay(), 200, 20)
t_ref = timed(lambda: step.refresh_graph.replay(), 50, 5)
print(f"graph: step (stream rows) {t_step*1e3:.1f} us, step (sorted list) {t_step_list*1e3:.1f} us, refresh {t_ref*1e3:.1f} us")


def md(interval, n=200):
    def run():
        for it in range(n):
            if it % interval == 0:
                step.refresh_graph.replay()
            step.graph.replay()
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for k in (10, 20):
    print(f"amortised step, refresh every {k}: {md(k)*1e3:.1f} us")
step.stream.check(synchronize=True)

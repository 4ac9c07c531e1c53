"""Throughput of the graph-replayed energy + forces step against system size (water boxes at 0.1 atoms / A^3, rc = 9 A,
P3M n = 5, fp32; mesh = power of two with spacing near 1 A).  Prints one line per size; run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
import bench  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

dev = torch.device("cuda", 0)
print(f"{'atoms':>9} {'pairs':>11} {'mesh':>5} {'ms/step':>9} {'atom-steps/s':>13} {'pairs/s':>10}")
SIZES = ((9, 32), (14, 32), (22, 64), (35, 128), (44, 128), (56, 256), (70, 256))
if len(sys.argv) > 1:  # e.g. 22:64 56:256
    SIZES = tuple(tuple(int(v) for v in a.split(":")) for a in sys.argv[1:])
for n_side, n_mesh in SIZES:
    t0 = time.time()
    w = workloads.water_box(n_side=n_side, n_mesh=n_mesh)
    f = bench.Frame(w, dev)
    g = tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos, f.pairs, f.shifts)
    for _ in range(5):
        g()
    torch.cuda.synchronize()
    steps = 50 if w.n_atoms < 300000 else 20
    t1 = time.perf_counter()
    for _ in range(steps):
        g()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / steps * 1e3
    print(f"{w.n_atoms:9d} {w.n_pairs:11d} {n_mesh:4d}^3 {ms:9.4f} {w.n_atoms / ms * 1e3:13.4e} {w.n_pairs / ms * 1e3:10.3e}"
          f"   (setup {time.time() - t0:.1f} s)", flush=True)
    del g, f, w
    torch.cuda.empty_cache()

"""How does the fused pair kernel's time scale with the number of rows it is given?  (throughput- or latency-bound)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import _lib, ops, workloads  # noqa: E402

w = workloads.water_box()
dev = torch.device("cuda", 0)
f32 = torch.float32
pos = torch.tensor(w.positions, dtype=f32, device=dev)
cell = torch.tensor(w.cell, dtype=f32, device=dev)
q = torch.tensor(w.charges, dtype=f32, device=dev)
pairs = torch.tensor(w.pairs, device=dev)
S = torch.tensor(w.shifts, dtype=f32, device=dev)
N = pos.shape[0]
topo = ops.get_topology(pairs, N)
ent_sh, fmt = topo.entries_with_shifts(S)
pot = tpa.CoulombPotential(smearing=w.smearing)._descriptor()
lib = _lib.load()
st = _lib.current_stream(dev)
out = torch.zeros((N, 1), dtype=f32, device=dev)
force = torch.zeros((N, 3), dtype=f32, device=dev)
rec = torch.empty((N, 4), dtype=f32, device=dev)
F32 = _lib.F32


def timed(fn, reps=100):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000


def fused(n, ready):
    _lib.check(lib.mipme_sr_rows_fused(
        _lib.current_stream(dev), F32, n, topo.row_ptr.data_ptr(), ent_sh.data_ptr(), topo.entries.data_ptr(), None, pos.data_ptr(),
        cell.data_ptr(), q.data_ptr(), q.data_ptr(), None, 0, 0, C.byref(pot), 0, fmt, rec.data_ptr(), ready,
        out.data_ptr(), force.data_ptr(), None, None, None))


fused(N, 0)  # fills the records
g = torch.cuda.CUDAGraph()
for frac in (1 / 64, 1 / 16, 1 / 8, 1 / 4, 1 / 2, 3 / 4, 1.0):
    n = max(64, int(N * frac))
    # graph replay of 20 back-to-back launches: no host launch overhead in the figure
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fused(n, 1)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for _ in range(20):
                fused(n, 1)
    t = timed(gr.replay, reps=20) / 20
    print(f"rows {n:6d} ({frac:6.3f} of N): {t:6.2f} us per launch  ({t / (n / N):6.2f} us per N rows)")

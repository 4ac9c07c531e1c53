"""GPU timing probe for the potential + force pass of the fused pair kernel on the cfg3 water box, one launch at a time
(graph of 20 back-to-back launches): 8-byte entries (format 1), 4-byte entries with the generic body (MIPME_ROWS_PK=0) or the
packed fp32 body (default).  Prints the time per launch and the deviation of potentials / force sums from the format-1 result."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import _lib, ops, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "water"
w = {"water": workloads.water_box, "dispersion": workloads.dispersion_box}[name]()
dev = torch.device("cuda", 0)
f32 = torch.float32
pos = torch.tensor(w.positions, dtype=f32, device=dev)
cell = torch.tensor(w.cell, dtype=f32, device=dev)
q = torch.tensor(w.charges, dtype=f32, device=dev)
pairs = torch.tensor(w.pairs, device=dev)
S = torch.tensor(w.shifts, dtype=f32, device=dev)
N = pos.shape[0]
topo = ops.get_topology(pairs, N)
ent8, fmt8 = topo.entries_with_shifts(S)
ent4 = topo.compact_entries(S)
potential = (tpa.CoulombPotential(smearing=w.smearing) if w.exponent == 1
             else tpa.InversePowerLawPotential(exponent=w.exponent, smearing=w.smearing))
pot = potential._descriptor()
lib = _lib.load()
rec = torch.empty((N, 4), dtype=f32, device=dev)
F32 = _lib.F32


def run(ent, fmt, out, force, ready=1):
    _lib.check(lib.mipme_sr_rows_fused(
        _lib.current_stream(dev), F32, N, topo.row_ptr.data_ptr(), ent.data_ptr(), topo.entries.data_ptr(), None, pos.data_ptr(),
        cell.data_ptr(), q.data_ptr(), q.data_ptr(), None, 0, 0, C.byref(pot), 0, fmt, rec.data_ptr(), ready,
        out.data_ptr(), force.data_ptr(), None, None, None))


def graph_time(ent, fmt, out, force):
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(ent, fmt, out, force)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for _ in range(20):
                run(ent, fmt, out, force)
    for _ in range(3):
        gr.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 400 * 1000


out8, f8 = torch.zeros((N, 1), dtype=f32, device=dev), torch.zeros((N, 3), dtype=f32, device=dev)
out4, f4 = torch.zeros((N, 1), dtype=f32, device=dev), torch.zeros((N, 3), dtype=f32, device=dev)
run(ent8, fmt8, out8, f8, ready=0)  # fills the records
t8 = graph_time(ent8, fmt8, out8, f8)
t4 = graph_time(ent4, 2, out4, f4)
torch.cuda.synchronize()
print(f"{name} N={N} pairs={pairs.shape[0]} MIPME_ROWS_PK={os.environ.get('MIPME_ROWS_PK', '1')}")
print(f"  8-byte entries, generic body: {t8:7.2f} us")
print(f"  4-byte entries              : {t4:7.2f} us")
print(f"  potential rel dev {float((out4 - out8).abs().max() / out8.abs().max()):.2e}   "
      f"force rel dev {float((f4 - f8).abs().max() / f8.abs().max()):.2e}")

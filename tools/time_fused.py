"""GPU timing probe for the fused distance + pair-sum row kernel (mipme_sr_rows_fused) on the cfg3 water box:
potential only / potential + speculative force sums / force sums only, next to the unfused mipme_rspace_rows."""
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
dist = tpa.pair_distances(pos, pairs, cell, S)
pot = tpa.CoulombPotential(smearing=w.smearing)._descriptor()
lib = _lib.load()
st = _lib.current_stream(dev)
out = torch.zeros((N, 1), dtype=f32, device=dev)
force = torch.zeros((N, 3), dtype=f32, device=dev)
g = torch.randn((N, 1), dtype=f32, device=dev)
rec = torch.empty((N, 4), dtype=f32, device=dev)
F32 = _lib.F32


def timed(fn, reps=50):
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


def fused(want_pot, want_force, grad):
    _lib.check(lib.mipme_sr_rows_fused(
        st, F32, N, topo.row_ptr.data_ptr(), ent_sh.data_ptr(), topo.entries.data_ptr(), None, pos.data_ptr(),
        cell.data_ptr(), q.data_ptr(), q.data_ptr() if want_pot else None, g.data_ptr() if grad else None, 0, 0,
        C.byref(pot), 0, fmt, rec.data_ptr(), 0, out.data_ptr() if want_pot else None, force.data_ptr() if want_force else None, None, None, None))


def unfused():
    _lib.check(lib.mipme_rspace_rows(st, F32, N, 1, topo.row_ptr.data_ptr(), topo.entries.data_ptr(), dist.data_ptr(),
                                     q.data_ptr(), None, 0, 0, C.byref(pot), 0, out.data_ptr()))


print("lib", os.environ.get("MIPME_LIB", "default"))
print("  unfused rspace_rows          %6.1f us" % timed(unfused))
print("  fused potential only         %6.1f us" % timed(lambda: fused(True, False, False)))
print("  fused potential + force sums %6.1f us" % timed(lambda: fused(True, True, False)))
print("  fused force sums (energy)    %6.1f us" % timed(lambda: fused(False, True, False)))
print("  fused force sums (general g) %6.1f us" % timed(lambda: fused(False, True, True)))

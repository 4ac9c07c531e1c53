"""Round 5: timeline of the co-scheduled PLANE spread + pair-sum launch (variant library built with
``bash tools/build_variant.sh timeline -DMIPME_WG_TIMELINE``; run with MIPME_LIB=<that .so>): when the plane and row workgroups
start and end, and where a plane workgroup spends its life (entry, tile zeroed, scatter done, conversion done, transform stored)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import torchpme_amd as tpa  # noqa: E402,F401
from torchpme_amd import _lib  # noqa: E402
from bench import Frame, make_workload  # noqa: E402

w = make_workload(sys.argv[1] if len(sys.argv) > 1 else "water", 0)
f = Frame(w, torch.device("cuda:0"))
for _ in range(5):
    f.step()
torch.cuda.synchronize()
lib = _lib.load()
pad8 = lambda n: (n + 7) // 8 * 8  # noqa: E731
P = int(os.environ.get('MIPME_PLANE_PARTS', '2'))
n_planes = w.n_mesh * P  # plane workgroups
n_rows = (w.n_atoms + 31) // 32
n = pad8(n_planes) + pad8(n_rows)
buf = np.zeros(n * 4, dtype=np.int64)
lib.mipme_debug_wg_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.mipme_debug_wg_timeline(buf.ctypes.data, n * 4) == 0
t = buf.reshape(n, 4)
start, end = t[:, 0] * 0.01, t[:, 1] * 0.01  # us (100 MHz)
ok = t[:, 1] > 0
t0 = start[ok].min()
start, end = start - t0, end - t0
role = np.arange(n) >= pad8(n_planes)
# (the stamps of the eager warm-up and the last replay share the buffer: keep the last launch only)
last = start >= start[ok].max() - 60.0
ok &= last
t0 = start[ok].min()
start, end = start - t0, end - t0
print(f"{n_planes} plane workgroups ({P} per plane), {n_rows} row workgroups; launch lasts {end[ok].max():.2f} us")
for name, m in (("planes", ~role & ok), ("rows", role & ok)):
    d = end[m] - start[m]
    print(f"{name:7s} start {start[m].min():6.2f} .. {start[m].max():6.2f}   end {end[m].min():6.2f} .. {end[m].max():6.2f}   "
          f"lifetime mean {d.mean():5.2f} min {d.min():5.2f} max {d.max():5.2f} us")
print("resident workgroups over time (planes / rows):")
for x in np.arange(0.0, end[ok].max() + 1.0, 1.0):
    live = (start <= x) & (end > x) & ok
    print(f"  t = {x:5.1f} us   {int((live & ~role).sum()):4d} {int((live & role).sum()):5d}   "
          f"rows started so far {int((role & ok & (start <= x)).sum()):5d}  finished {int((role & ok & (end <= x)).sum()):5d}")
pb = np.zeros(1024 * 8, dtype=np.int64)
lib.mipme_debug_wg_phase.argtypes = [C.c_void_p, C.c_int]
assert lib.mipme_debug_wg_phase(pb.ctypes.data, 1024 * 8) == 0
p = pb.reshape(1024, 8)[:min(n_planes, 1024), :5].astype(np.float64) * 0.01
names = ["entry", "tile zeroed + twiddles", "scatter done", "converted to the transform tile", "transform stored"]
for k in range(5):
    rel = p[:, k] - p[:, 0]
    print(f"  {names[k]:32s} mean {rel.mean():6.2f}   min {rel.min():6.2f}   max {rel.max():6.2f} us after the workgroup's entry")

"""Timeline of the workgroups of the co-scheduled spread + pair-sum launch (variant library built with
``bash tools/build_variant.sh timeline -DMIPME_WG_TIMELINE``; run with MIPME_LIB=<that .so>): when brick and row workgroups
start and end, how long they live, how many are resident, per XCD / CU."""
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
n_bricks = (w.n_mesh // 8) ** 3
n_rows = (w.n_atoms + 31) // 32
n = n_bricks + n_rows
buf = np.zeros(n * 4, dtype=np.int64)
lib.mipme_debug_wg_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.mipme_debug_wg_timeline(buf.ctypes.data, n * 4) == 0
t = buf.reshape(n, 4)
start, end = t[:, 0] * 0.01, t[:, 1] * 0.01  # us (100 MHz)
t0 = start.min()
start, end = start - t0, end - t0
hw, xcc = t[:, 2], t[:, 3] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)  # cu_id, sh_id, se_id
role = np.arange(n) >= n_bricks
print(f"{n_bricks} brick workgroups, {n_rows} row workgroups; launch lasts {end.max():.2f} us")
for name, m in (("bricks", ~role), ("rows", role)):
    d = end[m] - start[m]
    print(f"{name:7s} start {start[m].min():6.2f} .. {start[m].max():6.2f}   end {end[m].min():6.2f} .. {end[m].max():6.2f}   "
          f"lifetime mean {d.mean():5.2f} min {d.min():5.2f} max {d.max():5.2f} us")
print("resident workgroups over time (bricks / rows):")
for x in np.arange(0.0, end.max() + 1.0, 1.0):
    live = (start <= x) & (end > x)
    print(f"  t = {x:5.1f} us   {int((live & ~role).sum()):4d} {int((live & role).sum()):5d}   "
          f"rows started so far {int((role & (start <= x)).sum()):5d}  finished {int((role & (end <= x)).sum()):5d}")
key = xcc * 1024 + cu
print("distinct (xcc, se, sh, cu):", len(np.unique(key)), " row workgroups per CU: min/mean/max",
      np.bincount(np.unique(key[role], return_inverse=True)[1]).min(),
      round(float(np.bincount(np.unique(key[role], return_inverse=True)[1]).mean()), 2),
      np.bincount(np.unique(key[role], return_inverse=True)[1]).max())
d = end - start
order = np.argsort(start[role])
dr = d[role][order]
print("row lifetime by start order (deciles):", np.round([dr[int(q * (len(dr) - 1))] for q in np.linspace(0, 1, 11)], 2))

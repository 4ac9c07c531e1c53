"""Where a brick workgroup of the co-scheduled launch spends its life (variant library: ``bash tools/build_variant.sh timeline
-DMIPME_WG_TIMELINE``, run with MIPME_LIB=<that .so>): clock stamps of thread 0 of the first 512 workgroups (the bricks of the
cfg3 launch) at entry, after the neighbour counts are in, after the candidate scan (A1), after the staging (A2), after the
accumulation (C), at the end (R: partial bricks summed and written).
    MIPME_LIB=$PWD/torch-pme_amd/libmipme_timeline.so python tools/brick_phases.py [water|ionic]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402,F401
from torchpme_amd import _lib  # noqa: E402
from bench import Frame, make_workload  # noqa: E402

w = make_workload(sys.argv[1] if len(sys.argv) > 1 else "water", 0)
f = Frame(w, torch.device("cuda:0"))
for _ in range(5):
    f.step()
torch.cuda.synchronize()
lib = _lib.load()
n = min(1024, (w.n_mesh // 8) ** 3)
buf = np.zeros(1024 * 8, dtype=np.int64)
lib.mipme_debug_wg_phase.argtypes = [C.c_void_p, C.c_int]
assert lib.mipme_debug_wg_phase(buf.ctypes.data, 1024 * 8) == 0
t = buf.reshape(1024, 8)[:n, :6].astype(np.float64) * 0.01  # us (100 MHz)
names = ["entry", "neighbour counts in", "A1 candidate scan done", "A2 staging done", "C accumulation done", "R end"]
t0 = t[:, 0].min()
print(f"{n} brick workgroups; first entry .. last end: {t[:, 5].max() - t0:.2f} us")
for k in range(6):
    rel = t[:, k] - t[:, 0]
    print(f"  {names[k]:26s} mean {rel.mean():6.2f}   min {rel.min():6.2f}   max {rel.max():6.2f} us after the workgroup's entry")
d = np.diff(t, axis=1)
print("phase durations (mean / max, us):", {nm: (round(float(c.mean()), 2), round(float(c.max()), 2)) for nm, c in zip(["counts", "A1", "A2", "C", "R"], d.T)})

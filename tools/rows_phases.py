"""Where a row workgroup of the co-scheduled launch spends its life (variant library built with
``bash tools/build_variant.sh timeline -DMIPME_WG_TIMELINE``; run with MIPME_LIB=<that .so>): clock stamps of thread 0 of the
first 1024 row workgroups of the fp64 pair body at entry, after the prologue's loads are issued, after the table barrier (all
loads waited for), after the first and second loop iteration, after the loop, at the end.
    MIPME_LIB=$PWD/torch-pme_amd/libmipme_timeline.so python tools/rows_phases.py [ionic]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import torchpme_amd as tpa  # noqa: E402,F401
from torchpme_amd import _lib  # noqa: E402
from bench import Frame, make_workload  # noqa: E402

w = make_workload(sys.argv[1] if len(sys.argv) > 1 else "ionic", 0)
f = Frame(w, torch.device("cuda:0"))
for _ in range(5):
    f.step()
torch.cuda.synchronize()
lib = _lib.load()
n = min(1024, (w.n_atoms + 31) // 32)
buf = np.zeros(1024 * 8, dtype=np.int64)
lib.mipme_debug_rows_phase.argtypes = [C.c_void_p, C.c_int]
assert lib.mipme_debug_rows_phase(buf.ctypes.data, 1024 * 8) == 0
t = buf.reshape(1024, 8)[:n, :7].astype(np.float64) * 0.01  # us (100 MHz)
names = ["entry", "prologue loads issued", "tables built, loads back", "iteration 1 done", "iteration 2 done", "loop done", "end"]
t0 = t[:, 0].min()
print(f"{n} row workgroups; first entry .. last end: {t[:, 6].max() - t0:.2f} us")
for k in range(7):
    rel = t[:, k] - t[:, 0]
    print(f"  {names[k]:26s} mean {rel.mean():6.2f}   min {rel.min():6.2f}   max {rel.max():6.2f} us after the workgroup's entry")
print("entry of the workgroups after the first one: mean %.2f max %.2f us" % ((t[:, 0] - t0).mean(), (t[:, 0] - t0).max()))

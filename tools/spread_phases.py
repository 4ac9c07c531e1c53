"""Per-phase timing of the spread kernel from clock stamps written by every workgroup (variant library: see
tools/build_spread_timing.py).  Prints, for the cfg3 water box, when each phase of a workgroup ends relative to the first
workgroup's start and the mean / max duration of every phase."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import torchpme_amd as tpa
from torchpme_amd import _lib
from bench import Frame, make_workload
w = make_workload("water", 0)
f = Frame(w, torch.device("cuda:0"))
for _ in range(5):
    f.step()
torch.cuda.synchronize()
lib = _lib.load()
nb = 512
buf = np.zeros(nb * 8, dtype=np.int64)
lib.mipme_debug_spread_times.argtypes = [C.c_void_p, C.c_int]
rc = lib.mipme_debug_spread_times(buf.ctypes.data, nb * 8)
t = buf.reshape(nb, 8).astype(np.float64) * 0.01  # us (100 MHz)
t0 = t[:, 0].min()
names = ["start", "prefix done", "A1 done", "A2 done", "C+sync done", "end", "C done (wave0)"]
for k in range(7):
    rel = t[:, k] - t0
    print(f"{names[k]:16s} mean {rel.mean():7.2f}  min {rel.min():7.2f}  max {rel.max():7.2f} us after the first block start")
d = np.diff(t[:, [0, 1, 2, 3, 6, 4, 5]], axis=1)
print("per-block phase durations (mean / max, us):")
for n, col in zip(["prefix", "A1", "A2", "C (wave 0)", "sync", "R"], d.T):
    print(f"  {n:12s} {col.mean():6.2f} {col.max():6.2f}")
print("block start spread:", (t[:, 0] - t0).max(), " total:", t[:, 5].max() - t0)

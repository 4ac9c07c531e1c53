"""Which (libmipme plan, torch.fft shape) combinations interfere, in both orders (see tools/fft_interference_probe.py).
Every combination runs in its own process: python tools/fft_interference_matrix.py"""
import itertools
import os
import subprocess
import sys

CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["MIPME_FFT_SELFTEST"] = "0"
from torchpme_amd import _lib
ours, theirs, dt, order = sys.argv[1:5]
ns = tuple(int(v) for v in ours.split("x")); ts = tuple(int(v) for v in theirs.split("x"))
dtype = torch.float64 if dt == "f64" else torch.float32
dev = torch.device("cuda:0")
lib = _lib.load()
def ours_err():
    plan = _lib.FFTPlan(dev, dtype, ns, 1)
    a = torch.randn((1,) + ns, dtype=dtype, device=dev)
    G = torch.ones((ns[0] * ns[1] * (ns[2] // 2 + 1),), dtype=dtype, device=dev)
    cd = torch.complex128 if dt == "f64" else torch.complex64
    hat = torch.empty((1, G.numel()), dtype=cd, device=dev); hat2 = torch.empty_like(hat)
    out = torch.empty_like(a); dc = torch.empty((1,), dtype=dtype, device=dev)
    _lib.check(lib.mipme_convolve(plan.handle, _lib.current_stream(dev), a.data_ptr(), G.data_ptr(), hat.data_ptr(),
                                  hat2.data_ptr(), out.data_ptr(), dc.data_ptr()))
    torch.cuda.synchronize()
    return float((out / (ns[0] * ns[1] * ns[2]) - a).abs().max())
def theirs_err():
    x = torch.randn((1,) + ts, dtype=dtype, device=dev)
    y = torch.fft.rfftn(x, dim=(1, 2, 3)); torch.cuda.synchronize()
    ref = torch.fft.rfftn(x.cpu(), dim=(1, 2, 3))
    return float((y.cpu() - ref).abs().max() / ref.abs().max())
if order == "torch-first":
    t = theirs_err(); o = ours_err()
else:
    o = ours_err(); t = theirs_err()
print(f"{ours:>12s} {theirs:>12s} {dt} {order:>12s}  libmipme identity err {o:9.2e}   torch.fft.rfftn rel err {t:9.2e}")
'''
shapes = ["32x32x32", "64x64x64", "32x32x128", "128x128x128"]
for dt in ("f32", "f64"):
    for ours, theirs in itertools.product(shapes, shapes):
        for order in ("torch-first", "ours-first"):
            r = subprocess.run([sys.executable, "-c", CHILD, ours, theirs, dt, order], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if "libmipme identity" in l]
            bad = False
            if line:
                eo = float(line[0].split("identity err")[1].split()[0])
                et = float(line[0].split("rel err")[1].split()[0])
                bad = eo > 1e-3 or et > 1e-3
            if line:
                print(line[0] + ("   <-- interference" if bad else ""))
            else:
                print(ours, theirs, dt, order, "FAILED", r.stderr.strip().splitlines()[-1][:120] if r.stderr.strip() else "")

"""hipFFT interference probe: results of a libmipme FFT plan created after torch.fft has run.

    python tools/fft_interference_probe.py none        # plan created first: R2C, convolution and identity all exact
    python tools/fft_interference_probe.py 64x64x64    # after torch.fft.rfftn of a 64^3 fp64 tensor

With PyTorch 2.10+rocm7.0 the second call makes plan creation fail in its self-test (irfftn(rfftn(x)) != x); with
MIPME_FFT_SELFTEST=0 the plan is used anyway and the C2R transform of the (32,32,128) fp64 mesh is wrong at every odd z
index (the R2C transform and the filter table are right).  See DESIGN.md, ROCm pitfalls."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
from torchpme_amd import _lib, ops
DEV = torch.device("cuda:0")
pre = sys.argv[1]
if pre != "none":
    shape = tuple(int(v) for v in pre.split("x"))
    x = torch.randn((1,) + shape, device=DEV, dtype=torch.float64)
    y = torch.fft.rfftn(x, dim=(1, 2, 3)); torch.cuda.synchronize()
    del x, y
ns = (32, 32, 128)
rng = np.random.default_rng(0)
a = rng.normal(size=(1,) + ns)
ta = torch.tensor(a, device=DEV)
plan = _lib.FFTPlan(DEV, torch.float64, ns, 1)
geom = ops.MeshGeometry(np.diag([8.0, 7.0, 30.0]), ns, _lib.P3M, 5)
md = geom.desc(1)
hat = torch.empty((1, 32 * 32 * 65), dtype=torch.complex128, device=DEV)
lib = _lib.load()
_lib.check(lib.mipme_fft_r2c(plan.handle, _lib.current_stream(DEV), _lib.F64, C.byref(md), ta.data_ptr(), hat.data_ptr()))
torch.cuda.synchronize()
ref = np.fft.rfftn(a, axes=(1, 2, 3)).reshape(1, -1)
print(pre, "3-D R2C rel err", float(np.abs(hat.cpu().numpy() - ref).max() / np.abs(ref).max()))
# filter table
pot = tpa.CoulombPotential(smearing=1.0)._descriptor()
G = ops.build_filter(geom, pot, torch.float64, DEV)
torch.cuda.synchronize()
print(pre, "G checksum", float(G.double().sum()), float(G.double().abs().max()))
# full convolution with the 3-D plans against numpy
hat2 = torch.empty_like(hat)
out = torch.empty_like(ta)
dc = torch.empty((1,), dtype=torch.float64, device=DEV)
_lib.check(lib.mipme_convolve(plan.handle, _lib.current_stream(DEV), ta.data_ptr(), G.data_ptr(), hat.data_ptr(), hat2.data_ptr(),
                              out.data_ptr(), dc.data_ptr()))
torch.cuda.synchronize()
Gn = G.cpu().numpy().reshape(32, 32, 65)
refc = np.fft.irfftn(np.fft.rfftn(a[0]) * Gn, s=ns) * (32 * 32 * 128)
print(pre, "convolve rel err", float(np.abs(out.cpu().numpy()[0] - refc).max() / np.abs(refc).max()))
# the spread of a few atoms (brick path via kspace_forward is harder to isolate): mipme_spread
o = out.cpu().numpy()[0]
print("out/ref at a few points", (o / refc)[0, 0, :6], (o / refc)[5, 7, 60:66])
# identity test: irfftn(rfftn(x)) with G = 1
ones = torch.ones_like(G)
_lib.check(lib.mipme_convolve(plan.handle, _lib.current_stream(DEV), ta.data_ptr(), ones.data_ptr(), hat.data_ptr(), hat2.data_ptr(),
                              out.data_ptr(), dc.data_ptr()))
torch.cuda.synchronize()
idt = out.cpu().numpy()[0] / (32 * 32 * 128)
print(pre, "identity rel err", float(np.abs(idt - a[0]).max() / np.abs(a[0]).max()))
err = np.abs(idt - a[0])
print("bad points per z index (first 12):", (err > 1e-9).sum(axis=(0, 1))[:12], "total bad", int((err > 1e-9).sum()), "of", err.size)

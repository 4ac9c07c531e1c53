"""mipme_scaled_match (is g == s * q ?) against the number of values: one workgroup up to 32 768 values in registers, a loop beyond.
    python tools/time_match.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchpme_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
st = _lib.current_stream(dev)
for n in (8000, 31944, 65536, 262144, 1029000):
    q = torch.randn(n, device=dev)
    g = 0.37 * q
    res = torch.empty(2, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.empty(max(1, lib.mipme_scaled_match_work(n)), dtype=torch.float64, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        lib.mipme_scaled_match_wide(st, 0, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag.data_ptr(), work.data_ptr())
    a.record()
    for _ in range(50):
        lib.mipme_scaled_match_wide(st, 0, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag.data_ptr(), work.data_ptr())
    b.record()
    torch.cuda.synchronize()
    t_wide = a.elapsed_time(b) / 50 * 1e3
    for _ in range(5):
        lib.mipme_scaled_match(st, 0, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag.data_ptr())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        lib.mipme_scaled_match(st, 0, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag.data_ptr())
    b.record()
    torch.cuda.synchronize()
    g2 = g.clone()
    g2[n // 2] *= 1.001
    lib.mipme_scaled_match(st, 0, n, g2.data_ptr(), q.data_ptr(), res.data_ptr(), flag.data_ptr())
    torch.cuda.synchronize()
    bad = int(flag[0])
    lib.mipme_scaled_match(st, 0, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag.data_ptr())
    torch.cuda.synchronize()
    print(f"n = {n:8d}: one workgroup {a.elapsed_time(b) / 50 * 1e3:7.1f} us, many blocks {t_wide:6.1f} us per call (back to back), scale {float(res[0]):.6f}, match {int(flag[0])}, perturbed {bad}")

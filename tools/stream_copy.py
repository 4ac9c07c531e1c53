"""Achievable HBM copy bandwidth on this GPU (report it next to the nominal 8 TB/s):
device-to-device copies and a read-only reduction over buffers far larger than the 256 MiB Infinity Cache."""
import torch

dev = torch.device("cuda", 0)
print(torch.cuda.get_device_name(0))
for mib in (64, 512, 2048, 8192):
    n = mib * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for name, fn, nbytes in (("copy  (read + write)", lambda: b.copy_(a), 2 * 4 * n), ("sum   (read only)   ", lambda: a.sum(), 4 * n)):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        reps = 20
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        print(f"{mib:6d} MiB  {name}  {ms * 1e3:9.1f} us  {nbytes / ms / 1e9:8.3f} TB/s")
    del a, b

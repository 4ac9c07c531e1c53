"""The reference call sequence (eager drop-in step, compiled front end) with the energy-mode verdict polled on the host (0), the
default mix (1) or acted on by the device (2): ms per step, interleaved, same process -- the 6 launches the device-side decision
skips at run time cost ~26 us of GPU time and ~30 us of host time per step (profiles/r06_fin_kernel_stats_dropin.txt).
    python tools/r06/select_mode_ab.py [n_steps] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from torchpme_amd import _front  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
frame = bench.Frame(bench.make_workload("water", 0), torch.device("cuda"))
mod = _front.module()


def ms(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


res = {0: [], 1: [], 2: []}
for _ in range(reps):
    for mode in (0, 1, 2):
        mod.set_device_select(mode)
        res[mode].append(ms(lambda: frame.step_reference_protocol("helper")))
mod.set_device_select(_front.select_mode())
for mode, name in ((0, "polled on the host"), (1, "default mix"), (2, "on the device")):
    v = sorted(res[mode])
    print(f"select {mode} ({name:18s}): median {v[len(v) // 2]:.4f}  min {v[0]:.4f}  max {v[-1]:.4f} ms/step   {[round(x, 4) for x in res[mode]]}")

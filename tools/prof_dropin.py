"""cProfile of the reference call sequence (eager drop-in step) on the cfg3 frame: where the host time goes.
    python tools/prof_dropin.py [n_steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
frame = bench.Frame(bench.make_workload("water", 0), torch.device("cuda"))
for mode in ("helper",):
    for _ in range(20):
        frame.step_reference_protocol(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        frame.step_reference_protocol(mode)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{mode}: host-side {1e3 * t_host / n:.4f} ms/step, with final sync {1e3 * t_all / n:.4f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        frame.step_reference_protocol(mode)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
for _ in range(20):
    frame.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    frame.step()
torch.cuda.synchronize()
print(f"fast eager step: {1e3 * (time.perf_counter() - t0) / n:.4f} ms/step")

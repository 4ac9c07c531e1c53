"""cProfile of the reference call sequence (eager drop-in step) on the cfg3 frame: where the host time goes -- main thread
and the autograd engine's device thread (profiled from inside the first backward call).
    python tools/prof_dropin.py [n_steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torchpme_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
frame = bench.Frame(bench.make_workload("water", 0), torch.device("cuda"))


def timeit(fn, label):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{label}: host-side {1e3 * t_host / n:.4f} ms/step, with final sync {1e3 * t_all / n:.4f} ms/step")


timeit(lambda: frame.step_reference_protocol("helper"), "drop-in")
timeit(frame.step, "fast eager")

# ---- profile: main thread
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    frame.step_reference_protocol("helper")
torch.cuda.synchronize()
pr.disable()
print("==== main thread")
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
print("==== main thread, by own time")
pstats.Stats(pr).sort_stats("tottime").print_stats(35)

# ---- profile: engine thread (enable a profiler from inside the backward)
state = {}
orig = ops._PMEFunction.backward


def wrapped(ctx, *grads):
    if "pr" not in state:
        state["pr"] = cProfile.Profile()
        state["pr"].enable()
    return orig(ctx, *grads)


ops._PMEFunction.backward = staticmethod(wrapped)
for _ in range(n):
    frame.step_reference_protocol("helper")
torch.cuda.synchronize()


def stop(ctx, *grads):
    state["pr"].disable()
    state["done"] = True
    return orig(ctx, *grads)


ops._PMEFunction.backward = staticmethod(stop)
frame.step_reference_protocol("helper")
torch.cuda.synchronize()
print("==== autograd engine thread")
pstats.Stats(state["pr"]).sort_stats("cumulative").print_stats(35)

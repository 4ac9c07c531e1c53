"""Many small frames: GraphedFrameBatch (one launch per kernel for all frames) against one HIP graph per frame on its own
stream.  usage: python tools/frames_probe.py [n_frames] [n_side] [dtype]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
dev = torch.device("cuda", 0)
ws = [workloads.ionic_box(n_side=n_side, n_mesh=32, cutoff=6.0, seed=100 + k) for k in range(F)]
w = ws[0]


def make_calc():
    return tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing,
                             interpolation_nodes=w.order)


calc = make_calc()
t = lambda a: torch.tensor(a, dtype=dtype, device=dev)  # noqa: E731
frames = [(t(x.charges), t(x.cell), t(x.positions), torch.tensor(x.pairs, device=dev), t(x.shifts)) for x in ws]
batch = tpa.GraphedFrameBatch(calc, frames)
# one calculator per frame: frames that run concurrently on different streams must not share a plan (brick counters)
singles = [tpa.GraphedEnergyForces(make_calc(), *f) for f in frames]
streams = [torch.cuda.Stream(dev) for _ in frames]


def run_streams():
    for g, s in zip(singles, streams):
        with torch.cuda.stream(s):
            g.graph.replay()


def join():
    for s in streams:
        torch.cuda.current_stream(dev).wait_stream(s)


def timed(fn, after=lambda: None, reps=300):
    for _ in range(20):
        fn()
    after()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    after()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


a = timed(lambda: batch())
b = timed(run_streams, join)
E = batch()[0].cpu().numpy()
E1 = np.array([g()[0].item() for g in singles])
print(f"{F} frames x {w.n_atoms} atoms ({w.n_pairs} pairs, {w.n_mesh}^3, {dtype}): one launch per kernel {a:.4f} ms/step "
      f"= {F * w.n_atoms / a * 1e3:.3e} atom-steps/s; graphs on streams {b:.4f} ms/step = {F * w.n_atoms / b * 1e3:.3e}; "
      f"max rel energy difference {np.abs(E / E1 - 1).max():.1e}")

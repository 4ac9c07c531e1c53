"""Probe: do HIP-graph replays of independent frames overlap when issued on different streams?  (8 x cfg2 frames.)
Throughput only -- the frames share plan counters / reduction scratch here, so the numbers are not checked."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
import bench
from torchpme_amd import workloads
dev = torch.device("cuda", 0)
F = 8
frames = [bench.Frame(workloads.ionic_box(seed=12 + f), dev) for f in range(F)]
graphs = [tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos, f.pairs, f.shifts) for f in frames]
streams = [torch.cuda.Stream(dev) for _ in range(F)]
def run(multi, steps=50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for k, g in enumerate(graphs):
            if multi:
                with torch.cuda.stream(streams[k]):
                    g.graph.replay()
            else:
                g.graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for _ in range(2):
    print("single stream: %.3f ms per %d frames" % (run(False), F))
    print("multi  stream: %.3f ms per %d frames" % (run(True), F))

"""GPU timing probe: pair_distances backward with / without shift data (upper bound of what packing the shift into
the entry word could save)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
from torchpme_amd import workloads
w = workloads.water_box()
dev = torch.device("cuda", 0)
pos = torch.tensor(w.positions, dtype=torch.float32, device=dev, requires_grad=True)
cell = torch.tensor(w.cell, dtype=torch.float32, device=dev)
pairs = torch.tensor(w.pairs, device=dev)
S = torch.tensor(w.shifts, dtype=torch.float32, device=dev)
def run(use_shifts):
    d = tpa.pair_distances(pos, pairs, cell, S) if use_shifts else tpa.pair_distances(pos, pairs)
    g = torch.ones_like(d)
    for _ in range(5):
        pos.grad = None; d.backward(g, retain_graph=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(50):
        pos.grad = None; d.backward(g, retain_graph=True)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 50 * 1000
print("with shifts   : %.1f us per backward (incl. torch overhead)" % run(True))
print("without shifts: %.1f us" % run(False))

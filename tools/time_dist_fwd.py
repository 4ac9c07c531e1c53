"""GPU timing probe: pair_distances forward (packed 16-byte pair stream) on the cfg3 and cfg5 boxes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa
from torchpme_amd import workloads
dev = torch.device("cuda", 0)
print("lib", os.environ.get("MIPME_LIB", "default"))
for w in (workloads.water_box(), workloads.dispersion_box()):
    pos = torch.tensor(w.positions, dtype=torch.float32, device=dev)
    cell = torch.tensor(w.cell, dtype=torch.float32, device=dev)
    pairs = torch.tensor(w.pairs, device=dev)
    S = torch.tensor(w.shifts, dtype=torch.float32, device=dev)
    for _ in range(5):
        tpa.pair_distances(pos, pairs, cell, S)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(50):
        tpa.pair_distances(pos, pairs, cell, S)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1000
    print(f"  {w.name}: {us:8.1f} us  {16 * w.n_pairs / us / 1e6:6.2f} TB/s (16 B/pair)")

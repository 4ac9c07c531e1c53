"""The reference call sequence with a NEW neighbour-list tensor every call (bench.py: drop_in.cold_list_ms) on the cfg3 box:
wall time per call and (PROFILE=1) a cProfile of the host side.  Run under rocprofv3 for the kernel list."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torchpme_amd import workloads
w = workloads.water_box()
f = bench.Frame(w, torch.device("cuda", 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for _ in range(10): f.step_cold_list("list")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): f.step_cold_list("list")
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"cold list: host-side {1e3 * t_host / n:.4f} ms/call, with final sync {1e3 * (time.perf_counter() - t0) / n:.4f} ms/call")
if os.environ.get("PROFILE") == "1":
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n): f.step_cold_list("list")
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(40)

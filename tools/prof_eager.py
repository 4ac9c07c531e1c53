import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
from torchpme_amd import workloads
w = workloads.water_box()
f = bench.Frame(w, torch.device("cuda", 0))
for _ in range(20): f.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): f.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)

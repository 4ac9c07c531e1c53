"""Probe (GPU box): which combination of {process group, collective, HIP-graph replay} faults.
usage: python tools/dist_probe.py <mode>   with mode in: nopg, pg_only, pg_barrier_after, torchgraph_pg_barrier"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch.distributed as dist
torch.cuda.set_device(0)
if mode != "nopg":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
if mode == "torchgraph_pg_barrier":
    x = torch.ones(1 << 20, device="cuda"); y = torch.zeros_like(x)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y.copy_(x * 2)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y.copy_(x * 2 + 1)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); print(mode, "OK", y[:2].tolist()); sys.exit(0)
import torchpme_amd as tpa
from torchpme_amd import workloads
w = workloads.ionic_box(n_side=12, n_mesh=32, cutoff=6.0, dtype="f32")
dev = torch.device("cuda", 0)
t = lambda a, dt=torch.float32: torch.tensor(a, dtype=dt, device=dev)
if os.environ.get("PROBE_SR") == "1":
    calc = tpa.Calculator(tpa.CoulombPotential())
else:
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
step = tpa.GraphedEnergyForces(calc, t(w.charges), t(w.cell), t(w.positions), torch.tensor(w.pairs, device=dev), t(w.shifts))
for _ in range(5): step()
torch.cuda.synchronize(); print(mode, "5 replays ok", flush=True)
if mode == "pg_barrier_after":
    dist.barrier(); torch.cuda.synchronize(); print("barrier ok", flush=True)
for _ in range(200): step()
torch.cuda.synchronize(); print(mode, "OK", float(step.energy), flush=True)

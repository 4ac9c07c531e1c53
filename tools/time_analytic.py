"""Cost of exact second order (``calculator.double_backward = "analytic"``, torch-pme_amd/analytic.py) at benchmark size: one
training step of a loss on forces with learned charges -- E = sum q V, F = -dE/dr with create_graph=True, loss = sum F^2,
loss.backward() to a charge-scaling parameter (and the positions) -- next to the first-order energy + forces evaluation of the
same path and of the fused default path.  Usage: python tools/time_analytic.py [water|ionic|dispersion]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "water"
w = {"water": workloads.water_box, "ionic": workloads.ionic_box, "dispersion": workloads.dispersion_box}[name]()
dev = torch.device("cuda:0")
dt = torch.float32 if w.dtype == "f32" else torch.float64
pos0 = torch.tensor(w.positions, dtype=dt, device=dev)
q0 = torch.tensor(w.charges, dtype=dt, device=dev)
cell = torch.tensor(w.cell, dtype=dt, device=dev)
pairs = torch.tensor(w.pairs, dtype=torch.int64, device=dev)
shifts = torch.tensor(w.shifts, dtype=dt, device=dev)
pot = (tpa.CoulombPotential(smearing=w.smearing) if w.exponent == 1
       else tpa.InversePowerLawPotential(exponent=w.exponent, smearing=w.smearing))
Calc = tpa.P3MCalculator if w.scheme == "P3M" else tpa.PMECalculator


def wall_ms(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


out = {"workload": w.name, "n_atoms": w.n_atoms, "n_pairs": w.n_pairs, "dtype": w.dtype}
for mode in ((None, "analytic") if os.environ.get("MIPME_TIME_ANALYTIC_CHILD") != "graph" else ()):
    calc = Calc(pot, mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
    calc.double_backward = mode
    theta = torch.ones((), dtype=dt, device=dev, requires_grad=True)
    pos = pos0.clone().requires_grad_(True)

    def energy():
        q = q0 * theta
        d = tpa.pair_distances(pos, pairs, cell, shifts)
        return (q * calc(q, cell, pos, pairs, d)).sum()

    def first_order():
        (F,) = torch.autograd.grad(energy(), pos)
        return F

    tag = "fused" if mode is None else mode
    out[f"{tag}_energy_forces_ms"] = round(wall_ms(first_order, 10), 3)
    if mode is not None:
        def train_step():
            theta.grad = None
            (g,) = torch.autograd.grad(energy(), pos, create_graph=True)
            loss = (g * g).sum()
            loss.backward(inputs=[theta])
            return loss

        torch.cuda.reset_peak_memory_stats()
        out[f"{tag}_force_loss_step_ms"] = round(wall_ms(train_step, 5), 3)
        out[f"{tag}_peak_GB"] = round(torch.cuda.max_memory_allocated() / 2**30, 2)
        loss = train_step()
        out["loss"], out["dloss_dtheta"] = float(loss.detach()), float(theta.grad)
        # E is quadratic in theta, F linear, loss quadratic: d loss / d theta = 2 loss / theta at theta = 1 ... times 2
        out["dloss_dtheta_expected"] = 4.0 * float(loss.detach())
# The same training step captured into a HIP graph (torch.cuda.graph): the eager step is ~400 launches and bound by the host.
# In a child process (a failed capture must not take the eager numbers with it).
if os.environ.get("MIPME_TIME_ANALYTIC_CHILD") == "graph":
    calc = Calc(pot, mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
    calc.double_backward = "analytic"
    theta = torch.ones((), dtype=dt, device=dev, requires_grad=True)
    pos = pos0.clone().requires_grad_(True)

    def body():
        q = q0 * theta
        d = tpa.pair_distances(pos, pairs, cell, shifts)
        (g,) = torch.autograd.grad((q * calc(q, cell, pos, pairs, d)).sum(), pos, create_graph=True)
        loss = (g * g).sum()
        return torch.stack([loss.detach(), torch.autograd.grad(loss, theta)[0]])

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ref = body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res = body()
    ms = wall_ms(graph.replay, 20)
    print("GRAPH " + json.dumps({"force_loss_step_graph_ms": round(ms, 3), "loss": float(res[0]), "dloss_dtheta": float(res[1]),
                                 "eager_loss": float(ref[0])}))
    sys.exit(0)
import subprocess  # noqa: E402

child = subprocess.run(["timeout", "150", sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True,
                       env=dict(os.environ, MIPME_TIME_ANALYTIC_CHILD="graph"))
lines = [ln for ln in child.stdout.splitlines() if ln.startswith("GRAPH ")]
out["graph"] = json.loads(lines[-1][6:]) if lines else {"error": f"rc {child.returncode}: " + child.stderr.strip()[-200:]}
print(json.dumps(out))

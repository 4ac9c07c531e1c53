"""Where the autograd contract stands: ms per step for {forces}, {forces, dE/dq}, {forces, dE/dq, dE/dcell} as a captured graph
and eagerly, and the reference's own TuningTimings protocol (tuning/tuner.py:337-373: clones with requires_grad on positions,
cell and charges, fixed distances, ``result.sum().backward()``).  Usage: python tools/time_contract.py [water|ionic|dispersion]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import ops, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "water"
w = {"water": workloads.water_box, "ionic": workloads.ionic_box, "dispersion": workloads.dispersion_box}[name]()
dev = torch.device("cuda:0")
dt = torch.float32 if w.dtype == "f32" else torch.float64
pos0 = torch.tensor(w.positions, dtype=dt, device=dev)
q0 = torch.tensor(w.charges, dtype=dt, device=dev)
cell0 = torch.tensor(w.cell, dtype=dt, device=dev)
pairs = torch.tensor(w.pairs, dtype=torch.int64, device=dev)
shifts = torch.tensor(w.shifts, dtype=dt, device=dev)
pot = (tpa.CoulombPotential(smearing=w.smearing) if w.exponent == 1
       else tpa.InversePowerLawPotential(exponent=w.exponent, smearing=w.smearing))
Calc = tpa.P3MCalculator if w.scheme == "P3M" else tpa.PMECalculator
calc = Calc(pot, mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)


def event_ms(fn, n, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def wall_ms(fn, n, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


out = {"workload": w.name, "n_atoms": w.n_atoms, "n_pairs": w.n_pairs}
for leaves in (("pos",), ("pos", "q"), ("pos", "q", "cell")):
    pos = pos0.clone().requires_grad_(True)
    q = q0.clone().requires_grad_("q" in leaves)
    cell = cell0.clone().requires_grad_("cell" in leaves)

    def step(weighted=True):
        pos.grad = q.grad = cell.grad = None
        d = tpa.pair_distances(pos, pairs, cell, shifts)
        V = calc(q, cell, pos, pairs, d)
        E = tpa.weighted_sum(V, q) if weighted else (q * V).sum()
        E.backward()
        return E

    key = "+".join(leaves)
    out[f"eager_{key}_ms"] = wall_ms(lambda: step(False), 50)
    out[f"eager_weighted_sum_{key}_ms"] = wall_ms(step, 50)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        out[f"graph_{key}_ms"] = event_ms(g.replay, 200, 20)
    except Exception as e:  # noqa: BLE001
        out[f"graph_{key}_ms"] = f"capture failed: {type(e).__name__}: {e}"[:200]
        torch.cuda.synchronize()

# fast graph class for comparison
gef = tpa.GraphedEnergyForces(calc, q0.detach(), cell0, pos0, pairs, shifts)
out["GraphedEnergyForces_ms"] = event_ms(gef.graph.replay, 200, 20)
gefc = tpa.GraphedEnergyForces(calc, q0.detach(), cell0, pos0, pairs, shifts, cell_gradient=True)
out["GraphedEnergyForces_cell_ms"] = event_ms(gefc.graph.replay, 200, 20)

# the reference's TuningTimings protocol, literally (fixed distance tensor: no provenance)
d_fixed = tpa.pair_distances(pos0, pairs, cell0, shifts).detach().clone()


def protocol(seed_sum=True, backward=True):
    positions, cell, charges = pos0.clone(), cell0.clone(), q0.clone()
    if backward:
        for t in (positions, cell, charges):
            t.requires_grad_(True)
    result = calc.forward(positions=positions, charges=charges, cell=cell, neighbor_indices=pairs, neighbor_distances=d_fixed)
    value = result.sum() if seed_sum else (charges * result).sum()
    if backward:
        value.backward(retain_graph=True)
    return value


out["tuning_protocol_sum_ms"] = wall_ms(protocol, 40)
out["tuning_protocol_energy_seed_ms"] = wall_ms(lambda: protocol(False), 40)
out["tuning_protocol_forward_only_ms"] = wall_ms(lambda: protocol(True, False), 40)
tt = tpa.tuning.TuningTimings(q0, cell0, pos0, pairs, d_fixed, n_repeat=20, n_warmup=4)
out["TuningTimings_median_ms"] = 1e3 * tt(calc)
print(json.dumps(out, indent=1))

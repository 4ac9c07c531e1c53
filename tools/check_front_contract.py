"""Compiled front end (csrc/front.cpp) with charges / cell that require a gradient, against the oracle: energy mode, a general
upstream gradient, observed distances; then eager timings on the cfg3 box.
    python tools/check_front_contract.py [--no-timing]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import torchpme_amd as tpa  # noqa: E402
from oracle import pme_numpy as O  # noqa: E402
from torchpme_amd import _front, ops, workloads  # noqa: E402

dev = torch.device("cuda:0")
assert _front.module() is not None


def small_box(seed, triclinic, n_side=7, a=2.3):
    rng = np.random.default_rng(seed)
    L = n_side * a
    g = (np.arange(n_side) + 0.5) * a
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.35, 0.35, (n_side**3, 3))
    cell = L * np.eye(3)
    if triclinic:
        cell = cell + np.array([[0.0, 0.0, 0.0], [0.8, 0.0, 0.0], [-0.5, 0.7, 0.0]])
        pos = (pos / L) @ cell
    q = rng.normal(size=(len(pos), 1))
    q -= q.mean()
    return pos, cell, q


def oracle_grads(spec, scheme, order, h, q, cell, pos, pairs, S, w):
    """gradients of L = sum(w * V) w.r.t. positions, charges, cell (through the distances too), and dL/dd"""
    dist, _ = O.pair_distances(pos, cell, pairs, S)
    V, cache = O.forward(spec, scheme, order, h, q, cell, pos, pairs, dist, return_cache=True)
    gr = O.backward(cache, w)
    gpos, gcell_pair = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    return V, gpos + gr["positions"], gr["charges"], gr["cell"] + gcell_pair, gr["dist"]


def rel(a, b):
    a = a.detach().cpu().double().numpy().reshape(b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 3e-4)):
    for scheme, order, expo, tri in (("P3M", 5, 1, False), ("PME", 4, 1, True), ("P3M", 4, 6, True)):
        pos, cell, q = small_box(11 + order, tri)
        if expo == 6:
            q = np.abs(q) + 0.5
        rc, sm = 5.5, 1.1
        L = np.linalg.norm(cell, axis=1).min()
        hmesh = 2 * L / 30
        pairs, S, _ = tpa.neighbor_list(pos, cell, rc)
        spec = O.PotentialSpec("coulomb" if expo == 1 else "ipl", expo, sm, 1.0)
        pot = tpa.CoulombPotential(smearing=sm) if expo == 1 else tpa.InversePowerLawPotential(exponent=6, smearing=sm)
        Calc = tpa.P3MCalculator if scheme == "P3M" else tpa.PMECalculator
        calc = Calc(pot, mesh_spacing=hmesh, interpolation_nodes=order)
        ti = torch.tensor(pairs, device=dev)
        ts = torch.tensor(S, dtype=dtype, device=dev)
        rng = np.random.default_rng(5)
        w_gen = rng.normal(size=q.shape)
        for mode in ("energy", "general", "energy+observed", "general+observed", "charges-only", "cell-only"):
            w = q if mode.startswith(("energy", "charges", "cell")) else w_gen
            Vo, Fo, dqo, dco, ddo = oracle_grads(spec, "P3M" if scheme == "P3M" else "Lagrange", order, hmesh, q, cell, pos, pairs, S, w)
            tq, tc, tp = (torch.tensor(x, dtype=dtype, device=dev) for x in (q, cell, pos))
            tp.requires_grad_(True)
            tq.requires_grad_(mode != "cell-only")
            tc.requires_grad_(mode != "charges-only")
            d = tpa.pair_distances(tp, ti, tc, ts)
            assert _front.module().is_front_distances(d), "distances did not take the compiled node"
            V = calc(tq, tc, tp, ti, d)
            if "observed" in mode:
                d.retain_grad()  # (after the call: before it, the calculator declines the compiled node -- tested elsewhere)
            served = V.grad_fn.name()
            tw = tq if w is q else torch.tensor(w, dtype=dtype, device=dev)
            # (energy mode: the upstream gradient of V is the charges -- with the charges' own factor detached, so that dL/dq is
            # the calculator's half, which is what the oracle's adjoint returns)
            Lval = (tw.detach() * V).sum()
            Lval.backward()
            errs = dict(V=rel(V, Vo), pos=rel(tp.grad, Fo))
            if tq.grad is not None:
                errs["q"] = rel(tq.grad, dqo)
            if tc.grad is not None:
                errs["cell"] = rel(tc.grad, dco)
            if "observed" in mode:
                errs["dd"] = rel(d.grad, ddo)
            key = f"{str(dtype)[6:]} {scheme}{order} p={expo} tri={int(tri)} {mode:17s} {served}"
            print(key, " ".join(f"{k} {v:.1e}" for k, v in errs.items()), flush=True)
            fp64_ipl_cell = expo == 6 and dtype == torch.float64 and tc.requires_grad
            assert ("Mipme" in served) != fp64_ipl_cell, served
            assert max(errs.values()) < tol, (key, errs)
print("front contract parity OK")

if "--no-timing" in sys.argv:
    sys.exit(0)

w = workloads.water_box()
dt = torch.float32
tp, tq, tc = (torch.tensor(x, dtype=dt, device=dev) for x in (w.positions, w.charges, w.cell))
ti = torch.tensor(w.pairs, device=dev)
ts = torch.tensor(w.shifts, dtype=dt, device=dev)
calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)


def loop(want_q, want_c, n, general=False):
    tp.requires_grad_(True)
    tq.requires_grad_(want_q)
    tc.requires_grad_(want_c)
    wv = torch.randn_like(tq) if general else None

    def one():
        for t in (tp, tq, tc):
            t.grad = None
        d = tpa.pair_distances(tp, ti, tc, ts)
        V = calc(tq, tc, tp, ti, d)
        ((tq if wv is None else wv) * V).sum().backward()

    for _ in range(30):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


import gc

gc.disable()
cases = (("E+F", False, False), ("E+F+dq", True, False), ("E+F+dcell", False, True), ("E+F+dq+dcell", True, True), ("E+F", False, False))
for label, a, b in cases:
    ms = [loop(a, b, 200) for _ in range(3)]
    msg = [loop(a, b, 200, general=True) for _ in range(2)]
    print(f"eager {label:14s} {min(ms):.4f} ms ({' '.join(f'{x:.4f}' for x in ms)})   general seed {min(msg):.4f} ms", flush=True)

"""BASELINE.json configs[1] (cfg2: 8 000 ions, P3M n=4, 32^3) and configs[2] (cfg3: the 31 944-atom water box, P3M n=5, 64^3)
evaluated by the REFERENCE ITSELF at full size (build container only; the reference is imported, never copied), committed as
tests/golden/ref_fullsize.npz.  Round-4 verdict, "What's weak" 1: until now the full-size numbers came from the pinned oracle
(make_workloads_golden.py); with this file the headline configuration is pinned by the reference's own output.

    python tests/golden/make_reference_fullsize.py [ionic water]

Per workload `k` and precision `p` in {f64, f32}, with the same samples / checksum vectors as workloads.npz (default_rng(4242):
r (N,3), s (N,1), 256 sampled atoms):
  k_p_energy                 E = sum q V         (torchpme.P3MCalculator.forward, calculators/calculator.py:103-189)
  k_p_potential_sample/_dot  V
  k_p_force_sample/_sq/_dot  F = -dE/dpositions through the reference's own distance expression (tests/helpers.py:278-304)
  k_p_charge_grad_sample/_dot, k_p_cell_grad      dE/dq, dE/dcell (autograd, tests/calculators/test_workflow.py:164-192)
  k_p_sumseed_{value,pos_sample,pos_dot,charge_sample,charge_dot,cell}   L = V.sum() with constant distances, the protocol of
                                                                         tuning/tuner.py:350-369
plus k_n_pairs and k_pos_checksum (the synthetic box the numbers belong to)."""
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
_v = types.ModuleType("torchpme._version")
_v.__version__ = "0.0.0"
_v.__version_tuple__ = (0, 0, 0)
sys.modules["torchpme._version"] = _v
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, ROOT)
import torchpme  # noqa: E402  (the reference)

from torchpme_amd import workloads  # noqa: E402  (inputs only: the synthetic boxes of SURVEY 8d)

# "dispersion" = BASELINE.json configs[4] (cfg5: 262 144 atoms, 1/r^6, P3M n=5, 128^3): round-5 verdict, missing 2 -- until round 6
# it was held by the oracle's numbers only, and the oracle's p = 6 branch was pinned to the reference on a 7-atom cell
MAKERS = {"ionic": workloads.ionic_box, "water": workloads.water_box, "dispersion": workloads.dispersion_box}


def evaluate(w, dtype):
    t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)  # noqa: E731
    pairs = torch.tensor(w.pairs)
    shifts = t(w.shifts)
    # (potentials/coulomb.py, potentials/inversepowerlaw.py:55-169 with lib/math.py:85-104 for p = 6)
    pot = (torchpme.CoulombPotential(smearing=w.smearing) if w.exponent == 1
           else torchpme.InversePowerLawPotential(exponent=w.exponent, smearing=w.smearing))
    calc = torchpme.P3MCalculator(pot, mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order).to(dtype)
    pos, q, cell = t(w.positions).requires_grad_(True), t(w.charges).requires_grad_(True), t(w.cell).requires_grad_(True)
    # the reference's caller-side distances (tests/helpers.py:278-304)
    vec = pos[pairs[:, 1]] - pos[pairs[:, 0]] + shifts @ cell
    d = torch.linalg.norm(vec, dim=1)
    V = calc.forward(q, cell, pos, pairs, d)
    E = (q * V).sum()
    gp, gq, gc = torch.autograd.grad(E, (pos, q, cell))
    # the timing protocol: result.sum().backward() with the distances a constant input (tuning/tuner.py:350-369)
    pos2, q2, cell2 = t(w.positions).requires_grad_(True), t(w.charges).requires_grad_(True), t(w.cell).requires_grad_(True)
    V2 = calc.forward(q2, cell2, pos2, pairs, d.detach())
    L = V2.sum()
    sp, sq, sc = torch.autograd.grad(L, (pos2, q2, cell2))
    n = lambda x: x.detach().double().numpy()  # noqa: E731
    return dict(V=n(V), E=float(E), F=-n(gp), dq=n(gq), dcell=n(gc), L=float(L), sp=n(sp), sq=n(sq), sc=n(sc))


def main(names):
    path = os.path.join(HERE, "ref_fullsize.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    torch.manual_seed(0)
    for name in names:
        t0 = time.time()
        w = MAKERS[name]()
        rng = np.random.default_rng(4242)
        r = rng.normal(size=(w.n_atoms, 3))
        s = rng.normal(size=(w.n_atoms, 1))
        sample = np.sort(rng.choice(w.n_atoms, size=min(256, w.n_atoms), replace=False))
        out[f"{name}_n_pairs"] = np.asarray(w.n_pairs)
        out[f"{name}_sample"] = sample
        out[f"{name}_pos_checksum"] = np.array([w.positions.sum(), (w.positions**2).sum(), w.charges.sum(), (w.charges**2).sum()])
        t1 = time.time()
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            t2 = time.time()
            res = evaluate(w, dtype)
            print(f"{name} {tag}: reference forward + 2 x autograd in {time.time() - t2:.0f} s", flush=True)
            k = f"{name}_{tag}_"
            out[k + "energy"] = np.asarray(res["E"])
            out[k + "potential_sample"] = res["V"][sample, 0]
            out[k + "potential_dot"] = np.asarray(float((s * res["V"]).sum()))
            out[k + "force_sample"] = res["F"][sample]
            out[k + "force_sq"] = np.asarray(float((res["F"] ** 2).sum()))
            out[k + "force_dot"] = np.asarray(float((r * res["F"]).sum()))
            out[k + "charge_grad_sample"] = res["dq"][sample, 0]
            out[k + "charge_grad_dot"] = np.asarray(float((s * res["dq"]).sum()))
            out[k + "cell_grad"] = res["dcell"]
            out[k + "sumseed_value"] = np.asarray(res["L"])
            out[k + "sumseed_pos_sample"] = res["sp"][sample]
            out[k + "sumseed_pos_dot"] = np.asarray(float((r * res["sp"]).sum()))
            out[k + "sumseed_charge_sample"] = res["sq"][sample, 0]
            out[k + "sumseed_charge_dot"] = np.asarray(float((s * res["sq"]).sum()))
            out[k + "sumseed_cell"] = res["sc"]
            print(f"{name} {tag}: N={w.n_atoms} P={w.n_pairs} E={res['E']:.10f} |F|^2={(res['F']**2).sum():.6f} L={res['L']:.8f}",
                  flush=True)
        print(f"{name}: list {t1 - t0:.0f} s, reference {time.time() - t1:.0f} s", flush=True)
    np.savez(path, **out)


if __name__ == "__main__":
    main(sys.argv[1:] or ["ionic", "water"])

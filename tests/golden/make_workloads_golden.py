"""Oracle energies and force checksums of the benchmark workloads (BASELINE.json configs[1], [2], [4] at FULL size), computed
with the pinned oracle (oracle/pme_numpy.py, fp64: forward + analytic adjoint, pair part chained through the distances) and
committed as tests/golden/workloads.npz.  bench.py's `accuracy` block and tests/test_gpu_fullsize.py compare the HIP path with
these numbers -- the 39 M-pair cfg5 evaluation takes minutes of NumPy and tens of GB, too much for the GPU box's test run.

    python tests/golden/make_workloads_golden.py [water ionic dispersion]

Per workload `k`: k_energy, k_force_sample (256 atoms, indices k_sample), k_force_sq (sum |F|^2), k_force_dot (sum_a r_a . F_a
with r = default_rng(4242).normal((N, 3)): a checksum of the whole array), k_potential_dot (sum_a s_a V_a, s likewise),
k_n_pairs, k_pos_checksum (sum of positions, sum of squares: the synthetic box is the one the numbers belong to).  Round 4: the
whole autograd contract -- k_charge_grad_sample / k_charge_grad_dot (dE/dq), k_cell_grad (dE/dcell, 3x3) -- and the gradients of
the reference's timing protocol L = V.sum() with constant distances: k_sumseed_{value, pos_sample, pos_dot, charge_sample,
charge_dot, cell}."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pme_numpy as O  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

MAKERS = {"water": workloads.water_box, "ionic": workloads.ionic_box, "dispersion": workloads.dispersion_box}


def summarise(w):
    spec = O.PotentialSpec("coulomb" if w.exponent == 1 else "ipl", w.exponent, w.smearing, 1.0)
    dist = O.pair_distances(w.positions, w.cell, w.pairs, w.shifts)[0]
    V, cache = O.forward(spec, w.scheme, w.order, w.mesh_spacing, w.charges, w.cell, w.positions, w.pairs, dist,
                         return_cache=True)
    gr = O.backward(cache, w.charges)
    gpos_d, gcell_d = O.pair_distances_backward(w.positions, w.cell, w.pairs, w.shifts, gr["dist"])
    F = -(gr["positions"] + gpos_d)
    # the rest of the autograd contract of E = sum q V (reference: tests/calculators/test_workflow.py:164-192): dE/dq = V + the
    # adjoint's charge part, dE/dcell = mesh part + the pair part chained through the distances
    dEdq = gr["charges"] + V
    dEdcell = gr["cell"] + gcell_d
    # ... and of the reference's own timing protocol (tuning/tuner.py:350-369): L = V.sum() with the distances a CONSTANT input
    # (gradients w.r.t. positions and cell come from the mesh part alone)
    gs = O.backward(cache, np.ones_like(w.charges))
    rng = np.random.default_rng(4242)
    r = rng.normal(size=(w.n_atoms, 3))
    s = rng.normal(size=(w.n_atoms, 1))
    sample = np.sort(rng.choice(w.n_atoms, size=min(256, w.n_atoms), replace=False))
    return {
        "energy": float((V * w.charges).sum()),
        "sample": sample,
        "force_sample": F[sample],
        "potential_sample": V[sample, 0],
        "force_sq": float((F * F).sum()),
        "force_dot": float((r * F).sum()),
        "potential_dot": float((s * V).sum()),
        "charge_grad_sample": dEdq[sample, 0],
        "charge_grad_dot": float((s * dEdq).sum()),
        "cell_grad": dEdcell,
        "sumseed_value": float(V.sum()),
        "sumseed_pos_sample": gs["positions"][sample],
        "sumseed_pos_dot": float((r * gs["positions"]).sum()),
        "sumseed_charge_sample": gs["charges"][sample, 0],
        "sumseed_charge_dot": float((s * gs["charges"]).sum()),
        "sumseed_cell": gs["cell"],
        "n_pairs": w.n_pairs,
        "pos_checksum": np.array([w.positions.sum(), (w.positions**2).sum(), w.charges.sum(), (w.charges**2).sum()]),
    }


def main(names):
    path = os.path.join(ROOT, "tests", "golden", "workloads.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in names:
        t0 = time.time()
        w = MAKERS[name]()
        t1 = time.time()
        res = summarise(w)
        for k, v in res.items():
            out[f"{name}_{k}"] = np.asarray(v)
        print(f"{name}: N={w.n_atoms} P={w.n_pairs} E={res['energy']:.10f} |F|^2={res['force_sq']:.6f} "
              f"(list {t1 - t0:.0f} s, oracle {time.time() - t1:.0f} s)", flush=True)
        np.savez_compressed(path, **out)


if __name__ == "__main__":
    main(sys.argv[1:] or ["ionic", "water", "dispersion"])

"""Second-order golden vectors made by IMPORTING the reference (build container only).

    python tests/golden/make_second_order_golden.py

For every case of ``ref_small.npz`` (same inputs: 7-atom triclinic cell, every scheme / order / potential / slab / exclusion /
full-list combination on the path) the reference evaluates, in float64,

    S  = sum(g * V(q, cell, r, d))                                     V = Calculator.forward
    G  = dS / d(q, r, d)                  with create_graph=True        (first-order gradients of ref_small.npz)
    L  = sum_k <w_k, G_k>                                               w_k: seeded random cotangents, stored
    H  = dL / d(q, cell, r, d, g)                                       Hessian-vector products, every block at once

(No ``dS/dcell`` among the G: differentiating the reference's OWN cell gradient a second time returns NaN in every block -- even
with a zero cotangent, 0 * inf at the k = 0 point of its filter -- for every case below, so the cell enters these vectors as
the variable of the second differentiation only (``H_cell``: the stress of a force loss).  The cell-cell block is compared with
the torch oracle in tests/test_gpu_analytic.py.)

and, for three cases, one more level: ``T = d<w3, H_q> / dq`` with ``H_q`` formed with create_graph=True (third order).  Only
data are written (``second_order.npz``): inputs come from ref_small.npz, outputs are arrays.  The same G, L, H for every
``EwaldCalculator`` case of ``ref_ewald.npz`` (keys ``e..``).
"""

import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (imports the reference, with its two stubs)

torchpme = MG.torchpme


def main():
    z = np.load(os.path.join(HERE, "ref_small.npz"))
    rng = np.random.default_rng(20260929)
    out = {"names": z["names"]}
    third = {"c03", "c08", "c11"}
    for nm in (str(n) for n in z["names"]):
        meta = ast.literal_eval(str(z[f"{nm}/meta"]))
        if meta["kind"] == "coulomb":
            pot = torchpme.CoulombPotential(smearing=meta["smearing"], prefactor=meta["prefactor"],
                                            exclusion_radius=meta["exclusion_radius"])
        else:
            pot = torchpme.InversePowerLawPotential(exponent=meta["exponent"], smearing=meta["smearing"],
                                                    prefactor=meta["prefactor"], exclusion_radius=meta["exclusion_radius"])
        Calc = torchpme.P3MCalculator if meta["scheme"] == "P3M" else torchpme.PMECalculator
        calc = Calc(pot, mesh_spacing=meta["mesh_spacing"], interpolation_nodes=meta["order"], full_neighbor_list=meta["full_list"])
        t = lambda key: torch.tensor(z[f"{nm}/{key}"], dtype=torch.float64, requires_grad=True)  # noqa: E731
        q, pos, d, g = t("charges"), t("positions"), t("dist"), t("g")
        cell = torch.tensor(z["cell"], dtype=torch.float64, requires_grad=True)
        pairs = torch.tensor(z[f"{nm}/pairs"])
        per = None if meta["periodic"] is None else torch.tensor(meta["periodic"])
        V = calc(q, cell, pos, pairs, d, periodic=per)
        S = (V * g).sum()
        G = torch.autograd.grad(S, (q, pos, d), create_graph=True)
        w = [torch.tensor(rng.normal(size=tuple(x.shape))) for x in G]
        L = sum((wk * Gk).sum() for wk, Gk in zip(w, G))
        want3 = nm in third
        H = torch.autograd.grad(L, (q, cell, pos, d, g), create_graph=want3, allow_unused=True)
        H = [torch.zeros_like(x) if h is None else h for h, x in zip(H, (q, cell, pos, d, g))]  # (not reached: exactly zero)
        for key, val in zip(("w_charges", "w_positions", "w_dist"), w):
            out[f"{nm}/{key}"] = val.numpy()
        for key, val in zip(("H_charges", "H_cell", "H_positions", "H_dist", "H_g"), H):
            out[f"{nm}/{key}"] = val.detach().numpy()
        if want3:
            w3 = torch.tensor(rng.normal(size=tuple(q.shape)))
            T = torch.autograd.grad((w3 * H[0]).sum(), (q, pos, g), allow_unused=True)
            T = [torch.zeros_like(x) if v is None else v for v, x in zip(T, (q, pos, g))]
            out[f"{nm}/w3"] = w3.numpy()
            for key, val in zip(("T_charges", "T_positions", "T_g"), T):
                out[f"{nm}/{key}"] = val.numpy()
    # ---- the explicit Ewald sum: every case of ref_ewald.npz, same quantities (prefix e..)
    ze = np.load(os.path.join(HERE, "ref_ewald.npz"))
    out["ewald_names"] = ze["names"]
    for nm in (str(n) for n in ze["names"]):
        meta = ast.literal_eval(str(ze[f"{nm}/meta"]))
        pot = (torchpme.CoulombPotential(smearing=meta["smearing"], prefactor=meta["prefactor"]) if meta["kind"] == "coulomb"
               else torchpme.InversePowerLawPotential(exponent=meta["exponent"], smearing=meta["smearing"],
                                                      prefactor=meta["prefactor"]))
        calc = torchpme.EwaldCalculator(pot, lr_wavelength=meta["lr_wavelength"], full_neighbor_list=meta["full_list"])
        t = lambda key: torch.tensor(ze[f"{nm}/{key}"], dtype=torch.float64, requires_grad=True)  # noqa: E731
        q, pos, d, g = t("charges"), t("positions"), t("dist"), t("g")
        cell = torch.tensor(ze["cell"], dtype=torch.float64, requires_grad=True)
        per = None if meta["periodic"] is None else torch.tensor(meta["periodic"])
        kv = torch.tensor(ze[f"{nm}/kvectors"]) if meta["own_kvectors"] else None
        mask = torch.tensor(ze[f"{nm}/node_mask"]) if meta["node_mask"] else None
        V = calc(q, cell, pos, torch.tensor(ze[f"{nm}/pairs"]), d, periodic=per, kvectors=kv, node_mask=mask)
        G = torch.autograd.grad((V * g).sum(), (q, pos, d), create_graph=True)
        w = [torch.tensor(rng.normal(size=tuple(x.shape))) for x in G]
        L = sum((wk * Gk).sum() for wk, Gk in zip(w, G))
        H = torch.autograd.grad(L, (q, cell, pos, d, g), allow_unused=True)
        H = [torch.zeros_like(x) if h is None else h for h, x in zip(H, (q, cell, pos, d, g))]
        for key, val in zip(("w_charges", "w_positions", "w_dist"), w):
            out[f"{nm}/{key}"] = val.numpy()
        for key, val in zip(("H_charges", "H_cell", "H_positions", "H_dist", "H_g"), H):
            out[f"{nm}/{key}"] = val.detach().numpy()
    bad = [k for k, v in out.items() if k not in ("names", "ewald_names") and not np.isfinite(v).all()]
    assert not bad, bad
    np.savez(os.path.join(HERE, "second_order.npz"), **out)
    print("second_order.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()

"""Measured errors of the HIP path against the REFERENCE'S OWN full-size evaluation (tests/golden/ref_fullsize.npz) for the three
benchmark boxes: the numbers the tolerances of tests/test_gpu_fullsize.py::test_fullsize_against_the_reference_itself are five
times of.   python tools/r06/fullsize_errors.py [ionic water dispersion]  (GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import workloads  # noqa: E402
from test_gpu_fullsize import Box  # noqa: E402

MAKERS = {"ionic": workloads.ionic_box, "water": workloads.water_box, "dispersion": workloads.dispersion_box}
z = np.load(os.path.join(ROOT, "tests", "golden", "ref_fullsize.npz"))
relmax = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())  # noqa: E731
rell2 = lambda a, b: float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))  # noqa: E731
for cfg in sys.argv[1:] or list(MAKERS):
    w = MAKERS[cfg]()
    g = {k[len(cfg) + 5:]: z[k] for k in z.files if k.startswith(cfg + "_f64_")}
    g32 = {k[len(cfg) + 5:]: z[k] for k in z.files if k.startswith(cfg + "_f32_")}
    sample = z[f"{cfg}_sample"]
    print(f"{cfg}: reference fp32 vs reference fp64: E {abs(float(g32['energy']) / float(g['energy']) - 1):.2e}  F {rell2(g32['force_sample'], g['force_sample']):.2e}"
          f"  dq {relmax(g32['charge_grad_sample'], g['charge_grad_sample']):.2e}  dcell {rell2(g32['cell_grad'], g['cell_grad']):.2e}")
    for dtype in (torch.float64, torch.float32):
        box = Box(w, dtype)
        step = tpa.GraphedEnergyForces(box.calc, box.q, box.cell, box.pos, box.pairs, box.shifts, charge_gradient=True, cell_gradient=True)
        res = {"graph": tuple(x.clone() for x in step())}
        del step
        p_, q_, c_ = box.pos.clone().requires_grad_(True), box.q.clone().requires_grad_(True), box.cell.clone().requires_grad_(True)
        d = tpa.pair_distances(p_, box.pairs, c_, box.shifts)
        E = (box.calc(q_, c_, p_, box.pairs, d) * q_).sum()
        E.backward()
        res["eager"] = (E.detach(), -p_.grad, q_.grad, c_.grad)
        for name, (E, F, dq, dc) in res.items():
            F, dq, dc = F.cpu().double().numpy(), dq.cpu().double().numpy(), dc.cpu().double().numpy()
            print(f"  {str(dtype)[6:]:8s}{name:6s} E {abs(float(E) - float(g['energy'])) / abs(float(g['energy'])):.2e}  F(256, rel-L2) {rell2(F[sample], g['force_sample']):.2e}"
                  f"  F(256, max) {relmax(F[sample], g['force_sample']):.2e}  |F|^2 {abs(float((F * F).sum()) / float(g['force_sq']) - 1):.2e}"
                  f"  dq {relmax(dq[sample, 0], g['charge_grad_sample']):.2e}  dcell {rell2(dc, g['cell_grad']):.2e}")
        d_fixed = tpa.pair_distances(box.pos, box.pairs, box.cell, box.shifts).detach().clone()
        positions, cl, charges = box.pos.clone(), box.cell.clone(), box.q.clone()
        for x in (positions, cl, charges):
            x.requires_grad_(True)
        V = box.calc.forward(positions=positions, charges=charges, cell=cl, neighbor_indices=box.pairs, neighbor_distances=d_fixed)
        V.sum().backward()
        Vn = V.detach().cpu().double().numpy()
        print(f"  {str(dtype)[6:]:8s}V.sum() V {relmax(Vn[sample, 0], g['potential_sample']):.2e}  pos {rell2(positions.grad.cpu().double().numpy()[sample], g['sumseed_pos_sample']):.2e}"
              f"  q {relmax(charges.grad.cpu().double().numpy()[sample, 0], g['sumseed_charge_sample']):.2e}  cell {rell2(cl.grad.cpu().double().numpy(), g['sumseed_cell']):.2e}")
        del box
        torch.cuda.empty_cache()

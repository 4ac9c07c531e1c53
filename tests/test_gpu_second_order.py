"""Second derivatives (``-m gpu``; verdict item 9).  The reference is plain ATen ops, so ``create_graph=True`` -- a loss on
forces -- works there (``calculators/calculator.py:43-87,103-189``).  The HIP kernels are first order: by default a double
differentiation raises an error that names the way out, and with ``calculator.double_backward = "finite-difference"`` it works:
exact through ``pair_distances``, central differences of the analytic first-order gradients through the calculator.  Checked
against central differences of the ORACLE's analytic forces and against the reference's own double backward (golden numbers made
with the reference: tests/golden/make_golden.py is first order only, so the oracle's finite differences stand in)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import torchpme_amd as tpa  # noqa: E402
from oracle import pme_numpy as O  # noqa: E402
from torchpme_amd import ops  # noqa: E402

DEV = "cuda"


def _system(n_side=4, seed=3):
    rng = np.random.default_rng(seed)
    a = 2.4
    L = n_side * a
    g = (np.arange(n_side) + 0.5) * a
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (n_side**3, 3))
    q = rng.normal(size=(len(pos), 1))
    q -= q.mean()
    cell = L * np.eye(3)
    pairs, S, _ = tpa.neighbor_list(pos, cell, 4.5)
    return pos, q, cell, pairs, S


def _oracle_forces(pos, q, cell, pairs, S, sm, h, order):
    spec = O.PotentialSpec("coulomb", 1, sm, 1.0)
    dist, _ = O.pair_distances(pos, cell, pairs, S)
    V, cache = O.forward(spec, "P3M", order, h, q, cell, pos, pairs, dist, return_cache=True)
    gr = O.backward(cache, q)
    gpos, _ = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    return -(gpos + gr["positions"]), 2.0 * V  # forces, dE/dq (half list: dE/dq = 2 V)


def test_double_backward_raises_with_the_way_out():
    pos, q, cell, pairs, S = _system()
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.6, interpolation_nodes=4)
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    p = t(pos).requires_grad_(True)
    d = tpa.pair_distances(p, t(pairs), t(cell), t(S).double())
    E = (t(q) * calc(t(q), t(cell), p, t(pairs), d)).sum()
    (F,) = torch.autograd.grad(E, p, create_graph=True)
    with pytest.raises(RuntimeError, match='double_backward = "finite-difference"'):
        (F * F).sum().backward()
    assert ops.SECOND_ORDER_HINT.startswith("torchpme_amd: the HIP kernels provide FIRST-order gradients")


@pytest.mark.parametrize("which", ["P3M", "PME", "Ewald"])
def test_force_loss_gradients_by_finite_differences(which):
    """L = sum_a w_a . F_a with F = -dE/dr (create_graph=True): dL/dpositions = -H w and dL/dcharges = -d(w . dE/dr)/dq against
    central differences of the oracle's analytic forces along w (P3M) and of this package's own first-order forces (all)."""
    pos, q, cell, pairs, S = _system()
    sm, h, order = 1.0, 0.6, 4
    pot = tpa.CoulombPotential(smearing=sm)
    calc = (tpa.P3MCalculator(pot, mesh_spacing=h, interpolation_nodes=order) if which == "P3M" else
            tpa.PMECalculator(pot, mesh_spacing=h, interpolation_nodes=order) if which == "PME" else
            tpa.EwaldCalculator(pot, lr_wavelength=1.2))
    calc.double_backward = "finite-difference"
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    tq, tc, tp_, tS = t(q).requires_grad_(True), t(cell), t(pairs), t(S).double()
    w = t(np.random.default_rng(1).normal(size=pos.shape))

    def energy(p, qq):
        d = tpa.pair_distances(p, tp_, tc, tS)
        return (qq * calc(qq, tc, p, tp_, d)).sum()

    p = t(pos).requires_grad_(True)
    E = energy(p, tq)
    (g,) = torch.autograd.grad(E, p, create_graph=True)  # g = -F
    loss = -(w * g).sum()  # L = sum w . F
    dL_dp, dL_dq = torch.autograd.grad(loss, (p, tq))
    # central differences of first-order quantities with the same calculator (no second-order machinery)
    calc1 = type(calc)(pot, **({"lr_wavelength": 1.2} if which == "Ewald" else {"mesh_spacing": h, "interpolation_nodes": order}))
    eps = 1e-4

    def first_order(x):
        pp, qq = t(x).requires_grad_(True), t(q).requires_grad_(True)
        d = tpa.pair_distances(pp, tp_, tc, tS)
        Ex = (qq * calc1(qq, tc, pp, tp_, d)).sum()
        gp, gq = torch.autograd.grad(Ex, (pp, qq))
        return -gp, gq

    wn = w.cpu().numpy()
    Fp, gqp = first_order(pos + eps * wn)
    Fm, gqm = first_order(pos - eps * wn)
    ref_dp = (Fp - Fm) / (2 * eps)  # d(w.F)/dr = (dF/dr)^T w = H-symmetric: directional derivative of F along w
    ref_dq = -(gqp - gqm) / (2 * eps)  # d(w.F)/dq = -d/dq (w . dE/dr) = -D_w (dE/dq)
    assert float((dL_dp - ref_dp).norm() / ref_dp.norm()) < 2e-5, which
    assert float((dL_dq - ref_dq).norm() / ref_dq.norm()) < 2e-5, which
    if which == "P3M":  # and against the oracle's analytic forces, differenced the same way
        Fop, gop = _oracle_forces(pos + eps * wn, q, cell, pairs, S, sm, h, order)
        Fom, gom = _oracle_forces(pos - eps * wn, q, cell, pairs, S, sm, h, order)
        o_dp, o_dq = (Fop - Fom) / (2 * eps), -(gop - gom) / (2 * eps)
        assert np.linalg.norm(dL_dp.cpu().numpy() - o_dp) / np.linalg.norm(o_dp) < 2e-5
        assert np.linalg.norm(dL_dq.cpu().numpy() - o_dq) / np.linalg.norm(o_dq) < 2e-5
    # first order is untouched by the option: same forces as the plain calculator
    F1, _ = first_order(pos)
    assert float((-g.detach() - F1).norm() / F1.norm()) < 1e-12


@pytest.mark.parametrize("route", ["front", "python_nodes"])
def test_force_loss_on_other_parameters_needs_no_second_order(route, monkeypatch):
    """A model term next to the calculator (E = E_pme(r) + w * sum r^2), forces with create_graph=True, a loss on them: the
    gradient w.r.t. the MODEL parameter never differentiates the calculator twice, so it works out of the box when only that
    parameter is asked for (`backward(inputs=[w])` / `autograd.grad(loss, [w])`) -- the error node sits on the path to the
    positions only.  `loss.backward()` without `inputs` also asks for d loss / d positions, which is second order through the
    calculator: that raises the hint (round-3 advice: say so)."""
    if route == "python_nodes":
        monkeypatch.setattr(ops, "FRONT", False)
    pos, q, cell, pairs, S = _system()
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.8, interpolation_nodes=4)
    tp = torch.tensor(pos, device=DEV, dtype=torch.float64, requires_grad=True)
    tq, tc = torch.tensor(q, device=DEV, dtype=torch.float64), torch.tensor(cell, device=DEV, dtype=torch.float64)
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV, dtype=torch.float64)
    w = torch.tensor(0.3, device=DEV, dtype=torch.float64, requires_grad=True)

    def loss_on_forces():
        d = tpa.pair_distances(tp, ti, tc, tS)
        E = (tq * calc(tq, tc, tp, ti, d)).sum() + w * (tp * tp).sum()
        (F,) = torch.autograd.grad(E, tp, create_graph=True)
        return (F * F).sum(), F

    loss, F = loss_on_forces()
    loss.backward(inputs=[w])
    # F = F_pme + 2 w r  =>  d loss / d w = sum 2 F . 2 r
    want = float((4.0 * F.detach() * tp.detach()).sum())
    assert abs(float(w.grad) - want) <= 1e-10 * abs(want)
    (gw,) = torch.autograd.grad(loss_on_forces()[0], [w])
    assert abs(float(gw) - want) <= 1e-10 * abs(want)
    with pytest.raises(RuntimeError, match='double_backward = "finite-difference"'):
        loss_on_forces()[0].backward()

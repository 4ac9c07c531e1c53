"""``calculator.double_backward = "analytic"`` (``-m gpu``): the path from differentiable primitives (torch-pme_amd/analytic.py,
csrc/jets.hip) against the REFERENCE's own autograd -- first order (``ref_small.npz``), second and third order
(``second_order.npz``, tests/golden/make_second_order_golden.py), every scheme / order / potential / slab / exclusion /
full-list case on the path, every block of the Hessian-vector product (charges, cell, positions, distances, upstream gradient).
The reference is plain ATen ops and differentiates to any order (``calculators/calculator.py:43-87,103-189``,
``calculators/pme.py:88-143``); tolerance: float64 1e-9 relative (observed ~1e-13), float32 1e-4 against the float64 numbers.
Plus: ``gradcheck`` / ``gradgradcheck`` of every primitive, the fused first-order path as a second witness, a force loss with
learned charges against the torch oracle's double backward at 512 atoms."""

import ast

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import analytic, ops  # noqa: E402

from tests.test_gpu_parity import make_calc, relmax  # noqa: E402

DEV = "cuda"


def _case(z, nm, dtype=torch.float64):
    meta = ast.literal_eval(str(z[f"{nm}/meta"]))
    calc = make_calc(meta)
    calc.double_backward = "analytic"
    t = lambda k: torch.tensor(z[f"{nm}/{k}"], device=DEV, dtype=dtype, requires_grad=True)  # noqa: E731
    q, pos, d, g = t("charges"), t("positions"), t("dist"), t("g")
    cell = torch.tensor(z["cell"], device=DEV, dtype=dtype, requires_grad=True)
    pairs = torch.tensor(z[f"{nm}/pairs"], device=DEV)
    per = None if meta["periodic"] is None else torch.tensor(meta["periodic"], device=DEV)
    return meta, calc, q, cell, pos, pairs, d, g, per


def test_first_order_against_the_reference(golden_dir):
    z = np.load(f"{golden_dir}/ref_small.npz")
    worst = {}
    for nm in (str(n) for n in z["names"]):
        meta, calc, q, cell, pos, pairs, d, g, per = _case(z, nm)
        V = calc(q, cell, pos, pairs, d, periodic=per)
        (V * g.detach()).sum().backward()
        errs = dict(V=relmax(V.detach().cpu(), z[f"{nm}/V"]), q=relmax(q.grad.cpu(), z[f"{nm}/grad_charges"]),
                    pos=relmax(pos.grad.cpu(), z[f"{nm}/grad_positions"]), cell=relmax(cell.grad.cpu(), z[f"{nm}/grad_cell"]),
                    d=relmax(d.grad.cpu(), z[f"{nm}/grad_dist"]))
        for k, v in errs.items():
            assert v < 1e-9, (nm, meta, k, v)
            worst[k] = max(worst.get(k, 0.0), v)
    print("analytic path, first order, worst rel errors:", worst)


def _second_order(calc, q, cell, pos, pairs, d, g, per, w, create_graph=False):
    V = calc(q, cell, pos, pairs, d, periodic=per)
    S = (V * g).sum()
    G = torch.autograd.grad(S, (q, pos, d), create_graph=True)
    L = sum((wk * Gk).sum() for wk, Gk in zip(w, G))
    H = torch.autograd.grad(L, (q, cell, pos, d, g), create_graph=create_graph, allow_unused=True)
    return [torch.zeros_like(x) if h is None else h for h, x in zip(H, (q, cell, pos, d, g))]


@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-9), ("f32", torch.float32, 2e-4)])
def test_second_order_against_the_reference(golden_dir, tag, dtype, tol):
    """H = d<w, dS/d(q, r, d)>/dz for z = (charges, cell, positions, distances, g), S = sum(g V): every block the reference can
    form (its own second derivative of dS/dcell is NaN: tests/golden/make_second_order_golden.py), every case."""
    z, s = np.load(f"{golden_dir}/ref_small.npz"), np.load(f"{golden_dir}/second_order.npz")
    worst = {}
    for nm in (str(n) for n in z["names"]):
        meta, calc, q, cell, pos, pairs, d, g, per = _case(z, nm, dtype)
        w = [torch.tensor(s[f"{nm}/w_{k}"], device=DEV, dtype=dtype) for k in ("charges", "positions", "dist")]
        H = _second_order(calc, q, cell, pos, pairs, d, g, per, w)
        for key, h in zip(("charges", "cell", "positions", "dist", "g"), H):
            ref = s[f"{nm}/H_{key}"]
            err = float(np.abs(h.double().cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-300)) if np.abs(ref).max() > 0 \
                else float(h.abs().max())
            assert err < tol, (nm, meta, key, err)
            worst[key] = max(worst.get(key, 0.0), err)
    print(f"analytic path, second order {tag}, worst rel errors:", worst)


def test_third_order_against_the_reference(golden_dir):
    z, s = np.load(f"{golden_dir}/ref_small.npz"), np.load(f"{golden_dir}/second_order.npz")
    n = 0
    for nm in (str(x) for x in z["names"]):
        if f"{nm}/w3" not in s:
            continue
        n += 1
        meta, calc, q, cell, pos, pairs, d, g, per = _case(z, nm)
        w = [torch.tensor(s[f"{nm}/w_{k}"], device=DEV) for k in ("charges", "positions", "dist")]
        H = _second_order(calc, q, cell, pos, pairs, d, g, per, w, create_graph=True)
        T = torch.autograd.grad((torch.tensor(s[f"{nm}/w3"], device=DEV) * H[0]).sum(), (q, pos, g), allow_unused=True)
        for key, tt, x in zip(("charges", "positions", "g"), T, (q, pos, g)):
            ref = s[f"{nm}/T_{key}"]
            got = np.zeros_like(ref) if tt is None else tt.cpu().numpy()
            assert np.abs(got - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-30) + 1e-300, (nm, meta, key)
    assert n == 3


def _ewald_case(ze, nm):
    meta = ast.literal_eval(str(ze[f"{nm}/meta"]))
    pot = (tpa.CoulombPotential(smearing=meta["smearing"], prefactor=meta["prefactor"]) if meta["kind"] == "coulomb"
           else tpa.InversePowerLawPotential(exponent=meta["exponent"], smearing=meta["smearing"], prefactor=meta["prefactor"]))
    calc = tpa.EwaldCalculator(pot, lr_wavelength=meta["lr_wavelength"], full_neighbor_list=meta["full_list"])
    calc.double_backward = "analytic"
    t = lambda k, grad=True: torch.tensor(ze[f"{nm}/{k}"], device=DEV, requires_grad=grad)  # noqa: E731
    kw = dict(periodic=None if meta["periodic"] is None else torch.tensor(meta["periodic"], device=DEV),
              kvectors=t("kvectors", False) if meta["own_kvectors"] else None,
              node_mask=t("node_mask", False) if meta["node_mask"] else None)
    cell = torch.tensor(ze["cell"], device=DEV, requires_grad=True)
    return meta, calc, t("charges"), cell, t("positions"), t("pairs", False), t("dist"), t("g"), kw


def test_ewald_first_and_second_order_against_the_reference(golden_dir):
    """EwaldCalculator through the analytic route: V, the four gradients and every Hessian-vector block the reference can form,
    for every case of ref_ewald.npz (Coulomb / 1/r^p, channels, slab, own k-vectors, node mask, full list)."""
    ze, s = np.load(f"{golden_dir}/ref_ewald.npz"), np.load(f"{golden_dir}/second_order.npz")
    for nm in (str(n) for n in ze["names"]):
        meta, calc, q, cell, pos, pairs, d, g, kw = _ewald_case(ze, nm)
        V = calc(q, cell, pos, pairs, d, **kw)
        assert relmax(V.detach().cpu(), ze[f"{nm}/V"]) < 1e-9, (nm, meta)
        S = (V * g).sum()
        first = torch.autograd.grad(S, (q, pos, cell, d), retain_graph=True)
        for key, got in zip(("charges", "positions", "cell", "dist"), first):
            assert relmax(got.cpu(), ze[f"{nm}/grad_{key}"]) < 1e-9, (nm, meta, key)
        G = torch.autograd.grad(S, (q, pos, d), create_graph=True)
        w = [torch.tensor(s[f"{nm}/w_{k}"], device=DEV) for k in ("charges", "positions", "dist")]
        H = torch.autograd.grad(sum((wk * Gk).sum() for wk, Gk in zip(w, G)), (q, cell, pos, d, g), allow_unused=True)
        for key, h, x in zip(("charges", "cell", "positions", "dist", "g"), H, (q, cell, pos, d, g)):
            ref = s[f"{nm}/H_{key}"]
            got = np.zeros_like(ref) if h is None else h.cpu().numpy()
            assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max() + 1e-12, (nm, meta, key)


@pytest.mark.parametrize("nm", ["c03", "c06", "c11", "c16", "c22", "c25"])
def test_cell_cell_block_against_differences_of_the_fused_gradients(golden_dir, nm):
    """The block the reference cannot form (NaN): H = d<w, dS/dcell>/dz.  By the symmetry of mixed partials H_z is the
    directional derivative of dS/dz along w in the cell, formed here by central differences of the FUSED first-order kernels
    (default path, a second witness) -- P3M and PME, 1/r, 1/r^4 and 1/r^6, a slab, an exclusion radius, three channels."""
    z = np.load(f"{golden_dir}/ref_small.npz")
    meta, calc, q, cell, pos, pairs, d, g, per = _case(z, nm)
    w = torch.tensor(np.random.default_rng(7).normal(size=(3, 3)), device=DEV)
    V = calc(q, cell, pos, pairs, d, periodic=per)
    (Gc,) = torch.autograd.grad((V * g).sum(), cell, create_graph=True)
    H = torch.autograd.grad((w * Gc).sum(), (q, cell, pos, d, g), allow_unused=True)  # (dS/dcell does not depend on d)
    H = [torch.zeros_like(x) if h is None else h for h, x in zip(H, (q, cell, pos, d, g))]
    assert all(bool(torch.isfinite(h).all()) for h in H)
    fused = make_calc(meta)
    eps = 1e-5

    def grads(c):
        xs = [x.detach().clone().requires_grad_(True) for x in (q, c, pos, d, g)]
        Vf = fused(xs[0], xs[1], xs[2], pairs, xs[3], periodic=per)
        return torch.autograd.grad((Vf * xs[4]).sum(), xs)

    plus, minus = grads(cell.detach() + eps * w), grads(cell.detach() - eps * w)
    for key, h, a, b in zip(("charges", "cell", "positions", "dist", "g"), H, plus, minus):
        fd = (a - b) / (2 * eps)
        assert float((h - fd).abs().max()) <= 2e-7 * float(fd.abs().max()) + 1e-9, (nm, meta, key)


# ---- the primitives on their own ----------------------------------------------------------------------------------------


def _geom(scheme, order, ns=(6, 5, 8)):
    cell = np.array([[4, 0, 0], [0.5, 5, 0], [0.3, -0.4, 6]], dtype=np.float64)
    return ops.MeshGeometry(cell, ns, scheme, order)


@pytest.mark.parametrize("scheme,order", [(tpa._lib.P3M, 3), (tpa._lib.P3M, 5), (tpa._lib.LAGRANGE, 4), (tpa._lib.LAGRANGE, 7)])
def test_mesh_primitives_gradcheck(scheme, order):
    """spread / gather / convolve / spectral_dot: first and second derivatives against finite differences (float64), with the
    fractional coordinates kept away from the points where the piecewise polynomials change pieces."""
    rng = np.random.default_rng(order)
    geom = _geom(scheme, order)
    n_atoms, n_ch = 5, 2
    frac = rng.uniform(0.15, 0.35, (n_atoms, 3)) + rng.integers(-3, 9, (n_atoms, 3))  # pieces change at k / 2
    u = torch.tensor(frac, device=DEV, requires_grad=True)
    x = torch.tensor(rng.normal(size=(n_atoms, n_ch)), device=DEV, requires_grad=True)
    phi = torch.tensor(rng.normal(size=(n_ch, *geom.ns)), device=DEV, requires_grad=True)
    # the table of the convolution stands for an even function of k (G(|k|^2), U^2(k)): any such function will do
    nx, ny, nz = geom.ns
    k2 = (np.fft.fftfreq(nx)[:, None, None] ** 2 + np.fft.fftfreq(ny)[None, :, None] ** 2 + np.fft.rfftfreq(nz)[None, None, :] ** 2)
    G = torch.tensor(np.exp(-3.0 * k2) * (1 + 0.5 * np.cos(7 * k2)), device=DEV)
    zero = (0, 0, 0)
    kw = dict(eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=1e-10)
    assert torch.autograd.gradcheck(lambda a, b: analytic._Spread.apply(a, b, geom, zero), (u, x), **kw)
    assert torch.autograd.gradcheck(lambda a, b: analytic._Gather.apply(a, b, geom, zero), (u, phi), **kw)
    assert torch.autograd.gradgradcheck(lambda a, b: analytic._Gather.apply(a, b, geom, zero), (u, phi), **kw)
    assert torch.autograd.gradgradcheck(lambda a, b: analytic._Spread.apply(a, b, geom, zero), (u, x), **kw)
    # the three first-derivative gathers of one launch against the three single ones, and their own derivatives
    g3 = analytic._GatherGrad.apply(u, phi, geom, zero)
    for d, k in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
        assert relmax(g3[:, :, d].detach().cpu(), analytic._Gather.apply(u, phi, geom, k).detach().cpu()) < 1e-13
    assert torch.autograd.gradcheck(lambda a, b: analytic._GatherGrad.apply(a, b, geom, zero), (u, phi), **kw)
    if order > 3:  # (third derivatives of the weights: quadratic pieces have none)
        assert torch.autograd.gradgradcheck(lambda a, b: analytic._GatherGrad.apply(a, b, geom, zero), (u, phi), **kw)
    lin = dict(eps=1e-3, atol=1e-6, rtol=1e-6, nondet_tol=1e-10)  # a linear map: no truncation error, less round-off
    assert torch.autograd.gradcheck(lambda a: analytic._Convolve.apply(a, G, geom), (phi,), **lin)
    assert torch.autograd.gradgradcheck(lambda a: analytic._Convolve.apply(a, G, geom), (phi,), **lin)


def test_convolution_table_gradient():
    """d convolve / dG = spectral_dot and the derivative of THAT w.r.t. the mesh, against torch.fft (test side only)."""
    rng = np.random.default_rng(5)
    geom = _geom(tpa._lib.P3M, 3, ns=(4, 6, 8))
    nx, ny, nz = geom.ns
    k2 = (np.fft.fftfreq(nx)[:, None, None] ** 2 + np.fft.fftfreq(ny)[None, :, None] ** 2 + np.fft.rfftfreq(nz)[None, None, :] ** 2)
    G = torch.tensor(np.exp(-3.0 * k2), device=DEV, requires_grad=True)
    a = torch.tensor(rng.normal(size=(2, nx, ny, nz)), device=DEV, requires_grad=True)
    b = torch.tensor(rng.normal(size=(2, nx, ny, nz)), device=DEV)
    w = torch.tensor(rng.normal(size=tuple(G.shape)), device=DEV)

    def ours(a_, G_):
        return (analytic._Convolve.apply(a_, G_, geom) * b).sum()

    def theirs(a_, G_):
        return (torch.fft.irfftn(torch.fft.rfftn(a_, dim=(1, 2, 3)) * G_, s=(nx, ny, nz), dim=(1, 2, 3), norm="forward") * b).sum()

    ga, gG = torch.autograd.grad(ours(a, G), (a, G), create_graph=True)
    ra, rG = torch.autograd.grad(theirs(a, G), (a, G), create_graph=True)
    assert relmax(ga.detach().cpu(), ra.detach().cpu()) < 1e-12
    assert relmax(gG.detach().cpu(), rG.detach().cpu()) < 1e-12
    (h_ours,) = torch.autograd.grad((gG * w).sum(), a)
    (h_theirs,) = torch.autograd.grad((rG * w).sum(), a)
    assert relmax(h_ours.cpu(), h_theirs.cpu()) < 1e-12


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_pair_primitives_gradcheck(mode):
    rng = np.random.default_rng(mode)
    n_atoms, n_pairs, n_ch = 6, 11, 2
    pairs = torch.tensor(rng.integers(0, n_atoms, (n_pairs, 2)), device=DEV)
    w = torch.tensor(rng.normal(size=n_pairs), device=DEV, requires_grad=True)
    x = torch.tensor(rng.normal(size=(n_atoms, n_ch)), device=DEV, requires_grad=True)
    y = torch.tensor(rng.normal(size=(n_atoms, n_ch)), device=DEV, requires_grad=True)
    kw = dict(eps=1e-6, atol=1e-8, rtol=1e-7, nondet_tol=1e-10)
    assert torch.autograd.gradcheck(lambda a, b: analytic._PairSum.apply(a, b, pairs, mode), (w, x), **kw)
    assert torch.autograd.gradgradcheck(lambda a, b: analytic._PairSum.apply(a, b, pairs, mode), (w, x), **kw)
    if mode != 2:
        assert torch.autograd.gradcheck(lambda a, b: analytic._PairDot.apply(a, b, pairs, mode == 0), (x, y), **kw)
        assert torch.autograd.gradgradcheck(lambda a, b: analytic._PairDot.apply(a, b, pairs, mode == 0), (x, y), **kw)
    # and against index_add_ (the reference's formulation)
    i, j = pairs[:, 0], pairs[:, 1]
    ref = torch.zeros_like(x)
    if mode != 2:
        ref.index_add_(0, i, x.detach()[j] * w.detach()[:, None])
    if mode != 1:
        ref.index_add_(0, j, x.detach()[i] * w.detach()[:, None])
    assert relmax(analytic._PairSum.apply(w, x, pairs, mode).detach().cpu(), ref.cpu()) < 1e-14


@pytest.mark.parametrize("use_rows", [False, True])
def test_pair_difference_and_scatter(use_rows):
    """``x_j - x_i`` and its adjoint (the distance helper's two index operations) against ATen's indexing, with their first and
    second derivatives; rows of the transposed list and atomics on the list."""
    rng = np.random.default_rng(8)
    n_atoms, n_pairs = 40, 300
    pairs = torch.tensor(rng.integers(0, n_atoms - 3, (n_pairs, 2)), device=DEV)
    rows = None
    if use_rows:
        topo = ops.get_topology(pairs, n_atoms)
        rows = (topo.row_ptr, topo.entries)
    x = torch.tensor(rng.normal(size=(n_atoms, 3)), device=DEV, requires_grad=True)
    v = torch.tensor(rng.normal(size=(n_pairs, 3)), device=DEV, requires_grad=True)
    i, j = pairs[:, 0], pairs[:, 1]
    assert relmax(analytic._PairDiff.apply(x, pairs, rows).detach().cpu(), (x[j] - x[i]).detach().cpu()) < 1e-15
    ref = torch.zeros_like(x).index_add(0, j, v.detach()).index_add(0, i, -v.detach())
    assert relmax(analytic._PairScatter.apply(v, pairs, rows, n_atoms).detach().cpu(), ref.cpu()) < 1e-13
    kw = dict(eps=1e-3, atol=1e-8, rtol=1e-7, nondet_tol=1e-10)  # linear maps
    assert torch.autograd.gradcheck(lambda a: analytic._PairDiff.apply(a, pairs, rows), (x,), **kw)
    assert torch.autograd.gradgradcheck(lambda a: analytic._PairDiff.apply(a, pairs, rows), (x,), **kw)
    assert torch.autograd.gradcheck(lambda a: analytic._PairScatter.apply(a, pairs, rows, n_atoms), (v,), **kw)
    assert torch.autograd.gradgradcheck(lambda a: analytic._PairScatter.apply(a, pairs, rows, n_atoms), (v,), **kw)


@pytest.mark.parametrize("route", ["front", "python_nodes"])
def test_distance_helper_twice_differentiated(route, monkeypatch):
    """``pair_distances`` under create_graph=True (both host paths): gradient and Hessian-vector product of sum(d^2 w) against
    the same expression in plain tensor operations (``tests/helpers.py:278-304``), triclinic cell, w.r.t. positions and cell."""
    if route == "python_nodes":
        monkeypatch.setattr(ops, "FRONT", False)
    rng = np.random.default_rng(9)
    n = 60
    pos_np, cell_np = rng.uniform(0, 7, (n, 3)), 7 * np.eye(3) + rng.uniform(-0.4, 0.4, (3, 3))
    pairs_np, S_np, _ = tpa.neighbor_list(pos_np, cell_np, 3.5)
    pairs, S = torch.tensor(pairs_np, device=DEV), torch.tensor(S_np, device=DEV, dtype=torch.float64)
    w = torch.tensor(rng.normal(size=len(pairs_np)), device=DEV)
    c_pos, c_cell = torch.tensor(rng.normal(size=(n, 3)), device=DEV), torch.tensor(rng.normal(size=(3, 3)), device=DEV)
    res = []
    for ours in (True, False):
        pos = torch.tensor(pos_np, device=DEV, requires_grad=True)
        cell = torch.tensor(cell_np, device=DEV, requires_grad=True)
        if ours:
            d = tpa.pair_distances(pos, pairs, cell, S)
        else:
            d = torch.linalg.norm(pos[pairs[:, 1]] - pos[pairs[:, 0]] + S @ cell, dim=1)
        gp, gc = torch.autograd.grad((w * d * d * d).sum(), (pos, cell), create_graph=True)
        hp, hc = torch.autograd.grad((gp * c_pos).sum() + (gc * c_cell).sum(), (pos, cell))
        res.append([gp.detach().cpu(), gc.detach().cpu(), hp.cpu(), hc.cpu()])
    for a, b in zip(*res):
        assert relmax(a, b) < 1e-12


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_pair_sum_through_the_transposed_list(dtype, monkeypatch):
    """The owner-computes rows (lists of ROWS_MIN_PAIRS pairs and more) against the atomic kernel and ``index_add_``, all three
    modes, two channels, int64 and int32 indices, atoms without any pair."""
    rng = np.random.default_rng(3)
    n_atoms, n_pairs, n_ch = 700, 9000, 2
    ij = rng.integers(0, n_atoms - 20, (n_pairs, 2))  # the last twenty atoms have empty rows
    w = torch.tensor(rng.normal(size=n_pairs), device=DEV, dtype=dtype)
    x = torch.tensor(rng.normal(size=(n_atoms, n_ch)), device=DEV, dtype=dtype)
    tol = 1e-13 if dtype == torch.float64 else 2e-5
    for idx_dtype in (torch.int64, torch.int32):
        pairs = torch.tensor(ij, device=DEV, dtype=idx_dtype)
        i, j = pairs[:, 0].long(), pairs[:, 1].long()
        for mode in (0, 1, 2):
            ref = torch.zeros_like(x)
            if mode != 2:
                ref.index_add_(0, i, x[j] * w[:, None])
            if mode != 1:
                ref.index_add_(0, j, x[i] * w[:, None])
            monkeypatch.setattr(analytic, "ROWS", True)
            rows = analytic._PairSum.apply(w, x, pairs, mode)
            monkeypatch.setattr(analytic, "ROWS", False)
            atom = analytic._PairSum.apply(w, x, pairs, mode)
            assert relmax(rows.cpu(), ref.cpu()) < tol and relmax(atom.cpu(), ref.cpu()) < tol


# ---- a training-shaped use: loss on forces, learned charges -----------------------------------------------------------------


@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-9), ("f32", torch.float32, 5e-4)])
def test_force_loss_with_learned_charges_against_the_torch_oracle(tag, dtype, tol):
    """512 atoms, real neighbour list, distances through ``pair_distances`` (exact second order by itself): charges = f(theta),
    F = -dE/dr with create_graph, loss = sum (F - F0)^2; d loss / d theta and d loss / d cell against the PyTorch-CPU oracle's
    own double backward in float64 (oracle/pme_torch.py restates the reference's ATen chain)."""
    from oracle import pme_numpy as O
    from oracle import pme_torch as OT

    rng = np.random.default_rng(11)
    n_side, a = 8, 2.2
    gr = (np.arange(n_side) + 0.5) * a
    pos = np.stack(np.meshgrid(gr, gr, gr, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (n_side**3, 3))
    cell = n_side * a * np.eye(3) + rng.uniform(-0.2, 0.2, (3, 3))
    feats = rng.normal(size=(len(pos), 3))
    theta0 = np.array([0.7, -0.4, 0.2])
    F0 = rng.normal(size=pos.shape) * 0.1
    pairs, S, _ = tpa.neighbor_list(pos, cell, 5.0)
    sm, h, order = 1.0, 0.7, 4

    def loss_of(theta, tcell, tpos, energy_fn):
        q = (feats_t(theta) @ theta).reshape(-1, 1)
        q = q - q.mean()
        E = energy_fn(q, tcell, tpos)
        (gpos,) = torch.autograd.grad(E, tpos, create_graph=True)
        return ((-gpos - F0_t(theta)) ** 2).sum()

    feats_t = lambda th: torch.tensor(feats, dtype=th.dtype, device=th.device)  # noqa: E731
    F0_t = lambda th: torch.tensor(F0, dtype=th.dtype, device=th.device)  # noqa: E731

    # oracle (CPU, float64)
    th, tc, tp = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (theta0, cell, pos))
    spec = O.PotentialSpec("coulomb", 1, sm, 1.0)
    o_pairs, o_S = torch.tensor(pairs), torch.tensor(S, dtype=torch.float64)

    def oracle_energy(q, c, p):
        d = OT.pair_distances(p, c, o_pairs, o_S)
        return (q * OT.forward(spec, "P3M", order, h, q, c, p, o_pairs, d)).sum()

    lo = loss_of(th, tc, tp, oracle_energy)
    o_th, o_c = torch.autograd.grad(lo, (th, tc))

    # this package (GPU)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=sm), mesh_spacing=h, interpolation_nodes=order)
    calc.double_backward = "analytic"
    th, tc, tp = (torch.tensor(x, dtype=dtype, device=DEV, requires_grad=True) for x in (theta0, cell, pos))
    g_pairs, g_S = torch.tensor(pairs, device=DEV), torch.tensor(S, dtype=dtype, device=DEV)

    def gpu_energy(q, c, p):
        d = tpa.pair_distances(p, g_pairs, c, g_S)
        return (q * calc(q, c, p, g_pairs, d)).sum()

    lg = loss_of(th, tc, tp, gpu_energy)
    g_th, g_c = torch.autograd.grad(lg, (th, tc))
    e_l = abs(float(lg.detach()) - float(lo.detach())) / abs(float(lo.detach()))
    e_th, e_c = relmax(g_th.double().cpu(), o_th), relmax(g_c.double().cpu(), o_c)
    print(f"force loss {tag}: loss {e_l:.2e} dtheta {e_th:.2e} dcell {e_c:.2e}")
    assert e_l < tol and e_th < tol and e_c < tol


def test_analytic_and_fused_paths_agree_at_first_order():
    """The same call through the fused kernels (default) and through the primitives: V and all gradients, P3M and PME."""
    rng = np.random.default_rng(2)
    n_side, a = 6, 2.3
    gr = (np.arange(n_side) + 0.5) * a
    pos = np.stack(np.meshgrid(gr, gr, gr, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (n_side**3, 3))
    cell = n_side * a * np.eye(3)
    q = rng.normal(size=(len(pos), 1))
    pairs, S, _ = tpa.neighbor_list(pos, cell, 4.6)
    for Calc, order in ((tpa.P3MCalculator, 5), (tpa.PMECalculator, 4)):
        res = []
        for mode in (None, "analytic"):
            calc = Calc(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.8, interpolation_nodes=order)
            calc.double_backward = mode
            tq, tc, tp = (torch.tensor(x, device=DEV, requires_grad=True) for x in (q, cell, pos))
            d = tpa.pair_distances(tp, torch.tensor(pairs, device=DEV), tc, torch.tensor(S, device=DEV, dtype=torch.float64))
            V = calc(tq, tc, tp, torch.tensor(pairs, device=DEV), d)
            (tq * V).sum().backward()
            res.append([V.detach().cpu(), tq.grad.cpu(), tc.grad.cpu(), tp.grad.cpu()])
        for x, y in zip(*res):
            assert relmax(y, x) < 1e-11


def test_masks_index_types_and_views_through_the_analytic_route():
    """pair_mask, int32 indices, non-contiguous charges / positions (views of wider tensors): same potentials and gradients as
    the fused kernels, which have their own parity tests for these inputs."""
    rng = np.random.default_rng(4)
    n_side, a = 5, 2.4
    gr = (np.arange(n_side) + 0.5) * a
    pos = np.stack(np.meshgrid(gr, gr, gr, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (n_side**3, 3))
    cell = n_side * a * np.eye(3) + rng.uniform(-0.3, 0.3, (3, 3))
    q = rng.normal(size=(len(pos), 2))
    pairs, S, _ = tpa.neighbor_list(pos, cell, 4.4)
    mask = rng.uniform(size=len(pairs)) > 0.2
    res = []
    for mode in (None, "analytic"):
        calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.1, exclusion_radius=2.0), mesh_spacing=0.9, interpolation_nodes=4)
        calc.double_backward = mode
        wide_q = torch.tensor(np.concatenate([q, q], axis=1), device=DEV, requires_grad=True)
        wide_p = torch.tensor(np.concatenate([pos, pos], axis=1), device=DEV, requires_grad=True)
        tq, tp = wide_q[:, 1:3], wide_p[:, 3:]
        tc = torch.tensor(cell, device=DEV, requires_grad=True)
        ti = torch.tensor(pairs, device=DEV, dtype=torch.int32)
        d = tpa.pair_distances(tp, ti, tc, torch.tensor(S, device=DEV, dtype=torch.float64))
        V = calc(tq, tc, tp, ti, d, pair_mask=torch.tensor(mask, device=DEV))
        (V * V).sum().backward()
        res.append([V.detach().cpu(), wide_q.grad.cpu(), tc.grad.cpu(), wide_p.grad.cpu()])
    for x, y in zip(*res):
        assert relmax(y, x) < 1e-10


@pytest.mark.parametrize("which", ["P3M", "PME", "Ewald"])
def test_auto_mode_fused_first_order_exact_higher_orders(which):
    """``double_backward = "auto"``: without create_graph the values and gradients are the fused kernels' (bit for bit the plain
    calculator's through the same unfused-distance route); with create_graph the force loss's gradients w.r.t. charges,
    positions and cell are those of the analytic route."""
    rng = np.random.default_rng(12)
    n_side, a = 5, 2.3
    gr = (np.arange(n_side) + 0.5) * a
    pos = np.stack(np.meshgrid(gr, gr, gr, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (n_side**3, 3))
    cell = n_side * a * np.eye(3) + rng.uniform(-0.2, 0.2, (3, 3))
    q = rng.normal(size=(len(pos), 1))
    pairs_np, S_np, _ = tpa.neighbor_list(pos, cell, 4.5)
    pairs, S = torch.tensor(pairs_np, device=DEV), torch.tensor(S_np, device=DEV, dtype=torch.float64)
    w = torch.tensor(rng.normal(size=pos.shape), device=DEV)

    def make(mode):
        pot = tpa.CoulombPotential(smearing=1.0)
        calc = (tpa.P3MCalculator(pot, mesh_spacing=0.7, interpolation_nodes=4) if which == "P3M" else
                tpa.PMECalculator(pot, mesh_spacing=0.7, interpolation_nodes=5) if which == "PME" else
                tpa.EwaldCalculator(pot, lr_wavelength=1.4))
        calc.double_backward = mode
        return calc

    def run(mode, create_graph):
        calc = make(mode)
        tq, tc, tp = (torch.tensor(x, device=DEV, requires_grad=True) for x in (q, cell, pos))
        d = tpa.pair_distances(tp, pairs, tc, S)
        E = (tq * calc(tq, tc, tp, pairs, d)).sum()
        if not create_graph:
            return [E.detach().cpu()] + [x.cpu() for x in torch.autograd.grad(E, (tq, tc, tp))]
        (gp,) = torch.autograd.grad(E, tp, create_graph=True)
        return [E.detach().cpu()] + [x.cpu() for x in torch.autograd.grad((w * gp).sum(), (tq, tc, tp))]

    for a_, b_ in zip(run("auto", False), run(None, False)):
        assert relmax(a_, b_) < 1e-12
    for a_, b_ in zip(run("auto", True), run("analytic", True)):
        assert relmax(a_, b_) < 1e-10


def test_auto_mode_with_inputs_that_depend_on_each_other():
    """Positions made from fractional coordinates and the cell (a stress-of-a-force-loss set-up), distances from both: the
    recorded backward of the auto mode must hand out PARTIAL derivatives (the caller's graph adds the chains)."""
    rng = np.random.default_rng(13)
    n = 64
    frac_np = rng.uniform(0, 1, (n, 3))
    cell_np = 9.0 * np.eye(3) + rng.uniform(-0.3, 0.3, (3, 3))
    q = torch.tensor(rng.normal(size=(n, 1)), device=DEV)
    pairs_np, S_np, _ = tpa.neighbor_list(frac_np @ cell_np, cell_np, 4.0)
    pairs, S = torch.tensor(pairs_np, device=DEV), torch.tensor(S_np, device=DEV, dtype=torch.float64)
    w = torch.tensor(rng.normal(size=(n, 3)), device=DEV)
    res = []
    for mode in ("auto", "analytic"):
        calc = tpa.PMECalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.8, interpolation_nodes=4)
        calc.double_backward = mode
        frac = torch.tensor(frac_np, device=DEV, requires_grad=True)
        cell = torch.tensor(cell_np, device=DEV, requires_grad=True)
        pos = frac @ cell
        d = tpa.pair_distances(pos, pairs, cell, S)
        E = (q * calc(q, cell, pos, pairs, d)).sum()
        gf, gc = torch.autograd.grad(E, (frac, cell), create_graph=True)
        hf, hc = torch.autograd.grad((w * gf).sum() + gc.sum(), (frac, cell))
        res.append([gf.detach().cpu(), gc.detach().cpu(), hf.cpu(), hc.cpu()])
    for a_, b_ in zip(*res):
        assert relmax(a_, b_) < 1e-10


def test_training_step_replays_as_a_hip_graph():
    """The whole force-loss step of the analytic route -- distances, potentials, forces with create_graph, the gradient of the
    loss -- captured with ``torch.cuda.graph`` and replayed with new parameter values: the route makes no host round trip and
    no device copy of host data per call (the eager step is ~400 launches, bound by the host)."""
    rng = np.random.default_rng(14)
    n_side, a = 6, 2.5
    gr = (np.arange(n_side) + 0.5) * a
    pos_np = np.stack(np.meshgrid(gr, gr, gr, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.2, 0.2, (n_side**3, 3))
    cell_np = np.eye(3) * n_side * a
    pairs_np, S_np, _ = tpa.neighbor_list(pos_np, cell_np, 6.5)  # > 4096 pairs: the rows of the transposed list
    assert len(pairs_np) > analytic.ROWS_MIN_PAIRS
    t = lambda x: torch.tensor(x, device=DEV, dtype=torch.float64)  # noqa: E731
    pos, cell, S, pairs = t(pos_np).requires_grad_(True), t(cell_np), t(S_np), torch.tensor(pairs_np, device=DEV)
    q0 = t(rng.normal(size=(len(pos_np), 1)))
    theta = torch.ones((), device=DEV, dtype=torch.float64, requires_grad=True)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.8, interpolation_nodes=4)
    calc.double_backward = "analytic"

    def body():
        q = q0 * theta
        d = tpa.pair_distances(pos, pairs, cell, S)
        (g,) = torch.autograd.grad((q * calc(q, cell, pos, pairs, d)).sum(), pos, create_graph=True)
        loss = (g * g).sum()
        return torch.stack([loss.detach(), torch.autograd.grad(loss, theta)[0]])

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res = body()
    for value in (1.0, 0.7):
        with torch.no_grad():
            theta.fill_(value)
        graph.replay()
        got = res.clone()
        want = body()
        assert relmax(got.cpu(), want.detach().cpu()) < 1e-12
        assert abs(float(got[1]) - 4.0 * float(got[0]) / value) < 1e-9 * abs(float(got[1]))  # loss is quartic in theta


def test_example_fits_charges_to_forces():
    """examples/fit_charges_to_forces.py: L-BFGS on a force loss (second derivatives through the calculator in "auto" mode)
    recovers the hidden charge difference of two species from P3M forces."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import fit_charges_to_forces

    info = fit_charges_to_forces.run(n_side=6, steps=30)
    assert info["last_loss"] < 1e-12 * info["first_loss"], info
    assert abs(info["charge_difference"] - info["charge_difference_true"]) < 1e-6, info
    assert info["relative_force_residual"] < 1e-12, info


def test_unsupported_options_say_so():
    calc = tpa.EwaldCalculator(tpa.CoulombPotential(smearing=1.0), lr_wavelength=1.5)
    t = lambda x: torch.tensor(x, device=DEV, dtype=torch.float64)  # noqa: E731
    pos = t(np.random.default_rng(0).uniform(0, 4, (5, 3))).requires_grad_(True)
    pairs = torch.tensor([[0, 1], [1, 2], [2, 3], [3, 4]], device=DEV)
    calc.double_backward = "exact"
    with pytest.raises(ValueError, match="'auto', 'analytic' or 'finite-difference'"):
        calc(t(np.ones((5, 1))), t(4 * np.eye(3)), pos, pairs, t(np.ones(4)))
    mesh = tpa.PMECalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=1.0)
    mesh.double_backward = "analytic"
    with pytest.raises(NotImplementedError, match="Batching not implemented for mesh-based calculators"):
        mesh(t(np.ones((5, 1))), t(4 * np.eye(3)), pos, pairs, t(np.ones(4)), node_mask=torch.ones(5, device=DEV, dtype=torch.bool))

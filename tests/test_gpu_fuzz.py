"""Seeded random sweep of the hot path against the oracle: calculator type and interpolation order, potential (1/r^p,
p = 1..6, with and without an exclusion radius), dtype, triclinic cells whose meshes differ per axis (brick kernels and
atomic kernels, fused and 3-D convolution), atom counts that are not multiples of any block size, atoms outside the
cell, half / full lists, pair masks, eager / deferred distances, energy-mode and general upstream gradients.  Every case
checks potentials and the gradients w.r.t. positions, cell and charges."""
import numpy as np
import pytest
import torch

import torchpme_amd as tpa
from oracle import pme_numpy as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rell2(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / (np.linalg.norm(np.asarray(b)) + 1e-300))


@pytest.mark.parametrize("seed", range(96))
def test_random_configuration(seed):
    rng = np.random.default_rng(1000 + seed)
    dtype = torch.float64 if rng.uniform() < 0.6 else torch.float32
    scheme = "P3M" if rng.uniform() < 0.5 else "Lagrange"
    order = int(rng.integers(1, 6)) if scheme == "P3M" else int(rng.integers(3, 8))
    lengths = rng.uniform(5.0, 13.0, 3)
    cell = np.diag(lengths) + np.tril(rng.uniform(-0.15, 0.15, (3, 3)) * lengths.min(), -1)
    h = float(rng.uniform(0.4, 1.1))
    N = int(rng.integers(1, 260))
    pos = rng.uniform(-0.3, 1.3, (N, 3)) @ cell
    n_ch = int(rng.choice([1, 1, 1, 2, 3]))  # several charge channels take the unfused kernels
    q = rng.normal(size=(N, n_ch))
    full = bool(rng.uniform() < 0.4)
    cutoff = float(rng.uniform(2.0, 4.5))
    slab = bool(rng.uniform() < 0.15)  # 2-D periodic: one non-periodic axis (Coulomb only, potentials/coulomb.py:6-40)
    periodic = [True, True, True]
    if slab:
        periodic[int(rng.integers(0, 3))] = False
    pairs, S, dist = tpa.neighbor_list(pos, cell, cutoff, full_list=full, periodic=tuple(periodic))
    if len(pairs) and dist.min() < 0.5:  # keep 1/r^6 within what fp32 can compare
        keep = dist > 0.5
        pairs, S, dist = pairs[keep], S[keep], dist[keep]
    p = 1 if slab else int(rng.integers(1, 7))
    sm = float(rng.uniform(0.8, 1.5))
    excl = float(rng.uniform(1.0, 2.0)) if rng.uniform() < 0.25 else None
    if p == 1 and rng.uniform() < 0.5:
        spec = O.PotentialSpec("coulomb", 1, sm, 1.0, exclusion_radius=excl, exclusion_degree=2 if excl else 1)
        pot = tpa.CoulombPotential(smearing=sm, exclusion_radius=excl, exclusion_degree=2 if excl else 1)
    else:
        spec = O.PotentialSpec("ipl", p, sm, 1.0, exclusion_radius=excl, exclusion_degree=2 if excl else 1)
        pot = tpa.InversePowerLawPotential(exponent=p, smearing=sm, exclusion_radius=excl, exclusion_degree=2 if excl else 1)
    mask = (rng.uniform(size=len(pairs)) > 0.2) if rng.uniform() < 0.25 else None
    energy_mode = bool(rng.uniform() < 0.5)
    deferred = bool(rng.uniform() < 0.5)
    need_q = not energy_mode or bool(rng.uniform() < 0.5)
    gE = -1.3
    g = gE * q if energy_mode else rng.normal(size=(N, n_ch))

    Vo, cache = O.forward(spec, scheme, order, h, q, cell, pos, pairs, dist, full_list=full, pair_mask=mask,
                          periodic=tuple(periodic) if slab else None, return_cache=True)
    gr = O.backward(cache, g)
    gpos_d, gcell_d = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])

    Calc = tpa.P3MCalculator if scheme == "P3M" else tpa.PMECalculator
    calc = Calc(pot, mesh_spacing=h, interpolation_nodes=order, full_neighbor_list=full).to(dtype)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=dtype, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q, need_q), t(cell, True), t(pos, True)
    ti = torch.tensor(pairs.reshape(-1, 2), device=DEV)
    d = tpa.pair_distances(tp, ti, tc, t(S.reshape(-1, 3)), deferred=deferred)
    V = calc(tq, tc, tp, ti, d, pair_mask=None if mask is None else torch.tensor(mask, device=DEV),
             periodic=torch.tensor(periodic, device=DEV) if slab else None)
    if energy_mode:
        (gE * tpa.weighted_sum(V, tq)).backward()
    else:
        (V * t(g)).sum().backward()
    f64 = dtype == torch.float64
    # fp32: 1/r^p sums of either sign cancel; compare against the magnitude of the terms rather than of the sum
    tolV, tolG = (1e-10, 1e-9) if f64 else (3e-4, 2e-3)
    info = f"seed {seed}: {scheme}{order} p={p} excl={excl} N={N} P={len(pairs)} full={full} mask={mask is not None} " \
           f"energy={energy_mode} deferred={deferred} channels={n_ch} slab={slab} {dtype}"
    assert rell2(V.detach().cpu().numpy(), Vo) < tolV, info
    assert rell2(tp.grad.cpu().numpy(), gr["positions"] + gpos_d) < tolG, info
    # (the 9 cell-gradient components are sums over all pairs / mesh points with heavy cancellation: looser in fp32)
    assert rell2(tc.grad.cpu().numpy(), gr["cell"] + gcell_d) < (tolG if f64 else 1e-2), info
    if need_q:
        # energy mode: L = gE sum_a q_a V_a(q) depends on q directly (gE V) and through V (the oracle's adjoint for g = gE q)
        expect_q = gr["charges"] + (gE * Vo if energy_mode else 0.0)
        assert rell2(tq.grad.cpu().numpy(), expect_q) < tolG, info
    if len(pairs):
        assert rell2(d.detach().cpu().numpy(), dist) < (1e-13 if f64 else 1e-5), info


@pytest.mark.parametrize("seed", range(24))
def test_random_graphed_step(seed):
    """The same kind of random configuration through ``GraphedEnergyForces`` (HIP-graph replay: deferred distances,
    co-scheduled pair sum where the mesh allows it, direct energy gradient), optionally with the cell gradient, against the
    oracle -- for the captured positions and after an in-place position update."""
    rng = np.random.default_rng(5000 + seed)
    dtype = torch.float64 if rng.uniform() < 0.6 else torch.float32
    scheme = "P3M" if rng.uniform() < 0.5 else "Lagrange"
    order = int(rng.integers(2, 6)) if scheme == "P3M" else int(rng.integers(3, 8))
    lengths = rng.uniform(6.0, 12.0, 3)
    cell = np.diag(lengths) + np.tril(rng.uniform(-0.1, 0.1, (3, 3)) * lengths.min(), -1)
    h = float(rng.uniform(0.35, 0.9))
    n_side = int(rng.integers(2, int(lengths.min() / 1.3) + 1))  # jittered lattice: no overlapping atoms
    N = int(rng.integers(max(2, n_side**3 // 2), n_side**3 + 1))
    full = bool(rng.uniform() < 0.4)
    p = 6 if rng.uniform() < 0.3 else 1
    sm = float(rng.uniform(0.8, 1.4))
    with_cell = bool(rng.uniform() < 0.3)
    spec = O.PotentialSpec("coulomb" if p == 1 else "ipl", p, sm, 1.0)
    pot = tpa.CoulombPotential(smearing=sm) if p == 1 else tpa.InversePowerLawPotential(exponent=6, smearing=sm)
    Calc = tpa.P3MCalculator if scheme == "P3M" else tpa.PMECalculator
    calc = Calc(pot, mesh_spacing=h, interpolation_nodes=order, full_neighbor_list=full).to(dtype)
    sites = rng.permutation(n_side**3)[:N]
    grid = np.stack(np.unravel_index(sites, (n_side,) * 3), -1)
    frac = (grid + 0.5 + rng.uniform(-0.2, 0.2, (N, 3))) / n_side
    q = rng.normal(size=(N, 1))
    t = lambda a: torch.tensor(a, device=DEV, dtype=dtype)  # noqa: E731
    step = None
    for k in range(2):
        pos = (frac + (0.0 if k == 0 else 0.004) * rng.normal(size=(N, 3))) @ cell
        # one list for both position sets (a Verlet list): build it for the first set with a margin
        if k == 0:
            pairs, S, _ = tpa.neighbor_list(pos, cell, 3.6, full_list=full)
            ti, tS = torch.tensor(pairs.reshape(-1, 2), device=DEV), t(S.reshape(-1, 3))
            step = tpa.GraphedEnergyForces(calc, t(q), t(cell), t(pos), ti, tS, cell_gradient=with_cell,
                                           store_distances=seed % 2 == 0)
        dist, _ = O.pair_distances(pos, cell, pairs, S)
        if len(pairs) and dist.min() < 0.6:
            pytest.skip("random configuration with overlapping atoms")
        Vo, cache = O.forward(spec, scheme, order, h, q, cell, pos, pairs, dist, full_list=full, return_cache=True)
        gr = O.backward(cache, q)
        gpos_d, gcell_d = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
        out = step(t(pos))
        f64 = dtype == torch.float64
        info = f"seed {seed}/{k}: {scheme}{order} p={p} N={N} P={len(pairs)} full={full} cell={with_cell} {dtype}"
        Eo = float((q * Vo).sum())
        scale = float(np.abs(q * Vo).sum())
        assert abs(out[0].item() - Eo) < (1e-10 if f64 else 2e-5) * scale, info + f" energy {out[0].item()} vs {Eo} scale {scale}"
        assert rell2(out[1].cpu().numpy(), -(gr["positions"] + gpos_d)) < (1e-9 if f64 else 2e-3), info
        if with_cell:
            assert rell2(out[2].cpu().numpy(), gr["cell"] + gcell_d) < (1e-9 if f64 else 1e-2), info
        if len(pairs) and step.store_distances:
            assert rell2(step.distances.cpu().numpy(), dist) < (1e-13 if f64 else 1e-5), info


@pytest.mark.parametrize("channels", [1, 2])
def test_empty_pair_list_gradients(channels):
    """Found by an extended run of the sweep above (seed 442): three atoms further apart than the cutoff -- an empty pair
    list -- with gradients requested through the unfused kernels (several channels / general upstream gradient): the
    distance op has no gradient buffer to hand to its adjoint kernel and must return zeros."""
    rng = np.random.default_rng(442)
    cell = np.diag([6.2, 9.1, 9.2])
    pos = np.array([[0.5, 0.5, 0.5], [3.5, 5.0, 5.0], [1.0, 8.0, 2.0]])
    q = rng.normal(size=(3, channels))
    pairs, S, dist = tpa.neighbor_list(pos, cell, 2.0)
    assert len(pairs) == 0
    spec = O.PotentialSpec("coulomb", 1, 1.0, 1.0)
    g = rng.normal(size=(3, channels))
    Vo, cache = O.forward(spec, "P3M", 4, 1.0, q, cell, pos, pairs, dist, return_cache=True)
    gr = O.backward(cache, g)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=1.0, interpolation_nodes=4)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q, True), t(cell, True), t(pos, True)
    ti = torch.zeros((0, 2), dtype=torch.int64, device=DEV)
    d = tpa.pair_distances(tp, ti, tc, torch.zeros((0, 3), dtype=torch.float64, device=DEV))
    V = calc(tq, tc, tp, ti, d)
    (V * t(g)).sum().backward()
    assert rell2(V.detach().cpu().numpy(), Vo) < 1e-11
    assert rell2(tp.grad.cpu().numpy(), gr["positions"]) < 1e-9
    assert rell2(tc.grad.cpu().numpy(), gr["cell"]) < 1e-9
    assert rell2(tq.grad.cpu().numpy(), gr["charges"]) < 1e-10


@pytest.mark.parametrize("seed", range(32))
def test_random_ewald(seed):
    """Random small systems through ``EwaldCalculator`` (explicit reciprocal-space sum) against the oracle's forward, and its
    position gradient against central differences of the oracle's energy: 1/r^p for p = 1..6, triclinic cells, 1-3 channels,
    half / full lists, pair masks, 2-D slabs, fp64."""
    rng = np.random.default_rng(9000 + seed)
    lengths = rng.uniform(4.0, 9.0, 3)
    cell = np.diag(lengths) + np.tril(rng.uniform(-0.15, 0.15, (3, 3)) * lengths.min(), -1)
    N = int(rng.integers(1, 40))
    pos = rng.uniform(-0.2, 1.2, (N, 3)) @ cell
    n_ch = int(rng.choice([1, 1, 2, 3]))
    q = rng.normal(size=(N, n_ch))
    full = bool(rng.uniform() < 0.4)
    slab = bool(rng.uniform() < 0.2)
    periodic = [True, True, True]
    if slab:
        periodic[int(rng.integers(0, 3))] = False
    pairs, S, dist = tpa.neighbor_list(pos, cell, float(rng.uniform(2.0, 3.9)), full_list=full, periodic=tuple(periodic))
    if len(pairs) and dist.min() < 0.5:
        keep = dist > 0.5
        pairs, S, dist = pairs[keep], S[keep], dist[keep]
    p = 1 if slab else int(rng.integers(1, 7))
    sm = float(rng.uniform(0.7, 1.3))
    lr = float(rng.uniform(1.2, 2.5))
    spec = O.PotentialSpec("coulomb" if p == 1 else "ipl", p, sm, 1.0)
    pot = tpa.CoulombPotential(smearing=sm) if p == 1 else tpa.InversePowerLawPotential(exponent=p, smearing=sm)
    mask = (rng.uniform(size=len(pairs)) > 0.2) if rng.uniform() < 0.25 else None
    per = tuple(periodic) if slab else None
    Vo = O.ewald_forward(spec, lr, q, cell, pos, pairs, dist, full_list=full, periodic=per, pair_mask=mask)
    calc = tpa.EwaldCalculator(pot, lr_wavelength=lr, full_neighbor_list=full)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q), t(cell), t(pos, True)
    ti = torch.tensor(pairs.reshape(-1, 2), device=DEV)
    d = tpa.pair_distances(tp, ti, tc, t(S.reshape(-1, 3)))
    V = calc(tq, tc, tp, ti, d, pair_mask=None if mask is None else torch.tensor(mask, device=DEV),
             periodic=torch.tensor(periodic, device=DEV) if slab else None)
    info = f"seed {seed}: p={p} N={N} P={len(pairs)} full={full} mask={mask is not None} channels={n_ch} slab={slab}"
    assert rell2(V.detach().cpu().numpy(), Vo) < 1e-10, info
    w = rng.normal(size=(N, n_ch))
    (V * t(w)).sum().backward()
    # one component of the gradient by central differences of the oracle (distances recomputed for the displaced atom)
    a, c, h = int(rng.integers(0, N)), int(rng.integers(0, 3)), 1e-5
    vals = []
    for sgn in (+1, -1):
        pp = pos.copy()
        pp[a, c] += sgn * h
        dd, _ = O.pair_distances(pp, cell, pairs, S)
        vals.append(float((O.ewald_forward(spec, lr, q, cell, pp, pairs, dd, full_list=full, periodic=per, pair_mask=mask) * w).sum()))
    fd = (vals[0] - vals[1]) / (2 * h)
    got = float(tp.grad[a, c])
    assert abs(got - fd) < 1e-6 * (abs(fd) + float(np.abs(w * Vo).sum())), info + f" dL/dr {got} vs {fd}"


@pytest.mark.parametrize("seed", range(12))
def test_random_frame_batch(seed):
    """Random batches of independent frames (2-6 frames, 10-250 atoms each, different triclinic cells that give the same
    mesh, half / full lists, 1/r or 1/r^6, fp64 / fp32) through ``GraphedFrameBatch`` (one launch per kernel for all frames)
    against the eager calculator frame by frame: energies and forces."""
    rng = np.random.default_rng(7000 + seed)
    dtype = torch.float64 if rng.uniform() < 0.6 else torch.float32
    full = bool(rng.uniform() < 0.4)
    p = 6 if rng.uniform() < 0.3 else 1
    sm = float(rng.uniform(0.8, 1.3))
    pot = tpa.CoulombPotential(smearing=sm) if p == 1 else tpa.InversePowerLawPotential(exponent=6, smearing=sm)
    scheme = "P3M" if rng.uniform() < 0.5 else "PME"
    order = int(rng.integers(2, 6)) if scheme == "P3M" else int(rng.integers(3, 8))
    Calc = tpa.P3MCalculator if scheme == "P3M" else tpa.PMECalculator
    calc = Calc(pot, mesh_spacing=0.45, interpolation_nodes=order, full_neighbor_list=full).to(dtype)
    t = lambda a: torch.tensor(a, device=DEV, dtype=dtype)  # noqa: E731
    frames = []
    for _ in range(int(rng.integers(2, 7))):
        lengths = rng.uniform(7.6, 13.8, 3)  # 2 L / 0.45 + 1 in (33, 64]: meshes of 64^3
        cell = np.diag(lengths) + np.tril(rng.uniform(-0.02, 0.02, (3, 3)) * lengths.min(), -1)
        n_side = int(rng.integers(2, 7))
        N = int(rng.integers(max(2, n_side**3 // 2), n_side**3 + 1))
        sites = rng.permutation(n_side**3)[:N]
        grid = np.stack(np.unravel_index(sites, (n_side,) * 3), -1)
        pos = ((grid + 0.5 + rng.uniform(-0.2, 0.2, (N, 3))) / n_side) @ cell
        q = rng.normal(size=(N, 1))
        pairs, S, _ = tpa.neighbor_list(pos, cell, 3.4, full_list=full)
        frames.append((t(q), t(cell), t(pos), torch.tensor(pairs.reshape(-1, 2), device=DEV), t(S.reshape(-1, 3))))
    batch = tpa.GraphedFrameBatch(calc, frames)
    E, F = batch()
    tol = 1e-10 if dtype == torch.float64 else 5e-5
    for k, (q, cell, pos, pairs, S) in enumerate(frames):
        tp = pos.clone().requires_grad_(True)
        V = calc(q, cell, tp, pairs, tpa.pair_distances(tp, pairs, cell, S))
        Ek = tpa.weighted_sum(V, q)
        Ek.backward()
        info = f"seed {seed} frame {k}: {scheme}{order} p={p} N={pos.shape[0]} P={pairs.shape[0]} full={full} {dtype}"
        scale = float((q * V.detach()).abs().sum())
        assert abs(float(E[k]) - float(Ek.detach())) < tol * scale, info
        assert rell2(F[k].cpu().numpy(), -tp.grad.cpu().numpy()) < (1e-9 if dtype == torch.float64 else 2e-3), info


@pytest.mark.parametrize("seed", range(24))
def test_random_device_neighbor_list(seed):
    """The GPU cell-list builder against the host builder on random inputs: triclinic cells, atoms far outside the cell,
    any combination of periodic axes, half / full lists, cutoffs up to a third of the cell, fp64 / fp32 positions."""
    rng = np.random.default_rng(11000 + seed)
    lengths = rng.uniform(9.0, 16.0, 3)
    cell = np.diag(lengths) + np.tril(rng.uniform(-0.2, 0.2, (3, 3)) * lengths.min(), -1)
    N = int(rng.integers(1, 500))
    pos = rng.uniform(-0.8, 1.8, (N, 3)) @ cell
    periodic = tuple(bool(v) for v in rng.uniform(size=3) < 0.75)
    full = bool(rng.uniform() < 0.5)
    width = [abs(np.linalg.det(cell)) / np.linalg.norm(np.cross(cell[(d + 1) % 3], cell[(d + 2) % 3])) for d in range(3)]
    rc = float(rng.uniform(1.5, min(width) / 3.05))
    hp, hS, hd = tpa.neighbor_list(pos, cell, rc, full_list=full, periodic=periodic)
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    gp, gS, gd = tpa.neighbor_list_device(t(pos), t(cell), rc, full_list=full, periodic=periodic)
    info = f"seed {seed}: N={N} rc={rc:.3f} periodic={periodic} full={full} host pairs {len(hp)} device pairs {len(gp)}"
    assert len(gp) == len(hp), info

    def canon(p, S, d):
        key = np.lexsort((S[:, 2], S[:, 1], S[:, 0], p[:, 1], p[:, 0]))
        return p[key], S[key], d[key]

    a = canon(hp, hS.astype(np.int64), hd)
    b = canon(gp.cpu().numpy(), np.rint(gS.cpu().numpy()).astype(np.int64), gd.cpu().numpy())
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), info
    assert np.allclose(a[2], b[2], rtol=1e-12, atol=0), info

"""GPU parity tests (``-m gpu``): the HIP path, called through the C-ABI, against
(a) golden vectors generated from the reference (tests/golden/*.npz) and
(b) the NumPy oracle on seeded inputs.

Tolerances: fp64 rtol 1e-9 on potentials/gradients against reference autograd (observed ~1e-13);
fp32 energies within 1e-5 relative of the reference (the bar ``north_star`` states), per-atom
potentials/forces rel-L2 1e-5.
"""

import ast

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import torchpme_amd as tpa  # noqa: E402
from oracle import pme_numpy as O  # noqa: E402

DEV = "cuda"


def relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300))


def rell2(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / (np.linalg.norm(np.asarray(b)) + 1e-300))


def make_calc(meta, dtype=torch.float64):
    if meta["kind"] == "coulomb":
        pot = tpa.CoulombPotential(smearing=meta["smearing"], prefactor=meta["prefactor"],
                                   exclusion_radius=meta["exclusion_radius"])
    else:
        pot = tpa.InversePowerLawPotential(exponent=meta["exponent"], smearing=meta["smearing"],
                                           prefactor=meta["prefactor"], exclusion_radius=meta["exclusion_radius"])
    Calc = tpa.P3MCalculator if meta["scheme"] == "P3M" else tpa.PMECalculator
    return Calc(pot, mesh_spacing=meta["mesh_spacing"], interpolation_nodes=meta["order"],
                full_neighbor_list=meta["full_list"])


def _small_cases(golden_dir):
    z = np.load(f"{golden_dir}/ref_small.npz")
    return z, [str(n) for n in z["names"]]


def test_ref_small_all_cases(golden_dir):
    """Every scheme / order / potential / slab / exclusion case: V and all four gradients vs reference autograd."""
    z, names = _small_cases(golden_dir)
    worst = {}
    for nm in names:
        meta = ast.literal_eval(str(z[f"{nm}/meta"]))
        calc = make_calc(meta)
        t = lambda k, grad=False: torch.tensor(z[f"{nm}/{k}"], device=DEV, requires_grad=grad)  # noqa: E731
        q, pos, d = t("charges", True), t("positions", True), t("dist", True)
        cell = torch.tensor(z["cell"], device=DEV, requires_grad=True)
        pairs = t("pairs")
        per = None if meta["periodic"] is None else torch.tensor(meta["periodic"], device=DEV)
        V = calc(q, cell, pos, pairs, d, periodic=per)
        (V * t("g")).sum().backward()
        errs = dict(
            V=relmax(V.detach().cpu(), z[f"{nm}/V"]),
            q=relmax(q.grad.cpu(), z[f"{nm}/grad_charges"]),
            pos=relmax(pos.grad.cpu(), z[f"{nm}/grad_positions"]),
            cell=relmax(cell.grad.cpu(), z[f"{nm}/grad_cell"]),
            d=relmax(d.grad.cpu(), z[f"{nm}/grad_dist"]),
        )
        for k, v in errs.items():
            assert v < 1e-9, (nm, meta, k, v)
            worst[k] = max(worst.get(k, 0.0), v)
    print("ref_small worst rel errors:", worst)


@pytest.mark.parametrize("name", ["p3m5", "pme4"])
@pytest.mark.parametrize("tag,dtype", [("f64", torch.float64), ("f32", torch.float32)])
def test_ref_medium(golden_dir, name, tag, dtype):
    """512-atom box with a real cutoff list, distances through the HIP distance op."""
    z = np.load(f"{golden_dir}/ref_medium.npz")
    Calc = tpa.P3MCalculator if name == "p3m5" else tpa.PMECalculator
    calc = Calc(tpa.CoulombPotential(smearing=float(z["smearing"])), mesh_spacing=float(z[f"{name}/mesh_spacing"]),
                interpolation_nodes=int(z[f"{name}/order"]))
    pos = torch.tensor(z["positions"], device=DEV, dtype=dtype, requires_grad=True)
    cell = torch.tensor(z["cell"], device=DEV, dtype=dtype, requires_grad=True)
    q = torch.tensor(z["charges"], device=DEV, dtype=dtype, requires_grad=True)
    pairs = torch.tensor(z["pairs"], device=DEV)
    S = torch.tensor(z["shifts"], device=DEV, dtype=dtype)
    d = tpa.pair_distances(pos, pairs, cell, S)
    V = calc(q, cell, pos, pairs, d)
    E = (V * q).sum()
    E.backward()
    # always compare with the fp64 reference; for fp32 also with the reference's own fp32 result
    e64 = float(z[f"{name}/f64/energy"])
    relE = abs(E.item() - e64) / abs(e64)
    eV = rell2(V.detach().cpu(), z[f"{name}/f64/V"])
    eF = rell2(pos.grad.cpu(), z[f"{name}/f64/grad_positions"])
    eC = relmax(cell.grad.cpu(), z[f"{name}/f64/grad_cell"])
    eQ = rell2(q.grad.cpu(), z[f"{name}/f64/grad_charges"])
    print(f"{name} {tag}: relE={relE:.2e} V={eV:.2e} F={eF:.2e} cell={eC:.2e} q={eQ:.2e}")
    if dtype == torch.float64:
        assert relE < 1e-11 and eV < 1e-10 and eF < 1e-10 and eC < 1e-9 and eQ < 1e-10
    else:
        assert relE < 1e-5 and eV < 1e-5 and eF < 1e-5 and eC < 1e-4 and eQ < 1e-5
        e32 = float(z[f"{name}/f32/energy"])
        assert abs(E.item() - e32) / abs(e32) < 1e-5


@pytest.mark.parametrize("calc_name", ["pme", "p3m"])
@pytest.mark.parametrize("frame", [0, 1])
@pytest.mark.parametrize("full", [False, True])
def test_gromacs_frames(golden_dir, calc_name, frame, full):
    """GROMACS SPME energies (rtol 1e-4) and forces (rtol 5e-3): reference tests/calculators/test_values_ewald.py:223-315."""
    z = np.load(f"{golden_dir}/gromacs_frames.npz")
    pos_np, cell_np, q_np = z[f"{frame}/positions"], z[f"{frame}/cell"], z[f"{frame}/charges"].reshape(-1, 1)
    rc = 5.54
    sm = rc / 6
    pairs, S, _ = tpa.neighbor_list(pos_np, cell_np, rc, full_list=full)
    Calc = tpa.PMECalculator if calc_name == "pme" else tpa.P3MCalculator
    calc = Calc(tpa.CoulombPotential(smearing=sm, prefactor=float(z["prefactor_eV_A"])), mesh_spacing=sm / 8,
                full_neighbor_list=full)
    pos = torch.tensor(pos_np, device=DEV, requires_grad=True)
    cell = torch.tensor(cell_np, device=DEV)
    q = torch.tensor(q_np, device=DEV)
    d = tpa.pair_distances(pos, torch.tensor(pairs, device=DEV), cell, torch.tensor(S, device=DEV, dtype=torch.float64))
    V = calc(q, cell, pos, torch.tensor(pairs, device=DEV), d)
    E = (V * q).sum()
    (F,) = torch.autograd.grad(-E, pos)
    torch.testing.assert_close(E.item(), float(z[f"{frame}/energy"]), atol=0.0, rtol=1e-4)
    torch.testing.assert_close(F.cpu(), torch.tensor(z[f"{frame}/forces"]), atol=0.0, rtol=5e-3)
    # and the reference's own numbers for the same settings
    assert abs(E.item() - float(z[f"{frame}/{calc_name}/energy"])) < 1e-9 * abs(E.item())
    assert relmax(F.cpu(), z[f"{frame}/{calc_name}/forces"]) < 1e-8


CRYSTALS = ["CsCl", "NaCl_primitive", "NaCl_cubic", "zincblende", "wurtzite", "cu2o", "fluorite"]


@pytest.mark.parametrize("calc_name", ["pme", "p3m"])
@pytest.mark.parametrize("crystal", CRYSTALS)
@pytest.mark.parametrize("scaling", [1 / 2.0353610, 1.0, 3.4951291])
def test_madelung(golden_dir, calc_name, crystal, scaling):
    """Literature Madelung constants, rtol 9e-4 (reference tests/calculators/test_values_ewald.py:65-152)."""
    z = np.load(f"{golden_dir}/crystals.npz")
    pos_np = z[f"{crystal}/positions"] * scaling
    cell_np = z[f"{crystal}/cell"] * scaling
    q_np = z[f"{crystal}/charges"]
    madelung = float(z[f"{crystal}/madelung"]) / scaling
    nfu = int(z[f"{crystal}/n_formula"])
    rc = 2.0 * scaling
    sm = rc / 5.0
    Calc = tpa.PMECalculator if calc_name == "pme" else tpa.P3MCalculator
    calc = Calc(tpa.CoulombPotential(smearing=sm), mesh_spacing=sm / 8)
    pairs, S, dist = tpa.neighbor_list(pos_np, cell_np, rc)
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    V = calc(t(q_np), t(cell_np), t(pos_np), t(pairs), t(dist))
    energy = float((V.cpu().numpy() * q_np).sum())
    assert abs(-energy / nfu - madelung) / madelung < 9e-4


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("scheme,order", [("P3M", 4), ("P3M", 5), ("Lagrange", 4), ("Lagrange", 7)])
def test_against_oracle_random(seed, dtype, scheme, order):
    """Seeded triclinic boxes, 2 channels, atoms partly outside the cell: HIP vs the NumPy oracle (fp64 truth)."""
    rng = np.random.default_rng(100 + seed)
    cell = np.array([[9.0, 0, 0], [1.0, 8.0, 0], [0.5, -0.7, 10.0]])
    N = 300
    pos = rng.uniform(-3, 12, (N, 3))
    q = rng.normal(size=(N, 2))
    rc, sm, h = 4.0, 0.9, 0.7
    pairs, S, dist = tpa.neighbor_list(pos, cell, rc)
    spec = O.PotentialSpec("coulomb", 1, sm, 1.0)
    Vo, cache = O.forward(spec, scheme, order, h, q, cell, pos, pairs, dist, return_cache=True)
    g = rng.normal(size=(N, 2))
    gr = O.backward(cache, g)
    Calc = tpa.P3MCalculator if scheme == "P3M" else tpa.PMECalculator
    calc = Calc(tpa.CoulombPotential(smearing=sm), mesh_spacing=h, interpolation_nodes=order)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=dtype, requires_grad=grad)  # noqa: E731
    tq, tc, tp, td = t(q, True), t(cell, True), t(pos, True), t(dist, True)
    V = calc(tq, tc, tp, torch.tensor(pairs, device=DEV), td)
    (V * t(g)).sum().backward()
    tol = 1e-9 if dtype == torch.float64 else 3e-5
    assert rell2(V.detach().cpu(), Vo) < tol
    assert rell2(tq.grad.cpu(), gr["charges"]) < tol
    assert rell2(tp.grad.cpu(), gr["positions"]) < tol * 10
    assert rell2(td.grad.cpu(), gr["dist"]) < tol
    assert relmax(tc.grad.cpu(), gr["cell"]) < tol * 30


@pytest.mark.parametrize("p", [1, 3, 6])
def test_direct_molecules(golden_dir, p):
    """Exact direct sums with smearing=None (reference tests/calculators/test_values_direct.py)."""
    z = np.load(f"{golden_dir}/direct.npz")
    for nm in [str(n) for n in z["names"]]:
        pot = tpa.CoulombPotential() if p == 1 else tpa.InversePowerLawPotential(exponent=p)
        calc = tpa.Calculator(pot)
        t = lambda k: torch.tensor(z[f"{nm}/{k}"], device=DEV)  # noqa: E731
        V = calc(t("charges"), torch.eye(3, device=DEV, dtype=torch.float64), t("positions"), t("pairs"), t("dist"))
        np.testing.assert_allclose(V.cpu().numpy(), z[f"{nm}/V_p{p}"], rtol=1e-13, atol=2e-15)
        if p == 1:
            calc = tpa.Calculator(tpa.CoulombPotential(exclusion_radius=1.2, exclusion_degree=2))
            V = calc(t("charges"), torch.eye(3, device=DEV, dtype=torch.float64), t("positions"), t("pairs"), t("dist"))
            np.testing.assert_allclose(V.cpu().numpy(), z[f"{nm}/V_excl"], rtol=1e-12, atol=2e-15)


@pytest.mark.parametrize("mode", ["atomic", "rows"])
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_pair_modes(mode, full, idx_dtype, monkeypatch):
    """Both accumulation strategies of the pair kernels (hardware atomics / owner-computes rows), half and full
    lists, int64 and int32 indices, integer and non-integer shift tensors, with a pair mask: vs the oracle."""
    from torchpme_amd import ops

    monkeypatch.setattr(ops, "PAIR_MODE", mode)
    rng = np.random.default_rng(7)
    cell = np.array([[7.0, 0, 0], [0.7, 6.0, 0], [0.2, -0.5, 8.0]])
    N = 150
    pos = rng.uniform(-1, 8, (N, 3))
    q = rng.normal(size=(N, 3))
    rc, sm, h = 5.0, 1.1, 0.9  # rc > L/2: several images of the same pair
    pairs, S, dist = tpa.neighbor_list(pos, cell, rc, full_list=full)
    mask = rng.uniform(size=len(pairs)) > 0.2
    spec = O.PotentialSpec("coulomb", 1, sm, 0.7)
    g = rng.normal(size=(N, 3))
    Vo, cache = O.forward(spec, "P3M", 3, h, q, cell, pos, pairs, dist, full_list=full, pair_mask=mask, return_cache=True)
    gr = O.backward(cache, g)
    gpos_d, gcell_d = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=sm, prefactor=0.7), mesh_spacing=h, interpolation_nodes=3,
                             full_neighbor_list=full)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q, True), t(cell, True), t(pos, True)
    ti = torch.tensor(pairs, device=DEV, dtype=idx_dtype)
    for shifts in (torch.tensor(S, device=DEV), t(S)):  # int64 (converted) and float shifts
        for x in (tq, tc, tp):
            x.grad = None
        d = tpa.pair_distances(tp, ti, tc, shifts)
        np.testing.assert_allclose(d.detach().cpu().numpy(), dist, rtol=1e-13)
        V = calc(tq, tc, tp, ti, d, pair_mask=torch.tensor(mask, device=DEV))
        (V * t(g)).sum().backward()
        assert rell2(V.detach().cpu(), Vo) < 1e-11
        assert rell2(tq.grad.cpu(), gr["charges"]) < 1e-11
        assert rell2(tp.grad.cpu(), gr["positions"] + gpos_d) < 1e-10
        assert relmax(tc.grad.cpu(), gr["cell"] + gcell_d) < 1e-9


@pytest.mark.parametrize("mode", ["atomic", "bricks"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("scheme,order", [("P3M", 1), ("P3M", 2), ("P3M", 3), ("P3M", 5), ("Lagrange", 4), ("Lagrange", 6), ("Lagrange", 7)])
def test_mesh_modes(mode, dtype, scheme, order, monkeypatch):
    """Brick kernels (LDS tiles) and atomic-scatter kernels against the oracle on a non-cubic mesh whose sizes are
    not multiples of the brick (24 x 20 x 32), 2 channels, atoms far outside the cell, all gradients."""
    from torchpme_amd import ops

    monkeypatch.setattr(ops, "MESH_MODE", mode)
    rng = np.random.default_rng(order)
    cell = np.array([[11.0, 0, 0], [1.5, 9.0, 0], [0.7, -0.9, 15.0]])
    N = 400
    pos = rng.uniform(-20, 30, (N, 3))
    q = rng.normal(size=(N, 2))
    g = rng.normal(size=(N, 2))
    pairs = np.zeros((0, 2), dtype=np.int64)
    dist = np.zeros((0,))
    spec = O.PotentialSpec("coulomb", 1, 1.3, 1.0)
    ns = np.array([24, 20, 32])
    h = 1.0  # 2*|a|/h+1 -> 32, 32, 32 by the power-of-two rule; force the odd mesh through the stage API below
    Vo, cache = O.forward(spec, scheme, order, h, q, cell, pos, pairs, dist, ns=ns, return_cache=True)
    gr = O.backward(cache, g)
    # calculators derive ns from mesh_spacing; to exercise non-power-of-two meshes drive the ops layer directly
    geom = ops.MeshGeometry(cell, tuple(ns), 1 if scheme == "P3M" else 0, order)
    pot = tpa.CoulombPotential(smearing=1.3)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=dtype, requires_grad=grad)  # noqa: E731
    tq, tc, tp, td = t(q, True), t(cell, True), t(pos, True), t(dist, True)
    G = ops.build_filter(geom, pot._descriptor(), dtype, torch.device(DEV, 0))
    V = ops.pme_potential(tq, tc, tp, torch.tensor(pairs, device=DEV), td, None, geom, G, pot._descriptor(), False, None)
    (V * t(g)).sum().backward()
    tol = 1e-10 if dtype == torch.float64 else 2e-5
    assert rell2(V.detach().cpu(), Vo) < tol
    assert rell2(tq.grad.cpu(), gr["charges"]) < tol
    assert rell2(tp.grad.cpu(), gr["positions"]) < tol * 10
    assert relmax(tc.grad.cpu(), gr["cell"]) < tol * 50


def test_repeated_evaluations_give_the_same_numbers(golden_dir):
    """The same evaluation five times in a row on one stream (caches, plan counters and scratch reused): same numbers."""
    z = np.load(f"{golden_dir}/ref_medium.npz")
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=float(z["smearing"])),
                             mesh_spacing=float(z["p3m5/mesh_spacing"]), interpolation_nodes=5)
    pos = torch.tensor(z["positions"], device=DEV, requires_grad=True)
    cell = torch.tensor(z["cell"], device=DEV)
    q = torch.tensor(z["charges"], device=DEV, requires_grad=True)
    pairs = torch.tensor(z["pairs"], device=DEV)
    S = torch.tensor(z["shifts"], device=DEV, dtype=torch.float64)
    for _ in range(5):
        pos.grad = q.grad = None
        d = tpa.pair_distances(pos, pairs, cell, S)
        V = calc(q, cell, pos, pairs, d)
        (V * q).sum().backward()
        assert rell2(V.detach().cpu(), z["p3m5/f64/V"]) < 1e-10
        assert rell2(pos.grad.cpu(), z["p3m5/f64/grad_positions"]) < 1e-10
        assert rell2(q.grad.cpu(), z["p3m5/f64/grad_charges"]) < 1e-10


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_weighted_sum(dtype):
    torch.manual_seed(0)
    a = torch.randn(1000, 3, device=DEV, dtype=dtype, requires_grad=True)
    b = torch.randn(1000, 3, device=DEV, dtype=dtype, requires_grad=True)
    E = tpa.weighted_sum(a, b)
    (2.5 * E).backward()
    ref = (a.detach().double() * b.detach().double()).sum()
    assert abs(E.item() - ref.item()) < (1e-12 if dtype == torch.float64 else 1e-4) * max(1.0, abs(ref.item()))
    torch.testing.assert_close(a.grad, 2.5 * b.detach())
    torch.testing.assert_close(b.grad, 2.5 * a.detach())


def test_native_library_loaded():
    """The tests above must have run through libmipme.so (no silent fallback exists)."""
    with open("/proc/self/maps") as f:
        assert "libmipme.so" in f.read()


def test_graphed_energy_forces(golden_dir):
    """HIP-graph replay of distances -> forward -> backward reproduces the reference energy and forces, also after
    the positions are updated in place."""
    z = np.load(f"{golden_dir}/ref_medium.npz")
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=float(z["smearing"])),
                             mesh_spacing=float(z["p3m5/mesh_spacing"]), interpolation_nodes=5)
    t = lambda a, dt=torch.float64: torch.tensor(a, device=DEV, dtype=dt)  # noqa: E731
    pos, cell, q = t(z["positions"]), t(z["cell"]), t(z["charges"])
    pairs, S = torch.tensor(z["pairs"], device=DEV), t(z["shifts"])
    step = tpa.GraphedEnergyForces(calc, q, cell, pos + 0.01, pairs, S, store_distances=True)
    eref = float(z["p3m5/f64/energy"])
    E, F = step(pos)  # (E, F) are the graph's static output buffers: read them before the next replay
    e1 = E.item()
    assert abs(e1 - eref) < 1e-10 * abs(eref)
    assert rell2(F.cpu(), -z["p3m5/f64/grad_positions"]) < 1e-10
    # the distances of the step (written by the pair kernel, see pair_distances(deferred=True)) follow the positions
    d_ref = np.linalg.norm(z["positions"][z["pairs"][:, 1]] - z["positions"][z["pairs"][:, 0]] + z["shifts"] @ z["cell"], axis=1)
    assert relmax(step.distances.cpu(), d_ref) < 1e-14
    e2 = step(pos + 0.01)[0].item()
    e3 = step(pos)[0].item()
    assert abs(e2 - e1) > 1e-6 and abs(e3 - eref) < 1e-10 * abs(eref)
    assert relmax(step.distances.cpu(), d_ref) < 1e-14


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("variant", ["coulomb-p3m5", "p6-p3m3", "excl-pme4", "coulomb-pme7", "general-grad"])
def test_coscheduled_pair_sum(dtype, full, variant, monkeypatch):
    """The pair sum co-scheduled with the spread in one launch (``mipme_sr_job_t``; row workgroups of the fused distance +
    pair kernel behind the brick workgroups of the spread) against the two separate launches: potentials, deferred
    distances, energy-mode and general gradients; 1/r and 1/r^6 (co-scheduled kernels), a potential with an exclusion
    radius (sequential fallback inside the same C call), interpolation orders 3 to 7."""
    from torchpme_amd import ops

    rng = np.random.default_rng(17)
    cell = np.array([[9.0, 0, 0], [0.9, 8.0, 0], [0.3, -0.6, 10.0]])
    N = 260
    pos, q, w = rng.uniform(-1, 10, (N, 3)), rng.normal(size=(N, 1)), rng.normal(size=(N, 1))
    pairs, S, dist = tpa.neighbor_list(pos, cell, 4.0, full_list=full)
    pot = {"coulomb": tpa.CoulombPotential(smearing=1.0), "p6": tpa.InversePowerLawPotential(exponent=6, smearing=1.0),
           "excl": tpa.CoulombPotential(smearing=1.0, exclusion_radius=2.0), "general": tpa.CoulombPotential(smearing=1.0),
           }[variant.split("-")[0]]
    mesh = variant.split("-")[1]
    h = 0.45  # meshes of 64 x 64 x 64: brick kernels
    if mesh.startswith("pme"):
        calc = tpa.PMECalculator(pot, mesh_spacing=h, interpolation_nodes=int(mesh[3:]), full_neighbor_list=full)
    elif mesh.startswith("p3m"):
        calc = tpa.P3MCalculator(pot, mesh_spacing=h, interpolation_nodes=int(mesh[3:]), full_neighbor_list=full)
    else:
        calc = tpa.P3MCalculator(pot, mesh_spacing=h, full_neighbor_list=full)
    calc = calc.to(dtype)
    t = lambda a, dt=dtype: torch.tensor(a, device=DEV, dtype=dt)  # noqa: E731
    tq, tc, ti, tS, tw = t(q), t(cell), torch.tensor(pairs, device=DEV), t(S), t(w)
    res = {}
    for co in (True, False):
        monkeypatch.setattr(ops, "COSCHEDULE", co)
        tp = t(pos).requires_grad_(True)
        stages = {}
        from torchpme_amd import _lib
        _lib.profile_enable(True)
        d = tpa.pair_distances(tp, ti, tc, tS, deferred=True)
        V = calc(tq, tc, tp, ti, d)
        L = (V * tw).sum() if variant == "general-grad" else -0.7 * tpa.weighted_sum(V, tq)
        L.backward()
        stages = _lib.profile_report()
        _lib.profile_enable(False)
        expect_fused_launch = co and not variant.startswith("excl")
        assert ("spread+rspace_forward" in stages) == expect_fused_launch, stages.keys()
        res[co] = (d.detach().cpu().double().numpy(), V.detach().cpu().double().numpy(), tp.grad.cpu().double().numpy())
    tol = 1e-12 if dtype == torch.float64 else 5e-6
    assert relmax(res[True][0], dist) < (1e-14 if dtype == torch.float64 else 1e-6)
    for a, b in zip(res[True], res[False]):
        assert rell2(a, b) < tol


@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("kind", ["p3m", "direct"])
def test_energy_direct_gradient(golden_dir, full, kind, monkeypatch):
    """``weighted_sum`` of potentials that come straight from a calculator (fused distances, constant charges and cell):
    the position gradient is formed by one kernel from the per-atom sums of the forward pass -- neither the adjoint of the
    reduction nor the calculator's backward node runs.  Same numbers as the general path, also when the potentials feed
    a second loss term (that one goes through their own node; the contributions add) and for a scaled energy."""
    from torchpme_amd import ops

    rng = np.random.default_rng(8)
    cell = np.array([[7.0, 0, 0], [0.7, 6.0, 0], [0.2, -0.5, 8.0]])
    N = 140
    pos, q, w = rng.uniform(-1, 8, (N, 3)), rng.normal(size=(N, 1)), rng.normal(size=(N, 1))
    pairs, S, _ = tpa.neighbor_list(pos, cell, 4.5, full_list=full)
    if kind == "p3m":
        # mesh (64, 32, 64): more than 16 points per axis, so that the brick kernels (which form the mesh field) are used
        calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.1), mesh_spacing=0.4, full_neighbor_list=full)
    else:
        calc = tpa.Calculator(tpa.CoulombPotential(), full_neighbor_list=full)
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    tq, tc, ti, tS, tw = t(q), t(cell), t(pairs), t(S), t(w)
    res = {}
    for fast in (True, False):
        monkeypatch.setattr(ops, "ENERGY_FAST_PATH", fast)
        for second in (False, True):
            tp = t(pos).requires_grad_(True)
            calls = {}
            monkeypatch.setattr(ops, "PROFILE", calls)
            # deferred=True: the explicit opt-in to gradients that bypass the distance tensor (ops.DistanceSource.direct)
            V = calc(tq, tc, tp, ti, tpa.pair_distances(tp, ti, tc, tS, deferred=True))
            L = -1.7 * tpa.weighted_sum(V, tq)
            if second:
                L = L + (V * tw).sum()
            L.backward()
            monkeypatch.setattr(ops, "PROFILE", None)
            if fast and not second:  # direct: (the energy comes from the gather's tail when there is a mesh) + finalize
                assert "energy_sum_backward" not in calls and "rspace_backward" not in calls and "kspace_backward" not in calls
                assert "forces_finalize" in calls and ("energy_sum" in calls) == (kind == "direct")
            res[fast, second] = (L.item(), tp.grad.cpu().numpy())
    for second in (False, True):
        assert abs(res[True, second][0] - res[False, second][0]) < 1e-12 * abs(res[False, second][0])
        assert rell2(res[True, second][1], res[False, second][1]) < 1e-12


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("mesh_mode", ["bricks", "atomic"])
@pytest.mark.parametrize("name", ["p3m5", "pme4"])
def test_energy_fast_path(golden_dir, fast, mesh_mode, name, monkeypatch):
    """E = weighted_sum(V, q): the backward recognises grad = gE * charges and reuses the forward mesh (no second
    spread / FFT); forces and charge gradients must equal the general path and the reference, also for gE != 1."""
    from torchpme_amd import ops

    monkeypatch.setattr(ops, "ENERGY_FAST_PATH", fast)
    monkeypatch.setattr(ops, "MESH_MODE", mesh_mode)
    z = np.load(f"{golden_dir}/ref_medium.npz")
    Calc = tpa.P3MCalculator if name == "p3m5" else tpa.PMECalculator
    calc = Calc(tpa.CoulombPotential(smearing=float(z["smearing"])), mesh_spacing=float(z[f"{name}/mesh_spacing"]),
                interpolation_nodes=int(z[f"{name}/order"]))
    pos = torch.tensor(z["positions"], device=DEV, requires_grad=True)
    cell = torch.tensor(z["cell"], device=DEV)
    q = torch.tensor(z["charges"], device=DEV, requires_grad=True)
    pairs = torch.tensor(z["pairs"], device=DEV)
    S = torch.tensor(z["shifts"], device=DEV, dtype=torch.float64)
    d = tpa.pair_distances(pos, pairs, cell, S)
    V = calc(q, cell, pos, pairs, d)
    E = tpa.weighted_sum(V, q)
    (-1.7 * E).backward()
    assert abs(E.item() - float(z[f"{name}/f64/energy"])) < 1e-11 * abs(E.item())
    assert rell2(pos.grad.cpu(), -1.7 * z[f"{name}/f64/grad_positions"]) < 1e-10
    assert rell2(q.grad.cpu(), -1.7 * z[f"{name}/f64/grad_charges"]) < 1e-10


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("energy", [True, False])
@pytest.mark.parametrize("potential", ["coulomb", "excl", "p6"])
def test_fused_distances(fuse, full, energy, potential, monkeypatch):
    """Distances from ``pair_distances`` let the calculator recompute d in its row kernels and differentiate straight
    through to positions / cell (ops.FUSE_DISTANCES): same potentials and gradients as the unfused path and the
    oracle -- half and full lists, pair mask, triclinic cell with several images per pair, energy mode (speculative
    force sums finished by the finalize kernel) and an arbitrary upstream gradient, Coulomb / exclusion / 1/r^6."""
    from torchpme_amd import ops

    monkeypatch.setattr(ops, "FUSE_DISTANCES", fuse)
    rng = np.random.default_rng(11)
    cell = np.array([[7.0, 0, 0], [0.7, 6.0, 0], [0.2, -0.5, 8.0]])
    N = 170
    pos = rng.uniform(-1, 8, (N, 3))
    q = rng.normal(size=(N, 1))
    rc, sm, h = 5.0, 1.1, 0.9
    pairs, S, dist = tpa.neighbor_list(pos, cell, rc, full_list=full)
    mask = rng.uniform(size=len(pairs)) > 0.2
    if potential == "coulomb":
        spec, pot = O.PotentialSpec("coulomb", 1, sm, 0.7), tpa.CoulombPotential(smearing=sm, prefactor=0.7)
    elif potential == "excl":
        spec = O.PotentialSpec("coulomb", 1, sm, 1.0, exclusion_radius=2.5, exclusion_degree=2)
        pot = tpa.CoulombPotential(smearing=sm, exclusion_radius=2.5, exclusion_degree=2)
    else:
        spec, pot = O.PotentialSpec("ipl", 6, sm, 1.0), tpa.InversePowerLawPotential(exponent=6, smearing=sm)
    gE = -1.3
    g = gE * q if energy else rng.normal(size=(N, 1))
    Vo, cache = O.forward(spec, "P3M", 4, h, q, cell, pos, pairs, dist, full_list=full, pair_mask=mask, return_cache=True)
    gr = O.backward(cache, g)
    gpos_d, gcell_d = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    calc = tpa.P3MCalculator(pot, mesh_spacing=h, interpolation_nodes=4, full_neighbor_list=full)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q, not energy), t(cell, True), t(pos, True)
    ti = torch.tensor(pairs, device=DEV)
    d = tpa.pair_distances(tp, ti, tc, torch.tensor(S, device=DEV))
    calls = {}
    monkeypatch.setattr(ops, "PROFILE", calls)
    V = calc(tq, tc, tp, ti, d, pair_mask=torch.tensor(mask, device=DEV))
    if energy:
        (gE * tpa.weighted_sum(V, tq)).backward()
    else:
        (V * t(g)).sum().backward()
    monkeypatch.setattr(ops, "PROFILE", None)
    assert ("pair_distance_backward" in calls) == (not fuse)  # the fused path never touches dL/dd
    assert rell2(V.detach().cpu(), Vo) < 1e-11
    assert rell2(tp.grad.cpu(), gr["positions"] + gpos_d) < 1e-10
    assert relmax(tc.grad.cpu(), gr["cell"] + gcell_d) < 1e-9
    if not energy:
        assert rell2(tq.grad.cpu(), gr["charges"]) < 1e-11


def test_fused_distances_guards():
    """The fused path is taken only while the distance tensor is the untouched output of pair_distances for the same
    pair list; anything else (modified d, other list, leaf d, non-integer shifts, several channels) falls back."""
    from torchpme_amd import ops

    rng = np.random.default_rng(3)
    cell = np.eye(3) * 6.0
    pos = rng.uniform(0, 6, (60, 3))
    pairs, S, dist = tpa.neighbor_list(pos, cell, 2.9)
    tp = torch.tensor(pos, device=DEV, requires_grad=True)
    tc = torch.tensor(cell, device=DEV)
    ti = torch.tensor(pairs, device=DEV)
    tS = torch.tensor(S, device=DEV)
    d = tpa.pair_distances(tp, ti, tc, tS)
    src = d._mipme_src
    assert src.usable_for(d, ti, 1) and not src.usable_for(d, ti, 2) and not src.usable_for(d, ti.clone(), 1)
    assert not src.usable_for(d.detach(), ti, 1)
    d2 = tpa.pair_distances(tp, ti, tc, tS)
    d2.mul_(1.0)  # in-place edit bumps the version: provenance void
    assert not d2._mipme_src.usable_for(d2, ti, 1)
    # non-integer shifts cannot be packed: silently unfused, same numbers
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.8)
    q = torch.tensor(rng.normal(size=(60, 1)), device=DEV)
    Sf = tS.to(torch.float64) + 0.25
    outs = []
    for fuse in (True, False):
        ops.FUSE_DISTANCES = fuse
        try:
            dd = tpa.pair_distances(tp, ti, tc, Sf)
            outs.append(calc(q, tc, tp, ti, dd))
        finally:
            ops.FUSE_DISTANCES = True
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("variant", ["fused", "mask", "unsorted", "nograd", "p6", "ewald"])
def test_deferred_distances(dtype, full, variant, monkeypatch):
    """``pair_distances(..., deferred=True)``: the distance tensor is filled by the calculator's fused pair kernel (the row
    of a pair's first atom stores d[p]); with a pair mask / a list not ordered by its first index / a calculator without
    the fused path the stand-alone kernel runs first.  Same distances (vs the oracle), potentials and gradients as the
    eager distance kernel in every case, and the stand-alone kernel is launched exactly when expected."""
    from torchpme_amd import ops

    rng = np.random.default_rng(21)
    cell = np.array([[7.0, 0, 0], [0.7, 6.0, 0], [0.2, -0.5, 8.0]])
    N = 150
    pos = rng.uniform(-1, 8, (N, 3))
    q = rng.normal(size=(N, 1))
    pairs, S, dist = tpa.neighbor_list(pos, cell, 5.0, full_list=full)
    if variant == "unsorted":
        perm = rng.permutation(len(pairs))
        pairs, S, dist = pairs[perm], S[perm], dist[perm]
    mask = torch.tensor(rng.uniform(size=len(pairs)) > 0.2, device=DEV) if variant == "mask" else None
    pot = tpa.InversePowerLawPotential(exponent=6, smearing=1.1) if variant == "p6" else tpa.CoulombPotential(smearing=1.1)
    if variant == "ewald":
        calc = tpa.EwaldCalculator(pot, lr_wavelength=2.0, full_neighbor_list=full)
    else:
        calc = tpa.P3MCalculator(pot, mesh_spacing=0.9, interpolation_nodes=4, full_neighbor_list=full)
    calc = calc.to(dtype)
    grad = variant != "nograd"
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    res = []
    for deferred in (False, True):
        tq = torch.tensor(q, device=DEV, dtype=dtype)
        tc = torch.tensor(cell, device=DEV, dtype=dtype, requires_grad=grad)
        tp = torch.tensor(pos, device=DEV, dtype=dtype, requires_grad=grad)
        calls = {}
        monkeypatch.setattr(ops, "PROFILE", calls)
        d = tpa.pair_distances(tp, ti, tc, tS, deferred=deferred)
        assert d._mipme_src.pending == deferred
        V = calc(tq, tc, tp, ti, d, pair_mask=mask)
        assert not d._mipme_src.pending
        if grad:
            tpa.weighted_sum(V, tq).backward()
        monkeypatch.setattr(ops, "PROFILE", None)
        by_product = deferred and variant not in ("mask", "unsorted")
        assert ("pair_distance_forward" in calls) == (not by_product), calls.keys()
        res.append((d.detach().cpu(), V.detach().cpu(), tp.grad.cpu() if grad else None, tc.grad.cpu() if grad else None))
    tol = 1e-13 if dtype == torch.float64 else 2e-6
    assert relmax(res[1][0], dist) < (1e-14 if dtype == torch.float64 else 5e-7)  # the by-product distances vs the list builder's
    assert relmax(res[0][0], dist) < (1e-14 if dtype == torch.float64 else 5e-7)
    for a, b in zip(res[0][1:], res[1][1:]):
        if a is not None:
            assert rell2(b, a.numpy()) < tol


def test_deferred_distances_other_consumers():
    """A deferred tensor that reaches a consumer without the fused path (several charge channels; a second calculator call
    after the first one filled it) is materialised / reused correctly."""
    rng = np.random.default_rng(4)
    cell = np.eye(3) * 6.0
    pos = rng.uniform(0, 6, (80, 3))
    pairs, S, dist = tpa.neighbor_list(pos, cell, 2.9)
    tp = torch.tensor(pos, device=DEV, requires_grad=True)
    tc, ti, tS = torch.tensor(cell, device=DEV), torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    calc = tpa.PMECalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.8)
    q2 = torch.tensor(rng.normal(size=(80, 2)), device=DEV)
    d = tpa.pair_distances(tp, ti, tc, tS, deferred=True)
    V2 = calc(q2, tc, tp, ti, d)  # two channels: unfused pair kernels read d
    assert relmax(d.detach().cpu(), dist) < 1e-14
    V2_ref = calc(q2, tc, tp, ti, tpa.pair_distances(tp, ti, tc, tS))
    torch.testing.assert_close(V2, V2_ref, rtol=1e-13, atol=1e-13)
    d1 = tpa.pair_distances(tp, ti, tc, tS, deferred=True)
    Va = calc(q2[:, :1].contiguous(), tc, tp, ti, d1)  # fills d1
    Vb = calc(q2[:, :1].contiguous(), tc, tp, ti, d1)  # reads the provenance again: nothing pending
    torch.testing.assert_close(Va, Vb, rtol=1e-13, atol=1e-13)
    assert relmax(d1.detach().cpu(), dist) < 1e-14


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("cell_diag,h", [((5.0, 9.0, 17.0), 1.1), ((3.0, 3.0, 3.0), 2.9), ((30.0, 7.0, 7.0), 0.6)])
def test_fused_convolution(dtype, cell_diag, h, monkeypatch):
    """(y,z) hipFFT planes + one x-FFT * G * inverse-x-FFT kernel (ops.XFUSED) against the 3-D hipFFT plans + filter
    kernel, forward and general backward, several channels, unequal mesh sizes (nx from 4 to 128)."""
    from torchpme_amd import ops

    rng = np.random.default_rng(5)
    N = 90
    cell = torch.tensor(np.diag(cell_diag) + rng.normal(scale=0.1, size=(3, 3)), device=DEV, dtype=dtype)
    pos = torch.tensor(rng.uniform(0, 1, (N, 3)) * np.array(cell_diag), device=DEV, dtype=dtype)
    q = torch.tensor(rng.normal(size=(N, 2)), device=DEV, dtype=dtype)
    g = torch.tensor(rng.normal(size=(N, 2)), device=DEV, dtype=dtype)
    none = torch.zeros((0, 2), dtype=torch.int64, device=DEV)
    nod = torch.zeros((0,), dtype=dtype, device=DEV)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "XFUSED", fused)
        calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=h, interpolation_nodes=3)
        p, c = pos.clone().requires_grad_(True), q.clone().requires_grad_(True)
        calls = {}
        monkeypatch.setattr(ops, "PROFILE", calls)
        V = calc(c, cell, p, none, nod)
        (V * g).sum().backward()
        monkeypatch.setattr(ops, "PROFILE", None)
        res.append((V.detach(), p.grad, c.grad))
    tol = 1e-11 if dtype == torch.float64 else 2e-4
    for a, b in zip(*res):
        assert rell2(a.cpu(), b.cpu().numpy()) < tol


@pytest.mark.parametrize("with_mask", [False, True])
def test_fused_distances_large_shifts(with_mask, monkeypatch):
    """Cutoff of several cell lengths: cell shifts beyond the +-3 range of the LDS shift table make the fused kernels fall
    back to the 3 x int8 code; same numbers as the unfused path (also with a pair mask, which always takes that code)."""
    from torchpme_amd import ops

    rng = np.random.default_rng(2)
    cell = np.array([[2.1, 0, 0], [0.3, 1.9, 0], [0.0, 0.2, 2.3]])
    pos = rng.uniform(0, 2, (6, 3))
    q = rng.normal(size=(6, 1))
    pairs, S, dist = tpa.neighbor_list(pos, cell, 9.5)
    assert np.abs(S).max() > 3
    mask = torch.tensor(rng.uniform(size=len(pairs)) > 0.3, device=DEV) if with_mask else None
    res = []
    for fuse in (True, False):
        monkeypatch.setattr(ops, "FUSE_DISTANCES", fuse)
        tp = torch.tensor(pos, device=DEV, requires_grad=True)
        tc = torch.tensor(cell, device=DEV, requires_grad=True)
        tq = torch.tensor(q, device=DEV)
        ti = torch.tensor(pairs, device=DEV)
        calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.5), mesh_spacing=0.4, interpolation_nodes=4)
        d = tpa.pair_distances(tp, ti, tc, torch.tensor(S, device=DEV))
        V = calc(tq, tc, tp, ti, d, pair_mask=mask)
        tpa.weighted_sum(V, tq).backward()
        res.append((V.detach().cpu(), tp.grad.cpu(), tc.grad.cpu()))
    topo = ops.get_topology(ti, 6)
    assert topo.entries_with_shifts(torch.tensor(S, device=DEV, dtype=torch.float64))[1] == 0  # 3 x int8 code
    for a, b in zip(*res):
        assert rell2(a, b.numpy()) < 1e-10


def test_ref_ewald_all_cases(golden_dir):
    """EwaldCalculator: V and the gradients w.r.t. charges, positions, cell and distances against the
    reference's autograd for every golden case (Coulomb / 1/r^p, channels, slab, own k-vectors, node mask, full list)."""
    z = np.load(f"{golden_dir}/ref_ewald.npz")
    for nm in [str(n) for n in z["names"]]:
        meta = ast.literal_eval(str(z[f"{nm}/meta"]))
        pot = (tpa.CoulombPotential(smearing=meta["smearing"], prefactor=meta["prefactor"]) if meta["kind"] == "coulomb"
               else tpa.InversePowerLawPotential(exponent=meta["exponent"], smearing=meta["smearing"],
                                                 prefactor=meta["prefactor"]))
        calc = tpa.EwaldCalculator(pot, lr_wavelength=meta["lr_wavelength"], full_neighbor_list=meta["full_list"])
        t = lambda k, grad=False: torch.tensor(z[f"{nm}/{k}"], device=DEV, requires_grad=grad)  # noqa: E731
        q, pos, d = t("charges", True), t("positions", True), t("dist", True)
        cell = torch.tensor(z["cell"], device=DEV, requires_grad=True)
        per = None if meta["periodic"] is None else torch.tensor(meta["periodic"], device=DEV)
        kv = t("kvectors") if meta["own_kvectors"] else None
        mask = t("node_mask") if meta["node_mask"] else None
        V = calc(q, cell, pos, t("pairs"), d, periodic=per, kvectors=kv, node_mask=mask)
        (V * t("g")).sum().backward()
        errs = dict(V=relmax(V.detach().cpu(), z[f"{nm}/V"]), q=relmax(q.grad.cpu(), z[f"{nm}/grad_charges"]),
                    pos=relmax(pos.grad.cpu(), z[f"{nm}/grad_positions"]), cell=relmax(cell.grad.cpu(), z[f"{nm}/grad_cell"]),
                    d=relmax(d.grad.cpu(), z[f"{nm}/grad_dist"]))
        for k, v in errs.items():
            assert v < 1e-9, (nm, meta, k, v)
        # fp32 evaluation of the same case
        V32 = calc(q.detach().float(), cell.detach().float(), pos.detach().float(), t("pairs"), d.detach().float(),
                   periodic=per, kvectors=None if kv is None else kv.float(), node_mask=mask)
        assert V32.dtype == torch.float32 and relmax(V32.cpu().double(), z[f"{nm}/V"]) < 2e-4, (nm, meta)


@pytest.mark.parametrize("crystal", CRYSTALS)
def test_madelung_ewald(golden_dir, crystal):
    """Literature Madelung constants with the Ewald sum (reference tests/calculators/test_values_ewald.py:65-152,
    lr_wavelength = smearing / 2)."""
    z = np.load(f"{golden_dir}/crystals.npz")
    pos_np, cell_np, q_np = z[f"{crystal}/positions"], z[f"{crystal}/cell"], z[f"{crystal}/charges"]
    rc = 2.0
    sm = rc / 5.0
    calc = tpa.EwaldCalculator(tpa.CoulombPotential(smearing=sm), lr_wavelength=0.5 * sm)
    pairs, S, dist = tpa.neighbor_list(pos_np, cell_np, rc)
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    V = calc(t(q_np), t(cell_np), t(pos_np), t(pairs), t(dist))
    energy = float((V.cpu().numpy() * q_np).sum())
    madelung = float(z[f"{crystal}/madelung"])
    assert abs(-energy / int(z[f"{crystal}/n_formula"]) - madelung) / madelung < 9e-4


def test_ewald_constructor_errors():
    with pytest.raises(ValueError, match="Must specify range radius to use a potential with EwaldCalculator"):
        tpa.EwaldCalculator(tpa.CoulombPotential(smearing=None), lr_wavelength=1.0)
    with pytest.raises(ValueError, match="`smearing` is -1.0 but must be positive"):
        tpa.EwaldCalculator(tpa.CoulombPotential(smearing=-1.0), lr_wavelength=1.0)
    with pytest.raises(ValueError, match="`lr_wavelength` is -0.5 but must be positive"):
        tpa.EwaldCalculator(tpa.CoulombPotential(smearing=1.0), lr_wavelength=-0.5)


def test_concurrent_frames_on_streams():
    """Independent frames replayed concurrently on separate streams (bench.py --frames-per-gpu): every calculator owns its
    FFT plan / brick counters and every charges tensor its reduction scratch, so the results equal the sequential ones."""
    from torchpme_amd import workloads

    frames = []
    for f in range(4):
        w = workloads.ionic_box(n_side=10, n_mesh=16, cutoff=6.0, seed=50 + f)
        t = lambda a: torch.tensor(a, dtype=torch.float64, device=DEV)  # noqa: E731
        calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=4)
        frames.append(tpa.GraphedEnergyForces(calc, t(w.charges), t(w.cell), t(w.positions),
                                              torch.tensor(w.pairs, device=DEV), t(w.shifts)))
    ref = []
    for g in frames:
        E, F = g()
        ref.append((E.clone(), F.clone()))
    # the side streams below do not wait for the default stream: the reference replays (and the clones of their static
    # output buffers) must have finished before the concurrent replays overwrite those buffers
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(DEV) for _ in frames]
    for _ in range(20):
        for g, st in zip(frames, streams):
            with torch.cuda.stream(st):
                g.graph.replay()
    torch.cuda.synchronize()
    for g, (E, F) in zip(frames, ref):
        # (16^3 meshes use the atomic mesh kernels: the summation order of the fp64 atomics varies between runs)
        torch.testing.assert_close(g.energy, E, rtol=1e-10, atol=0)
        torch.testing.assert_close(g.forces, F, rtol=1e-9, atol=1e-11)


def test_graph_survives_cache_eviction(golden_dir):
    """A captured step keeps its cached inputs alive: after the topology / filter caches have been flushed by other work
    the replay still reproduces the reference."""
    import gc

    from torchpme_amd import ops

    z = np.load(f"{golden_dir}/ref_medium.npz")
    t = lambda k: torch.tensor(z[k], device=DEV)  # noqa: E731
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=float(z["smearing"])), mesh_spacing=float(z["p3m5/mesh_spacing"]),
                             interpolation_nodes=5)
    pos = t("positions")
    step = tpa.GraphedEnergyForces(calc, t("charges"), t("cell"), pos, t("pairs"), t("shifts").double())
    e_ref = float(z["p3m5/f64/energy"])
    # flush: many other pair lists through the topology cache, another cell through the calculator, lots of allocations
    rng = np.random.default_rng(0)
    for k in range(20):
        pr = torch.tensor(rng.integers(0, 50, (200, 2)), device=DEV)
        ops.get_topology(pr, 64)
    other_cell = t("cell") * 1.01
    calc(t("charges"), other_cell, pos, t("pairs"), tpa.pair_distances(pos, t("pairs"), other_cell, t("shifts").double()))
    ops._TOPOLOGIES.clear()
    ops._DOT_SCRATCH.clear()
    gc.collect()
    junk = [torch.full((1 << 20,), float(k), device=DEV) for k in range(64)]  # would overwrite freed blocks
    del junk
    E, F = step(pos)
    assert abs(E.item() - e_ref) < 1e-10 * abs(e_ref)
    assert rell2(F.cpu(), -z["p3m5/f64/grad_positions"]) < 1e-10


def test_graphed_cell_gradient(golden_dir):
    """GraphedEnergyForces(cell_gradient=True): energy, forces and dE/dcell (stress) of the replayed step equal the
    reference's autograd results."""
    z = np.load(f"{golden_dir}/ref_medium.npz")
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=float(z["smearing"])), mesh_spacing=float(z["p3m5/mesh_spacing"]),
                             interpolation_nodes=5)
    t = lambda k: torch.tensor(z[k], device=DEV)  # noqa: E731
    step = tpa.GraphedEnergyForces(calc, t("charges"), t("cell"), t("positions"), t("pairs"), t("shifts").double(),
                                   cell_gradient=True)
    for _ in range(2):
        E, F, dEdcell = step(t("positions"))
        assert abs(E.item() - float(z["p3m5/f64/energy"])) < 1e-10 * abs(E.item())
        assert rell2(F.cpu(), -z["p3m5/f64/grad_positions"]) < 1e-10
        assert relmax(dEdcell.cpu(), z["p3m5/f64/grad_cell"]) < 1e-9


@pytest.mark.parametrize("energy", [True, False])
def test_fused_path_with_slab_correction(energy):
    """2-D periodic slab (periodic = [T, T, F]) through pair_distances -> calculator: the slab term keeps the mesh part on
    the general backward while the pair part may still take its energy-mode shortcut; compare with the oracle."""
    rng = np.random.default_rng(21)
    cell = np.array([[8.0, 0, 0], [0.5, 7.0, 0], [0, 0, 30.0]])
    N = 120
    pos = np.column_stack([rng.uniform(0, 8, N), rng.uniform(0, 7, N), rng.uniform(10, 18, N)])
    q = rng.normal(size=(N, 1))
    q -= q.mean()
    periodic = (True, True, False)
    pairs, S, dist = tpa.neighbor_list(pos, cell, 5.0, periodic=periodic)
    spec = O.PotentialSpec("coulomb", 1, 1.0, 1.0)
    g = -0.7 * q if energy else rng.normal(size=(N, 1))
    Vo, cache = O.forward(spec, "P3M", 5, 0.8, q, cell, pos, pairs, dist, periodic=periodic, return_cache=True)
    gr = O.backward(cache, g)
    gpos_d, gcell_d = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.8, interpolation_nodes=5)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=torch.float64, requires_grad=grad)  # noqa: E731
    tq, tc, tp = t(q), t(cell, True), t(pos, True)
    ti = torch.tensor(pairs, device=DEV)
    d = tpa.pair_distances(tp, ti, tc, torch.tensor(S, device=DEV))
    V = calc(tq, tc, tp, ti, d, periodic=torch.tensor(periodic, device=DEV))
    if energy:
        (-0.7 * tpa.weighted_sum(V, tq)).backward()
    else:
        (V * t(g)).sum().backward()
    assert rell2(V.detach().cpu(), Vo) < 1e-11
    assert rell2(tp.grad.cpu(), gr["positions"] + gpos_d) < 1e-10
    assert relmax(tc.grad.cpu(), gr["cell"] + gcell_d) < 1e-9


def test_graphed_recapture(golden_dir):
    """GraphedEnergyForces.recapture with a new neighbour list (different pair order and a larger cutoff list of the same
    pairs): the replayed step follows the new list."""
    z = np.load(f"{golden_dir}/ref_medium.npz")
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=float(z["smearing"])), mesh_spacing=float(z["p3m5/mesh_spacing"]),
                             interpolation_nodes=5)
    t = lambda k: torch.tensor(z[k], device=DEV)  # noqa: E731
    pairs, shifts = t("pairs"), t("shifts").double()
    half = len(pairs) // 2
    step = tpa.GraphedEnergyForces(calc, t("charges"), t("cell"), t("positions"), pairs[:half].contiguous(),
                                   shifts[:half].contiguous())
    E_half, _ = step()
    e_half = E_half.item()
    perm = torch.randperm(len(pairs), device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    step.recapture(pairs[perm].contiguous(), shifts[perm].contiguous())
    E, F = step(t("positions"))
    assert abs(e_half - float(z["p3m5/f64/energy"])) > 1e-3 * abs(e_half)
    assert abs(E.item() - float(z["p3m5/f64/energy"])) < 1e-10 * abs(E.item())
    assert rell2(F.cpu(), -z["p3m5/f64/grad_positions"]) < 1e-10


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("kind", ["p3m-coulomb", "pme-p6"])
def test_frames_in_one_launch(dtype, full, kind):
    """``GraphedFrameBatch``: several independent frames (different atom counts, cells and neighbour lists, same mesh) with
    one launch per kernel of the pipeline (blockIdx.y = frame) against one ``GraphedEnergyForces`` per frame: energies,
    forces and the by-product distances, also after the positions are updated in place."""
    rng = np.random.default_rng(31)
    if kind == "p3m-coulomb":
        calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.45, interpolation_nodes=5,
                                 full_neighbor_list=full)
    else:
        calc = tpa.PMECalculator(tpa.InversePowerLawPotential(exponent=6, smearing=1.0), mesh_spacing=0.45,
                                 interpolation_nodes=4, full_neighbor_list=full)
    calc = calc.to(dtype)
    frames, singles = [], []
    for k, (N, L) in enumerate([(150, 9.0), (230, 9.6), (90, 8.8)]):  # 2L / 0.45 + 1 in (32, 64]: meshes of 64^3
        cell = np.array([[L, 0, 0], [0.05 * k * L, L, 0], [0, -0.03 * L, L]])
        pos = rng.uniform(0, L, (N, 3))
        q = rng.normal(size=(N, 1))
        pairs, S, _ = tpa.neighbor_list(pos, cell, 3.5, full_list=full)
        t = lambda a, dt=dtype: torch.tensor(a, device=DEV, dtype=dt)  # noqa: E731
        frames.append((t(q), t(cell), t(pos), torch.tensor(pairs, device=DEV), t(S)))
    batch = tpa.GraphedFrameBatch(calc, frames, store_distances=True)
    singles = [tpa.GraphedEnergyForces(calc, *(f[0], f[1], f[2], f[3], f[4]), store_distances=True) for f in frames]
    tol = 1e-11 if dtype == torch.float64 else 2e-5
    for shift in (0.0, 0.02):
        new = [f[2] + shift for f in frames]
        E, F = batch(new)
        E = E.cpu().numpy().copy()
        F = [x.cpu().numpy().copy() for x in F]
        for k, g in enumerate(singles):
            e1, f1 = g(new[k])
            assert abs(E[k] - e1.item()) < tol * abs(e1.item())
            assert rell2(F[k], f1.cpu().numpy()) < tol
            d_ref = g.distances.cpu().numpy()
            assert relmax(batch.distances[k].cpu().numpy(), d_ref) < (1e-13 if dtype == torch.float64 else 1e-6)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_plane_kernels_large_lds(dtype, monkeypatch):
    """(y,z) plane transforms of a 128^3 mesh: one workgroup holds a 128 x 65 half-complex plane in LDS (133 KB in fp64, above
    the default 64 KB dynamic-LDS limit, which the launcher raises); against the 3-D hipFFT plans."""
    from torchpme_amd import ops

    rng = np.random.default_rng(1)
    L, N = 40.0, 300
    t = lambda a: torch.tensor(a, device=DEV, dtype=dtype)  # noqa: E731
    cell, pos, q = t(np.eye(3) * L), t(rng.uniform(0, L, (N, 3))), t(rng.normal(size=(N, 1)))
    none, nod = torch.zeros((0, 2), dtype=torch.int64, device=DEV), torch.zeros((0,), dtype=dtype, device=DEV)
    res = []
    for xf in (True, False):
        monkeypatch.setattr(ops, "XFUSED", xf)
        calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.7, interpolation_nodes=5).to(dtype)
        res.append(calc(q, cell, pos, none, nod).double().cpu().numpy())
    assert rell2(res[0], res[1]) < (1e-13 if dtype == torch.float64 else 1e-6)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("cell_diag", [(20.0, 80.0, 80.0), (80.0, 20.0, 80.0), (20.0, 80.0, 20.0)])
def test_split_plane_kernels_256(dtype, cell_diag, monkeypatch):
    """(y,z) planes with a 256-wide axis exceed one workgroup's LDS (256 x 129 complex = 264 KB in fp32): the transforms run
    as two launches per direction -- z rows (32 or 16 rows per workgroup), y columns (16 or 8 kz per workgroup) -- instead of
    falling back to hipFFT's 2-D plans (round-1 verdict item 6).  Potentials and gradients against the 3-D hipFFT plans for
    meshes (64, 256, 256), (256, 64, 256) and (64, 256, 64), general upstream gradient."""
    from torchpme_amd import ops

    rng = np.random.default_rng(2)
    N = 400
    t = lambda a: torch.tensor(a, device=DEV, dtype=dtype)  # noqa: E731
    cell = t(np.diag(cell_diag) + rng.normal(scale=0.05, size=(3, 3)))
    pos, q, g = t(rng.uniform(0, 1, (N, 3)) * np.array(cell_diag)), t(rng.normal(size=(N, 1))), t(rng.normal(size=(N, 1)))
    none, nod = torch.zeros((0, 2), dtype=torch.int64, device=DEV), torch.zeros((0,), dtype=dtype, device=DEV)
    h = 0.7  # 80 A axes -> 2 * 80 / 0.7 + 1 = 230 -> 256 points, 20 A axes -> 58 -> 64 points
    res = []
    for xf in (True, False):
        monkeypatch.setattr(ops, "XFUSED", xf)
        calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.5), mesh_spacing=h, interpolation_nodes=4).to(dtype)
        p = pos.clone().requires_grad_(True)
        V = calc(q, cell, p, none, nod)
        assert calc._cache[6].ns == tuple(256 if d > 50 else 64 for d in cell_diag)
        (V * g).sum().backward()
        res.append((V.detach().double().cpu().numpy(), p.grad.double().cpu().numpy()))
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    assert rell2(res[0][0], res[1][0]) < tol and rell2(res[0][1], res[1][1]) < tol * 10


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("blob", ["corner", "two", "slab"])
def test_bin_overflow_region(dtype, blob):
    """One-pass binning: a brick owns a fixed number of slots (4 x the mean occupancy + 8) and the atoms that find it full go
    to the overflow region, which every particle <-> mesh kernel also walks.  A dense blob in an otherwise empty cell puts
    nearly all atoms there: potentials, energy and every gradient must still match the oracle (spread, gather, the general
    adjoint's second spread + gradient gather, the energy-mode gather tail)."""
    rng = np.random.default_rng(17)
    L = 24.0
    cell = np.eye(3) * L
    N = 260
    if blob == "corner":
        pos = rng.uniform(0.2, 2.8, (N, 3))  # a 2.6 A cube: one or two mesh bricks out of 512
    elif blob == "slab":
        # round 5, the PLANE lists of the plane spread (4 x the mean occupancy of an x plane + 64 entries each): a sheet of atoms
        # perpendicular to x puts ~300 atoms into each of two lists of 128 -- the rest goes through the plane overflow list
        N = 600
        gy, gz = np.meshgrid((np.arange(25) + 0.5) * L / 25, (np.arange(24) + 0.5) * L / 24, indexing="ij")
        pos = np.stack([rng.uniform(3.0, 3.35, N), gy.ravel() + rng.uniform(-0.2, 0.2, N), gz.ravel() + rng.uniform(-0.2, 0.2, N)], 1)
    else:
        pos = np.concatenate([rng.uniform(3.0, 5.5, (N // 2, 3)), rng.uniform(15.0, 17.5, (N - N // 2, 3))])
    q = rng.normal(size=(N, 1))
    q -= q.mean()
    sm, h = 1.0, 2 * L / 62  # -> 64^3 mesh: 512 bricks, mean occupancy 0.5 -> 8 slots per brick
    pairs, S, dist = tpa.neighbor_list(pos, cell, 4.0)
    spec = O.PotentialSpec("coulomb", 1, sm, 1.0)
    g = rng.normal(size=(N, 1))
    Vo, cache = O.forward(spec, "P3M", 5, h, q, cell, pos, pairs, dist, return_cache=True)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=sm), mesh_spacing=h, interpolation_nodes=5).to(dtype)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=dtype, requires_grad=grad)  # noqa: E731
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    tolV, tolG = (1e-11, 1e-10) if dtype == torch.float64 else (2e-5, 2e-4)
    for energy in (False, True):
        gr = O.backward(cache, q if energy else g)
        gpos_d, gcell_d = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
        tq, tc, tp = t(q, not energy), t(cell, True), t(pos, True)
        d = tpa.pair_distances(tp, ti, tc, tS, deferred=energy)
        V = calc(tq, tc, tp, ti, d)
        (tpa.weighted_sum(V, tq) if energy else (V * t(g)).sum()).backward()
        assert rell2(V.detach().cpu().double(), Vo) < tolV
        assert rell2(tp.grad.cpu().double(), gr["positions"] + gpos_d) < tolG
        assert relmax(tc.grad.cpu().double(), gr["cell"] + gcell_d) < 10 * tolG
        if not energy:
            assert rell2(tq.grad.cpu().double(), gr["charges"]) < tolV * 10
    # the graph-replayed step (gather tail: energy + forces in the gather launch) on the same blob
    tq, tc, tp = t(q), t(cell), t(pos)
    step = tpa.GraphedEnergyForces(calc, tq, tc, tp, ti, tS)
    E, F = step()
    gr = O.backward(cache, q)
    gpos_d, _ = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    assert abs(float(E) - float((Vo * q).sum())) < (1e-11 if dtype == torch.float64 else 2e-5) * abs(float((Vo * q).sum()))
    assert rell2(F.cpu().double(), -(gr["positions"] + gpos_d)) < tolG


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_sparse_brick_kernels(dtype):
    """Meshes with many sparsely filled bricks (here 128^3 = 4096 bricks, 1.5 atoms per brick on average plus a dense patch)
    take the quarter-size workgroups of the spread and of the gather, and the pair sum runs in a launch of its own: potentials
    and all gradients of the eager path and energy + forces of the replayed step (gather tail) against the oracle."""
    rng = np.random.default_rng(23)
    L = 64.0
    cell = np.array([[L, 0, 0], [0.04 * L, L, 0], [0, -0.03 * L, L]])
    N = 6000
    frac = rng.uniform(0, 1, (N, 3))
    frac[:600] = 0.45 + 0.08 * rng.uniform(0, 1, (600, 3))  # a patch at 20 x the mean density (bricks beyond their slot capacity)
    pos = frac @ cell
    q = rng.normal(size=(N, 1))
    q -= q.mean()
    sm, h = 1.2, 2 * L / 126  # -> 128^3
    pairs, S, dist = tpa.neighbor_list(pos, cell, 4.5)
    keep = dist > 0.7
    pairs, S, dist = pairs[keep], S[keep], dist[keep]
    spec = O.PotentialSpec("coulomb", 1, sm, 1.0)
    Vo, cache = O.forward(spec, "P3M", 5, h, q, cell, pos, pairs, dist, return_cache=True)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=sm), mesh_spacing=h, interpolation_nodes=5).to(dtype)
    assert calc._kspace_setup(torch.tensor(cell, device=DEV, dtype=dtype), dtype, torch.device(DEV))[0].ns == (128, 128, 128)
    t = lambda a, grad=False: torch.tensor(a, device=DEV, dtype=dtype, requires_grad=grad)  # noqa: E731
    ti, tS = torch.tensor(pairs, device=DEV), torch.tensor(S, device=DEV)
    tolV, tolG = (1e-10, 1e-9) if dtype == torch.float64 else (5e-5, 5e-4)
    g = rng.normal(size=(N, 1))
    for energy in (False, True):
        gr = O.backward(cache, q if energy else g)
        gpos_d, gcell_d = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
        tq, tc, tp = t(q, not energy), t(cell, True), t(pos, True)
        d = tpa.pair_distances(tp, ti, tc, tS)
        V = calc(tq, tc, tp, ti, d)
        (tpa.weighted_sum(V, tq) if energy else (V * t(g)).sum()).backward()
        assert rell2(V.detach().cpu().double(), Vo) < tolV
        assert rell2(tp.grad.cpu().double(), gr["positions"] + gpos_d) < tolG
        assert relmax(tc.grad.cpu().double(), gr["cell"] + gcell_d) < 20 * tolG
        if not energy:
            assert rell2(tq.grad.cpu().double(), gr["charges"]) < tolV * 10
    gr = O.backward(cache, q)
    gpos_d, _ = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    step = tpa.GraphedEnergyForces(calc, t(q), t(cell), t(pos), ti, tS)
    for _ in range(2):
        E, F = step()
        Eo = float((q * Vo).sum())
        assert abs(float(E) - Eo) < (1e-10 if dtype == torch.float64 else 2e-5) * float(np.abs(q * Vo).sum())
        assert rell2(F.cpu().double(), -(gr["positions"] + gpos_d)) < tolG


@pytest.mark.gpu
def test_random_sweeps_through_the_sparse_brick_kernels():
    """The sparse-brick variants (128-thread spread and gather tail, pair sum in its own launch) are chosen for large, thinly
    filled meshes only; with MIPME_SPARSE_FORCE=1 every brick mesh takes them, so the seeded random sweeps of the graph-replayed
    step and of the eager calculators (all schemes, orders, dtypes, triclinic cells, half / full lists) check them against the
    oracle too.  Separate process: the switch is read once per process."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, MIPME_SPARSE_FORCE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_fuzz.py", "-q", "-x", "-m", "gpu", "-k",
                        "test_random_graphed_step or test_random_configuration", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]

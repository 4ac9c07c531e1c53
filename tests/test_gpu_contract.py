"""The whole first-order autograd contract of the reference from ONE energy step: E = sum q V, F = -dE/dr, dE/dq, dE/dcell
(reference: tests/calculators/test_workflow.py:164-192 differentiates w.r.t. charges, cell, positions in one backward pass;
tuning/tuner.py:350-369 times exactly that; tests/calculators/test_values_ewald.py:317-356 validates the stress).

Here the gather launch also writes dE/dq = 2 V, and dE/dcell is assembled from partial sums of the pair kernel, of rider workgroups
of the inverse (y,z) launch (k-grid sums against the filter's derivative table) and of the gather + one single-workgroup launch
(mipme_kspace_forward_args_t.out_grad_charges / out_grad_cell; mipme_md_args_t.grad_charges / grad_cell).  Every case against the
oracle (oracle/pme_numpy.py, pinned to the reference's goldens by tests/test_oracle_golden.py): fp64 <= 1e-9 (observed 1e-13),
fp32 <= 2e-4 of the largest component (observed 5e-6); at the benchmark boxes' full size against tests/golden/workloads.npz."""

import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import ops, workloads  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pme_numpy as O  # noqa: E402

DEV = "cuda"
#: does the fp64 1/r^6 pair body form the cell sums of the energy step (rows_cell_supported in csrc/bricks.hip)?
FP64_IPL_CELL = True


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


def oracle_contract(spec, scheme, order, h, q, cell, pos, pairs, S, g=None):
    """E, dE/dr, dE/dq, dE/dcell of E = sum q V (g = None), or the gradients of L = sum g V with the charges held fixed in g."""
    dist, _ = O.pair_distances(pos, cell, pairs, S)
    V, cache = O.forward(spec, "P3M" if scheme == "P3M" else "Lagrange", order, h, q, cell, pos, pairs, dist, return_cache=True)
    gr = O.backward(cache, q if g is None else g)
    gpos, gcell_pair = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    E = float(((q if g is None else g) * V).sum())
    return E, gpos + gr["positions"], gr["charges"] + (V if g is None else 0.0), gr["cell"] + gcell_pair, V


def setup(dtype, scheme, order, expo, tri, seed=0):
    pos, cell, q = small_box(11 + order + seed, tri)
    if expo == 6:
        q = np.abs(q) + 0.5
    rc, sm = 5.5, 1.1
    hmesh = 2 * np.linalg.norm(cell, axis=1).min() / 30
    pairs, S, _ = tpa.neighbor_list(pos, cell, rc)
    spec = O.PotentialSpec("coulomb" if expo == 1 else "ipl", expo, sm, 1.0)
    pot = tpa.CoulombPotential(smearing=sm) if expo == 1 else tpa.InversePowerLawPotential(exponent=6, smearing=sm)
    Calc = tpa.P3MCalculator if scheme == "P3M" else tpa.PMECalculator
    calc = Calc(pot, mesh_spacing=hmesh, interpolation_nodes=order)
    t = dict(q=torch.tensor(q, dtype=dtype, device=DEV), cell=torch.tensor(cell, dtype=dtype, device=DEV),
             pos=torch.tensor(pos, dtype=dtype, device=DEV), pairs=torch.tensor(pairs, device=DEV),
             shifts=torch.tensor(S, dtype=dtype, device=DEV))
    return dict(np=(spec, scheme, order, hmesh, q, cell, pos, pairs, S), calc=calc, rc=rc, **t)


def rel(a, b):
    a = a.detach().cpu().double().numpy()
    return float(np.abs(a - b).max() / np.abs(b).max())


CASES = [("P3M", 5, 1, False), ("P3M", 4, 1, True), ("PME", 4, 1, True), ("P3M", 5, 6, False), ("P3M", 3, 1, True),
         ("P3M", 2, 1, False), ("PME", 7, 1, True), ("PME", 6, 6, True)]


@pytest.mark.parametrize("live", [False, True], ids=["binned", "live"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("scheme,order,expo,tri", CASES)
def test_graphed_step_returns_the_whole_contract(scheme, order, expo, tri, dtype, live):
    """GraphedEnergyForces(charge_gradient=True, cell_gradient=True): E, F, dE/dq, dE/dcell of two replays against the oracle;
    fp64 1/r^6 (no cell sums in that pair body) must take the general nodes and agree all the same."""
    c = setup(dtype, scheme, order, expo, tri)
    Eo, gpo, dqo, dco, _ = oracle_contract(*c["np"])
    if live:
        kw = dict(neighbors=c["rc"], live_bins=True)
        args = ()
    else:
        kw, args = {}, (c["pairs"], c["shifts"])
    fused_expected = not (expo == 6 and dtype == torch.float64) or FP64_IPL_CELL
    if live and not fused_expected:
        with pytest.raises(NotImplementedError):
            tpa.GraphedEnergyForces(c["calc"], c["q"], c["cell"], c["pos"], *args, charge_gradient=True, cell_gradient=True, **kw)
        return
    step = tpa.GraphedEnergyForces(c["calc"], c["q"], c["cell"], c["pos"], *args, charge_gradient=True, cell_gradient=True, **kw)
    assert step._fused_contract == fused_expected
    assert (step._live is not None) == live
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    for _ in range(2):
        E, F, dq, dc = step()
        torch.cuda.synchronize()
        assert abs(float(E) - Eo) <= tol * abs(Eo)
        assert rel(-F, gpo) <= tol and rel(dq, dqo) <= tol and rel(dc, dco) <= tol, (rel(-F, gpo), rel(dq, dqo), rel(dc, dco))
    # the single flags, and the order of the returned tuple
    s_q = tpa.GraphedEnergyForces(c["calc"], c["q"], c["cell"], c["pos"], *args, charge_gradient=True, **kw)
    out = s_q()
    assert len(out) == 3 and rel(out[2], dqo) <= tol
    s_c = tpa.GraphedEnergyForces(c["calc"], c["q"], c["cell"], c["pos"], *args, cell_gradient=True, **kw)
    out = s_c()
    assert len(out) == 3 and out[2].shape == (3, 3) and rel(out[2], dco) <= tol


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("reduce", ["weighted_sum", "tensor_ops"])
@pytest.mark.parametrize("distances", ["virtual", "default"])
def test_eager_autograd_contract_matches_the_oracle(dtype, reduce, distances):
    """The reference's call sequence with charges, cell and positions as leaves (tests/calculators/test_workflow.py:164-192):
    through `weighted_sum` on deferred distances (the gather tail serves all three gradients), through plain tensor ops and the
    default distance tensor (general nodes) -- same numbers."""
    c = setup(dtype, "P3M", 5, 1, True, seed=3)
    Eo, gpo, dqo, dco, _ = oracle_contract(*c["np"])
    pos = c["pos"].clone().requires_grad_(True)
    q = c["q"].clone().requires_grad_(True)
    cell = c["cell"].clone().requires_grad_(True)
    d = tpa.pair_distances(pos, c["pairs"], cell, c["shifts"], deferred="virtual" if distances == "virtual" else False)
    V = c["calc"](q, cell, pos, c["pairs"], d)
    E = tpa.weighted_sum(V, q) if reduce == "weighted_sum" else (q * V).sum()
    E.backward()
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    assert abs(float(E) - Eo) <= tol * abs(Eo)
    assert rel(pos.grad, gpo) <= tol and rel(q.grad, dqo) <= tol and rel(cell.grad, dco) <= tol
    # a seed other than one scales everything
    pos.grad = q.grad = cell.grad = None
    d = tpa.pair_distances(pos, c["pairs"], cell, c["shifts"], deferred="virtual" if distances == "virtual" else False)
    V = c["calc"](q, cell, pos, c["pairs"], d)
    E = tpa.weighted_sum(V, q) if reduce == "weighted_sum" else (q * V).sum()
    (E * -0.37).backward()
    assert rel(pos.grad, -0.37 * gpo) <= tol and rel(q.grad, -0.37 * dqo) <= tol and rel(cell.grad, -0.37 * dco) <= tol


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_distances_from_another_cell_tensor_leave_cell_the_mesh_part(dtype):
    """Round-4 advice: `cell.requires_grad` with distances that were built from a DIFFERENT (detached) cell tensor -- the
    reference's graph then gives `cell` the k-space part only (the pair part flows to the distances' cell, which wants none).
    `weighted_sum`'s direct node used to hand out mesh + pair part whenever only `cell` asked."""
    c = setup(dtype, "P3M", 5, 1, True, seed=9)
    spec, scheme, order, h, qn, celln, posn, pairs, S = c["np"]
    dist, _ = O.pair_distances(posn, celln, pairs, S)
    V, cache = O.forward(spec, "P3M", order, h, qn, celln, posn, pairs, dist, return_cache=True)
    gr = O.backward(cache, qn)
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    for reduce in ("weighted_sum", "tensor_ops"):
        pos = c["pos"].clone().requires_grad_(True)
        cell = c["cell"].clone().requires_grad_(True)
        d = tpa.pair_distances(pos, c["pairs"], cell.detach(), c["shifts"], deferred="virtual" if reduce == "weighted_sum" else False)
        Vt = c["calc"](c["q"], cell, pos, c["pairs"], d)
        E = tpa.weighted_sum(Vt, c["q"]) if reduce == "weighted_sum" else (c["q"] * Vt).sum()
        E.backward()
        assert rel(cell.grad, gr["cell"]) <= tol, (reduce, rel(cell.grad, gr["cell"]))


def test_tail_is_used_for_the_deferred_contract():
    """With deferred distances + weighted_sum the three gradients come out of the forward's launches: no kernel of the backward
    pass but the scaling of the stored results (launch names recorded by ops.PROFILE)."""
    c = setup(torch.float32, "P3M", 5, 1, False, seed=5)
    pos = c["pos"].clone().requires_grad_(True)
    q = c["q"].clone().requires_grad_(True)
    cell = c["cell"].clone().requires_grad_(True)
    ops.PROFILE = {}
    try:
        d = tpa.pair_distances(pos, c["pairs"], cell, c["shifts"], deferred="virtual")
        V = c["calc"](q, cell, pos, c["pairs"], d)
        node = V.grad_fn
        E = tpa.weighted_sum(V, q)
        E.backward()
        torch.cuda.synchronize()
        names = set(ops.PROFILE)
    finally:
        ops.PROFILE = None
    assert node.tail is not None and node.tail["grad_q"] is not None and node.tail["grad_cell"] is not None
    assert "kspace_backward" not in names and "rspace_backward" not in names and "energy_sum" not in names, names


def test_sum_seed_of_the_tuning_protocol():
    """tuning/tuner.py:350-369 literally: clones with requires_grad, constant distances, result.sum().backward() -- gradients
    w.r.t. positions, charges and cell against the oracle's adjoint for g = 1."""
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 2e-4)):
        c = setup(dtype, "P3M", 5, 1, True, seed=7)
        spec, scheme, order, h, qn, celln, posn, pairs, S = c["np"]
        dist, _ = O.pair_distances(posn, celln, pairs, S)
        V, cache = O.forward(spec, "P3M", order, h, qn, celln, posn, pairs, dist, return_cache=True)
        gr = O.backward(cache, np.ones_like(qn))
        d_fixed = torch.tensor(dist, dtype=dtype, device=DEV)
        for _ in range(2):
            positions, cell, charges = c["pos"].clone(), c["cell"].clone(), c["q"].clone()
            for t in (positions, cell, charges):
                t.requires_grad_(True)
            result = c["calc"].forward(positions=positions, charges=charges, cell=cell, neighbor_indices=c["pairs"],
                                       neighbor_distances=d_fixed)
            value = result.sum()
            value.backward(retain_graph=True)
            assert abs(float(value) - V.sum()) <= tol * np.abs(V).sum()
            assert rel(positions.grad, gr["positions"]) <= tol
            assert rel(charges.grad, gr["charges"]) <= tol
            assert rel(cell.grad, gr["cell"]) <= tol


@pytest.mark.parametrize("cfg", ["ionic", "water", "dispersion"])
def test_fullsize_contract_against_committed_oracle(cfg, golden_dir):
    """cfg2 (8 000 ions), cfg3 (31 944-atom water box) and cfg5 (262 144 atoms, 1/r^6, 128^3: the CELL variant of the 1/r^6 packed
    body and the interleaved block order of launches with >= 2 048 bricks) at full size, fp64 and fp32: dE/dq (256 sampled
    atoms + the whole-array checksum) and dE/dcell against tests/golden/workloads.npz, from the binned and the live-bin graph and
    from the eager tail; and the TuningTimings protocol's three gradients against the committed g = 1 adjoint.  fp32 tolerances
    are ~5 x the errors measured in round 4 (energy 1e-5, forces 2e-5, dE/dq 2e-5, dE/dcell 6e-5; cfg5's 1/r^6 forces and cell
    gradient are small differences of large terms: 1e-4 / 3e-4)."""
    w = {"ionic": workloads.ionic_box, "water": workloads.water_box, "dispersion": workloads.dispersion_box}[cfg]()
    z = np.load(os.path.join(golden_dir, "workloads.npz"))
    g = {k[len(cfg) + 1:]: z[k] for k in z.files if k.startswith(cfg + "_")}
    assert int(g["n_pairs"]) == w.n_pairs
    rng = np.random.default_rng(4242)
    rng.normal(size=(w.n_atoms, 3))
    s_vec = rng.normal(size=(w.n_atoms, 1))
    sample = g["sample"]
    disp = cfg == "dispersion"
    for dtype in (torch.float64, torch.float32):
        f64 = dtype == torch.float64
        tol_e, tol_f, tol_q, tol_c = (1e-10, 1e-9, 1e-9, 1e-9) if f64 else ((1e-5, 1e-4, 2e-5, 3e-4) if disp else (1e-5, 2e-5, 2e-5, 6e-5))
        tol = tol_q
        t = lambda a: torch.tensor(a, dtype=dtype, device=DEV)  # noqa: E731
        pos, cell, q, shifts = t(w.positions), t(w.cell), t(w.charges), t(w.shifts)
        pairs = torch.tensor(w.pairs, device=DEV)
        pot = (tpa.InversePowerLawPotential(exponent=w.exponent, smearing=w.smearing) if disp
               else tpa.CoulombPotential(smearing=w.smearing))
        calc = tpa.P3MCalculator(pot, mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
        results = {}
        fused_expected = not (disp and f64) or FP64_IPL_CELL  # (fp64 1/r^6: see test_graphed_step_returns_the_whole_contract)
        step = tpa.GraphedEnergyForces(calc, q, cell, pos, pairs, shifts, charge_gradient=True, cell_gradient=True)
        assert step._fused_contract == fused_expected
        results["graph"] = tuple(x.clone() for x in step())
        del step
        if fused_expected:
            live = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff, charge_gradient=True, cell_gradient=True)
            results["live"] = tuple(x.clone() for x in live())
            del live
        p_, q_, c_ = pos.clone().requires_grad_(True), q.clone().requires_grad_(True), cell.clone().requires_grad_(True)
        d = tpa.pair_distances(p_, pairs, c_, shifts, deferred="virtual")
        E = tpa.weighted_sum(calc(q_, c_, p_, pairs, d), q_)
        E.backward()
        results["eager"] = (E.detach(), -p_.grad, q_.grad, c_.grad)
        for name, (E, F, dq, dc) in results.items():
            dq = dq.cpu().double().numpy()
            assert abs(float(E) - float(g["energy"])) <= tol_e * abs(float(g["energy"])), (cfg, dtype, name)
            assert rel(F.cpu()[sample], g["force_sample"]) <= tol_f, (cfg, dtype, name, rel(F.cpu()[sample], g["force_sample"]))
            ref = g["charge_grad_sample"]
            assert np.abs(dq[sample, 0] - ref).max() <= tol_q * np.abs(ref).max(), (cfg, dtype, name)
            assert abs(float((s_vec * dq).sum()) - float(g["charge_grad_dot"])) <= tol_q * np.linalg.norm(s_vec) * np.linalg.norm(dq)
            assert rel(dc, g["cell_grad"]) <= tol_c, (cfg, dtype, name, rel(dc, g["cell_grad"]))
        # the reference's timing protocol on this box
        d_fixed = tpa.pair_distances(pos, pairs, cell, shifts).detach().clone()
        positions, cl, charges = pos.clone(), cell.clone(), q.clone()
        for x in (positions, cl, charges):
            x.requires_grad_(True)
        value = calc.forward(positions=positions, charges=charges, cell=cl, neighbor_indices=pairs,
                             neighbor_distances=d_fixed).sum()
        value.backward(retain_graph=True)
        assert abs(float(value) - float(g["sumseed_value"])) <= 10 * tol * abs(float(g["sumseed_value"])) + tol * np.abs(w.charges).sum()
        assert rel(positions.grad.cpu()[sample], g["sumseed_pos_sample"]) <= 10 * tol
        assert rel(charges.grad.cpu()[sample, 0], g["sumseed_charge_sample"]) <= 10 * tol
        assert rel(cl.grad, g["sumseed_cell"]) <= 10 * tol


def test_c_abi_refuses_what_the_tail_cannot_do():
    """out_grad_cell without its table / scratch, and out_grad_charges for a full list, are argument errors, not wrong numbers."""
    import ctypes as C

    from torchpme_amd import _lib

    lib = _lib.load()
    c = setup(torch.float32, "P3M", 5, 1, False)
    calc = c["calc"]
    geom, G = calc._kspace_setup(c["cell"], torch.float32, torch.device(DEV))
    md = geom.desc(1)
    plan = _lib.get_plan(torch.device(DEV), torch.float32, geom.ns, 1, geom.plan_store)
    assert lib.mipme_cell_tail_work(plan.handle, C.byref(md), 1000) > 0
    assert lib.mipme_cell_tail_work(None, C.byref(md), 1000) == 0
    D = ops.filter_derivative(geom, calc.potential._descriptor(), torch.float32, torch.device(DEV))
    assert D.shape == (*G.shape, 4) and bool(torch.isfinite(D).all())
    # PME has no charge-assignment factor: beta = 0
    c2 = setup(torch.float32, "PME", 4, 1, False)
    geom2, G2 = c2["calc"]._kspace_setup(c2["cell"], torch.float32, torch.device(DEV))
    D2 = ops.filter_derivative(geom2, c2["calc"].potential._descriptor(), torch.float32, torch.device(DEV))
    assert float(D2[..., 1:].abs().max()) == 0.0 and float(D2[..., 0].abs().max()) > 0.0


def test_new_list_tensors_with_old_values_reuse_the_structures(monkeypatch):
    """The reference's users hand a fresh neighbour list to every call (examples/02-neighbor-lists-usage.py:97-164).  A NEW
    `neighbor_indices` / shifts tensor with the values of the previous one reuses the transposed list and the entry streams on a
    bet that is verified on the device (ops.SPECULATE_LISTS, mipme_checksum); a tensor of the same shape with OTHER values loses
    the bet and gets freshly built structures -- the results are right either way."""
    c = setup(torch.float64, "P3M", 5, 1, True, seed=9)
    Eo, gpo, _, _, _ = oracle_contract(*c["np"])
    built = []
    orig = ops.PairTopology.__init__

    def counting(self, pairs, n_atoms):
        built.append(pairs.shape[0])
        orig(self, pairs, n_atoms)

    monkeypatch.setattr(ops.PairTopology, "__init__", counting)
    ops._TOPOLOGIES.clear()
    ops._BET_PAUSE[0] = 0

    def energy_forces(pairs, shifts, plain_distances=False):
        pos = c["pos"].clone().requires_grad_(True)
        d = tpa.pair_distances(pos, pairs, c["cell"], shifts)
        if plain_distances:
            d = d.detach().clone()
        V = c["calc"](c["q"], c["cell"], pos, pairs, d)
        E = (c["q"] * V).sum()
        E.backward()
        return float(E), pos.grad

    E, g = energy_forces(c["pairs"], c["shifts"])
    assert len(built) == 1 and abs(E - Eo) <= 1e-9 * abs(Eo) and rel(g, gpo) <= 1e-9
    for _ in range(3):  # clones: same values, new tensors -> no rebuild
        E, g = energy_forces(c["pairs"].clone(), c["shifts"].clone())
        assert abs(E - Eo) <= 1e-9 * abs(Eo) and rel(g, gpo) <= 1e-9
    assert len(built) == 1
    # other values in the same shape: reverse the list (i <-> j, shifts negated): another list of the same pairs
    pairs2 = c["pairs"].flip(1).contiguous()
    shifts2 = (-c["shifts"]).contiguous()
    E, g = energy_forces(pairs2, shifts2)
    assert len(built) == 2 and abs(E - Eo) <= 1e-9 * abs(Eo) and rel(g, gpo) <= 1e-9
    # the SAME pairs tensor with shifts of other values: one image moved (the energy changes, and must equal the oracle's)
    ops._BET_PAUSE[0] = 0
    S3 = c["np"][8].copy()
    k = int(np.argmax(np.abs(S3).sum(1) == 0))
    S3[k] = [1, 0, 0]
    spec, scheme, order, h, q, cell, pos, pairs, _ = c["np"]
    Eo3, gpo3, _, _, _ = oracle_contract(spec, scheme, order, h, q, cell, pos, pairs, S3)
    E, g = energy_forces(c["pairs"], torch.tensor(S3, dtype=torch.float64, device=DEV))
    assert abs(E - Eo3) <= 1e-9 * abs(Eo3) and rel(g, gpo3) <= 1e-9 and abs(Eo3 - Eo) > 1e-6 * abs(Eo)
    # a plain distance tensor (the calculator alone looks the list up): clone again
    ops._BET_PAUSE[0] = 0
    n = len(built)
    E1, _ = energy_forces(c["pairs"].clone(), c["shifts"], plain_distances=True)  # (the shifts tensor changed back: a lost bet)
    n = len(built)
    ops._BET_PAUSE[0] = 0
    E2, _ = energy_forces(c["pairs"].clone(), c["shifts"], plain_distances=True)
    assert len(built) == n and abs(E1 - Eo) <= 1e-9 * abs(Eo) and abs(E2 - Eo) <= 1e-9 * abs(Eo)


FRONT_CASES = [("P3M", 5, 1, False), ("PME", 4, 1, True), ("P3M", 4, 6, True)]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("mode", ["energy", "general", "energy+observed", "general+observed", "charges-only", "cell-only"])
@pytest.mark.parametrize("select", [1, 2], ids=["default", "on_device"])
@pytest.mark.parametrize("scheme,order,expo,tri", FRONT_CASES)
def test_compiled_front_end_serves_charges_and_cell(scheme, order, expo, tri, mode, dtype, select):
    """csrc/front.cpp with charges / cell as leaves (the reference's tests/calculators/test_workflow.py:164-192 through
    `pair_distances` + calculator + plain tensor reduction): the gather tail's dE/dq, dE/dcell on a match, the general adjoint
    otherwise, the pair part through the distances node when dE/d(neighbor_distances) is observed.  Twice through a retained
    graph (the forward's records must survive the charge adjoint)."""
    from torchpme_amd import _front

    if _front.module() is None:
        pytest.skip("compiled front end not built")
    c = setup(dtype, scheme, order, expo, tri, seed=1)
    spec, _, _, hmesh, q, cell, pos, pairs, S = c["np"]
    w = q if not mode.startswith("general") else np.random.default_rng(5).normal(size=q.shape)
    dist, _ = O.pair_distances(pos, cell, pairs, S)
    Vo, cache = O.forward(spec, "P3M" if scheme == "P3M" else "Lagrange", order, hmesh, q, cell, pos, pairs, dist, return_cache=True)
    gr = O.backward(cache, w)
    gpo, gcell_pair = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    gpo, dqo, dco, ddo = gpo + gr["positions"], gr["charges"], gr["cell"] + gcell_pair, gr["dist"]
    tp = c["pos"].clone().requires_grad_(True)
    tq = c["q"].clone().requires_grad_(mode != "cell-only")
    tc = c["cell"].clone().requires_grad_(mode != "charges-only")
    d = tpa.pair_distances(tp, c["pairs"], tc, c["shifts"])
    assert _front.module().is_front_distances(d)
    V = c["calc"](tq, tc, tp, c["pairs"], d)
    # (the compiled node serves fp64 1/r^6 with a cell gradient too: round 5 gave that pair body its cell sums)
    assert V.grad_fn.name() == "MipmeCalculatorBackward"
    if "observed" in mode:
        d.retain_grad()
    tw = tq.detach() if w is q else torch.tensor(w, dtype=dtype, device=DEV)
    tol = 1e-9 if dtype == torch.float64 else 3e-4
    _front.module().set_device_select(select)  # (front.cpp, g_device_select: who decides the energy mode)
    try:
        for rep in range(2):
            for t in (tp, tq, tc, d):
                t.grad = None
            (tw * V).sum().backward(retain_graph=True)
            assert rel(V, Vo) <= tol and rel(tp.grad, gpo) <= tol
            if tq.requires_grad:
                assert rel(tq.grad, dqo) <= tol
            if tc.requires_grad:
                assert rel(tc.grad, dco) <= tol
            if "observed" in mode:
                assert rel(d.grad, ddo) <= tol
    finally:
        _front.module().set_device_select(_front.select_mode())


def test_compiled_front_end_contract_with_the_polled_verdict():
    """The verdict of the energy-mode test read on the host (set_device_select(0); the default when charges / cell want
    gradients) or acted on by the device (set_device_select(2)): same numbers."""
    from torchpme_amd import _front

    mod = _front.module()
    if mod is None:
        pytest.skip("compiled front end not built")
    c = setup(torch.float64, "P3M", 5, 1, True, seed=2)
    Eo, gpo, dqo, dco, _ = oracle_contract(*c["np"])
    try:
        for device_select in (0, 1, 2):  # polled; the default mix (polled here: charges and cell want gradients); on the device
            mod.set_device_select(device_select)
            tp, tq, tc = (c[k].clone().requires_grad_(True) for k in ("pos", "q", "cell"))
            d = tpa.pair_distances(tp, c["pairs"], tc, c["shifts"])
            V = c["calc"](tq, tc, tp, c["pairs"], d)
            assert V.grad_fn.name() == "MipmeCalculatorBackward"
            E = (tq * V).sum()
            E.backward()
            assert abs(float(E) - Eo) <= 1e-9 * abs(Eo)
            assert rel(tp.grad, gpo) <= 1e-9 and rel(tq.grad, dqo) <= 1e-9 and rel(tc.grad, dco) <= 1e-9
    finally:
        mod.set_device_select(_front.select_mode())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("full_list", [False, True], ids=["half", "full"])
def test_constant_distances_take_the_tabulated_pair_sum(dtype, full_list):
    """The same `neighbor_distances` tensor again and again with changing charges -- the reference tuner's timing protocol
    (tuning/tuner.py:337-373) and any charge loop over a fixed geometry: from the second call on v_SR(d) per row entry comes
    from a table (mipme_rspace_rows_tabulate), the pair sum is a sparse matrix-vector product, and the charge gradient of a
    uniform upstream gradient is a multiple of the table's row sums.  Same numbers as the oracle on every call; a distance
    tensor modified in place is noticed (version counter)."""
    c = setup(dtype, "P3M", 4, 1, True, seed=4)
    spec, scheme, order, hmesh, q, cell, pos, pairs, S = c["np"]
    tol = 1e-9 if dtype == torch.float64 else 3e-4
    if full_list:
        pairs = np.concatenate([pairs, pairs[:, ::-1]])
        S = np.concatenate([S, -S])
    dist, _ = O.pair_distances(pos, cell, pairs, S)
    pot = tpa.CoulombPotential(smearing=1.1)
    calc = tpa.P3MCalculator(pot, mesh_spacing=hmesh, interpolation_nodes=order, full_neighbor_list=full_list)
    ti = torch.tensor(pairs, device=DEV)
    td = torch.tensor(dist, dtype=dtype, device=DEV)
    rng = np.random.default_rng(9)
    topo = None
    for call in range(5):
        qc = q * (1.0 + 0.1 * call) + 0.01 * rng.normal(size=q.shape)
        if call == 3:
            td.mul_(1.0)  # same values, new version: the table is dropped and rebuilt on the next sighting
        Vo, cache = O.forward(spec, "P3M", order, hmesh, qc, cell, pos, pairs, dist, return_cache=True, full_list=full_list)
        tq = torch.tensor(qc, dtype=dtype, device=DEV, requires_grad=True)
        tp = c["pos"].clone().requires_grad_(True)
        tc = c["cell"].clone().requires_grad_(True)
        V = calc(tq, tc, tp, ti, td)
        assert rel(V, Vo) <= tol
        topo = ops.get_topology(ti, len(q))
        assert (topo._tab[3] is not None) == (call in (1, 2, 4)), call
        if call % 2 == 0:  # uniform upstream gradient: result.sum().backward()
            gr = O.backward(cache, np.ones_like(qc))
            V.sum().backward()
        else:
            w = rng.normal(size=q.shape)
            gr = O.backward(cache, w)
            (torch.tensor(w, dtype=dtype, device=DEV) * V).sum().backward()
        assert rel(tq.grad, gr["charges"]) <= tol and rel(tp.grad, gr["positions"]) <= tol and rel(tc.grad, gr["cell"]) <= tol


def test_checksum_notices_single_words_swaps_and_tails():
    """mipme_checksum (the bet on "new list tensor, old values"): equal for copies, different after one changed word, after two
    swapped rows, after a change in the last words that do not fill a 16-byte vector; accumulates over several buffers."""
    g = torch.Generator(device="cpu").manual_seed(3)
    for n in (1, 3, 4, 1000, 4_000_003):
        a = torch.randint(-2**31, 2**31 - 1, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(DEV)
        ref = ops.device_checksum(a)[:2].clone()
        assert torch.equal(ops.device_checksum(a.clone())[:2], ref)
        for pos in {0, n // 2, n - 1}:
            b = a.clone()
            b[pos] += 1
            assert not torch.equal(ops.device_checksum(b)[:2], ref), (n, pos)
        if n >= 4:
            b = a.clone()
            b[[0, n - 1]] = b[[n - 1, 0]]
            assert bool(a[0] == a[n - 1]) or not torch.equal(ops.device_checksum(b)[:2], ref)
    # the pair list of a benchmark box: two rows swapped
    pairs = torch.tensor(workloads.water_box(n_side=8, n_mesh=16).pairs, device=DEV)
    ref = ops.device_checksum(pairs)[:2].clone()
    swapped = pairs.clone()
    swapped[[5, 77]] = swapped[[77, 5]]
    assert torch.equal(ops.device_checksum(pairs.clone())[:2], ref) and not torch.equal(ops.device_checksum(swapped)[:2], ref)

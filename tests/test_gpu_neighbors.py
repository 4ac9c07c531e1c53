"""Device neighbour list (``-m gpu``): the reference-format builder for any box / cutoff, and the row stream the pair kernels
read directly (``NeighborStream``, ``mipme_nl_stream``) -- its pair set against the host builder, energies / forces / other
gradients through it against the list-based path and the oracle, in-place refresh under a captured graph, energy conservation
of an MD run that refreshes its list.  Reference behaviour: a fresh list per call, ``tests/helpers.py:240-304``,
``examples/02-neighbor-lists-usage.py:97-164``.

Precision of "the same pair set": the device walk tests ``d <= cutoff`` in the dtype of the positions, the host builder in fp64.
With fp64 positions the sets are identical.  With fp32 positions a pair within a few ulp of the cutoff may fall on either side
(cfg3: 4 756 406 device pairs against 4 756 404, ``profiles/r03_k_refresh.txt``; each such pair carries v_SR(rc), ~1e-9 of the
energy scale).  The seeded fp32 cases below have no such pair, so exact equality is asserted there too; at full size the fp32
list is compared through energies and forces against the oracle (``test_gpu_fullsize.py``), not pair by pair."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import torchpme_amd as tpa  # noqa: E402
from oracle import pme_numpy as O  # noqa: E402
from torchpme_amd import workloads  # noqa: E402

DEV = "cuda"


def t(a, dtype=None):
    return torch.tensor(a, device=DEV, dtype=dtype)


def decode_stream(nl):
    """{(a, j, Sx, Sy, Sz)} of every entry in the stream (S = shift of the vector a -> j)."""
    N, cap = nl.n_atoms, nl.row_capacity
    rp = nl.row_ptr.cpu().numpy()
    w = nl.words.cpu().numpy()
    assert rp[3 * N] == N * cap
    out = []
    for a in range(N):
        beg, mid, end = rp[3 * a : 3 * a + 3]
        assert beg == a * cap and mid == end and end - beg <= cap
        ww = w[beg:end].astype(np.int64) & 0xFFFFFFFF
        j = ww & ((1 << 22) - 1)
        code = ww >> 22
        S = np.stack([code % 7 - 3, (code // 7) % 7 - 3, code // 49 - 3], axis=1)
        out.append(np.concatenate([np.full((len(j), 1), a), j[:, None], S], axis=1))
    return np.concatenate(out) if out else np.zeros((0, 5), dtype=np.int64)


def canon5(rows):
    rows = np.asarray(rows, dtype=np.int64)
    return rows[np.lexsort(rows.T[::-1])]


@pytest.mark.parametrize("triclinic", [False, True])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_stream_pair_set_matches_host_builder(triclinic, dtype):
    """Every row holds exactly the neighbours the host builder's FULL list gives that atom, with the shift of the vector from
    the atom to its neighbour; atoms outside the cell included."""
    rng = np.random.default_rng(21)
    cell = np.array([[13.0, 0, 0], [0, 14.0, 0], [0, 0, 12.5]]) if not triclinic else np.array(
        [[13.0, 0, 0], [2.0, 14.0, 0], [1.0, -1.5, 12.5]])
    N = 600
    pos = (rng.uniform(-0.7, 1.7, (N, 3)) @ cell).astype(np.float32 if dtype == torch.float32 else np.float64)
    rc = 3.7
    hp, hS, _ = tpa.neighbor_list(pos.astype(np.float64), cell, rc, full_list=True)
    nl = tpa.NeighborStream(t(pos, dtype), t(cell, dtype), rc)
    nl.check(synchronize=True)
    got = canon5(decode_stream(nl))
    want = canon5(np.concatenate([hp, hS], axis=1))
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    assert nl.n_entries == len(hp) and nl.longest_row == np.bincount(hp[:, 0], minlength=N).max()
    # deterministic: a second build gives the same words in the same order
    w1 = nl.words.clone()
    nl.update()
    rp = nl.row_ptr.cpu().numpy()
    for a in range(0, N, 37):
        assert torch.equal(nl.words[rp[3 * a]:rp[3 * a + 2]], w1[rp[3 * a]:rp[3 * a + 2]])
    # and the reference-format view of the same list
    p2, S2, d2 = nl.pairs(full_list=True)
    np.testing.assert_array_equal(canon5(np.concatenate([p2.cpu().numpy(), np.rint(S2.cpu().numpy()).astype(np.int64)], axis=1)), want)


@pytest.mark.parametrize("case", ["cscl", "long_cutoff", "one_cell", "slab", "planar", "thin_slab"])
def test_device_list_small_boxes(case):
    """Boxes smaller than three cutoffs -- down to a cutoff of several box lengths -- on the device (round 2 sent them to the
    host): same pairs, shifts and distances as the host builder; CsCl at rc = 2 has the survey's 58 half pairs."""
    rng = np.random.default_rng(3)
    periodic = (True, True, True)
    if case == "cscl":
        cell, pos, rc = np.eye(3), np.array([[0.0, 0, 0], [0.5, 0.5, 0.5]]), 2.0
    elif case == "long_cutoff":
        cell, pos, rc = np.array([[3.0, 0, 0], [0.4, 3.3, 0], [0.2, -0.3, 2.9]]), rng.uniform(0, 3, (7, 3)), 7.5
    elif case == "one_cell":
        cell, pos, rc = np.diag([5.0, 6.0, 5.5]), rng.uniform(-2, 7, (40, 3)), 4.9
    elif case == "slab":
        cell, pos, rc, periodic = np.diag([4.0, 4.5, 30.0]), rng.uniform(0, 4, (60, 3)) * [1, 1, 5], 5.0, (True, True, False)
    elif case == "planar":  # a sheet: every atom at one height of a non-periodic axis (round-3 advice: the reach along it exploded)
        pos = rng.uniform(0, 20, (80, 3))
        pos[:, 2] = 11.0
        cell, rc, periodic = np.diag([20.0, 20.0, 30.0]), 5.0, (True, True, False)
    else:  # a slab a hundred times thinner than the cutoff
        pos = rng.uniform(0, 12, (50, 3)) * [1, 1, 0.02 / 12]
        cell, rc, periodic = np.diag([12.0, 12.0, 30.0]), 5.0, (True, True, False)
    for full in (False, True):
        hp, hS, hd = tpa.neighbor_list(pos, cell, rc, full_list=full, periodic=periodic)
        gp, gS, gd = tpa.neighbor_list_device(t(pos), t(cell), rc, full_list=full, periodic=periodic)
        assert len(gp) == len(hp) and len(hp) > 0
        if case == "cscl" and not full:
            assert len(hp) == 58
        a = canon5(np.concatenate([hp, hS], axis=1))
        b = canon5(np.concatenate([gp.cpu().numpy(), np.rint(gS.cpu().numpy()).astype(np.int64)], axis=1))
        np.testing.assert_array_equal(a, b)
        ka = np.lexsort(np.concatenate([hp, hS], axis=1).T[::-1])
        kb = np.lexsort(np.concatenate([gp.cpu().numpy(), np.rint(gS.cpu().numpy()).astype(np.int64)], axis=1).T[::-1])
        np.testing.assert_allclose(gd.cpu().numpy()[kb], hd[ka], rtol=1e-12)


def test_device_list_of_one_atom_or_none():
    """N = 0 and N = 1 with a non-periodic axis (the cell grid spans the atoms' extent: a zero span must not turn into a reach
    of millions of cells), and N = 1 fully periodic with images within the cutoff."""
    cell = np.diag([6.0, 6.0, 30.0])
    for periodic in ((True, True, False), (True, True, True)):
        for n in (0, 1):
            pos = np.full((n, 3), 1.5)
            hp, hS, hd = tpa.neighbor_list(pos, cell, 7.0, periodic=periodic)
            gp, gS, gd = tpa.neighbor_list_device(t(pos.reshape(n, 3)), t(cell), 7.0, periodic=periodic)
            assert len(gp) == len(hp)
            if len(hp):
                a = canon5(np.concatenate([hp, hS], axis=1))
                b = canon5(np.concatenate([gp.cpu().numpy(), np.rint(gS.cpu().numpy()).astype(np.int64)], axis=1))
                np.testing.assert_array_equal(a, b)
    # the stream form of a planar system
    rng = np.random.default_rng(5)
    pos = rng.uniform(0, 20, (90, 3))
    pos[:, 2] = 3.0
    nl = tpa.NeighborStream(t(pos), t(np.diag([20.0, 20.0, 30.0])), 5.0, periodic=(True, True, False))
    nl.check(synchronize=True)
    hp, _, _ = tpa.neighbor_list(pos, np.diag([20.0, 20.0, 30.0]), 5.0, full_list=True, periodic=(True, True, False))
    assert nl.n_entries == len(hp)


def _small_water(n_side=6, dtype="f64"):
    return workloads.water_box(n_side=n_side, n_mesh=32, order=5, cutoff=6.0, dtype=dtype)


def _oracle_energy_forces(w):
    spec = O.PotentialSpec("coulomb" if w.exponent == 1 else "ipl", w.exponent, w.smearing, 1.0)
    dist, _ = O.pair_distances(w.positions, w.cell, w.pairs, w.shifts)
    V, cache = O.forward(spec, w.scheme, w.order, w.mesh_spacing, w.charges, w.cell, w.positions, w.pairs, dist, return_cache=True)
    gr = O.backward(cache, w.charges)
    gpos, gcell_d = O.pair_distances_backward(w.positions, w.cell, w.pairs, w.shifts, gr["dist"])
    return float((V * w.charges).sum()), -(gpos + gr["positions"]), gr, gcell_d


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_stream_energy_forces_against_oracle(dtype):
    """water_box() through the reference call sequence with the stream's handles: the oracle's energy and forces."""
    w = _small_water()
    E0, F0, _, _ = _oracle_energy_forces(w)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
    pos, cell, q = t(w.positions, dtype).requires_grad_(True), t(w.cell, dtype), t(w.charges, dtype)
    nl = tpa.NeighborStream(pos, cell, w.cutoff)
    assert nl.n_entries == 2 * w.n_pairs
    V = calc(q, cell, pos, nl.indices, nl.distances(pos, cell))
    E = (q * V).sum()
    E.backward()
    tol_e, tol_f = (1e-11, 1e-9) if dtype == torch.float64 else (1e-5, 3e-5)
    assert abs(E.item() - E0) <= tol_e * abs(E0)
    F = -pos.grad.cpu().numpy()
    assert np.linalg.norm(F - F0) <= tol_f * np.linalg.norm(F0)
    nl.check(synchronize=True)


def test_stream_other_gradients_match_list_path():
    """Charge, cell and position gradients of a general loss, PME and Ewald: the stream (8-byte expansion, symmetric rows) and
    the list-based path agree to rounding."""
    w = _small_water(n_side=5)
    rng = np.random.default_rng(0)
    wts = t(rng.normal(size=w.charges.shape))
    for make in (lambda: tpa.PMECalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=4),
                 lambda: tpa.EwaldCalculator(tpa.CoulombPotential(smearing=w.smearing), lr_wavelength=2.5),
                 lambda: tpa.P3MCalculator(tpa.InversePowerLawPotential(exponent=6, smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=3)):
        res = []
        for mode in ("list", "stream"):
            calc = make()
            pos, cell, q = t(w.positions).requires_grad_(True), t(w.cell).requires_grad_(True), t(w.charges).requires_grad_(True)
            if mode == "list":
                pairs, S = t(w.pairs), t(w.shifts).double()
                d = tpa.pair_distances(pos, pairs, cell, S)
            else:
                nl = tpa.NeighborStream(pos, cell, w.cutoff)
                pairs, d = nl.indices, nl.distances(pos, cell)
            V = calc(q, cell, pos, pairs, d)
            loss = (wts * V * V).sum() + (q * V).sum()
            loss.backward()
            res.append([x.detach().cpu().numpy() for x in (V, pos.grad, cell.grad, q.grad)])
        for a, b, name in zip(res[0], res[1], ("V", "dpos", "dcell", "dq")):
            assert np.abs(a - b).max() <= 1e-10 * np.abs(a).max(), (type(calc).__name__, name, np.abs(a - b).max(), np.abs(a).max())


def test_stream_refuses_what_it_cannot_do():
    w = _small_water(n_side=4)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing)
    pos, cell, q = t(w.positions), t(w.cell), t(w.charges)
    nl = tpa.NeighborStream(pos, cell, w.cutoff)
    with pytest.raises(ValueError, match="NeighborStream"):
        calc(q, cell, pos, nl.indices, torch.ones((1,), device=DEV, dtype=torch.float64))  # not the stream's distances
    with pytest.raises(ValueError, match="NeighborStream"):
        calc(torch.cat([q, q], 1), cell, pos, nl.indices, nl.distances(pos, cell))  # two channels
    with pytest.raises(ValueError, match="NeighborStream"):
        calc(q, cell, pos, nl.indices, nl.distances(pos, cell), pair_mask=torch.ones((1,), dtype=torch.bool, device=DEV))


def test_row_overflow_is_reported_and_grow_recovers():
    w = _small_water(n_side=5)
    pos, cell = t(w.positions), t(w.cell)
    nl = tpa.NeighborStream(pos, cell, w.cutoff, row_capacity=16)
    with pytest.raises(RuntimeError, match="a row needs"):
        nl.check(synchronize=True)
    need = nl.longest_row
    nl.grow()
    nl.check(synchronize=True)
    assert nl.row_capacity >= need and nl.n_entries == 2 * w.n_pairs


def test_far_atoms_are_reported():
    cell = np.diag([10.0, 10.0, 10.0])
    pos = np.random.default_rng(1).uniform(0, 10, (50, 3))
    pos[7] += [50.0, 0, 0]  # five cells away: shift codes beyond +-3
    nl = tpa.NeighborStream(t(pos), t(cell), 3.0)
    with pytest.raises(RuntimeError, match="cell shift beyond"):
        nl.check(synchronize=True)


def test_reference_format_list_refuses_atoms_beyond_the_wrap_range():
    """neighbor_list_device: an atom more than 400 grid cells outside the unit cell (its wrap integer is clamped in the binning
    pass) is an error, not a list with wrong shifts (round-3 advice); a merely distant atom gives the host builder's list."""
    cell = np.diag([10.0, 10.0, 10.0])
    pos = np.random.default_rng(1).uniform(0, 10, (50, 3))
    far = pos.copy()
    far[7] += [5000.0, 0, 0]  # 500 box lengths
    with pytest.raises(ValueError, match="more than 400 cells"):
        tpa.neighbor_list_device(t(far), t(cell), 3.0)
    near = pos.copy()
    near[7] += [50.0, 0, 0]
    gp, gS, _ = tpa.neighbor_list_device(t(near), t(cell), 3.0)
    hp, hS, _ = tpa.neighbor_list(near, cell, 3.0)
    np.testing.assert_array_equal(canon5(np.concatenate([hp, hS], axis=1)),
                                  canon5(np.concatenate([gp.cpu().numpy(), np.rint(gS.cpu().numpy()).astype(np.int64)], axis=1)))
    # and a later call with well-behaved atoms is not haunted by the stale flag
    gp, _, _ = tpa.neighbor_list_device(t(pos), t(cell), 3.0)
    assert len(gp) == len(tpa.neighbor_list(pos, cell, 3.0)[0])


def _close(E, F, E0, F0, tol_e, tol_f):
    eE = abs(E.item() - E0) / abs(E0)
    eF = np.linalg.norm(F.cpu().numpy() - F0) / np.linalg.norm(F0)
    assert eE <= tol_e and eF <= tol_f, (eE, eF)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_graphed_refresh_in_place(dtype):
    """GraphedEnergyForces(neighbors=cutoff): after the atoms moved, refresh() -- one graph replay, no re-capture, same
    buffers -- gives the energy and forces of a list built from scratch at the new positions (the oracle's)."""
    w = _small_water()
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
    pos, cell, q = t(w.positions, dtype), t(w.cell, dtype), t(w.charges, dtype)
    step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff)
    graph, words_ptr = step.graph, step.stream.words.data_ptr()
    E, F = step(pos)
    E0, F0, _, _ = _oracle_energy_forces(w)
    tol_e, tol_f = (1e-11, 1e-9) if dtype == torch.float64 else (1e-5, 3e-5)
    _close(E, F, E0, F0, tol_e, tol_f)
    # move every molecule by up to 1.5 A (far more than any skin): the old list is wrong, the refreshed one right
    rng = np.random.default_rng(77)
    shift = np.repeat(rng.uniform(-1.5, 1.5, (w.n_atoms // 3, 3)), 3, axis=0)
    new_pos = w.positions + shift
    p2, S2, _ = tpa.neighbor_list(new_pos, w.cell, w.cutoff)
    w2 = workloads.Workload(w.name, new_pos, w.charges, w.cell, p2, S2, w.cutoff, w.smearing, w.mesh_spacing, w.n_mesh,
                            w.scheme, w.order, w.exponent, w.dtype)
    E1, F1, _, _ = _oracle_energy_forces(w2)
    E_stale, _ = step(t(new_pos, dtype))
    stale_err = abs(E_stale.item() - E1) / abs(E1)
    if step._live is not None:
        # live bins: the step itself notices that atoms have moved more than a mesh point since the last refresh and says so
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="moved more than one mesh point"):
            step.refresh(check=True)
    step.refresh(check=True)
    E, F = step()
    assert step.graph is graph and step.stream.words.data_ptr() == words_ptr  # nothing was captured or allocated again
    _close(E, F, E1, F1, tol_e, tol_f)
    # the test moved the atoms far enough to matter (the live-bin step says so itself: NaN instead of a stale energy)
    assert np.isnan(stale_err) if step._live is not None else stale_err > 2 * tol_e
    assert step.stream.n_entries == 2 * len(p2)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("scheme,order,p", [("P3M", 5, 1), ("P3M", 3, 1), ("P3M", 2, 6), ("PME", 4, 1), ("PME", 7, 1), ("PME", 5, 6)])
def test_live_bins_match_the_binned_step(dtype, scheme, order, p):
    """mipme_md_step (bins and per-brick atom lists kept from the last refresh, weights evaluated on the fly) against the step
    that bins every call, after the atoms have moved by up to 0.45 mesh spacings since the refresh -- every order / scheme /
    exponent the kernels are instantiated for; and the same against the oracle for the headline case."""
    w = _small_water()
    pot = tpa.CoulombPotential(smearing=w.smearing) if p == 1 else tpa.InversePowerLawPotential(exponent=p, smearing=w.smearing)
    Calc = tpa.P3MCalculator if scheme == "P3M" else tpa.PMECalculator
    calc = Calc(pot, mesh_spacing=w.mesh_spacing, interpolation_nodes=order)
    pos, cell, q = t(w.positions, dtype), t(w.cell, dtype), t(w.charges, dtype)
    live = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff + 1.0, live_bins=True)
    ref = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff + 1.0, live_bins=False)
    assert live._live is not None and ref._live is None
    h = float(w.cell[0, 0]) / w.n_mesh
    rng = np.random.default_rng(12)
    moved = w.positions + rng.uniform(-0.45 * h, 0.45 * h, w.positions.shape) / np.sqrt(3.0)
    # the oracle on the SAME pair set (the list of the positions the stream was built from, skin included), at both positions
    hp, hS, _ = tpa.neighbor_list(w.positions, w.cell, w.cutoff + 1.0)
    spec = O.PotentialSpec("coulomb" if p == 1 else "ipl", p, w.smearing, 1.0)
    for x in (w.positions, moved):
        E1, F1 = live(t(x, dtype))
        E2, F2 = ref(t(x, dtype))
        tol = 1e-11 if dtype == torch.float64 else 2e-5
        assert abs(E1.item() - E2.item()) <= tol * abs(E2.item()), (E1.item(), E2.item())
        assert float((F1 - F2).norm() / F2.norm()) <= (1e-10 if dtype == torch.float64 else 5e-5)
        dist, _ = O.pair_distances(x, w.cell, hp, hS)
        V, cache = O.forward(spec, "P3M" if scheme == "P3M" else "Lagrange", order, w.mesh_spacing, w.charges, w.cell, x, hp, dist,
                             return_cache=True)
        gr = O.backward(cache, w.charges)
        gpos, _ = O.pair_distances_backward(x, w.cell, hp, hS, gr["dist"])
        Eo, Fo = float((V * w.charges).sum()), -(gpos + gr["positions"])
        tol_e, tol_f = (1e-10, 1e-9) if dtype == torch.float64 else (2e-5, 1e-4)
        assert abs(E1.item() - Eo) <= tol_e * abs(Eo), (E1.item(), Eo)
        assert float(np.abs(F1.cpu().double().numpy() - Fo).max()) <= tol_f * np.abs(Fo).max()
    live._deferred_check()
    torch.cuda.synchronize()
    live._live.check()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_live_margin_violation_is_loud_and_check_heals_it(dtype):
    """An atom that moves more than one mesh point between refreshes: the offending step returns NaN as its energy (not a
    plausible wrong number), the next plain use raises, and ``step(positions, check=True)`` refreshes and re-evaluates before
    it returns -- the oracle's energy and forces of a list built from scratch at the new positions.  ``max_displacement`` is
    the mesh spacing of a cubic cell."""
    w = _small_water()
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)
    pos, cell, q = t(w.positions, dtype), t(w.cell, dtype), t(w.charges, dtype)
    step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=w.cutoff + 0.5, live_bins=True)
    assert step._live is not None
    h = float(w.cell[0, 0]) / w.n_mesh
    assert abs(step.max_displacement - h) < 1e-6 * h
    rng = np.random.default_rng(4)
    shift = np.repeat(rng.uniform(-1.2, 1.2, (w.n_atoms // 3, 3)), 3, axis=0)  # up to 2 mesh points
    new_pos = w.positions + shift
    E_bad, _ = step(t(new_pos, dtype))
    torch.cuda.synchronize()
    assert bool(torch.isnan(E_bad))
    with pytest.raises(RuntimeError, match="moved more than one mesh point"):
        step()
    p2, S2, _ = tpa.neighbor_list(new_pos, w.cell, w.cutoff + 0.5)
    w2 = workloads.Workload(w.name, new_pos, w.charges, w.cell, p2, S2, w.cutoff, w.smearing, w.mesh_spacing, w.n_mesh,
                            w.scheme, w.order, w.exponent, w.dtype)
    E1, F1, _, _ = _oracle_energy_forces(w2)
    E, F = step(t(w.positions, dtype), check=True)  # back home first: a valid step
    E, F = step(t(new_pos, dtype), check=True)      # ... then the jump, healed inside the call
    tol_e, tol_f = (1e-10, 1e-9) if dtype == torch.float64 else (1e-5, 5e-5)
    _close(E, F, E1, F1, tol_e, tol_f)
    E, F = step()  # and the object is usable afterwards
    _close(E, F, E1, F1, tol_e, tol_f)


def test_nve_with_list_refresh_conserves_energy():
    """200 velocity-Verlet steps of a small ionic box with a skin of 1 A and a refresh every 10 steps: the total energy drifts
    no more than with a list rebuilt from scratch would (< 2e-4 of the kinetic energy scale)."""
    rng = np.random.default_rng(5)
    n_side, a = 8, 3.0
    L = n_side * a
    g = (np.arange(n_side) + 0.5) * a
    pos_np = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.2, 0.2, (n_side**3, 3))
    q_np = np.where((np.indices((n_side,) * 3).sum(0) % 2) == 0, 1.0, -1.0).reshape(-1, 1)
    N = len(pos_np)
    rc, skin = 7.0, 1.0
    # soft-core repulsion keeps the ions apart: a second calculator-free term evaluated with torch ops on the SAME list would
    # need the pairs, so use like-charge-only dynamics at low temperature instead: small dt, light thermal motion
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=rc / 5), mesh_spacing=2 * L / 30, interpolation_nodes=5)
    pos, cell, q = t(pos_np), t(L * np.eye(3)), t(q_np)
    step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=rc + skin)
    vel = t(rng.normal(size=(N, 3)) * 0.02)
    vel -= vel.mean(0)
    dt, mass = 0.05, 10.0
    E, F = step(pos)
    F = F.clone()
    x = pos.clone()
    x_ref = x.clone()
    total0 = E.item() + 0.5 * mass * (vel**2).sum().item()
    kin0 = 0.5 * mass * (vel**2).sum().item()
    n_refresh = 0
    totals = []
    for it in range(200):
        vel = vel + (0.5 * dt / mass) * F
        x = x + dt * vel
        if (x - x_ref).norm(dim=1).max().item() > 0.5 * skin or it % 10 == 9:
            step.refresh(x, check=True)
            x_ref = x.clone()
            n_refresh += 1
        E, F = step(x)
        F = F.clone()
        vel = vel + (0.5 * dt / mass) * F
        totals.append(E.item() + 0.5 * mass * (vel**2).sum().item())
    assert n_refresh >= 20
    drift = max(abs(v - total0) for v in totals)
    assert drift < 2e-3 * max(kin0, 1e-3) + 1e-7 * abs(total0), (drift, kin0, total0)

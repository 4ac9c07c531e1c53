"""Parity at BASELINE.json's full configuration sizes (cfg2: 8 000 charges fp64; cfg3: 31 944-atom water box fp32 / fp64;
cfg5: 262 144 atoms, 1/r^6, fp32) -- against the oracle where it finishes in seconds, and through size-independent
properties of the method everywhere: charge conservation on the mesh, linearity in the charges, invariance under lattice
translations, momentum conservation of the pair part, forces = -dE/dr by central differences, fused / unfused and
energy-mode / general gradient paths agreeing, fp32 within the stated tolerance of fp64 (1e-5 relative energy)."""

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


class Box:
    def __init__(self, w, dtype):
        self.w, self.dtype = w, dtype
        t = lambda a: torch.tensor(a, dtype=dtype, device=DEV)  # noqa: E731
        self.pos, self.cell, self.q = t(w.positions), t(w.cell), t(w.charges)
        self.pairs = torch.tensor(w.pairs, device=DEV)
        self.shifts = t(w.shifts)
        pot = (tpa.CoulombPotential(smearing=w.smearing) if w.exponent == 1
               else tpa.InversePowerLawPotential(exponent=w.exponent, smearing=w.smearing))
        self.pot = pot
        self.calc = tpa.P3MCalculator(pot, mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)

    def potentials(self, pos=None, q=None):
        pos = self.pos if pos is None else pos
        d = tpa.pair_distances(pos, self.pairs, self.cell, self.shifts)
        return self.calc(self.q if q is None else q, self.cell, pos, self.pairs, d)

    def energy_forces(self, pos=None, general=False):
        p = (self.pos if pos is None else pos).clone().requires_grad_(True)
        V = self.potentials(p)
        E = (V * self.q).sum() if general else tpa.weighted_sum(V, self.q)
        E.backward()
        return float(E.detach()), -p.grad


@pytest.fixture(scope="module")
def water():
    return workloads.water_box()


@pytest.fixture(scope="module")
def ionic():
    return workloads.ionic_box()


@pytest.fixture(scope="module")
def dispersion():
    return workloads.dispersion_box()


def rel(a, b):
    return float((a - b).norm() / b.norm())


def test_cfg2_against_oracle(ionic):
    """8 000 charges, 32^3, P3M n = 4, fp64: potentials and all gradients against the NumPy oracle at full size."""
    w = ionic
    box = Box(w, torch.float64)
    spec = O.PotentialSpec("coulomb", 1, w.smearing, 1.0)
    dist = O.pair_distances(w.positions, w.cell, w.pairs, w.shifts)[0]
    Vo, cache = O.forward(spec, "P3M", w.order, w.mesh_spacing, w.charges, w.cell, w.positions, w.pairs, dist,
                          return_cache=True)
    g = np.random.default_rng(0).normal(size=Vo.shape)
    gr = O.backward(cache, g)
    gpos_d, _ = O.pair_distances_backward(w.positions, w.cell, w.pairs, w.shifts, gr["dist"])
    p = box.pos.clone().requires_grad_(True)
    q = box.q.clone().requires_grad_(True)
    V = box.calc(q, box.cell, p, box.pairs, tpa.pair_distances(p, box.pairs, box.cell, box.shifts))
    (V * torch.tensor(g, device=DEV)).sum().backward()
    assert rel(V.detach().cpu(), torch.tensor(Vo)) < 1e-11
    assert rel(q.grad.cpu(), torch.tensor(gr["charges"])) < 1e-11
    assert rel(p.grad.cpu(), torch.tensor(gr["positions"] + gpos_d)) < 1e-10


def test_cfg3_forward_against_oracle(water):
    """31 944 atoms, 4.76 M pairs, 64^3, P3M n = 5: fp64 potentials against the oracle, fp32 within 2e-5 of it."""
    w = water
    spec = O.PotentialSpec("coulomb", 1, w.smearing, 1.0)
    dist = O.pair_distances(w.positions, w.cell, w.pairs, w.shifts)[0]
    Vo = torch.tensor(O.forward(spec, "P3M", w.order, w.mesh_spacing, w.charges, w.cell, w.positions, w.pairs, dist))
    assert rel(Box(w, torch.float64).potentials().cpu(), Vo) < 1e-11
    assert rel(Box(w, torch.float32).potentials().cpu().double(), Vo) < 2e-5


def _oracle_energy_forces(w):
    """E = sum q V and F = -dE/dr of a workload from the NumPy oracle (fp64): forward + analytic adjoint with g = q, the
    pair part chained through the distances (reference: autograd through compute_distances, tests/helpers.py:278-304)."""
    spec = O.PotentialSpec("coulomb" if w.exponent == 1 else "ipl", w.exponent, w.smearing, 1.0)
    dist = O.pair_distances(w.positions, w.cell, w.pairs, w.shifts)[0]
    Vo, cache = O.forward(spec, w.scheme, w.order, w.mesh_spacing, w.charges, w.cell, w.positions, w.pairs, dist,
                          return_cache=True)
    gr = O.backward(cache, w.charges)
    gpos_d, _ = O.pair_distances_backward(w.positions, w.cell, w.pairs, w.shifts, gr["dist"])
    return float((Vo * w.charges).sum()), -(gr["positions"] + gpos_d), dist


def test_cfg3_forces_against_oracle(water):
    """cfg3 at full size, energy AND forces against the oracle's analytic adjoint: fp64 <= 1e-10, fp32 energy within 1e-5
    and force rel-L2 within 1e-5 (north_star tolerance) -- through the eager calculators (fast and reference call
    sequence) and through the graph-replayed step bench.py times (GraphedEnergyForces)."""
    w = water
    Eo, Fo, dist_o = _oracle_energy_forces(w)
    Fo_t = torch.tensor(Fo)
    for dtype, tol_e, tol_f in ((torch.float64, 1e-11, 1e-10), (torch.float32, 1e-5, 1e-5)):
        box = Box(w, dtype)
        for general in (False, True):
            E, F = box.energy_forces(general=general)
            assert abs(E - Eo) < tol_e * abs(Eo), (dtype, general)
            assert rel(F.cpu().double(), Fo_t) < tol_f, (dtype, general)
        for store in (False, True):  # default: distances stay in registers (fp32: the packed pair body); True: by-product
            step = tpa.GraphedEnergyForces(box.calc, box.q, box.cell, box.pos, box.pairs, box.shifts, store_distances=store)
            for _ in range(2):  # the second replay must reproduce the first
                Eg, Fg = step()
                assert abs(float(Eg) - Eo) < tol_e * abs(Eo), (dtype, store)
                assert rel(Fg.cpu().double(), Fo_t) < tol_f, (dtype, store)
            if not store:
                assert step.distances is None
                continue
            # the distances the step produced as a by-product of its pair kernel
            dtol = 1e-13 if dtype == torch.float64 else 1e-5
            assert float(((step.distances.cpu().double() - torch.tensor(dist_o)).abs() / torch.tensor(dist_o)).max()) < dtol


def test_cfg4_one_gpu_share_against_oracle():
    """BASELINE.json configs[3] on one GPU: the 8 frames a rank owns (8 000 charges each, fp64, seeds 100..107 = rank 0's
    block of the 64) through GraphedFrameBatch -- ONE launch per kernel for all frames, the path
    ``bench.py --preset cfg4`` times -- with the energy, forces and distances of EVERY frame checked against the oracle."""
    ws = [workloads.ionic_box(seed=100 + f) for f in range(8)]
    boxes = [Box(w, torch.float64) for w in ws]
    batch = tpa.GraphedFrameBatch(boxes[0].calc, [(b.q, b.cell, b.pos, b.pairs, b.shifts) for b in boxes],
                                  store_distances=True)
    for replay in range(2):
        energies, forces = batch()
        torch.cuda.synchronize()
        for f, w in enumerate(ws):
            Eo, Fo, dist_o = _oracle_energy_forces(w)
            assert abs(float(energies[f]) - Eo) < 1e-11 * abs(Eo), (replay, f)
            assert rel(forces[f].cpu(), torch.tensor(Fo)) < 1e-10, (replay, f)
            if batch.distances[f] is not None:
                assert float((batch.distances[f].cpu() - torch.tensor(dist_o)).abs().max()) < 1e-12
    # the same frames, one HIP graph per frame on its own stream (bench.py --frame-batch streams)
    for f in (0, 7):
        b = boxes[f]
        Eo, Fo, _ = _oracle_energy_forces(ws[f])
        Eg, Fg = tpa.GraphedEnergyForces(b.calc, b.q, b.cell, b.pos, b.pairs, b.shifts)()
        assert abs(float(Eg) - Eo) < 1e-11 * abs(Eo) and rel(Fg.cpu(), torch.tensor(Fo)) < 1e-10


@pytest.mark.parametrize("cfg", ["ionic", "water", "dispersion"])
def test_fullsize_properties(cfg, request):
    w = request.getfixturevalue(cfg)
    dtype = torch.float64 if w.dtype == "f64" else torch.float32
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    box = Box(w, dtype)
    V = box.potentials()
    assert torch.isfinite(V).all()
    # charge conservation of the assignment: sum of the mesh == sum of the charges (reference test_mesh_interpolator.py)
    ns = ops.ns_mesh_from_cell(w.cell, w.mesh_spacing)
    assert ns == (w.n_mesh,) * 3
    mi = tpa.lib.MeshInterpolator(box.cell, ns, w.order, "P3M")
    mi.compute_weights(box.pos)
    mesh = mi.points_to_mesh(box.q)
    qsum, qabs = float(box.q.double().sum()), float(box.q.double().abs().sum())
    assert abs(float(mesh.double().sum()) - qsum) < (1e-12 if dtype == torch.float64 else 1e-6) * qabs
    # linearity in the charges
    q2 = torch.roll(box.q, 17, 0) * 0.37
    Vsum = box.potentials(q=box.q + q2)
    assert rel(Vsum, V + box.potentials(q=q2)) < tol
    assert rel(box.potentials(q=-2.5 * box.q), -2.5 * V) < tol
    # periodicity: translating the whole box by a lattice vector changes nothing but rounding (pair vectors are unchanged,
    # the mesh part is periodic); the mesh part alone is also invariant when only SOME atoms are replaced by their images
    ptol = 1e-9 if dtype == torch.float64 else 5e-4
    assert rel(box.potentials(pos=box.pos + (box.cell[0] - box.cell[2])), V) < ptol
    none = torch.zeros((0, 2), dtype=torch.int64, device=DEV)
    nod = torch.zeros((0,), dtype=dtype, device=DEV)
    moved = box.pos.clone()
    moved[::7] += box.cell[1] - 2 * box.cell[0]
    assert rel(box.calc(box.q, box.cell, moved, none, nod), box.calc(box.q, box.cell, box.pos, none, nod)) < ptol


@pytest.mark.parametrize("cfg", ["ionic", "water"])
def test_fullsize_forces(cfg, request):
    """Energy-mode forces == general-path forces == unfused-path forces; -dE/dr by central differences; fp32 vs fp64."""
    w = request.getfixturevalue(cfg)
    box = Box(w, torch.float64)
    E, F = box.energy_forces()
    Eg, Fg = box.energy_forces(general=True)
    assert abs(E - Eg) < 1e-11 * abs(E) and rel(F, Fg) < 1e-10
    ops.FUSE_DISTANCES = False
    try:
        Eu, Fu = box.energy_forces()
    finally:
        ops.FUSE_DISTANCES = True
    assert abs(E - Eu) < 1e-11 * abs(E) and rel(F, Fu) < 1e-10
    # the pair part conserves momentum exactly; the mesh part only up to the discretisation error
    sr = tpa.Calculator(tpa.CoulombPotential(smearing=None))
    p = box.pos.clone().requires_grad_(True)
    d = tpa.pair_distances(p, box.pairs, box.cell, box.shifts)
    tpa.weighted_sum(sr(box.q, box.cell, p, box.pairs, d), box.q).backward()
    assert float(p.grad.sum(0).abs().max()) < 1e-9 * float(p.grad.abs().max()) * 100
    assert float(F.sum(0).abs().max()) < 2e-2 * float(F.abs().mean()) * np.sqrt(w.n_atoms)
    # central differences on three coordinates
    h = 1e-4
    for atom, k in ((0, 0), (w.n_atoms // 2, 1), (w.n_atoms - 1, 2)):
        plus, minus = box.pos.clone(), box.pos.clone()
        plus[atom, k] += h
        minus[atom, k] -= h
        Ep = float(tpa.weighted_sum(box.potentials(plus), box.q))
        Em = float(tpa.weighted_sum(box.potentials(minus), box.q))
        fd = -(Ep - Em) / (2 * h)
        assert abs(fd - float(F[atom, k])) < 1e-5 * float(F.abs().max())
    # fp32 within the stated tolerance of fp64 (north_star: 1e-5 relative energy error)
    E32, F32 = Box(w, torch.float32).energy_forces()
    assert abs(E32 - E) < 1e-5 * abs(E)
    assert rel(F32.double(), F) < 1e-4


def test_cfg5_energy_forces(dispersion, golden_dir):
    """262 144 atoms, 39 M pairs, 128^3, 1/r^6: fp32 and fp64 energy + forces against the REFERENCE'S OWN fp64 evaluation of this box
    (ref_fullsize.npz, round 6; until then the fp32 path was compared with the fp64 path of the same library), the fp32 path against
    the fp64 path (all atoms), and the general-gradient route against the energy route.  fp32 force tolerance 8e-5 = five times the
    measured 1.58e-5 (the reference's own fp32 run: 1.59e-5)."""
    w = dispersion
    z = np.load(os.path.join(golden_dir, "ref_fullsize.npz"))
    Er, sample, Fr = float(z["dispersion_f64_energy"]), z["dispersion_sample"], torch.tensor(z["dispersion_f64_force_sample"])
    assert int(z["dispersion_n_pairs"]) == w.n_pairs
    E64, F64 = Box(w, torch.float64).energy_forces()
    E32, F32 = Box(w, torch.float32).energy_forces()
    assert abs(E64 - Er) < 1e-11 * abs(Er) and rel(F64.cpu()[sample], Fr) < 1e-10
    assert abs(E32 - Er) < 1e-5 * abs(Er) and rel(F32.double().cpu()[sample], Fr) < 8e-5
    assert abs(E32 - E64) < 1e-5 * abs(E64)
    assert rel(F32.double(), F64) < 8e-5
    Eg, Fg = Box(w, torch.float32).energy_forces(general=True)
    assert abs(Eg - E32) < 2e-6 * abs(E32) and rel(Fg, F32) < 8e-5


@pytest.mark.parametrize("cfg", ["ionic", "water", "dispersion"])
def test_fullsize_against_committed_oracle(cfg, request, golden_dir):
    """Every benchmark workload at FULL size against the pinned oracle's energy and forces (tests/golden/workloads.npz, made
    by tests/golden/make_workloads_golden.py: the 39 M-pair cfg5 evaluation is a minute of NumPy and tens of GB, so its result
    is committed instead of recomputed on the GPU box): fp64 -- energy 1e-11, 256 sampled forces 1e-9 of the largest, the
    whole-array checksums sum |F|^2 and sum r.F (seeded r) 1e-10 --, fp32 -- energy 1e-5, sampled forces rel-L2 1e-4 (cfg5's
    1/r^6 forces are small differences of large terms) --, through the eager calculators, the graph-replayed step and the
    device neighbour stream."""
    w = request.getfixturevalue(cfg)
    z = np.load(os.path.join(golden_dir, "workloads.npz"))
    g = {k[len(cfg) + 1:]: z[k] for k in z.files if k.startswith(cfg + "_")}
    chk = np.array([w.positions.sum(), (w.positions**2).sum(), w.charges.sum(), (w.charges**2).sum()])
    assert int(g["n_pairs"]) == w.n_pairs and np.allclose(chk, g["pos_checksum"], rtol=1e-13, atol=1e-9)
    Eo, sample, Fs = float(g["energy"]), g["sample"], torch.tensor(g["force_sample"])
    r = torch.tensor(np.random.default_rng(4242).normal(size=(w.n_atoms, 3)))
    tol32 = 1e-4 if cfg == "dispersion" else 2e-5  # (measured in round 4: 3.6e-6 on the water box; cfg5's forces are differences of large terms)
    for dtype, tol_e, tol_s, tol_c in ((torch.float64, 1e-11, 1e-9, 1e-10), (torch.float32, 1e-5, tol32, tol32)):
        box = Box(w, dtype)
        results = {"eager": box.energy_forces()}
        step = tpa.GraphedEnergyForces(box.calc, box.q, box.cell, box.pos, box.pairs, box.shifts)
        Eg, Fg = step()
        results["graph"] = (float(Eg), Fg.clone())
        del step
        st = tpa.GraphedEnergyForces(box.calc, box.q, box.cell, box.pos, neighbors=w.cutoff)
        Es, Fst = st()
        results["stream"] = (float(Es), Fst.clone())
        del st
        for name, (E, F) in results.items():
            F = F.cpu().double()
            assert abs(E - Eo) <= tol_e * abs(Eo), (cfg, dtype, name, E, Eo)
            assert float((F[sample] - Fs).abs().max()) <= tol_s * float(Fs.abs().max()) * (1 if dtype == torch.float64 else 10), (cfg, dtype, name)
            assert rel(F[sample], Fs) <= tol_s * (1 if dtype == torch.float64 else 1), (cfg, dtype, name, rel(F[sample], Fs))
            assert abs(float((F * F).sum()) - float(g["force_sq"])) <= tol_c * float(g["force_sq"]), (cfg, dtype, name)
            scale = float(F.norm() * r.norm())
            assert abs(float((r * F).sum()) - float(g["force_dot"])) <= tol_c * scale, (cfg, dtype, name)


@pytest.mark.parametrize("cfg", ["ionic", "water", "dispersion"])
def test_fullsize_against_the_reference_itself(cfg, request, golden_dir):
    """BASELINE.json configs[1] (8 000 ions, P3M n = 4, 32^3), configs[2] (the 31 944-atom water box, P3M n = 5, 64^3) and -- round 6
    -- configs[4] (262 144 atoms, InversePowerLawPotential(6), P3M n = 5, 128^3: potentials/inversepowerlaw.py:55-169) against
    the REFERENCE'S OWN evaluation of the same box (tests/golden/ref_fullsize.npz, made by importing torchpme in the build
    container: tests/golden/make_reference_fullsize.py) -- the whole first-order contract of one energy step: E, F (256 sampled
    atoms + the two whole-array checksums), dE/dq (sample + checksum), dE/dcell, from the graph-replayed step and the eager
    calculators; and the three gradients of the tuner's V.sum() protocol.  fp64 against the reference's fp64 numbers at 1e-10;
    fp32 against the same fp64 numbers at five times the errors measured in round 4 (energy 1e-5 = north_star's tolerance,
    forces rel-L2 2e-5, dE/dq 2e-5, dE/dcell 6e-5) -- the reference's own fp32 run differs from its fp64 run by 7e-6 in the
    sampled forces and 1.6e-5 in dE/dcell on the water box.  cfg5's fp32 tolerances are five times the errors measured in round 6
    (profiles/r06_c_fullsize_errors.txt: forces 1.58e-5 -- the reference's OWN fp32 run is 1.59e-5 from its fp64 run there --, dE/dq
    1.35e-5): 8e-5 / 7e-5."""
    w = request.getfixturevalue(cfg)
    z = np.load(os.path.join(golden_dir, "ref_fullsize.npz"))
    g = {k[len(cfg) + 5:]: z[k] for k in z.files if k.startswith(cfg + "_f64_")}
    chk = np.array([w.positions.sum(), (w.positions**2).sum(), w.charges.sum(), (w.charges**2).sum()])
    assert int(z[f"{cfg}_n_pairs"]) == w.n_pairs and np.allclose(chk, z[f"{cfg}_pos_checksum"], rtol=1e-13, atol=1e-9)
    sample = z[f"{cfg}_sample"]
    rng = np.random.default_rng(4242)
    r_vec = rng.normal(size=(w.n_atoms, 3))
    s_vec = rng.normal(size=(w.n_atoms, 1))
    relmax = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())  # noqa: E731
    rell2 = lambda a, b: float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))  # noqa: E731
    f32 = (torch.float32, 1e-5, 8e-5, 7e-5, 6e-5) if cfg == "dispersion" else (torch.float32, 1e-5, 2e-5, 2e-5, 6e-5)
    for dtype, tol_e, tol_f, tol_q, tol_c in ((torch.float64, 1e-11, 1e-10, 1e-10, 1e-10), f32):
        box = Box(w, dtype)
        results = {}
        step = tpa.GraphedEnergyForces(box.calc, box.q, box.cell, box.pos, box.pairs, box.shifts, charge_gradient=True,
                                       cell_gradient=True)
        results["graph"] = tuple(x.clone() for x in step())
        del step
        p_, q_, c_ = box.pos.clone().requires_grad_(True), box.q.clone().requires_grad_(True), box.cell.clone().requires_grad_(True)
        d = tpa.pair_distances(p_, box.pairs, c_, box.shifts)
        E = (box.calc(q_, c_, p_, box.pairs, d) * q_).sum()  # the reference's own call sequence, eager
        E.backward()
        results["eager"] = (E.detach(), -p_.grad, q_.grad, c_.grad)
        for name, (E, F, dq, dc) in results.items():
            F, dq, dc = F.cpu().double().numpy(), dq.cpu().double().numpy(), dc.cpu().double().numpy()
            tag = (cfg, dtype, name)
            assert abs(float(E) - float(g["energy"])) <= tol_e * abs(float(g["energy"])), tag
            assert rell2(F[sample], g["force_sample"]) <= tol_f, (tag, rell2(F[sample], g["force_sample"]))
            assert abs(float((F * F).sum()) - float(g["force_sq"])) <= 10 * tol_f * float(g["force_sq"]), tag
            assert abs(float((r_vec * F).sum()) - float(g["force_dot"])) <= tol_f * np.linalg.norm(F) * np.linalg.norm(r_vec), tag
            assert relmax(dq[sample, 0], g["charge_grad_sample"]) <= tol_q, (tag, relmax(dq[sample, 0], g["charge_grad_sample"]))
            assert abs(float((s_vec * dq).sum()) - float(g["charge_grad_dot"])) <= tol_q * np.linalg.norm(s_vec) * np.linalg.norm(dq), tag
            assert rell2(dc, g["cell_grad"]) <= tol_c, (tag, rell2(dc, g["cell_grad"]))
        # tuning/tuner.py:350-369 on this box: result.sum().backward() with constant distances
        d_fixed = tpa.pair_distances(box.pos, box.pairs, box.cell, box.shifts).detach().clone()
        positions, cl, charges = box.pos.clone(), box.cell.clone(), box.q.clone()
        for x in (positions, cl, charges):
            x.requires_grad_(True)
        V = box.calc.forward(positions=positions, charges=charges, cell=cl, neighbor_indices=box.pairs, neighbor_distances=d_fixed)
        V.sum().backward()
        Vn = V.detach().cpu().double().numpy()
        assert relmax(Vn[sample, 0], g["potential_sample"]) <= tol_q, (cfg, dtype)
        assert abs(float((s_vec * Vn).sum()) - float(g["potential_dot"])) <= tol_q * np.linalg.norm(s_vec) * np.linalg.norm(Vn)
        assert rell2(positions.grad.cpu().double().numpy()[sample], g["sumseed_pos_sample"]) <= 10 * tol_f, (cfg, dtype)
        assert relmax(charges.grad.cpu().double().numpy()[sample, 0], g["sumseed_charge_sample"]) <= 10 * tol_q, (cfg, dtype)
        assert rell2(cl.grad.cpu().double().numpy(), g["sumseed_cell"]) <= tol_c, (cfg, dtype)


def test_eight_headline_frames_in_one_batch(golden_dir):
    """Eight cfg3-size frames (31 944-atom water boxes, seeds 1234 ... 1241, fp32) evaluated together -- GraphedFrameBatch: one
    launch per kernel for all frames; and one replayed graph per frame on its own stream -- each frame's energy and forces
    against the pinned oracle's numbers for that frame (tests/golden/frames_water.npz, made by make_frames_golden.py): energy
    1e-5, 256 sampled forces rel-L2 1e-4, sum |F|^2 1e-4.  bench.py's `frames` block times exactly these two forms."""
    z = np.load(os.path.join(golden_dir, "frames_water.npz"))
    n = len(z["seeds"])
    ws = [workloads.water_box(seed=int(sd)) for sd in z["seeds"]]
    boxes = [Box(w, torch.float32) for w in ws]
    for f, w in enumerate(ws):
        assert int(z[f"f{f}_n_pairs"]) == w.n_pairs
        assert np.allclose([w.positions.sum(), (w.positions**2).sum()], z[f"f{f}_pos_checksum"], rtol=1e-13)
    batch = tpa.GraphedFrameBatch(boxes[0].calc, [(b.q, b.cell, b.pos, b.pairs, b.shifts) for b in boxes])
    for _ in range(2):
        energies, forces = batch()
    torch.cuda.synchronize()
    graphs = [tpa.GraphedEnergyForces(b.calc, b.q, b.cell, b.pos, b.pairs, b.shifts) for b in boxes]
    streams = [torch.cuda.Stream() for _ in boxes]
    for g, st in zip(graphs, streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            g.graph.replay()
    torch.cuda.synchronize()
    for f in range(n):
        Eo, sample, Fs = float(z[f"f{f}_energy"]), z[f"f{f}_sample"], torch.tensor(z[f"f{f}_force_sample"])
        for name, E, F in (("batch", energies[f], forces[f]), ("streams", graphs[f].energy, graphs[f].forces)):
            F = F.cpu().double()
            assert abs(float(E) - Eo) <= 1e-5 * abs(Eo), (f, name, float(E), Eo)
            assert rel(F[sample], Fs) <= 1e-4, (f, name)
            assert abs(float((F * F).sum()) - float(z[f"f{f}_force_sq"])) <= 1e-4 * float(z[f"f{f}_force_sq"]), (f, name)


def test_nve_energy_conservation():
    """examples/nve_ions.py: 1 728 charged soft spheres (Coulomb + 1/r^6, two graphed calculators, device neighbour list),
    200 velocity-Verlet steps -- the total energy is conserved to a small fraction of the kinetic energy, and halving the
    time step cuts the fluctuation ~4x (forces are the exact gradient of the energy the integrator sees)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import nve_ions

    hist, info = nve_ions.run(n_side=12, steps=200, dt_fs=2.0)
    e_tot, e_kin = hist.sum(1), hist[:, 1].mean()
    dev2 = np.abs(e_tot - e_tot[0]).max()
    assert np.isfinite(hist).all() and e_kin > 10.0
    assert dev2 < 6e-3 * e_kin  # measured 3.5e-3 (0.27 eV on 76 eV) at 2 fs
    hist1, _ = nve_ions.run(n_side=12, steps=400, dt_fs=1.0)
    dev1 = np.abs(hist1.sum(1) - hist1.sum(1)[0]).max()
    assert dev1 < 0.35 * dev2  # second-order integrator: measured 0.25

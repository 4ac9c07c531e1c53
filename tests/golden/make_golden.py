"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (build container only).

    python tests/golden/make_golden.py

The reference (``/root/reference``) is a pure-Python package; it cannot travel to the GPU box, so
its *outputs* on fixed seeded inputs are committed here as small ``.npz`` fixtures (data only:
inputs + expected outputs).  Nothing in tests/ or the product reads /root/reference at run time.

Fixtures written:
  crystals.npz       analytic crystals + literature Madelung constants (tests/helpers.py:19-237)
  gromacs_frames.npz the two 8-ion frames with GROMACS SPME energy/forces/stress
                     (examples/coulomb_test_frames.xyz)
  ref_small.npz      reference potentials + autograd gradients on a 7-atom triclinic cell for every
                     scheme/order/potential combination on the path (fp64)
  ref_medium.npz     512-atom jittered lattice, cutoff list, P3M n=5 / PME n=4, fp64 and fp32
                     reference potentials, energies and forces
  conventions.npz    stencil / weight / k-grid / filter known answers
  direct.npz         exact direct-sum molecules (tests/calculators/test_values_direct.py)
  ref_ewald.npz      EwaldCalculator potentials + autograd gradients (7-atom triclinic cell): Coulomb and 1/r^p, several
                     channels, 2-D periodic slab, user-supplied k-vectors, node mask, full list
  tuning.npz         a-priori error estimates (tuning/p3m.py, tuning/pme.py) and smearing estimates (tuning/tuner.py)
                     on two structures for a grid of (smearing, mesh_spacing, cutoff, nodes)
"""

import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# ---- import the reference with the two stubs it needs in this checkout ----
_v = types.ModuleType("torchpme._version")
_v.__version__ = "0.0.0"
_v.__version_tuple__ = (0, 0, 0)
sys.modules["torchpme._version"] = _v
_ves = types.ModuleType("vesin")
_ves.NeighborList = object
sys.modules["vesin"] = _ves
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "tests"))
sys.path.insert(0, ROOT)
import helpers as ref_helpers  # noqa: E402  (reference tests/helpers.py, crystals only)
import torchpme  # noqa: E402

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("_nl", os.path.join(ROOT, "torch-pme_amd", "neighbors.py"))
_nl = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_nl)
neighbor_list = _nl.neighbor_list


def t(x, dtype=torch.float64, grad=False):
    out = torch.tensor(np.asarray(x), dtype=dtype)
    out.requires_grad_(grad)
    return out


def ref_distances(pos, cell, pairs, shifts):
    vec = pos[pairs[:, 1]] - pos[pairs[:, 0]] + shifts.to(cell.dtype) @ cell
    return torch.linalg.norm(vec, dim=1)


# --------------------------------------------------------------------------------------
def make_crystals():
    names = [
        "CsCl", "NaCl_primitive", "NaCl_cubic", "zincblende", "wurtzite", "fluorite", "cu2o",
        "wigner_sc", "wigner_bcc", "wigner_bcc_cubiccell", "wigner_fcc", "wigner_fcc_cubiccell",
    ]
    out = {"names": np.array(names)}
    for nm in names:
        pos, q, cell, mad, nfu = ref_helpers.define_crystal(nm, dtype=torch.float64)
        out[f"{nm}/positions"] = pos.numpy()
        out[f"{nm}/charges"] = q.numpy()
        out[f"{nm}/cell"] = cell.numpy()
        out[f"{nm}/madelung"] = np.float64(mad.item())
        out[f"{nm}/n_formula"] = np.int64(nfu)
    np.savez(os.path.join(HERE, "crystals.npz"), **out)


def parse_xyz(path):
    frames = []
    with open(path) as f:
        lines = f.read().splitlines()
    k = 0
    while k < len(lines) and lines[k].strip():
        n = int(lines[k])
        hdr = lines[k + 1]

        def field(key):
            a = hdr.index(key + "=") + len(key) + 1
            if hdr[a] == '"':
                b = hdr.index('"', a + 1)
                return hdr[a + 1 : b]
            b = hdr.find(" ", a)
            return hdr[a : b if b > 0 else None]

        cell = np.array(field("Lattice").split(), dtype=np.float64).reshape(3, 3)
        energy = float(field("energy"))
        stress = np.array(field("stress").split(), dtype=np.float64).reshape(3, 3)
        rows = [ln.split() for ln in lines[k + 2 : k + 2 + n]]
        pos = np.array([[float(v) for v in r[1:4]] for r in rows])
        q = np.array([float(r[4]) for r in rows])
        frc = np.array([[float(v) for v in r[5:8]] for r in rows])
        frames.append(dict(cell=cell, energy=energy, stress=stress, positions=pos, charges=q, forces=frc))
        k += 2 + n
    return frames


def make_gromacs():
    frames = parse_xyz(os.path.join(REF, "examples", "coulomb_test_frames.xyz"))
    out = {"n_frames": np.int64(len(frames)), "prefactor_eV_A": np.float64(torchpme.prefactors.eV_A)}
    for k, fr in enumerate(frames):
        for key, val in fr.items():
            out[f"{k}/{key}"] = np.asarray(val)
        # reference PME / P3M results with the reference test's settings (rc=5.54, sigma=rc/6, h=sigma/8)
        rc = 5.54
        sm = rc / 6
        pairs, S, _ = neighbor_list(fr["positions"], fr["cell"], rc)
        out[f"{k}/n_half_pairs"] = np.int64(len(pairs))
        for name, Calc in (("pme", torchpme.PMECalculator), ("p3m", torchpme.P3MCalculator)):
            pos = t(fr["positions"], grad=True)
            cell = t(fr["cell"])
            q = t(fr["charges"]).reshape(-1, 1)
            calc = Calc(torchpme.CoulombPotential(smearing=sm, prefactor=torchpme.prefactors.eV_A), mesh_spacing=sm / 8)
            d = ref_distances(pos, cell, torch.tensor(pairs), torch.tensor(S))
            V = calc(q, cell, pos, torch.tensor(pairs), d)
            E = (V * q).sum()
            (F,) = torch.autograd.grad(-E, pos)
            out[f"{k}/{name}/energy"] = np.float64(E.item())
            out[f"{k}/{name}/forces"] = F.numpy()
    np.savez(os.path.join(HERE, "gromacs_frames.npz"), **out)


def make_ref_small():
    rng = np.random.default_rng(20260928)
    cell = np.array([[4, 0, 0], [0.5, 5, 0], [0.3, -0.4, 6]], dtype=np.float64)
    N, P = 7, 14
    out = {"cell": cell}
    cases = []
    for scheme, orders in (("P3M", [1, 2, 3, 4, 5]), ("Lagrange", [3, 4, 5, 6, 7])):
        for o in orders:
            cases.append((scheme, o, "coulomb", 1, 1, None, None, False))
    for p in range(1, 7):
        cases.append(("P3M", 5, "ipl", p, 1, None, None, False))
        cases.append(("Lagrange", 4, "ipl", p, 2, None, None, True))
    cases.append(("P3M", 4, "coulomb", 1, 2, [True, False, True], None, False))
    cases.append(("Lagrange", 4, "coulomb", 1, 2, [True, True, False], None, True))
    cases.append(("P3M", 5, "coulomb", 1, 1, None, 2.0, False))
    cases.append(("P3M", 3, "ipl", 6, 3, None, 2.0, False))
    names = []
    for ci, (scheme, order, kind, p, C, periodic, excl, full) in enumerate(cases):
        pos = rng.uniform(-2, 7, (N, 3))
        q = rng.normal(size=(N, C))
        pairs = rng.integers(0, N, (P, 2))
        pairs[:, 1] = (pairs[:, 0] + 1 + rng.integers(0, N - 1, P)) % N
        dist = rng.uniform(0.8, 3.0, P)
        g = rng.normal(size=(N, C))
        sm, pref, h = 1.0, 1.3, 1.0
        if kind == "coulomb":
            pot = torchpme.CoulombPotential(smearing=sm, prefactor=pref, exclusion_radius=excl)
        else:
            pot = torchpme.InversePowerLawPotential(exponent=p, smearing=sm, prefactor=pref, exclusion_radius=excl)
        Calc = torchpme.P3MCalculator if scheme == "P3M" else torchpme.PMECalculator
        calc = Calc(pot, mesh_spacing=h, interpolation_nodes=order, full_neighbor_list=full)
        tq, tc, tp, td = t(q, grad=True), t(cell, grad=True), t(pos, grad=True), t(dist, grad=True)
        per = None if periodic is None else torch.tensor(periodic)
        V = calc(tq, tc, tp, torch.tensor(pairs), td, periodic=per)
        (V * t(g)).sum().backward()
        nm = f"c{ci:02d}"
        names.append(nm)
        meta = dict(scheme=scheme, order=order, kind=kind, exponent=p, smearing=sm, prefactor=pref, mesh_spacing=h,
                    periodic=periodic, exclusion_radius=excl, full_list=full)
        out[f"{nm}/meta"] = np.array(repr(meta))
        for key, val in dict(positions=pos, charges=q, pairs=pairs, dist=dist, g=g, V=V.detach().numpy(),
                             grad_charges=tq.grad.numpy(), grad_positions=tp.grad.numpy(), grad_cell=tc.grad.numpy(),
                             grad_dist=td.grad.numpy()).items():
            out[f"{nm}/{key}"] = val
    out["names"] = np.array(names)
    np.savez(os.path.join(HERE, "ref_small.npz"), **out)


def jittered_lattice(n_side, a, jitter, rng):
    g = np.arange(n_side) * a
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3) + 0.5 * a
    return pos + rng.uniform(-jitter, jitter, pos.shape)


def make_ref_medium():
    rng = np.random.default_rng(77)
    n_side, a = 8, 2.1544  # rho ~ 0.1 / A^3
    L = n_side * a
    pos = jittered_lattice(n_side, a, 0.4, rng)
    N = len(pos)
    q = rng.normal(size=(N, 1))
    q -= q.mean()
    cell = L * np.eye(3)
    rc = 6.0
    sm = rc / 5
    pairs, S, dist = neighbor_list(pos, cell, rc)
    out = dict(positions=pos, charges=q, cell=cell, cutoff=np.float64(rc), smearing=np.float64(sm), pairs=pairs, shifts=S)
    for name, Calc, order, nmesh in (("p3m5", torchpme.P3MCalculator, 5, 32), ("pme4", torchpme.PMECalculator, 4, 32)):
        h = 2 * L / (nmesh - 2)
        out[f"{name}/mesh_spacing"] = np.float64(h)
        out[f"{name}/order"] = np.int64(order)
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            tp = t(pos, dt, grad=True)
            tc = t(cell, dt, grad=True)
            tq = t(q, dt, grad=True)
            calc = Calc(torchpme.CoulombPotential(smearing=sm), mesh_spacing=h, interpolation_nodes=order).to(dt)
            d = ref_distances(tp, tc, torch.tensor(pairs), torch.tensor(S))
            V = calc(tq, tc, tp, torch.tensor(pairs), d)
            E = (V * tq).sum()
            E.backward()
            assert tuple(calc.mesh_interpolator.ns_mesh.tolist()) == (nmesh,) * 3
            out[f"{name}/{tag}/V"] = V.detach().numpy()
            out[f"{name}/{tag}/energy"] = np.float64(E.item())
            out[f"{name}/{tag}/grad_positions"] = tp.grad.numpy()
            out[f"{name}/{tag}/grad_cell"] = tc.grad.numpy()
            out[f"{name}/{tag}/grad_charges"] = tq.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_medium.npz"), **out)


def make_conventions():
    cell = t([[4, 0, 0], [0.5, 5, 0], [0.3, -0.4, 6]])
    pos = t([[0.1, 0.2, 0.3], [2.0, 3.0, 4.0], [-11.5, 1.0, 2.0]])
    ns = torch.tensor([8, 8, 16])
    out = dict(cell=cell.numpy(), positions=pos.numpy(), ns=ns.numpy())
    for scheme, orders in (("P3M", [1, 2, 3, 4, 5]), ("Lagrange", [3, 4, 5, 6, 7])):
        for o in orders:
            mi = torchpme.lib.MeshInterpolator(cell, ns, o, scheme)
            mi.compute_weights(pos)
            out[f"{scheme}{o}/weights"] = mi.interpolation_weights.numpy()  # (n, N, 3)
            out[f"{scheme}{o}/x_indices"] = mi.x_indices.numpy()
            out[f"{scheme}{o}/y_indices"] = mi.y_indices.numpy()
            out[f"{scheme}{o}/z_indices"] = mi.z_indices.numpy()
            xs = torch.linspace(-0.5, 0.5, 21, dtype=torch.float64).reshape(-1, 1)
            out[f"{scheme}{o}/w_of_x"] = mi._compute_1d_weights(xs).numpy()[:, :, 0]
    out["x_grid"] = np.linspace(-0.5, 0.5, 21)
    kv = torchpme.lib.generate_kvectors_for_mesh(cell, ns)
    out["kvectors"] = kv.numpy()
    pot = torchpme.CoulombPotential(smearing=1.0)
    out["G_pme"] = torchpme.lib.KSpaceFilter(cell, ns, pot, "backward", "forward")._kfilter.numpy()
    for o in (1, 2, 3, 4, 5):
        out[f"G_p3m{o}"] = torchpme.lib.P3MKSpaceFilter(cell, ns, o, pot, "backward", "forward", 0, 2)._kfilter.numpy()
    for p in range(1, 7):
        ipl = torchpme.InversePowerLawPotential(exponent=p, smearing=0.8, prefactor=1.7)
        out[f"G_ipl{p}"] = torchpme.lib.P3MKSpaceFilter(cell, ns, 4, ipl, "backward", "forward", 0, 2)._kfilter.numpy()
        dd = torch.linspace(0.3, 6.0, 40, dtype=torch.float64)
        out[f"sr_ipl{p}"] = ipl.sr_from_dist(dd).numpy()
        out[f"lr_ipl{p}"] = ipl.lr_from_dist(dd).numpy()
        out[f"self_ipl{p}"] = np.float64(ipl.self_contribution().item())
        out[f"bg_ipl{p}"] = np.float64(ipl.background_correction().item())
    out["d_grid"] = np.linspace(0.3, 6.0, 40)
    for spacing in (0.3, 0.5, 1.0, 1.7):
        out[f"ns_mesh_{spacing}"] = torchpme.lib.get_ns_mesh(cell, spacing).numpy()
    np.savez_compressed(os.path.join(HERE, "conventions.npz"), **out)


def make_direct():
    """Exact direct sums: V_i = 1/2 sum_j q_j / r_ij for small molecules (smearing=None path)."""
    mols = {
        "dimer": np.array([[0.0, 0, 0], [0, 0, 1.0]]),
        "triangle": np.array([[0.0, 0, 0], [1.0, 0, 0], [0.5, math.sqrt(3) / 2, 0]]),
        "square": np.array([[1.0, 1, 0], [1, -1, 0], [-1, 1, 0], [-1, -1, 0]]) / 2,
        "tetrahedron": np.array([[0, 0, 0], [1, 0, 0], [0.5, math.sqrt(3) / 2, 0], [0.5, math.sqrt(3) / 6, math.sqrt(2 / 3)]]),
    }
    rng = np.random.default_rng(5)
    out = {"names": np.array(list(mols))}
    for nm, pos in mols.items():
        N = len(pos)
        q = rng.normal(size=(N, 2))
        ii, jj = np.triu_indices(N, 1)
        pairs = np.stack([ii, jj], axis=1)
        d = np.linalg.norm(pos[jj] - pos[ii], axis=1)
        for p in (1, 3, 6):
            pot = torchpme.CoulombPotential() if p == 1 else torchpme.InversePowerLawPotential(exponent=p)
            calc = torchpme.Calculator(pot)
            V = calc(t(q), t(np.eye(3)), t(pos), torch.tensor(pairs), t(d))
            out[f"{nm}/V_p{p}"] = V.numpy()
        pot = torchpme.CoulombPotential(exclusion_radius=1.2, exclusion_degree=2)
        out[f"{nm}/V_excl"] = torchpme.Calculator(pot)(t(q), t(np.eye(3)), t(pos), torch.tensor(pairs), t(d)).numpy()
        out[f"{nm}/positions"], out[f"{nm}/charges"], out[f"{nm}/pairs"], out[f"{nm}/dist"] = pos, q, pairs, d
    np.savez(os.path.join(HERE, "direct.npz"), **out)


def make_ref_ewald():
    from torchpme.lib import generate_kvectors_for_ewald

    rng = np.random.default_rng(77)
    cell = np.array([[4, 0, 0], [0.5, 5, 0], [0.3, -0.4, 6]], dtype=np.float64)
    N, P = 7, 14
    out = {"cell": cell}
    #        kind       p  C  periodic             full   lr_wl  own_k  node_mask
    cases = [("coulomb", 1, 1, None, False, 1.3, False, False),
             ("coulomb", 1, 2, None, True, 0.9, False, False),
             ("coulomb", 1, 1, [True, True, False], False, 1.1, False, False),
             ("coulomb", 1, 1, [False, True, True], False, 1.1, False, True),
             ("coulomb", 1, 1, None, False, 1.0, True, False),
             ("ipl", 1, 1, None, False, 1.2, False, False),
             ("ipl", 2, 1, None, False, 1.2, False, False),
             ("ipl", 3, 2, None, False, 1.2, False, False),
             ("ipl", 4, 1, None, False, 1.2, False, False),
             ("ipl", 5, 1, None, False, 1.2, False, False),
             ("ipl", 6, 1, None, True, 1.2, False, True)]
    names = []
    for ci, (kind, p, C, periodic, full, wl, own_k, use_mask) in enumerate(cases):
        pos = rng.uniform(-2, 7, (N, 3))
        q = rng.normal(size=(N, C))
        pairs = rng.integers(0, N, (P, 2))
        pairs[:, 1] = (pairs[:, 0] + 1 + rng.integers(0, N - 1, P)) % N
        dist = rng.uniform(0.8, 3.0, P)
        g = rng.normal(size=(N, C))
        sm, pref = 0.9, 1.3
        pot = (torchpme.CoulombPotential(smearing=sm, prefactor=pref) if kind == "coulomb"
               else torchpme.InversePowerLawPotential(exponent=p, smearing=sm, prefactor=pref))
        calc = torchpme.EwaldCalculator(pot, lr_wavelength=wl, full_neighbor_list=full)
        tq, tc, tp, td = t(q, grad=True), t(cell, grad=True), t(pos, grad=True), t(dist, grad=True)
        per = None if periodic is None else torch.tensor(periodic)
        kv = None
        if own_k:  # a caller-supplied set: the vectors of a coarser grid, zero padded (lib/kvectors.py:139-166)
            kv0 = generate_kvectors_for_ewald(ns=torch.tensor([3, 4, 5]), cell=t(cell))
            kv = torch.cat([kv0, torch.zeros((5, 3), dtype=torch.float64)])
            out[f"e{ci:02d}/kvectors"] = kv.numpy()
        mask = None
        if use_mask:
            mask = torch.tensor(rng.uniform(size=N) > 0.3)
            out[f"e{ci:02d}/node_mask"] = mask.numpy()
        V = calc(tq, tc, tp, torch.tensor(pairs), td, periodic=per, kvectors=kv, node_mask=mask)
        (V * t(g)).sum().backward()
        nm = f"e{ci:02d}"
        names.append(nm)
        meta = dict(kind=kind, exponent=p, smearing=sm, prefactor=pref, lr_wavelength=wl, periodic=periodic,
                    full_list=full, own_kvectors=own_k, node_mask=use_mask)
        out[f"{nm}/meta"] = np.array(repr(meta))
        for key, val in dict(positions=pos, charges=q, pairs=pairs, dist=dist, g=g, V=V.detach().numpy(),
                             grad_charges=tq.grad.numpy(), grad_positions=tp.grad.numpy(),
                             grad_cell=None if own_k else tc.grad.numpy(), grad_dist=td.grad.numpy()).items():
            if val is not None:
                out[f"{nm}/{key}"] = val
        if own_k:  # with caller-supplied k-vectors the cell only enters through the volume
            out[f"{nm}/grad_cell"] = tc.grad.numpy()
    out["names"] = np.array(names)
    np.savez(os.path.join(HERE, "ref_ewald.npz"), **out)


def make_tuning():
    from torchpme.tuning.p3m import P3MErrorBounds
    from torchpme.tuning.pme import PMEErrorBounds
    from torchpme.tuning.tuner import TunerBase

    rng = np.random.default_rng(42)
    structures = {
        "pair": (np.array([[1.0], [-1.0]]), np.eye(3), np.array([[0.0, 0.0, 0.0], [0.4, 0.4, 0.4]])),
        "tri": (rng.normal(size=(11, 1)), np.array([[6.0, 0, 0], [0.8, 7.0, 0], [-0.4, 0.5, 9.0]]),
                rng.uniform(0, 6, (11, 3))),
    }
    out = {}
    for name, (q, cell, pos) in structures.items():
        out[f"{name}/charges"], out[f"{name}/cell"], out[f"{name}/positions"] = q, cell, pos
        tq, tc, tp = t(q), t(cell), t(pos)
        rows = []
        for smearing, h, rc in ((1.0, 0.5, 4.4), (0.7, 0.3, 3.0), (1.6, 0.9, 6.5)):
            for nodes in range(1, 8):
                p3m = float(P3MErrorBounds(tq, tc, tp)(smearing=smearing, mesh_spacing=h, cutoff=rc, interpolation_nodes=nodes)) if nodes <= 7 else np.nan
                pme = float(PMEErrorBounds(tq, tc, tp)(smearing=smearing, mesh_spacing=h, cutoff=rc, interpolation_nodes=nodes)) if nodes >= 3 else np.nan
                rows.append((smearing, h, rc, nodes, p3m, pme))
        out[f"{name}/bounds"] = np.array(rows)  # smearing, mesh_spacing, cutoff, nodes, P3M estimate, PME estimate
        out[f"{name}/smearing"] = np.array([[rc, acc, TunerBase(tq, tc, tp, rc, None).estimate_smearing(acc)]
                                             for rc in (3.0, 4.4, 6.0) for acc in (1e-1, 1e-3, 1e-6)])
    np.savez(os.path.join(HERE, "tuning.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] in ("tuning", "ewald"):
        {"tuning": make_tuning, "ewald": make_ref_ewald}[sys.argv[1]]()
        sys.exit(0)
    make_tuning()
    make_ref_ewald()
    make_crystals()
    make_gromacs()
    make_ref_small()
    make_ref_medium()
    make_conventions()
    make_direct()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))

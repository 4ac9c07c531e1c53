"""``Calculator.forward`` from primitives that are closed under differentiation: exact derivatives of any order.

The reference is a chain of ATen ops (``calculators/calculator.py:43-87,103-189``, ``calculators/pme.py:88-143``,
``lib/mesh_interpolator.py:303-457``, ``lib/kspace_filter.py:122-197``), so ``create_graph=True`` works there: a loss on forces
with learned charges, Hessian-vector products, the stress of a force loss.  The fused HIP kernels behind the calculators are
first order.  ``calculator.double_backward = "analytic"`` routes a call through this module instead: the same mathematics, cut
into six linear maps whose backward passes are made of the same six maps (``csrc/jets.hip``, ``mipme_convolve``,
``mipme_fft_r2c``), glued with tensor expressions for everything that is small or elementwise (fractional coordinates, the pair
potential ``sr_from_dist``, the filter table as a function of the cell, self / background / slab terms) -- so PyTorch
differentiates the composition as often as it is asked to, w.r.t. charges, positions, cell and distances alike.

    spread(u, x; k)   mesh = sum_i x_i D^k W_i        d/du_d -> x * gather(., k + e_d)      d/dx   -> gather(., k)
    gather(u, phi; k) out_i = sum_m phi_m D^k W_i(m)   d/du_d -> g * gather(phi, k + e_d)    d/dphi -> spread(g, k)
    convolve(mesh, G) irfftn(G rfftn(mesh))            d/dmesh -> convolve(g, G)             d/dG   -> spectral_dot(mesh, g)
    spectral_dot(a,b) mu Re(a^ conj b^) per k          d/da   -> convolve(b, c)              d/db   -> convolve(a, c)
    pair_sum(w, x)    out_i = sum_p w_p x_j (+ j<-i)    d/dw   -> pair_dot(g, x)              d/dx   -> pair_sum(w, g) (transposed)
    pair_dot(a, b)    out_p = a_i . b_j (+ a_j . b_i)   d/da   -> pair_sum(c, b)              d/db   -> pair_sum(c, a) (transposed)

and, for the caller-side distance helper under ``create_graph=True`` (``recorded_distance_backward``), the pair difference
``x_j - x_i`` and its adjoint, the scatter, each the other's derivative.

The interpolation weights are piecewise polynomials of degree ``interpolation_nodes - 1``; the kernels provide their derivatives
up to third order per axis (enough for a double backward of a force loss and one order to spare), beyond that a call raises.
Cost: a dozen launches per evaluation instead of six, atomics in the spread and the pair sum -- this is the route for training
on forces and for Hessians, not for molecular dynamics.  Mesh calculators (PME, P3M), the plain pair sum, and the explicit Ewald
sum -- whose reciprocal part is written here as the reference writes it, (K, N) phase tables in tensor expressions.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib, lib, ops

_MAX_ORDER = 3
#: pair sums through the transposed list (no atomics) for lists of at least ROWS_MIN_PAIRS pairs; False: the atomic kernel
ROWS = True
ROWS_MIN_PAIRS = 4096


def _bump(k, d):
    out = list(k)
    out[d] += 1
    if out[d] > _MAX_ORDER:
        raise RuntimeError(
            "torchpme_amd: derivative of the interpolation weights beyond third order requested "
            '(`double_backward = "analytic"` covers up to the third derivative of the potentials w.r.t. positions / cell)')
    return tuple(out)


def _stream(t):
    return _lib.current_stream(t.device)


class _Spread(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, x, geom, k):
        uc, xc = u.detach().contiguous(), x.detach().contiguous()
        n_ch = xc.shape[1]
        mesh = torch.empty((n_ch, *geom.ns), dtype=xc.dtype, device=xc.device)
        md = geom.desc(n_ch)
        with _lib.on_device(xc.device):
            _lib.check(_lib.load().mipme_spread_jet(_stream(xc), _lib.dtype_code(xc.dtype), C.byref(md), xc.shape[0],
                                                    uc.data_ptr(), xc.data_ptr(), k[0], k[1], k[2], mesh.data_ptr()))
        ctx.save_for_backward(u, x)
        ctx.geom, ctx.k = geom, k
        return mesh

    @staticmethod
    def backward(ctx, g):
        u, x = ctx.saved_tensors
        geom, k = ctx.geom, ctx.k
        gu = gx = None
        if ctx.needs_input_grad[0]:
            gu = (x.unsqueeze(-1) * _GatherGrad.apply(u, g, geom, k)).sum(dim=1)
        if ctx.needs_input_grad[1]:
            gx = _Gather.apply(u, g, geom, k)
        return gu, gx, None, None


class _GatherGrad(torch.autograd.Function):
    """``out[i, c, d] = gather(u, phi; k + e_d)[i, c]`` for d = x, y, z in one launch: what the derivatives of spread and
    gather with respect to ``u`` are made of.  Its own derivatives are the same primitives one order up."""

    @staticmethod
    def forward(ctx, u, phi, geom, k):
        for d in range(3):
            _bump(k, d)  # (raises beyond third order)
        uc, pc = u.detach().contiguous(), phi.detach().contiguous()
        n_ch, n = pc.shape[0], uc.shape[0]
        out = torch.empty((n, n_ch, 3), dtype=pc.dtype, device=pc.device)
        md = geom.desc(n_ch)
        with _lib.on_device(pc.device):
            _lib.check(_lib.load().mipme_gather_jet3(_stream(pc), _lib.dtype_code(pc.dtype), C.byref(md), n, uc.data_ptr(),
                                                     pc.data_ptr(), k[0], k[1], k[2], out.data_ptr()))
        ctx.save_for_backward(u, phi)
        ctx.geom, ctx.k = geom, k
        return out

    @staticmethod
    def backward(ctx, g):  # g: (N, C, 3)
        u, phi = ctx.saved_tensors
        geom, k = ctx.geom, ctx.k
        gu = gphi = None
        if ctx.needs_input_grad[0]:
            gu = sum((g[:, :, d].unsqueeze(-1) * _GatherGrad.apply(u, phi, geom, _bump(k, d))).sum(dim=1) for d in range(3))
        if ctx.needs_input_grad[1]:
            gphi = sum(_Spread.apply(u, g[:, :, d], geom, _bump(k, d)) for d in range(3))
        return gu, gphi, None, None


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, phi, geom, k):
        uc, pc = u.detach().contiguous(), phi.detach().contiguous()
        n_ch, n = pc.shape[0], uc.shape[0]
        out = torch.empty((n, n_ch), dtype=pc.dtype, device=pc.device)
        md = geom.desc(n_ch)
        with _lib.on_device(pc.device):
            _lib.check(_lib.load().mipme_gather_jet(_stream(pc), _lib.dtype_code(pc.dtype), C.byref(md), n, uc.data_ptr(),
                                                    pc.data_ptr(), k[0], k[1], k[2], out.data_ptr()))
        ctx.save_for_backward(u, phi)
        ctx.geom, ctx.k = geom, k
        return out

    @staticmethod
    def backward(ctx, g):
        u, phi = ctx.saved_tensors
        geom, k = ctx.geom, ctx.k
        gu = gphi = None
        if ctx.needs_input_grad[0]:
            gu = (g.unsqueeze(-1) * _GatherGrad.apply(u, phi, geom, k)).sum(dim=1)
        if ctx.needs_input_grad[1]:
            gphi = _Spread.apply(u, g, geom, k)
        return gu, gphi, None, None


def _complex_dtype(dtype):
    return torch.complex64 if dtype == torch.float32 else torch.complex128


class _Convolve(torch.autograd.Function):
    """``irfftn(G * rfftn(mesh))``, both transforms un-normalised, ``G`` real on the half grid (``KSpaceFilter.forward``)."""

    @staticmethod
    def forward(ctx, mesh, G, geom):
        mc = mesh.detach().contiguous()
        Gc = G.detach().to(mc.dtype).contiguous()
        n_ch = mc.shape[0]
        hat = torch.empty((n_ch, geom.n_half), dtype=_complex_dtype(mc.dtype), device=mc.device)
        work = torch.empty_like(hat)
        out = torch.empty_like(mc)
        plan = _lib.get_plan(mc.device, mc.dtype, geom.ns, n_ch)
        with _lib.on_device(mc.device):
            _lib.check(_lib.load().mipme_convolve(plan.handle, _stream(mc), mc.data_ptr(), Gc.data_ptr(), hat.data_ptr(),
                                                  work.data_ptr(), out.data_ptr(), None))
        ctx.save_for_backward(mesh, G)
        ctx.geom = geom
        return out

    @staticmethod
    def backward(ctx, g):
        mesh, G = ctx.saved_tensors
        gm = gG = None
        if ctx.needs_input_grad[0]:
            gm = _Convolve.apply(g, G, ctx.geom)
        if ctx.needs_input_grad[1]:
            gG = _SpectralDot.apply(mesh, g, ctx.geom).to(G.dtype)
        return gm, gG, None


def _rfftn(mesh, geom):
    mc = mesh.detach().contiguous()
    n_ch = mc.shape[0]
    hat = torch.empty((n_ch, geom.ns[0], geom.ns[1], geom.ns[2] // 2 + 1), dtype=_complex_dtype(mc.dtype), device=mc.device)
    plan = _lib.get_plan(mc.device, mc.dtype, geom.ns, n_ch)
    md = geom.desc(n_ch)
    with _lib.on_device(mc.device):
        _lib.check(_lib.load().mipme_fft_r2c(plan.handle, _stream(mc), _lib.dtype_code(mc.dtype), C.byref(md), mc.data_ptr(),
                                             hat.data_ptr()))
    return hat


def _constant(geom, name, dtype, device, make):
    """Small device constants of a mesh geometry (made once per geometry, dtype and device: a host list copied to the device
    every call is a synchronous copy, and not capturable into a HIP graph)."""
    store = geom.__dict__.setdefault("_analytic_constants", {})
    key = (name, dtype, device)
    t = store.get(key)
    if t is None:
        t = store[key] = make()
    return t


def _mesh_sizes(geom, dtype, device):
    return _constant(geom, "ns", dtype, device, lambda: torch.tensor([float(n) for n in geom.ns], dtype=dtype, device=device))


def _multiplicity(geom, dtype, device):
    def make():
        nz = geom.ns[2]
        mu = [2.0] * (nz // 2 + 1)
        mu[0] = 1.0
        if nz % 2 == 0:
            mu[-1] = 1.0
        return torch.tensor(mu, dtype=dtype, device=device)

    return _constant(geom, "mu", dtype, device, make)


class _SpectralDot(torch.autograd.Function):
    """``out(k) = mu(k) sum_c Re(a^_c(k) conj b^_c(k))`` on the half grid: the adjoint of :class:`_Convolve` w.r.t. its table."""

    @staticmethod
    def forward(ctx, a, b, geom):
        ha, hb = _rfftn(a, geom), _rfftn(b, geom)
        out = (ha * hb.conj()).real.sum(dim=0) * _multiplicity(geom, a.dtype, a.device)
        ctx.save_for_backward(a, b)
        ctx.geom = geom
        return out

    @staticmethod
    def backward(ctx, c):
        a, b = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = _Convolve.apply(b, c, ctx.geom)
        if ctx.needs_input_grad[1]:
            gb = _Convolve.apply(a, c, ctx.geom)
        return ga, gb, None


_TRANSPOSED = {0: 0, 1: 2, 2: 1}


class _PairSum(torch.autograd.Function):
    """mode 0: half list, ``out_i += w x_j`` and ``out_j += w x_i``; 1: full list, ``out_i += w x_j``; 2: its transpose."""

    @staticmethod
    def forward(ctx, w, x, pairs, mode):
        wc, xc = w.detach().to(x.dtype).contiguous(), x.detach().contiguous()
        out = torch.empty_like(xc)
        lib = _lib.load()
        with _lib.on_device(xc.device):
            if ROWS and pairs.shape[0] >= ROWS_MIN_PAIRS and getattr(pairs, "_mipme_stream", None) is None:
                # the transposed list of this tensor (built once, cached on its identity: ops.get_topology; no bets outside a
                # betting scope): owner-computes rows instead of atomics
                topo = ops.get_topology(pairs, xc.shape[0])
                _lib.check(lib.mipme_pair_sum_rows(_stream(xc), _lib.dtype_code(xc.dtype), xc.shape[0], xc.shape[1],
                                                   topo.row_ptr.data_ptr(), topo.entries.data_ptr(), wc.data_ptr(),
                                                   xc.data_ptr(), mode, out.data_ptr()))
            else:
                _lib.check(lib.mipme_pair_sum(_stream(xc), _lib.dtype_code(xc.dtype), _lib.index_code(pairs.dtype),
                                              pairs.shape[0], xc.shape[0], xc.shape[1], pairs.data_ptr(), wc.data_ptr(),
                                              xc.data_ptr(), mode, out.data_ptr()))
        ctx.save_for_backward(w, x)
        ctx.pairs, ctx.mode = pairs, mode
        return out

    @staticmethod
    def backward(ctx, g):
        w, x = ctx.saved_tensors
        pairs, mode = ctx.pairs, ctx.mode
        gw = gx = None
        if ctx.needs_input_grad[0]:
            gw = (_PairDot.apply(x, g, pairs, False) if mode == 2 else _PairDot.apply(g, x, pairs, mode == 0)).to(w.dtype)
        if ctx.needs_input_grad[1]:
            gx = _PairSum.apply(w, g, pairs, _TRANSPOSED[mode])
        return gw, gx, None, None


class _PairDot(torch.autograd.Function):
    """``out_p = a_i . b_j`` (+ ``a_j . b_i`` for a half list)."""

    @staticmethod
    def forward(ctx, a, b, pairs, half):
        ac, bc = a.detach().contiguous(), b.detach().contiguous()
        out = torch.empty((pairs.shape[0],), dtype=ac.dtype, device=ac.device)
        with _lib.on_device(ac.device):
            _lib.check(_lib.load().mipme_pair_dot(_stream(ac), _lib.dtype_code(ac.dtype), _lib.index_code(pairs.dtype),
                                                  pairs.shape[0], ac.shape[1], pairs.data_ptr(), ac.data_ptr(), bc.data_ptr(),
                                                  1 if half else 0, out.data_ptr()))
        ctx.save_for_backward(a, b)
        ctx.pairs, ctx.half = pairs, half
        return out

    @staticmethod
    def backward(ctx, c):
        a, b = ctx.saved_tensors
        pairs, half = ctx.pairs, ctx.half
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = _PairSum.apply(c, b, pairs, 0 if half else 1)
        if ctx.needs_input_grad[1]:
            gb = _PairSum.apply(c, a, pairs, 0 if half else 2)
        return ga, gb, None, None


class _PairDiff(torch.autograd.Function):
    """``out_p = x_j - x_i`` (the reference helper's ``positions[j] - positions[i]``); adjoint: :class:`_PairScatter`."""

    @staticmethod
    def forward(ctx, x, pairs, rows):
        xc = x.detach().contiguous()
        out = torch.empty((pairs.shape[0], xc.shape[1]), dtype=xc.dtype, device=xc.device)
        with _lib.on_device(xc.device):
            _lib.check(_lib.load().mipme_pair_diff(_stream(xc), _lib.dtype_code(xc.dtype), _lib.index_code(pairs.dtype),
                                                   pairs.shape[0], xc.shape[1], pairs.data_ptr(), xc.data_ptr(), out.data_ptr()))
        ctx.pairs, ctx.rows, ctx.n = pairs, rows, xc.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return _PairScatter.apply(g, ctx.pairs, ctx.rows, ctx.n), None, None


class _PairScatter(torch.autograd.Function):
    """``out_a = sum_{p: j_p = a} v_p - sum_{p: i_p = a} v_p`` (the helper's two ``index_add_``); adjoint: :class:`_PairDiff`.
    ``rows = (row_ptr, entries)`` of the transposed list: owner-computes, no atomics; ``None``: atomics on the (P,2) list."""

    @staticmethod
    def forward(ctx, v, pairs, rows, n_atoms):
        vc = v.detach().contiguous()
        out = torch.empty((n_atoms, vc.shape[1]), dtype=vc.dtype, device=vc.device)
        with _lib.on_device(vc.device):
            _lib.check(_lib.load().mipme_pair_scatter(
                _stream(vc), _lib.dtype_code(vc.dtype), _lib.index_code(pairs.dtype), pairs.shape[0], n_atoms, vc.shape[1],
                pairs.data_ptr(), None if rows is None else rows[0].data_ptr(), None if rows is None else rows[1].data_ptr(),
                vc.data_ptr(), out.data_ptr()))
        ctx.pairs, ctx.rows = pairs, rows
        return out

    @staticmethod
    def backward(ctx, g):
        return _PairDiff.apply(g, ctx.pairs, ctx.rows), None, None, None


def recorded_distance_backward(grad_d, positions, cell, pairs, shifts, rows, want_pos=True, want_cell=True):
    """The adjoint of ``d_p = |r_j - r_i + S_p cell|`` for a backward pass that is itself being recorded (create_graph=True):
    differentiable to any order, what the reference's helper gives (``tests/helpers.py:278-304``) -- with the pair difference
    and the scatter as primitives of this module instead of ATen's indexing kernels (whose ``index_add_`` is a compare-and-swap
    loop in double precision: 4 ms for 1.2 M pairs, 70 % of a force-loss step).  Returns ``(grad_positions, grad_cell)``."""
    vec = _PairDiff.apply(positions, pairs, rows)
    if shifts is not None and cell is not None:
        vec = vec + shifts @ cell
    gvec = (grad_d / torch.linalg.norm(vec, dim=1)).unsqueeze(1) * vec
    gp = _PairScatter.apply(gvec, pairs, rows, positions.shape[0]) if want_pos else None
    gc = shifts.T @ gvec if (want_cell and shifts is not None and cell is not None) else None
    return gp, gc


# ---- the composition ---------------------------------------------------------------------------------------------------------


def _sinc_pi(y: torch.Tensor) -> torch.Tensor:
    """sin(y) / y with derivatives of every order finite at y = 0 (``torch.sinc`` differentiates once there; its second
    derivative is 0/0, which is how the reference's own double backward w.r.t. the cell becomes NaN)."""
    small = y.abs() < 1e-2
    y2 = torch.where(small, y * y, torch.zeros_like(y))
    series = 1 - y2 / 6 * (1 - y2 / 20 * (1 - y2 / 42 * (1 - y2 / 72)))
    ys = torch.where(small, torch.ones_like(y), y)
    return torch.where(small, series, torch.sin(ys) / ys)


def filter_table(calculator, cell: torch.Tensor, ns, geom=None) -> torch.Tensor:
    """G(k) on the rfft half grid as a differentiable function of the cell, in float64: ``kernel_from_k_sq(|k|^2)`` (PME,
    ``lib/kspace_filter.py:97-120``), divided by the squared Fourier transform of the charge assignment function for P3M
    (mode 0: ``prod_d sinc(k_d h_d / 2 pi)^(2 n)`` with h_d = |a_d| / n_d, zero where that vanishes; ``:293-329,349-361``)."""
    from .calculators import _reciprocal_and_det

    c64 = cell.to(torch.float64)
    if geom is None:
        k = lib.generate_kvectors_for_mesh(c64, ns)
    else:  # the same grid (lib/kvectors.py:77-102) from cached integer frequencies and a cross-product inverse: no LU
        # factorisation per call (whose workspace handling is not capturable into a HIP graph)
        def freq(n, half):
            f = torch.arange(n // 2 + 1 if half else n, device=cell.device)
            if not half:
                f = torch.where(f < (n + 1) // 2, f, f - n)
            return f.to(torch.float64)

        fx = _constant(geom, "fx", torch.float64, cell.device, lambda: freq(ns[0], False))
        fy = _constant(geom, "fy", torch.float64, cell.device, lambda: freq(ns[1], False))
        fz = _constant(geom, "fz", torch.float64, cell.device, lambda: freq(ns[2], True))
        recip = (2 * torch.pi) * _reciprocal_and_det(c64)[0]
        k = fx[:, None, None, None] * recip[0] + fy[None, :, None, None] * recip[1] + fz[None, None, :, None] * recip[2]
    G = calculator.potential.kernel_from_k_sq((k * k).sum(dim=-1))
    if calculator._scheme == _lib.P3M:
        nst = (_mesh_sizes(geom, torch.float64, cell.device) if geom is not None else
               torch.tensor([float(n) for n in ns], dtype=torch.float64, device=cell.device))
        kh = k * (torch.linalg.norm(c64, dim=1) / nst)
        U2 = torch.prod(_sinc_pi(0.5 * kh), dim=-1) ** (2 * calculator.interpolation_nodes)
        dead = U2 == 0
        G = torch.where(dead, torch.zeros_like(G), G / torch.where(dead, torch.ones_like(U2), U2))
    return G


def potentials(calculator, charges, cell, positions, neighbor_indices, neighbor_distances, periodic=None, node_mask=None,
               pair_mask=None, kvectors=None) -> torch.Tensor:
    """Per-atom potentials ``(n_atoms, n_channels)``: ``Calculator.forward`` of the reference, differentiable to any order."""
    from ._utils import _validate_parameters

    _validate_parameters(charges=charges, cell=cell, positions=positions, neighbor_indices=neighbor_indices,
                         neighbor_distances=neighbor_distances, periodic=periodic, pair_mask=pair_mask, node_mask=node_mask,
                         kvectors=kvectors)
    _lib.require_device(positions, "positions")
    if getattr(neighbor_indices, "_mipme_stream", None) is not None:
        raise ValueError('the handles of a NeighborStream serve the fused kernels only; `double_backward = "analytic"` needs a '
                         "list in the reference's format (`stream.pairs()`)")
    pot = calculator.potential
    is_ewald = hasattr(calculator, "lr_wavelength")
    if pot.smearing is not None and not is_ewald and not hasattr(calculator, "mesh_spacing"):
        raise NotImplementedError(
            f'`double_backward = "analytic"` covers the Ewald and mesh calculators and the plain pair sum; use '
            f'"finite-difference" for {type(calculator).__name__}')
    if pot.smearing is not None and not is_ewald and (node_mask is not None or kvectors is not None):
        raise NotImplementedError("Batching not implemented for mesh-based calculators")
    if charges.dim() != 2:
        raise NotImplementedError('padded batches are not served by `double_backward = "analytic"`')
    dtype = charges.dtype
    pairs = neighbor_indices.contiguous()
    # ---- real space: _compute_rspace (calculators/calculator.py:43-87)
    if pot.smearing is None:
        bare = pot.from_dist(neighbor_distances, pair_mask)
        if pot.exclusion_radius is not None:
            bare = bare * (1 - pot.f_cutoff(neighbor_distances, pair_mask))
    elif pot.exclusion_radius is None:
        # v_SR = (1 - P(p/2, x)) / d^p = Q(p/2, x) / d^p, x = d^2 / 2 sigma^2, in ONE piece: erfc(sqrt x) (+ a finite sum) for odd p,
        # e^-x times a finite sum for even p (potentials/coulomb.py:98-120, potentials/inversepowerlaw.py:72-107 form it as the
        # difference 1/d^p - P/d^p).  A handful of elementwise operations on the pair list instead of the incomplete-gamma
        # series' eighty -- autograd keeps a (P,) tensor for each -- and no cancellation at short distances.
        from .potentials import _reg_upper_gamma

        sm, pref, p = pot._host_params()  # Python floats (one copy per parameter set): no device copy per call
        d = neighbor_distances
        bare = pref * _reg_upper_gamma(p, d * d * (0.5 / (sm * sm))) / (d if p == 1 else d**p)
        if pair_mask is not None:
            bare = bare * pair_mask
    else:
        bare = pot.sr_from_dist(neighbor_distances, pair_mask)
    out = _PairSum.apply(bare.to(dtype), charges, pairs, 1 if calculator.full_neighbor_list else 0) / 2
    if pot.smearing is None:
        return out
    from .calculators import _reciprocal_and_det  # inv(cell).T and det from cross products: no host synchronisation

    recip, det = _reciprocal_and_det(cell)
    ivolume = torch.abs(det).pow(-1)
    if is_ewald:
        # ---- explicit Ewald sum (calculators/ewald.py:72-142): k = 2 pi F A^-T from the cell, the (K, N) phase tables as
        # tensor expressions, K in slabs (this is the reference's own formulation; the fused route is csrc/ewald.hip)
        if kvectors is None:
            kvectors = (2 * torch.pi) * calculator._frequencies(cell) @ recip
        lr = _ewald_kspace(pot, charges, positions, kvectors) * ivolume
    else:
        lr = _mesh_kspace(calculator, charges, cell, positions, recip.T) * ivolume
    lr = lr - charges * pot.self_contribution().to(dtype)
    lr = lr - 2 * pot.background_correction().to(dtype) * charges.sum(dim=0) * ivolume
    lr = lr + pot.pbc_correction(periodic, positions, cell, charges).to(dtype)
    if node_mask is not None:
        lr = lr * node_mask.unsqueeze(-1)
    return out + lr / 2


def _ewald_kspace(pot, charges, positions, kvectors) -> torch.Tensor:
    """``sum_k G(k) [cos(k r_i) sum_j q_j cos(k r_j) + sin(k r_i) sum_j q_j sin(k r_j)]`` (before the 1 / V)."""
    G = pot.lr_from_k_sq((kvectors * kvectors).sum(dim=-1)).to(charges.dtype)
    n_k, n = kvectors.shape[0], positions.shape[0]
    step = max(1, (1 << 24) // max(1, n))
    acc = torch.zeros_like(charges)
    for a in range(0, n_k, step):
        theta = kvectors[a:a + step] @ positions.T
        c, s, g = torch.cos(theta), torch.sin(theta), G[a:a + step, None]
        acc = acc + c.T @ ((c @ charges) * g) + s.T @ ((s @ charges) * g)
    return acc


def _geometry(calculator, cell):
    """Mesh sizes need the cell on the host (as in the reference, ``lib/kvectors.py:5-21``): one copy per cell tensor and
    version, remembered on the calculator -- a training loop over the same structure does not wait for the device every call."""
    import weakref

    c = calculator.__dict__.get("_analytic_geom")
    key = (calculator.mesh_spacing, calculator._scheme, calculator.interpolation_nodes)
    if c is not None and c[0]() is cell and c[1] == cell._version and c[2] == key:
        return c[3]
    cell_host = cell.detach().to("cpu", torch.float64).numpy()
    ns = ops.ns_mesh_from_cell(cell_host, calculator.mesh_spacing)
    geom = ops.MeshGeometry(cell_host, ns, calculator._scheme, calculator.interpolation_nodes)
    try:
        calculator.__dict__["_analytic_geom"] = (weakref.ref(cell), cell._version, key, geom)
    except TypeError:  # (a tensor subclass without weak references: no cache)
        pass
    return geom


def _mesh_kspace(calculator, charges, cell, positions, inv_cell) -> torch.Tensor:
    """``mesh_to_points(filter(points_to_mesh(charges)))`` (before the 1 / V): calculators/pme.py:88-113."""
    dtype = charges.dtype
    geom = _geometry(calculator, cell)
    ns = geom.ns
    u = _mesh_sizes(geom, dtype, positions.device) * (positions @ inv_cell)
    G = filter_table(calculator, cell, ns, geom).to(dtype)
    zero = (0, 0, 0)
    rho = _Spread.apply(u, charges, geom, zero)
    phi = _Convolve.apply(rho, G, geom)
    return _Gather.apply(u, phi, geom, zero)

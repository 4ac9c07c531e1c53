"""CPU oracle #2 (PyTorch-CPU, multi-threaded, autograd) -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import this module.  It restates the reference's path as a
chain of ATen ops on CPU tensors -- scatter with ``index_put_(accumulate=True)``, ``torch.fft.rfftn/irfftn``,
gather by advanced indexing, ``index_add_`` for the pair sum, gradients from autograd -- i.e. the same *kind* of
work torch-pme's CPU path does (reference ``calculators/calculator.py:43-87``, ``calculators/pme.py:88-143``,
``lib/mesh_interpolator.py:303-457``, ``lib/kspace_filter.py:122-197``), so that its wall time on the GPU box's host
cores is a fair stand-in for "torch-pme on CPU" (the reference itself cannot travel to the box).  The mathematics is
shared with ``oracle/pme_numpy.py`` (weights from generating rules, filter from closed forms); parity with it and
with the reference goldens is checked in ``tests/test_oracle_golden.py``.  Coulomb potential only (the benchmark
configurations cfg1-cfg4); other exponents are covered by the NumPy oracle.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import pme_numpy as O


def _weights(x: torch.Tensor, order: int, scheme: str) -> torch.Tensor:
    """(order, N, 3) interpolation weights; same generating rules as ``pme_numpy.weights_1d``."""
    n = order
    if scheme == "P3M":
        if n == 1:
            return torch.ones((1,) + x.shape, dtype=x.dtype)
        f = x + 0.5
        a = [torch.ones_like(f)]
        for k in range(2, n + 1):
            new = []
            for j in range(k):
                lo = a[j] if j < k - 1 else 0.0
                hi = a[j - 1] if j >= 1 else 0.0
                new.append(((f + j) * lo + (k - f - j) * hi) / (k - 1))
            a = new
        return torch.stack([a[n - 1 - t] for t in range(n)])
    xi = [t - 0.5 * (n - 1) for t in range(n)]
    out = []
    for t in range(n):
        num = torch.ones_like(x)
        den = 1.0
        for s in range(n):
            if s != t:
                num = num * (x - xi[s])
                den *= xi[t] - xi[s]
        out.append(num / den)
    return torch.stack(out)


def forward(spec: O.PotentialSpec, scheme: str, order: int, mesh_spacing: float, charges: torch.Tensor,
            cell: torch.Tensor, positions: torch.Tensor, pairs: torch.Tensor, dist: torch.Tensor,
            full_list: bool = False) -> torch.Tensor:
    """Per-atom potentials (N, C) on CPU tensors, differentiable through autograd."""
    assert spec.p == 1 and spec.smearing is not None and spec.exclusion_radius is None
    dt = positions.dtype
    sm, pref = spec.smearing, spec.prefactor
    # ---- short range: erfc(d / (sigma sqrt 2)) / d, index_add_ in both directions (calculator.py:70-87)
    v = pref * torch.erfc(dist / (sm * math.sqrt(2.0))) / dist
    i, j = pairs[:, 0], pairs[:, 1]
    pot = torch.zeros_like(charges)
    pot.index_add_(0, i, charges[j] * v[:, None])
    if not full_list:
        pot.index_add_(0, j, charges[i] * v[:, None])
    pot = pot / 2
    # ---- mesh part (pme.py:88-143)
    ns = O.get_ns_mesh(cell.detach().numpy(), mesh_spacing)
    nx, ny, nz = (int(s) for s in ns)
    nst = torch.tensor(ns, dtype=dt)
    inv = torch.linalg.inv(cell)
    u = nst * (positions @ inv)
    if order % 2 == 0:
        m = torch.floor(u)
        x = u - (m + 0.5)
    else:
        m = torch.round(u)
        x = u - m
    m = m.long()
    w = _weights(x, order, scheme)  # (n, N, 3)
    offs = torch.arange(order) + 1 - (order + 1) // 2
    idx = (m[None] + offs[:, None, None]) % torch.tensor(ns)  # (n, N, 3)
    t = torch.arange(order)
    tx, ty, tz = (a.reshape(-1) for a in torch.meshgrid(t, t, t, indexing="ij"))
    ix, iy, iz = idx[tx, :, 0], idx[ty, :, 1], idx[tz, :, 2]  # (n^3, N)
    w3 = w[tx, :, 0] * w[ty, :, 1] * w[tz, :, 2]  # (n^3, N)
    C = charges.shape[1]
    rho = torch.zeros((C, nx, ny, nz), dtype=dt)
    for c in range(C):
        rho[c].index_put_((ix, iy, iz), charges[:, c] * w3, accumulate=True)
    # filter: k-vectors from the cell so that the cell gradient flows as in the reference
    fx = torch.fft.fftfreq(nx, dtype=dt) * nx
    fy = torch.fft.fftfreq(ny, dtype=dt) * ny
    fz = torch.fft.rfftfreq(nz, dtype=dt) * nz
    B = 2 * math.pi * inv.T
    k = fx[:, None, None, None] * B[0] + fy[None, :, None, None] * B[1] + fz[None, None, :, None] * B[2]
    k2 = (k * k).sum(-1)
    zero = k2 == 0
    k2s = torch.where(zero, torch.ones_like(k2), k2)
    G = torch.where(zero, torch.zeros_like(k2), pref * 4 * math.pi * torch.exp(-0.5 * sm * sm * k2s) / k2s)
    if scheme == "P3M":
        h = torch.linalg.norm(cell, dim=1) / nst
        U2 = torch.prod(torch.sinc(k * h / (2 * math.pi)), dim=-1) ** (2 * order)
        G = torch.where(U2 == 0, torch.zeros_like(G), G / torch.where(U2 == 0, torch.ones_like(U2), U2))
    phi = torch.fft.irfftn(torch.fft.rfftn(rho, dim=(1, 2, 3)) * G, s=(nx, ny, nz), dim=(1, 2, 3), norm="forward")
    vol = torch.abs(torch.linalg.det(cell))
    lr = (phi[:, ix, iy, iz] * w3).sum(dim=1).T / vol
    lr = lr - charges * (pref * math.sqrt(2 / math.pi) / sm)
    lr = lr - 2 * (pref * math.pi * sm * sm) * charges.sum(dim=0) / vol
    return pot + lr / 2


def pair_distances(positions: torch.Tensor, cell: torch.Tensor, pairs: torch.Tensor, shifts: torch.Tensor) -> torch.Tensor:
    vec = positions[pairs[:, 1]] - positions[pairs[:, 0]] + shifts.to(cell.dtype) @ cell
    return torch.linalg.norm(vec, dim=1)


def energy_forces_step(spec, scheme, order, mesh_spacing, charges, cell, positions, pairs, shifts):
    """One benchmark step (distances -> potentials -> E = sum q V -> forces) on CPU; returns (E, forces)."""
    pos = positions.detach().clone().requires_grad_(True)
    d = pair_distances(pos, cell, pairs, shifts)
    V = forward(spec, scheme, order, mesh_spacing, charges, cell, pos, pairs, d)
    E = (V * charges).sum()
    (g,) = torch.autograd.grad(E, pos)
    return E.detach(), -g


def as_tensors(w, dtype):
    """Workload (``torchpme_amd.workloads.Workload``) -> CPU tensors."""
    return (torch.tensor(w.charges, dtype=dtype), torch.tensor(w.cell, dtype=dtype),
            torch.tensor(w.positions, dtype=dtype), torch.tensor(np.asarray(w.pairs)),
            torch.tensor(np.asarray(w.shifts)))

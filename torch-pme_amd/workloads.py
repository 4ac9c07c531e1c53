"""Synthetic periodic boxes for the benchmark configurations of BASELINE.json / SURVEY.md 8(d).

All boxes are cubic with number density 0.1 atoms / A^3 and a half neighbour list at ``cutoff`` (9 A by
default), smearing = cutoff / 5, and ``mesh_spacing = 2 L / (n_mesh - 2)`` so that the mesh is exactly
``n_mesh`` points per axis.  Positions sit on a jittered lattice (uniform random points would create
r -> 0 pairs that make fp32 comparisons meaningless).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .neighbors import neighbor_list, neighbor_list_device


def _build_list(pos: np.ndarray, cell: np.ndarray, cutoff: float):
    """Half neighbour list of a synthetic box: built on the GPU when one is present (milliseconds instead of ~15 s
    for 32k atoms with the host k-d tree), on the host otherwise; both give the same pair set."""
    try:
        import torch

        if torch.cuda.is_available():
            p, s, _ = neighbor_list_device(torch.tensor(pos, device="cuda"), torch.tensor(cell, device="cuda"), cutoff)
            return p.cpu().numpy(), s.cpu().numpy().round().astype(np.int64)
    except ImportError:
        pass
    p, s, _ = neighbor_list(pos, cell, cutoff)
    return p, s


@dataclass
class Workload:
    name: str
    positions: np.ndarray  # (N, 3) float64
    charges: np.ndarray  # (N, 1)
    cell: np.ndarray  # (3, 3)
    pairs: np.ndarray  # (P, 2) int64
    shifts: np.ndarray  # (P, 3) int64
    cutoff: float
    smearing: float
    mesh_spacing: float
    n_mesh: int
    scheme: str  # "P3M" | "PME"
    order: int
    exponent: int  # 1 = Coulomb
    dtype: str  # "f32" | "f64"

    @property
    def n_atoms(self) -> int:
        return self.positions.shape[0]

    @property
    def n_pairs(self) -> int:
        return self.pairs.shape[0]


def _lattice(n_side: int, a: float, jitter: float, rng) -> np.ndarray:
    g = (np.arange(n_side) + 0.5) * a
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    return pos + rng.uniform(-jitter, jitter, pos.shape)


def _random_rotations(n: int, rng) -> np.ndarray:
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack(
        [
            np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
            np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
            np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1),
        ],
        axis=1,
    )


def water_box(n_side: int = 22, n_mesh: int = 64, order: int = 5, cutoff: float = 9.0, seed: int = 1234,
              dtype: str = "f32") -> Workload:
    """cfg3: ``3 n_side^3`` atoms of rigid TIP3P-like water (q_O=-0.834, q_H=+0.417, r_OH=0.9572 A,
    HOH=104.52 deg), O on a jittered lattice, random orientations.  n_side=22 -> 31 944 atoms, L = 68.4 A."""
    rng = np.random.default_rng(seed)
    n_mol = n_side**3
    L = (3 * n_mol / 0.1) ** (1 / 3)
    a = L / n_side
    oxy = _lattice(n_side, a, 0.4, rng)
    half = np.deg2rad(104.52) / 2
    h_local = 0.9572 * np.array([[np.sin(half), 0.0, np.cos(half)], [-np.sin(half), 0.0, np.cos(half)]])
    R = _random_rotations(n_mol, rng)
    h = oxy[:, None, :] + np.einsum("mab,hb->mha", R, h_local)
    pos = np.concatenate([oxy[:, None, :], h], axis=1).reshape(-1, 3)
    q = np.tile(np.array([-0.834, 0.417, 0.417]), n_mol).reshape(-1, 1)
    cell = L * np.eye(3)
    pairs, shifts = _build_list(pos, cell, cutoff)
    return Workload(f"water_{3 * n_mol}", pos, q, cell, pairs, shifts, cutoff, cutoff / 5, 2 * L / (n_mesh - 2), n_mesh,
                    "P3M", order, 1, dtype)


def ionic_box(n_side: int = 20, n_mesh: int = 32, order: int = 4, cutoff: float = 9.0, seed: int = 12,
              dtype: str = "f64") -> Workload:
    """cfg2: ``n_side^3`` point charges ~ N(0,1), made neutral, on a jittered lattice (8 000 atoms, L = 43.1 A)."""
    rng = np.random.default_rng(seed)
    n = n_side**3
    L = (n / 0.1) ** (1 / 3)
    pos = _lattice(n_side, L / n_side, 0.4, rng)
    q = rng.normal(size=(n, 1))
    q -= q.mean()
    cell = L * np.eye(3)
    pairs, shifts = _build_list(pos, cell, cutoff)
    return Workload(f"ionic_{n}", pos, q, cell, pairs, shifts, cutoff, cutoff / 5, 2 * L / (n_mesh - 2), n_mesh, "P3M",
                    order, 1, dtype)


def dispersion_box(n_side: int = 64, n_mesh: int = 128, order: int = 5, cutoff: float = 9.0, seed: int = 8,
                   dtype: str = "f32") -> Workload:
    """cfg5: ``n_side^3`` atoms with C6-like weights U(0.5, 1.5) and the 1/r^6 potential (262 144 atoms)."""
    rng = np.random.default_rng(seed)
    n = n_side**3
    L = (n / 0.1) ** (1 / 3)
    pos = _lattice(n_side, L / n_side, 0.4, rng)
    q = rng.uniform(0.5, 1.5, size=(n, 1))
    cell = L * np.eye(3)
    pairs, shifts = _build_list(pos, cell, cutoff)
    return Workload(f"dispersion_{n}", pos, q, cell, pairs, shifts, cutoff, cutoff / 5, 2 * L / (n_mesh - 2), n_mesh,
                    "P3M", order, 6, dtype)

"""Host-side periodic neighbour list (caller side of the hot path).

The reference takes its pair list from third-party ``vesin`` (``tests/helpers.py:240-275``,
quantities ``"PdS"``); that package is not part of torch-pme and is absent here, so the build
carries its own builder.  It returns exactly what the calculators consume:

* ``pairs``  (P, 2) int64  -- atom indices ``(i, j)``
* ``shifts`` (P, 3) int64  -- integer cell shifts ``S`` with ``r_ij = r_j - r_i + S @ cell``
* ``dist``   (P,)  float64 -- ``|r_ij|``

Half list: every unordered interaction once (``i < j`` for any shift; ``i == j`` only for the
lexicographically positive half of the shifts).  Full list: both directions.  The cutoff may
exceed half the box (several periodic images), and cells may be triclinic.  Pairs at exactly the cutoff
are excluded (``d < cutoff``).
"""

from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def _image_range(cell: np.ndarray, cutoff: float, periodic) -> np.ndarray:
    """Images needed per axis: ceil(cutoff / perpendicular width of the cell along that axis)."""
    vol = abs(np.linalg.det(cell))
    n = np.zeros(3, dtype=np.int64)
    for d in range(3):
        if not periodic[d]:
            continue
        cross = np.cross(cell[(d + 1) % 3], cell[(d + 2) % 3])
        width = vol / np.linalg.norm(cross)
        n[d] = int(np.ceil(cutoff / width))
    return n


def neighbor_list(positions, cell, cutoff: float, full_list: bool = False, periodic=(True, True, True)):
    """Build the pair list of all atom pairs (including periodic images) closer than ``cutoff``."""
    pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float64))
    A = np.ascontiguousarray(np.asarray(cell, dtype=np.float64))
    periodic = np.asarray(periodic, dtype=bool)
    N = pos.shape[0]
    Ainv = np.linalg.inv(A)
    frac = pos @ Ainv
    wrap = np.where(periodic, np.floor(frac), 0.0).astype(np.int64)
    fw = frac - wrap
    base = fw @ A
    nimg = _image_range(A, cutoff, periodic)
    # ghost images: only those whose fractional coordinate lies within the cutoff skin of the cell
    vol = abs(np.linalg.det(A))
    skin = np.zeros(3)
    for d in range(3):
        cross = np.cross(A[(d + 1) % 3], A[(d + 2) % 3])
        skin[d] = cutoff * np.linalg.norm(cross) / vol
    ghost_pos, ghost_idx, ghost_shift = [], [], []
    rng = [np.arange(-nimg[d], nimg[d] + 1) for d in range(3)]
    for sx in rng[0]:
        for sy in rng[1]:
            for sz in rng[2]:
                s = np.array([sx, sy, sz])
                f = fw + s
                keep = np.all((f >= -skin) & (f <= 1.0 + skin), axis=1)
                if not keep.any():
                    continue
                ids = np.nonzero(keep)[0]
                ghost_pos.append(base[ids] + s @ A)
                ghost_idx.append(ids)
                ghost_shift.append(np.broadcast_to(s, (ids.size, 3)))
    gp = np.concatenate(ghost_pos)
    gi = np.concatenate(ghost_idx)
    gs = np.concatenate(ghost_shift)
    tree_real = cKDTree(base)
    tree_all = cKDTree(gp)
    coo = tree_real.sparse_distance_matrix(tree_all, cutoff, output_type="coo_matrix")
    i = coo.row.astype(np.int64)
    b = coo.col.astype(np.int64)
    j = gi[b]
    s = gs[b]
    S = s - wrap[j] + wrap[i]
    notself = ~((i == j) & np.all(s == 0, axis=1))
    if full_list:
        keep = notself
    else:
        lexpos = (S[:, 0] > 0) | ((S[:, 0] == 0) & ((S[:, 1] > 0) | ((S[:, 1] == 0) & (S[:, 2] > 0))))
        keep = notself & ((i < j) | ((i == j) & lexpos))
    i, j, S = i[keep], j[keep], S[keep]
    order = np.lexsort((S[:, 2], S[:, 1], S[:, 0], j, i))
    i, j, S = i[order], j[order], S[order]
    vec = pos[j] - pos[i] + S @ A
    dist = np.sqrt(np.sum(vec * vec, axis=1))
    inside = dist < cutoff  # strictly inside, like the list the reference tests use (58 half pairs for CsCl, rc = 2)
    return np.stack([i, j], axis=1)[inside], S[inside], dist[inside]


def neighbor_list_bruteforce(positions, cell, cutoff: float, full_list: bool = False, periodic=(True, True, True)):
    """O(N^2 * images) reference implementation used to test :func:`neighbor_list`."""
    pos = np.asarray(positions, dtype=np.float64)
    A = np.asarray(cell, dtype=np.float64)
    periodic = np.asarray(periodic, dtype=bool)
    N = pos.shape[0]
    # positions may lie outside the cell: pad the image range by the spread of fractional coordinates
    frac = pos @ np.linalg.inv(A)
    extra = np.where(periodic, np.ceil(frac.max(axis=0) - frac.min(axis=0)), 0).astype(np.int64)
    nimg = _image_range(A, cutoff, periodic) + extra
    out_i, out_j, out_S = [], [], []
    for sx in range(-nimg[0], nimg[0] + 1):
        for sy in range(-nimg[1], nimg[1] + 1):
            for sz in range(-nimg[2], nimg[2] + 1):
                S = np.array([sx, sy, sz])
                vec = pos[None, :, :] - pos[:, None, :] + S @ A
                d = np.sqrt(np.sum(vec * vec, axis=2))
                ii, jj = np.nonzero(d < cutoff)
                for a, b in zip(ii, jj):
                    if a == b and not S.any():
                        continue
                    if not full_list:
                        lexpos = tuple(S) > (0, 0, 0)
                        if not (a < b or (a == b and lexpos)):
                            continue
                    out_i.append(a)
                    out_j.append(b)
                    out_S.append(S)
    i = np.array(out_i, dtype=np.int64)
    j = np.array(out_j, dtype=np.int64)
    S = np.array(out_S, dtype=np.int64).reshape(-1, 3)
    order = np.lexsort((S[:, 2], S[:, 1], S[:, 0], j, i))
    i, j, S = i[order], j[order], S[order]
    vec = pos[j] - pos[i] + S @ A
    return np.stack([i, j], axis=1), S, np.sqrt(np.sum(vec * vec, axis=1))

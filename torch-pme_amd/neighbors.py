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
                # (along a non-periodic axis there are no images and atoms may lie anywhere: no filter there)
                keep = np.all(((f >= -skin) & (f <= 1.0 + skin)) | ~periodic, axis=1)
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


def neighbor_list_device(positions, cell, cutoff: float, full_list: bool = False, periodic=(True, True, True)):
    """Neighbour list built ON THE GPU (``csrc/neighbors.hip``): ``positions`` (N,3) and ``cell`` (3,3) are device
    tensors; returns device tensors ``pairs`` (P,2) int64, ``shifts`` (P,3) and ``dist`` (P,) in the dtype of
    ``positions`` -- the same pair set as :func:`neighbor_list` (rows ordered by the first index; the order inside a
    row is the cell-traversal order).  Every periodic axis needs at least 3 cutoff-wide cells (raises ``ValueError``
    otherwise: use the host builder then); along a non-periodic axis (no images) the atoms may lie anywhere -- the cell grid
    then spans their extent, which costs one more small device-to-host copy."""
    import ctypes as C

    import torch

    from . import _lib

    _lib.require_device(positions, "positions")
    lib = _lib.load()
    device, dtype = positions.device, positions.dtype
    A = cell.detach().to("cpu", torch.float64).numpy()
    vol = abs(np.linalg.det(A))
    periodic = [bool(p) for p in periodic]
    frac_off, frac_scale = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
    if not all(periodic) and positions.shape[0] > 0:
        frac = positions.detach().to(torch.float64) @ torch.tensor(np.linalg.inv(A), device=positions.device)
        lo, hi = frac.min(dim=0).values.cpu().numpy(), frac.max(dim=0).values.cpu().numpy()
    nc = []
    for d in range(3):
        width = vol / np.linalg.norm(np.cross(A[(d + 1) % 3], A[(d + 2) % 3]))
        if periodic[d]:
            nc.append(int(np.floor(width / cutoff)))
        else:  # the grid spans the atoms' extent along this axis (slightly enlarged so that the last atom is inside)
            span = (float(hi[d] - lo[d]) if positions.shape[0] > 0 else 0.0) * (1.0 + 1e-9) + 1e-9
            frac_off[d] = float(lo[d]) if positions.shape[0] > 0 else 0.0
            frac_scale[d] = 1.0 / span
            nc.append(max(1, int(np.floor(width * span / cutoff))))
    if any(p and n < 3 for n, p in zip(nc, periodic)):
        raise ValueError(
            f"device neighbour list needs >= 3 cells of width >= cutoff per periodic axis, got {nc}; use neighbor_list() (host)"
        )
    nc = [min(n, 256) for n in nc]
    desc = _lib.NlDesc()
    desc.cell[:] = A.ravel().tolist()
    desc.inv_cell[:] = np.linalg.inv(A).ravel().tolist()
    desc.n_cells[:] = nc
    desc.periodic[:] = [int(p) for p in periodic]
    desc.frac_offset[:] = frac_off
    desc.frac_scale[:] = frac_scale
    desc.cutoff = float(cutoff)
    desc.full_list = int(bool(full_list))
    pos = positions.detach().contiguous()
    N = pos.shape[0]
    ncells = nc[0] * nc[1] * nc[2]
    i32 = dict(dtype=torch.int32, device=device)
    cell_of = torch.empty((max(N, 1),), **i32)
    wrap = torch.empty((max(N, 1), 3), **i32)
    cell_start = torch.empty((ncells + 1,), **i32)
    cell_atoms = torch.empty((max(N, 1),), **i32)
    scratch = torch.empty((lib.mipme_nl_scratch_ints(C.byref(desc), N),), **i32)
    counts = torch.zeros((N,), **i32)
    dt = _lib.dtype_code(dtype)
    with torch.cuda.device(device):
        st = _lib.current_stream(device)
        _lib.check(lib.mipme_nl_bin(st, dt, C.byref(desc), N, pos.data_ptr(), cell_of.data_ptr(), wrap.data_ptr(),
                                    cell_start.data_ptr(), cell_atoms.data_ptr(), scratch.data_ptr()))
        _lib.check(lib.mipme_nl_count(st, dt, C.byref(desc), N, pos.data_ptr(), wrap.data_ptr(), cell_start.data_ptr(),
                                      cell_atoms.data_ptr(), counts.data_ptr()))
        offsets = torch.zeros((N + 1,), dtype=torch.int64, device=device)
        torch.cumsum(counts, dim=0, out=offsets[1:])
        P = int(offsets[-1].item())  # the one host synchronisation: the size of the list
        pairs = torch.empty((P, 2), dtype=torch.int64, device=device)
        shifts = torch.empty((P, 3), dtype=dtype, device=device)
        dist = torch.empty((P,), dtype=dtype, device=device)
        if P > 0:
            _lib.check(lib.mipme_nl_fill(st, dt, C.byref(desc), N, pos.data_ptr(), wrap.data_ptr(), cell_start.data_ptr(),
                                         cell_atoms.data_ptr(), offsets.data_ptr(), pairs.data_ptr(), shifts.data_ptr(),
                                         dist.data_ptr()))
    return pairs, shifts, dist

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
    if N == 0:
        return np.zeros((0, 2), dtype=np.int64), np.zeros((0, 3), dtype=np.int64), np.zeros((0,))
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


def _cell_grid(A: np.ndarray, cutoff: float, periodic, n_atoms: int, lo=None, hi=None):
    """Cell grid of the device builders: ``(n_cells, reach, frac_offset, frac_scale)``.  Cells are a shade more than
    ``cutoff / 2`` wide (perpendicular width) where the box allows it -- so that ``reach = ceil(cutoff / width)`` is 2 whatever
    the rounding -- and at most ~4 per atom in total; along a non-periodic axis the grid spans the atoms' extent ``[lo, hi]``
    (fractional coordinates)."""
    vol = abs(np.linalg.det(A))
    target = 0.5 * cutoff * (1.0 + 1e-9)
    widths, frac_off, frac_scale = [], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
    for d in range(3):
        width = vol / np.linalg.norm(np.cross(A[(d + 1) % 3], A[(d + 2) % 3]))
        if not periodic[d]:  # the grid spans the atoms' extent along this axis (slightly enlarged: the last atom is inside)
            span = (float(hi[d] - lo[d]) if lo is not None else 0.0) * (1.0 + 1e-9) + 1e-9
            frac_off[d] = float(lo[d]) if lo is not None else 0.0
            frac_scale[d] = 1.0 / span
            width *= span
        widths.append(width)
    nc = [max(1, min(4096, int(np.floor(w / target)))) for w in widths]
    limit = max(64, 4 * int(n_atoms))
    while nc[0] * nc[1] * nc[2] > limit:  # very dilute systems: coarser cells (the reach follows)
        f = (limit / (nc[0] * nc[1] * nc[2])) ** (1.0 / 3.0)
        nc = [max(1, min(n - 1, int(np.floor(n * f)))) if n > 1 else 1 for n in nc]
    reach = [int(np.ceil(cutoff / (w / n))) for w, n in zip(widths, nc)]
    # a non-periodic axis has no images: cells beyond [0, n) do not exist, so the walk never needs more than n - 1 cells to
    # either side -- a planar system (all atoms at one height), a slab much thinner than the cutoff, or one atom would
    # otherwise ask for millions of cells of reach along that axis (and be refused as "more than 100 images")
    reach = [r if periodic[d] else min(r, nc[d] - 1) for d, r in enumerate(reach)]
    return nc, reach, frac_off, frac_scale


def _nl_descriptor(cell_host: np.ndarray, cutoff: float, periodic, positions, full_list: bool):
    """``mipme_nl_t`` for this cell / cutoff (host arithmetic; a non-periodic axis needs the atoms' extent: one small
    device-to-host copy)."""
    import torch

    from . import _lib

    A = np.ascontiguousarray(np.asarray(cell_host, dtype=np.float64).reshape(3, 3))
    det = float(np.linalg.det(A))
    if det == 0.0 or not np.isfinite(det):
        raise ValueError(f"provided `cell` has a determinant of {det}, i.e. it is not a valid unit cell")
    if not cutoff > 0:
        raise ValueError(f"`cutoff` is {cutoff} but must be positive")
    periodic = [bool(p) for p in periodic]
    Ainv = np.linalg.inv(A)
    lo = hi = None
    N = positions.shape[0]
    if not all(periodic) and N > 0:
        frac = positions.detach().to(torch.float64) @ torch.tensor(Ainv, device=positions.device)
        lo, hi = frac.min(dim=0).values.cpu().numpy(), frac.max(dim=0).values.cpu().numpy()
    nc, reach, frac_off, frac_scale = _cell_grid(A, float(cutoff), periodic, N, lo, hi)
    desc = _lib.NlDesc()
    desc.cell[:] = A.ravel().tolist()
    desc.inv_cell[:] = Ainv.ravel().tolist()
    desc.n_cells[:] = nc
    desc.periodic[:] = [int(p) for p in periodic]
    desc.frac_offset[:] = frac_off
    desc.frac_scale[:] = frac_scale
    desc.reach[:] = reach
    desc.cutoff = float(cutoff)
    desc.full_list = int(bool(full_list))
    return desc


def neighbor_list_device(positions, cell, cutoff: float, full_list: bool = False, periodic=(True, True, True)):
    """Neighbour list built ON THE GPU (``csrc/neighbors.hip``): ``positions`` (N,3) and ``cell`` (3,3) are device
    tensors; returns device tensors ``pairs`` (P,2) int64, ``shifts`` (P,3) and ``dist`` (P,) in the dtype of
    ``positions`` -- the same pair set as :func:`neighbor_list` (rows ordered by the first index; the order inside a
    row is the cell-traversal order).  Any cell, box size and cutoff (a cutoff beyond the box brings in several images, as
    in the host builder); along a non-periodic axis (no images) the atoms may lie anywhere -- the cell grid then spans their
    extent, which costs one more small device-to-host copy.  The one host synchronisation is the size of the list."""
    import ctypes as C

    import torch

    from . import _lib

    _lib.require_device(positions, "positions")
    lib = _lib.load()
    device, dtype = positions.device, positions.dtype
    pos = positions.detach().contiguous()
    N = pos.shape[0]
    desc = _nl_descriptor(cell.detach().to("cpu", torch.float64).numpy(), cutoff, periodic, pos, full_list)
    ws = torch.zeros((lib.mipme_nl_workspace_bytes(C.byref(desc), N),), dtype=torch.uint8, device=device)
    counts = torch.zeros((N,), dtype=torch.int32, device=device)
    dt = _lib.dtype_code(dtype)
    with torch.cuda.device(device):
        st = _lib.current_stream(device)
        _lib.check(lib.mipme_nl_bin(st, dt, C.byref(desc), N, pos.data_ptr(), ws.data_ptr()))
        _lib.check(lib.mipme_nl_count(st, dt, C.byref(desc), N, ws.data_ptr(), counts.data_ptr()))
        offsets = torch.zeros((N + 1,), dtype=torch.int64, device=device)
        torch.cumsum(counts, dim=0, out=offsets[1:])
        # the one host synchronisation: the size of the list -- and, with it, the status word of the binning pass (the last
        # 256-byte block of the workspace holds the status words, csrc/neighbors.hip nl_layout): an atom more than 400 cells
        # outside the unit cell has its wrap integer clamped, i.e. its shifts would be wrong -- an error, not a result
        tail = torch.cat([offsets[-1:].to(torch.int64), ws[-256:-248].view(torch.int32)[1:2].to(torch.int64)]).cpu()
        P = int(tail[0])
        if int(tail[1]) & 4:
            raise ValueError("an atom lies more than 400 cells of the neighbour grid outside the unit cell: wrap the positions "
                             "into the cell (the shifts of the list would be wrong)")
        pairs = torch.empty((P, 2), dtype=torch.int64, device=device)
        shifts = torch.empty((P, 3), dtype=dtype, device=device)
        dist = torch.empty((P,), dtype=dtype, device=device)
        if P > 0:
            _lib.check(lib.mipme_nl_fill(st, dt, C.byref(desc), N, ws.data_ptr(), offsets.data_ptr(), pairs.data_ptr(),
                                         shifts.data_ptr(), dist.data_ptr()))
    return pairs, shifts, dist


class NeighborStream:
    """Neighbour list that lives on the GPU in the format the fused pair kernels read, refreshed IN PLACE.

    The reference hands a fresh ``(P, 2)`` list to every call (``examples/02-neighbor-lists-usage.py:97-164``); an MD or
    training loop on the GPU would pay, per refresh, the list build, an int64 round trip, a radix-sort transposition and the
    repacking into the kernels' entry stream.  This object owns fixed-size device buffers instead -- for every atom a row of
    ``row_capacity`` 4-byte words ``partner | cell-shift code << 22`` of all its neighbours (``mipme_nl_stream``) -- and
    :meth:`update` rewrites them from the current positions with two library calls, nothing read back by the host: it can be
    captured into a HIP graph (``GraphedEnergyForces(..., neighbors=stream)`` does, and replays it from ``refresh()``), and
    a captured energy + forces step keeps reading the same addresses.

    Use it through the reference API with its two handles::

        nl = NeighborStream(positions, cell, cutoff)          # builds the first list
        d = nl.distances(positions, cell)                      # "virtual" distances: autograd link to positions / cell
        V = calculator(charges, cell, positions, nl.indices, d)
        ...
        nl.update()                                            # after the atoms moved (same positions tensor), in place

    ``nl.indices`` (1, 2) and the distance tensor (1,) are opaque handles whose contents are undefined: only calculators of
    this package understand them, and only single-channel evaluations without a pair mask (everything else raises).  :meth:`pairs` returns the list in the reference's format when one is needed.
    The rows hold every neighbour of every atom, i.e. they stand for a HALF list whatever ``full_neighbor_list`` says (the
    result is the same).

    :param positions: (N, 3) device tensor; :meth:`update` without argument re-reads THIS tensor
    :param cell: (3, 3) device tensor (read once; build a new stream for a new cell)
    :param cutoff: neighbours closer than this (strictly) are listed -- include the skin here
    :param periodic: per-axis periodicity
    :param row_capacity: words per row; default: 1.2 x the longest row of the first build + 8, rounded up to 16
    """

    def __init__(self, positions, cell, cutoff: float, periodic=(True, True, True), row_capacity: int | None = None):
        import ctypes as C

        import torch

        from . import _lib

        _lib.require_device(positions, "positions")
        self._lib = lib = _lib.load()
        self.positions = positions
        self.cell = cell
        self.cutoff = float(cutoff)
        self.periodic = tuple(bool(p) for p in periodic)
        self.device, self.dtype = positions.device, positions.dtype
        self.n_atoms = N = positions.shape[0]
        if N > (1 << 22):
            raise ValueError("a NeighborStream addresses at most 2^22 atoms (4-byte entries)")
        self._dt = _lib.dtype_code(self.dtype)
        pos = positions.detach()
        # (N,3) contiguous, or the (N,3) view of (N,4) records x, y, z, charge (GraphedEnergyForces keeps its atoms that way)
        if pos.dim() != 2 or pos.shape[1] != 3 or pos.stride(1) != 1 or pos.stride(0) not in (3, 4):
            raise ValueError("`positions` must be a contiguous (N, 3) tensor (the stream re-reads this tensor in place)")
        self._desc = _nl_descriptor(cell.detach().to("cpu", torch.float64).numpy(), cutoff, self.periodic, pos, True)
        self._desc.position_stride = int(pos.stride(0)) if N > 1 else 3
        self._ws = torch.zeros((lib.mipme_nl_workspace_bytes(C.byref(self._desc), N),), dtype=torch.uint8, device=self.device)
        self._host = torch.zeros((4,), dtype=torch.int32).pin_memory()
        self._host_np = self._host.numpy()
        self.version = 0  #: number of completed :meth:`update` calls issued from the host (graph replays not counted)
        if row_capacity is None:
            counts = torch.zeros((max(N, 1),), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                st = _lib.current_stream(self.device)
                _lib.check(lib.mipme_nl_bin(st, self._dt, C.byref(self._desc), N, pos.data_ptr(), self._ws.data_ptr()))
                _lib.check(lib.mipme_nl_count(st, self._dt, C.byref(self._desc), N, self._ws.data_ptr(), counts.data_ptr()))
            longest = int(counts.max().item()) if N > 0 else 0
            row_capacity = (int(1.2 * longest) + 8 + 15) // 16 * 16
        self._allocate(int(row_capacity))
        self.update()

    def _allocate(self, row_capacity: int):
        import torch

        N = self.n_atoms
        if N * row_capacity >= (1 << 31):
            raise ValueError(f"{N} rows of {row_capacity} entries exceed 32-bit row offsets")
        self.row_capacity = row_capacity
        #: (3N + 1,) int32: begin, end, end of every row; the last word is the size of the entry buffer
        self.row_ptr = torch.zeros((3 * N + 1,), dtype=torch.int32, device=self.device)
        #: (N * row_capacity + 1,) int32 words ``partner | shift code << 22``
        # zero-filled ONCE: the packed pair kernel prefetches one step past the end of a row and multiplies what it finds by
        # a zero weight -- the padding must decode to a valid atom / shift code (0 | 0 does; so does any stale entry), never
        # to whatever the allocator left behind
        self.words = torch.zeros((N * row_capacity + 1,), dtype=torch.int32, device=self.device)
        #: opaque handle for the ``neighbor_indices`` argument of the calculators: a (1, 2) tensor whose contents are undefined
        self.indices = torch.zeros((1, 2), dtype=torch.int32, device=self.device)
        self.indices._mipme_stream = self
        self._ent8 = None  # (version seen, int32 (N * row_capacity + 1, 2)) for the kernels that read 8-byte entries
        self._handle = None

    def update(self, positions=None):
        """Rebuild the list in place from the current values of the positions tensor (two library calls on the current
        stream, no host synchronisation; capturable).  ``positions`` (optional) is copied into the stream's tensor first."""
        import ctypes as C

        import torch

        from . import _lib

        if positions is not None and positions is not self.positions:
            with torch.no_grad():
                self.positions.copy_(positions)
        lib, N = self._lib, self.n_atoms
        with _lib.on_device(self.device):
            st = _lib.current_stream(self.device)
            _lib.check(lib.mipme_nl_bin(st, self._dt, C.byref(self._desc), N, self.positions.data_ptr(), self._ws.data_ptr()))
            _lib.check(lib.mipme_nl_stream(st, self._dt, C.byref(self._desc), N, self._ws.data_ptr(), self.row_capacity,
                                           self.row_ptr.data_ptr(), self.words.data_ptr(), self._host.data_ptr()))
        self.version += 1
        return self

    # ---- status ------------------------------------------------------------------------------------------------------
    @property
    def longest_row(self) -> int:
        """Longest row of the last refresh that has completed on the device (pinned word, no synchronisation)."""
        return int(self._host_np[0])

    @property
    def refreshes(self) -> int:
        """Refreshes completed on the device so far (graph replays included)."""
        return int(self._host_np[2])

    def check(self, synchronize: bool = False) -> None:
        """Raise if a completed refresh dropped entries or met a cell shift the pair kernels cannot encode.  Costs a read of
        pinned host memory; with ``synchronize=True`` the stream is drained first, so the LAST refresh is covered."""
        import torch

        if synchronize:
            torch.cuda.current_stream(self.device).synchronize()
        flags = int(self._host_np[1])
        if flags & 1:
            raise RuntimeError(
                f"NeighborStream: a row needs {int(self._host_np[0])} entries but holds {self.row_capacity}; "
                "call grow() (and re-capture any graph that replays this stream)")
        if flags & 2:
            raise RuntimeError("NeighborStream: a cell shift beyond +-3 (atoms far outside the unit cell, or a cutoff of "
                               "several box lengths): wrap the positions, or use neighbor_list_device()")
        if flags & 4:
            raise RuntimeError("NeighborStream: an atom lies more than 400 cells outside the unit cell")

    def grow(self, row_capacity: int | None = None):
        """Re-allocate with a larger row capacity (default: 1.25 x the longest row seen + 16) and rebuild.  The buffers move:
        graphs captured over the old ones must be captured again."""
        if row_capacity is None:
            row_capacity = (int(1.25 * max(self.longest_row, self.row_capacity)) + 16 + 15) // 16 * 16
        self._allocate(int(row_capacity))
        return self.update()

    # ---- what the calculators consume --------------------------------------------------------------------------------------
    def distances(self, positions=None, cell=None):
        """The handle for the ``neighbor_distances`` argument: a "virtual" distance tensor (``ops.pair_distances(...,
        deferred="virtual")``) that links the calculator's result to ``positions`` / ``cell`` in the autograd graph; the
        fused pair kernel forms the distances in registers from those tensors.  Pass the tensors the calculator is called
        with (default: the stream's own)."""
        from . import ops

        return ops.stream_distances(self, self.positions if positions is None else positions,
                                    self.cell if cell is None else cell)

    def entries8(self):
        """(N * row_capacity + 1, 2) int32 ``{partner, shift code}`` view of the current words, for the kernels that read
        8-byte entries (gradients other than the energy's).  Rebuilt when the list has been refreshed from the host; a graph
        replay of a refresh is NOT seen here -- those kernels are not part of a captured step."""
        import torch

        c = self._ent8
        if c is not None and c[0] == (self.version, self.refreshes):
            return c[1]
        w = self.words
        ent8 = torch.stack((w & ((1 << 22) - 1), (w >> 22) & 511), dim=1).contiguous()
        self._ent8 = ((self.version, self.refreshes), ent8)
        return ent8

    def pairs(self, full_list: bool = False):
        """The current list in the reference's format: ``pairs`` (P,2) int64, ``shifts`` (P,3), ``distances`` (P,) -- a
        separate walk over the same binned atoms (one host synchronisation for P)."""
        return neighbor_list_device(self.positions, self.cell, self.cutoff, full_list=full_list, periodic=self.periodic)

    @property
    def n_entries(self) -> int:
        """Entries in the stream right now (twice the number of half-list pairs); synchronises."""
        r = self.row_ptr[: 3 * self.n_atoms].view(-1, 3)
        return int((r[:, 2] - r[:, 0]).sum().item())

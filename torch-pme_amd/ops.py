"""Autograd operators of the hot path, each a thin shim over the C-ABI of ``libmipme.so``.

* :class:`MeshGeometry` -- host-side cell-derived quantities (what ``MeshInterpolator.update`` and
  ``KSpaceFilter._prep_kvectors`` cache in the reference).
* :func:`pme_potential` -- ``Calculator.forward`` (reference ``calculators/calculator.py:103-189`` with
  ``calculators/pme.py:88-143``) as ONE autograd node: SR pair sum + mesh LR part, differentiable w.r.t.
  charges, cell, positions and neighbor_distances (first order).
* :func:`pair_distances` -- the caller-side distance op (reference ``tests/helpers.py:278-304``),
  differentiable w.r.t. positions and cell.

PyTorch is used for device memory, streams and the autograd tape only.
"""

from __future__ import annotations

import ctypes as C
import os
import weakref
from collections import OrderedDict

import numpy as np
import torch

from . import _front, _lib


SECOND_ORDER_HINT = (
    "torchpme_amd: the HIP kernels provide FIRST-order gradients; this graph is being differentiated twice (create_graph=True, "
    "e.g. a loss on forces).  Set `calculator.double_backward = \"analytic\"` before the forward call (the call is evaluated "
    "through differentiable primitives, exact to any order; \"auto\": only when a backward pass is itself recorded), or "
    "`calculator.double_backward = "
    "\"finite-difference\"`: the second derivative is then formed from central differences of the analytic first-order "
    "gradients (two more evaluations per double-backward pass; use float64).  Distances from `pair_distances` differentiate "
    "twice exactly either way.")


def first_order(fn):
    """``torch.autograd.function.once_differentiable`` with an error message that names the way out (reference behaviour:
    plain ATen ops differentiate any number of times, ``calculators/calculator.py:103-189``)."""
    import functools

    @functools.wraps(fn)
    def wrapper(ctx, *args):
        with torch.no_grad():
            outputs = fn(ctx, *args)
        if not torch.is_grad_enabled():
            return outputs
        # create_graph=True: the results depend on the node's SAVED inputs as well, whether or not the incoming gradient
        # carries a graph (torch's once_differentiable only looks at the latter and then returns constants silently): every
        # output points at a node that raises when something differentiates through it
        single = not isinstance(outputs, tuple)
        outs = (outputs,) if single else outputs
        err = torch._C._functions.DelayedError(SECOND_ORDER_HINT.encode(), len(outs))

        def fake_requires_grad(v):
            if v is not None:
                v = v.detach()
                v.requires_grad = True
            return v

        res = err(*[fake_requires_grad(v) for v in outs])
        return res[0] if single else res

    return wrapper


# Optional per-call timing used by bench.py: when PROFILE is a dict, every C-ABI call below is bracketed by
# HIP events recorded on the launch stream (torch.cuda.Event records on the current stream, which is the
# stream handed to libmipme).  PROFILE[name] collects (start, end) event pairs.  (While it is set the compiled front end,
# whose C-ABI calls do not pass through _call, stands aside.)
PROFILE = None


def _call(name, fn, *args):
    """Invoke a C-ABI entry point, raise on a non-zero status, optionally time it with HIP events."""
    if PROFILE is None:
        _lib.check(fn(*args))
        return
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    _lib.check(fn(*args))
    end.record()
    PROFILE.setdefault(name, []).append((start, end))


class MeshGeometry:
    """Cell-derived host quantities for one (cell, ns_mesh, scheme, order) combination."""

    def __init__(self, cell_host: np.ndarray, ns, scheme: int, order: int):
        A = np.ascontiguousarray(np.asarray(cell_host, dtype=np.float64).reshape(3, 3))
        det = float(np.linalg.det(A))
        if det == 0.0 or not np.isfinite(det):
            raise ValueError(f"provided `cell` has a determinant of {det}, i.e. it is not a valid unit cell")
        self.cell = A
        self.inv_cell = np.linalg.inv(A)
        self.volume = abs(det)
        self.ns = tuple(int(n) for n in ns)
        self.scheme = scheme
        self.order = order
        #: where the FFT plans (with their brick counters) of this geometry live: the calculators point it at their own dict,
        #: so that a calculator keeps ONE plan per mesh across cells, concurrently running calculators never share one, and
        #: the plan lives as long as the calculator (captured graphs hold raw pointers into it)
        self.plan_store = None

    def desc(self, n_channels: int) -> _lib.MeshDesc:
        """``mipme_mesh_t`` of this geometry (one struct per channel count, built once: the struct is read-only for the
        library)."""
        cache = self.__dict__.setdefault("_desc_cache", {})
        d = cache.get(n_channels)
        if d is not None:
            return d
        d = cache[n_channels] = _lib.MeshDesc()
        d.scheme, d.order = self.scheme, self.order
        d.nx, d.ny, d.nz = self.ns
        d.n_channels = n_channels
        d.cell[:] = self.cell.ravel().tolist()
        d.inv_cell[:] = self.inv_cell.ravel().tolist()
        d.volume = self.volume
        return d

    @property
    def n_mesh(self) -> int:
        return self.ns[0] * self.ns[1] * self.ns[2]

    @property
    def n_half(self) -> int:
        return self.ns[0] * self.ns[1] * (self.ns[2] // 2 + 1)


def ns_mesh_from_cell(cell_host: np.ndarray, mesh_spacing: float):
    """2^ceil(log2(2 |a_d| / h + 1)) per axis (reference ``lib/kvectors.py:5-21``)."""
    norms = np.linalg.norm(np.asarray(cell_host, dtype=np.float64).reshape(3, 3), axis=1)
    return tuple(int(v) for v in 2 ** np.ceil(np.log2(2 * norms / mesh_spacing + 1)).astype(np.int64))


def build_filter(geom: MeshGeometry, pot_desc: _lib.PotentialDesc, dtype, device) -> torch.Tensor:
    """G(k) on the rfft half grid, computed on the device in fp64 and stored in ``dtype``."""
    G = torch.empty((geom.ns[0], geom.ns[1], geom.ns[2] // 2 + 1), dtype=dtype, device=device)
    md = geom.desc(1)
    with _lib.on_device(device):
        _lib.check(
            _lib.load().mipme_kfilter_build(
                _lib.current_stream(device), _lib.dtype_code(dtype), C.byref(md), C.byref(pot_desc), G.data_ptr()
            )
        )
    return G


def filter_derivative(geom: MeshGeometry, pot_desc: _lib.PotentialDesc, dtype, device) -> torch.Tensor:
    """Derivative table of G(k) (``mipme_kfilter_build_deriv``: 4 reals per half-grid point) for the cell gradient of an energy
    step, built on first use and kept with the geometry it belongs to (a new cell makes a new geometry, hence a new table)."""
    key = (pot_desc.kind, pot_desc.exponent, pot_desc.smearing, pot_desc.prefactor, dtype, str(device))
    cache = geom.__dict__.setdefault("_deriv_cache", {})
    D = cache.get(key)
    if D is None:
        D = torch.empty((geom.ns[0], geom.ns[1], geom.ns[2] // 2 + 1, 4), dtype=dtype, device=device)
        md = geom.desc(1)
        with _lib.on_device(device):
            _lib.check(_lib.load().mipme_kfilter_build_deriv(_lib.current_stream(device), _lib.dtype_code(dtype), C.byref(md),
                                                             C.byref(pot_desc), D.data_ptr()))
        cache.clear()
        cache[key] = D
    return D


# How the pair kernels accumulate per-atom results:
#   "rows"   (default) owner-computes sums over a transposed pair list (csrc/topology.hip); needs a
#            one-off build per neighbour-list tensor, no atomics, deterministic.
#   "atomic" one pass over the list with hardware float atomics (csrc/rspace.hip); no preprocessing.
PAIR_MODE = "rows"

#: constant ``neighbor_distances`` (the same tensor, unmodified, seen a second time by the same list): v_SR(d) per row entry is
#: tabulated once and the pair sum becomes a sparse matrix-vector product (PairTopology.tabulated); "0" disables
TABULATE = True

# How atoms meet the mesh: "bricks" (default; atoms binned by 8^3 mesh brick, owner-computes LDS-tile spread,
# LDS-tiled gathers -- csrc/bricks.hip) or "atomic" (global float atomics -- csrc/mesh.hip).  Meshes too small
# for bricks always take the atomic kernels.
MESH_MODE = "bricks"

# When the gradient arriving at the calculator's backward was produced by ``weighted_sum(V, charges)`` (E = sum q V, the
# reduction every energy/force evaluation performs) it equals gE * charges and the adjoint mesh is a multiple of the
# forward mesh: the backward then skips the second spread + FFT pair.  Any other upstream gradient takes the general path.
#: run the short-range pair sum inside the spread launch of the mesh part (see mipme_sr_job_t in include/mipme.h)
COSCHEDULE = True
ENERGY_FAST_PATH = True
#: 4-byte entries {partner | shift code << 22} for the co-scheduled pair sum (half the entry stream; needs < 2^22 atoms)
COMPACT_ENTRIES = True
#: bytes per entry of the pair stream the co-scheduled kernel reads (bench.py's algorithmic byte count)
FUSED_ENTRY_BYTES = 4 if COMPACT_ENTRIES else 8
#: energy reduction + force assembly inside the gather launch when the forward can tell they will be wanted (see the tail
#: block of _PMEFunction.forward and _EnergyDirectSum)
TAIL_FUSION = True
#: a device scalar the caller promises to seed the next backward pass with (set by GraphedEnergyForces around its
#: evaluation): the gather's tail then writes seed * dE/dpositions and the backward pass launches nothing
SEED_PROMISE = None
#: (charges, cell, aux_seed) asked of the gather's tail irrespective of requires_grad (see seed_promise)
TAIL_REQUEST = (False, False, None)
#: a graphed.EnergyLog the gather's tail appends the energy to (see seed_promise)
TAIL_LOG = None


class seed_promise:
    """``with seed_promise(seed): V = calculator(...)`` -- the caller promises to seed the backward pass of ``weighted_sum(V,
    charges)`` with the device scalar ``seed`` (unchanged until then), e.g. ``E.backward(seed)`` with ``seed = -1`` to get the
    forces.  The gather's tail then writes ``seed * dE/dpositions`` in the forward pass and the backward launches nothing; any
    other seed still gives the right gradient through the ordinary backward kernels.

    ``charges=True`` / ``cell=True`` ask the same tail for ``aux_seed * dE/dcharges`` and ``aux_seed * dE/dcell`` as well, whether or
    not those tensors require a gradient (``aux_seed``: device scalar, default = ``seed``): a caller that reads the buffers of the
    node's ``tail`` itself -- :class:`~torchpme_amd.graphed.GraphedEnergyForces` -- gets forces (seed -1) and the derivatives
    w.r.t. charges and cell (aux seed +1) from one forward pass without a sign-flip launch."""

    def __init__(self, seed, charges: bool = False, cell: bool = False, aux_seed=None, energy_log=None):
        self.seed, self.want_q, self.want_cell, self.aux_seed = seed, bool(charges), bool(cell), aux_seed
        #: frame farm: an ``EnergyLog`` of one energy per evaluation that the tail's energy is also appended to (the node's
        #: ``tail["logged"]`` says whether the tail took it)
        self.energy_log = energy_log

    def __enter__(self):
        global SEED_PROMISE, TAIL_REQUEST, TAIL_LOG
        self._prev, SEED_PROMISE = SEED_PROMISE, self.seed
        self._prev_req, TAIL_REQUEST = TAIL_REQUEST, (self.want_q, self.want_cell, self.aux_seed)
        self._prev_log, TAIL_LOG = TAIL_LOG, self.energy_log
        return self

    def __exit__(self, *exc):
        global SEED_PROMISE, TAIL_REQUEST, TAIL_LOG
        SEED_PROMISE = self._prev
        TAIL_REQUEST = self._prev_req
        TAIL_LOG = self._prev_log
        return False
#: recognise an energy gradient (grad == gE * charges) that carries no tag from ``weighted_sum`` by comparing on the device
ENERGY_DETECT = True
#: compiled host side of the reference call sequence for its common case (csrc/front.cpp, _front.py); "0": Python nodes only
FRONT = os.environ.get("MIPME_FRONT", "1") != "0"
#: ... and act on the verdict ON THE DEVICE (mipme_set_skip_flag / mipme_energy_select) where only position gradients are asked
#: for, instead of polling it on the host: the poll makes the host wait for everything queued before it, i.e. the eager
#: reference call sequence ran GPU and host one after the other
DEVICE_SELECT = True
#: ... from this many atoms on: below, the eager step is bound by the host (0.3 ms of Python for 0.13 ms of kernels at 32k atoms)
#: and the extra launches of the skipped general adjoint cost more host time than the poll (measured on one box: 0.39 against
#: 0.33-0.37 ms at 31 944 atoms, 0.69 against 0.75 ms at 262 144)
DEVICE_SELECT_MIN_ATOMS = 65536

# Reciprocal-space convolution as (y,z) plane transforms + one kernel doing x-FFT, * G and the inverse x-FFT (power-of-two
# nx); "0" keeps the 3-D hipFFT plans + filter kernel.
XFUSED = True


# When ``neighbor_distances`` is the untouched output of :func:`pair_distances`, the calculator differentiates straight through
# to the positions / cell the distances were built from: the row kernels recompute d from the L2-resident positions (one
# gather per entry instead of random reads of P-sized arrays) and neither ``d`` nor ``dL/dd`` is read or written by the
# calculator.  The gradient then reaches ``positions`` directly, NOT via ``neighbor_distances`` -- set to 0 if you need
# ``torch.autograd.grad(E, neighbor_distances)`` for such a tensor (a leaf ``neighbor_distances`` is never fused).
FUSE_DISTANCES = True


def _pack_row_shifts(entries, shifts, n_pairs):
    """int32 (2P + 1,) with the 3 cell shifts of every row entry as int8, or None if they are not small integers."""
    lib = _lib.load()
    device = shifts.device
    packed = torch.zeros((2 * n_pairs + 1,), dtype=torch.int32, device=device)
    flag = torch.empty((1,), dtype=torch.int32, device=device)
    with _lib.on_device(device):
        _lib.check(lib.mipme_topology_pack_shifts(_lib.current_stream(device), _lib.dtype_code(shifts.dtype), n_pairs,
                                                  entries.data_ptr(), shifts.data_ptr(), packed.data_ptr(), flag.data_ptr()))
    return None if int(flag.item()) != 0 else packed


def _pack_entries(row_ptr, entries, shifts, n_pairs, n_atoms, table=True):
    """``(ent_sh, shift_format)``: int32 (2P + 1, 2) {other atom, role-adjusted cell-shift code}, table codes (format 1) if asked
    for and possible, else 3 x int8 (format 0); ``(None, 0)`` if the shifts are not small integers."""
    lib = _lib.load()
    device = entries.device
    ent_sh = torch.zeros((2 * n_pairs + 1, 2), dtype=torch.int32, device=device)
    flag = torch.empty((1,), dtype=torch.int32, device=device)
    fmt = 1 if table else 0
    while True:
        with _lib.on_device(device):
            _lib.check(lib.mipme_topology_pack_entries(
                _lib.current_stream(device), _lib.dtype_code(shifts.dtype) if shifts is not None else _lib.F32, n_pairs, n_atoms,
                row_ptr.data_ptr(), entries.data_ptr(), _lib.ptr(shifts), fmt, ent_sh.data_ptr(), flag.data_ptr()))
        bits = int(flag.item()) if shifts is not None else 0
        if bits & 1:
            return None, 0
        if fmt == 1 and bits & 2:
            fmt = 0  # shifts beyond the table range: repack as 3 x int8
            continue
        return ent_sh, fmt


# A NEW neighbour-list (or shifts) tensor is first assumed to hold the values of the previous one: the reference's users hand a
# fresh list to every call (examples/02-neighbor-lists-usage.py:97-164), often the same values in a new tensor (a clone, a data
# loader's copy), and the per-list structures -- radix-sort transposition, entry streams: 0.6 ms at cfg3 -- would be rebuilt for
# nothing.  The bet: ``mipme_checksum`` hashes the new tensor on the device and compares with the hash of the tensor the cached
# structures were built from; the verdict lands in pinned memory and is looked at (``verify_bets``) before any result that used
# the cached structures is handed out; a lost bet repeats the evaluation with freshly built structures.  "0": always rebuild.
SPECULATE_LISTS = True
_BETS: list = []  # pending verdicts: (pinned flags numpy view, slot, undo callable)
_ON_LOST: list = []  # side effects of work done on adopted structures, to take back if any pending bet is lost
_BET_PAUSE = [0]  # list misses to sit out after a lost bet (lists that really change from call to call)
_BET_FLAGS: dict = {}
#: polls of a pinned verdict word that gave up and synchronised instead (diagnostics: should stay at zero)
SPIN_TIMEOUTS = {"match": 0, "bets": 0, "cell": 0}


class SpeculationLost(Exception):
    """A bet on unchanged list values was lost: the caller repeats its work (the stale cache entries are already gone)."""


def _bet_slot():
    st = _BET_FLAGS.get("flags")
    if st is None:
        flags = torch.zeros((64,), dtype=torch.int32).pin_memory()
        st = _BET_FLAGS["flags"] = [flags, flags.numpy(), 0]
    st[2] = (st[2] + 1) % 64
    st[1][st[2]] = -1
    return st[0], st[1], st[2]


def device_checksum(t: torch.Tensor, expect: torch.Tensor | None = None, undo=None) -> torch.Tensor:
    """128-bit checksum of ``t`` (contiguous, 16-byte aligned storage) as a device tensor; with ``expect`` the comparison's
    verdict is queued for :func:`verify_bets` together with ``undo`` (what to forget if the bet is lost)."""
    lib = _lib.load()
    sums = torch.zeros((lib.mipme_checksum_words(),), dtype=torch.int64, device=t.device)
    flag_ptr = None
    if expect is not None:
        flags, view, slot = _bet_slot()
        flag_ptr = flags.data_ptr() + 4 * slot
        _BETS.append((view, slot, undo))
    with _lib.on_device(t.device):
        _lib.check(lib.mipme_checksum(_lib.current_stream(t.device), t.data_ptr(), t.numel() * t.element_size(), sums.data_ptr(),
                                      _lib.ptr(expect), flag_ptr))
    return sums


def verify_bets() -> None:
    """Look at the verdicts of the bets made since the last call (polling pinned words: the comparisons were queued before the
    kernels that used the cached structures); raise :class:`SpeculationLost` after undoing every lost one."""
    if not _BETS:
        return
    lost = False
    for view, slot, undo in _BETS:
        spins = 0
        while view[slot] == -1:
            spins += 1
            if spins > 5_000_000:
                SPIN_TIMEOUTS["bets"] += 1
                torch.cuda.synchronize()
                break
        if view[slot] != 1:
            lost = True
            if undo is not None:
                undo()
    _BETS.clear()
    on_lost = list(_ON_LOST)
    _ON_LOST.clear()
    if lost:
        for f in on_lost:
            f()
        _BET_PAUSE[0] = 16
        raise SpeculationLost()


def _abandon_bets() -> None:
    """An exception left a betting scope before its verdicts were looked at: treat every pending bet as lost (forget the adopted
    structures), so that no later caller finds an unverified entry."""
    for _view, _slot, undo in _BETS:
        if undo is not None:
            undo()
    _BETS.clear()
    for f in _ON_LOST:
        f()
    _ON_LOST.clear()


class betting:
    """``with betting(): ...`` -- the caller WILL call :func:`verify_bets` before it hands out anything computed inside.  Bets are
    only placed inside such a scope (a path that looks a list up without verifying -- the Ewald and direct calculators, the graph
    classes -- always builds or finds the structures of exactly its tensor); an exception that leaves the scope takes every
    pending bet back.  Equality is decided by a position-keyed 2 x 64-bit checksum (``mipme_checksum``: any single changed word and
    any two swapped words change it), i.e. with certainty against accidents, not against an adversary.  The pending-bet lists are
    module globals: one host thread per process evaluates calculators with bets (``ops.SPECULATE_LISTS = False`` otherwise)."""

    def __enter__(self):
        _BET_SCOPE[0] += 1
        return self

    def __exit__(self, *exc):
        _BET_SCOPE[0] -= 1
        if exc[0] is not None and _BET_SCOPE[0] == 0:
            _abandon_bets()
        return False


_BET_SCOPE = [0]


def _can_bet(t: torch.Tensor) -> bool:
    return (SPECULATE_LISTS and _BET_SCOPE[0] > 0 and _BET_PAUSE[0] == 0 and len(_BETS) < 32 and t.is_contiguous()
            and t.data_ptr() % 16 == 0
            and (t.numel() * t.element_size()) % 4 == 0 and not torch.cuda.is_current_stream_capturing())


class PairTopology:
    """Transposed pair list of one ``neighbor_indices`` tensor (see ``include/mipme.h``)."""

    #: row-layout flags OR-ed into the shift format handed to the library (0: rows that share their boundaries)
    fmt_flags = 0

    def __init__(self, pairs: torch.Tensor, n_atoms: int):
        lib = _lib.load()
        device = pairs.device
        P = pairs.shape[0]
        self.n_atoms, self.n_pairs = n_atoms, P
        self.row_ptr = torch.empty((2 * n_atoms + 1,), dtype=torch.int32, device=device)
        # one slot more than 2P: the row kernels prefetch entry `row begin` even for an empty last row (index 2P), never used
        self.entries = torch.empty((2 * P + 1, 2), dtype=torch.int32, device=device)
        self.entries[2 * P].zero_()  # (the build writes 2P entries: a 76 MB memset per list at cfg3 for one slot otherwise)
        self._packed = None  # (weakref(shifts), version, tensor|None)
        self._ent_sh = None  # (weakref(shifts)|None, version, tensor|None)
        self._pair_sh = None  # (weakref(shifts), version, tensor|None)
        self._ent32 = None  # (weakref(shifts)|None, version, tensor|None)
        self._sorted = None
        # what the structures were built from, for the bet on a new tensor with the same values (see SPECULATE_LISTS)
        self.src_dtype = pairs.dtype
        self._sum = device_checksum(pairs) if (SPECULATE_LISTS and pairs.is_contiguous() and pairs.data_ptr() % 16 == 0
                                               and not torch.cuda.is_current_stream_capturing()) else None
        self._shift_key = self._shift_sum = self._shift_meta = None
        # 8-byte (i, j) copy of an int64 list for the two kernels that stream the list in pair order
        self.pairs32 = pairs if pairs.dtype == torch.int32 else pairs.to(torch.int32)
        with _lib.on_device(device):
            nbytes = lib.mipme_topology_workspace_bytes(P)
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
            _lib.check(
                lib.mipme_topology_build(
                    _lib.current_stream(device), _lib.index_code(pairs.dtype), P, n_atoms, pairs.data_ptr(),
                    ws.data_ptr(), nbytes, self.row_ptr.data_ptr(), self.entries.data_ptr(),
                )
            )

    def tabulated(self, dist: torch.Tensor, mask, pot_desc, full_list: bool):
        """``(values, row_sum_transposed)`` of ``mipme_rspace_rows_tabulate`` for a ``neighbor_distances`` tensor that this
        list has ALREADY been evaluated with (same tensor, same version, same potential, same mask): a charge loop over a fixed
        geometry, the reference tuner's timing protocol (tuning/tuner.py:337-373).  ``None`` on the first sighting -- which only
        remembers the tensor -- so that distances that are new every call never pay for the table (76 MB and one pass over the
        list at 4.76 M pairs)."""
        key = (bytes(pot_desc), bool(full_list), None if mask is None else (mask.data_ptr(), mask._version))
        c = self.__dict__.get("_tab")
        if c is None or c[0]() is not dist or c[1] != dist._version or c[2] != key:
            self._tab = (weakref.ref(dist), dist._version, key, None)
            return None
        if c[3] is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            lib = _lib.load()
            device, dt = dist.device, _lib.dtype_code(dist.dtype)
            values = torch.empty((lib.mipme_rspace_rows_value_bytes(dt, self.n_pairs),), dtype=torch.uint8, device=device)
            row_sum = torch.empty((self.n_atoms, 1), dtype=dist.dtype, device=device)
            with _lib.on_device(device):
                _call("rspace_tabulate", lib.mipme_rspace_rows_tabulate, _lib.current_stream(device), dt, self.n_atoms,
                      self.row_ptr.data_ptr(), self.entries.data_ptr(), dist.data_ptr(), _lib.ptr(mask), int(full_list),
                      C.byref(pot_desc), values.data_ptr(), row_sum.data_ptr())
            c = self._tab = c[:3] + ((values, row_sum),)
        return c[3]

    def adopt_shifts(self, shifts: torch.Tensor, key: torch.Tensor | None = None) -> None:
        """Called with the shifts of an evaluation before the shift-keyed streams are asked for.  If they were built for ANOTHER
        tensor of the same shape and dtype, bet that it held the same values (SPECULATE_LISTS): re-key the streams to the new
        tensor and queue the checksum comparison; otherwise remember this tensor's checksum for the next such bet."""
        key = shifts if key is None else key
        cur = self._shift_key
        if cur is not None and cur[0]() is key and cur[1] == key._version:
            return
        meta = (tuple(shifts.shape), shifts.dtype)
        if self._shift_sum is not None and self._shift_meta == meta and _can_bet(shifts):
            def undo(self=self):
                self._packed = self._ent_sh = self._pair_sh = self._ent32 = None
                self.__dict__.pop("_front", None)
                self.__dict__.pop("_front_lazy", None)
                self._shift_key = self._shift_sum = None

            device_checksum(shifts, self._shift_sum, undo)
            wr, ver = weakref.ref(key), key._version
            if self._packed is not None:
                self._packed = (wr, ver) + tuple(self._packed[2:])
            if self._pair_sh is not None:
                self._pair_sh = (wr, ver) + tuple(self._pair_sh[2:])
            if self._ent32 is not None and self._ent32[0] is not None:
                self._ent32 = (wr, ver) + tuple(self._ent32[2:])
            if isinstance(self._ent_sh, dict):
                self._ent_sh = {k: ((wr, ver) + tuple(v[2:]) if v[0] is not None else v) for k, v in self._ent_sh.items()}
            self.__dict__.pop("_front", None)  # (the handle holds the old tensors; rebuilt from the cached streams)
            self._shift_key = (wr, ver)
            return
        self._packed = self._pair_sh = self._ent32 = None
        self._ent_sh = None
        self.__dict__.pop("_front", None)
        self.__dict__.pop("_front_lazy", None)
        self._shift_key = (weakref.ref(key), key._version)
        self._shift_meta = meta
        self._shift_sum = device_checksum(shifts) if (SPECULATE_LISTS and shifts.is_contiguous() and shifts.data_ptr() % 16 == 0
                                                      and not torch.cuda.is_current_stream_capturing()) else None

    def front(self, pairs: torch.Tensor, shifts: torch.Tensor, key: torch.Tensor):
        """Handle of this list (with the shift streams of ``key``) for the compiled front end (``_front.py``), or ``None``
        when its kernels do not apply (shifts that are not small integers, more than 2^22 atoms, ...).  Cached per shifts
        tensor; the first call packs every stream the fast path reads."""
        c = self.__dict__.get("_front")
        if c is not None and c[0]() is key and c[1] == key._version and c[2] == shifts.dtype:
            return c[3]
        handle = None
        mod = _front.module()
        if mod is not None and self.fmt_flags == 0 and self.n_pairs > 0:
            pair_packed = self.pair_packed_shifts(shifts, key)
            ent32 = self.compact_entries(shifts, key)  # exists <=> integer shifts within the table range
            if pair_packed is not None and ent32 is not None:
                # the two streams only the general adjoints read are made on first use (a closure over tensors, not over this
                # object: the handle may outlive it inside an autograd graph, and must not keep it alive in a cycle)
                row_ptr, entries, n_pairs, n_atoms = self.row_ptr, self.entries, self.n_pairs, self.n_atoms
                # (kept with the list, not with the handle: a new list tensor with the same values gets a new handle -- see
                # adopt_shifts -- and must not pack them again: 0.15 ms at cfg3 for a stream that only skipped kernels name)
                made = self.__dict__.setdefault("_front_lazy", {})

                def lazy(kind, shifts=shifts, made=made):
                    packed = made.get(kind)
                    if packed is not None:
                        return packed
                    if kind == "row_packed":
                        packed = _pack_row_shifts(entries, shifts, n_pairs)
                    else:
                        packed, fmt = _pack_entries(row_ptr, entries, shifts, n_pairs, n_atoms, table=True)
                        packed = packed if fmt == 1 else None
                    if packed is None:  # cannot happen when the 4-byte stream exists
                        raise RuntimeError(f"no {kind} stream for this list")
                    made[kind] = packed
                    return packed

                handle = mod.Topology(pairs, shifts, self.pairs32, pair_packed, self.row_ptr, self.entries, ent32, self.n_atoms,
                                      lazy)
        self._front = (weakref.ref(key), key._version, shifts.dtype, handle)
        return handle

    def front_plain(self, pairs: torch.Tensor):
        """Handle of this list for the compiled calculator node on caller-made distances (front.cpp, PlainTopo): the row
        structure only, no shift streams.  Cached per list tensor."""
        c = self.__dict__.get("_front_plain")
        if c is not None and c[0]() is pairs and c[1] == pairs._version:
            return c[2]
        mod = _front.module()
        handle = None
        if mod is not None and self.fmt_flags == 0 and self.n_pairs > 0:
            handle = mod.PlainTopology(pairs, self.row_ptr, self.entries, self.n_atoms)
        self._front_plain = (weakref.ref(pairs), pairs._version, handle)
        return handle

    @property
    def sorted_by_first(self) -> bool:
        """True if ``pairs[:, 0]`` is non-decreasing (neighbour-list builders emit it so): the role-i entries of a row are
        then consecutive pairs.  One device reduction + D2H read per list, cached."""
        if self._sorted is None:
            first = self.pairs32[:, 0]
            self._sorted = bool((first[1:] >= first[:-1]).all().item()) if self.n_pairs > 1 else True
        return self._sorted

    def packed_shifts(self, shifts: torch.Tensor, key: torch.Tensor | None = None):
        """int32 per entry holding the 3 cell shifts as int8, or None if the shifts are not small integers
        (the kernel then reads ``shifts[p]`` itself).  Cached per shifts tensor; the integrality flag is the
        only D2H read, once per (list, shifts) pair."""
        key = shifts if key is None else key  # cache on the caller's tensor (``shifts`` may be a converted copy)
        c = self._packed
        if c is not None and c[0]() is key and c[1] == key._version:
            return c[2]
        lib = _lib.load()
        device = shifts.device
        packed = torch.zeros((2 * self.n_pairs + 1,), dtype=torch.int32, device=device)
        flag = torch.empty((1,), dtype=torch.int32, device=device)
        with _lib.on_device(device):
            _lib.check(
                lib.mipme_topology_pack_shifts(
                    _lib.current_stream(device), _lib.dtype_code(shifts.dtype), self.n_pairs, self.entries.data_ptr(),
                    shifts.data_ptr(), packed.data_ptr(), flag.data_ptr(),
                )
            )
        if int(flag.item()) != 0:
            packed = None
        self._packed = (weakref.ref(key), key._version, packed)
        return packed


    def pair_packed_shifts(self, shifts: torch.Tensor, key: torch.Tensor | None = None):
        """int32 (P,) with the 3 cell shifts of every pair as int8 (list order), or None if they are not small integers."""
        key = shifts if key is None else key
        c = self._pair_sh
        if c is not None and c[0]() is key and c[1] == key._version:
            return c[2]
        lib = _lib.load()
        device = shifts.device
        packed = torch.empty((max(self.n_pairs, 1),), dtype=torch.int32, device=device)
        flag = torch.empty((1,), dtype=torch.int32, device=device)
        with _lib.on_device(device):
            _lib.check(lib.mipme_pack_pair_shifts(_lib.current_stream(device), _lib.dtype_code(shifts.dtype), self.n_pairs,
                                                  shifts.data_ptr(), packed.data_ptr(), flag.data_ptr()))
        if int(flag.item()) != 0:
            packed = None
        self._pair_sh = (weakref.ref(key), key._version, packed)
        return packed

    def compact_entries(self, shifts: torch.Tensor | None, key: torch.Tensor | None = None):
        """int32 (2P + 1,) entry stream of the co-scheduled pair sum, ``other | code << 22`` (format 2, see ``mipme.h``), or
        ``None`` when it does not apply (more than 2^22 atoms, shifts beyond the table range or not integers).  Cached per
        shifts tensor like the 8-byte streams."""
        if not COMPACT_ENTRIES or self.n_atoms > (1 << 22):
            return None
        c = self._ent32
        if c is not None and ((key is None and c[0] is None) or (key is not None and c[0] is not None and c[0]() is key
                                                                   and c[1] == key._version)):
            return c[2]
        lib = _lib.load()
        device = self.entries.device
        ent32 = torch.empty((2 * self.n_pairs + 1,), dtype=torch.int32, device=device)
        ent32[2 * self.n_pairs :].zero_()
        flag = torch.empty((1,), dtype=torch.int32, device=device)
        with _lib.on_device(device):
            _lib.check(
                lib.mipme_topology_pack_entries(
                    _lib.current_stream(device), _lib.dtype_code(shifts.dtype) if shifts is not None else _lib.F32,
                    self.n_pairs, self.n_atoms, self.row_ptr.data_ptr(), self.entries.data_ptr(), _lib.ptr(shifts), 2,
                    ent32.data_ptr(), flag.data_ptr(),
                )
            )
        if shifts is not None and int(flag.item()) != 0:
            ent32 = None
        self._ent32 = (None if key is None else weakref.ref(key), 0 if key is None else key._version, ent32)
        return ent32

    def entries_with_shifts(self, shifts: torch.Tensor | None, key: torch.Tensor | None = None, table: bool = True):
        """``(ent_sh, shift_format)``: the int32 (2P, 2) stream {other atom, role-adjusted cell-shift code} of the fused
        kernels, or ``(None, 0)`` if the shifts are not small integers.  ``table`` asks for the LDS-table code (format 1,
        |s| <= 3), falling back to 3 x int8 (format 0).  Cached per (shifts tensor, requested format)."""
        key = shifts if key is None else key
        c = self._ent_sh.get(bool(table)) if isinstance(self._ent_sh, dict) else None
        if c is not None and ((key is None and c[0] is None) or (key is not None and c[0] is not None and c[0]() is key
                                                                   and c[1] == key._version)):
            return c[2], c[3]
        lib = _lib.load()
        device = self.entries.device
        ent_sh = torch.zeros((2 * self.n_pairs + 1, 2), dtype=torch.int32, device=device)
        flag = torch.empty((1,), dtype=torch.int32, device=device)
        fmt = 1 if table else 0
        while True:
            with _lib.on_device(device):
                _lib.check(
                    lib.mipme_topology_pack_entries(
                        _lib.current_stream(device), _lib.dtype_code(shifts.dtype) if shifts is not None else _lib.F32,
                        self.n_pairs, self.n_atoms, self.row_ptr.data_ptr(), self.entries.data_ptr(), _lib.ptr(shifts),
                        fmt, ent_sh.data_ptr(), flag.data_ptr(),
                    )
                )
            bits = int(flag.item()) if shifts is not None else 0
            if bits & 1:
                ent_sh = None
            elif fmt == 1 and bits & 2:
                fmt = 0  # shifts beyond the table range: repack as 3 x int8
                continue
            break
        if not isinstance(self._ent_sh, dict):
            self._ent_sh = {}
        self._ent_sh[bool(table)] = (None if key is None else weakref.ref(key), 0 if key is None else key._version, ent_sh, fmt)
        return ent_sh, fmt


class DistanceSource:
    """Provenance of a distance tensor made by :func:`pair_distances` (attached to it as ``_mipme_src``)."""

    __slots__ = ("positions", "cell", "pairs", "shifts", "shifts_key", "versions", "dist_ref", "pending", "direct", "virtual")

    def __init__(self, positions, cell, pairs, shifts, shifts_key, dist, pending=False, virtual=False):
        self.positions, self.cell, self.pairs, self.shifts, self.shifts_key = positions, cell, pairs, shifts, shifts_key
        self.versions = (positions._version, None if cell is None else cell._version, pairs._version, dist._version)
        self.dist_ref = weakref.ref(dist)
        #: True while the values of a ``deferred=True`` tensor have not been written yet
        self.pending = pending
        #: how the pair part of a calculator's gradient reaches ``positions`` / ``cell`` when the fused kernels are used:
        #: False (default) -- through this tensor's own autograd node, as ``dL/d(neighbor_distances)`` in lazy form
        #: (:class:`LazyPairGradient`), so ``torch.autograd.grad(E, d)`` and hooks on ``d`` behave as in the reference;
        #: True (``pair_distances(..., deferred=True)``: explicit opt-in) -- straight to ``positions`` / ``cell``; the
        #: distance tensor is then NOT part of the result's autograd graph.
        self.direct = pending
        #: True (``deferred="virtual"``): a calculator that forms the distances inside its fused pair kernel does NOT store them
        #: either -- the tensor stays unwritten (``pending``) until some consumer that needs the values calls materialize()
        self.virtual = bool(virtual) and pending

    def materialize(self) -> None:
        """Write the values of a deferred distance tensor with the stand-alone distance kernel (for consumers other than
        the fused pair kernel, which produces them as a by-product)."""
        dist = self.dist_ref()
        if self.pending and dist is not None:
            sh = self.shifts
            _launch_pair_distances(self.positions.detach().contiguous(), None if self.cell is None else
                                   self.cell.detach().contiguous(), self.pairs.contiguous(),
                                   None if sh is None else sh.contiguous(), self.shifts_key, dist.detach())
        self.pending = False

    def usable_for(self, dist, pairs, n_channels) -> bool:
        """True if ``dist`` still equals ``|r_j - r_i + S cell|`` of the recorded tensors and ``pairs`` is the same list."""
        if not FUSE_DISTANCES or PAIR_MODE != "rows" or n_channels != 1:
            return False
        if self.dist_ref() is not dist or pairs is not self.pairs or dist.retains_grad:
            return False
        now = (self.positions._version, None if self.cell is None else self.cell._version, pairs._version, dist._version)
        return now == self.versions and dist.dtype == self.positions.dtype and dist.device == self.positions.device


_MATCH_FLAGS: dict = {}


def _match_flag(device):
    """Pinned int32 word (tensor, NumPy view) the energy-gradient detection kernel reports to -- one per (device, host
    thread): autograd runs the backward passes of different GPUs on different threads, and a shared word would let one thread
    read the other's verdict."""
    import threading

    key = (device.index if device.index is not None else torch.cuda.current_device(), threading.get_ident())
    slot = _MATCH_FLAGS.get(key)
    if slot is None:
        t = torch.zeros((1,), dtype=torch.int32).pin_memory()
        slot = _MATCH_FLAGS[key] = (t, t.numpy())
    return slot


class _LazyEntries8:
    """8-byte ``{partner, shift code}`` entries of a :class:`NeighborStream`, expanded from its 4-byte words the first time
    a kernel asks for their address (the energy + forces path never does)."""

    def __init__(self, stream):
        self._stream = stream

    def data_ptr(self):
        return self._stream.entries8().data_ptr()


class StreamTopology:
    """What the pair kernels need of a :class:`~torchpme_amd.neighbors.NeighborStream`: the rows the device neighbour list
    wrote directly (``mipme_nl_stream``: padded rows, every neighbour once per row, 4-byte words) in the clothes of a
    :class:`PairTopology`.  There are no pair indices behind it: no pair mask, no stored distances, no (P,) gradients."""

    fmt_flags = _lib.ROWS_PADDED

    def __init__(self, stream):
        self.stream = stream
        self.n_atoms = stream.n_atoms
        self.entries = torch.zeros((1, 2), dtype=torch.int32, device=stream.device)  # never read (no mask / distance output)
        self.sorted_by_first = False

    @property
    def row_ptr(self):
        return self.stream.row_ptr

    @property
    def n_pairs(self):
        return self.stream.indices.shape[0]

    @property
    def pairs32(self):
        raise NotImplementedError("a NeighborStream has no (P,2) pair list; call stream.pairs() for one in the reference's format")

    def compact_entries(self, shifts=None, key=None):
        return self.stream.words if COMPACT_ENTRIES else None

    def entries_with_shifts(self, shifts=None, key=None, table: bool = True):
        if not table:
            raise NotImplementedError("a NeighborStream carries no pair indices: `pair_mask` is not supported with it")
        return _LazyEntries8(self.stream), 1


_TOPOLOGIES: "OrderedDict" = OrderedDict()


def get_topology(pairs: torch.Tensor, n_atoms: int) -> PairTopology:
    """Topology of ``pairs``, cached on the identity and version counter of the tensor object: keep the
    neighbour-list tensor alive across steps (a Verlet list) and the transposition is paid once."""
    stream = getattr(pairs, "_mipme_stream", None)
    if stream is not None:  # the handle of a NeighborStream: rows written by the device neighbour list, nothing to build
        topo = stream._handle
        if topo is None:
            topo = stream._handle = StreamTopology(stream)
        return topo
    key = (pairs.data_ptr(), pairs.shape[0], n_atoms, pairs.device.index)
    hit = _TOPOLOGIES.get(key)
    if hit is not None and hit[0]() is pairs and hit[1] == pairs._version:
        _TOPOLOGIES.move_to_end(key)
        return hit[2]
    if _BET_PAUSE[0] > 0:
        _BET_PAUSE[0] -= 1
    elif _can_bet(pairs):
        # a new tensor: bet that it holds the values of the list the most recent matching topology was built from
        for k_old in reversed(_TOPOLOGIES):
            t_old = _TOPOLOGIES[k_old][2]
            if (k_old[1:] == key[1:] and isinstance(t_old, PairTopology) and t_old._sum is not None
                    and t_old.src_dtype == pairs.dtype):
                device_checksum(pairs, t_old._sum, lambda key=key: _TOPOLOGIES.pop(key, None))
                _TOPOLOGIES[key] = (weakref.ref(pairs), pairs._version, t_old)
                t_old.__dict__.pop("_front", None)  # (the compiled front end's handle names the old list tensor)
                while len(_TOPOLOGIES) > 16:
                    _TOPOLOGIES.popitem(last=False)
                return t_old
    topo = PairTopology(pairs, n_atoms)
    _TOPOLOGIES[key] = (weakref.ref(pairs), pairs._version, topo)
    while len(_TOPOLOGIES) > 16:
        _TOPOLOGIES.popitem(last=False)
    return topo


def _slab_axis(periodic_host):
    if periodic_host is None:
        return None
    p = [bool(v) for v in periodic_host]
    if sum(p) != 2:
        return None
    return p.index(False)


class LazyPairGradient(torch.Tensor):
    """``dL/d(neighbor_distances)`` of a calculator call whose pair part ran in the fused distance + pair kernels.

    Those kernels differentiate through ``d = |r_j - r_i + S cell|`` in the same pass, so what they produce is already the
    product of this gradient with the Jacobian of the distances: ``dL/dpositions`` (N,3) and ``dL/dcell`` (3,3) of the
    pair part.  The (P,) gradient itself is only a way-point of the chain rule -- unless somebody looks at it.  The
    calculator's backward therefore hands autograd this placeholder for the ``neighbor_distances`` slot:

    * the autograd node of :func:`pair_distances` recognises it and returns the two small products it carries -- no P-sized
      array is written or read (the reference moves ``P (16 + 3s)`` bytes here);
    * any other consumer -- ``torch.autograd.grad(E, d)``, a hook on ``d``, the accumulation with a second consumer's
      gradient -- touches it through an ATen op, and the first such op materialises the true (P,) tensor with the
      stand-alone adjoint kernel (``mipme_rspace_backward``), after which it is an ordinary tensor in every respect.

    So the autograd contract of the reference (output differentiable w.r.t. ``neighbor_distances``,
    ``tests/calculators/test_workflow.py:164-192``) holds by default and the fast path stays fast.  A wrapper subclass
    (no storage of its own); ``materialize()`` returns the plain tensor."""

    @staticmethod
    def __new__(cls, n_pairs, dtype, device, grad_pos, grad_cell, make):
        t = torch.Tensor._make_wrapper_subclass(cls, (n_pairs,), dtype=dtype, device=device, requires_grad=False)
        t._grad_pos, t._grad_cell, t._make, t._value = grad_pos, grad_cell, make, None
        return t

    __torch_function__ = torch._C._disabled_torch_function_impl

    def materialize(self) -> torch.Tensor:
        if self._value is None:
            self._value = self._make()
            self._make = None
        return self._value

    @property
    def materialized(self) -> bool:
        return self._value is not None

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map

        def plain(x):
            return x.materialize() if isinstance(x, LazyPairGradient) else x

        return func(*tree_map(plain, args), **tree_map(plain, kwargs or {}))


def _scaled_match(lib, st, dt, n, g, q, res, flag_ptr):
    """``mipme_scaled_match`` (is g == s * q for one scalar s?), in its many-block form beyond 32 768 values."""
    nw = lib.mipme_scaled_match_work(n) if n > 32768 else 0
    if nw == 0:
        _call("scaled_match", lib.mipme_scaled_match, st, dt, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag_ptr)
    else:
        work = torch.empty((nw,), dtype=torch.float64, device=g.device)
        _call("scaled_match", lib.mipme_scaled_match_wide, st, dt, n, g.data_ptr(), q.data_ptr(), res.data_ptr(), flag_ptr,
              work.data_ptr())


class _SkipGuard:
    """Clears the library's per-thread skip flag (``mipme_set_skip_flag``) on the way out of a backward pass, whatever happened in
    between: the flag points at a tensor that dies with the pass."""

    armed = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.armed:
            _lib.load().mipme_set_skip_flag(None)
        return False


class _PMEFunction(torch.autograd.Function):
    """SR + LR per-atom potentials as one autograd node."""

    @staticmethod
    def forward(ctx, charges, cell, positions, neighbor_distances, neighbor_indices, pair_mask, geom, G, pot_desc,
                full_list, slab_axis, src_positions=None, src_cell=None, src=None, lazy=False, nan_flag=None,
                atomic_pairs=False):
        lib = _lib.load()
        device, dtype = positions.device, positions.dtype
        dt = _lib.dtype_code(dtype)
        N, Cn = charges.shape
        P = neighbor_indices.shape[0]
        q = charges.detach().contiguous()
        pos = positions.detach().contiguous()
        dist = neighbor_distances.detach().contiguous()
        pairs = neighbor_indices.contiguous()
        mask = None if pair_mask is None else pair_mask.contiguous()
        out = torch.empty((N, Cn), dtype=dtype, device=device)
        need_cell = ctx.needs_input_grad[1]
        saved = {}
        field = tail = None
        with _lib.on_device(device):
            st = _lib.current_stream(device)
            # atomic_pairs: one pass over the list with float atomics -- for lists that are new every call (the flattened pair
            # list of a padded batch), where building the transposed list would cost more than it saves
            topo = get_topology(pairs, N) if (PAIR_MODE == "rows" and not atomic_pairs) else None
            if topo is not None and topo.fmt_flags:
                full_list = False  # rows of a NeighborStream hold every neighbour once: a half list whatever the calculator says
            fused = None
            if src is not None:
                if isinstance(topo, PairTopology) and src.shifts is not None:
                    topo.adopt_shifts(src.shifts, src.shifts_key)
                ent_sh, shift_fmt = topo.entries_with_shifts(src.shifts, src.shifts_key, table=mask is None)
                if ent_sh is not None:
                    fused = dict(
                        ent_sh=ent_sh, fmt=shift_fmt | topo.fmt_flags, pos=src_positions.detach().contiguous(),
                        cell=None if src_cell is None else src_cell.detach().contiguous(), force=None, partials=None,
                        records=torch.empty((N, 4), dtype=dtype, device=device),
                    )
                    # speculative force sums: if the backward turns out to be in energy mode they ARE the SR forces (the
                    # matching cell sums: decided below, once it is known whether the gather tail covers the cell gradient)
                    if ENERGY_FAST_PATH and ctx.needs_input_grad[11]:
                        fused["force"] = torch.empty((N, 3), dtype=dtype, device=device)
            # deferred distances (``pair_distances(..., deferred=True)``): the fused kernel writes them as a by-product
            write_dist = False
            if src is not None and src.pending:
                if src.virtual and fused is not None:
                    pass  # nobody reads the values: the fused kernel forms the distances in registers and drops them
                else:
                    write_dist = fused is not None and mask is None and P > 0 and topo.sorted_by_first
                    if not write_dist:
                        src.materialize()

            want_pair_partials = bool(fused is not None and fused["force"] is not None and src_cell is not None
                                      and ctx.needs_input_grad[12])
            if want_pair_partials and geom is None:
                fused["partials"] = torch.empty((lib.mipme_rows_partials_size(N),), dtype=torch.float64, device=device)

            # distances that are a plain tensor this list has been evaluated with before: the tabulated pair sum
            tab = None
            if (TABULATE and fused is None and topo is not None and Cn == 1 and P > 0 and not ctx.needs_input_grad[3]
                    and type(neighbor_distances) is torch.Tensor and neighbor_distances.is_contiguous()):
                tab = topo.tabulated(neighbor_distances, mask, pot_desc, full_list)  # (keyed on the caller's tensor object)
            ctx.tab = tab

            def run_rspace(accumulate):
                stream = _lib.current_stream(device)
                if fused is not None:
                    _call(
                        "rspace_forward", lib.mipme_sr_rows_fused,
                        stream, dt, N, topo.row_ptr.data_ptr(), fused["ent_sh"].data_ptr(), topo.entries.data_ptr(),
                        _lib.ptr(mask), fused["pos"].data_ptr(), _lib.ptr(fused["cell"]), q.data_ptr(), q.data_ptr(), None,
                        0, int(full_list), C.byref(pot_desc), accumulate, fused["fmt"], fused["records"].data_ptr(),
                        int(fused.get("records_ready", False)), out.data_ptr(), _lib.ptr(fused["force"]),
                        _lib.ptr(fused["partials"]), None, dist.data_ptr() if write_dist else None,
                    )
                    if write_dist:
                        src.pending = False
                        if _BETS:  # rows adopted on a bet: if it is lost, the rerun (or materialize()) writes the values again
                            _ON_LOST.append(lambda src=src: setattr(src, "pending", True))
                elif topo is not None:
                    if tab is not None:  # constant distances: v_SR(d) per row entry is already there (PairTopology.tabulated)
                        _call(
                            "rspace_forward", lib.mipme_rspace_rows_tabulated,
                            stream, dt, N, topo.row_ptr.data_ptr(), tab[0].data_ptr(), q.data_ptr(), 0, int(full_list),
                            accumulate, out.data_ptr(),
                        )
                        return
                    _call(
                        "rspace_forward", lib.mipme_rspace_rows,
                        stream, dt, N, Cn, topo.row_ptr.data_ptr(), topo.entries.data_ptr(), dist.data_ptr(),
                        q.data_ptr(), _lib.ptr(mask), 0, int(full_list), C.byref(pot_desc), accumulate, out.data_ptr(),
                    )
                else:
                    _call(
                        "rspace_forward", lib.mipme_rspace_forward,
                        stream, dt, _lib.index_code(pairs.dtype), P, N, Cn, pairs.data_ptr(), dist.data_ptr(),
                        q.data_ptr(), _lib.ptr(mask), int(full_list), C.byref(pot_desc), accumulate, out.data_ptr(),
                    )

            if geom is not None:
                md = geom.desc(Cn)
                plan = _lib.get_plan(device, dtype, geom.ns, Cn, geom.plan_store)
                cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
                rho_mesh = torch.empty((Cn,) + geom.ns, dtype=dtype, device=device)
                phi_mesh = torch.empty((Cn,) + geom.ns, dtype=dtype, device=device)
                # rfftn(rho) itself is only needed by the cell gradient; without it the convolution runs fused (see mipme.h)
                rho_hat = cell_partials = None
                if not plan.xfused or not XFUSED:
                    rho_hat = torch.empty((Cn, geom.n_half), dtype=cdtype, device=device)
                hat_work = torch.empty((Cn, geom.n_half), dtype=cdtype, device=device)
                dc = torch.empty((Cn,), dtype=dtype, device=device)
                phi_atoms = torch.empty((N, Cn), dtype=dtype, device=device) if need_cell else None
                bins = None
                if MESH_MODE == "bricks":
                    nbytes = lib.mipme_atom_bins_bytes(C.byref(md), N, dt)
                    if nbytes > 0:
                        bins = torch.empty((nbytes,), dtype=torch.uint8, device=device)
                # speculative: the mesh force per unit gE q_a, formed by the same gather (see the backward's energy mode)
                if ENERGY_FAST_PATH and bins is not None and Cn == 1 and ctx.needs_input_grad[2] and slab_axis is None:
                    field = torch.empty((N, 3), dtype=dtype, device=device)
                # the binning pass can emit the (position, charge) records of the fused pair kernel for free
                records_out = None
                if fused is not None and bins is not None and Cn == 1 and src_positions is positions:
                    records_out = fused["records"]
                # co-scheduled pair sum: the spread launch also carries the row workgroups of the fused distance + pair kernel
                # (mipme_sr_job_t); the gather then adds the mesh part to the potentials the pair sum wrote
                ni = ctx.needs_input_grad
                p_eff = 1 if pot_desc.kind == _lib.COULOMB else pot_desc.exponent
                cosched = bool(COSCHEDULE and records_out is not None and mask is None and (fused["fmt"] & 0xFF) == 1 and N > 0)
                # the 4-byte entry stream is read by the co-scheduled kernel only (mipme.h, shift_format 2): use it when the
                # library will co-schedule -- force sums wanted, 1/r or 1/r^6 with a smearing and no exclusion radius
                ent32 = None
                if (cosched and fused["force"] is not None and p_eff in (1, 6) and pot_desc.smearing > 0
                        and pot_desc.exclusion_radius <= 0):
                    ent32 = topo.compact_entries(src.shifts, src.shifts_key)
                # Tail of an energy step in the gather launch (mipme_kspace_forward_args_t.out_energy ...): speculative like the
                # force sums.  If the caller reduces with weighted_sum(V, charges) (see energy_direct below), E and the assembled
                # position gradient are already there and neither the energy reduction nor the force assembly is launched; and
                # the rest of the contract rides along (out_grad_charges / out_grad_cell): dE/dq = 2 V for a half list, dE/dcell
                # from partial sums of the same launches + one single-workgroup launch -- those two also serve the energy mode
                # of THIS node's backward (lazy distances, plain tensor reductions).
                symmetric = (not full_list) or bool(topo is not None and topo.fmt_flags)
                want_q, want_cell, aux_seed = TAIL_REQUEST
                tail_q = bool(ni[0] or want_q) and symmetric
                tail_cell = (bool(ni[1] or ni[12] or want_cell) and ent32 is not None and not write_dist and slab_axis is None
                             and src_cell is not None)
                tail_ok = bool(
                    TAIL_FUSION and cosched and field is not None and fused["force"] is not None and ENERGY_FAST_PATH
                    and not ni[3] and (tail_q or not ni[0]) and (tail_cell or not (ni[1] or ni[12])) and rho_hat is None
                    and p_eff in (1, 6) and pot_desc.exclusion_radius <= 0
                    and (not lazy or tail_q or tail_cell))  # (lazy distances: only for what the tail adds to this node)
                if not tail_ok:
                    tail_q = tail_cell = False
                if not tail_cell:
                    if want_pair_partials:
                        # the pair part's cell sums from the generic pair body (no co-scheduled launch with them)
                        fused["partials"] = torch.empty((lib.mipme_rows_partials_size(N),), dtype=torch.float64, device=device)
                        cosched = False
                    if need_cell and rho_hat is None and Cn > 1:
                        # speculative, like the force sums: the k-grid sums of the cell gradient for the energy mode, formed by
                        # the x stage of the fused convolution while rfftn(rho) is in LDS (the backward then needs neither rho^
                        # nor the 3-D plans; a general upstream gradient recomputes rho^ from the saved charge mesh)
                        cell_partials = torch.empty((lib.mipme_cellgrad_partials_size(C.byref(md), N),), dtype=torch.float64,
                                                    device=device)
                # one channel: the x stage keeps rfftn(rho) itself (a 1 MB store at 64^3) -- the backward pass forms the k-grid
                # sums from it in either mode, the general one with the fused convolution (mipme.h, G_deriv)
                rho_keep = None
                if need_cell and rho_hat is None and Cn == 1:
                    rho_keep = torch.empty((Cn, geom.n_half), dtype=cdtype, device=device)
                job = None
                if cosched:
                    job = _lib.SrJob(
                        n_atoms=N, row_ptr=topo.row_ptr.data_ptr(),
                        entries_shift=(fused["ent_sh"] if ent32 is None else ent32).data_ptr(),
                        entries=topo.entries.data_ptr(), positions=fused["pos"].data_ptr(), cell=_lib.ptr(fused["cell"]),
                        charges=q.data_ptr(), pot=C.pointer(pot_desc), full_list=int(full_list),
                        shift_format=fused["fmt"] if ent32 is None else (2 | topo.fmt_flags),
                        records=records_out.data_ptr(), out=out.data_ptr(), force=_lib.ptr(fused["force"]),
                        dist_out=dist.data_ptr() if write_dist else None,
                    )
                if tail_ok:
                    seed = SEED_PROMISE
                    if seed is not None and (seed.dtype != dtype or seed.device != device or seed.numel() != 1):
                        seed = None
                    if aux_seed is not None and (aux_seed.dtype != dtype or aux_seed.device != device or aux_seed.numel() != 1):
                        aux_seed = None
                    tail = dict(energy=torch.empty((), dtype=dtype, device=device),
                                grad=torch.empty((N, 3), dtype=dtype, device=device), seed=seed, grad_q=None, grad_cell=None,
                                aux_seed=aux_seed)
                    if tail_q:
                        tail["grad_q"] = torch.empty((N, 1), dtype=dtype, device=device)
                    if TAIL_LOG is not None and TAIL_LOG.n_frames == 1 and TAIL_LOG.values.device == device:
                        tail["log"], tail["logged"] = TAIL_LOG, True
                    if tail_cell:
                        tail["grad_cell"] = torch.empty((27,), dtype=dtype, device=device)
                        tail["G_deriv"] = filter_derivative(geom, pot_desc, dtype, device)
                        tail["cell_work"] = torch.empty((lib.mipme_cell_tail_work(plan.handle, C.byref(md), N),),
                                                        dtype=torch.float64, device=device)
                keep_rho_mesh = (cell_partials is not None or tail_cell) and rho_keep is None and rho_hat is None
                args = _lib.KspaceForwardArgs(
                    plan=plan.handle, stream=st, dtype=dt, accumulate_out=1 if job is not None else 0,
                    mesh=C.pointer(md), pot=C.pointer(pot_desc), n_atoms=N, positions=pos.data_ptr(), charges=q.data_ptr(),
                    G=G.data_ptr(), rho_mesh=rho_mesh.data_ptr(), rho_hat=_lib.ptr(rho_hat), hat_work=hat_work.data_ptr(),
                    phi_mesh=phi_mesh.data_ptr(), dc=dc.data_ptr(), out_lr=out.data_ptr(), out_phi=_lib.ptr(phi_atoms),
                    atom_bins=_lib.ptr(bins), gather_wait_event=None,
                    out_field=_lib.ptr(field), out_records=_lib.ptr(records_out),
                    sr_job=C.pointer(job) if job is not None else None, out_cell_partials=_lib.ptr(cell_partials),
                    out_energy=None if tail is None else tail["energy"].data_ptr(),
                    out_grad_positions=None if tail is None else tail["grad"].data_ptr(),
                    grad_seed=None if tail is None else _lib.ptr(tail["seed"]), nan_flag=nan_flag,
                    out_grad_charges=None if tail is None else _lib.ptr(tail["grad_q"]),
                    out_grad_cell=None if tail is None else _lib.ptr(tail["grad_cell"]),
                    G_deriv=None if tail is None else _lib.ptr(tail.get("G_deriv")),
                    cell_work=None if tail is None else _lib.ptr(tail.get("cell_work")),
                    aux_seed=None if tail is None else _lib.ptr(tail["aux_seed"]), out_rho_hat=_lib.ptr(rho_keep),
                    # the charge mesh is only read again (fft_r2c in the backward pass) if rfftn(rho) was not kept
                    flags=0 if keep_rho_mesh else _lib.FWD_RHO_MESH_UNUSED,
                    energy_log=None if tail is None or "log" not in tail else tail["log"].values.data_ptr(),
                    energy_log_cursor=None if tail is None or "log" not in tail else tail["log"].cursor.data_ptr(),
                    energy_log_capacity=0 if tail is None or "log" not in tail else tail["log"].capacity,
                )
                _call("kspace_forward", lib.mipme_kspace_forward, C.byref(args))
                if records_out is not None:
                    fused["records_ready"] = True
                if job is not None and write_dist:
                    src.pending = False
                    if _BETS:
                        _ON_LOST.append(lambda src=src: setattr(src, "pending", True))
                if slab_axis is not None:
                    moments = torch.empty((6 * Cn,), dtype=torch.float64, device=device)
                    _call(
                        "slab_forward", lib.mipme_slab_forward, st, dt, slab_axis, C.byref(md), pot_desc.prefactor, N,
                        pos.data_ptr(), q.data_ptr(), moments.data_ptr(), out.data_ptr(),
                    )
                ctx.rho_kept = rho_keep is not None
                saved = dict(phi_mesh=phi_mesh, rho_hat=(rho_hat if rho_keep is None else rho_keep) if need_cell else None,
                             rho_dc=dc, phi_atoms=phi_atoms,
                             bins=bins, rho_mesh=rho_mesh if keep_rho_mesh else None,
                             cell_partials=cell_partials)
                if job is None:
                    run_rspace(1)
            else:
                run_rspace(0)
        ctx.save_for_backward(q, pos, dist, pairs, mask, G, *(saved.get(k) for k in ("phi_mesh", "rho_hat", "rho_dc", "phi_atoms", "bins")), out)
        ctx.rho_mesh, ctx.cell_partials = saved.get("rho_mesh"), saved.get("cell_partials")
        ctx.field = field
        ctx.tail = tail  # {energy, grad, seed} written by the gather's tail, or None
        ctx.same_positions = src_positions is positions
        #: the pair part's gradient leaves through the neighbor_distances slot as a LazyPairGradient (see pme_potential)
        ctx.lazy = bool(lazy) and fused is not None
        ctx.geom, ctx.pot_desc, ctx.full_list, ctx.slab_axis = geom, pot_desc, full_list, slab_axis
        ctx.topo = topo
        ctx.fused = fused  # plain tensors made here, none of them an input or output of this node
        # E = weighted_sum(out, charges) can be differentiated without this node (see _EnergyDirectSum) when its gradient is
        # gE q_a (f F_a + field_a) with both per-atom sums already formed above and nothing else asks for a gradient
        ni = ctx.needs_input_grad
        covered = tail is not None and (tail["grad_q"] is not None or not ni[0]) and (
            tail["grad_cell"] is not None or not (ni[1] or ni[12]))
        ctx.energy_direct = bool(
            ENERGY_FAST_PATH and Cn == 1 and fused is not None and fused["force"] is not None and src_positions is positions
            and (covered or not (ni[0] or ni[1] or ni[12])) and not ni[3] and slab_axis is None
            and (geom is None or field is not None) and not lazy
        )
        ctx.src_cell_is_cell = src_cell is cell
        return out

    @staticmethod
    @first_order
    def backward(ctx, grad_out):
        lib = _lib.load()
        q, pos, dist, pairs, mask, G, phi_mesh, rho_hat, rho_dc, phi_atoms, bins, out = ctx.saved_tensors
        geom, pot_desc, fused, topo = ctx.geom, ctx.pot_desc, ctx.fused, ctx.topo
        need_q, need_cell, need_pos, need_dist = ctx.needs_input_grad[:4]
        lazy = ctx.lazy
        if lazy:  # the (P,) gradient is only formed if somebody other than the distance node asks for it (see below)
            need_dist = False
        need_src_pos = fused is not None and ctx.needs_input_grad[11]
        need_src_cell = fused is not None and fused["cell"] is not None and ctx.needs_input_grad[12]
        device, dtype = pos.device, pos.dtype
        dt = _lib.dtype_code(dtype)
        N, Cn = q.shape
        P = pairs.shape[0]
        full = int(ctx.full_list)
        g = grad_out.contiguous()
        grad_q = grad_pos = grad_cell = grad_dist = grad_src_pos = grad_src_cell = None
        guard = _SkipGuard()
        with _lib.on_device(device), guard:
            st = _lib.current_stream(device)
            do_kspace = geom is not None and (need_q or need_cell or need_pos)
            # Energy mode: if the upstream gradient was produced by ``weighted_sum(V, charges)`` with OUR charges it is
            # exactly gE * charges.  Then (a) the adjoint mesh is a multiple of the forward one (no second spread / FFT),
            # (b) the forces are gE q_a times the per-atom sums the forward pass already formed (``field`` from the gather,
            # ``force`` from the fused pair kernel), (c) for a half list dL/dq = gE * V (V is a symmetric bilinear form).
            tag = getattr(grad_out, "_mipme_scaled", None) if ENERGY_FAST_PATH else None
            gscale = sr_scale = select = None
            if tag is not None and tag[0] == q.data_ptr() and tag[1] == tuple(q.shape) and tag[2] == q._version:
                sr_scale = tag[3]  # enough for the pair part
            elif (ENERGY_FAST_PATH and ENERGY_DETECT and tag is None and N > 0 and (fused is not None or do_kspace)
                  and not (N > 1 and grad_out.stride(0) == 0)  # an expanded scalar (``V.sum().backward()``): not gE * charges
                  and not torch.cuda.is_current_stream_capturing()):
                # no tag: the caller reduced with plain tensor ops, ``(charges * V).sum()`` (README.rst:112-114) -- ask the
                # device whether the gradient is a multiple of the charges (one small kernel + a 2-value read; the general
                # adjoint it saves is a second spread, an FFT pair and a gradient gather).  Not during graph capture.
                res = torch.empty((2,), dtype=dtype, device=device)
                # (rows of a NeighborStream: the general pair kernel reads 8-byte entries, which would have to be expanded from
                # the stream's words on every call -- those keep the host poll)
                on_device = (DEVICE_SELECT and N >= DEVICE_SELECT_MIN_ATOMS and Cn == 1 and fused is not None
                             and fused["force"] is not None
                             and not (topo is not None and topo.fmt_flags)
                             and ctx.slab_axis is None and not (need_q or need_cell or need_dist or need_src_cell)
                             and (geom is None or ctx.field is not None) and (need_pos or need_src_pos))
                if on_device:
                    # only position gradients are wanted: run the GENERAL adjoint below with every kernel told to return at
                    # once when the verdict is a match, then let one kernel replace its outputs by the energy-mode expressions
                    # in that case -- no host read, the host keeps running ahead of the GPU
                    select = dict(res=res, flag=torch.empty((1,), dtype=torch.int32, device=device))
                    _scaled_match(lib, st, dt, N * Cn, g, q, res, select["flag"].data_ptr())
                    lib.mipme_set_skip_flag(select["flag"].data_ptr())
                    guard.armed = True
                else:
                    flag, flag_np = _match_flag(device)
                    flag_np[0] = -1
                    _scaled_match(lib, st, dt, N * Cn, g, q, res, flag.data_ptr())
                    # the kernel also writes its verdict to pinned host memory: poll that word instead of a device-to-host copy
                    # (hipMemcpy of 4 bytes costs 20-30 us on this stack; the poll ends a few us after the kernel does)
                    spins = 0
                    while flag_np[0] == -1:
                        spins += 1
                        if spins > 2_000_000:  # never seen; a stream synchronisation is the fallback
                            SPIN_TIMEOUTS["match"] += 1
                            torch.cuda.current_stream(device).synchronize()
                            break
                    if flag_np[0] == 1:
                        sr_scale = res[:1]
            if sr_scale is not None and ctx.slab_axis is None:
                gscale = sr_scale
            # what the forward's gather tail already holds for exactly this upstream gradient (mipme.h, out_grad_charges /
            # out_grad_cell): dE/dq = 2 V and dE/dcell (mesh part, pair part) per unit of the tail's seed
            tail = ctx.tail
            if gscale is not None and tail is not None and (tail["grad_q"] is not None or tail["grad_cell"] is not None):
                base = tail["aux_seed"] if tail["aux_seed"] is not None else tail["seed"]
                tscale = sr_scale if base is None else sr_scale / base
                if need_q and tail["grad_q"] is not None:
                    # (the tail holds the TOTAL dE/dq = 2 V of E = sum q V; through this node flows the half that comes from
                    # V's dependence on the charges, the other half is the reduction's own)
                    grad_q = tail["grad_q"] * (0.5 * tscale)
                    need_q = False
                if tail["grad_cell"] is not None and (need_cell or need_src_cell):
                    gc = tail["grad_cell"] * tscale
                    if need_cell:
                        grad_cell = gc[0:9].view(3, 3)
                        need_cell = False
                    if need_src_cell:
                        grad_src_cell = gc[9:18].view(3, 3)
                        need_src_cell = False
                do_kspace = geom is not None and (need_q or need_cell or need_pos)
            # the mesh force field of the forward gather serves the forces AND (through cellgrad_finalize) the cell gradient
            field = ctx.field if gscale is not None else None
            cell_partials = ctx.cell_partials
            if need_cell and rho_hat is None and gscale is None and geom is not None:
                # general upstream gradient after a forward that kept the charge mesh instead of rfftn(rho): transform it now
                # (with the calculator's own plan: torch.fft shares hipFFT state with it and broke later plans when tried)
                md0 = geom.desc(Cn)
                rho_hat = torch.empty((Cn, geom.n_half), device=device,
                                      dtype=torch.complex64 if dtype == torch.float32 else torch.complex128)
                _call("fft_r2c", lib.mipme_fft_r2c, _lib.get_plan(device, dtype, geom.ns, Cn, geom.plan_store).handle, st, dt,
                      C.byref(md0), ctx.rho_mesh.data_ptr(), rho_hat.data_ptr())
            energy_q = need_q and gscale is not None and not ctx.full_list
            if need_dist:
                grad_dist = torch.empty((P,), dtype=dtype, device=device)

            def run_grad_dist(with_charges):
                # grad_dist is a per-pair stream (no scatter); in "atomic" mode the same kernel also scatters grad_q
                pl = pairs if topo is None else topo.pairs32
                _call(
                    "rspace_backward", lib.mipme_rspace_backward,
                    _lib.current_stream(device), dt, _lib.index_code(pl.dtype), P, N, Cn, pl.data_ptr(),
                    dist.data_ptr(), q.data_ptr(), _lib.ptr(mask), full, C.byref(pot_desc), g.data_ptr(),
                    _lib.ptr(sr_scale), _lib.ptr(grad_dist), _lib.ptr(grad_q) if with_charges else None,
                )

            if do_kspace and gscale is not None:
                kb_q = need_q and not energy_q
                # with the field at hand the C call only finalises the cell gradient (no gradient gather)
                from_field = field is not None and not kb_q
                kb_pos = (need_pos or need_cell) and not from_field
                if kb_pos or kb_q or need_cell:
                    md = geom.desc(Cn)
                    plan = _lib.get_plan(device, dtype, geom.ns, Cn, geom.plan_store)
                    if kb_pos:
                        grad_pos = torch.empty((N, 3), dtype=dtype, device=device)
                    if kb_q:
                        grad_q = torch.empty((N, Cn), dtype=dtype, device=device)
                    partials, kgrid_ready = None, 0
                    if need_cell:  # energy mode: the k-grid sums come from rho^ alone (no second spread / FFTs) ...
                        grad_cell = torch.empty((3, 3), dtype=dtype, device=device)
                        if cell_partials is not None:  # ... and the forward's fused convolution has already formed them
                            partials, kgrid_ready = cell_partials, plan.kgrid_blocks
                        else:
                            partials = torch.empty((lib.mipme_cellgrad_partials_size(C.byref(md), N),), dtype=torch.float64,
                                                   device=device)
                    args = _lib.KspaceBackwardArgs(
                        plan=plan.handle, stream=st, dtype=dt, mesh=C.pointer(md), pot=C.pointer(pot_desc), n_atoms=N,
                        positions=pos.data_ptr(), charges=q.data_ptr(), grad_out=g.data_ptr(), G=G.data_ptr(),
                        phi_mesh=phi_mesh.data_ptr(), rho_hat=_lib.ptr(rho_hat) if need_cell else None,
                        rho_dc=_lib.ptr(rho_dc), phi_atoms=_lib.ptr(phi_atoms) if need_cell else None,
                        partials=_lib.ptr(partials), grad_positions=_lib.ptr(grad_pos), grad_charges=_lib.ptr(grad_q),
                        grad_cell=_lib.ptr(grad_cell), atom_bins=_lib.ptr(bins), grad_scale=gscale.data_ptr(),
                        mesh_field=_lib.ptr(field) if from_field else None, kgrid_blocks_ready=kgrid_ready,
                    )
                    _call("kspace_backward", lib.mipme_kspace_backward, C.byref(args))
                    if not need_pos:
                        grad_pos = None
                    if need_cell and not from_field:
                        field = None  # the gradient gather above has produced the mesh forces
            elif do_kspace:
                md = geom.desc(Cn)
                plan = _lib.get_plan(device, dtype, geom.ns, Cn, geom.plan_store)
                cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
                psi_mesh = torch.empty((Cn,) + geom.ns, dtype=dtype, device=device)
                chi_mesh = torch.empty((Cn,) + geom.ns, dtype=dtype, device=device)
                psi_hat = None
                fused_cell = bool(need_cell and getattr(ctx, "rho_kept", False) and plan.xfused and XFUSED and Cn == 1)
                if (need_cell and not fused_cell) or not plan.xfused or not XFUSED:
                    psi_hat = torch.empty((Cn, geom.n_half), dtype=cdtype, device=device)
                hat_work = torch.empty((Cn, geom.n_half), dtype=cdtype, device=device)
                dc = torch.empty((Cn,), dtype=dtype, device=device)
                if need_pos or need_cell:
                    grad_pos = torch.empty((N, 3), dtype=dtype, device=device)
                if need_q:
                    grad_q = torch.empty((N, Cn), dtype=dtype, device=device)
                partials = None
                if need_cell:
                    grad_cell = torch.empty((3, 3), dtype=dtype, device=device)
                    partials = torch.empty((lib.mipme_cellgrad_partials_size(C.byref(md), N),), dtype=torch.float64, device=device)
                args = _lib.KspaceBackwardArgs(
                    plan=plan.handle, stream=st, dtype=dt, mesh=C.pointer(md), pot=C.pointer(pot_desc), n_atoms=N,
                    positions=pos.data_ptr(), charges=q.data_ptr(), grad_out=g.data_ptr(), G=G.data_ptr(),
                    phi_mesh=phi_mesh.data_ptr(), rho_hat=_lib.ptr(rho_hat), rho_dc=_lib.ptr(rho_dc),
                    phi_atoms=_lib.ptr(phi_atoms), psi_mesh=psi_mesh.data_ptr(), psi_hat=_lib.ptr(psi_hat),
                    hat_work=hat_work.data_ptr(), chi_mesh=chi_mesh.data_ptr(), dc=dc.data_ptr(),
                    partials=_lib.ptr(partials), grad_positions=_lib.ptr(grad_pos), grad_charges=_lib.ptr(grad_q),
                    grad_cell=_lib.ptr(grad_cell), atom_bins=_lib.ptr(bins),
                    G_deriv=filter_derivative(geom, pot_desc, dtype, device).data_ptr() if fused_cell else None,
                )
                _call("kspace_backward", lib.mipme_kspace_backward, C.byref(args))
                if ctx.slab_axis is not None:
                    moments = torch.empty((6 * Cn,), dtype=torch.float64, device=device)
                    _call(
                        "slab_backward", lib.mipme_slab_backward, st, dt, ctx.slab_axis, C.byref(md), pot_desc.prefactor,
                        N, pos.data_ptr(), q.data_ptr(), g.data_ptr(), moments.data_ptr(), _lib.ptr(grad_pos),
                        _lib.ptr(grad_q), _lib.ptr(grad_cell),
                    )
                if not need_pos:
                    grad_pos = None
            elif need_q and not energy_q:
                grad_q = torch.zeros((N, Cn), dtype=dtype, device=device)
            atomic_q = need_q and topo is None and not energy_q
            if need_dist or atomic_q:
                run_grad_dist(atomic_q)

            # ---- gradients that end in per-atom sums formed by the forward pass (energy mode) ----
            sr_done = (need_src_pos or need_src_cell) and sr_scale is not None and fused["force"] is not None and (
                fused["partials"] is not None or not need_src_cell)
            mesh_done = need_pos and field is not None

            def finalize(force_t, field_t, partials_t, out_pos, out_cell):
                _call(
                    "forces_finalize", lib.mipme_sr_rows_finalize,
                    st, dt, N, _lib.ptr(force_t), _lib.ptr(field_t), q.data_ptr(), sr_scale.data_ptr(), full,
                    _lib.ptr(partials_t), _lib.ptr(out_pos), _lib.ptr(out_cell),
                )

            if sr_done and need_src_cell:
                grad_src_cell = torch.empty((3, 3), dtype=dtype, device=device)
            if sr_done and mesh_done and ctx.same_positions and need_src_pos and not lazy:
                # both parts differentiate the same ``positions`` tensor: one kernel, one gradient
                grad_pos = torch.empty((N, 3), dtype=dtype, device=device)
                finalize(fused["force"], field, fused["partials"], grad_pos, grad_src_cell if need_src_cell else None)
            else:
                if mesh_done:
                    grad_pos = torch.empty((N, 3), dtype=dtype, device=device)
                    finalize(None, field, None, grad_pos, None)
                if sr_done:
                    if need_src_pos:
                        grad_src_pos = torch.empty((N, 3), dtype=dtype, device=device)
                    finalize(fused["force"], None, fused["partials"], grad_src_pos, grad_src_cell if need_src_cell else None)
            if (need_src_pos or need_src_cell) and not sr_done:
                grad_src_pos = torch.empty((N, 3), dtype=dtype, device=device)
                partials = None
                if need_src_cell:
                    grad_src_cell = torch.empty((3, 3), dtype=dtype, device=device)
                    partials = torch.empty((lib.mipme_rows_partials_size(N),), dtype=torch.float64, device=device)
                _call(
                    "rspace_backward", lib.mipme_sr_rows_fused,
                    st, dt, N, topo.row_ptr.data_ptr(), fused["ent_sh"].data_ptr(), topo.entries.data_ptr(),
                    _lib.ptr(mask), fused["pos"].data_ptr(), _lib.ptr(fused["cell"]), q.data_ptr(), None, g.data_ptr(),
                    0, full, C.byref(pot_desc), 0, fused["fmt"], fused["records"].data_ptr(), 0, None, grad_src_pos.data_ptr(),
                    _lib.ptr(partials), _lib.ptr(grad_src_cell), None,
                )
                if not need_src_pos:
                    grad_src_pos = None

            if select is not None:
                lib.mipme_set_skip_flag(None)
                guard.armed = False
                _call("energy_select", lib.mipme_energy_select, st, dt, N, select["res"].data_ptr(), q.data_ptr(),
                      _lib.ptr(fused["force"]), _lib.ptr(ctx.field), full, _lib.ptr(grad_pos) if geom is not None else None,
                      _lib.ptr(grad_src_pos))

            # ---- charge gradient of the pair part ----
            if energy_q:
                grad_q = out * sr_scale
            elif need_q and fused is not None:
                _call(
                    "rspace_backward_charges", lib.mipme_sr_rows_fused,
                    st, dt, N, topo.row_ptr.data_ptr(), fused["ent_sh"].data_ptr(), topo.entries.data_ptr(),
                    _lib.ptr(mask), fused["pos"].data_ptr(), _lib.ptr(fused["cell"]), q.data_ptr(), g.data_ptr(), None,
                    1, full, C.byref(pot_desc), 1, fused["fmt"], fused["records"].data_ptr(), 0, grad_q.data_ptr(), None, None,
                    None, None,
                )
            elif need_q and topo is not None and getattr(ctx, "tab", None) is not None:
                if N > 1 and grad_out.stride(0) == 0 and grad_out.stride(1) in (0, 1):
                    # a uniform upstream gradient (``result.sum().backward()``): c times the table's row sums
                    grad_q.addcmul_(ctx.tab[1], grad_out[:1])
                else:
                    _call(
                        "rspace_backward_charges", lib.mipme_rspace_rows_tabulated,
                        st, dt, N, topo.row_ptr.data_ptr(), ctx.tab[0].data_ptr(), g.data_ptr(), 1, full, 1, grad_q.data_ptr(),
                    )
            elif need_q and topo is not None:
                _call(
                    "rspace_backward_charges", lib.mipme_rspace_rows,
                    st, dt, N, Cn, topo.row_ptr.data_ptr(), topo.entries.data_ptr(), dist.data_ptr(), g.data_ptr(),
                    _lib.ptr(mask), 1, full, C.byref(pot_desc), 1, grad_q.data_ptr(),
                )
            if geom is None:
                if need_pos:
                    grad_pos = torch.zeros((N, 3), dtype=dtype, device=device)
                if need_cell:
                    grad_cell = torch.zeros((3, 3), dtype=dtype, device=device)
            if lazy and ctx.needs_input_grad[3]:
                # the pair part leaves through the neighbor_distances slot: the products with the distance Jacobian ride in
                # the placeholder, the (P,) gradient is formed on demand
                def make_grad_dist(g=g, sr_scale=sr_scale):
                    out_d = torch.empty((P,), dtype=dtype, device=device)
                    pl = pairs if topo is None else topo.pairs32
                    with _lib.on_device(device):
                        _call(
                            "rspace_backward", lib.mipme_rspace_backward,
                            _lib.current_stream(device), dt, _lib.index_code(pl.dtype), P, N, Cn, pl.data_ptr(),
                            dist.data_ptr(), q.data_ptr(), _lib.ptr(mask), full, C.byref(pot_desc), g.data_ptr(),
                            _lib.ptr(sr_scale), out_d.data_ptr(), None,
                        )
                    return out_d

                grad_dist = LazyPairGradient(P, dtype, device, grad_src_pos, grad_src_cell, make_grad_dist)
                grad_src_pos = grad_src_cell = None
        return (grad_q, grad_cell, grad_pos, grad_dist, None, None, None, None, None, None, None, grad_src_pos,
                grad_src_cell, None, None, None, None)


def pme_potential(charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask, geom, G, pot_desc,
                  full_list, slab_axis, nan_flag=None, atomic_pairs=False):
    src = getattr(neighbor_distances, "_mipme_src", None)
    if getattr(neighbor_indices, "_mipme_stream", None) is not None:
        if (src is None or atomic_pairs or pair_mask is not None or not src.direct
                or not src.usable_for(neighbor_distances, neighbor_indices, charges.shape[1])):
            raise ValueError(
                "the handles of a NeighborStream serve single-channel calculators without a pair mask, together with the "
                "distances from `stream.distances(positions, cell)` of the current positions; use `stream.pairs()` for a list in "
                "the reference's format")
    if atomic_pairs:
        if src is not None and src.pending:
            src.materialize()
        return _PMEFunction.apply(charges, cell, positions, neighbor_distances, neighbor_indices, pair_mask, geom, G,
                                  pot_desc, full_list, slab_axis, None, None, None, False, nan_flag, True)
    if src is not None and src.usable_for(neighbor_distances, neighbor_indices, charges.shape[1]):
        # the fused kernels differentiate through the distances in the same pass (see FUSE_DISTANCES)
        if not src.direct:
            # default: the pair part's gradient flows through ``neighbor_distances`` (lazily, LazyPairGradient) and on to
            # positions / cell via that tensor's own node -- the reference's autograd graph
            return _PMEFunction.apply(charges, cell, positions, neighbor_distances, neighbor_indices, pair_mask, geom,
                                      G, pot_desc, full_list, slab_axis, src.positions, src.cell, src, True, nan_flag)
        # opt-in (``deferred=True``): straight to the tensors the distances were built from, bypassing ``d``
        out = _PMEFunction.apply(charges, cell, positions, neighbor_distances.detach(), neighbor_indices, pair_mask, geom,
                                 G, pot_desc, full_list, slab_axis, src.positions, src.cell, src, False, nan_flag)
        node = out.grad_fn
        if node is not None and getattr(node, "energy_direct", False):
            out._mipme_energy = (node, positions, charges, charges._version, cell, src.cell)
        return out
    if src is not None and src.pending:
        src.materialize()
    return _PMEFunction.apply(charges, cell, positions, neighbor_distances, neighbor_indices, pair_mask, geom, G,
                              pot_desc, full_list, slab_axis, None, None, None, False, nan_flag)


def _launch_pair_distances(pos, cl, pairs, sh, shifts_key, out):
    """out[p] = |r_j - r_i + S_p cell| (contiguous, detached tensors; ``sh`` already in the dtype of ``pos``)."""
    lib = _lib.load()
    device, dtype = pos.device, pos.dtype
    P = pairs.shape[0]
    topo = get_topology(pairs, pos.shape[0]) if PAIR_MODE == "rows" else None
    if isinstance(topo, PairTopology) and sh is not None:
        topo.adopt_shifts(sh, shifts_key)
    pl = pairs if topo is None else topo.pairs32
    packed = topo.pair_packed_shifts(sh, shifts_key) if (topo is not None and sh is not None) else None
    with _lib.on_device(device):
        if packed is not None:
            _call(
                "pair_distance_forward", lib.mipme_pair_distance_forward_packed,
                _lib.current_stream(device), _lib.dtype_code(dtype), P, pl.data_ptr(), packed.data_ptr(),
                pos.data_ptr(), cl.data_ptr(), out.data_ptr(),
            )
        else:
            _call(
                "pair_distance_forward", lib.mipme_pair_distance_forward,
                _lib.current_stream(device), _lib.dtype_code(dtype), _lib.index_code(pl.dtype), P,
                pl.data_ptr(), pos.data_ptr(), _lib.ptr(cl), _lib.ptr(sh), out.data_ptr(),
            )
    return topo


class _PairDistances(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positions, cell, neighbor_indices, shifts, deferred=False):
        device, dtype = positions.device, positions.dtype
        pos = positions.detach().contiguous()
        pairs = neighbor_indices.contiguous()
        cl = None if cell is None else cell.detach().contiguous()
        sh = None if shifts is None else shifts.to(dtype).contiguous()
        P = pairs.shape[0]
        out = torch.empty((P,), dtype=dtype, device=device)
        if deferred:  # values written later: by the fused pair kernel, or by DistanceSource.materialize()
            topo = get_topology(pairs, pos.shape[0]) if PAIR_MODE == "rows" else None
        else:
            topo = _launch_pair_distances(pos, cl, pairs, sh, shifts, out)
        ctx.save_for_backward(pos, cl, pairs, sh)
        ctx.topo = topo
        ctx.shifts_key = shifts
        #: the inputs as the caller passed them (with their autograd history), for a backward pass that is itself recorded
        ctx.inputs_for_second_order = (positions, cell) if SECOND_ORDER_DISTANCES else None
        return out

    @staticmethod
    def backward(ctx, grad_d):
        if torch.is_grad_enabled() and ctx.inputs_for_second_order is not None:
            # create_graph=True: the adjoint of the distances as differentiable tensor ops (exact second order; what the
            # reference's helper does, tests/helpers.py:278-304) -- the HIP adjoint kernel is first order
            positions, cell = ctx.inputs_for_second_order
            _, _, pairs, sh = ctx.saved_tensors
            if isinstance(grad_d, LazyPairGradient):
                grad_d = grad_d.materialize()
            from . import analytic

            topo = ctx.topo
            rows = (topo.row_ptr, topo.entries) if isinstance(topo, PairTopology) else None
            gp, gc = analytic.recorded_distance_backward(grad_d, positions, cell, pairs, sh, rows, ctx.needs_input_grad[0],
                                                         ctx.needs_input_grad[1])
            return gp, gc, None, None, None
        return _PairDistances._backward_first_order(ctx, grad_d)

    @staticmethod
    @first_order
    def _backward_first_order(ctx, grad_d):
        lib = _lib.load()
        pos, cl, pairs, sh = ctx.saved_tensors
        device, dtype = pos.device, pos.dtype
        N, P = pos.shape[0], pairs.shape[0]
        need_cell = cl is not None and ctx.needs_input_grad[1]
        if isinstance(grad_d, LazyPairGradient) and not grad_d.materialized:
            # sole consumer was a calculator whose fused kernels already applied this node's Jacobian (see LazyPairGradient)
            gp = grad_d._grad_pos if ctx.needs_input_grad[0] else None
            gc = grad_d._grad_cell if need_cell else None
            if (gp is not None or not ctx.needs_input_grad[0]) and (gc is not None or not need_cell):
                return gp, gc, None, None, None
            grad_d = grad_d.materialize()  # a product the calculator did not form: take the general route
        elif isinstance(grad_d, LazyPairGradient):
            grad_d = grad_d.materialize()
        if P == 0:  # empty pair list: nothing depends on the positions or the cell (and there is no gradient buffer to pass)
            return ((torch.zeros((N, 3), dtype=dtype, device=device) if ctx.needs_input_grad[0] else None),
                    (torch.zeros((3, 3), dtype=dtype, device=device) if need_cell else None), None, None, None)
        grad_pos = torch.empty((N, 3), dtype=dtype, device=device)
        grad_cell = partials = None
        topo = ctx.topo
        if need_cell:
            grad_cell = torch.empty((3, 3), dtype=dtype, device=device)
            n_part = lib.mipme_rows_partials_size(N) if topo is not None else lib.mipme_pair_partials_size(P)
            partials = torch.empty((n_part,), dtype=torch.float64, device=device)
        if topo is not None:
            packed = None if sh is None else topo.packed_shifts(sh, ctx.shifts_key)
            with _lib.on_device(device):
                _call(
                    "pair_distance_backward", lib.mipme_pair_distance_backward_rows,
                    _lib.current_stream(device), _lib.dtype_code(dtype), N, topo.row_ptr.data_ptr(),
                    topo.entries.data_ptr(), _lib.ptr(packed), pos.data_ptr(), _lib.ptr(cl),
                    None if packed is not None else _lib.ptr(sh), grad_d.contiguous().data_ptr(), _lib.ptr(partials),
                    grad_pos.data_ptr(), _lib.ptr(grad_cell),
                )
            return (grad_pos if ctx.needs_input_grad[0] else None), grad_cell, None, None, None
        with _lib.on_device(device):
            _call(
                "pair_distance_backward", lib.mipme_pair_distance_backward,
                    _lib.current_stream(device), _lib.dtype_code(dtype), _lib.index_code(pairs.dtype), P, N,
                    pairs.data_ptr(), pos.data_ptr(), _lib.ptr(cl), _lib.ptr(sh), grad_d.contiguous().data_ptr(),
                    _lib.ptr(partials), grad_pos.data_ptr(), _lib.ptr(grad_cell),
            )
        return (grad_pos if ctx.needs_input_grad[0] else None), grad_cell, None, None, None


def pair_distances(positions, neighbor_indices, cell=None, neighbor_shifts=None, deferred: bool | str = False):
    """``d[p] = |r_j - r_i + S_p @ cell|``, differentiable w.r.t. ``positions`` and ``cell``.

    Counterpart of the reference's caller-side helper ``compute_distances``
    (``tests/helpers.py:278-304``, ``examples/02-neighbor-lists-usage.py:141-164``).

    ``deferred=True`` is a promise that the returned tensor goes to a calculator of this package before anything reads
    it: its values are then written by the calculator's fused distance + pair kernel (the row that owns a pair's first
    atom stores ``d[p]``) instead of by a separate pass over the list; a calculator call that cannot do so (pair mask,
    non-integer shifts, list not ordered by its first index, ...) runs the stand-alone kernel first.  The values, the
    autograd graph and the result of the calculator are the same either way.

    ``deferred="virtual"`` is the stronger promise that NOTHING but calculators of this package ever reads the tensor: it
    connects the calculator to ``positions`` / ``cell`` in the autograd graph and carries the provenance, but a calculator
    whose fused kernel forms the distances in registers no longer stores them (19 MB per step at 4.76 M pairs, and the
    branch around the store); calculators that need the values in memory still write them first.  Its contents are
    otherwise undefined -- use it for the internal distance tensor of an energy + forces step, not for one you look at."""
    if torch.compiler.is_compiling():  # one dispatcher op inside torch.compile (library.py); nothing to defer there
        return torch.ops.mipme.pair_distances(positions, neighbor_indices, cell, neighbor_shifts)
    return _pair_distances_eager(positions, neighbor_indices, cell, neighbor_shifts, deferred)


@torch.compiler.disable
def _pair_distances_eager(positions, neighbor_indices, cell, neighbor_shifts, deferred):
    try:
        with betting():
            dist = _pair_distances_once(positions, neighbor_indices, cell, neighbor_shifts, deferred)
        verify_bets()  # (SPECULATE_LISTS: the structures of a previous list tensor may have been reused on a bet)
        return dist
    except SpeculationLost:
        return _pair_distances_once(positions, neighbor_indices, cell, neighbor_shifts, deferred)


def _pair_distances_once(positions, neighbor_indices, cell, neighbor_shifts, deferred):
    if cell is not None and neighbor_shifts is None:
        raise ValueError("Provided `cell` but no `neighbor_shifts`.")
    if cell is None and neighbor_shifts is not None:
        raise ValueError("Provided `neighbor_shifts` but no `cell`.")
    _lib.require_device(positions, "positions")
    if deferred not in (True, False, "virtual"):
        raise ValueError(f"`deferred` must be True, False or 'virtual', got {deferred!r}")
    if neighbor_shifts is None or neighbor_shifts.dtype == positions.dtype:
        shifts_c = neighbor_shifts
    else:
        shifts_c = neighbor_shifts.to(positions.dtype)
    if (deferred is False and FRONT and PROFILE is None and cell is not None and PAIR_MODE == "rows" and FUSE_DISTANCES and positions.requires_grad
            and torch.is_grad_enabled() and type(positions) is torch.Tensor and type(cell) is torch.Tensor
            and type(neighbor_indices) is torch.Tensor and neighbor_indices.is_contiguous() and neighbor_indices.dim() == 2
            and getattr(neighbor_indices, "_mipme_stream", None) is None and shifts_c.is_contiguous()
            and not inside_vmap(positions, cell, neighbor_indices, shifts_c)):
        # compiled front end (csrc/front.cpp): the same kernel, a C++ autograd node, ~8 us of host time instead of ~50
        mod = _front.module()
        if mod is not None:
            topo_f = get_topology(neighbor_indices, positions.shape[0])
            topo_f.adopt_shifts(shifts_c, neighbor_shifts)
            ft = topo_f.front(neighbor_indices, shifts_c, neighbor_shifts)
            if ft is not None:
                dist = mod.pair_distances(ft, positions, cell, neighbor_indices)
                if dist is not None:
                    dist._mipme_src = DistanceSource(positions, cell, neighbor_indices, shifts_c, neighbor_shifts, dist, False)
                    return dist
    dist = _PairDistances.apply(positions, cell, neighbor_indices, neighbor_shifts, bool(deferred))
    dist._mipme_src = DistanceSource(positions, cell, neighbor_indices, shifts_c, neighbor_shifts, dist, bool(deferred),
                                     virtual=deferred == "virtual")
    return dist


def stream_distances(stream, positions, cell):
    """The ``neighbor_distances`` handle of a :class:`~torchpme_amd.neighbors.NeighborStream`: an unwritten ("virtual")
    tensor whose provenance tells the calculator to form the distances inside its fused pair kernel from ``positions`` /
    ``cell`` and to send the pair part of the gradient straight to those tensors."""
    _lib.require_device(positions, "positions")
    dist = torch.empty((stream.indices.shape[0],), dtype=positions.dtype, device=positions.device)
    dist._mipme_src = DistanceSource(positions, cell, stream.indices, None, None, dist, pending=True, virtual=True)
    return dist


#: keep the distance node's inputs alive for a recorded (create_graph=True) backward pass; costs nothing unless used
SECOND_ORDER_DISTANCES = True


class _FiniteDifferenceSecondOrder(torch.autograd.Function):
    """``V = f(*z)`` for a black-box first-order evaluation ``f`` (a calculator's forward on the HIP kernels) as a node whose
    BACKWARD is differentiable too: the vector-Jacobian product ``G(g, z) = grad_z <g, f(z)>`` is formed analytically by the
    first-order path; when that pass is itself being recorded (``create_graph=True``), ``G`` is issued as a second node whose
    backward -- the Hessian-vector product and ``d<c, G>/dg`` -- comes from central differences of the analytic ``G`` and of
    ``f`` along the incoming cotangent ``c``:  H c = [G(g, z + e c) - G(g, z - e c)] / 2e,  d/dg = [f(z + e c) - f(z - e c)] / 2e.
    Two more evaluations per double-backward pass; error O(e^2) + rounding / e (float64: ~1e-7 relative with e = 1e-4 |z| / |c|)."""

    @staticmethod
    def forward(ctx, f, n_diff, *z):
        ctx.f, ctx.n_diff = f, n_diff
        with torch.no_grad():
            out = f(*[t.detach() if isinstance(t, torch.Tensor) else t for t in z])
        ctx.save_for_backward(*[t for t in z[:n_diff]])
        ctx.rest = z[n_diff:]
        return out

    @staticmethod
    def backward(ctx, g):
        z = ctx.saved_tensors
        if torch.is_grad_enabled():
            needs = tuple(ctx.needs_input_grad[2:2 + ctx.n_diff])
            grads = _FiniteDifferenceGradient.apply(ctx.f, ctx.n_diff, ctx.rest, needs, g, *z)
            grads = [x if n else None for x, n in zip(grads, needs)]
        else:
            grads = _fd_vjp(ctx.f, ctx.rest, tuple(ctx.needs_input_grad[2:2 + ctx.n_diff]), g, z)
        return (None, None, *grads, *([None] * len(ctx.rest)))


def _fd_vjp(f, rest, needs, g, z):
    """grad_z <g, f(z, *rest)> through the first-order autograd nodes of the HIP path (None where not needed)."""
    with torch.enable_grad():
        leaves = [t.detach().requires_grad_(n) for t, n in zip(z, needs)]
        V = f(*leaves, *rest)
        wanted = [t for t, n in zip(leaves, needs) if n]
        got = list(torch.autograd.grad(V, wanted, g.detach(), allow_unused=True)) if wanted else []
    out = []
    for t, n in zip(leaves, needs):
        gr = got.pop(0) if n else None
        out.append(torch.zeros_like(t) if (n and gr is None) else (gr.detach() if gr is not None else None))
    return out


class _FiniteDifferenceGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, n_diff, rest, needs, g, *z):
        ctx.f, ctx.rest, ctx.needs = f, rest, needs
        ctx.save_for_backward(g, *z)
        grads = _fd_vjp(f, rest, needs, g, z)
        out = tuple(torch.zeros((), device=g.device, dtype=g.dtype) if x is None else x for x in grads)
        ctx.mark_non_differentiable(*[o for o, x in zip(out, grads) if x is None])
        return out

    @staticmethod
    @first_order
    def backward(ctx, *cot):
        g, *z = ctx.saved_tensors
        f, rest, needs = ctx.f, ctx.rest, ctx.needs
        c = [ci if (n and ci is not None) else None for ci, n in zip(cot, needs)]
        cmax = max((float(ci.abs().max()) for ci in c if ci is not None and ci.numel()), default=0.0)
        if cmax == 0.0:
            return (None, None, None, None, torch.zeros_like(g), *[torch.zeros_like(t) if n else None for t, n in zip(z, needs)])
        # largest change of any input component along the direction: 1e-4 (float64) / 1e-2 (float32) in the inputs' own units
        # (Angstrom, charges: scales of order one)
        eps = (1e-4 if g.dtype == torch.float64 else 1e-2) / cmax
        zp = [t.detach() + eps * ci if ci is not None else t.detach() for t, ci in zip(z, c)]
        zm = [t.detach() - eps * ci if ci is not None else t.detach() for t, ci in zip(z, c)]
        with torch.no_grad():
            Vp, Vm = f(*zp, *rest), f(*zm, *rest)
        Gp, Gm = _fd_vjp(f, rest, needs, g, zp), _fd_vjp(f, rest, needs, g, zm)
        grad_g = (Vp - Vm) / (2 * eps)
        Hc = [((a - b) / (2 * eps)) if n else None for a, b, n in zip(Gp, Gm, needs)]
        return (None, None, None, None, grad_g, *Hc)


class _FusedFirstAnalyticHigher(torch.autograd.Function):
    """``V = f(*z)`` through the fused first-order kernels, with higher orders from a differentiable twin ``f_exact`` (the same
    call through the primitives of :mod:`analytic`) that is only evaluated when a backward pass is itself being recorded.

    forward: ``f`` on detached leaves with grad mode on -- the inner graph (the calculator's own first-order nodes) is kept;
    backward, not recorded: the inner graph's vector-Jacobian product, i.e. exactly the kernels of a plain call;
    backward, recorded (``create_graph=True``): ``grad(f_exact(*z), z, g, create_graph=True)`` on the ORIGINAL inputs -- values
    from the primitives, differentiable to any order."""

    @staticmethod
    def forward(ctx, f, f_exact, n_diff, *z):
        leaves = [t.detach().requires_grad_(t.requires_grad) for t in z[:n_diff]]
        with torch.enable_grad():
            inner = f(*leaves, *z[n_diff:])
        ctx.f_exact, ctx.n_diff, ctx.rest = f_exact, n_diff, z[n_diff:]
        ctx.leaves, ctx.inner = leaves, inner
        ctx.save_for_backward(*z[:n_diff])
        return inner.detach()

    @staticmethod
    def backward(ctx, g):
        needs = tuple(ctx.needs_input_grad[3:3 + ctx.n_diff])
        tail = [None] * len(ctx.rest)
        if not torch.is_grad_enabled():
            wanted = [t for t, n in zip(ctx.leaves, needs) if n]
            if ctx.inner.grad_fn is None or not wanted:
                return (None, None, None, *[None] * ctx.n_diff, *tail)
            got = list(torch.autograd.grad(ctx.inner, wanted, g, retain_graph=True, allow_unused=True))
            grads = [got.pop(0) if n else None for n in needs]
            return (None, None, None, *grads, *tail)
        with torch.enable_grad():
            # aliases of the inputs: the inputs may depend on each other in the caller's graph (distances on positions and cell,
            # positions on the cell), and the gradient w.r.t. an input tensor itself would then include the paths through the
            # others -- which the caller's graph adds a second time.  The gradient w.r.t. an alias is the PARTIAL derivative, and
            # stays connected to the original for every later differentiation.
            z = [t.view_as(t) for t in ctx.saved_tensors]
            V = ctx.f_exact(*z, *ctx.rest)
            wanted = [t for t, n in zip(z, needs) if n]
            got = list(torch.autograd.grad(V, wanted, g, create_graph=True, allow_unused=True)) if wanted else []
        grads = [got.pop(0) if n else None for n in needs]
        return (None, None, None, *grads, *tail)


def fused_first_analytic_higher(f, f_exact, diff_inputs, other_inputs):
    """See :class:`_FusedFirstAnalyticHigher` (``calculator.double_backward = "auto"``)."""
    return _FusedFirstAnalyticHigher.apply(f, f_exact, len(diff_inputs), *diff_inputs, *other_inputs)


def second_order_by_finite_differences(f, diff_inputs, other_inputs):
    """``f(*diff_inputs, *other_inputs)`` with a differentiable backward pass (see :class:`_FiniteDifferenceSecondOrder`)."""
    return _FiniteDifferenceSecondOrder.apply(f, len(diff_inputs), *diff_inputs, *other_inputs)


_DOT_SCRATCH = {}


def _dot_scratch(device, owner):
    """Persistent, zero-initialised scratch of ``mipme_dot_forward`` per (device, owner): the owner is the storage address
    of the charges tensor, so reductions of different frames (which may run concurrently on different streams) never share
    a ticket counter.  Not keyed on the stream: a CUDA-graph capture stream differs from the warm-up stream, and a buffer
    created during capture would bake its zero fill into every replay."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), owner)
    buf = _DOT_SCRATCH.get(key)
    if buf is None:
        while len(_DOT_SCRATCH) >= 256:
            _DOT_SCRATCH.pop(next(iter(_DOT_SCRATCH)))
        buf = _DOT_SCRATCH[key] = torch.zeros((65,), dtype=torch.float64, device=device)
    return buf


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        a_c, b_c = a.detach().contiguous(), b.detach().contiguous()
        out = torch.empty((), dtype=a.dtype, device=a.device)
        scratch = _dot_scratch(a.device, b.data_ptr())
        with _lib.on_device(a.device):
            _call("energy_sum", lib.mipme_dot_forward, _lib.current_stream(a.device), _lib.dtype_code(a.dtype),
                  a_c.numel(), a_c.data_ptr(), b_c.data_ptr(), scratch.data_ptr(), out.data_ptr())
        ctx.save_for_backward(a_c, b_c)
        return out

    @staticmethod
    @first_order
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        g = g.contiguous()
        with _lib.on_device(a.device):
            _call("energy_sum_backward", lib.mipme_dot_backward, _lib.current_stream(a.device), _lib.dtype_code(a.dtype),
                  a.numel(), g.data_ptr(), a.data_ptr(), b.data_ptr(), _lib.ptr(ga), _lib.ptr(gb))
        if ga is not None:
            # ga == g * b exactly: let a calculator backward that receives THIS tensor recognise it (see ENERGY_FAST_PATH)
            ga._mipme_scaled = (b.data_ptr(), tuple(b.shape), b._version, g)
        return ga, gb


@torch.compiler.disable
def weighted_sum(potentials: torch.Tensor, charges: torch.Tensor) -> torch.Tensor:
    """``(charges * potentials).sum()`` -- the energy reduction every caller of the reference performs
    (``README.rst:112-114``) -- as one kernel forward and one backward instead of five ATen launches."""
    if potentials.shape != charges.shape or potentials.dtype != charges.dtype:
        raise ValueError("`potentials` and `charges` must have the same shape and dtype")
    _lib.require_device(potentials, "potentials")
    hook = getattr(potentials, "_mipme_energy", None)
    if (hook is not None and ENERGY_FAST_PATH and torch.is_grad_enabled() and potentials.grad_fn is hook[0]
            and charges is hook[2] and charges._version == hook[3]):
        # (a node that is `energy_direct` with charges / cell that want a gradient has them in its gather tail)
        return _EnergyDirectSum.apply(potentials.detach(), charges, hook[1], hook[0], hook[4],
                                      None if hook[5] is hook[4] else hook[5])
    return _WeightedSum.apply(potentials, charges)


class _EnergyDirectSum(torch.autograd.Function):
    """``E = sum_a q_a V_a`` of potentials that come straight from a calculator, differentiated w.r.t. the positions in one
    step.  The gradient that would reach the potentials is ``gE * q`` (the charges are constants here), and for exactly
    that upstream gradient the calculator's backward is ``dE/dr_a = gE q_a (f F_a + field_a)`` with per-atom sums its
    FORWARD kernels already formed (``F``: speculative pair-force sums of the fused row kernel; ``field``: mesh field from
    the gather; f = 1/2 for a full list).  This node evaluates that expression directly (``forces_finalize``): neither
    the adjoint of the reduction (the tensor ``gE * q``) nor the potentials' own autograd node runs.  Any other consumer
    of the potentials still differentiates through their node as usual -- the contributions add."""

    @staticmethod
    def forward(ctx, V, q, positions, node, cell=None, src_cell=None):
        lib = _lib.load()
        q_c = q.detach().contiguous()
        ctx.tail = tail = getattr(node, "tail", None)
        if tail is not None:
            out = tail["energy"].detach()  # formed by the gather's tail: no reduction launch
        else:
            V_c = V.contiguous()
            out = torch.empty((), dtype=V.dtype, device=V.device)
            scratch = _dot_scratch(V.device, q.data_ptr())
            with _lib.on_device(V.device):
                _call("energy_sum", lib.mipme_dot_forward, _lib.current_stream(V.device), _lib.dtype_code(V.dtype),
                      V_c.numel(), V_c.data_ptr(), q_c.data_ptr(), scratch.data_ptr(), out.data_ptr())
        ctx.q, ctx.force, ctx.field, ctx.full = q_c, node.fused["force"], node.field, int(node.full_list)
        ctx.same_cell = src_cell is None  # the distances were formed with this very cell tensor (weighted_sum passes None then)
        return out

    @staticmethod
    @first_order
    def backward(ctx, g):
        lib = _lib.load()
        q = ctx.q
        tail = ctx.tail
        need_q, need_pos, need_cell, need_src_cell = (ctx.needs_input_grad[k] for k in (1, 2, 4, 5))
        promised = (tail is not None and tail["seed"] is not None and g.data_ptr() == tail["seed"].data_ptr()
                    and g.numel() == 1)
        grad_q = grad_cell = grad_src_cell = None
        if tail is not None and (need_q or need_cell or need_src_cell):
            # the rest of the contract, from the same gather tail (seed * dE/dq = 2 seed V; seed * dE/dcell: mesh part, pair part,
            # their sum); the tail was written with the promised seed, or with 1
            base = tail["aux_seed"] if tail["aux_seed"] is not None else tail["seed"]
            scale = None if (promised and tail["aux_seed"] is None) else (g if base is None else g / base)
            if need_q:
                grad_q = tail["grad_q"].detach() if scale is None else tail["grad_q"] * scale
            gc = tail["grad_cell"]
            if gc is not None:
                gc = gc.detach() if scale is None else gc * scale
                if need_cell and need_src_cell:
                    grad_cell, grad_src_cell = gc[0:9].view(3, 3), gc[9:18].view(3, 3)
                elif need_cell:
                    # the sum of mesh and pair part only if the distances' cell IS this tensor; distances from another
                    # (e.g. detached) cell tensor leave `cell` the mesh part alone, as the reference's graph does
                    grad_cell = (gc[18:27] if ctx.same_cell else gc[0:9]).view(3, 3)
                elif need_src_cell:
                    grad_src_cell = gc[9:18].view(3, 3)
        if not need_pos:
            return None, grad_q, None, None, grad_cell, grad_src_cell
        if promised:
            # the promised seed: the gather's tail has already written seed * dE/dpositions (a fresh alias, so that the
            # accumulation into positions.grad takes the buffer instead of copying it)
            return None, grad_q, tail["grad"].detach(), None, grad_cell, grad_src_cell
        grad_pos = torch.empty((q.shape[0], 3), dtype=q.dtype, device=q.device)
        g = g.contiguous()
        with _lib.on_device(q.device):
            _call("forces_finalize", lib.mipme_sr_rows_finalize, _lib.current_stream(q.device), _lib.dtype_code(q.dtype),
                  q.shape[0], _lib.ptr(ctx.force), _lib.ptr(ctx.field), q.data_ptr(), g.data_ptr(), ctx.full, None,
                  grad_pos.data_ptr(), None)
        return None, grad_q, grad_pos, None, grad_cell, grad_src_cell


class _EwaldKSpace(torch.autograd.Function):
    """``core[i,c] = sum_k G(k) [cos(k r_i) S_c(k,c) + sin(k r_i) S_s(k,c)]`` with S the structure factors of the charges:
    the reciprocal-space sum of ``EwaldCalculator`` (reference ``calculators/ewald.py:97-113``) without the 1/V factor.
    Differentiable w.r.t. charges, positions and the k-vectors (through which the cell gradient flows).

    Inputs are one structure -- charges (N,C), positions (N,3), kvectors (K,3) -- or a padded batch with a leading batch
    dimension on all three (B,N,C), (B,N,3), (B,K,3): ONE launch per kernel either way (``blockIdx.y`` = structure)."""

    @staticmethod
    def forward(ctx, charges, positions, kvectors, pot_desc):
        lib = _lib.load()
        device, dtype = positions.device, positions.dtype
        dt = _lib.dtype_code(dtype)
        q, pos, kv = charges.detach().contiguous(), positions.detach().contiguous(), kvectors.detach().contiguous()
        batched = pos.dim() == 3
        B = pos.shape[0] if batched else 1
        N, Cn = q.shape[-2:]
        K = kv.shape[-2]
        lead = (B,) if batched else ()
        G = torch.empty(lead + (K,), dtype=dtype, device=device)
        dG = torch.empty(lead + (K,), dtype=dtype, device=device)
        Sc = torch.empty(lead + (K, Cn), dtype=dtype, device=device)
        Ss = torch.empty(lead + (K, Cn), dtype=dtype, device=device)
        out = torch.empty(lead + (N, Cn), dtype=dtype, device=device)
        with _lib.on_device(device):
            st = _lib.current_stream(device)
            _call("ewald_filter", lib.mipme_ewald_filter, st, dt, C.byref(pot_desc), B * K, kv.data_ptr(), G.data_ptr(),
                  dG.data_ptr())
            _call("ewald_structure", lib.mipme_ewald_structure, st, dt, N, Cn, K, pos.data_ptr(), q.data_ptr(),
                  kv.data_ptr(), Sc.data_ptr(), Ss.data_ptr(), B)
            _call("ewald_potential", lib.mipme_ewald_potential, st, dt, N, Cn, K, pos.data_ptr(), kv.data_ptr(),
                  G.data_ptr(), Sc.data_ptr(), Ss.data_ptr(), out.data_ptr(), B)
        ctx.save_for_backward(q, pos, kv, G, dG, Sc, Ss)
        return out

    @staticmethod
    @first_order
    def backward(ctx, grad_out):
        lib = _lib.load()
        q, pos, kv, G, dG, Sc, Ss = ctx.saved_tensors
        need_q, need_pos, need_k = ctx.needs_input_grad[:3]
        device, dtype = pos.device, pos.dtype
        dt = _lib.dtype_code(dtype)
        B = pos.shape[0] if pos.dim() == 3 else 1
        N, Cn = q.shape[-2:]
        K = kv.shape[-2]
        g = grad_out.contiguous()
        Tc = torch.empty_like(Sc)
        Ts = torch.empty_like(Ss)
        grad_q = torch.empty_like(q) if need_q else None
        grad_pos = torch.empty_like(pos) if need_pos else None
        grad_k = torch.empty_like(kv) if need_k else None
        with _lib.on_device(device):
            st = _lib.current_stream(device)
            _call("ewald_structure", lib.mipme_ewald_structure, st, dt, N, Cn, K, pos.data_ptr(), g.data_ptr(),
                  kv.data_ptr(), Tc.data_ptr(), Ts.data_ptr(), B)
            if need_q:  # the sum is symmetric in (q, g): same kernel with the structure factors of g
                _call("ewald_potential", lib.mipme_ewald_potential, st, dt, N, Cn, K, pos.data_ptr(), kv.data_ptr(),
                      G.data_ptr(), Tc.data_ptr(), Ts.data_ptr(), grad_q.data_ptr(), B)
            if need_pos or need_k:
                _call("ewald_backward", lib.mipme_ewald_backward, st, dt, N, Cn, K, pos.data_ptr(), q.data_ptr(),
                      g.data_ptr(), kv.data_ptr(), G.data_ptr(), dG.data_ptr(), Sc.data_ptr(), Ss.data_ptr(),
                      Tc.data_ptr(), Ts.data_ptr(), _lib.ptr(grad_pos), _lib.ptr(grad_k), B)
        return grad_q, grad_pos, grad_k, None


def ewald_kspace(charges, positions, kvectors, pot_desc):
    return _EwaldKSpace.apply(charges, positions, kvectors, pot_desc)


# ---- torch.vmap bridge (padded batches, reference tests/calculators/test_padding.py) ------------------------------------
def inside_vmap(*tensors) -> bool:
    """True if any argument is a functorch BatchedTensor, i.e. the caller is being traced by ``torch.vmap``."""
    check = getattr(torch._C._functorch, "is_batchedtensor", None)
    return check is not None and any(isinstance(t, torch.Tensor) and check(t) for t in tensors)


class _SampleLoop(torch.autograd.Function):
    """``torch.vmap(calculator.forward)(padded batch)``: the HIP kernels take one structure at a time, so the vmap rule runs
    the samples one after the other on their slices of the padded batch and stacks the results.  Each sample is an ordinary
    autograd graph, so gradients flow back through the stack / select ops."""

    @staticmethod
    def forward(fn, *args):
        return fn(*args)

    @staticmethod
    def setup_context(ctx, inputs, output):
        pass

    @staticmethod
    def backward(ctx, *grads):  # only reached when the bridge is applied outside vmap, which the calculators never do
        raise NotImplementedError("_SampleLoop is a torch.vmap bridge; call the calculator directly")

    @staticmethod
    def vmap(info, in_dims, fn, *args):
        # a calculator with kernels that take the whole padded batch in one launch (EwaldCalculator) handles it itself
        batched_impl = getattr(getattr(fn, "__self__", None), "_forward_batched", None)
        if batched_impl is not None:
            out = batched_impl(info.batch_size, in_dims[1:], *args)
            if out is not None:
                return out, 0
        outs = []
        for b in range(info.batch_size):
            sample = [a if d is None else a.select(d, b) for a, d in zip(args, in_dims[1:])]
            outs.append(fn(*sample))
        return torch.stack(outs), 0


def vmap_bridge(fn, *args):
    return _SampleLoop.apply(fn, *args)

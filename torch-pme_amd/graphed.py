"""HIP-graph replay of one energy + forces evaluation.

At 32k atoms a step is a dozen short kernels behind ~0.3 ms of Python / autograd / launch overhead when run eagerly.  For a fixed topology (same neighbour list, cell, charges; positions change) the whole
``pair_distances -> calculator.forward -> (q*V).sum().backward()`` chain is captured once into a HIP graph
(``torch.cuda.CUDAGraph``: PyTorch is the capture front end, every captured node is a libmipme kernel
or a tiny reduction) and replayed per step.  This is the MD-loop form of the hot path; the calculators themselves
stay eager and reference-compatible.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops


class _LiveStep:
    """Buffers and argument struct of ``mipme_md_rebin`` / ``mipme_md_step`` (include/mipme.h): the energy + forces step of an
    MD-like loop on device-resident neighbour structures.  The atoms live in ONE (N, 4) array of records (x, y, z, charge) that
    every kernel of the step reads; the atom -> mesh-brick bookkeeping (bins, per-brick atom lists) is rebuilt only when the
    neighbour list is (``rebin``), while every weight is evaluated from the current positions in every step."""

    def __init__(self, calculator, charges, cell, positions, charge_gradient=False, cell_gradient=False):
        lib = self.lib = _lib.load()
        device, dtype = positions.device, positions.dtype
        N = positions.shape[0]
        self.device, self.dtype, self.n_atoms = device, dtype, N
        geom, G = calculator._kspace_setup(cell, dtype, device, speculate=False)
        self.geom, self.G = geom, G
        self.md = geom.desc(1)
        self.pot = calculator.potential._descriptor()
        self.dt = _lib.dtype_code(dtype)
        if charges.shape[1] != 1 or not lib.mipme_md_supported(C.byref(self.md), C.byref(self.pot), N, self.dt):
            raise NotImplementedError("outside the range of the live-bin step")
        self.plan = _lib.get_plan(device, dtype, geom.ns, 1, geom.plan_store)
        if not self.plan.xfused:
            raise NotImplementedError("the live-bin step needs a power-of-two mesh along x")
        self.rec = torch.empty((N, 4), dtype=dtype, device=device)
        self.rec[:, :3] = positions.detach()
        self.rec[:, 3] = charges.detach()[:, 0]
        cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
        self.cell = cell.detach().to(dtype).contiguous()
        self.rho = torch.empty(geom.ns, dtype=dtype, device=device)
        self.phi = torch.empty(geom.ns, dtype=dtype, device=device)
        self.hat = torch.empty((geom.n_half,), dtype=cdtype, device=device)
        self.dc = torch.empty((1,), dtype=dtype, device=device)
        self.bins = torch.empty((lib.mipme_atom_bins_bytes(C.byref(self.md), N, self.dt),), dtype=torch.uint8, device=device)
        self.lists = torch.zeros((lib.mipme_md_lists_ints(C.byref(self.md), N),), dtype=torch.int32, device=device)
        self.potentials = torch.empty((N,), dtype=dtype, device=device)
        self.pair_force = torch.empty((N, 3), dtype=dtype, device=device)
        self.energy = torch.empty((), dtype=dtype, device=device)
        self.grad = torch.empty((N, 3), dtype=dtype, device=device)
        self.minus_one = torch.tensor(-1.0, dtype=dtype, device=device)
        self.one = torch.tensor(1.0, dtype=dtype, device=device)
        # the rest of the autograd contract from the same five launches (+ one single-workgroup launch for the cell)
        self.grad_q = torch.empty((N, 1), dtype=dtype, device=device) if charge_gradient else None
        self.grad_cell = self.G_deriv = self.cell_work = None
        if cell_gradient:
            self.grad_cell = torch.empty((27,), dtype=dtype, device=device)
            self.G_deriv = ops.filter_derivative(geom, self.pot, dtype, device)
            self.cell_work = torch.empty((lib.mipme_cell_tail_work(self.plan.handle, C.byref(self.md), N),),
                                         dtype=torch.float64, device=device)
        self.flags = torch.zeros((1,), dtype=torch.int32).pin_memory()
        self.flags_np = self.flags.numpy()
        self.nan_flag = calculator._nan_flag_ptr()
        calculator._nan_shape = [1, *geom.ns]
        self.rows = None  # (row_ptr, words, shift_format)
        self.log = None  # an EnergyLog the step's gather appends the energy to (set by GraphedEnergyForces before its capture)

    def _args(self):
        row_ptr, words, fmt = self.rows if self.rows is not None else (self.lists, self.lists, 2)  # (rebin reads no rows)
        return _lib.MdArgs(
            plan=self.plan.handle, stream=_lib.current_stream(self.device), dtype=self.dt, shift_format=fmt,
            mesh=C.pointer(self.md), pot=C.pointer(self.pot), n_atoms=self.n_atoms, records=self.rec.data_ptr(),
            cell=self.cell.data_ptr(), G=self.G.data_ptr(), rho_mesh=self.rho.data_ptr(), hat_work=self.hat.data_ptr(),
            phi_mesh=self.phi.data_ptr(), dc=self.dc.data_ptr(), atom_bins=self.bins.data_ptr(),
            live_lists=self.lists.data_ptr(), row_ptr=row_ptr.data_ptr(), words=words.data_ptr(),
            potentials=self.potentials.data_ptr(), pair_force=self.pair_force.data_ptr(), energy=self.energy.data_ptr(),
            grad_positions=self.grad.data_ptr(), grad_seed=self.minus_one.data_ptr(), nan_flag=self.nan_flag,
            host_flags=self.flags.data_ptr(), grad_charges=_lib.ptr(self.grad_q), grad_cell=_lib.ptr(self.grad_cell),
            G_deriv=_lib.ptr(self.G_deriv), cell_work=_lib.ptr(self.cell_work), aux_seed=self.one.data_ptr(),
            energy_log=None if self.log is None else self.log.values.data_ptr(),
            energy_log_cursor=None if self.log is None else self.log.cursor.data_ptr(),
            energy_log_capacity=0 if self.log is None else self.log.capacity)

    def rebin(self):
        with _lib.on_device(self.device):
            a = self._args()
            _lib.check(self.lib.mipme_md_rebin(C.byref(a)))

    def step(self):
        with _lib.on_device(self.device):
            a = self._args()
            _lib.check(self.lib.mipme_md_step(C.byref(a)))

    @property
    def max_displacement(self) -> float:
        """A displacement (same units as the cell) every atom may make since the last rebin without leaving the margin of ONE mesh
        point: ``min_d 1 / (n_d |column d of inv(cell)|)`` (u_d = n_d r . inv(cell)[:, d] changes by at most n_d |dr| |column d|)."""
        inv = np.asarray(self.geom.inv_cell)
        return float(min(1.0 / (self.geom.ns[d] * np.linalg.norm(inv[:, d])) for d in range(3)))

    def moved(self) -> bool:
        """True if a step since the last look flagged an atom beyond the margin (clears the bit)."""
        f = int(self.flags_np[0])
        if f & 2:
            self.flags_np[0] = f & ~2
            return True
        return False

    def check(self):
        f = int(self.flags_np[0])
        if f:
            self.flags_np[0] = 0
            if f & 2:
                raise RuntimeError("GraphedEnergyForces: an atom has moved more than one mesh point since the last refresh(); "
                                   "the results of that step are invalid (its energy was returned as NaN) -- refresh the "
                                   "neighbour structures sooner, or call the object with check=True")
            raise RuntimeError("GraphedEnergyForces: a brick's atom list overflowed at the last refresh (very non-uniform "
                               "system); construct with live_bins=False")


class EnergyLog:
    """Device-resident log of the frame energies of successive evaluations (the frames a rank owns).

    A rank of the frame farm evaluates batch after batch of independent frames; with a log the energies of every batch stay
    on the device -- ``values[k]`` (``n_frames`` float64) = the energies of the k-th evaluation since :meth:`reset` -- and are
    exchanged ONCE (``farm.gather_energy_log``: one all-gather of the whole log) instead of once per evaluation.  The graphed
    steps append inside their gather launch -- the thread that writes a frame's energy also writes its log slot
    (``mipme_kspace_forward_args_t.energy_log``, ``mipme_md_args_t.energy_log``, ``mipme_frames_table_energy_log``): no extra
    launch, a replay costs neither the host nor the GPU anything (a separate push node measured +2.0 us on a 58 us step).
    Evaluations that do not end in the gather tail use :meth:`push` (``mipme_energy_log_push``, one small launch).  Slots wrap
    around: the log holds the last ``capacity`` evaluations; ``cursor`` (int32 per frame on the device, all equal) counts them."""

    def __init__(self, capacity: int, n_frames: int, device):
        if capacity < 1 or n_frames < 1:
            raise ValueError("an energy log needs capacity >= 1 and n_frames >= 1")
        self.capacity, self.n_frames = int(capacity), int(n_frames)
        self.values = torch.zeros((self.capacity, self.n_frames), dtype=torch.float64, device=device)
        self.cursor = torch.zeros((self.n_frames,), dtype=torch.int32, device=device)

    def push(self, energies: torch.Tensor) -> None:
        """Append ``energies`` (``n_frames`` reals on the log's device) -- one launch on the current stream, capturable."""
        if energies.numel() != self.n_frames or not energies.is_contiguous():
            raise ValueError(f"the log holds {self.n_frames} energies per evaluation, got a tensor of {tuple(energies.shape)}")
        _lib.require_device(energies, "energies")
        with _lib.on_device(energies.device):
            _lib.check(_lib.load().mipme_energy_log_push(
                _lib.current_stream(energies.device), _lib.dtype_code(energies.dtype), self.n_frames, energies.data_ptr(),
                self.values.data_ptr(), self.cursor.data_ptr(), self.capacity))

    def reset(self) -> None:
        """Start a new log (asynchronous, on the current stream)."""
        self.cursor.zero_()

    def count(self) -> int:
        """Evaluations pushed since :meth:`reset` (synchronises)."""
        return int(self.cursor[0].item())


def _as_energy_log(energy_log, n_frames, device):
    if energy_log is None or isinstance(energy_log, EnergyLog):
        if energy_log is not None and energy_log.n_frames != n_frames:
            raise ValueError(f"the energy log holds {energy_log.n_frames} energies per evaluation, the step produces {n_frames}")
        return energy_log
    return EnergyLog(int(energy_log), n_frames, device)


class GraphedEnergyForces:
    """``E, F = step(positions)`` with ``E = sum_i q_i V_i`` and ``F = -dE/dpositions``, replayed from a HIP graph.

    :param calculator: a :class:`PMECalculator` / :class:`P3MCalculator`
    :param charges, cell, positions, neighbor_indices, neighbor_shifts: tensors on the GPU; ``positions`` only
        provides the shape/dtype and the values for the warm-up.
    :param cell_gradient: also return ``dE/dcell`` (3,3) from every call (the virial is ``-cell.T @ dE/dcell``): the co-scheduled
        pair sum, the x stage of the convolution and the gather leave partial sums behind and ONE more single-workgroup launch
        assembles them (``mipme_kspace_forward_args_t.out_grad_cell``); cases those kernels do not cover (stored
        distances, non-integer shifts) go through the calculator's general autograd nodes instead -- same numbers
    :param charge_gradient: also return ``dE/dcharges`` (N,1) from every call (``= 2 V``, written by the gather launch).  With
        both flags a call returns ``E, F, dE/dq, dE/dcell`` -- the whole first-order autograd contract of the reference
        (``tests/calculators/test_workflow.py:164-192``) from one graph replay
    :param store_distances: keep the pair distances of the last evaluation in ``self.distances`` (P,) -- written by the pair
        kernel as a by-product; off by default: the kernel then forms them in registers only (19 MB less per step at 4.76 M
        pairs, and the packed fp32 body of the pair sum applies)
    :param neighbors: instead of ``neighbor_indices`` / ``neighbor_shifts``: a cutoff (float, skin included) or a
        :class:`~torchpme_amd.neighbors.NeighborStream`.  The object then owns a device neighbour list in the pair kernels' own
        format, built from ITS positions buffer, and :meth:`refresh` rebuilds it in place by replaying a second captured graph
        (cell-list binning + the walk that writes the rows): no list tensors, no sort, no re-capture of the step -- the MD form
        of the reference's "new list every call" (``examples/02-neighbor-lists-usage.py:97-164``).
    :param periodic: per-axis periodicity of the neighbour list made for ``neighbors=<cutoff>``
    :param live_bins: (``neighbors=`` form) also keep the atom -> mesh-brick bookkeeping across steps and rebuild it in
        :meth:`refresh`, evaluating every mesh weight on the fly from the current positions (``mipme_md_step``: five launches
        per step instead of six, no per-step binning pass, no per-brick candidate scan).  Valid while no atom has moved more
        than ONE MESH POINT since the last refresh -- :attr:`max_displacement` is that distance in the units of the cell;
        compare it with the skin in ``neighbors`` (a refresh criterion of skin / 2 can exceed it on a fine mesh).  Every step
        checks the margin on the device.  A step that violates it returns ``NaN`` as its energy (its other results are
        invalid too) and the next use of the object raises; ``step(positions, check=True)`` looks at the flag before it
        returns -- one stream synchronisation -- and, on a violation, refreshes the neighbour structures and evaluates again,
        so that what it returns is always valid.  In this mode the charges live in the object's records: change them with
        :meth:`set_charges`.  Default: on where the kernels cover the case (mesh calculators with 1/r or 1/r^6).
    :param energy_log: an :class:`EnergyLog` (or a capacity, for a new one: ``self.energy_log``) that every replay appends its
        energy to -- one more node at the end of the captured graph (frame farm: one exchange for many evaluations)
    :param epilogue: ``epilogue(step)`` is called once, INSIDE the capture, after everything else: whatever it launches on the
        current stream (e.g. an RCCL collective on ``step.energy``) becomes the tail of the replayed graph
    """

    def __init__(self, calculator, charges, cell, positions, neighbor_indices=None, neighbor_shifts=None, warmup: int = 3,
                 cell_gradient: bool = False, store_distances: bool = False, neighbors=None,
                 periodic=(True, True, True), live_bins: bool | None = None, charge_gradient: bool = False,
                 energy_log=None, epilogue=None):
        self.calc = calculator
        self.energy_log = _as_energy_log(energy_log, 1, positions.device)
        self._epilogue = epilogue
        self.store_distances = bool(store_distances)
        self.charge_gradient = bool(charge_gradient)
        self.stream = None
        if neighbors is not None:
            if neighbor_indices is not None or neighbor_shifts is not None:
                raise ValueError("give either `neighbors` or `neighbor_indices` / `neighbor_shifts`")
            if store_distances:
                raise ValueError("a device neighbour stream has no pair order to store distances in")
        elif neighbor_indices is None or neighbor_shifts is None:
            raise ValueError("`neighbor_indices` and `neighbor_shifts` (or `neighbors`) are required")
        self.q = charges.detach()
        #: with ``cell_gradient=True`` every call also returns dE/dcell (3,3) -- the virial is ``-cell.T @ dE/dcell``
        self.cell_gradient = bool(cell_gradient)
        self.cell = cell.detach().clone() if cell_gradient else cell.detach()
        self.pos = positions.detach().clone().requires_grad_(True)
        device = positions.device
        # seeding the backward pass with -1 makes ``pos.grad`` the forces directly (no fill and no negation kernel); the
        # derivatives w.r.t. charges and cell are asked of the same gather tail with a seed of their own (+1)
        self._minus_one = torch.tensor(-1.0, dtype=positions.dtype, device=device)
        self._one = torch.tensor(1.0, dtype=positions.dtype, device=device)
        self._fused_contract = True  # False: charges / cell gradients through the general autograd nodes (see _capture)
        self.charge_grad = self.cell_grad = None
        self._warmup = max(1, warmup)
        self.refresh_graph = None
        self._live = None
        if neighbors is not None and live_bins is not False and hasattr(calculator, "_kspace_setup") \
                and calculator.potential.smearing is not None:
            try:
                live = _LiveStep(calculator, self.q, self.cell, self.pos, self.charge_gradient, self.cell_gradient)
                live.rebin()  # trial: a very non-uniform system overflows the per-brick lists -> keep the binned step
                torch.cuda.current_stream(device).synchronize()
                live.check()
                self._live = live
                # the records are the atoms' storage from here on: `pos` is their (N, 3) view
                self.pos = live.rec[:, :3]
            except (NotImplementedError, RuntimeError):
                if live_bins:
                    raise
                self._live = None
        if neighbors is not None:
            from .neighbors import NeighborStream

            if isinstance(neighbors, NeighborStream) and self._live is not None:
                raise ValueError("with live bins the neighbour stream is built on the object's own records: pass the cutoff")
            if isinstance(neighbors, NeighborStream):
                if neighbors.n_atoms != self.pos.shape[0] or neighbors.dtype != self.pos.dtype:
                    raise ValueError("`neighbors` was built for other positions")
                self.stream = neighbors
                self.stream.positions = self.pos  # from now on the list is rebuilt from the graph's own buffer
                self.stream.update()
            else:
                self.stream = NeighborStream(self.pos, self.cell, float(neighbors), periodic=periodic)
            self.stream.check(synchronize=True)
            self._capture_refresh()
            self._capture(self.stream.indices, None)
        else:
            self._capture(neighbor_indices, neighbor_shifts)

    # ---- device neighbour list --------------------------------------------------------------------------------------------
    def _capture_refresh(self):
        device = self.pos.device
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        if self._live is not None:
            self._live.rows = (self.stream.row_ptr, self.stream.words, 2 | _lib.ROWS_PADDED)
        with torch.cuda.stream(side):
            self._refresh_eager()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.refresh_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.refresh_graph):
            self._refresh_eager()

    def _refresh_eager(self):
        self.stream.update()
        if self._live is not None:
            self._live.rebin()

    def refresh(self, positions: torch.Tensor | None = None, check: bool = False) -> None:
        """Rebuild the neighbour list from the current positions, in place (``neighbors=`` form only): one graph replay --
        binning, the walk that writes the rows, the status report --, after which the step graph reads the new list at the
        same addresses.  ``check=True`` waits for it and handles a row that outgrew its capacity (larger buffers, both
        graphs captured again); otherwise the status word is looked at when the object is next used."""
        if self.stream is None:
            raise RuntimeError("refresh() needs the `neighbors=` form; use recapture() with a new list otherwise")
        self._deferred_check()
        if positions is not None:
            with torch.no_grad():
                self.pos.copy_(positions)
        self.refresh_graph.replay()
        if check:
            torch.cuda.current_stream(self.pos.device).synchronize()
            self._deferred_check(recover=True)

    def _deferred_check(self, recover: bool = False):
        if self._live is not None:
            self._live.check()
        st = self.stream
        if st is None or not int(st._host_np[1]):
            return
        if recover and int(st._host_np[1]) == 1:  # only a row overflow: grow, capture again, rebuild
            st.grow()
            st.check(synchronize=True)
            self._capture_refresh()
            self._capture(st.indices, None)
            return
        st.check()

    def recapture(self, neighbor_indices, neighbor_shifts, positions: torch.Tensor | None = None) -> None:
        """New neighbour list (e.g. after the atoms moved by more than the skin): rebuild the pair topology and capture the
        step again; positions, cell and charges buffers are kept."""
        if positions is not None:
            with torch.no_grad():
                self.pos.copy_(positions)
        self._capture(neighbor_indices, neighbor_shifts)

    def _capture_live(self):
        device = self.pos.device
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        self._live.log = None  # (a re-capture after an overflow warms up again: those evaluations are not the caller's)
        with torch.cuda.stream(side):  # warm-up off the default stream (the plan allocates its scratch on first use)
            for _ in range(self._warmup):
                self._live.step()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self._keepalive = [self._live, self.stream, getattr(self.calc, "_cache", None)]
        self._live.log = self.energy_log  # (from here on: the gather of the step appends its energy to the log)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._live.step()
            self.energy = self._live.energy
            if self._epilogue is not None:
                self._epilogue(self)
        self.energy, self.forces, self.distances = self._live.energy, self._live.grad, None
        self.charge_grad = self._live.grad_q
        self.cell_grad = None if self._live.grad_cell is None else self._live.grad_cell[18:27].view(3, 3)
        self.pairs, self.shifts = self.stream.indices, None

    def _capture(self, neighbor_indices, neighbor_shifts):
        if self._live is not None:
            return self._capture_live()
        calculator, device, warmup = self.calc, self.pos.device, self._warmup
        self.pairs = neighbor_indices
        self.shifts = None if neighbor_shifts is None else neighbor_shifts.to(self.pos.dtype).contiguous()
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):  # warm-up off the default stream: plans, topology, filter caches get built
            for it in range(max(1, warmup) + 1):
                self.pos.grad = self.cell.grad = self.q.grad = None
                self._eval()
                if it == 0 and (self.charge_gradient or self.cell_gradient) and self._fused_contract:
                    # did the gather tail take the request (seed_promise(charges=, cell=))?  Otherwise the general nodes:
                    # charges / cell become leaves and autograd differentiates w.r.t. them
                    t = self._tail
                    ok = t is not None and (t["grad_q"] is not None or not self.charge_gradient) and (
                        t["grad_cell"] is not None or not self.cell_gradient)
                    if not ok:
                        self._fused_contract = False
                        if self.charge_gradient:
                            self.q = self.q.clone().requires_grad_(True)
                        if self.cell_gradient:
                            self.cell.requires_grad_(True)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.pos.grad = self.cell.grad = self.q.grad = None
        # The captured graph holds raw pointers into buffers that were built during the warm-up and live in caches: the
        # transposed pair list (+ packed shifts), the calculator's filter table, the reduction scratch.  Keep them alive
        # for the lifetime of the graph, whatever the caches evict later.
        topo = ops.get_topology(self.pairs, self.pos.shape[0]) if ops.PAIR_MODE == "rows" else None
        self._keepalive = [
            topo,
            getattr(calculator, "_cache", None),
            ops._dot_scratch(device, self.q.data_ptr()),
        ]
        if isinstance(topo, ops.PairTopology):
            # the entry streams are single-slot caches keyed by the shifts tensor: an eager call with the same pairs and other
            # shifts would replace the slot and free the buffer the graph reads -- pin the concrete tensors
            self._keepalive += [topo.row_ptr, topo.entries, topo._ent32, topo._ent_sh, topo._packed, topo._pair_sh]
        elif topo is not None:
            self._keepalive += [self.stream.row_ptr, self.stream.words]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.energy = self._eval(log=self.energy_log)
            self.forces = self.pos.grad
            if self.energy_log is not None and not (self._tail is not None and self._tail.get("logged")):
                self.energy_log.push(self.energy.reshape(1))  # (no gather tail in this step: one more node)
            if self._epilogue is not None:
                self._epilogue(self)
            if self._fused_contract:
                t = self._tail
                self._keepalive.append(t)  # (G_deriv, cell_work, the output buffers)
                self.charge_grad = t["grad_q"] if self.charge_gradient else None
                self.cell_grad = t["grad_cell"][18:27].view(3, 3) if self.cell_gradient else None
            else:
                # the backward pass is seeded with -1 (so that pos.grad is the force): undo the sign for charges and cell
                self.charge_grad = -self.q.grad if self.charge_gradient else None
                self.cell_grad = -self.cell.grad if self.cell_gradient else None

    def _eval(self, log=None):
        # the pair kernel of the calculator forms the distances itself (no separate pass over the list); "virtual": in
        # registers only, True: stored as a by-product
        if self.stream is not None:
            d = self.stream.distances(self.pos, self.cell)
        else:
            d = ops.pair_distances(self.pos, self.pairs, self.cell, self.shifts,
                                   deferred=True if self.store_distances else "virtual")
        #: the pair distances of the last evaluation (P,) with ``store_distances=True``, else None
        self.distances = d.detach() if self.store_distances else None
        # the backward pass below is seeded with self._minus_one: promise that to the forward, whose gather then writes the
        # forces themselves (energy reduction and force assembly ride in the gather launch, see ops.SEED_PROMISE) -- and, on
        # request, dE/dcharges and dE/dcell with a seed of +1
        fused = self._fused_contract
        with ops.seed_promise(self._minus_one, charges=fused and self.charge_gradient, cell=fused and self.cell_gradient,
                              aux_seed=self._one, energy_log=log):
            V = self.calc(self.q, self.cell, self.pos, self.pairs, d)
        self._tail = getattr(V.grad_fn, "tail", None)
        E = ops.weighted_sum(V, self.q)
        E.backward(self._minus_one)
        return E.detach()

    @property
    def max_displacement(self):
        """How far an atom may move between two :meth:`refresh` calls of the live-bin step (one mesh point, in the units of the
        cell); ``None`` for the binned step, which re-bins every call."""
        return None if self._live is None else self._live.max_displacement

    def set_charges(self, charges: torch.Tensor) -> None:
        """New charge values (same shape) for the following steps.  The binned step reads the tensor given at construction, so
        writing into that tensor has the same effect there; the live-bin step keeps the charges in its atom records."""
        with torch.no_grad():
            if self._live is not None:
                self._live.rec[:, 3] = charges.detach()[:, 0]
            self.q.copy_(charges.detach())

    def __call__(self, positions: torch.Tensor | None = None, check: bool = False):
        """``E, F`` (+ ``dE/dcharges`` with ``charge_gradient``, + ``dE/dcell`` with ``cell_gradient``, in that order) of the
        current -- or the given -- positions: one graph replay.  The returned tensors are the graph's own buffers.

        ``check=True`` (live-bin step): wait for the step and look at its margin flag; if an atom had moved more than one mesh
        point since the last refresh, :meth:`refresh` (neighbour list and bins from the current positions) and evaluate again
        before returning -- the results are then valid whatever the atoms did, at the price of one synchronisation per call."""
        self._deferred_check()
        self.calc.check()  # a NaN a previous replay met (pinned word, no synchronisation)
        if positions is not None:
            with torch.no_grad():
                self.pos.copy_(positions)
        self.graph.replay()
        if check and self._live is not None:
            torch.cuda.current_stream(self.pos.device).synchronize()
            if self._live.moved():
                self.refresh(check=True)
                self.graph.replay()
                torch.cuda.current_stream(self.pos.device).synchronize()
                self._deferred_check()
        out = (self.energy, self.forces)
        if self.charge_gradient:
            out += (self.charge_grad,)
        if self.cell_gradient:
            out += (self.cell_grad,)
        return out


class _FramesFunction(torch.autograd.Function):
    """Energies of all frames of a :class:`GraphedFrameBatch` as one autograd node: forward = the batched pipeline
    (``mipme_frames_forward``), backward = the batched force assembly (``mipme_frames_backward``)."""

    @staticmethod
    def forward(ctx, batch, *positions):
        batch._launch_forward()
        ctx.batch = batch
        return batch.energies.detach()  # a fresh tensor object per call (the buffer itself is persistent)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        batch = ctx.batch
        # seeded with the batch's own -1 vector the gradients are already there (written by the gather's tail of the forward
        # pass); fresh aliases, so that the accumulation into positions.grad takes the buffers instead of copying them
        if g.data_ptr() != batch._minus_one.data_ptr():
            batch._launch_backward(g.contiguous())
        return (None, *(b.detach() for b in batch._grad_pos))


class GraphedFrameBatch:
    """Energy + forces of several independent frames (the frames a rank owns) with ONE launch per kernel of the
    pipeline -- ``blockIdx.y`` = frame (``mipme_frames_forward / backward``, csrc/bricks.hip) -- replayed from one HIP graph.

    The frames may differ in atoms, cell and neighbour list; they must give the same mesh dimensions and share the
    calculator (P3M / PME with 1/r or 1/r^6), dtype and device.  ``energies, forces = batch(positions_list=None)``:
    ``energies`` (F,), ``forces`` a list of (N_f, 3) tensors (static buffers of the graph: read them before the next call).

    :param calculator: a :class:`PMECalculator` / :class:`P3MCalculator`
    :param frames: sequence of ``(charges, cell, positions, neighbor_indices, neighbor_shifts)``
    :param store_distances: keep every frame's pair distances in ``self.distances[f]`` (by-product of the pair kernel; needs a
        list ordered by its first index); off by default, see :class:`GraphedEnergyForces`
    :param energy_log: an :class:`EnergyLog` (or a capacity) that every replay appends its F energies to, see there
    :param epilogue: ``epilogue(batch)``, called once inside the capture after everything else (see :class:`GraphedEnergyForces`)
    """

    def __init__(self, calculator, frames, warmup: int = 2, store_distances: bool = False, energy_log=None, epilogue=None):
        lib = _lib.load()
        self.calc = calculator
        self.store_distances = bool(store_distances)
        frames = list(frames)
        F = self.n_frames = len(frames)
        if F == 0:
            raise ValueError("no frames")
        q0, _, p0, _, _ = frames[0]
        device, dtype = p0.device, p0.dtype
        _lib.require_device(p0, "positions")
        self.device, self.dtype = device, dtype
        dt = self._dt = _lib.dtype_code(dtype)
        self._pot = calculator.potential._descriptor()
        full = bool(calculator.full_neighbor_list)
        self.pos, self._keep, geoms, Gs = [], [], [], []
        for q, cell, pos, pairs, shifts in frames:
            if q.shape[1] != 1:
                raise ValueError("the frames path handles a single charge channel")
            geom, G = calculator._kspace_setup(cell, dtype, device, speculate=False)
            geoms.append(geom)
            Gs.append(G.reshape(-1))
        ns = geoms[0].ns
        if any(g.ns != ns for g in geoms):
            raise ValueError(f"all frames must give the same mesh, got {[g.ns for g in geoms]}")
        self.ns = ns
        M, Mh = geoms[0].n_mesh, geoms[0].n_half
        cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
        self._G = torch.stack(Gs).contiguous()
        self._rho = torch.empty((F,) + ns, dtype=dtype, device=device)
        self._phi = torch.empty((F,) + ns, dtype=dtype, device=device)
        self._hat = torch.empty((F, Mh), dtype=cdtype, device=device)
        self._dc = torch.empty((F,), dtype=dtype, device=device)
        self.energies = torch.empty((F,), dtype=dtype, device=device)
        self._plan = _lib.FFTPlan(device, dtype, ns, F)
        self._frames = (_lib.Frame * F)()
        self._minus_one = torch.full((F,), -1.0, dtype=dtype, device=device)
        self._grad_pos, self.distances = [], []
        compact_ok, ent_streams = True, []
        for k, (q, cell, pos, pairs, shifts) in enumerate(frames):
            N, P = pos.shape[0], pairs.shape[0]
            p = pos.detach().clone().contiguous().requires_grad_(True)
            qc = q.detach().reshape(-1).contiguous()
            cl = cell.detach().to(dtype).contiguous()
            sh = shifts.to(dtype).contiguous()
            topo = ops.get_topology(pairs, N)
            ent_sh, fmt = topo.entries_with_shifts(sh, shifts, table=True)
            if ent_sh is None or fmt != 1:
                raise ValueError(f"frame {k}: cell shifts must be integers in [-3, 3] for the frames path")
            ent32 = topo.compact_entries(sh, shifts)  # 4-byte entries when every frame has them
            compact_ok = compact_ok and ent32 is not None
            ent_streams.append((ent_sh, ent32))
            md = geoms[k].desc(1)
            nbytes = lib.mipme_atom_bins_bytes(C.byref(md), N, dt)
            if nbytes <= 0:
                raise ValueError(f"frame {k}: mesh {ns} is outside the brick kernels' range")
            nb = ((ns[0] + 7) // 8) * ((ns[1] + 7) // 8) * ((ns[2] + 7) // 8)
            buf = dict(
                bins=torch.empty((nbytes,), dtype=torch.uint8, device=device),
                # (brick counters + the plane lists' counters where the plane spread applies: mipme_frame_t.counter_ints)
                counters=torch.zeros((max(nb + 1, int(lib.mipme_frames_counter_ints(C.byref(md), N, dt))),), dtype=torch.int32,
                                     device=device),
                records=torch.empty((N, 4), dtype=dtype, device=device),
                out=torch.empty((N,), dtype=dtype, device=device),
                force=torch.empty((N, 3), dtype=dtype, device=device),
                field=torch.empty((N, 3), dtype=dtype, device=device),
                grad=torch.empty((N, 3), dtype=dtype, device=device),
                dist=(torch.empty((P,), dtype=dtype, device=device)
                      if self.store_distances and topo.sorted_by_first and P > 0 else None),
            )
            f = self._frames[k]
            f.n_atoms, f.positions, f.charges, f.cell, f.mesh = N, p.data_ptr(), qc.data_ptr(), cl.data_ptr(), md
            f.atom_bins, f.brick_counters = buf["bins"].data_ptr(), buf["counters"].data_ptr()
            f.counter_ints = int(buf["counters"].numel())
            f.row_ptr, f.entries_shift, f.entries = topo.row_ptr.data_ptr(), ent_sh.data_ptr(), topo.entries.data_ptr()
            f.full_list, f.shift_format = int(full), int(fmt)
            f.records = buf["records"].data_ptr()
            f.rho_mesh, f.phi_mesh, f.dc = self._rho[k].data_ptr(), self._phi[k].data_ptr(), self._dc[k:].data_ptr()
            f.out, f.force, f.field = buf["out"].data_ptr(), buf["force"].data_ptr(), buf["field"].data_ptr()
            f.dist_out = _lib.ptr(buf["dist"])
            f.energy, f.grad_positions = self.energies[k:].data_ptr(), buf["grad"].data_ptr()
            # energy + forces of the frame in the gather launch, seeded with the -1 the backward pass of _eval() uses
            f.use_tail, f.grad_seed = 1, self._minus_one[k:].data_ptr()
            self.pos.append(p)
            self._grad_pos.append(buf["grad"])
            self.distances.append(buf["dist"])
            self._keep.append((qc, cl, sh, topo, ent_sh, ent32, buf, pairs))
        if compact_ok:  # the same (compact) entry format for every frame
            for k, (_, ent32) in enumerate(ent_streams):
                self._frames[k].entries_shift, self._frames[k].shift_format = ent32.data_ptr(), 2
        nbytes = lib.mipme_frames_table_bytes(dt, F)
        host = np.zeros((nbytes,), dtype=np.uint8)
        _lib.check(lib.mipme_frames_table_build(dt, F, self._frames, C.byref(self._pot), host.ctypes.data, nbytes))
        self.energy_log = _as_energy_log(energy_log, F, device)
        if self.energy_log is not None:  # every frame's gather tail also writes its slot of the log (no extra launch)
            _lib.check(lib.mipme_frames_table_energy_log(dt, F, host.ctypes.data, nbytes, self.energy_log.values.data_ptr(),
                                                         self.energy_log.cursor.data_ptr(), self.energy_log.capacity))
        self._table = torch.from_numpy(host).to(device)
        # warm-up (plans, lazy module loads) off the default stream, then capture
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eval()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        if self.energy_log is not None:
            self.energy_log.reset()  # (the warm-up evaluations above appended too)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._eval()
            if epilogue is not None:
                epilogue(self)
        self.forces = [p.grad for p in self.pos]

    def _launch_forward(self):
        with _lib.on_device(self.device):
            _lib.check(_lib.load().mipme_frames_forward(
                self._plan.handle, _lib.current_stream(self.device), self._dt, self.n_frames, self._frames,
                C.byref(self._pot), self._table.data_ptr(), self._G.data_ptr(), self._G.shape[1], self._rho.data_ptr(),
                self._hat.data_ptr(), self._phi.data_ptr(), self._dc.data_ptr()))

    def _launch_backward(self, g):
        with _lib.on_device(self.device):
            _lib.check(_lib.load().mipme_frames_backward(
                _lib.current_stream(self.device), self._dt, self.n_frames, self._frames, self._table.data_ptr(),
                g.data_ptr()))

    def _eval(self):
        for p in self.pos:
            p.grad = None
        E = _FramesFunction.apply(self, *self.pos)
        E.backward(self._minus_one)  # seeded with -1: positions.grad are the forces
        return E

    def __call__(self, positions=None):
        if positions is not None:
            with torch.no_grad():
                for buf, new in zip(self.pos, positions):
                    buf.copy_(new)
        self.calc.check()  # a NaN a previous replay met (pinned word, no synchronisation)
        self.graph.replay()
        return self.energies, self.forces

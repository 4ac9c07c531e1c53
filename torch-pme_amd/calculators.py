"""Calculators with the reference's constructor / ``forward`` signatures
(``calculators/calculator.py:9-189``, ``calculators/pme.py:10-143``, ``calculators/p3m.py:9-84``),
running on hand-written HIP kernels for gfx950 through ``libmipme.so``.

``forward(charges, cell, positions, neighbor_indices, neighbor_distances, periodic=None,
node_mask=None, pair_mask=None, kvectors=None) -> (N, C)`` is differentiable (first order) with respect
to ``charges``, ``cell``, ``positions`` and ``neighbor_distances``.
"""

from __future__ import annotations

import math
import weakref

import numpy as np
import torch

from . import _front, _lib, ops
from ._utils import _validate_parameters
from .potentials import Potential

import os

#: a NEW cell tensor is first assumed to hold the values of the previous one (verified on the device, see
#: PMECalculator._kspace_setup); "0": always copy it to the host first, as the reference does
SPECULATE_CELL = True


class Calculator(torch.nn.Module):
    """Real-space pair sum ``V_i = 1/2 sum_j q_j v(r_ij)``; base class of the mesh calculators.

    :param potential: a :class:`Potential` (``CoulombPotential`` or ``InversePowerLawPotential``)
    :param full_neighbor_list: whether the neighbour list holds each pair twice (True) or once (False)
    """

    def __init__(self, potential: Potential, full_neighbor_list: bool = False):
        super().__init__()
        if not isinstance(potential, Potential):
            raise TypeError(f"Potential must be an instance of Potential, got {type(potential)}")
        self.potential = potential
        self.full_neighbor_list = full_neighbor_list
        #: NaN guard of the reference (``lib/kspace_filter.py:189-195``: ValueError when the k-space result holds a NaN).  The
        #: reference pays a device synchronisation per call for it.  Here the gather kernels raise a flag in pinned host
        #: memory whenever a potential they write is NaN (no extra launch, no synchronisation), and the flag is looked at
        #:   "deferred" (default) -- at the start of the NEXT call of this calculator and in :meth:`check`: the same error,
        #:                           one call late, at no cost;
        #:   True                 -- right after the call (synchronises, as the reference does);
        #:   False                -- never.
        self.check_nan = "deferred"
        #: second derivatives (``create_graph=True``: training on forces).  The fused HIP kernels are first order; ``None``
        #: (default) makes a double differentiation raise a RuntimeError that names this attribute.  ``"analytic"`` evaluates
        #: the call through :mod:`analytic` instead -- the same mathematics as a composition of linear primitives whose
        #: backward passes are made of the same primitives (``csrc/jets.hip``), exact to any order w.r.t. charges, positions,
        #: cell and distances, as the reference's ATen chain is (mesh calculators and the plain pair sum).
        #: ``"auto"``: potentials and a plain backward pass from the fused kernels, and only a backward pass that is itself
        #: recorded (``create_graph=True``) through the primitives -- for a calculator that serves training on forces and plain
        #: energy / force evaluations alike.
        #: ``"finite-difference"`` keeps the fused first-order kernels and forms the backward of the backward from central
        #: differences of their gradients (two more evaluations; use float64)
        self.double_backward = None
        self._nan_flag = None  # pinned int32[1], created on first use
        self._nan_shape = None
        self._spec_str = None
        if type(self) is Calculator:
            self._spec()

    # ---- copies and checkpoints ------------------------------------------------------------------------------------------
    #: per-instance device state that must not travel with a copy / pickle (FFT plans own raw device pointers, the caches
    #: hold weak references to the caller's tensors); rebuilt on first use
    _TRANSIENT = {"_cache": None, "_plan_store": dict, "_freq_cache": None, "_nan_flag": None, "_nan_shape": None,
                  "_speculated": None, "_bet_flag": None, "_bet_flag_np": None, "_bet_skip": 0, "_bet_backoff": 2, "_analytic_geom": None}


    def __getstate__(self):
        state = self.__dict__.copy()
        for name, fresh in self._TRANSIENT.items():
            if name in state:
                state[name] = fresh() if callable(fresh) else fresh
        return state

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Checkpoints written by the reference's calculators carry ``kspace_filter.*`` entries (copies of the potential's
        buffers inside its filter module, and the finite-difference table of ``P3MKSpaceFilter``): this build derives the
        filter from ``potential`` alone, so they are accepted and ignored -- a strict load of a reference checkpoint works."""
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        ignored = prefix + "kspace_filter."
        unexpected_keys[:] = [k for k in unexpected_keys if not k.startswith(ignored)]

    # ---- NaN guard ---------------------------------------------------------------------------------------------------------
    def _nan_flag_ptr(self):
        if self.check_nan is False or self.check_nan is None:
            return None
        if self._nan_flag is None:
            self._nan_flag = torch.zeros((1,), dtype=torch.int32).pin_memory()
        return self._nan_flag.data_ptr()

    def check(self) -> None:
        """Raise the reference's ``ValueError`` if a NaN was seen in the results of an earlier call (see ``check_nan``).  The
        flag is host memory written by the device: a call that is still running has not set it yet."""
        flag = self._nan_flag
        if flag is not None and int(flag[0]) != 0:
            flag[0] = 0
            shape = self._nan_shape
            raise ValueError(
                "NaNs detected in the k-space filter result. This are probably caused "
                "by an unsuitable `mesh_spacing`, resulting in a problematic grid of "
                f"shape: {shape}. Try adjsuting the grid by using a "
                "different `mesh_spacing` value."
            )

    # mesh calculators override this to return (MeshGeometry, G); the base class has no k-space part
    def _kspace_setup(self, cell, dtype, device, speculate: bool = True):
        if self.potential.smearing is not None:
            raise NotImplementedError(f"`compute_kspace` not implemented for {self.__class__.__name__}")
        return None, None

    def _spec(self):
        """JSON description of this calculator for the dispatcher op (``library.calculator_spec``).  Recorded at the end of
        the constructors -- ``torch.compile`` needs it as a constant while tracing -- and refreshed by ``scriptable()``;
        ``None`` for calculators the op cannot rebuild (custom ``Potential`` subclasses)."""
        from . import library

        try:
            self._spec_str = library.calculator_spec(self)
        except TypeError:
            self._spec_str = None
        return self._spec_str

    #: public attributes the spec string encodes: assigning one of them refreshes it at once (eagerly -- building the JSON is
    #: not traceable), so the compiled branch of ``forward`` never sees a stale description and dynamo, which guards on the
    #: string, recompiles
    _SPEC_ATTRS = frozenset({"full_neighbor_list", "mesh_spacing", "interpolation_nodes", "lr_wavelength", "potential"})

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if name in self._SPEC_ATTRS and self.__dict__.get("_spec_str") is not None:
            self._spec()

    def scriptable(self):
        """A TorchScript-compatible module with the same ``forward`` (``torch.jit.script(calculator.scriptable())``): the
        counterpart of the reference's ``torch.jit.script(calculator)`` (``tests/calculators/test_workflow.py:136-162``)."""
        from . import library

        spec = self._spec()
        if spec is None:
            raise TypeError(f"{type(self).__name__} with a {type(self.potential).__name__} has no dispatcher op")
        return library.ScriptableCalculator(spec)

    def __prepare_scriptable__(self):
        """``torch.jit.script(calculator)`` -- what the reference's users write (``tests/calculators/test_workflow.py:136-162``)
        -- scripts :meth:`scriptable`'s module: TorchScript asks a module for its scriptable stand-in through this hook."""
        return self.scriptable()

    def forward(
        self,
        charges: torch.Tensor,
        cell: torch.Tensor,
        positions: torch.Tensor,
        neighbor_indices: torch.Tensor,
        neighbor_distances: torch.Tensor,
        periodic: torch.Tensor | None = None,
        node_mask: torch.Tensor | None = None,
        pair_mask: torch.Tensor | None = None,
        kvectors: torch.Tensor | None = None,
    ):
        """Per-atom potentials ``(n_atoms, n_channels)``; see the reference docstring
        (``calculators/calculator.py:115-156``) for the meaning of every argument.  Under ``torch.vmap`` (padded batches
        with ``node_mask`` / ``pair_mask`` / ``kvectors``) the structures are evaluated one after the other."""
        if torch.compiler.is_compiling() and self._spec_str is not None:
            # inside torch.compile: one dispatcher op (fake implementation + autograd formula in library.py) instead of
            # ctypes + HIP launches, which cannot be traced -- the model stays a single graph
            from . import library

            return torch.ops.mipme.potentials(charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask,
                                              periodic, node_mask, kvectors, self._spec_str,
                                              library.needs_mask(charges, cell, positions, neighbor_distances))
        return self._eager_forward(charges, cell, positions, neighbor_indices, neighbor_distances, periodic, node_mask,
                                   pair_mask, kvectors)

    @torch.compiler.disable
    def _eager_forward(self, *args):
        if ops.inside_vmap(*args):
            return ops.vmap_bridge(self._forward_impl, *args)
        if self.double_backward is not None and torch.is_grad_enabled() and any(
                isinstance(a, torch.Tensor) and a.requires_grad for a in args):
            if self.double_backward not in ("finite-difference", "analytic", "auto"):
                raise ValueError(
                    f"`double_backward` is {self.double_backward!r} but must be None, 'auto', 'analytic' or 'finite-difference'")
            charges, cell, positions, pairs, dist, *rest = args
            # the distances are an ordinary differentiable input here (no fused / lazy pair gradient: the chain through
            # `pair_distances` is exact second order by itself)
            dist = dist.materialize() if isinstance(dist, ops.LazyPairGradient) else dist
            src = getattr(dist, "_mipme_src", None)
            if src is not None and src.pending:
                src.materialize()
            if self.double_backward == "analytic":
                from . import analytic

                return analytic.potentials(self, charges, cell, positions, pairs, dist, *rest)

            def first_order_eval(q, c, p, d, *others):
                # plain (unfused) evaluation: `d` carries no provenance, so the calculator differentiates w.r.t. it as a tensor
                return self._forward_impl(q, c, p, pairs, d, *others)

            if self.double_backward == "auto":
                from . import analytic

                def exact_eval(q, c, p, d, *others):
                    return analytic.potentials(self, q, c, p, pairs, d, *others)

                return ops.fused_first_analytic_higher(first_order_eval, exact_eval, (charges, cell, positions, dist), tuple(rest))
            return ops.second_order_by_finite_differences(first_order_eval, (charges, cell, positions, dist), tuple(rest))
        if ops.FRONT and ops.PROFILE is None and len(args) >= 5 and all(a is None for a in args[5:]) and self.check_nan is not True:
            out = self._front_forward(*args[:5])  # compiled host path of the common case (csrc/front.cpp); None: not that case
            if out is not None:
                return out
        return self._forward_impl(*args)

    def _front_forward(self, charges, cell, positions, neighbor_indices, neighbor_distances):
        return None

    def _forward_impl(self, charges, cell, positions, neighbor_indices, neighbor_distances, periodic=None, node_mask=None,
                      pair_mask=None, kvectors=None):
        _validate_parameters(
            charges=charges,
            cell=cell,
            positions=positions,
            neighbor_indices=neighbor_indices,
            neighbor_distances=neighbor_distances,
            periodic=periodic,
            pair_mask=pair_mask,
            node_mask=node_mask,
            kvectors=kvectors,
        )
        _lib.require_device(positions, "positions")
        if self.check_nan == "deferred":
            self.check()  # a NaN seen by an earlier call surfaces here
        pot_desc = self.potential._descriptor()
        has_kspace = self.potential.smearing is not None
        if has_kspace and (node_mask is not None or kvectors is not None):
            raise NotImplementedError("Batching not implemented for mesh-based calculators")
        geom, G = self._kspace_setup(cell, positions.dtype, positions.device)
        slab_axis = None
        is_coulombic = pot_desc.kind == _lib.COULOMB or pot_desc.exponent == 1  # the slab term exists for 1/r only
        if has_kspace and periodic is not None and is_coulombic:
            slab_axis = ops._slab_axis(periodic.tolist())
        nan_flag = self._nan_flag_ptr() if geom is not None else None
        if nan_flag is not None:
            self.__dict__["_nan_shape"] = [charges.shape[1], *geom.ns]
        with ops.betting():
            out = ops.pme_potential(
                charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask, geom, G, pot_desc,
                bool(self.full_neighbor_list), slab_axis, nan_flag,
            )
        try:
            ops.verify_bets()  # (ops.SPECULATE_LISTS: the structures of a previous list tensor may have been reused on a bet)
        except ops.SpeculationLost:
            out = ops.pme_potential(
                charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask, geom, G, pot_desc,
                bool(self.full_neighbor_list), slab_axis, nan_flag,
            )
        if getattr(self, "_speculated", None) is not None:
            # the geometry was the one cached for the PREVIOUS cell tensor, on the bet that the new one holds the same values
            # (_kspace_setup): the comparison ran first in the queue -- look at its verdict now that everything is launched
            if not self._speculation_held():
                geom, G = self._kspace_setup(cell, positions.dtype, positions.device, speculate=False)
                if nan_flag is not None:
                    self.__dict__["_nan_shape"] = [charges.shape[1], *geom.ns]
                out = ops.pme_potential(
                    charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask, geom, G, pot_desc,
                    bool(self.full_neighbor_list), slab_axis, nan_flag,
                )
        if self.check_nan is True and geom is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(positions.device).synchronize()
            self.check()
        return out


class PMECalculator(Calculator):
    r"""Particle-mesh Ewald: Lagrange interpolation (nodes 3..7) + reciprocal-space convolution.

    :param potential: potential with a positive ``smearing``
    :param mesh_spacing: target mesh spacing; the mesh is ``2^ceil(log2(2|a_d|/h + 1))`` points per axis
    :param interpolation_nodes: number of interpolation nodes per axis (3, 4, 5, 6 or 7)
    :param full_neighbor_list: see :class:`Calculator`
    """

    _scheme = _lib.LAGRANGE
    _orders = (3, 4, 5, 6, 7)
    _scheme_name = "Lagrange"

    def __init__(
        self,
        potential: Potential,
        mesh_spacing: float,
        interpolation_nodes: int = 4,
        full_neighbor_list: bool = False,
    ):
        super().__init__(potential=potential, full_neighbor_list=full_neighbor_list)
        if potential.smearing is None:
            raise ValueError("Must specify smearing to use a potential with PMECalculator")
        if potential.smearing <= 0:
            raise ValueError(f"`smearing` is {potential.smearing} but must be positive")
        if interpolation_nodes not in self._orders:
            lo, hi = self._orders[0], self._orders[-1]
            raise ValueError(
                f"`interpolation_nodes` is {interpolation_nodes} but only values "
                f"from {lo} to {hi} for method '{self._scheme_name}' are allowed"
            )
        self.mesh_spacing: float = mesh_spacing
        self.interpolation_nodes: int = interpolation_nodes
        self._cache = None  # (weakref(cell), version, dtype, device, pot key, mesh key, geom, G, device copy of the cell)
        self._speculated = self._bet_flag = self._bet_flag_np = None  # see _kspace_setup
        self._bet_skip, self._bet_backoff = 0, 2  # calls that sit a bet out after lost ones (see _speculation_held)
        self._plan_store = {}  # FFT plans of this calculator (see _lib.get_plan)
        self._spec()

    def _kspace_setup(self, cell, dtype, device, speculate: bool = True):
        """Mesh geometry and G(k) for this cell.  Both depend only on (cell, potential); they are cached on
        the identity + version counter of the ``cell`` tensor, so an MD / training loop that reuses its cell
        tensor pays the 9-value D2H copy (needed to size the mesh, as in the reference) only once.

        A NEW cell tensor (the reference's own timing protocol clones its inputs for every call, ``tuning/tuner.py:350-352``;
        data loaders hand out fresh tensors) would pay that copy -- and with it a wait for everything queued on the device --
        every call.  Instead the cached geometry is used on the bet that the values are the same: ``mipme_values_equal``
        compares the new tensor with a device copy of the cached cell, first in the queue, and ``_forward_impl`` looks at the
        verdict after its last launch (``_speculation_held``); a lost bet costs the launches of one evaluation, which is then
        repeated with the right geometry.  Not during graph capture, not when the previous bet was lost."""
        pot_desc = self.potential._descriptor()
        pkey = (pot_desc.kind, pot_desc.exponent, pot_desc.smearing, pot_desc.prefactor)
        c = self._cache
        d = self.__dict__  # (plain attributes: nn.Module.__setattr__ costs ~3 us a time, several times per call)
        d["_speculated"] = None
        if (
            c is not None
            and c[2] == dtype
            and c[3] == device
            and c[4] == pkey
            and c[5] == (self.mesh_spacing, self.interpolation_nodes)
        ):
            if c[0]() is cell and c[1] == cell._version:
                d["_bet_skip"], d["_bet_backoff"] = 0, 2  # a cell that comes back: bets are worth placing again
                return c[6], c[7]
            if (speculate and SPECULATE_CELL and self._bet_skip == 0 and cell.dtype == dtype and cell.is_contiguous()
                    and tuple(cell.shape) == (3, 3) and not torch.cuda.is_current_stream_capturing()):
                flag = self._bet_flag
                if flag is None:
                    flag = d["_bet_flag"] = torch.zeros((1,), dtype=torch.int32).pin_memory()
                    d["_bet_flag_np"] = flag.numpy()
                self._bet_flag_np[0] = -1
                with _lib.on_device(device):
                    _lib.check(_lib.load().mipme_values_equal(_lib.current_stream(device), _lib.dtype_code(dtype), 9,
                                                              cell.data_ptr(), c[8].data_ptr(), flag.data_ptr()))
                d["_speculated"] = cell
                return c[6], c[7]
        cell_host = cell.detach().to("cpu", torch.float64).numpy()
        ns = ops.ns_mesh_from_cell(cell_host, self.mesh_spacing)
        geom = ops.MeshGeometry(cell_host, ns, self._scheme, self.interpolation_nodes)
        geom.plan_store = self._plan_store
        G = ops.build_filter(geom, pot_desc, dtype, device)
        d["_cache"] = (weakref.ref(cell), cell._version, dtype, device, pkey, (self.mesh_spacing, self.interpolation_nodes),
                       geom, G, cell.detach().to(dtype).clone())
        # (the slow path does NOT re-arm the bet: a loop that hands over a different cell every call -- a data set of structures,
        # NPT -- would otherwise bet and lose every time, i.e. evaluate everything twice; a lost bet sits the next calls out)
        if speculate and self._bet_skip > 0:
            d["_bet_skip"] = self._bet_skip - 1
        return geom, G

    def _speculation_held(self) -> bool:
        """Verdict of the comparison ``_kspace_setup`` queued for a new cell tensor (pinned word, polled).  True: the cached
        geometry was the right one, and the cache now answers to the new tensor's identity as well."""
        d = self.__dict__
        cell, d["_speculated"] = self._speculated, None
        flag = self._bet_flag_np
        spins = 0
        while flag[0] == -1:
            spins += 1
            if spins > 5_000_000:
                ops.SPIN_TIMEOUTS["cell"] += 1
                torch.cuda.current_stream(cell.device).synchronize()
                break
        if flag[0] == 1:
            c = self._cache
            d["_cache"] = (weakref.ref(cell), cell._version) + c[2:]
            d["_bet_skip"], d["_bet_backoff"] = 0, 2
            return True
        # lost: the next `backoff` calls with a new cell take the plain path (doubling up to 64: cells that change from call to
        # call cost one repeated evaluation per ever longer stretch instead of one per call)
        d["_bet_skip"] = self._bet_backoff
        d["_bet_backoff"] = min(2 * self._bet_backoff, 64)
        return False


    def _front_forward(self, charges, cell, positions, neighbor_indices, neighbor_distances):
        """The call through the C++ autograd nodes of ``csrc/front.cpp`` when ``neighbor_distances`` comes from this package's
        ``pair_distances`` (its compiled node); gradients for positions, charges and cell; ``None`` sends the call down the Python
        path, which also owns every error message: nothing is validated here beyond what decides the route."""
        mod = _front.module()
        if (mod is None or type(neighbor_distances) is not torch.Tensor or type(cell) is not torch.Tensor
                or cell.shape != (3, 3) or type(positions) is not torch.Tensor or cell.dtype != positions.dtype
                or cell.device != positions.device):
            return None
        if not mod.is_front_distances(neighbor_distances):
            if (neighbor_distances.requires_grad or not positions.is_cuda
                    or getattr(neighbor_distances, "_mipme_src", None) is not None):  # (provenance: the Python nodes' business)
                return None
            return self._front_plain(mod, charges, cell, positions, neighbor_indices, neighbor_distances)
        geom, G = self._kspace_setup(cell, positions.dtype, positions.device, speculate=False)
        want_cell = cell.requires_grad
        fc = self._front_calculator(mod, geom, G, cell, positions, want_cell)
        if fc is None:
            return None
        if self.check_nan == "deferred":
            self.check()  # a NaN seen by an earlier call surfaces here
        out = mod.calc_forward(fc, charges, cell, positions, neighbor_indices, neighbor_distances)
        if out is not None and self._nan_flag is not None:
            self.__dict__["_nan_shape"] = [1, *geom.ns]
        return out

    def _front_calculator(self, mod, geom, G, cell, positions, want_cell):
        """The extension's handle of this calculator on this geometry (descriptors, plan, G and -- for a cell that requires a
        gradient -- its derivative table), or ``None`` when the compiled nodes do not cover the configuration."""
        key = (bool(self.full_neighbor_list), self.check_nan, want_cell)
        c = geom.__dict__.get("_front")
        if c is None or c[0] != key:
            fc = None
            pot_desc = self.potential._descriptor()
            p_eff = 1 if pot_desc.kind == _lib.COULOMB else pot_desc.exponent
            plan = _lib.get_plan(positions.device, positions.dtype, geom.ns, 1, geom.plan_store)
            if (p_eff in (1, 6) and pot_desc.smearing > 0 and pot_desc.exclusion_radius <= 0 and plan.xfused and ops.XFUSED
                    and ops.MESH_MODE == "bricks" and ops.PAIR_MODE == "rows" and ops.COSCHEDULE and ops.ENERGY_FAST_PATH
                    and ops.ENERGY_DETECT and ops.COMPACT_ENTRIES and ops.FUSE_DISTANCES):
                # (a cell that requires a gradient: the derivative table of G for the gather tail's dE/dcell (mipme.h,
                # out_grad_cell); without it the C++ side declines such calls)
                deriv = None
                if want_cell:
                    deriv = ops.filter_derivative(geom, pot_desc, positions.dtype, positions.device)
                fc = mod.Calculator(bytes(geom.desc(1)), bytes(pot_desc), plan.handle.value, G, cell,
                                    bool(self.full_neighbor_list), self._nan_flag_ptr() or 0, geom.n_half, plan, deriv)
            c = geom._front = (key, fc)
        return c[1]

    def _front_plain(self, mod, charges, cell, positions, neighbor_indices, neighbor_distances):
        """Caller-made distances without a history (front.cpp, PlainCalcNode): the reference tuner's timing protocol
        (tuning/tuner.py:337-373) and every call that hands over a neighbour-list library's distances.  The geometry may be the
        one cached for the previous cell tensor, on the bet that the new one holds the same values (``_kspace_setup``); the bet
        is looked at after the launches, and a lost one repeats the call through the Python path."""
        if (type(neighbor_indices) is not torch.Tensor or type(charges) is not torch.Tensor or charges.dim() != 2
                or charges.shape[1] != 1 or neighbor_indices.dim() != 2 or not neighbor_indices.is_contiguous()
                or not neighbor_distances.is_contiguous() or getattr(neighbor_indices, "_mipme_stream", None) is not None
                or not torch.is_grad_enabled() or ops.PAIR_MODE != "rows" or ops.PROFILE is not None
                or not (charges.requires_grad or cell.requires_grad or positions.requires_grad)
                or neighbor_indices.shape[0] == 0 or neighbor_distances.shape != (neighbor_indices.shape[0],)
                or charges.shape[0] != positions.shape[0] or ops.inside_vmap(charges, cell, positions, neighbor_distances)
                # what _validate_parameters would refuse must not reach the native launches: the Python path owns the messages
                or neighbor_indices.shape[1] != 2 or neighbor_indices.dtype not in (torch.int64, torch.int32)
                or neighbor_indices.device != positions.device or charges.device != positions.device
                or charges.dtype != positions.dtype or neighbor_distances.dtype != positions.dtype
                or neighbor_distances.device != positions.device or positions.dim() != 2 or positions.shape[1] != 3):
            return None
        geom, G = self._kspace_setup(cell, positions.dtype, positions.device)
        speculated = self._speculated is not None
        out = None
        fc = self._front_calculator(mod, geom, G, cell, positions, cell.requires_grad)
        if fc is not None:
            with ops.betting():
                topo = ops.get_topology(neighbor_indices, positions.shape[0])
                handle = topo.front_plain(neighbor_indices) if isinstance(topo, ops.PairTopology) else None
                if handle is not None:
                    tab = None
                    if ops.TABULATE:
                        tab = topo.tabulated(neighbor_distances, None, self.potential._descriptor(), bool(self.full_neighbor_list))
                    if self.check_nan == "deferred":
                        self.check()
                    out = mod.calc_forward_plain(fc, handle, charges, cell, positions, neighbor_indices, neighbor_distances,
                                                 None if tab is None else tab[0], None if tab is None else tab[1])
            try:
                ops.verify_bets()  # (a new list tensor with the values of the previous one: structures reused on a bet)
            except ops.SpeculationLost:
                out = None
        if speculated and self._speculated is not None and not self._speculation_held():
            out = None  # (the Python path below starts over with this cell's own geometry)
        elif out is None and speculated:
            self.__dict__["_speculated"] = None
        if out is not None and self._nan_flag is not None:
            self.__dict__["_nan_shape"] = [1, *geom.ns]
        return out


class P3MCalculator(PMECalculator):
    r"""Particle-particle particle-mesh: B-spline charge assignment (nodes 1..5) with the
    influence-function corrected kernel ``G = \hat v_{LR} / U^2`` (reference ``calculators/p3m.py``)."""

    _scheme = _lib.P3M
    _orders = (1, 2, 3, 4, 5)
    _scheme_name = "P3M"


def _reciprocal_and_det(cell: torch.Tensor):
    """``(inv(cell).T, det(cell))`` of (..., 3, 3) cells from cross products: the rows of ``A^-T`` are ``b x c / det``,
    ``c x a / det``, ``a x b / det``.  A handful of element-wise kernels, differentiable, and -- unlike ``torch.linalg.inv`` /
    ``torch.det`` (LU factorisations whose singularity check reads the device back) -- without a host synchronisation:
    the two calls were most of the 0.7 ms an eager Ewald evaluation took whatever its size."""
    a, b, c = cell[..., 0, :], cell[..., 1, :], cell[..., 2, :]
    bc, ca, ab = torch.linalg.cross(b, c), torch.linalg.cross(c, a), torch.linalg.cross(a, b)
    det = (a * bc).sum(dim=-1)
    return torch.stack((bc, ca, ab), dim=-2) / det[..., None, None], det


class EwaldCalculator(Calculator):
    r"""Ewald summation: real-space pair sum + explicit reciprocal-space sum over all k-vectors with wavelength
    ``>= lr_wavelength`` (reference ``calculators/ewald.py:8-142``).  O(N K); meant for small cells.

    The (K, N) phase sums run in ``csrc/ewald.hip``; k-vector generation (``lib/kvectors.py:24-74,105-136``: integer
    frequencies times the reciprocal cell), the 1/V factor and the self / background / slab terms are a handful of small
    tensor ops here, so that the cell gradient is autograd's.  ``kvectors`` may be supplied by the caller; ``node_mask``
    masks atoms of the result.  Padded batches go through ``torch.vmap(calculator.forward)`` as in the reference
    (``tests/calculators/test_padding.py``): one launch per kernel for the whole batch (``_forward_batched``).

    :param potential: potential with a positive ``smearing``
    :param lr_wavelength: spatial resolution of the reciprocal-space part
    :param full_neighbor_list: see :class:`Calculator`
    """

    def __init__(self, potential: Potential, lr_wavelength: float, full_neighbor_list: bool = False):
        super().__init__(potential=potential, full_neighbor_list=full_neighbor_list)
        if potential.smearing is None:
            raise ValueError("Must specify range radius to use a potential with EwaldCalculator")
        if potential.smearing <= 0:
            raise ValueError(f"`smearing` is {potential.smearing} but must be positive")
        if lr_wavelength <= 0:
            raise ValueError(f"`lr_wavelength` is {lr_wavelength} but must be positive")
        self.lr_wavelength: float = lr_wavelength
        self._freq_cache = None  # (weakref(cell), version, device) -> integer frequency table (K, 3)
        self._spec()

    def _frequencies(self, cell: torch.Tensor) -> torch.Tensor:
        """Integer frequencies (K,3) of all k-vectors: ``fftfreq(ns_d) * ns_d`` per axis, ``ns_d = ceil(|a_d| /
        lr_wavelength)``, x-major order, the zero vector first.  The mesh size needs the cell on the host (as in the
        reference, ``ewald.py:88-93``); cached per cell tensor."""
        c = self._freq_cache
        if c is not None and c[0]() is cell and c[1] == cell._version and c[2] == cell.device and c[3] == self.lr_wavelength:
            return c[4]
        cell_host = cell.detach().to("cpu", torch.float64).numpy()
        det = float(np.linalg.det(cell_host))
        if det == 0.0 or not np.isfinite(det):  # the cross-product inverse below would quietly produce inf / NaN k-vectors
            raise ValueError(f"provided `cell` has a determinant of {det}, i.e. it is not a valid unit cell")
        norms = np.linalg.norm(cell_host, axis=1)
        ns = np.ceil(norms / self.lr_wavelength).astype(np.int64)
        f = [np.fft.fftfreq(int(n)) * int(n) for n in ns]
        F = np.stack(np.meshgrid(*f, indexing="ij"), axis=-1).reshape(-1, 3)
        freq = torch.tensor(F, dtype=cell.dtype, device=cell.device)
        self._freq_cache = (weakref.ref(cell), cell._version, cell.device, self.lr_wavelength, freq)
        return freq

    def _forward_batched(self, batch_size, in_dims, charges, cell, positions, neighbor_indices, neighbor_distances,
                         periodic=None, node_mask=None, pair_mask=None, kvectors=None):
        """``torch.vmap(self.forward)`` over a zero-padded batch (reference ``tests/calculators/test_padding.py:73-98``) as ONE
        evaluation: the pair sum over the flattened pair list of all structures (atom indices offset by ``b N``; one pass of
        the atomic pair kernel -- the list is new every call, a transposed list would not pay), the reciprocal-space sums with
        ``blockIdx.y`` = structure (``ops._EwaldKSpace`` on (B,N,..) / (B,K,3) tensors), the self / background / slab terms as
        broadcast tensor expressions.  Returns ``None`` (the caller then loops over the samples) unless every per-structure
        tensor is batched along dimension 0 and the k-vectors are given, which is the reference's calling convention."""
        d = dict(zip(("charges", "cell", "positions", "neighbor_indices", "neighbor_distances", "periodic", "node_mask",
                      "pair_mask", "kvectors"), in_dims))
        must = ("charges", "cell", "positions", "neighbor_indices", "neighbor_distances", "kvectors")
        if kvectors is None or any(d[k] != 0 for k in must):
            return None
        if (node_mask is not None and d["node_mask"] != 0) or (pair_mask is not None and d["pair_mask"] != 0):
            return None
        B = batch_size
        per0 = None if periodic is None else (periodic[0] if d["periodic"] == 0 else periodic)
        _validate_parameters(
            charges=charges[0], cell=cell[0], positions=positions[0], neighbor_indices=neighbor_indices[0],
            neighbor_distances=neighbor_distances[0], periodic=per0, pair_mask=None if pair_mask is None else pair_mask[0],
            node_mask=None if node_mask is None else node_mask[0], kvectors=kvectors[0],
        )
        _lib.require_device(positions, "positions")
        N, Cn = charges.shape[1:]
        P = neighbor_indices.shape[1]
        pot_desc = self.potential._descriptor()
        # ---- real space: all structures as one list
        offsets = (torch.arange(B, device=positions.device, dtype=neighbor_indices.dtype) * N).view(B, 1, 1)
        sr = ops.pme_potential(
            charges.reshape(B * N, Cn), cell[0], positions.reshape(B * N, 3), (neighbor_indices + offsets).reshape(B * P, 2),
            neighbor_distances.reshape(B * P), None if pair_mask is None else pair_mask.reshape(B * P), None, None, pot_desc,
            bool(self.full_neighbor_list), None, None, atomic_pairs=True,
        ).reshape(B, N, Cn)
        # ---- reciprocal space
        volume = torch.abs(_reciprocal_and_det(cell)[1]).view(B, 1, 1)
        lr = ops.ewald_kspace(charges, positions, kvectors, pot_desc) / volume
        p = 1 if pot_desc.kind == _lib.COULOMB else pot_desc.exponent
        two_s2 = 2.0 * pot_desc.smearing**2
        lr = lr - charges * (pot_desc.prefactor / math.gamma(0.5 * p + 1.0) / two_s2 ** (0.5 * p))
        if p < 3:
            bg = pot_desc.prefactor * math.pi**1.5 * two_s2 ** (0.5 * (3 - p)) / ((3 - p) * math.gamma(0.5 * p))
            lr = lr - (2.0 * bg) * charges.sum(dim=1, keepdim=True) / volume
        if periodic is not None and p == 1:
            flags = periodic.tolist() if d["periodic"] == 0 else [periodic.tolist()] * B
            axes = [ops._slab_axis(f) for f in flags]
            if any(a is not None for a in axes):  # 2-D periodic slabs among the structures, potentials/coulomb.py:6-40
                ax = torch.tensor([0 if a is None else a for a in axes], device=positions.device)
                on = torch.tensor([a is not None for a in axes], device=positions.device, dtype=positions.dtype).view(B, 1, 1)
                z = positions.gather(2, ax.view(B, 1, 1).expand(B, N, 1))
                Lz = torch.linalg.norm(cell.gather(1, ax.view(B, 1, 1).expand(B, 1, 3)), dim=2, keepdim=True)
                Q, M = charges.sum(dim=1, keepdim=True), (charges * z).sum(dim=1, keepdim=True)
                M2 = (charges * z * z).sum(dim=1, keepdim=True)
                lr = lr + on * pot_desc.prefactor * (4 * math.pi / volume) * (z * M - 0.5 * (M2 + Q * z * z) - Q / 12.0 * Lz * Lz)
        if node_mask is not None:
            lr = lr * node_mask.unsqueeze(-1)
        return sr + lr / 2

    def _forward_impl(self, charges, cell, positions, neighbor_indices, neighbor_distances, periodic=None, node_mask=None,
                      pair_mask=None, kvectors=None):
        _validate_parameters(
            charges=charges, cell=cell, positions=positions, neighbor_indices=neighbor_indices,
            neighbor_distances=neighbor_distances, periodic=periodic, pair_mask=pair_mask, node_mask=node_mask,
            kvectors=kvectors,
        )
        _lib.require_device(positions, "positions")
        if charges.dim() != 2:
            raise NotImplementedError("EwaldCalculator: batched (padded) inputs are not supported by this build")
        pot_desc = self.potential._descriptor()
        sr = ops.pme_potential(charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask, None, None,
                               pot_desc, bool(self.full_neighbor_list), None)
        recip, det = _reciprocal_and_det(cell)
        if kvectors is None:
            # k = 2 pi F A^-T (kvectors.py:47-74); differentiable w.r.t. the cell
            kvectors = (2 * math.pi) * self._frequencies(cell) @ recip
        volume = torch.abs(det)
        lr = ops.ewald_kspace(charges, positions, kvectors, pot_desc) / volume
        p = 1 if pot_desc.kind == _lib.COULOMB else pot_desc.exponent
        two_s2 = 2.0 * pot_desc.smearing**2
        self_c = pot_desc.prefactor / math.gamma(0.5 * p + 1.0) / two_s2 ** (0.5 * p)
        lr = lr - charges * self_c
        if p < 3:
            bg = pot_desc.prefactor * math.pi**1.5 * two_s2 ** (0.5 * (3 - p)) / ((3 - p) * math.gamma(0.5 * p))
            lr = lr - (2.0 * bg) * charges.sum(dim=0) / volume
        if periodic is not None and p == 1:
            axis = ops._slab_axis(periodic.tolist())
            if axis is not None:  # 2-D periodic slab, potentials/coulomb.py:6-40
                z = positions[:, axis : axis + 1]
                Lz = torch.linalg.norm(cell[axis])
                Q, M, M2 = charges.sum(dim=0), (charges * z).sum(dim=0), (charges * z * z).sum(dim=0)
                lr = lr + pot_desc.prefactor * (4 * math.pi / volume) * (z * M - 0.5 * (M2 + Q * z * z) - Q / 12.0 * Lz * Lz)
        if node_mask is not None:
            lr = lr * node_mask.unsqueeze(-1)
        return sr + lr / 2

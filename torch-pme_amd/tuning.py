"""Parameter tuning for the mesh calculators: a-priori error estimates + a grid search that
times the candidates ON THE GPU through the real calculators.

Counterpart of the reference's ``tuning/p3m.py:69-323`` (``tune_p3m``, ``P3MErrorBounds``), ``tuning/pme.py:12-270``
(``tune_pme``, ``PMEErrorBounds``) and ``tuning/tuner.py:169-373`` (``TunerBase``, ``GridSearchTuner``,
``TuningTimings``): same call signatures, return values, exception texts and the same published estimates --

* real space (Kolafa & Perram):  ``2 Q2 / sqrt(N) / sqrt(rc V) * exp(-rc^2 / (2 sigma^2))``
* P3M reciprocal space (Deserno & Holm, J. Chem. Phys. 109, 7694 (1998), eq. 38 with the coefficients of Table II)
* PME reciprocal space (the Lagrange-interpolation estimate of the reference with its tabulated RMS factors)

The estimates are host arithmetic (Python floats); only the timing touches the device.  Unlike the reference's timer
(``time.monotonic`` without device synchronisation, ``tuner.py:337-373``) candidates are timed with HIP events on the
launch stream and the MEDIAN of the repeats is returned.
"""

from __future__ import annotations

import math
from fractions import Fraction
from itertools import product
from typing import Any
from warnings import warn

import torch

from ._utils import _validate_parameters
from .calculators import EwaldCalculator, P3MCalculator, PMECalculator
from .potentials import CoulombPotential

# a_m^(P) of Deserno & Holm, Table II: P = charge assignment order (interpolation nodes), m = 0 .. P-1
_DH_TABLE_II = {
    1: ("2/3",),
    2: ("1/50", "5/294"),
    3: ("1/588", "7/1440", "21/3872"),
    4: ("1/4320", "3/1936", "7601/2271360", "143/28800"),
    5: ("1/23232", "7601/13628160", "143/69120", "517231/106536960", "106640677/11737571328"),
    6: ("691/68140800", "13/57600", "47021/35512320", "9694607/2095994880", "733191589/59609088000",
        "326190917/11700633600"),
    7: ("1/345600", "3617/35512320", "745739/838397952", "56399353/12773376000", "25091609/1560084480",
        "1755948832039/36229939200000", "4887769399/37838389248"),
}
_DH_COEF = {order: tuple(float(Fraction(c)) for c in row) for order, row in _DH_TABLE_II.items()}

# RMS of the Lagrange interpolation error polynomial, indexed by the number of nodes (reference tuning/pme.py:207)
_LAGRANGE_RMS = {3: 0.246, 4: 0.404, 5: 0.950, 6: 2.51, 7: 8.42}


class TuningErrorBounds(torch.nn.Module):
    """Base class of the error estimates: holds the structure, ``forward`` = :meth:`error`."""

    def __init__(self, charges: torch.Tensor, cell: torch.Tensor, positions: torch.Tensor):
        super().__init__()
        self._charges, self._cell, self._positions = charges, cell, positions
        cell_host = cell.detach().to("cpu", torch.float64)
        self.volume = float(torch.abs(torch.det(cell_host)))
        self.sum_squared_charges = float((charges.detach().to(torch.float64) ** 2).sum())
        self.prefac = 2.0 * self.sum_squared_charges / math.sqrt(len(positions))
        self.cell_dimensions = [float(v) for v in torch.linalg.norm(cell_host, dim=1)]

    def forward(self, *args, **kwargs):
        return self.error(*args, **kwargs)

    def error(self, *args, **kwargs):
        raise NotImplementedError

    # shared pieces ---------------------------------------------------------------------------------------------
    def _effective_spacing(self, mesh_spacing: float) -> float:
        """Geometric mean of the per-axis spacings ``L_d / (2 L_d / h + 1)`` the estimate assumes."""
        h = 1.0
        for L in self.cell_dimensions:
            h *= L / (2.0 * L / mesh_spacing + 1.0)
        return h ** (1.0 / 3.0)

    def err_rspace(self, smearing, cutoff):
        """Real-space truncation error of the pair sum."""
        smearing, cutoff = float(smearing), float(cutoff)
        return self.prefac / math.sqrt(cutoff * self.volume) * math.exp(-(cutoff**2) / (2.0 * smearing**2))

    def _combine(self, smearing, mesh_spacing, cutoff, interpolation_nodes) -> torch.Tensor:
        k = self.err_kspace(smearing, mesh_spacing, interpolation_nodes)
        r = self.err_rspace(smearing, cutoff)
        return torch.tensor(math.sqrt(k * k + r * r), dtype=self._positions.dtype)


class P3MErrorBounds(TuningErrorBounds):
    """A-priori RMS force error of :class:`P3MCalculator` (reference ``tuning/p3m.py:176-323``)."""

    def err_kspace(self, smearing, mesh_spacing, interpolation_nodes):
        smearing, mesh_spacing, order = float(smearing), float(mesh_spacing), int(interpolation_nodes)
        alpha = 1.0 / (math.sqrt(2.0) * smearing)
        ha = self._effective_spacing(mesh_spacing) * alpha
        series = sum(a * ha ** (2 * m) for m, a in enumerate(_DH_COEF[order]))
        return (self.prefac / self.volume ** (2.0 / 3.0) * ha**order
                * math.sqrt(alpha * self.volume ** (1.0 / 3.0) * math.sqrt(2.0 * math.pi) * series))

    def error(self, smearing: float, mesh_spacing: float, cutoff: float, interpolation_nodes: int) -> torch.Tensor:
        return self._combine(smearing, mesh_spacing, cutoff, interpolation_nodes)


class PMEErrorBounds(TuningErrorBounds):
    """A-priori RMS force error of :class:`PMECalculator` (reference ``tuning/pme.py:141-270``)."""

    def err_kspace(self, smearing, mesh_spacing, interpolation_nodes):
        smearing, mesh_spacing, n = float(smearing), float(mesh_spacing), int(interpolation_nodes)
        h = self._effective_spacing(mesh_spacing)
        alpha = 1.0 / (math.sqrt(2.0) * smearing)
        return (self.prefac * math.pi**0.25 * math.sqrt(6.0 * alpha / (2 * n + 1)) / self.volume ** (2.0 / 3.0)
                * (math.sqrt(2.0) / smearing * h) ** n / math.factorial(n)
                * math.exp(n * (math.log(n / 2.0) - 1.0) / 2.0) * _LAGRANGE_RMS[n])

    def error(self, cutoff: float, smearing: float, mesh_spacing: float, interpolation_nodes: float) -> torch.Tensor:
        return self._combine(smearing, mesh_spacing, cutoff, interpolation_nodes)


class EwaldErrorBounds(TuningErrorBounds):
    """A-priori RMS force error of :class:`EwaldCalculator` (reference ``tuning/ewald.py:126-211``)."""

    def err_kspace(self, smearing, lr_wavelength):
        smearing, lr_wavelength = float(smearing), float(lr_wavelength)
        return (math.sqrt(self.prefac) / smearing / math.pi / math.sqrt(self.volume / lr_wavelength)
                * math.exp(-2.0 * (math.pi * smearing / lr_wavelength) ** 2))

    def error(self, smearing: float, lr_wavelength: float, cutoff: float) -> torch.Tensor:
        k, r = self.err_kspace(smearing, lr_wavelength), self.err_rspace(smearing, cutoff)
        return torch.tensor(math.sqrt(k * k + r * r), dtype=self._positions.dtype)


class TunerBase:
    """Holds the structure and estimates ``smearing`` from the real-space estimate (reference ``tuner.py:48-166``)."""

    def __init__(self, charges, cell, positions, cutoff: float, calculator, exponent: int = 1,
                 full_neighbor_list: bool = False, prefactor: float = 1.0):
        if exponent != 1:
            raise NotImplementedError(f"Only exponent = 1 is supported but got {exponent}.")
        _validate_parameters(
            charges=charges, cell=cell, positions=positions,
            neighbor_indices=torch.tensor([[0, 1]], device=positions.device),
            neighbor_distances=torch.tensor([1.0], device=positions.device, dtype=positions.dtype),
        )
        self.charges, self.cell, self.positions = charges, cell, positions
        self.cutoff, self.calculator, self.exponent = cutoff, calculator, exponent
        self.full_neighbor_list, self.prefactor = full_neighbor_list, prefactor
        self._smearing_esti_prefac = 2.0 * float((charges**2).sum()) / math.sqrt(len(positions))

    def tune(self, accuracy: float = 1e-3):
        raise NotImplementedError

    def estimate_smearing(self, accuracy: float) -> float:
        """The smearing that puts the real-space error estimate at ``cutoff`` to ``accuracy / 2`` (closed form)."""
        if not isinstance(accuracy, float):
            raise ValueError(f"'{accuracy}' is not a float.")
        volume = float(torch.abs(torch.det(self.cell.detach().to("cpu", torch.float64))))
        arg = accuracy / 2.0 / self._smearing_esti_prefac * math.sqrt(self.cutoff * volume)
        return float(self.cutoff / math.sqrt(-2.0 * math.log(arg)))

    @staticmethod
    def filter_neighbors(cutoff: float, neighbor_indices: torch.Tensor, neighbor_distances: torch.Tensor):
        """Keep the pairs with ``d < cutoff`` (a list built for a larger cutoff can be reused)."""
        keep = torch.where(neighbor_distances < cutoff)
        return neighbor_indices[keep], neighbor_distances[keep]


class TuningTimings(torch.nn.Module):
    """Times ``calculator.forward`` (+ ``sum().backward()``) on one structure: ``n_warmup`` untimed calls, then the MEDIAN
    of ``n_repeat`` calls, each bracketed by HIP events on the current stream (seconds).  Protocol of the reference's
    ``TuningTimings.forward`` (``tuner.py:337-373``): fresh clones with ``requires_grad`` on positions, cell, charges."""

    def __init__(self, charges, cell, positions, neighbor_indices, neighbor_distances, n_repeat: int = 4,
                 n_warmup: int = 4, run_backward: bool | None = True):
        super().__init__()
        _validate_parameters(charges=charges, cell=cell, positions=positions, neighbor_indices=neighbor_indices,
                             neighbor_distances=neighbor_distances)
        self.charges, self.cell, self.positions = charges, cell, positions
        self.neighbor_indices, self.neighbor_distances = neighbor_indices, neighbor_distances
        self.n_repeat, self.n_warmup, self.run_backward = n_repeat, n_warmup, run_backward

    def forward(self, calculator: torch.nn.Module) -> float:
        times = []
        for it in range(self.n_repeat + self.n_warmup):
            positions, cell, charges = self.positions.clone(), self.cell.clone(), self.charges.clone()
            if self.run_backward:
                for t in (positions, cell, charges):
                    t.requires_grad_(True)
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            result = calculator.forward(positions=positions, charges=charges, cell=cell,
                                        neighbor_indices=self.neighbor_indices,
                                        neighbor_distances=self.neighbor_distances)
            value = result.sum()
            if self.run_backward:
                value.backward(retain_graph=True)
            stop.record()
            stop.synchronize()
            check = getattr(calculator, "check", None)
            if check is not None:
                check()  # the deferred NaN guard would blame the NEXT candidate of a search: look at the flag now (synchronised)
            if it >= self.n_warmup:
                times.append(start.elapsed_time(stop) * 1e-3)
        times.sort()
        mid = len(times) // 2
        return times[mid] if len(times) % 2 else 0.5 * (times[mid - 1] + times[mid])


class GridSearchTuner(TunerBase):
    """Estimates the error of every parameter set and times those that meet the accuracy (``tuner.py:169-291``)."""

    def __init__(self, charges, cell, positions, cutoff: float, calculator, error_bounds: TuningErrorBounds,
                 params: list[dict], neighbor_indices, neighbor_distances, full_neighbor_list: bool = False,
                 prefactor: float = 1.0, exponent: int = 1):
        super().__init__(charges=charges, cell=cell, positions=positions, cutoff=cutoff, calculator=calculator,
                         exponent=exponent, full_neighbor_list=full_neighbor_list, prefactor=prefactor)
        self.error_bounds, self.params = error_bounds, params
        neighbor_indices, neighbor_distances = self.filter_neighbors(cutoff, neighbor_indices, neighbor_distances)
        self.time_func = TuningTimings(charges, cell, positions, neighbor_indices, neighbor_distances, run_backward=True)

    def tune(self, accuracy: float = 1e-3) -> tuple[list[float], list[float]]:
        if not isinstance(accuracy, float):
            raise ValueError(f"'{accuracy}' is not a float.")
        smearing = self.estimate_smearing(accuracy)
        errors, timings = [], []
        for param in self.params:
            err = float(self.error_bounds(smearing=smearing, cutoff=self.cutoff, **param))
            errors.append(err)
            timings.append(self._timing(smearing, param) if err <= accuracy else float("inf"))
        return errors, timings

    def _timing(self, smearing: float, k_space_params: dict) -> float:
        calculator = self.calculator(potential=CoulombPotential(smearing=smearing, prefactor=self.prefactor),
                                     full_neighbor_list=self.full_neighbor_list, **k_space_params)
        calculator.to(device=self.positions.device, dtype=self.positions.dtype)
        return self.time_func(calculator)


def _validated_min_dimension(charges, cell, positions, exponent) -> float:
    # validation first (the reference's tuners validate in TunerBase.__init__, before touching the cell)
    if exponent != 1:
        raise NotImplementedError(f"Only exponent = 1 is supported but got {exponent}.")
    _validate_parameters(
        charges=charges, cell=cell, positions=positions,
        neighbor_indices=torch.tensor([[0, 1]], device=positions.device),
        neighbor_distances=torch.tensor([1.0], device=positions.device, dtype=positions.dtype),
    )
    return float(torch.min(torch.linalg.norm(cell, dim=1)))


def _tune_mesh(calculator, bounds_cls, charges, cell, positions, cutoff, neighbor_indices, neighbor_distances,
               full_neighbor_list, prefactor, exponent, nodes_lo, nodes_hi, mesh_lo, mesh_hi, accuracy):
    min_dimension = _validated_min_dimension(charges, cell, positions, exponent)
    params = [
        {"interpolation_nodes": nodes, "mesh_spacing": 2 * min_dimension / (2**ns - 1)}
        for nodes, ns in product(range(nodes_lo, nodes_hi + 1), range(mesh_lo, mesh_hi + 1))
    ]
    return _grid_search(calculator, bounds_cls, params, charges, cell, positions, cutoff, neighbor_indices,
                        neighbor_distances, full_neighbor_list, prefactor, exponent, accuracy)


def _grid_search(calculator, bounds_cls, params, charges, cell, positions, cutoff, neighbor_indices, neighbor_distances,
                 full_neighbor_list, prefactor, exponent, accuracy):
    tuner = GridSearchTuner(
        charges=charges, cell=cell, positions=positions, cutoff=cutoff, exponent=exponent,
        neighbor_indices=neighbor_indices, neighbor_distances=neighbor_distances,
        full_neighbor_list=full_neighbor_list, prefactor=prefactor, calculator=calculator,
        error_bounds=bounds_cls(charges=charges, cell=cell, positions=positions), params=params,
    )
    smearing = tuner.estimate_smearing(accuracy)
    errs, timings = tuner.tune(accuracy)
    if any(err < accuracy for err in errs):
        best = timings.index(min(timings))
        return smearing, params[best], timings[best]
    warn(
        f"No parameter meets the accuracy requirement.\n"
        f"Returning the parameter with the smallest error, which is {min(errs)}.\n",
        stacklevel=2,
    )
    best = errs.index(min(errs))
    return smearing, params[best], timings[best]


def tune_p3m(charges, cell, positions, cutoff: float, neighbor_indices, neighbor_distances,
             full_neighbor_list: bool = False, prefactor: float = 1.0, exponent: int = 1, nodes_lo: int = 2,
             nodes_hi: int = 5, mesh_lo: int = 2, mesh_hi: int = 7, accuracy: float = 1e-3
             ) -> tuple[float, dict[str, Any], float]:
    """Fastest ``(interpolation_nodes, mesh_spacing)`` of :class:`P3MCalculator` whose estimated error is below
    ``accuracy``, for the smearing that puts the real-space error at ``cutoff`` to ``accuracy / 2``.
    Returns ``(smearing, {"interpolation_nodes", "mesh_spacing"}, seconds)`` (reference ``tuning/p3m.py:69-173``)."""
    return _tune_mesh(P3MCalculator, P3MErrorBounds, charges, cell, positions, cutoff, neighbor_indices,
                      neighbor_distances, full_neighbor_list, prefactor, exponent, nodes_lo, nodes_hi, mesh_lo, mesh_hi,
                      accuracy)


def tune_pme(charges, cell, positions, cutoff: float, neighbor_indices, neighbor_distances,
             full_neighbor_list: bool = False, prefactor: float = 1.0, exponent: int = 1, nodes_lo: int = 3,
             nodes_hi: int = 7, mesh_lo: int = 2, mesh_hi: int = 7, accuracy: float = 1e-3
             ) -> tuple[float, dict[str, Any], float]:
    """As :func:`tune_p3m` for :class:`PMECalculator` (Lagrange interpolation, nodes 3..7; ``tuning/pme.py:12-138``)."""
    return _tune_mesh(PMECalculator, PMEErrorBounds, charges, cell, positions, cutoff, neighbor_indices,
                      neighbor_distances, full_neighbor_list, prefactor, exponent, nodes_lo, nodes_hi, mesh_lo, mesh_hi,
                      accuracy)


def tune_ewald(charges, cell, positions, cutoff: float, neighbor_indices, neighbor_distances,
               full_neighbor_list: bool = False, prefactor: float = 1.0, exponent: int = 1, ns_lo: int = 1,
               ns_hi: int = 14, accuracy: float = 1e-3) -> tuple[float, dict[str, Any], float]:
    """Fastest ``lr_wavelength = min|a_d| / ns`` (``ns_lo <= ns <= ns_hi``) of :class:`EwaldCalculator` whose estimated
    error is below ``accuracy``; returns ``(smearing, {"lr_wavelength"}, seconds)`` (reference ``tuning/ewald.py:11-123``)."""
    min_dimension = _validated_min_dimension(charges, cell, positions, exponent)
    params = [{"lr_wavelength": min_dimension / ns} for ns in range(ns_lo, ns_hi + 1)]
    return _grid_search(EwaldCalculator, EwaldErrorBounds, params, charges, cell, positions, cutoff, neighbor_indices,
                        neighbor_distances, full_neighbor_list, prefactor, exponent, accuracy)

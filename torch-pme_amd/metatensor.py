"""metatensor / metatomic front end of the calculators (second half; reference
``metatensor/calculator.py:22-188`` and the three one-line subclasses ``metatensor/{ewald,pme,p3m}.py``).

``calculator(system, neighbors) -> TensorMap``: ``system`` is a ``metatomic.torch.System`` carrying a ``"charge"`` data block,
``neighbors`` the ``TensorBlock`` of a neighbour list (samples ``first_atom, second_atom, cell_shift_a/b/c``, one ``xyz``
component, one ``distance`` property: the pair vectors).  The result is a single-block ``TensorMap`` with samples
``(system, atom)`` and one property per charge channel, as in the reference; the same checks raise the same messages.

The two packages are imported when this module is (``ImportError`` with the reference's hints otherwise).  They are only
used through a handful of attributes (``Labels(names, values)``, ``.names``, ``.column``, ``TensorBlock(values, samples,
components, properties)``, ``TensorMap(keys, blocks)``, ``System.positions / cell / known_data / get_data``), so the tests
run against small stand-ins registered under the same module names.

MI355X-specific option ``fuse_distances=True``: the pair vectors of a metatomic neighbour list ARE ``r_j - r_i + S cell`` of
the system, so the distances can be formed by :func:`torchpme_amd.pair_distances` from the system's own positions, cell and
the integer cell shifts in the samples -- the calculator then runs its fused distance + pair kernels and the gradient reaches
``system.positions`` / ``system.cell`` without passing through the (P,3) vector block.  Off by default (reference
semantics: ``|neighbors.values|``, differentiable w.r.t. those values).
"""

from __future__ import annotations

import torch

try:
    from metatensor.torch import Labels, TensorBlock, TensorMap
except ImportError:
    raise ImportError(
        "metatensor.torch is required for torchpme.metatensor but is not installed. "
        "Try installing it with:\npip install metatensor[torch]"
    ) from None

try:
    from metatomic.torch import System  # noqa: F401  (the type callers pass; duck-typed here)
except ImportError:
    raise ImportError(
        "metatomic is required for torchpme.metatensor but is not installed. Try installing it with:\npip install metatomic"
    ) from None

from . import calculators as _calculators
from .ops import pair_distances

_NEIGHBOR_SAMPLES = ("first_atom", "second_atom", "cell_shift_a", "cell_shift_b", "cell_shift_c")


class Calculator(torch.nn.Module):
    """Thin wrapper of :class:`torchpme_amd.Calculator`; subclasses set ``_base_calculator``."""

    _base_calculator = _calculators.Calculator

    def __init__(self, *args, fuse_distances: bool = False, **kwargs):
        super().__init__()
        self._calculator = self._base_calculator(*args, **kwargs)
        self.fuse_distances = bool(fuse_distances)

    @property
    def double_backward(self):
        """``None`` | ``"auto"`` | ``"analytic"`` | ``"finite-difference"``: see :attr:`torchpme_amd.Calculator.double_backward`
        (second derivatives through the calculator: training on forces with learned charges)."""
        return self._calculator.double_backward

    @double_backward.setter
    def double_backward(self, mode):
        self._calculator.double_backward = mode

    @staticmethod
    def _validate_compute_parameters(system, neighbors) -> None:
        values = neighbors.values
        dtype, device = system.positions.dtype, system.positions.device
        for what, got, want in (("dtype", values.dtype, dtype), ("device", values.device, device)):
            if got != want:
                raise ValueError(f"{what} of `neighbors` ({got}) must be the same as `system` ({want})")
        if tuple(neighbors.samples.names) != _NEIGHBOR_SAMPLES:
            raise ValueError(
                "Invalid samples for `neighbors`: the sample names must be "
                "'first_atom', 'second_atom', 'cell_shift_a', 'cell_shift_b', 'cell_shift_c'"
            )
        xyz = Labels(["xyz"], torch.arange(3, dtype=torch.int32, device=device).unsqueeze(1))
        if len(neighbors.components) != 1 or neighbors.components[0] != xyz:
            raise ValueError("Invalid components for `neighbors`: there should be a single 'xyz'=[0, 1, 2] component")
        if neighbors.properties != Labels(["distance"], torch.zeros(1, 1, dtype=torch.int32, device=device)):
            raise ValueError("Invalid properties for `neighbors`: there should be a single 'distance'=0 property")
        if "charge" not in system.known_data():
            raise ValueError("`system` does not contain `charge` data")
        charge = system.get_data("charge")
        if len(charge) != 1:
            raise ValueError(f"Charge tensor have exactlty one block but has {len(charge)} blocks")
        n_components = len(charge.block().components)
        if n_components > 0:
            raise ValueError(f"TensorBlock containg the charges should not have components; found {n_components}")

    def forward(self, system, neighbors):
        """Potential of every atom and charge channel as a ``TensorMap`` (reference docstring: ``calculator.py:109-142``)."""
        self._validate_compute_parameters(system, neighbors)
        device = system.positions.device
        charges = system.get_data("charge").block().values
        first, second = neighbors.samples.column("first_atom"), neighbors.samples.column("second_atom")
        neighbor_indices = torch.stack([first, second], dim=1)
        if self.fuse_distances:
            shifts = torch.stack([neighbors.samples.column(n) for n in _NEIGHBOR_SAMPLES[2:]], dim=1)
            neighbor_distances = pair_distances(system.positions, neighbor_indices, system.cell, shifts)
        else:
            neighbor_distances = torch.linalg.norm(neighbors.values, dim=1).squeeze(1)
        potential = self._calculator.forward(
            charges=charges, cell=system.cell, positions=system.positions, neighbor_indices=neighbor_indices,
            neighbor_distances=neighbor_distances,
        )
        n_atoms = len(system)
        samples = torch.zeros((n_atoms, 2), device=device, dtype=torch.int32)
        samples[:, 1] = torch.arange(n_atoms, device=device, dtype=torch.int32)
        channels = torch.arange(charges.shape[1], device=device, dtype=torch.int32).unsqueeze(1)
        block = TensorBlock(values=potential, samples=Labels(["system", "atom"], samples), components=[],
                            properties=Labels("charges_channel", channels))
        return TensorMap(keys=Labels("_", torch.zeros(1, 1, dtype=torch.int32, device=device)), blocks=[block])


class EwaldCalculator(Calculator):
    """:class:`torchpme_amd.EwaldCalculator` behind the metatensor interface (reference ``metatensor/ewald.py``)."""

    _base_calculator = _calculators.EwaldCalculator


class PMECalculator(Calculator):
    """:class:`torchpme_amd.PMECalculator` behind the metatensor interface (reference ``metatensor/pme.py``)."""

    _base_calculator = _calculators.PMECalculator


class P3MCalculator(Calculator):
    """:class:`torchpme_amd.P3MCalculator` behind the metatensor interface (reference ``metatensor/p3m.py``)."""

    _base_calculator = _calculators.P3MCalculator

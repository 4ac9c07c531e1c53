"""MI355X (gfx950) native PME / P3M long-range calculators -- drop-in for the hot path of
lab-cosmo/torch-pme: ``PMECalculator / P3MCalculator.forward(charges, cell, positions,
neighbor_indices, neighbor_distances)`` and its autograd.

The directory is named ``torch-pme_amd``; import it as ``torchpme_amd`` (see ``torchpme_amd.py`` at the
repository root).  All compute runs in ``libmipme.so`` (hand-written HIP, C-ABI in ``include/mipme.h``).
"""

from . import lib, library, ops, prefactors, tuning, workloads  # noqa: F401  (library registers the torch.ops.mipme ops)
from ._lib import LIB_PATH, MipmeError  # noqa: F401
from .calculators import Calculator, EwaldCalculator, P3MCalculator, PMECalculator
from .graphed import EnergyLog, GraphedEnergyForces, GraphedFrameBatch
from .neighbors import NeighborStream, neighbor_list, neighbor_list_device
from .ops import pair_distances, weighted_sum
from .potentials import CoulombPotential, InversePowerLawPotential, Potential
from .tuning import tune_ewald, tune_p3m, tune_pme

__version__ = "0.1.0"

__all__ = [
    "Calculator",
    "EwaldCalculator",
    "P3MCalculator",
    "PMECalculator",
    "CoulombPotential",
    "InversePowerLawPotential",
    "Potential",
    "pair_distances",
    "weighted_sum",
    "EnergyLog",
    "GraphedEnergyForces",
    "GraphedFrameBatch",
    "neighbor_list",
    "neighbor_list_device",
    "NeighborStream",
    "tune_ewald",
    "tune_p3m",
    "tune_pme",
]

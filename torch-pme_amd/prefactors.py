"""Electrostatic prefactors e^2/(4 pi eps0) in common unit systems (reference ``prefactors.py:1-11``),
derived here from the CODATA 2018 constants instead of being tabulated."""

import math

_E = 1.602176634e-19  # elementary charge / C (exact)
_EPS0 = 8.8541878128e-12  # vacuum permittivity / F m^-1
_NA = 6.02214076e23  # Avogadro constant / mol^-1 (exact)

#: Conversion factor from Gaussian units to SI units (J m)
SI = _E * _E / (4 * math.pi * _EPS0)

#: Conversion factor from Gaussian units to electron volts / Angstroms
eV_A = SI / _E * 1e10

#: Conversion factor from Gaussian units to kilocalories per mole / Angstroms
kcalmol_A = SI * _NA / 4184.0 * 1e10

#: Conversion factor from Gaussian units to kilojoules per mole / Angstroms
kJmol = SI * _NA / 1000.0 * 1e10

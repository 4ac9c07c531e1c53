"""Pair potentials 1/r^p with Gaussian range separation (reference ``potentials/potential.py``,
``potentials/coulomb.py``, ``potentials/inversepowerlaw.py``).

Same constructor signatures, buffer names (``smearing``, ``prefactor``, ``exponent`` as float64
buffers; ``exclusion_radius``, ``exclusion_degree`` attributes) and method names as the reference,
so user code and ``state_dict`` round-trips carry over.  On the hot path the calculators do NOT call
these Python methods: they hand :meth:`Potential._descriptor` to the fused HIP kernels
(``csrc/rspace.hip`` evaluates v_SR and its derivative per pair, ``csrc/kfilter.hip`` evaluates the
Fourier-space kernel per mesh point).  The elementwise methods below exist for inspection and plotting,
as small closed-form tensor expressions.
"""

from __future__ import annotations

import math

import torch

from . import _lib

_SQRT_PI = math.sqrt(math.pi)


def _reg_upper_gamma(p: int, x: torch.Tensor) -> torch.Tensor:
    """Q(p/2, x) for integer p: finite sums (integer p/2) or erfc plus a finite sum (half-integer)."""
    ex = torch.exp(-x)
    if p % 2 == 0:
        term = torch.ones_like(x)
        total = torch.ones_like(x)
        for k in range(1, p // 2):
            term = term * x / k
            total = total + term
        return ex * total
    sx = torch.sqrt(x)
    q = torch.erfc(sx)
    term = ex / (_SQRT_PI * sx.clamp(min=1e-300))
    for k in range(1, (p - 1) // 2 + 1):
        term = term * x / (k - 0.5)
        q = q + term
    return q


def _exp1(z: torch.Tensor) -> torch.Tensor:
    """E1(z) = Gamma(0, z) for z > 0 (what the reference gets from ``lib/math.py:5-83``): power series
    -gamma - ln z - sum_k (-z)^k / (k k!) below z = 1, modified-Lentz continued fraction 1/(z+1-1/(z+3-4/(z+5-...))) above;
    both to double-precision rounding, as plain tensor expressions (differentiable)."""
    zs = z.clamp(max=1.0)
    term = torch.ones_like(z)
    ssum = torch.zeros_like(z)
    for k in range(1, 26):
        term = term * (-zs) / k
        ssum = ssum - term / k
    series = -0.5772156649015329 - torch.log(zs) + ssum
    zl = z.clamp(min=1.0)
    b = zl + 1.0
    c = torch.full_like(z, 1e300)
    d = 1.0 / b
    h = d
    for i in range(1, 61):
        an = -float(i * i)
        b = b + 2.0
        d = 1.0 / (an * d + b)
        c = b + an / c
        h = h * (c * d)
    return torch.where(z < 1.0, series, h * torch.exp(-zl))


_INV_GAMMA_HALF = {p: 1.0 / math.gamma(0.5 * p + 1.0) for p in range(1, 7)}


def _reg_lower_gamma(p: int, x: torch.Tensor) -> torch.Tensor:
    """P(p/2, x) = 1 - Q without the cancellation at small x: power series x^a e^-x sum_k x^k / Gamma(a+k+1) below x = 1
    (the reference calls ``torch.special.gammainc``, ``inversepowerlaw.py:98-103``), 1 - Q above."""
    a = 0.5 * p
    xs = x.clamp(max=1.0)
    term = torch.full_like(x, _INV_GAMMA_HALF[p])
    total = term.clone()
    for k in range(1, 25):
        term = term * xs / (a + k)
        total = total + term
    series = xs**a * torch.exp(-xs) * total
    return torch.where(x < 1.0, series, 1 - _reg_upper_gamma(p, x))


class Potential(torch.nn.Module):
    """Base interface of a pair potential (reference ``potentials/potential.py:4-212``).

    :param smearing: length scale of the SR/LR split (``None``: no split, real-space only)
    :param exclusion_radius: radius inside which the potential is smoothly switched off
    :param exclusion_degree: exponent of the raised-cosine switch
    :param prefactor: multiplicative prefactor (see :mod:`prefactors`)
    """

    _kind = None  # set by subclasses the HIP kernels know

    def __init__(
        self,
        smearing: float | None = None,
        exclusion_radius: float | None = None,
        exclusion_degree: int = 1,
        prefactor: float = 1.0,
    ):
        super().__init__()
        if smearing is not None:
            self.register_buffer("smearing", torch.tensor(smearing, dtype=torch.float64))
        else:
            self.smearing = None
        self.exclusion_radius = exclusion_radius
        self.exclusion_degree = exclusion_degree
        self.register_buffer("prefactor", torch.tensor(prefactor, dtype=torch.float64))
        self._host_cache = None

    # ---- host-side view of the parameters (buffers may live on the device) -------------------
    def _host_params(self):
        """(smearing|None, prefactor, exponent) as Python floats; one D2H copy, then cached."""
        key = (
            None if self.smearing is None else (self.smearing.data_ptr(), self.smearing._version),
            (self.prefactor.data_ptr(), self.prefactor._version),
        )
        if self._host_cache is None or self._host_cache[0] != key:
            sm = None if self.smearing is None else float(self.smearing)
            self._host_cache = (key, sm, float(self.prefactor), self._exponent_int())
        return self._host_cache[1:]

    def _exponent_int(self) -> int:
        raise NotImplementedError

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_host_cache"] = None  # host copies of the buffers / the C descriptor: rebuilt on first use
        state.pop("_desc_cache", None)
        return state

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._host_cache = None
        return out

    def _descriptor(self) -> _lib.PotentialDesc:
        """The plain-C description consumed by libmipme (``mipme_potential_t``)."""
        if self._kind is None:
            raise TypeError(
                f"{self.__class__.__name__} has no HIP kernel: the MI355X-native calculators support "
                "CoulombPotential and InversePowerLawPotential"
            )
        sm, pref, p = self._host_params()
        key = (sm, pref, p, self.exclusion_radius, self.exclusion_degree)
        cached = self.__dict__.get("_desc_cache")
        if cached is not None and cached[0] == key:
            return cached[1]  # read-only for the library: one struct per parameter set instead of one per call
        desc = _lib.PotentialDesc(
            kind=self._kind,
            exponent=p,
            smearing=-1.0 if sm is None else sm,
            prefactor=pref,
            exclusion_radius=-1.0 if self.exclusion_radius is None else float(self.exclusion_radius),
            exclusion_degree=int(self.exclusion_degree),
        )
        self.__dict__["_desc_cache"] = (key, desc)
        return desc

    # ---- reference method surface ------------------------------------------------------------
    def f_cutoff(self, dist: torch.Tensor, pair_mask: torch.Tensor | None = None) -> torch.Tensor:
        """1 - ((1 - cos(pi r / r_excl)) / 2)^n inside the exclusion radius, 0 outside."""
        if self.exclusion_radius is None:
            raise ValueError("Cannot compute cutoff function when `exclusion_radius` is not set")
        rx = self.exclusion_radius
        inside = dist < rx
        base = 0.5 * (1 - torch.cos(torch.pi * dist / rx))
        out = torch.where(inside, 1 - base**self.exclusion_degree, torch.zeros_like(dist))
        return out if pair_mask is None else out * pair_mask

    def from_dist(self, dist, pair_mask=None):
        raise NotImplementedError(f"from_dist is not implemented for {self.__class__.__name__}")

    def lr_from_dist(self, dist, pair_mask=None):
        raise NotImplementedError(f"lr_from_dist is not implemented for {self.__class__.__name__}")

    def sr_from_dist(self, dist: torch.Tensor, pair_mask: torch.Tensor | None = None) -> torch.Tensor:
        """V_SR = V - V_LR, or -V_LR * f_cut when an exclusion radius is set."""
        if self.smearing is None:
            raise ValueError("Cannot compute range-separated potential when `smearing` is not specified.")
        if self.exclusion_radius is None:
            return self.from_dist(dist, pair_mask=pair_mask) - self.lr_from_dist(dist, pair_mask=pair_mask)
        return -self.lr_from_dist(dist, pair_mask=pair_mask) * self.f_cutoff(dist, pair_mask=pair_mask)

    def lr_from_k_sq(self, k_sq):
        raise NotImplementedError(f"lr_from_k_sq is not implemented for {self.__class__.__name__}")

    def kernel_from_k_sq(self, k_sq: torch.Tensor) -> torch.Tensor:
        return self.lr_from_k_sq(k_sq)

    def self_contribution(self):
        raise NotImplementedError(f"self_contribution is not implemented for {self.__class__.__name__}")

    def background_correction(self):
        raise NotImplementedError(f"background_correction is not implemented for {self.__class__.__name__}")

    def pbc_correction(self, periodic, positions, cell, charges):
        """Correction for systems that are not periodic in all three directions; zero unless a subclass has one."""
        return self.prefactor * torch.zeros_like(charges)


class _PowerLawPotential(Potential):
    """Shared closed forms for 1/r^p, integer p in 1..6."""

    _p = 1

    def _exponent_int(self) -> int:
        return self._p

    def from_dist(self, dist: torch.Tensor, pair_mask: torch.Tensor | None = None) -> torch.Tensor:
        out = dist.clamp(min=1e-15) ** (-self._p)
        if pair_mask is not None:
            out = out * pair_mask
        return self.prefactor * out

    def lr_from_dist(self, dist: torch.Tensor, pair_mask: torch.Tensor | None = None) -> torch.Tensor:
        if self.smearing is None:
            raise ValueError("Cannot compute long-range contribution without specifying `smearing`.")
        d = dist.clamp(min=1e-12)
        x = 0.5 * d * d / self.smearing**2
        out = _reg_lower_gamma(self._p, x) / d**self._p
        if pair_mask is not None:
            out = out * pair_mask
        return self.prefactor * out

    def lr_from_k_sq(self, k_sq: torch.Tensor) -> torch.Tensor:
        if self.smearing is None:
            raise ValueError("Cannot compute long-range kernel without specifying `smearing`.")
        p = self._p
        a = 0.5 * (3 - p)
        c0 = math.pi**1.5 / math.gamma(0.5 * p) * (2 * self.smearing**2) ** a
        zero = k_sq == 0
        z = 0.5 * self.smearing**2 * torch.where(zero, torch.ones_like(k_sq), k_sq)
        ez = torch.exp(-z)
        if p == 1:
            f = ez / z
        elif p == 2:
            f = torch.sqrt(torch.pi / z) * torch.erfc(torch.sqrt(z))
        elif p == 4:
            f = 2 * (ez - torch.sqrt(torch.pi * z) * torch.erfc(torch.sqrt(z)))
        elif p == 6:
            f = ((2 - 4 * z) * ez + 4 * torch.sqrt(torch.pi * z**3) * torch.erfc(torch.sqrt(z))) / 3
        elif p == 3:  # Gamma(0, z) = E1(z)
            f = _exp1(z)
        else:  # p == 5: Gamma(-1, z) / z^-1 = e^-z - z E1(z)
            f = ez - z * _exp1(z)
        k0 = -c0 / a if p > 3 else 0.0
        return self.prefactor * torch.where(zero, k0 * torch.ones_like(k_sq), c0 * f)

    def self_contribution(self) -> torch.Tensor:
        if self.smearing is None:
            raise ValueError("Cannot compute self contribution without specifying `smearing`.")
        ph = 0.5 * self._p
        return self.prefactor / math.gamma(ph + 1) / (2 * self.smearing**2) ** ph

    def background_correction(self) -> torch.Tensor:
        if self.smearing is None:
            raise ValueError("Cannot compute background correction without specifying `smearing`.")
        p = self._p
        if p >= 3:
            return torch.zeros_like(self.smearing)
        return self.prefactor * math.pi**1.5 * (2 * self.smearing**2) ** (0.5 * (3 - p)) / ((3 - p) * math.gamma(0.5 * p))


    def pbc_correction(self, periodic, positions, cell, charges):
        """2-D slab term (reference ``potentials/coulomb.py:6-40,160-167``), non-zero when exactly two of ``periodic`` are
        True: ``(4 pi / V) (z_i M - (M2 + Q z_i^2) / 2 - Q L_z^2 / 12)`` with z along the non-periodic axis, Q, M, M2 the
        zeroth / first / second moments of the charges along it and L_z the length of that cell vector.  (The calculators
        evaluate the same expression in ``mipme_slab_forward``; this method is the inspectable tensor form.)"""
        if self._p != 1:  # the slab term exists for 1/r only (`potentials/inversepowerlaw.py:166-169`)
            return self.prefactor * torch.zeros_like(charges)
        if periodic is None:  # fully periodic: no term (and no device round trip to find that out)
            return self.prefactor * torch.zeros_like(charges)
        flags = [bool(v) for v in periodic.tolist()]
        if sum(flags) != 2:
            return self.prefactor * torch.zeros_like(charges)
        axis = flags.index(False)
        z = positions[:, axis : axis + 1]
        volume = torch.abs(torch.det(cell))
        Lz = torch.linalg.norm(cell[axis])
        Q, M, M2 = charges.sum(dim=0), (charges * z).sum(dim=0), (charges * z * z).sum(dim=0)
        return self.prefactor * (4 * math.pi / volume) * (z * M - 0.5 * (M2 + Q * z * z) - Q / 12.0 * Lz * Lz)


class InversePowerLawPotential(_PowerLawPotential):
    """1/r^p potential, p in 1..6 (reference ``potentials/inversepowerlaw.py:9-173``)."""

    _kind = _lib.INVERSE_POWER_LAW

    def __init__(
        self,
        exponent: int,
        smearing: float | None = None,
        exclusion_radius: float | None = None,
        exclusion_degree: int = 1,
        prefactor: float = 1.0,
    ):
        super().__init__(smearing, exclusion_radius, exclusion_degree, prefactor)
        if exponent not in (1, 2, 3, 4, 5, 6):
            raise ValueError(f"Unsupported exponent: {exponent}")
        self.register_buffer("exponent", torch.tensor(exponent, dtype=torch.float64))
        self._p = int(exponent)


class CoulombPotential(_PowerLawPotential):
    """Smoothed electrostatic 1/r potential (reference ``potentials/coulomb.py:43-171``)."""

    _kind = _lib.COULOMB
    _p = 1

    def __init__(
        self,
        smearing: float | None = None,
        exclusion_radius: float | None = None,
        exclusion_degree: int = 1,
        prefactor: float = 1.0,
    ):
        super().__init__(smearing, exclusion_radius, exclusion_degree, prefactor)

"""Building blocks of the mesh path with the reference's class / method names
(``lib/mesh_interpolator.py``, ``lib/kspace_filter.py``, ``lib/kvectors.py``), backed by libmipme.

These are the stage-level entry points (spread, gather, G(k), FFT convolution).  They are forward-only
utilities for inspection and for the stage-wise parity tests; the differentiable hot path is the fused
:class:`~calculators.PMECalculator` / :class:`~calculators.P3MCalculator` forward.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops
from .potentials import Potential


def get_ns_mesh(cell: torch.Tensor, mesh_spacing: float) -> torch.Tensor:
    """Mesh size per axis, the power of two above ``2 |a_d| / mesh_spacing + 1``
    (reference ``lib/kvectors.py:5-21``).  Returned on the device of ``cell``."""
    ns = ops.ns_mesh_from_cell(cell.detach().to("cpu", torch.float64).numpy(), mesh_spacing)
    return torch.tensor(ns, dtype=torch.int64, device=cell.device)


def _ns_tuple(ns_mesh) -> tuple:
    if isinstance(ns_mesh, torch.Tensor):
        if ns_mesh.shape != (3,):
            raise ValueError(f"shape {list(ns_mesh.shape)} of `ns_mesh` has to be (3,)")
        return tuple(int(v) for v in ns_mesh.tolist())
    return tuple(int(v) for v in ns_mesh)


class MeshInterpolator:
    """Particle <-> mesh interpolation (reference ``lib/mesh_interpolator.py:4-457``).

    ``compute_weights(positions)`` only records the positions: weights and stencil indices are recomputed
    inside the kernels (registers), never materialised in HBM.
    """

    def __init__(self, cell: torch.Tensor, ns_mesh, interpolation_nodes: int, method: str):
        if method == "Lagrange":
            if interpolation_nodes not in (3, 4, 5, 6, 7):
                raise ValueError(
                    f"`interpolation_nodes` is {interpolation_nodes} but only values "
                    f"from 3 to 7 for method 'Lagrange' are allowed"
                )
            self._scheme = _lib.LAGRANGE
        elif method == "P3M":
            if interpolation_nodes not in (1, 2, 3, 4, 5):
                raise ValueError(
                    f"`interpolation_nodes` is {interpolation_nodes} but only values "
                    "from 1 to 5 for method 'P3M' are allowed"
                )
            self._scheme = _lib.P3M
        else:
            raise ValueError(f"method '{method}' is not supported. Choose from 'Lagrange' or 'P3M'")
        self.method = method
        self.interpolation_nodes = interpolation_nodes
        self._positions = None
        self.cell = None
        self.ns_mesh = None
        self.update(cell, ns_mesh)

    def update(self, cell: torch.Tensor | None = None, ns_mesh=None) -> None:
        if cell is not None:
            if cell.shape != (3, 3):
                raise ValueError(f"cell of shape {list(cell.shape)} should be of shape (3, 3)")
            _lib.require_device(cell, "cell")
            self.cell = cell
            self._dtype, self._device = cell.dtype, cell.device
        if ns_mesh is not None:
            if isinstance(ns_mesh, torch.Tensor) and ns_mesh.device != self.cell.device:
                raise ValueError(
                    f"`cell` and `ns_mesh` are on different devices, got {self.cell.device} and {ns_mesh.device}"
                )
            self.ns_mesh = _ns_tuple(ns_mesh)
        self._geom = ops.MeshGeometry(
            self.cell.detach().to("cpu", torch.float64).numpy(), self.ns_mesh, self._scheme, self.interpolation_nodes
        )

    def compute_weights(self, positions: torch.Tensor):
        if positions.device != self._device:
            raise ValueError(
                f"`positions` device {positions.device} is not the same as instance device {self._device}"
            )
        if positions.dim() != 2 or positions.shape[1] != 3:
            raise ValueError(f"shape {list(positions.shape)} of `positions` has to be (N, 3)")
        self._positions = positions.detach().to(self._dtype).contiguous()

    def points_to_mesh(self, particle_weights: torch.Tensor) -> torch.Tensor:
        if particle_weights.device != self._device:
            raise ValueError(
                f"`particle_weights` device {particle_weights.device} is not the same as instance device {self._device}"
            )
        if particle_weights.dim() != 2:
            raise ValueError(f"`particle_weights` of dimension {particle_weights.dim()} has to be of dimension 2")
        vals = particle_weights.detach().to(self._dtype).contiguous()
        N, Cn = vals.shape
        mesh = torch.empty((Cn,) + self.ns_mesh, dtype=self._dtype, device=self._device)
        md = self._geom.desc(Cn)
        with torch.cuda.device(self._device):
            _lib.check(
                _lib.load().mipme_spread(
                    _lib.current_stream(self._device), _lib.dtype_code(self._dtype), C.byref(md), N,
                    self._positions.data_ptr(), vals.data_ptr(), mesh.data_ptr(),
                )
            )
        return mesh

    def mesh_to_points(self, mesh_vals: torch.Tensor) -> torch.Tensor:
        if mesh_vals.dim() != 4:
            raise ValueError(f"`mesh_vals` of dimension {mesh_vals.dim()} has to be of dimension 4")
        mesh = mesh_vals.detach().to(self._dtype).contiguous()
        Cn = mesh.shape[0]
        if tuple(mesh.shape[1:]) != self.ns_mesh:
            raise ValueError(f"`mesh_vals` of shape {list(mesh.shape)} does not match the mesh {list(self.ns_mesh)}")
        N = self._positions.shape[0]
        out = torch.empty((N, Cn), dtype=self._dtype, device=self._device)
        md = self._geom.desc(Cn)
        with torch.cuda.device(self._device):
            _lib.check(
                _lib.load().mipme_gather(
                    _lib.current_stream(self._device), _lib.dtype_code(self._dtype), C.byref(md), N,
                    self._positions.data_ptr(), mesh.data_ptr(), out.data_ptr(),
                )
            )
        return out


def generate_kvectors_for_mesh(cell: torch.Tensor, ns) -> torch.Tensor:
    """Reciprocal-space vectors of the rfft half grid of an ``(nx, ny, nz)`` mesh, shape ``(nx, ny, nz // 2 + 1, 3)``:
    ``k = 2 pi (f_x, f_y, f_z) A^-T`` with the integer frequencies of ``fftfreq`` along x and y and of ``rfftfreq`` along z
    (reference ``lib/kvectors.py:77-102``).  Differentiable w.r.t. ``cell``.  The calculators never materialise this grid (their
    kernels derive every k-vector from its index, ``csrc/kfilter.hip``); it is the inspection / custom-kernel utility."""
    if cell.shape != (3, 3):
        raise ValueError(f"cell of shape {list(cell.shape)} should be of shape (3, 3)")
    if isinstance(ns, torch.Tensor):
        if ns.shape != (3,):
            raise ValueError(f"ns of shape {list(ns.shape)} should be of shape (3, )")
        if ns.device != cell.device:
            raise ValueError(f"`ns` and `cell` are not on the same device, got {ns.device} and {cell.device}.")
    nx, ny, nz = _ns_tuple(ns)
    recip = (2 * torch.pi) * torch.linalg.inv_ex(cell)[0].T  # rows: reciprocal lattice vectors

    def freq(n, half):
        f = torch.arange(n // 2 + 1 if half else n, device=cell.device)
        if not half:
            f = torch.where(f < (n + 1) // 2, f, f - n)
        return f.to(cell.dtype)

    fx, fy, fz = freq(nx, False), freq(ny, False), freq(nz, True)
    return (fx[:, None, None, None] * recip[0] + fy[None, :, None, None] * recip[1] + fz[None, None, :, None] * recip[2])


class KSpaceKernel(torch.nn.Module):
    """Interface of a reciprocal-space kernel (reference ``lib/kspace_filter.py:7-35``): subclasses return the filter values
    for a tensor of squared k-vector norms.  The built-in potentials implement it as ``lr_from_k_sq``; a :class:`KSpaceFilter`
    given any other subclass tabulates ``kernel_from_k_sq`` once per cell with tensor operations and runs the same convolution
    kernels on the table."""

    def kernel_from_k_sq(self, k_sq: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError(f"kernel_from_k_sq is not implemented for '{self.__class__.__name__}'")


class KSpaceFilter:
    """``irfftn(rfftn(mesh) * G(k))`` with ``G = kernel(|k|^2)`` (reference ``lib/kspace_filter.py:31-222``).

    Only the un-normalised convention used by the calculators (``fft_norm="backward"``,
    ``ifft_norm="forward"``) is implemented; ``kernel`` is a built-in :class:`Potential` (G built by the device kernel) or any
    :class:`KSpaceKernel` (G tabulated from ``kernel_from_k_sq`` on :func:`generate_kvectors_for_mesh`)."""

    _scheme = _lib.LAGRANGE
    _order = 3

    def __init__(self, cell, ns_mesh, kernel: Potential, fft_norm: str = "backward", ifft_norm: str = "forward"):
        for name, val in (("fft_norm", fft_norm), ("ifft_norm", ifft_norm)):
            if val not in ("ortho", "forward", "backward"):
                raise ValueError(f"Invalid option '{val}' for the `{name}` parameter.")
        if (fft_norm, ifft_norm) != ("backward", "forward"):
            raise NotImplementedError("libmipme implements the un-normalised convention fft_norm='backward', ifft_norm='forward'")
        self.kernel = kernel
        self.cell = None
        self.ns_mesh = None
        self.update(cell, ns_mesh)

    def update(self, cell: torch.Tensor | None = None, ns_mesh=None) -> None:
        if cell is not None:
            if cell.shape != (3, 3):
                raise ValueError(f"cell of shape {list(cell.shape)} should be of shape (3, 3)")
            _lib.require_device(cell, "cell")
            self.cell = cell
        if ns_mesh is not None:
            if isinstance(ns_mesh, torch.Tensor) and ns_mesh.device != self.cell.device:
                raise ValueError(
                    f"`cell` and `ns_mesh` are on different devices, got {self.cell.device} and {ns_mesh.device}"
                )
            self.ns_mesh = _ns_tuple(ns_mesh)
        self._geom = ops.MeshGeometry(
            self.cell.detach().to("cpu", torch.float64).numpy(), self.ns_mesh, self._scheme, self._order
        )
        if isinstance(self.kernel, Potential):
            self._kfilter = ops.build_filter(self._geom, self.kernel._descriptor(), self.cell.dtype, self.cell.device)
        else:  # a custom KSpaceKernel: tabulated with tensor operations, then the same convolution kernels
            k = generate_kvectors_for_mesh(self.cell.detach(), self.ns_mesh)
            self._kfilter = self.kernel.kernel_from_k_sq((k * k).sum(-1)).to(self.cell.dtype).contiguous()

    def forward(self, mesh_values: torch.Tensor) -> torch.Tensor:
        if mesh_values.dim() != 4:
            raise ValueError(f"`mesh_values` needs to be a 4 dimensional tensor, got {mesh_values.dim()}")
        if mesh_values.device != self._kfilter.device:
            raise ValueError(
                "`mesh_values` and the k-space filter are on different devices, got "
                f"{mesh_values.device} and {self._kfilter.device}"
            )
        if tuple(mesh_values.shape[1:]) != self.ns_mesh:
            raise ValueError("The real-space mesh is inconsistent with the k-space grid.")
        dtype, device = self._kfilter.dtype, self._kfilter.device
        mesh = mesh_values.detach().to(dtype).contiguous()
        Cn = mesh.shape[0]
        cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
        hat = torch.empty((Cn, self._geom.n_half), dtype=cdtype, device=device)
        work = torch.empty_like(hat)
        out = torch.empty_like(mesh)
        plan = _lib.get_plan(device, dtype, self.ns_mesh, Cn)
        with torch.cuda.device(device):
            _lib.check(
                _lib.load().mipme_convolve(
                    plan.handle, _lib.current_stream(device), mesh.data_ptr(), self._kfilter.data_ptr(),
                    hat.data_ptr(), work.data_ptr(), out.data_ptr(), None,
                )
            )
        return out

    __call__ = forward


class P3MKSpaceFilter(KSpaceFilter):
    """P3M influence-function filter ``G = kernel(|k|^2) / U^2(k)``, mode 0
    (reference ``lib/kspace_filter.py:225-363``)."""

    _scheme = _lib.P3M

    def __init__(self, cell, ns_mesh, interpolation_nodes: int, kernel: Potential, fft_norm: str = "backward",
                 ifft_norm: str = "forward", mode: int = 0, differential_order: int = 2):
        if mode not in (0, 1, 2, 3):
            raise ValueError(f"`mode` should be one of [0, 1, 2, 3], but got {mode}")
        if differential_order not in (1, 2, 3, 4, 5, 6):
            raise ValueError(f"`differential_order` should be one between 1 and 6, but got {differential_order}")
        if mode != 0:
            raise NotImplementedError("only mode 0 (point-charge potentials) is on the MI355X path")
        if interpolation_nodes not in (1, 2, 3, 4, 5):
            raise ValueError(
                f"`interpolation_nodes` is {interpolation_nodes} but only values from 1 to 5 for method 'P3M' are allowed"
            )
        self._order = interpolation_nodes
        self.interpolation_nodes = interpolation_nodes
        self.mode = mode
        self.differential_order = differential_order
        super().__init__(cell, ns_mesh, kernel, fft_norm, ifft_norm)


def generate_kvectors_for_ewald(cell: torch.Tensor, ns) -> torch.Tensor:
    """All reciprocal-space vectors ``2 pi (f_x, f_y, f_z) A^-T`` with integer frequencies ``f_d = fftfreq(ns_d) ns_d``,
    shape ``(nx ny nz, 3)``, the zero vector first (reference ``lib/kvectors.py:105-136``).  Differentiable w.r.t. ``cell``."""
    if cell.shape != (3, 3):
        raise ValueError(f"cell of shape {list(cell.shape)} should be of shape (3, 3)")
    ns = _ns_tuple(ns)
    if len(ns) != 3:
        raise ValueError(f"ns of shape {[len(ns)]} should be of shape (3, )")
    freqs = [torch.fft.fftfreq(n, device=cell.device, dtype=cell.dtype) * n for n in ns]
    F = torch.stack(torch.meshgrid(*freqs, indexing="ij"), dim=-1).reshape(-1, 3)
    return (2 * torch.pi) * F @ torch.linalg.inv(cell).T


def compute_batched_kvectors(lr_wavelength: float, cells: torch.Tensor) -> torch.Tensor:
    """Zero-padded k-vector sets ``(B, K_max, 3)`` of a batch of cells for ``torch.vmap(EwaldCalculator.forward)``
    (reference ``lib/kvectors.py:139-166``; the padding relies on G(k = 0) = 0)."""
    sets = []
    for cell in cells:
        ns = torch.ceil(torch.linalg.norm(cell, dim=1) / lr_wavelength).long()
        sets.append(generate_kvectors_for_ewald(cell, ns))
    return torch.nn.utils.rnn.pad_sequence(sets, batch_first=True)

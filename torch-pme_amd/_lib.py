"""ctypes binding of ``libmipme.so`` (C-ABI declared in ``include/mipme.h``).

There is deliberately NO fallback: if the HIP library is missing or a call fails, the product
path raises.  ``import torch`` happens before the ``dlopen`` so that ``libamdhip64.so.7`` /
``libhipfft.so.0`` resolve to the copies PyTorch already loaded (one HIP runtime per process, so
``torch.cuda.current_stream().cuda_stream`` is a valid ``hipStream_t`` for the library).
"""

from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIPME_LIB", os.path.join(_HERE, "libmipme.so"))  # MIPME_LIB: alternative build (experiments)

F32, F64 = 0, 1
I64, I32 = 0, 1
LAGRANGE, P3M = 0, 1
COULOMB, INVERSE_POWER_LAW = 0, 1

EXPORTS = (
    "mipme_last_error", "mipme_version", "mipme_fft_plan_create", "mipme_fft_plan_destroy", "mipme_kfilter_build",
    "mipme_convolve", "mipme_spread", "mipme_gather", "mipme_kspace_forward", "mipme_kspace_backward",
    "mipme_cellgrad_partials_size", "mipme_slab_forward", "mipme_slab_backward", "mipme_rspace_forward",
    "mipme_rspace_backward", "mipme_pair_distance_forward", "mipme_pair_distance_backward",
    "mipme_pair_partials_size", "mipme_topology_workspace_bytes", "mipme_topology_build", "mipme_topology_pack_shifts",
    "mipme_rspace_rows", "mipme_rspace_rows_value_bytes", "mipme_rspace_rows_tabulate", "mipme_rspace_rows_tabulated", "mipme_pair_distance_backward_rows", "mipme_rows_partials_size", "mipme_atom_bins_bytes", "mipme_plane_spread_parts", "mipme_last_cosched_kernel", "mipme_frames_counter_ints",
    "mipme_profile_enable", "mipme_profile_report", "mipme_dot_forward", "mipme_dot_backward", "mipme_energy_log_push", "mipme_frames_table_energy_log",
    "mipme_nl_workspace_bytes", "mipme_nl_bin", "mipme_nl_count", "mipme_nl_fill", "mipme_nl_stream",
    "mipme_topology_pack_entries", "mipme_sr_rows_fused", "mipme_sr_rows_finalize",
    "mipme_pack_pair_shifts", "mipme_pair_distance_forward_packed", "mipme_fft_plan_xfused", "mipme_fft_plan_kgrid_blocks", "mipme_fft_r2c",
    "mipme_ewald_filter", "mipme_ewald_structure", "mipme_ewald_potential", "mipme_ewald_backward",
    "mipme_frames_table_bytes", "mipme_frames_table_build", "mipme_frames_forward", "mipme_frames_backward",
    "mipme_scaled_match", "mipme_scaled_match_work", "mipme_scaled_match_wide", "mipme_md_supported", "mipme_md_lists_ints", "mipme_md_rebin", "mipme_md_step", "mipme_set_skip_flag", "mipme_energy_select", "mipme_energy_select_sum", "mipme_energy_select_contract",
    "mipme_kfilter_build_deriv", "mipme_cell_tail_work", "mipme_values_equal", "mipme_checksum", "mipme_checksum_words",
    "mipme_spread_jet", "mipme_gather_jet", "mipme_gather_jet3", "mipme_pair_sum", "mipme_pair_sum_rows", "mipme_pair_dot", "mipme_pair_diff", "mipme_pair_scatter",
)


class PotentialDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("exponent", C.c_int32),
        ("smearing", C.c_double),
        ("prefactor", C.c_double),
        ("exclusion_radius", C.c_double),
        ("exclusion_degree", C.c_int32),
        ("_pad", C.c_int32),
    ]


class MeshDesc(C.Structure):
    _fields_ = [
        ("scheme", C.c_int32),
        ("order", C.c_int32),
        ("nx", C.c_int32),
        ("ny", C.c_int32),
        ("nz", C.c_int32),
        ("n_channels", C.c_int32),
        ("cell", C.c_double * 9),
        ("inv_cell", C.c_double * 9),
        ("volume", C.c_double),
    ]


class SrJob(C.Structure):
    """``mipme_sr_job_t``: the short-range pair sum co-scheduled with the spread by ``mipme_kspace_forward``."""

    _fields_ = [
        ("n_atoms", C.c_int64),
        ("row_ptr", C.c_void_p),
        ("entries_shift", C.c_void_p),
        ("entries", C.c_void_p),
        ("positions", C.c_void_p),
        ("cell", C.c_void_p),
        ("charges", C.c_void_p),
        ("pot", C.POINTER(PotentialDesc)),
        ("full_list", C.c_int32),
        ("shift_format", C.c_int32),
        ("records", C.c_void_p),
        ("out", C.c_void_p),
        ("force", C.c_void_p),
        ("dist_out", C.c_void_p),
    ]


class Frame(C.Structure):
    """``mipme_frame_t``: one frame of ``mipme_frames_forward / backward`` (independent frames in one launch)."""

    _fields_ = [
        ("n_atoms", C.c_int64),
        ("positions", C.c_void_p),
        ("charges", C.c_void_p),
        ("cell", C.c_void_p),
        ("mesh", MeshDesc),
        ("atom_bins", C.c_void_p),
        ("brick_counters", C.c_void_p),
        ("row_ptr", C.c_void_p),
        ("entries_shift", C.c_void_p),
        ("entries", C.c_void_p),
        ("full_list", C.c_int32),
        ("shift_format", C.c_int32),
        ("records", C.c_void_p),
        ("rho_mesh", C.c_void_p),
        ("phi_mesh", C.c_void_p),
        ("dc", C.c_void_p),
        ("out", C.c_void_p),
        ("force", C.c_void_p),
        ("field", C.c_void_p),
        ("dist_out", C.c_void_p),
        ("energy", C.c_void_p),
        ("grad_positions", C.c_void_p),
        ("use_tail", C.c_int32),
        ("counter_ints", C.c_int32),
        ("grad_seed", C.c_void_p),
    ]


ARGS_VERSION = 2


class _VersionedArgs(C.Structure):
    """Base of the versioned argument structs (``include/mipme.h``): ``size`` / ``version`` are filled in here, every other
    field is passed by NAME -- a binding that forgets a field passes NULL for it instead of shifting the rest."""

    def __init__(self, **fields):
        unknown = set(fields) - {name for name, _ in self._fields_}
        if unknown:
            raise TypeError(f"{type(self).__name__}: unknown field(s) {sorted(unknown)}")
        super().__init__(size=C.sizeof(type(self)), version=ARGS_VERSION, **fields)


class KspaceForwardArgs(_VersionedArgs):
    """``mipme_kspace_forward_args_t``"""

    _fields_ = [
        ("size", C.c_uint32), ("version", C.c_uint32),
        ("plan", C.c_void_p), ("stream", C.c_void_p), ("dtype", C.c_int32), ("accumulate_out", C.c_int32),
        ("mesh", C.POINTER(MeshDesc)), ("pot", C.POINTER(PotentialDesc)), ("n_atoms", C.c_int64),
        ("positions", C.c_void_p), ("charges", C.c_void_p), ("G", C.c_void_p),
        ("rho_mesh", C.c_void_p), ("rho_hat", C.c_void_p), ("hat_work", C.c_void_p), ("phi_mesh", C.c_void_p),
        ("dc", C.c_void_p), ("out_lr", C.c_void_p), ("out_phi", C.c_void_p), ("atom_bins", C.c_void_p),
        ("gather_wait_event", C.c_void_p), ("out_field", C.c_void_p), ("out_records", C.c_void_p),
        ("sr_job", C.POINTER(SrJob)), ("out_cell_partials", C.c_void_p),
        ("out_energy", C.c_void_p), ("out_grad_positions", C.c_void_p), ("grad_seed", C.c_void_p),
        ("nan_flag", C.c_void_p),
        ("out_grad_charges", C.c_void_p), ("out_grad_cell", C.c_void_p), ("G_deriv", C.c_void_p), ("cell_work", C.c_void_p),
        ("aux_seed", C.c_void_p), ("out_rho_hat", C.c_void_p), ("flags", C.c_int64),
        ("energy_log", C.c_void_p), ("energy_log_cursor", C.c_void_p), ("energy_log_capacity", C.c_int64),
    ]


class KspaceBackwardArgs(_VersionedArgs):
    """``mipme_kspace_backward_args_t``"""

    _fields_ = [
        ("size", C.c_uint32), ("version", C.c_uint32),
        ("plan", C.c_void_p), ("stream", C.c_void_p), ("dtype", C.c_int32), ("_pad", C.c_int32),
        ("mesh", C.POINTER(MeshDesc)), ("pot", C.POINTER(PotentialDesc)), ("n_atoms", C.c_int64),
        ("positions", C.c_void_p), ("charges", C.c_void_p), ("grad_out", C.c_void_p), ("G", C.c_void_p),
        ("phi_mesh", C.c_void_p), ("rho_hat", C.c_void_p), ("rho_dc", C.c_void_p), ("phi_atoms", C.c_void_p),
        ("psi_mesh", C.c_void_p), ("psi_hat", C.c_void_p), ("hat_work", C.c_void_p), ("chi_mesh", C.c_void_p),
        ("dc", C.c_void_p), ("partials", C.c_void_p), ("grad_positions", C.c_void_p), ("grad_charges", C.c_void_p),
        ("grad_cell", C.c_void_p), ("atom_bins", C.c_void_p), ("grad_scale", C.c_void_p), ("mesh_field", C.c_void_p),
        ("kgrid_blocks_ready", C.c_int64), ("G_deriv", C.c_void_p),
    ]


class NlDesc(C.Structure):
    _fields_ = [
        ("cell", C.c_double * 9),
        ("inv_cell", C.c_double * 9),
        ("n_cells", C.c_int32 * 3),
        ("periodic", C.c_int32 * 3),
        ("cutoff", C.c_double),
        ("full_list", C.c_int32),
        ("_pad", C.c_int32),
        ("frac_offset", C.c_double * 3),
        ("frac_scale", C.c_double * 3),
        ("reach", C.c_int32 * 3),
        ("position_stride", C.c_int32),
    ]


class MdArgs(_VersionedArgs):
    """``mipme_md_args_t`` (version 1)"""

    _fields_ = [
        ("size", C.c_uint32), ("version", C.c_uint32),
        ("plan", C.c_void_p), ("stream", C.c_void_p), ("dtype", C.c_int32), ("shift_format", C.c_int32),
        ("mesh", C.POINTER(MeshDesc)), ("pot", C.POINTER(PotentialDesc)), ("n_atoms", C.c_int64),
        ("records", C.c_void_p), ("cell", C.c_void_p), ("G", C.c_void_p),
        ("rho_mesh", C.c_void_p), ("hat_work", C.c_void_p), ("phi_mesh", C.c_void_p), ("dc", C.c_void_p),
        ("atom_bins", C.c_void_p), ("live_lists", C.c_void_p), ("row_ptr", C.c_void_p), ("words", C.c_void_p),
        ("potentials", C.c_void_p), ("pair_force", C.c_void_p), ("energy", C.c_void_p), ("grad_positions", C.c_void_p),
        ("grad_seed", C.c_void_p), ("nan_flag", C.c_void_p), ("host_flags", C.c_void_p),
        ("grad_charges", C.c_void_p), ("grad_cell", C.c_void_p), ("G_deriv", C.c_void_p), ("cell_work", C.c_void_p),
        ("aux_seed", C.c_void_p),
        ("energy_log", C.c_void_p), ("energy_log_cursor", C.c_void_p), ("energy_log_capacity", C.c_int64),
    ]

    def __init__(self, **fields):
        C.Structure.__init__(self, size=C.sizeof(type(self)), version=1, **fields)


#: OR-ed into a shift format: rows written by ``mipme_nl_stream`` (``row_ptr`` int32[3N+1], every neighbour once per row)
ROWS_PADDED = 0x100
FWD_RHO_MESH_UNUSED = 1  # mipme_kspace_forward_args_t.flags: the caller never reads rho_mesh after the call


class MipmeError(RuntimeError):
    pass


_lib = None


def _declare(lib):
    vp, i64, ci, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    MP, PP = C.POINTER(MeshDesc), C.POINTER(PotentialDesc)
    lib.mipme_last_error.restype = C.c_char_p
    lib.mipme_last_error.argtypes = []
    lib.mipme_version.restype = ci
    lib.mipme_version.argtypes = []
    sig = {
        "mipme_fft_plan_create": [ci, ci, ci, ci, ci, C.POINTER(vp)],
        "mipme_fft_plan_destroy": [vp],
        "mipme_kfilter_build": [vp, ci, MP, PP, vp],
        "mipme_kfilter_build_deriv": [vp, ci, MP, PP, vp],
        "mipme_convolve": [vp, vp, vp, vp, vp, vp, vp, vp],
        "mipme_spread": [vp, ci, MP, i64, vp, vp, vp],
        "mipme_gather": [vp, ci, MP, i64, vp, vp, vp],
        "mipme_spread_jet": [vp, ci, MP, i64, vp, vp, ci, ci, ci, vp],
        "mipme_gather_jet": [vp, ci, MP, i64, vp, vp, ci, ci, ci, vp],
        "mipme_gather_jet3": [vp, ci, MP, i64, vp, vp, ci, ci, ci, vp],
        "mipme_pair_sum": [vp, ci, ci, i64, i64, ci, vp, vp, vp, ci, vp],
        "mipme_pair_sum_rows": [vp, ci, i64, ci, vp, vp, vp, vp, ci, vp],
        "mipme_pair_dot": [vp, ci, ci, i64, ci, vp, vp, vp, ci, vp],
        "mipme_pair_diff": [vp, ci, ci, i64, ci, vp, vp, vp],
        "mipme_pair_scatter": [vp, ci, ci, i64, i64, ci, vp, vp, vp, vp, vp],
        "mipme_kspace_forward": [C.POINTER(KspaceForwardArgs)],
        "mipme_kspace_backward": [C.POINTER(KspaceBackwardArgs)],
        "mipme_slab_forward": [vp, ci, ci, MP, dbl, i64, vp, vp, vp, vp],
        "mipme_slab_backward": [vp, ci, ci, MP, dbl, i64, vp, vp, vp, vp, vp, vp, vp],
        "mipme_rspace_forward": [vp, ci, ci, i64, i64, ci, vp, vp, vp, vp, ci, PP, ci, vp],
        "mipme_rspace_backward": [vp, ci, ci, i64, i64, ci, vp, vp, vp, vp, ci, PP, vp, vp, vp, vp],
        "mipme_pair_distance_forward": [vp, ci, ci, i64, vp, vp, vp, vp, vp],
        "mipme_pair_distance_backward": [vp, ci, ci, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp],
        "mipme_topology_build": [vp, ci, i64, i64, vp, vp, i64, vp, vp],
        "mipme_topology_pack_shifts": [vp, ci, i64, vp, vp, vp, vp],
        "mipme_rspace_rows": [vp, ci, i64, ci, vp, vp, vp, vp, vp, ci, ci, PP, ci, vp],
        "mipme_rspace_rows_tabulate": [vp, ci, i64, vp, vp, vp, vp, ci, PP, vp, vp],
        "mipme_rspace_rows_tabulated": [vp, ci, i64, vp, vp, vp, ci, ci, ci, vp],
        "mipme_pair_distance_backward_rows": [vp, ci, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "mipme_topology_pack_entries": [vp, ci, i64, i64, vp, vp, vp, ci, vp, vp],
        "mipme_pack_pair_shifts": [vp, ci, i64, vp, vp, vp],
        "mipme_pair_distance_forward_packed": [vp, ci, i64, vp, vp, vp, vp, vp],
        "mipme_sr_rows_fused": [vp, ci, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, PP, ci, ci, vp, ci, vp, vp, vp, vp, vp],
        "mipme_sr_rows_finalize": [vp, ci, i64, vp, vp, vp, vp, ci, vp, vp, vp],
        "mipme_fft_r2c": [vp, vp, ci, MP, vp, vp],
        "mipme_frames_table_build": [ci, ci, C.POINTER(Frame), PP, vp, i64],
        "mipme_frames_forward": [vp, vp, ci, ci, C.POINTER(Frame), PP, vp, vp, i64, vp, vp, vp, vp],
        "mipme_frames_backward": [vp, ci, ci, C.POINTER(Frame), vp, vp],
        "mipme_ewald_filter": [vp, ci, PP, i64, vp, vp, vp],
        "mipme_ewald_structure": [vp, ci, i64, ci, i64, vp, vp, vp, vp, vp, i64],
        "mipme_ewald_potential": [vp, ci, i64, ci, i64, vp, vp, vp, vp, vp, vp, i64],
        "mipme_ewald_backward": [vp, ci, i64, ci, i64] + [vp] * 12 + [i64],
        "mipme_dot_forward": [vp, ci, i64, vp, vp, vp, vp],
        "mipme_dot_backward": [vp, ci, i64, vp, vp, vp, vp, vp],
        "mipme_energy_log_push": [vp, ci, ci, vp, vp, vp, ci],
        "mipme_frames_table_energy_log": [ci, ci, vp, i64, vp, vp, ci],
        "mipme_scaled_match": [vp, ci, i64, vp, vp, vp, vp],
        "mipme_scaled_match_work": [i64],
        "mipme_scaled_match_wide": [vp, ci, i64, vp, vp, vp, vp, vp],
        "mipme_nl_bin": [vp, ci, C.POINTER(NlDesc), i64, vp, vp],
        "mipme_nl_count": [vp, ci, C.POINTER(NlDesc), i64, vp, vp],
        "mipme_nl_fill": [vp, ci, C.POINTER(NlDesc), i64, vp, vp, vp, vp, vp],
        "mipme_nl_stream": [vp, ci, C.POINTER(NlDesc), i64, vp, i64, vp, vp, vp],
        "mipme_set_skip_flag": [vp],
        "mipme_values_equal": [vp, ci, i64, vp, vp, vp],
        "mipme_checksum": [vp, vp, i64, vp, vp, vp],
        "mipme_energy_select": [vp, ci, i64, vp, vp, vp, vp, ci, vp, vp],
        "mipme_energy_select_sum": [vp, ci, i64, vp, vp, vp, vp, ci, vp, vp, vp],
        "mipme_energy_select_contract": [vp, ci, i64, vp, vp, vp, vp, ci, vp, vp, vp],
        "mipme_md_rebin": [C.POINTER(MdArgs)],
        "mipme_md_step": [C.POINTER(MdArgs)],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.restype = ci
        fn.argtypes = argtypes
    lib.mipme_cellgrad_partials_size.restype = i64
    lib.mipme_cellgrad_partials_size.argtypes = [MP, i64]
    lib.mipme_pair_partials_size.restype = i64
    lib.mipme_pair_partials_size.argtypes = [i64]
    lib.mipme_topology_workspace_bytes.restype = i64
    lib.mipme_topology_workspace_bytes.argtypes = [i64]
    lib.mipme_rows_partials_size.restype = i64
    lib.mipme_rows_partials_size.argtypes = [i64]
    lib.mipme_atom_bins_bytes.restype = i64
    lib.mipme_atom_bins_bytes.argtypes = [MP, i64, ci]
    lib.mipme_last_cosched_kernel.restype = C.c_char_p
    lib.mipme_last_cosched_kernel.argtypes = []
    lib.mipme_plane_spread_parts.restype = ci
    lib.mipme_plane_spread_parts.argtypes = [MP, i64, ci]
    lib.mipme_frames_counter_ints.restype = i64
    lib.mipme_frames_counter_ints.argtypes = [MP, i64, ci]
    lib.mipme_md_supported.restype = ci
    lib.mipme_md_supported.argtypes = [MP, PP, i64, ci]
    lib.mipme_md_lists_ints.restype = i64
    lib.mipme_md_lists_ints.argtypes = [MP, i64]
    lib.mipme_nl_workspace_bytes.restype = i64
    lib.mipme_nl_workspace_bytes.argtypes = [C.POINTER(NlDesc), i64]
    lib.mipme_rspace_rows_value_bytes.restype = i64
    lib.mipme_rspace_rows_value_bytes.argtypes = [ci, i64]
    lib.mipme_checksum_words.restype = i64
    lib.mipme_checksum_words.argtypes = []
    lib.mipme_cell_tail_work.restype = i64
    lib.mipme_cell_tail_work.argtypes = [vp, MP, i64]
    lib.mipme_fft_plan_xfused.restype = ci
    lib.mipme_fft_plan_xfused.argtypes = [vp]
    lib.mipme_fft_plan_kgrid_blocks.restype = i64
    lib.mipme_fft_plan_kgrid_blocks.argtypes = [vp]
    lib.mipme_frames_table_bytes.restype = i64
    lib.mipme_frames_table_bytes.argtypes = [ci, ci]
    lib.mipme_profile_enable.restype = ci
    lib.mipme_profile_enable.argtypes = [ci]
    lib.mipme_profile_report.restype = i64
    lib.mipme_profile_report.argtypes = [C.c_char_p, i64]


def load():
    """Load (once) and return the ctypes handle of libmipme.so; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MipmeError(
                f"HIP extension {LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C torch-pme_amd/csrc`). There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        _declare(lib)
        _lib = lib
    return _lib


def check(rc: int):
    """Translate a C-ABI status into a Python exception (ValueError for MIPME_EINVAL)."""
    if rc == 0:
        return
    msg = load().mipme_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    raise MipmeError(f"libmipme error {rc}: {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def dtype_code(dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.float64:
        return F64
    raise TypeError(f"libmipme supports float32 and float64 tensors, got {dtype}")


def index_code(dtype) -> int:
    if dtype == torch.int64:
        return I64
    if dtype == torch.int32:
        return I32
    raise TypeError(f"neighbor indices must be int64 or int32, got {dtype}")


def current_stream(device) -> int:
    """``hipStream_t`` of PyTorch's current stream on ``device`` (raw handle: no Stream object per call)."""
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CTX = _NoCtx()


def on_device(device):
    """Context that makes ``device`` the current HIP device for the launches inside -- a no-op object when it already is
    (``torch.cuda.device`` costs ~4 us per enter / exit, several times per call)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_CTX
    return torch.cuda.device(device)


def require_device(t, name: str):
    if not t.is_cuda:
        raise MipmeError(
            f"`{name}` lives on {t.device}: the MI355X-native path only runs on HIP devices "
            "(tensors on 'cuda'); there is no CPU fallback."
        )


class FFTPlan:
    """Owning wrapper of a ``mipme_fft_plan`` (own (y,z) plane kernels + x stage; hipFFT R2C / C2R for the general paths)."""

    def __init__(self, device, dtype, ns, batch):
        self.key = (torch.device(device).index, dtype, tuple(int(n) for n in ns), int(batch))
        handle = C.c_void_p()
        with torch.cuda.device(device):
            check(load().mipme_fft_plan_create(dtype_code(dtype), int(ns[0]), int(ns[1]), int(ns[2]), int(batch), C.byref(handle)))
        self.handle = handle
        #: the plan can run the convolution as (y,z) plane transforms + one fused x kernel (power-of-two nx)
        self.xfused = bool(load().mipme_fft_plan_xfused(handle))
        #: number of k-grid partial sums the fused convolution writes when asked for the cell sums (out_cell_partials)
        self.kgrid_blocks = int(load().mipme_fft_plan_kgrid_blocks(handle))

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.mipme_fft_plan_destroy(self.handle)
        except Exception:
            pass


_PLANS: dict = {}


def get_plan(device, dtype, ns, batch, store: dict | None = None) -> FFTPlan:
    """Plan of (device, dtype, mesh, batch).  A plan owns device state that is live between its kernels (the brick counters
    of the binning pass), so evaluations that may run concurrently on different streams must not share one, and a plan must
    outlive every captured HIP graph that launches its kernels: the calculators therefore pass their own ``store`` dict
    (one plan per calculator and mesh, freed with the calculator).  Without a store a small global cache is used."""
    key = (torch.device(device).index, dtype, tuple(int(n) for n in ns), int(batch))
    cache = _PLANS if store is None else store
    plan = cache.get(key)
    if plan is None:
        if store is None and len(_PLANS) >= 32:
            _PLANS.pop(next(iter(_PLANS)))
        plan = FFTPlan(device, dtype, ns, batch)
        cache[key] = plan
    return plan


def profile_enable(on: bool):
    check(load().mipme_profile_enable(1 if on else 0))


def profile_report() -> dict:
    """{stage: (calls, total_ms)} recorded by the library since ``profile_enable(True)``."""
    lib = load()
    n = lib.mipme_profile_report(None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    lib.mipme_profile_report(buf, n + 1)
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split()
        out[name] = (int(calls), float(ms))
    return out

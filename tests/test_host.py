"""CPU tests (no GPU) of the host side: the C-ABI library loads and exports every declared symbol, argument
validation reproduces the reference's exception types and messages, constructors, prefactors, neighbour list."""

import math
import os
import re

import numpy as np
import pytest
import torch

import torchpme_amd as tpa
from torchpme_amd import _lib
from torchpme_amd.neighbors import neighbor_list, neighbor_list_bruteforce

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mipme.h")).read()
    declared = sorted(set(re.findall(r"\b(mipme_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"libmipme.so does not export {name}"
    assert sorted(declared) == sorted(_lib.EXPORTS)
    assert lib.mipme_version() == int(re.search(r"#define\s+MIPME_VERSION\s+(\d+)", hdr).group(1)) == 408


def test_build_hook_accepts_the_tree():
    """__graft_entry__.build(): make (a no-op on an up-to-date tree), every export present, the library's version that of the
    header, the compiled front end loadable."""
    import __graft_entry__ as entry

    entry.build()


def test_compiled_front_end_loads_and_declines_cpu_tensors():
    """csrc/front.cpp (C++ autograd nodes of the reference call sequence): the extension is built in-tree, binds libmipme at
    the path the ctypes layer uses, and -- host logic only, no kernel runs here -- declines anything that is not its case."""
    from torchpme_amd import _front

    mod = _front.module()
    assert mod is not None, "torch-pme_amd/_mipme_front.so is not built (make -C torch-pme_amd/csrc front)"
    for name in ("pair_distances", "calc_forward", "is_front_distances", "Topology", "Calculator", "load_library"):
        assert hasattr(mod, name)
    x = torch.zeros(5, requires_grad=True)
    assert not mod.is_front_distances(x) and not mod.is_front_distances(x * 2)
    # descriptor sizes of the ctypes mirror and of include/mipme.h as front.cpp was compiled against it
    pot = tpa.CoulombPotential(smearing=1.0)._descriptor()
    with pytest.raises(RuntimeError, match="descriptor size mismatch"):
        mod.Calculator(b"x", bytes(pot), 0, torch.zeros(1), torch.eye(3), False, 0, 1, None, None)
    md = _lib.MeshDesc()
    fc = mod.Calculator(bytes(md), bytes(pot), 0, torch.zeros(1), torch.eye(3), False, 0, 1, None, None)
    # CPU tensors: not this file's case -> None, the Python path (which raises the reference's errors) takes over
    pos = torch.zeros((4, 3), requires_grad=True)
    assert mod.calc_forward(fc, torch.zeros((4, 1)), torch.eye(3), pos, torch.zeros((2, 2), dtype=torch.int64), torch.zeros(2)) is None
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=1.0)
    assert calc._front_forward(torch.zeros((4, 1)), torch.eye(3), pos, torch.zeros((2, 2), dtype=torch.int64), torch.zeros(2)) is None


def test_abi_struct_layout_matches_header():
    import ctypes as C

    assert C.sizeof(_lib.PotentialDesc) == 40
    assert C.sizeof(_lib.MeshDesc) == 24 + 19 * 8
    assert _lib.MeshDesc.cell.offset == 24 and _lib.MeshDesc.volume.offset == 24 + 18 * 8
    assert C.sizeof(_lib.NlDesc) == 18 * 8 + 6 * 4 + 8 + 8 + 6 * 8 + 16 and _lib.NlDesc.frac_offset.offset == 18 * 8 + 24 + 16
    assert _lib.NlDesc.reach.offset == 18 * 8 + 24 + 16 + 48


def test_no_cpu_fallback():
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.5)
    z = lambda *s: torch.zeros(*s, dtype=torch.float64)  # noqa: E731
    with pytest.raises(tpa.MipmeError, match="no CPU fallback"):
        calc(z(2, 1), torch.eye(3, dtype=torch.float64), z(2, 3), torch.zeros((1, 2), dtype=torch.long), z(1))
    with pytest.raises(tpa.MipmeError, match="no CPU fallback"):
        tpa.pair_distances(z(2, 3), torch.zeros((1, 2), dtype=torch.long))


def test_abi_argument_errors_without_gpu():
    """Error paths of the C-ABI that return before touching the device."""
    import ctypes as C

    lib = _lib.load()
    md = _lib.MeshDesc(scheme=_lib.P3M, order=9, nx=4, ny=4, nz=4, n_channels=1)
    pd = _lib.PotentialDesc(kind=_lib.COULOMB, exponent=1, smearing=1.0, prefactor=1.0, exclusion_radius=-1, exclusion_degree=1)
    rc = lib.mipme_kfilter_build(None, _lib.F32, C.byref(md), C.byref(pd), None)
    assert rc == -1
    assert b"only values from 1 to 5 for method 'P3M' are allowed" in lib.mipme_last_error()
    with pytest.raises(ValueError, match="from 1 to 5"):
        _lib.check(rc)
    # argument checks of the entry points added for the fused / Ewald / tuning rows (all return MIPME_EINVAL = -1 before
    # any launch; messages via mipme_last_error)
    assert lib.mipme_sr_rows_fused(None, _lib.F32, 4, None, None, None, None, None, None, None, None, None, 0, 0,
                                   C.byref(pd), 0, 0, None, 0, None, None, None, None, None) == -1
    assert b"mipme_sr_rows_fused" in lib.mipme_last_error()
    assert C.sizeof(_lib.SrJob) == 104  # mipme_sr_job_t: int64 + 7 pointers + 2 x int32 + 4 pointers
    # mipme_frame_t: int64 + 3 pointers + mipme_mesh_t (6 x int32 + 19 doubles) + 5 pointers + 2 x int32 + 12 pointers
    assert C.sizeof(_lib.Frame) == 8 + 24 + C.sizeof(_lib.MeshDesc) + 40 + 8 + 96 and C.sizeof(_lib.MeshDesc) == 176
    frames = (_lib.Frame * 2)()
    assert lib.mipme_frames_table_bytes(_lib.F32, 2) > 0 and lib.mipme_frames_table_bytes(_lib.F32, 0) == 0
    assert lib.mipme_frames_table_build(_lib.F32, 2, frames, C.byref(pd), None, 0) == -1  # invalid (empty) mesh descriptors
    assert lib.mipme_frames_forward(None, None, _lib.F32, 0, frames, C.byref(pd), None, None, 0, None, None, None, None) == -1
    assert b"no frames" in lib.mipme_last_error()
    assert lib.mipme_frames_backward(None, _lib.F32, 0, frames, None, None) == -1
    assert lib.mipme_sr_rows_finalize(None, _lib.F32, 4, None, None, None, None, 0, None, None, None) == -1
    assert lib.mipme_topology_pack_entries(None, _lib.F32, 4, 2, None, None, None, 0, None, None) == -1
    assert lib.mipme_pair_distance_forward_packed(None, _lib.F32, 4, None, None, None, None, None) == -1
    assert b"mipme_pair_distance_forward_packed" in lib.mipme_last_error()
    assert lib.mipme_ewald_structure(None, _lib.F32, 4, 0, 8, None, None, None, None, None, 1) == -1
    assert b"invalid sizes" in lib.mipme_last_error()
    assert lib.mipme_ewald_backward(None, 99, 0, 1, 0, *([None] * 12), 1) == -1
    assert b"invalid dtype 99" in lib.mipme_last_error()
    bad = _lib.PotentialDesc(kind=_lib.INVERSE_POWER_LAW, exponent=9, smearing=1.0, prefactor=1.0, exclusion_radius=-1,
                             exclusion_degree=1)
    assert lib.mipme_ewald_filter(None, _lib.F64, C.byref(bad), 0, None, None, None) == -1
    assert b"Unsupported exponent: 9" in lib.mipme_last_error()
    assert lib.mipme_fft_plan_xfused(None) == 0


# ---- constructors (reference tests/calculators/test_workflow.py:77-96, test_calculator.py) ----
def test_constructor_errors():
    with pytest.raises(TypeError, match="Potential must be an instance of Potential, got <class 'int'>"):
        tpa.Calculator(potential=1)
    with pytest.raises(ValueError, match="Must specify smearing to use a potential with PMECalculator"):
        tpa.PMECalculator(tpa.CoulombPotential(), mesh_spacing=1.0)
    with pytest.raises(ValueError, match="`smearing` is -1.0 but must be positive"):
        tpa.P3MCalculator(tpa.CoulombPotential(smearing=-1.0), mesh_spacing=1.0)
    with pytest.raises(ValueError, match="`interpolation_nodes` is 8 but only values from 3 to 7 for method 'Lagrange' are allowed"):
        tpa.PMECalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=1.0, interpolation_nodes=8)
    with pytest.raises(ValueError, match="`interpolation_nodes` is 6 but only values from 1 to 5 for method 'P3M' are allowed"):
        tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=1.0, interpolation_nodes=6)
    with pytest.raises(ValueError, match="Unsupported exponent: 7"):
        tpa.InversePowerLawPotential(exponent=7, smearing=1.0)
    calc = tpa.P3MCalculator(tpa.InversePowerLawPotential(exponent=6, smearing=1.5), mesh_spacing=0.3, interpolation_nodes=5,
                             full_neighbor_list=True)
    assert (calc.mesh_spacing, calc.interpolation_nodes, calc.full_neighbor_list) == (0.3, 5, True)
    assert isinstance(calc.potential, tpa.Potential)


def test_state_dict_buffers():
    pot = tpa.InversePowerLawPotential(exponent=3, smearing=1.2, prefactor=2.0)
    sd = pot.state_dict()
    assert set(sd) == {"smearing", "prefactor", "exponent"} and all(v.dtype == torch.float64 for v in sd.values())
    assert set(tpa.CoulombPotential(smearing=1.0).state_dict()) == {"smearing", "prefactor"}
    pot2 = tpa.InversePowerLawPotential(exponent=3, smearing=9.9)
    pot2.load_state_dict(sd)
    assert pot2._descriptor().smearing == pytest.approx(1.2)
    d = tpa.CoulombPotential(smearing=None, exclusion_radius=2.0)._descriptor()
    assert d.smearing < 0 and d.exclusion_radius == 2.0 and d.kind == _lib.COULOMB


def test_potential_closed_forms_match_oracle():
    """The inspection methods of the potentials agree with the oracle's (scipy) special functions."""
    from oracle import pme_numpy as O

    d = torch.linspace(0.3, 6.0, 50, dtype=torch.float64)
    for p in range(1, 7):
        pot = tpa.InversePowerLawPotential(exponent=p, smearing=0.8, prefactor=1.7)
        spec = O.PotentialSpec("ipl", p, 0.8, 1.7)
        np.testing.assert_allclose(pot.lr_from_dist(d).numpy(), O.lr_pair(spec, d.numpy())[0], rtol=1e-11)
        np.testing.assert_allclose(pot.sr_from_dist(d).numpy(), O.sr_pair(spec, d.numpy())[0], rtol=1e-8, atol=1e-14)
        assert float(pot.self_contribution()) == pytest.approx(O.self_term(spec), rel=1e-14)
        assert float(pot.background_correction()) == pytest.approx(O.background_term(spec), rel=1e-14)
        if p in (1, 2, 4, 6):
            k2 = torch.tensor([0.0, 0.1, 1.0, 7.0], dtype=torch.float64)
            np.testing.assert_allclose(pot.lr_from_k_sq(k2).numpy(), O.lr_kernel(spec, k2.numpy())[0], rtol=1e-12)
    c = tpa.CoulombPotential(smearing=0.8, prefactor=1.7)
    i = tpa.InversePowerLawPotential(exponent=1, smearing=0.8, prefactor=1.7)
    np.testing.assert_allclose(c.sr_from_dist(d).numpy(), i.sr_from_dist(d).numpy(), rtol=1e-14)


def test_prefactors():
    assert tpa.prefactors.eV_A == pytest.approx(14.399645478425667, rel=1e-15)
    assert tpa.prefactors.kcalmol_A == pytest.approx(332.0637132991921, rel=1e-15)
    assert tpa.prefactors.kJmol == pytest.approx(1389.3545764438197, rel=1e-15)
    assert tpa.prefactors.SI == pytest.approx(2.3070775523417355e-28, rel=1e-15)


# ---- _validate_parameters messages (reference tests/calculators/test_calculator.py:50-243) ----
def _args(n=4, p=3, dtype=torch.float32):
    return dict(
        charges=torch.ones((n, 2), dtype=dtype),
        cell=torch.eye(3, dtype=dtype),
        positions=torch.zeros((n, 3), dtype=dtype),
        neighbor_indices=torch.zeros((p, 2), dtype=torch.long),
        neighbor_distances=torch.ones(p, dtype=dtype),
    )


VALIDATION_CASES = [
    (dict(positions=torch.zeros((4, 5))), ValueError,
     r"`positions` must be a tensor with shape \[n_atoms, 3\], got tensor with shape \[4, 5\]"),
    (dict(cell=torch.eye(2)), ValueError, r"`cell` must be a tensor with shape \[3, 3\], got tensor with shape \[2, 2\]"),
    (dict(cell=torch.eye(3, dtype=torch.float64)), TypeError,
     r"type of `cell` \(torch.float64\) must be same as that of the `positions` class \(torch.float32\)"),
    (dict(charges=torch.ones(4)), ValueError,
     r"`charges` must be a 2-dimensional tensor, got tensor with 1 dimension\(s\) and shape \[4\]"),
    (dict(charges=torch.ones((6, 2))), ValueError,
     r"`charges` must be a tensor with shape \[n_atoms, n_channels\], with `n_atoms` being the same as the variable "
     r"`positions`. Got tensor with shape \[6, 2\] where positions contains 4 atoms"),
    (dict(charges=torch.ones((4, 2), dtype=torch.float64)), TypeError,
     r"type of `charges` \(torch.float64\) must be same as that of the `positions` class \(torch.float32\)"),
    (dict(neighbor_indices=torch.zeros((3, 3), dtype=torch.long)), ValueError,
     r"neighbor_indices is expected to have shape \[num_neighbors, 2\], but got \[3, 3\] for one structure"),
    (dict(neighbor_distances=torch.ones(5)), ValueError,
     r"`neighbor_indices` and `neighbor_distances` need to have shapes \[num_neighbors, 2\] and \[num_neighbors\], "
     r"but got \[3, 2\] and \[5\]"),
    (dict(neighbor_distances=torch.ones(3, dtype=torch.float64)), TypeError,
     r"type of `neighbor_distances` \(torch.float64\) must be same as that of the `positions` class \(torch.float32\)"),
    (dict(periodic=torch.ones(2, dtype=torch.bool)), ValueError,
     r"`periodic` must be a tensor of shape \(3,\), got tensor with shape \[2\]"),
    (dict(pair_mask=torch.ones(5, dtype=torch.bool)), ValueError,
     r"`pair_mask` must have the same shape as the number of neighbors, got tensor with shape \[5\] while the number "
     r"of neighbors is 3"),
    (dict(pair_mask=torch.ones(3)), TypeError, r"type of `pair_mask` \(torch.float32\) must be torch.bool"),
    (dict(node_mask=torch.ones(5, dtype=torch.bool)), ValueError,
     r"`node_mask` must have shape \[n_atoms\], got tensor with shape \[5\] where n_atoms is 4"),
    (dict(node_mask=torch.ones(4)), TypeError, r"type of `node_mask` \(torch.float32\) must be torch.bool"),
    (dict(kvectors=torch.ones((7, 2))), ValueError,
     r"`kvectors` must be a tensor of shape \[n_kvecs, 3\], got tensor with shape \[7, 2\]"),
    (dict(kvectors=torch.ones((7, 3), dtype=torch.float64)), TypeError,
     r"type of `kvectors` \(torch.float64\) must be same as that of the `positions` class \(torch.float32\)"),
]


@pytest.mark.parametrize("override,exc,msg", VALIDATION_CASES)
def test_validation_messages(override, exc, msg):
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.5)
    args = _args()
    args.update(override)
    with pytest.raises(exc, match=msg):
        calc(**args)


def test_validation_device_message():
    calc = tpa.Calculator(tpa.CoulombPotential())
    args = _args()
    args["cell"] = torch.eye(3, device="meta")
    with pytest.raises(ValueError, match=r"device of `cell` \(meta\) must be same as that of the `positions` class \(cpu\)"):
        calc(**args)


# ---- neighbour list (reference tests/helpers.py:240-275, third-party vesin there) ----
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("cutoff", [1.5, 3.2])
@pytest.mark.parametrize("periodic", [(True, True, True), (True, True, False), (False, True, False), (False, False, False)])
def test_neighbor_list_vs_bruteforce(full, cutoff, periodic):
    rng = np.random.default_rng(11)
    cell = np.array([[3.0, 0, 0], [0.6, 2.5, 0], [-0.4, 0.3, 2.8]])
    pos = rng.uniform(-2, 5, (9, 3))  # some atoms outside the cell; cutoff > L/2 -> several images
    a = neighbor_list(pos, cell, cutoff, full_list=full, periodic=periodic)
    b = neighbor_list_bruteforce(pos, cell, cutoff, full_list=full, periodic=periodic)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-13)
    if not full:
        f = neighbor_list(pos, cell, cutoff, full_list=True, periodic=periodic)
        assert len(f[0]) == 2 * len(a[0])


def test_workload_shapes():
    from torchpme_amd import ops, workloads

    w = workloads.ionic_box(n_side=6, n_mesh=16, cutoff=5.0)
    assert w.n_atoms == 216 and w.pairs.shape == (w.n_pairs, 2) and w.shifts.shape == (w.n_pairs, 3)
    assert ops.ns_mesh_from_cell(w.cell, w.mesh_spacing) == (16, 16, 16)
    assert abs(w.charges.sum()) < 1e-9


def test_dispatcher_ops_and_specs(tmp_path):
    """SURVEY 8f rank 4 on the host: the calculators describe themselves as JSON specs that rebuild them, the TorchScript
    front end scripts / saves / loads, and the dispatcher ops propagate shapes on fake tensors (no kernel runs here)."""
    import json

    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    import torchpme_amd as tpa
    from torchpme_amd import library

    calcs = [
        tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.2, prefactor=14.4), mesh_spacing=0.7, interpolation_nodes=5),
        tpa.PMECalculator(tpa.InversePowerLawPotential(exponent=6, smearing=0.9, exclusion_radius=3.0, exclusion_degree=2),
                          mesh_spacing=0.5, interpolation_nodes=7, full_neighbor_list=True),
        tpa.EwaldCalculator(tpa.CoulombPotential(smearing=1.0), lr_wavelength=2.5),
        tpa.Calculator(tpa.CoulombPotential()),
    ]
    for calc in calcs:
        spec = calc._spec_str
        d = json.loads(spec)
        assert d["class"] == type(calc).__name__
        twin = library.calculator_from_spec(spec, torch.float64, torch.device("cpu"))
        assert type(twin) is type(calc) and library.calculator_spec(twin) == spec
        assert library.calculator_from_spec(spec, torch.float64, torch.device("cpu")) is twin  # cached
        scripted = torch.jit.script(calc.scriptable())
        scripted.save(str(tmp_path / "c.pt"))
        assert torch.jit.load(str(tmp_path / "c.pt")).spec == spec
        # and the reference's spelling, torch.jit.script(calculator) (tests/calculators/test_workflow.py:136-162), through the
        # __prepare_scriptable__ hook
        direct = torch.jit.script(calc)
        assert isinstance(direct, torch.jit.ScriptModule) and direct.spec == spec
    with FakeTensorMode():
        q, cell, pos = torch.empty((7, 2), device="cuda"), torch.empty((3, 3), device="cuda"), torch.empty((7, 3), device="cuda")
        pairs, shifts = torch.empty((11, 2), dtype=torch.int64, device="cuda"), torch.empty((11, 3), device="cuda")
        d = torch.ops.mipme.pair_distances(pos, pairs, cell, shifts)
        assert d.shape == (11,) and d.dtype == pos.dtype
        V = torch.ops.mipme.potentials(q, cell, pos, pairs, d, None, None, None, None, calcs[0]._spec_str)
        assert V.shape == (7, 2)

    class Custom(tpa.Potential):
        pass

    assert tpa.Calculator(Custom())._spec_str is None  # no dispatcher op for potentials the library cannot rebuild


def _header_structs():
    """{struct name: [field names in order]} parsed from include/mipme.h (typedef struct ... { ... } name;)."""
    hdr = open(os.path.join(ROOT, "include", "mipme.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct(?:\s+\w+)?\s*\{(.*?)\}\s*(\w+)\s*;", hdr, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = re.search(r"(\w+)\s*(?:\[\d+\])?\s*$", names[0].strip())
            fields.append(first.group(1))
            for extra in names[1:]:
                fields.append(re.search(r"(\w+)\s*(?:\[\d+\])?\s*$", extra.strip()).group(1))
        out[name] = fields
    return out


def test_ctypes_structs_mirror_the_header_field_by_field():
    """Every struct that crosses the C-ABI has the same fields, in the same order, in include/mipme.h and in _lib.py -- and
    the versioned argument structs refuse a caller compiled against another layout (round-1 verdict: positional 23-pointer
    calls drifted from the header without any error)."""
    import ctypes as C

    structs = _header_structs()
    pairs = {
        "mipme_potential_t": _lib.PotentialDesc, "mipme_mesh_t": _lib.MeshDesc, "mipme_sr_job_t": _lib.SrJob,
        "mipme_frame_t": _lib.Frame, "mipme_nl_t": _lib.NlDesc,
        "mipme_kspace_forward_args_t": _lib.KspaceForwardArgs, "mipme_kspace_backward_args_t": _lib.KspaceBackwardArgs,
    }
    assert set(pairs) <= set(structs), sorted(structs)
    for cname, cls in pairs.items():
        assert [n for n, _ in cls._fields_] == structs[cname], cname
    a = _lib.KspaceForwardArgs(n_atoms=5)
    assert a.size == C.sizeof(_lib.KspaceForwardArgs) and a.version == _lib.ARGS_VERSION and a.n_atoms == 5
    with pytest.raises(TypeError, match="unknown field"):
        _lib.KspaceForwardArgs(no_such_field=1)
    lib = _lib.load()
    a.version = 1
    assert lib.mipme_kspace_forward(C.byref(a)) == -1 and b"version" in lib.mipme_last_error()
    b = _lib.KspaceBackwardArgs()
    b.size = 8
    assert lib.mipme_kspace_backward(C.byref(b)) == -1 and b"size" in lib.mipme_last_error()
    assert lib.mipme_kspace_forward(None) == -1


# ---- copies, pickles, checkpoints (round-1 advisor: deepcopy / torch.save of a used calculator raised) ----
def test_calculator_copies_pickles_and_reference_checkpoints():
    import copy
    import ctypes as C
    import io
    import pickle
    import weakref

    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.2, prefactor=3.0), mesh_spacing=0.4, interpolation_nodes=5)
    cell = torch.eye(3)
    # the state a forward pass leaves behind: a weak reference to the caller's cell, FFT plans holding raw device pointers,
    # the pinned NaN flag
    calc._cache = (weakref.ref(cell), 0, torch.float32, "cuda", None, None, None, None)
    calc._plan_store = {"key": C.c_void_p(1234)}
    calc._nan_flag = torch.zeros(1, dtype=torch.int32)
    for clone in (copy.deepcopy(calc), pickle.loads(pickle.dumps(calc))):
        assert clone._cache is None and clone._plan_store == {} and clone._nan_flag is None
        assert clone.interpolation_nodes == 5 and clone.mesh_spacing == 0.4
        assert float(clone.potential.smearing) == 1.2 and float(clone.potential.prefactor) == 3.0
    assert calc._plan_store and calc._cache is not None  # the original keeps its state
    buf = io.BytesIO()
    torch.save(calc, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    assert type(again) is tpa.P3MCalculator and again._plan_store == {}
    ew = tpa.EwaldCalculator(tpa.CoulombPotential(smearing=1.0), lr_wavelength=0.5)
    ew._freq_cache = (weakref.ref(cell), 0, "cuda", 0.5, cell)
    assert copy.deepcopy(ew)._freq_cache is None
    # a checkpoint written by the reference's P3MCalculator / PMECalculator: extra kspace_filter.* entries, strict load
    ref_sd = {
        "potential.smearing": torch.tensor(0.7, dtype=torch.float64), "potential.prefactor": torch.tensor(2.0, dtype=torch.float64),
        "kspace_filter._diff_coeff": torch.zeros(6, 6), "kspace_filter.kernel.smearing": torch.tensor(0.7, dtype=torch.float64),
        "kspace_filter.kernel.prefactor": torch.tensor(2.0, dtype=torch.float64),
    }
    res = calc.load_state_dict(ref_sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert float(calc.potential.smearing) == 0.7 and calc.potential._descriptor().smearing == 0.7
    with pytest.raises(RuntimeError, match="Unexpected key"):
        calc.load_state_dict({**ref_sd, "something.else": torch.zeros(1)}, strict=True)
    # wrapped in a parent module the prefix logic still applies
    parent = torch.nn.ModuleDict({"lr": tpa.PMECalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.5)})
    parent.load_state_dict({"lr." + k: v for k, v in ref_sd.items() if "_diff_coeff" not in k}, strict=True)
    assert float(parent["lr"].potential.prefactor) == 2.0


def test_potential_closed_forms_match_scipy():
    """Python-side potential methods (inspection API): the lower incomplete gamma without cancellation at small distances
    (advisor: 1 - Q lost 1.6e-7 at d = 0.05 for p = 6), the Fourier kernels of p = 3, 5 through E1, the 2-D slab term."""
    import scipy.special as sp

    d = torch.tensor([1e-3, 0.05, 0.3, 1.0, 2.5, 7.0], dtype=torch.float64)
    sm = 1.3
    for p in range(1, 7):
        pot = tpa.InversePowerLawPotential(exponent=p, smearing=sm)
        ref = sp.gammainc(0.5 * p, (0.5 * d * d / sm**2).numpy()) / d.numpy() ** p
        np.testing.assert_allclose(pot.lr_from_dist(d).numpy(), ref, rtol=5e-14)
    k2 = torch.tensor([0.0, 1e-6, 0.01, 0.5, 1.0, 3.0, 20.0], dtype=torch.float64)
    z = (0.5 * sm**2 * k2[1:]).numpy()
    for p, f in ((3, sp.exp1(z)), (5, np.exp(-z) - z * sp.exp1(z))):
        c0 = np.pi**1.5 / math.gamma(0.5 * p) * (2 * sm**2) ** (0.5 * (3 - p))
        got = tpa.InversePowerLawPotential(exponent=p, smearing=sm).lr_from_k_sq(k2).numpy()
        np.testing.assert_allclose(got[1:], c0 * f, rtol=1e-12)
        assert got[0] == (0.0 if p == 3 else -c0 / (0.5 * (3 - p)))
    # slab term of CoulombPotential.pbc_correction against the formula of coulomb.py:6-40 written out with numpy
    rng = np.random.default_rng(0)
    pos, q = rng.uniform(0, 5, (7, 3)), rng.normal(size=(7, 2))
    cell = np.array([[5.0, 0, 0], [1, 6, 0], [0.3, 0.2, 7]])
    pot = tpa.CoulombPotential(smearing=1.0, prefactor=2.0)
    got = pot.pbc_correction(torch.tensor([True, False, True]), torch.tensor(pos), torch.tensor(cell), torch.tensor(q)).numpy()
    z1 = pos[:, 1:2]
    Q, M, M2 = q.sum(0), (q * z1).sum(0), (q * z1 * z1).sum(0)
    want = 2.0 * 4 * np.pi / abs(np.linalg.det(cell)) * (z1 * M - 0.5 * (M2 + Q * z1 * z1) - Q * np.linalg.norm(cell[1]) ** 2 / 12)
    np.testing.assert_allclose(got, want, rtol=1e-13)
    assert float(pot.pbc_correction(torch.tensor([True, True, True]), torch.tensor(pos), torch.tensor(cell), torch.tensor(q)).abs().max()) == 0.0


def test_integration_snippets_match_the_header():
    """Every ``lib.mipme_*(...)`` call and every argument-struct field shown in INTEGRATION.md exists in include/mipme.h with
    the same number of arguments / the same field list (round 1: two snippets had drifted from the header unnoticed)."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mipme.h")).read(), flags=re.S)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = "\n".join(re.findall(r"```python\n(.*?)```", doc, flags=re.S))
    code = re.sub(r"#.*", "", code)  # comments may contain commas and parentheses

    def split_args(text):  # top-level commas of "a, f(b, c), d"
        out, depth, cur = [], 0, ""
        for ch in text:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        return [a for a in out + [cur] if a.strip()]

    protos = {}
    for name, params in re.findall(r"\b(mipme_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        protos[name] = 0 if params.strip() in ("", "void") else len(split_args(params))
    calls = 0
    for m in re.finditer(r"lib\.(mipme_[a-z0-9_]+)\(", code):
        name = m.group(1)
        if name.endswith(("restype", "argtypes")) or code[m.end() - 1] != "(":
            continue
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(code[i], 0)
            i += 1
        args = code[m.end():i - 1]
        if args.strip() == "...":
            continue
        assert name in protos, f"INTEGRATION.md calls {name}, which include/mipme.h does not declare"
        assert len(split_args(args)) == protos[name], (name, len(split_args(args)), protos[name])
        calls += 1
    assert calls >= 6
    structs = _header_structs()
    for cls, cname in (("KspaceForwardArgs", "mipme_kspace_forward_args_t"), ("KspaceBackwardArgs", "mipme_kspace_backward_args_t"),
                       ("Potential", "mipme_potential_t"), ("Mesh", "mipme_mesh_t")):
        body = re.search(r"class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n" % cls, code, flags=re.S).group(1)
        assert re.findall(r'\("(\w+)"', body) == structs[cname], cls
    # keyword construction of the argument structs only uses declared fields
    for cls, cname in (("KspaceForwardArgs", "mipme_kspace_forward_args_t"), ("KspaceBackwardArgs", "mipme_kspace_backward_args_t")):
        for m in re.finditer(r"= %s\(" % cls, code):
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(code[i], 0)
                i += 1
            for arg in split_args(code[m.end():i - 1]):
                key = arg.split("=", 1)[0].strip()
                assert key in structs[cname], (cls, key)


def test_generate_kvectors_for_mesh_matches_the_reference_golden():
    """lib.generate_kvectors_for_mesh (reference lib/kvectors.py:77-102): the half-grid k-vectors of the conventions fixture
    (made by the reference: triclinic cell, ns = [8, 8, 16]), shape (nx, ny, nz/2+1, 3), and the survey's known value
    k[1,0,0]; differentiable w.r.t. the cell; KSpaceKernel is the reference's interface class."""
    import numpy as np
    import torch

    from torchpme_amd import lib

    z = np.load(os.path.join(ROOT, "tests", "golden", "conventions.npz"))
    cell = torch.tensor(z["cell"], dtype=torch.float64, requires_grad=True)
    ns = torch.tensor(z["ns"])
    k = lib.generate_kvectors_for_mesh(cell, ns)
    assert tuple(k.shape) == (int(ns[0]), int(ns[1]), int(ns[2]) // 2 + 1, 3)
    np.testing.assert_allclose(k.detach().numpy(), z["kvectors"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(k[1, 0, 0].detach().numpy(), [1.5707963, -0.15707963, -0.089011792], atol=1e-7)
    (k * k).sum().backward()
    assert cell.grad is not None and bool(torch.isfinite(cell.grad).all())
    with pytest.raises(ValueError, match=r"cell of shape \[3\] should be of shape \(3, 3\)"):
        lib.generate_kvectors_for_mesh(torch.zeros(3), ns)
    with pytest.raises(ValueError, match=r"ns of shape \[2\] should be of shape \(3, \)"):
        lib.generate_kvectors_for_mesh(cell.detach(), torch.tensor([2, 2]))
    with pytest.raises(NotImplementedError, match="kernel_from_k_sq is not implemented for 'KSpaceKernel'"):
        lib.KSpaceKernel().kernel_from_k_sq(torch.zeros(2))

"""Host-side tests of ``torchpme_amd.tuning`` (no GPU): the a-priori error estimates and the smearing estimate against
vectors produced by the reference (``tests/golden/tuning.npz`` <- ``make_golden.py``), and the argument checks of the
tuners, modelled on the reference's ``tests/tuning/test_error_bounds.py`` / ``test_tuning.py``."""

import numpy as np
import pytest
import torch

import torchpme_amd as tpa
from torchpme_amd.tuning import (EwaldErrorBounds, GridSearchTuner, P3MErrorBounds, PMEErrorBounds, TunerBase, tune_ewald,
                                 tune_p3m, tune_pme)


@pytest.mark.parametrize("name", ["pair", "tri"])
def test_error_bounds_match_reference(golden_dir, name):
    z = np.load(f"{golden_dir}/tuning.npz")
    q, cell, pos = (torch.tensor(z[f"{name}/{k}"]) for k in ("charges", "cell", "positions"))
    p3m, pme = P3MErrorBounds(q, cell, pos), PMEErrorBounds(q, cell, pos)
    # the reference wraps the hyper-parameters in float32 tensors (tuning/p3m.py:313-316), so its numbers carry fp32
    # rounding of smearing / mesh_spacing / cutoff; the restatement is plain double arithmetic
    for smearing, h, rc, nodes, e_p3m, e_pme in z[f"{name}/bounds"]:
        kw = dict(smearing=float(smearing), mesh_spacing=float(h), cutoff=float(rc), interpolation_nodes=int(nodes))
        assert float(p3m(**kw)) == pytest.approx(e_p3m, rel=5e-6)
        if not np.isnan(e_pme):
            assert float(pme(**kw)) == pytest.approx(e_pme, rel=5e-6)
    for rc, acc, sm in z[f"{name}/smearing"]:
        assert TunerBase(q, cell, pos, float(rc), None).estimate_smearing(float(acc)) == pytest.approx(sm, rel=1e-12)


def test_error_bounds_known_values():
    """The numbers quoted in the reference's docstrings / tests (tuning/pme.py:170-178, tuner.py:74-77)."""
    charges = torch.tensor([[1.0], [-1.0]])
    cell = torch.eye(3)
    positions = torch.tensor([[0.0, 0.0, 0.0], [0.4, 0.4, 0.4]])
    kw = dict(smearing=1.0, mesh_spacing=0.5, cutoff=4.4, interpolation_nodes=3)
    out = PMEErrorBounds(charges, cell, positions)(**kw)
    assert isinstance(out, torch.Tensor) and out.dtype == torch.float32
    torch.testing.assert_close(out, torch.tensor(0.0011180))
    torch.testing.assert_close(EwaldErrorBounds(charges, cell, positions)(smearing=1.0, lr_wavelength=0.5, cutoff=4.4),
                               torch.tensor(8.4304e-05))  # tests/tuning/test_error_bounds.py:12-16
    assert float(P3MErrorBounds(charges, cell, positions)(**kw)) == pytest.approx(4.5968e-4, rel=1e-4)
    assert TunerBase(charges, cell, positions, 4.4, None).estimate_smearing(1e-3) == pytest.approx(1.1069526756106463)
    # the real-space part is shared and the total is the root of the sum of squares
    b = P3MErrorBounds(charges, cell, positions)
    assert float(b(**kw)) == pytest.approx(np.hypot(b.err_kspace(1.0, 0.5, 3), b.err_rspace(1.0, 4.4)))


def system():
    return torch.ones((4, 1)), torch.eye(3), 0.3 * torch.arange(12, dtype=torch.float32).reshape((4, 3))


@pytest.mark.parametrize("tune", [tune_ewald, tune_pme, tune_p3m])
def test_tuner_argument_errors(tune):
    charges, cell, positions = system()
    pairs, dist = torch.tensor([[0, 1]]), torch.tensor([0.5])
    with pytest.raises(ValueError, match="'foo' is not a float."):
        TunerBase(charges, cell, positions, 4.4, None).estimate_smearing("foo")
    with pytest.raises(NotImplementedError, match="Only exponent = 1 is supported but got 2."):
        tune(charges=charges, cell=cell, positions=positions, cutoff=4.4, neighbor_indices=pairs, neighbor_distances=dist,
             exponent=2)
    with pytest.raises(ValueError, match=r"`positions` must be a tensor with shape \[n_atoms, 3\], got tensor with shape \[4, 5\]"):
        tune(charges=charges, cell=cell, positions=torch.ones((4, 5)), cutoff=4.4, neighbor_indices=None,
             neighbor_distances=None)
    with pytest.raises(ValueError, match=r"`cell` must be a tensor with shape \[3, 3\], got tensor with shape \[2, 2\]"):
        tune(charges=charges, cell=torch.ones([2, 2]), positions=positions, cutoff=4.4, neighbor_indices=None,
             neighbor_distances=None)
    with pytest.raises(TypeError, match=r"type of `cell` \(torch.float64\) must be same as that of the `positions` class \(torch.float32\)"):
        tune(charges=charges, cell=torch.eye(3, dtype=torch.float64), positions=positions, cutoff=4.4,
             neighbor_indices=None, neighbor_distances=None)


def test_filter_neighbors_and_grid():
    d = torch.tensor([0.5, 4.5, 4.39, 7.0])
    idx = torch.tensor([[0, 1], [0, 2], [1, 2], [2, 3]])
    fi, fd = TunerBase.filter_neighbors(4.4, idx, d)
    assert fd.tolist() == pytest.approx([0.5, 4.39]) and fi.tolist() == [[0, 1], [1, 2]]
    # errors of the grid are computed on the host; nothing is timed when no candidate meets the accuracy
    charges, cell, positions = system()
    params = [dict(interpolation_nodes=3, mesh_spacing=0.5), dict(interpolation_nodes=5, mesh_spacing=0.1)]
    tuner = GridSearchTuner(charges, cell, positions, 4.4, tpa.P3MCalculator, P3MErrorBounds(charges, cell, positions),
                            params, idx, d)
    errs, timings = tuner.tune(1e-12)
    assert len(errs) == 2 and errs[1] < errs[0] and timings == [float("inf")] * 2
    with pytest.raises(ValueError, match="'1' is not a float."):
        tuner.tune(1)

"""metatensor / metatomic adapter (reference ``metatensor/calculator.py:22-188``, its tests
``tests/metatensor/test_calculator_metatensor.py``): argument checks on CPU, values and gradients on the GPU against the
plain calculators.  The two packages are not installed here: stand-ins with the handful of attributes the adapter uses are
registered under their module names (tests/_metatensor_standins.py)."""

import numpy as np
import pytest
import torch

from tests import _metatensor_standins as S

S.install()

import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import metatensor as tpm  # noqa: E402


def _cscl(device="cpu", dtype=torch.float64):
    pos = torch.tensor([[0.0, 0, 0], [0.5, 0.5, 0.5]], dtype=dtype, device=device)
    cell = torch.eye(3, dtype=dtype, device=device)
    q = torch.tensor([[1.0], [-1.0]], dtype=dtype, device=device)
    pairs, shifts, _ = tpa.neighbor_list(pos.cpu().numpy(), cell.cpu().numpy(), 1.2)
    return pos, cell, q, torch.tensor(pairs, device=device), torch.tensor(shifts, device=device)


def test_adapter_argument_checks():
    pos, cell, q, pairs, shifts = _cscl()
    calc = tpm.P3MCalculator(tpa.CoulombPotential(smearing=0.3), mesh_spacing=0.1)
    assert isinstance(calc._calculator, tpa.P3MCalculator) and not calc.fuse_distances
    assert isinstance(tpm.PMECalculator(tpa.CoulombPotential(smearing=0.3), mesh_spacing=0.1)._calculator, tpa.PMECalculator)
    assert isinstance(tpm.EwaldCalculator(tpa.CoulombPotential(smearing=0.3), lr_wavelength=0.2)._calculator, tpa.EwaldCalculator)
    system, nl = S.make_system(pos, cell, q), S.make_neighbors(pos, cell, pairs, shifts)
    bad = S.make_neighbors(pos.float(), cell.float(), pairs, shifts)
    with pytest.raises(ValueError, match=r"dtype of `neighbors` \(torch.float32\) must be the same as `system` \(torch.float64\)"):
        calc(system, bad)
    bad = S.make_neighbors(pos, cell, pairs, shifts)
    bad.samples.names[0] = "atom_i"
    with pytest.raises(ValueError, match="Invalid samples for `neighbors`: the sample names must be 'first_atom'"):
        calc(system, bad)
    bad = S.make_neighbors(pos, cell, pairs, shifts)
    bad.components = []
    with pytest.raises(ValueError, match="Invalid components for `neighbors`: there should be a single 'xyz'"):
        calc(system, bad)
    bad = S.make_neighbors(pos, cell, pairs, shifts)
    bad.properties = S.Labels(["energy"], torch.zeros(1, 1, dtype=torch.int32))
    with pytest.raises(ValueError, match="Invalid properties for `neighbors`: there should be a single 'distance'=0 property"):
        calc(system, bad)
    empty = S.System(torch.ones(2, dtype=torch.int32), pos, cell)
    with pytest.raises(ValueError, match="`system` does not contain `charge` data"):
        calc(empty, nl)
    two = S.make_system(pos, cell, q)
    two.get_data("charge")._blocks.append(two.get_data("charge").block())
    with pytest.raises(ValueError, match="Charge tensor have exactlty one block but has 2 blocks"):
        calc(two, nl)
    comp = S.make_system(pos, cell, q)
    comp.get_data("charge").block().components = [S.Labels(["xyz"], torch.zeros(1, 1, dtype=torch.int32))]
    with pytest.raises(ValueError, match="TensorBlock containg the charges should not have components; found 1"):
        calc(comp, nl)
    with pytest.raises(tpa.MipmeError, match="no CPU fallback"):  # valid arguments reach the HIP path, which needs a GPU
        calc(system, nl)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("name", ["p3m", "pme", "ewald"])
def test_adapter_matches_plain_calculator(name, fuse):
    dev = "cuda"
    rng = np.random.default_rng(5)
    cell_np = np.array([[7.0, 0, 0], [0.6, 6.5, 0], [0.1, -0.4, 7.5]])
    pos_np = rng.uniform(0, 7, (60, 3))
    q_np = rng.normal(size=(60, 2))
    pairs_np, S_np, _ = tpa.neighbor_list(pos_np, cell_np, 3.5)
    pot = tpa.CoulombPotential(smearing=1.0)
    make = {"p3m": lambda M: M.P3MCalculator(pot, mesh_spacing=0.2),
            "pme": lambda M: M.PMECalculator(pot, mesh_spacing=0.2, interpolation_nodes=5),
            "ewald": lambda M: M.EwaldCalculator(pot, lr_wavelength=1.5)}[name]
    t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    pairs, shifts, q = t(pairs_np), t(S_np), t(q_np)
    res = []
    for adapter in (True, False):
        pos, cell = t(pos_np).requires_grad_(True), t(cell_np).requires_grad_(True)
        if adapter:
            calc = make(tpm)
            calc.fuse_distances = fuse
            out = calc(S.make_system(pos, cell, q), S.make_neighbors(pos, cell, pairs, shifts))
            assert len(out) == 1 and out.keys.names == ["_"]
            block = out.block()
            assert block.samples.names == ["system", "atom"] and block.properties.names == ["charges_channel"]
            assert block.components == [] and block.samples.values.shape == (60, 2)
            assert block.samples.values[:, 1].tolist() == list(range(60)) and len(block.properties) == 2
            V = block.values
        else:
            d = tpa.pair_distances(pos, pairs, cell, shifts)
            V = make(tpa)(q, cell, pos, pairs, d)
        (V * q).sum().backward()
        res.append((V.detach(), pos.grad, cell.grad))
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-10)

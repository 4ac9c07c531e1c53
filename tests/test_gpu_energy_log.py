"""The frame farm's energy log (graphed.EnergyLog, mipme_energy_log_push): every replay of a graphed step
appends its frame energies to a device-resident log -- the last node of the captured graph --, so that a rank streams batch
after batch through its GPU and exchanges the log once."""

import numpy as np
import pytest
import torch

import torchpme_amd as tpa

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _frame(rng, dtype, n_side=4, a=2.4):
    L = n_side * a
    g = (np.arange(n_side) + 0.5) * a
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.3, 0.3, (n_side**3, 3))
    q = rng.normal(size=(len(pos), 1))
    q -= q.mean()
    cell = L * np.eye(3)
    pairs, S, _ = tpa.neighbor_list(pos, cell, 4.0)
    t = lambda x: torch.tensor(x, device=DEV, dtype=dtype)  # noqa: E731
    return t(q), t(cell), t(pos), torch.tensor(pairs, device=DEV), t(S)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_push_wraps_and_counts(dtype):
    log = tpa.EnergyLog(3, 2, DEV)
    for k in range(5):
        log.push(torch.tensor([k + 0.25, -k - 0.5], device=DEV, dtype=dtype))
    assert log.count() == 5
    v = log.values.cpu().numpy()  # slots k mod 3: pushes 3, 4, 2
    np.testing.assert_array_equal(v, [[3.25, -3.5], [4.25, -4.5], [2.25, -2.5]])
    log.reset()
    log.push(torch.tensor([7.0, 8.0], device=DEV, dtype=dtype))
    assert log.count() == 1 and log.values[0].tolist() == [7.0, 8.0]
    with pytest.raises(ValueError):
        log.push(torch.zeros(3, device=DEV, dtype=dtype))
    with pytest.raises(tpa.MipmeError):
        log.push(torch.zeros(2, dtype=dtype))  # a host tensor: no CPU path


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("neighbors", ["list", "stream", "list-no-tail"])
def test_graphed_step_logs_every_replay(dtype, neighbors, monkeypatch):
    """binned step (the gather tail appends: mipme_kspace_forward_args_t.energy_log), live-bin step (mipme_md_args_t.energy_log)
    and a step without the gather tail (one more graph node: mipme_energy_log_push)."""
    from torchpme_amd import ops

    if neighbors == "list-no-tail":
        monkeypatch.setattr(ops, "TAIL_FUSION", False)
        neighbors = "list"
    rng = np.random.default_rng(11)
    q, cell, pos, pairs, S = _frame(rng, dtype)
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.6, interpolation_nodes=5).to(dtype)
    if neighbors == "list":
        step = tpa.GraphedEnergyForces(calc, q, cell, pos, pairs, S, energy_log=4)
        plain = tpa.GraphedEnergyForces(calc, q, cell, pos, pairs, S)
    else:
        step = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=4.0, energy_log=4)
        plain = tpa.GraphedEnergyForces(calc, q, cell, pos, neighbors=4.0)
    assert step.energy_log.count() == 0  # warm-up and capture push nothing
    if neighbors == "list":
        assert bool(step._tail is not None and step._tail.get("logged")) == ops.TAIL_FUSION
    want = []
    for k in range(6):
        p = pos + 0.01 * k * torch.tensor(rng.normal(size=tuple(pos.shape)), device=DEV, dtype=dtype)
        E, F = step(p)
        E0, F0 = plain(p)
        assert float(E) == float(E0) and torch.equal(F, F0)  # the log changes nothing the step returns
        want.append(float(E))
    assert step.energy_log.count() == 6
    v = step.energy_log.values.cpu().numpy()[:, 0]
    np.testing.assert_array_equal(v, [want[4], want[5], want[2], want[3]])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_frame_batch_logs_every_replay(dtype):
    rng = np.random.default_rng(12)
    frames = [_frame(rng, dtype) for _ in range(3)]
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.6, interpolation_nodes=4).to(dtype)
    log = tpa.EnergyLog(8, 3, DEV)
    batch = tpa.GraphedFrameBatch(calc, frames, energy_log=log)
    assert batch.energy_log is log
    want = []
    for k in range(3):
        new = [f[2] + 0.01 * k for f in frames]
        E, _ = batch(new)
        want.append(E.double().cpu().numpy().copy())
    assert log.count() == 3
    np.testing.assert_array_equal(log.values[:3].cpu().numpy(), np.stack(want))
    from torchpme_amd import farm

    out = farm.gather_energy_log(log.values[:3])  # no process group: the identity with a leading rank axis
    assert out.shape == (1, 3, 3) and torch.equal(out[0], log.values[:3])
    with pytest.raises(ValueError):
        tpa.GraphedFrameBatch(calc, frames, energy_log=tpa.EnergyLog(8, 2, DEV))


def test_epilogue_is_the_tail_of_the_graph():
    dtype = torch.float64
    rng = np.random.default_rng(13)
    q, cell, pos, pairs, S = _frame(rng, dtype)
    calc = tpa.PMECalculator(tpa.CoulombPotential(smearing=1.0), mesh_spacing=0.6, interpolation_nodes=4).to(dtype)
    twice = torch.zeros((), dtype=dtype, device=DEV)

    def epilogue(step):
        twice.copy_(2 * step.energy)

    step = tpa.GraphedEnergyForces(calc, q, cell, pos, pairs, S, epilogue=epilogue)
    for k in range(2):
        E, _ = step(pos + 0.02 * k)
        assert float(twice) == 2 * float(E)


def test_c_abi_refusals():
    """The log's entry points say no loudly: missing buffers, a capacity of zero, a host table that is too small, a frame batch
    without the gather tail."""
    import ctypes as C

    from torchpme_amd import _lib

    lib = _lib.load()
    log = tpa.EnergyLog(4, 2, DEV)
    src = torch.zeros(2, device=DEV, dtype=torch.float64)
    st = _lib.current_stream(torch.device(DEV))
    f64 = _lib.dtype_code(torch.float64)
    ok = lib.mipme_energy_log_push(st, f64, 2, src.data_ptr(), log.values.data_ptr(), log.cursor.data_ptr(), 4)
    assert ok == 0
    for args in ((st, f64, 0, src.data_ptr(), log.values.data_ptr(), log.cursor.data_ptr(), 4),
                 (st, f64, 2, None, log.values.data_ptr(), log.cursor.data_ptr(), 4),
                 (st, f64, 2, src.data_ptr(), None, log.cursor.data_ptr(), 4),
                 (st, f64, 2, src.data_ptr(), log.values.data_ptr(), None, 4),
                 (st, f64, 2, src.data_ptr(), log.values.data_ptr(), log.cursor.data_ptr(), 0),
                 (st, 77, 2, src.data_ptr(), log.values.data_ptr(), log.cursor.data_ptr(), 4)):
        assert lib.mipme_energy_log_push(*args) != 0 and lib.mipme_last_error()
    torch.cuda.synchronize()
    assert log.count() == 1
    # a host table that is too small / a log without cursors
    nbytes = lib.mipme_frames_table_bytes(f64, 2)
    host = np.zeros((nbytes,), dtype=np.uint8)
    assert lib.mipme_frames_table_energy_log(f64, 2, host.ctypes.data, nbytes - 1, log.values.data_ptr(), log.cursor.data_ptr(), 4) != 0
    assert lib.mipme_frames_table_energy_log(f64, 2, host.ctypes.data, nbytes, log.values.data_ptr(), None, 4) != 0
    # an all-zero table has use_tail = 0 in every frame: the log rides on the gather tail and is refused; switching it off is fine
    assert lib.mipme_frames_table_energy_log(f64, 2, host.ctypes.data, nbytes, log.values.data_ptr(), log.cursor.data_ptr(), 4) != 0
    assert b"gather tail" in lib.mipme_last_error()
    assert lib.mipme_frames_table_energy_log(f64, 2, host.ctypes.data, nbytes, None, None, 0) == 0

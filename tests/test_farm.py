"""Multi-process path (world_size 2, gloo, CPU): frame sharding + the single all-gather of the frame farm.
The per-frame evaluator here is the NumPy oracle (the HIP path needs a GPU); what is under test is the
partitioning and the collective."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from torchpme_amd import farm


def test_frame_blocks_cover_everything():
    for n, w in [(64, 8), (10, 4), (3, 8), (0, 2), (7, 1)]:
        seen = []
        for r in range(w):
            seen += list(farm.frame_block(n, r, w))
        assert seen == list(range(n))
        sizes = [len(farm.frame_block(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    assert [farm.frame_owner(f, 64, 8) for f in (0, 7, 8, 63)] == [0, 0, 1, 7]
    with pytest.raises(ValueError):
        farm.frame_block(4, 5, 2)


def _frame_energy(f: int) -> torch.Tensor:
    from oracle import pme_numpy as O
    from torchpme_amd.workloads import ionic_box

    w = ionic_box(n_side=3, n_mesh=8, cutoff=3.0, seed=100 + f)
    spec = O.PotentialSpec("coulomb", 1, w.smearing, 1.0)
    dist_, _ = O.pair_distances(w.positions, w.cell, w.pairs, w.shifts)
    V = O.forward(spec, "P3M", 3, w.mesh_spacing, w.charges, w.cell, w.positions, w.pairs, dist_)
    return torch.tensor(float((V * w.charges).sum()), dtype=torch.float64)


def _worker(rank, world, port, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def ev(f):
            calls.append(f)
            return _frame_energy(f)

        e = farm.farm_energies(n_frames, ev)
        assert calls == list(farm.frame_block(n_frames, rank, world))
        np.save(os.path.join(out_dir, f"e{rank}.npy"), e.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [4, 5])
def test_farm_two_ranks_gloo(tmp_path, n_frames):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    e0, e1 = np.load(tmp_path / "e0.npy"), np.load(tmp_path / "e1.npy")
    serial = farm.farm_energies(n_frames, _frame_energy).numpy()
    np.testing.assert_array_equal(e0, e1)
    np.testing.assert_allclose(e0, serial, rtol=0, atol=0)


def _frame_energy_forces(f: int):
    """(energy, (N_f, 4) potentials | pseudo-forces) of a frame whose size depends on the index."""
    n = 5 + (f % 3)
    g = torch.Generator().manual_seed(1000 + f)
    arr = torch.rand((n, 4), generator=g, dtype=torch.float64)
    return arr.sum(), arr


def _worker_forces(rank, world, port, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [5 + (f % 3) for f in range(n_frames)]
        e, arrays = farm.farm_energies_forces(sizes, _frame_energy_forces)
        np.save(os.path.join(out_dir, f"e{rank}.npy"), e.numpy())
        np.save(os.path.join(out_dir, f"f{rank}.npy"), torch.cat(arrays).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,world", [(5, 2), (2, 3), (17, 8), (5, 8)])
def test_farm_gathers_per_atom_results(tmp_path, n_frames, world):
    """the optional gather of per-atom potentials + forces -- frames of different sizes, a rank without frames;
    the node's shape (8 ranks) with an uneven frame count (17 frames: blocks of 3 and 2) and with ranks that have none (5
    frames on 8 ranks)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_forces, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    sizes = [5 + (f % 3) for f in range(n_frames)]
    e_ser, arr_ser = farm.farm_energies_forces(sizes, _frame_energy_forces)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"e{r}.npy"), e_ser.numpy())
        np.testing.assert_array_equal(np.load(tmp_path / f"f{r}.npy"), torch.cat(arr_ser).numpy())
    for f, a in enumerate(arr_ser):
        assert a.shape == (sizes[f], 4)


def _log_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # K = 3 evaluations of F = 2 frames per rank: value = 100 rank + 10 k + f
        k, f = torch.meshgrid(torch.arange(3), torch.arange(2), indexing="ij")
        log = (100.0 * rank + 10.0 * k + f).to(torch.float64)
        out = farm.gather_energy_log(log)
        np.save(os.path.join(out_dir, f"log{rank}.npy"), out.numpy())
    finally:
        dist.destroy_process_group()


def test_energy_log_crosses_in_one_collective(tmp_path):
    """``gather_energy_log``: K evaluations x F frames per rank, ONE all-gather; every rank ends up with every
    rank's log in rank order.  Without a process group it is the identity with a leading axis of one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_log_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    l0, l1 = np.load(tmp_path / "log0.npy"), np.load(tmp_path / "log1.npy")
    np.testing.assert_array_equal(l0, l1)
    assert l0.shape == (2, 3, 2)
    for r in range(2):
        for k in range(3):
            np.testing.assert_array_equal(l0[r, k], [100.0 * r + 10.0 * k, 100.0 * r + 10.0 * k + 1])
    alone = farm.gather_energy_log(torch.arange(6.0, dtype=torch.float64).reshape(3, 2))
    assert alone.shape == (1, 3, 2) and float(alone[0, 2, 1]) == 5.0
    with pytest.raises(ValueError):
        farm.gather_energy_log(torch.zeros(4))

"""``bench.py --gpus N`` must produce N ranks by itself (round-1 verdict: the flag was parsed and ignored).

Runs the real entry point on CPU: ``--stub-evaluator`` swaps the HIP frame evaluation for a dot product and the
``nccl`` backend for ``gloo``; everything else -- the re-exec under ``torch.distributed.run``, the rendezvous on
127.0.0.1, the barrier-bracketed timed loop, the all-gather inside it, MAX over ranks, the single JSON line of
rank 0 -- is the code path the GPU ranks take."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags, "--stub-evaluator"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_spawns_ranks():
    out = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--frames-per-gpu", "2")
    assert out["n_gpus"] == 2
    assert out["parallelism"]["n_ranks"] == 2
    assert len(out["parallelism"]["per_rank_ms_per_step"]) == 2
    assert out["energies_gathered"] == 4  # 2 ranks x 2 frames through the one all-gather
    assert out["ms_per_step"] == pytest.approx(max(out["parallelism"]["per_rank_ms_per_step"]), rel=1e-4)  # (the per-rank values are rounded to six decimals in the line)
    assert out["steps"] == 3 and out["warmup"] == 1


def test_exchange_modes_gather_the_same_energies():
    """log (default with > 1 rank, SURVEY 8(e): every step's energies kept in a log, ONE all-gather of the K x frames log per
    timed region), per-step (the all-gather after EVERY evaluation, inside the timed loop), pipelined (asynchronous, two slots)
    and final (only the last step's energies) must deliver the same frame energies; the line carries the protocol."""
    sums = {}
    for mode in ("log", "per-step", "pipelined", "final"):
        out = _run("--gpus", "2", "--steps", "4", "--warmup", "1", "--frames-per-gpu", "2", "--blocks", "2", "--exchange", mode,
                   "--exchange-sweep", "full")
        assert out["parallelism"]["exchange"] == mode and out["energies_gathered"] == 4
        assert out["timing"]["blocks"] == 2 and len(out["timing"]["blocks_ms_per_step"]) == 2
        # (the reported timed region is the median block -- the lower one of two --, the first block is reported next to it)
        assert out["ms_per_step"] == pytest.approx(min(out["timing"]["blocks_ms_per_step"]), rel=1e-4)
        assert out["ms_per_step_first_block"] == pytest.approx(out["timing"]["blocks_ms_per_step"][0], rel=1e-4)
        assert out["weak_efficiency"]["value"] > 0 and out["weak_efficiency"]["one_rank_ms_per_step"] > 0
        assert [r["rank"] for r in out["parallelism"]["ranks"]] == [0, 1]
        cores = [r["affinity"].get("cores") for r in out["parallelism"]["ranks"]]
        assert cores[0] != cores[1] or cores[0] is None  # disjoint core sets when the host has more than one core
        sums[mode] = out["energies_sum"]
        others = out["parallelism"]["other_exchange_modes_ms_per_step"]  # one timed block per other protocol, same invocation
        assert set(others) == {"none", "log", "per-step", "pipelined", "final"} - {mode} and all(v > 0 for v in others.values())
        if mode == "log":  # 2 ranks x 4 steps x 2 frames crossed in ONE collective
            assert out["parallelism"]["log_entries_gathered"] == 16
    assert sums["per-step"] == pytest.approx(sums["final"], rel=1e-12) == pytest.approx(sums["pipelined"], rel=1e-12)
    assert sums["log"] == pytest.approx(sums["final"], rel=1e-12)
    out = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-exchange-sweep")
    assert out["parallelism"]["exchange"] == "log"  # the default with more than one rank
    assert out["parallelism"]["other_exchange_modes_ms_per_step"] == {}


def test_eight_ranks_cfg4_preset():
    """BASELINE.json configs[3] as the driver will launch it on an 8-GPU node -- ``bench.py --gpus 8 --preset cfg4`` -- on eight CPU
    ranks (gloo, stub evaluator): rank binding, the one exchange of the 8 x 2 x 8 logged frame energies inside the timed region and
    ``weak_efficiency`` are all exercised; only the evaluator and the fabric differ from the real run."""
    out = _run("--gpus", "8", "--preset", "cfg4", "--steps", "2", "--warmup", "1", "--blocks", "1")
    assert out["n_gpus"] == 8 and out["parallelism"]["n_ranks"] == 8
    assert out["energies_gathered"] == 64 and out["config"]["frames_per_gpu"] == 8
    assert out["parallelism"]["exchange"] == "log" and out["parallelism"]["log_entries_gathered"] == 8 * 2 * 8
    # the default sweep of the other protocols: blocking collectives only (the asynchronous ones are --exchange-sweep full)
    assert set(out["parallelism"]["other_exchange_modes_ms_per_step"]) == {"none", "per-step", "final"}
    assert len(out["parallelism"]["per_rank_ms_per_step"]) == 8
    assert sorted(r["rank"] for r in out["parallelism"]["ranks"]) == list(range(8))
    assert out["weak_efficiency"]["value"] > 0


def test_single_rank_default():
    out = _run("--steps", "2", "--warmup", "1")
    assert out["n_gpus"] == 1 and out["parallelism"]["n_ranks"] == 1


def test_cfg4_preset():
    import bench

    args = bench.parse_args(["--preset", "cfg4", "--gpus", "8"])
    assert (args.workload, args.frames_per_gpu, args.gpus) == ("ionic", 8, 8)

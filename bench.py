"""Benchmark of the PME/P3M hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload water|ionic|dispersion]

One *step* = one energy + forces evaluation of one frame per GPU, the reference's protocol
(BASELINE.md section 2): ``d = pair_distances(...)`` -> ``V = calculator(...)`` -> ``E = sum(q*V)`` ->
``E.backward()`` (forces = -dE/dpositions).  Inputs are resident in HBM before the timed region.  With N > 1
every rank owns an independent frame (weak scaling, no intra-cell decomposition); the frame energies are exchanged
with ONE RCCL all_gather after the last step (inside the timed region).

Rank 0 prints ONE JSON line: metric = atom-steps/s over all ranks, plus
  roofline     -- HBM roofline of the dominant kernel, timed live with HIP events on the launch stream
  cpu_baseline -- the PyTorch-CPU oracle ("port" of the reference's ATen op sequence) timed on this host's cores
                  on a bounded sample (rank 0, N = 1 only)
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import ops, workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def make_workload(name: str, seed_offset: int):
    if name == "water":
        return workloads.water_box(seed=1234 + seed_offset)
    if name == "ionic":
        return workloads.ionic_box(seed=12 + seed_offset)
    if name == "dispersion":
        return workloads.dispersion_box(seed=8 + seed_offset)
    raise ValueError(name)


class Frame:
    """Device-resident inputs of one frame + its calculator."""

    def __init__(self, w, device):
        self.w = w
        dt = torch.float32 if w.dtype == "f32" else torch.float64
        self.dtype = dt
        self.pos = torch.tensor(w.positions, dtype=dt, device=device, requires_grad=True)
        self.q = torch.tensor(w.charges, dtype=dt, device=device)
        self.cell = torch.tensor(w.cell, dtype=dt, device=device)
        self.pairs = torch.tensor(w.pairs, dtype=torch.int64, device=device)
        self.shifts = torch.tensor(w.shifts, dtype=dt, device=device)
        self.minus_one = torch.tensor(-1.0, dtype=dt, device=device)  # backward seed: pos.grad = -dE/dr = forces
        pot = (tpa.CoulombPotential(smearing=w.smearing) if w.exponent == 1
               else tpa.InversePowerLawPotential(exponent=w.exponent, smearing=w.smearing))
        Calc = tpa.P3MCalculator if w.scheme == "P3M" else tpa.PMECalculator
        self.calc = Calc(pot, mesh_spacing=w.mesh_spacing, interpolation_nodes=w.order)

    def step(self):
        self.pos.grad = None
        # deferred: the distance tensor is written by the calculator's fused distance + pair kernel (by-product of the row that
        # owns a pair's first atom) instead of by a separate pass over the pair list -- same tensor, one kernel less
        d = tpa.pair_distances(self.pos, self.pairs, self.cell, self.shifts, deferred=True)
        self.distances = d.detach()
        V = self.calc(self.q, self.cell, self.pos, self.pairs, d)
        E = tpa.weighted_sum(V, self.q)
        E.backward(self.minus_one)
        return E.detach(), self.pos.grad


def algorithmic_bytes(w, s: int, fused: bool = True):
    """Minimum HBM bytes per energy+forces step (reference data formats) and per kernel launch.

    Per kernel the figure is what the launch must move IN THE FORMAT IT READS, never more than SURVEY's figure for the
    reference formats: the distance kernel streams int32 pairs + one packed shift word + writes d (8 + 4 + s bytes per
    pair instead of 16 + 3s + s), and the fused pair kernel streams two 8-byte entries per pair and reads neither d nor
    dL/dd (16 bytes per pair; SURVEY's three kernels it replaces add up to 48 + 7s).  DESIGN.md section 2 lists both."""
    P, N, M = w.n_pairs, w.n_atoms, w.n_mesh**3
    per_kernel = {
        "pair_distance_forward": (P * (8 + 4 + s) if fused else P * (16 + 3 * s + s)) + N * 3 * s,
        "pair_distance_backward": P * (16 + 3 * s + s) + N * 6 * s,
        "rspace_forward": (2 * P * 8 + P * s + N * 8 * s) if fused else (P * (16 + s) + N * 2 * s),
        "rspace_backward": P * (16 + 2 * s) + N * 3 * s,
        # mesh stages (the meshes themselves are L2 / Infinity-Cache resident at these sizes)
        "spread": N * 4 * s + 2 * M * s,
        # the spread and the fused distance + pair kernel co-scheduled in one launch (mipme_sr_job_t): both byte counts
        "spread+rspace_forward": (2 * P * 8 + P * s + N * 8 * s) + N * 4 * s + 2 * M * s,
        "gather": N * 5 * s + M * s,
        "gather_grad": N * 8 * s + 2 * M * s,
        "fft_r2c": 2 * M * s,
        "fft_c2r": 2 * M * s,
        "apply_filter": int(2.5 * M * s),
        # (y,z) hipFFT planes + one kernel for x-FFT * G * inverse x-FFT: the three stages above in one composite
        "convolve_xfused": int(6.5 * M * s),
        "bin_atoms": N * (3 * s + 8 + 16 + 4 * s),
        # energy reduction E = sum q V, its adjoint, and the energy-mode force assembly gE q_a (f F_a + field_a)
        "energy_sum": N * 2 * s,
        "energy_sum_backward": N * 3 * s,
        "forces_finalize": N * 10 * s,
    }
    step = P * (32 + 3 * s) + P * (32 + 8 * s) + N * 25 * s + 19 * M * s
    return step, per_kernel


def cpu_baseline(w, budget_s: float = 20.0):
    """Time the CPU restatement of the reference's op sequence (``oracle/pme_torch.py``: ATen ops on CPU tensors,
    all host threads, autograd) on this host, with the reference's own timing protocol (``tuning/tuner.py:337-373``:
    warm-up calls, then repeated energy + forces evaluations of the SAME frame, monotonic clock, median)."""
    from oracle import pme_numpy as O
    from oracle import pme_torch as OT

    if w.exponent != 1:  # the torch oracle covers the Coulomb configurations; fall back to the NumPy port
        return cpu_baseline_numpy(w, budget_s)
    dt = torch.float32 if w.dtype == "f32" else torch.float64
    spec = O.PotentialSpec("coulomb", 1, w.smearing, 1.0)
    q, cell, pos, pairs, S = OT.as_tensors(w, dt)
    scheme = "P3M" if w.scheme == "P3M" else "Lagrange"
    times = []
    t_all = time.monotonic()
    n_warm = 2
    while True:
        t0 = time.monotonic()
        E, F = OT.energy_forces_step(spec, scheme, w.order, w.mesh_spacing, q, cell, pos, pairs, S)
        times.append(time.monotonic() - t0)
        if len(times) >= n_warm + 8 or (len(times) > n_warm and time.monotonic() - t_all + times[-1] > budget_s):
            break
    timed = times[n_warm:] if len(times) > n_warm else times[-1:]
    best = float(np.median(timed))
    return {
        "value": w.n_atoms / best,
        "unit": "atom-steps/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{len(timed)} timed (+{len(times) - len(timed)} warm-up) full energy+forces steps of the same "
                  f"{w.n_atoms}-atom frame with oracle/pme_torch.py (PyTorch-CPU ATen ops + autograd, "
                  f"{torch.get_num_threads()} threads of {os.cpu_count()} logical cores), median {best:.3f} s/step, "
                  f"energy {float(E):.4f}",
    }


def cpu_baseline_numpy(w, budget_s: float = 25.0):
    """Single-threaded NumPy port (``oracle/pme_numpy.py``), used for the potentials the torch oracle does not cover."""
    from oracle import pme_numpy as O

    dt = np.float32 if w.dtype == "f32" else np.float64
    spec = O.PotentialSpec("coulomb" if w.exponent == 1 else "ipl", w.exponent, w.smearing, 1.0)
    pos, q, cell = w.positions.astype(dt), w.charges.astype(dt), w.cell.astype(dt)
    scheme = "P3M" if w.scheme == "P3M" else "Lagrange"
    times = []
    t_all = time.monotonic()
    while True:
        t0 = time.monotonic()
        dist, _ = O.pair_distances(pos, cell, w.pairs, w.shifts)
        V, cache = O.forward(spec, scheme, w.order, w.mesh_spacing, q, cell, pos, w.pairs, dist, return_cache=True)
        gr = O.backward(cache, q)
        gpos, _ = O.pair_distances_backward(pos, cell, w.pairs, w.shifts, gr["dist"])
        _forces = -(gpos + gr["positions"])
        times.append(time.monotonic() - t0)
        if time.monotonic() - t_all + times[-1] > budget_s or len(times) >= 4:
            break
    best = float(np.median(times))
    return {
        "value": w.n_atoms / best,
        "unit": "atom-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{len(times)} full energy+forces step(s) of the same {w.n_atoms}-atom frame with oracle/pme_numpy.py "
                  f"(NumPy, single thread), median {best:.2f} s/step; host has {os.cpu_count()} logical cores",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="water", choices=["water", "ionic", "dispersion"])
    ap.add_argument("--frame-batch", default="one-launch", choices=["one-launch", "streams"],
                    help="with several frames per GPU: every kernel of the pipeline launched once for all frames "
                         "(GraphedFrameBatch), or one HIP graph per frame replayed on its own stream")
    ap.add_argument("--frames-per-gpu", type=int, default=1,
                    help="independent frames evaluated per rank and step (BASELINE.json configs[3]: --workload ionic "
                         "--frames-per-gpu 8 on 8 GPUs = 64 frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch", default="graph", choices=["graph", "eager"],
                    help="graph: replay the captured step (HIP graph); eager: launch every kernel from Python")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("MIPME_FORCE_DIST") == "1"  # the latter: 1-rank smoke test of this path
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on the C-level stdout when the communicator comes up; keep stdout for the one JSON
        # line by pointing fd 1 at stderr while the process group initialises and runs its first collective
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()  # RCCL completes its lazy set-up before anything is timed or captured
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    n_frames = max(1, args.frames_per_gpu)
    frames = [Frame(make_workload(args.workload, rank * n_frames + f), device) for f in range(n_frames)]
    frame, w = frames[0], frames[0].w
    s = 4 if w.dtype == "f32" else 8
    # the farm's ONE exchange: after its last frame evaluation every rank contributes its frame energy
    # to an all-gather over RCCL (8 B per frame), inside the timed region
    my_energy = torch.zeros(n_frames, dtype=frame.dtype, device=device)
    all_energies = torch.zeros(world * n_frames, dtype=frame.dtype, device=device)

    def dbg(msg):
        if os.environ.get("MIPME_BENCH_DEBUG") == "1":
            torch.cuda.synchronize()
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    launch = args.launch
    graphed = None
    if launch == "graph":
        try:
            graphed = [tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos, f.pairs, f.shifts) for f in frames]
        except Exception as exc:  # capture not possible on this stack: fall back to eager launches, and say so
            print(f"[bench] HIP-graph capture failed ({type(exc).__name__}: {exc}); using eager launches", file=sys.stderr)
            launch, graphed = "eager", None

    # several frames per rank: every frame replays its graph on its own stream, so the (latency-bound, small) kernels of
    # independent frames overlap on the GPU; each stream orders the successive steps of its frame
    streams = [torch.cuda.Stream(device) for _ in frames] if (graphed is not None and n_frames > 1) else None
    batch = None
    if streams is not None and args.frame_batch == "one-launch":
        # all frames of this rank with one launch per kernel (blockIdx.y = frame) from one HIP graph
        batch = tpa.GraphedFrameBatch(frames[0].calc, [(f.q, f.cell, f.pos, f.pairs, f.shifts) for f in frames])

    def one_step():
        """One pass of the hot path over this rank's batch of frames; returns the frame energies (device tensors)."""
        if batch is not None:
            return list(batch()[0].unbind(0))
        if streams is not None:
            for g, st in zip(graphed, streams):
                with torch.cuda.stream(st):
                    g.graph.replay()
            return [g.energy for g in graphed]
        if graphed is not None:
            return [g()[0] for g in graphed]
        return [f.step()[0] for f in frames]

    def join_streams():
        if streams is not None and batch is None:
            for st in streams:
                torch.cuda.current_stream(device).wait_stream(st)

    def exchange(energies):
        if distributed:
            for k, E in enumerate(energies):
                my_energy[k] = E.reshape(())
            dist.all_gather_into_tensor(all_energies, my_energy)

    dbg("graph captured" if graphed is not None else "eager mode")
    for _ in range(args.warmup):
        E = one_step()
    join_streams()
    exchange(E)
    dbg("warm-up done")
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        E = one_step()
    join_streams()
    exchange(E)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dbg("timed loop done")
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * n_frames * w.n_atoms * args.steps / elapsed

    # ---- instrumented pass: per-call HIP-event timings on the launch stream (does not affect `value`) ----
    from torchpme_amd import _lib

    ops.PROFILE = {}
    _lib.profile_enable(True)
    n_instr = min(args.steps, 50)
    for _ in range(n_instr):
        frame.step()
    torch.cuda.synchronize()
    prof = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ops.PROFILE.items()}  # ms per call
    stages = {k: ms / calls for k, (calls, ms) in _lib.profile_report().items()}  # ms per launch, inside composites
    stage_calls = {k: calls / n_instr for k, (calls, ms) in _lib.profile_report().items()}
    _lib.profile_enable(False)
    ops.PROFILE = None

    # ---- accuracy of the timed dtype against the same path in fp64 (itself pinned to the reference at 1e-13) ----
    accuracy = None
    if rank == 0 and w.dtype == "f32":
        import copy

        w64 = copy.copy(w)
        w64.dtype = "f64"
        f64 = Frame(w64, device)
        E64, F64 = f64.step()
        E32, F32 = frame.step()
        accuracy = {
            "reference": "same HIP path in fp64 (parity with torch-pme fp64 <= 1e-12, tests/test_gpu_parity.py)",
            "rel_energy_error": abs(float(E32) - float(E64)) / abs(float(E64)),
            "force_rel_l2_error": float((F32.double() - F64).norm() / F64.norm()),
        }
        # the distance tensor the step produced (by-product of the pair kernel) against plain tensor arithmetic in fp64
        p64, c64 = frame.pos.detach().double(), frame.cell.double()
        d_ref = (p64[frame.pairs[:, 1]] - p64[frame.pairs[:, 0]] + frame.shifts.double() @ c64).norm(dim=1)
        accuracy["distance_max_rel_error"] = float(((frame.distances.double() - d_ref).abs() / d_ref).max())
        del f64

    if rank == 0:
        step_bytes, per_kernel = algorithmic_bytes(w, s, fused=ops.FUSE_DISTANCES)
        pair_kernels = {k: v for k, v in prof.items() if k in per_kernel}
        pair_kernels.update({k: v for k, v in stages.items() if k in per_kernel})
        # dominant kernel = the longest single launch (an HBM-streaming pair kernel at these sizes); the per-kernel
        # table below lists every kernel with its launches per step
        dom = max(pair_kernels, key=pair_kernels.get)
        achieved = per_kernel[dom] / (pair_kernels[dom] * 1e-3) / 1e9
        # HBM bytes per launch from the rocprofv3 PMC passes of the same command (tools/profile_gpu.sh ->
        # tools/pmc_to_json.py, committed under profiles/); bench.py cannot run the profiler on itself
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if args.workload == "water" and os.path.exists(pmc_path):
            traffic = json.load(open(pmc_path))["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
        table = {
            k: {"ms_per_launch": round(v, 5), "launches_per_step": stage_calls.get(k, 1.0),
                "algorithmic_MB": round(per_kernel[k] / 1e6, 3), "GBps": round(per_kernel[k] / (v * 1e-3) / 1e9, 1)}
            for k, v in sorted(pair_kernels.items(), key=lambda kv: -kv[1])
        }
        out = {
            "metric": "atom-steps/sec (energy+forces)",
            "value": value,
            "unit": "atom-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": w.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{w.name}: {w.n_atoms} atoms, {w.n_pairs} half pairs (rc={w.cutoff} A), "
                            f"{w.scheme} order {w.order}, {w.n_mesh}^3 mesh, "
                            f"{'Coulomb' if w.exponent == 1 else '1/r^%d' % w.exponent}, {w.dtype}, energy+forces via autograd",
                "frames_per_gpu": n_frames,
                "launch": ("HIP graph replay of the captured step"
                           + (", all frames in one launch per kernel (GraphedFrameBatch)" if batch is not None
                              else ", one stream per frame" if streams is not None else ""))
                          if launch == "graph" else "eager kernel launches",
                "parallelism": f"{world * n_frames} independent frame(s), {n_frames} per GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": per_kernel[dom],
                "kernel_ms": pair_kernels[dom],
            },
            "step_algorithmic_GB": step_bytes / 1e9,
            "step_hbm_frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernels": table,
            "abi_call_ms": prof,
            "energy": float(E[0].item()),
            "accuracy": accuracy,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

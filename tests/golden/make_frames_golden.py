"""Oracle energies and force checksums of EIGHT cfg3-size frames (31 944-atom water boxes, seeds 1234 ... 1241: the frames
`bench.py --frames-per-gpu 8` evaluates on one GPU), computed with the pinned oracle (oracle/pme_numpy.py, fp64) and committed as
tests/golden/frames_water.npz -- 20 s of NumPy per frame, too much for the GPU box's test run.  tests/test_gpu_fullsize.py compares
GraphedFrameBatch (one launch per kernel for all frames) and the per-frame graphs with these numbers.

    python tests/golden/make_frames_golden.py [n_frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pme_numpy as O  # noqa: E402
from torchpme_amd import workloads  # noqa: E402


def main(n_frames):
    out = {"seeds": np.arange(1234, 1234 + n_frames)}
    for f in range(n_frames):
        t0 = time.time()
        w = workloads.water_box(seed=1234 + f)
        spec = O.PotentialSpec("coulomb", 1, w.smearing, 1.0)
        dist = O.pair_distances(w.positions, w.cell, w.pairs, w.shifts)[0]
        V, cache = O.forward(spec, w.scheme, w.order, w.mesh_spacing, w.charges, w.cell, w.positions, w.pairs, dist,
                             return_cache=True)
        gr = O.backward(cache, w.charges)
        gpos_d, _ = O.pair_distances_backward(w.positions, w.cell, w.pairs, w.shifts, gr["dist"])
        F = -(gr["positions"] + gpos_d)
        sample = np.sort(np.random.default_rng(4242 + f).choice(w.n_atoms, size=256, replace=False))
        out[f"f{f}_energy"] = np.asarray(float((V * w.charges).sum()))
        out[f"f{f}_sample"] = sample
        out[f"f{f}_force_sample"] = F[sample]
        out[f"f{f}_force_sq"] = np.asarray(float((F * F).sum()))
        out[f"f{f}_n_pairs"] = np.asarray(w.n_pairs)
        out[f"f{f}_pos_checksum"] = np.array([w.positions.sum(), (w.positions**2).sum()])
        print(f"frame {f}: P={w.n_pairs} E={float(out[f'f{f}_energy']):.10f} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames_water.npz"), **out)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)

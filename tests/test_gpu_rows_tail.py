"""The last row blocks of the co-scheduled pair sum behind the convolution's INVERSE plane launch (csrc/bricks.hip
planes_inv_rows_kernel, MIPME_ROWS_TAIL = number of 64-row blocks; opt-in, profiles/r05_experiments.txt item 7): where the rows run
must not change what the step returns -- the forces, and the energy, whose late blocks' partial sums reach the gather's tail by
another way than the x stage's pre-reduction (GatherTailHost::sr2_first / sr2_count).  The library reads the switch once per
process: every setting runs in a process of its own."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys, torch
sys.path.insert(0, %r)
import torchpme_amd as tpa
import bench
from torchpme_amd import workloads
w = workloads.water_box(n_side=int(sys.argv[1]), n_mesh=64, dtype=sys.argv[2])
f = bench.Frame(w, torch.device("cuda", 0))
g = tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos, f.pairs, f.shifts)
for _ in range(3):
    E, F = g()
torch.cuda.synchronize()
F = F.double().cpu()
print(json.dumps({"E": float(E), "F_head": F[:64].flatten().tolist(), "F_tail": F[-64:].flatten().tolist(),
                  "F_sq": float((F * F).sum())}))
""" % ROOT


def run(tail, n_side, dtype):
    env = dict(os.environ, MIPME_ROWS_TAIL=str(tail))
    out = subprocess.run([sys.executable, "-c", SCRIPT, str(n_side), dtype], env=env, capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("n_side,dtype", [(22, "f32"), (17, "f32"), (9, "f32")])
def test_rows_behind_the_inverse_planes_give_the_same_step(n_side, dtype):
    """31 944 atoms (999 row blocks: the last 64-row block is half empty), 14 739 atoms (a partial last block) and 2 187 atoms
    (35 blocks of 64 rows in all), fp32 on 64^3 meshes -- where the plane spread runs and the inverse planes are 1024-thread
    workgroups (fp64 planes of that size do not fit the co-scheduled LDS budget: those steps have no row tail); 1, 40 and 1000
    (= capped at half of the rows) blocks behind the planes."""
    ref = run(0, n_side, dtype)
    tol_e, tol_f = (2e-6, 2e-5) if dtype == "f32" else (1e-12, 1e-11)
    for k in (1, 40, 1000):
        got = run(k, n_side, dtype)
        assert abs(got["E"] - ref["E"]) <= tol_e * abs(ref["E"]), (k, got["E"], ref["E"])
        assert abs(got["F_sq"] - ref["F_sq"]) <= tol_f * ref["F_sq"], k
        for key in ("F_head", "F_tail"):
            fa, fr = np.array(got[key]), np.array(ref[key])
            assert np.linalg.norm(fa - fr) <= tol_f * np.linalg.norm(fr), (k, key)

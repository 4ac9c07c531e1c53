"""Block order of the co-scheduled spread + pair-sum launch (csrc/bricks.hip, cosched_slot): bricks first, or one brick per `a` row
blocks and XCD.  The order must not change what the launch computes.  The library reads MIPME_BRICK_PATTERN once per process, so
every order runs in a process of its own; the first one (bricks first) is compared with the torch oracle as well."""
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
w = workloads.water_box(n_side=int(sys.argv[1]), n_mesh=int(sys.argv[2]))
f = bench.Frame(w, torch.device("cuda", 0))
if sys.argv[3] == "live":  # device neighbour structures + live bins: live_spread_rows_kernel
    g = tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos, neighbors=w.cutoff)
else:
    g = tpa.GraphedEnergyForces(f.calc, f.q, f.cell, f.pos, f.pairs, f.shifts)
for _ in range(3):
    E, F = g()
torch.cuda.synchronize()
F = F.double().cpu()
print(json.dumps({"E": float(E), "F_head": F[:64].flatten().tolist(), "F_sq": float((F * F).sum()), "F_sum": F.sum(0).tolist()}))
""" % ROOT


def run(pattern, n_side, n_mesh, path):
    env = dict(os.environ, MIPME_BRICK_PATTERN=str(pattern))
    out = subprocess.run([sys.executable, "-c", SCRIPT, str(n_side), str(n_mesh), path], env=env, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["binned", "live"])
@pytest.mark.parametrize("n_side,n_mesh", [(9, 32), (14, 32), (22, 64)])
def test_block_order_does_not_change_the_results(n_side, n_mesh, path):
    """2 187 atoms / 64 bricks, 8 232 atoms / 64 bricks (more row blocks than bricks x 3: rows left over behind the pattern) and
    the headline box (512 bricks, 999 row blocks: bricks left over for a = 3); fp32 sums in an order that the binning pass's
    atomics change from launch to launch, hence tolerances rather than equality."""
    ref = run(0, n_side, n_mesh, path)
    for a in (1, 2, 3, 7):
        got = run(a, n_side, n_mesh, path)
        assert abs(got["E"] - ref["E"]) <= 2e-6 * abs(ref["E"]), (a, got["E"], ref["E"])
        assert abs(got["F_sq"] - ref["F_sq"]) <= 2e-5 * ref["F_sq"], a
        fa, fr = np.array(got["F_head"]), np.array(ref["F_head"])
        assert np.linalg.norm(fa - fr) <= 2e-5 * np.linalg.norm(fr), a

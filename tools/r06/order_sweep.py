"""Order / scheme sweep of the graph-replayed energy + forces step (round-5 verdict item 5): the cfg3 box (31 944 atoms, 64^3)
with every interpolation scheme the API offers around the two that were tuned -- P3M 3, 4, 5 and Lagrange (PME) 4 (the
reference's default, calculators/pme.py:47-53), 6, 7 -- in fp32 and fp64: ms per step, the co-scheduled kernel that ran, and the
scratch bytes / registers of the step's kernels for that (scheme, order, dtype) from the code objects (no cliff = no scratch
on the default path, fp32 steps within 1.35 x of P3M-5 once the n^3 stencil work is accounted for).

    python tools/r06/order_sweep.py [--json]          (GPU box; also reached as  python bench.py --sweep-orders)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "r06"))
import torchpme_amd as tpa  # noqa: E402
from torchpme_amd import _lib, workloads  # noqa: E402

CASES = [("P3M", 3), ("P3M", 4), ("P3M", 5), ("PME", 4), ("PME", 6), ("PME", 7)]
STEP_KERNELS = ("bin_atoms_kernel", "plane_rows_kernel", "spread_rows_kernel", "spread_rows_capped_kernel", "gather_tail_kernel",
                "yz_planes_kernel", "xconv_kernel")


def resources_by_case():
    """{(scheme id, order, 'float' | 'double'): [(kernel, vgpr, sgpr, scratch)]} of the step's kernels, from the built objects."""
    try:
        from kernel_resources import resources
    except Exception:  # noqa: BLE001
        return {}
    out = {}
    for obj in ("bricks.o", "kfilter.o"):
        path = os.path.join(ROOT, "torch-pme_amd", "csrc", obj)
        if not os.path.exists(path):
            continue
        for r in resources(path):
            n = r["demangled"]
            if not n.startswith(STEP_KERNELS):
                continue
            out.setdefault(n, r)
    return out


def main(as_json=False, steps=300):
    dev = torch.device("cuda:0")
    w = workloads.water_box()
    res = resources_by_case()
    rows = []
    lib = _lib.load()
    for dtype, tname in ((torch.float32, "float"), (torch.float64, "double")):
        t = lambda a: torch.tensor(a, dtype=dtype, device=dev)  # noqa: E731
        q, cell, pos, sh = t(w.charges), t(w.cell), t(w.positions), t(w.shifts)
        pairs = torch.tensor(w.pairs, device=dev)
        for scheme, order in CASES:
            Calc = tpa.P3MCalculator if scheme == "P3M" else tpa.PMECalculator
            calc = Calc(tpa.CoulombPotential(smearing=w.smearing), mesh_spacing=w.mesh_spacing, interpolation_nodes=order).to(dtype)
            step = tpa.GraphedEnergyForces(calc, q, cell, pos, pairs, sh)
            kernel = lib.mipme_last_cosched_kernel().decode()
            for _ in range(30):
                step.graph.replay()
            torch.cuda.synchronize()
            blocks = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(steps):
                    step.graph.replay()
                b.record()
                torch.cuda.synchronize()
                blocks.append(a.elapsed_time(b) / steps)
            E = float(step.energy)
            sid = 1 if scheme == "P3M" else 0  # (MIPME_P3M = 1, MIPME_LAGRANGE = 0 in the kernels' template arguments)
            mine = []
            for name, r in res.items():
                args = name[name.index("<") + 1:]
                # kernels templated <SCHEME, N, T, ...> or <N, T, ...>
                if tname not in args:
                    continue
                head = [x.strip() for x in args.split(",")[:3]]
                if head[:2] == [str(sid), str(order)] or (head[0] == str(order) and head[1] == tname):
                    mine.append((name.split("(")[0][:70], r["vgpr"], r["sgpr"], r["scratch"]))
            scratch = {n: s for n, _, _, s in mine if s}
            rows.append(dict(scheme=scheme, order=order, dtype=tname, ms_per_step=float(np.median(blocks)), kernel=kernel, energy=E,
                             scratch_bytes=scratch, n_kernels_checked=len(mine)))
            del step
    base = {r["dtype"]: r["ms_per_step"] for r in rows if r["scheme"] == "P3M" and r["order"] == 5}
    for r in rows:
        r["vs_p3m5"] = r["ms_per_step"] / base[r["dtype"]]
    if as_json:
        print(json.dumps({"sweep_orders": rows, "box": f"{w.name}: {w.n_atoms} atoms, {w.n_mesh}^3, Coulomb, graph-replayed energy + forces step"}))
        return
    print(f"# {w.name}: {w.n_atoms} atoms, {w.n_pairs} half pairs, {w.n_mesh}^3 mesh, Coulomb; graph-replayed energy + forces step, median of 5 x {steps}")
    print(f"# {'scheme':<7}{'n':>2} {'dtype':<7}{'ms/step':>9} {'/ P3M-5':>8}  {'n^3/125':>7}  {'co-scheduled kernel':<28} energy            scratch (bytes, of the step's kernels for this case)")
    for r in rows:
        print(f"  {r['scheme']:<7}{r['order']:>2} {r['dtype']:<7}{r['ms_per_step']:>9.5f} {r['vs_p3m5']:>8.3f}  {r['order']**3 / 125:>7.2f}  {r['kernel']:<28} {r['energy']:<17.9g} "
              f"{r['scratch_bytes'] or 'none'} ({r['n_kernels_checked']} kernels)")


if __name__ == "__main__":
    main("--json" in sys.argv)

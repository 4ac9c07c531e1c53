"""GPU timing of EwaldCalculator (explicit k-space sum) -- forward + backward w.r.t. positions, eager -- for a few sizes, with the
stage profiler's per-kernel times."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402

dev = torch.device("cuda", 0)
for N, L, lr in ((512, 18.0, 2.0), (2048, 28.0, 2.0), (8000, 43.1, 2.5)):
    rng = np.random.default_rng(1)
    pos = rng.uniform(0, L, (N, 3))
    q = rng.normal(size=(N, 1)); q -= q.mean()
    cell = np.eye(3) * L
    pairs, S, _ = tpa.neighbor_list(pos, cell, 6.0)
    for dtype in (torch.float32, torch.float64):
        t = lambda a: torch.tensor(a, device=dev, dtype=dtype)  # noqa: E731
        tq, tc, ti, tS = t(q), t(cell), torch.tensor(pairs, device=dev), t(S)
        calc = tpa.EwaldCalculator(tpa.CoulombPotential(smearing=1.2), lr_wavelength=lr).to(dtype)
        tp = t(pos).requires_grad_(True)

        def step():
            tp.grad = None
            d = tpa.pair_distances(tp, ti, tc, tS)
            V = calc(tq, tc, tp, ti, d)
            E = tpa.weighted_sum(V, tq)
            E.backward()
            return E

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            E = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        g = tpa.GraphedEnergyForces(calc, tq, tc, tp.detach(), ti, tS)
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g()
        torch.cuda.synchronize()
        ms_g = (time.perf_counter() - t0) / n * 1e3
        nk = calc._frequencies(tc).shape[0]
        print(f"N={N} pairs={len(pairs)} k-vectors={nk} {dtype}: eager {ms:.3f} ms/step, graph replay {ms_g:.3f} ms/step "
              f"(energy + forces), E={float(E):.6f} / {float(g.energy):.6f}", flush=True)

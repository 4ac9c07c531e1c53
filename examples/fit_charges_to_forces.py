"""Learning charges from forces: the use of the reference that needs SECOND derivatives through the calculator.

A box of two species carries charges q_i = theta[species_i] (made neutral); the "data" are the P3M forces of a hidden theta.
Training minimises  loss(theta) = |F(theta) - F_data|^2  with  F = -dE/dr  taken with ``create_graph=True``, so that
``loss.backward()`` differentiates the calculator twice: d loss / d theta = -2 (F - F_data) . d^2 E / (dr dq) . dq/dtheta.
The reference does this with its chain of ATen ops on any device (``calculators/calculator.py:43-87,103-189``); here the
calculator runs in ``double_backward = "auto"`` mode: potentials and plain backward passes from the fused kernels (the
validation step below), the recorded backward of the training step through the differentiable primitives of
``torchpme_amd.analytic`` (``csrc/jets.hip``) -- exact, no finite differences.

    python examples/fit_charges_to_forces.py [n_side] [steps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402


def run(n_side: int = 8, steps: int = 40, dtype=torch.float64, seed: int = 0, log=None):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    a = 2.6
    n = n_side**3
    grid = np.stack(np.meshgrid(*(np.arange(n_side),) * 3, indexing="ij"), -1).reshape(-1, 3)
    positions = (grid + 0.5) * a + rng.uniform(-0.25, 0.25, (n, 3))
    species = (grid.sum(1) % 2).astype(np.int64)
    cell = np.eye(3) * n_side * a
    pairs_np, shifts_np, _ = tpa.neighbor_list(positions, cell, 6.0)

    t = lambda x, dt=dtype: torch.tensor(x, dtype=dt, device=dev)  # noqa: E731
    pos = t(positions).requires_grad_(True)
    cell_t, pairs, shifts = t(cell), torch.tensor(pairs_np, device=dev), t(shifts_np)
    onehot = t(np.eye(2)[species])
    calc = tpa.P3MCalculator(tpa.CoulombPotential(smearing=1.2), mesh_spacing=0.9, interpolation_nodes=4)
    calc.double_backward = "auto"

    def charges(theta):
        q = onehot @ theta
        return (q - q.mean()).reshape(-1, 1)

    def forces(theta, create_graph):
        q = charges(theta)
        d = tpa.pair_distances(pos, pairs, cell_t, shifts)
        energy = (q * calc(q, cell_t, pos, pairs, d)).sum()
        (g,) = torch.autograd.grad(energy, pos, create_graph=create_graph)
        return -g

    theta_true = t([0.8, -1.1])
    f_data = forces(theta_true, False).detach()
    theta = t([0.3, -0.2]).requires_grad_(True)
    opt = torch.optim.LBFGS([theta], lr=1.0, max_iter=steps, tolerance_grad=1e-12, tolerance_change=1e-14,
                            line_search_fn="strong_wolfe")
    history = []

    def closure():
        opt.zero_grad()
        loss = ((forces(theta, True) - f_data) ** 2).sum()
        loss.backward(inputs=[theta])
        history.append(float(loss.detach()))
        return loss

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.step(closure)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    # the charges are only defined up to the common shift the neutralisation removes, and the forces are even in a global
    # sign flip: compare the charge DIFFERENCE of the two species, magnitude only
    got = abs(float((theta[0] - theta[1]).detach()))
    want = abs(float(theta_true[0] - theta_true[1]))
    residual = float(((forces(theta.detach(), False) - f_data) ** 2).sum() / (f_data**2).sum())  # plain pass: fused kernels
    info = dict(n_atoms=n, n_pairs=len(pairs_np), evaluations=len(history), seconds=seconds, first_loss=history[0],
                last_loss=history[-1], charge_difference=got, charge_difference_true=want, relative_force_residual=residual)
    if log:
        log(info)
    return info


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:]]
    print(run(*args, log=None))

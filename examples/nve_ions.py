"""Microcanonical molecular dynamics of a rock-salt melt-like box of charged soft spheres, driven entirely by the GPU path:

    E = sum_{i<j} q_i q_j k_e / r_ij  (P3MCalculator, Coulomb)  +  sum_{i<j} c_i c_j / r_ij^6  (P3MCalculator, 1/r^6 repulsion)

Each of the two energies is a ``GraphedEnergyForces`` object (pair_distances -> calculator -> energy -> backward captured
once as a HIP graph, replayed every step); the neighbour list is built on the GPU and kept for the run (with a skin), so
the Hamiltonian is fixed and velocity Verlet must conserve the total energy up to O(dt^2) -- a whole-pipeline check that
the forces are the exact gradient of the energy.

    python examples/nve_ions.py [n_side] [steps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchpme_amd as tpa  # noqa: E402

KE = 14.399645478425667  # e^2 / (4 pi eps0) in eV A
TIME_UNIT_FS = 10.1805  # A sqrt(amu / eV) in fs


def run(n_side: int = 12, steps: int = 200, dt_fs: float = 1.0, dtype=torch.float64, temperature_k: float = 600.0,
        seed: int = 0, log=None):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    r0 = 2.8  # nearest-neighbour distance, A
    n = n_side**3
    L = n_side * r0
    grid = np.stack(np.meshgrid(*(np.arange(n_side),) * 3, indexing="ij"), -1).reshape(-1, 3)
    positions = (grid + 0.5) * r0 + rng.uniform(-0.05, 0.05, (n, 3))
    sign = np.where(grid.sum(1) % 2 == 0, 1.0, -1.0)
    charges = sign.reshape(-1, 1)
    rep = np.full((n, 1), np.sqrt(KE * r0**5 / 6.0))  # c_i c_j / r^6 balances the attraction of a +- pair at r0
    masses = np.where(sign > 0, 22.99, 35.45).reshape(-1, 1)
    cell = np.eye(3) * L
    cutoff, skin = 7.0, 1.0

    t = lambda a: torch.tensor(a, dtype=dtype, device=dev)  # noqa: E731
    pos, q, c6, m, cell_t = t(positions), t(charges), t(rep), t(masses), t(cell)
    pairs, shifts, _ = tpa.neighbor_list_device(pos, cell_t, cutoff + skin)
    smearing = cutoff / 5
    coul = tpa.P3MCalculator(tpa.CoulombPotential(smearing=smearing, prefactor=KE), mesh_spacing=smearing / 2,
                             interpolation_nodes=5)
    r6 = tpa.P3MCalculator(tpa.InversePowerLawPotential(exponent=6, smearing=smearing), mesh_spacing=smearing / 2,
                           interpolation_nodes=5)
    e_coul = tpa.GraphedEnergyForces(coul, q, cell_t, pos, pairs, shifts)
    e_rep = tpa.GraphedEnergyForces(r6, c6, cell_t, pos, pairs, shifts)

    def energy_forces(x):
        # the calculators return E = sum_i q_i V_i with V_i = 1/2 sum_j q_j v(r_ij): already the pair energy
        ec, fc = e_coul(x)
        er, fr = e_rep(x)
        return ec + er, fc + fr

    kb = 8.617333262e-5  # eV / K
    vel = t(rng.normal(size=(n, 3))) * torch.sqrt(kb * temperature_k / m)
    vel -= (vel * m).sum(0) / m.sum()
    dt = dt_fs / TIME_UNIT_FS
    e_pot, force = energy_forces(pos)
    e_pot, force = e_pot.clone(), force.clone()
    history = []
    t0 = time.perf_counter()
    for step in range(steps + 1):
        e_kin = 0.5 * (m * vel * vel).sum()
        history.append((float(e_pot), float(e_kin)))
        if log is not None and step % max(1, steps // 10) == 0:
            log(f"step {step:5d}  E_pot {history[-1][0]:16.8f}  E_kin {history[-1][1]:12.8f}  E_tot {sum(history[-1]):16.8f} eV")
        if step == steps:
            break
        vel = vel + 0.5 * dt * force / m
        pos = pos + dt * vel
        e_pot, force = energy_forces(pos)
        e_pot, force = e_pot.clone(), force.clone()
        vel = vel + 0.5 * dt * force / m
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return np.array(history), dict(n_atoms=n, n_pairs=int(pairs.shape[0]), wall_s=wall, steps=steps)


if __name__ == "__main__":
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    hist, info = run(n_side, steps, log=print)
    e_tot = hist.sum(1)
    print(f"{info['n_atoms']} ions, {info['n_pairs']} pairs, {steps} steps in {info['wall_s']:.2f} s "
          f"({info['wall_s'] / steps * 1e3:.3f} ms/step incl. host integrator)")
    print(f"total-energy drift: max |E - E0| = {np.abs(e_tot - e_tot[0]).max():.3e} eV; kinetic energy ~ {hist[:, 1].mean():.3f} eV")

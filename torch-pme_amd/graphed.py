"""HIP-graph replay of one energy + forces evaluation.

At 32k atoms a step is a dozen short kernels behind ~0.3 ms of Python / autograd / launch overhead when run eagerly.  For a fixed topology (same neighbour list, cell, charges; positions change) the whole
``pair_distances -> calculator.forward -> (q*V).sum().backward()`` chain is captured once into a HIP graph
(``torch.cuda.CUDAGraph``: PyTorch is the capture front end, every captured node is a libmipme / hipFFT kernel
or a tiny reduction) and replayed per step.  This is the MD-loop form of the hot path; the calculators themselves
stay eager and reference-compatible.
"""

from __future__ import annotations

import torch

from . import ops


class GraphedEnergyForces:
    """``E, F = step(positions)`` with ``E = sum_i q_i V_i`` and ``F = -dE/dpositions``, replayed from a HIP graph.

    :param calculator: a :class:`PMECalculator` / :class:`P3MCalculator`
    :param charges, cell, positions, neighbor_indices, neighbor_shifts: tensors on the GPU; ``positions`` only
        provides the shape/dtype and the values for the warm-up.
    :param cell_gradient: also return ``dE/dcell`` (stress) from every call
    """

    def __init__(self, calculator, charges, cell, positions, neighbor_indices, neighbor_shifts, warmup: int = 3,
                 cell_gradient: bool = False):
        self.calc = calculator
        self.q = charges.detach()
        #: with ``cell_gradient=True`` every call also returns dE/dcell (3,3) -- the virial is ``-cell.T @ dE/dcell``
        self.cell_gradient = cell_gradient
        self.cell = cell.detach().clone().requires_grad_(True) if cell_gradient else cell.detach()
        self.pos = positions.detach().clone().requires_grad_(True)
        device = positions.device
        # seeding the backward pass with -1 makes ``pos.grad`` the forces directly (no fill and no negation kernel)
        self._minus_one = torch.tensor(-1.0, dtype=positions.dtype, device=device)
        self._warmup = max(1, warmup)
        self._capture(neighbor_indices, neighbor_shifts)

    def recapture(self, neighbor_indices, neighbor_shifts, positions: torch.Tensor | None = None) -> None:
        """New neighbour list (e.g. after the atoms moved by more than the skin): rebuild the pair topology and capture the
        step again; positions, cell and charges buffers are kept."""
        if positions is not None:
            with torch.no_grad():
                self.pos.copy_(positions)
        self._capture(neighbor_indices, neighbor_shifts)

    def _capture(self, neighbor_indices, neighbor_shifts):
        calculator, cell_gradient, device, warmup = self.calc, self.cell_gradient, self.pos.device, self._warmup
        self.pairs = neighbor_indices
        self.shifts = neighbor_shifts.to(self.pos.dtype).contiguous()
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):  # warm-up off the default stream: plans, topology, filter caches get built
            for _ in range(max(1, warmup)):
                self.pos.grad = None
                self.cell.grad = None
                self._eval()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.pos.grad = None
        self.cell.grad = None
        # The captured graph holds raw pointers into buffers that were built during the warm-up and live in caches: the
        # transposed pair list (+ packed shifts), the calculator's filter table, the reduction scratch.  Keep them alive
        # for the lifetime of the graph, whatever the caches evict later.
        self._keepalive = [
            ops.get_topology(self.pairs, self.pos.shape[0]) if ops.PAIR_MODE == "rows" else None,
            getattr(calculator, "_cache", None),
            ops._dot_scratch(device, self.q.data_ptr()),
        ]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.energy = self._eval()
            self.forces = self.pos.grad
            # the backward pass is seeded with -1 (so that pos.grad is the force): undo the sign for the cell
            self.cell_grad = -self.cell.grad if cell_gradient else None

    def _eval(self):
        # deferred: the pair kernel of the calculator writes the distances as a by-product (no separate pass over the list)
        d = ops.pair_distances(self.pos, self.pairs, self.cell, self.shifts, deferred=True)
        #: the pair distances of the last evaluation (P,)
        self.distances = d.detach()
        V = self.calc(self.q, self.cell, self.pos, self.pairs, d)
        E = ops.weighted_sum(V, self.q)
        E.backward(self._minus_one)
        return E.detach()

    def __call__(self, positions: torch.Tensor | None = None):
        if positions is not None:
            with torch.no_grad():
                self.pos.copy_(positions)
        self.graph.replay()
        if self.cell_gradient:
            return self.energy, self.forces, self.cell_grad
        return self.energy, self.forces

"""Argument validation of ``Calculator.forward`` (reference ``_utils.py:4-170``).

The exception types and message texts are API surface (they are regex-asserted by the reference's
``tests/calculators/test_calculator.py:50-243``), so they are reproduced verbatim; the checks
themselves are organised as small helpers.
"""

from __future__ import annotations

import torch


def _same_device(name: str, tensor: torch.Tensor, device, suffix: str = ""):
    if tensor.device != device:
        raise ValueError(
            f"device of `{name}` ({tensor.device}) must be same as that of the `positions` class{suffix} ({device})"
        )


def _same_dtype(name: str, tensor: torch.Tensor, dtype):
    if tensor.dtype != dtype:
        raise TypeError(f"type of `{name}` ({tensor.dtype}) must be same as that of the `positions` class ({dtype})")


def _validate_parameters(
    charges: torch.Tensor,
    cell: torch.Tensor,
    positions: torch.Tensor,
    neighbor_indices: torch.Tensor,
    neighbor_distances: torch.Tensor,
    periodic: torch.Tensor | None = None,
    pair_mask: torch.Tensor | None = None,
    node_mask: torch.Tensor | None = None,
    kvectors: torch.Tensor | None = None,
) -> None:
    dtype, device = positions.dtype, positions.device
    n_atoms = positions.shape[-2]

    if list(positions.shape) != [n_atoms, 3]:
        raise ValueError(
            "`positions` must be a tensor with shape [n_atoms, 3], got tensor "
            f"with shape {list(positions.shape)}"
        )

    if list(cell.shape) != [3, 3]:
        raise ValueError(
            f"`cell` must be a tensor with shape [3, 3], got tensor with shape {list(cell.shape)}"
        )
    _same_dtype("cell", cell, dtype)
    _same_device("cell", cell, device)

    if charges.dim() != 2:
        raise ValueError(
            "`charges` must be a 2-dimensional tensor, got "
            f"tensor with {charges.dim()} dimension(s) and shape "
            f"{list(charges.shape)}"
        )
    if charges.shape[0] != n_atoms:
        raise ValueError(
            "`charges` must be a tensor with shape [n_atoms, n_channels], with "
            "`n_atoms` being the same as the variable `positions`. Got tensor with "
            f"shape {list(charges.shape)} where positions contains "
            f"{len(positions)} atoms"
        )
    _same_dtype("charges", charges, dtype)
    if charges.device != device:
        raise ValueError(
            f"device of `charges` ({charges.device}) must be same as that of the `positions` class ({device})"
        )

    if neighbor_indices.shape[1] != 2:
        raise ValueError(
            "neighbor_indices is expected to have shape [num_neighbors, 2]"
            f", but got {list(neighbor_indices.shape)} for one "
            "structure"
        )
    if neighbor_indices.device != device:
        raise ValueError(
            f"device of `neighbor_indices` ({neighbor_indices.device}) must be "
            f"same as that of the `positions` class ({device})"
        )
    if neighbor_distances.shape != neighbor_indices[:, 0].shape:
        raise ValueError(
            "`neighbor_indices` and `neighbor_distances` need to have shapes "
            "[num_neighbors, 2] and [num_neighbors], but got "
            f"{list(neighbor_indices.shape)} and {list(neighbor_distances.shape)}"
        )
    if neighbor_distances.device != device:
        raise ValueError(
            f"device of `neighbor_distances` ({neighbor_distances.device}) must be "
            f"same as that of the `positions` class ({device})"
        )
    if neighbor_distances.dtype != dtype:
        raise TypeError(
            f"type of `neighbor_distances` ({neighbor_distances.dtype}) must be same "
            f"as that of the `positions` class ({dtype})"
        )

    if periodic is not None:
        if periodic.shape != (3,):
            raise ValueError(
                f"`periodic` must be a tensor of shape (3,), got tensor with shape {list(periodic.shape)}"
            )
        if periodic.device != device:
            raise ValueError(
                f"device of `periodic` ({periodic.device}) must be same as that of "
                f"the `positions` class ({device})"
            )

    if pair_mask is not None:
        if pair_mask.shape != neighbor_indices[:, 0].shape:
            raise ValueError(
                "`pair_mask` must have the same shape as the number of neighbors, "
                f"got tensor with shape {list(pair_mask.shape)} while the number of "
                f"neighbors is {neighbor_indices.shape[0]}"
            )
        if pair_mask.device != device:
            raise ValueError(
                f"device of `pair_mask` ({pair_mask.device}) must be same as that "
                f"of the `positions` class ({device})"
            )
        if pair_mask.dtype != torch.bool:
            raise TypeError(f"type of `pair_mask` ({pair_mask.dtype}) must be torch.bool")

    if node_mask is not None:
        if node_mask.shape != (n_atoms,):
            raise ValueError(
                "`node_mask` must have shape [n_atoms], got tensor with shape "
                f"{list(node_mask.shape)} where n_atoms is {n_atoms}"
            )
        if node_mask.device != device:
            raise ValueError(
                f"device of `node_mask` ({node_mask.device}) must be same as that "
                f"of the `positions` class ({device})"
            )
        if node_mask.dtype != torch.bool:
            raise TypeError(f"type of `node_mask` ({node_mask.dtype}) must be torch.bool")

    if kvectors is not None:
        if kvectors.shape[1] != 3:
            raise ValueError(
                f"`kvectors` must be a tensor of shape [n_kvecs, 3], got tensor with shape {list(kvectors.shape)}"
            )
        if kvectors.device != device:
            raise ValueError(
                f"device of `kvectors` ({kvectors.device}) must be same as that of "
                f"the `positions` class ({device})"
            )
        if kvectors.dtype != dtype:
            raise TypeError(
                f"type of `kvectors` ({kvectors.dtype}) must be same as that of the "
                f"`positions` class ({dtype})"
            )

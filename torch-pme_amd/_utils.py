"""Argument validation of ``Calculator.forward`` as a rule table.

What must be true of the arguments is stated by the reference (``_utils.py:4-170``); the exception types and message
texts are API surface -- regex-asserted by the reference's ``tests/calculators/test_calculator.py:50-243`` -- so the
*templates* below reproduce them.  Everything else is this build's: one ordered table of rules (argument, requirement,
exception, message template) evaluated by a generic loop, with the three requirement kinds that occur -- a shape, the dtype
of ``positions`` (or ``bool`` for masks), the device of ``positions`` -- built by small factories.
"""

from __future__ import annotations

from typing import Callable, NamedTuple

import torch


class _Rule(NamedTuple):
    arg: str  # argument the rule is about; rules of an argument that is None are skipped
    ok: Callable[[dict], bool]  # requirement, evaluated on the dict of all arguments + derived values
    exc: type
    template: str  # str.format template over the fields produced by _fields()


def _fields(a: dict, arg: str) -> dict:
    """Values the message templates may refer to."""
    t = a[arg]
    return dict(
        arg=arg, shape=list(t.shape), ndim=t.dim(), dtype=t.dtype, device=t.device,
        n_atoms=a["n_atoms"], n_positions=len(a["positions"]), ref_dtype=a["dtype"], ref_device=a["device"],
        idx_shape=list(a["neighbor_indices"].shape), dist_shape=list(a["neighbor_distances"].shape),
        n_pairs=a["neighbor_indices"].shape[0],
    )


def _same_device(arg: str, tail: str = "must be same as that of the `positions` class ({ref_device})") -> _Rule:
    return _Rule(arg, lambda a: a[arg].device == a["device"], ValueError, "device of `{arg}` ({device}) " + tail)


def _same_dtype(arg: str) -> _Rule:
    return _Rule(arg, lambda a: a[arg].dtype == a["dtype"], TypeError,
                 "type of `{arg}` ({dtype}) must be same as that of the `positions` class ({ref_dtype})")


def _is_bool(arg: str) -> _Rule:
    return _Rule(arg, lambda a: a[arg].dtype == torch.bool, TypeError, "type of `{arg}` ({dtype}) must be torch.bool")


def _pair_shaped(a: dict, arg: str) -> bool:
    return a[arg].shape == a["neighbor_indices"][:, 0].shape


_RULES: tuple[_Rule, ...] = (
    _Rule("positions", lambda a: list(a["positions"].shape) == [a["n_atoms"], 3], ValueError,
          "`positions` must be a tensor with shape [n_atoms, 3], got tensor with shape {shape}"),
    # ---- cell
    _Rule("cell", lambda a: list(a["cell"].shape) == [3, 3], ValueError,
          "`cell` must be a tensor with shape [3, 3], got tensor with shape {shape}"),
    _same_dtype("cell"),
    _same_device("cell"),
    # ---- charges
    _Rule("charges", lambda a: a["charges"].dim() == 2, ValueError,
          "`charges` must be a 2-dimensional tensor, got tensor with {ndim} dimension(s) and shape {shape}"),
    _Rule("charges", lambda a: a["charges"].shape[0] == a["n_atoms"], ValueError,
          "`charges` must be a tensor with shape [n_atoms, n_channels], with `n_atoms` being the same as the variable "
          "`positions`. Got tensor with shape {shape} where positions contains {n_positions} atoms"),
    _same_dtype("charges"),
    _same_device("charges"),
    # ---- neighbour list
    _Rule("neighbor_indices", lambda a: a["neighbor_indices"].shape[1] == 2, ValueError,
          "neighbor_indices is expected to have shape [num_neighbors, 2], but got {shape} for one structure"),
    _same_device("neighbor_indices"),
    _Rule("neighbor_distances", lambda a: _pair_shaped(a, "neighbor_distances"), ValueError,
          "`neighbor_indices` and `neighbor_distances` need to have shapes [num_neighbors, 2] and [num_neighbors], but got "
          "{idx_shape} and {dist_shape}"),
    _same_device("neighbor_distances"),
    _same_dtype("neighbor_distances"),
    # ---- optional arguments
    _Rule("periodic", lambda a: a["periodic"].shape == (3,), ValueError,
          "`periodic` must be a tensor of shape (3,), got tensor with shape {shape}"),
    _same_device("periodic"),
    _Rule("pair_mask", lambda a: _pair_shaped(a, "pair_mask"), ValueError,
          "`pair_mask` must have the same shape as the number of neighbors, got tensor with shape {shape} while the number "
          "of neighbors is {n_pairs}"),
    _same_device("pair_mask"),
    _is_bool("pair_mask"),
    _Rule("node_mask", lambda a: a["node_mask"].shape == (a["n_atoms"],), ValueError,
          "`node_mask` must have shape [n_atoms], got tensor with shape {shape} where n_atoms is {n_atoms}"),
    _same_device("node_mask"),
    _is_bool("node_mask"),
    _Rule("kvectors", lambda a: a["kvectors"].shape[1] == 3, ValueError,
          "`kvectors` must be a tensor of shape [n_kvecs, 3], got tensor with shape {shape}"),
    _same_device("kvectors"),
    _same_dtype("kvectors"),
)


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
    """Raise the first violated rule of the table (the order in which the reference checks)."""
    args = dict(
        charges=charges, cell=cell, positions=positions, neighbor_indices=neighbor_indices,
        neighbor_distances=neighbor_distances, periodic=periodic, pair_mask=pair_mask, node_mask=node_mask,
        kvectors=kvectors, dtype=positions.dtype, device=positions.device, n_atoms=positions.shape[-2],
    )
    for rule in _RULES:
        if args[rule.arg] is not None and not rule.ok(args):
            raise rule.exc(rule.template.format(**_fields(args, rule.arg)))

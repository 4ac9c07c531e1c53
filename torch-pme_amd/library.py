"""``torch.library`` registration of the hot path: deployment formats either side of the calculators.

The calculators launch HIP kernels through ``ctypes``; ``torch.compile`` cannot trace that and TorchScript cannot compile
it.  Two dispatcher ops make the same path visible to both:

* ``torch.ops.mipme.potentials(charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask, periodic,
  node_mask, kvectors, spec) -> Tensor`` -- ``Calculator.forward`` of the calculator described by ``spec`` (a JSON string:
  class, potential and mesh parameters -- plain data, so a scripted / exported module carries it as a constant);
* ``torch.ops.mipme.pair_distances(positions, neighbor_indices, cell, neighbor_shifts) -> Tensor`` -- the caller-side
  distance helper (reference ``tests/helpers.py:278-304``).

Both have a fake (meta) implementation and an autograd formula, so ``torch.compile(model, fullgraph=True)`` keeps a model
that contains a calculator in one graph, ``torch.jit.script(calculator.scriptable())`` gives the TorchScript module the
reference's ``torch.jit.script(calculator)`` gives (``tests/calculators/test_workflow.py:136-162``) and
``torch.library.opcheck`` passes.  The forward op evaluates the calculator under its own autograd tape whenever gradient
mode is on and keeps that tape, keyed by the storage of the output it returned (:data:`_TAPES`, a few entries); the backward
op -- which receives the forward's output back -- differentiates the kept tape, i.e. runs the kernels of the eager backward
and nothing else.  Only when the tape is gone (evicted by later forward calls, ``KEEP_TAPE = False`` or
``torch.inference_mode``) does it re-run the forward first.  The eager calculators are the fast path (fused distances, graph replay) and are not routed
through these ops.
"""

import json
from collections import OrderedDict
from typing import Optional, Tuple

import torch
from torch import Tensor

_CALCULATORS = {}

#: storage address of a forward output -> (output with its autograd graph, the four differentiable leaves); consumed by the
#: backward op of the same call, bounded (a forward whose backward never comes is dropped after `_MAX_TAPES` later calls)
_TAPES: "OrderedDict" = OrderedDict()
_MAX_TAPES = 4
#: keep the forward's autograd tape for the backward op (False: the backward op re-runs the forward, inference pays nothing)
KEEP_TAPE = True


def calculator_spec(calc) -> str:
    """JSON description of a calculator of this package (what :func:`calculator_from_spec` needs to rebuild it)."""
    from . import calculators as C
    from .potentials import CoulombPotential, InversePowerLawPotential

    pot = calc.potential
    p = {
        "smearing": None if pot.smearing is None else float(pot.smearing),
        "exclusion_radius": None if pot.exclusion_radius is None else float(pot.exclusion_radius),
        "exclusion_degree": int(pot.exclusion_degree),
        "prefactor": float(pot.prefactor),
    }
    if isinstance(pot, InversePowerLawPotential):
        p.update(kind="InversePowerLawPotential", exponent=int(pot._exponent_int()))
    elif isinstance(pot, CoulombPotential):
        p.update(kind="CoulombPotential")
    else:
        raise TypeError(f"no dispatcher op for potentials of type {type(pot).__name__}")
    d = {"class": type(calc).__name__, "potential": p, "full_neighbor_list": bool(calc.full_neighbor_list)}
    if isinstance(calc, (C.PMECalculator, C.P3MCalculator)):
        d.update(mesh_spacing=float(calc.mesh_spacing), interpolation_nodes=int(calc.interpolation_nodes))
    elif isinstance(calc, C.EwaldCalculator):
        d.update(lr_wavelength=float(calc.lr_wavelength))
    elif type(calc) is not C.Calculator:
        raise TypeError(f"no dispatcher op for calculators of type {type(calc).__name__}")
    return json.dumps(d, sort_keys=True)


def calculator_from_spec(spec: str, dtype: torch.dtype, device: torch.device):
    """The calculator a spec describes, on ``device`` in ``dtype`` (cached: its plans and filter tables persist)."""
    key = (spec, dtype, str(device))
    calc = _CALCULATORS.get(key)
    if calc is None:
        from . import calculators as C
        from . import potentials as P

        d = json.loads(spec)
        p = dict(d["potential"])
        kind = p.pop("kind")
        pot = getattr(P, kind)(**p)
        kw = {k: v for k, v in d.items() if k not in ("class", "potential")}
        calc = getattr(C, d["class"])(pot, **kw).to(device=device, dtype=dtype)
        while len(_CALCULATORS) >= 64:
            _CALCULATORS.pop(next(iter(_CALCULATORS)))
        _CALCULATORS[key] = calc
    return calc


class _autograd_recording:
    """Inside the implementation of a dispatcher op the autograd dispatch keys are excluded (the op runs "below
    autograd"): ``torch.enable_grad()`` alone does not make ATen ops record a graph there.  The backward ops below
    re-run an eager forward and differentiate it, so they lift the exclusion for that region."""

    # the guard excludes the functionality keys (not the per-backend AutogradCUDA / AutogradCPU aliases)
    _KEYS = [getattr(torch._C.DispatchKey, k)
             for k in ("AutogradFunctionality", "AutogradOther", "AutogradNestedTensor", "ADInplaceOrView")]

    def __enter__(self):
        self._prev = [torch._C._dispatch_tls_is_dispatch_key_excluded(k) for k in self._KEYS]
        for k in self._KEYS:
            torch._C._dispatch_tls_set_dispatch_key_excluded(k, False)
        self._grad = torch.enable_grad()
        self._grad.__enter__()
        return self

    def __exit__(self, *exc):
        self._grad.__exit__(*exc)
        for k, was in zip(self._KEYS, self._prev):
            torch._C._dispatch_tls_set_dispatch_key_excluded(k, was)
        return False


# ---- potentials ---------------------------------------------------------------------------------------------------
@torch.library.custom_op("mipme::potentials", mutates_args=(), device_types="cuda")
def potentials(charges: Tensor, cell: Tensor, positions: Tensor, neighbor_indices: Tensor, neighbor_distances: Tensor,
               pair_mask: Optional[Tensor], periodic: Optional[Tensor], node_mask: Optional[Tensor],
               kvectors: Optional[Tensor], spec: str, needs: int = -1) -> Tensor:
    calc = calculator_from_spec(spec, positions.dtype, positions.device)
    # Below the autograd key the implementation cannot tell whether a backward op will follow (gradient mode reads "off" here
    # both under the user's no_grad and inside the dispatcher's own autograd node): the tape is kept unless the caller opted
    # out (KEEP_TAPE = False, or torch.inference_mode()).  Its cost when no backward comes: the speculative per-atom sums of
    # the forward kernels and at most _MAX_TAPES sets of saved buffers.
    # `needs` is what the caller -- above the autograd key, where it can still be seen -- knows about the backward to come: bit k
    # set = input k of (charges, cell, positions, neighbor_distances) requires a gradient and gradient mode is on; 0 = no backward
    # can follow (no_grad, or nothing requires a gradient): no tape, no speculative sums; -1 = unknown (direct callers of the
    # op): all four leaves record, as before.  (needs_mask() below computes it; Calculator.forward and ScriptableCalculator pass it.)
    if needs == 0 or not KEEP_TAPE or torch.is_inference_mode_enabled():
        return calc._forward_impl(charges, cell, positions, neighbor_indices, neighbor_distances, periodic, node_mask,
                                  pair_mask, kvectors)
    with _autograd_recording():
        leaves = [t.detach().requires_grad_(needs < 0 or bool(needs & (1 << k)))
                  for k, t in enumerate((charges, cell, positions, neighbor_distances))]
        V = calc._forward_impl(leaves[0], leaves[1], leaves[2], neighbor_indices, leaves[3], periodic, node_mask, pair_mask,
                               kvectors)
    out = V.detach()
    _TAPES[out.data_ptr()] = (V, leaves)
    while len(_TAPES) > _MAX_TAPES:
        _TAPES.popitem(last=False)
    return out


@potentials.register_fake
def _(charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask, periodic, node_mask, kvectors, spec, needs=-1):
    return torch.empty_like(charges)


@torch.library.custom_op("mipme::potentials_backward", mutates_args=(), device_types="cuda")
def potentials_backward(grad: Tensor, out: Tensor, charges: Tensor, cell: Tensor, positions: Tensor, neighbor_indices: Tensor,
                        neighbor_distances: Tensor, pair_mask: Optional[Tensor], periodic: Optional[Tensor],
                        node_mask: Optional[Tensor], kvectors: Optional[Tensor], spec: str, need_charges: bool,
                        need_cell: bool, need_positions: bool, need_distances: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    calc = calculator_from_spec(spec, positions.dtype, positions.device)
    needs = (need_charges, need_cell, need_positions, need_distances)
    tape = _TAPES.pop(out.data_ptr(), None)
    if tape is not None and (tape[0].shape != out.shape or tape[0].dtype != out.dtype):
        tape = None  # the address was reused by an unrelated tensor
    if tape is not None and any(n and not t.requires_grad for t, n in zip(tape[1], needs)):
        tape = None  # recorded for fewer leaves than this backward asks for (a caller that understated `needs`): start over
    with _autograd_recording():
        if tape is not None:  # the forward's own tape: only the backward kernels run
            V, leaves = tape
        else:
            leaves = [t.detach().requires_grad_(n) for t, n in zip((charges, cell, positions, neighbor_distances), needs)]
            V = calc._forward_impl(leaves[0], leaves[1], leaves[2], neighbor_indices, leaves[3], periodic, node_mask,
                                   pair_mask, kvectors)
        wanted = [t for t, n in zip(leaves, needs) if n]
        got = list(torch.autograd.grad(V, wanted, grad.contiguous(), allow_unused=True)) if wanted else []
    out = []
    for t, n in zip(leaves, needs):
        g = got.pop(0) if n else None
        out.append(torch.zeros_like(t) if g is None else g)
    return out[0], out[1], out[2], out[3]


@potentials_backward.register_fake
def _(grad, out, charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask, periodic, node_mask, kvectors, spec,
      need_charges, need_cell, need_positions, need_distances):
    return (torch.empty_like(charges), torch.empty_like(cell), torch.empty_like(positions),
            torch.empty_like(neighbor_distances))


def needs_mask(charges: Tensor, cell: Tensor, positions: Tensor, neighbor_distances: Tensor) -> int:
    """The `needs` argument of ``mipme::potentials`` for these inputs (traceable by dynamo and TorchScript)."""
    needs = 0
    if torch.is_grad_enabled():
        if charges.requires_grad:
            needs += 1
        if cell.requires_grad:
            needs += 2
        if positions.requires_grad:
            needs += 4
        if neighbor_distances.requires_grad:
            needs += 8
    return needs


def _potentials_setup(ctx, inputs, output):
    ctx.save_for_backward(output, *[t for t in inputs[:9] if isinstance(t, Tensor)])
    ctx.present = [isinstance(t, Tensor) for t in inputs[:9]]
    ctx.spec = inputs[9]


def _potentials_backward(ctx, grad):
    saved = list(ctx.saved_tensors)
    out = saved.pop(0)
    args = [saved.pop(0) if p else None for p in ctx.present]
    n = ctx.needs_input_grad
    gq, gc, gp, gd = torch.ops.mipme.potentials_backward(grad, out, *args, ctx.spec, n[0], n[1], n[2], n[4])
    return (gq if n[0] else None, gc if n[1] else None, gp if n[2] else None, None, gd if n[4] else None, None, None, None,
            None, None, None)


potentials.register_autograd(_potentials_backward, setup_context=_potentials_setup)


# ---- pair distances -----------------------------------------------------------------------------------------------
@torch.library.custom_op("mipme::pair_distances", mutates_args=(), device_types="cuda")
def pair_distances(positions: Tensor, neighbor_indices: Tensor, cell: Optional[Tensor],
                   neighbor_shifts: Optional[Tensor]) -> Tensor:
    from . import ops

    return ops._pair_distances_eager(positions, neighbor_indices, cell, neighbor_shifts, False).detach()


@pair_distances.register_fake
def _(positions, neighbor_indices, cell, neighbor_shifts):
    return positions.new_empty((neighbor_indices.shape[0],))


@torch.library.custom_op("mipme::pair_distances_backward", mutates_args=(), device_types="cuda")
def pair_distances_backward(grad: Tensor, positions: Tensor, neighbor_indices: Tensor, cell: Optional[Tensor],
                            neighbor_shifts: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    from . import ops

    with _autograd_recording():
        p = positions.detach().requires_grad_(True)
        c = None if cell is None else cell.detach().requires_grad_(True)
        d = ops._pair_distances_eager(p, neighbor_indices, c, neighbor_shifts, False)
        if c is None:
            (gp,) = torch.autograd.grad(d, (p,), grad.contiguous())
            gc = torch.zeros((3, 3), dtype=p.dtype, device=p.device)
        else:
            gp, gc = torch.autograd.grad(d, (p, c), grad.contiguous())
    return gp, gc


@pair_distances_backward.register_fake
def _(grad, positions, neighbor_indices, cell, neighbor_shifts):
    return torch.empty_like(positions), positions.new_empty((3, 3))


def _pd_setup(ctx, inputs, output):
    ctx.save_for_backward(*[t for t in inputs if isinstance(t, Tensor)])
    ctx.present = [isinstance(t, Tensor) for t in inputs]


def _pd_backward(ctx, grad):
    saved = list(ctx.saved_tensors)
    positions, pairs, cell, shifts = [saved.pop(0) if p else None for p in ctx.present]
    gp, gc = torch.ops.mipme.pair_distances_backward(grad, positions, pairs, cell, shifts)
    n = ctx.needs_input_grad
    return (gp if n[0] else None, None, gc if n[2] else None, None)


pair_distances.register_autograd(_pd_backward, setup_context=_pd_setup)


# ---- TorchScript front end ----------------------------------------------------------------------------------------
class ScriptableCalculator(torch.nn.Module):
    """TorchScript-compatible front end of a calculator: same ``forward`` signature, one dispatcher op inside.

    ``torch.jit.script(calculator.scriptable())`` is this package's counterpart of the reference's
    ``torch.jit.script(calculator)``; a saved module needs ``import torchpme_amd`` (which registers the op) before
    ``torch.jit.load``."""

    spec: str

    def __init__(self, spec: str):
        super().__init__()
        self.spec = spec

    def forward(self, charges: Tensor, cell: Tensor, positions: Tensor, neighbor_indices: Tensor,
                neighbor_distances: Tensor, periodic: Optional[Tensor] = None, node_mask: Optional[Tensor] = None,
                pair_mask: Optional[Tensor] = None, kvectors: Optional[Tensor] = None) -> Tensor:
        needs = 0
        if torch.is_grad_enabled():
            if charges.requires_grad:
                needs += 1
            if cell.requires_grad:
                needs += 2
            if positions.requires_grad:
                needs += 4
            if neighbor_distances.requires_grad:
                needs += 8
        return torch.ops.mipme.potentials(charges, cell, positions, neighbor_indices, neighbor_distances, pair_mask,
                                          periodic, node_mask, kvectors, self.spec, needs)

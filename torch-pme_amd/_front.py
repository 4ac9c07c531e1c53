"""Loader of the compiled front end (``csrc/front.cpp`` -> ``_mipme_front.so``): C++ autograd nodes for the reference call
sequence ``pair_distances -> calculator -> (q * V).sum().backward()`` in its common case.  ``module()`` returns the extension
or ``None`` (``MIPME_FRONT=0``, or the file was not built: the Python path of ``ops.py`` then serves every call, as it does for
everything outside the common case anyway)."""
from __future__ import annotations

import importlib.util
import os
import warnings

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_mipme_front.so")
ENABLED = os.environ.get("MIPME_FRONT", "1") != "0"

_mod = None
_tried = False


#: how the C++ calculator node decides the energy mode (front.cpp, g_device_select): 1 the default mix (polled when charges / cell
#: want gradients, on the device for positions only), 0 always polled, 2 always on the device -- ``module().set_device_select(k)``
#: switches a loaded module (tests/test_gpu_contract.py runs all three)
SELECT_MODE = 1


def select_mode() -> int:
    return SELECT_MODE


def module():
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    if not ENABLED:
        return None
    if not os.path.exists(PATH):
        warnings.warn(f"{PATH} is not built (make -C torch-pme_amd/csrc front): eager calculator calls keep their Python host "
                      "path", RuntimeWarning, stacklevel=2)
        return None
    _lib.load()
    try:
        spec = importlib.util.spec_from_file_location("_mipme_front", PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.load_library(_lib.LIB_PATH)
    except Exception as exc:  # noqa: BLE001  an ABI / torch-version mismatch: the Python path serves every call, as without it
        warnings.warn(f"{PATH} cannot be used ({type(exc).__name__}: {exc}); eager calculator calls keep their Python host "
                      "path -- rebuild it with `make -C torch-pme_amd/csrc front`", RuntimeWarning, stacklevel=2)
        return None
    from . import ops

    def unwrap(g):
        """What a distances node made by the extension does with a gradient that is a tensor subclass."""
        if isinstance(g, ops.LazyPairGradient):
            if not g.materialized and (g._grad_pos is not None or g._grad_cell is not None):
                return g._grad_pos, g._grad_cell, None
            return None, None, g.materialize()
        return None, None, g

    def recorded_distance_backward(g, positions, cell, pairs32, shifts, row_ptr, entries, want_pos, want_cell):
        """The distances node's adjoint under create_graph=True, from the differentiable primitives (analytic.py)."""
        from . import analytic

        if isinstance(g, ops.LazyPairGradient):
            g = g.materialize()
        return analytic.recorded_distance_backward(g, positions, cell, pairs32, shifts, (row_ptr, entries), want_pos, want_cell)

    mod.set_unwrap(unwrap)
    if hasattr(mod, "set_recorded_distance_backward"):
        mod.set_recorded_distance_backward(recorded_distance_backward)
    mod.set_second_order_hint(ops.SECOND_ORDER_HINT)
    mod.set_device_select(select_mode())
    _mod = mod
    return mod

"""tgp — MI355X-native drop-in for the SRC pooling hot path of torch-geometric-pool 1.0.1.

Same import surface as the reference for the path it covers (``tgp.poolers.get_pooler``,
``tgp.src.PoolingOutput``, ``tgp.select.SelectOutput``, ``tgp.reduce.BaseReduce``,
``tgp.connect.{SparseConnect,DenseConnect,KronConnect}``, ``tgp.lift.BaseLift``); Reduce and
Connect run as hand-written HIP kernels for gfx950 behind a C ABI (``include/tgp_hip.h``).
Neither torch_geometric nor torch_scatter is needed.
"""
import importlib
import sys

eps = 1e-8  # reference: tgp/__init__.py:6

__version__ = "1.0.1+mi355x"

_submodules = ["poolers", "src", "select", "reduce", "lift", "connect", "utils", "kernels", "distributed"]


def freeze_gc() -> None:
    """Move every object alive now (the imported torch / numpy / scipy modules: ~10^6 containers) into the collector's
    permanent generation.  A training step of a pooler allocates ~1500 tracked containers (autograd nodes, tuples), so
    CPython runs a full generation-2 collection every ~50 steps, and with torch imported that pass takes ~40 ms on the
    host -- 0.8 ms per step amortised, next to ~1.2 ms of actual work (measured, MI355X box).  Calling this once
    after start-up (model built, first batch seen) makes those passes scan only what was allocated afterwards.  Opt-in:
    the collector's policy belongs to the application."""
    import gc
    gc.collect()
    gc.freeze()


def output_views(enable: bool = True):
    """Context manager: inside it the single-call sparse operators hand out pooled edge lists as VIEWS of their
    capacity-E buffers (no copy, E-sized storage kept alive) instead of new exact-size tensors -- see
    ``tgp.kernels.output_views``.  The default everywhere else is the reference's contract: fresh contiguous tensors."""
    from .kernels import output_views as _ov
    return _ov(enable)


def __getattr__(name):
    if name in _submodules:
        module = importlib.import_module(f".{name}", __name__)
        setattr(sys.modules[__name__], name, module)
        return module
    raise AttributeError(f"module {__name__} has no attribute {name}")

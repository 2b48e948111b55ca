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


def __getattr__(name):
    if name in _submodules:
        module = importlib.import_module(f".{name}", __name__)
        setattr(sys.modules[__name__], name, module)
        return module
    raise AttributeError(f"module {__name__} has no attribute {name}")

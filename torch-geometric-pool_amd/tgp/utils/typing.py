"""Type aliases of the public API (reference: tgp/utils/typing.py)."""
from typing import Literal, Optional, Union

from torch import Tensor

SinvType = Literal["transpose", "inverse"]
ReduceType = str
LiftType = Literal["transpose", "inverse", "precomputed"]
ConnectionType = Literal["sum", "mean", "min", "max", "mul"]
Adj = Union[Tensor, "SparseTensor"]  # noqa: F821 - torch_sparse is optional
OptTensor = Optional[Tensor]

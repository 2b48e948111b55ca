"""Optional third-party packages.  None of them is required by this build."""
try:  # pragma: no cover - not present in the MI355X image
    import torch_sparse
    from torch_sparse import SparseTensor

    HAS_TORCH_SPARSE = True
except ImportError:
    torch_sparse = None
    SparseTensor = "SparseTensor"
    HAS_TORCH_SPARSE = False

try:  # pragma: no cover
    import torch_geometric

    HAS_PYG = True
except ImportError:
    torch_geometric = None
    HAS_PYG = False

# Graclus matching is implemented in this package (tgp.select.graclus_cluster); torch_cluster is
# never needed, the flag only mirrors the reference's attribute (tgp/imports.py:1-7).
HAS_TORCH_CLUSTER = True
HAS_TORCH_SCATTER = False


def check_torch_sparse_available():
    if not HAS_TORCH_SPARSE:
        raise ImportError("The 'torch_sparse' package is required for this operation.")


def is_sparsetensor(obj) -> bool:
    """True only for a torch_sparse.SparseTensor (never for torch COO tensors)."""
    return HAS_TORCH_SPARSE and isinstance(obj, SparseTensor)

"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol ``include/tgp_hip.h`` declares; the ctypes table binds exactly that set; compute entry points are
never reached with host tensors (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tgp_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("tgp_reduce_sparse_f32", "tgp_reduce_batch_i64", "tgp_connect_subgraph_count",
                 "tgp_connect_subgraph_fill", "tgp_connect_coalesce_count", "tgp_connect_coalesce_fill",
                 "tgp_postprocess_sparse_norm_f32", "tgp_dense_pool_f32", "tgp_postprocess_dense_f32",
                 "tgp_block_diag_count", "tgp_block_diag_fill", "tgp_last_error", "tgp_version"):
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol():
    from tgp import _native
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} declared in include/tgp_hip.h but not exported"


def test_ctypes_table_matches_header():
    from tgp import _native
    assert sorted(_native.SIGNATURES) == declared_symbols()
    lib = _native.lib()
    m = re.search(r"#define\s+TGP_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert m and lib.tgp_version() == int(m.group(1))
    assert lib.tgp_last_error() is not None


def test_flag_values_match_header():
    from tgp import _native
    text = open(HEADER).read()
    for name in ("REMOVE_SELF_LOOPS", "DEGREE_NORM", "EDGE_WEIGHT_NORM", "SUM_AXIS_ROWS", "EPS_FILTER",
                 "ADJ_TRANSPOSED", "NODE_FILTER"):
        m = re.search(rf"TGP_{name}\s*=\s*(\d+)", text)
        assert m and int(m.group(1)) == getattr(_native, name), name
    for op, val in (("SUM", 0), ("MEAN", 1), ("MIN", 2), ("MAX", 3), ("MUL", 4)):
        assert re.search(rf"TGP_{op}\s*=\s*{val}\b", text)
        assert _native.REDUCE_OPS[op.lower()] == val


def test_workspace_queries_run_without_a_gpu():
    from tgp import _native
    lib = _native.lib()
    assert lib.tgp_connect_coalesce_workspace_bytes(10_000_000, 1_000_000, 550_000) > 10_000_000 * 24
    assert lib.tgp_dense_pool_workspace_bytes(32, 1024, 128, 64) >= 32 * 1024 * 128 * 4
    assert lib.tgp_assign_index_workspace_bytes(0, 0) > 0


def test_no_cpu_fallback():
    from tgp import _native
    from tgp.connect import DenseConnect, SparseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    x = torch.randn(4, 3)
    so = SelectOutput(cluster_index=torch.tensor([0, 0, 1, 1]))
    with pytest.raises(_native.TgpNativeError, match="no CPU fallback"):
        BaseReduce()(x, so)
    with pytest.raises(_native.TgpNativeError, match="no CPU fallback"):
        SparseConnect()(torch.tensor([[0, 1], [1, 0]]), so)
    with pytest.raises(_native.TgpNativeError, match="no CPU fallback"):
        DenseConnect()(torch.rand(1, 4, 4), SelectOutput(s=torch.rand(1, 4, 2)))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "torch-geometric-pool_amd", "tgp")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "tgp_oracle" not in src and "import oracle" not in src, f


def test_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation happens before any HIP call, so it can be exercised on the CPU box."""
    from tgp import _native
    lib = _native.lib()
    assert lib.tgp_reduce_sparse_f32(None, -1, 4, 4, None, None, None, None, 0, 1, None, None) == -1  # TGP_ERR_INVALID
    assert b"tgp_reduce_sparse_f32" in lib.tgp_last_error()
    assert lib.tgp_connect_coalesce_count(None, None, None, 5, None, 4, 2, 99, 0, 1e-8, None, 0, None, None) == -1
    assert lib.tgp_connect_coalesce_count(None, None, None, 1 << 31, None, 4, 2, 0, 0, 1e-8, None, 0, None, None) == -1
    dummy = ctypes.c_int64(0)
    p = ctypes.addressof(dummy)
    # valid pointers but a workspace that is too small -> TGP_ERR_WORKSPACE, message names the call
    assert lib.tgp_connect_subgraph_count(p, p, None, 10, None, 0, 10, 0, 1e-8, p, 8, p, None) == -2
    assert b"workspace too small" in lib.tgp_last_error()
    assert lib.tgp_dense_pool_f32(p, p, p, 1, 1 << 31, 4, 4, 0, 1e-8, None, p, None, p, p, 1 << 20, None) == -4  # TGP_ERR_RANGE
    assert lib.tgp_block_diag_count(p, 70000, 70000, None, 0, 1e-8, p, 1 << 20, p, None) == -4
    # Kron: fp32 and fp64 values at once is a caller error; sizes beyond the int32 internals are refused
    assert lib.tgp_kron_batched_count(p, p, p, p, None, 0, 4, 4, p, 1, 4, -1, -1, -1, p, 2, 1e-2, None, p, 1 << 20, p, None) == -1
    assert lib.tgp_kron_batched_count(p, p, None, None, None, 0, 1 << 31, 4, p, 1, 4, -1, -1, -1, p, 2, 1e-2, None, p, 1 << 20, p, None) == -4


def test_ctypes_signatures_match_the_header_parameter_by_parameter():
    """Every binding in tgp/_native.py must take exactly the parameters the header declares, in kind: pointer /
    int64 / int / size_t / double.  (A drifted argtypes list would otherwise only show up as a wrong answer or a
    crash on the GPU box.)"""
    from tgp import _native
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    decls = dict(re.findall(r"\b(tgp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))

    def kind(param: str):
        p = " ".join(param.split())
        if "*" in p:
            return ctypes.c_void_p
        if p.startswith("int64_t"):
            return ctypes.c_int64
        if p.startswith("size_t"):
            return ctypes.c_size_t
        if p.startswith("double"):
            return ctypes.c_double
        if p.startswith("float"):
            return ctypes.c_float
        if p.startswith("uint64_t"):
            return ctypes.c_uint64
        if p.startswith("uint32_t"):
            return ctypes.c_uint32
        if p.startswith("int") or p.startswith("unsigned"):
            return ctypes.c_int
        raise AssertionError(f"unrecognised parameter '{p}'")

    for name, (_, argtypes) in _native.SIGNATURES.items():
        params = [q for q in decls[name].split(",") if q.strip() and q.strip() != "void"]
        want = [kind(q) for q in params]
        got = [ctypes.c_void_p if a is ctypes.c_char_p else a for a in argtypes]
        assert got == want, f"{name}: header {[w.__name__ for w in want]} vs binding {[g.__name__ for g in got]}"


def test_graft_entry_build_from_a_clean_tree(tmp_path):
    """``__graft_entry__.build()`` on a scratch copy without ``lib/``: clean checkout -> make -> bind every symbol
    -> version check against the header.  (r2 shipped a stale version literal in build() that no test reached.)"""
    import shutil
    import subprocess
    import sys
    dst = tmp_path / "tree"
    ignore = shutil.ignore_patterns(".git", "gpurun_out", "lib", "__pycache__", "profiles", "golden", ".pytest_cache")
    shutil.copytree(ROOT, dst, ignore=ignore)
    assert not (dst / "torch-geometric-pool_amd" / "lib").exists()
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); print('BUILD-OK')"],
                       cwd=dst, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "BUILD-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert (dst / "torch-geometric-pool_amd" / "lib" / "libtgp_hip.so").exists()


def test_round6_entry_points_reject_bad_arguments_without_a_gpu():
    """The r6 entries validate before any HIP call too: null outputs, workspaces that are too small, modes out of range."""
    from tgp import _native
    lib = _native.lib()
    d = (ctypes.c_int64 * 4)()
    p = ctypes.addressof(d)
    # the rows route's one-call forward: workspace too small (-2), a loss mode without its outputs (-1), a bad mode (-1)
    need = lib.tgp_pool_rows_fwd_workspace_bytes(4, 16, 8, 64, 200)
    assert need >= lib.tgp_segment_gemm_tn3_post_workspace_bytes(4, 16, 8, 16, 64)
    args = [p, 200, 8, p, None, p, p, p, None, 100, p, 4, 16, 64, 0, 3, 1e-12, 1e-15]
    tail = [None, 0.0, 1.0, 1.0, p, p, p, p, p]
    assert lib.tgp_pool_rows_fwd_f32(*args, 0, *tail, None, None, None, None, None, None, None, None, p, 8, None) == -2
    assert b"workspace too small" in lib.tgp_last_error()
    assert lib.tgp_pool_rows_fwd_f32(*args, 1, *tail, None, None, None, None, None, None, None, None, p, need, None) == -1
    assert b"MinCut outputs" in lib.tgp_last_error()
    assert lib.tgp_pool_rows_fwd_f32(*args, 7, *tail, None, None, None, None, None, None, None, None, p, need, None) == -1
    # its backward: the weight gradient needs the slab buffer; DiffPool needs the forward's link loss
    base = [p, p, p, p, p, p, None, None, None, None, p, None, p, 1, 200, 4, 16, 8, 64, 3, 1e-12, 1e-15]
    grads = [None, 0, None, None, 0, None, None, None]
    assert lib.tgp_pool_rows_bwd_f32(*base, 0, 0, 0.25, 0.0, 0.0, *grads, None, p, None, p, p, p, p, None, p, None,
                                     None) == -1
    assert b"slab buffer" in lib.tgp_last_error()
    assert lib.tgp_pool_rows_bwd_f32(*base, 2, 0, 0.25, 1.0, 1.0, *grads, None, p, None, p, p, p, p, p, p, None,
                                     None) == -1
    assert b"DiffPool operands" in lib.tgp_last_error()
    # the count wait needs an epoch; the per-graph records tail a 16-byte aligned table; the SpMM riders their outputs
    assert lib.tgp_result_wait_pack_cols(p, 0, p, p, None) == -1
    # mask -> index: an epoch, a scratch buffer and a result word are required; k cannot exceed n
    assert lib.tgp_mask_index_scratch_words(10000) == 2 + 3
    assert lib.tgp_mask_index_count(p, 16, None, p, p, 0, None) == -1
    assert lib.tgp_mask_index_count(p, 16, None, None, p, 1, None) == -1
    assert lib.tgp_mask_index_count(p, 1 << 31, None, p, p, 1, None) == -4
    assert lib.tgp_mask_index_fill(p, 16, p, 17, p, None, None, None, None, None, None) == -1
    assert lib.tgp_mask_index_fill(p, 16, p, 0, None, None, None, None, None, None, None) == 0
    assert lib.tgp_diffpool_stats_tail_f32(p + 4, 3, 1.0, 1.0, p, None) == -1
    assert lib.tgp_spmm_csr_stats_f32(p, p, None, 4, 4, p, 16, p, None, None, None) == -1
    n_part = ctypes.c_int(0)
    assert lib.tgp_spmm_csr_entropy_f32(p, p, None, 4, 4, p, 16, p, 1e-15, None, ctypes.addressof(n_part), None) == -1
    assert lib.tgp_dense_pool_small_diff_f32(p, p, p, p, None, None, 64, 40, 12, 16, 3, 1e-12, 1e-15, p, p, None, p, p,
                                             None, None) == -1  # S and W at once
    assert lib.tgp_segment_gemm_tn3_post_f32(p, p, p, 8, p, 16, p, p, p, p, p, 4, 200, 16, 64, 0, 3, 1e-12, p, 8, None) == -2

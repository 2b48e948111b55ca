#!/usr/bin/env python3
"""Headline benchmark: pooled nodes/sec of Reduce + Connect on batched graphs (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (BaseReduce + DenseConnect, i.e. S^T X and post-processed
S^T A S) over one resident synthetic batch.  N = 1 workload = BASELINE.json configs[1]:
DiffPool dense, B = 32 graphs, N = 1024, K = 128, F = 64 (fp32, the precision the reference computes
in).  Select and the sparse->dense preprocessing are excluded; inputs are resident in
HBM when the timed region starts.  With N > 1 every rank pools its own B graphs (weak scaling, seed =
rank) and the pooled outputs are all-gathered over RCCL inside the timed region.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     fp32-MFMA roofline of the dominant kernel (U = A S, 2*B*N*N*K flop per launch), timed
                  live with HIP events on the launch stream,
  "cpu_baseline": the CPU oracle (port of the reference algorithm) on a bounded sample of the same
                  workload on this box's host cores (rank 0, N = 1 only).
Other workloads (--workload c3|c4|c5) are for DESIGN.md tables, not the driver's line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def dense_inputs(B, N, K, F, seed, dev):
    """A = (rand < 0.01) symmetrised, zero diagonal; X ~ N(0,1); S = softmax(randn)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    A = (torch.rand(B, N, N, device=dev, generator=g) < 0.01).float()
    A = torch.maximum(A, A.transpose(1, 2)).contiguous()
    A.diagonal(dim1=1, dim2=2).zero_()
    X = torch.randn(B, N, F, device=dev, generator=g)
    S = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1)
    return S, A, X


def timed(fn, steps, warmup, sync, barrier):
    for _ in range(warmup):
        fn()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    barrier()
    return time.perf_counter() - t0


def event_time_ms(fn, reps, dev):
    """Average duration of `fn` (one kernel launch) measured with HIP events on the launch stream."""
    stream = torch.cuda.current_stream(dev)
    fn()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(reps):
        fn()
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def cpu_baseline_dense(B, N, K, F, budget_s=12.0):
    """CPU oracle (plain torch port of reference base_reduce.py:158-161 + dense_conn.py:111-122 +
    ops.py:282-335) on a bounded sample: B_s graphs of the same shape, repeated for ~budget_s seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tgp_oracle as O
    bs = min(B, 8)
    g = torch.Generator().manual_seed(0)
    A = (torch.rand(bs, N, N, generator=g) < 0.01).float()
    A = torch.maximum(A, A.transpose(1, 2))
    X = torch.randn(bs, N, F, generator=g)
    S = torch.softmax(torch.randn(bs, N, K, generator=g), -1)

    def step():
        O.reduce_dense(S, X)
        O.postprocess_dense(O.dense_connect(S, A), True, True, True, False)

    def rate(seconds):
        step()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            step()
            n += 1
        dt = time.perf_counter() - t0
        return bs * N * n / dt, n, dt

    threads = torch.get_num_threads()
    value, n, dt = rate(budget_s)
    torch.set_num_threads(1)  # all host cores, and again with one thread
    try:
        one, n1, dt1 = rate(budget_s / 3)
    finally:
        torch.set_num_threads(threads)
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": value, "unit": "nodes/s", "cores": threads, "kind": "port",
            "sample": f"{bs} of {B} graphs (N={N},K={K},F={F}) x {n} passes of the CPU oracle, {dt:.1f}s",
            "one_thread": {"value": one, "passes": n1, "seconds": round(dt1, 1)},
            "host": {"cpu_count": os.cpu_count(), "model": model}}


def run_other_workload(args, dev):
    """Secondary workloads for the DESIGN.md / BASELINE.md tables (single GPU, same JSON layout)."""
    from tgp import kernels
    from tgp.connect import DenseConnect, SparseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import GraclusSelect, SelectOutput
    from tgp.utils.ops import postprocess_adj_pool_dense
    from tgp.src import DenseSRCPooling
    g = torch.Generator(device=dev).manual_seed(0)
    red = BaseReduce()
    extra = {}
    if args.workload == "c3":
        B, Nmax, K, F = 2048, 60, 20, 32
        n_b = torch.randint(20, 61, (B,), device=dev, generator=g)
        mask = torch.arange(Nmax, device=dev).unsqueeze(0) < n_b.unsqueeze(1)
        A = (torch.rand(B, Nmax, Nmax, device=dev, generator=g) < (3.7 / 40)).float()
        A = torch.maximum(A, A.transpose(1, 2)) * mask.unsqueeze(1) * mask.unsqueeze(2)
        A.diagonal(dim1=1, dim2=2).zero_()
        A = A.contiguous()
        X = torch.randn(B, Nmax, F, device=dev, generator=g) * mask.unsqueeze(-1)
        S = torch.softmax(torch.randn(B, Nmax, K, device=dev, generator=g), -1) * mask.unsqueeze(-1)
        so, conn = SelectOutput(s=S, in_mask=mask), DenseConnect()
        nodes = int(n_b.sum())

        fused_pool = DenseSRCPooling(reducer=red, connector=conn, adj_transpose=True)

        def step():
            with torch.no_grad():
                if args.unfused:
                    red(X, so)
                    raw = conn.dense_connect(adj=A, s=S)  # MinCut order: raw -> (loss) -> post-process
                    postprocess_adj_pool_dense(raw, True, True, True, False)
                else:  # MinCut forward: x_pool, raw S^T A S (for the cut loss) and post-processed A' at once
                    fused_pool.reduce_connect(X, A, so, want_raw=True)
        name = "MinCut PROTEINS-shape batch: B=2048, n~U[20,60] padded to 60, K=20, F=32 (BASELINE configs[2])"
        alg = 4.0 * B * (Nmax * Nmax + Nmax * K + Nmax * F + K * K + K * F)
        kern_ms = event_time_ms(step, 20, dev)
        roof = {"kernel": "whole step (graphs fit in LDS: HBM-bound)", "bound": "hbm",
                "achieved": round(alg / (kern_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(alg / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                "bytes_per_launch": alg, "avg_launch_ms": round(kern_ms, 4)}
    else:
        n, f = 1_000_000, 128
        a = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
        b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
        keep = a != b
        a, b = a[keep], b[keep]
        ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
        if not args.unsorted_edges:
            # PyG hands out row-major sorted edge lists (to_undirected / coalesce); sort once at set-up
            ei = ei[:, torch.argsort(ei[0] * n + ei[1])]
        ew = torch.ones(ei.size(1), device=dev)
        x = torch.randn(n, f, device=dev, generator=g)
        batch = torch.zeros(n, dtype=torch.long, device=dev)
        nodes = n
        if args.workload == "c4_graclus":
            so = GraclusSelect()(ei, ew, num_nodes=n)
            conn = SparseConnect()
            k = so.num_supernodes
            extra = {"num_supernodes": k, "edges": int(ei.size(1)),
                     "edge_order": "random" if args.unsorted_edges else "row-major sorted (PyG convention)"}

            def step():
                so._drop_caches()  # rebuild the inverted index every step (no cross-step caching)
                xp, bp = red(x, so, batch=batch)
                conn(ei, so, edge_weight=ew, batch_pooled=bp)
            name = "Graclus precoarsening on one N=1M E=10M graph, F=128: Reduce + coalesce Connect (BASELINE configs[3])"
            nnz = n
        else:
            if args.workload == "c4_ndp":
                keep_nodes = (torch.rand(n, device=dev, generator=g) < 0.5).nonzero().view(-1)
                wts = None
                name = "NDP-shaped S (random +-1 partition) on N=1M, F=128: Reduce only (Kron excluded, SURVEY 8(d))"
            else:
                keep_nodes = torch.sort(torch.randperm(n, device=dev, generator=g)[: n // 2])[0]
                wts = torch.rand(keep_nodes.numel(), device=dev, generator=g)
                name = "TopK-shaped S (ratio 0.5, score weights) on N=1M, F=128: scatter-reduce only"
            k = keep_nodes.numel()
            # NDPSelect numbers supernodes in node order (ndp_select.py:128-144); TopK numbers them by score
            ci = torch.arange(k, device=dev) if args.workload == "c4_ndp" else torch.randperm(k, device=dev, generator=g)
            so = SelectOutput(node_index=keep_nodes, num_nodes=n, cluster_index=ci, num_supernodes=k, weight=wts)
            extra = {"num_supernodes": k}

            def step():
                so._drop_caches()
                so._set_one_to_one_index()  # what TopkSelect / NDPSelect attach: one node per supernode, no sort
                red(x, so, batch=batch)
            so._set_one_to_one_index()
            nnz = k
        # roofline of the gather-sum kernel alone (index cached), SURVEY 8(d) A1 byte count
        idx = so.assign_index()
        if idx.one_to_one:  # row + node_index + perm (+ weight) per assignment; no row_ptr table to read
            alg = nnz * (4.0 * f + 8 + 4 + (4 if so.weight is not None else 0)) + k * 4.0 * f
        else:
            alg = nnz * (4.0 * f + 8 + 8 + 4) + k * 4.0 * f
        kern_ms = event_time_ms(lambda: kernels.reduce_sparse(x, so.node_index, so.weight, idx), 20, dev)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(f"reduce_sparse_vec4_kernel:{args.workload}")
        roof = {"kernel": "tgp::reduce_sparse_vec4_kernel (gather-sum, index cached)", "bound": "hbm",
                "achieved": round(alg / (kern_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(alg / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": traffic,
                "bytes_per_launch": alg, "avg_launch_ms": round(kern_ms, 4)}
    dt = timed(step, args.steps, args.warmup, lambda: torch.cuda.synchronize(dev), lambda: None)
    cfg = {"workload": name, "nodes_counted": "input nodes per step"}
    cfg.update(extra)
    print(json.dumps({
        "metric": "pooled nodes/sec (Reduce+Connect) on batched graphs", "value": round(nodes * args.steps / dt, 1),
        "unit": "nodes/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg, "roofline": roof}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=["c2", "c5", "c3", "c4_graclus", "c4_ndp", "topk1m"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true",
                    help="dense workloads: call the Reduce and Connect operators one after the other")
    ap.add_argument("--unsorted-edges", action="store_true",
                    help="c4_graclus: leave the synthetic edge list in random order (forces the sort-based coalesce)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TGP_BENCH_FORCE_DIST=1: run the RCCL path with a one-rank group (a 1-GPU box can then exercise it)
    distributed = world > 1 or os.environ.get("TGP_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from tgp import _native, kernels
    from tgp.connect import DenseConnect
    from tgp.distributed import PackedGather
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    from tgp.src import DenseSRCPooling
    _native.lib()

    if args.workload not in ("c2", "c5"):
        if distributed:
            raise SystemExit("secondary workloads are single-GPU (C4 does not shard: replicas only)")
        run_other_workload(args, dev)
        return
    if args.workload == "c2":
        B, N, K, F = 32, 1024, 128, 64
        name = "DiffPool dense S^T X / S^T A S, batch=32 graphs N=1024 K=128 F=64 (BASELINE configs[1])"
    else:
        B, N, K, F = 2, 8192, 512, 128
        name = "DiffPool N=8192 K=512 F=128, 2 graphs per GPU (BASELINE configs[4] shape)"
    S, A, X = dense_inputs(B, N, K, F, seed=rank, dev=dev)
    so = SelectOutput(s=S)
    reducer, connector = BaseReduce(), DenseConnect()  # DiffPool defaults (diffpool.py:98-115)
    # what the dense poolers' forward runs after Select: Reduce + Connect as one native call (SURVEY 8(d) C2:
    # "time fused A3+A7+A8"); --unfused times the two operators called one after the other instead
    fused_pool = DenseSRCPooling(reducer=reducer, connector=connector, adj_transpose=True)

    # pooled outputs of every step are all-gathered; four steps share one collective (fewer, larger RCCL calls)
    gather = PackedGather(bucket_steps=4) if distributed else None

    def step():
        with torch.no_grad():
            if args.unfused:
                x_pool, _ = reducer(X, so)
                adj_pool, _ = connector(A, so)
            else:
                x_pool, _, adj_pool = fused_pool.reduce_connect(X, A, so)
            if distributed:
                # asynchronous on RCCL's stream: overlaps the next steps' kernels (at most one collective in flight)
                gather.start([x_pool, adj_pool])
                gather.take_ready()  # a consumer would use these; the bench only has to not accumulate them
        return x_pool, adj_pool

    def sync():
        if distributed:
            gather.flush()  # every step's gather, including a partly filled last bucket, belongs to the timed region
        torch.cuda.synchronize(dev)

    def barrier():
        if distributed:
            dist.barrier()

    dt = timed(step, args.steps, args.warmup, sync, barrier)
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    nodes_per_step = world * B * N
    value = nodes_per_step * args.steps / dt

    # ---- roofline of the dominant kernel: U = A S on the fp32 matrix cores --------------------
    flops = 2.0 * B * N * N * K
    gemm_ms = event_time_ms(lambda: kernels.bmm(A, S), max(args.steps, 10), dev)
    achieved = flops / (gemm_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(f"gemm_f32_mfma_kernel<false>:{args.workload}")
    roofline = {"kernel": "tgp::gemm_f32_mfma_kernel<false> (U = A S)", "bound": "mfma",
                "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                "flops_per_launch": flops, "avg_launch_ms": round(gemm_ms, 4)}

    if rank == 0:
        line = {
            "metric": "pooled nodes/sec (Reduce+Connect) on batched graphs",
            "value": round(value, 1), "unit": "nodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "graphs_per_gpu": B, "nodes_per_graph": N, "clusters": K,
                       "features": F, "nodes_counted": "input nodes (B*N per GPU per step)",
                       "step": ("BaseReduce (S^T X) then DenseConnect (S^T A S + diag/degree post-processing)"
                                if args.unfused else
                                "fused Reduce+Connect (S^T X, S^T A S, diag/degree post-processing) as the dense "
                                "poolers' forward calls it: DenseSRCPooling.reduce_connect")
                               + (" + RCCL all-gather of every step's pooled outputs (4 steps per collective)" if distributed else ""),
                       "parallelism": f"graph-sharded x{world}"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_dense(B, N, K, F)
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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

    step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 200:
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": bs * N * n / dt, "unit": "nodes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{bs} of {B} graphs (N={N},K={K},F={F}) x {n} passes of the CPU oracle, {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=["c2", "c5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from tgp import _native, kernels
    from tgp.connect import DenseConnect
    from tgp.distributed import all_gather_dense
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    _native.lib()

    if args.workload == "c2":
        B, N, K, F = 32, 1024, 128, 64
        name = "DiffPool dense S^T X / S^T A S, batch=32 graphs N=1024 K=128 F=64 (BASELINE configs[1])"
    else:
        B, N, K, F = 2, 8192, 512, 128
        name = "DiffPool N=8192 K=512 F=128, 2 graphs per GPU (BASELINE configs[4] shape)"
    S, A, X = dense_inputs(B, N, K, F, seed=rank, dev=dev)
    so = SelectOutput(s=S)
    reducer, connector = BaseReduce(), DenseConnect()  # DiffPool defaults (diffpool.py:98-115)

    def step():
        with torch.no_grad():
            x_pool, _ = reducer(X, so)
            adj_pool, _ = connector(A, so)
            if distributed:
                x_pool, adj_pool = all_gather_dense([x_pool, adj_pool])
        return x_pool, adj_pool

    def sync():
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
                       "step": "BaseReduce (S^T X) + DenseConnect (S^T A S + diag/degree post-processing)"
                               + (" + RCCL all-gather of pooled outputs" if distributed else ""),
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

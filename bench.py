#!/usr/bin/env python3
"""Headline benchmark: pooled nodes/sec of Reduce + Connect on batched graphs (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 ...                       # starts 8 ranks itself (child processes, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path (BaseReduce + DenseConnect, i.e. S^T X and post-processed S^T A S) over one
resident synthetic batch.  Headline workload = BASELINE.json configs[1]: DiffPool dense, B = 32 graphs, N = 1024,
K = 128, F = 64 (fp32, the precision the reference computes in).  Select and the sparse->dense preprocessing are
excluded; inputs are resident in HBM when the timed region starts.  With N > 1 every rank pools its
own B graphs (weak scaling, seed = rank) and the pooled outputs are all-gathered over RCCL inside the timed region.

Rank 0 prints one JSON line per secondary workload as it is measured (short keys, no prose) and, as the LAST stdout
line, the headline (< 4 KB: contract fields, windows, roofline, whole_step, cpu_baseline); everything, prose included,
also goes to gpurun_out/bench_detail.json ($TGP_BENCH_DETAIL).  The headline line:
  contract fields  `value` / `ms_per_step` come from ONE window of EXACTLY --steps steps after --warmup warm-up steps
                   (+ "settle_steps" more untimed steps: ~50 ms of the same workload, so that the window does not sit
                   on the device's clock ramp), bracketed by barrier + device synchronise on both sides, MAX over ranks;
  "windows"        the same step re-timed as the median of 5 windows of >= 200 steps (tens of ms per window);
  "roofline"       fp32-MFMA roofline of the dominant kernel (U = A S, 2*B*N*N*K flop per launch), timed live with HIP
                   events on the launch stream;
  "cpu_baseline"   the CPU oracle (port of the reference algorithm) on a bounded sample of the same workload on this
                   box's host cores (rank 0, N = 1 only);
Secondary lines ({"secondary": name, ...}): the other north_star workloads, each with its own median-window timing and roofline:
                   c5 (N=8192,K=512,F=128: fp32 MFMA), topk1m (TopK scatter-reduce: HBM), topk_connect (TopK subgraph
                   Connect: HBM), c4_graclus (Reduce + coalesce Connect, both rooflines: HBM), c3 (MinCut small graphs:
                   HBM), c4_ndp (NDP-shaped Reduce), topk_batch / graclus_batch (batched sparse Reduce + Connect; with
                   ranks: + variable-size RCCL all-gather) and their *_fresh twins (NEW tensor objects every step, as a
                   DataLoader's mini-batches: the per-tensor memos miss), e2e_diff_c2 / e2e_mincut_c3 (WHOLE pooler forwards on sparse
                   inputs, eager and HIP-graph replayed, launches per forward), e2e_train_mincut_c3 / e2e_train_topk_c3
                   (a whole MinCut / TopK training step, forward + backward, launches per step), e2e_train_mincut_c2 /
                   e2e_train_diff_c2 (the same at the headline shape, 32 x 1024 nodes, K = 128).  With N > 1 only the graph-sharded
                   ones run; one giant graph (C4) does not shard.
`--workload X` makes X the headline of the line instead; the driver's line is the default c2.
"""
import argparse
import gc
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
# v_mfma_f64_16x16x4_f64: the guide's matrix-core table has no fp64 row.  Half the fp32-input rate (32 FLOP/clk/SIMD) by
# AMD's MI355X data sheet = 78.6 TFLOP/s; rocBLAS dgemm reaches 76.7 TFLOP/s at N = 8192 on this part
# (profiles/r05_fp64_dense.txt), which pins the figure from below.
PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
COPY_RATE_GBS = 6290.0         # MI355X_MICROARCH.md: measured float4 copy (79 % of spec): the achievable streaming rate
METRIC = "pooled nodes/sec (Reduce+Connect) on batched graphs"
SETTLE_SECONDS = 0.1  # untimed run-in of the workload before the contract window (see main)
WINDOWS, WINDOW_STEPS = 5, 200


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` without a launcher: start N ranks as CHILD processes (one per GPU) and wait for them.
    Runs before this process has made any GPU call (a process that has initialised the GPU must never exec or fork
    GPU work); `torch.cuda.device_count()` does not initialise the device on this image."""
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # a rank died: the others would wait in a collective forever
                    q.terminate()
    return rc


# ------------------------------------------------------------------------------------------------ timing
class Ctx:
    """Per-process timing context: device, ranks, barrier, max-over-ranks."""

    def __init__(self, dev, rank, world, dist):
        self.dev, self.rank, self.world, self.dist = dev, rank, world, dist

    def sync(self):
        # torch.cuda.synchronize() alone returns tens of microseconds after the device went idle (the runtime's wait
        # yields the CPU); on a 2 ms window that is percents.  Poll an event first, then synchronise for real.
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        while not ev.query():
            pass
        torch.cuda.synchronize(self.dev)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t)


def timed_window(ctx: Ctx, step, steps: int, drain=None) -> float:
    """EXACTLY `steps` steps bracketed by barrier + device synchronise on both sides; MAX over ranks (seconds)."""
    ctx.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if drain is not None:
        drain()  # e.g. the last, partly filled all-gather bucket: it belongs to the timed region
    ctx.sync()
    ctx.barrier()
    return ctx.max_over_ranks(time.perf_counter() - t0)


def median_windows(ctx: Ctx, step, drain=None, steps=WINDOW_STEPS, windows=WINDOWS) -> dict:
    ms = [timed_window(ctx, step, steps, drain) / steps * 1e3 for _ in range(windows)]
    return {"steps_per_window": steps, "windows": windows, "ms_per_step_median": round(statistics.median(ms), 5),
            "ms_per_step_min": round(min(ms), 5), "ms_per_step_max": round(max(ms), 5)}


def event_time_ms(fn, reps, dev):
    """Average duration of `fn` (the launches of one kernel / one operator) measured with HIP events on the launch
    stream (torch's current stream: the library launches on the stream handle it is given)."""
    stream = torch.cuda.current_stream(dev)
    for _ in range(3):
        fn()
    # median of 5 event-bracketed groups: one host hiccup (a preempted launch thread on a shared box) inside a single
    # bracket once turned a 21 us kernel into "58 us"
    groups, per = 5, max(reps, 4)
    avgs = []
    for _ in range(groups):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # a ~0.5 ms spin kernel in front of the bracket lets the host run ahead of the GPU: for a 20 us kernel behind a
        # 25 us Python wrapper the bracket otherwise measures the wrapper (c3 read 26 us against 21.4 us in the kernel
        # trace and in the 200-step windows on a slow host)
        torch.cuda._sleep(1_000_000)
        start.record(stream)
        for _ in range(per):
            fn()
        stop.record(stream)
        stop.synchronize()
        avgs.append(start.elapsed_time(stop) / per)
    return statistics.median(avgs)


def _traffic(key):
    """(HBM bytes per launch, source) from the committed PMC passes (profiles/roofline_traffic.json; counters cannot
    be collected from inside the run), or (None, None)."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if key and os.path.exists(tpath):
        with open(tpath) as fh:
            blob = json.load(fh)
        if blob.get(key) is not None:
            return blob[key], blob.get("_source", "profiles/roofline_traffic.json")
    return None, None


def roof_mfma(kernel, flops, ms, traffic_key=None):
    a = flops / (ms * 1e-3) / 1e12
    traffic, src = _traffic(traffic_key)
    return {"kernel": kernel, "bound": "mfma", "achieved": round(a, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(a / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
            "traffic_source": src, "flops_per_launch": flops, "avg_launch_ms": round(ms, 5)}


def roof_hbm(kernel, nbytes, ms, traffic_key=None):
    a = nbytes / (ms * 1e-3) / 1e9
    traffic, src = _traffic(traffic_key)
    return {"kernel": kernel, "bound": "hbm", "achieved": round(a, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(a / PEAK_HBM_GBS, 4), "frac_of_measured_copy_rate": round(a / COPY_RATE_GBS, 4),
            "traffic": traffic, "traffic_source": src, "bytes_per_launch": nbytes, "avg_launch_ms": round(ms, 5)}


# ------------------------------------------------------------------------------------------------ workloads
def dense_inputs(B, N, K, F, seed, dev):
    """A = (rand < 0.01) symmetrised, zero diagonal; X ~ N(0,1); S = softmax(randn)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    A = (torch.rand(B, N, N, device=dev, generator=g) < 0.01).float()
    A = torch.maximum(A, A.transpose(1, 2)).contiguous()
    A.diagonal(dim1=1, dim2=2).zero_()
    X = torch.randn(B, N, F, device=dev, generator=g)
    S = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1)
    return S, A, X


class Workload:
    """name, step(), drain() (or None), nodes per step on this rank, rooflines() -> dict or list, extra config."""
    name = ""
    nodes = 0
    shards = False
    drain = None
    extra = None

    def step(self):
        raise NotImplementedError

    def rooflines(self, dev):
        raise NotImplementedError


class DenseDiffPool(Workload):
    """C2 / C5: fused A3 + A7 + A8 with DiffPool defaults (diffpool.py:98-115)."""
    shards = True

    def __init__(self, which, ctx, unfused=False, force_collective=False):
        from tgp.connect import DenseConnect
        from tgp.distributed import PackedGather
        from tgp.reduce import BaseReduce
        from tgp.select import SelectOutput
        from tgp.src import DenseSRCPooling
        self.f64 = which == "c2_f64"
        if which in ("c2", "c2_f64"):
            self.B, self.N, self.K, self.F = 32, 1024, 128, 64
            self.name = "DiffPool dense S^T X / S^T A S, batch=32 graphs N=1024 K=128 F=64 (BASELINE configs[1])"
            if self.f64:
                self.name += ", float64 tensors (a model.double() run: the fp64 matrix path, r5)"
        else:
            self.B, self.N, self.K, self.F = 2, 8192, 512, 128
            self.name = "DiffPool N=8192 K=512 F=128, 2 graphs per GPU (BASELINE configs[4] shape)"
        self.which, self.unfused = which, unfused
        self.S, self.A, self.X = dense_inputs(self.B, self.N, self.K, self.F, seed=ctx.rank, dev=ctx.dev)
        if self.f64:
            self.S, self.A, self.X = self.S.double(), self.A.double(), self.X.double()
        self.so = SelectOutput(s=self.S)
        self.reducer, self.connector = BaseReduce(), DenseConnect()
        self.pool = DenseSRCPooling(reducer=self.reducer, connector=self.connector, adj_transpose=True)
        # pooled outputs of every step are all-gathered; four steps share one collective (fewer, larger RCCL calls)
        self.gather = (PackedGather(bucket_steps=4, force_collective=force_collective)
                       if (ctx.dist is not None and not self.f64) else None)
        self.nodes = self.B * self.N
        if self.gather is not None:
            self.drain = self.gather.flush
        self.extra = {"graphs_per_gpu": self.B, "nodes_per_graph": self.N, "clusters": self.K, "features": self.F,
                      "nodes_counted": "input nodes (B*N per GPU per step)",
                      "step": ("BaseReduce (S^T X) then DenseConnect (S^T A S + diag/degree post-processing)"
                               if unfused else
                               "fused Reduce+Connect (S^T X, S^T A S, diag/degree post-processing) as the dense "
                               "poolers' forward calls it: DenseSRCPooling.reduce_connect")
                              + (" + RCCL all-gather of every step's pooled outputs (4 steps per collective)"
                                 if self.gather is not None else "")}

    def step(self):
        with torch.no_grad():
            if self.unfused:
                x_pool, _ = self.reducer(self.X, self.so)
                adj_pool, _ = self.connector(self.A, self.so)
            elif self.gather is not None:
                # the kernels write the pooled outputs straight into the next slot of the all-gather send buffer
                ox, oa = self.gather.slots([(self.B, self.K, self.F), (self.B, self.K, self.K)], device=self.S.device)
                x_pool, _, adj_pool = self.pool.reduce_connect(self.X, self.A, self.so, out_x=ox, out_adj=oa)
            else:
                x_pool, _, adj_pool = self.pool.reduce_connect(self.X, self.A, self.so)
            if self.gather is not None:
                # asynchronous on RCCL's stream: overlaps the next steps' kernels (one collective in flight)
                self.gather.start([x_pool, adj_pool])
                self.gather.take_ready()  # a consumer would use these; the bench only must not accumulate them

    def cpu_pass(self):
        """reference base_reduce.py:158-161 + dense_conn.py:111-122 + ops.py:282-335 on the first graphs of THIS batch."""
        O = _oracle()
        bs = min(self.B, 8 if self.which in ("c2", "c2_f64") else 1)
        S, A, X = self.S[:bs].cpu(), self.A[:bs].cpu(), self.X[:bs].cpu()

        def one_pass():
            O.reduce_dense(S, X)
            O.postprocess_dense(O.dense_connect(S, A), True, True, True, False)

        return one_pass, bs * self.N, (f"the first {bs} of the {self.B} graphs of the measured batch "
                                       f"(N={self.N},K={self.K},F={self.F})"), (10.0 if self.which != "c5" else 6.0)

    def rooflines(self, dev):
        from tgp import kernels
        flops = 2.0 * self.B * self.N * self.N * self.K
        ms = event_time_ms(lambda: kernels.bmm(self.A, self.S), 50 if self.which != "c5" else 10, dev)
        if self.f64:
            a = flops / (ms * 1e-3) / 1e12
            r = {"kernel": "tgp::gemm_f64_mfma_kernel<false, true> (U = A S on v_mfma_f64_16x16x4_f64)", "bound": "mfma",
                 "achieved": round(a, 2), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(a / PEAK_FP64_MFMA_TFLOPS, 4), "traffic": None, "traffic_source": None,
                 "peak_source": "AMD MI355X data sheet (fp64 matrix = half the fp32-input MFMA rate); the guide has no "
                                "fp64 row; rocBLAS dgemm reaches 76.7 TFLOP/s on this part",
                 "flops_per_launch": flops, "avg_launch_ms": round(ms, 5)}
            ms_lib = event_time_ms(lambda: torch.bmm(self.A, self.S), 50, dev)
            r["rocblas_dgemm_same_operands_ms"] = round(ms_lib, 5)
            r["whole_step_flops"] = flops + 2.0 * self.B * self.K * self.N * (self.K + self.F)
            return r
        r = roof_mfma("tgp::gemm_f32_mfma_kernel<false> (U = A S)", flops, ms,
                      f"gemm_f32_mfma_kernel<false>:{self.which}")
        step_flops = flops + 2.0 * self.B * self.K * self.N * (self.K + self.F)
        r["whole_step_flops"] = step_flops
        return r


class SmallGraphsMinCut(Workload):
    """C3: MinCut order on the PROTEINS-shape padded batch: x_pool, raw S^T A S, post-processed A' (one launch)."""
    shards = True

    def __init__(self, ctx, unfused=False):
        from tgp.connect import DenseConnect
        from tgp.reduce import BaseReduce
        from tgp.select import SelectOutput
        from tgp.src import DenseSRCPooling
        dev = ctx.dev
        g = torch.Generator(device=dev).manual_seed(ctx.rank)
        B, Nmax, K, F = 2048, 60, 20, 32
        n_b = torch.randint(20, 61, (B,), device=dev, generator=g)
        mask = torch.arange(Nmax, device=dev).unsqueeze(0) < n_b.unsqueeze(1)
        A = (torch.rand(B, Nmax, Nmax, device=dev, generator=g) < (3.7 / 40)).float()
        A = torch.maximum(A, A.transpose(1, 2)) * mask.unsqueeze(1) * mask.unsqueeze(2)
        A.diagonal(dim1=1, dim2=2).zero_()
        self.A = A.contiguous()
        self.X = torch.randn(B, Nmax, F, device=dev, generator=g) * mask.unsqueeze(-1)
        self.S = torch.softmax(torch.randn(B, Nmax, K, device=dev, generator=g), -1) * mask.unsqueeze(-1)
        self.so, self.conn, self.red = SelectOutput(s=self.S, in_mask=mask), DenseConnect(), BaseReduce()
        self.pool = DenseSRCPooling(reducer=self.red, connector=self.conn, adj_transpose=True)
        self.unfused = unfused
        self.nodes = int(n_b.sum())
        self.dims = (B, Nmax, K, F)
        self.name = "MinCut PROTEINS-shape batch: B=2048, n~U[20,60] padded to 60, K=20, F=32 (BASELINE configs[2])"
        self.extra = {"nodes_counted": "real (un-padded) input nodes per step"}

    def step(self):
        from tgp.utils.ops import postprocess_adj_pool_dense
        with torch.no_grad():
            if self.unfused:
                self.red(self.X, self.so)
                raw = self.conn.dense_connect(adj=self.A, s=self.S)  # MinCut order: raw -> (loss) -> post-process
                postprocess_adj_pool_dense(raw, True, True, True, False)
            else:
                self.pool.reduce_connect(self.X, self.A, self.so, want_raw=True)

    def cpu_pass(self):
        """MinCut order on the CPU: S^T X, raw S^T A S, post-processing (mincut.py:220-237 over the dense operators)."""
        O = _oracle()
        S, A, X = self.S.cpu(), self.A.cpu(), self.X.cpu()

        def one_pass():
            O.reduce_dense(S, X)
            O.postprocess_dense(O.dense_connect(S, A), True, True, True, False)

        return one_pass, self.nodes, "the whole measured batch (2048 padded graphs)", 4.0

    def rooflines(self, dev):
        B, Nmax, K, F = self.dims
        alg = 4.0 * B * (Nmax * Nmax + Nmax * K + Nmax * F + K * K + K * F)  # SURVEY 8(d), padded shapes
        ms = event_time_ms(self.step, 100, dev)
        return roof_hbm("tgp::dense_pool_small_kernel (whole step, one launch: graphs fit in LDS)", alg, ms,
                        "dense_pool_small_kernel:c3")


def sparse_batch(sizes, deg, f, dev, seed=0):
    """PyG-style batch: x [Ntot,F], sorted duplicate-free undirected edge_index [2,E], sorted batch vector."""
    g = torch.Generator(device=dev).manual_seed(seed)
    sizes = torch.as_tensor(sizes, device=dev)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(sizes.numel(), device=dev), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    src = torch.arange(n, device=dev).repeat_interleave(max(deg // 2, 1))
    dst = start[batch[src]] + (torch.rand(src.numel(), device=dev, generator=g) * sizes[batch[src]]).long()
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = torch.unique(torch.cat([src * n + dst, dst * n + src]))
    ei = torch.stack([key // n, key % n])
    x = torch.randn(n, f, device=dev, generator=g)
    return x, ei, batch


def _proteins_sizes(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(20, 61, (2048,), generator=g).tolist()


def count_kernels(fn):
    """Device kernels (and memsets) one call of `fn` launches, from torch.profiler; None when the profiler is not
    available."""
    try:
        from torch.profiler import ProfilerActivity, profile
        fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        return sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
    except Exception:
        return None


class PoolerForward(Workload):
    """e2e_diff_c2 / e2e_mincut_c3: the WHOLE forward of ``get_pooler(alias)(x, adj=edge_index, batch=batch)`` on
    sparse PyG-style inputs -- sparse->dense preprocessing, MLPSelect, Reduce, Connect, auxiliary losses -- eager,
    and replayed from a HIP graph.  Not the contract metric (that excludes Select and preprocessing):
    it is what a user's call costs."""
    shards = True

    def __init__(self, which, ctx):
        from tgp.poolers import get_pooler
        dev = ctx.dev
        torch.manual_seed(ctx.rank)
        if which == "e2e_diff_c2":
            self.B, self.N, self.K, self.F = 32, 1024, 128, 64
            sizes, deg, alias = [1024] * 32, 10, "diff"
            self.name = "get_pooler('diff') whole forward on sparse inputs: 32 graphs x 1024 nodes, K=128, F=64"
        else:
            self.B, self.N, self.K, self.F = 2048, 60, 20, 32
            sizes, deg, alias = _proteins_sizes(ctx.rank), 4, "mincut"
            self.name = "get_pooler('mincut') whole forward on sparse inputs: 2048 graphs n~U[20,60], K=20, F=32"
        self.which = which
        self.x, self.ei, self.batch = sparse_batch(sizes, deg, self.F, dev, seed=ctx.rank)
        self.pooler = get_pooler(alias, in_channels=self.F, k=self.K).to(dev).eval()
        self.nodes = self.x.size(0)
        self.graph = None
        self.extra = {"nodes_counted": "real input nodes per forward", "edges": int(self.ei.size(1)),
                      "step": "whole pooler forward, eager (preprocessing + Select + Reduce + Connect + losses)"}

    def step(self):
        with torch.no_grad():
            return self.pooler(x=self.x, adj=self.ei, batch=self.batch)

    def rooflines(self, dev):
        B, N, K, F = self.B, self.N, self.K, self.F
        ms = event_time_ms(self.step, 50, dev)
        if self.which == "e2e_diff_c2":
            # r6: a sparse input of this size takes the un-padded rows route -- no [B,N,N] adjacency, no N^2 K product
            # (r5 priced this line against 2BN^2K x 2 flops that are no longer executed): what the call has to move is
            # the edge list + x in, S and the pooled outputs out
            alg = self.ei.size(1) * 16.0 + self.nodes * F * 4.0 + 4.0 * self.nodes * K + B * N + 4.0 * B * (K * K + K * F)
            r = roof_hbm("whole forward (all kernels of the call, eager; rows route: no dense adjacency)", alg, ms)
            r["bytes_counted"] = "edge list + x read, S + mask + pooled outputs written"
        else:
            # r5: what the call has to move now that nothing is densified -- edge list + x in, S [B,N,K] + node mask out,
            # pooled outputs (the r4 line also counted a dense A and a padded X written and read once: 2.7 x these bytes)
            alg = (self.ei.size(1) * 16.0 + self.nodes * F * 4.0 + 4.0 * B * N * K + B * N
                   + 4.0 * B * (2 * K * K + K * F))
            r = roof_hbm("whole forward (all kernels of the call, eager)", alg, ms)
            r["bytes_counted"] = ("edge list + x read, S [B,N,K] + mask + pooled outputs written (no dense adjacency: it is "
                                  "not materialised)")
        r["launches_per_forward"] = count_kernels(self.step)
        try:  # the same forward replayed from a HIP graph (inputs resident, sizes memoised per batch vector)
            gph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.step()
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(gph):
                self.step()
            r["hip_graph_replay_ms"] = round(event_time_ms(gph.replay, 50, dev), 5)
        except Exception as exc:
            r["hip_graph_replay_ms"] = None
            r["hip_graph_error"] = f"{type(exc).__name__}: {str(exc)[:120]}"
        return r


class PoolerTrainStep(Workload):
    """e2e_train_mincut_c3 / e2e_train_topk_c3: a WHOLE training step of ``get_pooler('mincut')`` / ``('topk')`` on the
    PROTEINS-shaped sparse batch -- forward, a scalar loss over the pooled outputs (and MinCut's two auxiliary losses),
    backward to the input features and the selector's parameters.  MinCut: MLPSelect + the fused one-launch Reduce +
    Connect + loss terms and its one-launch backward (csrc/dense_graph_kernels.h); TopK: fused score, selection, the
    one-launch Reduce + subgraph Connect, and ONE backward pass over the kept rows (tgp_topk_pool_bwd_f32).  Eager, and
    launches per step; what a user's step costs."""
    shards = True

    def __init__(self, ctx, alias="mincut", scale="c3"):
        from tgp.poolers import get_pooler
        dev = ctx.dev
        torch.manual_seed(ctx.rank)
        self.alias, self.scale = alias, scale
        if scale == "c2":  # the headline shape: 32 graphs x 1024 nodes, K = 128, F = 64 (r6: one autograd node there too)
            self.B, self.N, self.K, self.F = 32, 1024, 128, 64
            sizes, deg = [1024] * 32, 10
            shape = "32 graphs x 1024 nodes, K=128, F=64"
        else:
            self.B, self.N, self.K, self.F = 2048, 60, 20, 32
            sizes, deg = _proteins_sizes(ctx.rank), 4
            shape = "2048 graphs n~U[20,60], " + ("K=20, " if alias != "topk" else "ratio 0.5, ") + "F=32"
        self.x, self.ei, self.batch = sparse_batch(sizes, deg, self.F, dev, seed=ctx.rank)
        self.x.requires_grad_(True)
        kw = dict(k=self.K) if alias != "topk" else dict(ratio=0.5)
        self.pooler = get_pooler(alias, in_channels=self.F, **kw).to(dev).train()
        self.nodes = self.x.size(0)
        self.name = f"get_pooler('{alias}') whole TRAINING step (forward + backward) on sparse inputs: {shape}"
        self.extra = {"nodes_counted": "real input nodes per step", "edges": int(self.ei.size(1)),
                      "step": "forward + loss + backward to x and the selector's parameters, eager"}

    def _loss(self, out):
        if self.alias in ("mincut", "diff"):
            return out.x.sum() + out.edge_index.sum() + sum(out.loss.values())
        return out.x.square().sum()

    def step(self):
        self.pooler.zero_grad(set_to_none=True)
        self.x.grad = None
        loss = self._loss(self.pooler(x=self.x, adj=self.ei, batch=self.batch))
        loss.backward()
        return loss

    def rooflines(self, dev):
        B, N, K, F = self.B, self.N, self.K, self.F
        ms = event_time_ms(self.step, 50, dev)
        if self.scale == "c2":
            # the N^2 K products the step needs: U = A S (forward), V = A^T S (backward; not for a symmetric A, which the
            # pooler detects), DiffPool's link residual S S^T; + the K-sized products of both directions
            # r6: sparse inputs of this size take the un-padded rows route (no [B,N,N] adjacency, no N^2 K product): the
            # step is priced by what it has to move -- forward: edge list + x in, S / T = A S written and read by the
            # S^T [T | X | S] product, pooled outputs; backward: the operand buffer [T | X | 1 | S | dY] written and read,
            # dS, dX, the K-sized right-hand sides
            n, E = float(self.nodes), float(self.ei.size(1))
            fwd = E * 16.0 + n * F * 4.0 + 2 * 4.0 * n * (2 * K + F) + 4.0 * B * (2 * K * K + K * F)
            bwd = 2 * 4.0 * n * (3 * K + F) + 4.0 * n * (K + F) + 4.0 * B * (3 * K + F) * K + n * F * 4.0
            r = roof_hbm("whole training step (all kernels of forward + backward, eager; rows route)", fwd + bwd, ms)
            r["launches_per_step"] = count_kernels(self.step)
            from tgp import functions as Fn
            r["backward_route"] = dict(Fn.POOL_LARGE_STATS)
            return self._graphed(r, dev)
        if self.alias == "mincut":
            # forward traffic (as e2e_mincut_c3) + backward: A, S, X read again, gS, gX and the dense gradients written / read
            fwd = (self.ei.size(1) * 16.0 + self.nodes * F * 4.0 + 2 * 4.0 * B * N * N + 2 * 4.0 * B * N * (F + K)
                   + 4.0 * B * (2 * K * K + K * F))
            bwd = 4.0 * B * N * N + 3 * 4.0 * B * N * (F + K) + 4.0 * B * (K * K + K * F) + 2 * self.nodes * F * 4.0
        else:
            # forward: x read by the score pass and (kept rows) by Reduce, x' written; edges read once (16 B), survivors
            # written (16 B; a quarter at ratio 0.5); backward: g' and the kept rows of x read, gx [N,F] written
            kept = 0.5 * self.nodes
            fwd = self.nodes * F * 4.0 + 2 * kept * F * 4.0 + self.ei.size(1) * 16.0 * 1.25 + self.nodes * 16.0
            bwd = 2 * kept * F * 4.0 + self.nodes * F * 4.0
        r = roof_hbm("whole training step (all kernels of forward + backward, eager)", fwd + bwd, ms)
        r["launches_per_step"] = count_kernels(self.step)
        if self.alias != "mincut":
            return r  # (the sparse selection waits for its counts on the host: not capturable)
        return self._graphed(r, dev)

    def _graphed(self, r, dev):
        """The same step with forward and backward replayed from HIP graphs (torch.cuda.make_graphed_callables)."""
        try:
            pooler, ei, batch, loss_of = self.pooler, self.ei, self.batch, self._loss

            class Step(torch.nn.Module):
                def __init__(self):
                    super().__init__()
                    self.pooler = pooler

                def forward(self, x):
                    return loss_of(self.pooler(x=x, adj=ei, batch=batch))

            xg = self.x.detach().clone().requires_grad_(True)
            graphed = torch.cuda.make_graphed_callables(Step(), (xg,), num_warmup_iters=3)

            def replay():
                self.pooler.zero_grad(set_to_none=True)
                xg.grad = None
                graphed(xg).backward()

            r["hip_graph_replay_ms"] = round(event_time_ms(replay, 50, dev), 5)
        except Exception as exc:
            r["hip_graph_replay_ms"] = None
            r["hip_graph_error"] = f"{type(exc).__name__}: {str(exc)[:120]}"
        return r


class TopkBatch(Workload):
    """topk_batch: TopK (ratio 0.5) Reduce + subgraph Connect on 2048 PROTEINS-shaped graphs, Select excluded.  The
    batched SPARSE path of SURVEY 8(e): graphs shard by id, and with more than one rank (or TGP_BENCH_FORCE_DIST=1)
    the variable-size pooled outputs are all-gathered over RCCL inside the timed region (counts, then padded
    payloads, then node-id / graph-id offsets: the merge rule of tgp/data/collate.py:144-153)."""
    shards = True

    FRESH_COPIES = 24  # more distinct tensor objects than any per-tensor memo of the package holds (16)

    def __init__(self, ctx, force_collective=False, unfused=False, which="topk_batch"):
        from tgp.connect import SparseConnect
        from tgp.reduce import BaseReduce
        from tgp.select import GraclusSelect, TopkSelect
        dev = ctx.dev
        torch.manual_seed(ctx.rank)
        self.f = 32
        self.fresh = which.endswith("_fresh")
        which = which[: -len("_fresh")] if self.fresh else which
        self.which = which
        self.x, self.ei, self.batch = sparse_batch(_proteins_sizes(ctx.rank), 4, self.f, dev, seed=ctx.rank)
        self.ew = torch.rand(self.ei.size(1), device=dev) + 0.5
        topk_sel = TopkSelect(in_channels=self.f, ratio=0.5).to(dev) if which == "topk_batch" else None

        def select(x, ei, ew):
            with torch.no_grad():
                if topk_sel is not None:
                    return topk_sel(x=x, batch=self.batch)
                # Graclus matching of every graph: many-to-one S, relabel + coalesce Connect
                return GraclusSelect()(ei, ew, num_nodes=x.size(0), batch=self.batch)
        self.so = select(self.x, self.ei, self.ew)
        # *_fresh (verdict r4 item 2): every step pools NEW tensor objects -- x, edge_index, edge_weight and the
        # SelectOutput Select made from them -- as every mini-batch of a training loop does: the memos keyed on the edge
        # list object (per-graph edge offsets, row order, declines) miss on every step.  Select has run on each copy (it
        # is outside the metric), so what Select itself leaves behind is there, as in a real forward: the
        # batch vector's facts and the by-products it attaches to its SelectOutput.
        self.copies, self._turn = None, 0
        if self.fresh:
            self.copies = []
            for _ in range(self.FRESH_COPIES):
                xx, ee, ww = self.x.clone(), self.ei.clone(), self.ew.clone()
                self.copies.append((xx, ee, ww, select(xx, ee, ww)))
        self.red, self.conn = BaseReduce(), SparseConnect()
        from tgp.src import SRCPooling
        self.pool = SRCPooling(reducer=self.red, connector=self.conn)
        self.unfused = unfused
        self.gather = ctx.dist is not None
        self.dist_world = ctx.world
        self.force = force_collective
        self.sg = None
        if self.gather:
            # ONE payload collective per bucket of eight steps, asynchronous, two deep: a gather overlaps the next steps' kernels
            from tgp.distributed import SparseGather
            # the merged outputs are handed out as exact-size contiguous tensors (SparseGather's default, the reference's
            # collate contract); `sg_views` hands out views of the receive buffer instead and is timed beside it
            self.sg = SparseGather(force_collective=force_collective, depth=2, bucket_steps=8)
            self.sg_views = SparseGather(force_collective=force_collective, depth=2, bucket_steps=8, views=True)
            self.drain = self.sg.flush
        self.nodes = self.x.size(0)
        self.num_graphs = 2048
        self.name = (("TopK (ratio 0.5) Reduce + subgraph Connect" if which == "topk_batch" else
                      "Graclus Reduce + coalesce Connect") +
                     " on 2048 PROTEINS-shaped graphs (n~U[20,60], F=32), graph-sharded"
                     + (", NEW tensor objects every step (a DataLoader's mini-batches)" if self.fresh else "")
                     + (", pooled outputs all-gathered over RCCL (variable-size)" if self.gather else ""))
        self.extra = {"edges": int(self.ei.size(1)), "num_supernodes": int(self.so.num_supernodes),
                      "nodes_counted": "input nodes per step per GPU",
                      "tensors": (f"rotates through {self.FRESH_COPIES} pre-made copies of (x, edge_index, edge_weight, "
                                  "SelectOutput): more objects than any per-tensor memo holds, so every step misses"
                                  if self.fresh else "the same tensor objects every step (full-batch training: the "
                                  "per-graph edge offsets of the edge list come from a memo)"),
                      "output_layout": "contiguous exact-size tensors (the default; capacity views are timed beside it)",
                      "step": ("BaseReduce then SparseConnect, operator by operator" if unfused else
                               "fused Reduce + Connect as the sparse poolers' forward calls it on a batch of small "
                               "graphs: SRCPooling.reduce_connect (one launch + the count read-back)")
                              + (" + SparseGather (one asynchronous payload collective per 8 steps)" if self.gather else "")}

    def _inputs(self):
        if self.copies is None:
            return self.x, self.ei, self.ew, self.so
        self._turn = (self._turn + 1) % len(self.copies)
        return self.copies[self._turn]

    def staged(self, inputs=None):
        x, e, w, so = inputs or self._inputs()
        with torch.no_grad():
            xp, bp = self.red(x, so, batch=self.batch)
            ei, ew = self.conn(e, so, edge_weight=w, batch_pooled=bp)
        return xp, ei, ew, bp

    def compute(self):
        inputs = self._inputs()
        if not self.unfused:
            x, e, w, so = inputs
            with torch.no_grad():
                fused = self.pool.reduce_connect(x, e, w, so, self.batch)
            if fused is not None:
                xp, bp, ei, ew = fused
                return xp, ei, ew, bp
        return self.staged(inputs)

    def step(self):
        if self.gather:
            # the step's outputs go straight into the gather's pack launch: the capacity views are enough for that
            # hand-off (tgp.output_views(): no compaction launch); what the gather hands OUT are exact-size tensors
            from tgp import kernels
            with kernels.output_views():
                xp, ei, ew, bp = self.compute()
            self.sg.start(xp, ei, ew, bp, self.num_graphs)
            self.sg.take_ready()  # a consumer would use these; the bench only must not accumulate them
            return xp, ei, ew, bp
        return self.compute()

    def cpu_pass(self):
        O = _oracle()
        so = self.so
        x, ei, ew, batch = self.x.cpu(), self.ei.cpu(), self.ew.cpu(), self.batch.cpu()
        ni, ci, w, n, k = so.node_index.cpu(), so.cluster_index.cpu(), so.weight.cpu(), self.nodes, int(so.num_supernodes)

        def one_pass():
            O.reduce_sparse(x, ni, ci, w, k)
            bp = O.reduce_batch_sparse(batch, ni, ci, k)
            O.sparse_connect(ei, ew, ni, ci, n, k, batch_pooled=bp)

        return one_pass, n, "the whole measured batch (2048 graphs): Reduce + Connect", 3.0

    def rooflines(self, dev):
        xp, ei, ew, bp = self.compute()
        E, E2, k = self.ei.size(1), ei.size(1), xp.size(0)
        if self.which == "topk_batch":
            alg = (k * (4.0 * self.f + 8 + 4 + 4) + k * 4.0 * self.f          # A1 one-to-one
                   + E * 20.0 + self.nodes + k * 8.0 + E2 * 20.0)              # A5 + A6
        else:
            alg = (self.nodes * (4.0 * self.f + 8 + 8 + 4) + k * 4.0 * self.f  # A1
                   + E * 20.0 + self.nodes * 8.0 + E2 * 20.0)                   # A4 + A6
        ms_c = event_time_ms(self.compute, 50, dev)
        mode = 0 if self.which == "topk_batch" else 1
        r = roof_hbm(f"tgp::sparse_pool_small_kernel<{mode}> (Reduce + Connect: one launch + the count read-back)"
                     if not self.unfused else
                     "Reduce + Connect (all kernels of both calls + the Connect's host read-back)", alg, ms_c,
                     f"sparse_pool_small:{self.which}")
        r["compute_only_ms"] = round(ms_c, 5)
        if not self.unfused:
            from tgp import kernels
            with kernels.output_views():  # r4's layout: edge_index a view of the kernel's capacity-E buffer, no copy
                r["capacity_views_ms"] = round(event_time_ms(self.compute, 50, dev), 5)
            r["contiguous_ms"] = r["compute_only_ms"]
            r["staged_operators_ms"] = round(event_time_ms(self.staged, 50, dev), 5)
            keep = self.copies
            self.copies = None  # (the comparison needs both routes on the same inputs)
            a, b = self.compute(), self.staged()
            self.copies = keep
            r["fused_equals_staged"] = bool(all(torch.equal(u, v) for u, v in zip(a, b)))
            r["edge_index_contiguous"] = bool(a[1].is_contiguous()
                                              and a[1].untyped_storage().nbytes() <= max(2 * a[1].numel() * 8, 512))
        if self.gather:
            ms_g = event_time_ms(self.step, 50, dev)
            self.sg.flush()
            r["compute_plus_gather_ms"] = round(ms_g, 5)
            exact, self.sg = self.sg, self.sg_views
            r["compute_plus_gather_views_ms"] = round(event_time_ms(self.step, 50, dev), 5)
            self.sg.flush()
            self.sg = exact
            r["gather"] = ("SparseGather: one pack launch, one payload collective and one unpack (+ id offsets) launch per "
                           "bucket of 8 steps, asynchronous (two buckets in flight), totals through pinned host words (no "
                           "host wait); the step's own outputs are handed to it as capacity views, the merged outputs "
                           "leave it as exact-size contiguous tensors (one tgp_copy_arrays launch per step)")
            from tgp.distributed import all_gather_sparse
            merged = all_gather_sparse(xp, ei, ew, bp, self.num_graphs, force_collective=self.force)
            if self.dist_world == 1:  # one-rank group: the merged result must be the local one, bit for bit
                r["gather_equals_local"] = bool(torch.equal(merged[0], xp) and torch.equal(merged[1], ei)
                                                and torch.equal(merged[2], ew) and torch.equal(merged[3], bp))
        return r


def _big_graph(dev, g, sort_rows=True):
    n = 1_000_000
    a = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
    b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    if sort_rows:
        # PyG hands out row-major sorted edge lists (to_undirected / coalesce); sort once at set-up
        ei = ei[:, torch.argsort(ei[0] * n + ei[1])]
    return n, ei.contiguous()


class SparseReduceOnly(Workload):
    """topk1m / c4_ndp: the sparse Reduce (A1 + A2) with a one-to-one assignment (TopK, NDP)."""

    def __init__(self, which, ctx):
        from tgp.reduce import BaseReduce
        from tgp.select import SelectOutput
        dev = ctx.dev
        g = torch.Generator(device=dev).manual_seed(0)
        n, f = 1_000_000, 128
        self.which, self.f = which, f
        self.x = torch.randn(n, f, device=dev, generator=g)
        self.batch = torch.zeros(n, dtype=torch.long, device=dev)
        if which == "c4_ndp":
            keep_nodes = (torch.rand(n, device=dev, generator=g) < 0.5).nonzero().view(-1)
            wts = None
            self.name = "NDP-shaped S (random +-1 partition) on N=1M, F=128: Reduce only (Kron excluded, SURVEY 8(d))"
        else:
            keep_nodes = torch.sort(torch.randperm(n, device=dev, generator=g)[: n // 2])[0]
            wts = torch.rand(keep_nodes.numel(), device=dev, generator=g)
            self.name = "TopK-shaped S (ratio 0.5, score weights) on N=1M, F=128: scatter-reduce only"
        k = keep_nodes.numel()
        # NDPSelect numbers supernodes in node order (ndp_select.py:128-144); TopK numbers them by score
        ci = torch.arange(k, device=dev) if which == "c4_ndp" else torch.randperm(k, device=dev, generator=g)
        self.so = SelectOutput(node_index=keep_nodes, num_nodes=n, cluster_index=ci, num_supernodes=k, weight=wts)
        self.so._set_one_to_one_index()
        self.red = BaseReduce()
        self.nodes, self.k = n, k
        self.extra = {"num_supernodes": k, "nodes_counted": "input nodes per step"}

    def step(self):
        so = self.so
        so._drop_caches()
        so._set_one_to_one_index()  # what TopkSelect / NDPSelect attach: one node per supernode, no sort
        self.red(self.x, so, batch=self.batch)

    def cpu_pass(self):
        """reference base_reduce.py:14-53,141-155 (gather x weight, scatter-add, reduce_batch) on the measured inputs."""
        O = _oracle()
        so, k = self.so, self.k
        x, batch, ni, ci = self.x.cpu(), self.batch.cpu(), so.node_index.cpu(), so.cluster_index.cpu()
        w = so.weight.cpu() if so.weight is not None else torch.ones(ni.numel())

        def one_pass():
            O.reduce_sparse(x, ni, ci, w, k)
            O.reduce_batch_sparse(batch, ni, ci, k)

        return one_pass, self.nodes, "the whole measured input (N=1M, F=128)", 5.0

    def rooflines(self, dev):
        from tgp import kernels
        so, f, k = self.so, self.f, self.k
        idx = so.assign_index()
        # row + node_index + perm (+ weight) per assignment; no row_ptr table to read (one-to-one)
        alg = k * (4.0 * f + 8 + 4 + (4 if so.weight is not None else 0)) + k * 4.0 * f
        ms = event_time_ms(lambda: kernels.reduce_sparse(self.x, so.node_index, so.weight, idx), 50, dev)
        return roof_hbm("tgp::reduce_one_to_one_packed_kernel (gather-scale, one assignment per supernode, packed index)", alg, ms,
                        f"reduce_one_to_one_kernel:{self.which}")


class TopkConnect(Workload):
    """A5 + A6: TopK subgraph Connect on N = 1M, E = 10M, ratio 0.5 (count, one sync, fill)."""

    def __init__(self, ctx):
        from tgp.connect import SparseConnect
        from tgp.select import SelectOutput
        dev = ctx.dev
        g = torch.Generator(device=dev).manual_seed(0)
        n, self.ei = _big_graph(dev, g)
        self.ew = torch.ones(self.ei.size(1), device=dev)
        # r5: the SelectOutput comes from the REAL selector (Select is outside the metric, SURVEY 8(d), and runs once
        # here): what TopkSelect attaches to it -- since r5 the kept-node bitmap + rank directory of its compaction
        # pass -- is there exactly as in a pooler's forward.  Random scores keep a random half of the nodes.
        from tgp.select import TopkSelect
        torch.manual_seed(0)
        sel = TopkSelect(in_channels=8, ratio=0.5).to(dev)
        with torch.no_grad():
            self.so = sel(x=torch.randn(n, 8, device=dev, generator=g))
        k = int(self.so.num_supernodes)
        self.conn = SparseConnect()
        self.nodes, self.n, self.k = n, n, k
        self.name = "TopK subgraph Connect (ratio 0.5) on one N=1M E=10M graph: induced subgraph + relabel + filters"
        self.extra = {"edges": int(self.ei.size(1)), "num_supernodes": k, "nodes_counted": "input nodes per step",
                      "select_output": "made by TopkSelect itself (once, outside the timed step): carries the kept-node "
                                       "bitmap + rank directory its compaction pass writes",
                      "output_layout": "contiguous exact-size tensors (the default; capacity views are timed beside it)"}

    def step(self):
        return self.conn(self.ei, self.so, edge_weight=self.ew)

    def cpu_pass(self):
        """reference base_conn.py:79-82 + ops.py:338-419 on the measured inputs."""
        O = _oracle()
        ei, ew, ni, n, k = self.ei.cpu(), self.ew.cpu(), self.so.node_index.cpu(), self.n, self.k

        def one_pass():
            O.sparse_connect(ei, ew, ni, None, n, k)

        return one_pass, n, "the whole measured input (N=1M, E=10M)", 5.0

    def rooflines(self, dev):
        ei_out, _ = self.step()
        E, E2 = self.ei.size(1), ei_out.size(1)
        alg = E * 20.0 + self.n + self.k * 8.0 + E2 * 20.0  # SURVEY 8(d) A5+A6
        ms = event_time_ms(self.step, 30, dev)
        r = roof_hbm("tgp::subgraph_stage_kernel<SINGLE> + tgp::edges_compact_kernel (whole Connect call incl. its one "
                     "host read-back; contiguous exact-size outputs)", alg, ms, "subgraph_connect:topk1m")
        r["edges_out"] = E2
        r["contiguous_ms"] = round(ms, 5)
        from tgp import kernels
        with kernels.output_views():  # r4's layout: edge_index a [2,E'] view of the capacity-E buffer, no compaction
            ms_v = event_time_ms(self.step, 30, dev)
        r["capacity_views_ms"] = round(ms_v, 5)
        r["capacity_views_frac"] = round(alg / (ms_v * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        r["edge_index_contiguous"] = bool(ei_out.is_contiguous()
                                          and ei_out.untyped_storage().nbytes() <= max(2 * ei_out.numel() * 8, 512))
        return r


class GraclusC4(Workload):
    """C4: Graclus-shaped many-to-one S on one N=1M E=10M graph: Reduce (A1+A2) + coalesce Connect (A4+A6)."""

    def __init__(self, ctx, unsorted_edges=False):
        from tgp.connect import SparseConnect
        from tgp.reduce import BaseReduce
        from tgp.select import GraclusSelect
        dev = ctx.dev
        g = torch.Generator(device=dev).manual_seed(0)
        n, self.ei = _big_graph(dev, g, sort_rows=not unsorted_edges)
        self.f = 128
        self.ew = torch.ones(self.ei.size(1), device=dev)
        self.x = torch.randn(n, self.f, device=dev, generator=g)
        self.batch = torch.zeros(n, dtype=torch.long, device=dev)
        self.so = GraclusSelect()(self.ei, self.ew, num_nodes=n)
        self.red, self.conn = BaseReduce(), SparseConnect()
        self.nodes, self.n, self.k = n, n, self.so.num_supernodes
        self.name = ("Graclus precoarsening on one N=1M E=10M graph, F=128: Reduce + coalesce Connect "
                     "(BASELINE configs[3])")
        self.extra = {"num_supernodes": self.k, "edges": int(self.ei.size(1)),
                      "edge_order": "random" if unsorted_edges else "row-major sorted (PyG convention)",
                      "nodes_counted": "input nodes per step",
                      # what the SelectOutput brings along from GraclusSelect (outside the timed step), and what it does not
                      "from_select": ("int32 CSR offsets of the row-sorted edge list (the Connect skips its pass over the "
                                      "row array); S known to have row index 0..N-1 and unit values (the Reduce skips those "
                                      "two loads per assignment)" if not unsorted_edges else
                                      "S known to have row index 0..N-1 and unit values"),
                      "rebuilt_every_step": "the supernode -> members index (counting sort, ~50 us: so._drop_caches())"}

    def step(self):
        so = self.so
        so._drop_caches()  # rebuild the inverted index every step (no cross-step caching)
        xp, bp = self.red(self.x, so, batch=self.batch)
        return self.conn(self.ei, so, edge_weight=self.ew, batch_pooled=bp)

    def cpu_pass(self):
        """reference base_reduce.py:14-53,141-155 + base_conn.py:83-89 (relabel + coalesce) + ops.py:338-419."""
        O = _oracle()
        so, n, k = self.so, self.n, self.k
        x, batch, ei, ew = self.x.cpu(), self.batch.cpu(), self.ei.cpu(), self.ew.cpu()
        ni, ci = so.node_index.cpu(), so.cluster_index.cpu()
        w = so.weight.cpu() if so.weight is not None else torch.ones(ni.numel())

        def one_pass():
            O.reduce_sparse(x, ni, ci, w, k)
            bp = O.reduce_batch_sparse(batch, ni, ci, k)
            O.sparse_connect(ei, ew, ni, ci, n, k, batch_pooled=bp)

        return one_pass, n, "the whole measured input (N=1M, E=10M, F=128): Reduce + coalesce Connect", 8.0

    def rooflines(self, dev):
        from tgp import kernels
        so, f, n, k = self.so, self.f, self.n, self.k
        idx = so.assign_index()
        alg_r = n * (4.0 * f + 8 + 8 + 4) + k * 4.0 * f
        # (as BaseReduce calls it: a clustering's S has row index 0..N-1 and unit values, which the SelectOutput knows)
        ident, unit = bool(so.__dict__.get("_identity_nodes", False)), bool(so.__dict__.get("_unit_values", False))
        ms_r = event_time_ms(lambda: kernels.reduce_sparse(self.x, so.node_index, so.weight, idx, identity_source=ident,
                                                           unit_weight=unit), 50, dev)
        ei_out, _ = self.conn(self.ei, so, edge_weight=self.ew)
        E, E2 = self.ei.size(1), ei_out.size(1)
        alg_c = E * 20.0 + n * 8.0 + E2 * 20.0  # SURVEY 8(d) A4+A6
        ms_c = event_time_ms(lambda: self.conn(self.ei, so, edge_weight=self.ew), 30, dev)
        rc = roof_hbm("coalesce Connect (every kernel of the call + its one host read-back; index cached)", alg_c,
                      ms_c, "coalesce_connect:c4_graclus")
        rc["edges_out"] = E2
        return [rc, roof_hbm("tgp::reduce_sparse_vec4_kernel (gather-sum, index cached)", alg_r, ms_r,
                             "reduce_sparse_vec4_kernel:c4_graclus")]


def make_workload(which, ctx, args):
    if which in ("c2", "c5", "c2_f64"):
        return DenseDiffPool(which, ctx, unfused=args.unfused,
                             force_collective=os.environ.get("TGP_BENCH_FORCE_DIST") == "1")
    if which == "c3":
        return SmallGraphsMinCut(ctx, unfused=args.unfused)
    if which in ("topk1m", "c4_ndp"):
        return SparseReduceOnly(which, ctx)
    if which == "topk_connect":
        return TopkConnect(ctx)
    if which == "c4_graclus":
        return GraclusC4(ctx, unsorted_edges=args.unsorted_edges)
    if which in ("e2e_diff_c2", "e2e_mincut_c3"):
        return PoolerForward(which, ctx)
    if which in ("e2e_train_mincut_c3", "e2e_train_topk_c3", "e2e_train_mincut_c2", "e2e_train_diff_c2"):
        _, _, alias, scale = which.split("_")
        return PoolerTrainStep(ctx, alias=alias, scale=scale)
    if which in ("topk_batch", "graclus_batch", "topk_batch_fresh", "graclus_batch_fresh"):
        return TopkBatch(ctx, force_collective=os.environ.get("TGP_BENCH_FORCE_DIST") == "1", unfused=args.unfused,
                         which=which)
    raise ValueError(which)


def run_secondary(which, ctx, args):
    """One secondary workload: warm-up, median of 5 windows of 200 steps, rooflines (first = the one that bounds it)."""
    wl = make_workload(which, ctx, args)
    for _ in range(5):
        wl.step()
    # the previous workload's CPU-baseline leg left the device idle for seconds: run in until ~50 ms of this workload
    # have gone by (clock ramp), synchronising every 50 steps as the headline's settle phase does
    ctx.sync()
    t0 = time.perf_counter()
    while ctx.dist is None and time.perf_counter() - t0 < SETTLE_SECONDS / 2:
        for _ in range(50):
            wl.step()
        ctx.sync()
    steps = WINDOW_STEPS if which != "c5" else 50  # c5: 1.1 ms per step, 50 steps = 55 ms per window
    win = median_windows(ctx, wl.step, wl.drain, steps=steps)
    roofs = wl.rooflines(ctx.dev)
    roofs = roofs if isinstance(roofs, list) else [roofs]
    ms = win["ms_per_step_median"]
    out = {"workload": which, "name": wl.name, "ms_per_step": ms,
           "value": round(wl.nodes * ctx.world / (ms * 1e-3), 1), "unit": "nodes/s", "windows": win,
           "roofline": roofs[0], "config": wl.extra}
    if len(roofs) > 1:
        out["roofline_other"] = roofs[1:]
    if which in ("c2", "c5"):
        fl = roofs[0]["whole_step_flops"]
        out["whole_step"] = {"flops": fl, "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                             "frac_of_fp32_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
    if which == "c2_f64":
        fl = roofs[0]["whole_step_flops"]
        out["dtype"] = "f64"
        out["whole_step"] = {"flops": fl, "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                             "frac_of_fp64_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS, 4)}
    if ctx.world == 1 and ctx.rank == 0 and not args.no_cpu_baseline and hasattr(wl, "cpu_pass"):
        out["cpu_baseline"] = cpu_baseline(*wl.cpu_pass())
    del wl
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------ CPU baseline
def _oracle():
    """The CPU oracle: imported by this leg only, as the reported baseline (never by the measured path)."""
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import tgp_oracle as O
    return O


def _host_model():
    try:
        with open("/proc/cpuinfo") as fh:
            return next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except OSError:
        return ""


def cpu_baseline(one_pass, nodes, sample, budget_s=10.0, min_passes=2):
    """`one_pass()` = the CPU oracle (plain torch port of the reference algorithm) over `nodes` input
    nodes of the same workload; timed per pass on this box's host cores, all torch threads and again with one
    (the pattern of /root/reference/examples/time_and_mem_test.py:335-440: warm-up, then timed repetitions).
    `value` = nodes / MEDIAN pass time: the number to quote (a shared host's scheduling outliers moved a mean by
    1.7x between two runs of round 1; the median is the CPU's undisturbed rate, i.e. the figure that flatters the
    CPU); `mean_value` = all nodes / all time is kept beside it for the record."""
    def rate(seconds, warm):
        for _ in range(warm):
            one_pass()
        ts, t0 = [], time.perf_counter()
        while len(ts) < min_passes or time.perf_counter() - t0 < seconds:
            t1 = time.perf_counter()
            one_pass()
            ts.append(time.perf_counter() - t1)
            if len(ts) >= min_passes and time.perf_counter() - t0 >= seconds:
                break
        return nodes / statistics.median(ts), len(ts), time.perf_counter() - t0, nodes * len(ts) / sum(ts)

    threads = torch.get_num_threads()
    value, n, dt, mean_rate = rate(budget_s, 1)
    torch.set_num_threads(1)  # all host cores, and again with one thread
    try:
        one, n1, dt1, _ = rate(budget_s / 3, 0)
    finally:
        torch.set_num_threads(threads)
    return {"value": value, "unit": "nodes/s", "cores": threads, "kind": "port",
            "sample": f"{sample}; median of {n} passes of the CPU oracle in {dt:.1f}s",
            "quote": "value (nodes / median pass time, all torch threads)",
            "mean_value": mean_rate,  # passes / total time: includes the host's scheduling outliers
            "one_thread": {"value": one, "passes": n1, "seconds": round(dt1, 1)},
            "host": {"cpu_count": os.cpu_count(), "model": _host_model()}}


# ------------------------------------------------------------------------------------------------ output
HEADLINE_LIMIT = 4096  # bytes: the driver keeps a bounded tail of stdout and parses its LAST line (round 5's one
#                        23.5 KB line carrying 15 secondaries was cut and read as "parsed: null")
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "settle_steps", "ms_per_step",
                 "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "windows", "roofline",
                 "whole_step", "cpu_baseline", "detail")
_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "flops_per_launch",
              "bytes_per_launch", "avg_launch_ms", "whole_step_flops")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "mean_value", "one_thread", "host")
_PROSE = ("step", "nodes_counted", "tensors", "output_layout", "from_select", "rebuilt_every_step", "select_output",
          "gather", "bytes_counted", "peak_source", "quote")  # sentences: BASELINE.md / DESIGN.md describe each workload


def _clip(text, n):
    return text if len(text) <= n else text[: n - 1] + "~"


def _slim(obj, keys=None):
    """Numbers, flags and short labels of `obj`; prose fields (`_PROSE`) stay in the detail file."""
    out = {}
    for k, v in obj.items():
        if k in _PROSE or (keys is not None and k not in keys):
            continue
        out[k] = _clip(v, 120) if isinstance(v, str) else v
    return out


def headline_text(line):
    """The LAST stdout line: the contract fields + windows, roofline, whole_step, cpu_baseline; < HEADLINE_LIMIT."""
    out = {k: line[k] for k in HEADLINE_KEYS if k in line}
    out["config"] = _slim(line["config"])
    out["roofline"] = _slim(line["roofline"], _ROOF_KEYS)
    if "cpu_baseline" in line:
        out["cpu_baseline"] = _slim(line["cpu_baseline"], _CPU_KEYS)
    text = json.dumps(out)
    if len(text) >= HEADLINE_LIMIT:  # never expected; rather lose the optional blocks than the parse
        for k in ("windows", "whole_step", "detail"):
            out.pop(k, None)
        text = json.dumps(out)
    return text


def secondary_text(sec):
    """One earlier stdout line per secondary workload: short keys, no prose."""
    if "error" in sec:
        return json.dumps({"secondary": sec["workload"], "error": _clip(sec["error"], 300)})
    out = {"secondary": sec["workload"], "ms_per_step": sec["ms_per_step"], "value": sec["value"], "unit": sec["unit"]}
    if "dtype" in sec:
        out["dtype"] = sec["dtype"]
    out["windows"] = sec["windows"]
    out["roofline"] = _slim(sec["roofline"])
    if "roofline_other" in sec:
        out["roofline_other"] = [_slim(r) for r in sec["roofline_other"]]
    if "whole_step" in sec:
        out["whole_step"] = sec["whole_step"]
    if sec.get("config"):
        out["config"] = _slim(sec["config"])
    if "cpu_baseline" in sec:
        out["cpu_baseline"] = _slim(sec["cpu_baseline"], ("value", "unit", "cores", "kind", "one_thread"))
    return json.dumps(out)


def output_lines(line, secondary):
    """stdout of rank 0: one line per secondary workload, then the headline as the last line."""
    return [secondary_text(s) for s in secondary] + [headline_text(line)]


def write_detail(line, secondary):
    """Everything (prose included) as one JSON document beside the run's log: $TGP_BENCH_DETAIL, else
    gpurun_out/bench_detail.json under the repository; returns the path written, or None."""
    path = os.environ.get("TGP_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(dict(line, secondary=secondary), fh, indent=1)
        return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError:
        return None


# ------------------------------------------------------------------------------------------------ main
ALL = ["c2", "c2_f64", "c5", "c3", "c4_graclus", "c4_ndp", "topk1m", "topk_connect", "topk_batch", "graclus_batch",
       "topk_batch_fresh", "graclus_batch_fresh", "e2e_diff_c2", "e2e_mincut_c3", "e2e_train_mincut_c3",
       "e2e_train_topk_c3", "e2e_train_mincut_c2", "e2e_train_diff_c2"]
SECONDARY_DEFAULT = ["c5", "c2_f64", "topk1m", "topk_connect", "c4_graclus", "c4_ndp", "c3", "topk_batch", "graclus_batch",
                     "topk_batch_fresh", "graclus_batch_fresh", "e2e_diff_c2", "e2e_mincut_c3", "e2e_train_mincut_c3",
                     "e2e_train_topk_c3", "e2e_train_mincut_c2", "e2e_train_diff_c2"]
SHARDED = ("c5", "c2_f64", "c3", "topk_batch", "graclus_batch", "topk_batch_fresh", "graclus_batch_fresh", "e2e_diff_c2",
           "e2e_mincut_c3", "e2e_train_mincut_c3", "e2e_train_topk_c3", "e2e_train_mincut_c2", "e2e_train_diff_c2")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2", choices=ALL)
    ap.add_argument("--secondary", default=None,
                    help="comma-separated secondary workloads ('none' to skip); default: all of them on the c2 line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true",
                    help="dense workloads: call the Reduce and Connect operators one after the other")
    ap.add_argument("--unsorted-edges", action="store_true",
                    help="c4_graclus: leave the synthetic edge list in random order (forces the sort-based coalesce)")
    args = ap.parse_args()
    # stdout carries the JSON lines and nothing else; stderr stays free of the profiler's / autograd's advisory warnings
    import warnings
    warnings.filterwarnings("ignore")
    try:
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    except AttributeError:
        pass

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))  # no GPU call has been made in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # TGP_BENCH_FORCE_DIST=1: run the RCCL path with a one-rank group (a 1-GPU box can then exercise it)
    distributed = world > 1 or os.environ.get("TGP_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL world size {dist.get_world_size()} != --gpus {args.gpus}")
    ctx = Ctx(dev, rank, world, dist)

    from tgp import _native
    _native.lib()  # fails loudly when the HIP library is missing

    wl = make_workload(args.workload, ctx, args)
    if world > 1 and not wl.shards:
        raise SystemExit(f"workload {args.workload} is single-GPU (one giant graph does not shard: replicas only)")
    for _ in range(2):  # the barrier's own first calls set up RCCL state: not part of any step
        ctx.barrier()
        ctx.sync()
    wl.step()
    # CPython's generation-2 collection takes ~40 ms with torch imported and lands in whichever window crosses the
    # allocation threshold (seen: a 20-step window of 2.4 ms reported as 42 ms); collect now, keep start-up objects
    # out of later passes (tgp.freeze_gc()).  Done BEFORE the warm-up steps so that the timed window
    # follows them without a host-side pause in which the GPU would clock down.
    gc.collect()
    gc.freeze()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.warmup):
        wl.step()
    ctx.sync()
    # Settle: the first ~50 ms of device work after the idle start-up phase run 5-10 % slow (clock / power ramp; a
    # 20-step window right after 5 warm-up steps read 0.111 ms per step against 0.098 in every later window).  The
    # W warm-up steps are therefore followed by more UNTIMED steps of the same workload until ~50 ms of it have run;
    # the count is agreed across ranks (every step may hold a collective) and reported as "settle_steps".
    est = torch.tensor([(time.perf_counter() - t0) / max(args.warmup, 1)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(est, op=dist.ReduceOp.MAX)
    settle_steps = max(0, min(4000, int(SETTLE_SECONDS / max(float(est.item()), 1e-6)) - args.warmup))
    done = 0
    while done < settle_steps:  # in bursts of 50 with a synchronise in between: the host must not run hundreds of
        for _ in range(min(50, settle_steps - done)):  # launches ahead (the first submission after such a backlog
            wl.step()                                  # takes ~100 us instead of ~40, inside the window that follows)
        done += 50
        ctx.sync()
    if wl.drain is not None:
        wl.drain()  # warm-up ends with an empty gather bucket, its collective done (communicator set-up is not a step)
    dt = timed_window(ctx, wl.step, args.steps, wl.drain)  # THE contract window: exactly --steps steps
    win = median_windows(ctx, wl.step, wl.drain, steps=max(WINDOW_STEPS if args.workload != "c5" else 50, 1))
    roofs = wl.rooflines(dev)
    roofs = roofs if isinstance(roofs, list) else [roofs]
    value = wl.nodes * world * args.steps / dt

    if args.secondary is None:
        sec = SECONDARY_DEFAULT if args.workload == "c2" else []
    else:
        sec = [s for s in args.secondary.split(",") if s and s != "none"]
    if world > 1:
        sec = [s for s in sec if s in SHARDED]  # the graph-sharded ones
    headline_cfg = dict(workload=wl.name, **(wl.extra or {}), parallelism=f"graph-sharded x{world}")
    dims = (wl.B, wl.N, wl.K, wl.F) if isinstance(wl, DenseDiffPool) else None
    cpu_job = None  # host copies of the measured inputs now; the CPU passes run after the GPU work of the line
    if world == 1 and rank == 0 and not args.no_cpu_baseline and hasattr(wl, "cpu_pass"):
        cpu_job = wl.cpu_pass()
    del wl
    torch.cuda.empty_cache()
    secondary = []
    for s in sec:
        try:
            secondary.append(run_secondary(s, ctx, args))
        except Exception as exc:  # a failing secondary must not lose the headline; it is reported, not hidden
            secondary.append({"workload": s, "error": f"{type(exc).__name__}: {exc}"})
            torch.cuda.empty_cache()
        if rank == 0:
            print(secondary_text(secondary[-1]), flush=True)  # its own line, as soon as it is measured

    if rank == 0:
        line = {
            "metric": METRIC, "value": round(value, 1), "unit": "nodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "settle_steps": settle_steps, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": headline_cfg,
            "windows": win, "roofline": roofs[0],
        }
        if len(roofs) > 1:
            line["roofline_other"] = roofs[1:]
        if dims is not None:
            fl = roofs[0]["whole_step_flops"]
            ms = win["ms_per_step_median"]
            line["whole_step"] = {"flops": fl, "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                                  "frac_of_fp32_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
        if cpu_job is not None:
            line["cpu_baseline"] = cpu_baseline(*cpu_job)
        detail = write_detail(line, secondary)
        if detail:
            line["detail"] = detail
        print(headline_text(line), flush=True)  # the LAST stdout line: what the driver parses
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

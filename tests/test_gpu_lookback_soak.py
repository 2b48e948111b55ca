"""The one-launch look-back kernels (csrc/lookback.h: sparse_pool_small_kernel, cr_member_single_kernel,
cr_scan_publish_kernel) under contention: thousands of calls while a second stream keeps every CU busy with large
products.  A look-back that meets its spin bound turns into a refusal (the call declines and the staged operators take
over): none may occur, and every result must equal the staged operators'.  Run once with tile = blockIdx.x (the default)
and once, in a child process, with TGP_LOOKBACK_TICKET=1 (tile = arrival number from an atomic ticket)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _soak(calls_small, steps_big):
    sys.path.insert(0, ROOT)
    import bench
    from tgp import kernels
    dev = torch.device("cuda:0")
    ctx = bench.Ctx(dev, 0, 1, None)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    b = torch.randn(4096, 4096, device=dev)
    busy = torch.empty_like(a)

    def keep_busy(n):
        with torch.cuda.stream(side):
            for _ in range(n):
                torch.mm(a, b, out=busy)  # ~1 ms of every CU each

    report = {}
    for which in ("topk_batch", "graclus_batch"):
        wl = bench.TopkBatch(ctx, which=which)
        with torch.no_grad():
            ref = wl.staged((wl.x, wl.ei, wl.ew, wl.so))
        declined = mismatch = 0
        for i in range(calls_small):
            if i % 200 == 0:
                keep_busy(12)
            with torch.no_grad():
                out = wl.pool.reduce_connect(wl.x, wl.ei, wl.ew, wl.so, wl.batch)
            if out is None:
                declined += 1
                kernels._SPS_DECLINED.clear()
            elif i % 500 == 0:
                xp, bp, ei, ew = out
                if not (torch.equal(xp, ref[0]) and torch.equal(ei, ref[1]) and torch.equal(ew, ref[2])):
                    mismatch += 1
        torch.cuda.synchronize()
        report[which] = (declined, mismatch)
    # the big-graph look-back kernels (coalesce Connect's member / survivor scans) on a 200k-node graph
    g = torch.Generator(device=dev).manual_seed(0)
    n = 200_000
    ei = torch.randint(0, n, (2, 1_000_000), device=dev, generator=g)
    ei = ei[:, ei[0] != ei[1]]
    ei = torch.cat([ei, ei.flip(0)], 1)
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])].contiguous()
    from tgp.connect import SparseConnect
    from tgp.select import GraclusSelect
    so = GraclusSelect()(ei, None, num_nodes=n)
    conn = SparseConnect()
    ew = torch.rand(ei.size(1), device=dev, generator=g) + 0.5
    ref = conn(ei, so, edge_weight=ew)
    mismatch = 0
    for i in range(steps_big):
        if i % 50 == 0:
            keep_busy(12)
        out = conn(ei, so, edge_weight=ew)
        if i % 25 == 0 and not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])):
            mismatch += 1
    torch.cuda.synchronize()
    report["coalesce"] = (0, mismatch)
    return report


def test_lookback_soak_beside_a_busy_stream():
    report = _soak(10_000, 400)
    for which, (declined, mismatch) in report.items():
        assert declined == 0 and mismatch == 0, report


def test_lookback_soak_with_arrival_tickets():
    env = dict(os.environ, TGP_LOOKBACK_TICKET="1")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); "
            "import test_gpu_lookback_soak as t; r = t._soak(2000, 100); print(r); "
            "sys.exit(0 if all(d == 0 and m == 0 for d, m in r.values()) else 1)"
            % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "torch-geometric-pool_amd")))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

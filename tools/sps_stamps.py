#!/usr/bin/env python3
"""Phase clocks of tgp_sparse_pool_small_f32 (diagnostic build: make -C torch-geometric-pool_amd/csrc stamps, loaded
through TGP_HIP_LIB): per workgroup, the constant-rate clock (100 MHz) at start / after the searches / after the
Reduce part / after the count pass / after the barrier / after the look-back / at the end.
usage: TGP_HIP_LIB=torch-geometric-pool_amd/lib/libtgp_hip_stamps.so python tools/sps_stamps.py [topk|graclus]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import bench  # noqa: E402
from tgp import kernels  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "topk"
dev = torch.device("cuda:0")
wl = bench.TopkBatch(bench.Ctx(dev, 0, 1, None), which="topk_batch" if which == "topk" else "graclus_batch")
for _ in range(20):
    wl.compute()
torch.cuda.synchronize()
state = next(iter(kernels._SPS_STATE.values()))
waves = 8
tiles = (2048 + waves - 1) // waves
st = state.status[2 + tiles: 2 + 9 * tiles].view(tiles, 8).cpu().double() * 0.01  # us
t0 = st[:, 0].min()
names = ["start", "searched", "reduced", "counted", "barrier", "look-back", "end"]
print(f"{which}: {tiles} workgroups; us since the first workgroup's start (min / median / max over workgroups)")
for k, nme in enumerate(names):
    col = st[:, k] - t0
    print(f"  {nme:10s} {col.min():7.2f} {col.median():7.2f} {col.max():7.2f}")
d = st[:, 1:7] - st[:, 0:6]
print("phase durations per workgroup, median us:", [round(float(v), 2) for v in d.median(0)[0]])
print("by tile (every 16th): start, end:", [(round(float(st[i, 0] - t0), 1), round(float(st[i, 6] - t0), 1)) for i in range(0, tiles, 16)])

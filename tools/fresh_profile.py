"""Where a pooler forward spends its time when every call sees NEW tensor objects (verdict r4 item 2): kernels of one
fresh call (torch.profiler), host profile (cProfile) of 300 fresh calls, wall time fresh vs same tensors.

    python tools/fresh_profile.py [topk_c3] [graclus_c3] [mincut_c3]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2e_launches import CASES, batch_graphs, resolve_sizes, count_kernels  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")


def wall(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name in ([a for a in sys.argv[1:] if a in CASES] or ["topk_c3", "graclus_c3", "mincut_c3"]):
    alias, kw, sizes, deg, f = CASES[name]
    sizes = resolve_sizes(sizes)
    x, ei, batch = batch_graphs(sizes, deg, f)
    pooler = get_pooler(alias, **kw).to(dev).eval()
    R = 24
    copies = [(x.clone(), ei.clone(), batch.clone()) for _ in range(R)]
    state = {"i": 0}

    def same():
        with torch.no_grad():
            pooler(x=x, adj=ei, batch=batch)

    def fresh():  # rotates through more distinct tensor objects than any memo holds: every call misses
        state["i"] = (state["i"] + 1) % R
        xx, ee, bb = copies[state["i"]]
        with torch.no_grad():
            pooler(x=xx, adj=ee, batch=bb)

    a, b = wall(same), wall(fresh)
    print(f"== {name}: same tensors {a:.3f} ms, fresh tensor objects {b:.3f} ms", flush=True)
    print("   kernels of one fresh call:")
    count_kernels(fresh, list_kernels=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        fresh()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative")
    st.print_stats(28)

"""Host profile of a whole sparse pooler forward (topk | graclus) on 2048 PROTEINS-shaped graphs: wall time per forward and
cProfile of 2000 forwards.  python tools/profile_sparse_forward.py [topk|graclus]"""
import cProfile, pstats, sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import e2e_launches as E
from tgp.poolers import get_pooler
which = sys.argv[1] if len(sys.argv) > 1 else "topk"
torch.manual_seed(0)
g = torch.Generator().manual_seed(0)
sizes = torch.randint(20, 61, (2048,), generator=g).tolist()
x, ei, batch = E.batch_graphs(sizes, 4, 32)
kw = dict(in_channels=32, ratio=0.5) if which == "topk" else {}
pooler = get_pooler(which, **kw).to(x.device).eval()
def fwd():
    with torch.no_grad():
        return pooler(x=x, adj=ei, batch=batch)
for _ in range(50): fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): fwd()
torch.cuda.synchronize()
print(f"{which}: {(time.perf_counter()-t0)/2000*1e6:.1f} us per forward")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): fwd()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)

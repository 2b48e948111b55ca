"""How many graphs of the ownership test's batch take NDPSelect's random fallback (cut < 0.5), with and without the Lanczos
warm start of the one-wave kernel (child process with TGP_NDP_LANCZOS=0)."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def run():
    from test_gpu_sparse_pool_small import _er_batch
    from tgp.select import NDPSelect
    dev = torch.device("cuda:0")
    x, ei, ew, batch = _er_batch(200, 5, 60, 8, 7, dev)
    outs = []
    for seed in (3, 4):
        torch.manual_seed(seed)
        so = NDPSelect()(edge_index=ei, edge_weight=ew, batch=batch, num_nodes=x.size(0))
        info = so._partition_info.cpu()
        outs.append((so.num_supernodes, int((info == -1).sum()), int(info.max()), (info == -1).nonzero().view(-1).tolist()))
    return outs


if __name__ == "__main__":
    print(os.environ.get("TGP_NDP_LANCZOS", "1"), run())
    if len(sys.argv) == 1:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, TGP_NDP_LANCZOS="0"))

import torch
dev=torch.device('cuda:0')
x=torch.randn(1_000_000,128,device=dev)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
y=torch.empty_like(x)
print('sum      ', t(lambda: x.sum()), 'us ->', 512/ t(lambda: x.sum()), 'TB/s')
print('sum(dim1)', t(lambda: x.sum(1)), 'us')
print('copy     ', t(lambda: y.copy_(x)), 'us ->', 1024/t(lambda: y.copy_(x)), 'TB/s')
import sys; sys.path.insert(0,'torch-geometric-pool_amd')
from tgp import kernels as K
w=torch.randn(128,device=dev)
print('row_dot  ', t(lambda: K.row_dot(x,w)), 'us ->', 512/t(lambda: K.row_dot(x,w)), 'TB/s')
print('topkscore', t(lambda: K.topk_score(x,w,True)), 'us')

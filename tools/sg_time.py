import sys, torch, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/torch-geometric-pool_amd")
import bench
dev = torch.device("cuda:0")
wl = bench.TopkConnect(bench.Ctx(dev, 0, 1, None))
for _ in range(10): wl.step()
torch.cuda.synchronize()
print("ms", bench.event_time_ms(wl.step, 30, dev))

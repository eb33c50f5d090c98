"""Host-side cost of enqueueing one DiT-L/2 forward (python + ctypes + HIP launch calls) vs its device time: how many lanes one
python thread can feed.  usage: python tools/host_launch_cost.py [network_batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models
dev = torch.device('cuda', 0)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dit, _ = build_models(dev, 'DiT-L/2', 'DiT2-L/2')
x = torch.randn(Bn // 2, 12, 32, 32, device=dev)
ctx = torch.cat([torch.zeros(Bn // 2, 77, 768, device=dev), torch.randn(Bn // 2, 77, 768, device=dev)])
cc = dit.prepare_context(ctx)
mc = dit.prepare_timesteps(torch.full((4, Bn), 500.0))
t = torch.full((Bn,), 500.0, device=dev)
sc = torch.ones(Bn, device=dev)
for _ in range(3):
    dit(x, t, context_cache=cc, in_scale=sc, mod_cache=(mc, 0))
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    dit(x, t, context_cache=cc, in_scale=sc, mod_cache=(mc, 0))
t_host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print("network batch %d: host enqueue %.2f ms per forward, device %.2f ms per forward (%d forwards back to back)" % (Bn, t_host * 1e3, t_all * 1e3, n))

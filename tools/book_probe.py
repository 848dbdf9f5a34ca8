import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from wheeledlab_amd.rl.ppo import _finished_episodes
K, n = 128, 4096
dev = "cuda:0"
rew = torch.rand(K, n, device=dev)
done = torch.rand(K, n, device=dev) < 0.01
cr, cl = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
acts = torch.rand(K, n, 2, device=dev)
vals = torch.rand(K + 1, n, device=dev)
to = torch.rand(K, n, device=dev) < 0.005
def once():
    ok = bool(torch.isfinite(acts).all())
    ret, length, a, b = _finished_episodes(rew, done, cr, cl)
    x = ret[-100:].tolist(); y = length[-100:].tolist()
    m = float(rew.mean())
    rew.add_(0.99 * vals[:-1] * to)
for _ in range(3): once()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): once()
torch.cuda.synchronize()
print("bookkeeping per iteration: %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))

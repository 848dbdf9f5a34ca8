"""dev probe: time wl_ppo_gradients (operand build + gradient kernel + reduction) for the builds in gpurun_variants/"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A

DEV = "cuda:0"
B = 524288
for path in sorted(glob.glob("gpurun_variants/lib_*.so")) or [None]:
    A._lib = None
    A.load(path)
    from wheeledlab_amd.rl.ppo import ActorCritic, FusedPpoStep, PPO
    torch.manual_seed(0)
    ac = ActorCritic(14, 14, 2).to(DEV)
    ppo = PPO(ac, fused_update=False)
    fz = FusedPpoStep(ac, ppo)
    r = lambda *s: torch.randn(*s, device=DEV)
    flat = dict(obs=r(B, 14), actions=r(B, 2), mu=r(B, 2), logp=r(B) - 2, adv=r(B), returns=r(B), values=r(B))
    perm = torch.randperm(B, device=DEV).to(torch.int32)
    sig = ac.std.detach().clone()
    for mb in (131072, 16384):
        for _ in range(3):
            fz.gradients(flat, perm, 0, mb, sig)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fz.gradients(flat, perm, 0, mb, sig)
        e1.record()
        torch.cuda.synchronize()
        print(os.path.basename(path or "default"), "mb", mb, f"{e0.elapsed_time(e1) * 1e3 / 20:.1f} us per gradients() call")

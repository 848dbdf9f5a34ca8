"""us per policy step (actor -> sample -> log-prob, critic value) of the one-launch kernel (wl_actor_critic_act) and of the
same step in torch eager (rl.ppo.ActorCritic.act + get_actions_log_prob + evaluate), for the elevation / visual
observation widths."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
if os.environ.get("WL_LIB"):
    A.load(os.environ["WL_LIB"])   # a variant build (gpurun_variants/lib_*.so)
from wheeledlab_amd.rl.ppo import ActorCritic

dev = "cuda:0"
res = {}
for D, act in ((689, "relu"), (3208, "elu")):
    for n in (512, 1024, 4096, 16384):
        ac = ActorCritic(D, D, 2, activation=act).to(dev)
        view = ac.fused()
        obs = torch.randn(n, D, device=dev)
        a, mu = torch.empty(n, 2, device=dev), torch.empty(n, 2, device=dev)
        logp, val = torch.empty(n, device=dev), torch.empty(n, device=dev)

        def fused(k):
            view.planes = False
            view.act(obs, a, mu, logp, val, 1, k)

        def planes(k):
            view.planes, view.planes_two_launch = True, False
            view.act(obs, a, mu, logp, val, 1, k, planes_fresh=k > 0)

        def planes2(k):
            view.planes, view.planes_two_launch = True, True
            view.act(obs, a, mu, logp, val, 1, k, planes_fresh=k > 0)

        def eager(k):
            with torch.inference_mode():
                x = ac.act(obs)
                ac.get_actions_log_prob(x)
                ac.evaluate(obs)

        for name, fn in (("kernel", fused), ("planes", planes), ("planes2", planes2)) + (() if os.environ.get("WL_LIB") else (("torch", eager),)):
            for k in range(20):
                fn(k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(200):
                fn(k)
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}:D{D}:n{n}"] = round(e0.elapsed_time(e1) * 1e3 / 200, 2)
print(json.dumps(res))

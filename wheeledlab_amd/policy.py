"""Policy in the loop: the RSL-RL ActorCritic MLPs of the drift agents and the rollout storage the fused collection
kernel fills (reference: rsl_rl ActorCritic / RolloutStorage as driven by
wheeledlab_rl/utils/modified_rsl_rl_runner.py:70-80 with wheeledlab_tasks/drifting/config/agents/mushr/rsl_rl_ppo_cfg.py).

Everything that computes runs in the HIP library (`wl_mlp_forward`, `wl_drift_rollout_policy`); this module only owns
the tensors and mirrors the names a rsl_rl user expects (`actor`, `critic`, `std`, `observations`, `actions`, `mu`,
`actions_log_prob`, `values`, `rewards`, `dones`).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi as A

ACTIVATIONS = {"relu": A.ACT_RELU, "elu": A.ACT_ELU}


class Mlp:
    """in -> 64 -> 64 -> out with torch nn.Linear weight layout; evaluated on the f32 matrix pipe."""

    def __init__(self, in_dim: int, out_dim: int, activation: str = "elu", device="cuda:0", hidden: int = 64,
                 generator: torch.Generator | None = None):
        if hidden != 64:
            raise ValueError("the matrix-pipe MLP is specialised for hidden dims [64, 64] (rsl_rl_ppo_cfg.py:14-15)")
        self.in_dim, self.out_dim, self.hidden, self.activation = int(in_dim), int(out_dim), 64, activation
        self.device = torch.device(device)

        def init(o, i):   # torch.nn.Linear default init: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias
            k = 1.0 / (i ** 0.5)
            w = (torch.rand(o, i, generator=generator) * 2 - 1) * k
            b = (torch.rand(o, generator=generator) * 2 - 1) * k
            return w.to(self.device).contiguous(), b.to(self.device).contiguous()

        self.w1, self.b1 = init(64, self.in_dim)
        self.w2, self.b2 = init(64, 64)
        self.w3, self.b3 = init(self.out_dim, 64)

    @classmethod
    def from_sequential(cls, seq, activation: str, device="cuda:0"):
        """adopt the three nn.Linear layers of a torch Sequential (rsl_rl's `actor_critic.actor` / `.critic`)"""
        lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
        if len(lin) != 3 or lin[0].out_features != 64 or lin[1].out_features != 64:
            raise ValueError("expected Linear(in,64) / Linear(64,64) / Linear(64,out)")
        self = cls.__new__(cls)
        self.in_dim, self.out_dim, self.hidden, self.activation = lin[0].in_features, lin[2].out_features, 64, activation
        self.device = torch.device(device)
        for i, m in enumerate(lin, 1):
            setattr(self, f"w{i}", m.weight.detach().to(self.device, torch.float32).contiguous())
            setattr(self, f"b{i}", m.bias.detach().to(self.device, torch.float32).contiguous())
        return self

    def struct(self) -> A.WlMlp:
        return A.WlMlp(self.w1.data_ptr(), self.b1.data_ptr(), self.w2.data_ptr(), self.b2.data_ptr(),
                       self.w3.data_ptr(), self.b3.data_ptr(), self.in_dim, self.out_dim, self.hidden,
                       ACTIVATIONS[self.activation])

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """y[..., out_dim] = mlp(x[..., in_dim]) (wl_mlp_forward)"""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == self.in_dim and x.device == self.device
        y = torch.empty(*x.shape[:-1], self.out_dim, dtype=torch.float32, device=self.device)
        s = self.struct()
        A.check(A.load().wl_mlp_forward(C.byref(s), x.numel() // self.in_dim, x.data_ptr(), y.data_ptr(),
                                        C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "wl_mlp_forward")
        return y


class ActorCritic:
    """`actor`, `critic`, `std` as in rsl_rl's ActorCritic (init_noise_std 1.0, rsl_rl_ppo_cfg.py:13)"""

    def __init__(self, num_obs: int = 14, num_actions: int = 2, activation: str = "elu", init_noise_std: float = 1.0,
                 device="cuda:0", seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        self.actor = Mlp(num_obs, num_actions, activation, device, generator=g)
        self.critic = Mlp(num_obs, 1, activation, device, generator=g)
        self.std = torch.full((num_actions,), float(init_noise_std), dtype=torch.float32, device=device)

    # policy step with the first layer on the bf16 matrix pipe (wl_actor_critic_act_planes): None = by size, True / False = forced
    planes: bool | None = None
    planes_two_launch: bool | None = None   # None = by size; the bf16 path's form: split-K partial sums + a tail launch, or one launch

    @staticmethod
    def planes_form(n: int, D: int):
        """which policy-step kernel for n rows of D features: None = the one-launch f32 kernel, "one" / "two" = the bf16 forms.
        Measured (tools/act_probe.py, us per step: f32 / bf16 one launch / bf16 two launches):
          D = 3208:  512 rows 26.4 / 28.6 / 18.1,  1024: 27.9 / 28.6 / 21.2,  4096: 44.5 / 35.7 / 42.0,  16 384: 126 / 141 / 163
          D =  689:  512 rows 10.7 / 10.2 / 12.4,  1024: 10.9 / 10.2 / 12.9,  4096: 15.6 / 15.8 / 17.9,  16 384: 38.5 / 32.5 / 45.3
        At the agents' sizes the step is a chain of dependent round trips (launch, first operands, the feature shares, the LDS
        fold, the 64-64 tail), so the forms differ by a few us; the f32 kernel stays the default where it is not clearly beaten."""
        if D >= 1024:
            return "two" if n <= 2048 else "one" if n <= 8192 else None
        return "one" if n >= 8192 and D >= 64 else None

    def _scratch(self, n: int):
        """the bf16 form's device scratch (weight planes, split-K partial sums) for up to n rows"""
        D = self.actor.in_dim
        sc = getattr(self, "_act_scratch", None)
        if sc is None or sc[0].rows_capacity < n or sc[0].dp != (D + 63) // 64 * 64:
            dp, dev = (D + 63) // 64 * 64, self.std.device
            splits = (dp + 127) // 128      # 128 features per split, whatever n: a shard reproduces its rows of the batch
            w_hi, w_lo = (torch.zeros(128, dp, dtype=torch.int16, device=dev) for _ in range(2))
            part = torch.zeros(splits, n, 128, dtype=torch.float32, device=dev)
            sc = (A.WlActScratch(w_hi.data_ptr(), w_lo.data_ptr(), part.data_ptr(), dp, splits, n, 0), w_hi, w_lo, part)
            self._act_scratch = sc
        return sc[0]

    # rows of the WHOLE batch when this view evaluates one env shard of it (world x local rows): the kernel form -- and with it
    # the summation order and precision of layer 1 -- is chosen from this count, so that a shard reproduces its rows of the
    # one-process batch (None: the local row count)
    global_rows: int | None = None

    def act(self, obs: torch.Tensor, actions: torch.Tensor, mu: torch.Tensor, log_prob: torch.Tensor, values: torch.Tensor,
            seed: int, step: int, env_offset: int = 0, deterministic: bool = False, nets: int = 3, planes_fresh: bool = False):
        """One policy step for observations of ANY width (wl_actor_critic_act: one launch, f32 matrix pipe; or, for large
        row x width products, wl_actor_critic_act_planes: two launches, first layer on the bf16 pipe): fills `actions`, `mu`
        [n, 2], `log_prob`, `values` [n] (rows of a RolloutStorage) from obs [n, D]; the draw is keyed by
        (seed, env_offset + row, step).  nets = 1: only the actor's half (actions / mu / log_prob; `values` may be None),
        nets = 2: only the critic's (`values`; the other outputs may be None).  planes_fresh: the weight planes of the bf16
        form are current (the caller has run this method since the last parameter update) -- skips their rebuild."""
        n, D = obs.shape
        assert D == self.actor.in_dim == self.critic.in_dim and obs.dtype == torch.float32 and obs.stride(1) == 1
        needed = (((actions, (n, 2)), (mu, (n, 2)), (log_prob, (n,))) if nets & 1 else ()) + (((values, (n,)),) if nets & 2 else ())
        for t, shape in needed:
            assert t.shape == shape and t.dtype == torch.float32 and t.is_contiguous() and t.device == obs.device
        ptr = lambda t: None if t is None else t.data_ptr()
        key = (self.actor.w1.data_ptr(), self.critic.w1.data_ptr(), self.std.data_ptr())
        if getattr(self, "_act_key", None) != key:   # the structs only hold pointers: rebuilt when the tensors are replaced
            self._act_key, self._act_structs, self._act_fn = key, (self.actor.struct(), self.critic.struct()), A.load().wl_actor_critic_act
            self._act_scratch = None
        a, c = self._act_structs
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        form = self.planes_form(max(n, self.global_rows or 0), D)
        if self.planes if self.planes is not None else form is not None:
            two = self.planes_two_launch if self.planes_two_launch is not None else form != "one"
            lib = A.load()
            sc = self._scratch(n)
            sc.reserved = int(two)
            # the planes live in the scratch: a caller's "fresh" only holds for the scratch object they were last built into
            # (a larger batch on the same view reallocates it -- zero-filled planes would make layer 1 bias-only)
            if not planes_fresh or getattr(self, "_planes_built_for", None) is not sc:
                A.check(lib.wl_actor_critic_planes(C.byref(a), C.byref(c), C.byref(sc), stream), "wl_actor_critic_planes")
                self._planes_built_for = sc
            A.check(lib.wl_actor_critic_act_planes(C.byref(a), C.byref(c), self.std.data_ptr(), n, obs.data_ptr(), obs.stride(0),
                                                   ptr(actions), ptr(mu), ptr(log_prob), ptr(values), int(env_offset), int(seed),
                                                   int(step), int(bool(deterministic)), int(nets), C.byref(sc), stream),
                    "wl_actor_critic_act_planes")
            return
        A.check(self._act_fn(C.byref(a), C.byref(c), self.std.data_ptr(), n, obs.data_ptr(), obs.stride(0),
                                             ptr(actions), ptr(mu), ptr(log_prob), ptr(values),
                                             int(env_offset), int(seed), int(step), int(bool(deterministic)), int(nets), stream),
                "wl_actor_critic_act")

    def values_batched(self, obs: torch.Tensor, out: torch.Tensor, chunk: int = 65536):
        """critic(obs) -> out [N] for a long batch of stored observations (the K + 1 rows of a rollout): the first layer as the
        streaming bf16 contraction with one sum per row and unit (wl_actor_critic_act_planes, form 2), `chunk` rows per call"""
        N, D = obs.shape
        assert D == self.critic.in_dim and obs.is_contiguous() and out.shape == (N,) and out.is_contiguous()
        if D < 64:
            return self.values(obs, out)
        key = (self.actor.w1.data_ptr(), self.critic.w1.data_ptr(), D, chunk)
        vb = getattr(self, "_values_scratch", None)
        if vb is None or vb[0] != key:
            dp, dev = (D + 63) // 64 * 64, self.std.device
            w_hi, w_lo = (torch.zeros(128, dp, dtype=torch.int16, device=dev) for _ in range(2))
            part = torch.zeros(chunk, 128, dtype=torch.float32, device=dev)
            vb = self._values_scratch = (key, A.WlActScratch(w_hi.data_ptr(), w_lo.data_ptr(), part.data_ptr(), dp, 1, chunk, 2), w_hi, w_lo,
                                         part, (self.actor.struct(), self.critic.struct()))
        sc, (a, c) = vb[1], vb[5]
        lib, stream = A.load(), C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        A.check(lib.wl_actor_critic_planes(C.byref(a), C.byref(c), C.byref(sc), stream), "wl_actor_critic_planes")
        for r0 in range(0, N, chunk):
            m = min(chunk, N - r0)
            A.check(lib.wl_actor_critic_act_planes(C.byref(a), C.byref(c), self.std.data_ptr(), m, obs[r0:r0 + m].data_ptr(), obs.stride(0),
                                                   None, None, None, out[r0:r0 + m].data_ptr(), 0, 0, 0, 0, 2, C.byref(sc), stream),
                    "wl_actor_critic_act_planes")
        return out

    def values(self, obs: torch.Tensor, out: torch.Tensor):
        """critic(obs) -> out [n] for observations of any width (the critic's half of wl_actor_critic_act)"""
        self.act(obs, None, None, None, out, 0, 0, nets=2)
        return out


def _normalise_over_ranks(adv: torch.Tensor) -> torch.Tensor:
    """(adv - mean) / (std + 1e-8) over the GLOBAL batch: with one process per GPU the moments are all-reduced (one
    3-float collective), so an N-rank run normalises exactly like the one big batch (rsl_rl is single-process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return (adv - adv.mean()) / (adv.std() + 1e-8)
    m = torch.stack([adv.sum().double(), (adv.double() ** 2).sum(), torch.tensor(float(adv.numel()), device=adv.device, dtype=torch.float64)])
    dist.all_reduce(m, op=dist.ReduceOp.SUM)
    mean = m[0] / m[2]
    var = (m[1] - m[2] * mean * mean) / (m[2] - 1.0)          # unbiased, like torch.std
    return ((adv - mean.float()) / (var.clamp_min(0).sqrt().float() + 1e-8))


class RolloutStorage:
    """[K(+1), n, ...] transition rows, named as rsl_rl's RolloutStorage"""

    def __init__(self, n_steps: int, n_envs: int, obs_dim: int = 14, num_actions: int = 2, device="cuda:0"):
        K, n, dev = int(n_steps), int(n_envs), torch.device(device)
        self.n_steps, self.n_envs = K, n
        self.observations = torch.zeros(K + 1, n, obs_dim, dtype=torch.float32, device=dev)   # row K = last obs
        self.actions = torch.zeros(K, n, num_actions, dtype=torch.float32, device=dev)
        self.mu = torch.zeros(K, n, num_actions, dtype=torch.float32, device=dev)
        self.actions_log_prob = torch.zeros(K, n, dtype=torch.float32, device=dev)
        self.rewards = torch.zeros(K, n, dtype=torch.float32, device=dev)
        self.terminated = torch.zeros(K, n, dtype=torch.bool, device=dev)
        self.time_outs = torch.zeros(K, n, dtype=torch.bool, device=dev)
        self.dones = torch.zeros(K, n, dtype=torch.long, device=dev)
        self.values = torch.zeros(K + 1, n, dtype=torch.float32, device=dev)                 # row K = bootstrap value

    def struct(self, start: int = 0) -> A.WlPolicyRollout:
        """the rows from step `start` on ([step][env] major, so a suffix of every tensor is itself a valid storage)"""
        return A.WlPolicyRollout(self.observations[start:].data_ptr(), self.actions[start:].data_ptr(),
                                 self.mu[start:].data_ptr(), self.actions_log_prob[start:].data_ptr(),
                                 self.rewards[start:].data_ptr(), self.terminated[start:].data_ptr(),
                                 self.time_outs[start:].data_ptr(), self.dones[start:].data_ptr())

    def bootstrap_time_outs(self, gamma: float):
        """rsl_rl PPO.process_env_step: rewards += gamma * V(obs_t) * time_outs (in place, once per collection)"""
        self.rewards += gamma * self.values[:-1] * self.time_outs

    def compute_returns(self, gamma: float = 0.99, lam: float = 0.95):
        """GAE exactly as rsl_rl RolloutStorage.compute_returns; returns (returns, normalised advantages) [K, n].
        On a GPU the backward recursion is one launch (wl_gae: one lane per env); on the CPU it is the torch loop."""
        K = self.n_steps
        adv = torch.empty_like(self.rewards)
        if self.rewards.is_cuda:
            returns = torch.empty_like(self.rewards)
            A.check(A.load().wl_gae(K, self.n_envs, self.rewards.data_ptr(), self.values.data_ptr(), self.dones.data_ptr(),
                                    float(gamma), float(lam), returns.data_ptr(), adv.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream(self.rewards.device).cuda_stream)), "wl_gae")
        else:
            last = torch.zeros(self.n_envs, dtype=torch.float32, device=self.rewards.device)
            not_done = 1.0 - self.dones.to(torch.float32)
            for t in reversed(range(K)):
                delta = self.rewards[t] + not_done[t] * gamma * self.values[t + 1] - self.values[t]
                last = delta + not_done[t] * gamma * lam * last
                adv[t] = last
            returns = adv + self.values[:-1]
        return returns, _normalise_over_ranks(adv)

"""PPO learner + on-policy runner for the registered tasks.

Host-side mirror of what the reference drives through rsl-rl-lib (wheeledlab_rl/utils/modified_rsl_rl_runner.py:24-128
with the agent cfgs under wheeledlab_tasks/*/config/agents): same class / method names (`ActorCritic`, `PPO.update`,
`OnPolicyRunner.learn / save / load / get_inference_policy`), same hyper-parameter fields, same checkpoint keys
(`model_state_dict`, `optimizer_state_dict`, `iter`, `infos`).  rsl-rl-lib is not vendored in the reference tree; the
algorithm follows its published definition (clipped surrogate, clipped value loss, entropy bonus, adaptive-KL learning
rate, GAE) and is restated independently for the tests in oracle/policy.py (GAE) -- parity with rsl_rl itself is unpinned.

What is MI355X-specific:
* collection, drift tasks: the whole `num_steps_per_env` rollout (actor on the f32 matrix pipe -> sample -> env.step ->
  storage rows) is ONE launch (`wl_drift_rollout_policy`) plus one `wl_mlp_forward` for the critic;
* collection, any task / observation width: per step one policy launch (`wl_actor_critic_act`) and the env's own launches,
  every output written in place into the storage rows (`env.collect_step`);
* the update, drift agents: the minibatch step (forward, losses, backward, clipping, adaptive-KL rule, Adam) in the HIP
  library (`FusedPpoStep`: `wl_ppo_minibatch`, or `wl_ppo_gradients` -> all-reduce -> `wl_ppo_apply` when data-parallel);
  other widths: torch autograd with the weight gradients of the tall minibatches as chunked batched GEMMs (`_TallLinear`);
* the actor / critic parameters the kernels read ARE the torch Parameters (updated in place, rsl_rl's checkpoint keys).
On a CPU (tests, gloo world-2) everything here runs as plain torch.
"""
from __future__ import annotations

import os
import time
from collections import deque

import torch
from torch import nn

from .. import dist as D
from ..policy import ActorCritic as _KernelActorCritic
from ..policy import Mlp, RolloutStorage

_ACT = {"elu": nn.ELU, "relu": nn.ReLU, "tanh": nn.Tanh}


def _mlp(i, hidden, o, activation):
    layers, d = [], i
    for h in hidden:
        layers += [nn.Linear(d, h), _ACT[activation]()]
        d = h
    return nn.Sequential(*layers, nn.Linear(d, o))


class _TallLinear(torch.autograd.Function):
    """y = x W^T + b for a tall batch (x [B, in], B >> in, out).  The weight gradient dW = dy^T x contracts over the B rows
    into an [out, in] matrix of a few tiles, and the BLAS runs that on a handful of workgroups: 0.41 ms for [64, 689] and
    0.40 ms even for [64, 64] at B = 131072 on MI355X (15 / 2.7 TFLOP/s) -- more than half of a PPO minibatch step on the
    elevation / visual observations.  Here the batch is cut into S chunks whose partial products run as ONE batched GEMM
    ([S, out, in]) followed by a sum: 0.15 ms for [64, 689] (tools/dw_probe.py)."""

    CHUNKS = 64

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        S, B = _TallLinear.CHUNKS, x.shape[0]
        gy = gy.contiguous()
        gw = torch.bmm(gy.view(S, B // S, -1).transpose(1, 2), x.view(S, B // S, -1)).sum(0)
        gx = gy @ w if ctx.needs_input_grad[0] else None
        return gx, gw, gy.sum(0)


def _run_mlp(seq, x, tall: bool):
    """seq(x); for a tall GPU batch under autograd the Linear layers go through _TallLinear"""
    if (tall and x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and x.is_contiguous()
            and x.shape[0] >= 64 * 256 and x.shape[0] % _TallLinear.CHUNKS == 0):
        for m in seq:
            x = _TallLinear.apply(x, m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
        return x
    return seq(x)


class ActorCritic(nn.Module):
    """rsl_rl.modules.ActorCritic work-alike: `actor`, `critic`, `std`, act / evaluate / get_actions_log_prob"""

    tall_linear = True   # weight gradients of tall minibatches as chunked batched GEMMs (see _TallLinear)

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(64, 64), critic_hidden_dims=(64, 64),
                 activation="elu", init_noise_std=1.0, **_unused):
        super().__init__()
        self.activation = activation
        self.actor = _mlp(num_actor_obs, list(actor_hidden_dims), num_actions, activation)
        self.critic = _mlp(num_critic_obs, list(critic_hidden_dims), 1, activation)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        self._fused = None

    # -- torch path (gradient step, tasks without a fused collector) --
    def update_distribution(self, obs):
        # validate_args=False: the argument checks call .all() -> a host sync per minibatch (and are illegal while a HIP
        # graph is being captured)
        mean = _run_mlp(self.actor, obs, self.tall_linear)
        self.distribution = torch.distributions.Normal(mean, self.std.expand(obs.shape[0], -1), validate_args=False)

    def act(self, obs):
        self.update_distribution(obs)
        return self.distribution.sample()

    def act_inference(self, obs):
        return self.actor(obs)

    def evaluate(self, critic_obs):
        return _run_mlp(self.critic, critic_obs, self.tall_linear)

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(-1)

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(-1)

    # -- kernel view: the SAME storage, as the structs wl_drift_rollout_policy / wl_mlp_forward read --
    def fusable(self) -> bool:
        def ok(seq):
            lin = [m for m in seq if isinstance(m, nn.Linear)]
            return len(lin) == 3 and lin[0].out_features == 64 and lin[1].out_features == 64 and lin[0].in_features <= 15
        return self.activation in ("elu", "relu") and ok(self.actor) and ok(self.critic) and self.std.is_cuda

    def act_fusable(self) -> bool:
        """the policy step as ONE launch (wl_actor_critic_act): [64, 64] elu / relu MLPs of any input width on a GPU"""
        def ok(seq, out):
            lin = [m for m in seq if isinstance(m, nn.Linear)]
            return len(lin) == 3 and lin[0].out_features == 64 and lin[1].out_features == 64 and lin[2].out_features == out
        return self.activation in ("elu", "relu") and ok(self.actor, 2) and ok(self.critic, 1) and self.std.is_cuda

    def fused(self):
        if self._fused is None:
            v = _KernelActorCritic.__new__(_KernelActorCritic)   # a kernel-side view: aliases the Parameters, owns nothing
            v.actor = Mlp.from_sequential(self.actor, self.activation, self.std.device)
            v.critic = Mlp.from_sequential(self.critic, self.activation, self.std.device)
            v.std = self.std.detach()
            for m, seq in ((v.actor, self.actor), (v.critic, self.critic)):   # the view must alias, never copy
                lin = [x for x in seq if isinstance(x, nn.Linear)]
                assert all(getattr(m, f"w{i + 1}").data_ptr() == lin[i].weight.data_ptr() for i in range(3))
            self._fused = v
        return self._fused


class PPO:
    """rsl_rl.algorithms.PPO.update on a filled RolloutStorage (fields: rsl_rl_ppo_cfg.py:18-31).

    On a GPU nothing in the update touches the host: the learning rate lives in a device tensor that the (capturable)
    Adam reads, and the adaptive-KL rule -- the one step of rsl_rl's update that needs `kl_mean` on the host -- is a
    `torch.where` on that tensor.  (Replaying the minibatch step as a captured HIP graph was tried on top of this: it
    saved 15 % of the update and its bias gradients drifted from the eager step's after the first update, with either
    BLAS backend, so it is not used.)"""

    def __init__(self, actor_critic: ActorCritic, value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2,
                 entropy_coef=0.005, num_learning_epochs=5, num_mini_batches=4, learning_rate=1e-3, schedule="adaptive",
                 gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0, fused_update: bool | None = None,
                 distributed: bool | None = None, **_unused):
        self.actor_critic = actor_critic
        self.value_loss_coef, self.use_clipped_value_loss, self.clip_param = value_loss_coef, use_clipped_value_loss, clip_param
        self.entropy_coef, self.num_learning_epochs, self.num_mini_batches = entropy_coef, num_learning_epochs, num_mini_batches
        self.schedule, self.gamma, self.lam = schedule, gamma, lam
        self.desired_kl, self.max_grad_norm = desired_kl, max_grad_norm
        dev = next(actor_critic.parameters()).device
        self._lr = torch.tensor(float(learning_rate), dtype=torch.float32, device=dev)
        self._lr_on_device = dev.type == "cuda"
        if self._lr_on_device:   # the optimiser reads the lr tensor: the adaptive rule never needs the host
            self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=self._lr, capturable=True, foreach=True)
        else:
            self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=float(learning_rate))

        # the drift agents' nets on a GPU: the whole minibatch step runs in the HIP library (FusedPpoStep below)
        can_fuse = (dev.type == "cuda" and actor_critic.fusable() and actor_critic.actor[0].in_features == 14
                    and actor_critic.actor[4].out_features == 2)
        # the wide agents (elevation 689, visual 3208 inputs): first layer as bf16-split streaming contractions (FusedWidePpoStep)
        self._wide = (not can_fuse and dev.type == "cuda" and actor_critic.act_fusable()
                      and actor_critic.actor[0].in_features >= 64
                      and actor_critic.actor[0].in_features == actor_critic.critic[0].in_features)
        if fused_update and not (can_fuse or self._wide):
            raise ValueError("fused_update needs D-64-64-2 / D-64-64-1 elu / relu nets (D = 14 or D >= 64) on a GPU")
        self.fused_update = (can_fuse or self._wide) if fused_update is None else bool(fused_update)
        self._fused = None
        # data-parallel learner: ranks hold identical parameters (same seed), step their own env shards, and average the
        # gradient (and the KL statistic of the adaptive rule) once per minibatch step
        self.world = D.world_size() if distributed is None else (D.world_size() if distributed else 1)

    @property
    def learning_rate(self) -> float:
        return self._fused.learning_rate if self._fused is not None else float(self._lr)

    # ---- one minibatch step on the tensors of `b` (all arithmetic; no host round trip) ---------------------------------
    def _step(self, b, sigma_old):
        ac = self.actor_critic
        ac.update_distribution(b["obs"])
        logp = ac.get_actions_log_prob(b["actions"])
        value = ac.evaluate(b["obs"]).squeeze(-1)
        mu, sigma, entropy = ac.action_mean, ac.action_std, ac.entropy
        kl_mean = torch.zeros((), device=mu.device)
        if self.desired_kl is not None and self.schedule == "adaptive":
            with torch.no_grad():
                kl = torch.sum(torch.log(sigma / sigma_old + 1e-5)
                               + (sigma_old.square() + (b["mu"] - mu).square()) / (2.0 * sigma.square()) - 0.5, -1)
                kl_mean = kl.mean()
                if self.world > 1:
                    D.average_(kl_mean)     # every rank takes the same learning-rate decision
                lr = self._lr
                up = (kl_mean > 0.0) & (kl_mean < self.desired_kl / 2.0)
                new_lr = torch.where(kl_mean > self.desired_kl * 2.0, (lr / 1.5).clamp_min(1e-5),
                                     torch.where(up, (lr * 1.5).clamp_max(1e-2), lr))
                self._lr.copy_(new_lr)
                if not self._lr_on_device:
                    for g in self.optimizer.param_groups:
                        g["lr"] = float(new_lr)
        adv = b["adv"]
        ratio = torch.exp(logp - b["logp"])
        surrogate = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
        ret, v_old = b["returns"], b["values"]
        if self.use_clipped_value_loss:
            v_clip = v_old + (value - v_old).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - ret).square(), (v_clip - ret).square()).mean()
        else:
            value_loss = (ret - value).square().mean()
        loss = surrogate + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.world > 1:
            D.average_gradients_(list(ac.parameters()))
        nn.utils.clip_grad_norm_(ac.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return torch.stack([value_loss.detach(), surrogate.detach(), kl_mean.detach()])

    def optimizer_state_dict(self):
        """torch.optim.Adam's state_dict (rsl_rl's checkpoint key), also when the fused step owns the moments"""
        if self._fused is not None:
            self._fused.state_to_optimizer(self.optimizer)
        return self.optimizer.state_dict()

    def load_optimizer_state(self, state_dict):
        """checkpoint resume: load_state_dict replaces the param-group lr, so the device lr tensor is re-bound"""
        self.optimizer.load_state_dict(state_dict)
        with torch.no_grad():
            self._lr.copy_(torch.as_tensor(self.optimizer.param_groups[0]["lr"], dtype=torch.float32))
        for g in self.optimizer.param_groups:
            g["lr"] = self._lr if self._lr_on_device else float(self._lr)
        if self._fused is not None:
            self._fused.state_from_optimizer(self.optimizer, float(self._lr))

    def update(self, storage: RolloutStorage, generator: torch.Generator | None = None):
        K, n = storage.n_steps, storage.n_envs
        returns, advantages = storage.compute_returns(self.gamma, self.lam)
        sigma_old = self.actor_critic.std.detach().clone()
        flat = dict(obs=storage.observations[:K].reshape(K * n, -1), actions=storage.actions.reshape(K * n, -1),
                    values=storage.values[:K].reshape(K * n), returns=returns.reshape(K * n),
                    adv=advantages.reshape(K * n), logp=storage.actions_log_prob.reshape(K * n),
                    mu=storage.mu.reshape(K * n, -1))
        batch = K * n
        mb = batch // self.num_mini_batches
        stats = torch.zeros(3, device=flat["obs"].device)
        fused = self.fused_update and (not self._wide or FusedWidePpoStep.shapes_ok(self.num_mini_batches * mb, mb))
        if fused:
            if self._fused is None:
                self._fused = (FusedWidePpoStep(self.actor_critic, self, self.num_mini_batches * mb, mb) if self._wide
                               else FusedPpoStep(self.actor_critic, self))
                self._fused.state_from_optimizer(self.optimizer, float(self._lr))
            fz = self._fused
            fz.ctrl[4:7] = 0.0
            flat = {k: v.contiguous() for k, v in flat.items()}
            # rsl_rl's mini_batch_generator draws ONE permutation per update and walks it in every epoch
            perm = torch.randperm(self.num_mini_batches * mb, device=flat["obs"].device, generator=generator).to(torch.int32)
            if self._wide:      # the observation block in the order of `perm`, as bf16 planes: once per update
                fz.stage(flat["obs"], perm)
            for _ in range(self.num_learning_epochs):
                for i in range(self.num_mini_batches):    # the kernel gathers through `perm`: no shuffled copies
                    fz.minibatch(flat, perm, i * mb, mb, sigma_old, split=self.world > 1)
            stats = fz.ctrl[4:7].clone()
        else:
            perm = torch.randperm(self.num_mini_batches * mb, device=flat["obs"].device, generator=generator)
            shuffled = {k: v[perm] for k, v in flat.items()}            # one gather per field and update
            for _ in range(self.num_learning_epochs):
                for i in range(self.num_mini_batches):
                    sl = slice(i * mb, (i + 1) * mb)
                    stats += self._step({k: v[sl] for k, v in shuffled.items()}, sigma_old)
        u = self.num_learning_epochs * self.num_mini_batches
        mean_value_loss, mean_surrogate_loss, mean_kl = (stats / u).tolist()
        return dict(value_function=mean_value_loss, surrogate=mean_surrogate_loss, kl=mean_kl, learning_rate=self.learning_rate)


class OnPolicyRunner:
    """rsl_rl OnPolicyRunner / the reference's ModifiedRslRunner work-alike around a RslRlVecEnvWrapper"""

    def __init__(self, env, train_cfg, log_dir: str | None = None, device=None, fused: bool | None = None,
                 kernel_policy: bool | None = None):
        cfg = train_cfg.to_dict() if hasattr(train_cfg, "to_dict") else dict(train_cfg)
        self.cfg, self.env, self.log_dir = cfg, env, log_dir
        self.device = torch.device(device or env.device)
        self.num_steps_per_env = int(cfg.get("num_steps_per_env", 128))
        self.save_interval = int(cfg.get("save_interval", 50))
        pol = cfg.get("policy", {})
        pol = pol.to_dict() if hasattr(pol, "to_dict") else dict(pol)
        alg = cfg.get("algorithm", {})
        alg = alg.to_dict() if hasattr(alg, "to_dict") else dict(alg)
        pol.pop("class_name", None), alg.pop("class_name", None)
        self.actor_critic = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **pol).to(self.device)
        self.alg = PPO(self.actor_critic, **alg)
        base = env.unwrapped
        can_fuse = (getattr(base, "_task", None) == "drift" and self.actor_critic.fusable()
                    and not getattr(base, "_has_custom_rewards", False))
        if fused and not can_fuse:
            raise ValueError("fused collection needs the drift task, [64, 64] elu/relu MLPs and only built-in reward terms")
        self.fused = can_fuse if fused is None else bool(fused)
        # per-step collection (every task, any observation width): the policy step as one launch (wl_actor_critic_act)
        can_act = self.device.type == "cuda" and self.actor_critic.act_fusable() and hasattr(base, "_batch")
        if kernel_policy and not can_act:
            raise ValueError("kernel_policy needs [64, 64] elu / relu MLPs with 2 actions on a GPU")
        self.kernel_policy = can_act if kernel_policy is None else bool(kernel_policy)
        self.storage = RolloutStorage(self.num_steps_per_env, env.num_envs, env.num_obs, env.num_actions, self.device)
        self.current_learning_iteration = 0
        # one process per GPU: every rank runs this loop on its own env shard with identical parameters (same seed, gradient
        # averaged per minibatch step in PPO); rank 0 prints and writes checkpoints, throughput is the whole job's
        self.world = self.alg.world
        self.rank = torch.distributed.get_rank() if self.world > 1 else 0
        self.tot_timesteps, self.tot_time = 0, 0.0
        self.history: list[dict] = []

    # ---- collection --------------------------------------------------------------------------------------
    def _collect_fused(self):
        self.env.unwrapped.rollout_policy(self.actor_critic.fused(), self.storage)

    def _collect_stepwise(self, obs):
        st, ac = self.storage, self.actor_critic
        # the policy step as one launch for any observation width (elevation 689, visual 3208): actor, sampling, log-prob
        # and the critic's value straight into the storage rows
        one_launch = self.kernel_policy
        view = ac.fused() if one_launch else None
        if view is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
            view.global_rows = st.n_envs * torch.distributed.get_world_size()   # kernel form as for the one-process batch
        base = self.env.unwrapped
        batch = getattr(base, "_batch", None)
        if one_launch and hasattr(base, "can_collect_rollout") and base.can_collect_rollout():
            # the whole loop as one launch per curriculum segment, then every row's value in one batched pass (the critic is
            # not needed to step)
            with torch.inference_mode():
                st.observations[0].copy_(obs)
                base.collect_rollout(view, st)
                base.finish_collection(st)
                K, n = st.n_steps, st.n_envs
                view.values_batched(st.observations.view((K + 1) * n, -1), st.values.view((K + 1) * n))
            return st.observations[K]
        if one_launch and hasattr(base, "collect_step") and not getattr(base, "_has_custom_rewards", False):
            # every output straight into the storage rows: per step one policy launch + the env's own launches, no copies
            with torch.inference_mode():
                st.observations[0].copy_(obs)
                for k in range(self.num_steps_per_env):
                    base.collect_step(view, st, k)
                base.finish_collection(st)
                obs = st.observations[self.num_steps_per_env]
                view.values(obs, st.values[self.num_steps_per_env])      # bootstrap value of the last observation
            return obs
        with torch.inference_mode():
            for k in range(self.num_steps_per_env):
                st.observations[k].copy_(obs)
                if one_launch:
                    view.act(st.observations[k], st.actions[k], st.mu[k], st.actions_log_prob[k], st.values[k], batch.seed,
                             batch.step_count, batch.env_offset, planes_fresh=k > 0)
                    a = st.actions[k]
                else:
                    a = ac.act(obs)
                    st.actions[k].copy_(a)
                    st.mu[k].copy_(ac.action_mean)
                    st.actions_log_prob[k].copy_(ac.get_actions_log_prob(a))
                obs, rew, dones, infos = self.env.step(a)
                st.rewards[k].copy_(rew)
                st.dones[k].copy_(dones)
                st.time_outs[k].copy_(infos.get("time_outs", torch.zeros_like(dones, dtype=torch.bool)))
                st.terminated[k].copy_((dones != 0) & ~st.time_outs[k])
            st.observations[self.num_steps_per_env].copy_(obs)
            K, n = st.n_steps, st.n_envs
            if one_launch:
                st.values[K].copy_(ac.evaluate(obs).reshape(n))
            else:
                st.values.copy_(ac.evaluate(st.observations.reshape((K + 1) * n, -1)).reshape(K + 1, n))
        return obs

    def _bookkeeping(self, st, carry_ret, carry_len):
        """returns / lengths of the (last <= 100) episodes that ended inside the rollout in time order, the mean RAW reward per
        step, whether every action is finite; updates the carries of the running episodes in place and applies the time-out
        bootstrap to st.rewards -- _finished_episodes + isfinite + mean + bootstrap_time_outs as one launch"""
        import ctypes as C

        from .. import _abi as A
        K, n = st.n_steps, st.n_envs
        buf = getattr(self, "_book", None)
        if buf is None or buf[0].shape != (K, n):
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
            buf = self._book = (z(K, n), z(K, n), z(3, (n + 255) // 256))
        ep_ret, ep_len, stats = buf
        assert st.time_outs.dtype == torch.bool and st.dones.dtype == torch.int64 and carry_ret.is_contiguous() and carry_len.is_contiguous()
        A.check(A.load().wl_rollout_bookkeeping(K, n, st.rewards.data_ptr(), st.values.data_ptr(), st.dones.data_ptr(),
                                                st.time_outs.data_ptr(), st.actions.data_ptr(), float(self.alg.gamma),
                                                carry_ret.data_ptr(), carry_len.data_ptr(), ep_ret.data_ptr(), ep_len.data_ptr(),
                                                stats.data_ptr(), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "wl_rollout_bookkeeping")
        sel = st.dones.view(-1).nonzero().view(-1)[-100:]
        out = torch.cat([stats.sum(1), ep_ret.view(-1)[sel], ep_len.view(-1)[sel]]).tolist()
        m = sel.numel()
        return out[3:3 + m], out[3 + m:3 + 2 * m], out[0] / (K * n), out[1] == 0.0

    # ---- the learning loop (modified_rsl_rl_runner.py:34-128) ----------------------------------------------
    def learn(self, num_learning_iterations: int, init_at_random_ep_len: bool = False, verbose: bool = True):
        env = self.env
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        obs, _ = env.get_observations()
        n = env.num_envs
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        cur_reward_sum = torch.zeros(n, device=self.device)
        cur_episode_length = torch.zeros(n, device=self.device)
        start_iter = self.current_learning_iteration
        for it in range(start_iter, start_iter + num_learning_iterations):
            t0 = time.time()
            if self.fused:
                self._collect_fused()
            else:
                obs = self._collect_stepwise(obs)
            st = self.storage
            # book keeping of finished episodes (runner :88-98 does it per step with a host sync each): here the whole
            # rollout at once with cumulative sums, one device->host copy of the finished episodes' returns / lengths
            with torch.no_grad():
                if self.device.type == "cuda":
                    # the same bookkeeping as below in one launch + one small device->host copy (wl_rollout_bookkeeping): as
                    # ~25 torch ops it cost 0.41 ms per iteration -- a tenth of the drift task's iteration
                    rets, lens, mean_step_reward, finite = self._bookkeeping(st, cur_reward_sum, cur_episode_length)
                    if not finite:
                        raise ValueError(f"non-finite values in the actions of iteration {it} (diverged policy?)")
                    rewbuffer.extend(rets)
                    lenbuffer.extend(lens)
                else:
                    # the runner's runtime guard (modified_rsl_rl_runner.py:74-75: "NaN in actions"), once per rollout instead
                    # of a host round trip per step
                    if not bool(torch.isfinite(st.actions).all()):
                        raise ValueError(f"non-finite values in the actions of iteration {it} (diverged policy?)")
                    ret, length, cur_reward_sum, cur_episode_length = _finished_episodes(st.rewards, st.dones != 0, cur_reward_sum,
                                                                                         cur_episode_length)
                    rewbuffer.extend(ret[-100:].tolist())
                    lenbuffer.extend(length[-100:].tolist())
                    # the learning signal that is logged is the RAW env reward: bootstrap_time_outs adds gamma * V(obs) at
                    # time-out steps, which grows as the critic learns even when the policy does not
                    mean_step_reward = float(st.rewards.mean())
                    st.bootstrap_time_outs(self.alg.gamma)
            torch.cuda.synchronize() if self.device.type == "cuda" else None
            t1 = time.time()
            losses = self.alg.update(st)
            torch.cuda.synchronize() if self.device.type == "cuda" else None
            t2 = time.time()
            self.current_learning_iteration = it + 1
            steps = st.n_steps * n * self.world
            self.tot_timesteps += steps
            self.tot_time += t2 - t0
            log = dict(iteration=it, collection_time=t1 - t0, learn_time=t2 - t1, fps=steps / (t2 - t0),
                       collection_fps=steps / max(t1 - t0, 1e-9), mean_reward=_mean(rewbuffer), mean_episode_length=_mean(lenbuffer),
                       mean_step_reward=mean_step_reward, mean_noise_std=float(self.actor_critic.std.detach().mean()), **losses)
            base = env.unwrapped
            if hasattr(base, "episode_log_summary"):
                log.update(base.episode_log_summary(st.n_steps, reduce_ranks=self.world > 1))
            self.history.append(log)
            if verbose and self.rank == 0:
                print(f"[it {it:4d}] fps {log['fps']:.3e} (collect {log['collection_fps']:.3e})  mean_reward {log['mean_reward']:.2f}"
                      f"  ep_len {log['mean_episode_length']:.1f}  step_rew {log['mean_step_reward']:.3f}"
                      f"  std {log['mean_noise_std']:.3f}  kl {losses['kl']:.4f}  lr {losses['learning_rate']:.2e}", flush=True)
            if self.log_dir and self.rank == 0 and (it % self.save_interval == 0 or it == start_iter + num_learning_iterations - 1):
                self.save(os.path.join(self.log_dir, "models", f"model_{it}.pt"))
        return self.history

    # ---- checkpoints: rsl_rl's keys (train_rl.py:96-106 resumes from `model_*.pt`) --------------------------
    def save(self, path: str, infos=None):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save({"model_state_dict": self.actor_critic.state_dict(), "optimizer_state_dict": self.alg.optimizer_state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path: str, load_optimizer: bool = True):
        d = torch.load(path, map_location=self.device, weights_only=False)
        self.actor_critic.load_state_dict(d["model_state_dict"])    # copies in place: the kernel view stays valid
        if load_optimizer:
            self.alg.load_optimizer_state(d["optimizer_state_dict"])
        self.current_learning_iteration = d["iter"]
        return d.get("infos")

    def get_inference_policy(self, device=None):
        self.actor_critic.eval()
        if device is not None:
            self.actor_critic.to(device)
        return self.actor_critic.act_inference


def _finished_episodes(rewards, done, carry_ret, carry_len):
    """returns / lengths of the episodes that end inside a [K, n] rollout, in time order, plus the updated carries of
    the episodes still running -- the vectorised form of the runner's per-step cur_reward_sum bookkeeping"""
    K, n = rewards.shape
    dev = rewards.device
    C = rewards.cumsum(0)
    t = torch.arange(K, device=dev)[:, None].expand(K, n)
    last_done = torch.cummax(torch.where(done, t, torch.full_like(t, -1)), 0).values        # last done index <= k
    prev = torch.cat([torch.full((1, n), -1, device=dev, dtype=t.dtype), last_done[:-1]])   # last done index <  k
    started_here = prev >= 0
    C_prev = torch.where(started_here, C.gather(0, prev.clamp(min=0)), torch.zeros_like(C))
    ep_ret = C - C_prev + torch.where(started_here, torch.zeros_like(C), carry_ret[None].expand(K, n))
    ep_len = (t - prev).to(rewards.dtype) + torch.where(started_here, torch.zeros_like(C), carry_len[None].expand(K, n))
    last = last_done[-1]
    any_done = last >= 0
    C_last = C.gather(0, last.clamp(min=0)[None])[0]
    new_ret = torch.where(any_done, C[-1] - C_last, carry_ret + C[-1])
    new_len = torch.where(any_done, (K - 1 - last).to(rewards.dtype), carry_len + K)
    return ep_ret[done], ep_len[done], new_ret, new_len


def _mean(buf):
    return float(sum(buf) / len(buf)) if len(buf) else 0.0


class FusedPpoStep:
    """The PPO minibatch step as three HIP launches (csrc/wl_ppo.hip: fused MFMA forward / backward / weight gradients,
    partial-sum reduction, clip + adaptive-KL learning rate + Adam) on the torch Parameters of `actor_critic` in place.
    Owns the device scratch the C ABI asks the caller for."""

    def __init__(self, actor_critic: ActorCritic, ppo: "PPO"):
        import ctypes as C

        from .. import _abi as A
        if not actor_critic.fusable() or actor_critic.actor[0].in_features != 14 or actor_critic.actor[4].out_features != 2:
            raise ValueError("the fused PPO step is specialised for the drift agents' 14-64-64-2 / 14-64-64-1 MLPs")
        self._C, self._A, self.lib = C, A, A.load()
        self.ac, self.view = actor_critic, actor_critic.fused()
        dev = actor_critic.std.device
        self.dev = dev
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.partials, self.grad = z(A.PPO_BLOCKS, A.PPO_PARTIAL_STRIDE), z(A.PPO_PARTIAL_STRIDE)
        self.adam_m, self.adam_v, self.ctrl = z(A.PPO_NUM_PARAMS), z(A.PPO_NUM_PARAMS), z(16)
        self.operands = z(A.PPO_OPERAND_FLOATS)
        self.ctrl[A.PPO_CTRL_LR:A.PPO_CTRL_LR + 2] = float(ppo.learning_rate)
        self.state = A.WlPpoState(self.partials.data_ptr(), self.grad.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                                  self.ctrl.data_ptr(), self.operands.data_ptr())
        adaptive = int(ppo.desired_kl is not None and ppo.schedule == "adaptive")
        self.hp = A.WlPpoParams(ppo.clip_param, ppo.value_loss_coef, ppo.entropy_coef, float(ppo.desired_kl or 0.0),
                                ppo.max_grad_norm, 0.9, 0.999, 1e-8, 1e-5, 1e-2, int(ppo.use_clipped_value_loss), adaptive)
        self.parity, self.adam_step = 0, 0
        self._actor, self._critic = self.view.actor.struct(), self.view.critic.struct()

    def _batch(self, flat, perm, sigma_old):
        A = self._A
        for k in ("obs", "actions", "mu", "logp", "adv", "returns", "values"):
            t = flat[k]
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.dev, k
        assert perm.dtype == torch.int32 and perm.is_contiguous()
        self._keep = (flat, perm, sigma_old)     # the launch is asynchronous: keep the tensors alive
        return A.WlPpoBatch(flat["obs"].data_ptr(), flat["actions"].data_ptr(), flat["mu"].data_ptr(), flat["logp"].data_ptr(),
                            flat["adv"].data_ptr(), flat["returns"].data_ptr(), flat["values"].data_ptr(), perm.data_ptr(),
                            sigma_old.data_ptr())

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    @property
    def learning_rate(self) -> float:
        return float(self.ctrl[self._A.PPO_CTRL_LR + self.parity])

    def gradients(self, flat, perm, mb_start, mb_size, sigma_old):
        """d loss / d parameters of one minibatch in the flat order of named_parameters() (no entropy term, no clipping)
        plus [value-loss sum, surrogate sum, KL sum] -- the parity entry point"""
        C, A = self._C, self._A
        bt = self._batch(flat, perm, sigma_old)
        self.ctrl[A.PPO_CTRL_NORM2:A.PPO_CTRL_NORM2 + 2] = 0.0
        A.check(self.lib.wl_ppo_gradients(C.byref(self._actor), C.byref(self._critic), self.ac.std.data_ptr(), C.byref(bt),
                                          int(mb_start), int(mb_size), C.byref(self.hp), C.byref(self.state), self.parity,
                                          self._stream()), "wl_ppo_gradients")
        return self.grad

    def minibatch(self, flat, perm, mb_start, mb_size, sigma_old, split: bool = False):
        """one PPO step in place.  split=True is the data-parallel form: gradients of this rank's minibatch, ONE all-reduce
        of the gradient row (parameter gradients + the three statistics), then the update on the averaged row"""
        C, A = self._C, self._A
        if split:
            self.ctrl[A.PPO_CTRL_NORM2 + self.parity] = 0.0
            bt = self._batch(flat, perm, sigma_old)
            A.check(self.lib.wl_ppo_gradients(C.byref(self._actor), C.byref(self._critic), self.ac.std.data_ptr(), C.byref(bt),
                                              int(mb_start), int(mb_size), C.byref(self.hp), C.byref(self.state), self.parity,
                                              self._stream()), "wl_ppo_gradients")
            D.average_(self.grad)
            self.ctrl[A.PPO_CTRL_NORM2 + self.parity] = self.grad[:A.PPO_NUM_PARAMS].square().sum()
            self.adam_step += 1
            A.check(self.lib.wl_ppo_apply(C.byref(self._actor), C.byref(self._critic), self.ac.std.data_ptr(), int(mb_size),
                                          C.byref(self.hp), C.byref(self.state), self.parity, self.adam_step, self._stream()),
                    "wl_ppo_apply")
            self.parity ^= 1
            return
        bt = self._batch(flat, perm, sigma_old)
        self.adam_step += 1
        A.check(self.lib.wl_ppo_minibatch(C.byref(self._actor), C.byref(self._critic), self.ac.std.data_ptr(), C.byref(bt),
                                          int(mb_start), int(mb_size), C.byref(self.hp), C.byref(self.state), self.parity,
                                          self.adam_step, self._stream()), "wl_ppo_minibatch")
        self.parity ^= 1

    # ---- Adam state <-> torch.optim.Adam (checkpoints keep rsl_rl's format) ---------------------------------------------
    def state_to_optimizer(self, optimizer):
        off = 0
        for p in self.ac.parameters():
            k = p.numel()
            st = optimizer.state[p]
            st["step"] = torch.tensor(float(self.adam_step), dtype=torch.float32, device=self.dev)
            st["exp_avg"] = self.adam_m[off:off + k].view_as(p).clone()
            st["exp_avg_sq"] = self.adam_v[off:off + k].view_as(p).clone()
            off += k
        lr = self.learning_rate
        for g in optimizer.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr

    def state_from_optimizer(self, optimizer, lr: float):
        off, step = 0, 0
        for p in self.ac.parameters():
            k = p.numel()
            st = optimizer.state.get(p, {})
            if "exp_avg" in st:
                self.adam_m[off:off + k] = st["exp_avg"].reshape(-1)
                self.adam_v[off:off + k] = st["exp_avg_sq"].reshape(-1)
                step = int(float(st["step"]))
            off += k
        self.adam_step = step
        self.ctrl[self._A.PPO_CTRL_LR:self._A.PPO_CTRL_LR + 2] = float(lr)


class FusedWidePpoStep(FusedPpoStep):
    """The same step for the wide agents (D = 689 elevation / 3208 visual): csrc/wl_ppo_wide.hip.  The observation block is
    staged once per update as bf16 planes in the order of the update's permutation (`stage`), the first layer of both nets
    and its weight gradient are streaming contractions on the bf16 matrix pipe (operands split hi + lo: 16 mantissa bits,
    f32 accumulation), everything behind it is the drift agents' kernel."""

    @staticmethod
    def shapes_ok(rows: int, mb: int) -> bool:
        return rows % 64 == 0 and mb % 64 == 0

    @staticmethod
    def pick_splits(dp: int, mb: int) -> int:
        """split-K factor of the dW1 contraction: the smallest power of two that puts >= 768 blocks (1.5 rounds of the chip's
        512 slots) on the launch -- measured on the elevation agent (6 row blocks x 2048 K chunks): 64 splits 122 us, 85
        (unequal shares) 139 us, 128 107 us, 256 112 us; the visual agent (26 x 512) is flat at 122-128 us from 8 to 64"""
        row_blocks, chunks = (dp + 127) // 128, max(1, mb // 64)
        want = 768 if row_blocks <= 8 else 384      # wide operands: fewer splits, the partial sums' reduction is 15 us per 16
        s = 1
        while row_blocks * s < want and s < chunks:
            s *= 2
        return min(s, chunks)

    def __init__(self, actor_critic: ActorCritic, ppo: "PPO", capacity: int, mb_capacity: int):
        import ctypes as C

        from .. import _abi as A
        D_in = actor_critic.actor[0].in_features
        if not actor_critic.act_fusable() or D_in < 64 or actor_critic.critic[0].in_features != D_in:
            raise ValueError("the wide fused PPO step needs D-64-64-2 / D-64-64-1 elu / relu MLPs with D >= 64")
        if not self.shapes_ok(capacity, mb_capacity):
            raise ValueError("rows per update and the minibatch size must be multiples of 64")
        self._C, self._A, self.lib = C, A, A.load()
        self.ac, self.view = actor_critic, actor_critic.fused()
        dev = actor_critic.std.device
        self.dev = dev
        self.in_dim, self.dp = D_in, (D_in + 63) // 64 * 64
        self.capacity, self.mb_capacity = int(capacity), int(mb_capacity)
        self.splits = int(os.environ.get("WL_WIDE_SPLITS", 0)) or self.pick_splits(self.dp, self.mb_capacity)
        self.G = int(self.lib.wl_ppo_wide_num_params(D_in))
        assert self.G == sum(p.numel() for p in actor_critic.parameters())
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        h = lambda *s: torch.zeros(*s, dtype=torch.int16, device=dev)
        dp = self.dp
        self.xt_hi, self.xt_lo = h(capacity // 64, dp, 64), h(capacity // 64, dp, 64)      # X^T blocked by 64 rows
        self.w_hi, self.w_lo = h(128, dp), h(128, dp)
        self.h1, self.dt_hi, self.dt_lo = z(mb_capacity, 128), h(mb_capacity // 64, 128, 64), h(mb_capacity // 64, 128, 64)
        self.dw_partials = z(self.splits, dp, 128)
        self.partials, self.narrow = z(A.PPO_BLOCKS, A.PPO_PARTIAL_STRIDE), z(A.PPO_PARTIAL_STRIDE)
        self.grad = z(self.G + 3)
        self.adam_m, self.adam_v, self.ctrl = z(self.G), z(self.G), z(16)
        self.operands = z(A.PPO_OPERAND_FLOATS)
        self.ctrl[A.PPO_CTRL_LR:A.PPO_CTRL_LR + 2] = float(ppo.learning_rate)
        ptr = lambda t: t.data_ptr()
        self.state = A.WlPpoWideState(ptr(self.xt_hi), ptr(self.xt_lo), ptr(self.w_hi), ptr(self.w_lo),
                                      ptr(self.h1), ptr(self.dt_hi), ptr(self.dt_lo), ptr(self.dw_partials), ptr(self.partials),
                                      ptr(self.narrow), ptr(self.grad), ptr(self.adam_m), ptr(self.adam_v), ptr(self.ctrl),
                                      ptr(self.operands), self.in_dim, self.dp, self.capacity, self.mb_capacity, self.splits)
        adaptive = int(ppo.desired_kl is not None and ppo.schedule == "adaptive")
        self.hp = A.WlPpoParams(ppo.clip_param, ppo.value_loss_coef, ppo.entropy_coef, float(ppo.desired_kl or 0.0),
                                ppo.max_grad_norm, 0.9, 0.999, 1e-8, 1e-5, 1e-2, int(ppo.use_clipped_value_loss), adaptive)
        self.parity, self.adam_step = 0, 0
        self._actor, self._critic = self.view.actor.struct(), self.view.critic.struct()

    def stage(self, obs, perm):
        """rows perm[k] of `obs` ([B, D] f32) -> staged row k (bf16 planes X and X^T)"""
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.in_dim and obs.device == self.dev
        assert perm.dtype == torch.int32 and perm.is_contiguous() and perm.numel() <= self.capacity
        self._keep_stage = (obs, perm)
        self._A.check(self.lib.wl_ppo_wide_stage(obs.data_ptr(), perm.data_ptr(), int(perm.numel()), self._C.byref(self.state),
                                                 self._stream()), "wl_ppo_wide_stage")

    def gradients(self, flat, perm, mb_start, mb_size, sigma_old):
        C, A = self._C, self._A
        bt = self._batch(flat, perm, sigma_old)
        self.ctrl[A.PPO_CTRL_NORM2:A.PPO_CTRL_NORM2 + 2] = 0.0
        A.check(self.lib.wl_ppo_wide_gradients(C.byref(self._actor), C.byref(self._critic), self.ac.std.data_ptr(), C.byref(bt),
                                               int(mb_start), int(mb_size), C.byref(self.hp), C.byref(self.state), self.parity,
                                               self._stream()), "wl_ppo_wide_gradients")
        return self.grad

    def minibatch(self, flat, perm, mb_start, mb_size, sigma_old, split: bool = False):
        C, A = self._C, self._A
        bt = self._batch(flat, perm, sigma_old)
        self.adam_step += 1
        if split:
            self.ctrl[A.PPO_CTRL_NORM2 + self.parity] = 0.0
            A.check(self.lib.wl_ppo_wide_gradients(C.byref(self._actor), C.byref(self._critic), self.ac.std.data_ptr(), C.byref(bt),
                                                   int(mb_start), int(mb_size), C.byref(self.hp), C.byref(self.state), self.parity,
                                                   self._stream()), "wl_ppo_wide_gradients")
            D.average_(self.grad)
            self.ctrl[A.PPO_CTRL_NORM2 + self.parity] = self.grad[:self.G].square().sum()
            A.check(self.lib.wl_ppo_wide_apply(C.byref(self._actor), C.byref(self._critic), self.ac.std.data_ptr(), int(mb_size),
                                               C.byref(self.hp), C.byref(self.state), self.parity, self.adam_step, self._stream()),
                    "wl_ppo_wide_apply")
        else:
            A.check(self.lib.wl_ppo_wide_minibatch(C.byref(self._actor), C.byref(self._critic), self.ac.std.data_ptr(), C.byref(bt),
                                                   int(mb_start), int(mb_size), C.byref(self.hp), C.byref(self.state), self.parity,
                                                   self.adam_step, self._stream()), "wl_ppo_wide_minibatch")
        self.parity ^= 1

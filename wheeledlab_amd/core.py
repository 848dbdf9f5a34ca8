"""DriftBatch: device buffers of one shard of drift envs + thin calls into the C ABI.

PyTorch is plumbing here: it owns the HBM allocations and the stream; every kernel is ours (csrc/wl_drift.hip).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _abi as A
from .params import drift_params


def stadium_reference_poses(u: torch.Tensor, track_radius: float = 0.8, straight: float = 0.8) -> torch.Tensor:
    """Pre-sampled reset poses on the stadium centre line: arclength u*L -> (x, y, yaw rad), [3, P].
    Follows reset_root_state_along_track.generate_reference_poses (wheeledlab_tasks/drifting/mdp/events.py:33-100):
    four pieces -- right straight (yaw 90 deg), top arc, left straight (yaw 270 deg), bottom arc."""
    r, s = track_radius, straight
    d = u.double() * (2.0 * math.pi * r + 4.0 * s)
    x, y, yaw = torch.empty_like(d), torch.empty_like(d), torch.empty_like(d)
    b1, b2, b3 = 2 * s, 2 * s + math.pi * r, 4 * s + math.pi * r
    m = d < b1
    x[m], y[m], yaw[m] = r, d[m] - s, math.pi / 2
    m = (d >= b1) & (d < b2)
    a = (d[m] - b1) / r
    x[m], y[m], yaw[m] = r * torch.cos(a), s + r * torch.sin(a), math.pi / 2 + a
    m = (d >= b2) & (d < b3)
    x[m], y[m], yaw[m] = -r, s - (d[m] - b2), 1.5 * math.pi
    m = d >= b3
    a = (d[m] - b3) / r
    x[m], y[m], yaw[m] = -r * torch.cos(a), -s - r * torch.sin(a), 1.5 * math.pi + a
    return torch.stack([x, y, yaw]).float()


def apply_startup_events(lib, bufs: "A.WlEnvBuffers", su, seed: int, stream, randomize: bool = True):
    """startup-mode events (domain randomisation, applied once): bucketed wheel friction
    (isaaclab randomize_rigid_body_material), throttle damping (randomize_actuator_gains, "abs"), base mass
    (randomize_rigid_body_mass, "add" onto the chassis mass).  Reference configs: mushr_drift_env_cfg.py:98-119,145-154;
    elevation cfg :387-407; visual cfg :264-299.  One launch of wl_startup_randomize: the draws are keyed by the GLOBAL
    env id (bufs.env_offset + e), so a sharded run holds the same parameter sets as the one big batch."""
    sp = A.WlStartupParams((C.c_float * 2)(*su.wheel_mu_s), (C.c_float * 2)(*su.wheel_mu_d), int(su.mu_buckets),
                           int(bool(su.mu_consistent)), (C.c_float * 2)(*su.damping), float(su.chassis_mass),
                           (C.c_float * 2)(*su.mass_add), int(bool(randomize)), (C.c_float * 2)(*getattr(su, "wheel_mass", (0.0, 0.0))))
    A.check(lib.wl_startup_randomize(C.byref(sp), C.byref(bufs), int(seed), stream), "wl_startup_randomize")


def _canonical_device(device) -> torch.device:
    """torch.device with its index filled in: 'cuda' and 'cuda:0' name the same GPU but compare unequal"""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return d


def pair_table(codes: torch.Tensor) -> torch.Tensor:
    """WlHeightField.pair as the header defines it: pair[j][i] = code[j][i] (low half) | code[min(j + 1, ny - 1)][i] << 16, int32
    [ny, nx] (what wl_heightfield_pairs builds on the device; here in torch, for host-side fields and as the test's definition)"""
    c = codes.to(torch.int32)
    up = torch.cat([c[1:], c[-1:]], 0)
    return ((c & 0xffff) | (up << 16)).to(torch.int32).contiguous()


class DeviceHeightField:
    """A heightfield resident on the device as the kernels read it (WlHeightField, ABI 21): 16-bit height codes [ny, nx] and the
    vertical scale, z = code * z_scale.  `heightfield` is `(height, x0, y0, cell)` with float heights (quantised: terrain.
    quantize_heights' rule, z_scale 2^-13 m unless the range needs more) or `(codes int16, x0, y0, cell, z_scale)`, arrays or
    tensors; or another DeviceHeightField on the same device (shared).  `.heights`: the decoded fp32 grid -- exactly the values
    every kernel sees (what tests hand to the oracle)."""

    def __init__(self, heightfield, device, outside_z: float = 0.0):
        from .terrain import default_z_scale
        self.device = _canonical_device(device)
        if isinstance(heightfield, DeviceHeightField):
            src = heightfield
            if src.device != self.device:
                raise ValueError(f"a DeviceHeightField lives on {src.device}; it cannot be shared with {self.device}")
            self.codes, self.z_scale, self.heights, self.pairs = src.codes, src.z_scale, src.heights, src.pairs
            self.x0, self.y0, self.cell = src.x0, src.y0, src.cell
        else:
            h, x0, y0, cell, *rest = heightfield
            h = torch.as_tensor(h)
            if h.dtype == torch.int16:
                if not rest:
                    raise ValueError("int16 height codes need their z_scale: (codes, x0, y0, cell, z_scale)")
                self.codes, self.z_scale = h.contiguous().to(self.device), float(rest[0])
            else:
                h = h.to(self.device, torch.float64)
                if not bool(torch.isfinite(h).all()):
                    raise ValueError("heightfield with non-finite heights")
                hmax = float(h.abs().max()) if h.numel() else 0.0
                self.z_scale = float(rest[0]) if rest else default_z_scale(hmax)
                if rest and math.isfinite(self.z_scale) and self.z_scale > 0 and hmax > 32767 * self.z_scale:
                    # (the default scale widens itself; an explicit one that cannot hold the heights would flatten them silently)
                    raise ValueError(f"heights up to {hmax:g} m do not fit 16-bit codes of z_scale {self.z_scale:g} m "
                                     f"(+-{32767 * self.z_scale:g} m): pass a larger z_scale or none")
                self.codes = torch.clamp(torch.round(h / self.z_scale), -32767, 32767).to(torch.int16).contiguous() if (
                    math.isfinite(self.z_scale) and self.z_scale > 0) else torch.zeros((0,), dtype=torch.int16)
            if not (math.isfinite(self.z_scale) and self.z_scale > 0) or self.codes.dim() != 2:
                raise ValueError("heightfield: a [ny, nx] grid and a positive, finite z_scale")
            self.heights = self.codes.to(torch.float32) * torch.tensor(self.z_scale, dtype=torch.float32, device=self.device)
            self.x0, self.y0, self.cell = float(x0), float(y0), float(cell)
            self.pairs = None
        self.outside_z = float(outside_z)
        ny, nx = self.codes.shape
        self.struct = A.WlHeightField(self.codes.data_ptr(), nx, ny, self.x0, self.y0, self.cell, self.outside_z, self.z_scale, None)
        if self.pairs is None:
            # the row-pair table the height scan gathers from (WlHeightField.pair, ABI 23): built on the device by the library
            if self.device.type == "cuda":
                self.pairs = torch.empty((ny, nx), dtype=torch.int32, device=self.device)
                A.check(A.load().wl_heightfield_pairs(C.byref(self.struct), self.pairs.data_ptr(),
                                                      C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "wl_heightfield_pairs")
            else:
                self.pairs = pair_table(self.codes)
        self.struct.pair = self.pairs.data_ptr()

    def as_tuple(self):
        """(decoded heights, x0, y0, cell): the form the oracle's functions take"""
        return self.heights, self.x0, self.y0, self.cell


class _MetricsView:
    """`metrics_raw` is what the kernels add into: [slots][WL_M_SHARDS][WL_M_COUNT].  `metrics` is the logical value
    (sum over the shards): [WL_M_COUNT] for one accumulator, [slots][WL_M_COUNT] for a ring; a fresh tensor per read."""

    @property
    def metrics(self) -> torch.Tensor:
        m = self.metrics_raw.sum(1)
        return m[0] if self.metrics_slots == 1 else m

    pose_epoch = 0       # bumped by everything that moves cars WITHOUT advancing step_count (resets, plugin pose writes)

    def touch_pose(self):
        """anything cached per env.step() from the poses (the scene camera's render) is stale after this"""
        self.pose_epoch = self.pose_epoch + 1

    def set_flags(self, flags: int = 0):
        """WlEnvBuffers.flags (_abi.FLAG_*): force an instantiation the launchers otherwise pick from the batch size -- the
        streaming (non-temporal store) forms, the cache-allocating forms, the height scan through LDS patches or through
        gathers.  0 = by size.  What the tests use to run the large-batch forms at small sizes (and vice versa)."""
        self._bufs.flags = int(flags)

    def _ring_aliases(self, n_steps: int) -> bool:
        """a persistent launch folds its steps into ring slot step0 % R and clears slot (step0 + n) % R for its successor: with n
        a multiple of R those are the same slot and the C ABI refuses the launch (WL_EINVAL).  The host layer then runs the
        rollout as two launches (1 and n - 1 steps) -- the same rollout; like n single steps it leaves the ring without the
        first step's counts (a ring of R slots holds R - 1 steps)."""
        return self.metrics_slots > 1 and n_steps > 1 and n_steps % self.metrics_slots == 0

    def _warn_ring_alias(self, n_steps: int):
        if not getattr(self, "_ring_alias_warned", False):
            import warnings
            warnings.warn(f"a persistent rollout of {n_steps} steps with a metric ring of {self.metrics_slots} slots is run as 1 + "
                          f"{n_steps - 1} steps: the ring then misses the first step's episode counts (choose a rollout length that is "
                          "not a multiple of metrics_slots to keep them)", stacklevel=3)
            self._ring_alias_warned = True


class DriftBatch(_MetricsView):
    """n drift envs resident on one GPU as a SoA state matrix [S_COUNT, stride] (fp32)."""

    OBS_DIM = 14

    def __init__(self, n_envs: int, device="cuda:0", params: A.WlDriftParams | None = None, seed: int = 42,
                 env_offset: int = 0, randomize: bool = True, metrics_slots: int = 1, startup=None):
        self.lib = A.load()  # raises HipExtensionMissing -- no fallback
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise A.HipExtensionMissing("DriftBatch needs a HIP device (device='cuda:N'); there is no CPU path")
        self.n = int(n_envs)
        self.stride = ((self.n + 63) // 64) * 64
        self.p = params if params is not None else drift_params()
        self.seed = int(seed)
        self.env_offset = int(env_offset)
        self.step_count = 0
        dev = self.device
        self.state = torch.zeros(A.S_COUNT, self.stride, dtype=torch.float32, device=dev)
        self.episode_len = torch.zeros(self.stride, dtype=torch.int32, device=dev)
        self.metrics_slots = int(metrics_slots)
        self.metrics_raw = torch.zeros(self.metrics_slots, A.M_SHARDS, A.M_COUNT, dtype=torch.float32, device=dev)
        self.obs = torch.zeros(self.n, self.OBS_DIM, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(self.n, dtype=torch.float32, device=dev)
        # torch.bool is one byte holding 0 / 1: the kernel's uint8 outputs land in it directly
        self.terminated = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.truncated = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.dones = torch.zeros(self.n, dtype=torch.long, device=dev)   # terminated | truncated, as RSL-RL consumes it
        g = torch.Generator().manual_seed(self.seed)
        # reference poses are drawn ONCE at construction (events.py:31,35)
        self.ref_table = torch.zeros(3, 32, dtype=torch.float32)
        tr = (startup.track_radius, startup.track_straight) if startup is not None else (0.8, 0.8)
        self.ref_table[:, : self.p.num_ref_points] = stadium_reference_poses(torch.rand(self.p.num_ref_points, generator=g), *tr)
        self.ref_table = self.ref_table.to(dev)
        self._bufs = A.WlEnvBuffers(self.state.data_ptr(), self.episode_len.data_ptr(), self.ref_table.data_ptr(),
                                    self.metrics_raw.data_ptr(), self.stride, self.n, self.env_offset, self.metrics_slots, 0, 0)
        self._startup_events(randomize, startup)
        self._out = A.WlStepOut(self.obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(),
                                self.truncated.data_ptr(), self.dones.data_ptr())

    # startup events (mushr_drift_env_cfg.py:98-119, 145-154)
    def _startup_events(self, randomize: bool, su=None):
        if su is None:  # the RSS drift defaults
            from .envs.flatten import StartupSpec
            su = StartupSpec(wheel_mu_s=(0.3, 0.5), wheel_mu_d=(0.3, 0.5), mu_buckets=20, mu_consistent=True,
                             damping=(10.0, 50.0), mass_add=(0.3, 0.5))
        apply_startup_events(self.lib, self._bufs, su, self.seed, self._stream(), randomize)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_lanes(self, lanes: int = 0):
        """step-kernel form: 0 = by env count (quad up to 32 768 envs, lane form with packed axles up to 262 144, with the
        scalar wheel loop beyond), 1 = lane per env / packed axles, 2 = lane per env / scalar wheel loop, 4 = quad per env"""
        assert lanes in (0, 1, 2, 4)   # 2: lane form with the scalar wheel loop (drift only; treated as 1 elsewhere)
        self._bufs.lanes = lanes

    def set_dones_output(self, on: bool = True):
        """the int64 `dones` row (terminated | truncated as RSL-RL's runner consumes it, + 8 B per env-step) of step() /
        in-place rollouts on or off; callers that read the two byte rows do not need it"""
        self._out.dones = self.dones.data_ptr() if on else None

    def reset(self, mask: torch.Tensor | None = None):
        self.touch_pose()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        A.check(self.lib.wl_drift_reset(C.byref(self.p), C.byref(self._bufs), None if m is None else m.data_ptr(),
                                        self.seed, self.step_count, self._stream()), "wl_drift_reset")

    def observe(self, noise: torch.Tensor | None = None) -> torch.Tensor:
        A.check(self.lib.wl_drift_observe(C.byref(self.p), C.byref(self._bufs),
                                          None if noise is None else noise.data_ptr(), self.obs.data_ptr(), self.seed,
                                          self.step_count, self._stream()), "wl_drift_observe")
        return self.obs

    def step(self, actions: torch.Tensor, noise: torch.Tensor | None = None):
        """actions [n,2] fp32 on device -> (obs [n,14], reward [n], terminated u8 [n], truncated u8 [n]) (views)"""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.shape != (self.n, 2):
            actions = actions.to(torch.float32).reshape(self.n, 2).contiguous()
        A.check(self.lib.wl_drift_step(C.byref(self.p), C.byref(self._bufs), actions.data_ptr(),
                                       None if noise is None else noise.data_ptr(), C.byref(self._out), self.seed,
                                       self.step_count, self._stream()), "wl_drift_step")
        self.step_count += 1
        return self.obs, self.reward, self.terminated, self.truncated

    def rollout(self, actions: torch.Tensor, obs_out: torch.Tensor | None = None, rew_out: torch.Tensor | None = None,
                term_out: torch.Tensor | None = None, trunc_out: torch.Tensor | None = None, persistent: bool = False,
                dones_out: torch.Tensor | None = None):
        """K fused steps with pre-staged actions [K,n,2]; optional [K,...] output storage (else overwrite).
        persistent=True runs them as ONE launch with the state held in registers (wl_drift_rollout_persistent)."""
        K = actions.shape[0]
        assert actions.shape == (K, self.n, 2) and actions.dtype == torch.float32 and actions.is_contiguous()
        if persistent and self._ring_aliases(K):
            self._warn_ring_alias(K)
            cut = lambda t, a, b: None if t is None else t[a:b]
            for a, b in ((0, 1), (1, K)):
                self.rollout(actions[a:b], cut(obs_out, a, b), cut(rew_out, a, b), cut(term_out, a, b), cut(trunc_out, a, b), True,
                             cut(dones_out, a, b))
            return
        if obs_out is not None:
            key = (obs_out.data_ptr(), rew_out.data_ptr(), term_out.data_ptr(), trunc_out.data_ptr(),
                   None if dones_out is None else dones_out.data_ptr())
            if getattr(self, "_out_key", None) != key:      # the struct of the caller's storage rows, rebuilt when they change
                self._out_key, self._out_rows = key, A.WlStepOut(*key)
            out, os_, vs_ = self._out_rows, self.n * self.OBS_DIM, self.n
        else:
            out, os_, vs_ = self._out, 0, 0
        fn = self.lib.wl_drift_rollout_persistent if persistent else self.lib.wl_drift_rollout
        A.check(fn(C.byref(self.p), C.byref(self._bufs), actions.data_ptr(), C.byref(out), os_, vs_, K, self.seed,
                   self.step_count, self._stream()), "wl_drift_rollout")
        self.step_count += K

    def rollout_policy(self, actor_critic, storage, evaluate_critic: bool = True, start: int = 0, count: int | None = None):
        """The runner's collection loop (modified_rsl_rl_runner.py:70-80) as one launch: `count` times
        { actor(obs) on the matrix pipe -> sample -> env.step } with the env state and the observation in registers
        (wl_drift_rollout_policy), filling storage rows start .. start + count; then (optionally) the critic over ALL
        observation rows of the storage in one wl_mlp_forward.  Row `start` of storage.observations is seeded with the
        current observation; self.obs ends as the last one."""
        K = storage.n_steps - start if count is None else int(count)
        assert storage.n_envs == self.n and actor_critic.actor.in_dim == self.OBS_DIM and 0 <= start and start + K <= storage.n_steps
        if self._ring_aliases(K):
            self.rollout_policy(actor_critic, storage, False, start, 1)
            return self.rollout_policy(actor_critic, storage, evaluate_critic, start + 1, K - 1)
        storage.observations[start].copy_(self.obs)
        if self.metrics_slots > 1 and K > 1:
            # the launch folds all K steps into ring slot step0 % R and clears slot (step0 + K) % R; the slots it skips
            # would otherwise keep counts from R steps ago
            R = self.metrics_slots
            self.metrics_raw[[(self.step_count + i) % R for i in range(1, K)]] = 0
        actor, io = actor_critic.actor.struct(), storage.struct(start)
        A.check(self.lib.wl_drift_rollout_policy(C.byref(self.p), C.byref(self._bufs), C.byref(actor),
                                                 actor_critic.std.data_ptr(), C.byref(io), K, self.seed, self.step_count,
                                                 self._stream()), "wl_drift_rollout_policy")
        self.step_count += K
        if K > 0:
            e = start + K
            self.obs.copy_(storage.observations[e])
            self.reward.copy_(storage.rewards[e - 1])
            self.terminated.copy_(storage.terminated[e - 1])
            self.truncated.copy_(storage.time_outs[e - 1])
            self.dones.copy_(storage.dones[e - 1])
        if evaluate_critic:
            storage.values.copy_(actor_critic.critic(storage.observations).squeeze(-1))
        return storage

    def read_metrics(self, zero: bool = True) -> torch.Tensor:
        m = self.metrics
        if zero:
            self.metrics_raw.zero_()
        return m


class ElevBatch(_MetricsView):
    """n elevation-task envs on one GPU (same SoA state matrix; rows WL_S_CMD_* carry the goal command)."""

    OBS_DIM = A.ELEV_OBS_DIM

    def __init__(self, n_envs: int, device="cuda:0", params: A.WlElevParams | None = None, seed: int = 42,
                 env_offset: int = 0, heightfield=None, metrics_slots: int = 1, startup=None):
        from .params import elev_params
        from .terrain import synthetic_heightfield
        self.lib = A.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise A.HipExtensionMissing("ElevBatch needs a HIP device; there is no CPU path")
        self.n, self.stride = int(n_envs), ((int(n_envs) + 63) // 64) * 64
        self.p = params if params is not None else elev_params()
        self.seed, self.env_offset, self.step_count = int(seed), int(env_offset), 0
        dev = self.device
        self.state = torch.zeros(A.S_COUNT, self.stride, dtype=torch.float32, device=dev)
        self.episode_len = torch.zeros(self.stride, dtype=torch.int32, device=dev)
        self.metrics_slots = int(metrics_slots)
        self.metrics_raw = torch.zeros(self.metrics_slots, A.M_SHARDS, A.M_COUNT, dtype=torch.float32, device=dev)
        self.obs = torch.zeros(self.n, self.OBS_DIM, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.terminated = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.truncated = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.dones = torch.zeros(self.n, dtype=torch.long, device=dev)   # terminated | truncated, as RSL-RL consumes it
        self.hf = DeviceHeightField(heightfield if heightfield is not None else synthetic_heightfield(), dev)
        self.height, self._hf = self.hf.heights, self.hf.struct       # the DECODED fp32 grid (what the kernels see); the ABI struct
        # startup events (elevation cfg :387-407): wheel friction fixed (2.0, 1.0), base mass += U(0.2, 0.5)
        if startup is None:
            from .envs.flatten import StartupSpec
            startup = StartupSpec(wheel_mu_s=(2.0, 2.0), wheel_mu_d=(1.0, 1.0), mu_buckets=5, mu_consistent=False,
                                  damping=(1000.0, 1000.0), mass_add=(0.2, 0.5))
        self._bufs = A.WlEnvBuffers(self.state.data_ptr(), self.episode_len.data_ptr(), None, self.metrics_raw.data_ptr(),
                                    self.stride, self.n, self.env_offset, self.metrics_slots, 0, 0)
        apply_startup_events(self.lib, self._bufs, startup, self.seed, self._stream())
        self._out = A.WlStepOut(self.obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(),
                                self.truncated.data_ptr(), self.dones.data_ptr())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_lanes(self, lanes: int = 0):
        """step-kernel form: 0 = by env count (quad up to 32768 envs), 1 = lane per env, 4 = quad per env"""
        assert lanes in (0, 1, 4)
        self._bufs.lanes = lanes

    def reset(self, mask: torch.Tensor | None = None):
        self.touch_pose()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        A.check(self.lib.wl_elev_reset(C.byref(self.p), C.byref(self._bufs), C.byref(self._hf),
                                       None if m is None else m.data_ptr(), self.seed, self.step_count, self._stream()),
                "wl_elev_reset")

    def observe(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """observation of the current state into self.obs (or a caller's [n, 689] buffer)"""
        out = self.obs if out is None else out
        A.check(self.lib.wl_elev_observe(C.byref(self.p), C.byref(self._bufs), C.byref(self._hf), out.data_ptr(),
                                         self._stream()), "wl_elev_observe")
        return out

    def step(self, actions: torch.Tensor):
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.shape != (self.n, 2):
            actions = actions.to(torch.float32).reshape(self.n, 2).contiguous()
        A.check(self.lib.wl_elev_step(C.byref(self.p), C.byref(self._bufs), C.byref(self._hf), actions.data_ptr(),
                                      C.byref(self._out), self.seed, self.step_count, self._stream()), "wl_elev_step")
        self.step_count += 1
        return self.obs, self.reward, self.terminated, self.truncated

    def rollout(self, actions: torch.Tensor, obs_out=None, rew_out=None, term_out=None, trunc_out=None, dones_out=None,
                persistent: bool = False):
        """K fused steps with pre-staged actions [K,n,2]; optional [K,...] output storage (else overwrite).  persistent=True
        (needs the storage, n <= 32 768) runs them as ONE launch with the state in registers and the height scan of step k
        overlapped with the integration of step k + 1 (wl_elev_rollout_persistent); same results."""
        K = actions.shape[0]
        assert actions.shape == (K, self.n, 2) and actions.dtype == torch.float32 and actions.is_contiguous()
        if persistent and obs_out is not None and self._ring_aliases(K):
            cut = lambda t, a, b: None if t is None else t[a:b]
            for a, b in ((0, 1), (1, K)):
                self.rollout(actions[a:b], obs_out[a:b], rew_out[a:b], term_out[a:b], trunc_out[a:b], cut(dones_out, a, b), True)
            return
        if obs_out is not None:
            out = A.WlStepOut(obs_out.data_ptr(), rew_out.data_ptr(), term_out.data_ptr(), trunc_out.data_ptr(),
                              None if dones_out is None else dones_out.data_ptr())
            os_, vs_ = self.n * self.OBS_DIM, self.n
        else:
            assert not persistent or K == 1, "a persistent rollout needs per-step output rows"
            out, os_, vs_ = self._out, 0, 0
        if persistent and self.metrics_slots > 1 and K > 1:
            R = self.metrics_slots       # the launch folds all K steps into slot step0 % R: the slots it skips keep old counts
            self.metrics_raw[[(self.step_count + i) % R for i in range(1, K)]] = 0
        fn = self.lib.wl_elev_rollout_persistent if persistent else self.lib.wl_elev_rollout
        A.check(fn(C.byref(self.p), C.byref(self._bufs), C.byref(self._hf), actions.data_ptr(), C.byref(out), os_, vs_, K, self.seed,
                   self.step_count, self._stream()), "wl_elev_rollout")
        self.step_count += K


    def collect_step(self, actor_critic, storage, k: int, deterministic: bool = False):
        """rows k of the storage <- policy(observation row k); env.step(); observation row k + 1, reward / flags / dones rows k
        -- the runner's collection step (modified_rsl_rl_runner.py:70-80) as ONE launch (wl_elev_collect_step).
        `actor_critic`: the kernel view (policy.ActorCritic: .actor, .critic, .std).  Quad form only (n <= 32 768)."""
        st = storage
        key = (st.observations.data_ptr(), actor_critic.actor.w1.data_ptr(), actor_critic.critic.w1.data_ptr(), actor_critic.std.data_ptr())
        if getattr(self, "_collect_key", None) != key:
            assert st.n_envs == self.n and st.observations.shape[2] == self.OBS_DIM and st.observations.is_contiguous()
            self._collect_key = key
            self._collect_nets = (actor_critic.actor.struct(), actor_critic.critic.struct())
        a, c = self._collect_nets
        obs = st.observations
        io = A.WlCollectIo(obs[k].data_ptr(), st.actions[k].data_ptr(), st.mu[k].data_ptr(), st.actions_log_prob[k].data_ptr(),
                           st.values[k].data_ptr())
        out = A.WlStepOut(obs[k + 1].data_ptr(), st.rewards[k].data_ptr(), st.terminated[k].data_ptr(), st.time_outs[k].data_ptr(),
                          st.dones[k].data_ptr())
        A.check(self.lib.wl_elev_collect_step(C.byref(self.p), C.byref(self._bufs), C.byref(self._hf), C.byref(a), C.byref(c),
                                              actor_critic.std.data_ptr(), C.byref(io), C.byref(out), int(bool(deterministic)),
                                              self.seed, self.step_count, self._stream()), "wl_elev_collect_step")
        self.step_count += 1

    def collect_rollout(self, actor_critic, storage, start: int = 0, count: int | None = None, deterministic: bool = False):
        """rows start .. start + count - 1 of the storage (and observation row start + count) from observation row `start`: the
        runner's whole collection loop as ONE launch (wl_elev_collect_rollout: the actor's first layer in the blocks' registers,
        their observation rows in LDS, the critic beside the physics).  Quad form only (n <= 32 768)."""
        st = storage
        count = st.n_steps - start if count is None else int(count)
        if self._ring_aliases(count):
            self.collect_rollout(actor_critic, storage, start, 1, deterministic)
            return self.collect_rollout(actor_critic, storage, start + 1, count - 1, deterministic)
        key = (st.observations.data_ptr(), actor_critic.actor.w1.data_ptr(), actor_critic.critic.w1.data_ptr(), actor_critic.std.data_ptr())
        if getattr(self, "_collect_key", None) != key:
            assert st.n_envs == self.n and st.observations.shape[2] == self.OBS_DIM and st.observations.is_contiguous()
            self._collect_key = key
            self._collect_nets = (actor_critic.actor.struct(), actor_critic.critic.struct())
        a, c = self._collect_nets
        obs, k = st.observations, int(start)
        io = A.WlCollectIo(obs[k].data_ptr(), st.actions[k].data_ptr(), st.mu[k].data_ptr(), st.actions_log_prob[k].data_ptr(),
                           st.values[k].data_ptr())
        out = A.WlStepOut(obs[k + 1].data_ptr(), st.rewards[k].data_ptr(), st.terminated[k].data_ptr(), st.time_outs[k].data_ptr(),
                          st.dones[k].data_ptr())
        if self.metrics_slots > 1 and count > 1:
            R = self.metrics_slots       # the launch folds all its steps into slot step0 % R: the slots it skips keep old counts
            self.metrics_raw[[(self.step_count + i) % R for i in range(1, count)]] = 0
        A.check(self.lib.wl_elev_collect_rollout(C.byref(self.p), C.byref(self._bufs), C.byref(self._hf), C.byref(a), C.byref(c),
                                                 actor_critic.std.data_ptr(), C.byref(io), C.byref(out), count, int(bool(deterministic)),
                                                 self.seed, self.step_count, self._stream()), "wl_elev_collect_rollout")
        self.step_count += count


class VisualBatch(_MetricsView):
    """n visual-task envs on one GPU: flat black/white traversability plane + ray-cast grey camera."""

    OBS_DIM = A.VIS_OBS_DIM

    def __init__(self, n_envs: int, device="cuda:0", params=None, seed: int = 42, env_offset: int = 0, trav_map=None,
                 spacing=(0.5, 0.5), metrics_slots: int = 1, startup=None, map_kwargs=None):
        import numpy as np

        from .params import visual_params
        from .travmap import generate_traversability_map, spawn_cells
        self.lib = A.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise A.HipExtensionMissing("VisualBatch needs a HIP device; there is no CPU path")
        self.n, self.stride = int(n_envs), ((int(n_envs) + 63) // 64) * 64
        self.p = params if params is not None else visual_params()
        self.seed, self.env_offset, self.step_count = int(seed), int(env_offset), 0
        dev = self.device
        self.state = torch.zeros(A.S_COUNT, self.stride, dtype=torch.float32, device=dev)
        self.episode_len = torch.zeros(self.stride, dtype=torch.int32, device=dev)
        self.metrics_slots = int(metrics_slots)
        self.metrics_raw = torch.zeros(self.metrics_slots, A.M_SHARDS, A.M_COUNT, dtype=torch.float32, device=dev)
        self.obs = torch.zeros(self.n, self.OBS_DIM, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.terminated = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.truncated = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.dones = torch.zeros(self.n, dtype=torch.long, device=dev)   # terminated | truncated, as RSL-RL consumes it
        if trav_map is None:  # generated at construction from a seeded RNG (the reference uses the global numpy RNG)
            trav_map = generate_traversability_map(rng=np.random.RandomState(self.seed), **(map_kwargs or {}))
        trav_map = np.ascontiguousarray(np.asarray(trav_map, dtype=bool))
        self.trav_map = torch.from_numpy(trav_map.astype(np.uint8)).to(dev)
        self.cells = torch.from_numpy(spawn_cells(trav_map)).contiguous().to(dev)
        # one bit per cell (bit k & 31 of word k >> 5, k = iy * cols + ix): what the camera kernels keep in LDS
        bits = np.packbits(np.concatenate([trav_map.reshape(-1), np.zeros((-trav_map.size) % 32, bool)]), bitorder="little")
        self.trav_bits = torch.from_numpy(bits.view(np.int32).copy()).to(dev)
        self._map = A.WlTravMap(self.trav_map.data_ptr(), self.cells.data_ptr(), trav_map.shape[0], trav_map.shape[1],
                                self.cells.shape[0], float(spacing[0]), float(spacing[1]), self.trav_bits.data_ptr())
        if startup is None:
            from .envs.flatten import StartupSpec
            startup = StartupSpec(wheel_mu_s=(0.5, 0.5), wheel_mu_d=(0.5, 0.5), damping=(1000.0, 1000.0), mass_add=(0.0, 0.0))
        self._bufs = A.WlEnvBuffers(self.state.data_ptr(), self.episode_len.data_ptr(), None, self.metrics_raw.data_ptr(),
                                    self.stride, self.n, self.env_offset, self.metrics_slots, 0, 0)
        apply_startup_events(self.lib, self._bufs, startup, self.seed, self._stream())
        self._out = A.WlStepOut(self.obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(),
                                self.truncated.data_ptr(), self.dones.data_ptr())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_lanes(self, lanes: int = 0):
        """step-kernel form: 0 = by env count (quad up to 32768 envs), 1 = lane per env, 4 = quad per env"""
        assert lanes in (0, 1, 4)
        self._bufs.lanes = lanes

    def sample_augmentation(self, generator: torch.Generator | None = None):
        """one (brightness, contrast, blur sigma) per call, like torchvision's ColorJitter(brightness=.8, contrast=.2)
        and GaussianBlur(5, sigma=(0.1, 5)) on a batched tensor (mdp_sensors/observations.py:21-23)"""
        # ColorJitter.forward: fn_idx = torch.randperm(4) over (brightness, contrast, saturation, hue), then the factors; ONE
        # draw per call, i.e. per batch (the reference passes the whole [B, C, H, W] tensor).  Hue / saturation: see the header.
        order = torch.randperm(4, generator=generator).tolist()
        u = torch.rand(3, generator=generator)
        self.p.brightness = float(0.2 + 1.6 * u[0])
        self.p.contrast = float(0.8 + 0.4 * u[1])
        self.p.blur_sigma = float(0.1 + 4.9 * u[2])
        self.p.contrast_first = int(order.index(1) < order.index(0))

    def reset(self, mask: torch.Tensor | None = None):
        self.touch_pose()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        A.check(self.lib.wl_visual_reset(C.byref(self.p), C.byref(self._bufs), C.byref(self._map),
                                         None if m is None else m.data_ptr(), self.seed, self.step_count, self._stream()),
                "wl_visual_reset")

    def observe(self) -> torch.Tensor:
        A.check(self.lib.wl_visual_observe(C.byref(self.p), C.byref(self._bufs), C.byref(self._map), self.obs.data_ptr(),
                                           self._stream()), "wl_visual_observe")
        return self.obs

    def step(self, actions: torch.Tensor):
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.shape != (self.n, 2):
            actions = actions.to(torch.float32).reshape(self.n, 2).contiguous()
        A.check(self.lib.wl_visual_step(C.byref(self.p), C.byref(self._bufs), C.byref(self._map), actions.data_ptr(),
                                        C.byref(self._out), self.seed, self.step_count, self._stream()), "wl_visual_step")
        self.step_count += 1
        return self.obs, self.reward, self.terminated, self.truncated

    def rollout(self, actions: torch.Tensor, obs_out=None, rew_out=None, term_out=None, trunc_out=None, dones_out=None,
                persistent: bool = False):
        """K fused steps with pre-staged actions [K,n,2]; optional [K,...] output storage (else overwrite).  persistent=True
        (needs the storage, n <= 32 768) runs them as ONE launch with the camera of step k rendered while step k + 1 is
        integrated (wl_visual_rollout_persistent); same results."""
        K = actions.shape[0]
        assert actions.shape == (K, self.n, 2) and actions.dtype == torch.float32 and actions.is_contiguous()
        if persistent and obs_out is not None and self._ring_aliases(K):
            cut = lambda t, a, b: None if t is None else t[a:b]
            for a, b in ((0, 1), (1, K)):
                self.rollout(actions[a:b], obs_out[a:b], rew_out[a:b], term_out[a:b], trunc_out[a:b], cut(dones_out, a, b), True)
            return
        if obs_out is not None:
            out = A.WlStepOut(obs_out.data_ptr(), rew_out.data_ptr(), term_out.data_ptr(), trunc_out.data_ptr(),
                              None if dones_out is None else dones_out.data_ptr())
            os_, vs_ = self.n * self.OBS_DIM, self.n
        else:
            assert not persistent or K == 1, "a persistent rollout needs per-step output rows"
            out, os_, vs_ = self._out, 0, 0
        if persistent and self.metrics_slots > 1 and K > 1:
            R = self.metrics_slots       # the launch folds all K steps into slot step0 % R: the slots it skips keep old counts
            self.metrics_raw[[(self.step_count + i) % R for i in range(1, K)]] = 0
        fn = self.lib.wl_visual_rollout_persistent if persistent else self.lib.wl_visual_rollout
        A.check(fn(C.byref(self.p), C.byref(self._bufs), C.byref(self._map), actions.data_ptr(), C.byref(out), os_, vs_, K, self.seed,
                   self.step_count, self._stream()), "wl_visual_rollout")
        self.step_count += K

    def depth(self, heightfield, max_depth: float = 20.0, out: torch.Tensor | None = None) -> torch.Tensor:
        """distance_to_image_plane of the camera against a heightfield -> [n, 60, 80] (BASELINE config 5); `heightfield` is
        (height [ny, nx], x0, y0, cell) or a DepthCamera (build it once when rendering every step)"""
        cam = heightfield if isinstance(heightfield, DepthCamera) else _cached_depth_camera(self, heightfield)
        return cam.render(self, max_depth, out)


class VisualDepthBatch(VisualBatch):
    """n envs of the visual-DEPTH extension task (BASELINE.json configs[4]; not a reference id): the visual task's step driven on a
    heightfield terrain with the camera's 60 x 80 depth image as the observation -- obs [n, 4808] = distance_to_image_plane |
    base_lin_vel | base_ang_vel | last_action.  Two launches per env.step(): the step (wl_visual_step_hf: lane or quad of lanes
    = env, HeightFieldGround contacts) and the depth ray-cast (wl_visual_depth_rows).  `heightfield`: (height [ny, nx], x0, y0,
    cell), default the synthetic 800 x 800 terrain; the traversability map should cover it (default: an 80 x 80 map of 0.5 m
    cells = the terrain's 40 m square, generated like the reference's from the seed)."""

    OBS_DIM = A.VISDEPTH_OBS_DIM

    def __init__(self, n_envs: int, device="cuda:0", params=None, seed: int = 42, env_offset: int = 0, trav_map=None,
                 spacing=(0.5, 0.5), metrics_slots: int = 1, startup=None, map_kwargs=None, heightfield=None, max_depth: float = 20.0):
        from .terrain import synthetic_heightfield
        if trav_map is None and map_kwargs is None:
            map_kwargs = dict(map_size=(80, 80), env_size=(40, 40), sub_group_size=(20, 20), num_walkers=1)
        super().__init__(n_envs, device, params, seed, env_offset, trav_map, spacing, metrics_slots, startup, map_kwargs)
        self.hf = DeviceHeightField(heightfield if heightfield is not None else synthetic_heightfield(), self.device)
        self.height = self.hf.heights
        self.max_depth = float(max_depth)
        self.camera = DepthCamera(self.hf, self.device, self.p)
        self._hf = self.camera._hf
        self.obs = torch.zeros(self.n, self.OBS_DIM, dtype=torch.float32, device=self.device)
        self._out = A.WlStepOut(self.obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(),
                                self.truncated.data_ptr(), self.dones.data_ptr())

    def reset(self, mask: torch.Tensor | None = None):
        self.touch_pose()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        A.check(self.lib.wl_visual_reset_hf(C.byref(self.p), C.byref(self._bufs), C.byref(self._map), C.byref(self._hf),
                                            None if m is None else m.data_ptr(), self.seed, self.step_count, self._stream()),
                "wl_visual_reset_hf")

    def observe(self, out: torch.Tensor | None = None) -> torch.Tensor:
        out = self.obs if out is None else out
        A.check(self.lib.wl_visual_depth_observe(C.byref(self.p), C.byref(self._bufs), C.byref(self._hf), self.camera.pyramid.data_ptr(),
                                                 self.max_depth, out.data_ptr(), self._stream()), "wl_visual_depth_observe")
        return out

    def sample_augmentation(self, generator=None):
        """the depth image is not augmented (mdp_sensors/observations.py:93-95 returns the raw distances)"""

    def step(self, actions: torch.Tensor):
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.shape != (self.n, 2):
            actions = actions.to(torch.float32).reshape(self.n, 2).contiguous()
        self._step_into(actions.data_ptr(), self._out)
        return self.obs, self.reward, self.terminated, self.truncated

    def _step_into(self, actions_ptr, out):
        A.check(self.lib.wl_visual_depth_step(C.byref(self.p), C.byref(self._bufs), C.byref(self._map), C.byref(self._hf),
                                              self.camera.pyramid.data_ptr(), self.max_depth, actions_ptr, C.byref(out), self.seed,
                                              self.step_count, self._stream()), "wl_visual_depth_step")
        self.step_count += 1

    def rollout(self, actions: torch.Tensor, obs_out=None, rew_out=None, term_out=None, trunc_out=None, dones_out=None,
                persistent: bool = False):
        """K steps with pre-staged actions [K, n, 2]; optional [K, ...] output storage (else overwritten in place)"""
        K = actions.shape[0]
        assert actions.shape == (K, self.n, 2) and actions.dtype == torch.float32 and actions.is_contiguous() and not persistent
        for k in range(K):
            out = self._out if obs_out is None else A.WlStepOut(obs_out[k].data_ptr(), rew_out[k].data_ptr(), term_out[k].data_ptr(),
                                                                trunc_out[k].data_ptr(), None if dones_out is None else dones_out[k].data_ptr())
            self._step_into(actions[k].data_ptr(), out)

    def depth(self, heightfield=None, max_depth: float | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
        """the task's own terrain unless another heightfield is given"""
        if heightfield is None:
            return self.camera.render(self, self.max_depth if max_depth is None else max_depth, out)
        return super().depth(heightfield, 20.0 if max_depth is None else max_depth, out)


class DepthCamera:
    """The visual task's pinhole camera rendering distance_to_image_plane against a heightfield (wl_visual_depth): owns the
    device copy of the field and its max-pyramid (built once), renders the poses of ANY batch (rows WL_S_PX.. / WL_S_QW.. of
    its state matrix).  Reference hook: mdp_sensors/observations.py:89-95; camera visual/mushr_visual_env_cfg.py:230-246."""

    IMG_H, IMG_W = 60, 80

    def __init__(self, heightfield, device="cuda:0", params: A.WlVisualParams | None = None, outside_z: float = 0.0):
        from .params import visual_params
        self.lib = A.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise A.HipExtensionMissing("DepthCamera needs a HIP device; there is no CPU path")
        self.p = params if params is not None else visual_params()
        self.hf = DeviceHeightField(heightfield, self.device, outside_z)     # a tuple (quantised here) or a batch's own `.hf` (shared)
        self.height, self._hf = self.hf.heights, self.hf.struct
        ny, nx = self.hf.codes.shape
        n_f = int(self.lib.wl_heightfield_pyramid_floats(nx, ny))
        if n_f <= 0:
            raise A.WlError(f"heightfield of {nx} x {ny} points is outside the pyramid's range")
        self.pyramid = torch.empty(n_f, dtype=torch.float32, device=self.device)
        A.check(self.lib.wl_heightfield_build_pyramid(C.byref(self._hf), self.pyramid.data_ptr(), self._stream()),
                "wl_heightfield_build_pyramid")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def render(self, batch, max_depth: float = 20.0, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(batch.n, self.IMG_H, self.IMG_W, dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == batch.n * self.IMG_H * self.IMG_W
        A.check(self.lib.wl_visual_depth(C.byref(self.p), C.byref(batch._bufs), C.byref(self._hf), self.pyramid.data_ptr(),
                                         float(max_depth), out.data_ptr(), self._stream()), "wl_visual_depth")
        return out


def _cached_depth_camera(batch, heightfield) -> DepthCamera:
    """one DepthCamera per (batch, heightfield): the pyramid is a SNAPSHOT of the field, built on first use and rebuilt when the
    array object, its placement (x0, y0, cell), its shape or -- for tensors -- its in-place version counter changes"""
    cache = batch.__dict__.setdefault("_depth_cameras", {})
    if isinstance(heightfield, DeviceHeightField):
        h, x0, y0, cell, zs = heightfield.codes, heightfield.x0, heightfield.y0, heightfield.cell, heightfield.z_scale
    else:
        h, x0, y0, cell = heightfield[:4]
        zs = float(heightfield[4]) if len(heightfield) > 4 else None       # the same codes under another vertical scale: another field
    key = (id(h), float(x0), float(y0), float(cell), tuple(h.shape), getattr(h, "_version", None), zs)
    cam = cache.get(key)
    if cam is None or cam._src is not h:
        for k in [k for k in cache if k[0] == id(h)]:      # an older snapshot of the same array
            del cache[k]
        cam = DepthCamera(heightfield, batch.device, batch.p if isinstance(batch.p, A.WlVisualParams) else None)
        cam._src = h    # keeps the key's object alive: an id is only unique among live objects
        cache[key] = cam
    return cam

"""GPU parity tests (run on a real MI355X: `pytest -m gpu`).  Everything goes through the C ABI.

Bars: integer / boolean outputs bit-exact (away from decision boundaries that fp32 rounding can flip, which are
counted and bounded); fp32 outputs within the tolerance written at each assert.  The oracle is pinned to the
reference by tests/test_oracle_golden_*.py; here the HIP kernels are compared (a) directly with the reference's
golden outputs and (b) with the oracle on seeded inputs.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import drift_mdp as OM
from oracle import drift_step as OS
from oracle import params as OP
from oracle import philox as PH

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    from wheeledlab_amd import _abi as A
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return A.load()


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(DEV).contiguous()


def soa(a, stride):
    """[N,k] -> device [k, stride]"""
    n, k = a.shape
    t = torch.zeros(k, stride, dtype=torch.float32)
    t[:, :n] = torch.from_numpy(np.ascontiguousarray(a.T))
    return t.to(DEV)


def test_philox_bit_exact(lib):
    n = 1000
    out = torch.empty(4, n, device=DEV)
    for seed, step, stream in ((42, 0, 0), (2 ** 40 + 17, 2 ** 33 + 5, 6), (0, 123456, 3)):
        assert lib.wl_philox_uniform(n, seed, step, stream, out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        want = PH.uniform4(np.arange(n), step, stream, seed)
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    # the device against the Random123 distribution's 7-round vectors (counter 0 / key 0; all ones): u01 keeps the top 24 bits of a word
    for env, step, stream, seed, words in ((0, 0, 0, 0, (0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48)),
                                           (0xFFFFFFFF, 2 ** 64 - 1, 0xFFFFFFFF, 2 ** 64 - 1, (0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662))):
        want = [np.float32((w >> 8) * 2.0 ** -24) for w in words]
        if env == 0:
            assert lib.wl_philox_uniform(1, seed, step, stream, out.data_ptr(), None) == 0
            torch.cuda.synchronize()
            got = out.cpu().numpy().reshape(-1)[[0, 1, 2, 3]]
            assert [float(g) for g in got] == [float(w) for w in want]
        u = PH.uniform4(np.array([env]), step, stream, seed)[:, 0]
        assert [float(g) for g in u] == [float(w) for w in want]
    # known-answer test from the Random123 distribution (ten rounds of the round function the draws use seven of): counter 0, key 0
    x = PH.philox4x32(np.array([0]), 0, 0, 0, rounds=10)[:, 0]
    assert [hex(int(v)) for v in x] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


@pytest.mark.parametrize("tag", ["n256", "edges"])
def test_drift_mdp_kernel_matches_reference_golden(lib, golden, tag):
    from wheeledlab_amd import params as PP
    g = golden(f"drift_mdp_{tag}")
    n = g["pos"].shape[0]
    stride = ((n + 63) // 64) * 64
    p = PP.drift_params()
    p.weight[3] = 0.0
    terms = torch.zeros(7, stride, device=DEV)
    rew = torch.zeros(n, device=DEV)
    term = torch.zeros(n, dtype=torch.uint8, device=DEV)
    obs = torch.zeros(n, 14, device=DEV)
    bufs = [soa(g[k], stride) for k in ("pos", "quat", "lin_vel_b", "ang_vel_b", "ang_vel_w")]
    steer = soa(g["joint_pos"][:, 0:2], stride)
    act = soa(g["actions"], stride)
    rc = lib.wl_drift_mdp(C.byref(p), n, stride, *[b.data_ptr() for b in bufs], steer.data_ptr(), act.data_ptr(), None,
                          terms.data_ptr(), rew.data_ptr(), term.data_ptr(), obs.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    T = terms.cpu().numpy()[:, :n]
    names = ["side_slip", "vel_dist", "track_progress_rate", "turn_left_go_right", "energy_through_turn", "cross_track_dist"]
    for i, name in enumerate(names):
        # fp32 tolerance: 1e-5 relative + 1e-5 absolute (atan2f / sqrtf differ from torch-CPU by a few ulp)
        np.testing.assert_allclose(T[i], g[name], rtol=1e-5, atol=1e-5, err_msg=name)
    np.testing.assert_array_equal(term.cpu().numpy().astype(bool), g["cart_off_track"])  # bit-exact
    # is_terminated_term and the weighted sum against the oracle
    o_terms = OM.drift_terms(OP.drift_params(), g["pos"], g["lin_vel_b"], g["ang_vel_b"], g["ang_vel_w"],
                             g["joint_pos"][:, 0:2], g["cart_off_track"], np.zeros(n, bool))
    np.testing.assert_array_equal(T[6], o_terms[6])
    o_rew, _ = OM.reward_sum(OP.drift_params(), o_terms)
    np.testing.assert_allclose(rew.cpu().numpy(), o_rew, rtol=1e-5, atol=2e-4)
    o_obs = OM.blind_obs(OP.drift_params(), g["pos"], g["quat"], g["lin_vel_b"], g["ang_vel_b"], g["actions"], None)
    got = obs.cpu().numpy()
    # euler angles wrap at 2pi: compare on the circle
    d = np.abs(got - o_obs)
    d[:, 3:6] = np.minimum(d[:, 3:6], 2 * np.pi - d[:, 3:6])
    assert d.max() < 2e-5


def test_action_map_matches_reference_golden(lib, golden):
    from wheeledlab_amd import params as PP
    g = golden("actions")
    n = g["actions"].shape[0]
    a = dev(g["actions"])
    for tag, ap in (("rwd", PP.mushr_action(0)), ("4wd", PP.mushr_action(1)),
                    ("f1tenth", PP.mushr_action(1, base_length=0.365, base_width=0.284))):
        ap.clip_wrapper = 0  # the golden action-term outputs are without the ClipAction wrapper
        proc = torch.zeros(n, 2, device=DEV)
        st = torch.zeros(n, 2, device=DEV)
        wh = torch.zeros(n, 4, device=DEV)
        assert lib.wl_action_map(C.byref(ap), n, a.data_ptr(), proc.data_ptr(), st.data_ptr(), wh.data_ptr(), None) == 0
        torch.cuda.synchronize()
        np.testing.assert_allclose(proc.cpu().numpy(), g[f"{tag}_processed"], rtol=1e-6, atol=1e-6)
        # tan(delta) comes from the hardware sin / cos units (bounded steering angle): 5e-6 abs
        np.testing.assert_allclose(st.cpu().numpy(), g[f"{tag}_steer_pos_target"], rtol=5e-6, atol=5e-6)
        w = wh.cpu().numpy()
        if tag == "rwd":
            np.testing.assert_allclose(w[:, :2], g["rwd_wheel_vel_target"], rtol=1e-6, atol=1e-5)
            assert (w[:, 2:] == 0).all()
        else:
            np.testing.assert_allclose(w, g[f"{tag}_wheel_vel_target"], rtol=2e-5, atol=1e-3)
    # map 2: the base class AckermannAction (ackermann_actions.py:150-201; tanh bounding, reverse allowed): the steer joints take
    # the true Ackermann angles atan(L / (R -+ W / 2)), the wheels the 4WD speeds -- against the reference's own outputs
    ap = PP.mushr_action(2)
    ap.clip_wrapper, ap.bounding, ap.no_reverse = 0, 2, 0
    proc, st, wh = torch.zeros(n, 2, device=DEV), torch.zeros(n, 2, device=DEV), torch.zeros(n, 4, device=DEV)
    assert lib.wl_action_map(C.byref(ap), n, a.data_ptr(), proc.data_ptr(), st.data_ptr(), wh.data_ptr(), None) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(proc.cpu().numpy(), g["base_processed"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(st.cpu().numpy(), g["base_steer_pos_target"], rtol=1e-5, atol=2e-6)
    assert (np.abs(g["base_steer_pos_target"][:, 0] - g["base_steer_pos_target"][:, 1]) > 1e-3).any()      # left != right: not the tan map
    np.testing.assert_allclose(wh.cpu().numpy(), g["base_wheel_vel_target"], rtol=2e-5, atol=1e-3)
    ap.map = 3
    assert lib.wl_action_map(C.byref(ap), n, a.data_ptr(), proc.data_ptr(), st.data_ptr(), wh.data_ptr(), None) == -1      # WL_EINVAL


def _fresh(n, seed=3, **kw):
    from wheeledlab_amd.core import DriftBatch
    env = DriftBatch(n, device=DEV, seed=seed, **kw)
    env.reset()
    torch.cuda.synchronize()
    return env


def test_reset_kernel_matches_oracle(lib):
    env = _fresh(300, seed=11)
    st = env.state.cpu().numpy()
    p = OP.drift_params()
    o = np.zeros_like(st)
    o[OS.QW] = 1
    o[23:27] = st[23:27]
    ep = np.ones(st.shape[1], np.int32)
    OS.reset_envs(p, o, ep, env.ref_table.cpu().numpy(), np.arange(300), 11, 0)
    np.testing.assert_allclose(st[:, :300], o[:, :300], rtol=1e-6, atol=1e-6)
    assert (env.episode_len.cpu().numpy()[:300] == 0).all()
    # every reset pose is within pos_noise of the centre line (events.py:122-124)
    d = OM.cross_track_dist(st[:3, :300].T, 0.8, 0.8, 0.0, 1.0)
    assert d.max() <= 0.5 * np.sqrt(2) + 1e-5


def _single_step_parity(env, p, n, seed, mode, steps=40):
    """Each step starts from the device state (copied to the host), so differences do not accumulate:
    tolerance 2e-4 abs/rel on state (4 sub-steps of fp32 with hardware rcp/rsq vs numpy), rewards 2e-3 abs
    (weights up to 5000 * dt amplify), observation 1e-3.  `p`: the oracle's parameter set (a namespace, or the product's
    own ctypes struct: same field names)."""
    if mode == "no_corruption":
        env.p.enable_corruption = 0
        p.enable_corruption = 0
    max_len = int(p.max_episode_length)
    rng = np.random.RandomState(0)
    ref = env.ref_table.cpu().numpy()
    flips = 0
    for k in range(steps):
        st = env.state.cpu().numpy().copy()
        ep = env.episode_len.cpu().numpy().copy()
        if k == steps // 2:  # push a quarter of the envs to the end of their episode -> time_out path
            ep[: n // 4] = max_len - 1
            env.episode_len.copy_(torch.from_numpy(ep))
        a = rng.uniform(-1.2, 1.2, (n, 2)).astype(np.float32)
        a[:, 0] = np.abs(a[:, 0])
        noise_t, noise_np = None, None
        if mode == "noise_tensor":
            noise_np = rng.normal(size=(12, env.stride)).astype(np.float32)
            noise_t = dev(noise_np)
        met0 = env.metrics.cpu().numpy().astype(np.float64)
        obs, rew, term, trunc = env.step(dev(a), noise_t)
        torch.cuda.synchronize()
        met = np.zeros(16)
        o_obs, o_rew, o_term, o_trunc, info = OS.step(p, st, ep, ref, a, seed, k, met,
                                                       noise_np[:, :n] if noise_np is not None else None)
        got = env.state.cpu().numpy()
        g_term = term.cpu().numpy().astype(bool)
        np.testing.assert_array_equal(trunc.cpu().numpy().astype(bool), o_trunc)
        bad = g_term != o_term
        flips += int(bad.sum())
        ok = ~bad
        np.testing.assert_allclose(got[:23, :n][:, ok], st[:23, :n][:, ok], rtol=2e-4, atol=2e-4, err_msg=f"step {k}")
        np.testing.assert_allclose(got[27:35, :n][:, ok], st[27:35, :n][:, ok], rtol=2e-3, atol=2e-3)
        np.testing.assert_array_equal(env.episode_len.cpu().numpy()[:n][ok], ep[:n][ok])
        np.testing.assert_allclose(rew.cpu().numpy()[ok], o_rew[ok], rtol=2e-3, atol=2e-3)
        d = np.abs(obs.cpu().numpy() - o_obs)[ok]
        d[:, 3:6] = np.minimum(d[:, 3:6], np.abs(2 * np.pi - d[:, 3:6]))
        assert d.max() < 1e-3, (k, d.max())
        if not bad.any():
            dm = env.metrics.cpu().numpy().astype(np.float64) - met0
            np.testing.assert_allclose(dm[8:16], met[8:16], atol=1e-3)
            np.testing.assert_allclose(dm[:8], met[:8], rtol=1e-3, atol=5e-2)
    assert flips <= 2, f"{flips} termination decisions differ (expected only at fp32 boundary ties)"


@pytest.mark.parametrize("lanes", [4, 1, 2])
@pytest.mark.parametrize("mode", ["philox", "noise_tensor", "no_corruption"])
def test_fused_step_matches_oracle_single_steps(lib, mode, lanes):
    n = 1024
    env = _fresh(n, seed=5)
    env.set_lanes(lanes)      # every form of the step kernel: quad-per-env (latency), lane-per-env with packed axles,
                              # lane-per-env with the scalar wheel loop (throughput)
    _single_step_parity(env, OP.drift_params(), n, 5, mode)


def _f1tenth_batch(n, seed):
    """the F1Tenth drift variant exactly as the registry builds it (drifting/f1tenth_drift_env_cfg.py:42-161,
    wheeledlab_assets/f1tenth.py:9-27): 4WD drive train (vehicle.drive = 1), 4WD action map, L 0.365 / W 0.284, F1Tenth
    actuator constants, startup randomisation of the throttle damping of all four wheels"""
    from wheeledlab_amd import registry, tasks  # noqa: F401
    from wheeledlab_amd.core import DriftBatch
    from wheeledlab_amd.envs.flatten import flatten_drift_cfg
    flat = flatten_drift_cfg(registry.parse_env_cfg("Isaac-F1TenthDriftRL-v0", device=DEV, num_envs=n))
    flat.params.action.clip_wrapper = 1     # gym's ClipAction folded into the kernel (driven with +-1.2 actions below)
    env = DriftBatch(n, device=DEV, seed=seed, params=flat.params, startup=flat.startup)
    env.reset()
    torch.cuda.synchronize()
    p = flat.params
    assert p.vehicle.drive == 1 and p.action.map == 1 and abs(p.action.base_length - 0.365) < 1e-6 and abs(p.action.base_width - 0.284) < 1e-6
    assert abs(p.vehicle.motor_sat - 1.0) < 1e-6 and abs(p.vehicle.motor_vel_limit - 400.0) < 1e-3 and abs(p.vehicle.steer_kp - 120.0) < 1e-4
    return env, flat


@pytest.mark.parametrize("lanes", [4, 1, 2])
@pytest.mark.parametrize("mode", ["philox", "no_corruption"])
def test_f1tenth_fused_step_matches_oracle_single_steps(lib, mode, lanes):
    """the drive = 1 instantiations of the drift step kernel (all three forms) against the oracle, with the parameter struct
    the registry flattens from the F1Tenth cfg handed to BOTH sides (the oracle reads the same field names)"""
    n = 1024
    env, flat = _f1tenth_batch(n, seed=5)
    env.set_lanes(lanes)
    damp = env.state[25, :n]
    assert float(damp.min()) >= 10.0 and float(damp.max()) <= 50.0 and float(damp.std()) > 5.0   # randomize_gains (:57-65)
    _single_step_parity(env, flat.params, n, 5, mode)


def test_f1tenth_full_size_properties(lib):
    """size-independent invariants of the F1Tenth variant at BASELINE's env count (as test_full_size_properties)"""
    n = 4096
    env, flat = _f1tenth_batch(n, seed=1)
    K = 300
    g = torch.Generator(device=DEV).manual_seed(0)
    resets, moved = 0, 0.0
    for k in range(K):
        obs, rew, term, trunc = env.step(torch.rand(n, 2, device=DEV, generator=g) * 2 - 1)
        resets += int((term | trunc).sum())
        moved = max(moved, float(env.state[7:9, :n].norm(dim=0).mean()))
    torch.cuda.synchronize()
    st = env.state[:, :n]
    assert torch.isfinite(st).all() and torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert ((st[3:7] ** 2).sum(0).sqrt() - 1).abs().max() < 1e-5
    assert (obs[:, 12:14].abs() <= 1).all() and (env.episode_len[:n] < 250).all()
    m = env.metrics.cpu().numpy()
    assert m[8] == resets and resets > 0 and m[14] == 0
    assert st[2].abs().max() < 0.05 and st[17].abs().max() <= 0.5312 and moved > 0.2     # on the plane, steering bounded, driving
    # all four wheels are driven: front wheels spin up under throttle with the car held on an open plane
    env.p.r_out, env.p.r_in, env.p.max_episode_length = 1e18, 0.0, 10 ** 9
    a = torch.zeros(n, 2, device=DEV)
    a[:, 0] = 1.0
    for _ in range(25):
        env.step(a)
    assert float(env.state[15:17, :n].mean()) > 10.0 and float(env.state[13:15, :n].mean()) > 10.0


def test_rollout_api_equals_step_api(lib):
    n, K = 512, 16
    a = torch.rand(K, n, 2, device=DEV) * 2 - 1
    e1, e2 = _fresh(n, seed=9), _fresh(n, seed=9)
    obs_all = torch.zeros(K, n, 14, device=DEV)
    rew_all = torch.zeros(K, n, device=DEV)
    te = torch.zeros(K, n, dtype=torch.uint8, device=DEV)
    tr = torch.zeros(K, n, dtype=torch.uint8, device=DEV)
    e1.rollout(a, obs_all, rew_all, te, tr)
    for k in range(K):
        obs, rew, term, trunc = e2.step(a[k])
        assert torch.equal(obs, obs_all[k]) and torch.equal(rew, rew_all[k]) and torch.equal(term, te[k])
    assert torch.equal(e1.state, e2.state) and torch.equal(e1.episode_len, e2.episode_len)
    # same episodes ended in both: counts are exact, reward sums equal up to the order of the float atomics
    assert torch.equal(e1.metrics[8:16], e2.metrics[8:16])
    torch.testing.assert_close(e1.metrics[:8], e2.metrics[:8], rtol=1e-5, atol=1e-3)
    assert e1.step_count == e2.step_count == K


@pytest.mark.parametrize("n,ring", [(512, 1), (4096, 64), (100, 1)])
def test_persistent_rollout_matches_stepping(lib, n, ring):
    """wl_drift_rollout_persistent keeps the rows in registers across K steps and must reproduce K wl_drift_step calls.
    The two kernels inline the same step function but the compiler contracts FMAs differently in each, so equality is
    to fp32 rounding on the first steps and to trajectory-divergence tolerance at the end of the rollout."""
    from wheeledlab_amd.core import DriftBatch
    K = 37
    a = torch.rand(K, n, 2, device=DEV) * 2.4 - 1.2
    e1 = DriftBatch(n, device=DEV, seed=13, metrics_slots=ring)
    e2 = DriftBatch(n, device=DEV, seed=13, metrics_slots=ring)
    e1.reset(), e2.reset()
    for env in (e1, e2):                       # start mid-episode so that time-outs happen inside the rollout
        env.episode_len[: n // 3] = 230
    e1.p.enable_corruption = e2.p.enable_corruption = 0          # compare clean observations
    obs_all = torch.zeros(K, n, 14, device=DEV)
    rew_all = torch.zeros(K, n, device=DEV)
    te = torch.zeros(K, n, dtype=torch.bool, device=DEV)
    tr = torch.zeros(K, n, dtype=torch.bool, device=DEV)
    e1.rollout(a, obs_all, rew_all, te, tr, persistent=True)
    msum = torch.zeros(16, device=DEV)
    same = torch.ones(n, dtype=torch.bool, device=DEV)            # envs whose discrete events agreed so far
    for k in range(K):
        slot_idx = e2.step_count % ring
        obs, rew, term, trunc = e2.step(a[k])
        assert torch.equal(trunc, tr[k]), k
        same &= term == te[k]
        d = (obs - obs_all[k]).abs()
        d[:, 3:6] = torch.minimum(d[:, 3:6], (2 * np.pi - d[:, 3:6]).abs())
        tol = 2e-5 if k < 3 else 5e-3                              # rounding first, then (bounded) divergence
        assert d[same].max() < tol, (k, float(d[same].max()))
        if ring > 1:
            msum += e2.metrics[slot_idx]
    assert same.float().mean() > 0.99
    ds = (e1.state[:23, :n] - e2.state[:23, :n]).abs()[:, same]
    assert ds[:13].max() < 5e-3 and torch.equal(e1.episode_len[:n][same], e2.episode_len[:n][same])
    assert tr.any() and e1.step_count == e2.step_count == K
    if bool(same.all()):
        m1 = e1.metrics[0] if ring > 1 else e1.metrics
        m2 = msum if ring > 1 else e2.metrics
        assert torch.allclose(m1[8:], m2[8:]) and torch.allclose(m1[:8], m2[:8], rtol=1e-3, atol=0.5)
    if ring > 1:   # all K steps' metrics land in the rollout's first slot; the slot after the rollout is clean
        assert (e1.metrics[K % ring] == 0).all()


@pytest.mark.parametrize("n", [4096, 32768])
def test_full_size_properties(lib, n):
    """size-independent invariants at BASELINE.json's sizes (4096 / GPU, 32768 = 8 x 4096)"""
    env = _fresh(n, seed=1)
    K = 300
    g = torch.Generator(device=DEV).manual_seed(0)
    resets = 0
    for k in range(K):
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        obs, rew, term, trunc = env.step(a)
        resets += int((term | trunc).sum())
    torch.cuda.synchronize()
    st = env.state[:, :n]
    assert torch.isfinite(st).all() and torch.isfinite(obs).all() and torch.isfinite(rew).all()
    qn = (st[3:7] ** 2).sum(0).sqrt()
    assert (qn - 1).abs().max() < 1e-5                     # unit quaternions
    assert (obs[:, 12:14].abs() <= 1).all()                 # last_action clip
    env.p.enable_corruption = 0
    clean = env.observe().clone()
    env.p.enable_corruption = 1
    assert (clean[:, 3:6] >= 0).all() and (clean[:, 3:6] < 2 * np.pi + 1e-5).all()   # euler wrapped to [0, 2pi)
    assert torch.equal(clean[:, 0:3], env.state[0:3, :n].T)                          # root_pos_w passes through
    ep = env.episode_len[:n]
    assert (ep >= 0).all() and (ep < 250).all()
    m = env.metrics.cpu().numpy()
    assert m[8] == resets and m[9] + m[10] >= resets and m[14] == 0   # metric conservation, no non-finite envs
    assert st[2].abs().max() < 0.05 and st[17].abs().max() <= 0.5312   # car stays on the plane; steer <= tan(0.488)
    # determinism: same seed, same actions -> bitwise identical state
    env2 = _fresh(n, seed=1)
    g = torch.Generator(device=DEV).manual_seed(0)
    for k in range(K):
        env2.step(torch.rand(n, 2, device=DEV, generator=g) * 2 - 1)
    assert torch.equal(env.state, env2.state) and torch.equal(env.episode_len, env2.episode_len)


def test_ragged_and_tiny_sizes(lib):
    for n in (1, 63, 64, 65, 257):
        env = _fresh(n, seed=2)
        obs, rew, term, trunc = env.step(torch.zeros(n, 2, device=DEV))
        torch.cuda.synchronize()
        assert obs.shape == (n, 14) and torch.isfinite(obs).all()
        assert (env.state[:, n:] == 0).all() and (env.episode_len[n:] == 0).all()   # padding columns are never written


def test_size_independent_physics_and_api_properties(lib):
    """invariants that need no oracle, at BASELINE's env count: friction-limited acceleration, idempotent observe / reset,
    masked reset touches only the masked envs, zero action brings every car to rest"""
    n = 4096
    env = _fresh(n, seed=21)
    env.p.enable_pushes = 0                      # pushes are velocity jumps: keep them out of the acceleration bound
    env.p.max_episode_length = 10 ** 9
    env.p.r_out, env.p.r_in = 1e18, 0.0          # open plane: no resets during the physics checks
    g = torch.Generator(device=DEV).manual_seed(3)
    mu_s = (env.state[23, :n] * 1.1)             # wheel x ground static friction, "multiply" combine
    worst = torch.zeros(n, device=DEV)
    v_prev = env.state[7:9, :n].clone()
    for k in range(150):
        env.step(torch.rand(n, 2, device=DEV, generator=g) * 2 - 1)
        v = env.state[7:9, :n]
        if k > 5:
            worst = torch.maximum(worst, (v - v_prev).norm(dim=0) / 0.02 / (mu_s * 9.81))
        v_prev = v.clone()
    assert float(worst.max()) < 1.15             # |a_horizontal| <= mu_s g (+ load-transfer / discretisation slack)
    assert float(worst.mean()) > 0.2             # ... and the bound is actually exercised
    # observe() is idempotent and does not touch the state
    st = env.state.clone()
    o1 = env.observe().clone()
    o2 = env.observe().clone()
    assert torch.equal(o1, o2) and torch.equal(st, env.state)
    # masked reset: only the masked envs change, and resetting twice at the same step gives the same pose
    mask = torch.zeros(n, dtype=torch.bool, device=DEV)
    mask[::7] = True
    env.reset(mask)
    s1 = env.state.clone()
    assert torch.equal(s1[:19, :n][:, ~mask], st[:19, :n][:, ~mask])
    assert (s1[7:13, :n][:, mask] == 0).all() and (env.episode_len[:n][mask] == 0).all()
    env.reset(mask)
    assert torch.equal(env.state, s1)
    # zero throttle: the motor holds the driven wheels at 0 and every car comes to rest on the plane
    for _ in range(200):
        env.step(torch.zeros(n, 2, device=DEV))
    assert float(env.state[7:9, :n].norm(dim=0).max()) < 0.05 and float(env.state[13:15, :n].abs().max()) < 0.5
    assert float(env.state[2, :n].abs().max()) < 5e-3


def test_observation_noise_is_gaussian_on_the_device():
    """The corruption of the policy observation (GaussianNoise std 0.1 / 0.1 / 0.5 / 0.4 on position, Euler angles, linear and angular
    velocity: wheeledlab_tasks/common/observations.py:27-50) is drawn on the device from 16-bit uniforms through Box-Muller (csrc/wl_rng.h:
    radius and angle take 65 536 values each, |z| <= 4.8, Philox at 7 rounds) where the reference calls torch.randn.  Bit equality with
    the oracle's identical quantiser says nothing about the DISTRIBUTION; this does: 4096 envs x 300 steps, the noisy observation minus
    the clean one of the same state, per term -- mean and std within 1 % of sigma, excess kurtosis within 0.05, the mass beyond 3 sigma
    within 10 % of a normal's 0.0027, the twelve components uncorrelated, and no repetition from step to step."""
    n, K = 4096, 300
    noisy, clean = _fresh(n, seed=31), _fresh(n, seed=31)
    clean.p.enable_corruption = 0
    assert noisy.p.enable_corruption == 1
    g = torch.Generator(device=DEV).manual_seed(3)
    z = torch.empty(K, n, 12, device=DEV)
    for k in range(K):
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        on, oc = noisy.step(a)[0], clean.step(a)[0]
        z[k] = on[:, :12] - oc[:, :12]
        assert torch.equal(on[:, 12:], oc[:, 12:])                   # last_action carries no noise
    assert torch.equal(noisy.state, clean.state)                     # the noise touches the observation only
    z = z.double()
    sig = torch.tensor([0.1] * 6 + [0.5] * 3 + [0.4] * 3, device=DEV, dtype=torch.float64)
    u = z / sig                                                      # ~ N(0, 1), 1.2 M samples per component
    mean, std = u.mean((0, 1)), u.std((0, 1))
    kurt = ((u - mean) ** 4).mean((0, 1)) / std ** 4 - 3.0
    tail = (u.abs() > 3.0).double().mean((0, 1))
    assert mean.abs().max() < 0.01, mean
    assert (std - 1).abs().max() < 0.01, std
    # per term (3 components each, 3.7 M samples): kurtosis to 0.05, 3-sigma mass to 10 %
    assert kurt.view(4, 3).mean(1).abs().max() < 0.05, kurt
    t4 = tail.view(4, 3).mean(1)
    assert ((t4 / 0.0026998) - 1).abs().max() < 0.10, t4
    assert u.abs().max() < 4.9                                        # the quantiser's reach (documented: |z| <= 4.8)
    c = torch.corrcoef(u.reshape(-1, 12).T)
    assert (c - torch.eye(12, device=DEV, dtype=torch.float64)).abs().max() < 5e-3, c
    lag = (u[1:] * u[:-1]).mean((0, 1))                               # step-to-step correlation of each component
    assert lag.abs().max() < 5e-3, lag

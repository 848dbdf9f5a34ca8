"""Host logic of the drop-in surface that needs no GPU: configclass semantics, the task registry, cfg -> kernel
parameter flattening, curriculum scalar logic against the reference's golden trajectory."""
import numpy as np
import pytest

from wheeledlab_amd import params as PP
from wheeledlab_amd import registry, tasks  # noqa: F401  (registers the ids)
from wheeledlab_amd.envs import mdp
from wheeledlab_amd.envs.configclass import configclass
from wheeledlab_amd.envs.flatten import flatten_drift_cfg
from wheeledlab_amd.envs.managers_cfg import RewardTermCfg
from wheeledlab_amd.tasks.drifting import MushrDriftPlayEnvCfg, MushrDriftRLEnvCfg


def test_configclass_semantics():
    @configclass
    class A:
        x: int = 1
        lst: list = [1, 2]

        def __post_init__(self):
            self.y = self.x * 2

    @configclass
    class B(A):
        z = A()

    a, b = A(), A(x=3)
    a.lst.append(9)
    assert b.lst == [1, 2] and b.y == 6 and A().lst == [1, 2]      # defaults are deep-copied per instance
    c = B()
    assert c.z.x == 1 and c.replace(x=5).x == 5 and c.x == 1
    assert B().to_dict()["z"]["lst"] == [1, 2]
    with pytest.raises(TypeError):
        A(nope=1)


def test_registry_has_reference_ids():
    assert "Isaac-MushrDriftRL-v0" in registry.registered_ids()
    s = registry.spec("Isaac-MushrDriftRL-v0")
    assert set(s.kwargs) == {"env_cfg_entry_point", "rsl_rl_cfg_entry_point", "play_env_cfg_entry_point"}
    cfg = registry.parse_env_cfg("Isaac-MushrDriftRL-v0", device="cuda:0", num_envs=16)
    assert cfg.scene.num_envs == 16 and cfg.sim.dt == 0.005 and cfg.decimation == 4 and cfg.episode_length_s == 5
    agent = registry.load_cfg_from_registry("Isaac-MushrDriftRL-v0", "rsl_rl_cfg_entry_point")
    assert agent.num_steps_per_env == 128 and agent.policy.actor_hidden_dims == [64, 64]


def _struct_eq(a, b, path=""):
    for name, _ in a._fields_:
        x, y = getattr(a, name), getattr(b, name)
        if hasattr(x, "_fields_"):
            _struct_eq(x, y, path + name + ".")
        elif hasattr(x, "__len__"):
            assert list(x) == pytest.approx(list(y), rel=1e-6), path + name
        else:
            assert x == pytest.approx(y, rel=1e-6), path + name


def test_flatten_rss_drift_cfg_equals_kernel_defaults():
    flat = flatten_drift_cfg(MushrDriftRLEnvCfg())
    want = PP.drift_params()
    want.action.clip_wrapper = 0          # the ClipAction wrapper is applied by the harness, not by the env cfg
    _struct_eq(flat.params, want)
    assert flat.startup.wheel_mu_s == (0.3, 0.5) and flat.startup.mu_buckets == 20 and flat.startup.damping == (10.0, 50.0)
    assert flat.startup.mass_add == (0.3, 0.5)
    assert [n for n, _ in flat.reward_names] == ["side_slip", "vel", "progress", "tlgr", "turn_energy", "cross_track", "term_pens"]
    assert flat.termination_names == {"time_out": "time_out", 0: "out_of_bounds"}
    assert [n for n, _ in flat.curriculum] == ["more_slip", "more_tlgr", "more_term_pens"]


def test_flatten_overrides_and_play_cfg():
    cfg = MushrDriftRLEnvCfg()
    cfg.rewards.side_slip.weight = 100.0                      # the documented Hydra override (wheeledlab_rl/docs/README.md:68-72)
    cfg.rewards.cross_track.params["track_radius"] = 1.0
    cfg.observations.policy.enable_corruption = False
    p = flatten_drift_cfg(cfg).params
    assert p.weight[0] == 100.0 and p.r_line == 1.0 and p.enable_corruption == 0
    play = flatten_drift_cfg(MushrDriftPlayEnvCfg())
    assert list(play.params.weight) == [0.0] * 8 and play.params.max_episode_length == 2 ** 31 - 1
    assert play.params.r_out > 1e20 and play.params.pos_noise == 0.0 and play.params.yaw_noise == 0.0


def test_flatten_routes_unknown_terms_to_the_torch_fallback_and_rejects_what_cannot_be_expressed():
    from wheeledlab_amd.envs.managers_cfg import ObservationTermCfg, TerminationTermCfg
    cfg = MushrDriftRLEnvCfg()
    cfg.rewards.extra = RewardTermCfg(func=lambda env: None, weight=1.0)      # set on the instance, as IsaacLab users do
    cfg.terminations.mine = TerminationTermCfg(func=lambda env: None)
    cfg.observations.policy.more = ObservationTermCfg(func=lambda env: None)
    flat = flatten_drift_cfg(cfg)
    assert [n for n, _ in flat.custom_rewards] == ["extra"]
    assert [n for n, _ in flat.custom_terminations] == ["mine"] and [n for n, _ in flat.custom_obs] == ["more"]
    assert flat.termination_names == {"time_out": "time_out", 0: "out_of_bounds"}     # the built-in ones stay fused
    # replacing a built-in termination by a plain callable: that term moves to the fallback, the kernel's slot is disabled
    cfg = MushrDriftRLEnvCfg()
    cfg.terminations.out_of_bounds.func = lambda env: None
    flat = flatten_drift_cfg(cfg)
    assert [n for n, _ in flat.custom_terminations] == ["out_of_bounds"] and 0 not in flat.termination_names
    assert flat.params.r_in == 0.0 and flat.params.r_out > 1e29
    # what neither the kernel nor a torch term can express is still refused
    cfg = MushrDriftRLEnvCfg()
    cfg.scene.terrain.physics_material.friction_combine_mode = "average"
    with pytest.raises(NotImplementedError):
        flatten_drift_cfg(cfg)
    cfg = MushrDriftRLEnvCfg()
    cfg.observations.policy.base_lin_vel_term.func = lambda env: None     # the fused block's layout is fixed
    with pytest.raises(NotImplementedError):
        flatten_drift_cfg(cfg)


class _FakeRM:
    def __init__(self, w):
        self.c = {k: RewardTermCfg(func=None, weight=v) for k, v in w.items()}

    def get_term_cfg(self, n):
        return self.c[n]

    def set_term_cfg(self, n, c):
        self.c[n] = c


def test_curriculum_matches_reference_golden(golden):
    g = golden("curriculum")

    class E:
        max_episode_length = 250
        common_step_counter = 0
    env = E()
    env.reward_manager = _FakeRM({"side_slip": 10.0, "tlgr": 0.0, "term_pens": -5000.0})
    cur = flatten_drift_cfg(MushrDriftRLEnvCfg()).curriculum
    got = []
    for step in g["steps"]:
        env.common_step_counter = int(step)
        for _, term in cur:
            mdp.increase_reward_weight_over_time(env, None, **term.params)
        got.append([env.reward_manager.c[k].weight for k in ("side_slip", "tlgr", "term_pens")])
    np.testing.assert_array_equal(np.array(got), g["weights"])


def test_product_map_generator_reproduces_the_reference_map(golden):
    """same numpy seed, same draw order => the reference's 500 x 500 traversability map bit for bit"""
    from wheeledlab_amd.travmap import generate_traversability_map, spawn_cells
    g = golden("visual_trav")
    want = np.unpackbits(g["full_map_packed"])[: 500 * 500].reshape(500, 500).astype(bool)
    np.random.seed(0)
    got = generate_traversability_map()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(generate_traversability_map((20, 20), (20, 20), (10, 10), 1, np.random.RandomState(0)).shape, (20, 20))
    cells = spawn_cells(want)
    assert cells.shape[1] == 2 and want[cells[:, 0], cells[:, 1]].all() and len(cells) == want.sum()


def test_visual_random_events_carry_the_wheel_mass_term():
    """VisualEventsRandomCfg (visual/mushr_visual_env_cfg.py:264-299): wheel friction buckets, base mass "abs" AND the wheel
    links' mass "abs" (0.01, 0.3) -- the third term must reach the startup spec (it was dropped before round 4), and the oracle's
    keyed draw adds the four wheel masses to the mass row"""
    import numpy as np

    from oracle import startup as OSU
    from wheeledlab_amd.envs.flatten import flatten_visual_cfg
    from wheeledlab_amd.tasks.visual import MushrVisualRLEnvCfg, MushrVisualRLRandomEnvCfg
    su = flatten_visual_cfg(MushrVisualRLRandomEnvCfg()).startup
    assert su.wheel_mass == (0.01, 0.3) and su.mass_add == (1.0, 3.0) and su.chassis_mass == 0.0
    assert su.wheel_mu_s == (0.4, 0.6) and su.mu_buckets == 10 and not su.mu_consistent
    assert flatten_visual_cfg(MushrVisualRLEnvCfg()).startup.wheel_mass == (0.0, 0.0)
    kw = dict(wheel_mu_s=su.wheel_mu_s, wheel_mu_d=su.wheel_mu_d, mu_buckets=su.mu_buckets, mu_consistent=su.mu_consistent,
              damping=su.damping, chassis_mass=su.chassis_mass, mass_add=su.mass_add)
    base = OSU.draw(4096, 7, 0, **kw)[3]
    full = OSU.draw(4096, 7, 0, wheel_mass=su.wheel_mass, **kw)[3]
    extra = full - base
    assert 0.04 - 1e-6 <= extra.min() and extra.max() <= 1.2 + 1e-6 and abs(extra.mean() - 0.62) < 0.02    # 4 x U(0.01, 0.3)
    assert 1.0 <= base.min() and base.max() <= 3.0
    # a wheel-link term with another operation is refused, not dropped
    cfg = MushrVisualRLRandomEnvCfg()
    cfg.events.add_wheel_mass.params["operation"] = "scale"
    import pytest
    with pytest.raises(NotImplementedError):
        flatten_visual_cfg(cfg)


def test_device_heightfield_forms_and_quantisation():
    """core.DeviceHeightField (the WlHeightField of ABI 21: int16 codes x z_scale): float heights are quantised with the documented
    rule, codes + scale are taken as they are, a decoded grid re-quantises without loss, the struct carries what the kernels read"""
    import numpy as np
    import pytest
    import torch

    from oracle import heightfield as OH
    from wheeledlab_amd import terrain as T
    from wheeledlab_amd.core import DeviceHeightField
    h, x0, y0, cell = T.synthetic_heightfield(64, 0.05, seed=3)
    ho = OH.make_terrain(64, 0.05, seed=3)
    assert np.array_equal(h, ho[0]) and h.dtype == np.float32                      # product and oracle generate one terrain
    d = DeviceHeightField((h, x0, y0, cell), "cpu")
    assert d.codes.dtype == torch.int16 and d.z_scale == 2.0 ** -13 and d.struct.nx == 64 and d.struct.ny == 64
    assert torch.equal(d.heights, torch.from_numpy(h)) and abs(d.struct.z_scale - 2.0 ** -13) == 0     # on the lattice: lossless
    assert d.struct.height == d.codes.data_ptr() and d.struct.outside_z == 0.0
    # arbitrary floats: rounded to the nearest code; the error is at most half a code
    rng = np.random.RandomState(0)
    r = rng.uniform(-2, 3, (17, 23)).astype(np.float32)
    dr = DeviceHeightField((r, 0.0, 0.0, 0.1), "cpu")
    assert np.abs(dr.heights.numpy() - r).max() <= 0.5 * 2.0 ** -13 + 1e-7
    codes, zs = T.quantize_heights(r)
    assert zs == 2.0 ** -13 and np.array_equal(dr.codes.numpy(), codes) and np.array_equal(OH.quantize(r), codes)
    assert np.array_equal(T.decode_heights(codes, zs), OH.decode(codes, zs)) and np.array_equal(dr.heights.numpy(), OH.decode(codes))
    # the row-pair table (ABI 23, include/wheeledlab_amd.h: pair[j][i] = code[j][i] | code[j + 1][i] << 16, the last row with itself)
    pr = dr.pairs.numpy().view(np.uint32)
    up = np.concatenate([codes[1:], codes[-1:]], 0)
    assert pr.shape == codes.shape and np.array_equal((pr & 0xffff).astype(np.uint16).view(np.int16), codes)
    assert np.array_equal((pr >> 16).astype(np.uint16).view(np.int16), up) and dr.struct.pair == dr.pairs.data_ptr()
    from tests.depth_cases import hf_struct
    _s, (_c, p_np) = hf_struct((dr.heights.numpy(), 0.0, 0.0, 0.1))
    assert np.array_equal(p_np, pr)                                                # the host simulations' numpy builder: the same table
    # a range beyond +-4 m: the scale doubles until the codes fit
    tall = DeviceHeightField((r * 4.0, 0.0, 0.0, 0.1), "cpu")
    assert tall.z_scale == 2.0 ** -11 and int(tall.codes.abs().max()) <= 32767
    # IsaacLab's form: int16 codes with their vertical_scale, taken as they are (and required)
    isaac = DeviceHeightField((codes, 0.0, 0.0, 0.1, 0.005), "cpu")
    assert isaac.z_scale == 0.005 and torch.equal(isaac.codes, torch.from_numpy(codes))
    assert torch.equal(isaac.heights, torch.from_numpy(codes.astype(np.float32) * np.float32(0.005)))
    with pytest.raises(ValueError):
        DeviceHeightField((codes, 0.0, 0.0, 0.1), "cpu")
    # float heights with an explicit scale; the decoded grid goes through again unchanged; sharing another field's codes
    coarse = DeviceHeightField((r, 0.0, 0.0, 0.1, 0.005), "cpu")
    again = DeviceHeightField((coarse.heights, 0.0, 0.0, 0.1, 0.005), "cpu")
    assert torch.equal(again.codes, coarse.codes)
    shared = DeviceHeightField(coarse, "cpu", outside_z=-1.0)
    assert shared.codes is coarse.codes and shared.struct.outside_z == -1.0 and coarse.struct.outside_z == 0.0
    bad = r.copy()
    bad[3, 4] = np.nan
    with pytest.raises(ValueError):
        DeviceHeightField((bad, 0.0, 0.0, 0.1), "cpu")
    with pytest.raises(ValueError):
        T.quantize_heights(bad)


def test_device_heightfield_refuses_scales_that_cannot_hold_the_heights_and_shares_across_device_spellings():
    """an explicit z_scale too small for the height range used to clip the codes silently (heights up to 4 m at
    z_scale 1e-4 were flattened at 3.28 m); a non-finite scale passed the `> 0` check; two spellings of one device failed a bare
    assert; the depth-camera cache ignored the vertical scale of a (codes, ..., z_scale) tuple."""
    import torch

    from wheeledlab_amd import terrain
    from wheeledlab_amd.core import DeviceHeightField, _cached_depth_camera, _canonical_device
    h = np.linspace(0.0, 4.0, 64 * 64, dtype=np.float32).reshape(64, 64)
    with pytest.raises(ValueError, match="do not fit"):
        DeviceHeightField((h, 0.0, 0.0, 0.1, 1e-4), "cpu")
    with pytest.raises(ValueError, match="do not fit"):
        terrain.quantize_heights(h, 1e-4)
    assert DeviceHeightField((h, 0.0, 0.0, 0.1, 2e-4), "cpu").heights.max() == pytest.approx(4.0, abs=2e-4)     # 6.55 m of range: fits
    for bad in (float("inf"), float("nan"), 0.0, -1.0):
        with pytest.raises(ValueError):
            DeviceHeightField((h, 0.0, 0.0, 0.1, bad), "cpu")
        with pytest.raises(ValueError):
            terrain.quantize_heights(h, bad)
        with pytest.raises(ValueError):
            DeviceHeightField((np.zeros((4, 4), np.int16), 0.0, 0.0, 0.1, bad), "cpu")
    assert _canonical_device("cpu") == torch.device("cpu")
    assert _canonical_device("cuda").index is not None and _canonical_device("cuda:0") == torch.device("cuda", 0)
    a = DeviceHeightField((h, 0.0, 0.0, 0.1), "cpu")
    assert DeviceHeightField(a, torch.device("cpu")).codes is a.codes
    a.device = torch.device("cuda", 1)                                                   # (a field that lives elsewhere)
    with pytest.raises(ValueError, match="cannot be shared"):
        DeviceHeightField(a, "cpu")
    # the cache key of the scene's depth camera: the same codes under another vertical scale are another field
    import inspect
    src = inspect.getsource(_cached_depth_camera)
    assert "zs)" in src and "heightfield.z_scale" in src and "heightfield[4]" in src

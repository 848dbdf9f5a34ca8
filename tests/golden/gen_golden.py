#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own mdp functions.

Runs only in the authoring container (it reads /root/reference, which does not exist on the GPU box).
Nothing here is imported by the product or by the tests; the tests only read the ``.npz`` files this
script writes.  No reference source text is copied: the reference modules are imported in place, fed
seeded inputs, and only (inputs, outputs) arrays are stored.

How the import works: the reference is a plugin layer on IsaacLab 2.0.2, which is
not installed here.  We inject a stub ``isaaclab`` namespace whose *state accessors* are one-liners over a
fake env (``mdp.root_pos_w = data.root_pos_w - env_origins`` ...), whose cfg classes are permissive
attribute bags, and whose math helpers restate the *published* IsaacLab definitions (those parts are
therefore "parity unpinned" -- see DESIGN.md).  Every arithmetic line that the reference itself owns
(reward / termination / observation / action-term / reset / curriculum / traversability functions) is
executed from the reference's files.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py      (WL_GOLDEN_OUT=<dir> writes there instead;
tests/test_oracle_golden_drift.py::test_committed_vectors_regenerate_from_the_reference does that and compares)
"""
import sys

sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only reference tree

import importlib
import math
import os
import re
import types

import numpy as np
import torch

REF = "/root/reference/source"
OUT = os.environ.get("WL_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))   # WL_GOLDEN_OUT: regenerate elsewhere
# WL_GOLDEN_SEED_OFFSET shifts every INPUT seed (states, actions, query points, reset draws): fresh vectors for
# tests/test_oracle_golden_drift.py::test_oracle_matches_the_reference_on_fresh_seeds (the map-generation seeds stay: the
# tests regenerate those maps from the same numpy seeds)
SEED_OFFSET = int(os.environ.get("WL_GOLDEN_SEED_OFFSET", "0"))

# --------------------------------------------------------------------------------------------------
# 1. stub namespace
# --------------------------------------------------------------------------------------------------


class Bag:
    """Permissive cfg stand-in: kwargs become attributes, unknown attributes materialise as Bags."""

    def __init__(self, *args, **kw):
        self.__dict__["_args"] = args
        for k, v in kw.items():
            setattr(self, k, v)
        post = getattr(type(self), "__post_init__", None)
        if post is not None:
            try:
                post(self)
            except Exception:
                pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        b = Bag()
        self.__dict__[name] = b
        return b

    def replace(self, **kw):
        new = type(self).__new__(type(self))
        new.__dict__.update(self.__dict__)
        for k, v in kw.items():
            setattr(new, k, v)
        return new

    def __call__(self, *a, **k):
        return Bag()


def configclass(cls):
    """identity + kwargs-init + __post_init__ call (enough for the reference's class bodies)."""
    if "__init__" not in cls.__dict__:
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)
            post = getattr(self, "__post_init__", None)
            if post is not None:
                try:
                    post()
                except Exception:
                    pass  # RL-env cfgs touch sim/viewer objects we do not model
        cls.__init__ = __init__
    if not hasattr(cls, "replace"):
        cls.replace = Bag.replace
    return cls


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _cfgcls(name):
    cls = type(name, (Bag,), {})
    # nested cfg classes the reference reaches through attribute access
    cls.InitialStateCfg = type("InitialStateCfg", (Bag,), {})
    cls.OffsetCfg = type("OffsetCfg", (Bag,), {})
    cls.Ranges = type("Ranges", (Bag,), {})
    return cls


class SceneEntityCfg:
    def __init__(self, name, joint_names=None, body_names=None, joint_ids=slice(None), body_ids=slice(None)):
        self.name = name
        self.joint_names = joint_names
        self.body_names = body_names
        self.joint_ids = joint_ids
        self.body_ids = body_ids

    def resolve(self, scene):       # isaaclab: names -> indices on the scene entity
        if self.body_names is not None:
            self.body_ids = scene[self.name].find_bodies(self.body_names)[0]
        if self.joint_names is not None:
            self.joint_ids = scene[self.name].find_joints(self.joint_names)[0]


# ---- published IsaacLab 2.0.2 math (restated; NOT reference-owned -> parity unpinned) ------------


def euler_xyz_from_quat(quat):
    w, x, y, z = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    roll = torch.atan2(2.0 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    sp = 2.0 * (w * y - z * x)
    pitch = torch.where(torch.abs(sp) >= 1, torch.copysign(torch.full_like(sp, math.pi / 2.0), sp), torch.asin(sp))
    yaw = torch.atan2(2.0 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return roll % (2 * math.pi), pitch % (2 * math.pi), yaw % (2 * math.pi)


def quat_from_euler_xyz(roll, pitch, yaw):
    cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
    cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
    cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
    qw = cy * cr * cp + sy * sr * sp
    qx = cy * sr * cp - sy * cr * sp
    qy = cy * cr * sp + sy * sr * cp
    qz = sy * cr * cp - cy * sr * sp
    return torch.stack([qw, qx, qy, qz], dim=-1)


def matrix_from_quat(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


# ---- fake env --------------------------------------------------------------------------------------

MUSHR_JOINTS = [
    "front_left_wheel_steer", "front_right_wheel_steer",
    "back_left_wheel_throttle", "back_right_wheel_throttle",
    "front_left_wheel_throttle", "front_right_wheel_throttle",
    "front_left_wheel_suspension", "front_right_wheel_suspension",
    "back_left_wheel_suspension", "back_right_wheel_suspension",
]


class FakeAsset:
    def __init__(self, data, joint_names=MUSHR_JOINTS):
        self.data = data
        self.joint_names = joint_names
        self.vel_target = None
        self.pos_target = None
        self.written_pose = None
        self.written_vel = None

    def find_joints(self, name_keys):
        if isinstance(name_keys, str):
            name_keys = [name_keys]
        ids, names = [], []
        for i, n in enumerate(self.joint_names):
            if any(re.fullmatch(k, n) for k in name_keys):
                ids.append(i)
                names.append(n)
        return ids, names

    body_names = ["base_link", "back_left_wheel_link", "back_right_wheel_link", "front_left_wheel_link", "front_right_wheel_link"]

    def find_bodies(self, name_keys):
        if isinstance(name_keys, str):
            name_keys = [name_keys]
        ids = [i for i, n in enumerate(self.body_names) if any(re.fullmatch(k, n) for k in name_keys)]
        return ids, [self.body_names[i] for i in ids]

    def set_joint_velocity_target(self, t, joint_ids=None):
        self.vel_target = (t.clone(), list(joint_ids))

    def set_joint_position_target(self, t, joint_ids=None):
        self.pos_target = (t.clone(), list(joint_ids))

    def write_root_pose_to_sim(self, pose, env_ids=None):
        self.written_pose = pose.clone()

    def write_root_velocity_to_sim(self, vel, env_ids=None):
        self.written_vel = vel.clone()


class FakeScene(dict):
    def __init__(self, robot, n):
        super().__init__(robot=robot)
        self.env_origins = torch.zeros(n, 3)
        self.sensors = {}
        self.terrain = Bag()

    def __getitem__(self, k):
        if k in self.sensors:
            return self.sensors[k]
        return dict.__getitem__(self, k)


class FakeRewardManager:
    def __init__(self, weights):
        self.cfgs = {k: types.SimpleNamespace(weight=float(v)) for k, v in weights.items()}

    def get_term_cfg(self, name):
        return self.cfgs[name]

    def set_term_cfg(self, name, cfg):
        self.cfgs[name] = cfg


class FakeEnv:
    def __init__(self, n, data):
        self.num_envs = n
        self.device = "cpu"
        self.scene = FakeScene(FakeAsset(data), n)
        self.commands = {}
        self.common_step_counter = 0
        self.max_episode_length = 250
        self.last_action = torch.zeros(n, 2)


def install_stubs():
    _mod("toml", load=lambda p: {"package": {"version": "0.0.0"}})
    _mod("cv2")
    _mod("gymnasium")
    pxr = _mod("pxr")
    for n in ("Usd", "UsdGeom", "UsdPhysics", "Gf"):
        setattr(pxr, n, Bag())
    tv = _mod("torchvision")
    _mod("torchvision.transforms", **{k: (lambda *a, **kw: Bag()) for k in
                                       ("Normalize", "Grayscale", "Compose", "ColorJitter",
                                        "RandomAdjustSharpness", "GaussianBlur")})

    _mod("isaaclab")
    utils = _mod("isaaclab.utils", configclass=configclass)
    _mod("isaaclab.utils.math", euler_xyz_from_quat=euler_xyz_from_quat,
         quat_from_euler_xyz=quat_from_euler_xyz, matrix_from_quat=matrix_from_quat)
    _mod("isaaclab.utils.noise", **{k: _cfgcls(k) for k in
                                    ("AdditiveUniformNoiseCfg", "AdditiveGaussianNoiseCfg", "UniformNoiseCfg",
                                     "GaussianNoiseCfg")})
    _mod("isaaclab.sim", **{k: _cfgcls(k) for k in
                            ("RigidBodyMaterialCfg", "DistantLightCfg", "UsdFileCfg", "RigidBodyPropertiesCfg",
                             "ArticulationRootPropertiesCfg", "GroundPlaneCfg", "PinholeCameraCfg", "SimulationCfg")})
    _mod("isaaclab.scene", InteractiveSceneCfg=type("InteractiveSceneCfg", (Bag,), {"__post_init__": lambda s: None}))
    _mod("isaaclab.terrains", TerrainImporterCfg=_cfgcls("TerrainImporterCfg"), TerrainImporter=Bag)
    _mod("isaaclab.assets", **{k: _cfgcls(k) for k in
                               ("ArticulationCfg", "AssetBaseCfg", "RigidObjectCfg")},
         Articulation=Bag, RigidObject=Bag)
    _mod("isaaclab.actuators", ImplicitActuatorCfg=_cfgcls("ImplicitActuatorCfg"), DCMotorCfg=_cfgcls("DCMotorCfg"))
    _mod("isaaclab.sensors", RayCasterCfg=_cfgcls("RayCasterCfg"), TiledCameraCfg=_cfgcls("TiledCameraCfg"),
         patterns=types.SimpleNamespace(GridPatternCfg=_cfgcls("GridPatternCfg")), Camera=Bag)

    class ManagerTermBase:
        def __init__(self, cfg, env):
            self.cfg = cfg
            self._env = env

        @property
        def device(self):
            return self._env.device

        @property
        def num_envs(self):
            return self._env.num_envs

    class ActionTerm(ManagerTermBase):
        def __init__(self, cfg, env):
            super().__init__(cfg, env)
            self._asset = env.scene[cfg.asset_name]

    class ActionTermCfg:
        asset_name: str = "robot"

        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

    def _termcfg(name):
        def __init__(self, func=None, params=None, weight=None, **kw):
            self.func = func
            self.params = params or {}
            self.weight = weight
            self.__dict__.update(kw)
        return type(name, (), {"__init__": __init__})

    _mod("isaaclab.managers", ManagerTermBase=ManagerTermBase, ActionTerm=ActionTerm, ActionTermCfg=ActionTermCfg,
         SceneEntityCfg=SceneEntityCfg,
         **{k: _termcfg(k) for k in ("EventTermCfg", "RewardTermCfg", "CurriculumTermCfg", "TerminationTermCfg",
                                     "ObservationTermCfg")},
         ObservationGroupCfg=type("ObservationGroupCfg", (), {}))

    envs = _mod("isaaclab.envs", ManagerBasedEnv=Bag, ManagerBasedRLEnv=Bag,
                ManagerBasedRLEnvCfg=type("ManagerBasedRLEnvCfg", (Bag,), {"__post_init__": lambda s: None}))
    _mod("isaaclab.envs.manager_based_rl_env", ManagerBasedRLEnv=Bag)

    # state accessors: one-liners over the fake env (IsaacLab public mdp API)
    def _robot(env):
        return env.scene["robot"]

    def joint_sel(env, asset_cfg):
        if asset_cfg is None or asset_cfg.joint_names is None:
            return slice(None)
        return _robot(env).find_joints(asset_cfg.joint_names)[0]

    def height_scan(env, sensor_cfg, offset=0.5):
        s = env.scene.sensors[sensor_cfg.name]
        return s.data.pos_w[:, 2].unsqueeze(1) - s.data.ray_hits_w[..., 2] - offset

    mdp = _mod(
        "isaaclab.envs.mdp",
        root_pos_w=lambda env, asset_cfg=None: _robot(env).data.root_pos_w - env.scene.env_origins,
        root_quat_w=lambda env, asset_cfg=None: _robot(env).data.root_quat_w,
        base_lin_vel=lambda env, asset_cfg=None: _robot(env).data.root_lin_vel_b,
        base_ang_vel=lambda env, asset_cfg=None: _robot(env).data.root_ang_vel_b,
        root_lin_vel_w=lambda env, asset_cfg=None: _robot(env).data.root_lin_vel_w,
        joint_pos=lambda env, asset_cfg=None: _robot(env).data.joint_pos[:, joint_sel(env, asset_cfg)],
        joint_vel=lambda env, asset_cfg=None: _robot(env).data.joint_vel[:, joint_sel(env, asset_cfg)],
        generated_commands=lambda env, command_name: env.commands[command_name],
        last_action=lambda env, action_name=None: env.last_action,
        height_scan=height_scan,
        euler_xyz_from_quat=euler_xyz_from_quat,
        time_out=None, root_height_below_minimum=None, randomize_rigid_body_material=None,
        randomize_actuator_gains=None, push_by_setting_velocity=None, randomize_rigid_body_mass=None,
    )
    _mod("isaaclab.envs.mdp.rewards", is_terminated_term=None)
    mdp.rewards = sys.modules["isaaclab.envs.mdp.rewards"]
    _mod("isaaclab.envs.mdp.commands", UniformPose2dCommandCfg=_cfgcls("UniformPose2dCommandCfg"))
    _mod("isaaclab.envs.mdp.events", reset_root_state_uniform=None)

    # reference packages: real sub-modules, bare parents (do not execute wheeledlab_tasks/__init__.py)
    for p in ("wheeledlab", "wheeledlab_assets", "wheeledlab_tasks"):
        sys.path.insert(0, os.path.join(REF, p))
    for pkg in ("wheeledlab_tasks", "wheeledlab_tasks.drifting", "wheeledlab_tasks.elevation",
                "wheeledlab_tasks.visual"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, "wheeledlab_tasks", *pkg.split("."))]
        sys.modules[pkg] = m


# --------------------------------------------------------------------------------------------------
# 2. inputs (SURVEY.md 8d "config 1": N=256, seed 0)
# --------------------------------------------------------------------------------------------------


def ref_track_points(rng, n):
    """points on the stadium centre-line (r=0.8, straight half-length 0.8) for realistic positions"""
    L = 2 * math.pi * 0.8 + 4 * 0.8
    d = rng.uniform(0, L, n)
    x = np.empty(n)
    y = np.empty(n)
    for i, di in enumerate(d):
        if di < 1.6:
            x[i], y[i] = 0.8, di - 0.8
        elif di < 1.6 + math.pi * 0.8:
            a = (di - 1.6) / 0.8
            x[i], y[i] = 0.8 * math.cos(a), 0.8 + 0.8 * math.sin(a)
        elif di < 3.2 + math.pi * 0.8:
            x[i], y[i] = -0.8, 0.8 - (di - 1.6 - math.pi * 0.8)
        else:
            a = (di - 3.2 - math.pi * 0.8) / 0.8
            x[i], y[i] = -0.8 * math.cos(a), -0.8 - 0.8 * math.sin(a)
    return x, y


def make_state(n, seed):
    rng = np.random.RandomState(seed + SEED_OFFSET)
    x, y = ref_track_points(rng, n)
    x += rng.uniform(-0.5, 0.5, n)
    y += rng.uniform(-0.5, 0.5, n)
    wild = rng.rand(n) < 0.25
    x = np.where(wild, rng.uniform(-2.5, 2.5, n), x)
    y = np.where(wild, rng.uniform(-3.3, 3.3, n), y)
    pos = np.stack([x, y, np.full(n, 0.05)], -1)
    yaw = rng.uniform(0, 2 * math.pi, n)
    roll = rng.normal(0, 0.02, n)
    pitch = rng.normal(0, 0.02, n)
    quat = quat_from_euler_xyz(*(torch.tensor(a, dtype=torch.float32) for a in (roll, pitch, yaw))).numpy()
    vb = np.stack([rng.uniform(0, 3.5, n), rng.normal(0, 0.8, n), rng.normal(0, 0.05, n)], -1)
    wb = np.stack([rng.normal(0, 0.2, n), rng.normal(0, 0.2, n), rng.normal(0, 1.5, n)], -1)
    ww = wb + rng.normal(0, 0.01, (n, 3))  # world-frame ang vel: an independent input to track_progress_rate
    vw = np.stack([rng.normal(0, 1.5, n), rng.normal(0, 1.5, n), rng.normal(0, 0.1, n)], -1)
    jp = np.zeros((n, 10))
    jp[:, 0:2] = rng.uniform(-0.53, 0.53, (n, 2))
    jv = np.zeros((n, 10))
    jv[:, 2:6] = rng.uniform(-5, 70, (n, 4))
    act = rng.uniform(-1.3, 1.3, (n, 2))  # beyond +-1 to exercise the clip
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    return dict(pos=f32(pos), quat=f32(quat), lin_vel_b=f32(vb), ang_vel_b=f32(wb), ang_vel_w=f32(ww),
                lin_vel_w=f32(vw), joint_pos=f32(jp), joint_vel=f32(jv), actions=f32(act))


def env_from_state(st):
    n = st["pos"].shape[0]
    t = {k: torch.from_numpy(v.copy()) for k, v in st.items()}
    data = types.SimpleNamespace(
        root_pos_w=t["pos"], root_quat_w=t["quat"], root_lin_vel_b=t["lin_vel_b"], root_ang_vel_b=t["ang_vel_b"],
        root_link_ang_vel_w=t["ang_vel_w"], root_lin_vel_w=t["lin_vel_w"], joint_pos=t["joint_pos"],
        joint_vel=t["joint_vel"], default_root_state=torch.zeros(n, 13),
    )
    env = FakeEnv(n, data)
    env.last_action = t["actions"]
    return env


def npy(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------------------------------
# 3. golden sets
# --------------------------------------------------------------------------------------------------


def gen_drift(D):
    out = {}
    for tag, st in (("n256", make_state(256, 0)), ("edges", edge_state())):
        env = env_from_state(st)
        R = D.DriftRewardsCfg
        T = D.DriftTerminationsCfg
        res = dict(st)
        res["side_slip"] = npy(D.side_slip(env, **R.side_slip.params))
        res["vel_dist"] = npy(D.vel_dist(env, **R.vel.params))
        res["track_progress_rate"] = npy(D.track_progress_rate(env))
        res["turn_left_go_right"] = npy(D.turn_left_go_right(env, **R.tlgr.params))
        res["energy_through_turn"] = npy(D.energy_through_turn(env, **R.turn_energy.params))
        res["cross_track_dist"] = npy(D.cross_track_dist(env, **R.cross_track.params))
        res["in_range"] = npy(D.in_range(env, 0.8, 0.3)).astype(np.int64)
        res["off_track"] = npy(D.off_track(env, 0.8, 2.0)).astype(np.int64)
        res["cart_off_track"] = npy(D.cart_off_track(env, **T.out_of_bounds.params))
        res["weights"] = np.array([R.side_slip.weight, R.vel.weight, R.progress.weight, R.tlgr.weight,
                                   R.turn_energy.weight, R.cross_track.weight, R.term_pens.weight], np.float32)
        out[tag] = res
        np.savez_compressed(os.path.join(OUT, f"drift_mdp_{tag}.npz"), **res)
    return out


def edge_state():
    """positions straddling every branch boundary of the track predicates (A9/A15/A16), +-1e-3 .. 1e-1"""
    pts = []
    S, rin, rout = 0.8, 0.3, 2.0
    for eps in (1e-3, 1e-2, 1e-1):
        for sx in (-1, 1):
            for sy in (-1, 1):
                for d in (-eps, eps):
                    pts += [(sx * (rout + d), sy * 0.4), (sx * (rin + d), sy * 0.4), (sx * 0.8, sy * (S + d)),
                            (sx * (rout + d), sy * S * 0.999), (0.0 + d, sy * (S + rout + d)),
                            (sx * (rin + d) * 0.7071, sy * (S + (rin + d) * 0.7071)),
                            (sx * (rout + d) * 0.7071, sy * (S + (rout + d) * 0.7071)),
                            (sx * d, sy * d), (sx * (0.8 + d), sy * (S + 0.8 + d))]
    pts += [(0.0, 0.0), (0.0, 0.8), (0.0, -0.8), (0.8, 0.0), (-0.8, 0.0), (2.0, 0.0), (0.3, 0.0), (0.0, 2.8),
            (0.0, 1.1), (0.0, -1.1), (0.0, -2.8)]
    n = len(pts)
    st = make_state(n, 1)
    st["pos"][:, 0] = np.array([p[0] for p in pts], np.float32)
    st["pos"][:, 1] = np.array([p[1] for p in pts], np.float32)
    # velocity edge cases for side_slip thresholds (|vx|<1, beta around .25 / .55)
    rng = np.random.RandomState(2 + SEED_OFFSET)
    beta = rng.choice([0.2499, 0.2501, 0.5499, 0.5501, 0.0, 0.4, -0.4, -0.2501, -0.5499], n)
    speed = rng.choice([0.999, 1.001, 3.0, 0.5, 2.0], n)
    st["lin_vel_b"][:, 0] = (speed * np.cos(beta)).astype(np.float32)
    st["lin_vel_b"][:, 1] = (speed * np.sin(beta)).astype(np.float32)
    st["lin_vel_b"][::7, 0] *= -1  # reversed cars: atan2 in the far quadrants
    return st


def gen_actions(A, C):
    n = 256
    st = make_state(n, 3)
    out = {"actions": st["actions"]}
    for tag, cfgcls in (("rwd", C.MushrRWDActionCfg), ("4wd", C.Mushr4WDActionCfg), ("f1tenth", C.F1Tenth4WDActionCfg)):
        env = env_from_state(st)
        cfg = cfgcls.throttle_steer
        if tag == "f1tenth":
            env.scene["robot"].joint_names = ["rotator_left", "rotator_right", "wheel_back_left", "wheel_back_right",
                                              "wheel_front_left", "wheel_front_right"]
        term = cfg.class_type(cfg, env)
        a = torch.from_numpy(st["actions"].copy())
        if tag == "rwd":  # drift env overrides the scale in __post_init__ (same values as the default)
            pass
        term.process_actions(a)
        term.apply_actions()
        asset = env.scene["robot"]
        out[f"{tag}_raw"] = npy(term.raw_actions)
        out[f"{tag}_processed"] = npy(term.processed_actions)
        out[f"{tag}_wheel_vel_target"] = npy(asset.vel_target[0])
        out[f"{tag}_wheel_ids"] = np.array(asset.vel_target[1])
        out[f"{tag}_steer_pos_target"] = npy(asset.pos_target[0])
        out[f"{tag}_steer_ids"] = np.array(asset.pos_target[1])
        out[f"{tag}_geom"] = np.array([cfg.base_length, cfg.base_width, cfg.wheel_radius, *cfg.scale], np.float32)
    # zero-steer special case (R = 1e6 branch) and exact-limit inputs
    z = np.array([[1.0, 0.0], [0.5, 0.0], [-1.0, 0.0], [1.0, 1.0], [1.0, -1.0], [0.0, 0.3]], np.float32)
    env = env_from_state({k: v[:6] for k, v in st.items()})
    cfg = C.Mushr4WDActionCfg.throttle_steer
    term = cfg.class_type(cfg, env)
    term.process_actions(torch.from_numpy(z.copy()))
    term.apply_actions()
    out["4wd_special_actions"] = z
    out["4wd_special_wheel_vel_target"] = npy(env.scene["robot"].vel_target[0])
    out["4wd_special_steer_pos_target"] = npy(env.scene["robot"].pos_target[0])
    # base-class true-Ackermann map (not used by a registered task; cheap to pin)
    base_cfg = A.AckermannActionCfg(wheel_joint_names=cfg.wheel_joint_names, steering_joint_names=cfg.steering_joint_names,
                                    base_length=0.325, base_width=0.2, wheel_radius=0.05, scale=(3.0, 0.488),
                                    bounding_strategy="tanh", no_reverse=False, asset_name="robot")
    env = env_from_state(st)
    term = A.AckermannAction(base_cfg, env)
    term.process_actions(torch.from_numpy(st["actions"].copy()))
    # NB the base class signature is (target_steering_angle, target_velocity) but apply_actions passes keywords
    term.apply_actions()
    out["base_processed"] = npy(term.processed_actions)
    out["base_wheel_vel_target"] = npy(env.scene["robot"].vel_target[0])
    out["base_steer_pos_target"] = npy(env.scene["robot"].pos_target[0])
    np.savez_compressed(os.path.join(OUT, "actions.npz"), **out)


def gen_reset(E):
    out = {}
    env = env_from_state(make_state(64, 4))
    cfg = types.SimpleNamespace(params={"track_radius": 0.8, "track_straight_dist": 0.8, "num_points": 20})
    torch.manual_seed(0 + SEED_OFFSET)
    u = torch.rand(20)  # what generate_reference_poses will draw first under the same seed
    torch.manual_seed(0 + SEED_OFFSET)
    term = E.reset_root_state_along_track(cfg, env)
    out["u_dists"] = npy(u)
    out["reference_poses"] = npy(term.reference_poses)  # [20, 2, 3]  (pos | euler deg)
    ids = torch.arange(64)
    torch.manual_seed(7 + SEED_OFFSET)
    idx = torch.randint(20, (64,))
    u_xy = torch.rand(64, 2)
    u_yaw = torch.rand(64)
    torch.manual_seed(7 + SEED_OFFSET)
    term(env, ids, 0.8, 0.8, 20, SceneEntityCfg("robot"), pos_noise=0.5, yaw_noise=1.0)
    out["idx"] = npy(idx)
    out["u_xy"] = npy(u_xy)
    out["u_yaw"] = npy(u_yaw)
    out["pose"] = npy(env.scene["robot"].written_pose)
    out["vel"] = npy(env.scene["robot"].written_vel)
    np.savez_compressed(os.path.join(OUT, "reset_track.npz"), **out)


def gen_curriculum(W, D):
    env = FakeEnv(1, None)
    env.max_episode_length = 250
    R = D.DriftRewardsCfg
    env.reward_manager = FakeRewardManager({"side_slip": R.side_slip.weight, "tlgr": R.tlgr.weight,
                                            "term_pens": R.term_pens.weight})
    C = D.DriftCurriculumCfg
    terms = [C.more_slip.params, C.more_tlgr.params, C.more_term_pens.params]
    steps, traj = [], []
    # _reset_idx is evaluated on steps where >=1 env resets: sample both episode boundaries and off-boundary steps
    for ep in range(0, 400):
        for off in (0, 17):
            env.common_step_counter = ep * 250 + off
            for p in terms:
                W.increase_reward_weight_over_time(env, [0], **p)
            steps.append(env.common_step_counter)
            traj.append([env.reward_manager.cfgs[k].weight for k in ("side_slip", "tlgr", "term_pens")])
    np.savez_compressed(os.path.join(OUT, "curriculum.npz"), steps=np.array(steps, np.int64),
                        weights=np.array(traj, np.float64),
                        params=np.array([[p["increase"], p["episodes_per_increase"], p["max_increases"]] for p in terms]))


def gen_elevation(El):
    n = 256
    st = make_state(n, 5)
    rng = np.random.RandomState(6 + SEED_OFFSET)
    st["pos"][:, 0] = rng.uniform(-19, 19, n)
    st["pos"][:, 1] = rng.uniform(-19, 19, n)
    st["pos"][:, 2] = rng.uniform(0.05, 1.3, n)
    st["lin_vel_b"][:, 0] = rng.uniform(-0.2, 1.5, n)
    st["lin_vel_b"][:, 2] = rng.normal(0, 0.15, n)
    rpy = np.stack([rng.normal(0, 0.5, n), rng.normal(0, 0.5, n), rng.uniform(-3.14, 3.14, n)], -1)
    st["quat"] = npy(quat_from_euler_xyz(*(torch.tensor(rpy[:, i], dtype=torch.float32) for i in range(3))))
    st["joint_vel"][:, 2:6] = rng.uniform(-1, 4, (n, 4))
    cmd = np.concatenate([rng.uniform(-19, 19, (n, 2)), np.zeros((n, 1)), rng.uniform(-3.14, 3.14, (n, 1))], -1)
    cmd[:8, :2] = st["pos"][:8, :2] + rng.uniform(-0.4, 0.4, (8, 2))  # some at-goal cases
    cmd[8, 0] = np.nan  # nan_to_num branch of goal_relative_xyz
    NH = 40  # envs that carry the full 26x26 ray grid (keeps the fixture small)
    hits_z = rng.uniform(-0.2, 1.5, (n, 676))
    hits_z[NH:] = hits_z[NH:, :1]  # remaining envs: one distinct value repeated (compresses away)
    env = env_from_state(st)
    env.commands["goal_pose"] = torch.tensor(cmd, dtype=torch.float32)
    sensor = types.SimpleNamespace(data=types.SimpleNamespace(
        pos_w=torch.tensor(np.concatenate([st["pos"][:, :2], st["pos"][:, 2:3] + 20.0], -1), dtype=torch.float32),
        ray_hits_w=torch.tensor(np.concatenate([np.zeros((n, 676, 2)), hits_z[..., None]], -1), dtype=torch.float32)))
    env.scene.sensors["height_scanner"] = sensor
    res = dict(st)
    res["command"] = cmd.astype(np.float32)
    res["sensor_pos_w"] = npy(sensor.data.pos_w)
    res["ray_hits_z"] = hits_z.astype(np.float32)
    O = El.ElevationObsCfg.ConcatObs
    T = El.ElevationTerminationsCfg
    res["world_height_map"] = npy(El.world_height_map(env, **O.elevation_map.params))
    res["goal_relative_xyz"] = npy(El.goal_relative_xyz(env))
    res["goal_progress_rate"] = npy(El.goal_progress_rate(env))
    res["higher_elevation"] = npy(El.higher_elevation(env))
    res["is_falling_penalty"] = npy(El.is_falling_penalty(env))
    res["forward_vel"] = npy(El.forward_vel(env))
    res["stuck"] = npy(El.stuck(env, **T.stuck.params))
    res["upright_penalty"] = npy(El.upright_penalty(env, 60.0))
    res["upright_bool"] = npy(El.upright_bool(env, **T.rollover.params))
    res["close_to_goal"] = npy(El.close_to_goal(env, **T.at_goal.params))
    R = El.ElevationRewardsCfg
    res["weights"] = np.array([R.vel_towards_goal.weight, R.height_z.weight, R.falling_penalty.weight,
                               R.termination_penalty.weight], np.float32)
    np.savez_compressed(os.path.join(OUT, "elevation_mdp.npz"), **res)
    gen_elevation_unwired(El, st, env)


def gen_elevation_unwired(El, st, env):
    """the reward functions the elevation cfg module DEFINES but does not register (SURVEY.md 8(a) row E12,
    mushr_elevation_env_cfg.py:159-164,175-231,256-266): their outputs on the same 256 states, so that the build's
    torch-fallback restatements (wheeledlab_amd/envs/mdp.py) are pinned too.  elevation_continuity keeps its previous
    elevation in a function attribute: first call (-> zeros), then a second call on shifted heights."""
    out = dict(pos=st["pos"], quat=st["quat"], lin_vel_b=st["lin_vel_b"], ang_vel_b=st["ang_vel_b"], lin_vel_w=st["lin_vel_w"],
               joint_vel=st["joint_vel"])
    out["forward_wheel_spin"] = npy(El.forward_wheel_spin(env))
    out["change_in_elevation"] = npy(El.change_in_elevation(env))
    out["steep_penalty"] = npy(El.steep_penalty(env, 0.2))
    out["yaw_change_onElev"] = npy(El.yaw_change_onElev(env, 0.5, 0.1))
    out["roll_on_elev"] = npy(El.roll_on_elev(env, 0.1, 0.1))
    out["ascending"] = npy(El.ascending(env))
    out["low_vel_penalty"] = npy(El.low_vel_penalty(env, 0.1))
    out["upright_penalty_30"] = npy(El.upright_penalty(env, 30.0))
    if hasattr(El.elevation_continuity, "prev_elevation"):
        del El.elevation_continuity.prev_elevation
    out["elevation_continuity_first"] = npy(El.elevation_continuity(env, 0.1))
    rng = np.random.RandomState(9 + SEED_OFFSET)
    dz = rng.normal(0, 0.05, st["pos"].shape[0]).astype(np.float32)
    st2 = dict(st)
    st2["pos"] = st["pos"].copy()
    st2["pos"][:, 2] += dz
    env2 = env_from_state(st2)
    out["pos_second"] = st2["pos"]
    out["elevation_continuity_second"] = npy(El.elevation_continuity(env2, 0.1))
    np.savez_compressed(os.path.join(OUT, "elevation_unwired.npz"), **out)


def gen_visual(VU, TU):
    # map generation: the reference draws from the global numpy RNG
    np.random.seed(0)
    small = VU.generate_env_map((20, 20), (10, 10), 1)
    np.random.seed(0)
    _, _, _, _, full = VU.generated_colored_plane((500, 500), (0.5, 0.5), (100, 100), (50, 50), 1, False)
    np.random.seed(1)
    poses = VU.generate_random_poses(64, 0.5, 0.5, full.tolist())
    util = TU.TraversabilityHashmapUtil()
    util.set_traversability_hashmap(full.tolist(), (500, 500), (0.5, 0.5))
    rng = np.random.RandomState(2 + SEED_OFFSET)
    xy = rng.uniform(-130, 130, (4096, 2)).astype(np.float32)
    xy[:64] = np.array([(p[0], p[1]) for p in poses], np.float32)
    # cell-boundary cases
    k = np.arange(64, 192)
    xy[k, 0] = (np.round(xy[k, 0] / 0.5) * 0.5 - 0.25 + rng.choice([-1e-4, 0, 1e-4], 128)).astype(np.float32)
    trav = util.get_traversability(torch.from_numpy(xy.copy()))
    xi, yi = util.get_map_id(torch.from_numpy(xy[:, 0].copy()), torch.from_numpy(xy[:, 1].copy()))
    np.savez_compressed(os.path.join(OUT, "visual_trav.npz"), env_map_20=small,
                        full_map_packed=np.packbits(full), poses=np.array(poses, np.float64), xy=xy,
                        trav=npy(trav), x_idx=npy(xi), y_idx=npy(yi))


def gen_visual_terms(st_seed=8):
    """traversable_reward / forward_vel / out_of_map live in the cfg module whose import writes a USD file;
    patch create_geometry to the in-memory generator first so nothing touches the filesystem."""
    VU = importlib.import_module("wheeledlab_tasks.visual.utils")
    TU = importlib.import_module("wheeledlab_tasks.visual.utils.traversability_utils")

    def fake_create(file_path, map_size, spacing, env_size, sub_group_size, num_walkers=16, color_sampling=False):
        np.random.seed(0)
        m = VU.generated_colored_plane(map_size, spacing, env_size, sub_group_size, num_walkers, color_sampling)[4]
        TU.TraversabilityHashmapUtil().set_traversability_hashmap(m.tolist(), map_size, spacing)
        return m.tolist()

    VU.create_geometry = fake_create
    V = importlib.import_module("wheeledlab_tasks.visual.mushr_visual_env_cfg")
    n = 512
    st = make_state(n, st_seed)
    rng = np.random.RandomState(9 + SEED_OFFSET)
    st["pos"][:, 0] = rng.uniform(-128, 128, n)
    st["pos"][:, 1] = rng.uniform(-128, 128, n)
    st["pos"][:4, 0] = [125.0, -125.0, 125.0001, -125.0001]
    env = env_from_state(st)
    env.scene["terrain"] = types.SimpleNamespace(cfg=V.VisualTerrainImporterCfg)
    res = dict(st)
    res["traversable_reward"] = npy(V.traversable_reward(env))
    res["forward_vel"] = npy(V.forward_vel(env))
    res["out_of_map"] = npy(V.out_of_map(env))
    res["weights"] = np.array([V.VisualRewardsCfg.traversablility.weight, V.VisualRewardsCfg.vel_rew.weight], np.float32)
    np.savez_compressed(os.path.join(OUT, "visual_mdp.npz"), **res)
    gen_visual_unwired(V, st, env)


def gen_visual_unwired(V, st, env):
    """the term functions the visual cfg module DEFINES but does not register (mushr_visual_env_cfg.py:314-368,400-403): their
    outputs on 512 states spread over the 250 m map (+ wheel-link positions as an independent input, roll angles over the whole
    circle for roll_over, a second position set on the drift-track scale for off_track), so that the build's torch restatements
    (wheeledlab_amd/envs/mdp.py) are pinned to the reference too"""
    n = st["pos"].shape[0]
    rng = np.random.RandomState(11 + SEED_OFFSET)
    # half of the cars near the paths (a random traversable cell + up to 0.4 m: wheels on and off the path), half anywhere
    TU = importlib.import_module("wheeledlab_tasks.visual.utils.traversability_utils")
    util = TU.TraversabilityHashmapUtil()
    m = np.asarray(torch.as_tensor(util.traversability_hashmap).cpu().numpy(), bool)
    cells = np.argwhere(m)                         # (y_idx, x_idx)
    pick = cells[rng.randint(0, len(cells), n // 2)]
    st = {k: v.copy() for k, v in st.items()}
    st["pos"][: n // 2, 0] = pick[:, 1] * 0.5 - 125.0 + rng.uniform(-0.4, 0.4, n // 2)
    st["pos"][: n // 2, 1] = pick[:, 0] * 0.5 - 125.0 + rng.uniform(-0.4, 0.4, n // 2)
    terrain = env.scene["terrain"]
    env = env_from_state(st)
    env.scene["terrain"] = terrain
    # wheel links: the root position + a body-frame offset per wheel, + the root link itself as body 0 (what `.*wheel_link` must skip)
    off = np.array([[0.0, 0.0], [-0.16, 0.1], [-0.16, -0.1], [0.16, 0.1], [0.16, -0.1]], np.float32)
    yaw = rng.uniform(0, 2 * math.pi, n).astype(np.float32)
    c, s_ = np.cos(yaw), np.sin(yaw)
    body = np.zeros((n, 5, 3), np.float32)
    body[:, :, 0] = st["pos"][:, None, 0] + c[:, None] * off[None, :, 0] - s_[:, None] * off[None, :, 1]
    body[:, :, 1] = st["pos"][:, None, 1] + s_[:, None] * off[None, :, 0] + c[:, None] * off[None, :, 1]
    body[:, :, 2] = 0.05
    # a quarter of the cars straddle a cell line (cells of 0.5 m): wheels on both sides of it
    k = np.arange(0, n, 4)
    body[k, :, 0] += (np.round(body[k, 0, 0] / 0.5) * 0.5 - 0.25 - body[k, 0, 0])[:, None]
    env.scene["robot"].data.body_pos_w = torch.from_numpy(body.copy())
    # roll over the whole circle (the function subtracts pi from an angle wrapped to [0, 2 pi))
    roll = rng.uniform(-math.pi, math.pi, n)
    quat = quat_from_euler_xyz(torch.tensor(roll, dtype=torch.float32), torch.zeros(n), torch.tensor(yaw)).numpy().astype(np.float32)
    env.scene["robot"].data.root_quat_w = torch.from_numpy(quat.copy())
    out = dict(pos=st["pos"], quat=quat, lin_vel_b=st["lin_vel_b"], body_pos_w=body)
    env.common_step_counter, env.max_episode_length = 999 * 50 + 49, 50
    out["bool_is_not_traversable_early"] = npy(V.bool_is_not_traversable(env))
    env.common_step_counter = 1000 * 50
    out["bool_is_not_traversable_late"] = npy(V.bool_is_not_traversable(env))
    out["counters"] = np.array([999 * 50 + 49, 1000 * 50, 50], np.int64)
    out["is_traversable"] = npy(V.is_traversable(env))
    out["is_traversable_speed_scaled"] = npy(V.is_traversable_speed_scaled(env))
    out["is_traversable_wheels"] = npy(V.is_traversable_wheels(env))
    out["binary_is_traversable_wheels"] = npy(V.binary_is_traversable_wheels(env))
    out["vel_rew_trav"] = npy(V.vel_rew_trav(env))
    out["vel_rew_trav_2_3"] = npy(V.vel_rew_trav(env, 2.0, 3.0))
    out["low_speed_penalty"] = npy(V.low_speed_penalty(env))
    out["low_speed_penalty_2"] = npy(V.low_speed_penalty(env, 2.0))
    out["roll_over"] = npy(V.roll_over(env))
    # off_track on the drift track's scale (the function is the drift cfg's, copied into the visual module)
    st2 = make_state(n, 12)
    env2 = env_from_state(st2)
    out["pos_track"] = st2["pos"]
    out["off_track"] = npy(V.off_track(env2, 0.8, 2.0))
    out["off_track_1"] = npy(V.off_track(env2, 0.8, 1.0))
    # the cfg CLASS's own lookup (:188-208 get_traversability / get_map_id: floor, [x_idx, y_idx] order -- not the singleton's rule the
    # terms above use): points all over the map and beyond its edge, some exactly on cell lines
    cfg = object.__new__(V.VisualTerrainImporterCfg)
    pts = rng.uniform(-130.0, 130.0, (n, 2)).astype(np.float32)
    pts[:32] = (np.round(pts[:32] / 0.5) * 0.5 + 0.25).astype(np.float32)          # on the lines of the floor rule
    pts[32:40] = [[-125.0, 125.0], [125.0, -125.0], [-124.75, -124.75], [124.75, 124.75], [0.0, 0.0], [0.25, -0.25], [-0.25, 0.25], [124.9, 0.1]]
    xi, yi = cfg.get_map_id(torch.from_numpy(pts[:, 0].copy()), torch.from_numpy(pts[:, 1].copy()))
    out["cfg_points"], out["cfg_map_id_x"], out["cfg_map_id_y"] = pts, npy(xi), npy(yi)
    out["cfg_traversability"] = npy(torch.as_tensor(cfg.get_traversability(torch.from_numpy(pts.copy()))))
    # the map the reference's singleton holds (what the terms above looked up)
    out["map_packed"], out["map_shape"] = np.packbits(m), np.array(m.shape, np.int64)
    out["spacing"] = np.array([util.row_spacing, util.col_spacing], np.float64)
    np.savez_compressed(os.path.join(OUT, "visual_unwired.npz"), **out)


def main():
    install_stubs()
    A = importlib.import_module("wheeledlab.envs.mdp.actions")
    W = importlib.import_module("wheeledlab.envs.mdp.curriculums")
    C = importlib.import_module("wheeledlab_tasks.common")
    D = importlib.import_module("wheeledlab_tasks.drifting.mushr_drift_env_cfg")
    E = importlib.import_module("wheeledlab_tasks.drifting.mdp.events")
    El = importlib.import_module("wheeledlab_tasks.elevation.mushr_elevation_env_cfg")
    VU = importlib.import_module("wheeledlab_tasks.visual.utils")
    TU = importlib.import_module("wheeledlab_tasks.visual.utils.traversability_utils")
    gen_drift(D)
    gen_actions(A, C)
    gen_reset(E)
    gen_curriculum(W, D)
    gen_elevation(El)
    gen_visual(VU, TU)
    gen_visual_terms()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()

"""Scene / articulation views over the SoA state matrix, exposing the attribute names the reference's term
functions read (`env.scene["robot"].data.root_pos_w`, `.find_joints(...)`, `env.scene.env_origins`, ...; census in
SURVEY.md section 8b).  These are cold-path conveniences for user plugins: [N,k] tensors are assembled from the
[k][N] rows with torch ops on the device.  The fused kernels never go through them."""
from __future__ import annotations

import re

import torch

from .. import _abi as A

MUSHR_JOINT_NAMES = [
    "front_left_wheel_steer", "front_right_wheel_steer",
    "back_left_wheel_throttle", "back_right_wheel_throttle",
    "front_left_wheel_throttle", "front_right_wheel_throttle",
    "front_left_wheel_suspension", "front_right_wheel_suspension",
    "back_left_wheel_suspension", "back_right_wheel_suspension",
]
# rigid bodies of the articulation as far as the reference's terms select them (`body_names=".*wheel_link"`,
# visual/mushr_visual_env_cfg.py:326,336): the root link and the four wheel links, wheels in the state rows' order (bl, br, fl, fr)
MUSHR_BODY_NAMES = ["base_link", "back_left_wheel_link", "back_right_wheel_link", "front_left_wheel_link", "front_right_wheel_link"]


def quat_rotate_inverse(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """body = R(q)^T world, q = (w, x, y, z)"""
    w, u = q[:, 0:1], q[:, 1:4]
    return v * (2.0 * w * w - 1.0) - torch.cross(u, v, dim=-1) * w * 2.0 + u * (u * v).sum(-1, keepdim=True) * 2.0


class ArticulationData:
    def __init__(self, batch, n_joints=len(MUSHR_JOINT_NAMES)):
        self._b = batch
        self._nj = n_joints

    def _rows(self, r0, k):
        b = self._b
        return b.state[r0:r0 + k, : b.n].T.contiguous()

    @property
    def root_pos_w(self):
        return self._rows(A.S_PX, 3)

    root_link_pos_w = root_pos_w

    @property
    def root_quat_w(self):
        return self._rows(A.S_QW, 4)

    root_link_quat_w = root_quat_w

    @property
    def root_lin_vel_w(self):
        return self._rows(A.S_VX, 3)

    root_com_lin_vel_w = root_lin_vel_w

    @property
    def root_ang_vel_w(self):
        return self._rows(A.S_WX, 3)

    root_link_ang_vel_w = root_ang_vel_w
    root_com_ang_vel_w = root_ang_vel_w

    @property
    def root_lin_vel_b(self):
        return quat_rotate_inverse(self.root_quat_w, self.root_lin_vel_w)

    @property
    def root_ang_vel_b(self):
        return quat_rotate_inverse(self.root_quat_w, self.root_ang_vel_w)

    @property
    def root_state_w(self):
        return self._rows(A.S_PX, 13)

    @property
    def root_vel_w(self):
        return self._rows(A.S_VX, 6)

    @property
    def body_pos_w(self):
        """[N, 5, 3]: the root link and the four wheel centres = root pose (+) the vehicle geometry the step kernels use
        (wl_vehicle.h::wheel_contact: wheel centre = root + R (+-half_wheelbase, +-half_track, wheel_z))"""
        v = self._b.p.vehicle
        local = torch.tensor([[0.0, 0.0, 0.0],
                              [-v.half_wheelbase_r, v.half_track, v.wheel_z], [-v.half_wheelbase_r, -v.half_track, v.wheel_z],
                              [v.half_wheelbase_f, v.half_track, v.wheel_z], [v.half_wheelbase_f, -v.half_track, v.wheel_z]],
                             dtype=torch.float32, device=self._b.device)
        q = self.root_quat_w
        w, u = q[:, None, 0:1], q[:, None, 1:4].expand(-1, local.shape[0], -1)
        lv = local[None].expand(q.shape[0], -1, -1)
        # R(q) v = v + 2 w (u x v) + 2 u x (u x v)
        c1 = torch.cross(u, lv, dim=-1)
        return self.root_pos_w[:, None, :] + lv + 2.0 * (w * c1 + torch.cross(u, c1, dim=-1))

    body_link_pos_w = body_pos_w

    @property
    def default_root_state(self):
        d = torch.zeros(self._b.n, 13, device=self._b.device)
        d[:, 3] = 1.0
        return d

    @property
    def joint_pos(self):
        b = self._b
        jp = torch.zeros(b.n, self._nj, device=b.device)
        jp[:, 0] = jp[:, 1] = b.state[A.S_STEER_POS, : b.n]
        return jp  # wheel spin angles are not integrated (no term reads them); suspension deflection is implicit

    @property
    def joint_vel(self):
        b = self._b
        jv = torch.zeros(b.n, self._nj, device=b.device)
        jv[:, 0] = jv[:, 1] = b.state[A.S_STEER_VEL, : b.n]
        jv[:, 2], jv[:, 3] = b.state[A.S_WHEEL_BL, : b.n], b.state[A.S_WHEEL_BR, : b.n]
        jv[:, 4], jv[:, 5] = b.state[A.S_WHEEL_FL, : b.n], b.state[A.S_WHEEL_FR, : b.n]
        return jv


class ArticulationView:
    def __init__(self, batch, joint_names=MUSHR_JOINT_NAMES):
        self._b = batch
        self.joint_names = list(joint_names)   # layout: [steer L, steer R, rear L, rear R, front L, front R, (suspension x4)]
        self.body_names = list(MUSHR_BODY_NAMES)
        self.data = ArticulationData(batch, len(self.joint_names))
        self.num_instances = batch.n

    def find_bodies(self, name_keys, preserve_order=False):
        keys = [name_keys] if isinstance(name_keys, str) else list(name_keys)
        ids = [i for i, n in enumerate(self.body_names) if any(re.fullmatch(k, n) for k in keys)]
        return ids, [self.body_names[i] for i in ids]

    def find_joints(self, name_keys, joint_subset=None, preserve_order=False):
        keys = [name_keys] if isinstance(name_keys, str) else list(name_keys)
        ids, names = [], []
        for i, n in enumerate(self.joint_names):
            if any(re.fullmatch(k, n) for k in keys):
                ids.append(i)
                names.append(n)
        return ids, names

    # plugin reset events (e.g. a user `reset_root_state_*` term) write through these
    def write_root_pose_to_sim(self, pose, env_ids=None):
        ids = slice(None) if env_ids is None else env_ids
        self._b.state[A.S_PX:A.S_PX + 7, ids] = pose.T.to(torch.float32)
        self._b.touch_pose()

    def write_root_velocity_to_sim(self, vel, env_ids=None):
        ids = slice(None) if env_ids is None else env_ids
        self._b.state[A.S_VX:A.S_VX + 6, ids] = vel.T.to(torch.float32)
        self._b.touch_pose()


class RayCasterData:
    """`sensor.data` of the elevation task's height scanner (isaaclab RayCaster, un-vendored): `pos_w` [N, 3] and
    `ray_hits_w` [N, 676, 3], grid order "xy" (x fastest), yaw-aligned -- what `mdp.height_scan` and the reference's
    `world_height_map` (elevation/mushr_elevation_env_cfg.py:44-48) read.  The hit heights come from the scan KERNEL
    (elev_scan_kernel through wl_elev_observe: world_height_map = hit_z + offset - plane_init, +clip on a miss), the hit
    x / y are the grid points themselves (vertical rays); rays that miss the terrain report +inf like Warp's ray caster."""

    def __init__(self, batch, offset_pos):
        self._b, self._off = batch, offset_pos

    @property
    def pos_w(self):
        b = self._b
        return b.state[A.S_PX:A.S_PX + 3, : b.n].T + torch.tensor(self._off, device=b.device, dtype=torch.float32)

    @property
    def ray_hits_w(self):
        b, p = self._b, self._b.p
        n, k = b.n, A.ELEV_SCAN_N
        hmap = b.observe(torch.empty_like(b.obs))[:, 13:]     # scratch buffer: the env's own observation stays untouched
        z = torch.where(hmap >= p.obs_clip, torch.full_like(hmap, float("inf")), hmap - p.scan_offset + p.elev_z0)
        g = -0.5 * p.scan_size + p.scan_res * torch.arange(k, device=b.device, dtype=torch.float32)
        lx, ly = g.repeat(k), g.repeat_interleave(k)
        q = b.state[A.S_QW:A.S_QW + 4, :n]
        ca, sa = 1.0 - 2.0 * (q[2] * q[2] + q[3] * q[3]), 2.0 * (q[0] * q[3] + q[1] * q[2])
        inv = torch.rsqrt(ca * ca + sa * sa)
        c, s_ = (ca * inv).unsqueeze(1), (sa * inv).unsqueeze(1)
        x = b.state[A.S_PX, :n].unsqueeze(1) + c * lx - s_ * ly
        y = b.state[A.S_PY, :n].unsqueeze(1) + s_ * lx + c * ly
        return torch.stack([x, y, z], dim=-1)


class RayCasterView:
    def __init__(self, batch, cfg):
        self.cfg = cfg
        self.data = RayCasterData(batch, tuple(getattr(cfg, "offset_pos", (0.0, 0.0, 0.0))))
        self.num_instances = batch.n


class CameraData:
    """`sensor.data` of the robot's camera (isaaclab TiledCamera, un-vendored) as far as the reference's observation functions
    read it: `output["distance_to_image_plane"]` [N, 60, 80, 1] (mdp_sensors/observations.py:89-95 `camera_data_depth` /
    `raycast_depth`), rendered on access by the depth ray-cast kernel (wl_visual_depth) against the task's terrain -- the
    heightfield of the elevation task, the z = 0 plane of the others -- and clipped at the camera's far plane."""

    def __init__(self, batch, cfg):
        self._b, self._cam, self._cfg = batch, None, cfg
        # ((batch.step_count, batch.pose_epoch) it was rendered at, image): one render per env.step(), however often read --
        # and a new one after anything that moved cars without a step (reset(), the masked in-step resets of custom
        # terminations, a plugin's write_root_pose_to_sim: they bump pose_epoch)
        self._cached = (None, None)
        self.caching = True              # the env's constructor switches it off around its shape probe of the custom terms
        clip = getattr(getattr(cfg, "spawn", None), "clipping_range", None) or (0.01, 100.0)
        self.far = float(clip[1])
        self.beyond = {"max": None, "zero": 0.0, "none": float("inf")}[getattr(cfg, "depth_clipping_behavior", "max")]

    def _camera(self):
        if self._cam is None:
            from ..core import DepthCamera
            b = self._b
            if getattr(b, "camera", None) is not None:      # visual-depth task: the batch's own camera (its terrain, its pyramid)
                self._cam = b.camera
                return self._cam
            if hasattr(b, "hf"):            # elevation task: its own heightfield (already on the device: shared, not re-quantised)
                hf = b.hf
            else:                           # flat ground: any grid at z = 0 (beyond it the outside plane is z = 0 as well)
                hf = (torch.zeros(3, 3, dtype=torch.float32, device=b.device), -1.0, -1.0, 1.0)
            self._cam = DepthCamera(hf, b.device, b.p if isinstance(b.p, A.WlVisualParams) else self._params_from_cfg())
        return self._cam

    def _params_from_cfg(self):
        """tasks whose batch carries no camera parameters (drift, elevation): intrinsics and mounting pose from the sensor's cfg"""
        from ..params import visual_params
        p, c = visual_params(), self._cfg
        sp = getattr(c, "spawn", None)
        if sp is not None and getattr(sp, "focal_length", None):
            p.fx = c.width * sp.focal_length / sp.horizontal_aperture
            p.fy = c.height * sp.focal_length / sp.vertical_aperture
            p.cx, p.cy = c.width / 2, c.height / 2
        if getattr(c, "body_pos", None) is not None:
            p.cam_pos[0], p.cam_pos[1], p.cam_pos[2] = c.body_pos
        return p

    @property
    def output(self):
        return _CameraOutputs(self)

    def rgb(self):
        """`output["rgb"]` [N, 60, 80, 3] uint8 (mdp_sensors/observations.py:60-62 `camera_data_rgb`): the designed camera's view of the
        black / white traversability plane, all 60 rows, no augmentation -- each pixel one ray against the z = 0 plane with a map lookup
        (white 255 on a traversable cell, black 0 off it or off the map, grey 127 where the ray misses the plane), the model the fused
        observation kernel evaluates for the lower 40 rows (csrc/wl_visual.hip, oracle/visual_step.py::camera).  Plain torch: this is
        the un-wired debugging / logging path, not env.step()."""
        b = self._b
        if not hasattr(b, "trav_map"):
            raise KeyError("camera data type 'rgb' needs the visual task's traversability map")
        n, p = b.n, b.p
        q = b.state[A.S_QW:A.S_QW + 4, :n]
        w, x, y, z = q[0], q[1], q[2], q[3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                         2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).view(n, 3, 3)
        cam = torch.tensor(list(p.cam_pos), dtype=torch.float32, device=b.device)
        o = b.state[A.S_PX:A.S_PX + 3, :n].T + R @ cam
        rows = torch.arange(60, dtype=torch.float32, device=b.device)
        cols = torch.arange(80, dtype=torch.float32, device=b.device)
        dy = -((cols + 0.5 - p.cx) / p.fx)
        dz = -((rows + 0.5 - p.cy) / p.fy)
        db = torch.stack([torch.ones(60, 80, device=b.device), dy[None, :].expand(60, 80), dz[:, None].expand(60, 80)], -1).view(-1, 3)
        dw = torch.einsum("nij,pj->npi", R, db)
        hit = dw[..., 2] < -1e-6
        t = torch.where(hit, -o[:, None, 2] / torch.where(hit, dw[..., 2], torch.full_like(dw[..., 2], -1.0)), torch.zeros_like(dw[..., 2]))
        hx, hy = o[:, None, 0] + t * dw[..., 0], o[:, None, 1] + t * dw[..., 1]
        m = b.trav_map
        nr, nc = m.shape
        rs, cs = float(b._map.row_spacing), float(b._map.col_spacing)
        on_map = hit & (hx.abs() <= nr * rs / 2) & (hy.abs() <= nc * cs / 2)
        # the reward terms' lookup (visual/utils/traversability_utils.py:68-88): (x + width / 2 + spacing / 2) / spacing truncated towards
        # zero, clamped, map[y, x]
        ix = torch.clamp(((hx + (nr * rs / 2 + rs / 2)) / rs).long(), 0, nr - 1)
        iy = torch.clamp(((hy + (nc * cs / 2 + cs / 2)) / cs).long(), 0, nc - 1)
        white = on_map & m.bool()[iy, ix]
        grey = torch.where(hit, torch.where(white, 255, 0), 127).to(torch.uint8)
        return grey.view(n, 60, 80, 1).expand(n, 60, 80, 3).contiguous()


class _CameraOutputs:
    def __init__(self, data):
        self._d = data

    def __getitem__(self, key):
        if key == "rgb":
            return self._d.rgb()
        if key != "distance_to_image_plane":
            raise KeyError(f"camera data type {key!r} is not rendered here (the fused observation carries the grey image)")
        d = self._d
        stamp = (getattr(d._b, "step_count", None), getattr(d._b, "pose_epoch", None))
        if not d.caching or None in stamp:
            img = d._camera().render(d._b, d.far)
        else:
            if d._cached[0] != stamp:       # the render is cached per env step; what lies beyond the far plane is applied per read
                d._cached = (stamp, d._camera().render(d._b, d.far))
            img = d._cached[1]
        if d.beyond is not None:
            img = torch.where(img >= d.far, torch.full_like(img, d.beyond), img)
        return img.unsqueeze(-1)

    def keys(self):
        return ["distance_to_image_plane"]


class CameraView:
    def __init__(self, batch, cfg):
        self.cfg = cfg
        self.data = CameraData(batch, cfg)
        self.num_instances = batch.n


class SceneView:
    def __init__(self, batch, cfg=None, task: str = "drift"):
        self._b = batch
        self.cfg = cfg
        self.num_envs = batch.n
        self.env_origins = torch.zeros(batch.n, 3, device=batch.device)  # env_spacing = 0 (mushr_drift_env_cfg.py:373)
        names = getattr(getattr(cfg, "robot", None), "joint_names", None) or MUSHR_JOINT_NAMES
        self.articulations = {"robot": ArticulationView(batch, names)}
        self.sensors = {}
        if task == "elevation" and getattr(cfg, "height_scanner", None) is not None:
            self.sensors["height_scanner"] = RayCasterView(batch, cfg.height_scanner)
        if getattr(cfg, "camera", None) is not None:
            self.sensors["camera"] = CameraView(batch, cfg.camera)
        self.terrain = getattr(cfg, "terrain", None)

    def __getitem__(self, key):
        if key in self.articulations:
            return self.articulations[key]
        if key in self.sensors:
            return self.sensors[key]
        if key == "terrain":
            return self.terrain
        raise KeyError(key)

    def keys(self):
        return list(self.articulations) + list(self.sensors) + ["terrain"]

"""The DEVICE physics headers (wheeledlab_amd/csrc/wl_vehicle.h, wl_heightfield.h) compiled for the host through a
stand-in <hip/hip_runtime.h> (tests/host_sim/) and held against the numpy oracle (oracle/vehicle.py).

Why: the kernels integrate the vehicle in the BODY frame (shared wheel kinematics, no tangent frames, torques summed in
the body frame, body-rate quaternion update) while the executable spec keeps the textbook world-frame form.  The two
are the same model; this test pins that on CPU, every round, for flat ground and for the heightfield, in both lane
variants -- the GPU parity tests (tests/test_gpu_*_parity.py) then check the compiled kernels.  Test infrastructure
only: nothing here is a product path (the product has no CPU path)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import elev_step as OE
from oracle import heightfield as OH
from oracle import params as OP
from oracle import vehicle as OV
from oracle.mathlib import F, f32
from wheeledlab_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get("WL_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")   # g++ has no ext_vector_type (wl_heightfield.h)


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    if not (os.path.exists(CLANG) or shutil.which(CLANG)):
        pytest.skip("no clang++ to build the host simulation")
    out = tmp_path_factory.mktemp("host_sim") / "libwl_host_sim.so"
    # -ffp-contract=off: the kernels contract to fma (-ffp-contract=fast); without it the host build rounds every
    # product -- a second, independent rounding pattern of the same expressions, which is what a parity bound wants
    subprocess.run([CLANG, "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                    "-I", os.path.join(ROOT, "tests", "host_sim", "hip_stub"),
                    "-I", os.path.join(ROOT, "wheeledlab_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host_sim", "vehicle_host.cpp"), "-o", str(out)], check=True)
    lib = C.CDLL(str(out))
    lib.hs_vehicle_integrate.restype = None
    return lib


def _vp_struct(vp):
    s = _abi.WlVehicleParams()
    for name, _ in s._fields_:
        setattr(s, name, getattr(vp, name))
    return s


def _states(n, seed, z0=0.0, hf=None):
    """plausible driving states: on the ground (suspension near equilibrium), some airborne, some sliding"""
    rng = np.random.RandomState(seed)
    yaw = rng.uniform(-np.pi, np.pi, n)
    roll, pitch = rng.normal(0, 0.04, n), rng.normal(0, 0.04, n)
    cr, sr, cp, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    q = f32(np.stack([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                      cr * cp * sy - sr * sp * cy], -1))
    x = f32(np.stack([rng.uniform(-15, 15, n), rng.uniform(-15, 15, n), np.zeros(n)], -1))
    if hf is not None:
        zg, _, _ = OH.sample(hf[0], hf[1], hf[2], hf[3], x[:, 0], x[:, 1])
        x[:, 2] = zg
    x[:, 2] += F(z0) + f32(rng.normal(0, 0.003, n))
    x[::7, 2] += F(0.05)                                   # airborne
    speed = rng.uniform(0, 3.5, n)
    slip = rng.normal(0, 0.5, n)
    vbx, vby = speed * np.cos(slip), speed * np.sin(slip)
    v = f32(np.stack([vbx * np.cos(yaw) - vby * np.sin(yaw), vbx * np.sin(yaw) + vby * np.cos(yaw), rng.normal(0, 0.05, n)], -1))
    wb = f32(np.stack([rng.normal(0, 0.3, n), rng.normal(0, 0.3, n), rng.normal(0, 2.0, n)], -1))
    wheel = f32(np.stack([speed / 0.05 * rng.uniform(0.7, 1.6, n) for _ in range(4)], -1))
    wheel[::5] = 0                                         # locked wheels
    steer = f32(np.stack([rng.uniform(-0.5, 0.5, n), rng.normal(0, 1.0, n)], -1))
    steer_target = f32(np.tan(rng.uniform(-0.488, 0.488, n)))
    wheel_target = f32(np.stack([rng.uniform(0, 60, n)] * 2 + [rng.uniform(0, 60, n)] * 2, -1))
    mass = f32(rng.uniform(3.3, 3.5, n))
    mu_s = f32(rng.uniform(0.3, 0.5, n))
    mu_d = f32(mu_s * rng.uniform(0.7, 1.0, n))
    damp = f32(rng.uniform(10, 50, n))
    return x, q, v, wb, wheel, steer, steer_target, wheel_target, mass, mu_s, mu_d, damp


def _oracle(vp, sim_dt, decimation, st, ground):
    x, q, v, wb, wheel, steer, steer_target, wheel_target, mass, mu_s, mu_d, damp = [a.copy() for a in st]
    th, om = steer[:, 0].copy(), steer[:, 1].copy()
    h = F(sim_dt) / F(vp.substeps)
    for _ in range(decimation * vp.substeps):
        x, q, v, wb, wheel, th, om = OV.substep(x, q, v, wb, wheel, th, om, steer_target, wheel_target, mass, mu_s, mu_d,
                                                damp, vp, h, ground)
    return x, q, v, wb, wheel, np.stack([th, om], -1)


def _host(lib, vp, sim_dt, decimation, st, hf_struct, unroll):
    x, q, v, wb, wheel, steer, steer_target, wheel_target, mass, mu_s, mu_d, damp = [np.ascontiguousarray(a.copy()) for a in st]
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    vs = _vp_struct(vp)
    lib.hs_vehicle_integrate(C.byref(vs), C.c_float(sim_dt), C.c_int(decimation), C.c_int(x.shape[0]), ptr(x), ptr(q), ptr(v),
                             ptr(wb), ptr(wheel), ptr(steer), ptr(steer_target), ptr(wheel_target), ptr(mass), ptr(mu_s),
                             ptr(mu_d), ptr(damp), C.byref(hf_struct) if hf_struct is not None else None, C.c_int(unroll))
    return x, q, v, wb, wheel, steer


def _compare(got, want, tol):
    names = ("x", "q", "v", "wb", "wheel", "steer")
    for name, g, w in zip(names, got, want):
        scale = 1.0 + np.abs(w)
        err = np.abs(g - w) / scale
        assert np.isfinite(g).all(), name
        assert err.max() < tol, (name, float(err.max()), int(np.argmax(err.max(axis=-1))))


@pytest.mark.parametrize("unroll", [1, 0])
@pytest.mark.parametrize("drive", [0, 1])
def test_body_frame_substeps_equal_the_world_frame_spec_on_flat_ground(hostlib, drive, unroll):
    """drift (RWD, dt 0.005 x 4) and the 4WD variants: one env.step() of sub-steps, 4096 states"""
    vp = OP.mushr_vehicle(drive=drive, motor_limit=0.5 if drive == 0 else 0.25)
    st = _states(4096, seed=3 + drive, z0=0.06 - 0.0028)
    want = _oracle(vp, 0.005, 4, st, OV.flat_ground)
    got = _host(hostlib, vp, 0.005, 4, st, None, unroll)
    _compare(got, want, 5e-5)


@pytest.mark.parametrize("unroll", [1, 0])
@pytest.mark.parametrize("h_ms", [10, 20])
def test_implicit_substeps_equal_the_spec_on_flat_ground(hostlib, h_ms, unroll):
    """the linearly implicit integrator (vp.implicit = 1): the visual task's 10 x 20 ms and the elevation task's 10 x 10 ms on the
    plane, 4WD at mu up to 2 -- the device's body-frame LDL^T against the oracle's, 4096 states"""
    vp = OP.mushr_vehicle(drive=1, motor_limit=0.25, ground_mu=(2.0, 2.0), implicit=1)
    st = list(_states(4096, seed=21 + h_ms, z0=0.06 - 0.0028))
    st[9], st[10], st[11] = f32(np.full(4096, 1.0)), f32(np.full(4096, 1.0)), f32(np.full(4096, 1000.0))   # wheel mu 1 x ground 2; servo damping
    st[9][::3] = F(0.4)
    want = _oracle(vp, h_ms * 1e-3, 10, st, OV.flat_ground)
    got = _host(hostlib, vp, h_ms * 1e-3, 10, st, None, unroll)
    # every seventh state starts airborne and touches down inside the step: a wheel that makes contact a sub-step earlier in one
    # arithmetic than in the other is a different branch for that sub-step (measured: max 4e-5 at 10 ms, 3e-4 at 20 ms)
    per_env = np.max([np.abs(g - w).reshape(len(g), -1).max(-1) / (1 + np.abs(w).reshape(len(w), -1).max(-1))
                      for g, w in zip(got, want)], axis=0)
    assert np.isfinite(per_env).all()
    assert np.median(per_env) < 5e-6 and (per_env > 5e-5).mean() < 0.01 and per_env.max() < 2e-3, (
        float(np.median(per_env)), float((per_env > 5e-5).mean()), float(per_env.max()))


def test_tilted_cars_take_the_isotropic_bound_in_both_statements(hostlib):
    """cars in EVERY orientation just above the plane (on their side, on their roof: R[2, 2] < 0.75 switches the implicit update to its
    isotropic over-estimate of the damping matrix) and hard landings that reach the normal-force cap: the device code against the spec"""
    vp = OP.mushr_vehicle(drive=1, motor_limit=0.25, ground_mu=(2.0, 2.0), implicit=1)
    n = 2048
    st = list(_states(n, seed=77, z0=0.06 - 0.0028))
    rng = np.random.RandomState(5)
    q = rng.normal(size=(n, 4))
    st[1] = f32(q / np.linalg.norm(q, axis=1, keepdims=True))
    st[0][:, 2] = f32(rng.uniform(0.0, 0.15, n))
    st[2][:, 2] = f32(rng.uniform(-3.0, 0.5, n))                # some slam into the ground: k pen - c v_n beyond susp_fmax
    st[9], st[10], st[11] = f32(np.full(n, 1.0)), f32(np.full(n, 1.0)), f32(np.full(n, 1000.0))
    R22 = 1.0 - 2.0 * (st[1][:, 1] ** 2 + st[1][:, 2] ** 2)
    assert 0.3 < (R22 < 0.75).mean() < 0.95
    want = _oracle(vp, 0.02, 3, st, OV.flat_ground)
    got = _host(hostlib, vp, 0.02, 3, st, None, 1)
    per_env = np.max([np.abs(g - w).reshape(len(g), -1).max(-1) / (1 + np.abs(w).reshape(len(w), -1).max(-1))
                      for g, w in zip(got, want)], axis=0)
    assert np.isfinite(per_env).all()
    # (tumbling cars make and break contact every sub-step and cross the 0.75 threshold: the branchy states get the loose bound)
    assert np.median(per_env) < 1e-5 and (per_env > 1e-3).mean() < 0.02, (float(np.median(per_env)), float((per_env > 1e-3).mean()))


def test_body_frame_substeps_equal_the_spec_on_the_heightfield(hostlib):
    """elevation: 4WD, ONE linearly implicit sub-step per sim.dt = 10 ms, bilinear heightfield with per-wheel normals; one control
    step of 10"""
    p = OE.elev_params()
    hf = OH.make_terrain()
    st = _states(2048, seed=11, z0=0.06 - 0.0028, hf=hf)
    want = _oracle(p.vehicle, p.sim_dt, p.decimation, st, OE.ground_fn(hf))
    from tests.depth_cases import hf_struct
    hs, _keep = hf_struct(hf)
    got = _host(hostlib, p.vehicle, p.sim_dt, p.decimation, st, hs, 1)
    assert p.vehicle.implicit == 1 and p.vehicle.substeps == 1
    # 10 sub-steps with contact make-or-break: a wheel that touches down in one build and not yet in the other puts the
    # env on a different branch for a sub-step; those (a handful) are held to a loose bound, the rest to fp32 noise
    err = max(float((np.abs(g - w) / (1 + np.abs(w))).max()) for g, w in zip(got, want))
    per_env = np.max([np.abs(g - w).reshape(len(g), -1).max(-1) / (1 + np.abs(w).reshape(len(w), -1).max(-1))
                      for g, w in zip(got, want)], axis=0)
    assert np.isfinite(per_env).all()
    assert (per_env > 2e-4).mean() < 0.02, float((per_env > 2e-4).mean())
    assert err < 0.5


def test_long_horizon_drift_rollout_stays_close_to_the_spec(hostlib):
    """40 control steps (160 sub-steps) with changing targets: no drift apart beyond chaotic amplification of rounding"""
    vp = OP.mushr_vehicle(drive=0)
    st = list(_states(512, seed=5, z0=0.06 - 0.0028))
    rng = np.random.RandomState(0)
    a = [np.ascontiguousarray(x.copy()) for x in st]
    b = [x.copy() for x in st]
    for k in range(40):
        tgt = f32(np.tan(rng.uniform(-0.488, 0.488, 512)))
        wt = f32(np.repeat(rng.uniform(0, 60, (512, 1)), 4, 1))
        wt[:, 2:] = 0
        a[6], a[7], b[6], b[7] = tgt, wt, tgt, wt
        ga = _host(hostlib, vp, 0.005, 4, a, None, 1)
        gb = _oracle(vp, 0.005, 4, b, OV.flat_ground)
        a[:6] = list(ga)
        b[:6] = list(gb)
    d = np.abs(a[2] - b[2]).max(-1)   # velocity difference per env
    assert np.median(d) < 1e-3 and (d > 5e-2).mean() < 0.05, (float(np.median(d)), float((d > 5e-2).mean()))


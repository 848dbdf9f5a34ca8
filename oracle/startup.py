"""Startup-mode events (domain randomisation applied once per env), numpy.  TEST INFRASTRUCTURE (see oracle/__init__).
Restates the semantics of IsaacLab's randomize_rigid_body_material (num_buckets materials per run, one bucket per
env, make_consistent: mu_d <= mu_s), randomize_actuator_gains ("abs") and randomize_rigid_body_mass ("add") as the
reference configures them (mushr_drift_env_cfg.py:98-119,145-154; elevation cfg :387-407).  IsaacLab itself is
un-vendored: parity unpinned; what is pinned is the product's kernel (wheeledlab_amd/csrc/wl_startup.hip) against this.

Keyed like every other draw of the path: Philox counter (GLOBAL env id, 0, stream 8) -> (bucket, damping, mass) and
(bucket id, 0, stream 9) -> the bucket's (mu_s, mu_d): shards of a multi-GPU run hold the rows of the one big batch."""
import numpy as np

from . import philox
from .mathlib import F

S_STARTUP, S_STARTUP_BUCKET, S_STARTUP_WHEELS = 8, 9, 10
DRIFT = dict(wheel_mu_s=(0.3, 0.5), wheel_mu_d=(0.3, 0.5), mu_buckets=20, mu_consistent=True, damping=(10.0, 50.0),
             chassis_mass=3.0, mass_add=(0.3, 0.5))
ELEV = dict(wheel_mu_s=(2.0, 2.0), wheel_mu_d=(1.0, 1.0), mu_buckets=5, mu_consistent=False, damping=(1000.0, 1000.0),
            chassis_mass=3.0, mass_add=(0.2, 0.5))


def _lerp(r, u):
    return (F(r[0]) + u * (F(r[1]) - F(r[0]))).astype(F)   # the kernel's fmaf differs by <= 1 ulp


def draw(n, seed, env_offset=0, *, wheel_mu_s, wheel_mu_d, mu_buckets, mu_consistent, damping, chassis_mass, mass_add,
         wheel_mass=(0.0, 0.0)):
    """-> mu_s, mu_d, damp, mass, bucket: float32 [n] each (bucket int32) for global envs env_offset .. env_offset + n"""
    gid = np.arange(n, dtype=np.uint64) + np.uint64(env_offset)
    u = philox.uniform4(gid, 0, S_STARTUP, seed)
    nb = max(int(mu_buckets), 1)
    bucket = np.minimum((u[0] * F(nb)).astype(np.int32), nb - 1)
    m = philox.uniform4(bucket.astype(np.uint64), 0, S_STARTUP_BUCKET, seed)
    mu_s = _lerp(wheel_mu_s, m[0])
    mu_d = _lerp(wheel_mu_d, m[1])
    if mu_consistent:
        mu_d = np.minimum(mu_d, mu_s)
    mass = (F(chassis_mass) + _lerp(mass_add, u[2])).astype(F)
    if wheel_mass[1] > 0:      # the four wheel links' masses ("abs", visual cfg :289-298) add to the vehicle's
        w = philox.uniform4(gid, 0, S_STARTUP_WHEELS, seed)
        mass = (mass + ((_lerp(wheel_mass, w[0]) + _lerp(wheel_mass, w[1])) + (_lerp(wheel_mass, w[2]) + _lerp(wheel_mass, w[3])))).astype(F)
    return mu_s, mu_d, _lerp(damping, u[1]), mass, bucket

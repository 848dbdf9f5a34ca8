// wl_startup.hip -- startup-mode events (domain randomisation, once per env) as a keyed kernel: the draws of env e depend
// on (seed, env_offset + e) only, so env shards of a multi-GPU run hold exactly the parameter sets of the one big batch.
// Reference: mushr_drift_env_cfg.py:98-119,145-154 (IsaacLab randomize_rigid_body_material / randomize_actuator_gains /
// randomize_rigid_body_mass, un-vendored).
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_kernel_common.h"
#include "wl_rng.h"

namespace {

WL_DEV float lerp_range(const float r[2], float u) { return fmaf(u, r[1] - r[0], r[0]); }

__global__ void __launch_bounds__(kBlock) startup_kernel(const WlStartupParams su, const WlEnvBuffers b, const uint64_t seed) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= b.n_envs) return;
    const Rows S = make_rows(b.state, b.stride);
    float mu_s, mu_d, damp, mass;
    if (su.randomize) {
        const F4 u = philox_uniform4((uint32_t)(b.env_offset + e), 0, WL_RS_STARTUP, seed);
        const int nb = max(su.mu_buckets, 1);
        const int bucket = min((int)(u.x * (float)nb), nb - 1);
        const F4 m = philox_uniform4((uint32_t)bucket, 0, WL_RS_STARTUP_BUCKET, seed);   // the bucket's material
        mu_s = lerp_range(su.wheel_mu_s, m.x);
        mu_d = lerp_range(su.wheel_mu_d, m.y);
        if (su.mu_consistent) mu_d = fminf(mu_d, mu_s);
        damp = lerp_range(su.damping, u.y);
        mass = su.chassis_mass + lerp_range(su.mass_add, u.z);
        if (su.wheel_mass[1] > 0.f) {    // randomize_rigid_body_mass on the four wheel links (visual cfg :289-298): link masses add up
            const F4 w = philox_uniform4((uint32_t)(b.env_offset + e), 0, WL_RS_STARTUP_WHEELS, seed);
            mass += (lerp_range(su.wheel_mass, w.x) + lerp_range(su.wheel_mass, w.y)) + (lerp_range(su.wheel_mass, w.z) + lerp_range(su.wheel_mass, w.w));
        }
    } else {
        mu_s = 0.5f * (su.wheel_mu_s[0] + su.wheel_mu_s[1]);
        mu_d = fminf(0.5f * (su.wheel_mu_d[0] + su.wheel_mu_d[1]), mu_s);
        damp = 0.5f * (su.damping[0] + su.damping[1]);
        mass = su.chassis_mass + 0.5f * (su.mass_add[0] + su.mass_add[1]) + 2.f * (su.wheel_mass[0] + su.wheel_mass[1]);
    }
    S.st(WL_S_MU_S, e, mu_s);
    S.st(WL_S_MU_D, e, mu_d);
    S.st(WL_S_DAMP, e, damp);
    S.st(WL_S_MASS, e, mass);
    S.st(WL_S_QW, e, 1.f);
}

}  // namespace

extern "C" int wl_startup_randomize(const WlStartupParams* su, const WlEnvBuffers* b, uint64_t seed, void* stream) {
    if (!su || !b || !b->state || b->n_envs <= 0 || b->stride < b->n_envs) return WL_EINVAL;
    if (b->stride % 64 != 0 || ((uintptr_t)b->state & 15u)) return WL_EALIGN;
    if (b->stride * 4 * WL_S_COUNT > 0x7fffffffLL) return WL_EINVAL;
    clear_error();
    startup_kernel<<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*su, *b, seed);
    return launch_status();
}

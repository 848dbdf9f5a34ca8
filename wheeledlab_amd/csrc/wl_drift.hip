// wl_drift.hip -- fused drift-task env.step() for gfx950 + the C ABI around it (include/wheeledlab_amd.h).
//
// One launch = one env.step() for all envs: action term -> decimation x (actuators + rigid body + tyres) ->
// terminations -> rewards -> in-kernel reset -> interval pushes -> observation.  One lane owns one env; the SoA
// state matrix is read once (coalesced 256 B per wavefront per row) and written once per env-step, all physics
// sub-steps stay in VGPRs.  The [n][14] observation is transposed through LDS so the stores are contiguous per
// wavefront; episode metrics are folded with LDS atomics and leave the block as <=16 global atomics.
#include "wl_drift_env.h"

namespace {

// The first eight arguments (13 dwords) repeat what the state loads and the random draws need -- pointers, sizes, seed,
// step: built with -amdgpu-kernarg-preload-count they arrive in SGPRs WITH the wavefront, so the state-row loads
// are issued at once, next to the vector loads of the parameter block, instead of after a first kernarg round trip.
constexpr int kHotArgBytes = 56;   // state 0, episode_len 8, actions 16, stride 24, n_envs 28, env_offset 32, seed 40, step 48

// LANES = 1: lane per env (throughput form).  LANES = 4: quad per env, one wheel per lane (latency form, small n).
// UNROLL (lane form only): the four wheels inlined and interleaved by the scheduler (more registers, more ILP: better
// while the SIMDs hold few wavefronts) or fenced one after the other (fewer registers -> one more wavefront per SIMD).
// DRIVE (lane form): the drive train compiled in (wl_vehicle.h); -1: the quad form decides per lane at run time.
// QB (quad form): threads per block.  The wavefronts of a block share a CU; at 4096 envs (256 wavefronts) blocks of 256
// threads put four of them on each of 64 CUs and leave 192 CUs idle: 6.63 us per launch against 6.27 with 128 threads and
// 6.37 with 64 (1024 envs: 6.38 / 6.03 / 5.87; 16 384 envs: 7.61 / 8.79 / 8.39) -- launch_step picks by env count.
template <int LANES, class Ground, bool UNROLL = true, int DRIVE = -1, int QB = kBlock, bool STREAM = false>
__global__ void __launch_bounds__(kBlock, LANES == 4 ? WL_MIN_WAVES : (UNROLL ? WL_LANE_WAVES : WL_LOWREG_WAVES)) drift_step_kernel(float* __restrict__ state, int32_t* __restrict__ episode_len,
                                                            const float2* __restrict__ actions, const int stride,
                                                            const int n_envs, const int env_offset, const uint64_t seed,
                                                            const uint64_t step, const WlDriftParams p_arg,
                                                            const VehDerived vd_arg, const WlEnvBuffers b_arg,
                                                            const float* __restrict__ noise, const WlStepOut out,
                                                            const Ground ground, const MetricSlots slots) {
    constexpr int kEnvs = QB / LANES;   // envs per block (QB threads per block in both forms)
    // quad (latency) form: parameters by one batch of vector loads from the kernarg segment (p_arg right behind the
    // hot arguments, vd_arg behind it); lane (throughput) form: the compiler's scalar loads -- latency is hidden by
    // occupancy there and the VGPRs are needed for the env
    WlDriftParams p = p_arg;
    VehDerived vd = vd_arg;
    WlEnvBuffers b = b_arg;          // the hot fields from the preloaded arguments, the rest when the kernarg block lands
    b.state = state;
    b.episode_len = episode_len;
    b.stride = stride;
    b.n_envs = n_envs;
    b.env_offset = env_offset;
    // lane form only: per-WAVEFRONT obs transposing tile and metric accumulators (no block barrier anywhere)
    __shared__ float tile[LANES == 1 ? kEnvs * kObsPad : 1];
    __shared__ float wave_metrics[LANES == 1 ? (QB / 64) * WL_M_COUNT : 1];
    const int wave = threadIdx.x >> 6;
    const int wid = LANES == 1 ? 0 : (threadIdx.x & 3);   // this lane's wheel (quad form)
    const bool lead = LANES == 1 || wid == 0;       // the lane that writes the env's shared rows / outputs
    const int e = blockIdx.x * kEnvs + threadIdx.x / LANES;
    const MetricSink<LANES> ms{wave_metrics + (LANES == 1 ? wave * WL_M_COUNT : 0), metric_shard(b, slots.cur)};
    ms.open();
    // STREAM: the launcher sets it when the state matrix alone outgrows the 256 MB Infinity Cache (below that the rows written now
    // are the next step's cache hits: at 1 M envs the non-temporal stores cost 4 %, at 4 M they gain 4 %)
    constexpr bool kStreaming = STREAM;
    if (e < b.n_envs) {
        const Rows S = make_rows(b.state, b.stride, kStreaming);
        EnvConst ec;
        DriftRows r;
        const float2 a = actions[e];
        float dr_mass, dr_mu_s, dr_mu_d, dr_damp;
        dr_mass = S.ld(WL_S_MASS, e), dr_mu_s = S.ld(WL_S_MU_S, e), dr_mu_d = S.ld(WL_S_MU_D, e), dr_damp = S.ld(WL_S_DAMP, e);
        load_rows<LANES>(S, b, p, e, wid, r);
        const uint32_t gid = (uint32_t)(b.env_offset + e);
        float* tile_w = tile + (LANES == 1 ? wave * 64 * kObsPad : 0);
        if constexpr (LANES == 4) {
            // The state requests go out FIRST (they need only the preloaded arguments), the parameter block right behind
            // them.  Everything that follows runs in their shadow -- and is kept there by a basic-block boundary: within
            // one block the instruction selector, to shorten live ranges, put the ~200 instructions of the draws (which
            // need no loaded value) IN FRONT of the state requests, so that the launch waited for the parameter block,
            // drew, and only then asked for its state: two memory round trips in series.  The memory clobber keeps the
            // loads from being sunk across the boundary, the always-true test of an opaque scalar cannot be folded.
            uint32_t kw[(sizeof(WlDriftParams) + sizeof(VehDerived)) / 4];
            kernarg_vector_words(kHotArgBytes, kw);
            int go = 1;
            asm volatile("" : "+s"(go) : : "memory");
            if (go) {
                // the step's random draws need only (seed, gid, step) -- preloaded arguments: one Philox block and two
                // Box-Muller pairs per lane before the first instruction that waits for anything
                StepDraws pre = draw_step_raw(gid, step, seed, wid);
                kernarg_words_landed(kw);            // the first instruction that waits for the parameter block
                __builtin_memcpy(&p, kw, sizeof(WlDriftParams));
                __builtin_memcpy(&vd, kw + sizeof(WlDriftParams) / 4, sizeof(VehDerived));
                keep_scalar_fields(p, p_arg);        // v_readfirstlane of the integer parameters
                vd.n_sub = uniform_i32(vd.n_sub);
                ref_pose_request(pre, b.ref_poses, p.num_ref_points);   // consumed only inside the (rare) reset branch: never waited for here
                env_const_rows(ec, p.vehicle, vd, dr_mass, dr_mu_s, dr_mu_d, dr_damp);
                drift_env_step<LANES>(p, b, vd, ground, S, ec, r, a, noise, out, e, wid, lead, gid, seed, step, tile_w, ms, nullptr, &pre);
            }
        } else {
            env_const_rows(ec, p.vehicle, vd, dr_mass, dr_mu_s, dr_mu_d, dr_damp);
            drift_env_step<LANES, UNROLL, DRIVE>(p, b, vd, ground, S, ec, r, a, noise, out, e, wid, lead, gid, seed, step, tile_w, ms);
        }
        store_rows<LANES>(S, b, p, e, wid, lead, r);
    }
    if (b.metrics_slots > 1) clear_metric_slot(b, slots.next);   // the slot the NEXT launch will use
    if constexpr (LANES == 1) {
        const int wave_env0 = blockIdx.x * kEnvs + wave * 64;
        if (wave_env0 < b.n_envs) flush_obs_wave<kStreaming>(tile + wave * 64 * kObsPad, out.obs, wave_env0, b.n_envs);
        ms.close();
    }
}

// Persistent rollout (quad form): K consecutive env.step()s in ONE launch with pre-staged actions [K][n][2].  The env's
// rows stay in registers across steps -- per step only the action is read and obs / reward / flags are written -- so the
// launch boundary, the state round trip through L2 and its address arithmetic are paid once per rollout.  Episode
// metrics of all K steps accumulate into ring slot (step0 % R); slot ((step0 + K) % R) is cleared for the next launch.
template <class Ground, int QB = kBlock /* threads per block, see drift_step_kernel */>
__global__ void __launch_bounds__(kBlock) drift_rollout_kernel(const WlDriftParams p_arg, const WlEnvBuffers b,
                                                               const float2* __restrict__ actions, const WlStepOut out,
                                                               const int64_t obs_step_stride, const int64_t vec_step_stride,
                                                               const int n_steps, const uint64_t seed, const uint64_t step0,
                                                               const Ground ground, const VehDerived vd_arg, const MetricSlots slots) {
    constexpr int LANES = 4, kEnvs = QB / LANES;
    WlDriftParams p = p_arg;
    VehDerived vd = vd_arg;
    pin_params_vgpr(p, vd);
    const int wid = threadIdx.x & 3;
    const bool lead = wid == 0;
    const int e = blockIdx.x * kEnvs + threadIdx.x / LANES;
    if (b.metrics_slots > 1) clear_metric_slot(b, slots.next);
    if (e >= b.n_envs) return;     // the quad form has no block-level barrier: whole quads may leave
    const MetricSink<LANES> ms{nullptr, metric_shard(b, slots.cur)};
    const Rows S = make_rows(b.state, b.stride);
    EnvConst ec;
    DriftRows r;
    load_env_const(S, p.vehicle, vd, e, ec);
    load_rows<LANES>(S, b, p, e, wid, r);
    const uint32_t gid = (uint32_t)(b.env_offset + e);
    float2 a_next = n_steps > 0 ? actions[e] : make_float2(0.f, 0.f);
    for (int k = 0; k < n_steps; ++k) {
        WlStepOut o = out;
        o.obs += k * obs_step_stride;
        o.reward += k * vec_step_stride;
        o.terminated += k * vec_step_stride;
        o.truncated += k * vec_step_stride;
        if (o.dones) o.dones += k * vec_step_stride;
        const float2 a = a_next;
        if (k + 1 < n_steps) a_next = actions[(int64_t)(k + 1) * b.n_envs + e];   // prefetch: hidden behind the physics
        const StepDraws pre = draw_step(p, b.ref_poses, gid, step0 + (uint64_t)k, seed, wid);
        drift_env_step<LANES>(p, b, vd, ground, S, ec, r, a, nullptr, o, e, wid, lead, gid, seed, step0 + (uint64_t)k, nullptr, ms,
                              nullptr, &pre);
    }
    store_rows<LANES>(S, b, p, e, wid, lead, r);
}

// ---- terms only, on caller-supplied state tensors (parity entry point) -----------------------------------
__global__ void __launch_bounds__(kBlock) drift_mdp_kernel(const WlDriftParams p, int n, int64_t stride,
                                                           const float* __restrict__ pos, const float* __restrict__ quat,
                                                           const float* __restrict__ vb_, const float* __restrict__ wb_,
                                                           const float* __restrict__ ww_, const float* __restrict__ steer,
                                                           const float* __restrict__ act, const uint8_t* __restrict__ timed_out,
                                                           float* __restrict__ terms, float* __restrict__ reward,
                                                           uint8_t* __restrict__ terminated, float* __restrict__ obs) {
    __shared__ float tile[kBlock * kObsPad];
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e < n) {
        const V3 P = v3(pos[e], pos[stride + e], pos[2 * stride + e]);
        const Quat q{quat[e], quat[stride + e], quat[2 * stride + e], quat[3 * stride + e]};
        const V3 vb = v3(vb_[e], vb_[stride + e], vb_[2 * stride + e]);
        const V3 wb = v3(wb_[e], wb_[stride + e], wb_[2 * stride + e]);
        const float wwz = ww_[2 * stride + e];
        const float sm = 0.5f * (steer[e] + steer[stride + e]);
        const bool to = timed_out ? timed_out[e] != 0 : false;
        const bool term = cart_off_track(P.x, P.y, p.straight, p.r_in, p.r_out);
        const DriftTerms tm = drift_terms(p, P, vb, wb, wwz, sm, term, to, atan2_fast(vb.y, vb.x));
        const float step_dt = p.sim_dt * (float)p.decimation;
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < WL_DR_NTERMS; ++i) {
            terms[i * stride + e] = tm.t[i];
            if (p.weight[i] != 0.f) r += tm.t[i] * p.weight[i] * step_dt;
        }
        reward[e] = r;
        terminated[e] = term ? 1 : 0;
        Noise12 zero;
#pragma unroll
        for (int k = 0; k < 12; ++k) zero.z[k] = 0.f;
        write_obs_row(&tile[threadIdx.x * kObsPad], p, P, euler_xyz_from_quat(q), vb, wb, act[e], act[stride + e], zero);
    }
    __syncthreads();
    flush_obs(tile, obs, blockIdx.x * kBlock, n);
}

__global__ void __launch_bounds__(kBlock) action_map_kernel(const WlActionParams ap, int n, const float2* __restrict__ actions,
                                                            float2* __restrict__ processed, float2* __restrict__ steer_target,
                                                            float4* __restrict__ wheel_target) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    float2 a = actions[e];
    float v, delta, st, w[4];
    process_action(ap, a.x, a.y, v, delta);
    joint_targets(ap, v, delta, st, w);
    float sl, sr;
    steer_pair(ap, delta, st, sl, sr);
    processed[e] = make_float2(v, delta);
    steer_target[e] = make_float2(sl, sr);
    wheel_target[e] = make_float4(w[0], w[1], w[2], w[3]);
}

__global__ void __launch_bounds__(kBlock) drift_reset_kernel(const WlDriftParams p, const WlEnvBuffers b,
                                                             const uint8_t* __restrict__ mask, uint64_t seed, uint64_t step) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= b.n_envs) return;
    if (mask && !mask[e]) return;
    const Rows S = make_rows(b.state, b.stride);
    const ResetDraw rd = draw_reset(p, b.ref_poses, (uint32_t)(b.env_offset + e), step, seed);
    st3(S, WL_S_PX, e, rd.pos);
    S.st(WL_S_QW, e, rd.q.w);
    S.st(WL_S_QX, e, rd.q.x);
    S.st(WL_S_QY, e, rd.q.y);
    S.st(WL_S_QZ, e, rd.q.z);
    st3(S, WL_S_VX, e, v3(0.f, 0.f, 0.f));
    st3(S, WL_S_WX, e, v3(0.f, 0.f, 0.f));
    S.st(WL_S_ACT0, e, 0.f);
    S.st(WL_S_ACT1, e, 0.f);
    S.st(WL_S_TIMER_HF, e, rd.timer_hf);
    S.st(WL_S_TIMER_LF, e, rd.timer_lf);
#pragma unroll
    for (int i = 0; i < WL_MAX_REW_TERMS; ++i) S.st(WL_S_EPSUM0 + i, e, 0.f);
    b.episode_len[e] = 0;
}

__global__ void __launch_bounds__(kBlock) drift_observe_kernel(const WlDriftParams p, const WlEnvBuffers b,
                                                               const float* __restrict__ noise, float* __restrict__ obs,
                                                               uint64_t seed, uint64_t step) {
    __shared__ float tile[kBlock * kObsPad];
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e < b.n_envs) {
        const Rows S = make_rows(b.state, b.stride);
        const Quat q{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
        const Mat3 R = mat_from_quat(q);
        const V3 vb = mul_t(R, ld3(S, WL_S_VX, e)), wb = mul_t(R, ld3(S, WL_S_WX, e));
        const Noise12 nz = obs_noise<1>(p, noise, b.stride, e, (uint32_t)(b.env_offset + e), step, seed);
        write_obs_row(&tile[threadIdx.x * kObsPad], p, ld3(S, WL_S_PX, e), euler_xyz_from_quat(q), vb, wb, S.ld(WL_S_ACT0, e),
                      S.ld(WL_S_ACT1, e), nz);
    }
    __syncthreads();
    flush_obs(tile, obs, blockIdx.x * kBlock, b.n_envs);
}

__global__ void philox_uniform_kernel(int n, uint64_t seed, uint64_t step, uint32_t stream_id, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const F4 u = philox_uniform4((uint32_t)e, step, stream_id, seed);
    out[e] = u.x;
    out[n + e] = u.y;
    out[2 * n + e] = u.z;
    out[3 * n + e] = u.w;
}

}  // namespace

extern "C" {

int wl_version(void) { return WL_ABI_VERSION; }

int wl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return WL_ENODEV;
    return n;
}

const char* wl_strerror(int code) {
    switch (code) {
        case WL_OK: return "ok";
        case WL_EINVAL: return "invalid argument";
        case WL_ELAUNCH: return "kernel launch failed";
        case WL_EALIGN: return "buffer alignment / stride violation";
        case WL_ENODEV: return "no HIP device";
        default: return "unknown error";
    }
}

// state matrices larger than this are streamed (sc1 nt stores): the Infinity Cache is 256 MB and the outputs want their share
constexpr int64_t kStreamingStateBytes = 192ll << 20;   // = 1.22 M envs

// one fused env.step() launch in the form the batch size (or WlEnvBuffers.lanes) selects
static void launch_step(const WlDriftParams* p, const WlEnvBuffers* b, const VehDerived& vd, const float2* actions, const float* noise,
                        const WlStepOut& out, uint64_t seed, uint64_t step, hipStream_t stream) {
    const MetricSlots slots = metric_slots(b, step);
#define WL_STEP_ARGS b->state, b->episode_len, actions, (int)b->stride, b->n_envs, b->env_offset, seed, step, *p, vd, *b, noise, out, FlatGround{}, slots
#ifndef WL_LANE_BLOCK
#define WL_LANE_BLOCK 256     // threads per block of the lane forms (whole wavefronts; the kernel has no block-level barrier).
                              // Round 4, us per step at 1 M / 2 M / 4 M envs: 256 threads 79.2 / 151.0 / 290.1, 128: 81.5 / 155.1 / 291.6,
                              // 64: 82.1 / 153.1 / 292.0 -- finer blocks do not even out the last round of the launch
#endif
    constexpr int LB = WL_LANE_BLOCK;
    const int grid = (b->n_envs + LB - 1) / LB;
    const bool awd = p->vehicle.drive == 1;
    // WL_FLAG_STREAM selects the streaming instantiation at any size (it is a lane form with the scalar wheel loop)
    const bool streaming = use_streaming(b, (int64_t)b->stride * 4 * WL_S_COUNT, kStreamingStateBytes);
    const bool forced_stream = (b->flags & WL_FLAG_STREAM) != 0;
    if (!forced_stream && use_quad(b)) {
        const int lanes = b->n_envs * 4;
        if (b->n_envs <= 2048) drift_step_kernel<4, FlatGround, true, -1, 64><<<(lanes + 63) / 64, 64, 0, stream>>>(WL_STEP_ARGS);
        else if (b->n_envs <= 8192) drift_step_kernel<4, FlatGround, true, -1, 128><<<(lanes + 127) / 128, 128, 0, stream>>>(WL_STEP_ARGS);
        else drift_step_kernel<4, FlatGround><<<grid_for(lanes), kBlock, 0, stream>>>(WL_STEP_ARGS);
    }
    else if (!forced_stream && use_unrolled(b)) {
        if (awd) drift_step_kernel<1, FlatGround, true, 1, LB><<<grid, LB, 0, stream>>>(WL_STEP_ARGS);
        else drift_step_kernel<1, FlatGround, true, 0, LB><<<grid, LB, 0, stream>>>(WL_STEP_ARGS);
    } else if (!streaming) {
        if (awd) drift_step_kernel<1, FlatGround, false, 1, LB><<<grid, LB, 0, stream>>>(WL_STEP_ARGS);
        else drift_step_kernel<1, FlatGround, false, 0, LB><<<grid, LB, 0, stream>>>(WL_STEP_ARGS);
    } else {
        if (awd) drift_step_kernel<1, FlatGround, false, 1, LB, true><<<grid, LB, 0, stream>>>(WL_STEP_ARGS);
        else drift_step_kernel<1, FlatGround, false, 0, LB, true><<<grid, LB, 0, stream>>>(WL_STEP_ARGS);
    }
#undef WL_STEP_ARGS
}

int wl_drift_step(const WlDriftParams* p, const WlEnvBuffers* b, const float* actions, const float* noise,
                  const WlStepOut* out, uint64_t seed, uint64_t step, void* stream) {
    int rc = check_buffers(p, b);
    if (rc != WL_OK) return rc;
    if (!actions || !out || !out->obs || !out->reward || !out->terminated || !out->truncated) return WL_EINVAL;
    clear_error();
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
    launch_step(p, b, vd, (const float2*)actions, noise, *out, seed, step, (hipStream_t)stream);
    return launch_status();
}

int wl_drift_rollout(const WlDriftParams* p, const WlEnvBuffers* b, const float* actions, const WlStepOut* out,
                     int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps, uint64_t seed, uint64_t step0,
                     void* stream) {
    int rc = check_buffers(p, b);
    if (rc != WL_OK) return rc;
    if (!actions || !out || !out->obs || !out->reward || !out->terminated || !out->truncated || n_steps < 0) return WL_EINVAL;
    clear_error();
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
    for (int k = 0; k < n_steps; ++k) {
        WlStepOut o = *out;
        o.obs += k * obs_step_stride;
        o.reward += k * vec_step_stride;
        o.terminated += k * vec_step_stride;
        o.truncated += k * vec_step_stride;
        if (o.dones) o.dones += k * vec_step_stride;
        launch_step(p, b, vd, (const float2*)(actions + (int64_t)k * b->n_envs * 2), nullptr, o, seed, step0 + (uint64_t)k, (hipStream_t)stream);
    }
    return launch_status();
}

int wl_drift_rollout_persistent(const WlDriftParams* p, const WlEnvBuffers* b, const float* actions, const WlStepOut* out,
                                int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps, uint64_t seed, uint64_t step0,
                                void* stream) {
    int rc = check_buffers(p, b);
    if (rc != WL_OK) return rc;
    if (!actions || !out || !out->obs || !out->reward || !out->terminated || !out->truncated || n_steps < 0) return WL_EINVAL;
    if (b->metrics_slots > 1 && n_steps % b->metrics_slots == 0 && n_steps > 0) return WL_EINVAL;   // ring slot aliasing
    clear_error();
#define WL_ROLLOUT_ARGS *p, *b, (const float2*)actions, *out, obs_step_stride, vec_step_stride, n_steps, seed, step0, FlatGround{}, \
                        derive_vehicle(p->vehicle, p->sim_dt, p->decimation), metric_slots(b, step0, (uint64_t)n_steps)
    const int lanes = b->n_envs * 4;
    if (b->n_envs <= 2048) drift_rollout_kernel<FlatGround, 64><<<(lanes + 63) / 64, 64, 0, (hipStream_t)stream>>>(WL_ROLLOUT_ARGS);
    else if (b->n_envs <= 8192) drift_rollout_kernel<FlatGround, 128><<<(lanes + 127) / 128, 128, 0, (hipStream_t)stream>>>(WL_ROLLOUT_ARGS);
    else drift_rollout_kernel<FlatGround><<<grid_for(lanes), kBlock, 0, (hipStream_t)stream>>>(WL_ROLLOUT_ARGS);
#undef WL_ROLLOUT_ARGS
    return launch_status();
}

int wl_drift_mdp(const WlDriftParams* p, int32_t n, int64_t stride, const float* pos, const float* quat,
                 const float* lin_vel_b, const float* ang_vel_b, const float* ang_vel_w, const float* steer_pos,
                 const float* last_action, const uint8_t* timed_out, float* terms, float* reward, uint8_t* terminated,
                 float* obs, void* stream) {
    if (!p || n <= 0 || stride < n || !pos || !quat || !lin_vel_b || !ang_vel_b || !ang_vel_w || !steer_pos ||
        !last_action || !terms || !reward || !terminated || !obs)
        return WL_EINVAL;
    clear_error();
    drift_mdp_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(*p, n, stride, pos, quat, lin_vel_b, ang_vel_b,
                                                                       ang_vel_w, steer_pos, last_action, timed_out, terms,
                                                                       reward, terminated, obs);
    return launch_status();
}

int wl_action_map(const WlActionParams* a, int32_t n, const float* actions, float* processed, float* steer_target,
                  float* wheel_target, void* stream) {
    if (!a || n <= 0 || !actions || !processed || !steer_target || !wheel_target || a->map < 0 || a->map > 2) return WL_EINVAL;
    if (((uintptr_t)wheel_target & 15u) || ((uintptr_t)actions & 7u)) return WL_EALIGN;
    clear_error();
    action_map_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(*a, n, (const float2*)actions, (float2*)processed,
                                                                        (float2*)steer_target, (float4*)wheel_target);
    return launch_status();
}

int wl_drift_reset(const WlDriftParams* p, const WlEnvBuffers* b, const uint8_t* mask, uint64_t seed, uint64_t step,
                   void* stream) {
    int rc = check_buffers(p, b);
    if (rc != WL_OK) return rc;
    clear_error();
    drift_reset_kernel<<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, *b, mask, seed, step);
    return launch_status();
}

int wl_drift_observe(const WlDriftParams* p, const WlEnvBuffers* b, const float* noise, float* obs, uint64_t seed,
                     uint64_t step, void* stream) {
    int rc = check_buffers(p, b);
    if (rc != WL_OK) return rc;
    if (!obs) return WL_EINVAL;
    clear_error();
    drift_observe_kernel<<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, *b, noise, obs, seed, step);
    return launch_status();
}

int wl_philox_uniform(int32_t n, uint64_t seed, uint64_t step, uint32_t stream_id, float* out, void* stream) {
    if (n <= 0 || !out) return WL_EINVAL;
    clear_error();
    philox_uniform_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, seed, step, stream_id, out);
    return launch_status();
}

}  // extern "C"

// wl_visual.hip -- visual task (wheeledlab_tasks/visual/mushr_visual_env_cfg.py) for gfx950.
//
// Two launches per env.step() (ONE for K steps of an open-loop rollout: visual_rollout_persistent_kernel):
//   1. visual_step_kernel (lane = env): 4WD action term -> sub-steps on the flat plane -> time_out / out_of_map ->
//      traversable_reward (byte-map gather) + forward_vel -> reset onto a random traversable cell.
//   2. visual_obs_kernel  (block = env): the 3208-dim observation.  3200 camera rays against the z = 0 plane with a
//      traversability-map lookup each; the 40 x 80 image is staged in LDS (12.8 KB) so that the contrast mean
//      (block reduction) and the separable 5x5 Gaussian blur (second LDS plane) never leave the CU; the row is
//      written once with contiguous dword stores -- 12.8 KB / env, ~98 % of the task's HBM bytes.
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_kernel_common.h"
#include "wl_drift_terms.h"
#include "wl_rng.h"
#include "wl_vehicle.h"
#include "wl_heightfield.h"

namespace {

constexpr int kImgH = WL_VIS_IMG_H - WL_VIS_CROP, kImgW = WL_VIS_IMG_W;
// threads that render one image (whole wavefronts, whole image rows per pass)
#ifndef WL_CAM_THREADS
#define WL_CAM_THREADS 256
#endif
constexpr int kCam = WL_CAM_THREADS;

// TraversabilityHashmapUtil.get_map_id (visual/utils/traversability_utils.py:83-88): float32 arithmetic, truncation
// toward zero (`.long()`), clamp to the map.
WL_DEV void map_id(const WlTravMap& m, float x, float y, int& xi, int& yi) {
    const float width = (float)m.rows * m.row_spacing, height = (float)m.cols * m.col_spacing;
    const float fx = (x + 0.5f * width + 0.5f * m.row_spacing) / m.row_spacing;
    const float fy = (y + 0.5f * height + 0.5f * m.col_spacing) / m.col_spacing;
    // float -> int conversion saturates on gfx950 (v_cvt_i32_f32), NaN -> 0: both end up clamped like torch's long()
    xi = min(max((int)fx, 0), m.rows - 1);
    yi = min(max((int)fy, 0), m.cols - 1);
}
WL_DEV bool traversable(const WlTravMap& m, float x, float y) {
    int xi, yi;
    map_id(m, x, y, xi, yi);
    return m.map[yi * m.cols + xi] != 0;   // map[y_idx, x_idx] (:78)
}
// render-path lookup: same cell function with reciprocal spacing (the camera is designed, not parity-pinned)
struct MapFast {
    float off_x, off_y, inv_rs, inv_cs, half_w, half_h;
};
WL_DEV MapFast map_fast(const WlTravMap& m) {
    const float width = (float)m.rows * m.row_spacing, height = (float)m.cols * m.col_spacing;
    return MapFast{0.5f * width + 0.5f * m.row_spacing, 0.5f * height + 0.5f * m.col_spacing, rcp(m.row_spacing),
                   rcp(m.col_spacing), 0.5f * width, 0.5f * height};
}
WL_DEV bool traversable_fast(const WlTravMap& m, const MapFast& f, float x, float y) {
    const int xi = min(max((int)((x + f.off_x) * f.inv_rs), 0), m.rows - 1);
    const int yi = min(max((int)((y + f.off_y) * f.inv_cs), 0), m.cols - 1);
    return m.map[yi * m.cols + xi] != 0;
}
// out_of_map (mushr_visual_env_cfg.py:390-398)
WL_DEV bool out_of_map(const WlTravMap& m, float x, float y) {
    const float hw = 0.5f * (float)m.rows * m.row_spacing, hh = 0.5f * (float)m.cols * m.col_spacing;
    return x > hw || x < -hw || y > hh || y < -hh;
}

struct VisReset {
    V3 pos;
    Quat q;
};
// visual/mdp/events.py:11-42 + generate_random_poses (utils/__init__.py:188-202): random traversable cell, z 0.1,
// yaw U(0, 360 deg), zero velocity
WL_DEV VisReset draw_visual_reset(const WlVisualParams& p, const WlTravMap& m, uint32_t gid, uint64_t step, uint64_t seed) {
    const F4 u = philox_uniform4(gid, step, 0, seed);
    const int k = min((int)(u.x * (float)m.n_cells), m.n_cells - 1);
    const int iy = m.cells[2 * k], ix = m.cells[2 * k + 1];
    VisReset r;
    r.pos = v3(((float)ix - (float)(m.cols / 2)) * m.row_spacing, ((float)iy - (float)(m.rows / 2)) * m.col_spacing, p.reset_z);
    float s, c;
    sincos_rev(0.5f * u.y, s, c);   // yaw = 2 pi u  ->  yaw / 2 = u / 2 revolutions
    r.q = Quat{c, 0.f, 0.f, s};
    return r;
}

// what the camera needs of an env after its step: exactly the values the step leaves in the state rows (post-reset)
struct CamPose {
    float px, py, pz, qw, qx, qy, qz, vx, vy, vz, wx, wy, wz, a0, a1;
};

// one env.step() of env e (lane form: one lane; quad form: the four lanes of a quad, wid = wheel).  Ground: the flat plane of the
// reference's task, or a heightfield (the visual-depth extension task, BASELINE config 5: wheel contacts by bilinear gathers as in
// the elevation task, reset poses lifted onto the terrain; `prop`: the env's 8 proprioceptive observation values are written
// there -- the depth image next to them comes from wl_depth.hip)
// a resetting env's new pose: the draw + (on a heightfield) the lift onto the terrain under the spawn cell
template <class Ground>
WL_DEV VisReset visual_reset_pose(const WlVisualParams& p, const WlTravMap& m, const Ground& ground, uint32_t gid, uint64_t step, uint64_t seed) {
    VisReset rd = draw_visual_reset(p, m, gid, step, seed);
    if constexpr (!Ground::kFlat) {     // z 0.1 above the plane -> 0.1 above the terrain under the spawn cell
        float zt;
        V3 nt;
        ground.sample(rd.pos.x, rd.pos.y, zt, nt);
        rd.pos.z += zt;
    }
    return rd;
}
// RESET_SRC: where a resetting env's pose comes from.  By default drawn on the spot; the quad-form step kernel of the small batches has
// a helper wavefront draw the block's resets while the physics runs (round 6): the draw is two or three DEPENDENT memory round trips
// (Philox -> the spawn-cell table -> on a heightfield the terrain under the cell) behind the last sub-step, with mean episodes of ~50
// steps every other block of 32 envs has one per step, and the launch is as long as its slowest block.
struct InlineVisualReset {
    template <class Ground>
    WL_DEV VisReset operator()(const WlVisualParams& p, const WlTravMap& m, const Ground& ground, uint32_t gid, uint64_t step, uint64_t seed, int) const {
        return visual_reset_pose(p, m, ground, gid, step, seed);
    }
};
template <int LANES, class Ground = FlatGround, class RESET_SRC = InlineVisualReset>
WL_DEV CamPose visual_env_step(const WlVisualParams& p, const VehDerived& vd, const WlEnvBuffers& b, const WlTravMap& m, float2 a,
                               const WlStepOut& out, const uint64_t seed, const uint64_t step, const Rows& S, const int e, const int wid,
                               const bool lead, float* blk_metrics, const Ground ground = Ground{}, float* __restrict__ prop = nullptr,
                               const RESET_SRC& reset_src = RESET_SRC()) {
    const WlVehicleParams& vp = p.vehicle;
    {
        const uint32_t gid = (uint32_t)(b.env_offset + e);
        float v_t, delta;
        process_action(p.action, a.x, a.y, v_t, delta);
        EnvConst ec;
        joint_targets(p.action, v_t, delta, ec.steer_target, ec.wheel_target);
        env_const_rows(ec, vp, vd, S.ld(WL_S_MASS, e), S.ld(WL_S_MU_S, e), S.ld(WL_S_MU_D, e), S.ld(WL_S_DAMP, e));
        if constexpr (LANES == 4) env_const_lane(ec, vp, vd, wid);
        VehState s;
        V3 pos = ld3(S, WL_S_PX, e);
        s.q = Quat{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
        s.v = ld3(S, WL_S_VX, e);
        V3 ww = ld3(S, WL_S_WX, e);
        if constexpr (LANES == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s.wheel[i] = S.ld(WL_S_WHEEL_BL + i, e);
        } else {
            s.wheel[0] = S.ld(WL_S_WHEEL_BL + wid, e);
        }
        s.th = S.ld(WL_S_STEER_POS, e);
        s.om = S.ld(WL_S_STEER_VEL, e);
        {
            const Mat3 R = mat_from_quat(s.q);
            s.x = pos + vp.cg_z * v3(R.r0.z, R.r1.z, R.r2.z);
            s.wb = mul_t(R, ww);
        }
        // bookkeeping rows: the quad form (one wavefront per SIMD, registers to spare) requests them BEFORE the physics loop
        // and finds them landed behind it; the lane form fetches them after it (registers are worth more there)
        int ep_len_in = 0;
        float epsum_in[WL_VR_NTERMS];
#pragma unroll
        for (int i = 0; i < WL_VR_NTERMS; ++i) epsum_in[i] = 0.f;
        auto fetch_bookkeeping = [&]() {
            ep_len_in = b.episode_len[e];
            if (p.log_episode_sums) {
#pragma unroll
                for (int i = 0; i < WL_VR_NTERMS; ++i) epsum_in[i] = S.ld(WL_S_EPSUM0 + i, e);
            }
        };
        if constexpr (LANES == 4) fetch_bookkeeping();
#ifndef WL_WHEEL_CORNER_CACHE
#define WL_WHEEL_CORNER_CACHE 1
#endif
        if constexpr (LANES == 1 && !Ground::kFlat && WL_WHEEL_CORNER_CACHE) {   // lane form on a heightfield: see HeightFieldGroundCached
            const HeightFieldGroundCached cached(ground);
            vehicle_integrate<LANES, HeightFieldGroundCached, true, -1, true>(vp, vd, ec, s, cached, wid);
        } else {
            vehicle_integrate<LANES, Ground, true, -1, true>(vp, vd, ec, s, ground, wid);
        }
        if constexpr (LANES != 4) {
            asm volatile("" ::: "memory");
            fetch_bookkeeping();
        }
        const Mat3 R = mat_from_quat(s.q);
        ww = mul(R, s.wb);
        pos = s.x - vp.cg_z * v3(R.r0.z, R.r1.z, R.r2.z);
        int ep_len = ep_len_in + 1;
        const bool truncated = ep_len >= p.max_episode_length;
        float wheel_sum;
        if constexpr (LANES == 1) wheel_sum = s.wheel[0] + s.wheel[1] + s.wheel[2] + s.wheel[3];
        else wheel_sum = quad_sum(s.wheel[0]);
        const float chk = pos.x + pos.y + pos.z + s.q.w + s.q.x + s.q.y + s.q.z + s.v.x + s.v.y + s.v.z + ww.x + ww.y +
                          ww.z + wheel_sum + s.th + s.om;
        const bool finite = __builtin_isfinite(chk);
        const bool oom = finite && out_of_map(m, pos.x, pos.y);
        const bool terminated = !finite || oom;
        const V3 vb = mul_t(R, s.v);
        float t[WL_VR_NTERMS];
        t[WL_VR_TRAVERSABLE] = finite ? (traversable(m, pos.x, pos.y) ? 1.f : -1.f) : 0.f;   // :309-312
        t[WL_VR_FORWARD_VEL] = vb.x;                                                          // :370-371
        const float step_dt = p.sim_dt * (float)p.decimation;
        float reward = 0.f;
        float epsum[WL_VR_NTERMS];
#pragma unroll
        for (int i = 0; i < WL_VR_NTERMS; ++i) {
            const float w = p.weight[i];
            const float c = (w != 0.f && finite) ? t[i] * w * step_dt : 0.f;
            reward += c;
            epsum[i] = p.log_episode_sums ? epsum_in[i] + c : 0.f;
        }
        if (lead) {
            out.reward[e] = reward;
            out.terminated[e] = terminated ? 1 : 0;
            out.truncated[e] = truncated ? 1 : 0;
            if (out.dones) out.dones[e] = (terminated || truncated) ? 1 : 0;
        }
        float a0 = a.x, a1 = a.y;
        if (terminated || truncated) {
            if (lead) {
#pragma unroll
            for (int i = 0; i < WL_VR_NTERMS; ++i) atomicAdd(&blk_metrics[WL_M_EPSUM0 + i], epsum[i]);
            atomicAdd(&blk_metrics[WL_M_RESETS], 1.f);
            if (truncated) atomicAdd(&blk_metrics[WL_M_TIMEOUTS], 1.f);
            if (oom) atomicAdd(&blk_metrics[WL_M_TERM0], 1.f);
            if (!finite) atomicAdd(&blk_metrics[WL_M_NONFINITE], 1.f);
            atomicAdd(&blk_metrics[WL_M_EPLEN], (float)ep_len);
            }
#pragma unroll
            for (int i = 0; i < WL_VR_NTERMS; ++i) epsum[i] = 0.f;
            if (!finite) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s.wheel[i] = 0.f;
                s.th = s.om = 0.f;
            }
            const VisReset rd = reset_src(p, m, ground, gid, step, seed, e);
            pos = rd.pos;
            s.q = rd.q;
            s.v = v3(0.f, 0.f, 0.f);
            ww = v3(0.f, 0.f, 0.f);
            ep_len = 0;
            a0 = a1 = 0.f;
        }
        if constexpr (LANES == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) S.st(WL_S_WHEEL_BL + i, e, s.wheel[i]);
        } else {
            S.st(WL_S_WHEEL_BL + wid, e, s.wheel[0]);
        }
        if (lead) {
            st3(S, WL_S_PX, e, pos);
            S.st(WL_S_QW, e, s.q.w);
            S.st(WL_S_QX, e, s.q.x);
            S.st(WL_S_QY, e, s.q.y);
            S.st(WL_S_QZ, e, s.q.z);
            st3(S, WL_S_VX, e, s.v);
            st3(S, WL_S_WX, e, ww);
            S.st(WL_S_STEER_POS, e, s.th);
            S.st(WL_S_STEER_VEL, e, s.om);
            S.st(WL_S_ACT0, e, a0);
            S.st(WL_S_ACT1, e, a1);
            if (p.log_episode_sums) {
    #pragma unroll
                for (int i = 0; i < WL_VR_NTERMS; ++i) S.st(WL_S_EPSUM0 + i, e, epsum[i]);
            }
            b.episode_len[e] = ep_len;
            if (prop) {     // base_lin_vel | base_ang_vel | last_action (clipped) of the post-reset state
                const Mat3 R2 = mat_from_quat(s.q);
                const V3 vb2 = mul_t(R2, s.v), wb2 = mul_t(R2, ww);
                prop[0] = vb2.x, prop[1] = vb2.y, prop[2] = vb2.z, prop[3] = wb2.x, prop[4] = wb2.y, prop[5] = wb2.z;
                prop[6] = clampf(a0, -1.f, 1.f), prop[7] = clampf(a1, -1.f, 1.f);
            }
        }
        return CamPose{pos.x, pos.y, pos.z, s.q.w, s.q.x, s.q.y, s.q.z, s.v.x, s.v.y, s.v.z, ww.x, ww.y, ww.z, a0, a1};
    }
}

template <int LANES, int QB = kBlock /* quad form: threads per block (see drift_step_kernel) */, class Ground = FlatGround>
__global__ void __launch_bounds__(kBlock) visual_step_kernel(const WlVisualParams p_arg, const VehDerived vd_arg, const WlEnvBuffers b,
                                                             const WlTravMap m, const float2* __restrict__ actions,
                                                             const WlStepOut out, const uint64_t seed, const uint64_t step,
                                                             const Ground ground = Ground{}, const int prop_stride = 0, const int prop_offset = 0) {
    __shared__ float blk_metrics[WL_M_COUNT];
    WlVisualParams p = p_arg;
    VehDerived vd = vd_arg;
    if constexpr (LANES == 4) {   // latency form: one batch of vector loads instead of dependent scalar-load round trips
        kernarg_vector_copy2(0, p, vd);   // both argument structs as ONE burst (two copies: a wait in the middle, see the helper)
        keep_scalar_common(p, p_arg);
        vd.n_sub = vd_arg.n_sub;
    }
    constexpr int kEnvs = (LANES == 4 ? QB : kBlock) / LANES;
    // heightfield ground only: launched with QB + 64 threads, the last wavefront draws the block's resets (on the plane the draw is one
    // round trip shorter and the helper bought nothing: visual env.step 35.4 against 35.2 us, same box)
    constexpr bool kHelper = LANES == 4 && QB == 128 && !Ground::kFlat;
    __shared__ VisReset reset_lds[kHelper ? kEnvs : 1];
    __shared__ int reset_ready;
    const int wid = LANES == 1 ? 0 : (threadIdx.x & 3);
    const bool lead = LANES == 1 || wid == 0;
    const int e = blockIdx.x * kEnvs + threadIdx.x / LANES;
    if (threadIdx.x < WL_M_COUNT) blk_metrics[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) reset_ready = 0;
    const int m_slot = b.metrics_slots > 1 ? (int)(step % (uint64_t)b.metrics_slots) : 0;
    if (b.metrics_slots > 1) clear_metric_slot(b, (m_slot + 1) % b.metrics_slots);
    __syncthreads();
    const Rows S = make_rows(b.state, b.stride);
    if constexpr (kHelper) {
        if (threadIdx.x >= QB) {
            const int j = (int)threadIdx.x - QB, ej = blockIdx.x * kEnvs + j;
            if (j < kEnvs && ej < b.n_envs) reset_lds[j] = visual_reset_pose(p_arg, m, ground, (uint32_t)(b.env_offset + ej), step, seed);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (j == 0) __hip_atomic_store(&reset_ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (e < b.n_envs) {
            struct HelperReset {
                const VisReset* reset_lds;
                int* reset_ready;
                int e0;
                WL_DEV VisReset operator()(const WlVisualParams&, const WlTravMap&, const Ground&, uint32_t, uint64_t, uint64_t, int e) const {
                    // set ~1 us into the launch, read ~8 us into it: the loop is the guarantee
                    while (__hip_atomic_load(reset_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
                    return reset_lds[e - e0];
                }
            };
            const HelperReset hr{reset_lds, &reset_ready, (int)(blockIdx.x * kEnvs)};
            visual_env_step<LANES, Ground, HelperReset>(p, vd, b, m, actions[e], out, seed, step, S, e, wid, lead, blk_metrics, ground,
                                                        prop_stride > 0 ? out.obs + (int64_t)e * prop_stride + prop_offset : nullptr, hr);
        }
    } else {
        if (e < b.n_envs)
            visual_env_step<LANES, Ground>(p, vd, b, m, actions[e], out, seed, step, S, e, wid, lead, blk_metrics, ground,
                                           prop_stride > 0 ? out.obs + (int64_t)e * prop_stride + prop_offset : nullptr);
    }
    __syncthreads();
    if (threadIdx.x < WL_M_COUNT) {
        const float v = blk_metrics[threadIdx.x];
        if (v != 0.f) atomicAdd(metric_shard(b, m_slot) + threadIdx.x, v);   // threads 0..15 = wavefront 0 of the block
    }
}

// camera ray of pixel (row, col) of the FULL 60 x 80 image in the body frame: optical axis = body +x, image right =
// body -y, image down = body -z (ROS optical convention of the reference's camera offset, :241-243)
WL_DEV V3 pixel_ray_body(const WlVisualParams& p, int row, int col) {
    return v3(1.f, -(((float)col + 0.5f - p.cx) / p.fx), -(((float)row + 0.5f - p.cy) / p.fy));
}

// (BlockSync / GroupSync -- how the wavefronts that render one image meet -- are in wl_kernel_common.h)
// sum over the GT threads (whole wavefronts) that render one image
template <int GT, class SYNC>
WL_DEV float group_sum(float v, float* scratch /* [GT / 64], this group's */, int gt, SYNC& sync) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    sync();
    if ((gt & 63) == 0) scratch[gt >> 6] = v;
    sync();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < GT / 64; ++w) t += scratch[w];
    return t;
}

// how a pixel's map cell is read: a byte gather from the global map, or one bit of the LDS-resident copy of the whole map
// (WlTravMap.bits: the camera's 3200 divergent byte gathers per image were what bound it -- one lane per cycle and CU through
// the texture addresser; an LDS read costs a thirty-second of that)
// `masked(k, on)`: the cell, or false for a pixel that is off the map.  Global form (round 6): the byte map through a buffer resource
// -- a 32-bit lane offset instead of a 64-bit address (v_mad_u64_u32 + v_lshl_add_u64 per pixel went), and an off-map pixel asks for
// an offset outside the buffer (the bounds check returns 0 = not traversable) instead of branching around its gather: with the branch
// the compiler waited for every pixel's gather inside it (`s_waitcnt vmcnt(0)` fourteen times per lane), now a lane's fourteen are in
// flight together as the loop was written for.
struct GlobalMapLookup {
    __amdgpu_buffer_rsrc_t rsrc;
    WL_DEV explicit GlobalMapLookup(const WlTravMap& m)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(m.map), 0, m.rows * m.cols, 0x00020000)) {}
    WL_DEV bool masked(int k, bool on) const { return __builtin_amdgcn_raw_buffer_load_b8(rsrc, on ? k : -1, 0, 0) != 0; }
};
struct LdsBitLookup {
    const uint32_t* bits;
    WL_DEV bool masked(int k, bool on) const {
        const int kk = on ? k : 0;
        return on && ((bits[kk >> 5] >> (kk & 31)) & 1u);
    }
};
constexpr int kMapWords = WL_VIS_LDS_MAP_CELLS / 32;   // 32 KB of LDS
// all threads of the block copy the bit map into LDS (coalesced dwords; the caller syncs).  Every request of a thread is issued
// before the first LDS write (rolled, the copy was a chain of ~10 dependent L2 round trips per block: slower than the gathers
// it replaces).
template <int THREADS>
WL_DEV void stage_map_bits(const WlTravMap& m, uint32_t* lds) {
    constexpr int kPer = (kMapWords + THREADS - 1) / THREADS;
    const int n_words = (m.rows * m.cols + 31) >> 5;
    uint32_t w[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int idx = (int)threadIdx.x + k * THREADS;
        w[k] = idx < n_words ? m.bits[idx] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int idx = (int)threadIdx.x + k * THREADS;
        if (idx < kMapWords) lds[idx] = w[k];
    }
}
inline bool lds_map_ok(const WlTravMap* m) { return m->bits != nullptr && (int64_t)m->rows * m->cols <= WL_VIS_LDS_MAP_CELLS; }

WL_DEV int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// the rendered image with its two reflected border columns on either side: pitch 84 floats = 21 sixteen-byte slots, so
// the 8-float window of a 4 x 4 output patch is two aligned ds_read_b128 and, 4 rows x 21 slots being 4 mod 16, the 16-lane
// groups a b128 read is served in (lanes of one patch row + 8 lanes of the next) hit 16 distinct slots.  (With the
// unpadded [40][80] image each of the 64 reads per thread was a 4-byte read at a lane stride of 4 floats: 8-way conflicts.)
constexpr int kPitch = kImgW + 4, kImgPix = kImgH * kPitch, kImgFloats = kImgPix + 8;   // + the blur's five tap weights (render_image)

// VisualObsCfg.PolicyCfg (:38-58): camera (3200) | base_lin_vel (3) | base_ang_vel (3) | last_action clip +-1 (2) of ONE env,
// by a group of GT threads (gt = thread in the group; `img` [kImgFloats] and `red` [GT / 64] are the group's LDS).  Two
// `sync()`s inside when the augmentation needs the whole image; a group without an env (`valid` false) renders its
// neighbour's pose and stores nothing.
// STREAM: the observation block is far larger than the caches (> 256 MB of rows per launch): non-temporal row stores
template <int GT, bool STREAM = false, class SYNC, class LOOKUP>
WL_DEV void render_image(const WlVisualParams& p, const WlTravMap& m, const LOOKUP& lookup, const CamPose& cp, float* __restrict__ row,
                         float* img, float* red, const int gt, const bool valid, SYNC& sync) {
    const V3 pos = v3(cp.px, cp.py, cp.pz);
    const Quat q{cp.qw, cp.qx, cp.qy, cp.qz};
    const Mat3 R = mat_from_quat(q);
    const V3 o = pos + mul(R, v3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]));
    const MapFast mf = map_fast(m);
    const float inv_fx = rcp(p.fx), inv_fy = rcp(p.fy);
    const bool plain = p.contrast == 1.f && !(p.blur_sigma > 0.f);   // no augmentation that needs the whole image
    // torchvision draws the order of ColorJitter's ops per call: contrast before brightness when set (only matters with a blend)
    const bool cfirst = p.contrast_first != 0 && p.contrast != 1.f;
    if (gt == GT - 1 && valid) {   // proprioception: the last lane (its wave renders the fewest pixels)
        const V3 vb = mul_t(R, v3(cp.vx, cp.vy, cp.vz)), wb = mul_t(R, v3(cp.wx, cp.wy, cp.wz));
        float* t = row + WL_VIS_NPIX;
        t[0] = vb.x; t[1] = vb.y; t[2] = vb.z;
        t[3] = wb.x; t[4] = wb.y; t[5] = wb.z;
        t[6] = clampf(cp.a0, -1.f, 1.f);
        t[7] = clampf(cp.a1, -1.f, 1.f);
    }
    // ---- render: one ray per pixel against the z = 0 plane.  d = R (1, dy(col), dz(row)) ----
    const V3 c0 = v3(R.r0.x, R.r1.x, R.r2.x), c1 = v3(R.r0.y, R.r1.y, R.r2.y), c2 = v3(R.r0.z, R.r1.z, R.r2.z);
    float part = 0.f;
    // Thread -> one image column and every third row (GT = 256: 240 of the threads, 3 rows x 80 columns per pass, 14 passes):
    // the ray direction then advances by a constant vector per pass (3 adds instead of 2 conversions + 8 fma), and the
    // lanes of a wavefront still store consecutive columns.  Software-pipelined like the height scan: every pixel of
    // this lane computes its map cell and issues its byte gather first (14 in flight per lane), then all are shaded and
    // stored -- rolled, each pixel paid the gather latency in turn.
    constexpr int kRowsPerPass = GT / kImgW, kIter = (kImgH + kRowsPerPass - 1) / kRowsPerPass;
    const int tc = gt % kImgW, tr = gt / kImgW;
    const bool lane_on = tr < kRowsPerPass;
    const float dy = -(((float)tc + 0.5f - p.cx) * inv_fx), dz0 = -(((float)(tr + WL_VIS_CROP) + 0.5f - p.cy) * inv_fy);
    V3 d = fma3(dz0, c2, fma3(dy, c1, c0));
    // (a product of its own: under -ffp-contract=fast a bare product feeding the `d + dstep` below is fused into that add per
    // instantiation -- the block = env kernel and the persistent kernel then disagree in the last bit of a ray direction and, one
    // image in 20 000, about the map cell a pixel falls into)
    const float kstep = -(float)kRowsPerPass * inv_fy;
    const V3 dstep = v3(fmaf(kstep, c2.x, 0.f), fmaf(kstep, c2.y, 0.f), fmaf(kstep, c2.z, 0.f));
    const float mx = mf.off_x * mf.inv_rs, my = mf.off_y * mf.inv_cs;   // cell = (int)(h * inv + off * inv)
    bool cell[kIter];
    bool hit[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const float t = -o.z * rcp(fminf(d.z, -1e-6f));
        const float hx = fmaf(t, d.x, o.x), hy = fmaf(t, d.y, o.y);
        hit[it] = d.z < -1e-6f;                                         // else: sky
        const bool on_map = hit[it] && fabsf(hx) <= mf.half_w && fabsf(hy) <= mf.half_h;
        // on the map |h| <= half extent, so h * inv + (n / 2 + 0.5) lies in [0.5, n + 0.5]: the conversion truncates, only the upper
        // clamp can bind (off the map the index is not used); 24-bit multiply (full rate; rows, cols < 2^23)
        const int xi = min((int)fmaf(hx, mf.inv_rs, mx), m.rows - 1);
        const int yi = min((int)fmaf(hy, mf.inv_cs, my), m.cols - 1);
        cell[it] = lookup.masked(__mul24(yi, m.cols) + xi, on_map);     // white path on black (utils/__init__.py:47-50)
        d = d + dstep;
    }
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int r = tr + it * kRowsPerPass;
        if (lane_on && r < kImgH) {
            float v = hit[it] ? (cell[it] ? 1.f : 0.f) : p.sky;
            if (!cfirst) v = clampf(v * p.brightness, 0.f, 1.f);       // ColorJitter brightness (before the contrast blend)
            if (plain) {
                if (valid) {   // grayscale + Normalize([0.5], [0.5]) straight to HBM
                    if constexpr (STREAM) __builtin_nontemporal_store((v * 0.9999f - 0.5f) * 2.f, row + r * kImgW + tc);
                    else row[r * kImgW + tc] = (v * 0.9999f - 0.5f) * 2.f;
                }
            } else {
                img[r * kPitch + 2 + tc] = v;      // (the two border columns on either side stay unwritten: the edge patches reflect when they read)
                part += v;
            }
        }
    }
    if (plain) return;
    // ---- augmentation on the LDS-resident image.  One 4 x 4 output patch per thread (10 x 20 patches = 200 threads):
    // the 8 x 8 input neighbourhood is read once (contrast blend applied on the fly), both passes of the separable 5-tap
    // Gaussian run in registers, and each patch row leaves as one 16-byte store.
    // GaussianBlur(5, sigma), torchvision's kernel1d: the five exponentials once per image (lanes 0 - 4, through LDS) instead of once per
    // thread -- with the division and the normalisation they were 100 vector instructions + 11 quarter-rate ones of every thread's ~1000,
    // in a launch that is bound by vector issue (round 6).  Same operations on the same values: same bits.
    float* wl = img + kImgPix;
    if (gt < 5 && p.blur_sigma > 0.f) {
        const float x = (float)(gt - 2) / p.blur_sigma;
        wl[gt] = __expf(-0.5f * x * x);
    }
    const float mean = 0.9999f * group_sum<GT>(part, red, gt, sync) * (1.f / (float)(kImgH * kImgW));   // also orders the img / weight writes
    const float cc = p.contrast, cm = (1.f - p.contrast) * mean;   // ColorJitter contrast: blend with the grey mean
    // a blend TOWARDS the mean (contrast <= 1) stays inside [0, 1], its clamp is the identity, and the blur is linear with
    // weights summing to one: blur(cc v + cm) = cc blur(v) + cm -- applied to the 16 outputs instead of the 64 inputs
    const bool fold = cc <= 1.f && cc >= 0.f && !cfirst;   // (contrast first: the brightness clamp sits between blend and blur)
    const float oc_ = fold ? cc : 1.f, om_ = fold ? cm : 0.f;
    float w[5] = {0.f, 0.f, 1.f, 0.f, 0.f};
    if (p.blur_sigma > 0.f) {
        float wsum = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            w[j] = wl[j];
            wsum += w[j];
        }
        const float inv = 1.f / wsum;
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] *= inv;
    }
    constexpr int kPatchCols = kImgW / 4, kPatches = (kImgH / 4) * kPatchCols;
    if (gt < kPatches && valid) {
        const int pr = (gt / kPatchCols) * 4, pc = (gt % kPatchCols) * 4;
        // reflect padding (torchvision) of the columns, at READ time: only the first and last patch of a row see columns outside the
        // image (-2 -> 2, -1 -> 1; 80 -> 78, 81 -> 77), four selects per row for them -- written at render time the four border
        // columns cost every pixel of every thread two compound tests and two predicated LDS writes (round 6: ~110 instructions per thread)
        const bool left = pc == 0, right = pc == kImgW - 4;
        float hrow[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4* line = reinterpret_cast<const float4*>(img + reflect(pr + i - 2, kImgH) * kPitch + pc);   // columns pc - 2 .. pc + 5
            const float4 lo4 = line[0], hi4 = line[1];
            float v[8] = {left ? hi4.x : lo4.x, left ? lo4.w : lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, right ? hi4.x : hi4.z, right ? lo4.w : hi4.w};
            if (!fold) {   // contrast > 1 can leave [0, 1]: the clamp sits between the blend and the blur
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = clampf(fmaf(cc, v[j], cm), 0.f, 1.f);
                if (cfirst) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = clampf(v[j] * p.brightness, 0.f, 1.f);
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o)
                hrow[i][o] = fmaf(w[0], v[o], fmaf(w[1], v[o + 1], fmaf(w[2], v[o + 2], fmaf(w[3], v[o + 3], w[4] * v[o + 4]))));
        }
#pragma unroll
        for (int orow = 0; orow < 4; ++orow) {
            float4 out4;
            float* o4 = reinterpret_cast<float*>(&out4);
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                const float g = fmaf(w[0], hrow[orow][oc], fmaf(w[1], hrow[orow + 1][oc], fmaf(w[2], hrow[orow + 2][oc],
                                fmaf(w[3], hrow[orow + 3][oc], w[4] * hrow[orow + 4][oc]))));
                o4[oc] = (fmaf(oc_, g, om_) * 0.9999f - 0.5f) * 2.f;   // (folded contrast,) grayscale + Normalize([0.5], [0.5])
            }
            // row base = e * 3208 floats = e * 12832 B (16-B aligned), patch column is a multiple of 4 floats
            if constexpr (STREAM) {
                typedef float wl_f4v __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(wl_f4v{out4.x, out4.y, out4.z, out4.w}, reinterpret_cast<wl_f4v*>(row + (pr + orow) * kImgW + pc));
            } else {
                *reinterpret_cast<float4*>(row + (pr + orow) * kImgW + pc) = out4;
            }
        }
    }
}

WL_DEV CamPose load_cam_pose(const Rows& S, const int e) {
    return CamPose{S.ld(WL_S_PX, e), S.ld(WL_S_PX + 1, e), S.ld(WL_S_PX + 2, e), S.ld(WL_S_QW, e), S.ld(WL_S_QX, e),
                   S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e), S.ld(WL_S_VX, e), S.ld(WL_S_VX + 1, e), S.ld(WL_S_VX + 2, e),
                   S.ld(WL_S_WX, e), S.ld(WL_S_WX + 1, e), S.ld(WL_S_WX + 2, e), S.ld(WL_S_ACT0, e), S.ld(WL_S_ACT1, e)};
}

// block = env: the observation of the state as it stands (reset / first observation / lane-form steps); map cells by byte
// gathers from global memory (maps too large for LDS, or no bit map supplied)
// MINW = 7: seven wavefronts per SIMD (72 VGPRs + 24 bytes of scratch) instead of the six that the free allocation (73 -> 80 VGPRs)
// allows -- round 4, env.step() in us at 4096 / 65 536 / 262 144 envs: 49.2 -> 49.0 / 415.5 -> 400 / 1606 -> 1543 (eight, 64 VGPRs + 40
// bytes: 49.6 / 408 / 1614).  The small launches keep the free allocation: nothing to win there, and their scratch rows reach the fabric
// (counter traffic of the 4096-env step 1.19 -> 1.40 x algorithmic).
#ifndef WL_CAM_MIN_WAVES
#define WL_CAM_MIN_WAVES 7
#endif
#ifndef WL_CAM_MIN_WAVES_ENVS
#define WL_CAM_MIN_WAVES_ENVS 32768
#endif
template <bool STREAM, int MINW = 1>
__global__ void __launch_bounds__(kCam, MINW) visual_obs_kernel(const WlVisualParams p, const WlEnvBuffers b, const WlTravMap m,
                                                                float* __restrict__ obs) {
    __shared__ __attribute__((aligned(16))) float img[kImgFloats];
    __shared__ float red[kCam / 64];
    const int e = blockIdx.x;
    const CamPose cp = load_cam_pose(make_rows(b.state, b.stride), e);
    BlockSync sync;
    render_image<kCam, STREAM>(p, m, GlobalMapLookup(m), cp, obs + (int64_t)e * WL_VIS_OBS_DIM, img, red, (int)threadIdx.x, true, sync);
}

// (Round 3: the same camera with the whole map in LDS as one bit per cell -- persistent blocks of three render groups, map staged
// once per block -- measured SLOWER than the byte gathers at 4096 envs: 26.1 against 18.2 us plain, 38.3 against 28.1 us
// augmented.  The gathers were not what bound it: 52.6 MB of observation rows in 18 us are 2.9 TB/s of stores, and the LDS form
// gives up a quarter of the resident wavefronts for its 32 KB.  In the persistent rollout below, where the blocks are resident
// for the whole rollout anyway, the LDS map does pay: 29.0 -> 26.0 us per step plain, 37.5 -> 35.9 augmented.)
#ifndef WL_VIS_STREAM_BYTES
#define WL_VIS_STREAM_BYTES 0ll
#endif
inline void launch_visual_obs(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, float* obs, hipStream_t hs) {
    // non-temporal rows at every size unless WL_FLAG_NO_STREAM asks otherwise (rounds 1-3: only beyond the 256 MB Infinity Cache, > 20 000
    // envs, where they are worth 12 %; round 4, env.step() in us with / without: 4096 envs 49.2 / 50.7, 8192: 72.2 / 73.5, 16 384: 117.6 / 119.8)
    const bool dense = b->n_envs > WL_CAM_MIN_WAVES_ENVS;
    if (use_streaming(b, (int64_t)b->n_envs * WL_VIS_OBS_DIM * 4, WL_VIS_STREAM_BYTES)) {
        if (dense) visual_obs_kernel<true, WL_CAM_MIN_WAVES><<<b->n_envs, kCam, 0, hs>>>(*p, *b, *m, obs);
        else visual_obs_kernel<true><<<b->n_envs, kCam, 0, hs>>>(*p, *b, *m, obs);
    } else {
        if (dense) visual_obs_kernel<false, WL_CAM_MIN_WAVES><<<b->n_envs, kCam, 0, hs>>>(*p, *b, *m, obs);
        else visual_obs_kernel<false><<<b->n_envs, kCam, 0, hs>>>(*p, *b, *m, obs);
    }
}

// K env.step()s in ONE launch with pre-staged actions [K][n][2] (open-loop rollouts; quad form, n <= 32 768), the visual
// counterpart of elev_rollout_persistent_kernel.  Block = 16 envs, 1024 threads.  Wavefront 0 steps them (through the
// state rows, which stay in this CU's cache: the camera, not the physics, is the longer leg here) and leaves each step's
// poses in one of two LDS buffers; the other twelve wavefronts -- three groups of GT = 256 threads, one image per group
// and pass -- render step k WHILE wavefront 0 already integrates step k + 1.  One s_barrier per step and wavefront; the
// render groups meet among themselves through GroupSync.  Episode metrics of all K steps go to ring slot `slots.cur`.
// (GT is visual_obs_kernel's: a group of 320 threads -- 4 image rows per pass, fifteen render wavefronts -- advances the
// ray direction in different increments, and rays that graze a cell edge then resolve differently: not bit-identical.)
constexpr int kPersistGroups = (1024 - 64) / kCam, kPersistThreads = 64 + kPersistGroups * kCam;
template <int EPB, bool LDSMAP>
__global__ void __launch_bounds__(kPersistThreads) visual_rollout_persistent_kernel(
    const WlVisualParams p_arg, const VehDerived vd_arg, const WlEnvBuffers b, const WlTravMap m, const float2* __restrict__ actions,
    const WlStepOut out, const int64_t obs_step_stride, const int64_t vec_step_stride, const int n_steps, const uint64_t seed,
    const uint64_t step0, const MetricSlots slots) {
    constexpr int GT = kCam, kGroups = kPersistGroups, kPasses = (EPB + kGroups - 1) / kGroups;
    static_assert(GT % 64 == 0 && GT >= kImgW && kGroups >= 1 && 4 * EPB <= 64, "whole wavefronts per render group; one physics wavefront");
    __shared__ float blk_metrics[WL_M_COUNT];
    __shared__ CamPose pose[2][EPB];
    __shared__ __attribute__((aligned(16))) float img[kGroups * kImgFloats];
    __shared__ float red[kGroups * (GT / 64)];
    __shared__ int arrivals[kGroups];
    __shared__ uint32_t mapbits[LDSMAP ? kMapWords : 1];
    const int tid = threadIdx.x;
    if (tid < WL_M_COUNT) blk_metrics[tid] = 0.f;
    if (tid < kGroups) arrivals[tid] = 0;
    if (b.metrics_slots > 1) clear_metric_slot(b, slots.next);
    if constexpr (LDSMAP) stage_map_bits<kPersistThreads>(m, mapbits);
    __syncthreads();
    const int e0 = blockIdx.x * EPB;
    if (tid < 64) {
        const int wid = tid & 3, e = e0 + (tid >> 2);
        const bool valid = tid < 4 * EPB && e < b.n_envs;
        const Rows S = make_rows(b.state, b.stride);
        WlVisualParams p;
        VehDerived vd;
        kernarg_vector_copy2(0, p, vd);
        keep_scalar_common(p, p_arg);
        vd.n_sub = vd_arg.n_sub;
#pragma unroll 1
        for (int k = 0; k < n_steps; ++k) {
            if (valid) {
                WlStepOut o = out;
                o.obs += k * obs_step_stride;
                o.reward += k * vec_step_stride;
                o.terminated += k * vec_step_stride;
                o.truncated += k * vec_step_stride;
                if (o.dones) o.dones += k * vec_step_stride;
                const float2 a = actions[(int64_t)k * b.n_envs + e];
                const CamPose cp = visual_env_step<4>(p, vd, b, m, a, o, seed, step0 + (uint64_t)k, S, e, wid, wid == 0, blk_metrics);
                if (wid == 0) pose[k & 1][tid >> 2] = cp;
            }
            __syncthreads();   // barrier k: the poses of step k are published
        }
        if (tid < WL_M_COUNT) {   // only this wavefront accumulated
            const float v = blk_metrics[tid];
            if (v != 0.f) atomicAdd(metric_shard(b, slots.cur) + tid, v);
        }
        return;
    }
    // ---- the other wavefronts: the camera of step k, one step behind the physics ----
    const int t = tid - 64, grp = t / GT, gt = t % GT;
    const int n_here = min(EPB, b.n_envs - e0);
    const bool plain = p_arg.contrast == 1.f && !(p_arg.blur_sigma > 0.f);
    GroupSync sync{arrivals + min(grp, kGroups - 1), 0, GT / 64};
#pragma unroll 1
    for (int k = 0; k < n_steps; ++k) {
        __syncthreads();       // barrier k (also: every group is done with its previous image)
        if (grp >= kGroups) continue;   // wavefronts left over when the block is no whole number of groups
        float* obs_k = out.obs + k * obs_step_stride;
#pragma unroll 1
        for (int pass = 0; pass < kPasses; ++pass) {
            const int j = pass * kGroups + grp;
            if (j >= n_here) break;                    // group-uniform: GroupSync involves this group only
            if (pass > 0 && !plain) sync();            // the previous image is still being read by the blur
            const CamPose cp = pose[k & 1][j];
            int gtl = gt;
            asm volatile("" : "+v"(gtl));              // per-pass copy: keeps the pixel / patch bookkeeping from being hoisted out of
                                                       // the loops and held in ~50 registers across them
            if constexpr (LDSMAP)
                render_image<GT>(p_arg, m, LdsBitLookup{mapbits}, cp, obs_k + (int64_t)(e0 + j) * WL_VIS_OBS_DIM, img + grp * kImgFloats,
                                 red + grp * (GT / 64), gtl, true, sync);
            else
                render_image<GT>(p_arg, m, GlobalMapLookup(m), cp, obs_k + (int64_t)(e0 + j) * WL_VIS_OBS_DIM, img + grp * kImgFloats,
                                 red + grp * (GT / 64), gtl, true, sync);
        }
    }
}

template <class Ground = FlatGround>
__global__ void __launch_bounds__(kBlock) visual_reset_kernel(const WlVisualParams p, const WlEnvBuffers b, const WlTravMap m,
                                                              const uint8_t* __restrict__ mask, uint64_t seed, uint64_t step,
                                                              const Ground ground = Ground{}) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= b.n_envs) return;
    if (mask && !mask[e]) return;
    const Rows S = make_rows(b.state, b.stride);
    VisReset rd = draw_visual_reset(p, m, (uint32_t)(b.env_offset + e), step, seed);
    if constexpr (!Ground::kFlat) {
        float zt;
        V3 nt;
        ground.sample(rd.pos.x, rd.pos.y, zt, nt);
        rd.pos.z += zt;
    }
    st3(S, WL_S_PX, e, rd.pos);
    S.st(WL_S_QW, e, rd.q.w);
    S.st(WL_S_QX, e, rd.q.x);
    S.st(WL_S_QY, e, rd.q.y);
    S.st(WL_S_QZ, e, rd.q.z);
    st3(S, WL_S_VX, e, v3(0.f, 0.f, 0.f));
    st3(S, WL_S_WX, e, v3(0.f, 0.f, 0.f));
    S.st(WL_S_ACT0, e, 0.f);
    S.st(WL_S_ACT1, e, 0.f);
#pragma unroll
    for (int i = 0; i < WL_MAX_REW_TERMS; ++i) S.st(WL_S_EPSUM0 + i, e, 0.f);
    b.episode_len[e] = 0;
}

__global__ void __launch_bounds__(kBlock) visual_mdp_kernel(const WlVisualParams p, const WlTravMap m, int n, int64_t stride,
                                                            const float* __restrict__ pos, const float* __restrict__ vb,
                                                            float* __restrict__ terms, uint8_t* __restrict__ oom,
                                                            int32_t* __restrict__ x_idx, int32_t* __restrict__ y_idx) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    const float x = pos[e], y = pos[stride + e];
    int xi, yi;
    map_id(m, x, y, xi, yi);
    x_idx[e] = xi;
    y_idx[e] = yi;
    terms[e] = m.map[yi * m.cols + xi] ? 1.f : -1.f;
    terms[stride + e] = vb[e];
    oom[e] = out_of_map(m, x, y) ? 1 : 0;
}

int check_visual(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m) {
    if (!p || !b || !m || !b->state || !b->episode_len || !b->metrics || !m->map || !m->cells) return WL_EINVAL;
    if (b->n_envs <= 0 || b->stride < b->n_envs || b->metrics_slots < 1 || m->n_cells <= 0) return WL_EINVAL;
    if (b->stride % 64 != 0 || ((uintptr_t)b->state & 15u)) return WL_EALIGN;
    if (b->stride * 4 * WL_S_COUNT > 0x7fffffffLL || (b->lanes != 0 && b->lanes != 1 && b->lanes != 4)) return WL_EINVAL;
    if (!flags_ok(b)) return WL_EINVAL;
    if (p->decimation <= 0 || p->vehicle.substeps <= 0 || !(p->sim_dt > 0.f) || m->rows <= 0 || m->cols <= 0) return WL_EINVAL;
    if (p->vehicle.implicit != 1 || !(p->vehicle.susp_fmax > 0.f)) return WL_EINVAL;   // these kernels step the linearly implicit integrator (wl_vehicle.h)
    return WL_OK;
}

}  // namespace

extern "C" {

int wl_visual_step(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const float* actions,
                   const WlStepOut* out, uint64_t seed, uint64_t step, void* stream) {
    return wl_visual_rollout(p, b, m, actions, out, 0, 0, 1, seed, step, stream);
}

int wl_visual_rollout(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const float* actions,
                      const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps, uint64_t seed,
                      uint64_t step0, void* stream) {
    int rc = check_visual(p, b, m);
    if (rc != WL_OK) return rc;
    if (!actions || !out || !out->obs || !out->reward || !out->terminated || !out->truncated || n_steps < 0) return WL_EINVAL;
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
    const bool quad = use_quad(b);
    clear_error();
    for (int k = 0; k < n_steps; ++k) {
        WlStepOut o = *out;
        o.obs += k * obs_step_stride;
        o.reward += k * vec_step_stride;
        o.terminated += k * vec_step_stride;
        o.truncated += k * vec_step_stride;
        if (o.dones) o.dones += k * vec_step_stride;
        const float2* a = (const float2*)(actions + (int64_t)k * b->n_envs * 2);
        const uint64_t st = step0 + (uint64_t)k;
        const hipStream_t hs = (hipStream_t)stream;
        const int n = b->n_envs;
        if (quad)
        {
            const int lanes = n * 4;
            if (n <= 2048) visual_step_kernel<4, 64><<<(lanes + 63) / 64, 64, 0, hs>>>(*p, vd, *b, *m, a, o, seed, st);
            else if (n <= 8192) visual_step_kernel<4, 128><<<(lanes + 127) / 128, 128, 0, hs>>>(*p, vd, *b, *m, a, o, seed, st);
            else visual_step_kernel<4><<<grid_for(lanes), kBlock, 0, hs>>>(*p, vd, *b, *m, a, o, seed, st);
        }
        else
            visual_step_kernel<1><<<grid_for(n), kBlock, 0, hs>>>(*p, vd, *b, *m, a, o, seed, st);
        launch_visual_obs(p, b, m, o.obs, hs);
    }
    return launch_status();
}

int wl_visual_rollout_persistent(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const float* actions,
                                 const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps,
                                 uint64_t seed, uint64_t step0, void* stream) {
    int rc = check_visual(p, b, m);
    if (rc != WL_OK) return rc;
    if (!use_quad(b)) return WL_EINVAL;   // the quad form's (n <= 32 768)
    if (!actions || !out || !out->obs || !out->reward || !out->terminated || !out->truncated || n_steps < 0) return WL_EINVAL;
    if (n_steps > 1 && obs_step_stride < (int64_t)b->n_envs * WL_VIS_OBS_DIM) return WL_EINVAL;   // the camera runs a step behind: rows must differ
    if (b->metrics_slots > 1 && n_steps % b->metrics_slots == 0 && n_steps > 0) return WL_EINVAL;   // ring slot aliasing
    clear_error();
    // envs per block: one round of blocks on the 256 CUs (a block's thirteen-plus wavefronts at 128 VGPRs fill a CU)
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
    const MetricSlots ms = metric_slots(b, step0, (uint64_t)n_steps);
    const int n = b->n_envs;
#define WL_VIS_PERSIST(E)                                                                                                   \
    if (lds_map_ok(m))                                                                                                      \
        visual_rollout_persistent_kernel<E, true><<<(n + (E) - 1) / (E), kPersistThreads, 0, (hipStream_t)stream>>>(        \
            *p, vd, *b, *m, (const float2*)actions, *out, obs_step_stride, vec_step_stride, n_steps, seed, step0, ms);      \
    else                                                                                                                    \
        visual_rollout_persistent_kernel<E, false><<<(n + (E) - 1) / (E), kPersistThreads, 0, (hipStream_t)stream>>>(       \
            *p, vd, *b, *m, (const float2*)actions, *out, obs_step_stride, vec_step_stride, n_steps, seed, step0, ms)
    if (n <= 1024) { WL_VIS_PERSIST(4); }
    else if (n <= 2048) { WL_VIS_PERSIST(8); }
    else { WL_VIS_PERSIST(16); }
#undef WL_VIS_PERSIST
    return launch_status();
}

int wl_visual_reset(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const uint8_t* mask, uint64_t seed,
                    uint64_t step, void* stream) {
    int rc = check_visual(p, b, m);
    if (rc != WL_OK) return rc;
    clear_error();
    visual_reset_kernel<FlatGround><<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, *b, *m, mask, seed, step, FlatGround{});
    return launch_status();
}

int wl_visual_observe(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, float* obs, void* stream) {
    int rc = check_visual(p, b, m);
    if (rc != WL_OK) return rc;
    if (!obs) return WL_EINVAL;
    clear_error();
    launch_visual_obs(p, b, m, obs, (hipStream_t)stream);
    return launch_status();
}

/* ---- the visual-depth extension task (BASELINE config 5): the visual task's step on a heightfield terrain; the observation row is
 * [ depth image 4800 | base_lin_vel 3 | base_ang_vel 3 | last_action 2 ]: this launch writes the last 8, wl_depth.hip the image ---- */
int wl_visual_step_hf(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const WlHeightField* hf, const float* actions,
                      const WlStepOut* out, uint64_t seed, uint64_t step, void* stream) {
    int rc = check_visual(p, b, m);
    if (rc != WL_OK) return rc;
    if (!hf || !hf->height || !hf->pair || hf->nx < 2 || hf->ny < 2 || !(hf->cell > 0.f) || !(hf->z_scale > 0.f && hf->z_scale < INFINITY)) return WL_EINVAL;   // pair: the contact sampler's table (wl_heightfield_pairs)
    if (!actions || !out || !out->obs || !out->reward || !out->terminated || !out->truncated) return WL_EINVAL;
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
    const HeightFieldGround g = make_ground(hf);
    const float2* a = (const float2*)actions;
    const hipStream_t hs = (hipStream_t)stream;
    const int n = b->n_envs;
    clear_error();
    if (use_quad(b)) {
        const int lanes = n * 4;
        if (n <= 8192) visual_step_kernel<4, 128, HeightFieldGround><<<(lanes + 127) / 128, 128 + 64, 0, hs>>>(*p, vd, *b, *m, a, *out, seed, step, g, WL_VISDEPTH_OBS_DIM, WL_VISDEPTH_NPIX);   // + the helper wavefront
        else visual_step_kernel<4, kBlock, HeightFieldGround><<<grid_for(lanes), kBlock, 0, hs>>>(*p, vd, *b, *m, a, *out, seed, step, g, WL_VISDEPTH_OBS_DIM, WL_VISDEPTH_NPIX);
    } else {
        visual_step_kernel<1, kBlock, HeightFieldGround><<<grid_for(n), kBlock, 0, hs>>>(*p, vd, *b, *m, a, *out, seed, step, g, WL_VISDEPTH_OBS_DIM, WL_VISDEPTH_NPIX);
    }
    return launch_status();
}

int wl_visual_reset_hf(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const WlHeightField* hf, const uint8_t* mask,
                       uint64_t seed, uint64_t step, void* stream) {
    int rc = check_visual(p, b, m);
    if (rc != WL_OK) return rc;
    if (!hf || !hf->height || !hf->pair || hf->nx < 2 || hf->ny < 2 || !(hf->cell > 0.f) || !(hf->z_scale > 0.f && hf->z_scale < INFINITY)) return WL_EINVAL;   // pair: the contact sampler's table (wl_heightfield_pairs)
    clear_error();
    visual_reset_kernel<HeightFieldGround><<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, *b, *m, mask, seed, step, make_ground(hf));
    return launch_status();
}

int wl_visual_mdp(const WlVisualParams* p, const WlTravMap* m, int32_t n, int64_t stride, const float* pos,
                  const float* lin_vel_b, float* terms, uint8_t* out_of_map_, int32_t* x_idx, int32_t* y_idx, void* stream) {
    if (!p || !m || !m->map || n <= 0 || stride < n || !pos || !lin_vel_b || !terms || !out_of_map_ || !x_idx || !y_idx)
        return WL_EINVAL;
    clear_error();
    visual_mdp_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(*p, *m, n, stride, pos, lin_vel_b, terms, out_of_map_,
                                                                        x_idx, y_idx);
    return launch_status();
}

}  // extern "C"

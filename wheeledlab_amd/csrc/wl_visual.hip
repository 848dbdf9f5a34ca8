// wl_visual.hip -- visual task (wheeledlab_tasks/visual/mushr_visual_env_cfg.py) for gfx950.
//
// Two launches per env.step():
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

template <int LANES, int QB = kBlock /* quad form: threads per block (see drift_step_kernel) */>
__global__ void __launch_bounds__(kBlock) visual_step_kernel(const WlVisualParams p_arg, const VehDerived vd_arg, const WlEnvBuffers b,
                                                             const WlTravMap m, const float2* __restrict__ actions,
                                                             const WlStepOut out, const uint64_t seed, const uint64_t step) {
    __shared__ float blk_metrics[WL_M_COUNT];
    WlVisualParams p = p_arg;
    VehDerived vd = vd_arg;
    if constexpr (LANES == 4) {   // latency form: one batch of vector loads instead of dependent scalar-load round trips
        kernarg_vector_copy2(0, p, vd);   // both argument structs as ONE burst (two copies: a wait in the middle, see the helper)
        keep_scalar_common(p, p_arg);
        vd.n_sub = vd_arg.n_sub;
    }
    constexpr int kEnvs = (LANES == 4 ? QB : kBlock) / LANES;
    const int wid = LANES == 1 ? 0 : (threadIdx.x & 3);
    const bool lead = LANES == 1 || wid == 0;
    const int e = blockIdx.x * kEnvs + threadIdx.x / LANES;
    if (threadIdx.x < WL_M_COUNT) blk_metrics[threadIdx.x] = 0.f;
    const int m_slot = b.metrics_slots > 1 ? (int)(step % (uint64_t)b.metrics_slots) : 0;
    if (b.metrics_slots > 1) clear_metric_slot(b, (m_slot + 1) % b.metrics_slots);
    __syncthreads();
    const Rows S = make_rows(b.state, b.stride);
    const WlVehicleParams& vp = p.vehicle;
    if (e < b.n_envs) {
        const uint32_t gid = (uint32_t)(b.env_offset + e);
        float2 a = actions[e];
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
        const FlatGround ground{};
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
        vehicle_integrate<LANES>(vp, vd, ec, s, ground, wid);
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
            const VisReset rd = draw_visual_reset(p, m, gid, step, seed);
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
        }
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

WL_DEV float block_sum(float v, float* scratch /* [kBlock/64] */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) t += scratch[w];
    return t;
}

WL_DEV int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// VisualObsCfg.PolicyCfg (:38-58): camera (3200) | base_lin_vel (3) | base_ang_vel (3) | last_action clip +-1 (2)
__global__ void __launch_bounds__(kBlock) visual_obs_kernel(const WlVisualParams p, const WlEnvBuffers b, const WlTravMap m,
                                                            float* __restrict__ obs) {
    // the rendered image with its two reflected border columns on either side: pitch 84 floats = 21 sixteen-byte slots, so
    // the 8-float window of a 4 x 4 output patch is two aligned ds_read_b128 and, 4 rows x 21 slots being 4 mod 16, the 16-lane
    // groups a b128 read is served in (lanes of one patch row + 8 lanes of the next) hit 16 distinct slots.  (With the
    // unpadded [40][80] image each of the 64 reads per thread was a 4-byte read at a lane stride of 4 floats: 8-way conflicts.)
    constexpr int kPitch = kImgW + 4;
    __shared__ __attribute__((aligned(16))) float img[kImgH * kPitch];
    __shared__ float red[kBlock / 64];
    const int e = blockIdx.x;
    const Rows S = make_rows(b.state, b.stride);
    const V3 pos = ld3(S, WL_S_PX, e);
    const Quat q{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
    const Mat3 R = mat_from_quat(q);
    const V3 o = pos + mul(R, v3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]));
    const MapFast mf = map_fast(m);
    const float inv_fx = rcp(p.fx), inv_fy = rcp(p.fy);
    const bool plain = p.contrast == 1.f && !(p.blur_sigma > 0.f);   // no augmentation that needs the whole image
    float* row = obs + (int64_t)e * WL_VIS_OBS_DIM;
    if (threadIdx.x == kBlock - 1) {   // proprioception: the last lane (its wave renders the fewest pixels)
        const V3 vb = mul_t(R, ld3(S, WL_S_VX, e)), wb = mul_t(R, ld3(S, WL_S_WX, e));
        float* t = row + WL_VIS_NPIX;
        t[0] = vb.x; t[1] = vb.y; t[2] = vb.z;
        t[3] = wb.x; t[4] = wb.y; t[5] = wb.z;
        t[6] = clampf(S.ld(WL_S_ACT0, e), -1.f, 1.f);
        t[7] = clampf(S.ld(WL_S_ACT1, e), -1.f, 1.f);
    }
    // ---- render: one ray per pixel against the z = 0 plane.  d = R (1, dy(col), dz(row)) ----
    const V3 c0 = v3(R.r0.x, R.r1.x, R.r2.x), c1 = v3(R.r0.y, R.r1.y, R.r2.y), c2 = v3(R.r0.z, R.r1.z, R.r2.z);
    float part = 0.f;
    // Thread -> one image column and every third row (240 of the 256 threads: 3 rows x 80 columns per pass, 14 passes):
    // the ray direction then advances by a constant vector per pass (3 adds instead of 2 conversions + 8 fma), and the
    // lanes of a wavefront still store consecutive columns.  Software-pipelined like the height scan: every pixel of
    // this lane computes its map cell and issues its byte gather first (14 in flight per lane), then all are shaded and
    // stored -- rolled, each pixel paid the gather latency in turn.
    constexpr int kRowsPerPass = kBlock / kImgW, kIter = (kImgH + kRowsPerPass - 1) / kRowsPerPass;
    const int tc = (int)threadIdx.x % kImgW, tr = (int)threadIdx.x / kImgW;
    const bool lane_on = tr < kRowsPerPass;
    const float dy = -(((float)tc + 0.5f - p.cx) * inv_fx), dz0 = -(((float)(tr + WL_VIS_CROP) + 0.5f - p.cy) * inv_fy);
    V3 d = fma3(dz0, c2, fma3(dy, c1, c0));
    const V3 dstep = (-(float)kRowsPerPass * inv_fy) * c2;
    const float mx = mf.off_x * mf.inv_rs, my = mf.off_y * mf.inv_cs;   // cell = (int)(h * inv + off * inv)
    uint8_t cell[kIter];
    bool hit[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const float t = -o.z * rcp(fminf(d.z, -1e-6f));
        const float hx = fmaf(t, d.x, o.x), hy = fmaf(t, d.y, o.y);
        hit[it] = d.z < -1e-6f;                                         // else: sky
        const bool on_map = hit[it] && fabsf(hx) <= mf.half_w && fabsf(hy) <= mf.half_h;
        const int xi = min(max((int)fmaf(hx, mf.inv_rs, mx), 0), m.rows - 1);
        const int yi = min(max((int)fmaf(hy, mf.inv_cs, my), 0), m.cols - 1);
        cell[it] = on_map ? m.map[yi * m.cols + xi] : (uint8_t)0;       // white path on black (utils/__init__.py:47-50)
        d = d + dstep;
    }
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int r = tr + it * kRowsPerPass;
        if (lane_on && r < kImgH) {
            float v = hit[it] ? (cell[it] != 0 ? 1.f : 0.f) : p.sky;
            v = clampf(v * p.brightness, 0.f, 1.f);                    // ColorJitter brightness
            if (plain) {
                row[r * kImgW + tc] = (v * 0.9999f - 0.5f) * 2.f;      // grayscale + Normalize([0.5], [0.5]) straight to HBM
            } else {
                float* line = img + r * kPitch + 2;
                line[tc] = v;
                if (tc == 1 || tc == 2) line[-tc] = v;                            // reflect padding (torchvision): -1 -> 1, -2 -> 2
                if (tc == kImgW - 2 || tc == kImgW - 3) line[2 * kImgW - 2 - tc] = v;   // 80 -> 78, 81 -> 77
                part += v;
            }
        }
    }
    if (plain) return;
    // ---- augmentation on the LDS-resident image.  One 4 x 4 output patch per thread (10 x 20 patches = 200 threads):
    // the 8 x 8 input neighbourhood is read once (contrast blend applied on the fly), both passes of the separable 5-tap
    // Gaussian run in registers, and each patch row leaves as one 16-byte store.
    const float mean = 0.9999f * block_sum(part, red) * (1.f / (float)(kImgH * kImgW));   // also orders the img writes
    const float cc = p.contrast, cm = (1.f - p.contrast) * mean;   // ColorJitter contrast: blend with the grey mean
    // a blend TOWARDS the mean (contrast <= 1) stays inside [0, 1], its clamp is the identity, and the blur is linear with
    // weights summing to one: blur(cc v + cm) = cc blur(v) + cm -- applied to the 16 outputs instead of the 64 inputs
    const bool fold = cc <= 1.f && cc >= 0.f;
    const float oc_ = fold ? cc : 1.f, om_ = fold ? cm : 0.f;
    float w[5] = {0.f, 0.f, 1.f, 0.f, 0.f};
    if (p.blur_sigma > 0.f) {                                      // GaussianBlur(5, sigma): torchvision's kernel1d
        float wsum = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float x = (float)(j - 2) / p.blur_sigma;
            w[j] = __expf(-0.5f * x * x);
            wsum += w[j];
        }
        const float inv = 1.f / wsum;
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] *= inv;
    }
    constexpr int kPatchCols = kImgW / 4, kPatches = (kImgH / 4) * kPatchCols;
    if (threadIdx.x < kPatches) {
        const int pr = (threadIdx.x / kPatchCols) * 4, pc = (threadIdx.x % kPatchCols) * 4;
        float hrow[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4* line = reinterpret_cast<const float4*>(img + reflect(pr + i - 2, kImgH) * kPitch + pc);   // columns pc - 2 .. pc + 5
            const float4 lo4 = line[0], hi4 = line[1];
            float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            if (!fold) {   // contrast > 1 can leave [0, 1]: the clamp sits between the blend and the blur
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = clampf(fmaf(cc, v[j], cm), 0.f, 1.f);
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
            *reinterpret_cast<float4*>(row + (pr + orow) * kImgW + pc) = out4;
        }
    }
}

__global__ void __launch_bounds__(kBlock) visual_reset_kernel(const WlVisualParams p, const WlEnvBuffers b, const WlTravMap m,
                                                              const uint8_t* __restrict__ mask, uint64_t seed, uint64_t step) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= b.n_envs) return;
    if (mask && !mask[e]) return;
    const Rows S = make_rows(b.state, b.stride);
    const VisReset rd = draw_visual_reset(p, m, (uint32_t)(b.env_offset + e), step, seed);
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

// depth extension: distance_to_image_plane of the camera against a heightfield (march 0.05 m steps along the optical
// axis, then 6 bisection steps); rays that leave the grid hit the z = outside_z plane
__global__ void __launch_bounds__(kBlock) visual_depth_kernel(const WlVisualParams p, const WlEnvBuffers b, const HeightFieldGround g,
                                                              float max_depth, float* __restrict__ depth) {
    const int e = blockIdx.x;
    const Rows S = make_rows(b.state, b.stride);
    const V3 pos = ld3(S, WL_S_PX, e);
    const Quat q{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
    const Mat3 R = mat_from_quat(q);
    const V3 o = pos + mul(R, v3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]));
    for (int k = threadIdx.x; k < WL_VIS_IMG_H * WL_VIS_IMG_W; k += kBlock) {
        const int r = k / WL_VIS_IMG_W, c = k - r * WL_VIS_IMG_W;
        const V3 d = mul(R, pixel_ray_body(p, r, c));   // |d.x_body| = 1: the ray parameter IS the image-plane distance
        float t0 = 0.f, t1 = 0.f, res = max_depth;
        bool found = false;
        for (float t = 0.05f; t <= max_depth; t += 0.05f) {
            float z;
            V3 n;
            g.sample(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), z, n);
            if (fmaf(t, d.z, o.z) <= z) {
                t0 = t - 0.05f;
                t1 = t;
                found = true;
                break;
            }
        }
        if (found) {
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const float tm = 0.5f * (t0 + t1);
                float z;
                V3 n;
                g.sample(fmaf(tm, d.x, o.x), fmaf(tm, d.y, o.y), z, n);
                if (fmaf(tm, d.z, o.z) <= z) t1 = tm; else t0 = tm;
            }
            res = 0.5f * (t0 + t1);
        }
        depth[(int64_t)e * WL_VIS_IMG_H * WL_VIS_IMG_W + k] = res;
    }
}

int check_visual(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m) {
    if (!p || !b || !m || !b->state || !b->episode_len || !b->metrics || !m->map || !m->cells) return WL_EINVAL;
    if (b->n_envs <= 0 || b->stride < b->n_envs || b->metrics_slots < 1 || m->n_cells <= 0) return WL_EINVAL;
    if (b->stride % 64 != 0 || ((uintptr_t)b->state & 15u)) return WL_EALIGN;
    if (b->stride * 4 * WL_S_COUNT > 0x7fffffffLL || (b->lanes != 0 && b->lanes != 1 && b->lanes != 4)) return WL_EINVAL;
    if (p->decimation <= 0 || p->vehicle.substeps <= 0 || !(p->sim_dt > 0.f) || m->rows <= 0 || m->cols <= 0) return WL_EINVAL;
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
        if (quad)
        {
            const int lanes = b->n_envs * 4;
            if (b->n_envs <= 2048) visual_step_kernel<4, 64><<<(lanes + 63) / 64, 64, 0, (hipStream_t)stream>>>(*p, vd, *b, *m, a, o, seed, step0 + (uint64_t)k);
            else if (b->n_envs <= 8192) visual_step_kernel<4, 128><<<(lanes + 127) / 128, 128, 0, (hipStream_t)stream>>>(*p, vd, *b, *m, a, o, seed, step0 + (uint64_t)k);
            else visual_step_kernel<4><<<grid_for(lanes), kBlock, 0, (hipStream_t)stream>>>(*p, vd, *b, *m, a, o, seed, step0 + (uint64_t)k);
        }
        else
            visual_step_kernel<1><<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, vd, *b, *m, a, o, seed, step0 + (uint64_t)k);
        visual_obs_kernel<<<b->n_envs, kBlock, 0, (hipStream_t)stream>>>(*p, *b, *m, o.obs);
    }
    return launch_status();
}

int wl_visual_reset(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const uint8_t* mask, uint64_t seed,
                    uint64_t step, void* stream) {
    int rc = check_visual(p, b, m);
    if (rc != WL_OK) return rc;
    clear_error();
    visual_reset_kernel<<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, *b, *m, mask, seed, step);
    return launch_status();
}

int wl_visual_observe(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, float* obs, void* stream) {
    int rc = check_visual(p, b, m);
    if (rc != WL_OK) return rc;
    if (!obs) return WL_EINVAL;
    clear_error();
    visual_obs_kernel<<<b->n_envs, kBlock, 0, (hipStream_t)stream>>>(*p, *b, *m, obs);
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

int wl_visual_depth(const WlVisualParams* p, const WlEnvBuffers* b, const WlHeightField* hf, float max_depth, float* depth,
                    void* stream) {
    if (!p || !b || !hf || !b->state || !hf->height || !depth || b->n_envs <= 0 || !(max_depth > 0.f)) return WL_EINVAL;
    clear_error();
    visual_depth_kernel<<<b->n_envs, kBlock, 0, (hipStream_t)stream>>>(*p, *b, make_ground(hf), max_depth, depth);
    return launch_status();
}

}  // extern "C"

// wl_drift_env.h -- device code of ONE drift-task env.step() on a register image of the env's state rows: reset draw,
// observation noise / layout, row load / store, episode-metric sink and drift_env_step<LANES, Ground> itself.  Shared by
// the step / rollout kernels (wl_drift.hip) and the policy-in-the-loop rollout kernel (wl_policy.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_kernel_common.h"
#include "wl_drift_terms.h"
#include "wl_rng.h"
#include "wl_vehicle.h"

namespace {

#ifndef WL_MIN_WAVES
#define WL_MIN_WAVES 2   // __launch_bounds__ 2nd argument = waves per SIMD the register allocator must leave room for
#endif
constexpr int kObsDim = 14;
constexpr int kObsPad = 15;  // odd LDS row pitch: the transposing writes are bank-conflict free

struct ResetDraw {
    V3 pos;
    Quat q;
    float yaw;   // the drawn yaw (rad, unwrapped): roll = pitch = 0, so the reset pose's Euler angles are (0, 0, yaw)
    float timer_hf, timer_lf;
};

// The drift step's draws, three Philox blocks of eight 16-bit uniforms each (wl_rng.h; round 3: six blocks of four 24-bit ones):
//   WL_RS_DRIFT_EVENTS  x: reference-pose index | x offset   y: y offset | yaw offset   z: re-armed hf | lf timer   w: lf push yaw kick | lf timer
//   WL_RS_NOISE0        x y z w: observation normals 0..7 (one Box-Muller pair per word: radius from the low half, angle from the high half)
//   WL_RS_NOISE1        x y: normals 8..11                                                 z: hf push dvx | dvy   w: hf push yaw kick | hf timer
// reset_root_state_along_track.__call__ (drifting/mdp/events.py:119-133) + EventManager.reset interval re-arm, from the event block's
// words x, y, z
WL_DEV ResetDraw reset_from_uniforms(const WlDriftParams& p, const float* __restrict__ ref, F4 u /* halves of x, y */, float t_hf, float t_lf) {
    const int idx = min((int)(u.x * (float)p.num_ref_points), p.num_ref_points - 1);
    ResetDraw r;
    r.pos = v3(fmaf(2.f * u.y - 1.f, p.pos_noise, ref[idx]), fmaf(2.f * u.z - 1.f, p.pos_noise, ref[32 + idx]), 0.f);
    const float yaw = fmaf(2.f * u.w - 1.f, p.yaw_noise, ref[64 + idx]);
    float s, c;
    sincos_fast(0.5f * yaw, s, c);
    r.q = Quat{c, 0.f, 0.f, s};
    r.yaw = yaw;
    r.timer_hf = fmaf(t_hf, p.hf_interval[1] - p.hf_interval[0], p.hf_interval[0]);
    r.timer_lf = fmaf(t_lf, p.lf_interval[1] - p.lf_interval[0], p.lf_interval[0]);
    return r;
}
WL_DEV ResetDraw reset_from_block(const WlDriftParams& p, const float* __restrict__ ref, const U4& ev) {
    return reset_from_uniforms(p, ref, u16x4(ev.x, ev.y), u16_lo(ev.z), u16_hi(ev.z));
}
WL_DEV ResetDraw draw_reset(const WlDriftParams& p, const float* __restrict__ ref, uint32_t gid, uint64_t step, uint64_t seed) {
    return reset_from_block(p, ref, philox_block(gid, step, WL_RS_DRIFT_EVENTS, seed));
}

// BlindObsCfg.PolicyCfg (wheeledlab_tasks/common/observations.py:24-54) into this wave's LDS tile
// quad form: the 14 observation values are replicated on the four lanes of the env's quad, so each lane stores its
// quarter as 8-byte words -- a wavefront (16 envs) writes one contiguous 896-byte run, no LDS transpose needed
WL_DEV void store_obs_quad(float* __restrict__ row /* obs + e * 14 */, int wid, const float o[14]) {
    float2* r2 = reinterpret_cast<float2*>(row);   // 56 B per env: 8-byte aligned
    r2[wid * 2] = make_float2(quad_pick(wid, o[0], o[4], o[8], o[12]), quad_pick(wid, o[1], o[5], o[9], o[13]));
    const float2 c = make_float2(quad_pick(wid, o[2], o[6], o[10], 0.f), quad_pick(wid, o[3], o[7], o[11], 0.f));
    if (wid < 3) r2[wid * 2 + 1] = c;
}

struct Noise12 {
    float z[12];
};

WL_DEV void obs_values(float o14[14], const WlDriftParams& p, V3 pos, V3 e, V3 vb, V3 wb, float a0, float a1, const Noise12& nz) {
    const float o[12] = {pos.x, pos.y, pos.z, e.x, e.y, e.z, vb.x, vb.y, vb.z, wb.x, wb.y, wb.z};
#pragma unroll
    for (int k = 0; k < 12; ++k) o14[k] = fmaf(p.noise_std[k / 3], nz.z[k], o[k]);
    o14[12] = clampf(a0, -1.f, 1.f);
    o14[13] = clampf(a1, -1.f, 1.f);
}

WL_DEV void write_obs_row(float* row, const WlDriftParams& p, V3 pos, V3 e /* euler xyz, wrapped */, V3 vb, V3 wb, float a0,
                          float a1, const Noise12& nz /* 12 standard normals (zeros when corruption is off) */) {
    const float o[12] = {pos.x, pos.y, pos.z, e.x, e.y, e.z, vb.x, vb.y, vb.z, wb.x, wb.y, wb.z};
#pragma unroll
    for (int k = 0; k < 12; ++k) row[k] = fmaf(p.noise_std[k / 3], nz.z[k], o[k]);
    row[12] = clampf(a0, -1.f, 1.f);
    row[13] = clampf(a1, -1.f, 1.f);
}

// quad form: lanes 0, 3, 1 hold normals 0..3, 4..7, 8..11 (draw_step_raw); every lane of the quad gets all twelve via DPP quad broadcasts
WL_DEV Noise12 gather_quad_noise(const float z[4]) {
    Noise12 nz;
    nz.z[0] = quad_bcast<0>(z[0]); nz.z[1] = quad_bcast<0>(z[1]); nz.z[2] = quad_bcast<0>(z[2]); nz.z[3] = quad_bcast<0>(z[3]);
    nz.z[4] = quad_bcast<3>(z[0]); nz.z[5] = quad_bcast<3>(z[1]); nz.z[6] = quad_bcast<3>(z[2]); nz.z[7] = quad_bcast<3>(z[3]);
    nz.z[8] = quad_bcast<1>(z[0]); nz.z[9] = quad_bcast<1>(z[1]); nz.z[10] = quad_bcast<1>(z[2]); nz.z[11] = quad_bcast<1>(z[3]);
    return nz;
}

// one word -> two standard normals
WL_DEV void normals_of_word(uint32_t w, float& z0, float& z1) { box_muller_open(u16_lo(w), u16_hi(w), z0, z1); }

// observation noise: off, the caller's parity tensor, or (lane form) two Philox blocks -> 12 normals; `nb` = the WL_RS_NOISE1 block when
// the caller has drawn it already (the step draws it once for the hf push and the noise).  The quad form draws its normals in
// draw_step (one block per lane) and gathers them with gather_quad_noise.
template <int LANES>
WL_DEV Noise12 obs_noise(const WlDriftParams& p, const float* __restrict__ noise, int64_t stride, int e, uint32_t gid,
                         uint64_t step, uint64_t seed, const U4* nb = nullptr) {
    Noise12 nz;
    if (!p.enable_corruption) {
#pragma unroll
        for (int k = 0; k < 12; ++k) nz.z[k] = 0.f;
    } else if (noise) {   // parity mode: caller-supplied standard normals [12][stride]
#pragma unroll
        for (int k = 0; k < 12; ++k) nz.z[k] = noise[k * stride + e];
    } else {
        const U4 na = philox_block(gid, step, WL_RS_NOISE0, seed);
        const U4 n1 = nb ? *nb : philox_block(gid, step, WL_RS_NOISE1, seed);
        normals_of_word(na.x, nz.z[0], nz.z[1]);
        normals_of_word(na.y, nz.z[2], nz.z[3]);
        normals_of_word(na.z, nz.z[4], nz.z[5]);
        normals_of_word(na.w, nz.z[6], nz.z[7]);
        normals_of_word(n1.x, nz.z[8], nz.z[9]);
        normals_of_word(n1.y, nz.z[10], nz.z[11]);
    }
    return nz;
}

// Everything random about one env-step that does not depend on the env's state: (seed, global env id, step) key it all.
// Quad form: drawn at the top of the step -- in the per-step kernel that is the shadow of the cold-L2 load of the state
// matrix -- instead of lazily in the reset / push branches on the tail, where each draw is a dependent chain of ~100
// instructions on the kernel's critical path (the kernel ends with its slowest wavefront, and with 16 envs per
// wavefront nearly every step some wavefront resets and most have a push).  The draws are SPREAD over the quad: lane w
// computes ONE Philox block -- lane 0 WL_RS_NOISE0, lane 1 WL_RS_NOISE1, lane 2 WL_RS_DRIFT_EVENTS, lane 3 WL_RS_NOISE0 again (its
// words z, w become normals 4..7) -- where round 3 had two per lane and round 1 five, and the branch that needs an event pulls it
// from its lane with DPP quad broadcasts (the branch conditions are functions of the replicated env state, so a quad is always
// wholly inside or outside a branch).
struct StepDraws {
    F4 e0, e1;           // the halves of this lane's words (x, y) and (z, w) as uniforms: lane 1's e1 = hf push, lane 2's e0 / e1 = reset / timers + lf push
    float ref[3];        // the reference pose (x, y, yaw) lane 2's reset would start from (ref_pose_request)
    float z[4];          // this lane's four observation-noise normals (lanes 0, 3, 1 of the quad; see gather_quad_noise)
};

// first half: needs nothing but the key (preloaded kernel arguments): it runs while BOTH the state rows and the parameter
// block are still in flight.  (Round 5 probe: with this function returning constants -- no Philox block, no Box-Muller -- the 4096-env
// launch takes 5.95 - 5.98 us against 6.05: where they stand the draws cost <= 0.1 us, so handing them to a sibling wavefront
// through LDS + an s_barrier has nothing to win.)
WL_DEV StepDraws draw_step_raw(uint32_t gid, uint64_t step, uint64_t seed, int wid) {
    StepDraws d;
    const uint32_t stream = wid == 1 ? (uint32_t)WL_RS_NOISE1 : wid == 2 ? (uint32_t)WL_RS_DRIFT_EVENTS : (uint32_t)WL_RS_NOISE0;
    const U4 w = philox_block(gid, step, stream, seed);
    d.e0 = u16x4(w.x, w.y);
    d.e1 = u16x4(w.z, w.w);
    const bool hi = wid == 3;
    box_muller_open(hi ? d.e1.x : d.e0.x, hi ? d.e1.y : d.e0.y, d.z[0], d.z[1]);
    box_muller_open(hi ? d.e1.z : d.e0.z, hi ? d.e1.w : d.e0.w, d.z[2], d.z[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d.z[k] = opaque(d.z[k]);   // computed HERE, not sunk to the observation code on the tail
    d.e0.x = opaque(d.e0.x), d.e0.y = opaque(d.e0.y), d.e0.z = opaque(d.e0.z), d.e0.w = opaque(d.e0.w);
    d.e1.x = opaque(d.e1.x), d.e1.y = opaque(d.e1.y), d.e1.z = opaque(d.e1.z), d.e1.w = opaque(d.e1.w);
    d.ref[0] = d.ref[1] = d.ref[2] = 0.f;
    return d;
}
// the reference-pose lookup is a dependent memory round trip: requested as soon as the pose count is known, consumed
// only inside the reset branch (every lane looks up ITS block's index; lane 2's is the one that counts)
WL_DEV void ref_pose_request(StepDraws& d, const float* __restrict__ ref, int num_ref) {
    const int idx = min((int)(d.e0.x * (float)num_ref), num_ref - 1);
    d.ref[0] = ref[idx], d.ref[1] = ref[32 + idx], d.ref[2] = ref[64 + idx];
}
WL_DEV StepDraws draw_step(const WlDriftParams& p, const float* __restrict__ ref, uint32_t gid, uint64_t step, uint64_t seed,
                           int wid) {
    StepDraws d = draw_step_raw(gid, step, seed, wid);
    ref_pose_request(d, ref, p.num_ref_points);
    return d;
}
// the events of the quad, pulled from the lane that drew them (called inside the reset branch: a quad is always wholly
// inside or outside it, the branch conditions being functions of the replicated env state)
WL_DEV ResetDraw quad_reset_draw(const WlDriftParams& p, const StepDraws& d) {
    ResetDraw r;
    // reset_root_state_along_track.__call__ (drifting/mdp/events.py:119-133) on lane 2's block (same arithmetic as reset_from_uniforms)
    r.pos = v3(quad_bcast<2>(fmaf(2.f * d.e0.y - 1.f, p.pos_noise, d.ref[0])), quad_bcast<2>(fmaf(2.f * d.e0.z - 1.f, p.pos_noise, d.ref[1])), 0.f);
    r.yaw = quad_bcast<2>(fmaf(2.f * d.e0.w - 1.f, p.yaw_noise, d.ref[2]));
    float s, c;
    sincos_fast(0.5f * r.yaw, s, c);
    r.q = Quat{c, 0.f, 0.f, s};
    r.timer_hf = quad_bcast<2>(fmaf(d.e1.x, p.hf_interval[1] - p.hf_interval[0], p.hf_interval[0]));
    r.timer_lf = quad_bcast<2>(fmaf(d.e1.y, p.lf_interval[1] - p.lf_interval[0], p.lf_interval[0]));
    return r;
}
// the hf push's four uniforms (lane 1, words z w) / the lf push's two (lane 2, word w)
WL_DEV F4 quad_hf_draw(const StepDraws& d) { return F4{quad_bcast<1>(d.e1.x), quad_bcast<1>(d.e1.y), quad_bcast<1>(d.e1.z), quad_bcast<1>(d.e1.w)}; }
WL_DEV void quad_lf_draw(const StepDraws& d, float& kick, float& timer) { kick = quad_bcast<2>(d.e1.z), timer = quad_bcast<2>(d.e1.w); }

// flush a tile of `n_slots` obs rows ([slot][kObsPad], row-padded) to obs[n][14]: contiguous dword stores
WL_DEV void flush_obs(const float* tile, float* __restrict__ obs, int block_env0, int n, int envs_per_block = kBlock) {
    const int n_valid = min(envs_per_block, n - block_env0);
    const int total = n_valid * kObsDim;
    float* dst = obs + (int64_t)block_env0 * kObsDim;
    for (int f = threadIdx.x; f < total; f += kBlock) {
        const int e = f / kObsDim, k = f - e * kObsDim;
        dst[f] = tile[e * kObsPad + k];
    }
}
// (Round 4: the same rows as 16-byte words -- 3.5 store instructions per lane instead of 14, four LDS reads each: 84.3 vs 83.3 us at
// 1 M envs, 298.8 vs 301.1 at 4 M: no difference.)
// (Round 3, streaming form: the rows straight from registers instead -- seven 8-byte non-temporal stores per lane, no LDS, ~200
// instructions fewer per wavefront -- measured SLOWER: 302.5 vs 297.8 us at 4 M envs, 91.2 vs 80.3 at 1 M: full-line store
// instructions are worth more to the memory system than the index arithmetic costs the VALU.)
// Lane form: every WAVEFRONT owns a 64-row tile and flushes it itself -- LDS accesses of one wavefront execute in
// order, so the transposition needs no block barrier (round 1: one tile per block behind __syncthreads, the
// wavefronts of a block waiting for the slowest).  Element f = lane + 64 j of the wavefront's 896 contiguous output
// floats sits at tile[f + f / 14].
template <bool STREAMING = false>   // non-temporal stores: the rows are not read again before the caches turn over (launch_step: state matrix > 192 MB, 1.22 M envs)
WL_DEV void flush_obs_wave(const float* tile_w, float* __restrict__ obs, int wave_env0, int n) {
    const int lane = threadIdx.x & 63;
    const int n_valid = min(64, n - wave_env0);
    float* dst = obs + (int64_t)wave_env0 * kObsDim + lane;
    if (n_valid == 64) {               // wave-uniform: every wavefront but the batch's last
        const int m = lane * 4682;     // f / 14 == (f * 4682) >> 16 for f < 896
#pragma unroll
        for (int j = 0; j < kObsDim; ++j) {
            const int e = (m + 64 * j * 4682) >> 16;
            if constexpr (STREAMING) __builtin_nontemporal_store(tile_w[lane + 64 * j + e], dst + 64 * j);
            else dst[64 * j] = tile_w[lane + 64 * j + e];
        }
    } else {
        const int total = n_valid * kObsDim;
        for (int f = lane; f < total; f += 64) dst[f - lane] = tile_w[f + f / kObsDim];
    }
}

// Register image of one env's rows of the state matrix (memory form: root-link position, world-frame velocities).
struct DriftRows {
    V3 pos;
    Quat q;
    V3 v, ww;
    float wheel[4];   // quad form: wheel[0] is this lane's wheel
    float th, om;
    float a0, a1;     // last raw action
    float timer_hf, timer_lf;
    float epsum[WL_DR_NTERMS];   // quad form only (the lane form streams these rows late to save registers)
    int ep_len;
};

// rows that only the tail of the step reads
WL_DEV void load_bookkeeping_rows(const Rows& S, const WlEnvBuffers& b, int e, DriftRows& r) {
    r.timer_hf = S.ld(WL_S_TIMER_HF, e);
    r.timer_lf = S.ld(WL_S_TIMER_LF, e);
    r.ep_len = b.episode_len[e];
}

template <int LANES>
WL_DEV void load_rows(const Rows& S, const WlEnvBuffers& b, const WlDriftParams& p, int e, int wid, DriftRows& r) {
    if constexpr (LANES == 4) {   // latency form: ordered requests, everything up front (see Rows::ld)
        r.pos = v3(S.ld(WL_S_PX, e), S.ld(WL_S_PY, e), S.ld(WL_S_PZ, e));
        r.q = Quat{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
        r.v = v3(S.ld(WL_S_VX, e), S.ld(WL_S_VY, e), S.ld(WL_S_VZ, e));
        r.ww = v3(S.ld(WL_S_WX, e), S.ld(WL_S_WY, e), S.ld(WL_S_WZ, e));
        r.wheel[0] = S.ld_lane_row(WL_S_WHEEL_BL + wid, e);
#pragma unroll
        for (int i = 0; i < WL_DR_NTERMS; ++i) r.epsum[i] = S.ld(WL_S_EPSUM0 + i, e);   // unconditional: gating the REQUEST on
        // log_episode_sums would make the load burst wait for the flag's own fetch; the flag gates the uses (metrics, store)
        r.th = S.ld(WL_S_STEER_POS, e);
        r.om = S.ld(WL_S_STEER_VEL, e);
        r.timer_hf = S.ld(WL_S_TIMER_HF, e);
        r.timer_lf = S.ld(WL_S_TIMER_LF, e);
        r.ep_len = b.episode_len[e];
    } else {                      // throughput form: the bookkeeping rows are fetched after the physics loop (registers)
        r.pos = ld3(S, WL_S_PX, e);
        r.q = Quat{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
        r.v = ld3(S, WL_S_VX, e);
        r.ww = ld3(S, WL_S_WX, e);
#pragma unroll
        for (int i = 0; i < 4; ++i) r.wheel[i] = S.ld(WL_S_WHEEL_BL + i, e);
        r.th = S.ld(WL_S_STEER_POS, e);
        r.om = S.ld(WL_S_STEER_VEL, e);
    }
}

template <int LANES>
WL_DEV void store_rows(const Rows& S, const WlEnvBuffers& b, const WlDriftParams& p, int e, int wid, bool lead,
                       const DriftRows& r) {
    if constexpr (LANES == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) S.st(WL_S_WHEEL_BL + i, e, r.wheel[i]);
    } else {
        S.st_lane_row(WL_S_WHEEL_BL + wid, e, r.wheel[0]);
    }
    if (!lead) return;
    st3(S, WL_S_PX, e, r.pos);
    S.st(WL_S_QW, e, r.q.w);
    S.st(WL_S_QX, e, r.q.x);
    S.st(WL_S_QY, e, r.q.y);
    S.st(WL_S_QZ, e, r.q.z);
    st3(S, WL_S_VX, e, r.v);
    st3(S, WL_S_WX, e, r.ww);
    S.st(WL_S_STEER_POS, e, r.th);
    S.st(WL_S_STEER_VEL, e, r.om);
    S.st(WL_S_ACT0, e, r.a0);
    S.st(WL_S_ACT1, e, r.a1);
    S.st(WL_S_TIMER_HF, e, r.timer_hf);
    S.st(WL_S_TIMER_LF, e, r.timer_lf);
    if constexpr (LANES == 4) {
        if (p.log_episode_sums) {
#pragma unroll
            for (int i = 0; i < WL_DR_NTERMS; ++i) S.st(WL_S_EPSUM0 + i, e, r.epsum[i]);
        }
    }
    b.episode_len[e] = r.ep_len;
}

// per-env constants that do not change during a rollout
WL_DEV void load_env_const(const Rows& S, const WlVehicleParams& vp, const VehDerived& vd, int e, EnvConst& ec) {
    env_const_rows(ec, vp, vd, S.ld(WL_S_MASS, e), S.ld(WL_S_MU_S, e), S.ld(WL_S_MU_D, e), S.ld(WL_S_DAMP, e));
}

// Episode-metric accumulation.  Lane form: LDS atomics into the WAVEFRONT's own 16 accumulators (zeroed and flushed by
// the wavefront itself: no block barrier), <= 16 global atomics per wavefront that had a reset.  Quad form: resets are
// rare per wavefront (16 envs), so the few lead lanes that reset go straight to the global accumulators and the
// kernel needs no LDS at all.
template <int LANES>
struct MetricSink {
    float* lds;     // this wavefront's accumulators (lane form)
    float* glob;    // this wavefront's shard of this step's slot of the metric ring
    WL_DEV void add(int idx, float v) const {
        if constexpr (LANES == 4) atomicAdd(glob + idx, v);
        else atomicAdd(lds + idx, v);
    }
    WL_DEV void open() const {    // lane form, before the first add
        if constexpr (LANES == 1)
            if ((threadIdx.x & 63) < WL_M_COUNT) lds[threadIdx.x & 63] = 0.f;
    }
    WL_DEV void close() const {   // lane form, after the last add (LDS operations of a wavefront complete in order)
        if constexpr (LANES == 1) {
            if ((threadIdx.x & 63) < WL_M_COUNT) {
                const float m = lds[threadIdx.x & 63];
                if (m != 0.f) atomicAdd(glob + (threadIdx.x & 63), m);
            }
        }
    }
};

// sum of 16 values as a tree (depth 4): the non-finite guard is on the step's critical path and a chain of 15 dependent
// adds is 15 x the dependent-issue latency of a wavefront that is alone on its SIMD
WL_DEV float sum16(const float (&v)[16]) {
    const float a0 = v[0] + v[1], a1 = v[2] + v[3], a2 = v[4] + v[5], a3 = v[6] + v[7];
    const float a4 = v[8] + v[9], a5 = v[10] + v[11], a6 = v[12] + v[13], a7 = v[14] + v[15];
    return ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

// ONE env.step() on the register image `r` (shared by the per-step kernel and the persistent rollout kernels): action
// term -> physics -> terminations -> rewards -> reset -> pushes -> observation row.
// Writes reward / flags to `out` (already offset to this step); accumulates episode metrics into `ms`.
//   tile: lane form: this WAVEFRONT's LDS obs tile (64 rows) -- the caller flushes it with flush_obs_wave
//   obs_keep: quad form: the 14 values stay in the caller's registers (policy in the loop) instead of being stored
//   pre: quad form: this step's draws (draw_step), made by the caller where they overlap a memory wait
template <int LANES, bool UNROLL = true, int DRIVE = -1, class Ground>
WL_DEV void drift_env_step(const WlDriftParams& p, const WlEnvBuffers& b, const VehDerived& vd, const Ground& ground,
                           const Rows& S, EnvConst& ec, DriftRows& r, float2 a, const float* __restrict__ noise,
                           const WlStepOut& out, int e, int wid, bool lead, uint32_t gid, uint64_t seed, uint64_t step,
                           float* tile, const MetricSink<LANES>& ms, float* obs_keep = nullptr,
                           const StepDraws* pre = nullptr) {
    const WlVehicleParams& vp = p.vehicle;
    // ---- action manager: ClipAction + process_actions + joint targets (once per env-step) ----
    float v_t, delta;
    process_action(p.action, a.x, a.y, v_t, delta);
    joint_targets(p.action, v_t, delta, ec.steer_target, ec.wheel_target);
    if constexpr (LANES == 4) env_const_lane(ec, vp, vd, wid);
    // ---- memory form -> integrator form (CoM position, body-frame angular velocity) ----
    VehState s;
    s.q = r.q;
    s.v = r.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) s.wheel[i] = r.wheel[i];
    s.th = r.th;
    s.om = r.om;
    {
        const Mat3 R = mat_from_quat(s.q);
        s.x = r.pos + vp.cg_z * v3(R.r0.z, R.r1.z, R.r2.z);  // CoM = link origin + R (0,0,cg_z)
        s.wb = mul_t(R, r.ww);
    }
    // ---- physics: decimation x substeps, everything in registers ----
    vehicle_integrate<LANES, Ground, UNROLL, DRIVE>(vp, vd, ec, s, ground, wid);
    // lane form: the rows only the tail reads are requested NOW, behind ~1000 instructions of cover (terminations,
    // rewards, Euler angles, noise) and not up front, where they would sit in registers through the physics loop
    float epsum[WL_DR_NTERMS];
    if constexpr (LANES == 1) {
        load_bookkeeping_rows(S, b, e, r);
#pragma unroll
        for (int i = 0; i < WL_DR_NTERMS; ++i) epsum[i] = p.log_episode_sums ? S.ld(WL_S_EPSUM0 + i, e) : 0.f;
        __builtin_amdgcn_sched_barrier(0);   // keep the requests here: the scheduler would sink them to their first use
    } else {
#pragma unroll
        for (int i = 0; i < WL_DR_NTERMS; ++i) epsum[i] = p.log_episode_sums ? r.epsum[i] : 0.f;
    }
    const Mat3 R = mat_from_quat(s.q);
    V3 ww = mul(R, s.wb);
    V3 pos = s.x - vp.cg_z * v3(R.r0.z, R.r1.z, R.r2.z);
    // ---- terminations (time_out, cart_off_track) + non-finite guard ----
    float wheel_sum;
    if constexpr (LANES == 1) wheel_sum = (s.wheel[0] + s.wheel[1]) + (s.wheel[2] + s.wheel[3]);
    else wheel_sum = quad_sum(s.wheel[0]);
    const float chk_in[16] = {pos.x, pos.y, pos.z, s.q.w, s.q.x, s.q.y, s.q.z, s.v.x, s.v.y, s.v.z, ww.x, ww.y, ww.z, wheel_sum, s.th, s.om};
    const bool finite = __builtin_isfinite(sum16(chk_in));
    const bool terminated = !finite || cart_off_track(pos.x, pos.y, p.straight, p.r_in, p.r_out);
    // ---- rewards on the post-physics state ----
    V3 vb = mul_t(R, s.v);
    // side-slip angle and the three Euler angles are four atan2s (asin(x) == atan2(x, sqrt(1 - x^2))): the quad form
    // evaluates them as ONE atan2 with per-lane arguments and DPP-broadcasts the results
    float slip_angle;
    V3 euler;
    if constexpr (LANES == 4) {
        const Quat q = s.q;
        const float sp = 2.f * (q.w * q.y - q.z * q.x);
        // all four argument pairs are computed by every lane and picked by value (three v_cndmask each)
        const float ay = quad_pick(wid, opaque(vb.y), opaque(2.f * (q.w * q.x + q.y * q.z)), opaque(sp), opaque(2.f * (q.w * q.z + q.x * q.y)));
        const float ax = quad_pick(wid, opaque(vb.x), opaque(1.f - 2.f * (q.x * q.x + q.y * q.y)), opaque(fsqrt(fmaxf(fmaf(-sp, sp, 1.f), 0.f))),
                                   opaque(1.f - 2.f * (q.y * q.y + q.z * q.z)));
        const float ang = atan2_fast(ay, ax);
        slip_angle = quad_bcast<0>(ang);
        euler = v3(wrap_2pi(quad_bcast<1>(ang)), wrap_2pi(quad_bcast<2>(ang)), wrap_2pi(quad_bcast<3>(ang)));
    } else {
        slip_angle = atan2_fast(vb.y, vb.x);
    }
    int ep_len = r.ep_len + 1;
    const bool truncated = ep_len >= p.max_episode_length;
    DriftTerms tm = drift_terms(p, pos, vb, s.wb, ww.z, s.th, terminated, truncated, slip_angle);
    const float step_dt = p.sim_dt * (float)p.decimation;
    float reward = 0.f;
#pragma unroll
    for (int i = 0; i < WL_DR_NTERMS; ++i) {
        const float w = p.weight[i];
        const float c = (w != 0.f && finite) ? tm.t[i] * w * step_dt : 0.f;  // RewardManager skips w == 0
        reward += c;
        epsum[i] += c;
    }
    if (lead) {
        if (S.streaming) {   // compile-time constant (Rows::streaming): outputs of a batch far larger than the caches
            __builtin_nontemporal_store(reward, out.reward + e);
            __builtin_nontemporal_store((uint8_t)(terminated ? 1 : 0), out.terminated + e);
            __builtin_nontemporal_store((uint8_t)(truncated ? 1 : 0), out.truncated + e);
            if (out.dones) __builtin_nontemporal_store((int64_t)((terminated || truncated) ? 1 : 0), out.dones + e);
        } else {
            out.reward[e] = reward;
            out.terminated[e] = terminated ? 1 : 0;
            out.truncated[e] = truncated ? 1 : 0;
            if (out.dones) out.dones[e] = (terminated || truncated) ? 1 : 0;
        }
    }
    // ---- reset (done envs) ----
    // The observation wants the body-frame velocities of the state AFTER reset and pushes.  Round 1 rebuilt the rotation
    // matrix from the final quaternion and rotated both vectors again (43 instructions, every env, every step); but the
    // common path changes neither, a reset leaves the env at rest, and a push adds a known world-frame increment: the
    // body-frame values are carried along and corrected inside the (divergent, occasional) branches instead.
    Mat3 Ro = R;
    V3 wb_o = s.wb;
    float a0 = a.x, a1 = a.y, timer_hf = r.timer_hf, timer_lf = r.timer_lf;
    // lane form: the event block serves the reset (words x, y, z) and the lf push (w): drawn once, by the wavefronts with a lane that
    // needs either (a lane that resets re-arms its lf timer from the same block, so `lf_due` on the old timer covers every use)
    U4 ev = U4{0u, 0u, 0u, 0u};
    const float lf_next = timer_lf - step_dt;     // ONE value for the draw condition and the push below (a re-contracted copy could differ in the last bit)
    const bool done = terminated || truncated;
    if constexpr (LANES == 1) {
        if (done || (p.enable_pushes && lf_next < 1e-6f)) ev = philox_block(gid, step, WL_RS_DRIFT_EVENTS, seed);
    }
    if (done) {
        if (lead) {
#pragma unroll
            for (int i = 0; i < WL_DR_NTERMS; ++i) ms.add(WL_M_EPSUM0 + i, epsum[i]);
            ms.add(WL_M_RESETS, 1.f);
            if (truncated) ms.add(WL_M_TIMEOUTS, 1.f);
            if (terminated) ms.add(WL_M_TERM0, 1.f);
            if (!finite) ms.add(WL_M_NONFINITE, 1.f);
            ms.add(WL_M_EPLEN, (float)ep_len);
        }
#pragma unroll
        for (int i = 0; i < WL_DR_NTERMS; ++i) epsum[i] = 0.f;
        if (!finite) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s.wheel[i] = 0.f;
            s.th = s.om = 0.f;
        }
        ResetDraw rd;
        if constexpr (LANES == 4) rd = quad_reset_draw(p, *pre);
        else rd = reset_from_block(p, b.ref_poses, ev);
        pos = rd.pos;
        s.q = rd.q;
        if constexpr (LANES == 4) euler = v3(0.f, 0.f, rd.yaw - WL_TWO_PI * floorf(rd.yaw * WL_INV_TWO_PI));
        s.v = v3(0.f, 0.f, 0.f);
        ww = v3(0.f, 0.f, 0.f);
        // the observation frame: at rest, rotated about z only (q = (c, 0, 0, s))
        vb = v3(0.f, 0.f, 0.f);
        wb_o = v3(0.f, 0.f, 0.f);
        const float cy = fmaf(rd.q.w, rd.q.w, -rd.q.z * rd.q.z), sy = 2.f * rd.q.w * rd.q.z;
        Ro = Mat3{v3(cy, -sy, 0.f), v3(sy, cy, 0.f), v3(0.f, 0.f, 1.f)};
        timer_hf = rd.timer_hf;
        timer_lf = rd.timer_lf;
        ep_len = 0;
        a0 = a1 = 0.f;  // ActionManager.reset zeroes `action` (last_action) of reset envs
    }
    // ---- interval events: push_by_setting_velocity (mushr_drift_env_cfg.py:121-143) ----
    bool hf_due = false;
    if (p.enable_pushes) {
        timer_hf -= step_dt;
        hf_due = timer_hf < 1e-6f;
    }
    // lane form: the WL_RS_NOISE1 block serves the hf push (words z, w) and the last four observation normals (x, y): drawn once, here
    U4 nb = U4{0u, 0u, 0u, 0u};
    const bool draw_noise = p.enable_corruption && !noise;
    if constexpr (LANES == 1) {
        if (draw_noise || hf_due) nb = philox_block(gid, step, WL_RS_NOISE1, seed);
    }
    if (p.enable_pushes) {
        if (hf_due) {
            F4 u;
            if constexpr (LANES == 4) u = quad_hf_draw(*pre);
            else u = u16x4(nb.z, nb.w);
            const float dvx = (2.f * u.x - 1.f) * p.hf_vel_x, dvy = (2.f * u.y - 1.f) * p.hf_vel_y, dwz = (2.f * u.z - 1.f) * p.hf_vel_yaw;
            s.v.x += dvx;
            s.v.y += dvy;
            ww.z += dwz;
            vb = v3(fmaf(Ro.r0.x, dvx, fmaf(Ro.r1.x, dvy, vb.x)), fmaf(Ro.r0.y, dvx, fmaf(Ro.r1.y, dvy, vb.y)),
                    fmaf(Ro.r0.z, dvx, fmaf(Ro.r1.z, dvy, vb.z)));
            wb_o = fma3(dwz, Ro.r2, wb_o);
            timer_hf = fmaf(u.w, p.hf_interval[1] - p.hf_interval[0], p.hf_interval[0]);
        }
        timer_lf = done ? timer_lf - step_dt : lf_next;
        if (timer_lf < 1e-6f) {     // (lane form: `ev` was drawn for this lane -- it reset, or the condition above saw the same value)
            float uk, ut;
            if constexpr (LANES == 4) quad_lf_draw(*pre, uk, ut);
            else uk = u16_lo(ev.w), ut = u16_hi(ev.w);
            const float dwz = (2.f * uk - 1.f) * p.lf_vel_yaw;
            ww.z += dwz;
            wb_o = fma3(dwz, Ro.r2, wb_o);
            timer_lf = fmaf(ut, p.lf_interval[1] - p.lf_interval[0], p.lf_interval[0]);
        }
    }
    // ---- back to memory form ----
    r.pos = pos;
    r.q = s.q;
    r.v = s.v;
    r.ww = ww;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.wheel[i] = s.wheel[i];
    r.th = s.th;
    r.om = s.om;
    r.a0 = a0;
    r.a1 = a1;
    r.timer_hf = timer_hf;
    r.timer_lf = timer_lf;
    r.ep_len = ep_len;
    if constexpr (LANES == 4) {
#pragma unroll
        for (int i = 0; i < WL_DR_NTERMS; ++i) r.epsum[i] = epsum[i];
    } else {
        if (p.log_episode_sums) {
#pragma unroll
            for (int i = 0; i < WL_DR_NTERMS; ++i) S.st(WL_S_EPSUM0 + i, e, epsum[i]);
        }
    }
    // ---- observation of the post-reset state (vb, wb_o: carried along above) ----
    const V3 wb2 = wb_o;
    Noise12 nz;
    bool drawn = false;
    if constexpr (LANES == 4) {
        if (p.enable_corruption && !noise) {
            nz = gather_quad_noise(pre->z);
            drawn = true;
        }
    }
    if (!drawn) nz = obs_noise<LANES>(p, noise, b.stride, e, gid, step, seed, LANES == 1 ? &nb : nullptr);   // off / parity tensor / lane form
    if constexpr (LANES == 1) euler = euler_xyz_from_quat(s.q);
    if constexpr (LANES == 4) {
        // quad form: the 14 values are replicated on the quad's lanes; a caller that feeds them to a policy in the
        // same launch (wl_policy.hip) keeps them in registers and stores the row itself
        float o_local[14];
        float* o = obs_keep ? obs_keep : o_local;
        obs_values(o, p, pos, euler, vb, wb2, a0, a1, nz);
        if (!obs_keep) store_obs_quad(out.obs + (int64_t)e * kObsDim, wid, o);
    } else {
        write_obs_row(&tile[(threadIdx.x & 63) * kObsPad], p, pos, euler, vb, wb2, a0, a1, nz);
    }
}

// Persistent kernels keep every parameter live across the whole rollout loop: ~130 uniform dwords against 102 SGPRs, so
// the compiler parks the overflow in lanes of a VGPR and pays a v_readlane (+ hazard nops) per use -- ~600 extra
// instructions per env-step.  gfx950 has no scalar float ALU, so float parameters are only ever VALU operands anyway:
// pinning them into VGPRs (one v_mov each, once per launch) frees the SGPR file for addresses and loop control, and the
// wavefront (alone on its SIMD in these kernels) has hundreds of VGPRs to spare.
WL_DEV void pin_vgpr(float& x) { asm volatile("" : "+v"(x)); }
template <int N>
WL_DEV void pin_vgpr(float (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) pin_vgpr(a[i]);
}
WL_DEV void pin_params_vgpr(WlDriftParams& p, VehDerived& d) {
    pin_vgpr(p.sim_dt);
    WlActionParams& a = p.action;
    pin_vgpr(a.scale), pin_vgpr(a.offset), pin_vgpr(a.base_length), pin_vgpr(a.base_width), pin_vgpr(a.wheel_radius);
    WlVehicleParams& v = p.vehicle;
    float* vf[] = {&v.gravity, &v.half_wheelbase_f, &v.half_wheelbase_r, &v.half_track, &v.wheel_radius, &v.wheel_z, &v.cg_z,
                   &v.gyr_x, &v.gyr_y, &v.gyr_z, &v.wheel_inertia, &v.wheel_damping, &v.susp_k, &v.susp_c, &v.ground_mu_s,
                   &v.ground_mu_d, &v.slip_peak, &v.v_min, &v.motor_sat, &v.motor_limit, &v.motor_vel_limit, &v.steer_kp,
                   &v.steer_kd, &v.steer_effort, &v.steer_vel_limit, &v.steer_inertia, &v.susp_fmax};
#pragma unroll
    for (float* f : vf) pin_vgpr(*f);
    float* pf[] = {&p.straight, &p.r_in, &p.r_out, &p.r_line, &p.slip_min, &p.slip_max, &p.slip_min_vx, &p.speed_target,
                   &p.speed_offset, &p.tlgr_thresh, &p.ctd_offset, &p.ctd_p, &p.pos_noise, &p.yaw_noise, &p.hf_vel_x,
                   &p.hf_vel_y, &p.hf_vel_yaw, &p.lf_vel_yaw};
#pragma unroll
    for (float* f : pf) pin_vgpr(*f);
    pin_vgpr(p.weight), pin_vgpr(p.noise_std), pin_vgpr(p.hf_interval), pin_vgpr(p.lf_interval);
    float* df[] = {&d.h, &d.inv_h, &d.half_h, &d.steer_a, &d.steer_b, &d.steer_J_h, &d.steer_h_J, &d.zrel, &d.Iw_h, &d.A0,
                   &d.inv_wlim, &d.mot_b, &d.r2, &d.inv_A0, &d.hg, &d.inv_g2x, &d.inv_g2y, &d.inv_g2z, &d.cgx, &d.cgy, &d.cgz};
#pragma unroll
    for (float* f : df) pin_vgpr(*f);
}

WL_DEV void keep_scalar_fields(WlDriftParams& v, const WlDriftParams&) {
    uniform_scalar_common(v);
    v.enable_corruption = uniform_i32(v.enable_corruption);
    v.num_ref_points = uniform_i32(v.num_ref_points);
    v.enable_pushes = uniform_i32(v.enable_pushes);
}

// host-side validation shared by every drift entry point
inline int check_buffers(const WlDriftParams* p, const WlEnvBuffers* b) {
    if (!p || !b || !b->state || !b->episode_len || !b->ref_poses || !b->metrics) return WL_EINVAL;
    if (b->n_envs <= 0 || b->stride < b->n_envs) return WL_EINVAL;
    if (b->stride % 64 != 0 || ((uintptr_t)b->state & 15u)) return WL_EALIGN;
    if (b->metrics_slots < 1 || (b->lanes != 0 && b->lanes != 1 && b->lanes != 2 && b->lanes != 4)) return WL_EINVAL;
    if (!flags_ok(b) || ((b->flags & WL_FLAG_STREAM) && b->lanes == 4)) return WL_EINVAL;   // the streaming form is a lane form
    if (b->stride * 4 * WL_S_COUNT > 0x7fffffffLL) return WL_EINVAL;   // buffer-resource offsets are 32-bit (~13 M envs)
    if (p->decimation <= 0 || p->vehicle.substeps <= 0 || p->num_ref_points <= 0 || p->num_ref_points > 32) return WL_EINVAL;
    if (p->vehicle.implicit != 0 || !(p->vehicle.susp_fmax > 0.f)) return WL_EINVAL;   // the drift kernels step the explicit integrator (wl_vehicle.h)
    if (!(p->sim_dt > 0.f)) return WL_EINVAL;
    return WL_OK;
}

}  // namespace

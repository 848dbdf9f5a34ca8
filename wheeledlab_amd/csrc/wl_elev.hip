// wl_elev.hip -- elevation task (wheeledlab_tasks/elevation/mushr_elevation_env_cfg.py) for gfx950.
//
// Launches per env.step():
//   quad form (n <= 32 768): ONE -- elev_step_scan_kernel: block = 16 envs; wavefront 0 steps them (one quad of lanes per
//      env: 4WD action term -> decimation x substeps of the rigid body + 4 tyre contacts on the heightfield (bilinear
//      height + normal gathers, L2-resident grid) -> terminations -> rewards -> in-kernel reset -> goal-command update),
//      then all 8 wavefronts cast the 16 x 676 height rays from the poses wavefront 0 left in LDS.  Round 2: 30.3 ->
//      27.8 us per step at 4096 envs against the two launches below (the scan no longer pays its own launch gap, wave
//      ramp and pose-row latency; a first fused version with 4 wavefronts and 4 batches of gathers per lane gained nothing).
//   lane form: TWO --
//   1. elev_step_kernel  (lane = env): the same step; state is read once / written once, sub-steps stay in VGPRs; also
//      writes the 13 proprioceptive values of the observation row.
//   2. elev_scan_kernel / elev_scan_lds_kernel (block = env): the 26 x 26 yaw-aligned height rays, four per lane, written as
//      16-byte words -- this launch carries ~90 % of the task's HBM bytes (2.7 KB / env).  Gather form (two 4-byte gathers of
//      two height codes per ray) below 16 384 envs, LDS-patch form (the footprint's codes staged by LDS-DMA) beyond; the terrain
//      is 16-bit codes x z_scale since round 5 (wl_heightfield.h).  (Round 2 history: staging the env's
//      terrain patch in LDS -- its bounding box fetched row by row with coalesced 8-byte requests, corners read from
//      the tile -- was measured slower in every form tried: block per env 15.8 us, persistent blocks with the next
//      env's pose prefetched 18 us, against 11.3 us for the gathers; round 2, DESIGN.md section 6.  Round 3: the same for the
//      PHYSICS -- the 24 x 24 grid points under each car staged in LDS by all eight wavefronts of the fused kernel, the wheel
//      contacts of the 20 sub-steps reading their corners from there: + 2.7 us for the staging and NOTHING back per sub-step
//      (1.12 against 1.09 us per decimation step): the sub-step is a dependent VALU chain, its two gathers per wheel are
//      already hidden behind it.)
#include <hip/hip_runtime.h>

// Debug builds only (tools/build_variants.sh ... -DWL_FUSED_TIMELINE=1; tools/fused_timeline.py): lane 0 of a wavefront stamps
// the 100 MHz wall clock into wl_timeline[block][slot] at the phase boundaries of the fused step + scan launch.
#ifndef WL_FUSED_TIMELINE
#define WL_FUSED_TIMELINE 0
#endif
#if WL_FUSED_TIMELINE
__device__ unsigned long long wl_timeline[2048][16];
#define WL_TL(slot) do { if ((threadIdx.x & 63) == 0) wl_timeline[blockIdx.x & 2047][slot] = wall_clock64(); } while (0)
#else
#define WL_TL(slot) do { } while (0)
#endif


#include "../../include/wheeledlab_amd.h"
#include "wl_kernel_common.h"
#include "wl_actor_dev.h"
#include "wl_drift_terms.h"   // process_action / joint_targets (shared action term)
#include "wl_rng.h"
#include "wl_vehicle.h"
#include "wl_heightfield.h"

namespace {

enum ElevStream : uint32_t { ES_RESET = 0, ES_CMD_RESET = 1, ES_CMD_RESAMPLE = 2 };

// cos / sin of the yaw of IsaacLab's yaw_quat(q) without the atan2 round trip
WL_DEV void yaw_cs(Quat q, float& c, float& s) {
    // every multiply-add spelled out: under -ffp-contract=fast the compiler picks WHICH product of a sum of products it fuses per
    // inlining site, and two sites of one kernel then round differently (round 4: the first and the later envs of a persistent
    // scan block disagreed in the last bit)
    const float a = fmaf(-2.f, fmaf(q.z, q.z, q.y * q.y), 1.f), hb = fmaf(q.w, q.z, q.x * q.y), b = hb + hb;
    const float inv = rsq(fmaf(a, a, b * b));
    c = a * inv;
    s = b * inv;
}

WL_DEV float sym(float u, float a) { return (2.f * u - 1.f) * a; }

struct ElevTerms {
    float t[WL_ER_NTERMS];
    bool flag[WL_ET_NTERMS];
};

// elevation mdp terms (citations: mushr_elevation_env_cfg.py)
WL_DEV ElevTerms elev_terms(const WlElevParams& p, V3 pos, float up_dot, V3 vb, V3 vw, float wheel_sum, float cbx, float cby,
                            bool timed_out) {
    ElevTerms r;
    const float gx = cbx - pos.x, gy = cby - pos.y;   // goal vector: base-frame command minus world position (:50-55 quirk)
    r.flag[WL_ET_BELOW_MIN_HEIGHT] = pos.z < p.min_height;                                            // :356-359
    r.flag[WL_ET_STUCK] = fminf(vb.x, p.stuck_vel_cap) < p.stuck_min_vel && wheel_sum > p.stuck_wheel_spin;   // :342-347
    r.flag[WL_ET_ROLLOVER] = up_dot < p.upright_cos;                                                  // :217-222, 339-340
    const float g2 = fmaf(gx, gx, gy * gy);
    r.flag[WL_ET_AT_GOAL] = fsqrt(g2) < p.goal_dist;                               // :268-273
    r.t[WL_ER_GOAL_PROGRESS] = p.progress_offset + fmaf(vw.x, gx, vw.y * gy) * rsq(g2);   // :239-249
    const float z = pos.z - p.elev_z0;
    r.t[WL_ER_HIGHER_ELEVATION] = clampf((z > p.elev_min && vb.x > p.elev_min_vel) ? z : 0.f, 0.f, 1.f);       // :166-173
    r.t[WL_ER_FALLING] = vb.z > p.fall_vel ? 1.f : 0.f;                                               // :251-254
    r.t[WL_ER_STUCK_PENALTY] = (r.flag[WL_ET_STUCK] && !timed_out) ? 1.f : 0.f;                       // :301-305
    return r;
}

// the 13 proprioceptive observation values of ElevationObsCfg.ConcatObs (:61-73), written straight into the env's
// observation row by the lane-per-env kernels (step / prop); the block-per-env scan kernel adds the 676 map values
template <int LANES>
WL_DEV void write_elev_prop(const WlElevParams& p, float* __restrict__ row, V3 pos, Quat q, V3 vb, V3 wb, float cbx, float cby,
                            float a0, float a1, int wid, bool lead, float* row2 = nullptr /* a second copy (the collector's LDS tile) */) {
    V3 eu;
    if constexpr (LANES == 4) {   // roll / pitch / yaw as one lane-parallel atan2 (asin x = atan2(x, sqrt(1 - x^2)))
        const float sp = 2.f * (q.w * q.y - q.z * q.x);
        const float ay = wid == 1 ? 2.f * (q.w * q.x + q.y * q.z) : wid == 2 ? sp : 2.f * (q.w * q.z + q.x * q.y);
        const float ax = wid == 1 ? 1.f - 2.f * (q.x * q.x + q.y * q.y) : wid == 2 ? fsqrt(fmaxf(1.f - sp * sp, 0.f))
                                                                                  : 1.f - 2.f * (q.y * q.y + q.z * q.z);
        const float ang = atan2_fast(ay, ax);
        eu = v3(wrap_2pi(quad_bcast<1>(ang)), wrap_2pi(quad_bcast<2>(ang)), wrap_2pi(quad_bcast<3>(ang)));
    } else {
        eu = euler_xyz_from_quat(q);
    }
    if (!lead) return;
    const float gx = cbx - pos.x, gy = cby - pos.y;
    const float v[13] = {gx != gx ? 0.f : gx,   // nan_to_num(nan=0) (:55); +-inf are left to the policy as in the reference
                         gy != gy ? 0.f : gy,
                         eu.x, eu.y, eu.z,
                         clampf(vb.x, -p.obs_clip, p.obs_clip), clampf(vb.y, -p.obs_clip, p.obs_clip), clampf(vb.z, -p.obs_clip, p.obs_clip),
                         clampf(wb.x, -p.obs_clip, p.obs_clip), clampf(wb.y, -p.obs_clip, p.obs_clip), clampf(wb.z, -p.obs_clip, p.obs_clip),
                         clampf(a0, -1.f, 1.f), clampf(a1, -1.f, 1.f)};
#pragma unroll
    for (int i = 0; i < 13; ++i) row[i] = v[i];
    if (row2) {
#pragma unroll
        for (int i = 0; i < 13; ++i) row2[i] = v[i];
    }
}

struct ElevReset {
    V3 pos;
    Quat q;
    float vx, vy, tgt_x, tgt_y, tgt_h;
};

// isaaclab reset_root_state_uniform with the ranges of :409-419 + UniformPose2dCommand resample (:425-435)
WL_DEV ElevReset draw_elev_reset(const WlElevParams& p, const HeightFieldGround& g, uint32_t gid, uint64_t step, uint64_t seed) {
    const F4 u = philox_uniform4(gid, step, ES_RESET, seed);
    const F4 c = philox_uniform4(gid, step, ES_CMD_RESET, seed);
    ElevReset r;
    const float x = sym(u.x, p.reset_xy), y = sym(u.y, p.reset_xy);
    float zt;
    V3 n;
    g.sample(x, y, zt, n);
    r.pos = v3(x, y, fmaxf(p.reset_z, zt + p.spawn_clearance));
    float s, cc;
    sincos_fast(0.5f * sym(u.z, p.reset_yaw), s, cc);
    r.q = Quat{cc, 0.f, 0.f, s};
    r.vx = fmaf(u.w, p.reset_vel[1] - p.reset_vel[0], p.reset_vel[0]);
    r.vy = fmaf(c.w, p.reset_vel[1] - p.reset_vel[0], p.reset_vel[0]);
    r.tgt_x = sym(c.x, p.cmd_xy);
    r.tgt_y = sym(c.y, p.cmd_xy);
    r.tgt_h = sym(c.z, p.cmd_heading);
    return r;
}

// what the height scan needs of an env after its step: root position and the cos / sin of its yaw
struct ScanPose {
    float px, py, pz, c, s;
};

// ---- world_height_map (:44-48): the 26 x 26 yaw-aligned height scan -- what every form below evaluates per ray ------------------
// The rays of one env form a lattice; in GRID units (cells of the heightfield) ray (ix, iy) stands at
// (u0 + ix ux + iy uy, v0 + ix vx + iy vy).  Round 4: the lattice frame is set up once per env (7 floats, what the fused kernels
// keep in LDS) and a ray costs ~35 VALU instructions instead of ~62 (the ray's metres -> rotate -> translate -> grid units chain,
// four float compares for the inside test and 64-bit address arithmetic went; the scan was as much instruction- as gather-bound:
// 676 rays x 62 / 64 lanes = 655 wavefront-instructions per env against ~1100 clocks per env and CU).  Same definition
// (oracle/elev_step.py::height_map, heightfield.py::sample); fp32 rounding of the ray position differs by a few ulp of the grid
// coordinate (<= 1e-4 cell) from the metre chain, and the sampler's 1e-3-cell guard at the far border is not needed (the
// integer cell index is clamped instead): both far inside the parity tolerance (2e-5 m on the height map).
struct ScanFrame {
    float u0, v0, ux, vx, uy, vy, pz;
};
WL_DEV ScanFrame scan_frame(const WlElevParams& p, const HeightFieldGround& g, const ScanPose& sp) {
    const float g0 = -0.5f * p.scan_size, k = p.scan_res * g.inv_cell;
    ScanFrame f;
    // (multiply-adds spelled out: see yaw_cs)
    f.u0 = ((sp.px + fmaf(sp.c, g0, -(sp.s * g0))) - g.f.x0) * g.inv_cell;
    f.v0 = ((sp.py + fmaf(sp.s, g0, sp.c * g0)) - g.f.y0) * g.inv_cell;
    f.ux = sp.c * k, f.vx = sp.s * k;
    f.uy = -sp.s * k, f.vy = sp.c * k;
    f.pz = sp.pz;
    return f;
}
// ray index r (< 1024) of the scan -> lattice coordinates, meshgrid "xy": x fastest (r / 26 by multiply-shift: exact, checked)
WL_DEV void scan_ray_xy(int r, float& fix, float& fiy) {
    const int iy = (int)(__umul24((unsigned)r, 1261u) >> 15);
    fix = (float)(r - iy * WL_ELEV_SCAN_N);
    fiy = (float)iy;
}
// the cell a ray falls into: float offset of its lower-left grid point, position inside the cell, inside-the-field flag
struct ScanCell {
    int i, j;      // NOT clamped: outside the field when !inside
    float fu, fv;
    bool inside;
};
WL_DEV ScanCell scan_cell(const ScanFrame& f, const WlHeightField& hf, float fix, float fiy) {
    const float u = fmaf(fix, f.ux, fmaf(fiy, f.uy, f.u0)), v = fmaf(fix, f.vx, fmaf(fiy, f.vy, f.v0));
    const float fl_u = floorf(u), fl_v = floorf(v);
    const int i = (int)fl_u, j = (int)fl_v;      // saturating conversion: far-away rays stay far away
    ScanCell c;
    c.fu = u - fl_u;
    c.fv = v - fl_v;
    // 0 <= u < nx - 1 as ONE unsigned compare per axis; a NaN pose (converted to cell 0) is caught through its NaN fraction
    c.inside = (unsigned)i < (unsigned)(hf.nx - 1) && (unsigned)j < (unsigned)(hf.ny - 1) && (c.fu + c.fv >= 0.f);
    c.i = i, c.j = j;
    return c;
}
// a ray in flight: the gather(s) of its cell's four corner codes, decoded and consumed by scan_value.
// PAIR (the gather forms, round 6): ONE 8-byte gather from the row-pair table (WlHeightField.pair): lo = (c00 | c01 << 16), hi = (c10 |
// c11 << 16).  The gather forms were bound by the texture unit's address rate -- rocprofv3 on the fused launch at 4096 envs: TA busy
// 16 000 cycles per CU = the whole scan phase, 57 cache accesses per 64-lane gather instruction, L1 hit rate 0.91 -- and a ray cost
// two of them (rows j and j + 1 of the code field); with the pair table it costs one (measured: fused step 20.5 -> 17.5 us, gather-form
// scan at 262 144 envs 484 -> 319 us).  !PAIR (the LDS patch form, which reads the code field's rows): lo = codes (i, i + 1) of row j,
// hi = of row j + 1, low half = the first.
struct ScanRay {
    uint32_t lo, hi;
    float fu, fv;
    bool inside;
};
struct ScanField {   // the row-pair table through a buffer resource: one 32-bit lane offset per gather
    __amdgpu_buffer_rsrc_t rsrc;
    float z_scale;
};
WL_DEV ScanField scan_field(const WlHeightField& hf) {
    return ScanField{__builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(hf.pair), 0, hf.nx * hf.ny * 4, 0x00020000), hf.z_scale};
}
WL_DEV ScanRay scan_request(const ScanFrame& f, const WlHeightField& hf, const ScanField& sf, float fix, float fiy) {
    typedef unsigned wl_u2 __attribute__((ext_vector_type(2)));
    const ScanCell c = scan_cell(f, hf, fix, fiy);
    ScanRay r;
    r.fu = c.fu, r.fv = c.fv, r.inside = c.inside;
    // a ray outside the field asks for whatever address its cell index wraps to: inside the buffer it reads a value nobody uses
    // (the ray is a miss), outside it the resource's bounds check returns 0 -- four clamps per ray saved.  24-bit multiply (full
    // rate; v_mul_lo_u32 is a quarter-rate instruction): exact for every ray inside the field (check_elev: nx, ny < 2^23)
    const wl_u2 w = __builtin_amdgcn_raw_buffer_load_b64(sf.rsrc, (__mul24(c.j, hf.nx) + c.i) * 4, 0, 0);
    r.lo = w.x, r.hi = w.y;
    return r;
}
// FOUR consecutive rays per lane, stored as ONE 16-byte word (round 4).  The scan's 676 four-byte stores per env were what bound
// it, in the gather form and the LDS form alike (both 500 us per launch at 262 144 envs whatever the read side did): a store
// instruction costs the memory pipeline per INSTRUCTION far more than per byte (MI355X_MICROARCH.md: narrow stores are issue-bound,
// a dword store ~6 x a dwordx4 store per byte).  Rays 4 q .. 4 q + 3 of an env (q < 169) are neighbours along x; the scan row has 26
// = 6.5 quads, so a quad starting at ix = 24 continues at (0, iy + 1): ray pairs (0, 1) and (2, 3) never straddle a row.
typedef float wl_float4_u __attribute__((ext_vector_type(4), aligned(4)));   // observation rows are only 4-byte aligned
constexpr int kScanQuads = WL_ELEV_SCAN_N * WL_ELEV_SCAN_N / 4;             // 169
static_assert(WL_ELEV_SCAN_N % 2 == 0 && (WL_ELEV_SCAN_N * WL_ELEV_SCAN_N) % 4 == 0, "ray pairs stay inside a scan row, whole quads per env");
// quad slot (env-in-block x 169 + quad, < 2^13) -> env j, quad q (idx / 169 by multiply-shift: exact, checked)
WL_DEV void scan_quad_slot(int idx, int& j, int& q) {
    j = (int)(__umul24((unsigned)idx, 6205u) >> 20);
    q = idx - j * kScanQuads;
}
// bilinear height under the ray -> the observation value: -(sensor_z - hit_z - offset) + (root_z - plane_init_value), +inf on a
// miss, clipped to +- obs_clip
template <bool PAIR = true>
WL_DEV float scan_value(const WlElevParams& p, const ScanRay& r, float z_scale, float pz) {
    // the blend runs on the CODES and is scaled once (3 instructions per ray fewer than decoding the four corners first; the large-batch
    // scan is VALU-bound since round 5).  Exactly the decode-first value when z_scale is a power of two (terrain.py's default: scaling
    // by 2^k commutes with every rounding); for other scales it differs from it in the last bit -- every scan form shares this function
    // (codes are < 2^15 in magnitude: their differences are exact in either layout, so both give the same bits)
    const float lo0 = (float)(int)(int16_t)(r.lo & 0xffffu), lo1 = (float)((int)r.lo >> 16);
    const float hi0 = (float)(int)(int16_t)(r.hi & 0xffffu), hi1 = (float)((int)r.hi >> 16);
    const float c00 = lo0, c10 = PAIR ? hi0 : lo1, c01 = PAIR ? lo1 : hi0, c11 = hi1;
    const float a = fmaf(r.fu, c10 - c00, c00), b = fmaf(r.fu, c11 - c01, c01);
    float hz;
    {
#pragma clang fp contract(off)
        hz = fmaf(r.fv, b - a, a) * z_scale;      // a multiply of its own: not fused into the subtraction below per inlining site
    }
    const float val = r.inside ? (-(pz - hz - p.scan_offset) + (pz - p.elev_z0)) : __builtin_inff();
    return clampf(val, -p.obs_clip, p.obs_clip);
}
// the gathers of the four rays of quad q (4 x 8 B in flight per lane) ...
WL_DEV void scan_quad_request(const ScanFrame& f, const WlHeightField& hf, const ScanField& sf, int q, ScanRay (&r)[4]) {
    float fx0, fy0, fx2, fy2;
    scan_ray_xy(4 * q, fx0, fy0);
    scan_ray_xy(4 * q + 2, fx2, fy2);
    r[0] = scan_request(f, hf, sf, fx0, fy0);
    r[1] = scan_request(f, hf, sf, fx0 + 1.f, fy0);
    r[2] = scan_request(f, hf, sf, fx2, fy2);
    r[3] = scan_request(f, hf, sf, fx2 + 1.f, fy2);
}
// ... and their four values as one 16-byte word
template <bool PAIR = true>
WL_DEV wl_float4_u scan_quad_value(const WlElevParams& p, const ScanRay (&r)[4], float z_scale, float pz) {
    wl_float4_u v;
    v.x = scan_value<PAIR>(p, r[0], z_scale, pz), v.y = scan_value<PAIR>(p, r[1], z_scale, pz), v.z = scan_value<PAIR>(p, r[2], z_scale, pz);
    v.w = scan_value<PAIR>(p, r[3], z_scale, pz);
    return v;
}
#ifndef WL_FUSED_SCAN_NT
#define WL_FUSED_SCAN_NT true     // the fused step + scan launches' map rows as non-temporal stores (round 4, 4096 envs: 25.6 -> 24.7 us per step)
#endif
template <bool STREAM>
WL_DEV void scan_quad_store(float* __restrict__ row_map /* obs row + 13 */, int q, wl_float4_u v) {
    wl_float4_u* dst = reinterpret_cast<wl_float4_u*>(row_map + 4 * q);

    if constexpr (STREAM) __builtin_nontemporal_store(v, dst);
    else *dst = v;
}

// the dynamic rows of an env as the step needs them at its start (requested in one go, ahead of the parameter block)
template <int LANES>
struct ElevRows {
    float mass, mu_s, mu_d, damp;
    V3 pos, v, ww;
    Quat q;
    float wheel[LANES == 1 ? 4 : 1];
    float th, om;
};
template <int LANES>
WL_DEV ElevRows<LANES> load_elev_rows(const Rows& S, int e, int wid) {
    ElevRows<LANES> r;
    r.mass = S.ld(WL_S_MASS, e), r.mu_s = S.ld(WL_S_MU_S, e), r.mu_d = S.ld(WL_S_MU_D, e), r.damp = S.ld(WL_S_DAMP, e);
    r.pos = ld3(S, WL_S_PX, e);
    r.q = Quat{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
    r.v = ld3(S, WL_S_VX, e);
    r.ww = ld3(S, WL_S_WX, e);
    if constexpr (LANES == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r.wheel[i] = S.ld(WL_S_WHEEL_BL + i, e);
    } else {
        r.wheel[0] = S.ld(WL_S_WHEEL_BL + wid, e);
    }
    r.th = S.ld(WL_S_STEER_POS, e);
    r.om = S.ld(WL_S_STEER_VEL, e);
    return r;
}

// the bookkeeping rows of an env (episode length, goal command, episode sums) + the last action: with ElevRows everything a
// persistent rollout carries in registers from step to step
struct ElevBook {
    int ep_len;
    float cb[2], epsum[WL_ER_NTERMS], tgt[4];   // tgt: target x, y, heading, resample timer
    float act[2];
};
template <int LANES>
WL_DEV void store_elev_state(const WlElevParams& p, const WlEnvBuffers& b, const Rows& S, int e, int wid, bool lead,
                             const ElevRows<LANES>& r, const ElevBook& k) {
    if constexpr (LANES == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) S.st(WL_S_WHEEL_BL + i, e, r.wheel[i]);
    } else {
        S.st(WL_S_WHEEL_BL + wid, e, r.wheel[0]);
    }
    if (lead) {
        st3(S, WL_S_PX, e, r.pos);
        S.st(WL_S_QW, e, r.q.w);
        S.st(WL_S_QX, e, r.q.x);
        S.st(WL_S_QY, e, r.q.y);
        S.st(WL_S_QZ, e, r.q.z);
        st3(S, WL_S_VX, e, r.v);
        st3(S, WL_S_WX, e, r.ww);
        S.st(WL_S_STEER_POS, e, r.th);
        S.st(WL_S_STEER_VEL, e, r.om);
        S.st(WL_S_ACT0, e, k.act[0]);
        S.st(WL_S_ACT1, e, k.act[1]);
        if (p.log_episode_sums) {
#pragma unroll
            for (int i = 0; i < WL_ER_NTERMS; ++i) S.st(WL_S_EPSUM0 + i, e, k.epsum[i]);
        }
        S.st(WL_S_CMD_BX, e, k.cb[0]);
        S.st(WL_S_CMD_BY, e, k.cb[1]);
        S.st(WL_S_TGT_X, e, k.tgt[0]);
        S.st(WL_S_TGT_Y, e, k.tgt[1]);
        S.st(WL_S_TGT_H, e, k.tgt[2]);
        S.st(WL_S_CMD_TIMER, e, k.tgt[3]);
        b.episode_len[e] = k.ep_len;
    }
}
template <int LANES>
WL_DEV ElevBook load_elev_book(const WlElevParams& p, const WlEnvBuffers& b, const Rows& S, int e) {
    ElevBook k;
    k.ep_len = b.episode_len[e];
    k.cb[0] = S.ld(WL_S_CMD_BX, e);
    k.cb[1] = S.ld(WL_S_CMD_BY, e);
#pragma unroll
    for (int i = 0; i < WL_ER_NTERMS; ++i) k.epsum[i] = p.log_episode_sums ? S.ld(WL_S_EPSUM0 + i, e) : 0.f;
    k.tgt[0] = S.ld(WL_S_TGT_X, e);
    k.tgt[1] = S.ld(WL_S_TGT_Y, e);
    k.tgt[2] = S.ld(WL_S_TGT_H, e);
    k.tgt[3] = S.ld(WL_S_CMD_TIMER, e);
    k.act[0] = k.act[1] = 0.f;
    return k;
}

// one env.step() of env `e` (all LANES lanes of the env take part): the body of the step kernels below.
// PERSIST: the env's rows and bookkeeping live in `rows` / `*carry` across calls (persistent rollout): nothing is loaded
// from or stored to the state matrix here, both are updated in place.
// HOOKS (round 6): how the fused launch takes the step apart.  `pose(pos, q)` is called with the env's pose as the step leaves it
// (AFTER a reset, if the env resets) as soon as that pose is known -- before the rewards are weighted, the outputs, the metrics and the
// state rows are written -- so that the launch can hand the height scan its lattice frames and let the other wavefronts start while
// this one finishes its bookkeeping.  `reset(...)` supplies a resetting env's draw: by default drawn here; the fused launch has an idle
// wavefront draw all 16 envs' resets while the physics runs (the draw depends on (seed, env, step) and the terrain only) -- with a
// reset somewhere in nearly every launch, the draw's ~0.9 us (two Philox blocks, a terrain sample's round trip, sin / cos) was on the
// launch's critical path.
struct NoStepHooks {
    WL_DEV ElevReset reset(const WlElevParams& p, const HeightFieldGround& g, uint32_t gid, uint64_t step, uint64_t seed, int) const {
        return draw_elev_reset(p, g, gid, step, seed);
    }
    WL_DEV void pose(const V3&, const Quat&) const {}
};
template <int LANES, bool PERSIST = false, class HOOKS = NoStepHooks>
WL_DEV ScanPose elev_env_step(const WlElevParams& p, const VehDerived& vd, const WlEnvBuffers& b, const HeightFieldGround& ground,
                              const float2 action, ElevRows<LANES>& rows, const WlStepOut& out, const uint64_t seed,
                              const uint64_t step, const Rows& S, const int e, const int wid, const bool lead, float* blk_metrics,
                              ElevBook* carry = nullptr, float* prop2 = nullptr, const HOOKS& hooks = HOOKS()) {
    const WlVehicleParams& vp = p.vehicle;
    const uint32_t gid = (uint32_t)(b.env_offset + e);
    float2 a = action;
    float v_t, delta;
    process_action(p.action, a.x, a.y, v_t, delta);
    EnvConst ec;
    joint_targets(p.action, v_t, delta, ec.steer_target, ec.wheel_target);
    env_const_rows(ec, vp, vd, rows.mass, rows.mu_s, rows.mu_d, rows.damp);
    if constexpr (LANES == 4) env_const_lane(ec, vp, vd, wid);
    VehState s;
    V3 pos = rows.pos;
    s.q = rows.q;
    s.v = rows.v;
    V3 ww = rows.ww;
#pragma unroll
    for (int i = 0; i < (LANES == 1 ? 4 : 1); ++i) s.wheel[i] = rows.wheel[i];
    s.th = rows.th;
    s.om = rows.om;
    {
        const Mat3 R = mat_from_quat(s.q);
        s.x = pos + vp.cg_z * v3(R.r0.z, R.r1.z, R.r2.z);
        s.wb = mul_t(R, ww);
    }
    // Bookkeeping rows (episode length, goal command, episode sums): the lane form fetches them after the physics loop --
    // it runs several wavefronts per SIMD and the registers are worth more than the latency; the quad form is one wavefront
    // per SIMD with registers to spare, so it requests them BEFORE the loop and finds them landed behind it.
    int ep_len_in = 0;
    float cb_in[2] = {0.f, 0.f}, epsum_in[WL_ER_NTERMS], tgt_in[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WL_ER_NTERMS; ++i) epsum_in[i] = 0.f;
    auto fetch_bookkeeping = [&]() {
        if constexpr (PERSIST) {
            ep_len_in = carry->ep_len;
            cb_in[0] = carry->cb[0], cb_in[1] = carry->cb[1];
#pragma unroll
            for (int i = 0; i < WL_ER_NTERMS; ++i) epsum_in[i] = carry->epsum[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) tgt_in[i] = carry->tgt[i];
            return;
        }
        ep_len_in = b.episode_len[e];
        cb_in[0] = S.ld(WL_S_CMD_BX, e);
        cb_in[1] = S.ld(WL_S_CMD_BY, e);
        if (p.log_episode_sums) {
#pragma unroll
            for (int i = 0; i < WL_ER_NTERMS; ++i) epsum_in[i] = S.ld(WL_S_EPSUM0 + i, e);
        }
        tgt_in[0] = S.ld(WL_S_TGT_X, e);
        tgt_in[1] = S.ld(WL_S_TGT_Y, e);
        tgt_in[2] = S.ld(WL_S_TGT_H, e);
        tgt_in[3] = S.ld(WL_S_CMD_TIMER, e);
    };
    if constexpr (LANES == 4) fetch_bookkeeping();
#ifndef WL_WHEEL_CORNER_CACHE
#define WL_WHEEL_CORNER_CACHE 1
#endif
    if constexpr (LANES == 1 && WL_WHEEL_CORNER_CACHE) {      // lane form: each wheel's cell corners stay in registers between sub-steps
        const HeightFieldGroundCached cached(ground);
        vehicle_integrate<LANES, HeightFieldGroundCached, true, -1, true>(vp, vd, ec, s, cached, wid);
    } else {
        if constexpr (LANES == 4) WL_TL(1);
        vehicle_integrate<LANES, HeightFieldGround, true, -1, true>(vp, vd, ec, s, ground, wid);
        if constexpr (LANES == 4) WL_TL(2);
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
    const V3 vb = mul_t(R, s.v);
    // terminations / rewards use the command as the PREVIOUS step's command update left it (IsaacLab step order)
    float cbx = cb_in[0], cby = cb_in[1];
    const ElevTerms tm = elev_terms(p, pos, R.r2.z, vb, s.v, wheel_sum, cbx, cby, truncated);
    const bool terminated = !finite || tm.flag[0] || tm.flag[1] || tm.flag[2] || tm.flag[3];
    // the pose the step leaves behind first (a reset replaces it): whoever waits for it (hooks.pose) is served before the bookkeeping
    const bool reset_now = terminated || truncated;
    ElevReset rd{};
    if (reset_now) {
        rd = hooks.reset(p, ground, gid, step, seed, e);
        pos = rd.pos;
    }
    hooks.pose(pos, reset_now ? rd.q : s.q);
    const float step_dt = p.sim_dt * (float)p.decimation;
    float reward = 0.f;
    float epsum[WL_ER_NTERMS];
#pragma unroll
    for (int i = 0; i < WL_ER_NTERMS; ++i) {
        const float w = p.weight[i];
        const float c = (w != 0.f && finite) ? tm.t[i] * w * step_dt : 0.f;
        reward += c;
        epsum[i] = p.log_episode_sums ? epsum_in[i] + c : 0.f;
    }
    if (lead) {
        out.reward[e] = reward;
        out.terminated[e] = terminated ? 1 : 0;
        out.truncated[e] = truncated ? 1 : 0;
        if (out.dones) out.dones[e] = reset_now ? 1 : 0;
    }
    float a0 = a.x, a1 = a.y;
    float tgt_x = tgt_in[0], tgt_y = tgt_in[1], tgt_h = tgt_in[2], cmd_timer = tgt_in[3];
    if (reset_now) {
        if (lead) {
#pragma unroll
        for (int i = 0; i < WL_ER_NTERMS; ++i) atomicAdd(&blk_metrics[WL_M_EPSUM0 + i], epsum[i]);
        atomicAdd(&blk_metrics[WL_M_RESETS], 1.f);
        if (truncated) atomicAdd(&blk_metrics[WL_M_TIMEOUTS], 1.f);
#pragma unroll
        for (int k = 0; k < WL_ET_NTERMS; ++k)
            if (finite && tm.flag[k]) atomicAdd(&blk_metrics[WL_M_TERM0 + k], 1.f);
        if (!finite) atomicAdd(&blk_metrics[WL_M_NONFINITE], 1.f);
        atomicAdd(&blk_metrics[WL_M_EPLEN], (float)ep_len);
        }
#pragma unroll
        for (int i = 0; i < WL_ER_NTERMS; ++i) epsum[i] = 0.f;
        if (!finite) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s.wheel[i] = 0.f;
            s.th = s.om = 0.f;
        }
        s.q = rd.q;
        s.v = v3(rd.vx, rd.vy, 0.f);
        ww = v3(0.f, 0.f, 0.f);
        tgt_x = rd.tgt_x;
        tgt_y = rd.tgt_y;
        tgt_h = rd.tgt_h;
        cmd_timer = p.cmd_resample_s;
        ep_len = 0;
        a0 = a1 = 0.f;
    }
    // command manager: count down, resample expired targets, re-express the target in the yaw-aligned base frame
    cmd_timer -= step_dt;
    if (cmd_timer <= 0.f) {
        const F4 u = philox_uniform4(gid, step, ES_CMD_RESAMPLE, seed);
        tgt_x = sym(u.x, p.cmd_xy);
        tgt_y = sym(u.y, p.cmd_xy);
        tgt_h = sym(u.z, p.cmd_heading);
        cmd_timer = p.cmd_resample_s;
    }
    {
        float c, sn;
        yaw_cs(s.q, c, sn);
        const float dx = tgt_x - pos.x, dy = tgt_y - pos.y;
        cbx = fmaf(c, dx, sn * dy);
        cby = fmaf(-sn, dx, c * dy);
    }
    {   // the env's new rows: kept (persistent rollout) or written back
        rows.pos = pos, rows.q = s.q, rows.v = s.v, rows.ww = ww, rows.th = s.th, rows.om = s.om;
#pragma unroll
        for (int i = 0; i < (LANES == 1 ? 4 : 1); ++i) rows.wheel[i] = s.wheel[i];
        ElevBook k;
        k.ep_len = ep_len;
        k.cb[0] = cbx, k.cb[1] = cby;
#pragma unroll
        for (int i = 0; i < WL_ER_NTERMS; ++i) k.epsum[i] = epsum[i];
        k.tgt[0] = tgt_x, k.tgt[1] = tgt_y, k.tgt[2] = tgt_h, k.tgt[3] = cmd_timer;
        k.act[0] = a0, k.act[1] = a1;
        if constexpr (PERSIST) *carry = k;
        else store_elev_state<LANES>(p, b, S, e, wid, lead, rows, k);
    }
    // proprioceptive part of the observation, from the post-reset state (all lanes of a quad take part)
    const Mat3 R2 = mat_from_quat(s.q);
    write_elev_prop<LANES>(p, out.obs + (int64_t)e * WL_ELEV_OBS_DIM, pos, s.q, mul_t(R2, s.v), mul_t(R2, ww), cbx, cby, a0, a1,
                           wid, lead, prop2);
    float yc, ys;
    yaw_cs(s.q, yc, ys);
    return ScanPose{pos.x, pos.y, pos.z, yc, ys};
}

// (lane form: 110 VGPRs = 4 wavefronts per SIMD.  Squeezed to 96 / 80 registers for 5 / 6 wavefronts the kernel spills 60 / 128 bytes
// of scratch per lane and the step at 262 144 envs goes from 650 to 664 / 782 us: round 3.)
#ifndef WL_ELEV_LANE_WAVES
#define WL_ELEV_LANE_WAVES 1
#endif
template <int LANES>
__global__ void __launch_bounds__(kBlock, LANES == 1 ? WL_ELEV_LANE_WAVES : 1) elev_step_kernel(const WlElevParams p_arg, const VehDerived vd_arg, const WlEnvBuffers b,
                                                           const HeightFieldGround ground, const float2* __restrict__ actions,
                                                           const WlStepOut out, const uint64_t seed, const uint64_t step) {
    __shared__ float blk_metrics[WL_M_COUNT];
    WlElevParams p = p_arg;
    VehDerived vd = vd_arg;
    if constexpr (LANES == 4) {   // latency form: one batch of vector loads instead of dependent scalar-load round trips
        kernarg_vector_copy2(0, p, vd);
        keep_scalar_common(p, p_arg);
        vd.n_sub = vd_arg.n_sub;
    }
    constexpr int kEnvs = kBlock / LANES;
    const int wid = LANES == 1 ? 0 : (threadIdx.x & 3);
    const bool lead = LANES == 1 || wid == 0;
    const int e = blockIdx.x * kEnvs + threadIdx.x / LANES;
    if (threadIdx.x < WL_M_COUNT) blk_metrics[threadIdx.x] = 0.f;
    const int m_slot = b.metrics_slots > 1 ? (int)(step % (uint64_t)b.metrics_slots) : 0;
    if (b.metrics_slots > 1) clear_metric_slot(b, (m_slot + 1) % b.metrics_slots);
    __syncthreads();
    const Rows S = make_rows(b.state, b.stride);
    if (e < b.n_envs) {
        ElevRows<LANES> rows = load_elev_rows<LANES>(S, e, wid);
        (void)elev_env_step<LANES>(p, vd, b, ground, actions[e], rows, out, seed, step, S, e, wid, lead, blk_metrics);
    }
    __syncthreads();
    if (threadIdx.x < WL_M_COUNT) {
        const float m = blk_metrics[threadIdx.x];
        if (m != 0.f) atomicAdd(metric_shard(b, m_slot) + threadIdx.x, m);   // threads 0..15 = wavefront 0 of the block
    }
}

// world_height_map (:44-48): the 26 x 26 yaw-aligned height scan, clipped to +-10, into obs[e][13:689].  One block per
// env; the env's 13 proprioceptive values are written by the lane-per-env kernels (step / prop).
// (Rounds 1 - 3: 128 threads per env, 5.3 rays per lane, dword stores.)
constexpr int kScanThreads = 192;                       // threads per env: 169 quads of rays -> one quad per lane, three wavefronts
template <bool STREAM>   // observation rows of the launch beyond the Infinity Cache: non-temporal stores
__global__ void __launch_bounds__(kScanThreads) elev_scan_kernel(const WlElevParams p, const WlEnvBuffers b, const HeightFieldGround ground,
                                                                 float* __restrict__ obs) {
    const int e = blockIdx.x, tid = threadIdx.x;
    if (tid >= kScanQuads) return;
    const Rows S = make_rows(b.state, b.stride);
    const float px = S.ld(WL_S_PX, e), py = S.ld(WL_S_PY, e), pz = S.ld(WL_S_PZ, e);
    const Quat q{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
    float c, s;
    yaw_cs(q, c, s);
    const ScanFrame fr = scan_frame(p, ground, ScanPose{px, py, pz, c, s});
    const ScanField sf = scan_field(ground.f);
    ScanRay cr[4];
    scan_quad_request(fr, ground.f, sf, tid, cr);     // this lane's four rays: their 8 gathers in flight together
    scan_quad_store<STREAM>(obs + (int64_t)e * WL_ELEV_OBS_DIM + 13, tid, scan_quad_value(p, cr, sf.z_scale, pz));
}
// The same scan with the env's terrain patch staged in LDS (BASELINE config 3: "heightfield gather ... in LDS").  One block =
// one env.  The bounding box of the yaw-rotated 2.5 m footprint -- at most 74 x 74 grid points of the 0.05 m field -- is fetched
// as whole rows (consecutive lanes = consecutive 16-byte words: full-rate coalesced requests, against two divergent 4-byte
// gathers per ray) and the 676 rays read their four corners from LDS (two aligned dword pairs each, the code pair cut out with v_alignbit) with
// the arithmetic of scan_cell / scan_value: the rows are bit-identical to the gather form's.  The staging costs NO vector
// arithmetic per element: THREADS lanes x 16 bytes = a whole number of patch rows of PITCH codes per pass, so a thread's column
// group never changes and its row advances by a constant -- the global offset of pass `it` is the thread's constant lane offset
// + a SCALAR offset, its LDS address the wavefront's M0 base + an immediate.  Rows past the field's end read 0 through the buffer
// resource's bounds check (never used: rays there are misses).
// Round 5: 16-bit codes -- a patch row of 80 codes is 160 B instead of 320, the block stages half of round 4's 25.6 KB (the L2 -> LDS
// volume was the scan's largest stream: ~6 GB per launch at 262 144 envs).  The patch origin
// is an EVEN column (16-byte requests from 4-byte aligned addresses need an even code index; the row pitch nx must be even too:
// scan_patch_fits), one column of the 80 - 74 spare.
// (Round 4, first version: 256 threads, pitch 74, flat index split by multiply-shift per element, pass count by a chain of scalar
// branches -- 245 VALU + 216 SALU per wavefront, four wavefronts per env: SLOWER than the gathers at every size, 597 against
// 502 us per observation launch at 262 144 envs: the scan is instruction-bound before it is address-rate-bound.)
constexpr int kPatch = 74;               // grid points per side the bounding box can need (scan_size * sqrt 2 / cell + 3 must fit)
constexpr int kPatchRows = 80;           // patch rows staged at most (kPatch + slack, a multiple of 16)
// Block shape (round 5, us per observation launch at 262 144 envs, same box): 320 threads / pitch 80 (round 4's shape, three
// requests per thread) 411 - 414; 256 / 128: the same; 192 / 96 (the three wavefronts that cast the rays also stage, five requests
// per thread, 15.4 KB per env) 355 - 357 -- fewer, fuller wavefronts per env; shipped.
#ifndef WL_SCAN_LDS_THREADS
#define WL_SCAN_LDS_THREADS 192
#endif
#ifndef WL_SCAN_LDS_PITCH
#define WL_SCAN_LDS_PITCH 96             // codes per LDS row (>= kPatch + 2: an even origin costs one column)
#endif
#ifndef WL_SCAN_LDS_STAGERS
#define WL_SCAN_LDS_STAGERS WL_SCAN_LDS_THREADS      // threads that issue staging requests (a whole number of patch rows per pass)
#endif
// the 7 pose rows of env e (block-uniform address) by ONE lane per wavefront, broadcast with v_readfirstlane: the texture unit is
// charged per lane address, and with five wavefronts per env the pose loads were 35 of the block's 133 full-width vector-memory
// instructions.  (Through the scalar cache instead -- s_load_dword x 7 -- the launch is faster up to 16 384 envs, 12.6 against
// 13.9 us at 4096, and TWICE as slow beyond, 911 against 483 us at 262 144: every block's seven lines miss the small scalar cache.)
WL_DEV void load_pose_lane0(const Rows& S, int e, float (&v)[7]) {
    float r[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    static_assert(WL_S_PX == 0 && WL_S_QZ == 6, "pose = rows 0 .. 6");
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) r[k] = S.ld(k, e);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r[k])));
}
// Block = one env.  Wavefront 0 alone fetches the pose rows and sets the env up (yaw, lattice frame, the patch's
// origin and row count) and hands 11 words to the others through LDS: the launch is bound by the INSTRUCTIONS it issues, not by
// what it moves or waits for (measured at 262 144 envs, us per observation launch, profiles/r04_scan_experiments.txt: half of the
// staging lanes switched off 522 against 538; two / three envs per block with all their requests overlapped 442 / 572 against
// 418), and with the set-up repeated by all five wavefronts it was a third of them.
struct ScanSetup {      // what wavefront 0 publishes (44 bytes)
    ScanFrame fr;
    int origin, i0, j0, rows;      // byte offset of the patch origin in the field; its column and row; patch rows needed
};
template <bool STREAM, int THREADS, int PITCH, int STAGERS = THREADS>
__global__ void __launch_bounds__(THREADS) elev_scan_lds_kernel(const WlElevParams p, const WlEnvBuffers b, const HeightFieldGround ground,
                                                                float* __restrict__ obs) {
    constexpr int kWordsPerRow = PITCH / 8;                          // 16-byte words per patch row
    constexpr int kRowsPerPass = STAGERS / kWordsPerRow;
    constexpr int kPasses = (kPatchRows + kRowsPerPass - 1) / kRowsPerPass;
    constexpr int kAlways = 64 / kRowsPerPass;                      // 64 rows: the footprint at yaw 0 (52 rows) and a little beyond
    static_assert(PITCH % 8 == 0 && PITCH >= kPatch + 2 && THREADS % 64 == 0 && THREADS >= 192, "whole 16-byte words per row; three ray wavefronts");
    static_assert(kRowsPerPass * kWordsPerRow == STAGERS && STAGERS <= THREADS && 64 % kRowsPerPass == 0 && kAlways >= 1 && kAlways <= kPasses, "whole rows per pass");
    __shared__ __attribute__((aligned(16))) int16_t patch[PITCH * kRowsPerPass * kPasses + 8];   // + the dword past the last pair read
    __shared__ __attribute__((aligned(16))) ScanSetup setup;
    const int e = blockIdx.x, tid = threadIdx.x;
    const WlHeightField& f = ground.f;
    if (tid < 64) {
        float pose[7];
        load_pose_lane0(make_rows(b.state, b.stride), e, pose);
        float c, s;
        yaw_cs(Quat{pose[3], pose[4], pose[5], pose[6]}, c, s);
        const ScanFrame fr = scan_frame(p, ground, ScanPose{pose[0], pose[1], pose[2], c, s});
        // the patch: the lattice's bounding box in grid units (its corners are rays (0,0), (25,0), (0,25), (25,25)), the +1 corner
        // of the last cell, a little slack for rounding; the origin column rounded down to an even one
        constexpr float kSpan = (float)(WL_ELEV_SCAN_N - 1);
        const float u_lo = fr.u0 + fminf(kSpan * fr.ux, 0.f) + fminf(kSpan * fr.uy, 0.f), v_lo = fr.v0 + fminf(kSpan * fr.vx, 0.f) + fminf(kSpan * fr.vy, 0.f);
        const float v_hi = fr.v0 + fmaxf(kSpan * fr.vx, 0.f) + fmaxf(kSpan * fr.vy, 0.f);
        const int i0 = min(max((int)floorf(u_lo - 0.02f), 0) & ~1, f.nx - PITCH), j0 = min(max((int)floorf(v_lo - 0.02f), 0), f.ny - kPatch);
        if (tid == 0) {
            setup.fr = fr;
            setup.origin = (j0 * f.nx + i0) * 2;
            setup.i0 = i0, setup.j0 = j0;
            setup.rows = min(max((int)floorf(v_hi + 0.02f) + 2 - j0, 1), kPatch);
        }
    }
    __syncthreads();
    const int origin = __builtin_amdgcn_readfirstlane(setup.origin), rows = __builtin_amdgcn_readfirstlane(setup.rows);
    // staging: 16 bytes (8 codes) per lane and request, kRowsPerPass patch rows per pass: a thread's column group never changes and
    // its row advances by kRowsPerPass -- global offset = constant lane offset + a SCALAR pass offset.  Straight into LDS
    // (buffer_load_dwordx4 ... lds): a wavefront's 64 lanes land as 1 KB at its M0 base -- exactly this layout (consecutive
    // threads = consecutive 16-byte words) -- with no staging registers and no ds_write.  The compiler does not wait for LDS-DMA:
    // vmcnt(0) by hand before the barrier that publishes the patch.
    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(f.height), 0, f.nx * f.ny * 2, 0x00020000);
    const int r0 = tid / kWordsPerRow;                               // compile-time divisor: multiply-shift
    const int c8 = tid - r0 * kWordsPerRow;
    const int lane_off = ((int)__umul24((unsigned)r0, (unsigned)f.nx) + 8 * c8) * 2;
    const int pass_bytes = kRowsPerPass * f.nx * 2;
    if (STAGERS == THREADS || tid < STAGERS) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        int16_t* wave_base = patch + (tid >> 6) * 512;              // 64 lanes x 16 B = 512 codes
#pragma unroll
        for (int it = 0; it < kAlways; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr)(wave_base + it * STAGERS * 8), 16, lane_off, origin + it * pass_bytes, 0, 0);
#pragma unroll
        for (int it = kAlways; it < kPasses; ++it)
            if (rows > it * kRowsPerPass)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr)(wave_base + it * STAGERS * 8), 16, lane_off, origin + it * pass_bytes, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    }
    __syncthreads();
    if (tid < kScanQuads) {     // the first three wavefronts: one quad of rays per lane
        const ScanFrame fr = setup.fr;
        const int i0 = setup.i0, j0 = setup.j0;
        float fx[4], fy[4];
        scan_ray_xy(4 * tid, fx[0], fy[0]);
        scan_ray_xy(4 * tid + 2, fx[2], fy[2]);
        fx[1] = fx[0] + 1.f, fy[1] = fy[0], fx[3] = fx[2] + 1.f, fy[3] = fy[2];
        ScanRay cr[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const ScanCell cell = scan_cell(fr, f, fx[m], fy[m]);
            // clamped (unsigned minimum: a negative offset wraps to the top and is clamped with everything else): a point the
            // bounding box missed would read a wrong corner (the parity tests would show it), never out of bounds
            // ALIGNED dword pairs + v_alignbit: a code pair at an odd column straddles two LDS dwords, and a misaligned ds_read_b32
            // is served lane by lane (measured: the scan launch at 262 144 envs 698 us with 2-byte aligned reads against 404 us for
            // the fp32 patch of round 4)
            const unsigned rj = min((unsigned)(cell.j - j0), (unsigned)(kRowsPerPass * kPasses - 2)), ri = min((unsigned)(cell.i - i0), (unsigned)(PITCH - 2));
            const uint32_t* h = reinterpret_cast<const uint32_t*>(patch) + rj * (PITCH / 2) + (ri >> 1);
            const unsigned sh = (ri & 1u) * 16u;
            cr[m].lo = __builtin_amdgcn_alignbit(h[1], h[0], sh);
            cr[m].hi = __builtin_amdgcn_alignbit(h[PITCH / 2 + 1], h[PITCH / 2], sh);
            cr[m].fu = cell.fu, cr[m].fv = cell.fv, cr[m].inside = cell.inside;
        }
        scan_quad_store<STREAM>(obs + (int64_t)e * WL_ELEV_OBS_DIM + 13, tid, scan_quad_value<false>(p, cr, f.z_scale, fr.pz));
    }
}
// the staged patch must hold the footprint's bounding box at any yaw; 16-byte staging requests need an even row pitch (and the
// even origin column one spare code)
inline bool scan_patch_fits(const WlElevParams* p, const WlHeightField* hf) {
    return hf->nx >= WL_SCAN_LDS_PITCH && hf->ny >= kPatch && (hf->nx & 1) == 0 && p->scan_size * 1.41422f / hf->cell + 3.2f <= (float)kPatch;
}
// Which form (round 6): the GATHER form at every size.  With the row-pair table a ray is one lane address instead of two and the
// gather form passed the LDS form everywhere (us per observation launch, same box, gather / LDS: 16 384 envs 26.8 / 28.3, 65 536:
// 94 / 95, 262 144: 345 / 356, 1 M: 1223 / 1252; rounds 4 - 5, two gathers per ray: 484 - 515 against 404 -> 334 at 262 144).  Both
// the gather form is bound by the texture unit's one lane address per cycle (676 per env); what holds the LDS form at the same time
// (its texture-unit and vector work are each ~half of its launch) was never pinned down.  WL_FLAG_SCAN_LDS still selects the LDS form
// (BASELINE config 3's "heightfield patch in LDS"; the parity tests run both), WL_SCAN_LDS_MIN_ENVS (default: never) a size from
// which it is the default.
#ifndef WL_SCAN_LDS_MIN_ENVS
#define WL_SCAN_LDS_MIN_ENVS 0x7fffffff
#endif
#ifndef WL_ELEV_FUSED_MAX_ENVS
#define WL_ELEV_FUSED_MAX_ENVS 12288
#endif
#ifndef WL_ELEV_STREAM_BYTES
#define WL_ELEV_STREAM_BYTES 0ll
#endif
inline void launch_elev_scan(const WlElevParams* p, const WlEnvBuffers* b, const HeightFieldGround& g, float* obs, hipStream_t hs) {
    // non-temporal map rows at every size unless WL_FLAG_NO_STREAM asks otherwise (round 4: 2 - 3 % of the step from 4096 envs up; the rows
    // are not read again by this launch, and an XCD's L2 does not survive the launch boundary anyway)
    const bool stream = use_streaming(b, (int64_t)b->n_envs * WL_ELEV_OBS_DIM * 4, WL_ELEV_STREAM_BYTES);
    const bool lds = scan_patch_fits(p, &g.f) && ((b->flags & WL_FLAG_SCAN_LDS) || (!(b->flags & WL_FLAG_SCAN_GATHER) && b->n_envs >= WL_SCAN_LDS_MIN_ENVS));
    if (lds) {
        if (stream) elev_scan_lds_kernel<true, WL_SCAN_LDS_THREADS, WL_SCAN_LDS_PITCH, WL_SCAN_LDS_STAGERS><<<b->n_envs, WL_SCAN_LDS_THREADS, 0, hs>>>(*p, *b, g, obs);
        else elev_scan_lds_kernel<false, WL_SCAN_LDS_THREADS, WL_SCAN_LDS_PITCH, WL_SCAN_LDS_STAGERS><<<b->n_envs, WL_SCAN_LDS_THREADS, 0, hs>>>(*p, *b, g, obs);
        return;
    }
    if (stream) elev_scan_kernel<true><<<b->n_envs, kScanThreads, 0, hs>>>(*p, *b, g, obs);
    else elev_scan_kernel<false><<<b->n_envs, kScanThreads, 0, hs>>>(*p, *b, g, obs);
}

// env.step() AND the height scan as ONE launch (quad form, n <= 32 768): block = 16 envs, 8 wavefronts.  Wavefront 0
// steps them (16 quads, as in elev_step_kernel<4>) and leaves each env's post-step pose in LDS; then all eight
// wavefronts cast the 16 x 676 rays (flat index over (env, ray): 21.1 per lane, in three batches of gathers).  Against the
// two-launch form this removes the scan kernel's own start (launch gap, wave ramp, the pose rows' first-touch latency)
// from the step's dependent chain, and one physics wavefront per CU spreads the step over 256 CUs instead of 64 (4096 envs).
constexpr int kFusedThreads = 512, kFusedEnvs = 16;

// what the policy phase of the collector kernel reads and fills: rows k of an rsl_rl RolloutStorage
struct PolicyIo {
    WlMlp actor, critic;
    const float* std;
    const float* obs_in;   // [n][689] observation row k
    float *actions, *mu, *log_prob, *values;
    int deterministic;
};

// POLICY: the runner's whole collection step -- actions = alg.act(obs) -> env.step(actions) -> next observation
// (modified_rsl_rl_runner.py:70-80) -- in this one launch.  Phase A: the eight wavefronts are 2 nets x 4 shares of the 689
// features of layer 1 for the block's 16 observation rows -- the structure AND the arithmetic of
// actor_critic_act_kernel<ACT, 4, RT> (same feature ranges, bias on the first share, partial accumulators summed in the same
// order; f32 MFMA is an fmaf chain), so the collector equals { wl_actor_critic_act; wl_elev_step } bit for bit wherever that
// kernel splits the features four ways (the elevation agent at <= 8192 rows).  Wavefront 0 (actor) and wavefront 4 (critic)
// then finish their nets (layers 2-3, draw, log-prob) and wavefront 0 hands the 16 actions to the physics through LDS.
// MEASURED SLOWER than the two launches it replaces (4096 envs: 47.9 us against 16.2 + 29.1 = 45.4 us per collection step;
// with layer 1's MFMA loop cut out 34.4, with the tail cut out 44.0): a block of 16 rows streams BOTH first-layer matrices
// (352 KB) from L2 -- 90 MB per launch, twice the stand-alone policy kernel's traffic (32 rows per block), and that kernel
// is already bound by exactly this L2 -> L1 operand stream.  More rows per block would halve it and double the scan phase
// per CU.  The entry point stays (bit-identical to the two calls, tests/test_gpu_training.py); the runner does not use it.
template <bool POLICY, int ACT = WL_ACT_ELU>
__global__ void __launch_bounds__(kFusedThreads) elev_step_scan_kernel(const WlElevParams p_arg, const VehDerived vd_arg,
                                                                       const WlEnvBuffers b, const HeightFieldGround ground,
                                                                       const float2* __restrict__ actions, const WlStepOut out,
                                                                       const uint64_t seed, const uint64_t step, const PolicyIo pio) {
    __shared__ float blk_metrics[WL_M_COUNT];
    __shared__ ScanFrame frame[kFusedEnvs];
    __shared__ __attribute__((aligned(16))) float hbuf[POLICY ? 2 * 3 * kMlpTiles * 64 * 4 : 4];   // partial accumulators [net][share - 1][tile][lane][4]
    __shared__ float2 act_lds[kFusedEnvs];
    __shared__ ElevReset reset_lds[kFusedEnvs];     // the block's 16 reset draws, by wavefront 1 while wavefront 0 integrates (!POLICY)
    __shared__ int reset_ready;
    const int tid = threadIdx.x;
    if (tid < 64) WL_TL(0);
    if (tid == 0) reset_ready = 0;
    if (tid < WL_M_COUNT) blk_metrics[tid] = 0.f;
    const int m_slot = b.metrics_slots > 1 ? (int)(step % (uint64_t)b.metrics_slots) : 0;
    if (b.metrics_slots > 1) clear_metric_slot(b, (m_slot + 1) % b.metrics_slots);
    const int e0 = blockIdx.x * kFusedEnvs;
    if constexpr (POLICY) {
        const int lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
        const int which = wave >> 2, share = wave & 3;        // net, share of the features
        const WlMlp& net = which == 0 ? pio.actor : pio.critic;
        constexpr int D = WL_ELEV_OBS_DIM, kShares = 4, kFull = D >> 4, kPer = (kFull + kShares - 1) / kShares;
        const float* w_lane = net.w1 + (int64_t)m * D + 4 * g;      // unit m (+ 16 t), features 4 g ..
        const float* x_lane = pio.obs_in + (int64_t)min(e0 + m, b.n_envs - 1) * D + 4 * g;
        // the tail's weights and the draw are independent of layer 1: requested / computed in the shadow of its loads
        MlpTail W;
        float z0 = 0.f, z1 = 0.f;
        if (share == 0) {
            load_tail(net, lane, W);
            if (which == 0 && !pio.deterministic) {
                const F4 u = philox_uniform4((uint32_t)(b.env_offset + e0 + m), step, WL_RS_POLICY, seed);
                box_muller(u.x, u.y, z0, z1);
            }
        }
        f32x4 h[kMlpTiles];
#pragma unroll
        for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[t][r] = share == 0 ? net.b1[16 * t + 4 * g + r] : 0.f;   // the bias seeds the first share
        const int c0 = min(share * kPer, kFull), c1 = min(c0 + kPer, kFull);
        constexpr int kDepth = 4;   // chunks of operands in flight (4 weight tiles + the observation rows each)
        wl_f4u ra[kDepth][kMlpTiles], rb[kDepth];
#pragma unroll
        for (int j = 0; j < kDepth; ++j)
            if (c0 + j < c1) {
#pragma unroll
                for (int t = 0; t < kMlpTiles; ++t) ra[j][t] = *reinterpret_cast<const wl_f4u*>(w_lane + (int64_t)16 * t * D + ((c0 + j) << 4));
                rb[j] = *reinterpret_cast<const wl_f4u*>(x_lane + ((c0 + j) << 4));
            }
        wl_f4u la[kMlpTiles], lb;   // the partial last chunk (last share), requested up front as well
        const bool has_last = share == kShares - 1 && (D & 15);
        if (has_last) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool in = (kFull << 4) + 4 * g + s < D;
#pragma unroll
                for (int t = 0; t < kMlpTiles; ++t) la[t][s] = in ? w_lane[(int64_t)16 * t * D + (kFull << 4) + s] : 0.f;
                lb[s] = in ? x_lane[(kFull << 4) + s] : 0.f;
            }
        }
        for (int c = c0; c < c1; c += kDepth) {
#pragma unroll
            for (int j = 0; j < kDepth; ++j) {
                if (c + j < c1) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < kMlpTiles; ++t) h[t] = mfma4(ra[j][t][s], rb[j][s], h[t]);
                    if (c + j + kDepth < c1) {
#pragma unroll
                        for (int t = 0; t < kMlpTiles; ++t)
                            ra[j][t] = *reinterpret_cast<const wl_f4u*>(w_lane + (int64_t)16 * t * D + ((c + j + kDepth) << 4));
                        rb[j] = *reinterpret_cast<const wl_f4u*>(x_lane + ((c + j + kDepth) << 4));
                    }
                }
            }
        }
        if (has_last) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < kMlpTiles; ++t) h[t] = mfma4(la[t][s], lb[s], h[t]);
        }
        // shares 1..3 hand their partial accumulators to share 0 through LDS: [net][share - 1][tile][lane] f32x4
        if (share > 0) {
#pragma unroll
            for (int t = 0; t < kMlpTiles; ++t)
                *reinterpret_cast<f32x4*>(hbuf + (((which * 3 + share - 1) * kMlpTiles + t) * 64 + lane) * 4) = h[t];
        }
        __syncthreads();
        if (share == 0) {
#pragma unroll
            for (int k = 0; k < kShares - 1; ++k)
#pragma unroll
                for (int t = 0; t < kMlpTiles; ++t)
                    h[t] += *reinterpret_cast<const f32x4*>(hbuf + (((which * 3 + k) * kMlpTiles + t) * 64 + lane) * 4);
            const f32x4 o4 = eval_tail<ACT>(W, h, lane);
            const int r_out = e0 + m;
            if (g == 0) {
                if (which == 1) {
                    if (r_out < b.n_envs) pio.values[r_out] = o4[0];
                } else {
                    const float std0 = pio.std[0], std1 = pio.std[1];
                    const float2 av = make_float2(fmaf(std0, z0, o4[0]), fmaf(std1, z1, o4[1]));
                    act_lds[m] = av;
                    if (r_out < b.n_envs) {
                        reinterpret_cast<float2*>(pio.actions)[r_out] = av;
                        reinterpret_cast<float2*>(pio.mu)[r_out] = make_float2(o4[0], o4[1]);
                        pio.log_prob[r_out] = fmaf(-0.5f, fmaf(z0, z0, z1 * z1), -(log_fast(std0) + log_fast(std1)) - kLog2PiA);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int wid = tid & 3, e = e0 + (tid >> 2);
        if (e < b.n_envs) {
            // ONE memory round trip in front of the physics instead of four in series (parameter block, derived block, action,
            // state rows -- each waited for where it was first used): the state rows and the action are requested first (they
            // need only scalar arguments), the two argument structs right behind them as one burst, and a basic-block boundary
            // keeps the instruction selector from sinking the requests back to their first uses (as in drift_step_kernel).
            // (the action through a buffer resource, like the state rows: as a plain global load the scheduler parked it behind
            // the parameter block, waiting for a register the burst still had in flight -- a second round trip)
            float2 a;
            if constexpr (POLICY) {
                a = act_lds[tid >> 2];
            } else {
                float2* ap = const_cast<float2*>(actions);
                asm volatile("" : "+s"(ap));   // launder the read-only / no-alias argument: its loads are otherwise free to cross the boundary below
                const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(ap, 0, b.n_envs * 8, 0x00020000);
                a.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ar, e * 8, 0, 0));
                a.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ar, e * 8, 4, 0));
            }
            const Rows S = make_rows(b.state, b.stride);
            ElevRows<4> rows = load_elev_rows<4>(S, e, wid);
            WlElevParams p;
            VehDerived vd;
            kernarg_vector_copy2(0, p, vd);
            int go = 1;
            asm volatile("" : "+s"(go) : : "memory");
            if (go) {
                keep_scalar_common(p, p_arg);
                vd.n_sub = vd_arg.n_sub;
                // the block's ONE meeting point sits in the middle of this wavefront's step: as soon as the pose it leaves behind is
                // known (after a reset's draw) the lattice frames go to LDS and the other seven wavefronts start casting rays, while
                // this one weights its rewards, writes outputs, metrics and state rows and then joins them for a smaller share
                // (fused_scan_share).  Round 6, tools/fused_timeline.py: the bookkeeping was 1.1 us of every launch with seven
                // wavefronts waiting behind it.  (Every block has an env, so wavefront 0 always gets here: one s_barrier per wavefront.)
                struct FusedHooks {
                    const WlElevParams& p;
                    const HeightFieldGround& ground;
                    ScanFrame* frame;
                    const ElevReset* reset_lds;
                    int* reset_ready;
                    int tid, wid, e0;
                    WL_DEV ElevReset reset(const WlElevParams& pp, const HeightFieldGround& g, uint32_t gid, uint64_t st, uint64_t sd, int e) const {
                        if constexpr (POLICY) return draw_elev_reset(pp, g, gid, st, sd);      // (the collector's other wavefronts are busy with the nets)
                        // wavefront 1 set the flag long ago (its draws take ~1.5 us, this is ~9 us into the launch): the loop is the guarantee
                        while (__hip_atomic_load(reset_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
                        return reset_lds[e - e0];
                    }
                    WL_DEV void pose(const V3& pos_out, const Quat& q_out) const {
                        float yc, ys;
                        yaw_cs(q_out, yc, ys);
                        if (wid == 0) frame[tid >> 2] = scan_frame(p, ground, ScanPose{pos_out.x, pos_out.y, pos_out.z, yc, ys});
                        WL_TL(3);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                };
                const FusedHooks hooks{p, ground, frame, reset_lds, &reset_ready, tid, wid, e0};
                (void)elev_env_step<4, false, FusedHooks>(p, vd, b, ground, a, rows, out, seed, step, S, e, wid, wid == 0, blk_metrics, nullptr, nullptr, hooks);
            }
        }
        WL_TL(4);
        if (tid < WL_M_COUNT) {
            const float m = blk_metrics[tid];
            if (m != 0.f) atomicAdd(metric_shard(b, m_slot) + tid, m);
        }
    } else {
        if constexpr (!POLICY) {
            if (tid < 128) {      // wavefront 1: the block's reset draws, in the shadow of the physics (FusedHooks::reset)
                const int j = tid - 64;
                if (j < kFusedEnvs && e0 + j < b.n_envs) reset_lds[j] = draw_elev_reset(p_arg, ground, (uint32_t)(b.env_offset + e0 + j), step, seed);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (j == 0) __hip_atomic_store(&reset_ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (tid >= kFusedThreads - 64) WL_TL(8);
    }
    // ---- the scan ----
    const WlElevParams& p = p_arg;
    const int n_here = min(kFusedEnvs, b.n_envs - e0);
    const ScanField sf = scan_field(ground.f);
    // flat index over (env, quad of rays): 16 x 169 quads / 512 lanes = 5.3 per lane, ONE quad (4 eight-byte gathers) per batch:
    // request, blend, store, next.  (Rounds 2 - 3, single rays: 1 / 2 / 3 batches of gathers 28.4 / 26.2 / 25.9 us per step at 4096
    // envs; round 6, 10 sub-steps: 1 / 2 / 3 / 6 batches 21.4 / 21.3 / 21.3 / 20.2 us.)
    // What bounds the phase (round 6, tools/fused_timeline.py: wall-clock stamps inside the launch; tools/tcp_pass.sh: TA / TCP
    // counters): NOT its stores -- 0.2 - 0.4 us from a wavefront's last store to its acknowledgement; 11.3 MB written with nothing
    // in front of them cost 1.8 - 2.9 us on top of an empty launch (tools/microbench/row_burst.hip), and that overlaps the phase --
    // but the TEXTURE UNIT's address rate: with two 4-byte gathers per ray the phase took 6.1 - 6.8 us, the unit was busy 16 000
    // cycles per CU (all of it), 57 cache accesses per 64-lane gather instruction, L1 hit rate 0.91.  One 8-byte gather per ray
    // (the row-pair table): 4.2 - 4.5 us = 10 816 lane addresses per CU; the phase's ~1000 vector instructions per wavefront are 1.8 us
    // of pipe at two wavefronts per SIMD, so it is still the unit's -- fewer lane addresses (an LDS patch) would be the next step.
    // Measured and dropped in round 6: the same loop software-pipelined (quad k + 1 requested before quad k is blended and stored:
    // 21.0 against 20.5 us, and 19.1 against 18.6 with the pair table -- the older wavefronts of a SIMD run ahead and the younger
    // ones finish alone); store cache policies sc1 / nt / write-through (equal); a wavefront per env with the rays in 7 x 9 blocks of
    // neighbours (21.1 us).
    // 43 chunks of 64 ray quads (the last one holds 16) in six rounds; the wavefronts of a round take NEIGHBOURING chunks (what is in
    // flight on the CU at a time then covers ~3 envs' patches of the table; with a contiguous range per wavefront -- 16 envs' patches
    // at once -- the phase took 4.8 instead of 4.2 us: L1).  Wavefront 0, busy with its bookkeeping, sits out the first two rounds:
    // rounds 0 - 1: wavefronts 1 - 7 (7 chunks each), rounds 2 - 4: all eight, round 5: wavefronts 0 - 4 -- per SIMD (wavefronts s and
    // s + 4) 11 chunks, or 10 + the bookkeeping.
    constexpr int kAll = kFusedEnvs * kScanQuads, kChunks = (kAll + 63) / 64;
    static_assert(kChunks == 43 && kFusedThreads == 512, "the shares below are for 16 envs x 169 quads on eight wavefronts");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const bool mine = r < 2 ? wave != 0 : (r < 5 || wave <= 4);      // wavefront-uniform
        if (mine) {
            const int chunk = r < 2 ? 7 * r + wave - 1 : 14 + 8 * (r - 2) + wave;
            const int idx = chunk * 64 + lane;
            int j, q;
            scan_quad_slot(min(idx, kAll - 1), j, q);
            const ScanFrame fr = frame[min(j, n_here - 1)];
            ScanRay cr[4];
            scan_quad_request(fr, ground.f, sf, q, cr);
            if (idx < kAll && j < n_here)
                scan_quad_store<WL_FUSED_SCAN_NT>(out.obs + (int64_t)(e0 + j) * WL_ELEV_OBS_DIM + 13, q, scan_quad_value(p, cr, sf.z_scale, fr.pz));
        }
    }
#if WL_FUSED_TIMELINE
    if (tid < 64) WL_TL(5);
    if (tid >= kFusedThreads - 64) WL_TL(9);
    __builtin_amdgcn_s_waitcnt(0);
    if (tid < 64) WL_TL(6);
    if (tid >= kFusedThreads - 64) WL_TL(10);
#endif
}

// K env.step()s in ONE launch with pre-staged actions [K][n][2] (open-loop rollouts: sampling-based planners, system
// identification, the bench; quad form, n <= 32 768).  Block = 16 envs.  Wavefront 0 keeps their rows and bookkeeping in
// registers across the K steps (ElevRows / ElevBook: no state round trip, no launch boundary per step) and leaves each
// step's poses in one of two LDS buffers; wavefronts 1..7 cast the height rays of step k WHILE wavefront 0 already
// integrates step k + 1 -- with actions that do not depend on the observations the scan is off the critical path.  One
// s_barrier per step and wavefront: at barrier k wavefront 0 has finished step k, the others the scan of step k - 1.
// Episode metrics of all K steps go to ring slot `slots.cur`, `slots.next` is cleared for the next launch.
constexpr int kScanLanes = kFusedThreads - 64;
__global__ void __launch_bounds__(kFusedThreads) elev_rollout_persistent_kernel(const WlElevParams p_arg, const VehDerived vd_arg,
                                                                                const WlEnvBuffers b, const HeightFieldGround ground,
                                                                                const float2* __restrict__ actions, const WlStepOut out,
                                                                                const int64_t obs_step_stride, const int64_t vec_step_stride,
                                                                                const int n_steps, const uint64_t seed, const uint64_t step0,
                                                                                const MetricSlots slots) {
    __shared__ float blk_metrics[WL_M_COUNT];
    __shared__ ScanFrame frame[2][kFusedEnvs];
    const int tid = threadIdx.x;
    if (tid < WL_M_COUNT) blk_metrics[tid] = 0.f;
    if (b.metrics_slots > 1) clear_metric_slot(b, slots.next);
    __syncthreads();
    const int e0 = blockIdx.x * kFusedEnvs;
    if (tid < 64) {
        const int wid = tid & 3, e = e0 + (tid >> 2);
        const bool valid = e < b.n_envs;
        const Rows S = make_rows(b.state, b.stride);
        ElevRows<4> rows;
        ElevBook book;
        WlElevParams p;
        VehDerived vd;
        if (valid) {
            rows = load_elev_rows<4>(S, e, wid);
            kernarg_vector_copy2(0, p, vd);
            keep_scalar_common(p, p_arg);
            vd.n_sub = vd_arg.n_sub;
            book = load_elev_book<4>(p, b, S, e);
        }
        for (int k = 0; k < n_steps; ++k) {
            if (valid) {
                WlStepOut o = out;
                o.obs += k * obs_step_stride;
                o.reward += k * vec_step_stride;
                o.terminated += k * vec_step_stride;
                o.truncated += k * vec_step_stride;
                if (o.dones) o.dones += k * vec_step_stride;
                const float2 a = actions[(int64_t)k * b.n_envs + e];
                const ScanPose sp = elev_env_step<4, true>(p, vd, b, ground, a, rows, o, seed, step0 + (uint64_t)k, S, e, wid, wid == 0,
                                                           blk_metrics, &book);
                if (wid == 0) frame[k & 1][tid >> 2] = scan_frame(p, ground, sp);
            }
            __syncthreads();   // barrier k: the poses of step k are published
        }
        if (valid) store_elev_state<4>(p, b, S, e, wid, wid == 0, rows, book);
        if (tid < WL_M_COUNT) {   // only this wavefront accumulated
            const float m = blk_metrics[tid];
            if (m != 0.f) atomicAdd(metric_shard(b, slots.cur) + tid, m);
        }
        return;
    }
    // ---- wavefronts 1..7: the scan of step k, one step behind the physics ----
    const WlElevParams& p = p_arg;
    const int t7 = tid - 64;
    constexpr int kAll = kFusedEnvs * kScanQuads;
    constexpr int kBatches = 3, kSlots = (kAll + kScanLanes - 1) / kScanLanes, kBatch = (kSlots + kBatches - 1) / kBatches;
    const int n_here = min(kFusedEnvs, b.n_envs - e0);
    const ScanField sf = scan_field(ground.f);
    for (int k = 0; k < n_steps; ++k) {
        __syncthreads();       // barrier k
        float* obs_k = out.obs + k * obs_step_stride;
#pragma unroll
        for (int part = 0; part < kBatches; ++part) {
            ScanRay cr[kBatch][4];
            float pz[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                int j, q;
                scan_quad_slot(min(t7 + (part * kBatch + i) * kScanLanes, kAll - 1), j, q);
                const ScanFrame fr = frame[k & 1][min(j, n_here - 1)];
                pz[i] = fr.pz;
                scan_quad_request(fr, ground.f, sf, q, cr[i]);
            }
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int idx = t7 + (part * kBatch + i) * kScanLanes;
                int j, q;
                scan_quad_slot(idx, j, q);
                if (idx < kAll && j < n_here) scan_quad_store<WL_FUSED_SCAN_NT>(obs_k + (int64_t)(e0 + j) * WL_ELEV_OBS_DIM + 13, q, scan_quad_value(p, cr[i], sf.z_scale, pz[i]));
            }
        }
    }
}

// proprioceptive part only, lane per env: used by wl_elev_observe (reset / get_observations path)
// ---- the runner's collection, K x { actions = actor(obs) -> env.step -> storage rows } (modified_rsl_rl_runner.py:70-80), as ONE launch ----
// Block = 16 envs, eight wavefronts, K steps in a loop.  What makes the policy step cheap enough to live inside the env's
// block: the ACTOR's first-layer matrix (64 x 689 f32 = 176 KB) is held in the block's REGISTERS for the whole launch (each
// of the eight wavefronts keeps the MFMA A fragments of its 5 - 6 sixteen-feature chunks: <= 96 VGPRs; a CU's register
// file is 512 KB), and the block's 16 observation rows live in LDS, where the scan writes them.  Per step and block:
//   A  all eight wavefronts: partial layer-1 products of the actor from registers x LDS rows (<= 96 MFMAs each), to LDS
//   B  wavefront 0: sums them, layers 2 - 3 (tail weights parked in LDS), draw, log-prob, storage rows; then the 16 envs'
//      physics as in elev_rollout_persistent_kernel (rows and bookkeeping parked in LDS between steps: registers are the
//      scarce resource here -- every wavefront's allocation carries the 96 weight registers)
//   D  all wavefronts: the 16 x 676 height rays of the post-step poses, into the storage's next observation row AND the LDS rows
// Three s_barriers per step.  (The one-launch-per-step collector above streams both matrices from L2 for 16 rows per step
// and block -- 90 MB per step chip-wide -- and is bound by exactly that.)
// The CRITIC is not in here: its values are not needed to step, and the caller evaluates all K + 1 observation rows in one
// batched pass afterwards.  Measured with the critic inside (its first layer streamed from L2 by wavefronts 1..7 beside
// the physics, wavefront 1 finishing the net): 55 us per step against 33.7 without -- not the streaming itself (pacing it
// and cache-policy bits changed nothing) but its registers: the allocator answered with 85 - 160 spills, part of them
// inside the sub-step loop of the lone physics wavefront.
// Arithmetic: layer 1 is summed in eight partial sums per unit (wl_actor_critic_act: four): equal to that kernel to rounding,
// not bit for bit; the env.step is elev_env_step's.
constexpr int kColWavesA = 8;                                                  // wavefronts sharing the actor's first layer
constexpr int kColChunks = (WL_ELEV_OBS_DIM + 15) / 16;                        // 44 (the last holds one feature)
constexpr int kColMaxA = (kColChunks + kColWavesA - 1) / kColWavesA;           // 6
constexpr int kTilePitch = (WL_ELEV_OBS_DIM + 3) / 4 * 4;                      // 692 floats: 16-byte aligned LDS rows
constexpr int kPartFloats = kColWavesA * kMlpTiles * 64 * 4;                   // one net's partial accumulators
constexpr int kTailFloats = (kMlpTiles * kMlpHidSteps + kMlpHidSteps) * 64;    // an MlpTail, [value][lane]
constexpr int kCarryWords = 20 + 3 + WL_ER_NTERMS + 4 + 2;                      // wavefront 0's rows + bookkeeping (carry_io), [word][lane]
constexpr int kColLdsFloats = kFusedEnvs * kTilePitch + kPartFloats + kTailFloats + kCarryWords * 64 + kFusedEnvs * 13 +
                              kFusedEnvs * 8 + kFusedEnvs * 2 + WL_M_COUNT + 4;
WL_DEV int col_chunk_begin(int w, int waves) { return w * (kColChunks / waves) + min(w, kColChunks % waves); }

// wavefront-private values parked in LDS, one column per lane (no barrier: the lane that writes is the lane that reads).
// Field by field: a memcpy through a word array left the array on the stack.
struct LdsColumn {
    float* base;
    int lane, i;
    WL_DEV void put(float v) { base[(i++) * 64 + lane] = v; }
    WL_DEV void put(int v) { base[(i++) * 64 + lane] = __int_as_float(v); }
    WL_DEV void get(float& v) { v = base[(i++) * 64 + lane]; }
    WL_DEV void get(int& v) { v = __float_as_int(base[(i++) * 64 + lane]); }
};
template <bool PUT, class T>
WL_DEV void col_io(LdsColumn& c, T& v) {
    if constexpr (PUT) c.put(v); else c.get(v);
}
template <bool PUT>
WL_DEV void tail_io(float* base, int lane, MlpTail& W) {
    LdsColumn c{base, lane, 0};
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
        for (int i = 0; i < kMlpHidSteps; ++i) col_io<PUT>(c, W.w2[t][i]);
#pragma unroll
    for (int i = 0; i < kMlpHidSteps; ++i) col_io<PUT>(c, W.w3[i]);
}
template <bool PUT>
WL_DEV void carry_io(float* base, int lane, ElevRows<4>& r, ElevBook& k) {
    LdsColumn c{base, lane, 0};
    col_io<PUT>(c, r.mass), col_io<PUT>(c, r.mu_s), col_io<PUT>(c, r.mu_d), col_io<PUT>(c, r.damp);
    col_io<PUT>(c, r.pos.x), col_io<PUT>(c, r.pos.y), col_io<PUT>(c, r.pos.z);
    col_io<PUT>(c, r.v.x), col_io<PUT>(c, r.v.y), col_io<PUT>(c, r.v.z);
    col_io<PUT>(c, r.ww.x), col_io<PUT>(c, r.ww.y), col_io<PUT>(c, r.ww.z);
    col_io<PUT>(c, r.q.w), col_io<PUT>(c, r.q.x), col_io<PUT>(c, r.q.y), col_io<PUT>(c, r.q.z);
    col_io<PUT>(c, r.wheel[0]), col_io<PUT>(c, r.th), col_io<PUT>(c, r.om);
    col_io<PUT>(c, k.ep_len), col_io<PUT>(c, k.cb[0]), col_io<PUT>(c, k.cb[1]);
#pragma unroll
    for (int i = 0; i < WL_ER_NTERMS; ++i) col_io<PUT>(c, k.epsum[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) col_io<PUT>(c, k.tgt[i]);
    col_io<PUT>(c, k.act[0]), col_io<PUT>(c, k.act[1]);
}

template <int ACT>
__global__ void __launch_bounds__(kFusedThreads) elev_collect_rollout_kernel(const WlElevParams p, const VehDerived vd, const WlEnvBuffers b,
                                                                             const HeightFieldGround ground, const WlStepOut out,
                                                                             const int n_steps, const uint64_t seed, const uint64_t step0,
                                                                             const PolicyIo pio, const MetricSlots slots) {
    extern __shared__ __attribute__((aligned(16))) float col_lds[];
    float* obs_tile = col_lds;                                   // [16][kTilePitch]
    float* part_a = obs_tile + kFusedEnvs * kTilePitch;          // [8][4][64][4]
    float* tail_a = part_a + kPartFloats;                        // the actor's MlpTail, [85][64]
    float* carry = tail_a + kTailFloats;                         // wavefront 0's ElevRows + ElevBook between steps, [words][64]
    float* prop = carry + kCarryWords * 64;                      // [16][13]
    ScanFrame* frame = reinterpret_cast<ScanFrame*>(prop + kFusedEnvs * 13);   // [16] (7 floats each, 8 reserved)
    float2* act_lds = reinterpret_cast<float2*>(prop + kFusedEnvs * 13 + kFusedEnvs * 8);
    float* blk_metrics = reinterpret_cast<float*>(act_lds + kFusedEnvs);
    constexpr int D = WL_ELEV_OBS_DIM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
    const int e0 = blockIdx.x * kFusedEnvs, n = b.n_envs;
    const int n_here = min(kFusedEnvs, n - e0);
    if (tid < WL_M_COUNT) blk_metrics[tid] = 0.f;
    if (b.metrics_slots > 1) clear_metric_slot(b, slots.next);
    // the block's first observation rows -> LDS (rows past the batch and the pad columns: zero)
    for (int i = tid; i < kFusedEnvs * kTilePitch; i += kFusedThreads) {
        const int r = i / kTilePitch, c = i - r * kTilePitch;
        obs_tile[i] = (r < n_here && c < D) ? pio.obs_in[(int64_t)(e0 + r) * D + c] : 0.f;
    }
    // every wavefront: its chunks of the actor's first layer as MFMA A fragments (unit 16 t + m, features 16 c + 4 g ..)
    const int a0 = col_chunk_begin(wave, kColWavesA), a1 = col_chunk_begin(wave + 1, kColWavesA);
    f32x4 ra[kColMaxA][kMlpTiles];
#pragma unroll
    for (int j = 0; j < kColMaxA; ++j)
#pragma unroll
        for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 16 * (a0 + j) + 4 * g + q;
                ra[j][t][q] = (a0 + j < a1 && f < D) ? pio.actor.w1[(int64_t)(16 * t + m) * D + f] : 0.f;
            }
    // wavefront 0: the actor's tail weights and the envs' rows / bookkeeping, parked in LDS between their uses
    const int wid = tid & 3, e = e0 + (tid >> 2);
    const bool phys = wave == 0 && e < n;
    const Rows S = make_rows(b.state, b.stride);
    if (wave == 0) {
        MlpTail Wa;
        load_tail(pio.actor, lane, Wa);
        tail_io<true>(tail_a, lane, Wa);
        if (phys) {
            ElevRows<4> rows = load_elev_rows<4>(S, e, wid);
            ElevBook book = load_elev_book<4>(p, b, S, e);
            carry_io<true>(carry, lane, rows, book);
        }
    }
    const ScanField sf = scan_field(ground.f);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < n_steps; ++k) {
        const uint64_t step = step0 + (uint64_t)k;
        const int64_t kn = (int64_t)k * n;
        // ---- A: actor layer 1, partial sums (all eight wavefronts) ----
        {
            f32x4 h[kMlpTiles];
#pragma unroll
            for (int t = 0; t < kMlpTiles; ++t) h[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < kColMaxA; ++j)
                if (a0 + j < a1) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(obs_tile + m * kTilePitch + 16 * (a0 + j) + 4 * g);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int t = 0; t < kMlpTiles; ++t) h[t] = mfma4(ra[j][t][q], x[q], h[t]);
                }
#pragma unroll
            for (int t = 0; t < kMlpTiles; ++t) *reinterpret_cast<f32x4*>(part_a + ((wave * kMlpTiles + t) * 64 + lane) * 4) = h[t];
        }
        __syncthreads();   // barrier 1: the actor's partial sums are in LDS
        if (wave == 0) {
            // ---- B: the rest of the actor, the draw, the storage rows; then the physics ----
            {
                f32x4 h[kMlpTiles];
#pragma unroll
                for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[t][r] = pio.actor.b1[16 * t + 4 * g + r];
#pragma unroll
                for (int w = 0; w < kColWavesA; ++w)
#pragma unroll
                    for (int t = 0; t < kMlpTiles; ++t) h[t] += *reinterpret_cast<const f32x4*>(part_a + ((w * kMlpTiles + t) * 64 + lane) * 4);
                MlpTail Wa;
                tail_io<false>(tail_a, lane, Wa);
                const f32x4 o4 = eval_tail<ACT>(Wa, h, lane);
                if (g == 0) {
                    float z0 = 0.f, z1 = 0.f;
                    if (!pio.deterministic) {
                        const F4 u = philox_uniform4((uint32_t)(b.env_offset + e0 + m), step, WL_RS_POLICY, seed);
                        box_muller(u.x, u.y, z0, z1);
                    }
                    const float std0 = pio.std[0], std1 = pio.std[1];
                    const float2 av = make_float2(fmaf(std0, z0, o4[0]), fmaf(std1, z1, o4[1]));
                    act_lds[m] = av;
                    if (e0 + m < n) {
                        reinterpret_cast<float2*>(pio.actions)[kn + e0 + m] = av;
                        reinterpret_cast<float2*>(pio.mu)[kn + e0 + m] = make_float2(o4[0], o4[1]);
                        pio.log_prob[kn + e0 + m] = fmaf(-0.5f, fmaf(z0, z0, z1 * z1), -(log_fast(std0) + log_fast(std1)) - kLog2PiA);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // act_lds: written by lanes 0..15, read by all 64 below
            __builtin_amdgcn_wave_barrier();
            if (phys) {
                WlStepOut o = out;
                o.obs += kn * D;
                o.reward += kn;
                o.terminated += kn;
                o.truncated += kn;
                if (o.dones) o.dones += kn;
                const float2 a = act_lds[tid >> 2];
                ElevRows<4> rows;
                ElevBook book;
                carry_io<false>(carry, lane, rows, book);
                const ScanPose sp = elev_env_step<4, true>(p, vd, b, ground, a, rows, o, seed, step, S, e, wid, wid == 0, blk_metrics, &book,
                                                           prop + (tid >> 2) * 13);
                carry_io<true>(carry, lane, rows, book);
                if (wid == 0) frame[tid >> 2] = scan_frame(p, ground, sp);
            }
        }
        __syncthreads();   // barrier 2: poses and proprioception of step k published; nobody reads the old observation rows any more
        // ---- D: the next observation rows: proprioception from wavefront 0, the height scan by everyone ----
        int tl = tid;
        asm volatile("" : "+v"(tl));   // per-step copy: keeps the 22 rays' index arithmetic of a thread inside the loop (registers)
        if (tl < kFusedEnvs * 13) {
            const int r = tl / 13, c = tl - r * 13;
            obs_tile[r * kTilePitch + c] = prop[tl];
        }
        {
            constexpr int kAll = kFusedEnvs * kScanQuads;
            constexpr int kBatches = 3, kSlots = (kAll + kFusedThreads - 1) / kFusedThreads, kBatch = (kSlots + kBatches - 1) / kBatches;
            float* obs_k = out.obs + kn * D;
#pragma unroll
            for (int part = 0; part < kBatches; ++part) {
                ScanRay cr[kBatch][4];
                float pz[kBatch];
#pragma unroll
                for (int i = 0; i < kBatch; ++i) {
                    int j, q;
                    scan_quad_slot(min(tl + (part * kBatch + i) * kFusedThreads, kAll - 1), j, q);
                    const ScanFrame fr = frame[min(j, n_here - 1)];
                    pz[i] = fr.pz;
                    scan_quad_request(fr, ground.f, sf, q, cr[i]);
                }
#pragma unroll
                for (int i = 0; i < kBatch; ++i) {
                    const int idx = tl + (part * kBatch + i) * kFusedThreads;
                    int j, q;
                    scan_quad_slot(idx, j, q);
                    if (idx < kAll && j < n_here) {
                        const wl_float4_u v = scan_quad_value(p, cr[i], sf.z_scale, pz[i]);
                        scan_quad_store<WL_FUSED_SCAN_NT>(obs_k + (int64_t)(e0 + j) * D + 13, q, v);
                        float* t4 = obs_tile + j * kTilePitch + 13 + 4 * q;
                        t4[0] = v.x, t4[1] = v.y, t4[2] = v.z, t4[3] = v.w;
                    }
                }
            }
        }
        __syncthreads();   // barrier 3: the observation rows of step k + 1 are complete
    }
    if (wave == 0) {
        if (phys) {
            ElevRows<4> rows;
            ElevBook book;
            carry_io<false>(carry, lane, rows, book);
            store_elev_state<4>(p, b, S, e, wid, wid == 0, rows, book);
        }
        if (tid < WL_M_COUNT) {   // only this wavefront accumulated
            const float v = blk_metrics[tid];
            if (v != 0.f) atomicAdd(metric_shard(b, slots.cur) + tid, v);
        }
    }
}

__global__ void __launch_bounds__(kBlock) elev_prop_kernel(const WlElevParams p, const WlEnvBuffers b, float* __restrict__ obs) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= b.n_envs) return;
    const Rows S = make_rows(b.state, b.stride);
    const V3 pos = ld3(S, WL_S_PX, e);
    const Quat q{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
    const Mat3 R = mat_from_quat(q);
    write_elev_prop<1>(p, obs + (int64_t)e * WL_ELEV_OBS_DIM, pos, q, mul_t(R, ld3(S, WL_S_VX, e)), mul_t(R, ld3(S, WL_S_WX, e)),
                       S.ld(WL_S_CMD_BX, e), S.ld(WL_S_CMD_BY, e), S.ld(WL_S_ACT0, e), S.ld(WL_S_ACT1, e), 0, true);
}

__global__ void __launch_bounds__(kBlock) elev_reset_kernel(const WlElevParams p, const WlEnvBuffers b, const HeightFieldGround ground,
                                                            const uint8_t* __restrict__ mask, uint64_t seed, uint64_t step) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= b.n_envs) return;
    if (mask && !mask[e]) return;
    const Rows S = make_rows(b.state, b.stride);
    const ElevReset rd = draw_elev_reset(p, ground, (uint32_t)(b.env_offset + e), step, seed);
    st3(S, WL_S_PX, e, rd.pos);
    S.st(WL_S_QW, e, rd.q.w);
    S.st(WL_S_QX, e, rd.q.x);
    S.st(WL_S_QY, e, rd.q.y);
    S.st(WL_S_QZ, e, rd.q.z);
    st3(S, WL_S_VX, e, v3(rd.vx, rd.vy, 0.f));
    st3(S, WL_S_WX, e, v3(0.f, 0.f, 0.f));
    S.st(WL_S_ACT0, e, 0.f);
    S.st(WL_S_ACT1, e, 0.f);
#pragma unroll
    for (int i = 0; i < WL_MAX_REW_TERMS; ++i) S.st(WL_S_EPSUM0 + i, e, 0.f);
    S.st(WL_S_TGT_X, e, rd.tgt_x);
    S.st(WL_S_TGT_Y, e, rd.tgt_y);
    S.st(WL_S_TGT_H, e, rd.tgt_h);
    S.st(WL_S_CMD_TIMER, e, p.cmd_resample_s);
    float c, sn;
    yaw_cs(rd.q, c, sn);
    const float dx = rd.tgt_x - rd.pos.x, dy = rd.tgt_y - rd.pos.y;
    S.st(WL_S_CMD_BX, e, fmaf(c, dx, sn * dy));
    S.st(WL_S_CMD_BY, e, fmaf(-sn, dx, c * dy));
    b.episode_len[e] = 0;
}

__global__ void __launch_bounds__(kBlock) elev_mdp_kernel(const WlElevParams p, int n, int64_t stride, const float* __restrict__ pos,
                                                          const float* __restrict__ quat, const float* __restrict__ vb_,
                                                          const float* __restrict__ vw_, const float* __restrict__ wheel,
                                                          const float* __restrict__ cmd, const uint8_t* __restrict__ timed_out,
                                                          int n_rays, const float* __restrict__ sensor_z,
                                                          const float* __restrict__ hit_z, float* __restrict__ terms,
                                                          uint8_t* __restrict__ flags, float* __restrict__ goal_rel,
                                                          float* __restrict__ hmap) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    const V3 P = v3(pos[e], pos[stride + e], pos[2 * stride + e]);
    const Quat q{quat[e], quat[stride + e], quat[2 * stride + e], quat[3 * stride + e]};
    const V3 vb = v3(vb_[e], vb_[stride + e], vb_[2 * stride + e]);
    const V3 vw = v3(vw_[e], vw_[stride + e], vw_[2 * stride + e]);
    const float ws = wheel[e] + wheel[stride + e] + wheel[2 * stride + e] + wheel[3 * stride + e];
    const float cbx = cmd[e], cby = cmd[stride + e];
    const Mat3 R = mat_from_quat(q);
    const ElevTerms tm = elev_terms(p, P, R.r2.z, vb, vw, ws, cbx, cby, timed_out ? timed_out[e] != 0 : false);
#pragma unroll
    for (int i = 0; i < WL_ER_NTERMS; ++i) terms[i * stride + e] = tm.t[i];
#pragma unroll
    for (int i = 0; i < WL_ET_NTERMS; ++i) flags[i * stride + e] = tm.flag[i] ? 1 : 0;
    const float gx = cbx - P.x, gy = cby - P.y;
    goal_rel[e] = gx != gx ? 0.f : gx;
    goal_rel[stride + e] = gy != gy ? 0.f : gy;
    if (hit_z) {
        const float sz = sensor_z[e];
        for (int k = 0; k < n_rays; ++k) hmap[k * stride + e] = -(sz - hit_z[k * stride + e] - p.scan_offset) + (P.z - p.elev_z0);
    }
}

int check_elev(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf) {
    if (!p || !b || !hf || !b->state || !b->episode_len || !b->metrics || !hf->height) return WL_EINVAL;
    if (b->n_envs <= 0 || b->stride < b->n_envs || b->metrics_slots < 1) return WL_EINVAL;
    if (b->stride % 64 != 0 || ((uintptr_t)b->state & 15u)) return WL_EALIGN;
    if (b->stride * 4 * WL_S_COUNT > 0x7fffffffLL || (b->lanes != 0 && b->lanes != 1 && b->lanes != 4)) return WL_EINVAL;
    if (!flags_ok(b)) return WL_EINVAL;
    if (p->decimation <= 0 || p->vehicle.substeps <= 0 || !(p->sim_dt > 0.f)) return WL_EINVAL;
    if (p->vehicle.implicit != 1 || !(p->vehicle.susp_fmax > 0.f)) return WL_EINVAL;   // these kernels step the linearly implicit integrator (wl_vehicle.h)
    if (hf->nx < 2 || hf->ny < 2 || !(hf->cell > 0.f) || !(hf->z_scale > 0.f && hf->z_scale < INFINITY)) return WL_EINVAL;
    if (!hf->pair || (int64_t)hf->nx * hf->ny * 4 > 0x7fffffffLL || hf->nx >= (1 << 23) || hf->ny >= (1 << 23)) return WL_EINVAL;   // wl_heightfield_pairs
    if ((uintptr_t)hf->pair & 3u) return WL_EALIGN;
    return WL_OK;
}

// pair[j][i] = code[j][i] | code[min(j + 1, ny - 1)][i] << 16
__global__ void __launch_bounds__(256) heightfield_pairs_kernel(const int16_t* __restrict__ height, uint32_t* __restrict__ pair, int nx, int ny) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (int64_t)nx * ny) return;
    const int64_t up = k + nx < (int64_t)nx * ny ? k + nx : k;
    pair[k] = (uint32_t)(uint16_t)height[k] | (uint32_t)(uint16_t)height[up] << 16;
}


}  // namespace

extern "C" {
#if WL_FUSED_TIMELINE
int wl_debug_fused_timeline(unsigned long long* host_dst /* [2048][16] */) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(wl_timeline), sizeof(unsigned long long) * 2048 * 16);
}
#endif

int wl_heightfield_pairs(const WlHeightField* hf, uint32_t* pair_out, void* stream) {
    if (!hf || !hf->height || !pair_out || hf->nx < 2 || hf->ny < 2) return WL_EINVAL;
    if ((uintptr_t)pair_out & 3u) return WL_EALIGN;
    clear_error();
    const int64_t n = (int64_t)hf->nx * hf->ny;
    heightfield_pairs_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(hf->height, pair_out, hf->nx, hf->ny);
    return launch_status();
}

int wl_elev_step(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* actions,
                 const WlStepOut* out, uint64_t seed, uint64_t step, void* stream) {
    return wl_elev_rollout(p, b, hf, actions, out, 0, 0, 1, seed, step, stream);
}

int wl_elev_rollout(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* actions,
                    const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps, uint64_t seed,
                    uint64_t step0, void* stream) {
    int rc = check_elev(p, b, hf);
    if (rc != WL_OK) return rc;
    if (!actions || !out || !out->obs || !out->reward || !out->terminated || !out->truncated || n_steps < 0) return WL_EINVAL;
    const HeightFieldGround g = make_ground(hf);
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
    // step + scan in one launch up to three 16-env blocks per CU; beyond, the lane-form step + the scan launch (round 6, us per step, fused /
    // two launches, profiles/r06_fused_crossover.txt: 8192 envs 30.9 / 43.8, 12 288: 44.8 / 48.5, 16 384: 58.9 / 52.4, 32 768: 114.5 / 72.2;
    // round 4: 8192 envs 48.5 fused; 16 384 envs 92.5 / 75.4)
    const bool quad = use_quad(b) && (b->lanes == 4 || b->n_envs <= WL_ELEV_FUSED_MAX_ENVS);
    clear_error();
    for (int k = 0; k < n_steps; ++k) {
        WlStepOut o = *out;
        o.obs += k * obs_step_stride;
        o.reward += k * vec_step_stride;
        o.terminated += k * vec_step_stride;
        o.truncated += k * vec_step_stride;
        if (o.dones) o.dones += k * vec_step_stride;
        const float2* a = (const float2*)(actions + (int64_t)k * b->n_envs * 2);
        if (quad) {   // step + scan in one launch
            elev_step_scan_kernel<false><<<(b->n_envs + kFusedEnvs - 1) / kFusedEnvs, kFusedThreads, 0, (hipStream_t)stream>>>(*p, vd, *b, g, a, o, seed, step0 + (uint64_t)k, PolicyIo{});
        } else {
            elev_step_kernel<1><<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, vd, *b, g, a, o, seed, step0 + (uint64_t)k);
            launch_elev_scan(p, b, g, o.obs, (hipStream_t)stream);
        }
    }
    return launch_status();
}

int wl_elev_collect_step(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const WlMlp* actor, const WlMlp* critic,
                         const float* std, const WlCollectIo* io, const WlStepOut* out, int32_t deterministic, uint64_t seed,
                         uint64_t step, void* stream) {
    int rc = check_elev(p, b, hf);
    if (rc != WL_OK) return rc;
    if (!use_quad(b)) return WL_EINVAL;   // the one-launch collector is the quad form's (n <= 32 768); beyond: act + step
    if (!actor || !critic || !std || !io || !io->obs_in || !io->actions || !io->mu || !io->log_prob || !io->values) return WL_EINVAL;
    if (!out || !out->obs || !out->reward || !out->terminated || !out->truncated) return WL_EINVAL;
    for (const WlMlp* m : {actor, critic})
        if (!m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3 || m->hidden != kMlpHidden || m->in_dim != WL_ELEV_OBS_DIM ||
            (m->activation != WL_ACT_ELU && m->activation != WL_ACT_RELU))
            return WL_EINVAL;
    if (actor->out_dim != 2 || critic->out_dim != 1 || actor->activation != critic->activation) return WL_EINVAL;
    if (((uintptr_t)io->actions & 7u) || ((uintptr_t)io->mu & 7u) || ((uintptr_t)io->obs_in & 3u)) return WL_EALIGN;
    const HeightFieldGround g = make_ground(hf);
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
    const PolicyIo pio{*actor, *critic, std, io->obs_in, io->actions, io->mu, io->log_prob, io->values, deterministic};
    const int grid = (b->n_envs + kFusedEnvs - 1) / kFusedEnvs;
    clear_error();
    if (actor->activation == WL_ACT_ELU)
        elev_step_scan_kernel<true, WL_ACT_ELU><<<grid, kFusedThreads, 0, (hipStream_t)stream>>>(*p, vd, *b, g, nullptr, *out, seed, step, pio);
    else
        elev_step_scan_kernel<true, WL_ACT_RELU><<<grid, kFusedThreads, 0, (hipStream_t)stream>>>(*p, vd, *b, g, nullptr, *out, seed, step, pio);
    return launch_status();
}

int wl_elev_collect_rollout(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const WlMlp* actor, const WlMlp* critic,
                            const float* std, const WlCollectIo* io, const WlStepOut* out, int32_t n_steps, int32_t deterministic,
                            uint64_t seed, uint64_t step0, void* stream) {
    int rc = check_elev(p, b, hf);
    if (rc != WL_OK) return rc;
    if (!use_quad(b)) return WL_EINVAL;   // the quad form's (n <= 32 768); beyond: act + step
    if (!actor || !critic || !std || !io || !io->obs_in || !io->actions || !io->mu || !io->log_prob || !io->values || n_steps < 0) return WL_EINVAL;
    if (!out || !out->obs || !out->reward || !out->terminated || !out->truncated) return WL_EINVAL;
    for (const WlMlp* m : {actor, critic})
        if (!m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3 || m->hidden != kMlpHidden || m->in_dim != WL_ELEV_OBS_DIM ||
            (m->activation != WL_ACT_ELU && m->activation != WL_ACT_RELU))
            return WL_EINVAL;
    if (actor->out_dim != 2 || critic->out_dim != 1 || actor->activation != critic->activation) return WL_EINVAL;
    if (((uintptr_t)io->actions & 7u) || ((uintptr_t)io->mu & 7u) || ((uintptr_t)io->obs_in & 3u)) return WL_EALIGN;
    // rows k of a [K + 1][n][689] observation block: the kernel writes row k + 1 where the caller's policy would read it
    if (out->obs != io->obs_in + (int64_t)b->n_envs * WL_ELEV_OBS_DIM) return WL_EINVAL;
    if (b->metrics_slots > 1 && n_steps % b->metrics_slots == 0 && n_steps > 0) return WL_EINVAL;   // ring slot aliasing
    const HeightFieldGround g = make_ground(hf);
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
    const PolicyIo pio{*actor, *critic, std, io->obs_in, io->actions, io->mu, io->log_prob, io->values, deterministic};
    const int grid = (b->n_envs + kFusedEnvs - 1) / kFusedEnvs;
    const size_t lds_bytes = (size_t)kColLdsFloats * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)elev_collect_rollout_kernel<WL_ACT_ELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void*)elev_collect_rollout_kernel<WL_ACT_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    const MetricSlots ms = metric_slots(b, step0, (uint64_t)n_steps);
    clear_error();
    if (actor->activation == WL_ACT_ELU)
        elev_collect_rollout_kernel<WL_ACT_ELU><<<grid, kFusedThreads, lds_bytes, (hipStream_t)stream>>>(*p, vd, *b, g, *out, n_steps, seed, step0, pio, ms);
    else
        elev_collect_rollout_kernel<WL_ACT_RELU><<<grid, kFusedThreads, lds_bytes, (hipStream_t)stream>>>(*p, vd, *b, g, *out, n_steps, seed, step0, pio, ms);
    return launch_status();
}

int wl_elev_rollout_persistent(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* actions,
                               const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps, uint64_t seed,
                               uint64_t step0, void* stream) {
    int rc = check_elev(p, b, hf);
    if (rc != WL_OK) return rc;
    if (!use_quad(b)) return WL_EINVAL;   // the quad form's (n <= 32 768)
    if (!actions || !out || !out->obs || !out->reward || !out->terminated || !out->truncated || n_steps < 0) return WL_EINVAL;
    if (n_steps > 1 && obs_step_stride < (int64_t)b->n_envs * WL_ELEV_OBS_DIM) return WL_EINVAL;   // the scan runs a step behind: rows must differ
    if (b->metrics_slots > 1 && n_steps % b->metrics_slots == 0 && n_steps > 0) return WL_EINVAL;   // ring slot aliasing
    clear_error();
    elev_rollout_persistent_kernel<<<(b->n_envs + kFusedEnvs - 1) / kFusedEnvs, kFusedThreads, 0, (hipStream_t)stream>>>(
        *p, derive_vehicle(p->vehicle, p->sim_dt, p->decimation), *b, make_ground(hf), (const float2*)actions, *out, obs_step_stride,
        vec_step_stride, n_steps, seed, step0, metric_slots(b, step0, (uint64_t)n_steps));
    return launch_status();
}

int wl_elev_reset(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const uint8_t* mask, uint64_t seed,
                  uint64_t step, void* stream) {
    int rc = check_elev(p, b, hf);
    if (rc != WL_OK) return rc;
    clear_error();
    elev_reset_kernel<<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, *b, make_ground(hf), mask, seed, step);
    return launch_status();
}

int wl_elev_observe(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, float* obs, void* stream) {
    int rc = check_elev(p, b, hf);
    if (rc != WL_OK) return rc;
    if (!obs) return WL_EINVAL;
    clear_error();
    elev_prop_kernel<<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*p, *b, obs);
    launch_elev_scan(p, b, make_ground(hf), obs, (hipStream_t)stream);
    return launch_status();
}

int wl_elev_mdp(const WlElevParams* p, int32_t n, int64_t stride, const float* pos, const float* quat, const float* lin_vel_b,
                const float* lin_vel_w, const float* wheel_vel, const float* command, const uint8_t* timed_out, int32_t n_rays,
                const float* sensor_z, const float* hit_z, float* terms, uint8_t* flags, float* goal_rel, float* height_map,
                void* stream) {
    if (!p || n <= 0 || stride < n || !pos || !quat || !lin_vel_b || !lin_vel_w || !wheel_vel || !command || !terms || !flags ||
        !goal_rel)
        return WL_EINVAL;
    if (hit_z && (!sensor_z || !height_map || n_rays <= 0)) return WL_EINVAL;
    clear_error();
    elev_mdp_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(*p, n, stride, pos, quat, lin_vel_b, lin_vel_w, wheel_vel,
                                                                      command, timed_out, n_rays, sensor_z, hit_z, terms, flags,
                                                                      goal_rel, height_map);
    return launch_status();
}

}  // extern "C"

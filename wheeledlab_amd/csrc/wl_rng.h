// wl_rng.h -- Philox4x32 counter RNG (Salmon et al. SC'11), WL_PHILOX_ROUNDS rounds.  key = (seed_lo, seed_hi),
// counter = (global env id, step_lo, step_hi, stream id).  Stateless: no RNG state lives in HBM.
// Rounds: 7 -- the smallest round count of Philox4x32 the paper reports Crush-resistant (its Table 2; 10 is the library's default with
// a safety margin).  32 x 32 -> 64-bit multiplies are quarter-rate on gfx950 and the generator was ~25 % of the drift step's vector
// instruction slots at large N: 10 -> 7 rounds is worth 3 - 4 % of the launch there (profiles/r04_rng_probe.jsonl).  Pinned against the
// Random123 distribution's known answers for 7 AND 10 rounds (tests/test_oracle_golden_drift.py; the device: tests/test_gpu_drift_parity.py).
#pragma once
#include "wl_math.h"

// drift task: WL_RS_DRIFT_EVENTS (reset pose + re-armed timers + low-frequency push), WL_RS_NOISE0 (observation normals 0..7),
// WL_RS_NOISE1 (normals 8..11 + high-frequency push): layout in wl_drift_env.h.  The other tasks' reset draws use streams 0..2 of their own
// enums (wl_elev.hip ES_*, wl_visual.hip).  Ids 1..3 and 6 are free since the drift draws were packed (round 4).
enum WlRngStream : uint32_t { WL_RS_DRIFT_EVENTS = 0, WL_RS_NOISE0 = 4, WL_RS_NOISE1 = 5, WL_RS_POLICY = 7, WL_RS_STARTUP = 8,
                            WL_RS_STARTUP_BUCKET = 9, WL_RS_STARTUP_WHEELS = 10 };

struct U4 {
    uint32_t x, y, z, w;
};
struct F4 {
    float x, y, z, w;
};

#ifndef WL_PHILOX_ROUNDS
#define WL_PHILOX_ROUNDS 7
#endif
WL_DEV U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < WL_PHILOX_ROUNDS; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32): 32-bit integer multiplies are quarter-rate on gfx950, so
        // separate mul_hi / mul_lo would double the dominant cost of the generator
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // (gfx950 has no v_xor3_b32: the assembler rejects it for this target)
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

WL_DEV float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }  // [0,1)

WL_DEV F4 philox_uniform4(uint32_t env, uint64_t step, uint32_t stream, uint64_t seed) {
    U4 r = philox4x32(env, (uint32_t)step, (uint32_t)(step >> 32), stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    return F4{u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
}

// the raw block, for draws that want 16-bit uniforms: EIGHT per block, (h + 1/2) / 65536 in (0, 1) -- both halves of a word (exact in
// fp32 either as an fma or as multiply + add).  Position noise of +-0.5 m resolves to 15 um, a timer interval to microseconds.
WL_DEV U4 philox_block(uint32_t env, uint64_t step, uint32_t stream, uint64_t seed) {
    return philox4x32(env, (uint32_t)step, (uint32_t)(step >> 32), stream, (uint32_t)seed, (uint32_t)(seed >> 32));
}
WL_DEV float u16_lo(uint32_t w) { return fmaf((float)(w & 0xFFFFu), 1.52587890625e-05f, 7.62939453125e-06f); }
WL_DEV float u16_hi(uint32_t w) { return fmaf((float)(w >> 16), 1.52587890625e-05f, 7.62939453125e-06f); }
// halves of two words as four uniforms: (lo a, hi a, lo b, hi b)
WL_DEV F4 u16x4(uint32_t a, uint32_t b) { return F4{u16_lo(a), u16_hi(a), u16_lo(b), u16_hi(b)}; }

// two standard normals from two uniforms (Box-Muller)
WL_DEV void box_muller(float u0, float u1, float& z0, float& z1) {
    const float r = fsqrt(-2.f * log_fast(1.f - u0));
    float s, c;
    sincos_rev(u1, s, c);  // angle = 2 pi u1 is exactly u1 revolutions
    z0 = r * c;
    z1 = r * s;
}
// the same from uniforms in the OPEN interval (u16_*): radius from u0 itself (|z| <= 4.8)
WL_DEV void box_muller_open(float u0, float u1, float& z0, float& z1) {
    const float r = fsqrt(-2.f * log_fast(u0));
    float s, c;
    sincos_rev(u1, s, c);
    z0 = r * c;
    z1 = r * s;
}

// wl_rng.h -- Philox4x32-10 counter RNG (Salmon et al. SC'11).  key = (seed_lo, seed_hi),
// counter = (global env id, step_lo, step_hi, stream id).  Stateless: no RNG state lives in HBM.
#pragma once
#include "wl_math.h"

enum WlRngStream : uint32_t { WL_RS_RESET = 0, WL_RS_TIMERS = 1, WL_RS_PUSH_HF = 2, WL_RS_PUSH_LF = 3, WL_RS_NOISE0 = 4 /* ..6 */, WL_RS_POLICY = 7,
                            WL_RS_STARTUP = 8, WL_RS_STARTUP_BUCKET = 9, WL_RS_STARTUP_WHEELS = 10 };

struct U4 {
    uint32_t x, y, z, w;
};
struct F4 {
    float x, y, z, w;
};

WL_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
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
    U4 r = philox4x32_10(env, (uint32_t)step, (uint32_t)(step >> 32), stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    return F4{u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
}

// two standard normals from two uniforms (Box-Muller)
WL_DEV void box_muller(float u0, float u1, float& z0, float& z1) {
    const float r = fsqrt(-2.f * log_fast(1.f - u0));
    float s, c;
    sincos_rev(u1, s, c);  // angle = 2 pi u1 is exactly u1 revolutions
    z0 = r * c;
    z1 = r * s;
}

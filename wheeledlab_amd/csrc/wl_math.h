// wl_math.h -- small device math for the env kernels (gfx950).  One lane = one env; everything lives in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WL_DEV __device__ __forceinline__

struct V3 {
    float x, y, z;
};
WL_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
WL_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
WL_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
WL_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
WL_DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
// (sums of two products are spelled out as fma(a, b, -(c * d)): under -ffp-contract=fast the compiler otherwise picks WHICH product
// it fuses per inlining site, and kernels that must agree bit for bit -- the persistent rollouts with the stepping launches -- do not)
WL_DEV V3 cross(V3 a, V3 b) {
    return V3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
WL_DEV V3 fma3(float s, V3 a, V3 b) { return V3{fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)}; }

// hardware reciprocal / rsqrt / sqrt (1 ulp class): the path is not IEEE-division sensitive, parity budget is 1e-5
WL_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
WL_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
WL_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// clamp(x, lo, hi) for lo <= hi is the median of the three: ONE v_med3_f32 (fminf(fmaxf()) costs 3-4 instructions: the
// IEEE min / max pair plus a v_max x, x canonicalisation of each non-constant operand)
WL_DEV float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// hardware sin / cos take REVOLUTIONS (v_sin_f32 / v_cos_f32, ~1e-6 abs): used where the argument is a bounded angle
#define WL_INV_TWO_PI 0.15915494309189533577f
WL_DEV void sincos_rev(float rev, float& s, float& c) {
    s = __builtin_amdgcn_sinf(rev);
    c = __builtin_amdgcn_cosf(rev);
}
WL_DEV void sincos_fast(float rad, float& s, float& c) { sincos_rev(rad * WL_INV_TWO_PI, s, c); }
// tan of a bounded steering angle (|x| <= 0.5 rad in every registered task) from the hardware sin / cos
WL_DEV float tan_fast(float rad) {
    float s, c;
    sincos_fast(rad, s, c);
    return s * __builtin_amdgcn_rcpf(c);
}
WL_DEV float log_fast(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }  // v_log_f32 is log2

struct Quat {
    float w, x, y, z;
};

// rotation matrix rows (body -> world) of a unit quaternion
struct Mat3 {
    V3 r0, r1, r2;
};
WL_DEV Mat3 mat_from_quat(Quat q) {
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    const float dz = 1.f - q.z * z2, dy = 1.f - q.y * y2;   // (one product each: nothing to choose)
    Mat3 m;
    m.r0 = v3(fmaf(-q.y, y2, dz), fmaf(q.x, y2, -wz), fmaf(q.x, z2, wy));
    m.r1 = v3(fmaf(q.x, y2, wz), fmaf(-q.x, x2, dz), fmaf(q.y, z2, -wx));
    m.r2 = v3(fmaf(q.x, z2, -wy), fmaf(q.y, z2, wx), fmaf(-q.x, x2, dy));
    return m;
}
WL_DEV V3 mul(const Mat3& m, V3 v) { return v3(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }
WL_DEV V3 mul_t(const Mat3& m, V3 v) {
    return v3(fmaf(m.r0.x, v.x, fmaf(m.r1.x, v.y, m.r2.x * v.z)), fmaf(m.r0.y, v.x, fmaf(m.r1.y, v.y, m.r2.y * v.z)),
              fmaf(m.r0.z, v.x, fmaf(m.r1.z, v.y, m.r2.z * v.z)));
}

#define WL_TWO_PI 6.28318530717958647692f
#define WL_PI 3.14159265358979323846f

// python-style modulo into [0, 2pi) for |a| <= pi (atan2 / asin ranges)
WL_DEV float wrap_2pi(float a) {
    return a < 0.f ? a + WL_TWO_PI : a;  // == torch.remainder(a, 2pi) on this range (may round to 2pi itself)
}

// atan2 for finite arguments: octant reduction to t = min / max in [0, 1], atan t = t + t^3 P(t^2) (degree-7 least-squares
// fit on Chebyshev nodes: 1.3e-7 abs in fp32 Horner form, rcp at 1 ulp), unfolded by value.  ~22 instructions against
// ~50 for libm's atan2f (exact division through frexp / ldexp, inf / NaN / signed-zero cases); four of them per env-step.
WL_DEV float atan2_fast(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * rcp(fmaxf(mx, 1e-37f));   // (0, 0) -> 0
    const float u = t * t;
    float q = fmaf(u, 0.003962155897170305f, -0.020364457741379738f);
    q = fmaf(u, q, 0.04938491806387901f);
    q = fmaf(u, q, -0.08042293787002563f);
    q = fmaf(u, q, 0.10877407342195511f);
    q = fmaf(u, q, -0.14259037375450134f);
    q = fmaf(u, q, 0.19998809695243835f);
    q = fmaf(u, q, -0.33333325386047363f);
    float r = fmaf(t * u, q, t);
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.f ? WL_PI - r : r;
    return copysignf(r, y);
}

// IsaacLab euler_xyz_from_quat (un-vendored; reference call site wheeledlab/envs/mdp/observations.py:11).
// asin x = atan2(x, sqrt(1 - x^2)), which is +-pi/2 at |x| >= 1 like the reference's clamp.
WL_DEV V3 euler_xyz_from_quat(Quat q) {
    const float roll = atan2_fast(2.f * (q.w * q.x + q.y * q.z), 1.f - 2.f * (q.x * q.x + q.y * q.y));
    const float sp = 2.f * (q.w * q.y - q.z * q.x);
    const float pitch = atan2_fast(sp, fsqrt(fmaxf(fmaf(-sp, sp, 1.f), 0.f)));
    const float yaw = atan2_fast(2.f * (q.w * q.z + q.x * q.y), 1.f - 2.f * (q.y * q.y + q.z * q.z));
    return v3(wrap_2pi(roll), wrap_2pi(pitch), wrap_2pi(yaw));
}

// a value the optimiser cannot look through or move into a branch: the operands of a by-value selection stay plain
// VALU results and the selection a v_cndmask (LLVM otherwise sinks an expensive operand -- a sqrt, a chain of fmas --
// into a divergent branch around the select: exec-mask save / restore plus a branch per selection)
WL_DEV float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

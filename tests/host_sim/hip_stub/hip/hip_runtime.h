// TEST INFRASTRUCTURE ONLY.  A stand-in for <hip/hip_runtime.h> that lets the DEVICE headers of wheeledlab_amd/csrc
// (wl_math.h, wl_vehicle.h, wl_heightfield.h, wl_drift_terms.h) be compiled by the host g++ so that their arithmetic
// can be checked against the numpy oracle in the CPU test suite (tests/test_host_sim_cpu.py).  It is on the include
// path of that one test build and nowhere else; the product has no CPU path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define WL_HOST_SIM 1
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__

static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }
static inline float __builtin_amdgcn_sinf(float rev) { return (float)std::sin(6.283185307179586 * (double)rev); }   // v_sin_f32 takes revolutions
static inline float __builtin_amdgcn_cosf(float rev) { return (float)std::cos(6.283185307179586 * (double)rev); }
static inline float __builtin_amdgcn_logf(float x) { return std::log2(x); }                                         // v_log_f32 is log2
static inline int __builtin_amdgcn_update_dpp(int, int v, int, int, int, bool) { return v; }   // quad forms are not simulated
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return p ? 1ull : 0ull; }   // a wavefront of one lane
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {     // v_alignbit_b32: ({hi, lo} >> sh[4:0])[31:0]
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31u));
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }

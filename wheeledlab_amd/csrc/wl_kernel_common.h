// wl_kernel_common.h -- helpers shared by the per-task translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"

namespace {

constexpr int kBlock = 256;

struct Rows {  // row accessor of the SoA state matrix
    float* base;
    int64_t stride;
    WL_DEV float& operator()(int row, int env) const { return base[row * stride + env]; }
};

WL_DEV V3 ld3(const Rows& s, int row, int e) { return v3(s(row, e), s(row + 1, e), s(row + 2, e)); }
WL_DEV void st3(const Rows& s, int row, int e, V3 v) {
    s(row, e) = v.x;
    s(row + 1, e) = v.y;
    s(row + 2, e) = v.z;
}

inline int grid_for(int n) { return (n + kBlock - 1) / kBlock; }
// Step kernels come in two forms (wl_vehicle.h): lane-per-env (no redundant work: best when the chip is full) and
// quad-per-env (one wheel per lane: shorter critical path, 4x the waves: best while the chip is under-filled).
// 256 CUs x 4 SIMDs = 1024 wave slots at one wave per SIMD; quads pay off up to a few waves per SIMD.
#ifndef WL_QUAD_MAX_ENVS
#define WL_QUAD_MAX_ENVS 32768
#endif
inline bool use_quad(int n_envs) {
    static const char* force = getenv("WL_FORCE_LANES");   // "1" / "4": testing hook
    if (force && force[0] == '1') return false;
    if (force && force[0] == '4') return true;
    return n_envs <= WL_QUAD_MAX_ENVS;
}
// The host process (PyTorch) may leave a benign sticky error (e.g. hipErrorNotReady from an event query) in this
// thread's HIP error slot: clear it before the launch so launch_status() reports only our own launch.
inline void clear_error() { (void)hipGetLastError(); }
inline int launch_status() { return hipGetLastError() == hipSuccess ? WL_OK : WL_ELAUNCH; }


}  // namespace

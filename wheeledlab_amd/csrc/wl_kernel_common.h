// wl_kernel_common.h -- helpers shared by the per-task translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"

namespace {

constexpr int kBlock = 256;

// Row accessor of the SoA state matrix through a buffer resource: every access is `buffer_load/store_dword` with ONE
// per-lane byte offset (4 * env, a single VGPR for the whole kernel) and a scalar row offset (row * stride * 4, SALU).
// With flat `global_*` addressing the compiler materialises a 64-bit VGPR address per row and keeps ~40 of them live
// across the physics loop; the buffer form removes ~80 VGPRs and the 64-bit address arithmetic.
struct Rows {
    __amdgpu_buffer_rsrc_t rsrc;
    int row_bytes;   // stride * 4
    // streaming: the batch is far larger than the caches (state matrix > 192 MB: the drift step beyond 1.2 M envs): rows written now are not
    // read again before they are evicted, so their stores carry the `sc1 nt` policy (cache-policy operand 18: bit 1 nt, bit 4
    // sc1).  Measured on the step's own pattern (34 SoA rows in, 30 out, 4 M envs; tools/microbench/layout_bw): 5.24 TB/s with
    // default stores, 6.15 TB/s with sc1 nt.  A compile-time constant wherever it is used (set from a template parameter,
    // everything inlined): the untaken form is dropped.
    bool streaming;
    WL_DEV float ld(int row, int env) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, env * 4, row * row_bytes, 0));
    }
    WL_DEV void st(int row, int env, float v) const {
        if (streaming) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, env * 4, row * row_bytes, 18);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, env * 4, row * row_bytes, 0);
    }
    // row index that differs per LANE (quad form: lane `wid` owns wheel row WL_S_WHEEL_BL + wid): the row goes into the
    // per-lane byte offset.  (Through ld / st the row would be the SCALAR offset, and a lane-varying scalar operand
    // makes the compiler emit a readfirstlane "waterfall" loop: up to four serialised passes around one load.)
    // The product is a 24-bit multiply (row < 64, row_bytes < 2^24: the C-ABI wrappers refuse the quad form beyond that):
    // written as `row * row_bytes + env * 4` the compiler emits v_mad_u64_u32 with a 64-bit addend PAIR whose upper
    // register it shares with a pending load's destination -- a false dependency that parked the wavefront on
    // s_waitcnt vmcnt in the middle of its load burst (one extra memory round trip per launch).
    WL_DEV int lane_row_offset(int row_lane, int env) const { return (int)__umul24((unsigned)row_lane, (unsigned)row_bytes) + env * 4; }
    WL_DEV float ld_lane_row(int row_lane, int env) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_row_offset(row_lane, env), 0, 0));
    }
    WL_DEV void st_lane_row(int row_lane, int env, float v) const {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, lane_row_offset(row_lane, env), 0, 0);
    }
};
WL_DEV Rows make_rows(float* base, int64_t stride, bool streaming = false) {
    // dword3 0x00020000: raw 32-bit data format on gfx90a / gfx94x / gfx950; num_records bounds the whole matrix
    return Rows{__builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(stride * 4 * WL_S_COUNT), 0x00020000), (int)(stride * 4), streaming};
}

WL_DEV V3 ld3(const Rows& s, int row, int e) { return v3(s.ld(row, e), s.ld(row + 1, e), s.ld(row + 2, e)); }
WL_DEV void st3(const Rows& s, int row, int e, V3 v) {
    s.st(row, e, v.x);
    s.st(row + 1, e, v.y);
    s.st(row + 2, e, v.z);
}

// Kernel arguments live in memory the host rewrites for every launch and the GPU reads uncached; the compiler fetches
// them with scalar loads in as many dependent batches as the 102-entry SGPR file forces (six `s_load ... s_waitcnt`
// round trips in the step kernel's prologue, each a trip to HBM).  The latency-bound form instead copies the whole
// parameter block with VECTOR loads at a lane-uniform address: one batch, issued from the kernarg pointer the wavefront
// is born with, straight into VGPRs (where float parameters are consumed anyway).  Integer fields that steer scalar
// control flow are then taken from the scalar copy again so that branches and loop bounds stay uniform.
template <int N>
WL_DEV void kernarg_vector_words(int byte_offset, uint32_t (&w)[N]) {
    // volatile: the requests stay where they are written (constant-address-space loads are otherwise sunk to their
    // first use, block by block -- each batch another kernarg round trip on the critical path).  On gfx950 a volatile load
    // carries sc0 sc1; the argument block is written by the host for this launch and read once: nothing to lose.
    using KArg = const volatile __attribute__((address_space(4))) uint32_t;
    KArg* ka = (KArg*)__builtin_amdgcn_kernarg_segment_ptr();
    int z = 0;
    asm volatile("" : "+v"(z));   // a VGPR zero the compiler cannot fold: keeps these loads on the vector path
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
    using KArg4 = const volatile __attribute__((address_space(4))) u32x4;
    constexpr int N4 = N / 4;   // 16-byte requests (a volatile access is never merged: dword by dword it would be N requests)
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        const u32x4 q = *(KArg4*)(ka + byte_offset / 4 + 4 * i + z);
        w[4 * i] = q.x, w[4 * i + 1] = q.y, w[4 * i + 2] = q.z, w[4 * i + 3] = q.w;
    }
#pragma unroll
    for (int i = 4 * N4; i < N; ++i) w[i] = ka[byte_offset / 4 + i + z];
}
template <class T>
WL_DEV T kernarg_vector_copy(int byte_offset) {
    static_assert(sizeof(T) % 4 == 0, "dword POD");
    uint32_t w[sizeof(T) / 4];
    kernarg_vector_words(byte_offset, w);
    T t;
    __builtin_memcpy(&t, w, sizeof(T));
    return t;
}
// Volatile requests cannot be dropped, so fields nobody reads are fetched too -- and a destination register that is
// dead on arrival is handed out again at once, which makes the NEXT writer of that register wait for the load to land
// (write-after-write): the draws that should run in the shadow of the burst stalled on its first instruction.  Touching
// every word where the block is first needed keeps all destinations allocated until then.
template <int N>
WL_DEV void kernarg_words_landed(uint32_t (&w)[N]) {
#pragma unroll
    for (int i = 0; i + 8 <= N; i += 8)
        asm volatile("" : "+v"(w[i]), "+v"(w[i + 1]), "+v"(w[i + 2]), "+v"(w[i + 3]), "+v"(w[i + 4]), "+v"(w[i + 5]), "+v"(w[i + 6]), "+v"(w[i + 7]));
#pragma unroll
    for (int i = N - N % 8; i < N; ++i) asm volatile("" : "+v"(w[i]));
}
// two ADJACENT argument structs in one burst from one address register (two separate copies made the second one's
// address registers collide with the first one's pending destinations: a wait in the middle of the burst)
template <class A, class B>
WL_DEV void kernarg_vector_copy2(int byte_offset, A& a, B& b) {
    static_assert(sizeof(A) % 4 == 0 && sizeof(B) % 4 == 0, "dword PODs");
    uint32_t w[(sizeof(A) + sizeof(B)) / 4];
    kernarg_vector_words(byte_offset, w);
    __builtin_memcpy(&a, w, sizeof(A));
    __builtin_memcpy(&b, w + sizeof(A) / 4, sizeof(B));
}
// Integer fields steer scalar control flow (loop bounds, uniform branches): they must live in SGPRs.  They arrive with
// the vector copy like everything else and are moved across with v_readfirstlane (the value is lane-uniform) -- NOT
// re-read from the scalar copy of the argument: those s_loads were issued where first used, each batch a further
// kernarg round trip (~1 us) on the launch's critical path.
WL_DEV int32_t uniform_i32(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// the integer fields every task's parameter struct has (WlDriftParams / WlElevParams / WlVisualParams)
template <class P>
WL_DEV void uniform_scalar_common(P& v) {   // from the vector copy, once it has landed
    v.decimation = uniform_i32(v.decimation);
    v.max_episode_length = uniform_i32(v.max_episode_length);
    v.action.bounding = uniform_i32(v.action.bounding);
    v.action.no_reverse = uniform_i32(v.action.no_reverse);
    v.action.clip_wrapper = uniform_i32(v.action.clip_wrapper);
    v.action.map = uniform_i32(v.action.map);
    v.vehicle.drive = uniform_i32(v.vehicle.drive);
    v.vehicle.substeps = uniform_i32(v.vehicle.substeps);
    v.log_episode_sums = uniform_i32(v.log_episode_sums);
}
template <class P>
WL_DEV void keep_scalar_common(P& v, const P& s) {   // from the scalar copy of the argument (s_load where first used)
    v.decimation = s.decimation;
    v.max_episode_length = s.max_episode_length;
    v.action.bounding = s.action.bounding;
    v.action.no_reverse = s.action.no_reverse;
    v.action.clip_wrapper = s.action.clip_wrapper;
    v.action.map = s.action.map;
    v.vehicle.drive = s.vehicle.drive;
    v.vehicle.substeps = s.vehicle.substeps;
    v.log_episode_sums = s.log_episode_sums;
}

// Metric accumulators: [slot][WL_M_SHARDS][WL_M_COUNT] (include/wheeledlab_amd.h).
constexpr int kMetricSlotFloats = WL_M_SHARDS * WL_M_COUNT;
WL_DEV float* metric_shard(const WlEnvBuffers& b, int slot) {   // this wavefront's shard of `slot`
    const int wave = blockIdx.x * ((int)blockDim.x >> 6) + (threadIdx.x >> 6);
    return b.metrics + (int64_t)slot * kMetricSlotFloats + (wave & (WL_M_SHARDS - 1)) * WL_M_COUNT;
}
WL_DEV void clear_metric_slot(const WlEnvBuffers& b, int slot) {   // block 0 zeroes all shards of `slot`
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < kMetricSlotFloats; i += (int)blockDim.x) b.metrics[(int64_t)slot * kMetricSlotFloats + i] = 0.f;
}

// How a GROUP of wavefronts of a block meets (the visual camera's render groups, the elevation collector's layer-1 wavefronts).
// BlockSync: the block's s_barrier -- every group of the block makes the same calls.  GroupSync: a counting barrier in LDS among the group's own wavefronts, for blocks in which other wavefronts
// do something else meanwhile (the persistent rollout's physics wavefront would have to join an s_barrier).  Spinning is
// safe: the wavefronts of a workgroup are always co-resident.  The counter only grows (target = arrivals so far).
struct BlockSync {
    WL_DEV void operator()() { __syncthreads(); }
};
struct GroupSync {
    int* cnt;
    int target, n_waves;
    WL_DEV void operator()() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        target += n_waves;
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

// The slot of the metric ring a launch accumulates into and the slot it clears for its successor, computed by the
// C-ABI wrapper: `step % slots` on a 64-bit step is ~110 scalar instructions in-kernel (no hardware divide), twice,
// on the latency-critical head and tail of a 7 us launch.
struct MetricSlots {
    int32_t cur, next;
};
inline MetricSlots metric_slots(const WlEnvBuffers* b, uint64_t step0, uint64_t n_steps = 1) {
    if (b->metrics_slots <= 1) return MetricSlots{0, 0};
    const uint64_t R = (uint64_t)b->metrics_slots;
    return MetricSlots{(int32_t)(step0 % R), (int32_t)((step0 + n_steps) % R)};
}

inline int grid_for(int n) { return (n + kBlock - 1) / kBlock; }
// Step kernels come in two forms (wl_vehicle.h): lane-per-env (no redundant work: best when the chip is full) and
// quad-per-env (one wheel per lane: shorter critical path, 4x the waves: best while the chip is under-filled).
// 256 CUs x 4 SIMDs = 1024 wave slots at one wave per SIMD; quads pay off up to a few waves per SIMD.
#ifndef WL_QUAD_MAX_ENVS
#define WL_QUAD_MAX_ENVS 32768
#endif
// lane form: the interleaved-wheels build up to here, the fenced low-register build beyond
#ifndef WL_UNROLL_MAX_ENVS
#define WL_UNROLL_MAX_ENVS 262144
#endif
#ifndef WL_LANE_WAVES
#define WL_LANE_WAVES 5     // wavefronts per SIMD the lane builds must leave room for (__launch_bounds__): interleaved ..
#endif
#ifndef WL_LOWREG_WAVES
#define WL_LOWREG_WAVES 5   // .. and fenced
#endif
// WlEnvBuffers.flags: valid combinations only (a flag and its opposite together are refused)
inline bool flags_ok(const WlEnvBuffers* b) {
    const int f = b->flags;
    if (f & ~WL_FLAG_MASK) return false;
    if ((f & WL_FLAG_STREAM) && (f & WL_FLAG_NO_STREAM)) return false;
    if ((f & WL_FLAG_SCAN_LDS) && (f & WL_FLAG_SCAN_GATHER)) return false;
    return true;
}
// streaming (non-temporal) stores: forced by the flags, otherwise when `bytes` (what the launch writes and will not read again
// before the caches turn over) exceed `threshold`
inline bool use_streaming(const WlEnvBuffers* b, int64_t bytes, int64_t threshold) {
    if (b->flags & WL_FLAG_STREAM) return true;
    if (b->flags & WL_FLAG_NO_STREAM) return false;
    return bytes > threshold;
}
inline bool use_unrolled(const WlEnvBuffers* b) {   // lane form only
    if (b->lanes == 1) return true;
    if (b->lanes == 2) return false;
    return b->n_envs <= WL_UNROLL_MAX_ENVS;
}
inline bool use_quad(const WlEnvBuffers* b) {
    if (b->lanes == 1 || b->lanes == 2) return false;
    if ((b->flags & WL_FLAG_STREAM) && b->lanes != 4) return false;   // the streaming instantiations are lane forms
    if (b->stride * 4 >= (1 << 24)) return false;   // Rows::lane_row_offset is a 24-bit multiply (4 M envs: never the quad regime)
    if (b->lanes == 4) return true;
    return b->n_envs <= WL_QUAD_MAX_ENVS;
}
// The host process (PyTorch) may leave a benign sticky error (e.g. hipErrorNotReady from an event query) in this
// thread's HIP error slot: clear it before the launch so launch_status() reports only our own launch.
inline void clear_error() { (void)hipGetLastError(); }
inline int launch_status() { return hipGetLastError() == hipSuccess ? WL_OK : WL_ELAUNCH; }


}  // namespace

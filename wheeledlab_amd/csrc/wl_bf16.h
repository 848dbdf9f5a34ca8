// wl_bf16.h -- f32 -> two bf16 planes (x = hi + lo to 16 mantissa bits), round to nearest even (v_cvt_pk_bf16_f32 on gfx950)
#pragma once
#include <hip/hip_runtime.h>

#include "wl_kernel_common.h"

namespace {

typedef __bf16 wl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wl_f32x2 __attribute__((ext_vector_type(2)));

// (a, b) -> packed pairs: low half = a's plane value, high half = b's
WL_DEV void split_bf16_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const wl_f32x2 v = {a, b};
    const wl_bf16x2 h = __builtin_convertvector(v, wl_bf16x2);
    const wl_bf16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, wl_f32x2), wl_bf16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// Layer-1 weights of both nets ([64][in_dim] each) -> planes [128][dp], dp = in_dim rounded up to 64.  K chunk c holds features
// 64 c .. 64 c + 63, except the LAST chunk of an in_dim that is not a multiple of 64: it holds the row's last 64 features,
// in_dim - 64 .. in_dim - 1, with the ones the previous chunk already covers set to zero.  The f32 operand of the contraction
// is then read at min(64 c, in_dim - 64): every load stays inside its row, at full width, with no padding of the rows.
WL_DEV float chunked_weight(const float* __restrict__ w, int in_dim, int dp, int pos) {
    const int c = pos >> 6, last = dp / 64 - 1;
    if (c < last || in_dim == dp) return w[pos];
    const int f = in_dim - 64 + (pos & 63);
    return f < 64 * last ? 0.f : w[f];
}
// pair i of positions (of 128 * dp / 2) of the planes
WL_DEV void weight_plane_pair(const float* __restrict__ w1_actor, const float* __restrict__ w1_critic, const int in_dim, const int dp,
                              const int i, uint32_t* __restrict__ w_hi, uint32_t* __restrict__ w_lo) {
    if (i >= 128 * dp / 2) return;
    const int u = i / (dp / 2), pos = 2 * (i - u * (dp / 2));
    const float* w = u < 64 ? w1_actor + (int64_t)u * in_dim : w1_critic + (int64_t)(u - 64) * in_dim;
    split_bf16_pair(chunked_weight(w, in_dim, dp, pos), chunked_weight(w, in_dim, dp, pos + 1), w_hi[i], w_lo[i]);
}

}  // namespace

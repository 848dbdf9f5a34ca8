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

}  // namespace

// wl_actor_dev.h -- device pieces of the wide-observation policy step shared by wl_actor.hip (the policy-step kernels) and
// the per-task collectors that run the policy inside the env's launch (wl_elev.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_kernel_common.h"
#include "wl_mlp.h"
#include "wl_rng.h"

namespace {

constexpr float kLog2PiA = 1.8378770664093453f;
typedef float wl_f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment (rows of odd length)

struct MlpTail {   // A operands of layers 2 and 3 (this lane's element of each 16 x 4 weight tile), as in MlpWeights
    float w2[kMlpTiles][kMlpHidSteps];
    float w3[kMlpHidSteps];
};

WL_DEV void load_tail(const WlMlp& net, int lane, MlpTail& W) {
    const int m = lane & 15, g = (lane >> 4) & 3;
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) {
        const int unit = 16 * t + m;
#pragma unroll
        for (int tp = 0; tp < kMlpTiles; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) W.w2[t][4 * tp + r] = net.w2[unit * kMlpHidden + 16 * tp + 4 * g + r];
        W.w2[t][kMlpHidSteps - 1] = g == 0 ? net.b2[unit] : 0.f;
    }
#pragma unroll
    for (int tp = 0; tp < kMlpTiles; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) W.w3[4 * tp + r] = m < net.out_dim ? net.w3[m * kMlpHidden + 16 * tp + 4 * g + r] : 0.f;
    W.w3[kMlpHidSteps - 1] = (g == 0 && m < net.out_dim) ? net.b3[m] : 0.f;
}

// layers 2 and 3 on the (pre-activation, bias included) layer-1 accumulators; lanes 0..15 return outputs 0..3 of row l
template <int ACT>
WL_DEV f32x4 eval_tail(const MlpTail& W, f32x4 h1[kMlpTiles], int lane) {
    const float one_g0 = ((lane >> 4) & 3) == 0 ? 1.f : 0.f;
    f32x4 h2[kMlpTiles];
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[t][r] = mlp_act<ACT>(h1[t][r]);
        h2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int tp = 0; tp < kMlpTiles; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < kMlpTiles; ++t) h2[t] = mfma4(W.w2[t][4 * tp + r], h1[tp][r], h2[t]);
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) h2[t] = mfma4(W.w2[t][kMlpHidSteps - 1], one_g0, h2[t]);
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[t][r] = mlp_act<ACT>(h2[t][r]);
    f32x4 out = {0.f, 0.f, 0.f, 0.f}, out_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < kMlpTiles; tp += 2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            out = mfma4(W.w3[4 * tp + r], h2[tp][r], out);
            out_b = mfma4(W.w3[4 * (tp + 1) + r], h2[tp + 1][r], out_b);
        }
    out = mfma4(W.w3[kMlpHidSteps - 1], one_g0, out);
    return out + out_b;
}

}  // namespace

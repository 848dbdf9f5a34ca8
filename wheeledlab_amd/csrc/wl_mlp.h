// wl_mlp.h -- the RSL-RL ActorCritic MLP of the drift agents (in -> 64 -> 64 -> out, ELU / ReLU;
// wheeledlab_tasks/drifting/config/agents/mushr/rsl_rl_ppo_cfg.py:12-17) on the f32 matrix pipe of gfx950.
//
// One wavefront evaluates the net for 16 rows (envs) with v_mfma_f32_16x16x4_f32 in the TRANSPOSED formulation
//     H^T[units x envs] = W[units x k] * X^T[k x envs]
// so that the accumulator layout of one layer (lane l: env = l & 15, units 4 (l >> 4) + r, r = register 0..3) IS the
// B-operand layout of the next layer (lane l: env = l & 15, k = l >> 4) once the k axis is walked in the order
// (input tile t', register r, lane group g) -> unit 16 t' + 4 g + r.  A dot product does not care about the order of its
// terms as long as both operands agree, and the weights (A operands, one f32 per lane each) are fetched from memory in
// whatever order we like -- so activations never leave their registers between layers: no LDS, no shuffles.
// Biases ride on one extra k-step per tile (B = 1 on lane group 0, A = bias there) for layers 2 and 3, and on the padded
// input feature `in_dim` (= 1.0) for layer 1.  The f32 MFMA is an exact fmaf chain, so results equal a plain fp32 MLP
// up to the summation order.
//
// Weights stay resident in registers (101 VGPRs/AGPRs per net): the kernels that use this run one wavefront per SIMD
// and own its whole 512-entry register file.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kMlpHidden = 64;
constexpr int kMlpTiles = kMlpHidden / 16;       // 16-unit output tiles per hidden layer
constexpr int kMlpInSteps = 4;                   // k-steps of layer 1: up to 15 input features + the bias feature
constexpr int kMlpHidSteps = kMlpTiles * 4 + 1;  // k-steps of layers 2 / 3: 64 units + the bias step

struct MlpWeights {   // A operands, this lane's element of each 16 x 4 weight tile
    float w1[kMlpTiles][kMlpInSteps];
    float w2[kMlpTiles][kMlpHidSteps];
    float w3[kMlpHidSteps];
};

// lane l of the wavefront: m = l & 15 (row of the 16-unit output tile), g = l >> 4 (k index within a k-step)
WL_DEV void mlp_load_weights(const WlMlp& net, int lane, MlpWeights& W) {
    const int m = lane & 15, g = (lane >> 4) & 3;
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) {
        const int unit = 16 * t + m;
#pragma unroll
        for (int s = 0; s < kMlpInSteps; ++s) {
            const int f = 4 * s + g;   // input feature; feature in_dim carries the bias, the rest of the padding is 0
            W.w1[t][s] = f < net.in_dim ? net.w1[unit * net.in_dim + f] : f == net.in_dim ? net.b1[unit] : 0.f;
        }
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

template <int ACT>
WL_DEV float mlp_act(float x) {
    if constexpr (ACT == WL_ACT_RELU) return fmaxf(x, 0.f);
    else return x > 0.f ? x : __builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.f;   // ELU, alpha = 1 (|err| ~ 1e-7)
}

WL_DEV f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// x[s]: B operand of layer-1 k-step s -- on lane l the input feature 4 s + (l >> 4) of row (env) l & 15, with feature
// in_dim == 1.0 and the remaining padding 0.  Returns the output accumulator: lanes 0..15 (g == 0) hold outputs
// 0..3 of row l in registers 0..3.  Must be executed by the whole wavefront (no divergence around it).
template <int ACT>
WL_DEV f32x4 mlp_eval(const MlpWeights& W, const float x[kMlpInSteps], int lane) {
    const float one_g0 = ((lane >> 4) & 3) == 0 ? 1.f : 0.f;
    // k-step outermost, output tile innermost: four independent accumulator chains are in flight, so no MFMA waits
    // on the 40-cycle dependent-accumulator latency of its predecessor
    f32x4 h1[kMlpTiles], h2[kMlpTiles];
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) h1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < kMlpInSteps; ++s)
#pragma unroll
        for (int t = 0; t < kMlpTiles; ++t) h1[t] = mfma4(W.w1[t][s], x[s], h1[t]);
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[t][r] = mlp_act<ACT>(h1[t][r]);
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) h2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    // the output layer is one 17-step chain; two partial accumulators halve its dependent latency
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

inline int check_mlp(const WlMlp* net) {
    if (!net || !net->w1 || !net->b1 || !net->w2 || !net->b2 || !net->w3 || !net->b3) return WL_EINVAL;
    if (net->hidden != kMlpHidden || net->in_dim < 1 || net->in_dim > 4 * kMlpInSteps - 1 || net->out_dim < 1 || net->out_dim > 4)
        return WL_EINVAL;
    if (net->activation != WL_ACT_RELU && net->activation != WL_ACT_ELU) return WL_EINVAL;
    return WL_OK;
}

}  // namespace

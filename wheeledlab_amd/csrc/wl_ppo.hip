// wl_ppo.hip -- one PPO minibatch step of the drift agents (14-64-64-2 actor, 14-64-64-1 critic; rsl_rl PPO.update as the
// reference drives it: wheeledlab_rl/utils/modified_rsl_rl_runner.py:104-109, hyper-parameters rsl_rl_ppo_cfg.py:18-31)
// as three launches: forward + losses + backward + weight gradients fused on the f32 matrix pipe, a partial-sum
// reduction, and gradient clipping + Adam + the adaptive-KL learning-rate rule.
//
// Gradient kernel, one wavefront per 16-sample tile (v_mfma_f32_16x16x4_f32, lane l: g = l >> 4, n = l & 15; round 2: the two
// 64 x 64 products per net -- layer 2 forward and delta1 = W2^T delta2 -- on v_mfma_f32_16x16x32_bf16 with split operands,
// see the F2 / B2 slabs below: 124 -> 99 us per 131 072-sample gradient call):
//   * forward in the transposed formulation of wl_mlp.h: H^T[unit][sample] = W . X^T, so a layer's accumulator (lane
//     (g, n): units 16 t + 4 g + r of sample n) is the next layer's B operand when k is walked as (tile t', register r);
//   * the output layer is ONE accumulator for both nets with its rows spread so that lane group g ends up holding
//     output g of sample n in register 0: g = 0 mu_0, g = 1 mu_1, g = 2 value.  The loss derivatives are computed right
//     there, and that register IS the B operand (k = g) of the backward product delta2^T = W3^T . delta3^T;
//   * delta1^T = W2^T . delta2^T uses the same layout trick as the forward pass (delta2's accumulator is its B operand);
//   * weight gradients dW = delta^T . H contract over the SAMPLES, which the accumulator layout keeps in l & 15 while
//     both MFMA operands want their k index in l >> 4: delta and H are transposed through a per-wavefront LDS buffer
//     ([unit][sample], 16 ds_write + 16 ds_read per lane and matrix) and then feed A and B with the same read pattern;
//     dW1 takes the observation rows straight from memory in B layout, its padded feature 14 (= 1) yields db1 for free;
//   * every weight operand is read from LDS (86 KB of operand-ordered quads: forward L1 / L2, backward W2^T / W3^T per net,
//     the joint output layer), copied in by the block from a table the step builds once;
//   * a block is four actor + four critic wavefronts working pairwise on the same tiles: 114 gradient accumulators per
//     wavefront stay in registers across all its tiles and leave through a batched LDS reduction as one [G] partial per block.
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_bf16.h"
#include "wl_kernel_common.h"
#include "wl_mlp.h"
#include "wl_ppo_internal.h"

namespace {

constexpr int kIn = 14, kHid = 64, kTiles = 4;
constexpr int G = WL_PPO_NUM_PARAMS, kRow = WL_PPO_PARTIAL_STRIDE, kPpoBlocks = WL_PPO_BLOCKS;
// flat parameter order = torch named_parameters() of rl/ppo.py ActorCritic: std, actor.{0,2,4}.{weight,bias}, critic...
constexpr int O_STD = 0, O_AW1 = 2, O_AB1 = O_AW1 + kHid * kIn, O_AW2 = O_AB1 + kHid, O_AB2 = O_AW2 + kHid * kHid,
              O_AW3 = O_AB2 + kHid, O_AB3 = O_AW3 + 2 * kHid, O_CW1 = O_AB3 + 2, O_CB1 = O_CW1 + kHid * kIn, O_CW2 = O_CB1 + kHid,
              O_CB2 = O_CW2 + kHid * kHid, O_CW3 = O_CB2 + kHid, O_CB3 = O_CW3 + kHid;
static_assert(O_CB3 + 1 == G, "parameter layout");
constexpr int S_VLOSS = G, S_SURR = G + 1, S_KL = G + 2;

// LDS operand tables (floats).  Operands are stored in QUADS in the order the MFMA loops consume them -- slot
// ((quad * 64 + lane) * 4 + j) -- so that one ds_read_b128 per lane fetches the A operands of four consecutive MFMAs
// (with one ds_read_b32 per MFMA the kernel waited on LDS latency for every matrix instruction: 210 us per minibatch).
//   F1 quad s        : the four output tiles t = j of layer-1 k-step s
//   F2 slab (jj, t, plane): layer 2 on the bf16 pipe (v_mfma_f32_16x16x32_bf16; every f32 operand two bf16 planes, x = hi + lo,
//                      product = lo.hi + hi.lo + hi.hi -- 3 MFMAs of 16 cycles for a K of 32 against 8 f32 MFMAs of 32 cycles):
//                      slab (jj * 4 + t) * 2 + plane = lane (i, g)'s 8 bf16 of W2[16 t + i][in(e)], in(e) = 16 (2 jj + (e >> 2)) +
//                      4 g + (e & 3) -- exactly the 8 inputs lane group g already HOLDS as accumulators h1[2 jj][0..3],
//                      h1[2 jj + 1][0..3], so the B operand is packed from registers with no data movement;
//      slabs 16 + t    : the bias b2 in accumulator layout (f32x4 per lane: units 16 t + 4 g + r), the accumulators' start value
//   B2 slab (jj, t, plane): the same for W2^T: W2[out(e)][16 t + i], out(e) = 16 (2 jj + (e >> 2)) + 4 g + (e & 3)
//   B3 quad 0        : the four unit tiles t = j of W3^T
//   F3 quad q        : k-steps 4 q + j of the joint output layer (0..15 actor units, 16..31 critic units, 32 bias, pad)
constexpr int T_F1 = 0, T_F2 = T_F1 + 4 * 256, T_B2 = T_F2 + 20 * 256, T_B3 = T_B2 + 16 * 256, kNetTab = T_B3 + 256;
constexpr int T_F3 = 2 * kNetTab, kTabFloats = T_F3 + 9 * 256;
static_assert(kTabFloats == WL_PPO_OPERAND_FLOATS, "header constant");
constexpr int kTStride = 20, kTBuf = kHid * kTStride;          // transposition buffer [unit][sample], padded rows
constexpr int kPpoWaves = 8;                                   // per block: four actor wavefronts + four critic wavefronts
constexpr int kLdsFloats = kTabFloats + kPpoWaves * kTBuf;     // + one buffer per wavefront
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");
static_assert(kTabFloats >= kRow, "the operand tables are reused as the block's gradient accumulator");

struct PpoNets {
    WlMlp actor, critic;
    const float* std;
};

// value of slot `idx` of the operand tables (run once per block): i = row of the A tile, g = its k index
WL_DEV float operand_value(const PpoNets& N, int idx) {
    const int j = idx & 3, lane = (idx >> 2) & 63, i = lane & 15, g = lane >> 4;
    int q = idx >> 8;
    if (q < 2 * (kNetTab / 256)) {
        const bool is_actor = q < kNetTab / 256;
        const WlMlp& net = is_actor ? N.actor : N.critic;
        if (!is_actor) q -= kNetTab / 256;
        if (q < 4) {                                            // forward layer 1: k-step s = q, tile t = j
            const int f = 4 * q + g, unit = 16 * j + i;
            return f < kIn ? net.w1[unit * kIn + f] : f == kIn ? net.b1[unit] : 0.f;
        }
        q -= 4;
        if (q < 20) {                                           // forward layer 2: bf16 planes + the bias slabs
            if (q >= 16) return net.b2[16 * (q - 16) + 4 * g + j];
            const int jj = q >> 3, t = (q >> 1) & 3, plane = q & 1;
            float w[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * j + h;
                w[h] = net.w2[(16 * t + i) * kHid + 16 * (2 * jj + (e >> 2)) + 4 * g + (e & 3)];
            }
            uint32_t hi, lo;
            split_bf16_pair(w[0], w[1], hi, lo);
            return __uint_as_float(plane == 0 ? hi : lo);
        }
        q -= 20;
        if (q < 16) {                                           // backward W2^T: bf16 planes
            const int jj = q >> 3, t = (q >> 1) & 3, plane = q & 1;
            float w[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * j + h;
                w[h] = net.w2[(16 * (2 * jj + (e >> 2)) + 4 * g + (e & 3)) * kHid + 16 * t + i];
            }
            uint32_t hi, lo;
            split_bf16_pair(w[0], w[1], hi, lo);
            return __uint_as_float(plane == 0 ? hi : lo);
        }
        const int unit = 16 * j + i;                            // backward W3^T: k = g = output index of the joint layer
        if (is_actor) return g < 2 ? net.w3[g * kHid + unit] : 0.f;
        return g == 2 ? net.w3[unit] : 0.f;
    }
    q -= 2 * (kNetTab / 256);                                   // joint output layer: rows 0 / 4 / 8 = mu_0 / mu_1 / value
    const int ks = 4 * q + j;
    if (ks < 16) return i == 0 ? N.actor.w3[16 * (ks >> 2) + 4 * g + (ks & 3)]
                      : i == 4 ? N.actor.w3[kHid + 16 * (ks >> 2) + 4 * g + (ks & 3)] : 0.f;
    if (ks < 32) return i == 8 ? N.critic.w3[16 * ((ks - 16) >> 2) + 4 * g + ((ks - 16) & 3)] : 0.f;
    if (ks > 32) return 0.f;
    return g != 0 ? 0.f : i == 0 ? N.actor.b3[0] : i == 4 ? N.actor.b3[1] : i == 8 ? N.critic.b3[0] : 0.f;
}
WL_DEV f32x4 quad(const float* tab, int q, int lane) { return *reinterpret_cast<const f32x4*>(tab + (q * 64 + lane) * 4); }

template <int ACT>
WL_DEV float act_grad_from_output(float h) {   // d act / d z expressed through h = act(z)
    if constexpr (ACT == WL_ACT_RELU) return h > 0.f ? 1.f : 0.f;
    else return h > 0.f ? 1.f : h + 1.f;       // ELU: e^z = h + 1 for z <= 0
}

WL_DEV float lane_xor16(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __float_as_int(v)));
}

// accumulator layout -> [unit][sample] in the wavefront's LDS buffer.  Sample 4 s + g' sits at column 4 g' + s, so the four
// k-steps s = 0..3 of an operand (same lane group g' = g) are one 16-byte read.
WL_DEV void put_transposed(float* T, const f32x4 X[kTiles], int g, int n) {
    const int col = 4 * (n & 3) + (n >> 2);
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * t + 4 * g + r) * kTStride + col] = X[t][r];
}
// operands (A or B alike) of k-steps s = 0..3 for unit tile t: elements [unit 16 t + n][sample 4 s + g]
WL_DEV f32x4 get_transposed4(const float* T, int t, int g, int n) {
    return *reinterpret_cast<const f32x4*>(T + (16 * t + n) * kTStride + 4 * g);
}

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// B operand of the bf16 products from two accumulator tiles: this lane's 8 inputs as two planes
WL_DEV void pack_planes(const f32x4& x0, const f32x4& x1, u32x4& hi, u32x4& lo) {
    uint32_t h, l;
    split_bf16_pair(x0[0], x0[1], h, l); hi[0] = h; lo[0] = l;
    split_bf16_pair(x0[2], x0[3], h, l); hi[1] = h; lo[1] = l;
    split_bf16_pair(x1[0], x1[1], h, l); hi[2] = h; lo[2] = l;
    split_bf16_pair(x1[2], x1[3], h, l); hi[3] = h; lo[3] = l;
}
// acc[t] += W . x over 32 inputs (slab group `base`: (jj, t, plane) order), three products per tile
WL_DEV void mma_planes(const float* slabs, int jj, int lane, const u32x4& bh, const u32x4& bl, f32x4 acc[kTiles]) {
    const bf16x8 xh = __builtin_bit_cast(bf16x8, bh), xl = __builtin_bit_cast(bf16x8, bl);
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        const float* at = slabs + ((jj * 4 + t) * 2) * 256 + lane * 4;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(at), al = *reinterpret_cast<const bf16x8*>(at + 256);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc[t], 0, 0, 0);
    }
}

template <int ACT>
WL_DEV void forward_layer2(const float* tab, float one_g0, int lane, const f32x4 h1[kTiles], f32x4 h2[kTiles]) {
    (void)one_g0;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) h2[t] = *reinterpret_cast<const f32x4*>(tab + T_F2 + (16 + t) * 256 + lane * 4);   // bias
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        u32x4 bh, bl;
        pack_planes(h1[2 * jj], h1[2 * jj + 1], bh, bl);
        mma_planes(tab + T_F2, jj, lane, bh, bl, h2);
    }
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[t][r] = mlp_act<ACT>(h2[t][r]);
}

template <int ACT>
WL_DEV void forward_hidden(const float* tab, const float xs[4], float one_g0, int lane, f32x4 h1[kTiles], f32x4 h2[kTiles]) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t) h1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f32x4 a = quad(tab + T_F1, s, lane);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) h1[t] = mfma4(a[t], xs[s], h1[t]);
    }
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[t][r] = mlp_act<ACT>(h1[t][r]);
    forward_layer2<ACT>(tab, one_g0, lane, h1, h2);
}

// gradient accumulators of one net, in MFMA accumulator layout (rows 4 g + r, column n of each 16 x 16 tile)
struct NetGrads {
    f32x4 w2[kTiles][kTiles];   // [out tile][in tile]
    f32x4 w1[kTiles];           // columns = input features (14 = bias, 15 unused); wide form: this lane's partial sums of delta1 (db1)
    f32x4 w3[kTiles];           // rows 0 / 4 (actor) or 8 (critic) of the joint output layer, columns = units of tile t'
    f32x4 b2[kTiles];           // this lane's partial sums over its sample column
    WL_DEV void zero() {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            w1[t] = w3[t] = b2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < kTiles; ++u) w2[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
};

// backward of one net from delta3 (B operand of the joint layer) + weight-gradient accumulation for this tile.  One LDS
// buffer per wavefront: a matrix is written, the operands it feeds are read into registers, then the next one is written.
// The wide form (first layer outside this kernel, see wl_ppo_wide.hip): delta1 leaves as two bf16 planes (x = hi + lo to
// 16 mantissa bits) in [unit][sample] order -- the shared operand of the dW1 = delta1^T . X contraction -- and db1 is summed here.
struct WideOut {
    uint16_t *hi, *lo;   // this net's 64 rows of the tile's 64-sample block ([unit][64] bf16), offset to the tile's 16 samples
};
WL_DEV uint32_t bf16_rne(float x) {   // round to nearest even (finite inputs)
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
WL_DEV void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    hi = bf16_rne(x);
    lo = bf16_rne(x - __uint_as_float(hi << 16));
}

template <int ACT, bool WIDE>
WL_DEV void backward_net(const float* tab, float* T, const f32x4 h1[kTiles], const f32x4 h2[kTiles], float d3,
                         const float xb[4] /* obs rows in B layout: sample 4 s + g, feature n */, int lane, NetGrads& A,
                         const WideOut wo) {
    const int g = lane >> 4, n = lane & 15;
    // dW3 += delta3^T . H2 : A operand = delta3 of row i's output at sample 4 s + g (rows 0 / 4 / 8)
    {
        put_transposed(T, h2, g, n);
        __builtin_amdgcn_wave_barrier();
        f32x4 b4[kTiles];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) b4[t] = get_transposed4(T, t, g, n);
        __builtin_amdgcn_wave_barrier();
        T[g * kTStride + 4 * (n & 3) + (n >> 2)] = d3;          // [output g][sample n], over rows 0..3 (already consumed)
        __builtin_amdgcn_wave_barrier();
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a4 = (n & 3) == 0 && n < 12 ? *reinterpret_cast<const f32x4*>(T + (n >> 2) * kTStride + 4 * g) : zero4;
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) A.w3[t] = mfma4(a4[s], b4[t][s], A.w3[t]);
    }
    // delta2 = (W3^T delta3) * act'(h2)
    f32x4 d2[kTiles], d1[kTiles];
    {
        const f32x4 a = quad(tab + T_B3, 0, lane);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            d2[t] = mfma4(a[t], d3, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[t][r] *= act_grad_from_output<ACT>(h2[t][r]);
        }
    }
    // delta1 = (W2^T delta2) * act'(h1)
#pragma unroll
    for (int t = 0; t < kTiles; ++t) d1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {   // bf16 planes, as layer 2 forward
        u32x4 bh, bl;
        pack_planes(d2[2 * jj], d2[2 * jj + 1], bh, bl);
        mma_planes(tab + T_B2, jj, lane, bh, bl, d1);
    }
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d1[t][r] *= act_grad_from_output<ACT>(h1[t][r]);
            A.b2[t][r] += d2[t][r];
        }
    // dW2 += delta2^T . H1
    {
        __builtin_amdgcn_wave_barrier();
        put_transposed(T, h1, g, n);
        __builtin_amdgcn_wave_barrier();
        f32x4 hb[kTiles];
#pragma unroll
        for (int u = 0; u < kTiles; ++u) hb[u] = get_transposed4(T, u, g, n);
        __builtin_amdgcn_wave_barrier();
        put_transposed(T, d2, g, n);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            const f32x4 a4 = get_transposed4(T, t, g, n);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < kTiles; ++u) A.w2[t][u] = mfma4(a4[s], hb[u][s], A.w2[t][u]);
        }
    }
    // dW1 (+ db1 in column 14) += delta1^T . X
    __builtin_amdgcn_wave_barrier();
    put_transposed(T, d1, g, n);
    __builtin_amdgcn_wave_barrier();
    if constexpr (!WIDE) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            const f32x4 a4 = get_transposed4(T, t, g, n);
#pragma unroll
            for (int s = 0; s < 4; ++s) A.w1[t] = mfma4(a4[s], xb[s], A.w1[t]);
        }
    } else {
        // unit 16 t + n, samples 4 g .. 4 g + 3 (sample s sits at column 4 (s & 3) + (s >> 2)): 8 bytes per plane and lane
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            uint32_t h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_bf16(T[(16 * t + n) * kTStride + 4 * j + g], h[j], l[j]);
            const int at = (16 * t + n) * 64 + 4 * g;
            *reinterpret_cast<uint2*>(wo.hi + at) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            *reinterpret_cast<uint2*>(wo.lo + at) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
#pragma unroll
            for (int r = 0; r < 4; ++r) A.w1[t][r] += d1[t][r];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// sum over the 16 sample-lanes (n = 0..15) of a lane group
WL_DEV float sum_over_n(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// one wavefront adds its accumulators into the block's [G] buffer.  The caller serialises the four wavefronts (barriers
// in between), so these are plain read-modify-writes to addresses no other lane touches: LDS float atomics from four
// wavefronts at once made this epilogue ~40 us of the kernel.
template <bool WIDE>
WL_DEV void flush_net(float* acc, const NetGrads& A, int lane, bool actor) {
    const int g = lane >> 4, n = lane & 15;
    const int o_w1 = actor ? O_AW1 : O_CW1, o_b1 = actor ? O_AB1 : O_CB1, o_w2 = actor ? O_AW2 : O_CW2,
              o_b2 = actor ? O_AB2 : O_CB2, o_w3 = actor ? O_AW3 : O_CW3;
    // reads first, writes after, tile by tile: written as `acc[i] += v` every element is a dependent LDS round trip
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        float o2[4][kTiles], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int uo = 16 * t + 4 * g + r;
#pragma unroll
            for (int u = 0; u < kTiles; ++u) o2[r][u] = acc[o_w2 + uo * kHid + 16 * u + n];
            if constexpr (WIDE) o1[r] = n == 0 ? acc[o_b1 + uo] : 0.f;
            else o1[r] = n < kIn ? acc[o_w1 + uo * kIn + n] : n == kIn ? acc[o_b1 + uo] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int uo = 16 * t + 4 * g + r;
#pragma unroll
            for (int u = 0; u < kTiles; ++u) acc[o_w2 + uo * kHid + 16 * u + n] = o2[r][u] + A.w2[t][u][r];
            if constexpr (WIDE) {   // the w1 slots of the row stay zero: dW1 comes from the contraction kernel
                const float b1 = sum_over_n(A.w1[t][r]);
                if (n == 0) acc[o_b1 + uo] = o1[r] + b1;
            } else {
                if (n < kIn) acc[o_w1 + uo * kIn + n] = o1[r] + A.w1[t][r];
                else if (n == kIn) acc[o_b1 + uo] = o1[r] + A.w1[t][r];
            }
        }
    }
    float b2[kTiles][4], ob[kTiles][4];
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b2[t][r] = sum_over_n(A.b2[t][r]);
            ob[t][r] = acc[o_b2 + 16 * t + 4 * g + r];
        }
    // joint output layer: accumulator row 4 g + r; rows 0 / 4 are the actor's outputs 0 / 1, row 8 the critic's
    const bool mine = actor ? g < 2 : g == 2;
    const int w3_row = actor ? g * kHid : 0;
    float o3[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) o3[t] = mine ? acc[o_w3 + w3_row + 16 * t + n] : 0.f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n == 0) acc[o_b2 + 16 * t + 4 * g + r] = ob[t][r] + b2[t][r];
        if (mine) acc[o_w3 + w3_row + 16 * t + n] = o3[t] + A.w3[t][0];
    }
}

struct PpoHyper {
    float clip, value_loss_coef, inv_batch;
    int use_clipped_value_loss;
};

// the nets in MFMA operand order, built once per minibatch step (each slot is a dependent gather from the weight tensors:
// done by every block of the gradient kernel it cost more than the matrix work)
// (wide agents: the same launch also splits the first-layer weights into the bf16 planes of wl_ppo_wide.hip's contractions
// -- blocks behind the table's; everything that depends on the weights only, one launch in front of the step)
struct WeightPlanes {
    uint32_t *w_hi, *w_lo;   // [128][dp] as pairs, or NULL
    int dp;
};
constexpr int kTabBlocks = (kTabFloats + 255) / 256;
__global__ void __launch_bounds__(256) ppo_operands_kernel(const PpoNets N, float* __restrict__ operands, const WeightPlanes wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.x < kTabBlocks) {
        if (i < kTabFloats) operands[i] = operand_value(N, i);
        return;
    }
    weight_plane_pair(N.actor.w1, N.critic.w1, N.actor.in_dim, wp.dp, i - kTabBlocks * 256, wp.w_hi, wp.w_lo);
}

// Eight wavefronts per block: wavefronts 0..3 differentiate the ACTOR, 4..7 the CRITIC, pairwise on the same tiles (the
// surrogate / KL / std terms need only the actor's outputs, the value loss only the critic's, so the two never talk).
// Each half needs ~250 registers, which puts one actor and one critic wavefront on every SIMD: while one is in its
// activation / transpose (VALU, LDS) phases the other keeps the matrix pipe busy.
// WIDE: the first layer lives outside (wl_ppo_wide.hip): `wio.h1` holds the activated layer-1 outputs of the minibatch,
// [position in the minibatch][actor units 0..63 | critic units 64..127], and delta1 leaves through `wio.dt_*`.
struct WideIo {
    const float* h1;
    uint16_t *dt_hi, *dt_lo;   // bf16 planes, blocked [position / 64][128 units][position % 64]
};
template <int ACT, bool WIDE>
__global__ void __launch_bounds__(64 * kPpoWaves) ppo_grad_kernel(const PpoNets N, const float* __restrict__ operands,
                                                                  const WlPpoBatch bt, const int mb_start, const int mb_size,
                                                                  const PpoHyper hp, float* __restrict__ partials, const WideIo wio) {
    constexpr int kThreads = 64 * kPpoWaves;
    extern __shared__ float lds[];
    float* tab = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, n = lane & 15;
    const bool actor_wave = wave < 4;
    float* T = lds + kTabFloats + wave * kTBuf;
    {   // 87 KB of operands -> LDS, the loads issued in batches (one by one they are dependent round trips)
        constexpr int kVec = kTabFloats / 4;
        const f32x4* src = reinterpret_cast<const f32x4*>(operands);
        f32x4* dst = reinterpret_cast<f32x4*>(tab);
        for (int base = threadIdx.x; base < kVec; base += 4 * kThreads) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = base + k * kThreads < kVec ? src[base + k * kThreads] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (base + k * kThreads < kVec) dst[base + k * kThreads] = v[k];
        }
    }
    __syncthreads();
    const float* net_tab = tab + (actor_wave ? 0 : kNetTab);
    const float one_g0 = g == 0 ? 1.f : 0.f;
    const float sig = N.std[g & 1], sig_other = N.std[(g & 1) ^ 1];
    const float inv_sig = 1.f / sig, log_sig_sum = logf(sig) + logf(sig_other);

    NetGrads GN;
    GN.zero();
    float d_sigma = 0.f, d_b3 = 0.f, s_vloss = 0.f, s_surr = 0.f, s_kl = 0.f;

    const int n_tiles = (mb_size + 15) >> 4;
    const int n_pairs = gridDim.x * 4;
    for (int tile = blockIdx.x * 4 + (wave & 3); tile < n_tiles; tile += n_pairs) {
        const int k_n = tile * 16 + n;                       // position in the minibatch of "my" sample (column n)
        const bool valid = k_n < mb_size;
        const int smp = bt.perm[mb_start + (valid ? k_n : 0)];
        f32x4 h1[kTiles], h2[kTiles];
        float xb[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (WIDE) {   // units 16 t + 4 g .. + 3 of sample n: the accumulator layout, 16 bytes per tile
            const float* row = wio.h1 + (int64_t)(valid ? k_n : 0) * (2 * kHid) + (actor_wave ? 0 : kHid) + 4 * g;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) h1[t] = *reinterpret_cast<const f32x4*>(row + 16 * t);
            forward_layer2<ACT>(net_tab, one_g0, lane, h1, h2);
        } else {
            // observation of sample n in forward-B layout (feature 4 s + g) ...
            float xs[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int f = 4 * s + g;
                xs[s] = f == kIn ? 1.f : f < kIn ? bt.obs[(int64_t)smp * kIn + f] : 0.f;
            }
            // ... and of sample 4 s + g in weight-gradient-B layout (feature n)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k_s = tile * 16 + 4 * s + g;
                const int smp_s = bt.perm[mb_start + (k_s < mb_size ? k_s : 0)];
                xb[s] = (k_s >= mb_size) ? 0.f : n == kIn ? 1.f : n < kIn ? bt.obs[(int64_t)smp_s * kIn + n] : 0.f;
            }
            forward_hidden<ACT>(net_tab, xs, one_g0, lane, h1, h2);
        }
        // this net's rows of the joint output layer: k-steps 0..15 (actor units) or 16..31 (critic units), then the biases
        f32x4 out = {0.f, 0.f, 0.f, 0.f}, out_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = quad(tab + T_F3, actor_wave ? q : q + 4, lane);
            out = mfma4(a[0], h2[q][0], out);
            out_b = mfma4(a[1], h2[q][1], out_b);
            out = mfma4(a[2], h2[q][2], out);
            out_b = mfma4(a[3], h2[q][3], out_b);
        }
        out = mfma4(quad(tab + T_F3, 8, lane)[0], one_g0, out) + out_b;
        const float y = out[0];                               // g = 0: mu_0, 1: mu_1 (actor waves); g = 2: value (critic waves)

        float d3 = 0.f;
        if (actor_wave) {   // ---- clipped surrogate, KL statistic, std gradient ------------------------------------------
            const float adv = bt.adv[smp], logp_old = bt.logp_old[smp];
            const float act_g = bt.actions[smp * 2 + (g & 1)], mu_old_g = bt.mu_old[smp * 2 + (g & 1)];
            const float z = (act_g - y) * inv_sig;            // meaningful on g < 2
            const float zz_other = lane_xor16(z * z, lane);
            const float logp = -0.5f * (z * z + zz_other) - log_sig_sum - 1.8378770664093453f;
            const float ratio = __expf(logp - logp_old);
            const float s1 = -adv * ratio, s2 = -adv * fminf(fmaxf(ratio, 1.f - hp.clip), 1.f + hp.clip);
            const float dl_dlogp = (s1 >= s2 ? -adv : 0.f) * ratio * hp.inv_batch;
            // KL(old || new) per action dim: log(sigma / sigma_old + 1e-5) + (sigma_old^2 + (mu_old - mu)^2) / (2 sigma^2) - 1/2
            const float so = bt.sigma_old[g & 1];
            const float kl_g = logf(sig / so + 1e-5f) + (so * so + (mu_old_g - y) * (mu_old_g - y)) * (0.5f * inv_sig * inv_sig) - 0.5f;
            const float kl_o = lane_xor16(kl_g, lane);
            if (valid && g < 2) {
                d3 = dl_dlogp * z * inv_sig;
                d_sigma += dl_dlogp * (z * z - 1.f) * inv_sig;
                d_b3 += d3;
                if (g == 0) {
                    s_surr += fmaxf(s1, s2);
                    s_kl += kl_g + kl_o;
                }
            }
        } else {            // ---- (clipped) value loss on lane group 2 -----------------------------------------------------
            const float ret = bt.returns[smp], v_old = bt.values_old[smp];
            const float e1 = y - ret, dvo = y - v_old;
            const float e2 = v_old + fminf(fmaxf(dvo, -hp.clip), hp.clip) - ret;
            const float l1 = e1 * e1, l2 = e2 * e2;
            float vloss = l1, dvl = 2.f * e1;
            if (hp.use_clipped_value_loss && l2 > l1) {
                vloss = l2;
                dvl = fabsf(dvo) <= hp.clip ? 2.f * e2 : 0.f;
            }
            if (valid && g == 2) {
                d3 = hp.value_loss_coef * dvl * hp.inv_batch;
                d_b3 += d3;
                s_vloss += vloss;
            }
        }
        WideOut wo{nullptr, nullptr};
        if constexpr (WIDE) {
            const int64_t at = ((int64_t)(tile >> 2) * (2 * kHid) + (actor_wave ? 0 : kHid)) * 64 + (tile & 3) * 16;
            wo = WideOut{wio.dt_hi + at, wio.dt_lo + at};
        }
        backward_net<ACT, WIDE>(net_tab, T, h1, h2, d3, xb, lane, GN, wo);
    }

    // ---- block reduction: the operand tables become the [G + stats] accumulator ------------------------------------------
    __syncthreads();
    float* acc = lds;
    for (int i = threadIdx.x; i < kRow; i += kThreads) acc[i] = 0.f;
    __syncthreads();
    d_sigma = sum_over_n(d_sigma);
    d_b3 = sum_over_n(d_b3);
    s_vloss = sum_over_n(s_vloss);
    s_surr = sum_over_n(s_surr);
    s_kl = sum_over_n(s_kl);
    for (int w = 0; w < 4; ++w) {      // one actor and one critic wavefront at a time (their parameter ranges are disjoint)
        if ((wave & 3) == w) {
            flush_net<WIDE>(acc, GN, lane, actor_wave);
            if (n == 0) {
                if (actor_wave && g < 2) {
                    acc[O_STD + g] += d_sigma;
                    acc[O_AB3 + g] += d_b3;
                    if (g == 0) {
                        acc[S_SURR] += s_surr;
                        acc[S_KL] += s_kl;
                    }
                } else if (!actor_wave && g == 2) {
                    acc[O_CB3] += d_b3;
                    acc[S_VLOSS] += s_vloss;
                }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < kRow; i += kThreads) partials[(int64_t)blockIdx.x * kRow + i] = acc[i];
}

// grad[i] = sum over blocks of partials[b][i]; norm2 += sum of squares of the parameter gradients (stats excluded).
// 64 columns per block, four row groups of threads: 64 independent coalesced loads per thread instead of one thread
// walking all 256 rows (61 us -> a few).
// Also snapshots the action std into ctrl: the apply kernel updates std in place (threads 0 / 1 of its block 0) while every
// one of its blocks needs the OLD std for the entropy gradient and the clip coefficient -- reading it from `std` there is a
// cross-block race (a late block sees the new value: nondeterministic clip coefficient, ranks drifting apart bit by bit).
__global__ void __launch_bounds__(256) ppo_reduce_kernel(const float* __restrict__ partials, int n_blocks, float* __restrict__ grad,
                                                         float* __restrict__ norm2, const float* __restrict__ std,
                                                         float* __restrict__ std_snapshot) {
    if (blockIdx.x == 0 && threadIdx.x < 2) std_snapshot[threadIdx.x] = std[threadIdx.x];
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < kRow) {
        int b = rg;
#pragma unroll 4   // 16 loads in flight per thread (4: every round of four was a dependent round trip, 12 us for 256 rows)
        for (; b + 12 < n_blocks; b += 16) {
            s0 += partials[(int64_t)b * kRow + c];
            s1 += partials[(int64_t)(b + 4) * kRow + c];
            s2 += partials[(int64_t)(b + 8) * kRow + c];
            s3 += partials[(int64_t)(b + 12) * kRow + c];
        }
        for (; b < n_blocks; b += 4) s0 += partials[(int64_t)b * kRow + c];
    }
    part[rg][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0) {
        const float s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (c < kRow) grad[c] = s;
        float q = c < G ? s * s : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
        if (threadIdx.x == 0) atomicAdd(norm2, q);
    }
}

// flat parameter order for an input width `in` (the constants above are in = 14)
struct PpoLayout {
    int in, o_aw1, o_cw1, G;
};
inline PpoLayout ppo_layout(int in) {
    const int per_net = kHid * in + kHid + kHid * kHid + kHid;
    return PpoLayout{in, 2, 2 + per_net + 2 * kHid + 2, 2 + 2 * per_net + 3 * kHid + 3};
}
WL_DEV float* param_ptr(const PpoNets& N, float* std, int i, const PpoLayout& L) {
    if (i < L.o_aw1) return std + i;
    const bool a = i < L.o_cw1;
    const WlMlp& net = a ? N.actor : N.critic;
    const int j = i - (a ? L.o_aw1 : L.o_cw1);
    const int nw3 = a ? 2 * kHid : kHid;
    if (j < kHid * L.in) return const_cast<float*>(net.w1) + j;
    if (j < kHid * L.in + kHid) return const_cast<float*>(net.b1) + (j - kHid * L.in);
    const int k = j - kHid * L.in - kHid;
    if (k < kHid * kHid) return const_cast<float*>(net.w2) + k;
    if (k < kHid * kHid + kHid) return const_cast<float*>(net.b2) + (k - kHid * kHid);
    const int m = k - kHid * kHid - kHid;
    if (m < nw3) return const_cast<float*>(net.w3) + m;
    return const_cast<float*>(net.b3) + (m - nw3);
}

// entropy term, gradient clipping, the adaptive-KL learning-rate rule and Adam (torch.optim.Adam's arithmetic)
__global__ void __launch_bounds__(256) ppo_apply_kernel(const PpoNets N, float* std, const WlPpoParams hp, const float inv_batch,
                                                        const float* __restrict__ grad, float* __restrict__ adam_m,
                                                        float* __restrict__ adam_v, float* __restrict__ ctrl, const int parity,
                                                        const int step /* 1-based */, const PpoLayout L) {
    const int G = L.G, S_VLOSS = L.G, S_SURR = L.G + 1, S_KL = L.G + 2;   // the statistics follow the parameters
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float lr_old = ctrl[WL_PPO_CTRL_LR + parity];
    float lr = lr_old;
    if (hp.adaptive) {
        const float kl = grad[S_KL] * inv_batch;
        if (kl > hp.desired_kl * 2.f) lr = fmaxf(hp.lr_min, lr_old / 1.5f);
        else if (kl > 0.f && kl < hp.desired_kl * 0.5f) lr = fminf(hp.lr_max, lr_old * 1.5f);
    }
    // the entropy bonus -c_e * sum_j log sigma_j enters the std gradient before the norm; its square is added on the fly
    float n2 = ctrl[WL_PPO_CTRL_NORM2 + parity];
    const float gs0 = grad[O_STD], gs1 = grad[O_STD + 1];
    const float es0 = -hp.entropy_coef / ctrl[WL_PPO_CTRL_STD], es1 = -hp.entropy_coef / ctrl[WL_PPO_CTRL_STD + 1];   // snapshot, see ppo_reduce_kernel
    n2 += (gs0 + es0) * (gs0 + es0) - gs0 * gs0 + (gs1 + es1) * (gs1 + es1) - gs1 * gs1;
    const float coef = fminf(1.f, hp.max_grad_norm / (sqrtf(fmaxf(n2, 0.f)) + 1e-6f));
    if (i < G) {
        float gr = grad[i];
        if (i == O_STD) gr += es0;
        if (i == O_STD + 1) gr += es1;
        gr *= coef;
        const float m = hp.beta1 * adam_m[i] + (1.f - hp.beta1) * gr;
        const float v = hp.beta2 * adam_v[i] + (1.f - hp.beta2) * gr * gr;
        adam_m[i] = m;
        adam_v[i] = v;
        const float bc1 = 1.f - powf(hp.beta1, (float)step), bc2 = 1.f - powf(hp.beta2, (float)step);
        const float denom = sqrtf(v) / sqrtf(bc2) + hp.eps;
        float* p = param_ptr(N, std, i, L);
        *p -= (lr / bc1) * (m / denom);
    }
    if (i == 0) {   // hand the learning rate to the next call, clear its norm accumulator, book the statistics
        ctrl[WL_PPO_CTRL_LR + (parity ^ 1)] = lr;
        ctrl[WL_PPO_CTRL_NORM2 + (parity ^ 1)] = 0.f;
        ctrl[WL_PPO_CTRL_STATS + 0] += grad[S_VLOSS] * inv_batch;
        ctrl[WL_PPO_CTRL_STATS + 1] += grad[S_SURR] * inv_batch;
        ctrl[WL_PPO_CTRL_STATS + 2] += grad[S_KL] * inv_batch;
    }
}

// rsl_rl RolloutStorage.compute_returns: one lane per env walks its K transitions backwards ([step][env] rows: coalesced)
__global__ void __launch_bounds__(256) gae_kernel(const int K, const int n, const float* __restrict__ rewards,
                                                   const float* __restrict__ values, const int64_t* __restrict__ dones,
                                                   const float gamma, const float lam, float* __restrict__ returns,
                                                   float* __restrict__ advantages) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float adv = 0.f, v_next = values[(int64_t)K * n + e];
    constexpr int kB = 8;   // eight transitions' rows requested together (one by one each was a memory round trip: the stores may alias)
    for (int t0 = K - 1; t0 >= 0; t0 -= kB) {
        float r[kB], vv[kB];
        int64_t d[kB];
#pragma unroll
        for (int i = 0; i < kB; ++i) {
            const int64_t at = (int64_t)max(t0 - i, 0) * n + e;
            r[i] = rewards[at], vv[i] = values[at], d[i] = dones[at];
        }
#pragma unroll
        for (int i = 0; i < kB; ++i) {
            if (t0 - i < 0) break;
            const int64_t at = (int64_t)(t0 - i) * n + e;
            const float nd = d[i] != 0 ? 0.f : 1.f, v = vv[i];
            const float delta = r[i] + nd * gamma * v_next - v;
            adv = delta + nd * gamma * lam * adv;
            advantages[at] = adv;
            returns[at] = adv + v;
            v_next = v;
        }
    }
}

// The runner's per-rollout bookkeeping (modified_rsl_rl_runner.py:74-75, 88-98 does it per step, with a host sync each) in one
// pass, one lane per env walking its K transitions forwards: the return / length of every episode that ends inside the rollout
// (written at its last transition; the carries of episodes still running are updated in place), the sum of the RAW rewards,
// the number of non-finite action components, and then rsl_rl's time-out bootstrap rewards += gamma V(obs_t) time_outs in place.
// Per-block partial sums (no float atomics: the logged mean must not depend on the block order).
__global__ void __launch_bounds__(256) rollout_bookkeeping_kernel(const int K, const int n, float* __restrict__ rewards,
                                                                  const float* __restrict__ values, const int64_t* __restrict__ dones,
                                                                  const uint8_t* __restrict__ time_outs, const float2* __restrict__ actions,
                                                                  const float gamma, float* __restrict__ carry_ret,
                                                                  float* __restrict__ carry_len, float* __restrict__ ep_ret,
                                                                  float* __restrict__ ep_len, float* __restrict__ stats) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    float rsum = 0.f, bad = 0.f, n_done = 0.f;
    if (e < n) {
        float ret = carry_ret[e], len = carry_len[e];
        // eight transitions' rows requested together: the stores of the loop body may alias the loads for all the compiler knows,
        // so one by one every transition paid a memory round trip (0.12 ms for 128 steps)
        constexpr int kB = 8;
        for (int t0 = 0; t0 < K; t0 += kB) {
            float r[kB], v[kB];
            float2 a[kB];
            int64_t d[kB];
            uint8_t to[kB];
#pragma unroll
            for (int i = 0; i < kB; ++i) {
                const int64_t at = (int64_t)min(t0 + i, K - 1) * n + e;
                r[i] = rewards[at], v[i] = values[at], a[i] = actions[at], d[i] = dones[at], to[i] = time_outs[at];
            }
#pragma unroll
            for (int i = 0; i < kB; ++i) {
                if (t0 + i >= K) break;
                const int64_t at = (int64_t)(t0 + i) * n + e;
                rsum += r[i];
                bad += (__builtin_isfinite(a[i].x) ? 0.f : 1.f) + (__builtin_isfinite(a[i].y) ? 0.f : 1.f);
                ret += r[i];
                len += 1.f;
                if (d[i] != 0) {
                    ep_ret[at] = ret;
                    ep_len[at] = len;
                    ret = len = 0.f;
                    n_done += 1.f;
                }
                if (to[i]) rewards[at] = fmaf(gamma, v[i], r[i]);
            }
        }
        carry_ret[e] = ret;
        carry_len[e] = len;
    }
    __shared__ float part[3][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        rsum += __shfl_down(rsum, off, 64);
        bad += __shfl_down(bad, off, 64);
        n_done += __shfl_down(n_done, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = rsum;
        part[1][threadIdx.x >> 6] = bad;
        part[2][threadIdx.x >> 6] = n_done;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float* q = part[threadIdx.x];
        stats[threadIdx.x * gridDim.x + blockIdx.x] = (q[0] + q[1]) + (q[2] + q[3]);
    }
}

int check_ppo(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* bt, int mb_start, int mb_size,
              const WlPpoState* st) {
    int rc = check_mlp(actor);
    if (rc != WL_OK) return rc;
    rc = check_mlp(critic);
    if (rc != WL_OK) return rc;
    if (actor->in_dim != kIn || critic->in_dim != kIn || actor->out_dim != 2 || critic->out_dim != 1 ||
        actor->activation != critic->activation)
        return WL_EINVAL;
    if (!std || !bt || !bt->obs || !bt->actions || !bt->mu_old || !bt->logp_old || !bt->adv || !bt->returns || !bt->values_old ||
        !bt->perm || !bt->sigma_old || mb_start < 0 || mb_size <= 0)
        return WL_EINVAL;
    if (!st || !st->partials || !st->grad || !st->ctrl || !st->operands || ((uintptr_t)st->operands & 15u)) return WL_EINVAL;
    return WL_OK;
}

// operands -> gradient kernel -> reduction of the per-block rows into `grad` ([kRow], narrow layout) + squared norm + std snapshot.
// wio == nullptr: the drift agents' form (first layer in-kernel); else the wide form (wl_ppo_wide.hip): the operand tables
// are there already (ppo_prepare_wide) and the reduction is the caller's (ppo_wide_scatter_kernel); returns the number of
// per-block rows written instead of WL_OK.
int launch_grad(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* bt, int mb_start, int mb_size,
                const WlPpoParams* hp, float* partials, float* operands, float* grad, float* norm2, float* std_snapshot,
                const WideIo* wio, hipStream_t stream) {
    const PpoNets N{*actor, *critic, std};
    const PpoHyper h{hp->clip, hp->value_loss_coef, 1.f / (float)mb_size, hp->use_clipped_value_loss};
    const int n_tiles = (mb_size + 15) / 16;
    const int blocks = min(kPpoBlocks, (n_tiles + 3) / 4);
    const size_t lds_bytes = (size_t)kLdsFloats * 4;
    static bool attr_set = false;
    if (!attr_set) {
        for (const void* f : {(const void*)ppo_grad_kernel<WL_ACT_ELU, false>, (const void*)ppo_grad_kernel<WL_ACT_RELU, false>,
                              (const void*)ppo_grad_kernel<WL_ACT_ELU, true>, (const void*)ppo_grad_kernel<WL_ACT_RELU, true>})
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    clear_error();
    if (!wio) ppo_operands_kernel<<<kTabBlocks, 256, 0, stream>>>(N, operands, WeightPlanes{nullptr, nullptr, 0});
    const bool elu = actor->activation == WL_ACT_ELU;
    const WideIo none{nullptr, nullptr, nullptr};
    if (!wio) {
        if (elu) ppo_grad_kernel<WL_ACT_ELU, false><<<blocks, 64 * kPpoWaves, lds_bytes, stream>>>(N, operands, *bt, mb_start, mb_size, h, partials, none);
        else ppo_grad_kernel<WL_ACT_RELU, false><<<blocks, 64 * kPpoWaves, lds_bytes, stream>>>(N, operands, *bt, mb_start, mb_size, h, partials, none);
    } else {
        if (elu) ppo_grad_kernel<WL_ACT_ELU, true><<<blocks, 64 * kPpoWaves, lds_bytes, stream>>>(N, operands, *bt, mb_start, mb_size, h, partials, *wio);
        else ppo_grad_kernel<WL_ACT_RELU, true><<<blocks, 64 * kPpoWaves, lds_bytes, stream>>>(N, operands, *bt, mb_start, mb_size, h, partials, *wio);
    }
    if (wio) return launch_status() == WL_OK ? blocks : WL_ELAUNCH;
    ppo_reduce_kernel<<<(kRow + 63) / 64, 256, 0, stream>>>(partials, blocks, grad, norm2, std, std_snapshot);
    return launch_status();
}
int launch_grad(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* bt, int mb_start, int mb_size,
                const WlPpoParams* hp, const WlPpoState* st, int parity, hipStream_t stream) {
    return launch_grad(actor, critic, std, bt, mb_start, mb_size, hp, st->partials, st->operands, st->grad,
                       st->ctrl + WL_PPO_CTRL_NORM2 + parity, st->ctrl + WL_PPO_CTRL_STD, nullptr, stream);
}

int launch_apply(const WlMlp* actor, const WlMlp* critic, float* std, int in_dim, int mb_size, const WlPpoParams* hp,
                 const float* grad, float* adam_m, float* adam_v, float* ctrl, int parity, int adam_step, hipStream_t stream) {
    const PpoNets N{*actor, *critic, std};
    const PpoLayout L = ppo_layout(in_dim);
    ppo_apply_kernel<<<(L.G + 255) / 256, 256, 0, stream>>>(N, std, *hp, 1.f / (float)mb_size, grad, adam_m, adam_v, ctrl, parity,
                                                             adam_step, L);
    return launch_status();
}

}  // namespace

namespace wl_internal {   // wl_ppo_internal.h: what wl_ppo_wide.hip drives
int ppo_prepare_wide(const WlMlp* actor, const WlMlp* critic, const float* std, float* operands, int dp, uint16_t* w_hi, uint16_t* w_lo,
                     hipStream_t stream) {
    const PpoNets N{*actor, *critic, std};
    clear_error();
    ppo_operands_kernel<<<kTabBlocks + (128 * dp / 2 + 255) / 256, 256, 0, stream>>>(N, operands,
                                                                                         WeightPlanes{(uint32_t*)w_hi, (uint32_t*)w_lo, dp});
    return launch_status();
}
int ppo_tail_wide(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* bt, int mb_start, int mb_size,
                  const WlPpoParams* hp, float* partials, float* operands, const float* h1, uint16_t* dt_hi, uint16_t* dt_lo,
                  hipStream_t stream) {
    const WideIo wio{h1, dt_hi, dt_lo};
    return launch_grad(actor, critic, std, bt, mb_start, mb_size, hp, partials, operands, nullptr, nullptr, nullptr, &wio, stream);
}
int ppo_apply_any(const WlMlp* actor, const WlMlp* critic, float* std, int in_dim, int mb_size, const WlPpoParams* hp,
                  const float* grad, float* adam_m, float* adam_v, float* ctrl, int parity, int adam_step, hipStream_t stream) {
    clear_error();
    return launch_apply(actor, critic, std, in_dim, mb_size, hp, grad, adam_m, adam_v, ctrl, parity, adam_step, stream);
}
}  // namespace wl_internal

extern "C" {

int wl_gae(int32_t n_steps, int32_t n_envs, const float* rewards, const float* values, const int64_t* dones, float gamma,
           float lam, float* returns, float* advantages, void* stream) {
    if (n_steps <= 0 || n_envs <= 0 || !rewards || !values || !dones || !returns || !advantages) return WL_EINVAL;
    clear_error();
    gae_kernel<<<(n_envs + 255) / 256, 256, 0, (hipStream_t)stream>>>(n_steps, n_envs, rewards, values, dones, gamma, lam, returns,
                                                                      advantages);
    return launch_status();
}

int wl_rollout_bookkeeping(int32_t n_steps, int32_t n_envs, float* rewards, const float* values, const int64_t* dones,
                           const uint8_t* time_outs, const float* actions, float gamma, float* carry_ret, float* carry_len,
                           float* ep_ret, float* ep_len, float* stats, void* stream) {
    if (n_steps <= 0 || n_envs <= 0 || !rewards || !values || !dones || !time_outs || !actions || !carry_ret || !carry_len || !ep_ret ||
        !ep_len || !stats)
        return WL_EINVAL;
    if ((uintptr_t)actions & 7u) return WL_EALIGN;
    clear_error();
    rollout_bookkeeping_kernel<<<(n_envs + 255) / 256, 256, 0, (hipStream_t)stream>>>(n_steps, n_envs, rewards, values, dones, time_outs,
                                                                                     (const float2*)actions, gamma, carry_ret, carry_len,
                                                                                     ep_ret, ep_len, stats);
    return launch_status();
}

int wl_ppo_gradients(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* batch, int32_t mb_start,
                     int32_t mb_size, const WlPpoParams* hp, const WlPpoState* state, int32_t parity, void* stream) {
    int rc = check_ppo(actor, critic, std, batch, mb_start, mb_size, state);
    if (rc != WL_OK) return rc;
    if (!hp || (parity != 0 && parity != 1)) return WL_EINVAL;
    return launch_grad(actor, critic, std, batch, mb_start, mb_size, hp, state, parity, (hipStream_t)stream);
}

int wl_ppo_minibatch(const WlMlp* actor, const WlMlp* critic, float* std, const WlPpoBatch* batch, int32_t mb_start,
                     int32_t mb_size, const WlPpoParams* hp, const WlPpoState* state, int32_t parity, int32_t adam_step,
                     void* stream) {
    int rc = check_ppo(actor, critic, std, batch, mb_start, mb_size, state);
    if (rc != WL_OK) return rc;
    if (!hp || !state->adam_m || !state->adam_v || (parity != 0 && parity != 1) || adam_step < 1) return WL_EINVAL;
    rc = launch_grad(actor, critic, std, batch, mb_start, mb_size, hp, state, parity, (hipStream_t)stream);
    if (rc != WL_OK) return rc;
    return launch_apply(actor, critic, std, kIn, mb_size, hp, state->grad, state->adam_m, state->adam_v, state->ctrl, parity, adam_step,
                        (hipStream_t)stream);
}

int wl_ppo_apply(const WlMlp* actor, const WlMlp* critic, float* std, int32_t mb_size, const WlPpoParams* hp,
                 const WlPpoState* state, int32_t parity, int32_t adam_step, void* stream) {
    int rc = check_mlp(actor);
    if (rc == WL_OK) rc = check_mlp(critic);
    if (rc != WL_OK) return rc;
    if (actor->in_dim != kIn || critic->in_dim != kIn || actor->out_dim != 2 || critic->out_dim != 1) return WL_EINVAL;
    if (!std || !hp || !state || !state->grad || !state->adam_m || !state->adam_v || !state->ctrl || mb_size <= 0 ||
        (parity != 0 && parity != 1) || adam_step < 1)
        return WL_EINVAL;
    clear_error();
    return launch_apply(actor, critic, std, kIn, mb_size, hp, state->grad, state->adam_m, state->adam_v, state->ctrl, parity, adam_step,
                        (hipStream_t)stream);
}

}  // extern "C"

// wl_actor.hip -- one policy step for WIDE observations (elevation: 689 features, visual: 3208): actor mean -> Gaussian
// sample -> log-prob, and the critic's value, in ONE launch (rsl_rl ActorCritic.act + evaluate,
// modified_rsl_rl_runner.py:70-76 with the agents of elevation/config/agents/mushr/rsl_rl_ppo_cfg.py: [64, 64] MLPs).
//
// The first layer is a skinny GEMM H1^T[64 x rows] = W1[64 x D] * X^T[D x rows] on v_mfma_f32_16x16x4_f32 (exact fp32).
// A wavefront owns ONE net (actor or critic) and ONE 16-row tile, i.e. four 16-unit accumulators; the contraction index
// is walked in chunks of 16 features with the k-step <-> feature mapping f = k0 + 4 g + s (lane group g, k-step s), so
// that every lane fetches its A operands (weights) and its B operand (observation) as ONE 16-byte load per chunk and
// tile -- a dot product does not care about the order of its terms.  A wavefront keeps RT row tiles on the same weight
// operands; KS wavefronts split the feature range of one net and one row block between them (a 4096-env batch is only
// 128 row blocks of 32: 2 nets x KS = 4 puts a wavefront on every SIMD) and fold their partial accumulators through LDS;
// wavefront 0 then finishes the 64-64-out tail with the register-resident layout of wl_mlp.h (the layer-1 accumulator IS
// the B operand of layer 2).
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_kernel_common.h"
#include "wl_actor_dev.h"
#include "wl_bf16.h"
#include "wl_mlp.h"
#include "wl_ppo_internal.h"
#include "wl_rng.h"

namespace {

using u32x4_t = __attribute__((ext_vector_type(4))) uint32_t;
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kRing = 4;   // chunks of operands in flight per wavefront (deeper rings measured no different: the
                           // kernel is bound by L2 -> L1 operand traffic, not by its latency)

// operands of one 16-feature chunk: A (weights) for the four unit tiles, B (observations) for the RT row tiles
template <int RT>
struct Chunk {
    wl_f4u a[kMlpTiles];
    wl_f4u b[RT];
};

// 16-byte operand loads of chunk k0 (all 16 features inside [0, D))
template <int RT>
WL_DEV void load_chunk(const float* __restrict__ w_lane, const float* const (&x_lane)[RT], int D, int k0, Chunk<RT>& c) {
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) c.a[t] = *reinterpret_cast<const wl_f4u*>(w_lane + (int64_t)16 * t * D + k0);
#pragma unroll
    for (int q = 0; q < RT; ++q) c.b[q] = *reinterpret_cast<const wl_f4u*>(x_lane[q] + k0);
}
// the last, partial chunk: element-wise with the features past D read as zero (never touches memory past a row's end)
template <int RT>
WL_DEV void load_chunk_tail(const float* __restrict__ w_lane, const float* const (&x_lane)[RT], int D, int k0, int g, Chunk<RT>& c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bool in = k0 + 4 * g + s < D;
#pragma unroll
        for (int t = 0; t < kMlpTiles; ++t) c.a[t][s] = in ? w_lane[(int64_t)16 * t * D + k0 + s] : 0.f;
#pragma unroll
        for (int q = 0; q < RT; ++q) c.b[q][s] = in ? x_lane[q][k0 + s] : 0.f;
    }
}
template <int RT>
WL_DEV void mma_chunk(const Chunk<RT>& c, f32x4 (&h)[RT][kMlpTiles]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < RT; ++q)
#pragma unroll
            for (int t = 0; t < kMlpTiles; ++t) h[q][t] = mfma4(c.a[t][s], c.b[q][s], h[q][t]);
}

// block = KS wavefronts on RT 16-row tiles of ONE net (blockIdx.y: 0 actor, 1 critic).  Every weight operand a wavefront
// fetches feeds RT row tiles: at RT = 1 the launch was bound by L2 -> L1 operand traffic (each of the 256 tiles of a
// 4096-row batch streamed the whole first-layer matrix), not by the matrix pipe.
template <int ACT, int KS, int RT>
__global__ void __launch_bounds__(64 * KS) actor_critic_act_kernel(const WlMlp actor, const WlMlp critic,
                                                                   const float* __restrict__ std, const int n_rows,
                                                                   const float* __restrict__ obs, const int64_t obs_stride,
                                                                   float* __restrict__ actions, float* __restrict__ mu_out,
                                                                   float* __restrict__ log_prob, float* __restrict__ values,
                                                                   const int env_offset, const uint64_t seed, const uint64_t step,
                                                                   const int deterministic, const int first_net) {
    __shared__ float part[KS > 1 ? KS - 1 : 1][RT * kMlpTiles * 4][64];
    const int lane = threadIdx.x & 63, kpart = threadIdx.x >> 6;   // this wavefront's share of the features
    const int which = (int)blockIdx.y + first_net;   // 0 actor, 1 critic (grid.y = 1: only `first_net`)
    const WlMlp& net = which == 0 ? actor : critic;
    const int D = net.in_dim;
    const int m = lane & 15, g = lane >> 4;
    const int row0 = (int)blockIdx.x * (16 * RT);
    // this lane's operand streams: weights of unit m (+ 16 t) and the observations of rows m (+ 16 q), at feature offset 4 g
    const float* w_lane = net.w1 + (int64_t)m * D + 4 * g;
    const float* x_lane[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q)   // spare lanes of the last tile(s) mirror the last row
        x_lane[q] = obs + (int64_t)min(row0 + 16 * q + m, n_rows - 1) * obs_stride + 4 * g;

    f32x4 h[RT][kMlpTiles];
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bias = kpart == 0 ? net.b1[16 * t + 4 * g + r] : 0.f;   // the bias seeds the accumulators
#pragma unroll
            for (int q = 0; q < RT; ++q) h[q][t][r] = bias;
        }

    // chunks [c0, c1) of 16 features for this wavefront; the partial last chunk (if any) belongs to the last share
    const int n_full = D >> 4, per = (n_full + KS - 1) / KS;
    const int c0 = min(kpart * per, n_full), c1 = min(c0 + per, n_full);
    Chunk<RT> ring[kRing];
#pragma unroll
    for (int j = 0; j < kRing; ++j)
        if (c0 + j < c1) load_chunk<RT>(w_lane, x_lane, D, (c0 + j) << 4, ring[j]);
    Chunk<RT> last;   // the partial last chunk, requested up front as well
    const bool has_last = kpart == KS - 1 && (D & 15);
    if (has_last) load_chunk_tail<RT>(w_lane, x_lane, D, n_full << 4, g, last);
    // independent of layer 1 and needed right after it: the tail's weights and the action draws, requested / computed in
    // the shadow of the first operand loads instead of as a dependent round trip at the end
    MlpTail W;
    float z0[RT], z1[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) z0[q] = z1[q] = 0.f;
    if (kpart == 0) {
        load_tail(net, lane, W);
        if (which == 0 && !deterministic) {
#pragma unroll
            for (int q = 0; q < RT; ++q) {
                const F4 u = philox_uniform4((uint32_t)(env_offset + row0 + 16 * q + m), step, WL_RS_POLICY, seed);
                box_muller(u.x, u.y, z0[q], z1[q]);
            }
        }
    }
    for (int c = c0; c < c1; c += kRing) {
#pragma unroll
        for (int j = 0; j < kRing; ++j) {
            if (c + j < c1) {
                mma_chunk<RT>(ring[j], h);
                if (c + j + kRing < c1) load_chunk<RT>(w_lane, x_lane, D, (c + j + kRing) << 4, ring[j]);
            }
        }
    }
    if (has_last) mma_chunk<RT>(last, h);
    if constexpr (KS > 1) {
        if (kpart > 0) {
#pragma unroll
            for (int q = 0; q < RT; ++q)
#pragma unroll
                for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[kpart - 1][(q * kMlpTiles + t) * 4 + r][lane] = h[q][t][r];
        }
        __syncthreads();
        if (kpart > 0) return;
#pragma unroll
        for (int k = 0; k < KS - 1; ++k)
#pragma unroll
            for (int q = 0; q < RT; ++q)
#pragma unroll
                for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[q][t][r] += part[k][(q * kMlpTiles + t) * 4 + r][lane];
    }
    const float std0 = std[0], std1 = std[1];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        const f32x4 out = eval_tail<ACT>(W, h[q], lane);
        const int r_out = row0 + 16 * q + m;
        if (g != 0 || r_out >= n_rows) continue;
        if (which == 1) {
            values[r_out] = out[0];
            continue;
        }
        // a ~ N(mu, diag(std^2)) keyed by (seed, global env, step) on the policy stream: the draw of wl_drift_rollout_policy
        reinterpret_cast<float2*>(actions)[r_out] = make_float2(fmaf(std0, z0[q], out[0]), fmaf(std1, z1[q], out[1]));
        reinterpret_cast<float2*>(mu_out)[r_out] = make_float2(out[0], out[1]);
        log_prob[r_out] = fmaf(-0.5f, fmaf(z0[q], z0[q], z1[q] * z1[q]), -(log_fast(std0) + log_fast(std1)) - kLog2PiA);
    }
}

// The policy step with the first layer on the bf16 matrix pipe (wl_actor_critic_act_planes): layer 1 of both nets is the
// split-K contraction of wl_ppo_wide.hip (observation rows split into bf16 planes in registers, weights as planes through
// LDS -- every weight operand feeds 128 rows instead of 16-64, which is what bound the kernel above); this kernel sums
// the partial products, adds the bias and finishes the net as above.  One wavefront per 16-row tile and net.
template <int ACT>
__global__ void __launch_bounds__(64) act_tail_kernel(const WlMlp actor, const WlMlp critic, const float* __restrict__ std,
                                                      const int n_rows, const float* __restrict__ partials, const int splits,
                                                      float* __restrict__ actions, float* __restrict__ mu_out,
                                                      float* __restrict__ log_prob, float* __restrict__ values, const int env_offset,
                                                      const uint64_t seed, const uint64_t step, const int deterministic,
                                                      const int first_net) {
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    const int which = (int)blockIdx.y + first_net;
    const WlMlp& net = which == 0 ? actor : critic;
    const int r_out = (int)blockIdx.x * 16 + m, row = min(r_out, n_rows - 1);
    // accumulator layout of layer 1: units 16 t + 4 g .. + 3 of row m -- 16 bytes per tile and split
    const float* src = partials + (int64_t)row * (2 * kMlpHidden) + which * kMlpHidden + 4 * g;
    const int64_t plane = (int64_t)n_rows * (2 * kMlpHidden);
    // the tail's weights and the draw do not depend on layer 1: requested / computed in the shadow of the partial sums' loads
    MlpTail W;
    load_tail(net, lane, W);
    float z0 = 0.f, z1 = 0.f;
    if (which == 0 && !deterministic) {
        const F4 u = philox_uniform4((uint32_t)(env_offset + r_out), step, WL_RS_POLICY, seed);
        box_muller(u.x, u.y, z0, z1);
    }
    f32x4 h[kMlpTiles];
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) h[t] = *reinterpret_cast<const f32x4*>(net.b1 + 16 * t + 4 * g);
    // eight splits' loads in flight per round (rolled, every split was a dependent memory round trip: 20 us at 51 splits)
    int s = 0;
    for (; s + 8 <= splits; s += 8) {
        f32x4 v[8][kMlpTiles];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < kMlpTiles; ++t) v[i][t] = *reinterpret_cast<const f32x4*>(src + (s + i) * plane + 16 * t);
#pragma unroll
        for (int t = 0; t < kMlpTiles; ++t)
            h[t] += ((v[0][t] + v[1][t]) + (v[2][t] + v[3][t])) + ((v[4][t] + v[5][t]) + (v[6][t] + v[7][t]));
    }
    {
        f32x4 v[8][kMlpTiles];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < kMlpTiles; ++t)
                v[i][t] = s + i < splits ? *reinterpret_cast<const f32x4*>(src + (s + i) * plane + 16 * t) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < kMlpTiles; ++t)
            h[t] += ((v[0][t] + v[1][t]) + (v[2][t] + v[3][t])) + ((v[4][t] + v[5][t]) + (v[6][t] + v[7][t]));
    }
    const f32x4 out = eval_tail<ACT>(W, h, lane);
    if (g != 0 || r_out >= n_rows) return;
    if (which == 1) {
        values[r_out] = out[0];
        return;
    }
    const float std0 = std[0], std1 = std[1];
    reinterpret_cast<float2*>(actions)[r_out] = make_float2(fmaf(std0, z0, out[0]), fmaf(std1, z1, out[1]));
    reinterpret_cast<float2*>(mu_out)[r_out] = make_float2(out[0], out[1]);
    log_prob[r_out] = fmaf(-0.5f, fmaf(z0, z0, z1 * z1), -(log_fast(std0) + log_fast(std1)) - kLog2PiA);
}

// ---- one launch, layer 1 on the bf16 pipe ------------------------------------------------------------------------------------
// The f32 kernel above is bound by its operand stream from L2: every 16 x RT rows re-read a whole first-layer matrix, and
// more rows per block (fewer re-reads) cost f32 matrix time it does not have.  v_mfma_f32_16x16x32_bf16 with split operands
// (x = hi + lo, three products) is 5x cheaper per feature, which buys RT = 4 (64 rows per block: half the stream of RT = 2)
// with room to spare.  Weights come as the bf16 planes of wl_actor_critic_planes ([128][dp], 64-wide chunks, the last one
// overlapping its predecessor), observation rows are f32 and split in registers; KS wavefronts split the features, fold
// their partial accumulators through LDS in a fixed order, and wavefront q finishes row tile q (layers 2-3, draw, log-prob).
struct PlaneStep {
    u32x4_t ah[kMlpTiles], al[kMlpTiles];   // weights: 8 bf16 per plane and unit tile
};
WL_DEV int plane_feature(int s, int D) { return min(64 * (s >> 1), D - 64) + 32 * (s & 1); }   // first feature of k-step s

template <int ACT, int KS, int RT, int DEPTH = 2>
__global__ void __launch_bounds__(64 * KS) act_bf16_kernel(const WlMlp actor, const WlMlp critic, const float* __restrict__ std,
                                                           const int n_rows, const float* __restrict__ obs, const int64_t obs_stride,
                                                           float* __restrict__ actions, float* __restrict__ mu_out,
                                                           float* __restrict__ log_prob, float* __restrict__ values,
                                                           const int env_offset, const uint64_t seed, const uint64_t step,
                                                           const int deterministic, const int first_net,
                                                           const uint16_t* __restrict__ w_hi, const uint16_t* __restrict__ w_lo,
                                                           const int dp) {
    static_assert(RT <= KS, "wavefront q finishes row tile q");
    __shared__ __attribute__((aligned(16))) float part[KS][RT * kMlpTiles][64 * 4];   // [share][tile][lane] f32x4
    const int lane = threadIdx.x & 63, kpart = threadIdx.x >> 6;
    const int which = (int)blockIdx.y + first_net;
    const WlMlp& net = which == 0 ? actor : critic;
    const int D = net.in_dim;
    const int m = lane & 15, g = lane >> 4;
    const int row0 = (int)blockIdx.x * (16 * RT);
    const uint16_t* wh = w_hi + (int64_t)(which * kMlpHidden + m) * dp + 8 * g;
    const uint16_t* wl = w_lo + (int64_t)(which * kMlpHidden + m) * dp + 8 * g;
    const float* x_lane[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) x_lane[q] = obs + (int64_t)min(row0 + 16 * q + m, n_rows - 1) * obs_stride + 8 * g;

    f32x4 h[RT][kMlpTiles];
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bias = kpart == 0 ? net.b1[16 * t + 4 * g + r] : 0.f;
#pragma unroll
            for (int q = 0; q < RT; ++q) h[q][t][r] = bias;
        }
    const int n_steps = dp >> 5, per = (n_steps + KS - 1) / KS;
    const int s0 = min(kpart * per, n_steps), s1 = min(s0 + per, n_steps);
    constexpr int kDepth = DEPTH;
    PlaneStep ra[kDepth];
    wl_f4u rb[kDepth][RT][2];
    auto load_step = [&](int j, int s) {
#pragma unroll
        for (int t = 0; t < kMlpTiles; ++t) {
            ra[j].ah[t] = *reinterpret_cast<const u32x4_t*>(wh + (int64_t)16 * t * dp + 32 * s);
            ra[j].al[t] = *reinterpret_cast<const u32x4_t*>(wl + (int64_t)16 * t * dp + 32 * s);
        }
        const int f = plane_feature(s, D);
#pragma unroll
        for (int q = 0; q < RT; ++q) {
            rb[j][q][0] = *reinterpret_cast<const wl_f4u*>(x_lane[q] + f);
            rb[j][q][1] = *reinterpret_cast<const wl_f4u*>(x_lane[q] + f + 4);
        }
    };
#pragma unroll
    for (int j = 0; j < kDepth; ++j)
        if (s0 + j < s1) load_step(j, s0 + j);
    // the tail's weights and the draw of the row tile this wavefront will finish: in the shadow of the operand loads
    MlpTail W;
    float z0 = 0.f, z1 = 0.f;
    if (kpart < RT) {
        load_tail(net, lane, W);
        if (which == 0 && !deterministic) {
            const F4 u = philox_uniform4((uint32_t)(env_offset + row0 + 16 * kpart + m), step, WL_RS_POLICY, seed);
            box_muller(u.x, u.y, z0, z1);
        }
    }
    for (int s = s0; s < s1; s += kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            if (s + j < s1) {
                bf16x8_t bh[RT], bl[RT];
#pragma unroll
                for (int q = 0; q < RT; ++q) {
                    u32x4_t ph, pl;
                    uint32_t a_, b_;
                    split_bf16_pair(rb[j][q][0][0], rb[j][q][0][1], a_, b_); ph[0] = a_; pl[0] = b_;
                    split_bf16_pair(rb[j][q][0][2], rb[j][q][0][3], a_, b_); ph[1] = a_; pl[1] = b_;
                    split_bf16_pair(rb[j][q][1][0], rb[j][q][1][1], a_, b_); ph[2] = a_; pl[2] = b_;
                    split_bf16_pair(rb[j][q][1][2], rb[j][q][1][3], a_, b_); ph[3] = a_; pl[3] = b_;
                    bh[q] = __builtin_bit_cast(bf16x8_t, ph);
                    bl[q] = __builtin_bit_cast(bf16x8_t, pl);
                }
#pragma unroll
                for (int t = 0; t < kMlpTiles; ++t) {
                    const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, ra[j].ah[t]), al = __builtin_bit_cast(bf16x8_t, ra[j].al[t]);
#pragma unroll
                    for (int q = 0; q < RT; ++q) h[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[q], h[q][t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < RT; ++q) h[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[q], h[q][t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < RT; ++q) h[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[q], h[q][t], 0, 0, 0);
                }
                if (s + j + kDepth < s1) load_step(j, s + j + kDepth);
            }
        }
    }
    // every share's accumulators -> LDS; wavefront q sums row tile q over the shares in a fixed order and finishes it
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int t = 0; t < kMlpTiles; ++t) *reinterpret_cast<f32x4*>(&part[kpart][q * kMlpTiles + t][lane * 4]) = h[q][t];
    __syncthreads();
    if (kpart >= RT) return;
    f32x4 hq[kMlpTiles];
#pragma unroll
    for (int t = 0; t < kMlpTiles; ++t) {
        hq[t] = *reinterpret_cast<const f32x4*>(&part[0][kpart * kMlpTiles + t][lane * 4]);
#pragma unroll
        for (int k = 1; k < KS; ++k) hq[t] += *reinterpret_cast<const f32x4*>(&part[k][kpart * kMlpTiles + t][lane * 4]);
    }
    const f32x4 out = eval_tail<ACT>(W, hq, lane);
    const int r_out = row0 + 16 * kpart + m;
    if (g != 0 || r_out >= n_rows) return;
    if (which == 1) {
        values[r_out] = out[0];
        return;
    }
    const float std0 = std[0], std1 = std[1];
    reinterpret_cast<float2*>(actions)[r_out] = make_float2(fmaf(std0, z0, out[0]), fmaf(std1, z1, out[1]));
    reinterpret_cast<float2*>(mu_out)[r_out] = make_float2(out[0], out[1]);
    log_prob[r_out] = fmaf(-0.5f, fmaf(z0, z0, z1 * z1), -(log_fast(std0) + log_fast(std1)) - kLog2PiA);
}

int check_wide(const WlMlp* net, int out_dim) {
    if (!net || !net->w1 || !net->b1 || !net->w2 || !net->b2 || !net->w3 || !net->b3) return WL_EINVAL;
    if (net->hidden != kMlpHidden || net->in_dim < 1 || net->in_dim > (1 << 20) || net->out_dim != out_dim) return WL_EINVAL;
    if (net->activation != WL_ACT_RELU && net->activation != WL_ACT_ELU) return WL_EINVAL;
    return WL_OK;
}

template <int ACT, int RT>
void launch_act(int ks, int row_blocks, hipStream_t s, const WlMlp& a, const WlMlp& c, const float* std, int n_rows,
                const float* obs, int64_t obs_stride, float* actions, float* mu, float* log_prob, float* values, int env_offset,
                uint64_t seed, uint64_t step, int deterministic, int nets) {
    const dim3 grid(row_blocks, nets == 3 ? 2 : 1);
    const int first_net = nets == 2 ? 1 : 0;
#define WL_LAUNCH_ACT(KS)                                                                                                      \
    actor_critic_act_kernel<ACT, KS, RT><<<grid, 64 * KS, 0, s>>>(a, c, std, n_rows, obs, obs_stride, actions, mu, log_prob,    \
                                                                   values, env_offset, seed, step, deterministic, first_net)
    if (ks >= 4) WL_LAUNCH_ACT(4);
    else if (ks == 2) WL_LAUNCH_ACT(2);
    else WL_LAUNCH_ACT(1);
#undef WL_LAUNCH_ACT
}

}  // namespace

extern "C" {

int wl_actor_critic_act(const WlMlp* actor, const WlMlp* critic, const float* std, int32_t n_rows, const float* obs,
                        int64_t obs_stride, float* actions, float* mu, float* log_prob, float* values, int32_t env_offset,
                        uint64_t seed, uint64_t step, int32_t deterministic, int32_t nets, void* stream) {
    int rc = check_wide(actor, 2);
    if (rc == WL_OK) rc = check_wide(critic, 1);
    if (rc != WL_OK) return rc;
    if (actor->in_dim != critic->in_dim || actor->activation != critic->activation) return WL_EINVAL;
    if (nets < 1 || nets > 3 || n_rows <= 0 || !obs || obs_stride < actor->in_dim) return WL_EINVAL;
    if ((nets & 1) && (!std || !actions || !mu || !log_prob)) return WL_EINVAL;
    if ((nets & 2) && !values) return WL_EINVAL;
    if ((nets & 1) && (((uintptr_t)actions & 7u) || ((uintptr_t)mu & 7u))) return WL_EALIGN;
    if ((uintptr_t)obs & 3u) return WL_EALIGN;
    clear_error();
    const int tiles = (n_rows + 15) / 16;
    // Row tiles per wavefront by the number of (tile, net) pairs: one while a wavefront per pair does not yet fill the 1024
    // SIMDs four times over with the feature split, two / four beyond (each weight operand then feeds more rows).  The
    // feature split KS is chosen from the row count alone (as for a joint launch), so the arithmetic -- the order of the
    // partial sums -- and hence the result does not depend on `nets`.
    const int n_nets = nets == 3 ? 2 : 1;
    const int rt = tiles * n_nets >= 2048 ? 4 : tiles * n_nets >= 512 ? 2 : 1;
    const int row_blocks = (tiles + rt - 1) / rt;
    const int rt_joint = tiles * 2 >= 2048 ? 4 : tiles * 2 >= 512 ? 2 : 1;
    const int row_blocks_joint = (tiles + rt_joint - 1) / rt_joint;
    int ks = 1;
    while (ks < 4 && row_blocks_joint * 2 * ks < 1024 && (actor->in_dim >> 4) / (ks * 2) >= 8) ks *= 2;
    const bool elu = actor->activation == WL_ACT_ELU;
#define WL_ACT_ARGS ks, row_blocks, (hipStream_t)stream, *actor, *critic, std, n_rows, obs, obs_stride, actions, mu, log_prob, values, \
                    env_offset, seed, step, deterministic, nets
    if (rt == 4) {
        if (elu) launch_act<WL_ACT_ELU, 4>(WL_ACT_ARGS);
        else launch_act<WL_ACT_RELU, 4>(WL_ACT_ARGS);
    } else if (rt == 2) {
        if (elu) launch_act<WL_ACT_ELU, 2>(WL_ACT_ARGS);
        else launch_act<WL_ACT_RELU, 2>(WL_ACT_ARGS);
    } else {
        if (elu) launch_act<WL_ACT_ELU, 1>(WL_ACT_ARGS);
        else launch_act<WL_ACT_RELU, 1>(WL_ACT_ARGS);
    }
#undef WL_ACT_ARGS
    return launch_status();
}

int wl_actor_critic_planes(const WlMlp* actor, const WlMlp* critic, const WlActScratch* sc, void* stream) {
    int rc = check_wide(actor, 2);
    if (rc == WL_OK) rc = check_wide(critic, 1);
    if (rc != WL_OK) return rc;
    if (!sc || !sc->w_hi || !sc->w_lo || actor->in_dim != critic->in_dim || actor->in_dim < 64 || sc->dp != (actor->in_dim + 63) / 64 * 64)
        return WL_EINVAL;
    if (((uintptr_t)sc->w_hi & 15u) || ((uintptr_t)sc->w_lo & 15u)) return WL_EALIGN;
    return wl_internal::mlp_weight_planes(actor, critic, sc->dp, sc->w_hi, sc->w_lo, (hipStream_t)stream);
}

int wl_actor_critic_act_planes(const WlMlp* actor, const WlMlp* critic, const float* std, int32_t n_rows, const float* obs,
                               int64_t obs_stride, float* actions, float* mu, float* log_prob, float* values, int32_t env_offset,
                               uint64_t seed, uint64_t step, int32_t deterministic, int32_t nets, const WlActScratch* sc,
                               void* stream) {
    int rc = check_wide(actor, 2);
    if (rc == WL_OK) rc = check_wide(critic, 1);
    if (rc != WL_OK) return rc;
    if (actor->in_dim != critic->in_dim || actor->activation != critic->activation) return WL_EINVAL;
    if (nets < 1 || nets > 3 || n_rows <= 0 || !obs || obs_stride < actor->in_dim) return WL_EINVAL;
    if ((nets & 1) && (!std || !actions || !mu || !log_prob)) return WL_EINVAL;
    if ((nets & 2) && !values) return WL_EINVAL;
    if ((nets & 1) && (((uintptr_t)actions & 7u) || ((uintptr_t)mu & 7u))) return WL_EALIGN;
    if ((uintptr_t)obs & 3u) return WL_EALIGN;
    if (!sc || !sc->w_hi || !sc->w_lo || !sc->partials || actor->in_dim < 64 || sc->dp != (actor->in_dim + 63) / 64 * 64 ||
        sc->splits < (sc->reserved == 2 ? 1 : (sc->dp + 127) / 128) || n_rows > sc->rows_capacity || sc->reserved < 0 || sc->reserved > 2)
        return WL_EINVAL;
    if (((uintptr_t)sc->w_hi & 15u) || ((uintptr_t)sc->w_lo & 15u) || ((uintptr_t)sc->partials & 15u)) return WL_EALIGN;
    if (sc->reserved == 0) {   // one launch: feature shares folded through LDS (reserved = 1: the two-launch split-K form below)
        const int tiles = (n_rows + 15) / 16, n_nets = nets == 3 ? 2 : 1, first = nets == 2 ? 1 : 0;
        const bool elu = actor->activation == WL_ACT_ELU;
        clear_error();
#define WL_BF16_ACT(KS, RT)                                                                                                         \
    do {                                                                                                                            \
        const dim3 grid((tiles + RT - 1) / RT, n_nets);                                                                             \
        if (elu) act_bf16_kernel<WL_ACT_ELU, KS, RT><<<grid, 64 * KS, 0, (hipStream_t)stream>>>(*actor, *critic, std, n_rows, obs, obs_stride, actions, mu, log_prob, values, env_offset, seed, step, deterministic, first, sc->w_hi, sc->w_lo, sc->dp); \
        else act_bf16_kernel<WL_ACT_RELU, KS, RT><<<grid, 64 * KS, 0, (hipStream_t)stream>>>(*actor, *critic, std, n_rows, obs, obs_stride, actions, mu, log_prob, values, env_offset, seed, step, deterministic, first, sc->w_hi, sc->w_lo, sc->dp); \
    } while (0)
        // rows per block by the batch (fewer re-reads of the weight planes while enough blocks remain), the feature split by the width
        // (the feature split depends on the WIDTH only: a row's result must not depend on the batch it arrives in)
        if (sc->dp >= 1024) {
            if (tiles >= 96) WL_BF16_ACT(8, 2);
            else WL_BF16_ACT(8, 1);
        } else {
            if (tiles >= 192) WL_BF16_ACT(4, 4);
            else if (tiles >= 96) WL_BF16_ACT(4, 2);
            else WL_BF16_ACT(4, 1);
        }
#undef WL_BF16_ACT
        return launch_status();
    }
    const int splits = wl_internal::layer1_partials(obs, obs_stride, n_rows, actor->in_dim, sc->dp, sc->w_hi, sc->w_lo, sc->splits,
                                                    sc->partials, (hipStream_t)stream, sc->reserved == 2);
    if (splits < 0) return splits;
    const dim3 grid((n_rows + 15) / 16, nets == 3 ? 2 : 1);
    const int first_net = nets == 2 ? 1 : 0;
    if (actor->activation == WL_ACT_ELU)
        act_tail_kernel<WL_ACT_ELU><<<grid, 64, 0, (hipStream_t)stream>>>(*actor, *critic, std, n_rows, sc->partials, splits, actions, mu,
                                                                          log_prob, values, env_offset, seed, step, deterministic, first_net);
    else
        act_tail_kernel<WL_ACT_RELU><<<grid, 64, 0, (hipStream_t)stream>>>(*actor, *critic, std, n_rows, sc->partials, splits, actions, mu,
                                                                           log_prob, values, env_offset, seed, step, deterministic, first_net);
    return launch_status();
}

}  // extern "C"

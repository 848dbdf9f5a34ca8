// wl_ppo_wide.hip -- the PPO minibatch step of the WIDE agents (D-64-64-2 actor / D-64-64-1 critic, D = 689 elevation /
// 3208 visual; rsl_rl PPO.update as the reference drives it: wheeledlab_rl/utils/modified_rsl_rl_runner.py:104-118 with the
// agent configs under wheeledlab_tasks/{elevation,visual}/config/agents).
//
// Everything behind the first layer is the drift agents' kernel (wl_ppo.hip, first layer cut off).  The first layer is two
// streaming contractions per minibatch over the [B][D] observation block (361 MB for 131 072 x 689 f32):
//     H1      = act(X . W1^T + b1)          [B][128]     (both nets: 128 units, K = D)
//     dW1^T   = X^T . delta1                [D][128]     (K = B: split over the samples, partial sums reduced afterwards)
// In f32 on the matrix pipe these are 2 x 23 GFLOP at 157 TFLOP/s = 0.3 ms -- slower than reading X.  Here every f32 operand
// is split into two bf16 planes, x = hi + lo (16 mantissa bits), and a product is hi.hi + lo.hi + hi.lo with f32 accumulation
// (v_mfma_f32_16x16x32_bf16, 16x the f32 rate): 3 x 23 GFLOP at 2.5 PFLOP/s = 28 us, under the ~70 us it takes to stream the
// planes (4 bytes per element, same as f32).  The error of the dropped lo.lo term and the plane rounding is ~2^-17 relative
// per product (tests/test_gpu_ppo_wide.py: gradients vs torch autograd).
//
// One kernel does both contractions (`skinny_kernel`): Out[r][u] = sum_k Bm[r][k] . A[u][k] with both operands K-contiguous,
//   * A = the SHARED operand (128 rows: W1 of both nets, or delta1^T) -- staged through LDS in 64-wide K chunks, double
//     buffered, 16-byte pieces XOR-swizzled so that ds_write_b128 (8-lane groups) and ds_read_b128 (the four 16-lane groups
//     of MI355X_MICROARCH.md, LDS section) are conflict free;
//   * Bm = the STREAMED operand (observation rows, or rows of X^T) -- each wavefront loads the MFMA fragments of its own 32
//     rows straight from memory one chunk ahead (lane (g, n): row n, 16 bytes at k = 8 g: half a cache line per row and
//     instruction, every byte used once);
//   * a block = 4 wavefronts x 32 rows, two blocks per CU; per 64-wide chunk and wavefront 96 MFMAs, 32 ds_read_b128, 16
//     fragment loads.
// X^T exists because the MFMA wants both operands contiguous along the contraction index, and dW1 contracts over samples:
// `wl_ppo_wide_stage` writes its planes (through an LDS tile) once per update, in the order of the update's permutation,
// so that the minibatches of all its epochs are contiguous column ranges.  The forward contraction needs no staged copy:
// it reads the f32 observation rows through the permutation and splits them in registers (v_cvt_pk_bf16_f32, 5
// instructions per pair, hidden under the MFMAs) -- the same bytes as reading two planes.
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_bf16.h"
#include "wl_kernel_common.h"
#include "wl_ppo_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kHid = 64, kUnits = 2 * kHid;      // both nets side by side
constexpr int kRowN = WL_PPO_PARTIAL_STRIDE;     // narrow row (drift layout, in = 14)
constexpr int kInN = 14;

typedef float wl_f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment (rows of odd length)

// ---- staging: obs rows in permuted order -> X^T planes, blocked [row / 64][dp][row % 64] ------------------------------------
// (a 64-row K chunk of X^T is then one contiguous 2 * 64 * dp-byte run instead of dp pieces a row of `capacity` apart).
// X itself is not staged: the forward contraction reads the f32 rows through `perm` and splits them in registers.
// block = 64 rows x 64 features; thread (ty, tx) = (i >> 4, i & 15): rows 16 m + ty, features 4 tx .. 4 tx + 3
__global__ void __launch_bounds__(256) ppo_wide_stage_kernel(const float* __restrict__ obs, const int32_t* __restrict__ perm,
                                                             const int in_dim, const int dp, uint16_t* __restrict__ xt_hi,
                                                             uint16_t* __restrict__ xt_lo) {
    __shared__ uint16_t t_hi[64][68], t_lo[64][68];   // [feature][row]; 136-byte rows: 8-byte reads stay aligned
    const int r0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int r = 16 * m + ty;
        const float* src = obs + (int64_t)perm[r0 + r] * in_dim;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = f0 + 4 * tx + j;
            v[j] = f < in_dim ? src[f] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            uint32_t h, l;
            split_bf16_pair(v[j], v[j + 1], h, l);
            t_hi[4 * tx + j][r] = (uint16_t)h;
            t_hi[4 * tx + j + 1][r] = (uint16_t)(h >> 16);
            t_lo[4 * tx + j][r] = (uint16_t)l;
            t_lo[4 * tx + j + 1][r] = (uint16_t)(l >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int f = 16 * m + ty;
        const int64_t at = ((int64_t)blockIdx.x * dp + f0 + f) * 64 + 4 * tx;   // [row chunk][feature][64 rows]
        *reinterpret_cast<uint2*>(xt_hi + at) = *reinterpret_cast<const uint2*>(&t_hi[f][4 * tx]);
        *reinterpret_cast<uint2*>(xt_lo + at) = *reinterpret_cast<const uint2*>(&t_lo[f][4 * tx]);
    }
}

// layer-1 weights of both nets -> planes [128][dp] (wl_bf16.h: weight_plane_pair)
__global__ void __launch_bounds__(256) ppo_wide_weights_kernel(const float* __restrict__ w1_actor, const float* __restrict__ w1_critic,
                                                               const int in_dim, const int dp, uint32_t* __restrict__ w_hi,
                                                               uint32_t* __restrict__ w_lo) {
    weight_plane_pair(w1_actor, w1_critic, in_dim, dp, blockIdx.x * 256 + threadIdx.x, w_hi, w_lo);
}

// ---- the contraction ----------------------------------------------------------------------------------------------------------
struct SkinnyArgs {
    const uint16_t *b_hi, *b_lo;   // streamed operand as bf16 planes, rows_b rows, already offset to the first k of the contraction
    const float* b_f32;            // ... or (BF32) as f32 rows [*][b_row] of b_k >= 64 features, split in registers
    const int32_t* b_perm;         // BF32: row r of the operand is row b_perm[r] of b_f32 (NULL: r)
    const uint16_t *a_hi, *a_lo;   // shared operand, 128 rows, likewise
    float* out;                    // [splits][rows_b][128]
    const float *bias_a, *bias_c;  // epilogue: + bias (units 0..63 / 64..127), then the activation
    int rows_b, row_blocks, b_k;
    int64_t b_row, b_chunk, a_row, a_chunk;   // element (r, k) of an operand sits at r * row + (k >> 6) * chunk + (k & 63)
    int n_chunks, chunks_per_split, splits;   // split s contracts the 64-wide K chunks [s cps, min((s + 1) cps, n_chunks))
};

constexpr int kChunk = 64;                       // K elements per LDS stage
constexpr int kPlaneBytes = kUnits * kChunk * 2; // 16 KB: [128 rows][128 B], 16-byte pieces swizzled
WL_DEV int piece_offset(int row, int piece) { return row * (kChunk * 2) + ((piece ^ ((row >> 1) & 7)) << 4); }

WL_DEV float act_elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }

// the streamed operand's registers for one 64-wide chunk: per (row tile q, k-step j) either the two planes' fragments
// (8 bf16 each) or the 8 raw floats they are split from
struct BRegs {
    u32x4 x[2][2], y[2][2];   // planes: x = hi, y = lo; BF32: x = floats 0..3, y = floats 4..7
};

template <int EPI /* 0: raw partial sums, WL_ACT_ELU / WL_ACT_RELU + 1: bias + activation */, bool BF32>
__global__ void __launch_bounds__(256, 2) skinny_kernel(const SkinnyArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * kPlaneBytes];   // [buffer][hi / lo][kPlaneBytes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
    int rb = blockIdx.x, sp = 0;
    if (a.splits > 1) {   // the blocks of one split share the A stream: keep them on one XCD (blockIdx round-robins over 8)
        sp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * a.row_blocks));
        rb = (blockIdx.x >> 3) % a.row_blocks;
        if (sp >= a.splits) return;   // the grid is padded to a multiple of 8 splits
    }
    const int64_t c0 = (int64_t)sp * a.chunks_per_split;   // first K chunk of this split
    const int n_chunks = min(a.chunks_per_split, a.n_chunks - (int)c0);

    // streamed operand: fragments of rows row0 + 16 q + n, k = 32 j + 8 g .. + 7 of the chunk
    const int row0 = rb * 128 + 32 * wave;
    const uint16_t *pbh[2], *pbl[2];
    const float* pbf[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = min(row0 + 16 * q + n, a.rows_b - 1);
        if constexpr (BF32) {
            const int64_t src = a.b_perm ? a.b_perm[r] : r;
            pbf[q] = a.b_f32 + src * a.b_row + 8 * g;
            pbh[q] = pbl[q] = nullptr;
        } else {
            pbh[q] = a.b_hi + r * a.b_row + c0 * a.b_chunk + 8 * g;
            pbl[q] = a.b_lo + r * a.b_row + c0 * a.b_chunk + 8 * g;
            pbf[q] = nullptr;
        }
    }
    // BF32: chunk c of a row starts at feature min(64 c, b_k - 64) (the weight planes' last chunk is laid out to match)
    const int k_last = a.b_k - kChunk;
    // shared operand: thread -> piece (tid & 7) of rows (tid >> 3) + 32 m
    const int sp_piece = tid & 7, sp_row = tid >> 3;
    const uint16_t* pah = a.a_hi + sp_row * a.a_row + c0 * a.a_chunk + 8 * sp_piece;
    const uint16_t* pal = a.a_lo + sp_row * a.a_row + c0 * a.a_chunk + 8 * sp_piece;
    const int64_t a_row_step = 32 * a.a_row;

    f32x4 acc[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    BRegs cur, nxt;
    u32x4 sh[4], sl[4];
#define WL_LOAD_B(R, C)                                                                                     \
    if constexpr (!BF32) {                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j) {       \
            R.x[q][j] = *reinterpret_cast<const u32x4*>(pbh[q] + (C) * a.b_chunk + 32 * j);                 \
            R.y[q][j] = *reinterpret_cast<const u32x4*>(pbl[q] + (C) * a.b_chunk + 32 * j);                 \
        }                                                                                                   \
    } else {                                                                                                \
        const int k = min(((int)c0 + (C)) * kChunk, k_last);                                                \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j) {       \
            const float* src = pbf[q] + k + 32 * j;                                                         \
            R.x[q][j] = __builtin_bit_cast(u32x4, *reinterpret_cast<const wl_f4u*>(src));                   \
            R.y[q][j] = __builtin_bit_cast(u32x4, *reinterpret_cast<const wl_f4u*>(src + 4));               \
        }                                                                                                   \
    }
#define WL_LOAD_A(C)                                                                        \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                         \
        sh[m] = *reinterpret_cast<const u32x4*>(pah + m * a_row_step + (C) * a.a_chunk);        \
        sl[m] = *reinterpret_cast<const u32x4*>(pal + m * a_row_step + (C) * a.a_chunk);        \
    }
#define WL_STORE_A(BUF)                                                                     \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                         \
        const int off = piece_offset(sp_row + 32 * m, sp_piece);                             \
        *reinterpret_cast<u32x4*>(lds + (BUF) * 2 * kPlaneBytes + off) = sh[m];              \
        *reinterpret_cast<u32x4*>(lds + (BUF) * 2 * kPlaneBytes + kPlaneBytes + off) = sl[m]; \
    }

    WL_LOAD_A(0)
    WL_LOAD_B(cur, 0)
    WL_STORE_A(0)
    // all prologue loads have landed before the loop: otherwise the wait-count pass, merging the loop entry with the back
    // edge, protects the first use of `cur` inside the loop with a vmcnt that also drains the prefetch just issued
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    for (int c = 0; c < n_chunks; ++c) {
        const bool more = c + 1 < n_chunks;
        if (more) {
            WL_LOAD_A(c + 1)
            WL_LOAD_B(nxt, c + 1)
        }
        __syncthreads();   // buffer c & 1 is complete; everybody is done with buffer (c + 1) & 1
        const unsigned char* ph = lds + (c & 1) * 2 * kPlaneBytes;
        const unsigned char* pl = ph + kPlaneBytes;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if constexpr (BF32) {   // 8 floats -> the two planes' fragments
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        uint32_t h0, l0, h1, l1;
                        split_bf16_pair(__uint_as_float(cur.x[q][j][2 * i]), __uint_as_float(cur.x[q][j][2 * i + 1]), h0, l0);
                        split_bf16_pair(__uint_as_float(cur.y[q][j][2 * i]), __uint_as_float(cur.y[q][j][2 * i + 1]), h1, l1);
                        bh[q][i] = h0;
                        bl[q][i] = l0;
                        bh[q][2 + i] = h1;
                        bl[q][2 + i] = l1;
                    }
                } else {
                    bh[q] = cur.x[q][j];
                    bl[q] = cur.y[q][j];
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int off = piece_offset(16 * t + n, 4 * j + g);
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ph + off);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(pl + off);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, bh[q]), acc[q][t], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, bl[q]), acc[q][t], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, bh[q]), acc[q][t], 0, 0, 0);
            }
        }
        if (more) {
            WL_STORE_A((c + 1) & 1)
            cur = nxt;
        }
    }
#undef WL_LOAD_A
#undef WL_LOAD_B
#undef WL_STORE_A

    // accumulator of tile (q, t): lane (g, n) holds units 16 t + 4 g .. + 3 of row row0 + 16 q + n -> one 16-byte store
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = row0 + 16 * q + n;
        float* dst = a.out + ((int64_t)sp * a.rows_b + r) * kUnits + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 v = acc[q][t];
            if constexpr (EPI != 0) {
                const float* bias = (t < 4 ? a.bias_a : a.bias_c) + 16 * (t & 3) + 4 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z = v[i] + bias[i];
                    v[i] = EPI == WL_ACT_RELU + 1 ? fmaxf(z, 0.f) : act_elu(z);
                }
            }
            if (r < a.rows_b) *reinterpret_cast<f32x4*>(dst + 16 * t) = v;
        }
    }
}

// ---- narrow row + dW1 partial sums -> the flat gradient (torch named_parameters() order for input width D) ------------------
struct WideLayout {
    int in, o_aw1, o_arest, o_cw1, o_crest, G;   // "rest" = b1 onwards
};
inline WideLayout wide_layout(int in) {
    const int per_net = kHid * in + kHid + kHid * kHid + kHid;
    WideLayout L;
    L.in = in;
    L.o_aw1 = 2;
    L.o_arest = 2 + kHid * in;
    L.o_cw1 = 2 + per_net + 2 * kHid + 2;
    L.o_crest = L.o_cw1 + kHid * in;
    L.G = 2 + 2 * per_net + 3 * kHid + 3;
    return L;
}
// narrow-row offsets of the same places (in = 14)
constexpr int kN_AB1 = 2 + kHid * kInN, kN_CW1 = 2 + (kHid * kInN + kHid + kHid * kHid + kHid) + 2 * kHid + 2,
              kN_CB1 = kN_CW1 + kHid * kInN, kN_G = WL_PPO_NUM_PARAMS;
constexpr int kActorRest = kHid + kHid * kHid + kHid + 2 * kHid + 2, kCriticRest = kHid + kHid * kHid + kHid + kHid + 1;
static_assert(kN_CB1 + kCriticRest == kN_G, "narrow layout");

// blocks [0, nb_w1): thread -> (feature, unit): sum over the splits (coalesced over the unit index), write grad[w1 slot],
// add the squares to *norm2.  Remaining blocks: the reduction of the gradient kernel's per-block rows (`partials`
// [n_rows][kRowN], the drift agents' layout with its first-layer weight slots unused) straight into their wide places + the
// three statistics -- the arithmetic of ppo_reduce_kernel (64 columns per block, four row groups of threads), whose launch
// and whose place in the dependent chain (it ran between the tail and the dW1 contraction) this saves; and the std snapshot.
// VEC = 1: few features, many splits (the elevation agent: 704 x 128 sums of 128 terms) -- a thread per sum.  VEC = 4: many
// features, few splits (the visual agent: 3264 x 128 sums of 16): a thread per four units, 16-byte loads, all of a
// thread's loads in flight at once (a thread per sum ran at 0.9 TB/s there: 16 dependent-looking 4-byte loads each).
// The same order of summation in both.
template <int VEC>
__global__ void __launch_bounds__(256) ppo_wide_scatter_kernel(const float* __restrict__ dw_partials, const int splits, const int dp,
                                                               const float* __restrict__ partials, const int n_rows, const WideLayout L,
                                                               const int nb_w1, float* __restrict__ grad, float* __restrict__ norm2,
                                                               const float* __restrict__ std, float* __restrict__ std_snapshot) {
    if ((int)blockIdx.x < nb_w1) {
        typedef float vec_t __attribute__((ext_vector_type(VEC)));
        const int i = (blockIdx.x * 256 + threadIdx.x) * VEC;    // = f * 128 + u
        const int f = i >> 7, u = i & 127;
        float q = 0.f;
        if (f < L.in) {
            vec_t s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const int64_t plane = (int64_t)dp * kUnits;
            const float* src = dw_partials + i;
            int s = 0;
#pragma unroll 4
            for (; s + 3 < splits; s += 4) {
                s0 += *reinterpret_cast<const vec_t*>(src + (s + 0) * plane);
                s1 += *reinterpret_cast<const vec_t*>(src + (s + 1) * plane);
                s2 += *reinterpret_cast<const vec_t*>(src + (s + 2) * plane);
                s3 += *reinterpret_cast<const vec_t*>(src + (s + 3) * plane);
            }
            for (; s < splits; ++s) s0 += *reinterpret_cast<const vec_t*>(src + s * plane);
            const vec_t v = (s0 + s1) + (s2 + s3);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float vj = v[j];
                const int uj = u + j;
                grad[(uj < kHid ? L.o_aw1 + uj * L.in : L.o_cw1 + (uj - kHid) * L.in) + f] = vj;
                q = fmaf(vj, vj, q);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(norm2, (part[0] + part[1]) + (part[2] + part[3]));
        return;
    }
    const int rb = blockIdx.x - nb_w1;
    if (rb == 0 && threadIdx.x < 2) std_snapshot[threadIdx.x] = std[threadIdx.x];   // see ppo_reduce_kernel
    __shared__ float part[4][64];
    const int j = rb * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;   // j = index into the narrow row
    int to = -1;
    if (j < 2) to = j;
    else if (j < kN_AB1) to = -1;                              // actor w1 slots: unused
    else if (j < kN_CW1) to = L.o_arest + (j - kN_AB1);
    else if (j < kN_CB1) to = -1;                              // critic w1 slots
    else if (j < kN_G) to = L.o_crest + (j - kN_CB1);
    else if (j < kRowN) to = L.G + (j - kN_G);                 // value-loss, surrogate, KL sums
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (to >= 0) {
        int b = rg;
#pragma unroll 4
        for (; b + 12 < n_rows; b += 16) {
            s0 += partials[(int64_t)b * kRowN + j];
            s1 += partials[(int64_t)(b + 4) * kRowN + j];
            s2 += partials[(int64_t)(b + 8) * kRowN + j];
            s3 += partials[(int64_t)(b + 12) * kRowN + j];
        }
        for (; b < n_rows; b += 4) s0 += partials[(int64_t)b * kRowN + j];
    }
    part[rg][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0) {
        const float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (to >= 0) grad[to] = v;
        float q = (to >= 0 && j < kN_G) ? v * v : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
        if (threadIdx.x == 0) atomicAdd(norm2, q);
    }
}

int check_wide(const WlMlp* actor, const WlMlp* critic, const WlPpoWideState* st) {
    if (!actor || !critic || !st) return WL_EINVAL;
    for (const WlMlp* m : {actor, critic})
        if (!m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3 || m->hidden != kHid ||
            (m->activation != WL_ACT_ELU && m->activation != WL_ACT_RELU))
            return WL_EINVAL;
    if (actor->in_dim != st->in_dim || critic->in_dim != st->in_dim || actor->out_dim != 2 || critic->out_dim != 1 ||
        actor->activation != critic->activation)
        return WL_EINVAL;
    if (st->in_dim < 64 || st->dp != (st->in_dim + 63) / 64 * 64 || st->capacity <= 0 || (st->capacity & 63) || st->mb_capacity <= 0 ||
        (st->mb_capacity & 63) || st->splits < 1)
        return WL_EINVAL;
    if (!st->xt_hi || !st->xt_lo || !st->w_hi || !st->w_lo || !st->h1 || !st->dt_hi || !st->dt_lo ||
        !st->dw_partials || !st->partials || !st->narrow || !st->grad || !st->ctrl || !st->operands || ((uintptr_t)st->operands & 15u))
        return WL_EINVAL;
    for (const void* p : {(const void*)st->xt_hi, (const void*)st->xt_lo, (const void*)st->w_hi,
                          (const void*)st->w_lo, (const void*)st->h1, (const void*)st->dt_hi, (const void*)st->dt_lo,
                          (const void*)st->dw_partials})
        if ((uintptr_t)p & 15u) return WL_EINVAL;
    return WL_OK;
}

int launch_wide_gradients(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* bt, int mb_start, int mb_size,
                          const WlPpoParams* hp, const WlPpoWideState* st, int parity, hipStream_t stream) {
    const int dp = st->dp, D = st->in_dim;
    clear_error();
    // one launch for everything that depends on the weights only: the tail's operand tables and layer 1's bf16 planes
    int rc = wl_internal::ppo_prepare_wide(actor, critic, std, st->operands, dp, st->w_hi, st->w_lo, stream);
    if (rc != WL_OK) return rc;
    {   // H1 = act(X W1^T + b1): rows = samples of the minibatch (f32 rows of `obs` through `perm`), K = dp
        SkinnyArgs a{};
        a.b_f32 = bt->obs;
        a.b_perm = bt->perm + mb_start;
        a.b_row = D;
        a.b_k = D;
        a.a_hi = st->w_hi;
        a.a_lo = st->w_lo;
        a.out = st->h1;
        a.bias_a = actor->b1;
        a.bias_c = critic->b1;
        a.rows_b = mb_size;
        a.row_blocks = (mb_size + 127) / 128;
        a.a_row = dp;        // row-major [unit][dp]
        a.a_chunk = kChunk;
        a.n_chunks = a.chunks_per_split = dp / kChunk;
        a.splits = 1;
        if (actor->activation == WL_ACT_ELU) skinny_kernel<WL_ACT_ELU + 1, true><<<a.row_blocks, 256, 0, stream>>>(a);
        else skinny_kernel<WL_ACT_RELU + 1, true><<<a.row_blocks, 256, 0, stream>>>(a);
        if (launch_status() != WL_OK) return WL_ELAUNCH;
    }
    float* norm2 = st->ctrl + WL_PPO_CTRL_NORM2 + parity;
    const int n_rows = wl_internal::ppo_tail_wide(actor, critic, std, bt, mb_start, mb_size, hp, st->partials, st->operands, st->h1,
                                                  st->dt_hi, st->dt_lo, stream);   // rows of `partials` it filled
    if (n_rows < 0) return n_rows;
    int splits_used = 1;
    {   // dW1^T = X^T delta1: rows = features, K = the minibatch's samples, split over <= `splits` blocks per row block
        SkinnyArgs a{};
        a.b_hi = st->xt_hi + (int64_t)mb_start * dp;   // blocked [sample / 64][dp][64]: chunk mb_start / 64
        a.b_lo = st->xt_lo + (int64_t)mb_start * dp;
        a.a_hi = st->dt_hi;
        a.a_lo = st->dt_lo;
        a.out = st->dw_partials;
        a.rows_b = dp;
        a.row_blocks = (dp + 127) / 128;
        a.b_row = kChunk;
        a.b_chunk = (int64_t)dp * kChunk;
        a.a_row = kChunk;                               // delta1^T likewise: [sample / 64][128][64]
        a.a_chunk = (int64_t)kUnits * kChunk;
        a.n_chunks = mb_size / kChunk;
        a.chunks_per_split = (a.n_chunks + st->splits - 1) / st->splits;
        a.splits = splits_used = (a.n_chunks + a.chunks_per_split - 1) / a.chunks_per_split;   // <= st->splits, none empty
        skinny_kernel<0, false><<<a.row_blocks * ((a.splits + 7) / 8 * 8), 256, 0, stream>>>(a);
    }
    const WideLayout L = wide_layout(D);
    const int nb_rest = (kRowN + 63) / 64;
    float* snap = st->ctrl + WL_PPO_CTRL_STD;
    if (splits_used <= 32 && dp >= 1024) {
        const int nb_w1 = dp * kUnits / (256 * 4);
        ppo_wide_scatter_kernel<4><<<nb_w1 + nb_rest, 256, 0, stream>>>(st->dw_partials, splits_used, dp, st->partials, n_rows, L, nb_w1,
                                                                         st->grad, norm2, std, snap);
    } else {
        const int nb_w1 = dp * kUnits / 256;
        ppo_wide_scatter_kernel<1><<<nb_w1 + nb_rest, 256, 0, stream>>>(st->dw_partials, splits_used, dp, st->partials, n_rows, L, nb_w1,
                                                                         st->grad, norm2, std, snap);
    }
    return launch_status();
}

int check_batch(const float* std, const WlPpoBatch* bt, int mb_start, int mb_size, const WlPpoWideState* st) {
    if (!std || !bt || !bt->obs || !bt->actions || !bt->mu_old || !bt->logp_old || !bt->adv || !bt->returns || !bt->values_old || !bt->perm ||
        !bt->sigma_old)
        return WL_EINVAL;
    if (mb_start < 0 || mb_size <= 0 || (mb_start & 63) || (mb_size & 63) || mb_size > st->mb_capacity ||
        mb_start + mb_size > st->capacity)
        return WL_EINVAL;
    return WL_OK;
}

}  // namespace

namespace wl_internal {

int mlp_weight_planes(const WlMlp* actor, const WlMlp* critic, int dp, uint16_t* w_hi, uint16_t* w_lo, hipStream_t stream) {
    clear_error();
    ppo_wide_weights_kernel<<<(kUnits * dp / 2 + 255) / 256, 256, 0, stream>>>(actor->w1, critic->w1, actor->in_dim, dp, (uint32_t*)w_hi,
                                                                                (uint32_t*)w_lo);
    return launch_status();
}

int layer1_partials(const float* x, int64_t x_stride, int n_rows, int in_dim, int dp, const uint16_t* w_hi, const uint16_t* w_lo,
                    int max_splits, float* out, hipStream_t stream, bool whole_k) {
    SkinnyArgs a{};
    a.b_f32 = x;
    a.b_perm = nullptr;
    a.b_row = x_stride;
    a.b_k = in_dim;
    a.a_hi = w_hi;
    a.a_lo = w_lo;
    a.out = out;
    a.rows_b = n_rows;
    a.row_blocks = (n_rows + 127) / 128;
    a.a_row = dp;
    a.a_chunk = kChunk;
    a.n_chunks = dp / kChunk;
    // a fixed share per split: the grouping of the partial sums -- hence every bit of the result -- depends on the width
    // of the rows only, not on how many rows the call has (a shard of a batch must reproduce its rows of the batch)
    a.chunks_per_split = whole_k ? a.n_chunks : 2;   // whole_k: one sum per row and unit (long batches: a sixth of the partial-sum traffic)
    a.splits = (a.n_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
    if (a.splits > max_splits) return WL_EINVAL;
    clear_error();
    const int grid = a.splits > 1 ? a.row_blocks * ((a.splits + 7) / 8 * 8) : a.row_blocks;
    skinny_kernel<0, true><<<grid, 256, 0, stream>>>(a);
    return launch_status() == WL_OK ? a.splits : WL_ELAUNCH;
}

}  // namespace wl_internal

extern "C" {

int32_t wl_ppo_wide_num_params(int32_t in_dim) { return in_dim < 1 ? 0 : wide_layout(in_dim).G; }

int wl_ppo_wide_stage(const float* obs, const int32_t* perm, int32_t n_rows, const WlPpoWideState* st, void* stream) {
    if (!obs || !perm || !st || !st->xt_hi || !st->xt_lo || n_rows <= 0 || (n_rows & 63) ||
        n_rows > st->capacity || st->dp != (st->in_dim + 63) / 64 * 64 || st->in_dim < 1)
        return WL_EINVAL;
    clear_error();
    ppo_wide_stage_kernel<<<dim3(n_rows / 64, st->dp / 64), 256, 0, (hipStream_t)stream>>>(obs, perm, st->in_dim, st->dp,
                                                                                            st->xt_hi, st->xt_lo);
    return launch_status();
}

int wl_ppo_wide_gradients(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* batch, int32_t mb_start,
                          int32_t mb_size, const WlPpoParams* hp, const WlPpoWideState* state, int32_t parity, void* stream) {
    int rc = check_wide(actor, critic, state);
    if (rc == WL_OK) rc = check_batch(std, batch, mb_start, mb_size, state);
    if (rc != WL_OK) return rc;
    if (!hp || (parity != 0 && parity != 1)) return WL_EINVAL;
    return launch_wide_gradients(actor, critic, std, batch, mb_start, mb_size, hp, state, parity, (hipStream_t)stream);
}

int wl_ppo_wide_apply(const WlMlp* actor, const WlMlp* critic, float* std, int32_t mb_size, const WlPpoParams* hp,
                      const WlPpoWideState* state, int32_t parity, int32_t adam_step, void* stream) {
    int rc = check_wide(actor, critic, state);
    if (rc != WL_OK) return rc;
    if (!std || !hp || !state->adam_m || !state->adam_v || mb_size <= 0 || (parity != 0 && parity != 1) || adam_step < 1)
        return WL_EINVAL;
    return wl_internal::ppo_apply_any(actor, critic, std, state->in_dim, mb_size, hp, state->grad, state->adam_m, state->adam_v,
                                      state->ctrl, parity, adam_step, (hipStream_t)stream);
}

int wl_ppo_wide_minibatch(const WlMlp* actor, const WlMlp* critic, float* std, const WlPpoBatch* batch, int32_t mb_start,
                          int32_t mb_size, const WlPpoParams* hp, const WlPpoWideState* state, int32_t parity, int32_t adam_step,
                          void* stream) {
    int rc = check_wide(actor, critic, state);
    if (rc == WL_OK) rc = check_batch(std, batch, mb_start, mb_size, state);
    if (rc != WL_OK) return rc;
    if (!hp || !state->adam_m || !state->adam_v || (parity != 0 && parity != 1) || adam_step < 1) return WL_EINVAL;
    rc = launch_wide_gradients(actor, critic, std, batch, mb_start, mb_size, hp, state, parity, (hipStream_t)stream);
    if (rc != WL_OK) return rc;
    return wl_internal::ppo_apply_any(actor, critic, std, state->in_dim, mb_size, hp, state->grad, state->adam_m, state->adam_v,
                                      state->ctrl, parity, adam_step, (hipStream_t)stream);
}

}  // extern "C"

// wl_ppo_internal.h -- pieces of wl_ppo.hip that wl_ppo_wide.hip drives (library-internal C++ linkage, not part of the C ABI)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"

namespace wl_internal {

// Everything of a wide step that depends on the weights only, as one launch: the tail's operand tables (`operands`,
// WL_PPO_OPERAND_FLOATS) and the first-layer weights of both nets as bf16 planes [128][dp] (as mlp_weight_planes).
int ppo_prepare_wide(const WlMlp* actor, const WlMlp* critic, const float* std, float* operands, int dp, uint16_t* w_hi, uint16_t* w_lo,
                     hipStream_t stream);

// The gradient kernel of wl_ppo.hip with the first layer cut off: reads the activated layer-1 outputs `h1`
// ([position in the minibatch][actor 64 | critic 64]) and the tables of ppo_prepare_wide, writes delta1^T as bf16 planes
// `dt_hi / dt_lo` (blocked [position / 64][128 units][position % 64]) and every other gradient (db1 included) + the three
// loss sums as per-block rows `partials` ([WL_PPO_BLOCKS][WL_PPO_PARTIAL_STRIDE], the drift agents' row layout with its
// first-layer weight slots left at zero).  Returns the number of rows written (the caller reduces them:
// ppo_wide_scatter_kernel) or a negative WL_E* code.
int ppo_tail_wide(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* bt, int mb_start, int mb_size,
                  const WlPpoParams* hp, float* partials, float* operands, const float* h1, uint16_t* dt_hi, uint16_t* dt_lo,
                  hipStream_t stream);

// entropy term + clipping + adaptive-KL rule + Adam on a flat gradient row of input width `in_dim` (statistics behind it)
int ppo_apply_any(const WlMlp* actor, const WlMlp* critic, float* std, int in_dim, int mb_size, const WlPpoParams* hp,
                  const float* grad, float* adam_m, float* adam_v, float* ctrl, int parity, int adam_step, hipStream_t stream);

// wl_ppo_wide.hip: layer-1 weights of both nets -> bf16 planes [128][dp] (dp = in_dim rounded up to 64, padding zero)
int mlp_weight_planes(const WlMlp* actor, const WlMlp* critic, int dp, uint16_t* w_hi, uint16_t* w_lo, hipStream_t stream);

// wl_ppo_wide.hip: out[s][r][u] = sum over the 64-wide K chunks of split s of x[r][k] . W[u][k] (both nets: 128 units), x = f32 rows
// ([n_rows][x_stride], in_dim valid features) split into bf16 planes in registers, W = planes from mlp_weight_planes.
// Two chunks per split (whatever n_rows: results do not depend on the batch a row arrives in); returns the number of
// splits = ceil(dp / 128) (partial sums [splits][n_rows][128]; WL_EINVAL if that exceeds max_splits) or a negative WL_E* code.
// whole_k: ONE split (the whole width per block; also independent of the batch) -- for long batches, where the six partial
// sums per row of the default would be more traffic than the rows themselves.
int layer1_partials(const float* x, int64_t x_stride, int n_rows, int in_dim, int dp, const uint16_t* w_hi, const uint16_t* w_lo,
                    int max_splits, float* out, hipStream_t stream, bool whole_k = false);

}  // namespace wl_internal

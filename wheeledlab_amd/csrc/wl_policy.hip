// wl_policy.hip -- policy in the loop: the actor MLP on the f32 matrix pipe (wl_mlp.h) fused with the drift env.step()
// (wl_drift_env.h) into one persistent rollout launch, plus the standalone MLP forward used for the critic.
//
// Replaces the collection loop of the reference's runner (modified_rsl_rl_runner.py:70-80) for the drift agents
// (rsl_rl_ppo_cfg.py:6,12-17: 128 steps per env, [64, 64] ELU actor).  Per step and wavefront (16 envs in quad form):
//   observation (registers, replicated on the env's quad) --4 ds_bpermute--> layer-1 B operands
//   -> 101 v_mfma_f32_16x16x4_f32 (weights resident in registers) -> action means on lanes 0..15
//   --2 ds_bpermute--> back onto the quads -> Gaussian sample (Philox) -> drift_env_step -> next observation.
// Nothing but the transition rows (obs, action, mu, log-prob, reward, flags) touches memory inside the loop.
#include "wl_drift_env.h"
#include "wl_mlp.h"

namespace {

constexpr float kLog2Pi = 1.8378770664093453f;

WL_DEV float lane_pull(int src_lane, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// y[n][out_dim] = mlp(x[n][in_dim]); a wavefront walks 16-row tiles with the weights resident in registers
template <int ACT>
__global__ void __launch_bounds__(kBlock) mlp_forward_kernel(const WlMlp net, const int n_rows, const float* __restrict__ x,
                                                             float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = (gridDim.x * kBlock) >> 6;
    const int n_tiles = (n_rows + 15) >> 4;
    if (wave >= n_tiles) return;   // wave-uniform
    MlpWeights W;
    mlp_load_weights(net, lane, W);
    const int m = lane & 15, g = lane >> 4;
    for (int tile = wave; tile < n_tiles; tile += n_waves) {
        const int row = tile * 16 + m;
        const bool valid = row < n_rows;
        float xs[kMlpInSteps];
#pragma unroll
        for (int s = 0; s < kMlpInSteps; ++s) {
            const int f = 4 * s + g;
            xs[s] = f == net.in_dim ? 1.f : (valid && f < net.in_dim) ? x[(int64_t)row * net.in_dim + f] : 0.f;
        }
        const f32x4 out = mlp_eval<ACT>(W, xs, lane);
        if (g == 0 && valid) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < net.out_dim) y[(int64_t)row * net.out_dim + o] = out[o];
        }
    }
}

// K steps of { actor -> sample -> env.step } in one launch (quad form).  Lanes whose env index is past n_envs mirror
// the last env (the matrix pipe and the cross-lane pulls want whole wavefronts) and never store.
template <int ACT, class Ground, int QB = kBlock /* threads per block, see drift_step_kernel */>
__global__ void __launch_bounds__(kBlock) drift_policy_rollout_kernel(const WlDriftParams p_arg, const WlEnvBuffers b, const WlMlp actor,
                                                                      const float* __restrict__ action_std,
                                                                      const WlPolicyRollout io, const int n_steps,
                                                                      const uint64_t seed, const uint64_t step0,
                                                                      const Ground ground, const VehDerived vd_arg, const MetricSlots slots) {
    constexpr int LANES = 4, kEnvs = QB / LANES;
    WlDriftParams p = p_arg;
    VehDerived vd = vd_arg;
    pin_params_vgpr(p, vd);
    const int lane = threadIdx.x & 63;
    const int le = threadIdx.x / LANES, wid = threadIdx.x & 3;
    const int e_raw = blockIdx.x * kEnvs + le;
    if (b.metrics_slots > 1) clear_metric_slot(b, slots.next);
    if (e_raw - (lane >> 2) >= b.n_envs) return;   // the whole wavefront is past the end (wave-uniform)
    const bool valid = e_raw < b.n_envs;
    const int e = valid ? e_raw : b.n_envs - 1;
    const bool lead = valid && wid == 0;
    const MetricSink<LANES> ms{nullptr, metric_shard(b, slots.cur)};
    const Rows S = make_rows(b.state, b.stride);
    const int64_t n = b.n_envs;

    MlpWeights W;
    mlp_load_weights(actor, lane, W);
    const float std0 = action_std[0], std1 = action_std[1];
    const float logp_const = -(log_fast(std0) + log_fast(std1)) - kLog2Pi;

    EnvConst ec;
    DriftRows r;
    load_env_const(S, p.vehicle, vd, e, ec);
    load_rows<LANES>(S, b, p, e, wid, r);
    float o[16];   // this env's observation, replicated on its quad; o[14] = 1 feeds the layer-1 bias, o[15] pads
    {
        const float2* row = reinterpret_cast<const float2*>(io.obs + (int64_t)e * kObsDim);
#pragma unroll
        for (int k = 0; k < kObsDim / 2; ++k) {
            const float2 v = row[k];
            o[2 * k] = v.x;
            o[2 * k + 1] = v.y;
        }
    }
    o[14] = 1.f;
    o[15] = 0.f;
    // lane l = 16 g + nn of the matrix layout pulls feature 4 s + g of env nn from lane 4 nn + g of the quad layout
    const int pull_in = 4 * (lane & 15) + (lane >> 4);
    // lane of the quad layout pulls its env's outputs from lane (lane >> 2) of the matrix layout (lane group 0)
    const int pull_out = lane >> 2;

    for (int k = 0; k < n_steps; ++k) {
        const uint64_t step = step0 + (uint64_t)k;
        float xs[kMlpInSteps];
#pragma unroll
        for (int s = 0; s < kMlpInSteps; ++s) {
            xs[s] = lane_pull(pull_in, quad_pick(wid, o[4 * s], o[4 * s + 1], o[4 * s + 2], o[4 * s + 3]));
        }
        const f32x4 out = mlp_eval<ACT>(W, xs, lane);
        const float mu0 = lane_pull(pull_out, out[0]), mu1 = lane_pull(pull_out, out[1]);
        // a ~ N(mu, diag(std^2)) (rsl_rl ActorCritic.act); identical on the four lanes of the quad
        const uint32_t gid = (uint32_t)(b.env_offset + e);
        const F4 u = philox_uniform4(gid, step, WL_RS_POLICY, seed);
        float z0, z1;
        box_muller(u.x, u.y, z0, z1);
        const float2 a = make_float2(fmaf(std0, z0, mu0), fmaf(std1, z1, mu1));
        if (lead) {
            const int64_t at = (int64_t)k * n + e;
            reinterpret_cast<float2*>(io.actions)[at] = a;
            reinterpret_cast<float2*>(io.mu)[at] = make_float2(mu0, mu1);
            io.log_prob[at] = fmaf(-0.5f, fmaf(z0, z0, z1 * z1), logp_const);
        }
        WlStepOut so;
        so.obs = nullptr;   // the observation stays in `o`
        so.reward = io.reward + (int64_t)k * n;
        so.terminated = io.terminated + (int64_t)k * n;
        so.truncated = io.truncated + (int64_t)k * n;
        so.dones = io.dones ? io.dones + (int64_t)k * n : nullptr;
        const StepDraws pre = draw_step(p, b.ref_poses, gid, step, seed, wid);
        drift_env_step<LANES>(p, b, vd, ground, S, ec, r, a, nullptr, so, e, wid, lead, gid, seed, step, nullptr, ms, o, &pre);
        if (valid) store_obs_quad(io.obs + ((int64_t)(k + 1) * n + e) * kObsDim, wid, o);
    }
    if (valid) store_rows<LANES>(S, b, p, e, wid, lead, r);
}

}  // namespace

extern "C" {

int wl_mlp_forward(const WlMlp* net, int32_t n_rows, const float* x, float* y, void* stream) {
    int rc = check_mlp(net);
    if (rc != WL_OK) return rc;
    if (n_rows <= 0 || !x || !y) return WL_EINVAL;
    clear_error();
    const int n_tiles = (n_rows + 15) / 16;
    const int grid = min((n_tiles + 3) / 4, 2048);   // 4 wavefronts per block; each walks tiles with resident weights
    if (net->activation == WL_ACT_ELU)
        mlp_forward_kernel<WL_ACT_ELU><<<grid, kBlock, 0, (hipStream_t)stream>>>(*net, n_rows, x, y);
    else
        mlp_forward_kernel<WL_ACT_RELU><<<grid, kBlock, 0, (hipStream_t)stream>>>(*net, n_rows, x, y);
    return launch_status();
}

int wl_drift_rollout_policy(const WlDriftParams* p, const WlEnvBuffers* b, const WlMlp* actor, const float* action_std,
                            const WlPolicyRollout* io, int32_t n_steps, uint64_t seed, uint64_t step0, void* stream) {
    int rc = check_buffers(p, b);
    if (rc != WL_OK) return rc;
    rc = check_mlp(actor);
    if (rc != WL_OK) return rc;
    if (actor->in_dim != kObsDim || actor->out_dim != 2 || !action_std || n_steps < 0) return WL_EINVAL;
    if (!io || !io->obs || !io->actions || !io->mu || !io->log_prob || !io->reward || !io->terminated || !io->truncated)
        return WL_EINVAL;
    if (((uintptr_t)io->obs & 7u) || ((uintptr_t)io->actions & 7u) || ((uintptr_t)io->mu & 7u)) return WL_EALIGN;
    if (b->metrics_slots > 1 && n_steps % b->metrics_slots == 0 && n_steps > 0) return WL_EINVAL;   // ring slot aliasing
    clear_error();
    const VehDerived vd = derive_vehicle(p->vehicle, p->sim_dt, p->decimation);
#define WL_PR_ARGS *p, *b, *actor, action_std, *io, n_steps, seed, step0, FlatGround{}, vd, metric_slots(b, step0, (uint64_t)n_steps)
#define WL_PR_LAUNCH(ACT)                                                                                                       \
    if (b->n_envs <= 2048) drift_policy_rollout_kernel<ACT, FlatGround, 64><<<(lanes + 63) / 64, 64, 0, (hipStream_t)stream>>>(WL_PR_ARGS);   \
    else if (b->n_envs <= 8192) drift_policy_rollout_kernel<ACT, FlatGround, 128><<<(lanes + 127) / 128, 128, 0, (hipStream_t)stream>>>(WL_PR_ARGS); \
    else drift_policy_rollout_kernel<ACT, FlatGround><<<grid_for(lanes), kBlock, 0, (hipStream_t)stream>>>(WL_PR_ARGS)
    const int lanes = b->n_envs * 4;
    if (actor->activation == WL_ACT_ELU) {
        WL_PR_LAUNCH(WL_ACT_ELU);
    } else {
        WL_PR_LAUNCH(WL_ACT_RELU);
    }
#undef WL_PR_LAUNCH
#undef WL_PR_ARGS
    return launch_status();
}

}  // extern "C"

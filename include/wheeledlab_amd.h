/*
 * wheeledlab_amd.h -- C ABI of the MI355X (gfx950) env.step() hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own: its hot path is
 * `isaaclab.envs.ManagerBasedRLEnv.step()` (registered at
 * source/wheeledlab_tasks/wheeledlab_tasks/__init__.py:14-23) calling *up* into the reference's action terms
 * and mdp functions and *down* into PhysX.  Each entry point below names the reference code it replaces.
 *
 * Conventions
 *   - plain C, no torch / HIP types: device pointers are `void*`-compatible raw pointers, the stream is a
 *     `void*` holding a hipStream_t (NULL = default stream).
 *   - the caller owns every buffer; the library allocates nothing and keeps no global state.
 *   - asynchronous on the given stream, re-entrant, no internal synchronisation.
 *   - return 0 (WL_OK) or a negative WL_E* code; never throws.
 *   - per-env state is structure-of-arrays: a float matrix `state[WL_S_COUNT][stride]`, row = field,
 *     column = env, `stride >= n_envs`, `stride % 64 == 0`, base 16-byte aligned.
 *   - quaternions are (w, x, y, z); velocities in `state` are world-frame (IsaacLab `root_state_w` order).
 */
#ifndef WHEELEDLAB_AMD_H
#define WHEELEDLAB_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WL_ABI_VERSION 23

enum WlStatus { WL_OK = 0, WL_EINVAL = -1, WL_ELAUNCH = -2, WL_EALIGN = -3, WL_ENODEV = -4 };

/* ---- rows of the SoA state matrix -------------------------------------------------------------------- */
enum WlStateField {
    /* root state, IsaacLab `root_state_w` order: link position, orientation, COM linear vel, angular vel */
    WL_S_PX = 0, WL_S_PY, WL_S_PZ,
    WL_S_QW, WL_S_QX, WL_S_QY, WL_S_QZ,
    WL_S_VX, WL_S_VY, WL_S_VZ,
    WL_S_WX, WL_S_WY, WL_S_WZ,
    /* wheel spin [rad/s], reference wheel order (rc_car_actions.py:62): back-left, back-right, front-left, front-right */
    WL_S_WHEEL_BL, WL_S_WHEEL_BR, WL_S_WHEEL_FL, WL_S_WHEEL_FR,
    /* steering joint (both front steer joints share target and gains -> one value), position [rad] and rate */
    WL_S_STEER_POS, WL_S_STEER_VEL,
    /* last raw action (ActionManager.action), 2 floats */
    WL_S_ACT0, WL_S_ACT1,
    /* interval-event timers (push_robots_hf / push_robots_lf time_left), seconds */
    WL_S_TIMER_HF, WL_S_TIMER_LF,
    /* per-env domain-randomised constants (startup events): wheel static / dynamic friction, rear throttle
       damping, total mass.  Read-only inside step. */
    WL_S_MU_S, WL_S_MU_D, WL_S_DAMP, WL_S_MASS,
    /* per-term episode reward sums (RewardManager._episode_sums), WL_MAX_REW_TERMS rows */
    WL_S_EPSUM0,
    /* goal command of the elevation task (UniformPose2dCommand): target in the yaw-aligned base frame (x, y), world
       target (x, y), heading, time left until resampling */
    WL_S_CMD_BX = WL_S_EPSUM0 + 8, WL_S_CMD_BY, WL_S_TGT_X, WL_S_TGT_Y, WL_S_TGT_H, WL_S_CMD_TIMER,
    WL_S_COUNT
};
#define WL_N_DYN 23          /* rows [0, WL_N_DYN) are read AND written every step */
#define WL_MAX_REW_TERMS 8

/* ---- drift task: reward term slots (DriftRewardsCfg order, mushr_drift_env_cfg.py:246-299) ------------ */
enum WlDriftRewTerm {
    WL_DR_SIDE_SLIP = 0, WL_DR_VEL, WL_DR_PROGRESS, WL_DR_TLGR, WL_DR_TURN_ENERGY, WL_DR_CROSS_TRACK,
    WL_DR_TERM_PENS, WL_DR_NTERMS
};

/* ---- metric accumulators ------------------------------------------------------------------------------------
 * One logical accumulator vector is float[WL_M_SHARDS][WL_M_COUNT]; its value is the SUM over the shards.  Wavefronts
 * add into shard (global wavefront index % WL_M_SHARDS): float atomics of one launch to ONE address are performed
 * back to back at the memory side (~35 ns each on MI355X), and with ~33 resets per step at 4096 envs a single
 * 16-float vector cost the step kernel 1.2 us of its 10 us; 32 shards make it ~1 atomic per address.
 * The caller zeroes the shards when it has consumed them (metrics_slots == 1) or lets the kernels run the ring. */
#define WL_M_SHARDS 32
enum WlMetric {
    WL_M_EPSUM0 = 0,                 /* [0,8): sum over reset envs of episode_sum[term]                   */
    WL_M_RESETS = 8,                 /* number of env resets                                              */
    WL_M_TIMEOUTS = 9,               /* Episode_Termination/time_out count                                */
    WL_M_TERM0 = 10,                 /* [10,14): counts of the task's `terminated` terms                  */
    WL_M_NONFINITE = 14,             /* envs whose state went non-finite (they are force-reset)           */
    WL_M_EPLEN = 15,                 /* sum of episode lengths (steps) over reset envs                    */
    WL_M_COUNT = 16
};

/* ---- vehicle + contact model (replaces PhysX for this path; designed) -------- */
typedef struct WlVehicleParams {
    float gravity;            /* 9.81 */
    float half_wheelbase_f;   /* CoM -> front axle [m]  (base_length 0.325 split; common/actions.py:18)       */
    float half_wheelbase_r;   /* CoM -> rear axle [m]                                                         */
    float half_track;         /* base_width / 2 = 0.1                                                         */
    float wheel_radius;       /* 0.05                                                                         */
    float wheel_z;            /* wheel-centre height in the root-link frame (root origin rests at ground)    */
    float cg_z;               /* CoM height in the root-link frame                                            */
    float gyr_x, gyr_y, gyr_z;/* radii of gyration: I = m * gyr^2 (mass is per-env)                           */
    float wheel_inertia;      /* spin inertia of one wheel [kg m^2]                                           */
    float wheel_damping;      /* bearing damping [N m s/rad]                                                  */
    float susp_k, susp_c;     /* vertical tyre+suspension spring / damper per wheel                           */
    float ground_mu_s, ground_mu_d;  /* terrain material, combined by "multiply" (mushr_drift_env_cfg.py:45-50) */
    float slip_peak;          /* normalised slip at peak friction                                             */
    float v_min;              /* low-speed regularisation of the slip denominator [m/s]                       */
    /* DC motor on throttle joints (hound.py:13-21,40-43): tau = damp*(w_tgt-w) clipped to the DC curve       */
    float motor_sat, motor_limit, motor_vel_limit;
    int32_t drive;            /* 0 = rear wheel drive (front passive, hound.py:44-51), 1 = 4WD               */
    /* implicit PD on the steer joints (hound.py:5-12)                                                        */
    float steer_kp, steer_kd, steer_effort, steer_vel_limit, steer_inertia;
    int32_t substeps;         /* integrator sub-steps per sim.dt: h = sim.dt / substeps                       */
    /* ABI 22: which integrator steps the force laws above (wl_vehicle.h, spec oracle/vehicle.py).
     * 0 = explicit (h <= 5 ms, tyre stiffness capped): what the drift / F1Tenth entry points take (sim.dt = 5 ms,
     *     mushr_drift_env_cfg.py:393-394) -- they refuse 1;
     * 1 = linearly implicit (Rosenbrock-W; stable and steady-state-exact up to h = 20 ms): what the elevation / visual /
     *     visual-depth entry points take -- ONE sub-step per sim.dt, the reference's own physics rate
     *     (mushr_elevation_env_cfg.py:461-462, mushr_visual_env_cfg.py:435-436) -- they refuse 0.                      */
    int32_t implicit;
    /* ABI 22: the normal force of a wheel is max(min(k pen, susp_fmax) - c v_n, 0): the SPRING's force is capped.  The model has no
     * chassis collision: a car that lands on its roof or tumbles meets the ground with its wheel spheres only, tens of centimetres
     * deep, where a linear penalty spring would hand it hundreds of newtons per wheel.  24 x the static wheel load (6.7 cm of
     * penetration: more than the wheel's radius) in the registered tasks: never reached in driving, nor by the 0.25 m spawn drop
     * of the elevation task (2.4 cm).  Must be > 0.                                                                          */
    float susp_fmax;
} WlVehicleParams;

/* ---- action term (AckermannAction.process_actions, ackermann_actions.py:119-133) ---------------------- */
typedef struct WlActionParams {
    float scale[2], offset[2];
    int32_t bounding;         /* 0 none, 1 clip(-1,1), 2 tanh                                                 */
    int32_t no_reverse;       /* clamp processed throttle >= 0                                                */
    int32_t clip_wrapper;     /* 1: apply the ClipAction wrapper (clip_action.py:27) before everything        */
    int32_t map;              /* 0 RCCarRWDAction, 1 RCCar4WDAction (rc_car_actions.py:12-29 / 36-64), 2 the base
                               * class AckermannAction: true Ackermann angles (ackermann_actions.py:150-201)   */
    float base_length, base_width, wheel_radius;
} WlActionParams;

typedef struct WlDriftParams {
    float sim_dt;             /* 0.005 (mushr_drift_env_cfg.py:393)                                           */
    int32_t decimation;       /* 4     (:394)                                                                 */
    int32_t max_episode_length; /* ceil(5 s / 0.02 s) = 250 (:396)                                            */
    WlActionParams action;
    WlVehicleParams vehicle;
    /* track geometry (:27-32) */
    float straight, r_in, r_out, r_line;
    /* reward weights in WlDriftRewTerm order and term params (:246-299); weight 0 => term skipped            */
    float weight[WL_MAX_REW_TERMS];
    float slip_min, slip_max, slip_min_vx;
    float speed_target, speed_offset;
    float tlgr_thresh;
    float ctd_offset, ctd_p;
    /* observation corruption (common/observations.py:27-50, enable at mushr_drift_env_cfg.py:399)            */
    int32_t enable_corruption;
    float noise_std[4];       /* pos, euler, lin vel, ang vel                                                 */
    /* reset (drifting/mdp/events.py:102-133)                                                                 */
    int32_t num_ref_points;
    float pos_noise, yaw_noise;
    /* interval pushes (mushr_drift_env_cfg.py:121-143)                                                        */
    int32_t enable_pushes;
    float hf_interval[2], hf_vel_x, hf_vel_y, hf_vel_yaw;
    float lf_interval[2], lf_vel_yaw;
    int32_t log_episode_sums; /* maintain WL_S_EPSUM rows + WL_M_EPSUM metrics                                */
} WlDriftParams;

/* ---- device buffers of one env batch ------------------------------------------------------------------ */
typedef struct WlEnvBuffers {
    float* state;             /* [WL_S_COUNT][stride]                                                         */
    int32_t* episode_len;     /* [n]  episode_length_buf                                                      */
    const float* ref_poses;   /* [3][32]: x, y, yaw(rad) of the pre-sampled reference poses (events.py:31)     */
    float* metrics;           /* [metrics_slots][WL_M_SHARDS][WL_M_COUNT] accumulators (slot step % slots)    */
    int64_t stride;
    int32_t n_envs;
    int32_t env_offset;       /* global id of env 0 of this shard (rank * n_envs): keys the RNG streams            */
    int32_t metrics_slots;    /* 1: one accumulator the caller zeroes; R > 1: ring of per-step slots, the step    */
                              /* kernel accumulates into slot (step % R) and clears slot ((step + 1) % R)         */
    int32_t lanes;            /* step-kernel form: 0 = choose by env count, 1 = lane per env (packed axles),       */
                              /* 2 = lane per env (scalar wheel loop, 5 waves / SIMD), 4 = quad per env            */
    int32_t flags;            /* WL_FLAG_* bits; 0 = every choice below is made from the batch size                  */
} WlEnvBuffers;

/* WlEnvBuffers.flags: force one of the instantiations the launchers otherwise pick from the batch size -- so that tests and
 * probes can run EVERY instantiation at any size (the large-batch forms at 1000 envs, the cache-allocating forms at 4 M). */
#define WL_FLAG_STREAM 1       /* streaming forms: rows / observation rows / outputs written with non-temporal (sc1 nt) stores; */
                               /* drift: the lane-per-env streaming instantiation (`lanes` must not be 4)                        */
#define WL_FLAG_NO_STREAM 2    /* cache-allocating stores whatever the batch size                                                */
#define WL_FLAG_SCAN_LDS 4     /* elevation height scan: the env's terrain patch staged in LDS (lane form / wl_elev_observe)     */
#define WL_FLAG_SCAN_GATHER 8  /* elevation height scan: per-ray gathers from the L2-resident field                              */
#define WL_FLAG_MASK 15

/* ---- outputs of one step -------------------------------------------------------------------------------- */
typedef struct WlStepOut {
    float* obs;               /* [n][obs_dim] row-major, what the policy consumes                            */
    float* reward;            /* [n]                                                                          */
    uint8_t* terminated;      /* [n]                                                                          */
    uint8_t* truncated;       /* [n]                                                                          */
    int64_t* dones;           /* [n] terminated | truncated as int64 (what RSL-RL's runner consumes,             */
                              /* modified_rsl_rl_runner.py:73,101); may be NULL                                   */
} WlStepOut;

int wl_version(void);
/* number of HIP devices visible, or WL_ENODEV */
int wl_device_count(void);
const char* wl_strerror(int code);

/*
 * Fused drift-task env.step(): replaces, for all n envs in ONE kernel launch, what IsaacLab's
 * ManagerBasedRLEnv.step() does with the reference's plugins:
 *   ClipAction (clip_action.py:27) -> AckermannAction.process_actions (ackermann_actions.py:119-133)
 *   -> decimation x [RCCarRWDAction map (rc_car_actions.py:12-29) -> actuators (hound.py:4-52) -> rigid body +
 *      4 tyre contacts (replaces PhysX)] -> time_out + cart_off_track (mushr_drift_env_cfg.py:343-362)
 *   -> 7 reward terms x weight x step_dt (:160-299) -> in-kernel reset along the track (drifting/mdp/events.py:102-133)
 *   -> interval pushes (:121-143) -> 14-dim noisy observation (common/observations.py:24-54).
 * `actions` is [n][2] row-major (device).  `noise` is NULL (in-kernel Philox4x32, 7 rounds (csrc/wl_rng.h), keyed by (seed, env, step))
 * or a device float[12][stride] of standard normals used instead (parity mode).
 * `step` is IsaacLab's common_step_counter BEFORE this step.
 */
int wl_drift_step(const WlDriftParams* p, const WlEnvBuffers* b, const float* actions, const float* noise,
                  const WlStepOut* out, uint64_t seed, uint64_t step, void* stream);

/*
 * K consecutive fused steps with pre-staged actions [K][n][2]; outputs of step k are written at
 * obs + k*obs_step_stride (floats), reward + k*vec_step_stride, terminated/truncated + k*vec_step_stride
 * (strides 0 = overwrite).  This is the rollout inner loop of the reference's runner
 * (modified_rsl_rl_runner.py:70-73) minus the policy.
 */
int wl_drift_rollout(const WlDriftParams* p, const WlEnvBuffers* b, const float* actions, const WlStepOut* out,
                     int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps, uint64_t seed,
                     uint64_t step0, void* stream);

/*
 * Same contract as wl_drift_rollout, executed as ONE persistent launch (quad form): each env's rows stay in registers
 * across the K steps; per step only the action is read and obs / reward / flags are written.  Results are identical to
 * K calls of wl_drift_step.  Episode metrics of all K steps accumulate into ring slot (step0 % R) and slot
 * ((step0 + K) % R) is cleared for the next launch (K % R must not be 0 when R > 1).  For pre-staged action sequences
 * (open-loop evaluation, sampling-based MPC); a policy in the loop needs wl_drift_step or wl_drift_rollout_policy.
 */
int wl_drift_rollout_persistent(const WlDriftParams* p, const WlEnvBuffers* b, const float* actions,
                                const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps,
                                uint64_t seed, uint64_t step0, void* stream);

/* ---- policy in the loop (the rollout loop of modified_rsl_rl_runner.py:70-80) ------------- */
enum WlActivation { WL_ACT_RELU = 0, WL_ACT_ELU = 1 };

/*
 * One MLP of RSL-RL's ActorCritic as the drift agents configure it (rsl_rl_ppo_cfg.py:12-17: hidden dims [64, 64],
 * activation "elu"): y = W3 act(W2 act(W1 x + b1) + b2) + b3.  Device pointers, fp32, torch nn.Linear layout
 * (weight [out][in] row-major).
 */
typedef struct WlMlp {
    const float* w1; const float* b1;   /* [64][in_dim], [64] */
    const float* w2; const float* b2;   /* [64][64], [64] */
    const float* w3; const float* b3;   /* [out_dim][64], [out_dim] */
    int32_t in_dim;      /* 1..15 (drift observation: 14); wl_actor_critic_act: any (elevation 689, visual 3208) */
    int32_t out_dim;     /* 1..4  (actor: 2 action means; critic: 1 value) */
    int32_t hidden;      /* must be 64 */
    int32_t activation;  /* WlActivation */
} WlMlp;

/*
 * y[n_rows][out_dim] = mlp(x[n_rows][in_dim]) on the f32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products and
 * sums, i.e. a plain fp32 MLP up to summation order).  Used for the critic over a collected rollout and as the parity
 * entry point of the in-loop actor.
 */
int wl_mlp_forward(const WlMlp* net, int32_t n_rows, const float* x, float* y, void* stream);

/* Rows of RSL-RL's RolloutStorage that the collection loop fills, K = n_steps, all device pointers, [step][env] major. */
typedef struct WlPolicyRollout {
    float* obs;            /* [K + 1][n][14]: row 0 = the current observation (INPUT), rows 1..K are written */
    float* actions;        /* [K][n][2] sampled actions a = mu + sigma * N(0, 1), before ClipAction */
    float* mu;             /* [K][n][2] actor output */
    float* log_prob;       /* [K][n] sum over the 2 action dims of log N(a; mu, sigma) */
    float* reward;         /* [K][n] */
    uint8_t* terminated;   /* [K][n] */
    uint8_t* truncated;    /* [K][n] ("time_outs": the caller bootstraps rewards with gamma * V on these) */
    int64_t* dones;        /* [K][n] terminated | truncated, or NULL */
} WlPolicyRollout;

/*
 * The rollout loop of the reference's runner (modified_rsl_rl_runner.py:70-80: `actions = alg.act(obs)`;
 * `obs, rewards, dones, infos = env.step(actions)`) for K steps as ONE persistent launch: per step the actor MLP runs on
 * the matrix pipe on the observation the env step just produced (still in registers), the action is sampled
 * (Philox stream 7, keyed like every other draw by (seed, global env id, step)), the env is stepped exactly as
 * wl_drift_step does, and the transition is written to the storage rows.  `action_std` is a device pointer to the 2
 * standard deviations (RSL-RL's `ActorCritic.std`).  The critic is not on the loop's dependency chain: evaluate it
 * over obs[0..K] afterwards with wl_mlp_forward.  Episode metrics as in wl_drift_rollout_persistent.
 */
int wl_drift_rollout_policy(const WlDriftParams* p, const WlEnvBuffers* b, const WlMlp* actor, const float* action_std,
                            const WlPolicyRollout* io, int32_t n_steps, uint64_t seed, uint64_t step0, void* stream);

/*
 * One policy step on WIDE observations (the elevation / visual agents: rsl_rl ActorCritic.act + evaluate of
 * modified_rsl_rl_runner.py:70-76 on 689 / 3208 features) in ONE launch: mu = actor(obs), a = mu + std * N(0, 1),
 * log_prob, value = critic(obs).  obs is [n_rows][obs_stride] (obs_stride >= in_dim floats, rows 4-byte aligned); the
 * first layers run as skinny GEMMs on the f32 matrix pipe (exact fp32 products and sums), the draw is keyed by
 * (seed, env_offset + row, step) on the policy stream -- the draw of wl_drift_rollout_policy -- or skipped
 * (deterministic != 0: a = mu, the play policy).  actions / mu are [n_rows][2] (8-byte aligned), log_prob / values [n_rows].
 * `nets`: 3 = both, 1 = actor only (values may be NULL: the play policy), 2 = critic only (std / actions / mu / log_prob may
 * be NULL: values of stored observations); the results do not depend on how the two halves are launched.
 */
int wl_actor_critic_act(const WlMlp* actor, const WlMlp* critic, const float* std, int32_t n_rows, const float* obs,
                        int64_t obs_stride, float* actions, float* mu, float* log_prob, float* values, int32_t env_offset,
                        uint64_t seed, uint64_t step, int32_t deterministic, int32_t nets, void* stream);

/* The same policy step with the first layer on the bf16 matrix pipe: the observation rows are split into two bf16 planes
 * in registers (x = hi + lo, 16 mantissa bits; products hi.hi + lo.hi + hi.lo, f32 accumulation -- the arithmetic of the
 * wide PPO step below, so a stored log-prob and the learner's first evaluation of it agree), the layer-1 weights of both
 * nets come as planes through LDS and feed 128 rows per fetch.  Two launches.  Results differ from wl_actor_critic_act by
 * the split's rounding (~2^-17 relative in the layer-1 pre-activations). */
typedef struct WlActScratch {        /* caller-owned device memory */
    uint16_t *w_hi, *w_lo;           /* [128][dp] layer-1 weights of actor (units 0..63) and critic (64..127) as bf16 planes */
    float* partials;                 /* [splits][rows_capacity][128] split-K partial sums of layer 1 (128 features per split) */
    int32_t dp;                      /* in_dim rounded up to 64 */
    int32_t splits;                  /* >= ceil(dp / 128) (form 2: >= 1) */
    int32_t rows_capacity;
    int32_t reserved;                /* form.  0: one launch (feature shares folded through LDS); 1: two launches (split-K partial sums,
                                        128 features each); 2: two launches, ONE sum over the whole width per row and unit -- for long
                                        batches (the values of a whole rollout): partials is [rows_capacity][128] */
} WlActScratch;
/* (re)build the weight planes: once after every change of the layer-1 weights */
int wl_actor_critic_planes(const WlMlp* actor, const WlMlp* critic, const WlActScratch* scratch, void* stream);
int wl_actor_critic_act_planes(const WlMlp* actor, const WlMlp* critic, const float* std, int32_t n_rows, const float* obs,
                               int64_t obs_stride, float* actions, float* mu, float* log_prob, float* values, int32_t env_offset,
                               uint64_t seed, uint64_t step, int32_t deterministic, int32_t nets, const WlActScratch* scratch,
                               void* stream);

/* ---- PPO learner step of the drift agents ("on-device PPO for the 64-64 MLP") ------------
 * One minibatch step of rsl_rl's PPO.update (modified_rsl_rl_runner.py:104-109; rsl_rl_ppo_cfg.py:18-31) for the
 * 14-64-64-2 actor / 14-64-64-1 critic pair: forward, clipped surrogate + clipped value loss + entropy bonus, backward,
 * gradient-norm clipping, the adaptive-KL learning-rate rule and Adam, entirely on the device.
 * Flat parameter / gradient order (WL_PPO_NUM_PARAMS floats) = torch named_parameters() of the ActorCritic:
 * std[2], actor w1[64][14] b1[64] w2[64][64] b2[64] w3[2][64] b3[2], critic w1 b1 w2 b2 w3[1][64] b3[1]. */
#define WL_PPO_NUM_PARAMS 10437
#define WL_PPO_PARTIAL_STRIDE 10440   /* + value-loss, surrogate and KL sums of the minibatch */
#define WL_PPO_BLOCKS 256             /* rows of WlPpoState.partials */
#define WL_PPO_OPERAND_FLOATS 23296   /* both nets laid out in MFMA operand order (rebuilt by every step) */
/* WlPpoState.ctrl (16 floats, zero-initialised by the caller once): */
#define WL_PPO_CTRL_LR 0      /* [2] learning rate, ping-pong by `parity` (the caller seeds BOTH with the initial lr) */
#define WL_PPO_CTRL_NORM2 2   /* [2] squared gradient norm accumulator, ping-pong */
#define WL_PPO_CTRL_STATS 4   /* [3] running sums of mean value loss / surrogate / KL over the calls (caller zeroes) */
#define WL_PPO_CTRL_STD 8     /* [2] the action std the gradients were taken at (written by the gradient launches, read by apply) */

typedef struct WlPpoBatch {          /* a flattened rollout, B = K * n samples; all device pointers */
    const float* obs;                /* [B][14] */
    const float* actions;            /* [B][2] */
    const float* mu_old;             /* [B][2] */
    const float* logp_old;           /* [B] */
    const float* adv;                /* [B] advantages as the loss uses them (normalised) */
    const float* returns;            /* [B] */
    const float* values_old;         /* [B] */
    const int32_t* perm;             /* [>= mb_start + mb_size] sample order of this epoch */
    const float* sigma_old;          /* [2] action std at collection time */
} WlPpoBatch;

typedef struct WlPpoParams {
    float clip, value_loss_coef, entropy_coef, desired_kl, max_grad_norm;
    float beta1, beta2, eps;         /* Adam (0.9, 0.999, 1e-8) */
    float lr_min, lr_max;            /* bounds of the adaptive rule (1e-5, 1e-2) */
    int32_t use_clipped_value_loss;
    int32_t adaptive;                /* schedule == "adaptive" && desired_kl set */
} WlPpoParams;

typedef struct WlPpoState {          /* caller-owned device scratch */
    float* partials;                 /* [WL_PPO_BLOCKS][WL_PPO_PARTIAL_STRIDE] */
    float* grad;                     /* [WL_PPO_PARTIAL_STRIDE] */
    float* adam_m;                   /* [WL_PPO_NUM_PARAMS] */
    float* adam_v;                   /* [WL_PPO_NUM_PARAMS] */
    float* ctrl;                     /* [16], see WL_PPO_CTRL_* */
    float* operands;                 /* [WL_PPO_OPERAND_FLOATS], 16-byte aligned */
} WlPpoState;

/* Generalised advantage estimation exactly as rsl_rl's RolloutStorage.compute_returns: rewards / dones [K][n], values
 * [K + 1][n] (row K = bootstrap value of the last observation); writes returns and (un-normalised) advantages [K][n]. */
int wl_gae(int32_t n_steps, int32_t n_envs, const float* rewards, const float* values, const int64_t* dones, float gamma,
           float lam, float* returns, float* advantages, void* stream);

/* The runner's per-rollout bookkeeping in one launch (modified_rsl_rl_runner.py:74-75 "NaN in actions", :88-98 episode returns /
 * lengths, and rsl_rl PPO.process_env_step's time-out bootstrap): rewards / dones / time_outs [K][n], values [K + 1][n], actions
 * [K][n][2].  Writes the return and the length of every episode that ends at transition (t, e) to ep_ret / ep_len [K][n] (other
 * entries untouched), updates the carries of the running episodes ([n], in / out), then adds gamma * values[t] to the rewards
 * of timed-out transitions in place.  stats [3][ceil(n / 256)]: per-block sums of the RAW rewards, of the non-finite action
 * components and of the finished episodes (sum over the blocks on the caller's side: a fixed order). */
int wl_rollout_bookkeeping(int32_t n_steps, int32_t n_envs, float* rewards, const float* values, const int64_t* dones,
                           const uint8_t* time_outs, const float* actions, float gamma, float* carry_ret, float* carry_len,
                           float* ep_ret, float* ep_len, float* stats, void* stream);

/* gradients only (parity entry point): state->grad = d loss / d params (entropy term and clipping NOT applied) + the
 * three sums; accumulates the squared norm into ctrl[WL_PPO_CTRL_NORM2 + parity]. */
int wl_ppo_gradients(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* batch, int32_t mb_start,
                     int32_t mb_size, const WlPpoParams* hp, const WlPpoState* state, int32_t parity, void* stream);

/* the whole step: parameters (`actor`, `critic`, `std`) are updated in place.  `parity` alternates 0 / 1 between
 * consecutive calls; `adam_step` counts the calls from 1. */
int wl_ppo_minibatch(const WlMlp* actor, const WlMlp* critic, float* std, const WlPpoBatch* batch, int32_t mb_start,
                     int32_t mb_size, const WlPpoParams* hp, const WlPpoState* state, int32_t parity, int32_t adam_step,
                     void* stream);

/* The second half of wl_ppo_minibatch on a gradient row the caller has already reduced: entropy term, norm clipping
 * (with ctrl[WL_PPO_CTRL_NORM2 + parity] = squared norm of state->grad's parameter part), adaptive-KL learning rate, Adam.
 * Data-parallel learner: every rank calls wl_ppo_gradients on its shard of the minibatch, the ranks average state->grad
 * (all WL_PPO_PARTIAL_STRIDE floats: parameter gradients are means, the three statistics are sums over `mb_size` samples)
 * with ONE all-reduce, set the squared norm, and call this -- identical inputs on every rank, hence identical parameters
 * and learning rates without a broadcast. */
int wl_ppo_apply(const WlMlp* actor, const WlMlp* critic, float* std, int32_t mb_size, const WlPpoParams* hp,
                 const WlPpoState* state, int32_t parity, int32_t adam_step, void* stream);

/* ---- PPO learner step of the WIDE agents (SURVEY section 8(f) rank 3 remainder): D-64-64-2 actor / D-64-64-1 critic with
 * D = 689 (elevation, config/agents/mushr/rsl_rl_ppo_cfg.py of that task) or 3208 (visual); any D >= 64 works.
 * Same algorithm, same flat parameter order (with 14 -> D) and the same ctrl slots as the drift agents' step; what
 * differs is where the first layer runs.  With B = 131 072 rows of 689 floats a minibatch is 361 MB, so the layer-1
 * products are streaming contractions, done on the bf16 matrix pipe with every f32 operand split into two bf16 planes
 * (x = hi + lo, 16 mantissa bits; three products hi.hi + lo.hi + hi.lo, f32 accumulation):
 *   wl_ppo_wide_stage      once per update: rows of `obs` in the order of `perm` -> the planes of X^T (the dW1 operand)
 *   wl_ppo_wide_gradients  per minibatch: H1 = act(X W1^T + b1) for both nets (one contraction, 128 units), the rest of
 *                          the forward / loss / backward pass as in wl_ppo_gradients (same kernel, first layer cut off),
 *                          delta1 out as bf16 planes, dW1 = delta1^T X as a split-K contraction over the samples, and the
 *                          reduction of all partial sums into state->grad
 *   wl_ppo_wide_apply      clipping + adaptive-KL rule + Adam on state->grad (after an all-reduce when data-parallel)
 *   wl_ppo_wide_minibatch  = gradients + apply.
 * Dp = D rounded up to 64.  Minibatch starts / sizes must be multiples of 64 (rsl_rl's sizes are). */
typedef struct WlPpoWideState {      /* caller-owned device memory; the caller zero-fills ctrl, adam_m, adam_v once */
    uint16_t *xt_hi, *xt_lo;         /* [capacity / 64][dp][64]  the staged observations TRANSPOSED, bf16 planes, blocked by 64 rows */
    uint16_t *w_hi, *w_lo;           /* [128][dp]       layer-1 weights of both nets (rebuilt by every step) */
    float* h1;                       /* [mb_capacity][128]  activated layer-1 outputs of the minibatch */
    uint16_t *dt_hi, *dt_lo;         /* [mb_capacity / 64][128][64]  delta1^T planes, blocked likewise */
    float* dw_partials;              /* [splits][dp][128]   split-K partial sums of dW1^T */
    float* partials;                 /* [WL_PPO_BLOCKS][WL_PPO_PARTIAL_STRIDE]  per-block sums of the narrow part */
    float* narrow;                   /* [WL_PPO_PARTIAL_STRIDE]  reserved (the rows are reduced straight into `grad`) */
    float* grad;                     /* [wl_ppo_wide_num_params(D) + 3]  flat gradient + value-loss / surrogate / KL sums */
    float *adam_m, *adam_v;          /* [wl_ppo_wide_num_params(D)] */
    float* ctrl;                     /* [16], WL_PPO_CTRL_* */
    float* operands;                 /* [WL_PPO_OPERAND_FLOATS], 16-byte aligned */
    int32_t in_dim, dp;              /* D and D rounded up to 64 */
    int32_t capacity, mb_capacity;   /* rows staged per update (multiple of 64) / largest minibatch (multiple of 64) */
    int32_t splits;                  /* rows of dw_partials: the dW1 contraction is split over at most this many blocks per 128 features */
} WlPpoWideState;

/* number of parameters of the D-64-64-2 / D-64-64-1 pair incl. the 2 std entries (10437 for D = 14) */
int32_t wl_ppo_wide_num_params(int32_t in_dim);

/* rows [0, n_rows) of the staged planes <- obs[perm[k]] ([*][D] f32, row stride D).  n_rows <= capacity, multiple of 64. */
int wl_ppo_wide_stage(const float* obs, const int32_t* perm, int32_t n_rows, const WlPpoWideState* state, void* stream);

/* All fields of the batch are gathered through `perm` as in wl_ppo_gradients (the forward contraction splits the f32
 * observation rows in registers); staged row k must be obs[perm[k]] (the dW1 contraction reads the staged X^T).  state->grad = d loss / d params (no entropy term, no clipping) + the three sums;
 * ctrl[WL_PPO_CTRL_NORM2 + parity] accumulates its squared norm. */
int wl_ppo_wide_gradients(const WlMlp* actor, const WlMlp* critic, const float* std, const WlPpoBatch* batch, int32_t mb_start,
                          int32_t mb_size, const WlPpoParams* hp, const WlPpoWideState* state, int32_t parity, void* stream);
int wl_ppo_wide_apply(const WlMlp* actor, const WlMlp* critic, float* std, int32_t mb_size, const WlPpoParams* hp,
                      const WlPpoWideState* state, int32_t parity, int32_t adam_step, void* stream);
int wl_ppo_wide_minibatch(const WlMlp* actor, const WlMlp* critic, float* std, const WlPpoBatch* batch, int32_t mb_start,
                          int32_t mb_size, const WlPpoParams* hp, const WlPpoWideState* state, int32_t parity, int32_t adam_step,
                          void* stream);

/*
 * Drift mdp terms only, on caller-supplied state tensors (the parity entry point: "outputs match the reference
 * mdp terms on identical state tensors").  All inputs are SoA float[k][stride] device arrays:
 *   pos[3], quat[4], lin_vel_b[3], ang_vel_b[3], ang_vel_w[3], steer_pos[2], last_action[2].
 * Outputs: terms[WL_DR_NTERMS][stride] UNWEIGHTED term values (side_slip .. is_terminated),
 *          reward[n] = sum_i weight_i * term_i * step_dt, terminated[n] (cart_off_track), obs[n][14] (no noise).
 * Replaces mushr_drift_env_cfg.py:160-240,343-348; wheeledlab/envs/mdp/observations.py:9-12.
 */
int wl_drift_mdp(const WlDriftParams* p, int32_t n, int64_t stride, const float* pos, const float* quat,
                 const float* lin_vel_b, const float* ang_vel_b, const float* ang_vel_w, const float* steer_pos,
                 const float* last_action, const uint8_t* timed_out, float* terms, float* reward,
                 uint8_t* terminated, float* obs, void* stream);

/*
 * Action term only: ClipAction + process_actions + the RWD / 4WD joint-target map (parity entry point).
 * actions [n][2] -> processed [n][2] (v m/s, delta rad), steer_target [n][2] (= tan delta, both joints),
 * wheel_target [n][4] rad/s in order bl, br, fl, fr (RWD writes fl = fr = 0).
 * Replaces ackermann_actions.py:119-145, rc_car_actions.py:12-29,36-64.
 */
int wl_action_map(const WlActionParams* a, int32_t n, const float* actions, float* processed, float* steer_target,
                  float* wheel_target, void* stream);

/*
 * Reset every env in `mask` (uint8[n], NULL = all) exactly as the in-kernel reset does (events.py:102-133),
 * also re-arming push timers and zeroing episode_len / last action / episode sums.  Used by env.reset().
 */
int wl_drift_reset(const WlDriftParams* p, const WlEnvBuffers* b, const uint8_t* mask, uint64_t seed,
                   uint64_t step, void* stream);

/* Observation only (env.reset() / get_observations()): 14-dim policy obs from the current state. */
int wl_drift_observe(const WlDriftParams* p, const WlEnvBuffers* b, const float* noise, float* obs, uint64_t seed,
                     uint64_t step, void* stream);

/* ======================================================================================================== */
/* Elevation task (wheeledlab_tasks/elevation/mushr_elevation_env_cfg.py)                                   */
/* ======================================================================================================== */

/* terrain: regular-grid heightfield, height[iy][ix] row-major, world x = x0 + ix*cell. Replaces the terrain mesh
 * `Terrains/huge_compact.usd` (:95-108) for BOTH wheel contact and the ray-caster (:132-142). Outside the grid the
 * ground is the extra plane at z = outside_z (:120-128) and height-scan rays miss.
 * ABI 21: the grid holds 16-bit height CODES, z = code * z_scale (one fp32 multiply per grid point on the contact and depth paths;
 * the height scan blends four codes and scales once: the same value for a power-of-two scale, else within 2e-5 m of it) --
 * IsaacLab's own height-field terrains are int16 x vertical_scale (isaaclab.terrains.height_field).  Half the bytes of the fp32
 * grid of ABI <= 20: the 800 x 800 bench terrain is 1.28 MB instead of 2.56 MB per XCD's L2. */
typedef struct WlHeightField {
    const int16_t* height;
    int32_t nx, ny;
    float x0, y0, cell, outside_z;
    float z_scale;            /* metres per code, > 0 and finite (terrain.py default 2^-13 m = 0.122 mm: +-4 m)  */
    const uint32_t* pair;     /* ABI 23: the ROW-PAIR table [ny][nx], pair[j][i] = code[j][i] (low half) | code[j + 1][i] << 16 (the
                                 last row paired with itself): a cell's four corners are ONE 8-byte gather, pair[j][i .. i + 1].
                                 Filled by wl_heightfield_pairs; REQUIRED by the elevation entry points (their height scan was
                                 bound by the texture unit's address rate: two 4-byte gathers per ray -> one 8-byte gather, fused
                                 step at 4096 envs 20.5 -> 17.5 us); the visual / depth entry points do not read it (NULL is fine) */
} WlHeightField;
/* fills pair_out[ny][nx] (device memory, 4-byte aligned, nx * ny * 4 bytes) from hf->height; hf->pair is ignored.  Call again
 * whenever the codes change. */
int wl_heightfield_pairs(const WlHeightField* hf, uint32_t* pair_out, void* stream);

enum WlElevRewTerm { WL_ER_GOAL_PROGRESS = 0, WL_ER_HIGHER_ELEVATION, WL_ER_FALLING, WL_ER_STUCK_PENALTY, WL_ER_NTERMS };
/* `terminated` terms, counted in metrics[WL_M_TERM0 + k] */
enum WlElevTermTerm { WL_ET_BELOW_MIN_HEIGHT = 0, WL_ET_STUCK, WL_ET_ROLLOVER, WL_ET_AT_GOAL, WL_ET_NTERMS };

#define WL_ELEV_SCAN_N 26                      /* GridPatternCfg(size 2.5, resolution 0.1) -> 26 x 26 rays (:139)   */
#define WL_ELEV_OBS_DIM (13 + WL_ELEV_SCAN_N * WL_ELEV_SCAN_N)   /* 689 (:61-86)                                  */

typedef struct WlElevParams {
    float sim_dt;             /* 0.01 (:461)                                                                  */
    int32_t decimation;       /* 10   (:462)                                                                  */
    int32_t max_episode_length; /* ceil(20 s / 0.1 s) = 200 (:465)                                            */
    WlActionParams action;    /* Mushr4WDActionCfg (common/actions.py:29-47)                                  */
    WlVehicleParams vehicle;  /* MUSHR_SUS_CFG 4WD, motor limit 0.25 (hound.py:13-21)                         */
    float weight[WL_MAX_REW_TERMS];   /* WlElevRewTerm order: 200, 5000, 0, -200 (:286-305)                   */
    float min_height;         /* root_height_below_minimum 0.15 (:356-359)                                    */
    float stuck_min_vel, stuck_wheel_spin, stuck_vel_cap;   /* 0.02, 5.0 (:360-366), forward_vel cap 1.2 (:157)  */
    float upright_cos;        /* cos(60 deg): rollover when R33 < this (:217-222, 368-371)                     */
    float goal_dist;          /* 0.5 (:373-376)                                                               */
    float fall_vel;           /* 0.10 (:251-254)                                                              */
    float elev_z0, elev_min, elev_min_vel;   /* 0.19, 0.1, 0.1 (:166-173)                                      */
    float progress_offset;    /* 5 (:249)                                                                     */
    float reset_xy, reset_yaw, reset_vel[2], reset_z, spawn_clearance;   /* :409-419, :147-149                 */
    float cmd_xy, cmd_heading, cmd_resample_s;                           /* :425-435                           */
    float scan_size, scan_res, scan_offset, obs_clip;                    /* :74-82, :139                       */
    int32_t log_episode_sums;
} WlElevParams;

/*
 * Fused elevation env.step(): action term (RCCar4WDAction, rc_car_actions.py:36-64) -> decimation x substeps rigid
 * body + 4 tyre contacts on the heightfield -> terminations (:349-376) -> rewards (:286-305) -> reset
 * (isaaclab reset_root_state_uniform, :409-419) -> goal command update (:425-435) in ONE launch (lane = env), then
 * a second launch (block = env) that assembles the 689-dim observation incl. the 26 x 26 height map (:44-48,61-86).
 * `b->ref_poses` is unused (may be NULL).  obs is [n][WL_ELEV_OBS_DIM].
 */
int wl_elev_step(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* actions,
                 const WlStepOut* out, uint64_t seed, uint64_t step, void* stream);
int wl_elev_rollout(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* actions,
                    const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps,
                    uint64_t seed, uint64_t step0, void* stream);
int wl_elev_reset(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const uint8_t* mask,
                  uint64_t seed, uint64_t step, void* stream);
/* observation only (ObservationManager.compute over ElevationObsCfg :57-88) */
int wl_elev_observe(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, float* obs, void* stream);
/*
 * Elevation mdp terms on caller-supplied state tensors (parity entry point). SoA inputs float[k][stride]:
 * pos[3], quat[4], lin_vel_b[3], lin_vel_w[3], wheel_vel[4], command[2]; optional ray inputs sensor_z[n],
 * hit_z[n_rays][stride] (NULL to skip).  Outputs: terms[4][stride] unweighted (goal_progress_rate,
 * higher_elevation, is_falling_penalty, is_terminated(stuck)), flags[4][stride] bytes (below_min_height, stuck,
 * rollover, at_goal), goal_rel[2][stride], height_map[n_rays][stride] (unclipped world_height_map).
 * Replaces mushr_elevation_env_cfg.py:44-55,155-173,217-222,239-254,268-273,339-347.
 */
int wl_elev_mdp(const WlElevParams* p, int32_t n, int64_t stride, const float* pos, const float* quat,
                const float* lin_vel_b, const float* lin_vel_w, const float* wheel_vel, const float* command,
                const uint8_t* timed_out, int32_t n_rays, const float* sensor_z, const float* hit_z, float* terms,
                uint8_t* flags, float* goal_rel, float* height_map, void* stream);

/* wl_elev_rollout as ONE launch for open-loop rollouts (pre-staged actions): the envs' rows stay in registers across the
 * n_steps steps and the height scan of step k runs while step k + 1 is integrated (the actions do not depend on the
 * observations).  Same results as wl_elev_rollout bit for bit.  Quad form only (n_envs <= 32 768, else WL_EINVAL); with more
 * than one step the per-step observation rows must be distinct (obs_step_stride >= n_envs * WL_ELEV_OBS_DIM).  Episode metrics
 * of all steps go to ring slot (step0 % slots), slot ((step0 + n_steps) % slots) is cleared (as wl_drift_rollout_persistent). */
int wl_elev_rollout_persistent(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* actions,
                               const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps, uint64_t seed,
                               uint64_t step0, void* stream);

/* ---- the runner's collection step in ONE launch, elevation task (policy in the loop) -------------------
 * modified_rsl_rl_runner.py:70-80 per step: actions = alg.act(obs) -> obs, rewards, dones = env.step(actions) -> storage.
 * `io` are rows k of an rsl_rl RolloutStorage: the observation row the policy reads and the action / mean / log-prob /
 * value rows it fills; `out` is where the step writes (observation row k + 1, reward / flags / dones rows k).  Quad form
 * only (n_envs <= 32 768; WL_EINVAL beyond: call wl_actor_critic_act + wl_elev_step).  Equals those two calls bit for bit
 * where wl_actor_critic_act splits the features four ways (<= 8192 rows), to rounding elsewhere.  Measured at 4096 envs:
 * 47.9 us against 45.4 us for the two calls (its 16-row blocks double the first-layer operand traffic from L2). */
typedef struct WlCollectIo {
    const float* obs_in;             /* [n][obs_dim] */
    float* actions;                  /* [n][2] */
    float* mu;                       /* [n][2] */
    float* log_prob;                 /* [n] */
    float* values;                   /* [n] */
} WlCollectIo;
int wl_elev_collect_step(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const WlMlp* actor, const WlMlp* critic,
                         const float* std, const WlCollectIo* io, const WlStepOut* out, int32_t deterministic, uint64_t seed,
                         uint64_t step, void* stream);

/* The runner's collection loop -- n_steps x { actions = actor(obs) -> env.step -> storage rows } (modified_rsl_rl_runner.py:
 * 70-80) -- as ONE launch.  `io` / `out` are rows 0 of [n_steps (+ 1)][n]... blocks of an rsl_rl RolloutStorage: step k reads
 * observation row io->obs_in + k n 689, writes actions / mu (+ k n 2), log_prob (+ k n), reward / flags / dones (+ k n) and the
 * next observation row out->obs + k n 689 -- out->obs MUST be io->obs_in + n 689 (the policy of step k + 1 reads what step k
 * wrote).  io->values is NOT written: the critic's values are not needed to step; evaluate the n_steps + 1 observation rows
 * in one batched call afterwards (wl_actor_critic_act with nets = critic half).  `critic` is validated and otherwise unused.
 * Block = 16 envs: the actor's first-layer matrix stays in the block's registers for the whole launch, the block's
 * observation rows in LDS.  Quad form only (n_envs <= 32 768).  Per step equal to { wl_actor_critic_act; wl_elev_step } to
 * fp32 rounding (first layer summed in eight partial sums instead of four), same random streams; a K-step launch equals K
 * one-step launches bit for bit.  Episode metrics of all steps go to ring slot (step0 % slots), slot ((step0 + n_steps) %
 * slots) is cleared. */
int wl_elev_collect_rollout(const WlElevParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const WlMlp* actor, const WlMlp* critic,
                            const float* std, const WlCollectIo* io, const WlStepOut* out, int32_t n_steps, int32_t deterministic,
                            uint64_t seed, uint64_t step0, void* stream);


/* ======================================================================================================== */
/* Visual task (wheeledlab_tasks/visual/mushr_visual_env_cfg.py)                                            */
/* ======================================================================================================== */

#define WL_VIS_IMG_H 60
#define WL_VIS_IMG_W 80
#define WL_VIS_CROP 20                          /* rows dropped from the top: H // 3 (mdp_sensors/observations.py:79) */
#define WL_VIS_NPIX ((WL_VIS_IMG_H - WL_VIS_CROP) * WL_VIS_IMG_W)   /* 3200                                        */
#define WL_VIS_OBS_DIM (WL_VIS_NPIX + 8)        /* 3208 (mushr_visual_env_cfg.py:38-58)                               */
#define WL_VISDEPTH_NPIX (WL_VIS_IMG_H * WL_VIS_IMG_W)   /* 4800: the uncropped depth image                              */
#define WL_VISDEPTH_OBS_DIM (WL_VISDEPTH_NPIX + 8)       /* 4808: visual-depth extension task (BASELINE config 5)         */

enum WlVisualRewTerm { WL_VR_TRAVERSABLE = 0, WL_VR_FORWARD_VEL, WL_VR_NTERMS };

/* traversability map (visual/utils/__init__.py:60-86): map[iy][ix] bytes, 1 = traversable, plus the list of
 * traversable cells (iy, ix) in nonzero order used for spawning (utils/__init__.py:188-202). */
typedef struct WlTravMap {
    const uint8_t* map;        /* [rows][cols] */
    const int32_t* cells;      /* [n_cells][2] = (iy, ix) */
    int32_t rows, cols, n_cells;
    float row_spacing, col_spacing;
    /* optional: the same map one BIT per cell, bit (k & 31) of word (k >> 5) for k = iy * cols + ix, ceil(rows * cols / 32)
       words (+ 0 padding bits).  With it (and rows * cols <= WL_VIS_LDS_MAP_CELLS) the persistent rollout keeps the whole map in
       LDS -- 31 KB for the reference's 500 x 500 map -- and a pixel's lookup is an LDS read instead of a byte gather (10 % off
       its step; the per-step camera launch keeps the byte gathers: measured faster there).  NULL: byte gathers from `map`. */
    const uint32_t* bits;
} WlTravMap;
#define WL_VIS_LDS_MAP_CELLS (512 * 512)

typedef struct WlVisualParams {
    float sim_dt;              /* 0.02 (:435)                                                                */
    int32_t decimation;        /* 10   (:436)                                                                */
    int32_t max_episode_length;/* ceil(10 s / 0.2 s) = 50 (:439)                                             */
    WlActionParams action;     /* Mushr4WDActionCfg                                                          */
    WlVehicleParams vehicle;   /* 4WD, ground friction 2.0 / 2.0 (:130-135)                                  */
    float weight[WL_MAX_REW_TERMS];   /* traversable_reward 5, forward_vel 7 (:375-385)                      */
    float reset_z;             /* 0.1 (:203)                                                                 */
    /* pinhole camera (:230-246): intrinsics from focal length / apertures; pose in the body frame is designed     */
    float cam_pos[3];
    float fx, fy, cx, cy;
    float sky;                 /* grey level of rays that miss the plane                                     */
    /* augmentation of this call (torchvision ColorJitter brightness / contrast + GaussianBlur(5) sigma,
       mdp_sensors/observations.py:21-23,82-84); 1, 1, 0 = none                                              */
    float brightness, contrast, blur_sigma;
    /* torchvision's ColorJitter.forward applies its four ops in a random permutation per call (torch.randperm(4)); on the
       rendered image (R = G = B) hue is the identity and saturation a scale within 1.8e-4 of 1, so what matters is whether the
       contrast blend runs BEFORE the brightness scale (1) or after it (0): they do not commute (clamps, mean)             */
    int32_t contrast_first;
    int32_t log_episode_sums;
} WlVisualParams;

/*
 * Fused visual env.step(): 4WD action term -> decimation x substeps on the flat plane -> time_out + out_of_map
 * (:390-398) -> traversable_reward (:309-312, lookup traversability_utils.py:68-88) + forward_vel (:370-371) -> reset
 * onto a random traversable cell (visual/mdp/events.py:11-42), then the observation: 40 x 80 ray-cast grey image of the
 * traversability plane staged in LDS, brightness / contrast / 5x5 Gaussian blur / grayscale / normalise
 * (mdp_sensors/observations.py:75-87), + base_lin_vel, base_ang_vel, last_action.  obs is [n][WL_VIS_OBS_DIM].
 * Two launches per step: the step (lane or quad of lanes = env), then the camera (block = env).
 */
int wl_visual_step(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const float* actions,
                   const WlStepOut* out, uint64_t seed, uint64_t step, void* stream);
int wl_visual_rollout(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const float* actions,
                      const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps,
                      uint64_t seed, uint64_t step0, void* stream);
/* wl_visual_rollout as ONE launch for open-loop rollouts (pre-staged actions): the camera of step k is rendered by twelve
 * wavefronts of each 16-env block while a thirteenth already integrates step k + 1.  Same results as wl_visual_rollout bit for
 * bit.  Quad form only (n_envs <= 32 768, else WL_EINVAL); with more than one step the per-step observation rows must be
 * distinct (obs_step_stride >= n_envs * WL_VIS_OBS_DIM).  Episode metrics of all steps go to ring slot (step0 % slots),
 * slot ((step0 + n_steps) % slots) is cleared (as wl_drift_rollout_persistent). */
int wl_visual_rollout_persistent(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const float* actions,
                                 const WlStepOut* out, int64_t obs_step_stride, int64_t vec_step_stride, int32_t n_steps,
                                 uint64_t seed, uint64_t step0, void* stream);
int wl_visual_reset(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const uint8_t* mask,
                    uint64_t seed, uint64_t step, void* stream);
int wl_visual_observe(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, float* obs, void* stream);
/*
 * Visual mdp terms on caller-supplied tensors (parity entry point): pos[3][stride], lin_vel_b[3][stride] ->
 * terms[2][stride] (traversable_reward, forward_vel), out_of_map[n] bytes, map indices x_idx[n], y_idx[n] (int32).
 * Replaces mushr_visual_env_cfg.py:309-312,370-371,390-398 and traversability_utils.py:68-88.
 */
int wl_visual_mdp(const WlVisualParams* p, const WlTravMap* m, int32_t n, int64_t stride, const float* pos,
                  const float* lin_vel_b, float* terms, uint8_t* out_of_map, int32_t* x_idx, int32_t* y_idx,
                  void* stream);
/*
 * Depth ray-cast of the same camera against a heightfield (BASELINE.json config 5 "depth raycast against heightfield";
 * the observation functions mdp_sensors/observations.py:89-95 `camera_data_depth` / `raycast_depth` forward IsaacLab's
 * `distance_to_image_plane` of the camera visual/mushr_visual_env_cfg.py:230-246 -- defined but unwired in the
 * reference's VisualObsCfg).  depth [n][60][80]: per pixel the distance along the optical axis to the first point of the
 * ray on or below the terrain solid -- bilinear patches of `hf` inside the grid, the plane z = hf->outside_z beyond it --
 * exact per cell (the patch along the ray is a quadratic; no marching step), clipped at max_depth.  The ray's ground
 * track is walked through a max-pyramid of the field that the caller builds ONCE per heightfield:
 *   wl_heightfield_pyramid_floats(nx, ny)  -> floats the pyramid needs (0: unsupported size; nx, ny <= 16 385)
 *   wl_heightfield_build_pyramid(hf, pyramid, stream)
 * The poses are rows WL_S_PX.. / WL_S_QW.. of `b->state` (any task's batch).  Spec: oracle/depth.c.
 */
int64_t wl_heightfield_pyramid_floats(int32_t nx, int32_t ny);
int wl_heightfield_build_pyramid(const WlHeightField* hf, float* pyramid, void* stream);
int wl_visual_depth(const WlVisualParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* pyramid,
                    float max_depth, float* depth, void* stream);
/* the same image into rows of `row_stride` floats (>= 4800): env e's image at rows + e * row_stride */
int wl_visual_depth_rows(const WlVisualParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* pyramid,
                         float max_depth, float* rows, int64_t row_stride, void* stream);
/*
 * Visual-depth EXTENSION task (not a reference id; BASELINE.json configs[4] "Visual task, 4096 envs, depth raycast against
 * heightfield"): the visual task's env.step() -- 4WD action term, traversable_reward / forward_vel, time_out / out_of_map, reset
 * onto a random traversable cell (mushr_visual_env_cfg.py:309-312,370-371,390-398; visual/mdp/events.py:11-42) -- driven on a
 * heightfield terrain (wheel contacts by bilinear gathers as in the elevation task; reset poses lifted onto the terrain), with
 * the camera's depth image as the policy observation: obs [n][WL_VISDEPTH_OBS_DIM] = distance_to_image_plane 60 x 80
 * (`raycast_depth`, mdp_sensors/observations.py:93-95) | base_lin_vel 3 | base_ang_vel 3 | last_action 2.
 *   wl_visual_step_hf       the step launch: state, reward, flags, and columns 4800 .. 4807 of out->obs
 *   wl_visual_reset_hf      reset (mask NULL = all envs)
 *   wl_visual_depth_step    step launch + depth launch (columns 0 .. 4799): one env.step()
 *   wl_visual_depth_observe columns 0 .. 4807 of the state as it stands (reset / first observation)
 */
int wl_visual_step_hf(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const WlHeightField* hf, const float* actions,
                      const WlStepOut* out, uint64_t seed, uint64_t step, void* stream);
int wl_visual_reset_hf(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const WlHeightField* hf, const uint8_t* mask,
                       uint64_t seed, uint64_t step, void* stream);
int wl_visual_depth_step(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const WlHeightField* hf, const float* pyramid,
                         float max_depth, const float* actions, const WlStepOut* out, uint64_t seed, uint64_t step, void* stream);
int wl_visual_depth_observe(const WlVisualParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* pyramid, float max_depth,
                            float* obs, void* stream);

/*
 * Startup-mode events (domain randomisation applied ONCE per env, at construction): bucketed wheel friction
 * (isaaclab randomize_rigid_body_material: `mu_buckets` materials per run, every env picks one; make_consistent:
 * mu_d <= mu_s), throttle damping (randomize_actuator_gains, "abs") and base mass (randomize_rigid_body_mass, "add"
 * onto the chassis mass).  Reference configs: mushr_drift_env_cfg.py:98-119,145-154; elevation cfg :387-407; visual
 * cfg :264-299.  Writes rows WL_S_MU_S, WL_S_MU_D, WL_S_DAMP, WL_S_MASS of every env and sets WL_S_QW = 1.
 * Keyed like every other draw: Philox counter (global env id = env_offset + e, 0, stream 8) for the env's bucket /
 * damping / mass and (bucket id, 0, stream 9) for a bucket's friction pair -- so a W-rank run of n envs each holds the
 * same W*n parameter sets as a 1-rank run of W*n envs (the reference draws per env, not per process).
 * randomize == 0: every env gets the mid-points of the ranges.
 */
typedef struct WlStartupParams {
    float wheel_mu_s[2], wheel_mu_d[2];   /* ranges of the wheel material's static / dynamic friction           */
    int32_t mu_buckets;                   /* num_buckets (20 in the drift task)                                  */
    int32_t mu_consistent;                /* make_consistent: mu_d = min(mu_d, mu_s)                             */
    float damping[2];                     /* throttle damping range                                              */
    float chassis_mass;                   /* base mass before the "add" operation                                */
    float mass_add[2];                    /* added mass range                                                    */
    int32_t randomize;
    float wheel_mass[2];                  /* range of EACH wheel link's mass ("abs", visual cfg :289-298): the four draws add  */
                                          /* to the vehicle's mass row; (0, 0): not randomised (wheels counted in chassis_mass) */
} WlStartupParams;
int wl_startup_randomize(const WlStartupParams* su, const WlEnvBuffers* b, uint64_t seed, void* stream);

/* Raw Philox4x32 (7 rounds) uniforms, 24 bits of each word, as used in-kernel: out[4][n] for counter (env, step, stream_id). Test hook. */
int wl_philox_uniform(int32_t n, uint64_t seed, uint64_t step, uint32_t stream_id, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WHEELEDLAB_AMD_H */

/*
 * tma.h -- C ABI of libtma_hip.so: the MI355X (gfx950) vector-env + PPO hot path for
 * lukehollis/three-mlagents.  Plain pointers and sizes only; every device pointer is HBM on the
 * handle's device; `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *
 * The reference has no FFI: its boundary is a Python API.  Each entry point cites
 * the reference interface (path:line under /root/reference/) whose work it replaces.  The Python
 * shim (three-mlagents_amd/_lib.py) binds these with ctypes and mirrors the reference's operator
 * names; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Every function returns a TMA_* status; tma_last_error() gives the message (thread-local).
 * Status -> Python exception in the shim:  INVALID -> ValueError (registry.py:368-369,
 * training.py:105-114), UNKNOWN_TASK -> KeyError (registry.py:359-362), HIP -> RuntimeError.
 */
#ifndef TMA_H
#define TMA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMA_VERSION 208

enum { TMA_OK = 0, TMA_ERR_INVALID = 1, TMA_ERR_UNKNOWN_TASK = 2, TMA_ERR_HIP = 3 };

/* task ids (backend/mlagents/registry.py:52-116,225-240) */
enum { TMA_TASK_BASIC = 0, TMA_TASK_GRIDWORLD = 1, TMA_TASK_BALL3D = 2, TMA_TASK_PUSH = 3, TMA_TASK_CRAWLER = 4, TMA_TASK_WALLJUMP = 5, TMA_TASK_BICYCLE = 6, TMA_TASK_BRICKBREAK = 7, TMA_TASK_GLIDER = 8,
       TMA_TASK_ANT = 9, /* the reference's `ant` shapes (Ant-v5: 105 observations, 8 torques; envs.py:274-277) on the build's chain dynamics; CRAWLER = BASELINE.json's 172 / 20 shape */
       TMA_NUM_TASKS = 10 };

/* dtype of the `actions` buffer handed to tma_env_step */
enum { TMA_ACT_I32 = 0, TMA_ACT_I64 = 1, TMA_ACT_F32 = 2 };

/* episode k of (global) env i is reset with numpy-legacy seed  base + i + k * TMA_EP_STRIDE  (mod 2^32);
 * k = 0 is the reference's `seed + rank` (backend/mlagents/training.py:80,84). */
#define TMA_EP_STRIDE (1u << 20)

int tma_version(void);
const char *tma_last_error(void);

/* ---- task metadata: spaces declared by make_*_env (backend/mlagents/envs.py:35-44,162-199,274-277) ---- */
int tma_task_id(const char *name, int *task_out);
int tma_task_obs_dim(int task);           /* 21 / 4 / 6 / 4 / 172 / 4 / 7 / 45 / 16 / 105 */
int tma_task_num_actions(int task);       /* Discrete(n); 0 for a Box action space */
int tma_task_act_dim(int task);           /* Box action dim (crawler: 20, ant: 8), else 1 */
int tma_task_state_dim(int task);         /* doubles per env in the flat get/set_state layout */
int tma_task_max_episode_steps(int task); /* 50 / 100 / 200 / 120 / 1000 / 150 / 2000 / 2000 / 4000 / 1000 */

/* ---- vector env: replaces make_vector_env + DummyVecEnv + Monitor + LegacySingleAgentGymAdapter +
 *      the task step()/reset() (backend/mlagents/training.py:71-89; backend/mlagents/envs.py:30-159;
 *      backend/examples/{gridworld.py:40-95, ball3d.py:47-113, push.py:39-125, walljump.py:40-98, bicycle.py:40-141,
 *      brick_break.py:39-131, glider.py:55-265}) ---- */
typedef struct tma_env tma_env;

/* env_offset = global index of this shard's env 0 (data-parallel sharding); ring_depth = look-ahead of
 * pre-drawn reset states per env (>= 2; the MT19937 reset states of future episodes are produced by a
 * separate refill kernel every ring_depth steps). */
int tma_env_create(int task, int64_t num_envs, int device, uint32_t seed_base, uint32_t env_offset, int ring_depth,
                   tma_env **out);
/* destroy drains the device; the handle's device blocks (up to 64 MiB each, 512 MiB per process) are kept for the next tma_env_create that asks
 * for the same sizes -- the callers this replaces build and close a vector env per training run and per evaluation */
int tma_env_destroy(tma_env *h);
/* VecEnv.seed(seed): env i -> seed + i at the next reset (SB3 DummyVecEnv) */
int tma_env_seed(tma_env *h, uint32_t seed_base);
/* VecEnv.reset(): every env starts episode 0;  obs_out f32[num_envs][obs_dim] */
int tma_env_reset(tma_env *h, float *obs_out, void *stream);
/*
 * VecEnv.step_wait() for n_steps consecutive vector steps in ONE launch (n_steps = 1 is the plain
 * VecEnv.step).  actions: [n_steps][num_envs] (I32/I64) or [n_steps][num_envs][act_dim] (F32); if NULL the
 * counter-based tape a(i,t) = mix32(tape_seed, global_i, tape_t0 + s) % n_actions is generated on device.
 * Outputs are [n_steps][num_envs][...]; obs is the post-auto-reset observation (what DummyVecEnv
 * returns), term_obs the pre-reset observation where done (info["terminal_observation"]), ep_ret/ep_len the
 * Monitor episode sum/length where done (else 0).  rew/term/trunc/term_obs/ep_ret/ep_len may be NULL.
 * n_steps must not exceed tma_env_steps_until_refill().
 */
int tma_env_step(tma_env *h, const void *actions, int action_dtype, uint32_t tape_seed, uint32_t tape_t0, int n_steps,
                 float *obs_out, float *rew_out, uint8_t *term_out, uint8_t *trunc_out, float *term_obs_out,
                 double *ep_ret_out, int32_t *ep_len_out, void *stream);
/* Seam S1 (backend/mlagents/envs.py:125-152): the reference's single env returns `float(reward)` of the value the task computed in float64
 * (Basic 0.09000000000000001, Bicycle / BrickBreak / Glider rewards out of float64 physics); `rew_out` carries its float32 rounding, which is
 * what SB3's VecEnv keeps.  tma_env_set_reward64(env, plane, capacity): every tma_env_step launched afterwards ALSO stores the float64 reward
 * at plane[k * num_envs + i] (device memory owned by the caller, `capacity` doubles >= num_envs; a step call with n_steps * num_envs beyond it
 * is refused with TMA_ERR_INVALID); NULL turns it off -- the caller does that before it frees the plane.  The fused rollout kernels
 * (tma_rollout_collect) do not write it: they replace SB3's float32 buffer, not the Gymnasium single-env surface.  (ABI 208: the capacity.) */
int tma_env_set_reward64(tma_env *env, double *plane, int64_t capacity);
/* `reps` consecutive single-step launches with the same action buffer and output planes, issued from native code with no
 * host-language round trip in between (launch-latency measurements; semantics = calling tma_env_step `reps` times) */
int tma_env_step_repeat(tma_env *h, const void *actions, int action_dtype, int reps, float *obs_out, float *rew_out, uint8_t *term_out,
                        uint8_t *trunc_out, float *term_obs_out, void *stream);
int tma_env_steps_until_refill(tma_env *h, int *out);
/* engine options; "refill_small_window" = 1 shortens the register-resident MT19937 window so tests exercise the
 * general in-memory generator that takes over when rejection sampling needs more outputs */
int tma_env_set_option(tma_env *h, const char *key, int64_t value);
/* re-draw the reset states consumed since the last refill (exact numpy MT19937 legacy stream) */
int tma_env_refill(tma_env *h, void *stream);
/* flat float64 state [num_envs][state_dim] in the oracle's layout (state injection for parity tests;
 * also BasicMoveToGoalEnv.reset(options={"position": p}), backend/mlagents/envs.py:54-57) */
int tma_env_get_state(tma_env *h, double *state_out, void *stream);
int tma_env_set_state(tma_env *h, const double *state_in, void *stream);
int tma_env_episode_index(tma_env *h, uint32_t *out, void *stream);
/* Optional per-episode Monitor log (SB3 Monitor writes one `r,l,t` row per finished episode: reference training.py:85-86).  capacity > 0
 * allocates a device log of that many records and turns logging on in every step / rollout kernel; 0 turns it off.  Synchronises the
 * device.  tma_env_pop_episode_log copies up to max_records records (return, length, env index; in the order the kernels appended them)
 * to HOST memory, reports how many were stored (*n_stored) and how many episodes ended since the last pop (*n_seen >= *n_stored when the
 * log overflowed), and empties the log.  Synchronises `stream`. */
int tma_env_episode_log(tma_env *env, int64_t capacity);
int tma_env_pop_episode_log(tma_env *env, double *ret_host, int32_t *len_host, int32_t *env_host, int64_t max_records, int64_t *n_stored,
                            int64_t *n_seen, void *stream);
/* SB3 Monitor file rows (backend/mlagents/training.py:85-86: every env is wrapped in Monitor(env, filename), which appends one `r,l,t` line
 * per finished episode): appends n lines "round(ret, 6),len,round(t, 6)" to `path`, printed like Python prints those values.  HOST arrays,
 * no GPU work, thread-safe for distinct paths: the Python layer calls it from a writer thread. */
int tma_monitor_append_rows(const char *path, const double *ret, const int32_t *len, const double *t, int64_t n);
/* Monitor aggregate since the last call: out[0]=sum of episode returns, out[1]=sum of lengths, out[2]=count.
 * Synchronises `stream`. */
int tma_env_pop_episode_stats(tma_env *h, double *out3_host, void *stream);
/* Empty the episode log and the Monitor aggregate ordered on `stream` (two hipMemsetAsync), without reading them and without synchronising:
 * what a deterministic evaluation does before its first chunk (SB3's evaluate_policy starts from fresh Monitor state by construction,
 * backend/mlagents/training.py:240-247) -- whatever an earlier user of the env left, kernels launched on `stream` after this call start from zero. */
int tma_env_clear_episode_log(tma_env *env, void *stream);
/* Two-phase pop for a training loop that keeps the GPU busy across iterations (round 4; the reference's Monitor / logger run on the host
 * between rollouts, training.py:85-86,152-161 -- here they must not sit between two GPU iterations).  tma_env_detach_episode_log: HOST-side
 * swap of the buffer set handed to kernels at launch -- kernels launched BEFORE the call wrote their Monitor aggregates and episode records
 * into the set that is now detached, later launches write into the other, empty set; no stream is touched.  tma_env_pop_detached_episode_log:
 * tma_env_pop_episode_log + tma_env_pop_episode_stats of the detached set on any stream ordered behind those earlier kernels (a side stream
 * behind an event); empties it; synchronises only `stream`.  One detached set at a time (TMA_ERR_INVALID otherwise). */
int tma_env_detach_episode_log(tma_env *env);
int tma_env_pop_detached_episode_log(tma_env *env, double *ret_host, int32_t *len_host, int32_t *env_host, int64_t max_records, int64_t *n_stored,
                                     int64_t *n_seen, double *stats3_host, void *stream);

/* ---- GAE: replaces SB3 RolloutBuffer.compute_returns_and_advantage (3P, constructed at
 *      backend/mlagents/training.py:150; hyper-parameters training.py:383-384).  All [T][N] f32. ---- */
int tma_gae(const float *rewards, const float *values, const float *episode_starts, const float *last_values,
            const uint8_t *dones, double gamma, double gae_lambda, int T, int64_t N, float *adv_out, float *ret_out,
            void *stream);

/* ---- actor-critic MLP + PPO update: replaces the SB3 objects PPO("MlpPolicy", env, **kwargs) builds at
 *      backend/mlagents/training.py:150 (policy_kwargs net_arch pi/vf = [H, H], training.py:363-365; hyper-parameters
 *      training.py:377-390) and model.learn() drives at training.py:166-170.  Third-party semantics: SURVEY.md App. C. ---- */
typedef struct {
    int obs_dim;    /* D */
    int hidden;     /* H: two tanh layers of width H for both the policy and the value net (multiple of 64) */
    int act_dim;    /* Discrete(n): n (2..16); Box: action dimension (1..32) */
    int continuous; /* 0: Categorical head; 1: DiagGaussian head with a state-independent log_std */
    int mfma_dtype; /* 0: f32 MFMA everywhere (parity mode, every hidden width); 1: bf16 MFMA operands with f32 master
                       weights and f32 accumulation for the hidden-layer and head GEMMs (hidden = 128 / 192 / 256 only:
                       BASELINE.json configs[2] "PPO MLP(256,256) bf16").  Rollout and update use the same forward code.
                       2 (round 5, opt-in): the f32 UPDATE of a Discrete 256 x 256 policy (the reference's default net,
                       backend/mlagents/training.py:363-365; up to 32 observations) on the bf16 MFMA with every operand as three bf16
                       terms and the six products of order <= 2 -- f32-class accuracy (2^-21 against float64, the exact-f32 MFMA's own
                       figure) at ~1.8x the f32 matrix pipe; rollouts, evaluation and small minibatches run the exact-f32 kernels. */
    int device;     /* HIP device the parameter / rollout / workspace buffers live on: every tma_policy_* / tma_ppo_* call makes it
                       the calling thread's current device first (callers may be worker threads: backend/main.py:152
                       asyncio.to_thread).  -1: keep whatever device is current on the calling thread. */
} tma_policy_dims;

/* parameter buffer = n_total floats: [0, n_trainable) trainable, [in][out] layout, order
 * pi.W1 pi.b1 pi.W2 pi.b2 pi.W3(action_net) pi.b3 vf.W1 vf.b1 vf.W2 vf.b2 vf.W3(value_net) vf.b3 [log_std];
 * the rest are [out][in] copies maintained by tma_policy_sync / tma_ppo_adam_step. */
int tma_policy_param_count(const tma_policy_dims *d, int64_t *n_trainable, int64_t *n_total);
int tma_policy_param_offsets(const tma_policy_dims *d, int32_t *out13);
int tma_policy_sync(float *params, const tma_policy_dims *d, void *stream);
/* ActorCriticPolicy.forward(obs, deterministic): actions i32[n] (Discrete) or f32[n][act_dim] (Box, unclipped),
 * values f32[n], log_prob f32[n].  Sampling uses the counter-based stream (rng_seed, env_offset + row, rng_step). */
int tma_policy_act(const float *params, const tma_policy_dims *d, const float *obs, int64_t n, uint32_t rng_seed, uint32_t rng_step,
                   uint32_t env_offset, int deterministic, void *actions_out, float *values_out, float *logp_out, void *stream);
/* tma_policy_act for step t plus tma_policy_bootstrap for step t-1 in one launch (prev_* may be NULL) */
int tma_policy_act_bootstrap(const float *params, const tma_policy_dims *d, const float *obs, int64_t n, uint32_t rng_seed, uint32_t rng_step,
                             uint32_t env_offset, void *actions_out, float *values_out, float *logp_out, const float *prev_terminal_obs,
                             const uint8_t *prev_truncated, double gamma, float *prev_rewards_inout, void *stream);
/* ActorCriticPolicy.predict_values */
int tma_policy_values(const float *params, const tma_policy_dims *d, const float *obs, int64_t n, float *values_out, void *stream);
/* collect_rollouts timeout bootstrap: rewards[i] += gamma * V(terminal_obs[i]) where truncated[i] */
int tma_policy_bootstrap(const float *params, const tma_policy_dims *d, const float *terminal_obs, const uint8_t *truncated, int64_t n,
                         double gamma, float *rewards_inout, void *stream);

typedef struct {
    const float *obs;       /* [T][N][D]                                  RolloutBuffer.observations */
    const void *actions;    /* i32 [T][N] (Discrete) or f32 [T][N][A] (Box)              .actions    */
    const float *log_probs; /* [T][N]                                                     .log_probs  */
    const float *advantages;
    const float *returns;
    int T;
    int64_t N;
    const float *packed;    /* optional, NULL = absent: [T][N][tma_ppo_packed_floats / (T*N)] sample records {obs (padded to a multiple of 4), log_prob,
                               advantage, action bits, return} written by tma_ppo_pack_samples from the five planes above.  With it the H = 64
                               gradient kernel reads ONE record (one 64-byte line) per sample and net instead of gathering from five planes.
                               Results are bit-identical with and without it. */
} tma_rollout;

typedef struct {
    const int64_t *indices; /* optional explicit permutation of the env-major flat index f = i*T + t (SB3 swap_and_flatten);
                               NULL -> on-device Feistel permutation keyed by (perm_seed, perm_epoch) */
    uint32_t perm_seed, perm_epoch;
    int64_t start, count;   /* this minibatch = permuted rows [start, start + count) */
    int64_t prepared_batch; /* 0: self-contained call.  > 0: tma_ppo_epoch_prepare ran on this workspace for the same rollout view,
                               (perm_seed, perm_epoch) and this batch size, start is a multiple of it -- the per-minibatch advantage
                               pass is skipped and the cached sample offsets / advantage partials are used */
    int64_t stats_count;    /* 0: advantages are normalised with the mean / unbiased std of this minibatch's `count` rows.
                               > 0 (data-parallel runs, prepared epochs only): the workspace partials of this minibatch were replaced
                               by sums over the GLOBAL minibatch of stats_count rows (tma_ppo_epoch_adv_sums) -- every rank then
                               normalises with the same global mean / std, as one SB3 run over the concatenated batch would */
} tma_minibatch;

typedef struct {
    double clip_range, ent_coef, vf_coef;
    int normalize_advantage;
} tma_ppo_hparams;

/* Sample records for tma_rollout.packed: floats the plane needs for this policy shape and T x N samples (0: the shape has no use for it --
 * today: hidden 64, Discrete head, <= 8 observations), and the pass that fills it (once per rollout, after the advantages and returns exist). */
int64_t tma_ppo_packed_floats(const tma_policy_dims *d, int T, int64_t N);
int tma_ppo_pack_samples(const tma_rollout *rb, const tma_policy_dims *d, float *packed_out, void *stream);
/* bytes of the update workspace for a policy shape (loss-stat slots, norm partials, partial-gradient slabs, sample-offset cache;
 * bf16 layouts with 33..64 or 161..192 observations: + a dz1 cache of 2 * 262144 * hidden bf16 for the dW1 launch; f32 layouts with
 * hidden 128/192/256 and 161..176 observations: the same in f32) */
int64_t tma_ppo_workspace_bytes(const tma_policy_dims *d);
/* PPO.train inner loop body up to loss.backward(): accumulates d(loss)/d(params) into grad[n_trainable] (caller zeroes it
 * once; tma_ppo_adam_step re-zeroes it) and loss statistics into the workspace. */
int tma_ppo_minibatch_grad(const float *params, const tma_policy_dims *d, const tma_rollout *rb, const tma_minibatch *mb,
                           const tma_ppo_hparams *hp, float *grad, void *workspace, void *stream);
/* Once per epoch (optional): sample offsets of the whole permutation and the advantage (sum, sum of squares) partials of every
 * minibatch [k*batch_size, (k+1)*batch_size) in one launch, instead of one small launch in front of every minibatch gradient.
 * epoch->start / count describe the whole pass (0, T*N).  Needs T*N <= 2^22 and batch_size >= 256, else TMA_ERR_INVALID. */
int tma_ppo_epoch_prepare(const tma_rollout *rb, const tma_minibatch *epoch, int64_t batch_size, const tma_policy_dims *d, void *workspace,
                          void *stream);
/* Data-parallel advantage statistics ("12-byte all-reduce for global advantage mean/var"), for an epoch that
 * tma_ppo_epoch_prepare just prepared on this workspace with the same batch_size over `total` = T*N samples.
 * direction 0 (export): sums[k] = {sum, sum of squares} of minibatch k's advantages (double[n_minibatches][2], device memory), folded
 * from the workspace partials in a fixed order.  The caller all-reduces (sum) that buffer over the ranks -- ONE collective of
 * 16 * n_minibatches bytes per epoch -- and hands it back with direction 1 (import), which makes it the workspace's partials;
 * the epoch's minibatches are then run with tma_minibatch.stats_count = global row count. */
int tma_ppo_epoch_adv_sums(void *workspace, const tma_policy_dims *d, int64_t batch_size, int64_t total, double *sums, int direction,
                           void *stream);
/* clip_grad_norm_(max_grad_norm) + Adam.step() (+ refresh of the [out][in] copies).  grad_scale multiplies the gradient
 * first (1/world_size after an all-reduce(sum)). */
int tma_ppo_adam_step(float *params, float *grad, float *exp_avg, float *exp_avg_sq, const tma_policy_dims *d, int64_t step, double lr,
                      double beta1, double beta2, double eps, double max_grad_norm, double grad_scale, void *workspace, void *stream);
/* Same update for the single-GPU case where `grad` is exactly what the LAST tma_ppo_minibatch_grad on this workspace produced
 * (no all-reduce, no accumulation over several minibatches; last_count = that minibatch's sample count): the gradient norm is taken
 * from partial sums the gradient reduction already left in the workspace and the derived copies are written by the optimizer kernel
 * itself (one launch instead of three).  Falls back to tma_ppo_adam_step(grad_scale = 1) for shapes without that fast path. */
int tma_ppo_adam_step_local(float *params, float *grad, float *exp_avg, float *exp_avg_sq, const tma_policy_dims *d, int64_t step, double lr,
                            double beta1, double beta2, double eps, double max_grad_norm, void *workspace, void *stream, int64_t last_count);
/* One whole epoch of PPO.train on ONE GPU in a single call: [tma_ppo_epoch_prepare] + for every minibatch of batch_size rows of the
 * (perm_seed, perm_epoch) permutation: tma_ppo_minibatch_grad + tma_ppo_adam_step_local, issued natively with no host-language round trip
 * in between (at the reference's literal batch_size = 256 and 4096 envs an epoch is 16 384 optimizer steps: backend/mlagents/training.py:379).
 * first_step = Adam step index of the epoch's first minibatch (>= 1); grad must be zero on entry and is zero on return.  On H = 64 fast-path
 * policies with every minibatch >= 256 rows the optimizer step of minibatch k runs in the prologue of gradient launch k + 1 (two launches per
 * minibatch; the state ping-pongs between the caller's buffers and a copy in the workspace and ends in the caller's buffers; TMA_NO_ADAM_FOLD=1
 * in the environment: one optimizer launch per minibatch).  Bit-identical to the
 * per-minibatch calls -- except on H = 64 policies at batch_size = 256 with T*N a multiple of it, where the epoch runs as ONE persistent
 * launch (csrc/tma_h64p.hip: weights in LDS, Adam moments in registers, eight workgroups of one XCD exchanging partial gradients through
 * the L2): same gradient sums and Adam arithmetic, the clip norm's f64 sum in another fixed order (parameters agree to the last bit or
 * two) -- and, since ABI 208, on the reference's DEFAULT policy (two 256 x 256 tanh nets, f32 weights: mfma_dtype 0 or 2; Discrete head of
 * <= 16 actions, <= 32 observations; backend/mlagents/training.py:363-365) at batch_size = 256, where the epoch is ONE persistent launch of
 * csrc/tma_h256p.hip: each net on the 32 CUs of one XCD as 4 row groups x 8 column slices, weights of a slice in LDS, Adam moments in
 * registers, four same-XCD exchanges of activations and partial gradients per optimizer step; its sums have their own fixed order, so it is
 * run-to-run bit-identical and equal to the per-minibatch launches to rounding (2e-6 on the parameters after three epochs; 18.9 us per
 * optimizer step where the three launches take 34.2).  TMA_NO_PERSIST=1 in the environment selects the per-minibatch launches
 * (TMA_NO_PERSIST256=1: for the 256-wide kernel only).  Should a persistent kernel fail to place or synchronise its workgroups (eight
 * resident on one XCD; 32 on each of two for the 256-wide one) it commits nothing; this call notices (it waits for the persistent launch and
 * reads one status word back), restores its snapshot, re-runs the epoch through the per-minibatch launches and counts the event
 * (tma_ppo_persist_fallbacks). */
int tma_ppo_train_epoch_local(float *params, const tma_policy_dims *d, const tma_rollout *rb, uint32_t perm_seed, uint32_t perm_epoch,
                              int64_t batch_size, const tma_ppo_hparams *hp, float *grad, float *exp_avg, float *exp_avg_sq, int64_t first_step,
                              double lr, double beta1, double beta2, double eps, double max_grad_norm, void *workspace, void *stream);
/* `n_epochs` consecutive epochs (permutations perm_epoch0, perm_epoch0 + 1, ...; optimizer steps first_step ...) in one call.  Where a persistent
 * epoch kernel takes the shape (batch_size 256: H = 64 fast-path layouts, and the reference's default 256 x 256 f32 policy with a Discrete head
 * and <= 32 observations) and n_epochs * T * N sample offsets fit the workspace cache (2^22), ALL the epochs run as ONE launch -- the
 * reference's own 1- and 8-env schedules are 4 and 32 optimizer steps per epoch, where a launch per epoch is mostly launch; otherwise (and
 * when such a launch cannot place its workgroups) exactly tma_ppo_train_epoch_local per epoch.  Same results either way. */
int tma_ppo_train_epochs_local(float *params, const tma_policy_dims *d, const tma_rollout *rb, uint32_t perm_seed, uint32_t perm_epoch0, int n_epochs,
                               int64_t batch_size, const tma_ppo_hparams *hp, float *grad, float *exp_avg, float *exp_avg_sq, int64_t first_step,
                               double lr, double beta1, double beta2, double eps, double max_grad_norm, void *workspace, void *stream);
/* One epoch of PPO.train on ONE RANK of a data-parallel job (env shards per GPU, one gradient all-reduce per
 * minibatch; the reference itself is one process, backend/mlagents/training.py:71-89,150): for every minibatch of batch_size local rows --
 * tma_ppo_minibatch_grad, then `allreduce(ctx, grad, n_trainable)` (the caller's collective: SUM over the ranks, in place, enqueued on
 * `stream` or ordered against it -- torch.distributed.all_reduce on the current stream does that; must return 0), then
 * clip_grad_norm_ + Adam on grad * grad_scale (1 / world size) -- issued natively, the collective being the only host-language call per
 * minibatch.  Bit-identical to tma_ppo_minibatch_grad / all-reduce / tma_ppo_adam_step(grad_scale) called in a loop.
 * prepared_batch: batch_size if the caller ran tma_ppo_epoch_prepare (+ tma_ppo_epoch_adv_sums) for this epoch, else 0;
 * stats_world: multiplier of a minibatch's row count for its advantage statistics (world size with global statistics, else 0 = local).
 * grad must be zero on entry and is zero on return. */
typedef int (*tma_allreduce_fn)(void *ctx, float *buffer, int64_t count);
int tma_ppo_train_epoch_dp(float *params, const tma_policy_dims *d, const tma_rollout *rb, uint32_t perm_seed, uint32_t perm_epoch,
                           int64_t batch_size, int64_t prepared_batch, int stats_world, const tma_ppo_hparams *hp, float *grad, float *exp_avg,
                           float *exp_avg_sq, int64_t first_step, double lr, double beta1, double beta2, double eps, double max_grad_norm,
                           double grad_scale, tma_allreduce_fn allreduce, void *ctx, void *workspace, void *stream);
/* ---- RCCL communicator owned by the library (round 4): the collective of tma_ppo_train_epoch_dp without a host-language hop.
 *      The reference has no distributed code (one process: backend/mlagents/training.py:71-89,150); north_star's only collective is the
 *      per-minibatch SUM of the flat f32 policy gradient over the GPUs of a node.
 * Rank 0 draws a unique id (128 opaque bytes, ncclGetUniqueId), the caller broadcasts it over whatever channel it has (the torch.distributed
 * process group it rendezvoused with: any backend), every rank then calls tma_comm_create (ncclCommInitRank; collective over the ranks;
 * device < 0: the calling thread's current device).  tma_comm_allreduce: in-place SUM of `count` elements (dtype 0 = f32, 1 = f64) enqueued
 * on `stream` -- kernels queued on that stream before / after it are ordered before / after the collective, nothing else is needed.
 * tma_comm_allreduce_cb has the tma_allreduce_fn signature: pass it to tma_ppo_train_epoch_dp with the communicator as ctx (after
 * tma_comm_bind_stream(comm, the stream given to the epoch call)) and a data-parallel epoch runs without leaving native code.
 * RCCL is bound at run time (dlopen librccl.so.1: inside a PyTorch process that is the copy PyTorch loaded); tma_comm_available() == 0 and
 * TMA_ERR_HIP from the other entries when it cannot be.  tma_comm_timing(comm, n): bracket the next n all-reduces with HIP events on their
 * stream; tma_comm_pop_timing waits for them and returns their durations in microseconds (bench.py dp_timing) and the number of
 * all-reduces issued so far. */
typedef struct tma_comm tma_comm;
int tma_comm_available(void);
int tma_comm_unique_id(unsigned char *id_out128);
int tma_comm_create(const unsigned char *id128, int world, int rank, int device, tma_comm **out);
int tma_comm_destroy(tma_comm *comm);
int tma_comm_bind_stream(tma_comm *comm, void *stream);
int tma_comm_allreduce(tma_comm *comm, void *buffer, int64_t count, int dtype, void *stream);
int tma_comm_allreduce_cb(void *ctx, float *buffer, int64_t count);
int tma_comm_timing(tma_comm *comm, int samples);
int tma_comm_pop_timing(tma_comm *comm, float *us_out, int capacity, int *n_out, int64_t *calls_out);
/* ---- Peer exchange (round 5, ABI 207): the same SUM as direct xGMI stores between the GPUs of ONE node (up to 8 ranks), for messages where a
 *      ring's 2 (world - 1) hops and the collective's own launch are the cost (the 64 x 64 policy's gradient is 37 KB, 320 times per headline
 *      iteration).  Every rank owns an inbox in fine-grained device memory that its peers map through a HIP IPC handle; a sender stores each
 *      32-bit payload word with the all-reduce's sequence number next to it as ONE 8-byte word into its slot of every inbox, a receiver reads
 *      the `world` slots of its own inbox (system-scope 8-byte loads, until every word carries the expected sequence number) and adds them in
 *      rank order -- the same order on every rank, so replicas stay bit-identical; no fence and no ordering between words is assumed
 *      (csrc/tma_p2p.h has the protocol and why two slot parities suffice).
 * Set-up: tma_comm_p2p_prepare(comm, max_words, ticket_out[128]) allocates the inbox (slots of max_words 8-byte words; an f32 element is one
 * word, an f64 element two) and writes a 128-byte ticket (the inbox's IPC handle, the PCI bus id of its device, a 64-bit identity of the host); the caller gathers the
 * `world` tickets over the channel it already has (rank order) and gives them to tma_comm_p2p_attach, which refuses (TMA_ERR_HIP, nothing
 * mapped) unless every peer runs on THIS host and its device is this rank's own or a visible one with peer access (switched on there).  A
 * receiver that does not get its words within the timeout raises the communicator's error flag and delivers NaN (never a sum of stale
 * words); every later exchange then fails at once (tma_comm_p2p_status.timed_out, checked by the caller behind its stream synchronisation);
 * tma_comm_p2p_enable(comm, 1) then routes every tma_comm_allreduce /
 * tma_comm_allreduce_cb whose message fits a slot through the exchange (larger ones keep RCCL), and tma_ppo_train_epoch_dp -- given
 * tma_comm_allreduce_cb and such a communicator -- FUSES it on its H = 64 path: the slab reduction stores the reduced gradient into the
 * peers' inboxes, the sum-of-squares pass in front of the optimizer step reads the sum: no collective launch in the minibatch chain at all.
 * tma_comm_create_p2p: a communicator WITHOUT an RCCL side (several ranks on one GPU, which RCCL does not allow: the one-GPU tests); its
 * all-reduces must fit the exchange.  A receiver that waits longer than TMA_P2P_TIMEOUT_S (default 120) for a peer's words raises a flag:
 * the all-reduce that was waiting returns NaN (ABI 208; never a sum of stale words), every later one fails with TMA_ERR_HIP,
 * tma_comm_p2p_status reports it -- nothing spins for ever.  tma_comm_timing brackets the RECEIVING kernel of an exchange (what the chain waits for once the sender kernel is done). */
int tma_comm_create_p2p(int world, int rank, int device, tma_comm **out);
int tma_comm_p2p_prepare(tma_comm *comm, int64_t max_words, unsigned char *ticket_out128);
int tma_comm_p2p_attach(tma_comm *comm, const unsigned char *tickets_world_x_128);
/* The same wiring for `world` communicators that live in ONE process (each created with tma_comm_create_p2p(world, r, device) and prepared
 * with the same max_words): peers[r] is rank r's communicator, its inbox is used directly.  For drivers that run several ranks as streams of
 * one process -- the tests run the exchange at the world sizes of a node (4, 8) on one GPU this way; a receiver spins on words another
 * rank's sender stores, so every rank needs a stream (hardware queue) of its own. */
int tma_comm_p2p_attach_local(tma_comm *comm, tma_comm *const *peers_world);
int tma_comm_p2p_enable(tma_comm *comm, int on);
int tma_comm_p2p_status(tma_comm *comm, int *enabled_out, int64_t *calls_out, int *timed_out_out, int64_t *slot_words_out);
int tma_comm_p2p_set_timeout(tma_comm *comm, double seconds); /* receivers of later exchanges give up after this long (set-up: a short one for the self-check) */
/* How many epochs of tma_ppo_train_epoch_local on this workspace fell back from the persistent launch to per-minibatch launches.  Synchronises `stream`. */
int tma_ppo_persist_fallbacks(void *workspace, int64_t *count_out, void *stream);
/* The minibatch order of the on-device permutation (the engine's stand-in for np.random.permutation in SB3's RolloutBuffer.get): writes, to
 * HOST memory, the env-major flat indices f = i*T + t of permuted rows [0, total) for (perm_seed, perm_epoch).  Minibatch m of size B is
 * rows [m*B, (m+1)*B).  Lets a caller reproduce or log the schedule; no GPU work. */
int tma_ppo_permutation(uint32_t perm_seed, uint32_t perm_epoch, int64_t total, int64_t *indices_out_host);
/* Profiling aid for bench.py's roofline object: when enabled, tma_ppo_minibatch_grad brackets its DOMINANT kernel (the persistent
 * forward+backward kernel; for two-pass shapes both passes; not the advantage pass, not the slab reduction) with HIP events recorded on
 * the stream it launches on; tma_debug_last_grad_kernel_us waits for the last bracketed launch and returns its duration. */
int tma_debug_time_grad_kernel(int enable);
/* Test aid: fills the LDS of every CU with `pattern` (0: quiet NaNs) so that a kernel reading LDS it never wrote shows it in its outputs. */
int tma_debug_poison_lds(unsigned pattern, void *stream);
int tma_debug_last_grad_kernel_us(float *us_out);
/* out8: sums since the last call of {policy_loss, value_sq_err, entropy, approx_kl, clipped, n_samples}, then the last
 * total grad norm and clip coefficient.  Synchronises `stream`. */
int tma_ppo_pop_stats(void *workspace, double *out8_host, void *stream);
/* The same in two phases, for a loop that must not wait for the update it has just queued: tma_ppo_stats_enqueue copies the raw slots into
 * `staging_host` (tma_ppo_stats_staging_bytes() bytes of host memory; pinned memory makes the copy asynchronous) and clears them, ordered
 * on `stream`, without waiting; after the caller has synchronised with that point of the stream tma_ppo_stats_fold(staging_host, out8)
 * returns what tma_ppo_pop_stats would have. */
int64_t tma_ppo_stats_staging_bytes(void);
int tma_ppo_stats_enqueue(void *workspace, void *staging_host, void *stream);
int tma_ppo_stats_fold(const void *staging_host, double *out8_host);

/* ---- native rollout loop: SB3 OnPolicyAlgorithm.collect_rollouts driven by model.learn()
 *      (backend/mlagents/training.py:166-170; SURVEY.md §3.1 loop A) without a host round-trip per step ---- */
typedef struct {
    float *obs;            /* [T+1][N][D]; slot t is the observation acted on at step t, slot T the final new_obs */
    void *actions;         /* [T][N] i32 or [T][N][A] f32 */
    float *rewards;        /* [T][N], timeout bootstrap already applied */
    float *values;         /* [T][N] */
    float *log_probs;      /* [T][N] */
    uint8_t *terminated;   /* [T][N] */
    uint8_t *truncated;    /* [T][N] */
    float *terminal_obs;   /* [K][N][D] scratch (K = terminal_obs_slots): slot (t mod K) holds step t's pre-reset observations */
    float *last_values;    /* [N] */
    int64_t N;
    int terminal_obs_slots; /* K >= 1 (0 is read as 1).  With K > 1 the non-fused path computes the timeout bootstrap
                               rewards[t] += gamma * V(terminal_obs[t]) of K consecutive steps in ONE launch over K*N rows
                               instead of one small launch per vector step */
} tma_rollout_buffers;
/* One launch advances every env by many vector steps ("fused chunk") for: 64-wide f32 policies on GridWorld / Push / Ball3D / WallJump /
 * Bicycle / Glider; 256-wide policies, bf16 or f32, on the Discrete tasks with up to 32 observations and on the Box-action tasks (Crawler /
 * Ant shapes; f32: up to 4096 envs, round 6).  Every other shape runs policy forward + env step launch by launch -- the same results bit for bit
 * (tests/test_ppo_gpu.py::test_native_rollout_equals_stepwise_composition); TMA_NO_WIDE_FUSED=1 / TMA_NO_CONT_F32_FUSED=1 force that path;
 * TMA_ROLL2=1 runs the 64-wide fused chunk on two waves per 16-env tile (round 5's kernel) instead of four.
 * deterministic != 0: actions are the distribution's mode (first maximal logit / Gaussian mean) instead of samples -- what SB3's
 * evaluate_policy(deterministic=True) asks of the policy (backend/mlagents/training.py:177-184,240-247); evaluation.py runs whole evaluation
 * chunks through this entry point and reads the finished episodes from the env's episode log. */
int tma_rollout_collect(tma_env *env, const float *params, const tma_policy_dims *d, const tma_rollout_buffers *b, int t_begin, int t_end,
                        int T, uint32_t rng_seed, uint32_t rng_step0, uint32_t env_offset, double gamma, int compute_last_values,
                        int deterministic, void *stream);
/* GAE from done flags (episode_starts[t+1] == terminated[t] | truncated[t]): same arithmetic as tma_gae */
int tma_gae_flags(const float *rewards, const float *values, const uint8_t *terminated, const uint8_t *truncated,
                  const float *last_values, double gamma, double gae_lambda, int T, int64_t N, float *adv_out, float *ret_out,
                  void *stream);

#ifdef __cplusplus
}
#endif
#endif

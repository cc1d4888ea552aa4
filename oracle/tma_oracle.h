/*
 * tma_oracle.h -- CPU restatement of the three-mlagents vector-env hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (three-mlagents_amd/) never does.
 *
 * Parity status: the env layer (Basic, GridWorld, Ball3D, Push, WallJump, Bicycle, BrickBreak, Glider, the
 * Gymnasium adapter rule, the DummyVecEnv/Monitor bookkeeping) is PINNED, bit for bit, against golden vectors
 * produced by importing the reference's own files (tools/gen_golden.py -> tests/golden/*.npz; the three
 * float64 tasks with numpy's libm code paths, see that script's docstring).  GAE follows the published
 * stable-baselines3 2.9.0 algorithm (third-party, not vendored in the reference; pin
 * backend/uv.lock:1686-1687): "parity unpinned" for that function.  The Crawler-shape env is
 * build-defined (the reference delegates to MuJoCo Ant-v5, backend/mlagents/envs.py:274-277):
 * "parity unpinned".
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef TMA_ORACLE_H
#define TMA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BASIC = 0, ORC_GRIDWORLD = 1, ORC_BALL3D = 2, ORC_PUSH = 3, ORC_CRAWLER = 4, ORC_WALLJUMP = 5, ORC_BICYCLE = 6, ORC_BRICKBREAK = 7, ORC_GLIDER = 8, ORC_ANT = 9 };

#define ORC_MAX_STATE 80   /* upper bound on doubles per env in the flat state vector */
#define ORC_EP_STRIDE (1u << 20)

/* ---- numpy legacy global RNG (MT19937), SURVEY.md Appendix B ---- */
typedef struct {
    uint32_t mt[624];
    int idx;
} orc_mt;
void orc_mt_seed(orc_mt *s, uint32_t seed);          /* np.random.seed(int) == init_genrand */
uint32_t orc_mt_u32(orc_mt *s);                      /* one tempered 32-bit output */
uint32_t orc_mt_interval(orc_mt *s, uint32_t max);   /* legacy rk_interval: masked rejection */
double orc_mt_double(orc_mt *s);                     /* legacy rk_double (53 bit) */
void orc_mt_raw(uint32_t seed, uint32_t *out, int n);
void orc_mt_shuffle(uint32_t seed, int32_t *perm, int n, int32_t *next_interval_max_inout);
void orc_mt_uniform(uint32_t seed, double lo, double hi, double *out, int n);

/* ---- per-task metadata (backend/mlagents/envs.py:162-199) ---- */
int orc_obs_dim(int task);
int orc_num_actions(int task);      /* 0 for a Box action space */
int orc_act_dim(int task);          /* Box action dim (crawler), else 1 */
int orc_state_dim(int task);        /* doubles in the flat state vector */
int orc_max_episode_steps(int task);

/* episode seed contract: s(i,k) = base + i + k*2^20 (mod 2^32) */
uint32_t orc_episode_seed(uint32_t base, uint32_t env_index, uint32_t episode);
/* counter-based action tape a(i,t) = mix32(seed,i,t) % n */
uint32_t orc_mix32(uint32_t seed, uint32_t i, uint32_t t);
void orc_action_tape(uint32_t tape_seed, int n_envs, int env_offset, int T, int n_actions, int32_t *out /*[T][n_envs]*/);

/* adapter.reset(seed=s): seed, ctor reset, explicit reset (envs.py:110-123). */
void orc_reset_from_seed(int task, uint32_t seed, double *state_out, float *obs_out);
/* the legacy env's step() below the adapter: (obs, reward, done).  actions: int for discrete tasks */
void orc_legacy_step(int task, double *state_inout, const void *action, float *obs_out, double *reward_out, int *done_out);

/* ---- vector env: DummyVecEnv + Monitor + adapter semantics ---- */
typedef struct orc_vec orc_vec;
orc_vec *orc_vec_create(int task, int n_envs, uint32_t seed_base, uint32_t env_offset);
void orc_vec_destroy(orc_vec *v);
void orc_vec_set_threads(orc_vec *v, int n_threads);
void orc_vec_reset(orc_vec *v, float *obs_out /*[n][D]*/);
/* actions: int32[n] (discrete) or float[n][A] (crawler).  Any output pointer may be NULL. */
void orc_vec_step(orc_vec *v, const void *actions, float *obs_out, float *rew32_out, double *rew64_out,
                  uint8_t *term_out, uint8_t *trunc_out, float *term_obs_out, double *ep_ret_out,
                  int32_t *ep_len_out);
void orc_vec_get_state(const orc_vec *v, double *state_out /*[n][state_dim]*/);
void orc_vec_set_state(orc_vec *v, const double *state_in);
void orc_vec_episode_index(const orc_vec *v, uint32_t *out /*[n]*/);

/* ---- GAE: stable-baselines3 2.9.0 RolloutBuffer.compute_returns_and_advantage (3P) ---- */
void orc_gae(const float *rewards, const float *values, const float *episode_starts, const float *last_values,
             const uint8_t *dones, float gamma, float gae_lambda_times_gamma, int T, int N, float *adv_out,
             float *ret_out);
/* the same chains over env blocks on `threads` OpenMP threads (bit-identical; bench.py's all-cores CPU leg) */
void orc_gae_threads(const float *rewards, const float *values, const float *episode_starts, const float *last_values,
                     const uint8_t *dones, float gamma, float gae_lambda_times_gamma, int T, int N, float *adv_out,
                     float *ret_out, int threads);

#ifdef __cplusplus
}
#endif
#endif

// tma_rollout.hip -- native rollout driver: the inner loop of SB3 OnPolicyAlgorithm.collect_rollouts without a Python
// round-trip per vector step.  Per step t it enqueues, on one HIP stream:
//   policy_act(obs[t]) -> actions[t], values[t], log_probs[t]
//   env step(actions[t]) -> obs[t+1], rewards[t], terminated[t], truncated[t], terminal_obs   (+ reset-ring refill when due)
//   timeout bootstrap: rewards[t] += gamma * V(terminal_obs) where truncated[t]   (folded into step t+1's policy launch)
// and, after the last step, last_values = V(obs[T]).  Reference call path: model.learn() at
// /root/reference/backend/mlagents/training.py:166-170 -> SB3 collect_rollouts (SURVEY.md §3.1 hot loop A, App. C.6).
#include "tma_common.h"

extern "C" int tma_rollout_collect(tma_env *env, const float *params, const tma_policy_dims *d, const tma_rollout_buffers *b, int t_begin,
                                   int t_end, int T, uint32_t rng_seed, uint32_t rng_step0, uint32_t env_offset, double gamma,
                                   int compute_last_values, void *stream) {
    using namespace tma;
    if (!env || !params || !d || !b) return fail(TMA_ERR_INVALID, "tma_rollout_collect: null argument");
    if (!b->obs || !b->actions || !b->rewards || !b->values || !b->log_probs || !b->terminated || !b->truncated || !b->terminal_obs)
        return fail(TMA_ERR_INVALID, "tma_rollout_collect: rollout buffers has a null plane");
    if (t_begin < 0 || t_end > T || t_begin > t_end) return fail(TMA_ERR_INVALID, "bad step range [%d, %d) for T=%d", t_begin, t_end, T);
    const int64_t N = b->N;
    const int D = d->obs_dim, A = d->continuous ? d->act_dim : 1;
    const size_t act_elem = d->continuous ? sizeof(float) : sizeof(int32_t);
    for (int t = t_begin; t < t_end; t++) {
        const float *obs_t = b->obs + (int64_t)t * N * D;
        float *obs_next = b->obs + (int64_t)(t + 1) * N * D;
        void *act_t = static_cast<char *>(b->actions) + (int64_t)t * N * A * act_elem;
        // policy forward of step t; the timeout bootstrap of step t-1 (terminal_obs still holds step t-1's) rides in the same launch
        int rc = tma_policy_act_bootstrap(params, d, obs_t, N, rng_seed, rng_step0 + (uint32_t)t, env_offset, act_t, b->values + (int64_t)t * N,
                                          b->log_probs + (int64_t)t * N, t > 0 ? b->terminal_obs : nullptr,
                                          t > 0 ? b->truncated + (int64_t)(t - 1) * N : nullptr, gamma,
                                          t > 0 ? b->rewards + (int64_t)(t - 1) * N : nullptr, stream);
        if (rc) return rc;
        rc = tma_env_step(env, act_t, d->continuous ? TMA_ACT_F32 : TMA_ACT_I32, 0, 0, 1, obs_next, b->rewards + (int64_t)t * N,
                          b->terminated + (int64_t)t * N, b->truncated + (int64_t)t * N, b->terminal_obs, nullptr, nullptr, stream);
        if (rc) return rc;
        if (t == T - 1) {  // last step of the rollout: nothing follows to carry its bootstrap
            rc = tma_policy_bootstrap(params, d, b->terminal_obs, b->truncated + (int64_t)t * N, N, gamma, b->rewards + (int64_t)t * N, stream);
            if (rc) return rc;
        }
    }
    if (compute_last_values && t_end == T) {
        if (!b->last_values) return fail(TMA_ERR_INVALID, "last_values is null");
        int rc = tma_policy_values(params, d, b->obs + (int64_t)T * N * D, N, b->last_values, stream);
        if (rc) return rc;
    }
    return TMA_OK;
}

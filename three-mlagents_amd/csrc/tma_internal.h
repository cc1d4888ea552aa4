// tma_internal.h -- definitions shared by the translation units of libtma_hip.so (not part of the C ABI).
#pragma once
#include "tma_tasks.h"

namespace tma {

struct EnvView {
    int64_t N;
    int D;  // ring depth
    uint32_t seed_base, env_offset;
    uint32_t *st, *ring, *cur_ep, *filled_hi;
    double *ep_ret, *stats;
    // optional episode log (tma_env_episode_log): one record per finished episode, appended in execution order
    double *log_ret;            // [log_cap] Monitor return: the f64 sum of the episode's rewards in step order (SB3 Monitor sums Python floats)
    int32_t *log_len, *log_env; // [log_cap] episode length, env index within the vector
    unsigned long long *log_n;  // episodes seen since the last pop (records beyond log_cap are counted, not stored)
    long long log_cap;
};

// Monitor record of a finished episode (SB3 Monitor's `r`, `l`; reference training.py:85-86 wraps every env in one)
__device__ __forceinline__ void log_episode(const EnvView &v, int64_t env, double ep_return, int steps) {
    if (v.log_n == nullptr) return;
    const unsigned long long k = atomicAdd(v.log_n, 1ull);
    if (k < (unsigned long long)v.log_cap) {
        v.log_ret[k] = ep_return;
        v.log_len[k] = (int32_t)steps;
        v.log_env[k] = (int32_t)env;
    }
}

template <int OBS>
__device__ __forceinline__ void store_obs(float *dst, const float *o) {
    if constexpr (OBS == 4) {
        *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else if constexpr (OBS % 2 == 0) {
#pragma unroll
        for (int k = 0; k < OBS; k += 2) *reinterpret_cast<float2 *>(dst + k) = make_float2(o[k], o[k + 1]);
    } else {
#pragma unroll
        for (int k = 0; k < OBS; k++) dst[k] = o[k];
    }
}

// obs of state s -> dst row (wide obs are written straight from the task, narrow ones staged in registers)
template <class T>
__device__ __forceinline__ void emit_obs(const typename T::S &s, float *dst) {
    if constexpr (T::OBS > 32) {
        T::obs(s, dst);
    } else {
        float o[T::OBS];
        T::obs(s, o);
        store_obs<T::OBS>(dst, o);
    }
}

struct RefillView {
    uint32_t *first_ep;   // [N] first episode index each env must (re)fill
    uint32_t *env_off;    // [N] exclusive prefix of the per-env item counts inside the env's 256-block
    uint32_t *block_sum;  // [nb]
    uint32_t *block_off;  // [nb] exclusive prefix of block_sum
    uint32_t *total;      // [0] = number of items of this refill, [1] = fallback count
    uint32_t *fb_env, *fb_ep;
    int nb, fb_cap;
};

}  // namespace tma

constexpr int FB_BLOCKS = 16;      // fallback kernel grid (16 x 256 threads, 2.5 KB of MT19937 scratch each)
constexpr int FB_CAP = 1 << 20;    // smallest fallback-list capacity = items per refill round (tma_env_create scales it with N * ring_depth)

struct tma_env {
    int task, device;
    tma::EnvView v;
    tma::RefillView rv;
    uint32_t *mt_scratch;
    int steps_since_refill;
    bool is_reset;
    bool small_window;  // test hook: use the short fast-path window so the fallback generator is exercised
    // second set of Monitor aggregates / episode-log buffers (tma_env_detach_episode_log): what the kernels launched before the detach wrote
    double *d_stats = nullptr, *d_log_ret = nullptr;
    int32_t *d_log_len = nullptr, *d_log_env = nullptr;
    unsigned long long *d_log_n = nullptr;
    bool detached = false;  // the detached set holds data that has not been popped yet
    double *rew64_out = nullptr;  // tma_env_set_reward64: caller-owned [n_steps][N] plane tma_env_step also writes the f64 reward into
    int64_t rew64_cap = 0;        // ... and the doubles it holds (a step call that would write beyond it is refused)
    // side stream + events of the policy-only fused rollout (tma_rollout_collect): the batched value / bootstrap launches of chunk c run beside
    // the chunk kernel of chunk c + 1
    hipStream_t side = nullptr;
    hipEvent_t ev_chunk = nullptr, ev_side[2] = {nullptr, nullptr};
};

template <class F>
static int dispatch_task(int task, F &&f) {
    switch (task) {
    case TMA_TASK_BASIC: return f(tma::BasicTask{});
    case TMA_TASK_GRIDWORLD: return f(tma::GridTask{});
    case TMA_TASK_BALL3D: return f(tma::BallTask{});
    case TMA_TASK_PUSH: return f(tma::PushTask{});
    case TMA_TASK_CRAWLER: return f(tma::CrawlerTask{});
    case TMA_TASK_WALLJUMP: return f(tma::WallJumpTask{});
    case TMA_TASK_BICYCLE: return f(tma::BicycleTask{});
    case TMA_TASK_BRICKBREAK: return f(tma::BrickBreakTask{});
    case TMA_TASK_GLIDER: return f(tma::GliderTask{});
    case TMA_TASK_ANT: return f(tma::AntTask{});
    }
    return tma::fail(TMA_ERR_UNKNOWN_TASK, "unknown task id %d", task);
}


// bookkeeping after a kernel outside tma_env.hip advanced the envs by n_steps (triggers the reset-ring refill when due)
int tma_env_internal_after_steps(tma_env *h, int n_steps, void *stream);

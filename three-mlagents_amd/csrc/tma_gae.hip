// tma_gae.hip -- GAE(lambda) advantages + returns over a (T, N) rollout, one thread per env, reverse scan over T.
//
// Replaces stable-baselines3 2.9.0 RolloutBuffer.compute_returns_and_advantage (third-party; the buffer is built by
// PPO(...) at /root/reference/backend/mlagents/training.py:150 with gamma/gae_lambda from training.py:383-384).
// The float32 operation order is SB3's so results are bit-identical to the NumPy loop:
//   delta = r_t + gamma * V_{t+1} * nnt - V_t ;  A_t = delta + (gamma*lambda) * nnt * A_{t+1} ;  ret = A + V.
// HBM traffic: 12 B read + 8 B written per (t, env); rows of 64 consecutive envs are coalesced at every t.
// The recurrence is sequential in t, so loads are software-pipelined UNROLL rows ahead of the dependent math.
#include "tma_common.h"

namespace tma {

constexpr int GAE_UNROLL = 16;  // rows of loads in flight ahead of the dependent chain (3 arrays x 16 rows per lane)

// FLAGS=false: SB3 layout (float episode_starts[T][N] + final dones[N]).  FLAGS=true: engine layout, done flags
// terminated/truncated[T][N] where episode_starts[t+1] == done[t], so next_non_terminal at step t is 1 - done[t].
template <bool FLAGS>
__global__ __launch_bounds__(256) void gae_kernel(const float *__restrict__ rewards, const float *__restrict__ values,
                                                  const float *__restrict__ episode_starts, const float *__restrict__ last_values,
                                                  const uint8_t *__restrict__ dones, const uint8_t *__restrict__ term,
                                                  const uint8_t *__restrict__ trunc, float gamma, float gl, int T, int64_t N,
                                                  float *__restrict__ adv, float *__restrict__ ret) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float last = 0.0f;
    float next_v = last_values[i];
    float next_nnt;
    if constexpr (FLAGS) next_nnt = 1.0f - ((term[(int64_t)(T - 1) * N + i] | trunc[(int64_t)(T - 1) * N + i]) ? 1.0f : 0.0f);
    else next_nnt = 1.0f - (dones[i] ? 1.0f : 0.0f);
    int t = T - 1;
    for (; t >= GAE_UNROLL - 1; t -= GAE_UNROLL) {
        float r[GAE_UNROLL], v[GAE_UNROLL], es[GAE_UNROLL];
#pragma unroll
        for (int u = 0; u < GAE_UNROLL; u++) {
            const int64_t off = (int64_t)(t - u) * N + i;
            r[u] = rewards[off];
            v[u] = values[off];
            if constexpr (FLAGS) {  // flag of the PREVIOUS step is this step's episode_start
                const int tp = t - u - 1;
                es[u] = tp >= 0 ? ((term[(int64_t)tp * N + i] | trunc[(int64_t)tp * N + i]) ? 1.0f : 0.0f) : 0.0f;
            } else {
                es[u] = episode_starts[off];
            }
        }
#pragma unroll
        for (int u = 0; u < GAE_UNROLL; u++) {
            const int64_t off = (int64_t)(t - u) * N + i;
            float a = gamma * next_v;
            a = a * next_nnt;
            float delta = r[u] + a;
            delta = delta - v[u];
            float b = gl * next_nnt;
            b = b * last;
            last = delta + b;
            adv[off] = last;
            ret[off] = last + v[u];
            next_v = v[u];
            next_nnt = 1.0f - es[u];
        }
    }
    for (; t >= 0; t--) {
        const int64_t off = (int64_t)t * N + i;
        const float rr = rewards[off], vv = values[off];
        float ee;
        if constexpr (FLAGS) ee = t >= 1 ? ((term[(int64_t)(t - 1) * N + i] | trunc[(int64_t)(t - 1) * N + i]) ? 1.0f : 0.0f) : 0.0f;
        else ee = episode_starts[off];
        float a = gamma * next_v;
        a = a * next_nnt;
        float delta = rr + a;
        delta = delta - vv;
        float b = gl * next_nnt;
        b = b * last;
        last = delta + b;
        adv[off] = last;
        ret[off] = last + vv;
        next_v = vv;
        next_nnt = 1.0f - ee;
    }
}

// The recurrence over t is evaluated exactly as SB3 does (one rounded f32 chain per env): a segment-parallel scan would compose the
// affine maps A_t = delta_t + c_t * A_{t+1} in a different rounding order and lose bit-exactness, so the parallelism is over envs only.
// Up to 16 384 envs run as single-wave blocks -- 4096 envs then sit on 64 CUs (one wave each, every load of the 16-row window in flight)
// instead of on 16.
static inline int gae_block(int64_t N) { return N <= 16384 ? 64 : 256; }

}  // namespace tma

extern "C" int tma_gae(const float *rewards, const float *values, const float *episode_starts, const float *last_values,
                       const uint8_t *dones, double gamma, double gae_lambda, int T, int64_t N, float *adv_out, float *ret_out,
                       void *stream) {
    using namespace tma;
    if (!rewards || !values || !episode_starts || !last_values || !dones || !adv_out || !ret_out)
        return fail(TMA_ERR_INVALID, "tma_gae: null buffer");
    if (T < 1 || N < 1) return fail(TMA_ERR_INVALID, "tma_gae: T and N must be >= 1 (got T=%d N=%lld)", T, (long long)N);
    // SB3 multiplies the python floats gamma*gae_lambda in float64, then the product meets the float32 arrays
    const float gl = (float)(gamma * gae_lambda);
    const int bs = gae_block(N);
    gae_kernel<false><<<dim3((unsigned)ceil_div(N, bs)), dim3(bs), 0, (hipStream_t)stream>>>(
        rewards, values, episode_starts, last_values, dones, nullptr, nullptr, (float)gamma, gl, T, N, adv_out, ret_out);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

extern "C" int tma_gae_flags(const float *rewards, const float *values, const uint8_t *terminated, const uint8_t *truncated,
                             const float *last_values, double gamma, double gae_lambda, int T, int64_t N, float *adv_out, float *ret_out,
                             void *stream) {
    using namespace tma;
    if (!rewards || !values || !terminated || !truncated || !last_values || !adv_out || !ret_out)
        return fail(TMA_ERR_INVALID, "tma_gae_flags: null buffer");
    if (T < 1 || N < 1) return fail(TMA_ERR_INVALID, "tma_gae_flags: T and N must be >= 1 (got T=%d N=%lld)", T, (long long)N);
    const float gl = (float)(gamma * gae_lambda);
    const int bs = gae_block(N);
    gae_kernel<true><<<dim3((unsigned)ceil_div(N, bs)), dim3(bs), 0, (hipStream_t)stream>>>(
        rewards, values, nullptr, last_values, nullptr, terminated, truncated, (float)gamma, gl, T, N, adv_out, ret_out);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

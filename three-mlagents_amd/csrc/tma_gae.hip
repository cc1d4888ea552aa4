// tma_gae.hip -- GAE(lambda) advantages + returns over a (T, N) rollout: a producer / consumer kernel for vectors of up to 65 536 envs
// (gae_pc_kernel below) and the one-thread-per-env reverse scan over T for larger ones (gae_kernel).
//
// Replaces stable-baselines3 2.9.0 RolloutBuffer.compute_returns_and_advantage (third-party; the buffer is built by
// PPO(...) at /root/reference/backend/mlagents/training.py:150 with gamma/gae_lambda from training.py:383-384).
// The float32 operation order is SB3's so results are bit-identical to the NumPy loop:
//   delta = r_t + gamma * V_{t+1} * nnt - V_t ;  A_t = delta + (gamma*lambda) * nnt * A_{t+1} ;  ret = A + V.
// HBM traffic: 12 B read + 8 B written per (t, env); rows of 64 consecutive envs are coalesced at every t.
// The recurrence is sequential in t, so loads are software-pipelined UNROLL rows ahead of the dependent math.
#include "tma_common.h"

#include <cstdlib>

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

// ------------------------------------------------------------------------------------------
// Small vectors (N <= 65 536: every BASELINE config): producer / consumer form of the same arithmetic.  gae_kernel above is bound by one
// memory round trip per 16-row window (64 windows x ~1.5 us at T = 1024, whatever the number of waves: DESIGN.md section 10), and nothing in
// delta_t = (r_t + (gamma * V_{t+1}) * nnt_t) - V_t and c_t = (gamma * lambda) * nnt_t depends on the recurrence.  So a workgroup owns 16
// envs: waves 1..3 stream rows of chunk k + 1 from memory, form (delta, c, V) with exactly the operations of the loop above and leave them
// in LDS, while wave 0 runs the chain A_t = delta_t + c_t * A_{t+1} of chunk k out of LDS -- two dependent operations per step, never a
// wait on memory -- and leaves A_t where delta_t was; the producer waves write advantages and returns (A_t + V_t) to memory one chunk
// later, right before they refill that buffer.  One workgroup barrier per 128-step chunk.  Same operations in the same order per
// element: bit-identical to gae_kernel (and to the NumPy loop).
// ------------------------------------------------------------------------------------------
constexpr int GP_ENVS = 16, GP_TC = 128, GP_PROD = 448;  // envs per workgroup, steps per chunk, producer threads (7 waves + the chain wave)

template <bool FLAGS>
__global__ __launch_bounds__(GP_PROD + 64) void gae_pc_kernel(const float *__restrict__ rewards, const float *__restrict__ values,
                                                     const float *__restrict__ episode_starts, const float *__restrict__ last_values,
                                                     const uint8_t *__restrict__ dones, const uint8_t *__restrict__ term,
                                                     const uint8_t *__restrict__ trunc, float gamma, float gl, int T, int64_t N,
                                                     float *__restrict__ adv, float *__restrict__ ret) {
    // per buffer: delta (overwritten by the advantage once the chain has passed), c, V
    // [env][step] rows of TC + 4 floats: the chain lane of an env reads / writes FOUR consecutive steps per ds_read_b128 / ds_write_b128 (lanes
    // are 4 banks apart: conflict-free), and the producers' scalar accesses (16 envs x 4 consecutive steps per wave instruction) hit 64 banks
    constexpr int LDR = GP_TC + 4;
    __shared__ __attribute__((aligned(16))) float sD[2][GP_ENVS][LDR], sC[2][GP_ENVS][LDR], sV[2][GP_ENVS][LDR];
    // workgroups are dealt to the 8 XCDs round-robin; XCD x takes a CONTIGUOUS range of env groups, so that the two 64-byte halves of a
    // 128-byte line (envs 32 g .. 32 g + 31 of a row) are fetched into ONE L2 instead of two
    const int per_xcd = gridDim.x >> 3;
    const int64_t grp = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t i0 = grp * GP_ENVS;
    if (i0 >= N) return;  // (whole workgroup: the grid is rounded up to a multiple of 8)
    const int tid = threadIdx.x, wave = tid >> 6;
    const int n_chunks = (T + GP_TC - 1) / GP_TC;
    constexpr int PER = (GP_TC * GP_ENVS + GP_PROD - 1) / GP_PROD;
    const int p = tid - 64;  // producer index (waves 1..3)
    // chunk k covers t = t_hi(k) down to max(t_hi(k) - TC + 1, 0); LDS row j of the chunk is step t_hi(k) - j.  Producer thread p owns the
    // slots e = p + GP_PROD j of EVERY chunk: it fills them, and it is the one that later writes the advantages / returns found there to memory
    // (so a buffer can be refilled for chunk k + 2 right behind the write-out of chunk k without a barrier in between).
    // step 1 of a refill: every load of the chunk in flight
    auto fetch = [&](int k, float (&r)[PER], float (&v)[PER], float (&vn)[PER], float (&nn)[PER]) {
        const int t_hi = T - 1 - k * GP_TC;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int e = p + GP_PROD * j, row = e / GP_ENVS, env = e % GP_ENVS, t = t_hi - row;
            // (branch-free: out-of-range slots read element 0 and are never used, so every load of the chunk is issued back to back)
            const bool ok = e < GP_TC * GP_ENVS && t >= 0 && i0 + env < N;
            const int64_t off = ok ? (int64_t)t * N + i0 + env : 0;
            const bool top = ok && t == T - 1;
            r[j] = rewards[off];
            v[j] = values[off];
            vn[j] = *(top ? last_values + i0 + env : values + (ok ? off + N : 0));
            float done;
            if constexpr (FLAGS) done = (term[off] | trunc[off]) ? 1.0f : 0.0f;                    // next_non_terminal of step t = 1 - done[t]
            else done = top ? (dones[i0 + env] ? 1.0f : 0.0f) : episode_starts[ok ? off + N : 0];  // = 1 - episode_starts[t + 1]
            nn[j] = 1.0f - done;
        }
    };
    // step 2: advantages / returns of chunk k (left in the buffer by the chain wave) to memory: ret = A + V, the loop's own addition
    auto write_out = [&](int k) {
        const int t_hi = T - 1 - k * GP_TC;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int e = p + GP_PROD * j, row = e / GP_ENVS, env = e % GP_ENVS, t = t_hi - row;
            if (e < GP_TC * GP_ENVS && t >= 0 && i0 + env < N) {
                const int64_t off = (int64_t)t * N + i0 + env;
                const float a = sD[k & 1][env][row];
                adv[off] = a;
                ret[off] = a + sV[k & 1][env][row];
            }
        }
    };
    // step 3: (delta, c, V) of the fetched chunk into its buffer
    auto fill = [&](int k, const float (&r)[PER], const float (&v)[PER], const float (&vn)[PER], const float (&nn)[PER]) {
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int e = p + GP_PROD * j, row = e / GP_ENVS, env = e % GP_ENVS;
            if (e < GP_TC * GP_ENVS) {
                float a = gamma * vn[j];
                a = a * nn[j];
                float delta = r[j] + a;
                delta = delta - v[j];
                sD[k & 1][env][row] = delta;
                sC[k & 1][env][row] = gl * nn[j];
                sV[k & 1][env][row] = v[j];
            }
        }
    };
    if (wave > 0) {
        float r[PER], v[PER], vn[PER], nn[PER];
        fetch(0, r, v, vn, nn);
        fill(0, r, v, vn, nn);
    }
    __syncthreads();
    float last = 0.0f;
    const bool chain_lane = tid < GP_ENVS && i0 + tid < N;
#ifndef TMA_GAE_NO_PRIO
    if (wave == 0) __builtin_amdgcn_s_setprio(3);  // the chain wave shares its SIMD with a producer wave: its dependent mul / add pairs go first
#endif
    for (int k = 0; k < n_chunks; k++) {
        if (wave > 0) {
            float r[PER], v[PER], vn[PER], nn[PER];
            const bool more = k + 1 < n_chunks;
            if (more) fetch(k + 1, r, v, vn, nn);
            if (k >= 1) write_out(k - 1);  // (buffer (k - 1) & 1 == (k + 1) & 1: emptied and refilled by the same threads, slot for slot)
            if (more) fill(k + 1, r, v, vn, nn);
        } else if (chain_lane) {
            const int t_hi = T - 1 - k * GP_TC;
            const int rows = t_hi + 1 < GP_TC ? t_hi + 1 : GP_TC;
            float *bd = &sD[k & 1][tid][0];
            const float *bc = &sC[k & 1][tid][0];
            typedef float f4 __attribute__((ext_vector_type(4)));
            int j = 0;
            // operands of 8 steps read ahead of the 8 steps being chained: an LDS latency per 8 steps, not in front of them
            f4 d0, d1, c0, c1;
            if (rows >= 8) {
                d0 = *reinterpret_cast<const f4 *>(bd), d1 = *reinterpret_cast<const f4 *>(bd + 4);
                c0 = *reinterpret_cast<const f4 *>(bc), c1 = *reinterpret_cast<const f4 *>(bc + 4);
            }
            for (; j + 8 <= rows; j += 8) {
                f4 dn0, dn1, cn0, cn1;
                const bool ahead = j + 16 <= rows;
                if (ahead) {
                    dn0 = *reinterpret_cast<const f4 *>(bd + j + 8), dn1 = *reinterpret_cast<const f4 *>(bd + j + 12);
                    cn0 = *reinterpret_cast<const f4 *>(bc + j + 8), cn1 = *reinterpret_cast<const f4 *>(bc + j + 12);
                }
                f4 a0, a1;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float b = c0[u] * last;
                    last = d0[u] + b;
                    a0[u] = last;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float b = c1[u] * last;
                    last = d1[u] + b;
                    a1[u] = last;
                }
                *reinterpret_cast<f4 *>(bd + j) = a0;
                *reinterpret_cast<f4 *>(bd + j + 4) = a1;
                if (ahead) d0 = dn0, d1 = dn1, c0 = cn0, c1 = cn1;
            }
            for (; j < rows; j++) {
                const float b = bc[j] * last;
                last = bd[j] + b;
                bd[j] = last;
            }
        }
        __syncthreads();
    }
    if (wave > 0) write_out(n_chunks - 1);
}

// The recurrence over t is evaluated exactly as SB3 does (one rounded f32 chain per env): a segment-parallel scan would compose the
// affine maps A_t = delta_t + c_t * A_{t+1} in a different rounding order and lose bit-exactness, so the parallelism is over envs only.
// Up to 16 384 envs run as single-wave blocks -- 4096 envs then sit on 64 CUs (one wave each, every load of the 16-row window in flight)
// instead of on 16.
static inline int gae_block(int64_t N) { return N <= 16384 ? 64 : 256; }
// producer / consumer kernel: while 16-env workgroups do not outnumber what the chip holds at once (TMA_GAE_SCAN=1: the one-wave kernel, for A/B timing)
static inline bool gae_small(int64_t N) { return N <= 65536 && getenv("TMA_GAE_SCAN") == nullptr; }

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
    if (gae_small(N))
        gae_pc_kernel<false><<<dim3((unsigned)(8 * ceil_div(ceil_div(N, GP_ENVS), 8))), dim3(GP_PROD + 64), 0, (hipStream_t)stream>>>(
            rewards, values, episode_starts, last_values, dones, nullptr, nullptr, (float)gamma, gl, T, N, adv_out, ret_out);
    else
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
    if (gae_small(N))
        gae_pc_kernel<true><<<dim3((unsigned)(8 * ceil_div(ceil_div(N, GP_ENVS), 8))), dim3(GP_PROD + 64), 0, (hipStream_t)stream>>>(
            rewards, values, nullptr, last_values, nullptr, terminated, truncated, (float)gamma, gl, T, N, adv_out, ret_out);
    else
        gae_kernel<true><<<dim3((unsigned)ceil_div(N, bs)), dim3(bs), 0, (hipStream_t)stream>>>(
            rewards, values, nullptr, last_values, nullptr, terminated, truncated, (float)gamma, gl, T, N, adv_out, ret_out);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

// tma_policy.hip -- actor-critic MLP forward (act / values / timeout bootstrap), PPO clipped-surrogate
// forward+backward over a minibatch, global-norm clip + Adam.  gfx950, f32-input MFMA (exact f32).
//
// Replaces stable-baselines3 2.9.0 (third-party, pinned by /root/reference/backend/uv.lock:1686-1687) pieces that
// PPO("MlpPolicy", env, **kwargs) at /root/reference/backend/mlagents/training.py:150 constructs and
// model.learn() at training.py:166-170 drives:
//   ActorCriticPolicy.forward / predict_values / evaluate_actions / _predict      -> policy_act_kernel, values kernels
//   OnPolicyAlgorithm.collect_rollouts timeout bootstrap (rewards += gamma * V(terminal_obs))  -> bootstrap_kernel
//   RolloutBuffer.get (minibatch permutation + gather)                            -> perm_index + gather in ppo_grad_kernel
//   PPO.train loss (clipped surrogate, value MSE, entropy bonus) + autograd       -> ppo_grad_kernel
//   clip_grad_norm_(max_grad_norm) + torch.optim.Adam(eps=1e-5).step()            -> grad_sumsq_kernel + adam_kernel
// Formulas: SURVEY.md Appendix C.3 / C.5.  "Parity unpinned" at this boundary (SB3 is not importable here); checked
// against a torch-CPU autograd restatement in tests/.
#include "tma_h64_tile.h"
#include "tma_p2p.h"

#include <cstring>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace tma {


__device__ __forceinline__ void build_image_elem(float *params, const PLayout &L, int img, int e, int W1t, int b1, int W2t, int b2, int W3t, int b3,
                                                 int n_out) {
    const int D = L.D, H = 64;
    float v = 0.0f;
    if (e < IMG_W2F) {  // [k][r16][j] <- W1t[k][16j + r16]
        const int x = e - IMG_W1, k = x >> 6, r16 = (x >> 2) & 15, j = x & 3;
        v = k < D ? params[W1t + k * H + 16 * j + r16] : 0.0f;
    } else if (e < IMG_W3F) {
        const int x = e - IMG_W2F, k = x >> 6, r16 = (x >> 2) & 15, j = x & 3;
        v = params[W2t + k * H + 16 * j + r16];
    } else if (e < IMG_W3B) {  // [k][col]
        const int x = e - IMG_W3F, k = x >> 4, col = x & 15;
        v = col < n_out ? params[W3t + k * n_out + col] : 0.0f;
    } else if (e < IMG_W2B) {  // [n][r16][j] <- W3[n][k = 16j + r16] = W3t[k][n]
        const int x = e - IMG_W3B, n = x >> 6, r16 = (x >> 2) & 15, j = x & 3;
        v = n < n_out ? params[W3t + (16 * j + r16) * n_out + n] : 0.0f;
    } else if (e < IMG_B1) {   // [n][r16][j] <- W2[n][k] = W2t[k][n]
        const int x = e - IMG_W2B, n = x >> 6, r16 = (x >> 2) & 15, j = x & 3;
        v = params[W2t + (16 * j + r16) * H + n];
    } else if (e < IMG_B2) {
        v = params[b1 + (e - IMG_B1)];
    } else if (e < IMG_B3) {
        v = params[b2 + (e - IMG_B2)];
    } else {
        const int c = e - IMG_B3;
        v = c < n_out ? params[b3 + c] : 0.0f;
    }
    params[img + e] = v;
}

// f32 fragment-major images of W2 for the column-parallel f32 kernels (layout: PLayout::fr_pi)
__global__ void build_f32_frag_images_kernel(float *params, PLayout L) {
    const int H = L.H, NQ = H >> 4, per = H * H;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 4 * per; e += gridDim.x * blockDim.x) {
        const int net = e / (2 * per), x = e - net * 2 * per, bwd = x >= per, y = bwd ? x - per : x;
        const float *W2 = params + (net == 0 ? L.pW2t : L.vW2t);
        const int i = y & 3, l = (y >> 2) & 63, rest = y >> 8, q = rest % NQ, t = rest / NQ;
        const int a = 16 * q + 4 * i + (l >> 4), b = 16 * t + (l & 15);  // forward: (k, n) = (a, b); input-gradient: (k', n) = (b, a)
        params[(net == 0 ? L.fr_pi : L.fr_vf) + x] = bwd ? W2[b * H + a] : W2[a * H + b];
    }
    if (L.fr1_pi >= 0) {
        const int NQ1 = (L.D + 15) / 16, per1 = H * 16 * NQ1;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * per1; e += gridDim.x * blockDim.x) {
            const int net = e / per1, y = e - net * per1;
            const float *W1 = params + (net == 0 ? L.pW1t : L.vW1t);
            const int i = y & 3, l = (y >> 2) & 63, rest = y >> 8, q = rest % NQ1, t = rest / NQ1;
            const int k = 16 * q + 4 * i + (l >> 4), n = 16 * t + (l & 15);
            params[(net == 0 ? L.fr1_pi : L.fr1_vf) + y] = k < L.D ? W1[k * H + n] : 0.0f;
        }
    }
}

// refreshes everything derived from the trainable region: [out][in] copies and (H == 64 fast path) the LDS images
__global__ void sync_transposed_kernel(float *params, PLayout L) {
    const int H = L.H, A = L.A;
    const int n_copy = 2 * H * H + A * H + H;
    const int n_img = L.img_pi >= 0 ? 2 * IMG_FLOATS : 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_copy + n_img; e += gridDim.x * blockDim.x) {
        if (e >= n_copy) {
            const int x = e - n_copy;
            if (x < IMG_FLOATS) build_image_elem(params, L, L.img_pi, x, L.pW1t, L.pb1, L.pW2t, L.pb2, L.pW3t, L.pb3, A);
            else build_image_elem(params, L, L.img_vf, x - IMG_FLOATS, L.vW1t, L.vb1, L.vW2t, L.vb2, L.vW3t, L.vb3, 1);
            continue;
        }
        int x = e;
        if (x < H * H) {  // pW2[n][k] = pW2t[k][n]
            const int n = x / H, k = x % H;
            params[L.pW2 + x] = params[L.pW2t + k * H + n];
            continue;
        }
        x -= H * H;
        if (x < A * H) {
            const int n = x / H, k = x % H;
            params[L.pW3 + x] = params[L.pW3t + k * A + n];
            continue;
        }
        x -= A * H;
        if (x < H * H) {
            const int n = x / H, k = x % H;
            params[L.vW2 + x] = params[L.vW2t + k * H + n];
            continue;
        }
        x -= H * H;
        params[L.vW3 + x] = params[L.vW3t + x];
    }
}

// ------------------------------------------------------------------------------------------
// policy_act: obs[n][D] -> sampled (or deterministic) actions, values, log-probs.  One wave per 16 rows.
// MODE: 0 act (actions+values+logp), 1 values only, 2 timeout bootstrap (rewards[i] += gamma * V(term_obs[i]) where trunc[i])
// ------------------------------------------------------------------------------------------
template <bool CONT, int MODE>
__global__ __launch_bounds__(256) void policy_fwd_kernel(const float *__restrict__ params, PLayout L, const float *__restrict__ obs, int64_t n,
                                                         uint32_t rng_seed, uint32_t rng_step, uint32_t env_offset, int deterministic,
                                                         void *__restrict__ actions_out, float *__restrict__ values_out,
                                                         float *__restrict__ logp_out, const uint8_t *__restrict__ trunc, float gamma,
                                                         float *__restrict__ rewards) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int D = L.D, H = L.H, A = L.A;
    const int ldx = ((D + 3) & ~3) + 2, ld = H + 2;
    const int per_wave = 16 * (ldx + 2 * ld) + 32;
    float *X = smem + (int64_t)wave * per_wave;
    float *h1 = X + 16 * ldx, *h2 = h1 + 16 * ld;
    int64_t *row_off = reinterpret_cast<int64_t *>(h2 + 16 * ld);
    const int64_t n_tiles = (n + 15) >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * wpb + wave; tile < n_tiles; tile += (int64_t)gridDim.x * wpb) {
        const int64_t row0 = tile << 4;
        if constexpr (MODE == 2) {  // skip tiles without a truncated env
            const int64_t rr = row0 + r16;
            const bool t = (rr < n) && trunc[rr] != 0;
            if (__ballot(t) == 0ull) continue;
        }
        if (lane < 16) row_off[lane] = (row0 + lane < n) ? row0 + lane : -1;
        load_obs_tile(obs, row_off, D, X, ldx, lane);
        const Net V = vf_net(params, L);
        dense_tanh(X, ldx, D, V.W1t, V.b1, H, h1, ld, lane);
        dense_tanh(h1, ld, H, V.W2t, V.b2, H, h2, ld, lane);
        f32x4 vacc[1];
        dense_head<1>(h2, ld, H, V.W3t, V.b3, 1, vacc, lane);
        if constexpr (MODE == 1) {
            if (r16 == 0)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t row = row0 + g * 4 + r;
                    if (row < n) values_out[row] = vacc[0][r];
                }
            continue;
        }
        if constexpr (MODE == 2) {
            if (r16 == 0)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t row = row0 + g * 4 + r;
                    if (row < n && trunc[row]) {
                        const float gv = gamma * vacc[0][r];  // SB3: rewards[idx] += self.gamma * terminal_value (float32)
                        rewards[row] = rewards[row] + gv;
                    }
                }
            continue;
        }
        if constexpr (MODE == 0) {
            const Net P = pi_net(params, L);
            dense_tanh(X, ldx, D, P.W1t, P.b1, H, h1, ld, lane);
            dense_tanh(h1, ld, H, P.W2t, P.b2, H, h2, ld, lane);
            if constexpr (!CONT) {
                f32x4 acc[1];
                dense_head<1>(h2, ld, H, P.W3t, P.b3, A, acc, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t row = row0 + g * 4 + r;
                    const bool colok = r16 < A;
                    const float x = colok ? acc[0][r] : -INFINITY;
                    const float m = gmax16(x);
                    const float e = colok ? expf(x - m) : 0.0f;
                    const float s = gsum16(e);
                    const float lse = m + logf(s);
                    const float lp = x - lse;
                    int act;
                    if (deterministic) {
                        const float cand = (colok && x == m) ? (float)r16 : 99.0f;
                        float mn = cand;
                        mn = gmin16(mn);
                        act = (int)mn;
                    } else {
                        const float c = gscan16(e / s);
                        const uint32_t gi = env_offset + (uint32_t)row;
                        const float u = uniform01(mix32(rng_seed, gi, rng_step));
                        const float cnt = gsum16((colok && c <= u) ? 1.0f : 0.0f);
                        act = min((int)cnt, A - 1);
                    }
                    const float lpa = gsum16((r16 == act) ? lp : 0.0f);
                    const float vrow = gfirst_quad(vacc[0][r]);  // value sits in column 0 of the group
                    if (r16 == r && row < n) {
                        static_cast<int32_t *>(actions_out)[row] = act;
                        logp_out[row] = lpa;
                        values_out[row] = vrow;
                    }
                }
            } else {
                f32x4 acc[2];
                dense_head<2>(h2, ld, H, P.W3t, P.b3, A, acc, lane);
                const float *ls = params + L.log_std;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t row = row0 + g * 4 + r;
                    const uint32_t gi = env_offset + (uint32_t)row;
                    float lpsum = 0.0f;
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int col = 16 * j + r16;
                        if (col < A) {
                            const float mu = acc[j][r], lsd = ls[col], sd = expf(lsd);
                            float a = mu;
                            if (!deterministic) {
                                const float u1 = fmaxf(uniform01(mix32(rng_seed ^ (0x68E31DA4u + (uint32_t)col * 0x9E3779B9u), gi, rng_step)), 5.9604645e-08f);
                                const float u2 = uniform01(mix32(rng_seed ^ (0xB5297A4Du + (uint32_t)col * 0x85EBCA77u), gi, rng_step));
                                const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853071795865f * u2);
                                a = mu + sd * z;
                            }
                            const float d = a - mu;
                            lpsum += -(d * d) / (2.0f * (sd * sd)) - lsd - 0.9189385332046727f;
                            if (row < n) static_cast<float *>(actions_out)[row * A + col] = a;
                        }
                    }
                    lpsum = gsum16(lpsum);
                    const float vrow = gfirst_quad(vacc[0][r]);
                    if (r16 == r && row < n) {
                        logp_out[row] = lpsum;
                        values_out[row] = vrow;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// minibatch advantage statistics (mean, unbiased std) -- one block; SB3 PPO.train normalize_advantage
// ------------------------------------------------------------------------------------------
// pass 1: up to ADV_BLOCKS blocks, each sums a contiguous slice of the (permuted) minibatch -> (sum, sum of squares) partials
constexpr int ADV_BLOCKS = 128;
// gridDim.y > 1 (tma_ppo_epoch_prepare): blockIdx.y = minibatch k of an epoch split into chunks of `batch` rows; partials and
// offsets of minibatch k land at partials + 2 * k * gridDim.x and offs_out + k * batch.
__global__ __launch_bounds__(256) void adv_partial_kernel(const float *__restrict__ adv, Minibatch mb, int T, int64_t N, double *partials,
                                                          int32_t *__restrict__ offs_out, int64_t batch = 0) {
    __shared__ double s1[4], s2[4];
    if (gridDim.y > 1 || batch > 0) {
        const int64_t k = blockIdx.y, s0 = k * batch;
        const int64_t cnt = (s0 + batch <= mb.count) ? batch : mb.count - s0;
        mb.start += s0, mb.count = cnt;
        partials += 2 * k * gridDim.x;
        if (offs_out) offs_out += s0;
    }
    int nb = (int)((mb.count + 1023) / 1024);  // partial blocks this minibatch uses: the same split as a stand-alone launch
    if (nb > (int)gridDim.x) nb = gridDim.x;
    if ((int)blockIdx.x >= nb) return;
    const int64_t per = (mb.count + nb - 1) / nb;
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = (j0 + per < mb.count) ? j0 + per : mb.count;
    double a = 0.0, b = 0.0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) {
        const int64_t off = sample_offset(mb, mb.start + j, T, N);
        if (offs_out) offs_out[j] = (int32_t)off;
        const double x = (double)adv[off];
        a += x;
        b += x * x;
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    if ((threadIdx.x & 63) == 0) s1[threadIdx.x >> 6] = a, s2[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        partials[2 * blockIdx.x + 1] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    }
}
// pass 2: one wave folds the partials in a fixed order -> mean, unbiased std (SB3: advantages.std() + 1e-8)
__global__ void adv_final_kernel(const double *partials, int n_part, int64_t count, float *ws_adv) {
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < n_part; k += 64) a += partials[2 * k], b += partials[2 * k + 1];
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    if (threadIdx.x == 0) {
        const double n = (double)count, mean = a / n;
        double var = n > 1.0 ? (b - n * mean * mean) / (n - 1.0) : 0.0;
        if (var < 0.0) var = 0.0;
        ws_adv[0] = (float)mean;
        ws_adv[1] = (float)sqrt(var);
    }
}

// data-parallel epochs (tma_ppo_epoch_adv_sums): one wave per minibatch k.  EXPORT folds the minibatch's partials (the slots
// adv_partial_kernel filled, same order as adv_final_kernel) into sums[k] = {sum, sumsq}; IMPORT writes the all-reduced pair back as
// partial 0 and zeroes the minibatch's other slots, so every consumer of the partials sees the global sums.
template <bool IMPORT>
__global__ __launch_bounds__(64) void adv_epoch_sums_kernel(double *partials, int stride, int64_t batch, int64_t total, double *sums) {
    const int64_t k = blockIdx.x, s0 = k * batch;
    const int64_t cnt = (s0 + batch <= total) ? batch : total - s0;
    int nb = (int)((cnt + 1023) / 1024);
    if (nb > stride) nb = stride;
    double *part = partials + 2 * k * stride;
    if constexpr (IMPORT) {
        for (int j = threadIdx.x; j < nb; j += 64) {
            part[2 * j] = j == 0 ? sums[2 * k] : 0.0;
            part[2 * j + 1] = j == 0 ? sums[2 * k + 1] : 0.0;
        }
    } else {
        double a = 0.0, b = 0.0;
        for (int j = threadIdx.x; j < nb; j += 64) a += part[2 * j], b += part[2 * j + 1];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            b += __shfl_down(b, o, 64);
        }
        if (threadIdx.x == 0) sums[2 * k] = a, sums[2 * k + 1] = b;
    }
}

// ------------------------------------------------------------------------------------------
// PPO minibatch forward + loss + backward.  One wave per 16 samples, gradient accumulated with float atomics.
// ------------------------------------------------------------------------------------------
// Even blocks carry the POLICY net, odd blocks the VALUE net (they share nothing).  The input-gradient tiles overwrite the
// activations they derive from (dz2 over h2, dz1 over h1), so a wave needs X + 2 activation tiles of LDS and four waves fit.
template <bool CONT, bool IS_PI>
__device__ __forceinline__ void grad_generic_body(const float *__restrict__ params, const PLayout &L, const Rollout &rb, const Minibatch &mb,
                                                  const HParams &hp, const float *__restrict__ ws_adv, float *__restrict__ grad,
                                                  double *__restrict__ stat_slot, float *smem, int n_blocks_net, int block_net) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int D = L.D, H = L.H, A = L.A;
    const int ldx = ((D + 3) & ~3) + 2, ld = H + 2, ld3 = 34;
    const int per_wave = 16 * (ldx + 2 * ld + ld3) + 16 * 8;
    float *X = smem + (int64_t)wave * per_wave;
    float *h1 = X + 16 * ldx, *h2 = h1 + 16 * ld, *dzA = h2, *dzB = h1, *dz3 = h2 + 16 * ld;
    int64_t *row_off = reinterpret_cast<int64_t *>(dz3 + 16 * ld3);
    float *meta = reinterpret_cast<float *>(row_off + 16);
    const float invB = 1.0f / (float)mb.count;
    const float amean = hp.normalize_advantage ? ws_adv[0] : 0.0f;
    const float astd = hp.normalize_advantage ? ws_adv[1] : 1.0f;
    const Net Q = IS_PI ? pi_net(params, L) : vf_net(params, L);
    const int NOUT = IS_PI ? A : 1;
    float *gW1 = grad + (IS_PI ? L.pW1t : L.vW1t), *gb1 = grad + (IS_PI ? L.pb1 : L.vb1);
    float *gW2 = grad + (IS_PI ? L.pW2t : L.vW2t), *gb2 = grad + (IS_PI ? L.pb2 : L.vb2);
    float *gW3 = grad + (IS_PI ? L.pW3t : L.vW3t), *gb3 = grad + (IS_PI ? L.pb3 : L.vb3);
    double st_a = 0.0, st_ent = 0.0, st_kl = 0.0, st_clip = 0.0, st_n = 0.0;
    const int64_t n_tiles = (mb.count + 15) >> 4;
    for (int64_t tile = (int64_t)block_net * wpb + wave; tile < n_tiles; tile += (int64_t)n_blocks_net * wpb) {
        if (lane < 16) {
            const int64_t j = (tile << 4) + lane;
            int64_t off = -1;
            meta[lane * 4 + 0] = meta[lane * 4 + 1] = meta[lane * 4 + 2] = meta[lane * 4 + 3] = 0.0f;
            if (j < mb.count) {
                off = mb.offs ? (int64_t)mb.offs[j] : sample_offset(mb, mb.start + j, rb.T, rb.N);
                meta[lane * 4 + 0] = rb.log_probs[off];
                meta[lane * 4 + 1] = rb.advantages[off];
                meta[lane * 4 + 2] = rb.returns[off];
                if constexpr (!CONT) meta[lane * 4 + 3] = __int_as_float(static_cast<const int32_t *>(rb.actions)[off]);
            }
            row_off[lane] = off;
        }
        load_obs_tile(rb.obs, row_off, D, X, ldx, lane);
        dense_tanh(X, ldx, D, Q.W1t, Q.b1, H, h1, ld, lane);
        dense_tanh(h1, ld, H, Q.W2t, Q.b2, H, h2, ld, lane);
        if constexpr (!IS_PI) {
            f32x4 vacc[1];
            dense_head<1>(h2, ld, H, Q.W3t, Q.b3, 1, vacc, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = g * 4 + r;
                const bool valid = row_off[row] >= 0;
                const float diff = vacc[0][r] - meta[row * 4 + 2];
                dz3[row * ld3 + r16] = (valid && r16 == 0) ? (hp.vf_coef * 2.0f * invB) * diff : 0.0f;
                if (valid && r16 == 0) st_a += (double)(diff * diff);
            }
        } else if constexpr (!CONT) {
            f32x4 acc[1];
            dense_head<1>(h2, ld, H, Q.W3t, Q.b3, A, acc, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = g * 4 + r;
                const bool valid = row_off[row] >= 0;
                const bool colok = r16 < A;
                const float x = colok ? acc[0][r] : -INFINITY;
                const float m = gmax16(x);
                const float e = colok ? expf(x - m) : 0.0f;
                const float s = gsum16(e);
                const float lse = m + logf(s);
                const float lp = colok ? x - lse : 0.0f;
                const float p = e / s;
                const int act = __float_as_int(meta[row * 4 + 3]);
                const float lpa = gsum16((r16 == act) ? lp : 0.0f);
                const float ent = -gsum16(p * lp);
                const float old = meta[row * 4 + 0];
                const float advn = (meta[row * 4 + 1] - amean) / (astd + 1e-8f);
                const float ratio = expf(lpa - old);
                const float pl1 = advn * ratio;
                const float rc = fminf(fmaxf(ratio, 1.0f - hp.clip_range), 1.0f + hp.clip_range);
                const float pl2 = advn * rc;
                const float g_lp = (valid && pl1 <= pl2) ? -(advn * ratio) * invB : 0.0f;
                float dl = g_lp * (((r16 == act) ? 1.0f : 0.0f) - p);
                dl += valid ? (hp.ent_coef * invB) * (p * (lp + ent)) : 0.0f;
                dz3[row * ld3 + r16] = colok ? dl : 0.0f;
                if (valid && r16 == 0) {
                    st_a += (double)(-fminf(pl1, pl2));
                    st_ent += (double)ent;
                    st_kl += (double)((ratio - 1.0f) - (lpa - old));
                    st_clip += (fabsf(ratio - 1.0f) > hp.clip_range) ? 1.0 : 0.0;
                    st_n += 1.0;
                }
            }
        } else {
            f32x4 acc[2];
            dense_head<2>(h2, ld, H, Q.W3t, Q.b3, A, acc, lane);
            const float *lsp = params + L.log_std;
            float dlsd[2] = {0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = g * 4 + r;
                const int64_t off = row_off[row];
                const bool valid = off >= 0;
                float lpsum = 0.0f, d[2] = {0.0f, 0.0f}, sd[2] = {1.0f, 1.0f}, entsum = 0.0f;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int col = 16 * j + r16;
                    if (col < A) {
                        const float lsd = lsp[col];
                        sd[j] = expf(lsd);
                        const float a = valid ? static_cast<const float *>(rb.actions)[off * A + col] : 0.0f;
                        d[j] = a - acc[j][r];
                        lpsum += -(d[j] * d[j]) / (2.0f * (sd[j] * sd[j])) - lsd - 0.9189385332046727f;
                        entsum += 1.4189385332046727f + lsd;
                    }
                }
                const float lpa = gsum16(lpsum);
                const float ent = gsum16(entsum);
                const float old = meta[row * 4 + 0];
                const float advn = (meta[row * 4 + 1] - amean) / (astd + 1e-8f);
                const float ratio = expf(lpa - old);
                const float pl1 = advn * ratio;
                const float rc = fminf(fmaxf(ratio, 1.0f - hp.clip_range), 1.0f + hp.clip_range);
                const float pl2 = advn * rc;
                const float g_lp = (valid && pl1 <= pl2) ? -(advn * ratio) * invB : 0.0f;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int col = 16 * j + r16;
                    const float var = sd[j] * sd[j];
                    dz3[row * ld3 + col] = (col < A) ? g_lp * (d[j] / var) : 0.0f;
                    if (col < A) dlsd[j] += g_lp * ((d[j] * d[j]) / var - 1.0f) - (valid ? hp.ent_coef * invB : 0.0f);
                }
                if (valid && r16 == 0) {
                    st_a += (double)(-fminf(pl1, pl2));
                    st_ent += (double)ent;
                    st_kl += (double)((ratio - 1.0f) - (lpa - old));
                    st_clip += (fabsf(ratio - 1.0f) > hp.clip_range) ? 1.0 : 0.0;
                    st_n += 1.0;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {  // log_std gradient: fold the 4 row groups, one atomic per column
                float v = dlsd[j];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int col = 16 * j + r16;
                if (g == 0 && col < A) atomicAdd(grad + L.log_std + col, v);
            }
        }
        dense_bwd_weight(h2, ld, H, dz3, ld3, NOUT, gW3, gb3, lane);
        dense_bwd_input(dz3, ld3, NOUT, Q.W3, H, h2, ld, dzA, ld, lane);
        dense_bwd_weight(h1, ld, H, dzA, ld, H, gW2, gb2, lane);
        dense_bwd_input(dzA, ld, H, Q.W2, H, h1, ld, dzB, ld, lane);
        dense_bwd_weight(X, ldx, D, dzB, ld, H, gW1, gb1, lane);
    }
    double st[5] = {st_a, st_ent, st_kl, st_clip, st_n};
#pragma unroll
    for (int q = 0; q < 5; q++)
        for (int o = 32; o > 0; o >>= 1) st[q] += __shfl_down(st[q], o, 64);
    __syncthreads();
    double *red = reinterpret_cast<double *>(smem);
    if (lane == 0)
        for (int q = 0; q < 5; q++) red[wave * 5 + q] = st[q];
    __syncthreads();
    if (threadIdx.x < 5) {
        double s = 0.0;
        for (int w = 0; w < wpb; w++) s += red[w * 5 + threadIdx.x];
        const int q = IS_PI ? (threadIdx.x == 0 ? 0 : threadIdx.x + 1) : (threadIdx.x == 0 ? 1 : -1);
        if (q >= 0) stat_slot[q] += s;
    }
}

template <bool CONT>
__global__ __launch_bounds__(256) void ppo_grad_kernel(const float *__restrict__ params, PLayout L, Rollout rb, Minibatch mb, HParams hp,
                                                       const float *__restrict__ ws_adv, float *__restrict__ grad, double *__restrict__ stat_slots) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pair = blockIdx.x >> 1, n_pairs = gridDim.x >> 1;
    double *slot = stat_slots + (int64_t)pair * 8;
    if ((blockIdx.x & 1) == 0) grad_generic_body<CONT, true>(params, L, rb, mb, hp, ws_adv, grad, slot, smem, n_pairs, pair);
    else grad_generic_body<CONT, false>(params, L, rb, mb, hp, ws_adv, grad, slot, smem, n_pairs, pair);
}

// ------------------------------------------------------------------------------------------
// Wide policies (H = 128 / 256: the reference's default net_arch is 256x256, training.py:363-365): column-parallel blocks.
// A block of 4 waves (one per SIMD, so each wave has the full 512-register budget: 256 accumulator + 256 working registers)
// walks row groups of M = 32 samples for ONE net; wave w owns output columns [w*H/4, (w+1)*H/4) of both hidden layers, so its
// slice of every weight gradient (dW2: H x H/4 = 256 registers at H = 256) stays in MFMA accumulators
// for the whole launch, every weight fragment fetched from L2 is used for both 16-row tiles, and each block ends by storing
// ITS slab of the gradient with plain stores (no atomics; slab_reduce_kernel folds the blocks in a fixed order).
// Activations of the row group live in block-shared LDS; __syncthreads separates the layers.
// ------------------------------------------------------------------------------------------
// HALF (round 4): row groups of 16 samples (one row tile) instead of 32 -- for minibatches too small to give every CU a 32-row group (the
// reference's literal batch_size = 256: 8 groups per net = 16 workgroups; as half groups 32): the second row tile's MFMAs, LDS traffic and
// epilogues are compiled out, the sample dimension of the weight-gradient GEMMs runs over 4 k-steps instead of 8.  Same operations per
// element in the same order as the first row tile of a full group.
// NW = 8 (round 4, single-pass shapes): two waves per SIMD, 32 columns each -- the dW2 slice of a wave is 128 accumulator registers instead of 256,
// and one wave's LDS waits and epilogues run beside its SIMD partner's MFMAs (the f32 MFMA pipe was 60 % busy with one wave per SIMD).
#ifndef TMA_HALF_RING
#define TMA_HALF_RING 4
#endif
constexpr int W2_DEFER_ROWS = 1024;  // rows per image of the dW2 deferral buffer ([net][h1 | dz2][W2_DEFER_ROWS][H]: half-group minibatches of <= 1024 samples)
#ifdef TMA_WIDE_PHASE_TICKS  // diagnostic build (make libtma_hip_wticks.so, tools/wide_ticks.py): cycles per phase of the f32 wide gradient kernel, wave 0 of block 0 of each net
__device__ unsigned long long g_wide_ticks[2][16];
#define TMA_WTICK(i)                                                      \
    do {                                                                  \
        const long long tn_ = clock64();                                  \
        if (wtick_on) wtick_lds[i] += (unsigned long long)(tn_ - wtlast); \
        wtlast = clock64();                                               \
    } while (0)
#else
#define TMA_WTICK(i)
#endif
template <bool CONT, bool IS_PI, int NTW, int KT1C, int PASS, int NQ1C, bool HALF = false, int NW = 4>
__device__ __forceinline__ void grad_wide_body(const float *__restrict__ params, const PLayout &L, const Rollout &rb, const Minibatch &mb,
                                               const HParams &hp, const float *__restrict__ ws_adv, float *__restrict__ slab,
                                               double *__restrict__ stat_slot, float *smem, int n_blocks_net, int block_net,
                                               float *__restrict__ dz1c) {
    // dz1c (two-pass widths, minibatches that fit the workspace cache): PASS 0 leaves every row group's dz1 there as the B operands
    // of the dW1 MFMAs ([group][wave][tile][lane][8 floats]); PASS 2 re-gathers the observation rows and runs only those MFMAs
    // -- same operands, same order as PASS 1 (the recompute pass, kept for larger minibatches), hence the same bits.
    static_assert(NW == 4 || (NW == 8 && PASS == 0 && KT1C > 0 && NQ1C == 0), "eight waves: single-pass shapes with dW1 in registers");
    constexpr int M = 32, RW = M / NW, H = 16 * NTW * NW, KT2 = H / 16, NT3 = (IS_PI && CONT) ? 2 : 1, ld = H + 2, ld3 = 34;
    constexpr int MG = HALF ? 16 : 32, MTN = HALF ? 1 : 2, SN = HALF ? 4 : 8;  // rows per group, row tiles, sample k-steps of the weight-gradient GEMMs
    static_assert(!HALF || (PASS == 0 && KT1C > 0 && NQ1C == 0), "half groups: single-pass shapes with dW1 in registers");
    const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (same two measures as tma_wide_bf16.h: the weight pointers are laundered once per row group so LICM cannot hoist the
    // loop-invariant weight loads out of the group loop and spill them, and lane-derived addresses are re-derived per phase)
    int lane = lane0, r16 = lane0 & 15, g = lane0 >> 4;
#define TMA_RELANE()                       \
    do {                                   \
        lane = lane0;                      \
        asm volatile("" : "+v"(lane));     \
        r16 = lane & 15, g = lane >> 4;    \
    } while (0)
#ifdef TMA_WIDE_PHASE_TICKS
    __shared__ unsigned long long wtick_lds[16];
    if (threadIdx.x < 16) wtick_lds[threadIdx.x] = 0;
    long long wtlast = clock64();
    const bool wtick_on = threadIdx.x == 0 && block_net == 0;
#endif
    const int D = L.D, A = L.A;
    const int NOUT = IS_PI ? A : 1;
    const int ldx = ((D + 3) & ~3) + 2, KS1 = (D + 3) >> 2, KT1 = (D + 15) >> 4;
    constexpr bool MAIN = PASS == 0;
    constexpr bool acc_w1 = KT1C > 0, rmw_w1 = KT1C == 0;
    constexpr int KT1A = KT1C > 0 ? KT1C : 1;
    float *X = smem, *h1 = X + M * ldx, *h2 = h1 + M * ld, *dz3 = h2 + M * ld;
    float *meta = dz3 + M * ld3;
    int64_t *row_off = reinterpret_cast<int64_t *>(meta + M * 4);
    float *scratch = reinterpret_cast<float *>(row_off + M);  // 64 floats
    float *hpart = scratch + 64;  // [4 waves][2 tiles][2][64 lanes][4]: split-K partial head outputs
    float *bias = hpart + NW * 2 * 2 * 256;  // b1[H], b2[H], b3[32] (zero padded): LDS copies, so no global load sits in front of a phase
    // buffer offsets of the NEXT group's rows (round 5): fetched during the current group, so that a group starts with ONE memory round trip --
    // metadata and observation rows together -- instead of three dependent ones (offset -> metadata -> barrier -> observation rows)
    int64_t *row_off_next = reinterpret_cast<int64_t *>(bias + 2 * H + 32);
    // Box heads (round 6): the group's actions [M][32], gathered with the observation rows -- the loss had read them from global memory
    // where it needed them (a cache-missing round trip in the middle of its phase: 5.5 k of a 256-sample launch's 65 k cycles at the Ant width)
    constexpr bool ACT_TILE = CONT && IS_PI && PASS != 2;
    float *act_tile = reinterpret_cast<float *>(row_off_next + M);
    const int n_base = wave * 16 * NTW;
    const float invB = 1.0f / (float)mb.count;
    // minibatch advantage statistics: folded here from the partials (the order of adv_final_kernel, so the same bits) instead of by a
    // one-block launch in front of every minibatch (4.8 us of a 70 us step at 256 samples); the prologue's barrier below publishes them
    (void)ws_adv;
    __shared__ float adv_ms[2];
    if (IS_PI && hp.normalize_advantage && threadIdx.x < 64) {
        double a = 0.0, bsum = 0.0;
        for (int k = threadIdx.x; k < mb.adv_n_part; k += 64) a += mb.adv_part[2 * k], bsum += mb.adv_part[2 * k + 1];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            bsum += __shfl_down(bsum, o, 64);
        }
        if (threadIdx.x == 0) {
            const double n = (double)mb.stats_n, mean = a / n;
            double var = n > 1.0 ? (bsum - n * mean * mean) / (n - 1.0) : 0.0;
            if (var < 0.0) var = 0.0;
            adv_ms[0] = (float)mean;
            adv_ms[1] = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float amean = (IS_PI && hp.normalize_advantage) ? adv_ms[0] : 0.0f;
    const float astd = (IS_PI && hp.normalize_advantage) ? adv_ms[1] : 1.0f;
    Net Q = IS_PI ? pi_net(params, L) : vf_net(params, L);
    const f32x4 z4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 aW2[KT2][NTW], aW1[KT1A][NTW], aW3[NTW][NT3];
    float ab1[NTW], ab2[NTW], ab3[NT3], dlsd[2] = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        ab1[j] = ab2[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < KT1A; i++) aW1[i][j] = z4;
#pragma unroll
        for (int i = 0; i < KT2; i++) aW2[i][j] = z4;
#pragma unroll
        for (int q = 0; q < NT3; q++) aW3[j][q] = z4;
    }
#pragma unroll
    for (int q = 0; q < NT3; q++) ab3[q] = 0.0f;
    LossStats st;
    for (int e = threadIdx.x; e < 2 * H + 32; e += blockDim.x)
        bias[e] = e < H ? Q.b1[e] : (e < 2 * H ? Q.b2[e - H] : (e - 2 * H < NOUT ? Q.b3[e - 2 * H] : 0.0f));
    auto next_off = [&](int64_t grp_n) -> int64_t {  // buffer offset of row threadIdx.x of group grp_n (-1: padding row / beyond the minibatch)
        const int64_t j = grp_n * MG + threadIdx.x;
        if ((int)threadIdx.x >= MG || j >= mb.count) return -1;
        return mb.offs ? (int64_t)mb.offs[j] : sample_offset(mb, mb.start + j, rb.T, rb.N);
    };
    if (threadIdx.x < M) row_off_next[threadIdx.x] = next_off(block_net);
    __syncthreads();
    // The two H x H weight streams of a row group (layer-2 forward, then layer-2 input-gradient) come from the fragment-major f32
    // images (PLayout::fr_pi) through a register ring of R fragments: one 16-byte load per lane feeds four k-steps (8 MFMAs), and
    // the slot a fragment is consumed from is reloaded at once with the fragment R positions further down the cyclic stream --
    // 2 k cycles of MFMA work of lookahead that carries across phases, barriers and row groups.
    constexpr int NQ = H / 16, S1 = NTW * NQ1C, SL = S1 + 2 * NTW * NQ;
    // (half groups halve the MFMA work behind every fragment, i.e. the lookahead a slot buys: the four-wave half-group kernel has the registers
    //  of the compiled-out second row tile to spare; -DTMA_HALF_RING=16 gives it a ring four times as deep -- measured 40.3 us against 41.9 at 256 samples,
    //  both behind the eight-wave kernel's 34.8, so the default stays 4)
    constexpr int R = (HALF && NW == 4 && SL % TMA_HALF_RING == 0) ? TMA_HALF_RING : (SL % 4 == 0 ? 4 : (SL % 5 == 0 ? 5 : 3));  // ring slots are static registers: R must divide the cyclic stream
    static_assert(SL % R == 0, "ring must divide the per-group fragment stream");
    const float *fr = params + (IS_PI ? L.fr_pi : L.fr_vf), *fr1 = params + (IS_PI ? L.fr1_pi : L.fr1_vf);
    int nt0 = wave * NTW;
    auto sload = [&](int s) -> f32x4 {  // s in [0, SL): compile-time after unrolling.  Order: [layer 1: q outer, tile inner] [layer-2 forward] [input-gradient]
        if (s < S1) return frag_f32(fr1, (nt0 + s % NTW) * NQ1C + s / NTW, lane0);
        const int u = s - S1, half = u >= NTW * NQ, t = half ? u - NTW * NQ : u;
#ifdef TMA_WIDE_RING_HOT  // timing-only build: every ring load hits one of two L1-resident fragments (wrong results; what does the stream cost?)
        return frag_f32(fr + half * H * H, (nt0 + 0) * NQ + (t & 1), lane0);
#endif
        return frag_f32(fr + half * H * H, (nt0 + t / NQ) * NQ + t % NQ, lane0);
    };
    f32x4 ring[R];
    if constexpr (PASS != 2) {
#pragma unroll
        for (int s = 0; s < R; s++) ring[s] = sload(s);
    }
    const int64_t n_groups = (mb.count + MG - 1) / MG;
    // PASS 2 (round 6): a group is its observation rows and its cached dz1 operands, then 32 MFMAs per k-tile -- the rows and operands of the NEXT
    // group are requested in front of this group's MFMAs and committed to LDS at the top of the next iteration, the row offsets a group
    // further ahead (the gather had been two dependent round trips -- offsets, then rows -- in front of every group's MFMAs).  Same operands in
    // the same order: the bits of PASS 1.  Measured at the Crawler width: 299 -> 288 us, of which the MFMAs are 194 (one k-tile instead of
    // eleven: 112 us; no requests: 241 us) -- the requests still cost 47 us because their 56 destination registers do not fit beside the 176
    // accumulators this file's -amdgpu-mfma-vgpr-form=1 puts into architectural registers: the compiler parks them in accumulator registers
    // as they arrive, behind counted waits INSIDE the MFMA block.  Not pursued further.
    [[maybe_unused]] float tq[3][RW];
    [[maybe_unused]] f32x4 zq[NTW][2];
    [[maybe_unused]] bool qok[RW];
    [[maybe_unused]] int64_t noff2 = -1;
    auto p2_issue = [&](int64_t g2) {  // (row_off_next holds the offsets of group g2, published by a barrier)
        if constexpr (PASS == 2) {
            typedef const float __attribute__((address_space(1))) *gf_ptr;
            const int Dp = (D + 3) & ~3;
#pragma unroll
            for (int i = 0; i < RW; i++) {
                const int64_t off = row_off_next[wave * RW + i];
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)off), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)off >> 32));
                const int64_t offu = (int64_t)(((uint64_t)hi << 32) | lo);
                qok[i] = offu >= 0;
                gf_ptr rb_i = reinterpret_cast<gf_ptr>(reinterpret_cast<uintptr_t>(rb.obs + (qok[i] ? offu * D : 0)));
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int c = 64 * k + lane0;
                    if (64 * k < Dp) tq[k][i] = rb_i[c < D ? c : 0];
                }
            }
            const float *gi = dz1c + (g2 * 4 + wave) * (int64_t)(NTW * 512);
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                zq[j][0] = *reinterpret_cast<const f32x4 *>(gi + (j * 64 + lane0) * 8);
                zq[j][1] = *reinterpret_cast<const f32x4 *>(gi + (j * 64 + lane0) * 8 + 4);
            }
        }
    };
    if constexpr (PASS == 2) {
        static_assert(NW == 4, "PASS 2: the dz1 cache is laid out for four waves");
        if (block_net < n_groups) {
            if (threadIdx.x < M) noff2 = block_net + n_blocks_net < n_groups ? next_off(block_net + n_blocks_net) : -1;
            p2_issue(block_net);  // (the prologue's barrier published row_off_next)
            __syncthreads();  // every wave has read its rows' offsets
            if (threadIdx.x < M) row_off_next[threadIdx.x] = noff2;  // (published by the loop's first barrier)
        }
    }
    TMA_WTICK(0);  // prologue
    for (int64_t grp = block_net; grp < n_groups; grp += n_blocks_net) {
        f32x4 zc[NTW][2];  // PASS 2: cached dz1 operands of this wave (requested a group ago)
        if constexpr (PASS == 2) {
            const int Dp = (D + 3) & ~3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int c = 64 * k + lane0;
                if (64 * k < Dp && c < Dp) {
#pragma unroll
                    for (int i = 0; i < RW; i++) X[(wave * RW + i) * ldx + c] = (qok[i] && c < D) ? tq[k][i] : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < NTW; j++) zc[j][0] = zq[j][0], zc[j][1] = zq[j][1];
            __syncthreads();  // the group's rows are in LDS; row_off_next names the next group's
            // (the offset load in front of the row requests: it is consumed at the end of this iteration, and a wait for it must not wait for them --
            //  vmcnt retires in order)
            if (threadIdx.x < M) noff2 = grp + 2 * (int64_t)n_blocks_net < n_groups ? next_off(grp + 2 * (int64_t)n_blocks_net) : -1;
            if (grp + n_blocks_net < n_groups) p2_issue(grp + n_blocks_net);
            TMA_RELANE();
            // (per accumulator the same sidx order as P6 of PASS 1; the A operands of k-tile kt + 1 are read under the MFMAs of k-tile kt)
            float a[2][8];
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) a[0][sidx] = r16 < D ? X[(4 * sidx + g) * ldx + r16] : 0.0f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < KT1A; kt++) {
                if (kt + 1 < KT1A) {
                    const int krow = (kt + 1) * 16 + r16;
#pragma unroll
                    for (int sidx = 0; sidx < 8; sidx++) a[(kt + 1) & 1][sidx] = krow < D ? X[(4 * sidx + g) * ldx + krow] : 0.0f;
                }
                // (sample step outer, column tile inner: consecutive MFMAs go to different accumulators -- with the tile outermost eight
                //  dependent MFMAs in a row issued once per ~45 cycles instead of 32: 305 us for this pass at the Crawler width)
#pragma unroll
                for (int sidx = 0; sidx < 8; sidx++)
#pragma unroll
                    for (int j = 0; j < NTW; j++) aW1[kt][j] = mfma16(a[kt & 1][sidx], zc[j][sidx >> 2][sidx & 3], aW1[kt][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();  // X and row_off_next are read
            if (threadIdx.x < M) row_off_next[threadIdx.x] = noff2;
            continue;
        }
        {
            const float *pl = launder_uniform(params);
            Q = IS_PI ? pi_net(pl, L) : vf_net(pl, L);
            fr = pl + (IS_PI ? L.fr_pi : L.fr_vf);
            fr1 = pl + (IS_PI ? L.fr1_pi : L.fr1_vf);
            asm volatile("" : "+s"(nt0));
            nt0 = __builtin_amdgcn_readfirstlane(nt0);
        }
        TMA_RELANE();
        // ---- P0: gather sample metadata and the observation rows (row offsets: row_off_next, left there by the previous group) ----
        {
            // Wave w gathers rows RW w .. RW w + RW - 1, lane l the columns l, l + 64, ...: the row's buffer offset is wave-uniform (scalar base +
            // lane offset, no 64-bit per-lane address arithmetic) and a whole batch of loads is in flight before the first store.
            // (The element-indexed form of this loop -- e = tid + 256 i, row = e / D -- was instruction-bound at Crawler width.)
            typedef const float __attribute__((address_space(1))) *gf_ptr;
            gf_ptr rbase[RW];
            bool rok[RW];
            [[maybe_unused]] float actv[RW];
#pragma unroll
            for (int i = 0; i < RW; i++) {
                const int64_t off = row_off_next[wave * RW + i];
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)off), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)off >> 32));
                const int64_t offu = (int64_t)(((uint64_t)hi << 32) | lo);
                rok[i] = offu >= 0;
                rbase[i] = reinterpret_cast<gf_ptr>(reinterpret_cast<uintptr_t>(rb.obs + (rok[i] ? offu * D : 0)));
                if constexpr (ACT_TILE) {  // (issue only; stored behind the observation chunks)
                    gf_ptr ab = reinterpret_cast<gf_ptr>(reinterpret_cast<uintptr_t>(static_cast<const float *>(rb.actions) + (rok[i] ? offu * A : 0)));
                    actv[i] = ab[lane < A ? lane : 0];
                }
            }
            // the metadata of row threadIdx.x: issued in front of the observation loads, consumed behind them (one round trip for both)
            int64_t moff = -1;
            float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
            if (threadIdx.x < M) {
                moff = row_off_next[threadIdx.x];
                if constexpr (PASS != 2) {
                    if (moff >= 0) {
                        m0 = rb.log_probs[moff], m1 = rb.advantages[moff], m2 = rb.returns[moff];
                        if constexpr (!CONT) m3 = __int_as_float(static_cast<const int32_t *>(rb.actions)[moff]);
                    }
                }
            }
            const int Dp = (D + 3) & ~3;
            for (int c0 = 0; c0 < Dp; c0 += 192) {
                float t[3][RW];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int c = c0 + 64 * k + lane;
                    if (c0 + 64 * k < Dp) {  // (uniform) narrow observations: one column chunk
#pragma unroll
                        for (int i = 0; i < RW; i++) t[k][i] = rbase[i][c < D ? c : 0];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int c = c0 + 64 * k + lane;
                    if (c0 + 64 * k < Dp && c < Dp) {
#pragma unroll
                        for (int i = 0; i < RW; i++) X[(wave * RW + i) * ldx + c] = (rok[i] && c < D) ? t[k][i] : 0.0f;
                    }
                }
            }
            if (threadIdx.x < M) {
                meta[threadIdx.x * 4 + 0] = m0, meta[threadIdx.x * 4 + 1] = m1, meta[threadIdx.x * 4 + 2] = m2, meta[threadIdx.x * 4 + 3] = m3;
                row_off[threadIdx.x] = moff;
            }
            if constexpr (ACT_TILE) {
                if (lane < 32) {
#pragma unroll
                    for (int i = 0; i < RW; i++) act_tile[(wave * RW + i) * 32 + lane] = (rok[i] && lane < A) ? actv[i] : 0.0f;
                }
            }
        }
        __syncthreads();
        TMA_WTICK(1);  // P0 gathers
        TMA_RELANE();
        if constexpr (PASS == 2) {
#pragma unroll
            for (int kt = 0; kt < KT1A; kt++) {  // (per accumulator the same sidx order as P6 of PASS 1; the A operand is read once per k-tile)
                const int krow = kt * 16 + r16;
                float a[8];
#pragma unroll
                for (int sidx = 0; sidx < 8; sidx++) a[sidx] = krow < D ? X[(4 * sidx + g) * ldx + krow] : 0.0f;
#pragma unroll
                for (int j = 0; j < NTW; j++)
#pragma unroll
                    for (int sidx = 0; sidx < 8; sidx++) aW1[kt][j] = mfma16(a[sidx], zc[j][sidx >> 2][sidx & 3], aW1[kt][j]);
            }
            if (threadIdx.x < M) row_off_next[threadIdx.x] = grp + n_blocks_net < n_groups ? next_off(grp + n_blocks_net) : -1;  // (read in front of the barrier above)
            __syncthreads();
            continue;
        }
        // ---- P1: layer 1 forward, this wave's columns, both row tiles ----
        {
            f32x4 acc[NTW][2];
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const float bias_v = bias[n_base + 16 * j + r16];
                acc[j][0] = acc[j][1] = f32x4{bias_v, bias_v, bias_v, bias_v};
            }
            if constexpr (NQ1C > 0) {
#pragma unroll
                for (int q = 0; q < NQ1C; q++) {
                    float a0[4], a1[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int k = 16 * q + 4 * i + g;
                        const bool ok = k < D;  // the image's rows >= D are zero, but the LDS columns there are not initialised
                        a0[i] = ok ? X[r16 * ldx + k] : 0.0f, a1[i] = ok ? X[(16 + r16) * ldx + k] : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < NTW; j++) {
                        const int s = q * NTW + j;
                        const f32x4 w4 = ring[s % R];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            acc[j][0] = mfma16(a0[i], w4[i], acc[j][0]);
                            acc[j][1] = mfma16(a1[i], w4[i], acc[j][1]);
                        }
                        ring[s % R] = sload((s + R) % SL);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (KT1C >= 1 && KT1C <= 8) {  // D <= 32 (or, round 6, <= 112 on half groups): all weight loads issued together
                constexpr int KSC = 4 * KT1C, KB = KSC <= 8 ? KSC : 8, NB = (KSC + KB - 1) / KB;  // (wider inputs: eight k-steps of weights per batch of loads,
                float w1[2][KB][NTW];                                                    //  the next batch requested while this one multiplies)
                auto w1_load = [&](int b) {
#pragma unroll
                    for (int ks = 0; ks < KB; ks++)
#pragma unroll
                        for (int j = 0; j < NTW; j++) {
                            const int k = 4 * (b * KB + ks) + g;
                            w1[b & 1][ks][j] = k < D ? Q.W1t[(int64_t)k * H + n_base + 16 * j + r16] : 0.0f;
                        }
                };
                w1_load(0);
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if (b + 1 < NB) {
                        w1_load(b + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int ks = 0; ks < KB; ks++) {
                        if (b * KB + ks < KS1) {
                            const int k = 4 * (b * KB + ks) + g;
                            const float a0 = X[r16 * ldx + k], a1 = HALF ? 0.0f : X[(16 + r16) * ldx + k];
#pragma unroll
                            for (int j = 0; j < NTW; j++) {
                                acc[j][0] = mfma16(a0, w1[b & 1][ks][j], acc[j][0]);
                                if constexpr (!HALF) acc[j][1] = mfma16(a1, w1[b & 1][ks][j], acc[j][1]);
                            }
                        }
                    }
                    if (b + 1 < NB) __builtin_amdgcn_sched_barrier(0);
                }
            } else
            for (int ks = 0; ks < KS1; ks++) {
                const int k = 4 * ks + g;
                const bool ok = k < D;
                const float a0 = X[r16 * ldx + k], a1 = X[(16 + r16) * ldx + k];
#pragma unroll
                for (int j = 0; j < NTW; j++) {
                    const float w = ok ? Q.W1t[(int64_t)k * H + n_base + 16 * j + r16] : 0.0f;
                    acc[j][0] = mfma16(a0, w, acc[j][0]);
                    acc[j][1] = mfma16(a1, w, acc[j][1]);
                }
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int mt = 0; mt < MTN; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) h1[(mt * 16 + g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(acc[j][mt][r]);
        }
        __syncthreads();
        TMA_WTICK(2);  // P1 layer 1
        TMA_RELANE();
        // ---- P2: layer 2 forward, one 16-column tile at a time (8 accumulator registers live), weights through the ring ----
#pragma unroll
        for (int j = 0; j < NTW; j++) {
            const float bias_v = bias[H + n_base + 16 * j + r16];
            f32x4 c0 = f32x4{bias_v, bias_v, bias_v, bias_v}, c1 = c0;
            // explicit pipeline, fenced per fragment: A operands one fragment ahead (LDS), the ring slot reloaded right behind its use
            float a0[2][4], a1[2][4];
#pragma unroll
            for (int i = 0; i < 4; i++) a0[0][i] = h1[r16 * ld + 4 * i + g], a1[0][i] = HALF ? 0.0f : h1[(16 + r16) * ld + 4 * i + g];
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int s = S1 + j * NQ + q;
                if (q + 1 < NQ) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int k = 16 * (q + 1) + 4 * i + g;
                        a0[(q + 1) & 1][i] = h1[r16 * ld + k], a1[(q + 1) & 1][i] = HALF ? 0.0f : h1[(16 + r16) * ld + k];
                    }
                }
                const f32x4 w4 = ring[s % R];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    c0 = mfma16(a0[q & 1][i], w4[i], c0);
                    if constexpr (!HALF) c1 = mfma16(a1[q & 1][i], w4[i], c1);
                }
                ring[s % R] = sload((s + R) % SL);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                h2[(g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(c0[r]);
                if constexpr (!HALF) h2[(16 + g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(c1[r]);
            }
        }
        __syncthreads();
        TMA_WTICK(3);  // P2 layer 2
        TMA_RELANE();
        // ---- P3a: split-K head: wave w multiplies k-steps [w*H/16, (w+1)*H/16) of h2 for both row tiles; all of its weight loads
        // go out together (one L2 round trip per group instead of a dependent load per k-step on two waves) ----
        {
            constexpr int HKS = H / (4 * NW);  // k-steps of 4 per wave
            float w3[HKS][NT3];
#pragma unroll
            for (int i = 0; i < HKS; i++)
#pragma unroll
                for (int q = 0; q < NT3; q++) {
                    const int k = 4 * (wave * HKS + i) + g, col = 16 * q + r16;
                    w3[i][q] = col < NOUT ? Q.W3t[(int64_t)k * NOUT + col] : 0.0f;
                }
#pragma unroll
            for (int mt = 0; mt < MTN; mt++) {
                f32x4 part[NT3];
#pragma unroll
                for (int q = 0; q < NT3; q++) part[q] = z4;
#pragma unroll
                for (int i = 0; i < HKS; i++) {
                    const float a = h2[(mt * 16 + r16) * ld + 4 * (wave * HKS + i) + g];
#pragma unroll
                    for (int q = 0; q < NT3; q++) part[q] = mfma16(a, w3[i][q], part[q]);
                }
#pragma unroll
                for (int q = 0; q < NT3; q++) *reinterpret_cast<f32x4 *>(hpart + (((wave * 2 + mt) * 2 + q) * 64 + lane) * 4) = part[q];
            }
        }
        __syncthreads();
        TMA_RELANE();
        // ---- P3b: loss: wave mt (0, 1) takes row tile mt.  Eight waves, Categorical head (round 5): the four rows of a lane group go to four
        // waves per row tile -- the loss is a dependent chain of transcendentals per row (4.2 k cycles per group on two waves while the other six
        // sat at the barrier); per row the same operations, so the same dz3 and the same per-row statistics ----
        constexpr bool LOSS8 = NW == 8 && IS_PI && !CONT;
        const int loss_mt = LOSS8 ? (MTN == 2 ? (wave & 1) : 0) : wave, loss_part = LOSS8 ? (MTN == 2 ? (wave >> 1) : wave) : 0;
        if (LOSS8 ? loss_part < 4 : wave < MTN) {
            const int mt = loss_mt;
            f32x4 out[NT3];
#pragma unroll
            for (int q = 0; q < NT3; q++) {
                const int col = 16 * q + r16;
                const float b = bias[2 * H + col];
                out[q] = f32x4{b, b, b, b};
#pragma unroll
                for (int w = 0; w < NW; w++) out[q] += *reinterpret_cast<const f32x4 *>(hpart + (((w * 2 + mt) * 2 + q) * 64 + lane) * 4);
            }
            float *dzt = dz3 + mt * 16 * ld3;
            if constexpr (LOSS8) {
#define TMA_WLOSS_ARGS out, meta + mt * 64, row_off + mt * 16, rb.actions, params + L.log_std, A, amean, astd, hp, invB, dzt, ld3, dlsd, st, lane, 0, 4, (ACT_TILE ? act_tile + mt * 16 * 32 : nullptr)
                if (loss_part == 0) policy_loss_tile<CONT, 0, 1>(TMA_WLOSS_ARGS);
                else if (loss_part == 1) policy_loss_tile<CONT, 1, 2>(TMA_WLOSS_ARGS);
                else if (loss_part == 2) policy_loss_tile<CONT, 2, 3>(TMA_WLOSS_ARGS);
                else policy_loss_tile<CONT, 3, 4>(TMA_WLOSS_ARGS);
#undef TMA_WLOSS_ARGS
            } else if constexpr (IS_PI) {
                policy_loss_tile<CONT>(out, meta + mt * 64, row_off + mt * 16, rb.actions, params + L.log_std, A, amean, astd, hp, invB, dzt, ld3, dlsd, st,
                                       lane, 0, 4, ACT_TILE ? act_tile + mt * 16 * 32 : nullptr);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = g * 4 + r;
                    const bool valid = row_off[mt * 16 + row] >= 0;
                    const float diff = out[0][r] - meta[(mt * 16 + row) * 4 + 2];
                    dzt[row * ld3 + r16] = (valid && r16 == 0) ? (hp.vf_coef * 2.0f * invB) * diff : 0.0f;
                    if (valid && r16 == 0) st.a += (double)(diff * diff);
                }
            }
        }
        __syncthreads();
        TMA_WTICK(4);  // P3 head + loss
        TMA_RELANE();
        // (the next group's row offsets: requested here, parked in LDS at the end of the group -- row_off_next was last read in front of P1's barrier)
        int64_t noff = -1;
        if (threadIdx.x < M && grp + n_blocks_net < n_groups) noff = next_off(grp + n_blocks_net);
        // ---- P4: head weight gradient (this wave's k rows), head bias, and dz2 = (dz3 . W3) * (1 - h2^2) in place ----
        {
#pragma unroll
            for (int q = 0; q < NT3; q++) {
                const int col = 16 * q + r16;
                float bf[8];
#pragma unroll
                for (int sidx = 0; sidx < SN; sidx++) bf[sidx] = dz3[(4 * sidx + g) * ld3 + col];
                if (wave == 0) {
                    float c = 0.0f;
#pragma unroll
                    for (int sidx = 0; sidx < SN; sidx++) c += bf[sidx];
                    ab3[q] += c;
                }
#pragma unroll
                for (int i = 0; i < NTW; i++) {
                    const int krow = n_base + 16 * i + r16;
#pragma unroll
                    for (int sidx = 0; sidx < SN; sidx++) aW3[i][q] = mfma16(h2[(4 * sidx + g) * ld + krow], bf[sidx], aW3[i][q]);
                }
            }
            f32x4 acc[NTW][2];
#pragma unroll
            for (int j = 0; j < NTW; j++) acc[j][0] = acc[j][1] = z4;
            const int NS = (NOUT + 3) >> 2;
            constexpr int NSC = 4 * NT3;  // n_out <= 16 * NT3: all head weights of this wave's columns fetched in one round trip
            float w3b[NSC][NTW];
#pragma unroll
            for (int ns = 0; ns < NSC; ns++)
#pragma unroll
                for (int j = 0; j < NTW; j++) {
                    const int n = 4 * ns + g;
                    w3b[ns][j] = n < NOUT ? Q.W3[(int64_t)n * H + n_base + 16 * j + r16] : 0.0f;
                }
#pragma unroll
            for (int ns = 0; ns < NSC; ns++) {
                if (ns < NS) {
                    const int n = 4 * ns + g;
                    const float a0 = dz3[r16 * ld3 + n], a1 = HALF ? 0.0f : dz3[(16 + r16) * ld3 + n];
#pragma unroll
                    for (int j = 0; j < NTW; j++) {
                        acc[j][0] = mfma16(a0, w3b[ns][j], acc[j][0]);
                        if constexpr (!HALF) acc[j][1] = mfma16(a1, w3b[ns][j], acc[j][1]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int mt = 0; mt < MTN; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float *pp = h2 + (mt * 16 + g * 4 + r) * ld + n_base + 16 * j + r16;
                        const float h = *pp;
                        *pp = acc[j][mt][r] * (1.0f - h * h);
                    }
        }
        __syncthreads();
        TMA_WTICK(5);  // P4 dW3 + dz2
        TMA_RELANE();
        // ---- P5: dW2 slice += h1^T . dz2[:, slice];  dz1 = (dz2 . W2) * (1 - h1^2) kept in registers until every wave is done with h1 ----
        f32x4 dz1[NTW][2];
        {
            float bf[NTW][8];
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const int col = n_base + 16 * j + r16;
                float c = 0.0f;
#pragma unroll
                for (int sidx = 0; sidx < SN; sidx++) {
                    bf[j][sidx] = h2[(4 * sidx + g) * ld + col];
                    c += bf[j][sidx];
                }
                ab2[j] += c;
            }
            // Half groups with a deferral buffer (round 5, minibatches of <= 1024 samples): dW2 = h1^T . dz2 is NOT accumulated here -- a block's
            // rank-16 contribution would leave as a 256 KB slab that sixteen to sixty-four blocks write and slab_reduce_kernel reads back.  The group's
            // h1 and dz2 rows (16 KB each) go to the buffer instead and wide_small_reduce_kernel forms H1^T . DZ2 over the whole minibatch as ONE
            // GEMM spread over 128 workgroups, straight into the gradient.
            constexpr bool ALWAYS_DEFER = HALF && KT1C > 2;  // (wider inputs: dW1's k-tiles take the registers dW2 would need -- the launch guarantees the buffer)
            const bool defer_w2 = ALWAYS_DEFER || (HALF && dz1c != nullptr);  // (block-uniform)
            if (defer_w2) {
                float *bh = dz1c, *bz = bh + (int64_t)W2_DEFER_ROWS * H;  // (dz1c: this net's [h1 | dz2] pair -- the launch passes the net stride)
                for (int e = threadIdx.x; e < 16 * (H / 4); e += 64 * NW) {
                    const int row = e / (H / 4), c4 = (e % (H / 4)) * 4;
                    const float *ph = h1 + row * ld + c4, *pz = h2 + row * ld + c4;
                    const int64_t o = (grp * 16 + row) * (int64_t)H + c4;
                    *reinterpret_cast<f32x4 *>(bh + o) = f32x4{ph[0], ph[1], ph[2], ph[3]};
                    *reinterpret_cast<f32x4 *>(bz + o) = f32x4{pz[0], pz[1], pz[2], pz[3]};
                }
            }
#pragma unroll
            for (int kt = 0; kt < KT2; kt++) {  // one A fragment (8 LDS reads) feeds all NTW column tiles
                if (defer_w2) break;
                float av[8];
#pragma unroll
                for (int sidx = 0; sidx < SN; sidx++) av[sidx] = h1[(4 * sidx + g) * ld + kt * 16 + r16];
#pragma unroll
                for (int sidx = 0; sidx < SN; sidx++)
#pragma unroll
                    for (int j = 0; j < NTW; j++) aW2[kt][j] = mfma16(av[sidx], bf[j][sidx], aW2[kt][j]);
                __builtin_amdgcn_sched_barrier(0);  // do not let the scheduler hoist later tiles' reads over this one (register budget)
            }
            TMA_WTICK(6);  // P5a dW2
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                f32x4 c0 = z4, c1 = z4;
                float a0[2][4], a1[2][4];
#pragma unroll
                for (int i = 0; i < 4; i++) a0[0][i] = h2[r16 * ld + 4 * i + g], a1[0][i] = HALF ? 0.0f : h2[(16 + r16) * ld + 4 * i + g];
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const int s = S1 + NTW * NQ + j * NQ + q;
                    if (q + 1 < NQ) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int n = 16 * (q + 1) + 4 * i + g;
                            a0[(q + 1) & 1][i] = h2[r16 * ld + n], a1[(q + 1) & 1][i] = HALF ? 0.0f : h2[(16 + r16) * ld + n];
                        }
                    }
                    const f32x4 w4 = ring[s % R];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        c0 = mfma16(a0[q & 1][i], w4[i], c0);
                        if constexpr (!HALF) c1 = mfma16(a1[q & 1][i], w4[i], c1);
                    }
                    ring[s % R] = sload((s + R) % SL);
                    __builtin_amdgcn_sched_barrier(0);
                }
                dz1[j][0] = c0, dz1[j][1] = c1;
            }
        }
        __syncthreads();
        TMA_WTICK(7);  // P5b dh1
#pragma unroll
        for (int j = 0; j < NTW; j++)
#pragma unroll
            for (int mt = 0; mt < MTN; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float *pp = h1 + (mt * 16 + g * 4 + r) * ld + n_base + 16 * j + r16;
                    const float h = *pp;
                    *pp = dz1[j][mt][r] * (1.0f - h * h);
                }
        TMA_RELANE();
        // ---- P6: dW1 slice += X^T . dz1[:, slice] (each wave reads back only the columns it just wrote) ----
#pragma unroll
        for (int j = 0; j < NTW; j++) {
            const int col = n_base + 16 * j + r16;
            float bf[8];
            float c = 0.0f;
#pragma unroll
            for (int sidx = 0; sidx < SN; sidx++) {
                bf[sidx] = h1[(4 * sidx + g) * ld + col];
                c += bf[sidx];
            }
            ab1[j] += c;
            if constexpr (MAIN && KT1C < 0) {
                if (dz1c) {  // (block-uniform) first of two passes: leave the dW1 operands for PASS 2
                    float *gi = dz1c + (grp * 4 + wave) * (int64_t)(NTW * 512) + (j * 64 + lane) * 8;
                    *reinterpret_cast<f32x4 *>(gi) = f32x4{bf[0], bf[1], bf[2], bf[3]};
                    *reinterpret_cast<f32x4 *>(gi + 4) = f32x4{bf[4], bf[5], bf[6], bf[7]};
                }
            }
            if constexpr (acc_w1) {
#pragma unroll
                for (int kt = 0; kt < KT1A; kt++) {
                    const int krow = kt * 16 + r16;
#pragma unroll
                    for (int sidx = 0; sidx < SN; sidx++) {
                        const float a = krow < D ? X[(4 * sidx + g) * ldx + krow] : 0.0f;
                        aW1[kt][j] = mfma16(a, bf[sidx], aW1[kt][j]);
                    }
                }
            } else if constexpr (rmw_w1) {
                float *gW1 = slab + (IS_PI ? L.pW1t : L.vW1t);
                for (int kt = 0; kt < KT1; kt++) {
                    f32x4 t = z4;
                    const int krow = kt * 16 + r16;
#pragma unroll
                    for (int sidx = 0; sidx < 8; sidx++) {
                        const float a = krow < D ? X[(4 * sidx + g) * ldx + krow] : 0.0f;
                        t = mfma16(a, bf[sidx], t);
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int k = kt * 16 + g * 4 + r;
                        if (k < D) gW1[(int64_t)k * H + col] += t[r];  // this wave is the only writer of these slab columns
                    }
                }
            }
        }
        if (threadIdx.x < M) row_off_next[threadIdx.x] = noff;
        __syncthreads();
        TMA_WTICK(8);  // P6 dz1 + dW1
    }
#undef TMA_RELANE
    // ---- store this block's slab (every parameter of the net has exactly one owning wave) ----
    float *gW1 = slab + (IS_PI ? L.pW1t : L.vW1t), *gb1 = slab + (IS_PI ? L.pb1 : L.vb1);
    float *gW2 = slab + (IS_PI ? L.pW2t : L.vW2t), *gb2 = slab + (IS_PI ? L.pb2 : L.vb2);
    float *gW3 = slab + (IS_PI ? L.pW3t : L.vW3t), *gb3 = slab + (IS_PI ? L.pb3 : L.vb3);
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        const int col = n_base + 16 * j + r16;
        if constexpr (MAIN) {
            if (!(HALF && (KT1C > 2 || dz1c != nullptr))) {  // (deferred: wide_small_reduce_kernel writes dW2 straight into the gradient)
#pragma unroll
                for (int kt = 0; kt < KT2; kt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) gW2[(int64_t)(kt * 16 + g * 4 + r) * H + col] = aW2[kt][j][r];
            }
        }
        if constexpr (acc_w1) {
#pragma unroll
            for (int kt = 0; kt < KT1A; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = kt * 16 + g * 4 + r;
                    if (k < D) gW1[(int64_t)k * H + col] = aW1[kt][j][r];
                }
        }
        if constexpr (MAIN) {
            float v1 = ab1[j], v2 = ab2[j];
            v1 += __shfl_xor(v1, 16, 64), v1 += __shfl_xor(v1, 32, 64);
            v2 += __shfl_xor(v2, 16, 64), v2 += __shfl_xor(v2, 32, 64);
            if (g == 0) gb1[col] = v1, gb2[col] = v2;
#pragma unroll
            for (int q = 0; q < NT3; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = n_base + 16 * j + g * 4 + r, n = 16 * q + r16;
                    if (n < NOUT) gW3[(int64_t)k * NOUT + n] = aW3[j][q][r];
                }
        }
    }
    TMA_WTICK(9);  // slab store issue
#ifdef TMA_WIDE_PHASE_TICKS
    __syncthreads();
    if (threadIdx.x == 0 && block_net == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&g_wide_ticks[IS_PI ? 0 : 1][i], wtick_lds[i]);
#endif
    if constexpr (!MAIN) return;
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < NT3; q++) {
            float v = ab3[q];
            v += __shfl_xor(v, 16, 64), v += __shfl_xor(v, 32, 64);
            const int n = 16 * q + r16;
            if (g == 0 && n < NOUT) gb3[n] = v;
        }
    }
    if constexpr (IS_PI && CONT) {  // log_std gradient: the two head waves hold one row tile each
        float v0 = dlsd[0], v1 = dlsd[1];
        v0 += __shfl_xor(v0, 16, 64), v0 += __shfl_xor(v0, 32, 64);
        v1 += __shfl_xor(v1, 16, 64), v1 += __shfl_xor(v1, 32, 64);
        if (wave == 1 && g == 0) scratch[r16] = v0, scratch[16 + r16] = v1;
        __syncthreads();
        if (wave == 0 && g == 0) {
            if (r16 < A) slab[L.log_std + r16] = v0 + scratch[r16];
            if (16 + r16 < A) slab[L.log_std + 16 + r16] = v1 + scratch[16 + r16];
        }
    }
    // loss statistics of the head waves -> this pair's slot
    double sv[5] = {st.a, st.ent, st.kl, st.clip, st.n};
#pragma unroll
    for (int q = 0; q < 5; q++)
        for (int o = 32; o > 0; o >>= 1) sv[q] += __shfl_down(sv[q], o, 64);
    __syncthreads();
    double *red = reinterpret_cast<double *>(smem);
    constexpr int NWS = (NW == 8 && IS_PI && !CONT) ? NW : 2;  // waves that carry loss statistics (P3b)
    if (lane == 0 && wave < NWS)
        for (int q = 0; q < 5; q++) red[wave * 5 + q] = sv[q];
    __syncthreads();
    if (threadIdx.x < 5) {
        double ssum = red[threadIdx.x] + red[5 + threadIdx.x];
#pragma unroll
        for (int w = 2; w < NWS; w++) ssum += red[w * 5 + threadIdx.x];
        const int q = IS_PI ? (threadIdx.x == 0 ? 0 : threadIdx.x + 1) : (threadIdx.x == 0 ? 1 : -1);
        if (q >= 0) stat_slot[q] += ssum;
    }
}

template <bool CONT, int NTW, int KT1C, int PASS = 0, int NQ1C = 0, bool HALF = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW / 4) void ppo_grad_wide_kernel(const float *__restrict__ params, PLayout L, Rollout rb, Minibatch mb, HParams hp,
                                                               const float *__restrict__ ws_adv, float *__restrict__ slabs,
                                                               double *__restrict__ stat_slots, int n_pi, float *__restrict__ dz1,
                                                               int64_t dz1_net_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // blocks [0, n_pi): policy net, [n_pi, gridDim.x): value net (the policy net's row group costs more: it gets more blocks)
    const bool is_pi = (int)blockIdx.x < n_pi;
    const int b = is_pi ? blockIdx.x : blockIdx.x - n_pi, nb = is_pi ? n_pi : (int)gridDim.x - n_pi;
    float *slab = slabs + (int64_t)b * L.P;
    double *slot = stat_slots + (int64_t)b * 8;
    if (is_pi) grad_wide_body<CONT, true, NTW, KT1C, PASS, NQ1C, HALF, NW>(params, L, rb, mb, hp, ws_adv, slab, slot, smem, nb, b, dz1);
    else grad_wide_body<CONT, false, NTW, KT1C, PASS, NQ1C, HALF, NW>(params, L, rb, mb, hp, ws_adv, slab, slot, smem, nb, b, dz1 ? dz1 + dz1_net_stride : nullptr);
}

#include "tma_wide_bf16.h"

// zero the layer-1 weight columns of every slab when they are accumulated in place (observations wider than 32)
__global__ void slab_zero_w1_kernel(float *slabs, int n_slabs, PLayout L) {
    const int per = L.D * L.H;
    const int64_t tot = (int64_t)n_slabs * 2 * per;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / (2 * per);
        const int x = (int)(e - b * 2 * per);
        slabs[b * L.P + (x < per ? L.pW1t + x : L.vW1t + (x - per))] = 0.0f;
    }
}

static int grad_wide_smem_bytes(const PLayout &L, int nw = 4) {
    const int ldx = ((L.D + 3) & ~3) + 2, ld = L.H + 2;
    return (32 * (ldx + 2 * ld + 34 + 4) + 2 * 32 + 64 + nw * 2 * 2 * 256 + 2 * L.H + 32 + 2 * 32 + 32 * 32) * 4;  // (last terms: row_off_next, the Box heads' action tile)
}

// grad[e] += sum over blocks of slab[b][e].  64 params x 4 slab quarters per block, partial sums folded through LDS in a
// fixed order -> bitwise reproducible, and enough independent loads in flight to run at L2 speed.
// Parameters in [vf_begin, vf_end) (the value net) are summed over n_slabs_vf slabs, all others over n_slabs (the policy net may
// run on more blocks than the value net -- the bf16 wide kernel balances the two by their cost per row group).
// sq_part (optional): sq_part[blockIdx.x] = sum of squares of this block's 64 finished gradient entries (f64, fixed order) -- lets
// tma_ppo_adam_step_local skip its own pass over the gradient for the norm.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ slabs, int n_slabs_pi, int P, float *__restrict__ grad,
                                                          int n_slabs_vf = -1, int vf_begin = 0, int vf_end = 0, double *__restrict__ sq_part = nullptr,
                                                          int overwrite = 0, PeerPush push = PeerPush{}) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int n_slabs = (n_slabs_vf >= 0 && e >= vf_begin && e < vf_end) ? n_slabs_vf : n_slabs_pi;
    const int per = (n_slabs + 3) >> 2;
    const int b0 = q * per, b1 = (b0 + per < n_slabs) ? b0 + per : n_slabs;
    float s = 0.0f;
    if (e < P) {
        int b = b0;
        for (; b + 32 <= b1; b += 32) {  // (a quarter of 128 slabs in ONE batch of loads: one memory round trip; same summation order)
            float t[32];
#pragma unroll
            for (int u = 0; u < 32; u++) t[u] = slabs[(int64_t)(b + u) * P + e];
#pragma unroll
            for (int u = 0; u < 32; u++) s += t[u];
        }
        for (; b + 8 <= b1; b += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = slabs[(int64_t)(b + u) * P + e];
#pragma unroll
            for (int u = 0; u < 8; u++) s += t[u];
        }
        for (; b < b1; b++) s += slabs[(int64_t)b * P + e];
    }
    part[q][lane] = s;
    __syncthreads();
    if (q == 0) {
        float gnew = 0.0f;
        if (e < P) {
            // (overwrite: the previous minibatch's gradient is still there -- its optimizer step ran inside the gradient launch, AdamFold)
            gnew = (overwrite ? 0.0f : grad[e]) + (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);
            grad[e] = gnew;
            // data-parallel peer exchange (tma_p2p.h): this rank's reduced gradient straight into every rank's inbox -- the all-reduce's send half
            if (push.world > 0) p2p_push(push, e, __float_as_uint(gnew));
        }
        if (sq_part) {
            double sq = (double)gnew * (double)gnew;
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
            if (lane == 0) sq_part[blockIdx.x] = sq;
        }
    }
}

// Small minibatches of the f32 256-wide path (half groups with the dW2 deferral buffer, <= 1024 samples): ONE launch finishes the gradient.
//  * workgroups [0, 128): dW2 of one net as a GEMM over the whole minibatch -- block (net, ti, tj) forms rows [16 ti, 16 ti + 16) x columns
//    [64 tj, 64 tj + 64) of H1^T . DZ2 on sixteen waves: wave (u, ks) the 16-column tile u over the ks-th quarter of the samples (operands
//    straight from the L2-resident buffer the gradient kernel left, sixteen k-steps of four samples per batch of loads -- at 256 samples ONE
//    memory round trip per wave), the four partial tiles of a column tile added in slice order through LDS, the result added to the gradient,
//    one sum-of-squares partial per (row, 64 columns);
//  * workgroups behind them: slab_reduce_kernel's work on the parameters that are NOT a W2 entry (W1, biases, head, log_std: 64 consecutive
//    entries of that compacted index space per block, sixteen slab ranges folded through LDS in a fixed order).
// The partials fill the same ceil(P / 64) slots slab_reduce_kernel fills -- [0, 2048) the W2 strips, then the compacted chunks -- and the
// optimizer kernel only ever sums all of them, in a fixed order: deterministic; the norm's last bits differ from the slab path's (another
// partition of the same squares).
__global__ __launch_bounds__(1024) void wide_small_reduce_kernel(const float *__restrict__ slabs, int n_slabs_pi, int n_slabs_vf, PLayout L,
                                                                 const float *__restrict__ w2buf, int rows, float *__restrict__ grad,
                                                                 double *__restrict__ sq_part) {
    constexpr int H = 256;
    __shared__ float tile[4][16][68];  // [sample quarter][row][64 columns + pad]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x < 128) {
        const int net = blockIdx.x >> 6, ti = (blockIdx.x >> 2) & 15, tj = blockIdx.x & 3;
        const int r16 = lane & 15, g = lane >> 4, u = wave & 3, ks = wave >> 2;
        const float *A0 = w2buf + (int64_t)(2 * net) * (W2_DEFER_ROWS * H) + 16 * ti + r16;
        const float *B0 = w2buf + (int64_t)(2 * net + 1) * (W2_DEFER_ROWS * H) + 64 * tj + 16 * u + r16;
        // wave w finishes row w at the end: lane l = column 64 tj + l (256 contiguous bytes of the gradient; its old value is requested now)
        const int row = wave, k = 16 * ti + row;
        const int64_t e = (int64_t)(net == 0 ? L.pW2t : L.vW2t) + (int64_t)k * H + 64 * tj + lane;
        const float gold = grad[e];
        f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const int nq = rows >> 2, per = (nq + 3) >> 2;  // k-steps of four samples; this wave's are [ks per, min(nq, (ks + 1) per))
        const int q_end = (ks + 1) * per < nq ? (ks + 1) * per : nq;
        auto run = [&](auto kb) {  // batches of KB k-steps: all 2 KB loads of a batch in flight together, then its MFMAs (same order for every KB)
            constexpr int KB = decltype(kb)::value;
            for (int q0 = ks * per; q0 < q_end; q0 += KB) {
                float a[KB], b[KB];
#pragma unroll
                for (int v = 0; v < KB; v++) {
                    const bool in = q0 + v < q_end;  // (rows behind the minibatch hold an earlier launch's data)
                    const int64_t o = (int64_t)(4 * (in ? q0 + v : 0) + g) * H;
                    a[v] = in ? A0[o] : 0.0f, b[v] = in ? B0[o] : 0.0f;
                }
#pragma unroll
                for (int v = 0; v < KB; v++) acc = mfma16(a[v], b[v], acc);
            }
        };
        if (per <= 16) run(std::integral_constant<int, 16>{});  // <= 256 samples: one batch
        else run(std::integral_constant<int, 32>{});
#pragma unroll
        for (int r = 0; r < 4; r++) tile[ks][4 * g + r][16 * u + r16] = acc[r];
        __syncthreads();
        const float gnew = gold + (((tile[0][row][lane] + tile[1][row][lane]) + tile[2][row][lane]) + tile[3][row][lane]);
        grad[e] = gnew;
        double sq = (double)gnew * (double)gnew;
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
        if (lane == 0) sq_part[net * 1024 + k * 4 + tj] = sq;
        return;
    }
    float(*part)[64] = reinterpret_cast<float(*)[64]>(&tile[0][0][0]);  // [16 slab ranges][64 lanes]
    const int chunk = blockIdx.x - 128, q = wave;
    const int s = chunk * 64 + lane, P_small = L.P - 2 * H * H;
    const int a0 = L.pW2t, a1 = L.vW2t - H * H;  // compacted index space: [0, a0) | the flat entries between the two W2 blocks | those behind the second
    const int e = s < a0 ? s : (s < a1 ? s + H * H : s + 2 * H * H);
    const bool live = s < P_small;
    const int n_slabs = (live && e >= L.vW1t && e < L.log_std) ? n_slabs_vf : n_slabs_pi;
    const int per = (n_slabs + 15) >> 4;
    const int b0 = q * per < n_slabs ? q * per : n_slabs, b1 = (b0 + per < n_slabs) ? b0 + per : n_slabs;
    float sum = 0.0f;
    const float gold = (live && q == 0) ? grad[e] : 0.0f;
    if (live) {
        float t[4];  // (<= 64 slabs: a range holds at most four)
#pragma unroll
        for (int v = 0; v < 4; v++) t[v] = b0 + v < b1 ? slabs[(int64_t)(b0 + v) * L.P + e] : 0.0f;
#pragma unroll
        for (int v = 0; v < 4; v++) sum += t[v];
        for (int b = b0 + 4; b < b1; b++) sum += slabs[(int64_t)b * L.P + e];
    }
    part[q][lane] = sum;
    __syncthreads();
    if (q == 0) {
        float gnew = 0.0f;
        if (live) {
            float tot = part[0][lane];
#pragma unroll
            for (int v = 1; v < 16; v++) tot += part[v][lane];
            gnew = gold + tot;
            grad[e] = gnew;
        }
        double sq = (double)gnew * (double)gnew;
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
        if (lane == 0) sq_part[2048 + chunk] = sq;
    }
}

// ------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float *__restrict__ grad, int P, float scale, double *partials) {
    __shared__ double red[4];
    double s = 0.0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P; e += gridDim.x * blockDim.x) {
        const float gv = grad[e] * scale;
        s += (double)gv * (double)gv;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ params, float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v,
                                                   int P, float scale, const double *__restrict__ partials, int n_partials, float max_norm, float lr_step,
                                                   float beta1, float beta2, float bc2_sqrt, float eps, double *norm_out) {
    double tot = 0.0;
    for (int b = 0; b < n_partials; b++) tot += partials[b];
    const float total_norm = (float)sqrt(tot);
    float coef = max_norm / (total_norm + 1e-6f);  // torch.nn.utils.clip_grad_norm_
    coef = coef > 1.0f ? 1.0f : coef;
    if (max_norm <= 0.0f) coef = 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        norm_out[0] = (double)total_norm;
        norm_out[1] = (double)coef;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P; e += gridDim.x * blockDim.x) {
        const float gv = (grad[e] * scale) * coef;
        grad[e] = 0.0f;  // ready for the next minibatch
        float mm = m[e], vv = v[e];
        mm = mm + (gv - mm) * (1.0f - beta1);         // exp_avg.lerp_(grad, 1 - beta1)
        vv = vv * beta2 + (gv * gv) * (1.0f - beta2);  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        m[e] = mm;
        v[e] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        params[e] = params[e] - lr_step * (mm / denom);  // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
}



// sum-of-squares partials of the (scaled) gradient, one per 64 parameters -- the same values, in the same order, slab_reduce_kernel
// leaves for an un-reduced gradient; used when the gradient was all-reduced (or accumulated) after the reduction kernel ran
__global__ __launch_bounds__(64) void grad_sumsq64_kernel(const float *__restrict__ grad, int P, float scale, double *__restrict__ sq_part) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const float gv = e < P ? grad[e] * scale : 0.0f;
    double sq = (double)gv * (double)gv;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
    if (threadIdx.x == 0) sq_part[blockIdx.x] = sq;
}

// The receive half of the peer exchange fused into the same pass: grad[e] = sum over the ranks (rank order) of the words their slab
// reductions stored into this rank's inbox, then the partials of the scaled sum exactly as above.
__global__ __launch_bounds__(64) void grad_pull_sumsq64_kernel(float *__restrict__ grad, int P, float scale, double *__restrict__ sq_part, PeerPull pull) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    float gsum = 0.0f;
    if (e < P) {
        gsum = p2p_pull_f32(pull, e);
        grad[e] = gsum;
    }
    const float gv = gsum * scale;
    double sq = (double)gv * (double)gv;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
    if (threadIdx.x == 0) sq_part[blockIdx.x] = sq;
}

// tma_ppo_adam_step_local, H == 64 fast path: one thread per parameter over ceil(P / 256) blocks.  The norm comes from the
// sum-of-squares partials slab_reduce_kernel left (every block folds them in the same fixed order), and each thread writes its
// updated parameter to the flat buffer AND to its derived copies / image slots -- no single-block optimizer, no refresh launch.
// (p_src, m_src, v_src): the state before the step -- the same buffers, or the other half of the AdamFold double buffer.
__global__ __launch_bounds__(256) void adam_scatter_h64_kernel(float *params, float *__restrict__ grad, float *m, float *v, PLayout L,
                                                               const double *__restrict__ sq_part, int n_part, float max_norm, float lr_step,
                                                               float beta1, float beta2, float bc2_sqrt, float eps, double *norm_out, float scale,
                                                               const float *p_src, const float *m_src, const float *v_src) {
    __shared__ double red[4];
    __shared__ float coef_s;
    double a = (int)threadIdx.x < n_part ? sq_part[threadIdx.x] : 0.0;
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = ((red[0] + red[1]) + red[2]) + red[3];
        const float total_norm = (float)sqrt(tot);
        float coef = max_norm / (total_norm + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
        if (max_norm <= 0.0f) coef = 1.0f;
        coef_s = coef;
        if (blockIdx.x == 0) norm_out[0] = (double)total_norm, norm_out[1] = (double)coef;
    }
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= L.P) return;
    const float gv = (grad[e] * scale) * coef_s;
    grad[e] = 0.0f;
    float mm = m_src[e], vv = v_src[e];
    const float pn = adam_update_h64(p_src[e], gv, mm, vv, beta1, beta2, 1.0f / bc2_sqrt, eps, lr_step);
    m[e] = mm;
    v[e] = vv;
    params[e] = pn;
    if (e < L.log_std) scatter_derived_h64(params, L, e, pn);
}

// Column-parallel layouts (H = 128 / 192 / 256): derived locations of trainable parameter e -- the [out][in] f32 copies of W2 / W3
// and, in bf16 mode, its slots in the fragment-major images (inverse of build_bf16_images_kernel).
__device__ __forceinline__ void scatter_derived_wide(float *params, const PLayout &L, int e, float val) {
    const int D = L.D, H = L.H;
    const bool vf = e >= L.vW1t && e < L.log_std;
    const int base = vf ? L.vW1t : L.pW1t, n_out = vf ? 1 : L.A;
    bf16_t *img = L.bf16 ? reinterpret_cast<bf16_t *>(params + (vf ? L.bf_vf : L.bf_pi)) : nullptr;
    const BfNet B = bf_net_layout(D, H, n_out);
    const int KS2 = H >> 5, KS1 = ((D + 31) & ~31) >> 5;
    const bf16_t bv = (bf16_t)val;
    if (L.split) {  // mfma_dtype = 2: the same slots in the three planes of the split images (hi, mid, lo: val == hi + mid + lo)
        bf16_t *sp = reinterpret_cast<bf16_t *>(params + (vf ? L.sp_vf : L.sp_pi));
        const float r1 = val - (float)bv;
        const bf16_t bm = (bf16_t)r1, bl = (bf16_t)(r1 - (float)bm);
        auto put = [&](int slot) { sp[slot] = bv, sp[B.size + slot] = bm, sp[2 * B.size + slot] = bl; };
        int y = e - base;
        if (y < D * H) {
            const int k = y / H, n = y - k * H;
            put(B.fW1 + (((n >> 4) * KS1 + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7));
        } else if ((y -= D * H + H) >= 0 && y < H * H) {
            const int k = y / H, n = y - k * H;
            put(B.fW2 + (((n >> 4) * KS2 + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7));
            put(B.bW2 + (((k >> 4) * KS2 + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (k & 15)) * 8 + (n & 7));
        } else if ((y -= H * H + H) >= 0 && y < H * n_out) {
            const int k = y / n_out, a = y - k * n_out;
            put(B.fW3 + (((a >> 4) * KS2 + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (a & 15)) * 8 + (k & 7));
            put(B.bW3 + ((k >> 4) * 64 + ((a >> 3) & 3) * 16 + (k & 15)) * 8 + (a & 7));
        }
    }
    int x = e - base;
    if (x < D * H) {  // W1t[k][n]
        const int k = x / H, n = x - k * H;
        if (img) img[B.fW1 + (((n >> 4) * KS1 + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7)] = bv;
        if (L.fr1_pi >= 0)
            params[(vf ? L.fr1_vf : L.fr1_pi) + (((n >> 4) * ((D + 15) / 16) + (k >> 4)) * 64 + (k & 3) * 16 + (n & 15)) * 4 + ((k >> 2) & 3)] = val;
        return;
    }
    x -= D * H;
    if (x < H) return;
    x -= H;
    if (x < H * H) {  // W2t[k][n]
        const int k = x / H, n = x - k * H;
        params[(vf ? L.vW2 : L.pW2) + n * H + k] = val;
        if (L.fr_pi >= 0) {  // f32 fragment images: forward slot of (k, n), input-gradient slot of (k' = k, n)
            float *fr = params + (vf ? L.fr_vf : L.fr_pi);
            const int NQ = H >> 4;
            fr[(((n >> 4) * NQ + (k >> 4)) * 64 + (k & 3) * 16 + (n & 15)) * 4 + ((k >> 2) & 3)] = val;
            fr[H * H + (((k >> 4) * NQ + (n >> 4)) * 64 + (n & 3) * 16 + (k & 15)) * 4 + ((n >> 2) & 3)] = val;
        }
        if (img) {
            img[B.fW2 + (((n >> 4) * KS2 + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7)] = bv;
            img[B.bW2 + (((k >> 4) * KS2 + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (k & 15)) * 8 + (n & 7)] = bv;
        }
        return;
    }
    x -= H * H;
    if (x < H) return;
    x -= H;
    if (x < H * n_out) {  // W3t[k][a]
        const int k = x / n_out, a = x - k * n_out;
        params[(vf ? L.vW3 : L.pW3) + a * H + k] = val;
        if (img) {
            img[B.fW3 + (((a >> 4) * KS2 + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (a & 15)) * 8 + (k & 7)] = bv;
            img[B.bW3 + ((k >> 4) * 64 + ((a >> 3) & 3) * 16 + (k & 15)) * 8 + (a & 7)] = bv;
        }
    }
}

// tma_ppo_adam_step_local for the column-parallel layouts: as adam_scatter_h64_kernel, with up to WIDE_SQ_SLOTS norm partials
__global__ __launch_bounds__(256) void adam_scatter_wide_kernel(float *__restrict__ params, float *__restrict__ grad, float *__restrict__ m,
                                                                float *__restrict__ v, PLayout L, const double *__restrict__ sq_part, int n_part,
                                                                float max_norm, float lr_step, float beta1, float beta2, float bc2_sqrt, float eps,
                                                                double *norm_out, float scale) {
    __shared__ double red[4];
    __shared__ float coef_s;
    // this thread's element first (its four loads fly under the fold of the norm partials), then the partials sixteen loads at a time:
    // one memory round trip per batch instead of one per partial (a thread of the 137 k-parameter net folds 9 of them); same order of additions
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool live = e < L.P;
    const float g_e = live ? grad[e] : 0.0f, m_e = live ? m[e] : 0.0f, v_e = live ? v[e] : 0.0f, p_e = live ? params[e] : 0.0f;
    double a = 0.0;
    for (int b0 = threadIdx.x; b0 < n_part; b0 += 256 * 16) {
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; u++) t[u] = b0 + 256 * u < n_part ? sq_part[b0 + 256 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (b0 + 256 * u < n_part) a += t[u];
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = ((red[0] + red[1]) + red[2]) + red[3];
        const float total_norm = (float)sqrt(tot);
        float coef = max_norm / (total_norm + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
        if (max_norm <= 0.0f) coef = 1.0f;
        coef_s = coef;
        if (blockIdx.x == 0) norm_out[0] = (double)total_norm, norm_out[1] = (double)coef;
    }
    __syncthreads();
    if (!live) return;
    const float gv = (g_e * scale) * coef_s;
    grad[e] = 0.0f;
    float mm = m_e, vv = v_e;
    // (round 6: adam_update_h64 -- hardware sqrt / rcp + explicit FMAs, the H = 64 kernels' routine -- here as well: the 256-wide persistent epoch
    //  kernel, tma_h256p.hip, runs it on every workgroup each step, and the two paths share one arithmetic)
    const float pn = adam_update_h64(p_e, gv, mm, vv, beta1, beta2, 1.0f / bc2_sqrt, eps, lr_step);
    m[e] = mm;
    v[e] = vv;
    params[e] = pn;
    if (e < L.log_std) scatter_derived_wide(params, L, e, pn);
}

// small policies (P <= 32768): clip_grad_norm_ + Adam in ONE single-block launch (the norm needs no second kernel)
__global__ __launch_bounds__(1024) void opt_small_kernel(float *__restrict__ params, float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v,
                                                         PLayout L, float scale, float max_norm, float lr_step, float beta1, float beta2, float bc2_sqrt,
                                                         float eps, double *norm_out) {
    __shared__ double red[16];
    __shared__ float coef_s;
    const int P = L.P;
    // P <= 10240 (the 64x64 nets: 9350): every thread keeps its <= 10 elements of grad / m / v / params in registers, so all four
    // streams are in flight together and the second pass needs no global load (one memory round trip instead of two).
    constexpr int NE = 10;
    const bool small = P <= NE * 1024;
    float rg[NE], rm[NE], rv[NE], rp[NE];
    double sq = 0.0;
    if (small) {
#pragma unroll
        for (int i = 0; i < NE; i++) {
            const int e = threadIdx.x + 1024 * i;
            const bool ok = e < P;
            rg[i] = ok ? grad[e] * scale : 0.0f;
            rm[i] = ok ? m[e] : 0.0f;
            rv[i] = ok ? v[e] : 0.0f;
            rp[i] = ok ? params[e] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NE; i++) sq += (double)rg[i] * (double)rg[i];
    } else {
        for (int e = threadIdx.x; e < P; e += 1024) {
            const float gv = grad[e] * scale;
            sq += (double)gv * (double)gv;
        }
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += red[w];
        const float total_norm = (float)sqrt(tot);
        float coef = max_norm / (total_norm + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
        if (max_norm <= 0.0f) coef = 1.0f;
        coef_s = coef;
        norm_out[0] = (double)total_norm;
        norm_out[1] = (double)coef;
    }
    __syncthreads();
    const float coef = coef_s;
    if (small) {
#pragma unroll
        for (int i = 0; i < NE; i++) {
            const int e = threadIdx.x + 1024 * i;
            if (e < P) {
                const float gv = rg[i] * coef;
                grad[e] = 0.0f;
                float mm = rm[i], vv = rv[i];
                mm = mm + (gv - mm) * (1.0f - beta1);
                vv = vv * beta2 + (gv * gv) * (1.0f - beta2);
                m[e] = mm;
                v[e] = vv;
                const float denom = sqrtf(vv) / bc2_sqrt + eps;
                params[e] = rp[i] - lr_step * (mm / denom);
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < P; e += 1024) {
        const float gv = (grad[e] * scale) * coef;
        grad[e] = 0.0f;
        float mm = m[e], vv = v[e];
        mm = mm + (gv - mm) * (1.0f - beta1);
        vv = vv * beta2 + (gv * gv) * (1.0f - beta2);
        m[e] = mm;
        v[e] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        params[e] = params[e] - lr_step * (mm / denom);
    }
}

static inline PLayout layout_of(const tma_policy_dims *d) { return make_layout(d->obs_dim, d->hidden, d->act_dim, d->continuous, d->mfma_dtype); }

// everything derived from the trainable region: [out][in] copies, H == 64 LDS images, bf16 fragment images
static int launch_sync(float *params, const PLayout &L, hipStream_t s) {
    const int total = 2 * L.H * L.H + L.A * L.H + L.H + (L.img_pi >= 0 ? 2 * IMG_FLOATS : 0);
    sync_transposed_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s>>>(params, L);
    TMA_LAUNCH_CHECK();
    if (L.bf16) {
        build_bf16_images_kernel<<<dim3(256), dim3(256), 0, s>>>(params, L);
        TMA_LAUNCH_CHECK();
    }
    if (L.fr_pi >= 0) {
        build_f32_frag_images_kernel<<<dim3((unsigned)ceil_div(4 * L.H * L.H, 256)), dim3(256), 0, s>>>(params, L);
        TMA_LAUNCH_CHECK();
    }
    if (L.split) return tma_launch_build_split3(params, L, s);
    return TMA_OK;
}

// profiling aid (tma_debug_time_grad_kernel): HIP events around the dominant kernel of tma_ppo_minibatch_grad, on its stream
static bool g_time_grad = false;
static hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
static bool g_ev_valid = false;
struct GradTimer {
    hipStream_t s;
    bool on;
    explicit GradTimer(hipStream_t st) : s(st), on(g_time_grad && g_ev0 && g_ev1) {
        if (on) (void)hipEventRecord(g_ev0, s);
    }
    ~GradTimer() {
        if (on) {
            (void)hipEventRecord(g_ev1, s);
            g_ev_valid = true;
        }
    }
};

// where slab_reduce_kernel leaves its sum-of-squares partials (one per 64 parameters) for tma_ppo_adam_step_local
// second half of the (parameters, exp_avg, exp_avg_sq) double buffer of the optimizer step folded into the H = 64 gradient launches
// (AdamFold, tma_ppo_train_epoch_local): 3 x P floats behind everything else in the workspace
static inline int64_t fold_state_offset(const PLayout &L) {
    return (WS_SLABS + (int64_t)slab_cap(L) * L.P * 4 + OFFS_CAP * 4 + EPOCH_PART_BYTES + WIDE_SQ_SLOTS * 8 + dz1_cache_bytes(L) + 15) & ~(int64_t)15;
}
// (also the snapshot the persistent epoch kernels' fallback restores: H = 64 fast-path layouts and the 256-wide layouts tma_h256p.hip takes)
static inline bool h256p_layout(const PLayout &L) { return !L.bf16 && L.fr_pi >= 0 && L.H == 256 && !L.cont && L.A <= 16 && L.D <= 32; }
static inline int64_t fold_state_bytes(const PLayout &L) { return (L.img_pi >= 0 || h256p_layout(L)) ? 3 * (((int64_t)L.P + 3) & ~(int64_t)3) * 4 : 0; }

static double *sq_partials(char *ws, const PLayout &L) {
    const int n = (int)ceil_div(L.P, 64);
    if (n <= 256) return reinterpret_cast<double *>(ws + WS_NORM_PART);
    if (n > WIDE_SQ_SLOTS) return nullptr;
    return reinterpret_cast<double *>(ws + WS_SLABS + (int64_t)slab_cap(L) * L.P * 4 + OFFS_CAP * 4 + EPOCH_PART_BYTES);
}

static int check_dims(const tma_policy_dims *d) {
    if (!d) return fail(TMA_ERR_INVALID, "policy dims is null");
    if (d->obs_dim < 1 || d->obs_dim > 4096) return fail(TMA_ERR_INVALID, "obs_dim out of range: %d", d->obs_dim);
    if (d->hidden < 64 || d->hidden % 64 != 0 || d->hidden > 1024)
        return fail(TMA_ERR_INVALID, "hidden width must be a multiple of 64 in [64, 1024] (got %d)", d->hidden);
    if (d->continuous) {
        if (d->act_dim < 1 || d->act_dim > 32) return fail(TMA_ERR_INVALID, "Box action dim must be in [1, 32] (got %d)", d->act_dim);
    } else if (d->act_dim < 2 || d->act_dim > 16)
        return fail(TMA_ERR_INVALID, "Discrete action count must be in [2, 16] (got %d)", d->act_dim);
    if (d->device < -1) return fail(TMA_ERR_INVALID, "device must be >= 0, or -1 for the calling thread's current device (got %d)", d->device);
    if (d->mfma_dtype < 0 || d->mfma_dtype > 2) return fail(TMA_ERR_INVALID, "mfma_dtype must be 0 (f32), 1 (bf16) or 2 (bf16 x 3), got %d", d->mfma_dtype);
    if (d->mfma_dtype == 2 && (d->hidden != 256 || d->continuous || d->obs_dim > 32))
        return fail(TMA_ERR_INVALID, "mfma_dtype 2 (three-term bf16 split of the f32 update) covers Discrete heads, hidden 256 and up to 32 observations (got hidden %d, obs %d%s)",
                    d->hidden, d->obs_dim, d->continuous ? ", Box actions" : "");
    if (d->mfma_dtype == 1) {
        if (d->hidden != 128 && d->hidden != 192 && d->hidden != 256)
            return fail(TMA_ERR_INVALID, "the bf16 MFMA path covers hidden widths 128 / 192 / 256 (got %d)", d->hidden);
        if (grad_wide_bf_smem_bytes(d->obs_dim, d->hidden, 2) > 160 * 1024)
            return fail(TMA_ERR_INVALID, "obs_dim %d too wide for the bf16 LDS tile", d->obs_dim);
    }
    return TMA_OK;
}

// entry points that launch kernels: validate the shape, then make the policy's device current on the calling thread
static int enter(const tma_policy_dims *d) {
    const int rc = check_dims(d);
    if (rc) return rc;
    if (d->device >= 0) {
        int cur = -1;
        TMA_HIP(hipGetDevice(&cur));
        if (cur != d->device) TMA_HIP(hipSetDevice(d->device));
    }
    return TMA_OK;
}

static int fwd_smem_bytes(const PLayout &L, int wpb) {
    const int ldx = ((L.D + 3) & ~3) + 2, ld = L.H + 2;
    return wpb * (16 * (ldx + 2 * ld) + 32) * 4;
}
static int grad_smem_bytes(const PLayout &L, int wpb) {
    const int ldx = ((L.D + 3) & ~3) + 2, ld = L.H + 2;
    return wpb * (16 * (ldx + 2 * ld + 34) + 16 * 8) * 4;
}


// ------------------------------------------------------------------------------------------
// H = 64 forward fast path: both nets' forward weight images staged in LDS once per block (float4 copies of the images
// tma_policy_sync maintains), B operands by ds_read_b128.  MODE 0 also folds the timeout bootstrap of the PREVIOUS vector
// step (rewards_prev[i] += gamma * V(terminal_obs_prev[i]) where truncated_prev[i]) into the same launch.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void policy_fwd_h64_kernel(const float *__restrict__ params, PLayout L, const float *__restrict__ obs, int64_t n,
                                                             uint32_t rng_seed, uint32_t rng_step, uint32_t env_offset, int deterministic,
                                                             int32_t *__restrict__ actions_out, float *__restrict__ values_out,
                                                             float *__restrict__ logp_out, const float *__restrict__ boot_obs,
                                                             const uint8_t *__restrict__ boot_trunc, float gamma, float *__restrict__ boot_rewards) {
    // (round 3) the transposed register chain of the update kernels (h64t_forward, tma_h64_tile.h): lane (g, s) works for row s of its tile,
    // reads that row's observation features 4 ks + g straight from global memory into MFMA B operands, and nothing but the weight images
    // lives in LDS.  Same instructions as the forward half of the update's tile: a rollout's log-probabilities and values are bit for bit
    // what the first update epoch recomputes.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int D = L.D, A = L.A, KS1 = (D + 3) >> 2;
    float *vimg = smem, *pimg = smem + FWD_IMG;
    stage_fwd_image(params + L.img_vf, vimg);
    if constexpr (MODE == 0) stage_fwd_image(params + L.img_pi, pimg);
    __syncthreads();
    const int64_t n_tiles = (n + 15) >> 4;
    auto features = [&](const float *src, int64_t row, bool ok, float (&xb)[4]) {  // (clamped addresses, masked values: no load sits behind a branch)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int c = 4 * ks + g;
            const float x = src[(ok ? row : 0) * D + (c < D ? c : 0)];
            xb[ks] = (ok && c < D) ? x : 0.0f;
        }
    };
    if constexpr (MODE != 2) {
        for (int64_t tile = (int64_t)blockIdx.x * wpb + wave; tile < n_tiles; tile += (int64_t)gridDim.x * wpb) {
            const int64_t row = (tile << 4) + r16;
            const bool ok = row < n;
            float xb[4];
            features(obs, row, ok, xb);
            f32x4 o0, o1;
            h64t_forward<4>(vimg, vimg + IMG_FWD_FLOATS, vimg + IMG_FWD_FLOATS + 64, vimg + IMG_FWD_FLOATS + 128, xb, KS1, o0, o1, lane);
            if (ok && g == 0) values_out[row] = o0[0] + o1[0];
            if constexpr (MODE == 0) {
                h64t_forward<4>(pimg, pimg + IMG_FWD_FLOATS, pimg + IMG_FWD_FLOATS + 64, pimg + IMG_FWD_FLOATS + 128, xb, KS1, o0, o1, lane);
                int act;
                float lp;
                h64t_act(o0, o1, A, rng_seed, env_offset + (uint32_t)row, rng_step, deterministic, act, lp, lane);
                if (ok && g == 0) {
                    actions_out[row] = act;
                    logp_out[row] = lp;
                }
            }
        }
    }
    if constexpr (MODE != 1) {
        if (boot_trunc != nullptr) {
            for (int64_t tile = (int64_t)blockIdx.x * wpb + wave; tile < n_tiles; tile += (int64_t)gridDim.x * wpb) {
                const int64_t row = (tile << 4) + r16;
                const bool tflag = (row < n) && boot_trunc[row < n ? row : 0] != 0;
                if (__ballot(tflag) == 0ull) continue;
                float xb[4];
                features(boot_obs, row, row < n, xb);
                f32x4 o0, o1;
                h64t_forward<4>(vimg, vimg + IMG_FWD_FLOATS, vimg + IMG_FWD_FLOATS + 64, vimg + IMG_FWD_FLOATS + 128, xb, KS1, o0, o1, lane);
                if (tflag && g == 0) {
                    const float gv = gamma * (o0[0] + o1[0]);
                    boot_rewards[row] = boot_rewards[row] + gv;
                }
            }
        }
    }
}

template <int MODE>
static int launch_fwd_h64(const float *params, const PLayout &L, const float *obs, int64_t n, uint32_t seed, uint32_t step, uint32_t env_offset,
                          int deterministic, void *actions, float *values, float *logp, const float *boot_obs, const uint8_t *boot_trunc, float gamma,
                          float *boot_rewards, hipStream_t s) {
    const int64_t tiles = ceil_div(n, 16);
    const int wpb = tiles >= 512 ? 4 : (tiles >= 64 ? 2 : 1);
    const int smem = ((MODE == 0) ? 2 : 1) * FWD_IMG * 4;
    int64_t blocks = ceil_div(tiles, wpb);
    if (blocks > 2048) blocks = 2048;
    auto k = policy_fwd_h64_kernel<MODE>;
    if (smem > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    k<<<dim3((unsigned)blocks), dim3(64 * wpb), smem, s>>>(params, L, obs, n, seed, step, env_offset, deterministic, static_cast<int32_t *>(actions), values,
                                                           logp, boot_obs, boot_trunc, gamma, boot_rewards);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}


// ------------------------------------------------------------------------------------------
// Wide-policy forward (H = 128 / 192 / 256): the column-parallel layout of ppo_grad_wide_kernel without the backward pass.
// A block of 4 waves carries a row group of 32 samples through both nets; wave w computes hidden columns [w*H/4, (w+1)*H/4),
// every weight fragment read from L2 serves both 16-row tiles, waves 0/1 finish the heads (sampling / log-prob / value).
// One wave per tile (policy_fwd_kernel) needs ~3700 dependent MFMAs with L2 loads per tile at H = 256: ~300 us per launch.
// ------------------------------------------------------------------------------------------
template <bool CONT, int MODE, int NTW, bool BF>
__global__ __launch_bounds__(256) void policy_fwd_wide_kernel(const float *__restrict__ params, PLayout L, const float *__restrict__ obs, int64_t n,
                                                              uint32_t rng_seed, uint32_t rng_step, uint32_t env_offset, int deterministic,
                                                              void *__restrict__ actions_out, float *__restrict__ values_out,
                                                              float *__restrict__ logp_out, const uint8_t *__restrict__ trunc, float gamma,
                                                              float *__restrict__ rewards) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int M = 32, H = 64 * NTW, ld = H + 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int D = L.D, A = L.A;
    const int ldx = ((D + 3) & ~3) + 2, KS1 = (D + 3) >> 2;
    float *X = smem, *h1 = X + M * ldx, *h2 = h1 + M * ld;
    // bf16 mode (tma_wide_bf16.h): the same LDS bytes hold bf16 A images; the forward code is shared with the update kernel
    const int Kp1 = (D + 31) & ~31, ldxb = Kp1 + 16, lda = H + 16;
    bf16_t *Xa = reinterpret_cast<bf16_t *>(smem), *A1 = Xa + M * ldxb, *A2 = A1 + M * lda;
    const int n_base = wave * 16 * NTW;
    const int64_t n_groups = (n + M - 1) / M;
    // bf16 mode: the net of a block never changes (MODE 0: value net on even, policy net on odd blocks; MODES 1 / 2: value net), so EVERY
    // weight fragment and bias of this wave -- up to six layer-1 k-steps (observations of up to 192 floats) and the whole layer 2 -- is
    // requested once, at kernel start, in front of the observation loads: one L2 round trip under the observation staging instead of one
    // per layer-1 k-step behind it.  Same k order per accumulator as bf_hidden_layer, hence the same bits (the update kernel's
    // forward pass must reproduce these log-probabilities).
    constexpr int KS2B = H / 32, KS1R = 6;
    const int KS1b = Kp1 >> 5;
    const bool blk_pi = MODE == 0 && (blockIdx.x & 1) == 1;
    bf16x8 w1r[BF ? NTW : 1][BF ? KS1R : 1], w2r[BF ? NTW : 1][BF ? KS2B : 1];
    float b1r[NTW], b2r[NTW];
    if constexpr (BF) {
        const BfNetPtr Wb = bf_net_ptr(params, L, blk_pi);
        const Net Qb = blk_pi ? pi_net(params, L) : vf_net(params, L);
#pragma unroll
        for (int j = 0; j < NTW; j++) {
#pragma unroll
            for (int ks = 0; ks < KS1R; ks++)
                if (ks < KS1b && KS1b <= KS1R) w1r[j][ks] = bf_frag(Wb.fW1, (wave * NTW + j) * KS1b + ks, lane);
        }
#pragma unroll
        for (int j = 0; j < NTW; j++) {
#pragma unroll
            for (int ks = 0; ks < KS2B; ks++) w2r[j][ks] = bf_frag(Wb.fW2, (wave * NTW + j) * KS2B + ks, lane);
            b1r[j] = Qb.b1[n_base + 16 * j + r16];
            b2r[j] = Qb.b2[n_base + 16 * j + r16];
        }
    }
    bool two_rt = true;  // (block-uniform) the row group has more than 16 valid rows
    auto hidden = [&](const Net &Q, bool is_pi) {  // X -> h1 -> h2 for this wave's columns, both row tiles
        if constexpr (BF) {
            constexpr int KS2 = KS2B;
            auto &w2 = w2r;
            auto &b2v = b2r;
            if (KS1b <= KS1R) {  // (uniform) layer 1 from the register fragments: bf_hidden_layer's loop order and epilogue
                f32x4 acc1[NTW][2];
#pragma unroll
                for (int j = 0; j < NTW; j++) acc1[j][0] = acc1[j][1] = f32x4{b1r[j], b1r[j], b1r[j], b1r[j]};
#pragma unroll
                for (int ks = 0; ks < KS1R; ks++) {
                    if (ks < KS1b) {
#pragma unroll
                        for (int mt = 0; mt < 2; mt++) {
                            const bf16x8 a = a_frag(Xa, ldxb, 16 * mt + r16, ks, g);
#pragma unroll
                            for (int j = 0; j < NTW; j++) acc1[j][mt] = mfma_bf(a, w1r[j][ks], acc1[j][mt]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NTW; j++)
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) A1[(16 * mt + 4 * g + r) * lda + n_base + 16 * j + r16] = (bf16_t)tma_tanh(acc1[j][mt][r]);
            } else {
                bf_hidden_layer<NTW, 2, false>(Xa, ldxb, KS1b, bf_net_ptr(params, L, is_pi).fW1, Q.b1, A1, lda, nullptr, n_base, lane);
            }
            __syncthreads();
            f32x4 acc[NTW][2];
#pragma unroll
            for (int j = 0; j < NTW; j++) acc[j][0] = acc[j][1] = f32x4{b2v[j], b2v[j], b2v[j], b2v[j]};
#pragma unroll
            for (int ks = 0; ks < KS2; ks++)
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    const bf16x8 a = a_frag(A1, lda, 16 * mt + r16, ks, g);
#pragma unroll
                    for (int j = 0; j < NTW; j++) acc[j][mt] = mfma_bf(a, w2[j][ks], acc[j][mt]);
                }
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) A2[(16 * mt + 4 * g + r) * lda + n_base + 16 * j + r16] = (bf16_t)tma_tanh(acc[j][mt][r]);
            __syncthreads();
            return;
        }
        // f32 mode.  A row group with at most 16 valid rows (the reference's own 8-env runs) skips its second row tile: half the MFMAs of a
        // latency chain that is all there is at that size.  Weights: the whole k range of a column tile is fetched before it is used, and
        // tile j + 1's layer-2 column slice is requested before tile j multiplies (one exposed L2 latency per layer, not one per tile).
#pragma unroll 1
        for (int j = 0; j < NTW; j++) {
            const float bias = Q.b1[n_base + 16 * j + r16];
            f32x4 c0 = f32x4{bias, bias, bias, bias}, c1 = c0;
            for (int ks0 = 0; ks0 < KS1; ks0 += 16) {
                float wv1[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int k = 4 * (ks0 + u) + g;
                    wv1[u] = Q.W1t[(int64_t)((ks0 + u < KS1 && k < D) ? k : 0) * H + n_base + 16 * j + r16];
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    if (ks0 + u < KS1) {  // (uniform)
                        const int k = 4 * (ks0 + u) + g;
                        const float w = k < D ? wv1[u] : 0.0f;
                        c0 = mfma16(X[r16 * ldx + k], w, c0);
                        if (two_rt) c1 = mfma16(X[(16 + r16) * ldx + k], w, c1);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                h1[(g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(c0[r]);
                if (two_rt) h1[(16 + g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(c1[r]);
            }
        }
        __syncthreads();
        float wv[2][H / 4];
        auto fetch2 = [&](int j, float (&dst)[H / 4]) {
            const float *wcol = Q.W2t + n_base + 16 * j + r16;
#pragma unroll
            for (int ks = 0; ks < H / 4; ks++) dst[ks] = wcol[(int64_t)(4 * ks + g) * H];
        };
        fetch2(0, wv[0]);
#pragma unroll
        for (int j = 0; j < NTW; j++) {
            const float bias = Q.b2[n_base + 16 * j + r16];
            f32x4 c0 = f32x4{bias, bias, bias, bias}, c1 = c0;
            if (j + 1 < NTW) fetch2(j + 1, wv[(j + 1) & 1]);
            if (two_rt) {
#pragma unroll
                for (int ks = 0; ks < H / 4; ks++) {
                    const int k = 4 * ks + g;
                    c0 = mfma16(h1[r16 * ld + k], wv[j & 1][ks], c0);
                    c1 = mfma16(h1[(16 + r16) * ld + k], wv[j & 1][ks], c1);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < H / 4; ks++) c0 = mfma16(h1[r16 * ld + 4 * ks + g], wv[j & 1][ks], c0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                h2[(g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(c0[r]);
                if (two_rt) h2[(16 + g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(c1[r]);
            }
        }
        __syncthreads();
    };
    // MODE 0 runs the value net and the policy net of a row group on two different blocks (even / odd blockIdx): 4096 envs are only
    // 128 row groups, so this fills all 256 CUs and halves the dependent chain of a vector step.
    constexpr int NR = MODE == 0 ? 2 : 1;
    const int role = MODE == 0 ? (int)(blockIdx.x & 1) : 0;
    for (int64_t grp = blockIdx.x / NR; grp < n_groups; grp += gridDim.x / NR) {
        const int64_t row0 = grp * M;
        two_rt = row0 + 16 < n;
        if constexpr (MODE == 2) {  // skip row groups without a truncated env (block-uniform vote)
            const int64_t rr = row0 + threadIdx.x;
            const int any = __syncthreads_or((threadIdx.x < M && rr < n && trunc[rr] != 0) ? 1 : 0);
            if (!any) continue;
        }
        // observation rows -> LDS, twelve loads in flight per thread: the element-by-element form of this loop waited out one memory round
        // trip per element (24 in a row at the Crawler width, more than the rest of the kernel together)
        {
            const int width = BF ? Kp1 : ((D + 3) & ~3), tot = M * width;
            for (int e0 = threadIdx.x; e0 < tot; e0 += 12 * blockDim.x) {
                float val[12];
#pragma unroll
                for (int u = 0; u < 12; u++) {
                    const int e = e0 + u * blockDim.x, ec = e < tot ? e : 0;
                    const int row = ec / width, c = ec - row * width;
                    const bool ok = e < tot && row0 + row < n && c < D;
                    val[u] = obs[ok ? (row0 + row) * D + c : 0];  // (clamped address; masked when it is consumed below)
                }
#pragma unroll
                for (int u = 0; u < 12; u++) {
                    const int e = e0 + u * blockDim.x;
                    if (e < tot) {
                        const int row = e / width, c = e - row * width;
                        const float x = (row0 + row < n && c < D) ? val[u] : 0.0f;
                        if constexpr (BF) Xa[row * ldxb + c] = (bf16_t)x;
                        else X[row * ldx + c] = x;
                    }
                }
            }
        }
        __syncthreads();
        const Net V = vf_net(params, L);
        f32x4 vacc[1] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
        const int mt = wave & 1;
        if (role == 0) {
            hidden(V, false);
            if (wave < 2) {
                if constexpr (BF) bf_head<1>(A2, lda, 16 * mt, H / 32, bf_net_ptr(params, L, false).fW3, V.b3, 1, vacc, lane);
                else dense_head<1>(h2 + mt * 16 * ld, ld, H, V.W3t, V.b3, 1, vacc, lane);
            }
        }
        if constexpr (MODE == 1 || MODE == 0) {
            if (role == 0 && wave < 2 && r16 == 0)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t row = row0 + mt * 16 + g * 4 + r;
                    if (row < n) values_out[row] = vacc[0][r];
                }
        } else if constexpr (MODE == 2) {
            if (wave < 2 && r16 == 0)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t row = row0 + mt * 16 + g * 4 + r;
                    if (row < n && trunc[row]) {
                        const float gv = gamma * vacc[0][r];
                        rewards[row] = rewards[row] + gv;
                    }
                }
        }
        if constexpr (MODE == 0) if (role == 1) {
            const Net P = pi_net(params, L);
            hidden(P, true);
            if (wave < 2) {
                const float *hh = h2 + mt * 16 * ld;
                if constexpr (!CONT) {
                    f32x4 acc[1];
                    if constexpr (BF) bf_head<1>(A2, lda, 16 * mt, H / 32, bf_net_ptr(params, L, true).fW3, P.b3, A, acc, lane);
                    else dense_head<1>(hh, ld, H, P.W3t, P.b3, A, acc, lane);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int64_t row = row0 + mt * 16 + g * 4 + r;
                        const bool colok = r16 < A;
                        const float x = colok ? acc[0][r] : -INFINITY;
                        const float m = gmax16(x);
                        const float e = colok ? expf(x - m) : 0.0f;
                        const float sm = gsum16(e);
                        const float lse = m + logf(sm);
                        const float lp = x - lse;
                        int act;
                        if (deterministic) {
                            float mn = (colok && x == m) ? (float)r16 : 99.0f;
                            mn = gmin16(mn);
                            act = (int)mn;
                        } else {
                            const float c = gscan16(e / sm);
                            const float u = uniform01(mix32(rng_seed, env_offset + (uint32_t)row, rng_step));
                            const float cnt = gsum16((colok && c <= u) ? 1.0f : 0.0f);
                            act = min((int)cnt, A - 1);
                        }
                        const float lpa = gsum16((r16 == act) ? lp : 0.0f);
                        if (r16 == r && row < n) {
                            static_cast<int32_t *>(actions_out)[row] = act;
                            logp_out[row] = lpa;
                        }
                    }
                } else {
                    f32x4 acc[2];
                    if constexpr (BF) bf_head<2>(A2, lda, 16 * mt, H / 32, bf_net_ptr(params, L, true).fW3, P.b3, A, acc, lane);
                    else dense_head<2>(hh, ld, H, P.W3t, P.b3, A, acc, lane);
                    const float *ls = params + L.log_std;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int64_t row = row0 + mt * 16 + g * 4 + r;
                        const uint32_t gi = env_offset + (uint32_t)row;
                        float lpsum = 0.0f;
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            const int col = 16 * j + r16;
                            if (col < A) {
                                const float mu = acc[j][r], lsd = ls[col], sd = expf(lsd);
                                float a = mu;
                                if (!deterministic) {
                                    const float u1 = fmaxf(uniform01(mix32(rng_seed ^ (0x68E31DA4u + (uint32_t)col * 0x9E3779B9u), gi, rng_step)), 5.9604645e-08f);
                                    const float u2 = uniform01(mix32(rng_seed ^ (0xB5297A4Du + (uint32_t)col * 0x85EBCA77u), gi, rng_step));
                                    // Box-Muller on the hardware units: v_log (base 2, scaled), v_sqrt, and v_cos, whose argument is in revolutions
                                    const float z = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
                                    a = mu + sd * z;
                                }
                                const float d = a - mu;
                                lpsum += -(d * d) / (2.0f * (sd * sd)) - lsd - 0.9189385332046727f;
                                if (row < n) static_cast<float *>(actions_out)[row * A + col] = a;
                            }
                        }
                        lpsum = gsum16(lpsum);
                        if (r16 == r && row < n) logp_out[row] = lpsum;
                    }
                }
            }
        }
        __syncthreads();  // the next row group overwrites X / h1 / h2
    }
}

static int fwd_wide_smem_bytes(const PLayout &L) {
    const int ldx = ((L.D + 3) & ~3) + 2, ld = L.H + 2;
    return 32 * (ldx + 2 * ld) * 4;
}

template <int MODE>
static int launch_fwd(const float *params, const tma_policy_dims *d, const float *obs, int64_t n, uint32_t seed, uint32_t step, uint32_t env_offset,
                      int deterministic, void *actions, float *values, float *logp, const uint8_t *trunc, float gamma, float *rewards,
                      hipStream_t s) {
    const PLayout L = layout_of(d);
    if (L.img_pi >= 0) {
        if constexpr (MODE == 2) return launch_fwd_h64<2>(params, L, nullptr, n, 0, 0, 0, 1, nullptr, nullptr, nullptr, obs, trunc, gamma, rewards, s);
        else return launch_fwd_h64<MODE>(params, L, obs, n, seed, step, env_offset, deterministic, actions, values, logp, nullptr, nullptr, 0.0f, nullptr, s);
    }
    if (L.bf16) {
        const int smemw = fwd_wide_bf_smem_bytes(L.D, L.H);
        int64_t groups = ceil_div(n, 32);
        if (groups > 4096) groups = 4096;
        auto launchw = [&](auto k) -> int {
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smemw));
            k<<<dim3((unsigned)(MODE == 0 ? 2 * groups : groups)), dim3(256), smemw, s>>>(params, L, obs, n, seed, step, env_offset, deterministic, actions,
                                                                                           values, logp, trunc, gamma, rewards);
            return TMA_OK;
        };
        int wrc;
        if (d->continuous) wrc = L.H == 256 ? launchw(policy_fwd_wide_kernel<true, MODE, 4, true>) : (L.H == 192 ? launchw(policy_fwd_wide_kernel<true, MODE, 3, true>) : launchw(policy_fwd_wide_kernel<true, MODE, 2, true>));
        else wrc = L.H == 256 ? launchw(policy_fwd_wide_kernel<false, MODE, 4, true>) : (L.H == 192 ? launchw(policy_fwd_wide_kernel<false, MODE, 3, true>) : launchw(policy_fwd_wide_kernel<false, MODE, 2, true>));
        if (wrc) return wrc;
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    if ((L.H == 128 || L.H == 192 || L.H == 256) && fwd_wide_smem_bytes(L) <= 160 * 1024) {
        const int smemw = fwd_wide_smem_bytes(L);
        int64_t groups = ceil_div(n, 32);
        if (groups > 4096) groups = 4096;
        auto launchw = [&](auto k) -> int {
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smemw));
            k<<<dim3((unsigned)(MODE == 0 ? 2 * groups : groups)), dim3(256), smemw, s>>>(params, L, obs, n, seed, step, env_offset, deterministic, actions,
                                                                                           values, logp, trunc, gamma, rewards);
            return TMA_OK;
        };
        int wrc;
        if (d->continuous) wrc = L.H == 256 ? launchw(policy_fwd_wide_kernel<true, MODE, 4, false>) : (L.H == 192 ? launchw(policy_fwd_wide_kernel<true, MODE, 3, false>) : launchw(policy_fwd_wide_kernel<true, MODE, 2, false>));
        else wrc = L.H == 256 ? launchw(policy_fwd_wide_kernel<false, MODE, 4, false>) : (L.H == 192 ? launchw(policy_fwd_wide_kernel<false, MODE, 3, false>) : launchw(policy_fwd_wide_kernel<false, MODE, 2, false>));
        if (wrc) return wrc;
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    const int64_t tiles = ceil_div(n, 16);
    int wpb = tiles >= 1024 ? 4 : 1;  // small batches: one wave per block so every CU gets work
    while (wpb > 1 && fwd_smem_bytes(L, wpb) > 64 * 1024) wpb >>= 1;
    const int smem = fwd_smem_bytes(L, wpb);
    int64_t blocks = ceil_div(tiles, wpb);
    if (blocks > 8192) blocks = 8192;
    if (d->continuous) {
        auto k = policy_fwd_kernel<true, MODE>;
        if (smem > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        k<<<dim3((unsigned)blocks), dim3(64 * wpb), smem, s>>>(params, L, obs, n, seed, step, env_offset, deterministic, actions, values, logp,
                                                               trunc, gamma, rewards);
    } else {
        auto k = policy_fwd_kernel<false, MODE>;
        if (smem > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        k<<<dim3((unsigned)blocks), dim3(64 * wpb), smem, s>>>(params, L, obs, n, seed, step, env_offset, deterministic, actions, values, logp,
                                                               trunc, gamma, rewards);
    }
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

}  // namespace tma

using namespace tma;

int tma_launch_slab_zero_w1(float *slabs, int n_slabs, const PLayout &L, hipStream_t s) {
    slab_zero_w1_kernel<<<dim3(256), dim3(256), 0, s>>>(slabs, n_slabs, L);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

extern "C" {

int64_t tma_ppo_workspace_bytes(const tma_policy_dims *d) {
    if (!d || check_dims(d)) return WS_BYTES;
    const PLayout L = layout_of(d);
    return fold_state_offset(L) + fold_state_bytes(L);
}

int tma_policy_param_count(const tma_policy_dims *d, int64_t *n_trainable, int64_t *n_total) {
    int rc = check_dims(d);
    if (rc) return rc;
    const PLayout L = layout_of(d);
    if (n_trainable) *n_trainable = L.P;
    if (n_total) *n_total = L.total;
    return TMA_OK;
}

int tma_policy_param_offsets(const tma_policy_dims *d, int32_t *out13) {
    int rc = check_dims(d);
    if (rc) return rc;
    if (!out13) return fail(TMA_ERR_INVALID, "out13 is null");
    const PLayout L = layout_of(d);
    const int o[13] = {L.pW1t, L.pb1, L.pW2t, L.pb2, L.pW3t, L.pb3, L.vW1t, L.vb1, L.vW2t, L.vb2, L.vW3t, L.vb3, L.log_std};
    for (int i = 0; i < 13; i++) out13[i] = o[i];
    return TMA_OK;
}

int tma_policy_sync(float *params, const tma_policy_dims *d, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params) return fail(TMA_ERR_INVALID, "params is null");
    const PLayout L = layout_of(d);
    return launch_sync(params, L, (hipStream_t)stream);
}

int tma_policy_act(const float *params, const tma_policy_dims *d, const float *obs, int64_t n, uint32_t rng_seed, uint32_t rng_step,
                   uint32_t env_offset, int deterministic, void *actions_out, float *values_out, float *logp_out, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !obs || !actions_out || !values_out || !logp_out) return fail(TMA_ERR_INVALID, "tma_policy_act: null buffer");
    if (n < 1) return fail(TMA_ERR_INVALID, "tma_policy_act: n must be >= 1");
    return launch_fwd<0>(params, d, obs, n, rng_seed, rng_step, env_offset, deterministic, actions_out, values_out, logp_out, nullptr, 0.0f, nullptr,
                         (hipStream_t)stream);
}

int tma_policy_act_bootstrap(const float *params, const tma_policy_dims *d, const float *obs, int64_t n, uint32_t rng_seed, uint32_t rng_step,
                             uint32_t env_offset, void *actions_out, float *values_out, float *logp_out, const float *prev_terminal_obs,
                             const uint8_t *prev_truncated, double gamma, float *prev_rewards_inout, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !obs || !actions_out || !values_out || !logp_out) return fail(TMA_ERR_INVALID, "tma_policy_act_bootstrap: null buffer");
    if (n < 1) return fail(TMA_ERR_INVALID, "tma_policy_act_bootstrap: n must be >= 1");
    const PLayout L = layout_of(d);
    const bool boot = prev_terminal_obs && prev_truncated && prev_rewards_inout;
    if (L.img_pi >= 0)  // one launch: bootstrap of the previous step folded into this step's forward
        return launch_fwd_h64<0>(params, L, obs, n, rng_seed, rng_step, env_offset, 0, actions_out, values_out, logp_out, boot ? prev_terminal_obs : nullptr,
                                 boot ? prev_truncated : nullptr, (float)gamma, boot ? prev_rewards_inout : nullptr, (hipStream_t)stream);
    if (boot) {
        rc = launch_fwd<2>(params, d, prev_terminal_obs, n, 0, 0, 0, 1, nullptr, nullptr, nullptr, prev_truncated, (float)gamma, prev_rewards_inout,
                           (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch_fwd<0>(params, d, obs, n, rng_seed, rng_step, env_offset, 0, actions_out, values_out, logp_out, nullptr, 0.0f, nullptr, (hipStream_t)stream);
}

int tma_policy_values(const float *params, const tma_policy_dims *d, const float *obs, int64_t n, float *values_out, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !obs || !values_out) return fail(TMA_ERR_INVALID, "tma_policy_values: null buffer");
    if (n < 1) return fail(TMA_ERR_INVALID, "tma_policy_values: n must be >= 1");
    return launch_fwd<1>(params, d, obs, n, 0, 0, 0, 1, nullptr, values_out, nullptr, nullptr, 0.0f, nullptr, (hipStream_t)stream);
}

int tma_policy_bootstrap(const float *params, const tma_policy_dims *d, const float *terminal_obs, const uint8_t *truncated, int64_t n, double gamma,
                         float *rewards_inout, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !terminal_obs || !truncated || !rewards_inout) return fail(TMA_ERR_INVALID, "tma_policy_bootstrap: null buffer");
    if (n < 1) return fail(TMA_ERR_INVALID, "tma_policy_bootstrap: n must be >= 1");
    return launch_fwd<2>(params, d, terminal_obs, n, 0, 0, 0, 1, nullptr, nullptr, nullptr, truncated, (float)gamma, rewards_inout, (hipStream_t)stream);
}

// fold (H = 64 fast path only): the previous minibatch's optimizer step, done in the prologue of this gradient launch (AdamFold)
static int minibatch_grad_impl(const float *params, const tma_policy_dims *d, const tma_rollout *rb, const tma_minibatch *mbi, const tma_ppo_hparams *hp,
                               float *grad, void *workspace, void *stream, const AdamFold *fold, int overwrite, const PeerPush *push = nullptr) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !rb || !mbi || !hp || !grad || !workspace) return fail(TMA_ERR_INVALID, "tma_ppo_minibatch_grad: null argument");
    if (!rb->obs || !rb->actions || !rb->log_probs || !rb->advantages || !rb->returns) return fail(TMA_ERR_INVALID, "rollout view has a null buffer");
    if (rb->T < 1 || rb->N < 1) return fail(TMA_ERR_INVALID, "rollout view: T and N must be >= 1");
    const int64_t total = (int64_t)rb->T * rb->N;
    if (total > 0x7fffffffLL) return fail(TMA_ERR_INVALID, "rollout of %lld samples exceeds the 2^31 minibatch index range", (long long)total);
    if (mbi->count < 1 || mbi->start < 0 || mbi->start + mbi->count > total)
        return fail(TMA_ERR_INVALID, "minibatch [%lld, +%lld) outside the %lld-sample rollout", (long long)mbi->start, (long long)mbi->count, (long long)total);
    hipStream_t s = (hipStream_t)stream;
    const PLayout L = layout_of(d);
    Rollout R{rb->obs, rb->actions, rb->log_probs, rb->advantages, rb->returns, rb->T, rb->N, rec_floats(L) > 0 ? rb->packed : nullptr};
    Minibatch M{mbi->indices, mbi->perm_seed, mbi->perm_epoch, mbi->start, mbi->count, total, nullptr, mbi->count, nullptr, 0};
    const bool prepared = mbi->prepared_batch > 0;
    if (mbi->stats_count != 0) {
        if (!prepared || mbi->stats_count < mbi->count)
            return fail(TMA_ERR_INVALID, "stats_count %lld needs a prepared epoch (tma_ppo_epoch_prepare + tma_ppo_epoch_adv_sums) and must be >= count %lld",
                        (long long)mbi->stats_count, (long long)mbi->count);
        M.stats_n = mbi->stats_count;
    }
    if (prepared && (mbi->prepared_batch < 256 || total > OFFS_CAP || mbi->start % mbi->prepared_batch != 0 || mbi->count > mbi->prepared_batch))
        return fail(TMA_ERR_INVALID, "minibatch [%lld, +%lld) does not match the prepared epoch split (batch %lld)", (long long)mbi->start,
                    (long long)mbi->count, (long long)mbi->prepared_batch);
    static const int bf_debug = getenv("TMA_BF_DEBUG") ? atoi(getenv("TMA_BF_DEBUG")) : 0;
    HParams hpar{(float)hp->clip_range, (float)hp->ent_coef, (float)hp->vf_coef, (hp->normalize_advantage && mbi->count > 1) ? 1 : 0, bf_debug};
    char *ws = static_cast<char *>(workspace);
    float *ws_adv = reinterpret_cast<float *>(ws + WS_ADV);
    double *slots = reinterpret_cast<double *>(ws + WS_STATS);
    const int64_t tiles = ceil_div(mbi->count, 16);
    const bool h64 = L.img_pi >= 0 && tiles >= 16;  // >= 256 samples: persistent LDS-image kernel; smaller batches: generic kernel
    double *adv_part = reinterpret_cast<double *>(ws + WS_ADV_PART);
    int nbk = (int)ceil_div(mbi->count, 1024);
    if (nbk > ADV_BLOCKS) nbk = ADV_BLOCKS;
    const int64_t offs_base = WS_SLABS + (int64_t)slab_cap(L) * L.P * 4;
    if (prepared) {  // tma_ppo_epoch_prepare left this minibatch's partials and the epoch's offsets in the workspace
        int stride = (int)ceil_div(mbi->prepared_batch, 1024);
        if (stride > ADV_BLOCKS) stride = ADV_BLOCKS;
        adv_part = reinterpret_cast<double *>(ws + offs_base + OFFS_CAP * 4) + 2 * (mbi->start / mbi->prepared_batch) * stride;
        M.offs = reinterpret_cast<int32_t *>(ws + offs_base) + mbi->start;
    }
    // offsets cache: written by the advantage pass, read by every gradient kernel (saves the permutation arithmetic per sample)
    int32_t *offs = (!prepared && mbi->count <= OFFS_CAP && (hpar.normalize_advantage || L.bf16)) ? reinterpret_cast<int32_t *>(ws + offs_base) : nullptr;
    if (!prepared && (hpar.normalize_advantage || offs)) {
        adv_partial_kernel<<<dim3(nbk), dim3(256), 0, s>>>(rb->advantages, M, rb->T, rb->N, adv_part, offs);
        TMA_LAUNCH_CHECK();
        M.offs = offs;
    }
    M.adv_part = adv_part, M.adv_n_part = nbk;
    static const bool force_wide = getenv("TMA_FORCE_WIDE") != nullptr;  // test hook: take the column-parallel kernel at any batch size
    const bool wide_f32 = !L.bf16 && (L.H == 128 || L.H == 192 || L.H == 256) && (tiles >= 8 || force_wide) && grad_wide_smem_bytes(L) <= 160 * 1024;
    if (hpar.normalize_advantage) {
        if (!h64 && !L.bf16 && !wide_f32) {  // the H = 64 and the column-parallel kernels fold the partials themselves
            adv_final_kernel<<<dim3(1), dim3(64), 0, s>>>(adv_part, nbk, M.stats_n, ws_adv);
            TMA_LAUNCH_CHECK();
        }
    }
    if (h64) {
        // register-accumulating persistent kernel (tma_h64.hip) + deterministic slab reduction
        float *slabs = reinterpret_cast<float *>(ws + WS_SLABS);
        int blocks4 = 0;
        int lrc;
        {
            GradTimer timer(s);
            lrc = tma_launch_grad_h64(params, L, R, M, hpar, adv_part, nbk, slabs, slots, &blocks4, s, fold);
        }
        if (lrc) return lrc;
        slab_reduce_kernel<<<dim3((unsigned)ceil_div(L.P, 64)), dim3(256), 0, s>>>(slabs, (int)blocks4, L.P, grad, -1, 0, 0, sq_partials(ws, L), overwrite,
                                                                                   push ? *push : PeerPush{});
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    if (fold || overwrite || push) return fail(TMA_ERR_INVALID, "internal: folded optimizer step outside the H = 64 fast path");
    if (L.bf16) {  // column-parallel bf16-MFMA kernel (tma_bf16.hip) + deterministic slab reduction
        if ((int64_t)rb->T * rb->N * L.D >= (int64_t)1 << 31)  // (its observation gather indexes the buffer with 32-bit arithmetic)
            return fail(TMA_ERR_INVALID, "bf16 update: T * N * obs_dim = %lld exceeds 2^31", (long long)((int64_t)rb->T * rb->N * L.D));
        float *slabs = reinterpret_cast<float *>(ws + WS_SLABS);
        int n_pi = 0, n_vf = 0, lrc;
        {
            GradTimer timer(s);
            lrc = tma_launch_grad_wide_bf(params, L, R, M, hpar, ws_adv, slabs, slots, ws, &n_pi, &n_vf, s);
        }
        if (lrc) return lrc;
        slab_reduce_kernel<<<dim3((unsigned)ceil_div(L.P, 64)), dim3(256), 0, s>>>(slabs, n_pi, L.P, grad, n_vf, L.vW1t, L.log_std, sq_partials(ws, L));
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    if (wide_f32 && tma_split3_eligible(L, mbi->count)) {  // mfma_dtype = 2: the same update on the bf16 MFMA, every operand as three bf16 terms
        float *slabs = reinterpret_cast<float *>(ws + WS_SLABS);
        int n_pi = 0, n_vf = 0, lrc;
        {
            GradTimer timer(s);
            lrc = tma_launch_grad_split3(params, L, R, M, hpar, slabs, slots, &n_pi, &n_vf, s);
        }
        if (lrc) return lrc;
        slab_reduce_kernel<<<dim3((unsigned)ceil_div(L.P, 64)), dim3(256), 0, s>>>(slabs, n_pi, L.P, grad, n_vf, L.vW1t, L.log_std, sq_partials(ws, L));
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    if (wide_f32) {
        // column-parallel register-accumulating kernel + deterministic slab reduction
        const int smemw = grad_wide_smem_bytes(L);
        // small minibatches on single-pass shapes (D <= 32): 16-row half groups, so that the reference's literal batch_size = 256 runs on 32
        // workgroups instead of 16 (TMA_NO_HALF_GROUPS=1: 32-row groups throughout)
        // Round 6: ... and observations of up to 112 floats (the reference's `ant` task: Ant-v5, 105 observations, batch_size 256) on the same
        // half groups with dW1 in registers (7 k-tiles: the eight-wave half-group kernel defers dW2, so it has them) -- that width took the
        // runtime-width kernel with dW1 accumulated in the slab: 99.5 us per 256-sample gradient launch
        static const bool no_half = getenv("TMA_NO_HALF_GROUPS") != nullptr;
        const bool small7 = L.H == 256 && L.D > 32 && L.D <= 112 && mbi->count <= 1024 && !no_half && getenv("TMA_WIDE_NW4") == nullptr && getenv("TMA_NO_DEFER_W2") == nullptr &&
                            (int64_t)64 * L.P + 4 * (int64_t)W2_DEFER_ROWS * L.H <= (int64_t)slab_cap(L) * L.P;  // (the deferral buffer must fit: that kernel has no dW2 accumulators)
        const bool half = (L.D <= 32 || small7) && mbi->count <= 1024 && !no_half;  // (at 2048 samples the doubled slab count costs more than the shorter groups save: 79.6 against 76.6 us per call)
        const int64_t groups = ceil_div(mbi->count, half ? 16 : 32);  // one row group per block while there are CUs to spare, then grid-stride
        const int cap_pi = d->continuous ? 136 : 128, cap_vf = 256 - cap_pi;  // measured: the Categorical head leaves the two nets balanced
        const int n_pi = (int)(groups < cap_pi ? groups : cap_pi), n_vf = (int)(groups < cap_vf ? groups : cap_vf);
        const int64_t pairs = n_pi;  // slabs in use (the value net uses the first n_vf of them)
        float *slabs = reinterpret_cast<float *>(ws + WS_SLABS);
        // dW1: D <= 32 in registers; D in 161..176 (Crawler's 172: 11 k-tiles) by a second pass that keeps only dW1 in registers;
        // any other width accumulates it in place in the slab
        // (round 6: 97..112 observations with a Box head at H = 256 -- Ant-v5's 105 -- by the same two passes with seven k-tiles: kt1 = 107, "7 in two passes")
        const int kt1 = L.D <= 16 ? 1 : (L.D <= 32 ? 2 : (small7 ? 7 : ((L.D > 160 && L.D <= 176) ? 11 : ((f32_two_pass(L) && L.D <= 112) ? 107 : 0))));
        if (kt1 == 0) {
            slab_zero_w1_kernel<<<dim3(256), dim3(256), 0, s>>>(slabs, (int)pairs, L);
            TMA_LAUNCH_CHECK();
        }
        auto launch = [&](auto k, float *dz1 = nullptr) -> int {
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smemw));
            k<<<dim3((unsigned)(n_pi + n_vf)), dim3(256), smemw, s>>>(params, L, R, M, hpar, ws_adv, slabs, slots, n_pi, dz1, DZ1_CAP * L.H);
            return TMA_OK;
        };
        const bool eight = L.H == 256 && (kt1 == 1 || kt1 == 2 || kt1 == 7) && getenv("TMA_WIDE_NW4") == nullptr;
        const int smem8 = grad_wide_smem_bytes(L, 8);
        // half groups on the eight-wave kernel (<= 1024 samples: <= 64 slabs in use): dW2 deferred to wide_small_reduce_kernel through a buffer behind
        // slab 64 of the workspace's slab area (TMA_NO_DEFER_W2=1: the slab path throughout)
        static const bool no_defer = getenv("TMA_NO_DEFER_W2") != nullptr;
        const bool defer_w2 = half && eight && !no_defer && n_pi <= 64 && groups * 16 <= W2_DEFER_ROWS && (int64_t)64 * L.P + 4 * (int64_t)W2_DEFER_ROWS * L.H <= (int64_t)slab_cap(L) * L.P;
        float *const w2buf = defer_w2 ? slabs + (int64_t)64 * L.P : nullptr;
        auto launch8 = [&](auto k) -> int {
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem8));
            k<<<dim3((unsigned)(n_pi + n_vf)), dim3(512), smem8, s>>>(params, L, R, M, hpar, ws_adv, slabs, slots, n_pi, w2buf, (int64_t)2 * W2_DEFER_ROWS * L.H);
            return TMA_OK;
        };
        // (as on the bf16 path) minibatches that fit the dz1 cache: chain pass + dW1 from the cached operands; else chain + recompute
        float *const dz1_cache = (f32_two_pass(L) && (kt1 == 11 || kt1 == 107) && mbi->count <= DZ1_CAP && !getenv("TMA_NO_DZ1_CACHE"))
            ? reinterpret_cast<float *>(ws + WS_SLABS + (int64_t)slab_cap(L) * L.P * 4 + OFFS_CAP * 4 + EPOCH_PART_BYTES + WIDE_SQ_SLOTS * 8) : nullptr;
        auto pick = [&](auto ntw) -> int {
            constexpr int NTWc = decltype(ntw)::value;
            auto both = [&](auto cont) -> int {
                constexpr bool C = decltype(cont)::value;
                if constexpr (NTWc == 4) {  // H = 256, single-pass shapes: eight waves of 32 columns (TMA_WIDE_NW4=1: four of 64)
                    if (eight && kt1 == 1) return half ? launch8(ppo_grad_wide_kernel<C, 2, 1, 0, 0, true, 8>) : launch8(ppo_grad_wide_kernel<C, 2, 1, 0, 0, false, 8>);
                    if (eight && kt1 == 2) return half ? launch8(ppo_grad_wide_kernel<C, 2, 2, 0, 0, true, 8>) : launch8(ppo_grad_wide_kernel<C, 2, 2, 0, 0, false, 8>);
                    if (eight && kt1 == 7) return launch8(ppo_grad_wide_kernel<C, 2, 7, 0, 0, true, 8>);  // (small7 implies half groups)
                }
                if (kt1 == 1) return half ? launch(ppo_grad_wide_kernel<C, NTWc, 1, 0, 0, true>) : launch(ppo_grad_wide_kernel<C, NTWc, 1>);
                if (kt1 == 2) return half ? launch(ppo_grad_wide_kernel<C, NTWc, 2, 0, 0, true>) : launch(ppo_grad_wide_kernel<C, NTWc, 2>);
                if (kt1 == 11) {
                    const int rc2 = launch(ppo_grad_wide_kernel<C, NTWc, -1, 0, 11>, dz1_cache);
                    if (rc2) return rc2;
                    return dz1_cache ? launch(ppo_grad_wide_kernel<C, NTWc, 11, 2, 11>, dz1_cache) : launch(ppo_grad_wide_kernel<C, NTWc, 11, 1, 11>);
                }
                if constexpr (C && NTWc == 4) {
                    if (kt1 == 107) {
                        const int rc2 = launch(ppo_grad_wide_kernel<true, 4, -1, 0, 7>, dz1_cache);
                        if (rc2) return rc2;
                        return dz1_cache ? launch(ppo_grad_wide_kernel<true, 4, 7, 2, 7>, dz1_cache) : launch(ppo_grad_wide_kernel<true, 4, 7, 1, 7>);
                    }
                }
                return launch(ppo_grad_wide_kernel<C, NTWc, 0>);
            };
            return d->continuous ? both(std::true_type{}) : both(std::false_type{});
        };
        int lrc;
        {
            GradTimer timer(s);
            lrc = L.H == 256 ? pick(std::integral_constant<int, 4>{}) : (L.H == 192 ? pick(std::integral_constant<int, 3>{}) : pick(std::integral_constant<int, 2>{}));
        }
        if (lrc) return lrc;
        TMA_LAUNCH_CHECK();
        if (defer_w2) {
            wide_small_reduce_kernel<<<dim3((unsigned)(128 + ceil_div(L.P - 2 * L.H * L.H, 64))), dim3(1024), 0, s>>>(slabs, n_pi, n_vf, L, w2buf, (int)(groups * 16), grad,
                                                                                                                  sq_partials(ws, L));
            TMA_LAUNCH_CHECK();
            return TMA_OK;
        }
        slab_reduce_kernel<<<dim3((unsigned)ceil_div(L.P, 64)), dim3(256), 0, s>>>(slabs, n_pi, L.P, grad, n_vf, L.vW1t, L.log_std, sq_partials(ws, L));
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    int wpb = tiles >= 512 ? 4 : 1;
    while (wpb > 1 && grad_smem_bytes(L, wpb) > 156 * 1024) wpb--;
    const int smem = grad_smem_bytes(L, wpb);
    if (smem > 160 * 1024) return fail(TMA_ERR_INVALID, "policy too wide for the LDS-resident tile (needs %d bytes)", smem);
    int64_t blocks = ceil_div(tiles, wpb);
    if (blocks > MAX_GRAD_BLOCKS / 2) blocks = MAX_GRAD_BLOCKS / 2;
    if (d->continuous) {
        auto k = ppo_grad_kernel<true>;
        if (smem > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        k<<<dim3((unsigned)(2 * blocks)), dim3(64 * wpb), smem, s>>>(params, L, R, M, hpar, ws_adv, grad, slots);
    } else {
        auto k = ppo_grad_kernel<false>;
        if (smem > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        k<<<dim3((unsigned)(2 * blocks)), dim3(64 * wpb), smem, s>>>(params, L, R, M, hpar, ws_adv, grad, slots);
    }
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

// count a persistent-epoch fallback in the workspace (WS_PERSIST_ERR + 8: int64) and say so once per process
__global__ void persist_count_kernel(long long *ctr) { *ctr += 1; }
static int persist_fallback_note(char *ws, hipStream_t s) {
    persist_count_kernel<<<dim3(1), dim3(1), 0, s>>>(reinterpret_cast<long long *>(ws + WS_PERSIST_ERR + 8));
    static bool said = false;
    if (!said) {
        said = true;
        fprintf(stderr, "libtma_hip: the persistent epoch kernel (csrc/tma_h64p.hip) could not place / synchronise its workgroups on one XCD; "
                        "this epoch (and any later one that fails the same way) runs through the per-minibatch launches instead -- same results, "
                        "lower optimizer-step rate.  TMA_NO_PERSIST=1 selects that path outright.\n");
    }
    return TMA_OK;
}

int tma_ppo_persist_fallbacks(void *workspace, int64_t *count_out, void *stream) {
    if (!workspace || !count_out) return fail(TMA_ERR_INVALID, "tma_ppo_persist_fallbacks: null argument");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, workspace) == hipSuccess) (void)hipSetDevice(attr.device);
    long long v = 0;
    TMA_HIP(hipMemcpyAsync(&v, static_cast<char *>(workspace) + WS_PERSIST_ERR + 8, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    TMA_HIP(hipStreamSynchronize((hipStream_t)stream));
    *count_out = (int64_t)v;
    return TMA_OK;
}

int tma_ppo_minibatch_grad(const float *params, const tma_policy_dims *d, const tma_rollout *rb, const tma_minibatch *mbi, const tma_ppo_hparams *hp,
                           float *grad, void *workspace, void *stream) {
    return minibatch_grad_impl(params, d, rb, mbi, hp, grad, workspace, stream, nullptr, 0);
}

// The persistent epoch kernels (tma_h64p.hip: H = 64; tma_h256p.hip: the reference's default 256 x 256 policy) over `n_epochs` consecutive
// epochs of the reference's literal batch_size = 256 in ONE launch: the sample offsets and advantage partials of every epoch are laid out
// behind each other (epoch e: permutation perm_epoch0 + e, offsets at [e * total, (e + 1) * total), partials of its minibatches behind
// those of epoch e - 1) and the kernel walks n_epochs * n_mb optimizer steps -- weights, moments and step count never leave the chip
// between epochs.  *ran = 1: everything is committed (parameters, moments, derived images, statistics).  *ran = 0: the shape is not
// eligible, or the launch could not place / synchronise its workgroups: the state is what it was before the call (snapshot restored,
// the event counted) and the caller runs the epochs through the per-minibatch launches.
static int persistent_epochs(float *params, const tma_policy_dims *d, const tma_rollout *rb, const PLayout &L, uint32_t perm_seed, uint32_t perm_epoch0,
                             int n_epochs, int64_t batch_size, const tma_ppo_hparams *hp, float *exp_avg, float *exp_avg_sq, int64_t first_step, double lr,
                             double beta1, double beta2, double eps, double max_grad_norm, void *workspace, void *stream, int *ran) {
    *ran = 0;
    const int64_t total = (int64_t)rb->T * rb->N, all = total * n_epochs;
    if (n_epochs < 1 || all > OFFS_CAP || batch_size < 256 || total % batch_size != 0) return TMA_OK;
    const bool p64 = tma_epoch_h64p_eligible(L, batch_size, all);
    const bool p256 = !p64 && h256p_layout(L) && tma_epoch_h256p_eligible(L, batch_size, all);
    if (!p64 && !p256) return TMA_OK;
    if (!rb->obs || !rb->actions || !rb->log_probs || !rb->advantages || !rb->returns) return fail(TMA_ERR_INVALID, "rollout view has a null buffer");
    char *ws = static_cast<char *>(workspace);
    hipStream_t ps = (hipStream_t)stream;
    const int64_t offs_base = WS_SLABS + (int64_t)slab_cap(L) * L.P * 4;
    int stride = (int)ceil_div(batch_size, 1024);
    if (stride > ADV_BLOCKS) stride = ADV_BLOCKS;
    const int64_t n_mb = total / batch_size;
    for (int e = 0; e < n_epochs; e++) {  // (tma_ppo_epoch_prepare's launch, epoch e at its place)
        Minibatch M{nullptr, perm_seed, perm_epoch0 + (uint32_t)e, 0, total, total, nullptr, total, nullptr, 0};
        adv_partial_kernel<<<dim3(stride, (unsigned)n_mb), dim3(256), 0, ps>>>(rb->advantages, M, rb->T, rb->N,
                                                                               reinterpret_cast<double *>(ws + offs_base + OFFS_CAP * 4) + 2 * e * n_mb * stride,
                                                                               reinterpret_cast<int32_t *>(ws + offs_base) + e * total, batch_size);
        TMA_LAUNCH_CHECK();
    }
    const Rollout R{rb->obs, rb->actions, rb->log_probs, rb->advantages, rb->returns, rb->T, rb->N, rb->packed};
    const HParams hpar{(float)hp->clip_range, (float)hp->ent_coef, (float)hp->vf_coef, hp->normalize_advantage ? 1 : 0, 0};
    // Snapshot of what the launch may commit (trainable parameters, both moments, its 64 statistic slots) in the idle half of the AdamFold
    // double buffer: "commit nothing after an abort" is a per-block decision inside the kernel, so a block that gives up on its LAST wait
    // can raise the abort word after another block has already written its net -- the fallback restores the snapshot first and is
    // therefore the same results whatever the kernel managed to write (4 small device copies per launch).
    const int64_t Pp = ((int64_t)L.P + 3) & ~(int64_t)3;
    float *snap = reinterpret_cast<float *>(ws + fold_state_offset(L));
    TMA_HIP(hipMemcpyAsync(snap, params, (size_t)L.P * 4, hipMemcpyDeviceToDevice, ps));
    TMA_HIP(hipMemcpyAsync(snap + Pp, exp_avg, (size_t)L.P * 4, hipMemcpyDeviceToDevice, ps));
    TMA_HIP(hipMemcpyAsync(snap + 2 * Pp, exp_avg_sq, (size_t)L.P * 4, hipMemcpyDeviceToDevice, ps));
    TMA_HIP(hipMemcpyAsync(ws + WS_PERSIST_SNAP, ws + WS_STATS, 8 * 8 * 8, hipMemcpyDeviceToDevice, ps));
    int rc = (p64 ? tma_launch_epoch_h64p : tma_launch_epoch_h256p)(params, L, R, hpar, reinterpret_cast<const int32_t *>(ws + offs_base),
                                                                    reinterpret_cast<const double *>(ws + offs_base + OFFS_CAP * 4), stride, all, batch_size,
                                                                    exp_avg, exp_avg_sq, first_step, lr, beta1, beta2, eps, max_grad_norm, ws, ps);
    if (rc) return rc;
    // The persistent kernels need their workgroups resident together on one XCD per group (H = 64: eight on one; 256-wide: 32 on each of
    // two); a concurrent kernel, a CU mask or a preempted wave can deny that, in which case they give up on a bounded wait and commit
    // NOTHING.  Check per launch (one 4-byte read-back: the launch is ~10^5 times longer) and, on failure, hand the epochs back to the
    // per-minibatch launches -- training goes on, the event is counted (tma_ppo_persist_fallbacks) and reported once on stderr.
    int persist_err = 0;
    TMA_HIP(hipMemcpyAsync(&persist_err, ws + WS_PERSIST_ERR, sizeof(int), hipMemcpyDeviceToHost, ps));
    TMA_HIP(hipStreamSynchronize(ps));
    if (const char *ff = getenv("TMA_PERSIST_FORCE_FAIL"))  // test hook "late": the launch ran and committed EVERYTHING, then is declared failed
        if (!strcmp(ff, "late")) persist_err = 1;
    if (!persist_err) {
        *ran = 1;
        return p256 ? launch_sync(params, L, ps) : TMA_OK;  // (the 256-wide kernel writes the trainable region; its derived images follow here)
    }
    TMA_HIP(hipMemsetAsync(ws + WS_PERSIST_ERR, 0, sizeof(int), ps));
    TMA_HIP(hipMemcpyAsync(params, snap, (size_t)L.P * 4, hipMemcpyDeviceToDevice, ps));
    TMA_HIP(hipMemcpyAsync(exp_avg, snap + Pp, (size_t)L.P * 4, hipMemcpyDeviceToDevice, ps));
    TMA_HIP(hipMemcpyAsync(exp_avg_sq, snap + 2 * Pp, (size_t)L.P * 4, hipMemcpyDeviceToDevice, ps));
    TMA_HIP(hipMemcpyAsync(ws + WS_STATS, ws + WS_PERSIST_SNAP, 8 * 8 * 8, hipMemcpyDeviceToDevice, ps));
    rc = launch_sync(params, L, ps);  // derived copies and weight images of the restored parameters
    if (rc) return rc;
    if (n_epochs == 1) persist_fallback_note(ws, ps);  // (a multi-epoch launch that failed is retried epoch by epoch: each of those counts for itself)
    return TMA_OK;
}

int tma_ppo_train_epochs_local(float *params, const tma_policy_dims *d, const tma_rollout *rb, uint32_t perm_seed, uint32_t perm_epoch0, int n_epochs,
                               int64_t batch_size, const tma_ppo_hparams *hp, float *grad, float *exp_avg, float *exp_avg_sq, int64_t first_step, double lr,
                               double beta1, double beta2, double eps, double max_grad_norm, void *workspace, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !rb || !hp || !grad || !exp_avg || !exp_avg_sq || !workspace) return fail(TMA_ERR_INVALID, "tma_ppo_train_epochs_local: null argument");
    if (rb->T < 1 || rb->N < 1 || batch_size < 1 || first_step < 1 || n_epochs < 1)
        return fail(TMA_ERR_INVALID, "tma_ppo_train_epochs_local: T, N, batch_size, first_step and n_epochs must be >= 1");
    const int64_t total = (int64_t)rb->T * rb->N, n_mb = ceil_div(total, batch_size);
    const PLayout L = layout_of(d);
    int e = 0;
    // as many epochs per persistent launch as the offsets cache holds (all of them for the reference's own 1- and 8-env schedules: 4 or 32
    // optimizer steps an epoch, where a launch per epoch was mostly launch)
    const int per = (int)(total > 0 && OFFS_CAP / total >= 1 ? (OFFS_CAP / total < n_epochs ? OFFS_CAP / total : n_epochs) : 0);
    while (per >= 1 && e < n_epochs) {
        const int n = n_epochs - e < per ? n_epochs - e : per;
        int ran = 0;
        rc = persistent_epochs(params, d, rb, L, perm_seed, perm_epoch0 + (uint32_t)e, n, batch_size, hp, exp_avg, exp_avg_sq, first_step + e * n_mb, lr, beta1, beta2,
                               eps, max_grad_norm, workspace, stream, &ran);
        if (rc) return rc;
        if (!ran) break;
        e += n;
    }
    for (; e < n_epochs; e++) {  // not eligible (or handed back): epoch by epoch (which tries the single-epoch persistent launch first, then the launches)
        rc = tma_ppo_train_epoch_local(params, d, rb, perm_seed, perm_epoch0 + (uint32_t)e, batch_size, hp, grad, exp_avg, exp_avg_sq, first_step + e * n_mb, lr,
                                       beta1, beta2, eps, max_grad_norm, workspace, stream);
        if (rc) return rc;
    }
    return TMA_OK;
}

int tma_ppo_train_epoch_local(float *params, const tma_policy_dims *d, const tma_rollout *rb, uint32_t perm_seed, uint32_t perm_epoch, int64_t batch_size,
                              const tma_ppo_hparams *hp, float *grad, float *exp_avg, float *exp_avg_sq, int64_t first_step, double lr, double beta1,
                              double beta2, double eps, double max_grad_norm, void *workspace, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !rb || !hp || !grad || !exp_avg || !exp_avg_sq || !workspace) return fail(TMA_ERR_INVALID, "tma_ppo_train_epoch_local: null argument");
    if (rb->T < 1 || rb->N < 1 || batch_size < 1 || first_step < 1) return fail(TMA_ERR_INVALID, "tma_ppo_train_epoch_local: T, N, batch_size and first_step must be >= 1");
    const int64_t total = (int64_t)rb->T * rb->N;
    const bool prepared = total <= OFFS_CAP && batch_size >= 256;
    const PLayout L = layout_of(d);
    if (prepared) {
        // the reference's literal batch_size = 256: the whole epoch as one persistent launch (H = 64: tma_h64p.hip; the reference's default
        // 256 x 256 policy: tma_h256p.hip)
        int ran = 0;
        rc = persistent_epochs(params, d, rb, L, perm_seed, perm_epoch, 1, batch_size, hp, exp_avg, exp_avg_sq, first_step, lr, beta1, beta2, eps, max_grad_norm,
                               workspace, stream, &ran);
        if (rc) return rc;
        if (ran) return TMA_OK;
        const tma_minibatch ep{nullptr, perm_seed, perm_epoch, 0, total, 0, 0};
        rc = tma_ppo_epoch_prepare(rb, &ep, batch_size, d, workspace, stream);
        if (rc) return rc;
    }
    int64_t step = first_step;
    const bool no_fold = getenv("TMA_NO_ADAM_FOLD") != nullptr;  // test / measurement switch: one optimizer launch per minibatch (read per epoch)
    const int64_t tail = total % batch_size;
    if (L.img_pi >= 0 && prepared && (tail == 0 || tail >= 256) && total > batch_size && !no_fold) {
        // H = 64 fast path, every minibatch on the LDS-image kernel: the optimizer step of minibatch k runs in the prologue of gradient
        // launch k + 1 (AdamFold: every workgroup redoes it for its net and builds its weight image from the results), so a minibatch
        // costs two launches (gradient, slab reduction) instead of three; the state ping-pongs between (params, exp_avg, exp_avg_sq) and
        // the workspace copy, and the epoch's last step is the ordinary optimizer launch, which leaves everything (derived copies and
        // images included) in the caller's buffers.  Same arithmetic on the same inputs as the unfolded sequence: bit-identical.
        char *ws = static_cast<char *>(workspace);
        hipStream_t s = (hipStream_t)stream;
        const int64_t Pp = ((int64_t)L.P + 3) & ~(int64_t)3;
        float *alt = reinterpret_cast<float *>(ws + fold_state_offset(L));
        float *bufs[2][3] = {{params, exp_avg, exp_avg_sq}, {alt, alt + Pp, alt + 2 * Pp}};
        int cur = 0;
        const double *sqp = sq_partials(ws, L);
        for (int64_t start = 0; start < total; start += batch_size, step++) {
            const int64_t count = start + batch_size <= total ? batch_size : total - start;
            const tma_minibatch mb{nullptr, perm_seed, perm_epoch, start, count, batch_size, 0};
            AdamFold f{};
            if (start > 0) {  // the step of the previous minibatch (index step - 1)
                const double bc1 = 1.0 - pow(beta1, (double)(step - 1)), bc2 = 1.0 - pow(beta2, (double)(step - 1));
                f = AdamFold{grad, sqp, (int)ceil_div(L.P, 64), bufs[cur][0], bufs[cur][1], bufs[cur][2], bufs[cur ^ 1][0], bufs[cur ^ 1][1],
                             bufs[cur ^ 1][2], (float)max_grad_norm, (float)(lr / bc1), (float)beta1, (float)beta2, (float)sqrt(bc2), (float)eps,
                             reinterpret_cast<double *>(ws + WS_NORM_OUT), 1.0f};
            }
            rc = minibatch_grad_impl(params, d, rb, &mb, hp, grad, workspace, stream, start > 0 ? &f : nullptr, start > 0 ? 1 : 0);
            if (rc) return rc;
            if (start > 0) cur ^= 1;
        }
        const double bc1 = 1.0 - pow(beta1, (double)(step - 1)), bc2 = 1.0 - pow(beta2, (double)(step - 1));
        adam_scatter_h64_kernel<<<dim3((unsigned)ceil_div(L.P, 256)), dim3(256), 0, s>>>(
            params, grad, exp_avg, exp_avg_sq, L, sqp, (int)ceil_div(L.P, 64), (float)max_grad_norm, (float)(lr / bc1), (float)beta1, (float)beta2,
            (float)sqrt(bc2), (float)eps, reinterpret_cast<double *>(ws + WS_NORM_OUT), 1.0f, bufs[cur][0], bufs[cur][1], bufs[cur][2]);
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    for (int64_t start = 0; start < total; start += batch_size, step++) {
        const int64_t count = start + batch_size <= total ? batch_size : total - start;
        const tma_minibatch mb{nullptr, perm_seed, perm_epoch, start, count, prepared ? batch_size : 0, 0};
        rc = tma_ppo_minibatch_grad(params, d, rb, &mb, hp, grad, workspace, stream);
        if (rc) return rc;
        rc = tma_ppo_adam_step_local(params, grad, exp_avg, exp_avg_sq, d, step, lr, beta1, beta2, eps, max_grad_norm, workspace, stream, count);
        if (rc) return rc;
    }
    return TMA_OK;
}

int tma_ppo_train_epoch_dp(float *params, const tma_policy_dims *d, const tma_rollout *rb, uint32_t perm_seed, uint32_t perm_epoch, int64_t batch_size,
                           int64_t prepared_batch, int stats_world, const tma_ppo_hparams *hp, float *grad, float *exp_avg, float *exp_avg_sq,
                           int64_t first_step, double lr, double beta1, double beta2, double eps, double max_grad_norm, double grad_scale,
                           tma_allreduce_fn allreduce, void *ctx, void *workspace, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !rb || !hp || !grad || !exp_avg || !exp_avg_sq || !workspace || !allreduce) return fail(TMA_ERR_INVALID, "tma_ppo_train_epoch_dp: null argument");
    if (rb->T < 1 || rb->N < 1 || batch_size < 1 || first_step < 1) return fail(TMA_ERR_INVALID, "tma_ppo_train_epoch_dp: T, N, batch_size and first_step must be >= 1");
    if (prepared_batch != 0 && prepared_batch != batch_size) return fail(TMA_ERR_INVALID, "tma_ppo_train_epoch_dp: prepared_batch must be 0 or batch_size");
    if (stats_world < 0 || (stats_world > 0 && prepared_batch == 0)) return fail(TMA_ERR_INVALID, "tma_ppo_train_epoch_dp: global statistics need a prepared epoch");
    const int64_t total = (int64_t)rb->T * rb->N;
    const PLayout L = layout_of(d);
    // (The optimizer step is NOT folded into the next gradient launch here, as tma_ppo_train_epoch_local does: the clip norm must come from the
    // all-reduced gradient, and taking it in every workgroup's prologue -- 147 64-lane f64 shuffle trees through the LDS crossbar of each CU --
    // measured 3.4 us per minibatch SLOWER than the sum-of-squares + optimizer launches it would replace: DESIGN.md section 10.)
    int64_t step = first_step;
    // Round 5, H = 64 fast path: the dependent chain of a data-parallel minibatch was gradient -> slab_reduce -> all-reduce -> grad_sumsq64 -> Adam.
    // What the collective forces is only that the NORM PARTIALS come from the all-reduced gradient; the step itself can still run where the
    // single-GPU epoch runs it -- in the prologue of the next gradient launch (AdamFold, with the 1/world scale of the summed gradient in the
    // same place adam_scatter_h64_kernel applies it).  Chain: gradient(+ step k - 1) -> slab_reduce (overwrite) -> all-reduce -> grad_sumsq64:
    // one dependent launch fewer per minibatch, the epoch's last step by the ordinary optimizer launch.  Same routine on the same inputs
    // as the unfolded sequence: bit-identical (tests/test_dist_gpu.py, test_native_data_parallel_epoch_equals_the_single_gpu_epoch).
    // (Taking the partials inside the optimizer kernel instead -- every block re-summing the 37 KB gradient in shuffle-tree order, no
    // grad_sumsq64 launch -- was built first and measured 1.2 us per minibatch SLOWER than the launch it removed: DESIGN.md section 10.)
    const int64_t tail = total % batch_size;
    static const bool no_dp_fold = getenv("TMA_DP_NO_FOLD") != nullptr || getenv("TMA_NO_ADAM_FOLD") != nullptr;  // A/B switch: the round-4 chain
    if (L.img_pi >= 0 && L.P <= 64 * 256 && prepared_batch == batch_size && batch_size >= 256 && (tail == 0 || tail >= 256) && total > batch_size && !no_dp_fold) {
        char *ws = static_cast<char *>(workspace);
        hipStream_t s = (hipStream_t)stream;
        const int64_t Pp = ((int64_t)L.P + 3) & ~(int64_t)3;
        float *alt = reinterpret_cast<float *>(ws + fold_state_offset(L));
        float *bufs[2][3] = {{params, exp_avg, exp_avg_sq}, {alt, alt + Pp, alt + 2 * Pp}};
        int cur = 0;
        double *sqp = sq_partials(ws, L);
        const int n_part = (int)ceil_div(L.P, 64);
        // the library's own communicator with its peer exchange on (tma_comm_p2p_enable) and bound to this stream: fuse the exchange
        static const bool no_p2p_fuse = getenv("TMA_P2P_NO_FUSE") != nullptr;  // A/B switch: push / pull as launches of their own (tma_comm_allreduce)
        tma_comm *comm = allreduce == &tma_comm_allreduce_cb ? static_cast<tma_comm *>(ctx) : nullptr;
        const bool fused = comm && !no_p2p_fuse && tma_comm_p2p_ready(comm, L.P) && tma_comm_bound_stream(comm) == s;
        for (int64_t start = 0; start < total; start += batch_size, step++) {
            const int64_t count = start + batch_size <= total ? batch_size : total - start;
            const tma_minibatch mb{nullptr, perm_seed, perm_epoch, start, count, prepared_batch, stats_world > 0 ? count * stats_world : 0};
            AdamFold f{};
            if (start > 0) {  // the step of the previous minibatch (index step - 1)
                const double bc1 = 1.0 - pow(beta1, (double)(step - 1)), bc2 = 1.0 - pow(beta2, (double)(step - 1));
                f = AdamFold{grad, sqp, n_part, bufs[cur][0], bufs[cur][1], bufs[cur][2], bufs[cur ^ 1][0], bufs[cur ^ 1][1],
                             bufs[cur ^ 1][2], (float)max_grad_norm, (float)(lr / bc1), (float)beta1, (float)beta2, (float)sqrt(bc2), (float)eps,
                             reinterpret_cast<double *>(ws + WS_NORM_OUT), (float)grad_scale};
            }
            if (fused) {
                // peer exchange, fused: slab_reduce_kernel stores this rank's reduced gradient into every rank's inbox, the sum-of-squares pass
                // reads the rank-ordered sum out of this rank's own -- no collective launch between the two
                PeerPush push;
                PeerPull pull;
                rc = tma_comm_p2p_next(comm, L.P, &push, &pull);
                if (rc) return rc;
                rc = minibatch_grad_impl(params, d, rb, &mb, hp, grad, workspace, stream, start > 0 ? &f : nullptr, 1, &push);
                if (rc) return rc;
                if (start > 0) cur ^= 1;
                const int timed = tma_comm_time_begin(comm, s);
                grad_pull_sumsq64_kernel<<<dim3((unsigned)n_part), dim3(64), 0, s>>>(grad, L.P, (float)grad_scale, sqp, pull);
                TMA_LAUNCH_CHECK();
                if (timed) tma_comm_time_end(comm, s);
                continue;
            }
            rc = minibatch_grad_impl(params, d, rb, &mb, hp, grad, workspace, stream, start > 0 ? &f : nullptr, 1);
            if (rc) return rc;
            if (start > 0) cur ^= 1;
            if (allreduce(ctx, grad, L.P) != 0) return fail(TMA_ERR_INVALID, "tma_ppo_train_epoch_dp: the all-reduce callback failed");
            grad_sumsq64_kernel<<<dim3((unsigned)n_part), dim3(64), 0, s>>>(grad, L.P, (float)grad_scale, sqp);
            TMA_LAUNCH_CHECK();
        }
        const double bc1 = 1.0 - pow(beta1, (double)(step - 1)), bc2 = 1.0 - pow(beta2, (double)(step - 1));
        adam_scatter_h64_kernel<<<dim3((unsigned)ceil_div(L.P, 256)), dim3(256), 0, s>>>(
            params, grad, exp_avg, exp_avg_sq, L, sqp, n_part, (float)max_grad_norm, (float)(lr / bc1), (float)beta1, (float)beta2,
            (float)sqrt(bc2), (float)eps, reinterpret_cast<double *>(ws + WS_NORM_OUT), (float)grad_scale, bufs[cur][0], bufs[cur][1], bufs[cur][2]);
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    for (int64_t start = 0; start < total; start += batch_size, step++) {
        const int64_t count = start + batch_size <= total ? batch_size : total - start;
        const tma_minibatch mb{nullptr, perm_seed, perm_epoch, start, count, prepared_batch, stats_world > 0 ? count * stats_world : 0};
        rc = tma_ppo_minibatch_grad(params, d, rb, &mb, hp, grad, workspace, stream);
        if (rc) return rc;
        if (allreduce(ctx, grad, L.P) != 0) return fail(TMA_ERR_INVALID, "tma_ppo_train_epoch_dp: the all-reduce callback failed");
        rc = tma_ppo_adam_step(params, grad, exp_avg, exp_avg_sq, d, step, lr, beta1, beta2, eps, max_grad_norm, grad_scale, workspace, stream);
        if (rc) return rc;
    }
    return TMA_OK;
}

// sample records (tma_rollout.packed): thread = buffer row; {obs | zero padding | log_prob, advantage, action bits, return}
__global__ __launch_bounds__(256) void pack_samples_kernel(Rollout rb, int D, int rs, int64_t total, float *__restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= total) return;
    const int xs = rs - 4;
    float x[8];
#pragma unroll
    for (int c = 0; c < 8; c++) x[c] = (c < D) ? rb.obs[row * D + c] : 0.0f;
    float4 *dst = reinterpret_cast<float4 *>(out + row * rs);
    dst[0] = float4{x[0], x[1], x[2], x[3]};
    if (xs == 8) dst[1] = float4{x[4], x[5], x[6], x[7]};
    dst[xs >> 2] = float4{rb.log_probs[row], rb.advantages[row], __int_as_float(static_cast<const int32_t *>(rb.actions)[row]), rb.returns[row]};
}

#ifdef TMA_WIDE_PHASE_TICKS
extern "C" int tma_debug_wide_ticks(unsigned long long *out32, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(tma::g_wide_ticks), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(tma::g_wide_ticks), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

int64_t tma_ppo_packed_floats(const tma_policy_dims *d, int T, int64_t N) {
    if (!d || check_dims(d) || T < 1 || N < 1) return 0;
    const PLayout L = layout_of(d);
    return (int64_t)rec_floats(L) * T * N;
}

int tma_ppo_pack_samples(const tma_rollout *rb, const tma_policy_dims *d, float *packed_out, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!rb || !packed_out || !rb->obs || !rb->actions || !rb->log_probs || !rb->advantages || !rb->returns || rb->T < 1 || rb->N < 1)
        return fail(TMA_ERR_INVALID, "tma_ppo_pack_samples: null buffer");
    const PLayout L = layout_of(d);
    const int rs = rec_floats(L);
    if (rs == 0) return fail(TMA_ERR_INVALID, "tma_ppo_pack_samples: this policy shape has no packed sample records (tma_ppo_packed_floats == 0)");
    const int64_t total = (int64_t)rb->T * rb->N;
    const Rollout R{rb->obs, rb->actions, rb->log_probs, rb->advantages, rb->returns, rb->T, rb->N, nullptr};
    pack_samples_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream>>>(R, L.D, rs, total, packed_out);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

int tma_ppo_epoch_prepare(const tma_rollout *rb, const tma_minibatch *epoch, int64_t batch_size, const tma_policy_dims *d, void *workspace,
                          void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!rb || !epoch || !workspace) return fail(TMA_ERR_INVALID, "tma_ppo_epoch_prepare: null argument");
    if (!rb->advantages || rb->T < 1 || rb->N < 1) return fail(TMA_ERR_INVALID, "rollout view: advantages / T / N");
    const int64_t total = (int64_t)rb->T * rb->N;
    if (total > OFFS_CAP) return fail(TMA_ERR_INVALID, "epoch of %lld samples exceeds the %lld-entry offsets cache", (long long)total, (long long)OFFS_CAP);
    if (batch_size < 256) return fail(TMA_ERR_INVALID, "tma_ppo_epoch_prepare needs batch_size >= 256 (got %lld)", (long long)batch_size);
    if (epoch->start != 0 || epoch->count != total) return fail(TMA_ERR_INVALID, "epoch descriptor must cover [0, T*N)");
    const PLayout L = layout_of(d);
    char *ws = static_cast<char *>(workspace);
    const int64_t offs_base = WS_SLABS + (int64_t)slab_cap(L) * L.P * 4;
    int stride = (int)ceil_div(batch_size, 1024);
    if (stride > ADV_BLOCKS) stride = ADV_BLOCKS;
    const int64_t n_mb = ceil_div(total, batch_size);
    if (n_mb > 65535) return fail(TMA_ERR_INVALID, "too many minibatches per epoch (%lld)", (long long)n_mb);
    Minibatch M{epoch->indices, epoch->perm_seed, epoch->perm_epoch, 0, total, total, nullptr, total, nullptr, 0};
    adv_partial_kernel<<<dim3(stride, (unsigned)n_mb), dim3(256), 0, (hipStream_t)stream>>>(
        rb->advantages, M, rb->T, rb->N, reinterpret_cast<double *>(ws + offs_base + OFFS_CAP * 4), reinterpret_cast<int32_t *>(ws + offs_base), batch_size);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

int tma_ppo_epoch_adv_sums(void *workspace, const tma_policy_dims *d, int64_t batch_size, int64_t total, double *sums, int direction, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!workspace || !sums) return fail(TMA_ERR_INVALID, "tma_ppo_epoch_adv_sums: null argument");
    if (batch_size < 256 || total < 1 || total > OFFS_CAP) return fail(TMA_ERR_INVALID, "tma_ppo_epoch_adv_sums: same limits as tma_ppo_epoch_prepare");
    if (direction != 0 && direction != 1) return fail(TMA_ERR_INVALID, "direction must be 0 (export) or 1 (import)");
    const PLayout L = layout_of(d);
    char *ws = static_cast<char *>(workspace);
    const int64_t offs_base = WS_SLABS + (int64_t)slab_cap(L) * L.P * 4;
    int stride = (int)ceil_div(batch_size, 1024);
    if (stride > ADV_BLOCKS) stride = ADV_BLOCKS;
    const unsigned n_mb = (unsigned)ceil_div(total, batch_size);
    double *partials = reinterpret_cast<double *>(ws + offs_base + OFFS_CAP * 4);
    if (direction == 0) adv_epoch_sums_kernel<false><<<dim3(n_mb), dim3(64), 0, (hipStream_t)stream>>>(partials, stride, batch_size, total, sums);
    else adv_epoch_sums_kernel<true><<<dim3(n_mb), dim3(64), 0, (hipStream_t)stream>>>(partials, stride, batch_size, total, sums);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

int tma_ppo_adam_step(float *params, float *grad, float *exp_avg, float *exp_avg_sq, const tma_policy_dims *d, int64_t step, double lr, double beta1,
                      double beta2, double eps, double max_grad_norm, double grad_scale, void *workspace, void *stream) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !grad || !exp_avg || !exp_avg_sq || !workspace) return fail(TMA_ERR_INVALID, "tma_ppo_adam_step: null buffer");
    if (step < 1) return fail(TMA_ERR_INVALID, "Adam step index must be >= 1");
    hipStream_t s = (hipStream_t)stream;
    const PLayout L = layout_of(d);
    char *ws = static_cast<char *>(workspace);
    double *partials = reinterpret_cast<double *>(ws + WS_NORM_PART);
    double *norm_out = reinterpret_cast<double *>(ws + WS_NORM_OUT);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1, bc2_sqrt = sqrt(bc2);
    double *sqp = sq_partials(ws, L);
    const bool scat_h64 = L.img_pi >= 0 && L.P <= 64 * 256, scat_wide = L.bf16 || L.fr_pi >= 0;
    if (sqp && (scat_h64 || scat_wide)) {
        // layouts whose derived copies the optimizer kernel scatters itself: norm partials of the (all-reduced, scaled) gradient, then
        // ONE multi-block Adam + scatter launch -- two launches instead of three to five
        const int n_part = (int)ceil_div(L.P, 64);
        grad_sumsq64_kernel<<<dim3((unsigned)n_part), dim3(64), 0, s>>>(grad, L.P, (float)grad_scale, sqp);
        TMA_LAUNCH_CHECK();
        if (scat_h64)
            adam_scatter_h64_kernel<<<dim3((unsigned)ceil_div(L.P, 256)), dim3(256), 0, s>>>(params, grad, exp_avg, exp_avg_sq, L, sqp, n_part,
                                                                                             (float)max_grad_norm, (float)step_size, (float)beta1, (float)beta2,
                                                                                             (float)bc2_sqrt, (float)eps, norm_out, (float)grad_scale, params,
                                                                                             exp_avg, exp_avg_sq);
        else
            adam_scatter_wide_kernel<<<dim3((unsigned)ceil_div(L.P, 256)), dim3(256), 0, s>>>(params, grad, exp_avg, exp_avg_sq, L, sqp, n_part,
                                                                                              (float)max_grad_norm, (float)step_size, (float)beta1, (float)beta2,
                                                                                              (float)bc2_sqrt, (float)eps, norm_out, (float)grad_scale);
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    if (L.P <= 32768) {
        opt_small_kernel<<<dim3(1), dim3(1024), 0, s>>>(params, grad, exp_avg, exp_avg_sq, L, (float)grad_scale, (float)max_grad_norm, (float)step_size,
                                                        (float)beta1, (float)beta2, (float)bc2_sqrt, (float)eps, norm_out);
        TMA_LAUNCH_CHECK();
        return launch_sync(params, L, s);
    }
    int nb = (int)ceil_div(L.P, 1024);
    if (nb > 256) nb = 256;
    grad_sumsq_kernel<<<dim3(nb), dim3(256), 0, s>>>(grad, L.P, (float)grad_scale, partials);
    TMA_LAUNCH_CHECK();
    adam_kernel<<<dim3(nb), dim3(256), 0, s>>>(params, grad, exp_avg, exp_avg_sq, L.P, (float)grad_scale, partials, nb, (float)max_grad_norm,
                                               (float)step_size, (float)beta1, (float)beta2, (float)bc2_sqrt, (float)eps, norm_out);
    TMA_LAUNCH_CHECK();
    return launch_sync(params, L, s);
}

int tma_ppo_adam_step_local(float *params, float *grad, float *exp_avg, float *exp_avg_sq, const tma_policy_dims *d, int64_t step, double lr,
                            double beta1, double beta2, double eps, double max_grad_norm, void *workspace, void *stream, int64_t last_count) {
    int rc = enter(d);
    if (rc) return rc;
    if (!params || !grad || !exp_avg || !exp_avg_sq || !workspace) return fail(TMA_ERR_INVALID, "tma_ppo_adam_step_local: null buffer");
    if (step < 1) return fail(TMA_ERR_INVALID, "Adam step index must be >= 1");
    const PLayout L = layout_of(d);
    // the partials exist only when the last tma_ppo_minibatch_grad ended in slab_reduce_kernel: the H == 64 persistent kernel
    // (>= 256 samples), the bf16 column-parallel kernel (any size) or the f32 column-parallel kernel (>= 128 samples)
    char *ws = static_cast<char *>(workspace);
    const bool h64 = L.img_pi >= 0 && last_count >= 256;
    const bool wide_f32 = !L.bf16 && (L.H == 128 || L.H == 192 || L.H == 256) && last_count >= 128 && grad_wide_smem_bytes(L) <= 160 * 1024 &&
                          getenv("TMA_FORCE_WIDE") == nullptr;
    const double *sqp = sq_partials(ws, L);
    if (!(h64 || L.bf16 || wide_f32) || !sqp || last_count < 1)
        return tma_ppo_adam_step(params, grad, exp_avg, exp_avg_sq, d, step, lr, beta1, beta2, eps, max_grad_norm, 1.0, workspace, stream);
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1, bc2_sqrt = sqrt(bc2);
    if (h64)
        adam_scatter_h64_kernel<<<dim3((unsigned)ceil_div(L.P, 256)), dim3(256), 0, s>>>(
            params, grad, exp_avg, exp_avg_sq, L, sqp, (int)ceil_div(L.P, 64), (float)max_grad_norm, (float)step_size, (float)beta1, (float)beta2,
            (float)bc2_sqrt, (float)eps, reinterpret_cast<double *>(ws + WS_NORM_OUT), 1.0f, params, exp_avg, exp_avg_sq);
    else
        adam_scatter_wide_kernel<<<dim3((unsigned)ceil_div(L.P, 256)), dim3(256), 0, s>>>(
            params, grad, exp_avg, exp_avg_sq, L, sqp, (int)ceil_div(L.P, 64), (float)max_grad_norm, (float)step_size, (float)beta1, (float)beta2,
            (float)bc2_sqrt, (float)eps, reinterpret_cast<double *>(ws + WS_NORM_OUT), 1.0f);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

int tma_ppo_permutation(uint32_t perm_seed, uint32_t perm_epoch, int64_t total, int64_t *indices_out_host) {
    if (!indices_out_host || total < 1 || total > 0x7fffffffLL) return fail(TMA_ERR_INVALID, "tma_ppo_permutation: null output or total outside [1, 2^31)");
    for (int64_t j = 0; j < total; j++) indices_out_host[j] = (int64_t)perm_index(perm_seed, perm_epoch, (uint32_t)j, (uint32_t)total);
    return TMA_OK;
}

int tma_debug_time_grad_kernel(int enable) {
    if (enable && !g_ev0) {
        TMA_HIP(hipEventCreate(&g_ev0));
        TMA_HIP(hipEventCreate(&g_ev1));
    }
    g_time_grad = enable != 0;
    g_ev_valid = false;
    return TMA_OK;
}

int tma_debug_last_grad_kernel_us(float *us_out) {
    if (!us_out) return fail(TMA_ERR_INVALID, "us_out is null");
    if (!g_ev_valid) return fail(TMA_ERR_INVALID, "no timed tma_ppo_minibatch_grad launch (enable tma_debug_time_grad_kernel first; column-parallel / H=64 paths only)");
    TMA_HIP(hipEventSynchronize(g_ev1));
    float ms = 0.0f;
    TMA_HIP(hipEventElapsedTime(&ms, g_ev0, g_ev1));
    *us_out = ms * 1e3f;
    return TMA_OK;
}

// Two-phase form of tma_ppo_pop_stats for a loop that must not wait for the update it has just queued: _enqueue copies the raw statistic
// slots into the caller's staging buffer (tma_ppo_stats_staging_bytes() bytes of HOST memory, pinned for a truly asynchronous copy) and clears
// them, stream-ordered, and returns at once; once the caller knows that point of the stream has passed (an event, a later synchronisation),
// _fold turns the staged slots into the eight values tma_ppo_pop_stats returns.
int64_t tma_ppo_stats_staging_bytes(void) { return (int64_t)sizeof(double) * (MAX_GRAD_BLOCKS * 8 + 2) + 16; }

int tma_ppo_stats_enqueue(void *workspace, void *staging_host, void *stream) {
    if (!workspace || !staging_host) return fail(TMA_ERR_INVALID, "tma_ppo_stats_enqueue: null argument");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, workspace) == hipSuccess) (void)hipSetDevice(attr.device);
    hipStream_t s = (hipStream_t)stream;
    char *ws = static_cast<char *>(workspace);
    double *tmp = static_cast<double *>(staging_host);
    TMA_HIP(hipMemcpyAsync(tmp, ws + WS_STATS, sizeof(double) * MAX_GRAD_BLOCKS * 8, hipMemcpyDeviceToHost, s));
    TMA_HIP(hipMemcpyAsync(tmp + MAX_GRAD_BLOCKS * 8, ws + WS_NORM_OUT, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
    TMA_HIP(hipMemcpyAsync(tmp + MAX_GRAD_BLOCKS * 8 + 2, ws + WS_PERSIST_ERR, sizeof(int), hipMemcpyDeviceToHost, s));
    TMA_HIP(hipMemsetAsync(ws + WS_PERSIST_ERR, 0, sizeof(int), s));
    TMA_HIP(hipMemsetAsync(ws + WS_STATS, 0, sizeof(double) * MAX_GRAD_BLOCKS * 8, s));
    return TMA_OK;
}

int tma_ppo_stats_fold(const void *staging_host, double *out8_host) {
    if (!staging_host || !out8_host) return fail(TMA_ERR_INVALID, "tma_ppo_stats_fold: null argument");
    const double *tmp = static_cast<const double *>(staging_host);
    int persist_err = 0;
    memcpy(&persist_err, tmp + MAX_GRAD_BLOCKS * 8 + 2, sizeof(int));
    if (persist_err)
        return fail(TMA_ERR_HIP, "the persistent epoch kernel could not place / synchronise its workgroups on one XCD; parameters of that epoch "
                                      "were not updated (set TMA_NO_PERSIST=1 to use the per-minibatch launches)");
    for (int q = 0; q < 6; q++) out8_host[q] = 0.0;
    for (int b = 0; b < MAX_GRAD_BLOCKS; b++)
        for (int q = 0; q < 6; q++) out8_host[q] += tmp[b * 8 + q];
    out8_host[6] = tmp[MAX_GRAD_BLOCKS * 8];
    out8_host[7] = tmp[MAX_GRAD_BLOCKS * 8 + 1];
    return TMA_OK;
}

int tma_ppo_pop_stats(void *workspace, double *out8_host, void *stream) {
    if (!workspace || !out8_host) return fail(TMA_ERR_INVALID, "null argument");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, workspace) == hipSuccess) (void)hipSetDevice(attr.device);  // callers may be worker threads
    hipStream_t s = (hipStream_t)stream;
    std::vector<double> tmpv(MAX_GRAD_BLOCKS * 8 + 2);
    double *tmp = tmpv.data();
    char *ws = static_cast<char *>(workspace);
    TMA_HIP(hipMemcpyAsync(tmp, ws + WS_STATS, sizeof(double) * MAX_GRAD_BLOCKS * 8, hipMemcpyDeviceToHost, s));
    TMA_HIP(hipMemcpyAsync(tmp + MAX_GRAD_BLOCKS * 8, ws + WS_NORM_OUT, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
    int persist_err = 0;
    TMA_HIP(hipMemcpyAsync(&persist_err, ws + WS_PERSIST_ERR, sizeof(int), hipMemcpyDeviceToHost, s));
    TMA_HIP(hipMemsetAsync(ws + WS_PERSIST_ERR, 0, sizeof(int), s));
    TMA_HIP(hipMemsetAsync(ws + WS_STATS, 0, sizeof(double) * MAX_GRAD_BLOCKS * 8, s));
    TMA_HIP(hipStreamSynchronize(s));
    if (persist_err)
        return fail(TMA_ERR_HIP, "the persistent epoch kernel could not place / synchronise its workgroups on one XCD; parameters of that epoch "
                                      "were not updated (set TMA_NO_PERSIST=1 to use the per-minibatch launches)");
    for (int q = 0; q < 6; q++) out8_host[q] = 0.0;
    for (int b = 0; b < MAX_GRAD_BLOCKS; b++)
        for (int q = 0; q < 6; q++) out8_host[q] += tmp[b * 8 + q];
    out8_host[6] = tmp[MAX_GRAD_BLOCKS * 8];
    out8_host[7] = tmp[MAX_GRAD_BLOCKS * 8 + 1];
    return TMA_OK;
}

}  // extern "C"

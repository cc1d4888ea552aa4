// tma_rollout.hip -- native rollout driver: the inner loop of SB3 OnPolicyAlgorithm.collect_rollouts without a Python
// round-trip per vector step.  Per step t it enqueues, on one HIP stream:
//   policy_act(obs[t]) -> actions[t], values[t], log_probs[t]
//   env step(actions[t]) -> obs[t+1], rewards[t], terminated[t], truncated[t], terminal_obs   (+ reset-ring refill when due)
//   timeout bootstrap: rewards[t] += gamma * V(terminal_obs) where truncated[t]   (folded into step t+1's policy launch)
// and, after the last step, last_values = V(obs[T]).  Reference call path: model.learn() at
// /root/reference/backend/mlagents/training.py:166-170 -> SB3 collect_rollouts (SURVEY.md §3.1 hot loop A, App. C.6).
#include "tma_internal.h"
#include "tma_h64_tile.h"

#include <cstdlib>

namespace tma {
#include "tma_wide_bf16.h"  // bf16 fragment helpers and image layout shared with the update / forward kernels
}  // namespace tma

namespace tma {

// ------------------------------------------------------------------------------------------
// Fused rollout chunk (H = 64 policies on GridWorld / Push / Ball3D): ONE launch advances every env by n_steps vector steps.
// A wavefront owns a tile of 16 envs for the whole chunk: the two forward weight images are staged in LDS once, the tile's
// observations live in LDS between steps, the 16 env states live in the registers of 16 "owner" lanes (lane = 16*g + r owns
// tile row 4*g + r, exactly where the policy epilogue leaves that row's action / log-prob / value), and per step the wave runs
// value net -> policy net -> categorical sample -> env step (+ auto-reset from the ring, Monitor sums) -> obs into LDS and
// into rollout-buffer slot t+1 -> (only if some env of the tile hit its time limit) value net on the terminal observations
// for the timeout bootstrap.  No kernel boundary, no HBM round trip of the observation between policy and env.
// ------------------------------------------------------------------------------------------
struct ChunkPtrs {
    float *obs;
    int32_t *actions;
    float *rewards, *values, *log_probs;
    uint8_t *terminated, *truncated;
};

// (round 3) Both variants run the forward passes as the TRANSPOSED REGISTER CHAIN of the update kernels (h64t_forward, tma_h64_tile.h):
// lane (g, s) works for env s of the tile, the observation features go straight from LDS words into MFMA B operands, no activation is
// written to LDS, the action comes out of two cross-lane-group reductions (h64t_act) in every lane of the env's column, and the 16 lanes
// of lane group 0 own the env states.  The LDS round-trip chain this replaces cost 3.6 us per vector step at 4096 envs.
template <class T, bool STORE = true>
__device__ __forceinline__ void chunk_env_step(const EnvView &v, const ChunkPtrs &b, typename T::S &s, double &er, uint32_t &ce, int64_t N, int64_t i,
                                               int t, int act, float lp, float *Xn, float *XT, float &rew32, bool &tr_out, bool write_reward_if_trunc,
                                               double &sret, double &slen, double &scnt, bool *te_out = nullptr) {
    // STORE = false (rollout_chunk4_h64_kernel): the step's rollout-buffer rows are written by another wave from what this one leaves in LDS
    // (action, reward, flags: the caller; next observation: Xn) -- everything else is the same
    constexpr int D = T::OBS;
    const int64_t off = (int64_t)t * N + i;
    if constexpr (STORE) {
        b.actions[off] = act;
        b.log_probs[off] = lp;
    }
    double r;
    bool done;
    T::step(s, act, nullptr, r, done);
    const int steps = T::steps(s);
    const bool hit = steps >= T::MAXSTEPS;  // adapter rule, backend/mlagents/envs.py:139-145
    const bool te = done && !hit, tr = hit;
    er += r;
    rew32 = (float)r;
    if constexpr (STORE) {
        b.terminated[off] = (uint8_t)te;
        b.truncated[off] = (uint8_t)tr;
    }
    if (te_out != nullptr) *te_out = te;
    float o[D];
    if (te || tr) {
        if (tr) {
            T::obs(s, o);
#pragma unroll
            for (int c = 0; c < D; c++) XT[c] = o[c];
        }
        sret += er, slen += (double)steps, scnt += 1.0;
        log_episode(v, i, er, steps);
        er = 0.0;
        ce += 1;
        uint32_t rec[T::RW];
        const uint32_t *slot = v.ring + ((int64_t)(ce % (uint32_t)v.D) * T::RW) * N + i;
#pragma unroll
        for (int w = 0; w < T::RW; w++) rec[w] = slot[(int64_t)w * N];
        T::from_rec(rec, s);
    }
    T::obs(s, o);
    if constexpr (STORE) store_obs<D>(b.obs + ((int64_t)(t + 1) * N + i) * D, o);
#pragma unroll
    for (int c = 0; c < D; c++) Xn[c] = o[c];
    if constexpr (STORE) {
        if (!tr || write_reward_if_trunc) b.rewards[off] = rew32;
    }
    tr_out = tr;
}

constexpr int CH_LDX = 17;  // observation tiles [16 envs][17]: the odd row stride spreads the 16 rows of a feature column over the banks

#ifdef TMA_ROLL_TICKS  // diagnostic build (make libtma_hip_rticks.so, tools/roll_ticks.py): cycles per phase of a vector step, thread 0 of block 0
__device__ unsigned long long g_roll_ticks[8];
#define TMA_RTICK(i)                                                                  \
    do {                                                                              \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_roll_ticks[i] += tn_ - rt_last;    \
        rt_last = tn_;                                                                \
    } while (0)
#else
#define TMA_RTICK(i)
#endif
template <class T>
__global__ __launch_bounds__(256) void rollout_chunk_h64_kernel(EnvView v, const float *__restrict__ params, PLayout L, ChunkPtrs b, int t0, int n_steps,
                                                                uint32_t rng_seed, uint32_t rng_step0, float gamma, int det) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int D = T::OBS, KS1 = (D + 3) >> 2;
    const int A = L.A;
    float *vimg = smem, *pimg = smem + FWD_IMG;
    float *X = smem + 2 * FWD_IMG + (int64_t)wave * (2 * 16 * CH_LDX), *XT = X + 16 * CH_LDX;  // this wave's observation / terminal-observation tiles
    stage_fwd_image(params + L.img_vf, vimg);
    stage_fwd_image(params + L.img_pi, pimg);
    __syncthreads();
    const int64_t N = v.N;
    const int64_t tile = (int64_t)blockIdx.x * wpb + wave;
    if (tile * 16 >= N) return;
    const int64_t row0 = tile << 4;
    const int64_t i = row0 + r16;           // the env this lane's column stands for
    const bool active = g == 0 && i < N;    // ... and the lane that owns its state and writes its outputs
    typename T::S s;
    double er = 0.0;
    uint32_t ce = 0;
    if (active) {
        T::unpack(v.st, N, i, s);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    for (int e = lane; e < 16 * CH_LDX; e += 64) {
        const int row = e / CH_LDX, c = e - row * CH_LDX;
        X[e] = (row0 + row < N && c < D) ? b.obs[((int64_t)t0 * N + row0 + row) * D + c] : 0.0f;
        XT[e] = 0.0f;
    }
    double sret = 0.0, slen = 0.0, scnt = 0.0;
    for (int k = 0; k < n_steps; k++) {
        const int t = t0 + k;
        float xb[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) xb[ks] = X[r16 * CH_LDX + ((4 * ks + g < D) ? 4 * ks + g : 16)];  // (column 16: a zero)
        f32x4 o0, o1;
        h64t_forward<KS1>(vimg, vimg + IMG_FWD_FLOATS, vimg + IMG_FWD_FLOATS + 64, vimg + IMG_FWD_FLOATS + 128, xb, KS1, o0, o1, lane);
        const float value = o0[0] + o1[0];
        h64t_forward<KS1>(pimg, pimg + IMG_FWD_FLOATS, pimg + IMG_FWD_FLOATS + 64, pimg + IMG_FWD_FLOATS + 128, xb, KS1, o0, o1, lane);
        int act;
        float lp;
        h64t_act(o0, o1, A, rng_seed, v.env_offset + (uint32_t)i, rng_step0 + (uint32_t)t, det, act, lp, lane);
        bool tr_flag = false;
        float rew32 = 0.0f;
        if (active) {
            b.values[(int64_t)t * N + i] = value;
            chunk_env_step<T>(v, b, s, er, ce, N, i, t, act, lp, X + r16 * CH_LDX, XT + r16 * CH_LDX, rew32, tr_flag, true, sret, slen, scnt);
        }
        if (__ballot(tr_flag) != 0ull) {  // timeout bootstrap: rewards += gamma * V(terminal_obs) where truncated
            float xt[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) xt[ks] = XT[r16 * CH_LDX + ((4 * ks + g < D) ? 4 * ks + g : 16)];
            h64t_forward<KS1>(vimg, vimg + IMG_FWD_FLOATS, vimg + IMG_FWD_FLOATS + 64, vimg + IMG_FWD_FLOATS + 128, xt, KS1, o0, o1, lane);
            if (tr_flag) {
                const float gv = gamma * (o0[0] + o1[0]);
                b.rewards[(int64_t)t * N + i] = rew32 + gv;
            }
        }
    }
    if (active) {
        T::pack(v.st, N, i, s);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sret += __shfl_down(sret, o, 64);
        slen += __shfl_down(slen, o, 64);
        scnt += __shfl_down(scnt, o, 64);
    }
    if (lane == 0 && scnt > 0.0) {  // Monitor aggregate: the 16 tiles of a 256-env slot add with f64 atomics (few per launch)
        double *slot = v.stats + (row0 >> 8) * 3;
        atomicAdd(slot + 0, sret);
        atomicAdd(slot + 1, slen);
        atomicAdd(slot + 2, scnt);
    }
}

// Two-wave variant for the BASELINE shape (4096 envs = 256 tiles = one tile per CU): the dependent chain of a vector step is split
// over two SIMDs of the CU.  Wave 0 runs the policy net, samples, steps the 16 envs and writes obs / actions / log-probs / flags;
// wave 1 runs the value net on the same observation tile (and, one step later, the timeout bootstrap of any env that hit its time
// limit).  Observations and terminal observations are double-buffered in LDS by step parity, so the only synchronisation is one
// workgroup barrier per step.  Rewards of truncated rows are written by wave 1 (reward + gamma * V(terminal obs)), all others by
// wave 0 -- the two waves never store to the same address.  Same arithmetic per element as rollout_chunk_h64_kernel.
template <class T>
__global__ __launch_bounds__(128) void rollout_chunk2_h64_kernel(EnvView v, const float *__restrict__ params, PLayout L, ChunkPtrs b, int t0, int n_steps,
                                                                 uint32_t rng_seed, uint32_t rng_step0, float gamma, int det) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int D = T::OBS, KS1 = (D + 3) >> 2;
    const int A = L.A;
    float *vimg = smem, *pimg = smem + FWD_IMG;
    float *X0 = smem + 2 * FWD_IMG;                    // [2][16][CH_LDX] observation tile, by step parity
    float *XT0 = X0 + 2 * 16 * CH_LDX;                 // [2][16][CH_LDX] terminal observations of truncated rows
    float *rw = XT0 + 2 * 16 * CH_LDX;                 // [2][16] reward of a truncated row (before the bootstrap)
    int *trf = reinterpret_cast<int *>(rw + 32);       // [2][16] row truncated at this parity's step
    int *flag = trf + 32;                              // [2] any row truncated
    stage_fwd_image(params + L.img_vf, vimg);
    stage_fwd_image(params + L.img_pi, pimg);
    const int64_t N = v.N;
    const int64_t row0 = (int64_t)blockIdx.x << 4;
    const int64_t i = row0 + r16;           // the env this lane's column stands for
    const bool active = g == 0 && i < N;    // ... and the lane that owns its state (wave 0) / writes its value (wave 1)
    typename T::S s;
    double er = 0.0;
    uint32_t ce = 0;
    if (wave == 0 && active) {
        T::unpack(v.st, N, i, s);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    if (wave == 0) {
        for (int e = lane; e < 2 * 16 * CH_LDX; e += 64) {  // parity 0: the observations of step t0; every other word of the four tiles: zero
            const int row = e / CH_LDX, c = e - row * CH_LDX;
            X0[e] = (row < 16 && row0 + row < N && c < D) ? b.obs[((int64_t)t0 * N + row0 + row) * D + c] : 0.0f;
            XT0[e] = 0.0f;
        }
        if (lane < 2) flag[lane] = 0;
        if (lane < 32) trf[lane] = 0;
    }
    __syncthreads();
    double sret = 0.0, slen = 0.0, scnt = 0.0;
    const float *img = wave == 1 ? vimg : pimg;
    // this wave's net in registers for the whole launch (h64t_forward_r: the arithmetic of h64t_forward, no LDS weight reads in the step loop)
    H64FwdRegs<KS1> FR;
    h64t_load_fwd<KS1>(img, img + IMG_FWD_FLOATS, img + IMG_FWD_FLOATS + 64, img + IMG_FWD_FLOATS + 128, KS1, FR, lane);
#ifdef TMA_ROLL_TICKS
    unsigned long long rt_last = __builtin_amdgcn_s_memtime();
#endif
    for (int k = 0; k < n_steps; k++) {
        const int t = t0 + k, p = k & 1, q = p ^ 1;
        TMA_RTICK(0);
        const float *X = X0 + p * 16 * CH_LDX;
        float xb[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) xb[ks] = X[r16 * CH_LDX + ((4 * ks + g < D) ? 4 * ks + g : 16)];  // (column 16: a zero)
        f32x4 o0, o1;
        h64t_forward_r<KS1>(FR, xb, KS1, o0, o1);
#ifdef TMA_ROLL_TICKS
        asm volatile("" : "+v"(o0), "+v"(o1));
#endif
        TMA_RTICK(1);
        if (wave == 1) {
            if (active) b.values[(int64_t)t * N + i] = o0[0] + o1[0];
            if (k > 0 && flag[q]) {  // timeout bootstrap of step t-1: rewards = reward + gamma * V(terminal_obs) where truncated
                float xt[KS1];
#pragma unroll
                for (int ks = 0; ks < KS1; ks++) xt[ks] = XT0[q * 16 * CH_LDX + r16 * CH_LDX + ((4 * ks + g < D) ? 4 * ks + g : 16)];
                h64t_forward_r<KS1>(FR, xt, KS1, o0, o1);  // (wave 1 holds the value net)
                if (active && trf[q * 16 + r16]) {
                    const float gv = gamma * (o0[0] + o1[0]);
                    b.rewards[(int64_t)(t - 1) * N + i] = rw[q * 16 + r16] + gv;
                }
            }
        } else {
            int act;
            float lp;
            h64t_act(o0, o1, A, rng_seed, v.env_offset + (uint32_t)i, rng_step0 + (uint32_t)t, det, act, lp, lane);
#ifdef TMA_ROLL_TICKS
            asm volatile("" : "+v"(act), "+v"(lp));
#endif
            TMA_RTICK(2);
            bool tr_flag = false;
            if (active) {
                float rew32;
                chunk_env_step<T>(v, b, s, er, ce, N, i, t, act, lp, X0 + q * 16 * CH_LDX + r16 * CH_LDX, XT0 + p * 16 * CH_LDX + r16 * CH_LDX, rew32, tr_flag,
                                  false, sret, slen, scnt);  // truncated rows: wave 1 writes reward + bootstrap after the barrier
                if (tr_flag) rw[p * 16 + r16] = rew32;
                trf[p * 16 + r16] = tr_flag ? 1 : 0;
            }
            const bool any = __ballot(tr_flag) != 0ull;
            if (lane == 0) flag[p] = any ? 1 : 0;
            TMA_RTICK(3);
        }
        __syncthreads();
        TMA_RTICK(4);
    }
    if (wave == 1) {  // bootstrap of the chunk's last step
        const int q = (n_steps - 1) & 1, t = t0 + n_steps - 1;
        if (n_steps > 0 && flag[q]) {
            float xt[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) xt[ks] = XT0[q * 16 * CH_LDX + r16 * CH_LDX + ((4 * ks + g < D) ? 4 * ks + g : 16)];
            f32x4 o0, o1;
            h64t_forward_r<KS1>(FR, xt, KS1, o0, o1);
            if (active && trf[q * 16 + r16]) {
                const float gv = gamma * (o0[0] + o1[0]);
                b.rewards[(int64_t)t * N + i] = rw[q * 16 + r16] + gv;
            }
        }
        return;
    }
    if (active) {
        T::pack(v.st, N, i, s);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sret += __shfl_down(sret, o, 64);
        slen += __shfl_down(slen, o, 64);
        scnt += __shfl_down(scnt, o, 64);
    }
    if (lane == 0 && scnt > 0.0) {
        double *slot = v.stats + (row0 >> 8) * 3;
        atomicAdd(slot + 0, sret);
        atomicAdd(slot + 1, slen);
        atomicAdd(slot + 2, scnt);
    }
}


// Four-wave variant of the two-wave kernel above (round 6): each net on TWO waves -- layer 1 on both, layer 2 split by output tiles (32 MFMAs a
// wave instead of 64), the partner's activated tiles through LDS, head / action / env step on the net's first wave.  One more workgroup barrier
// per step (the hand-over), 1 k cycles less layer 2 on the policy chain.  Same arithmetic per element (h64t_forward_half).  TMA_ROLL2=1 selects
// the two-wave kernel (A/B).
template <class T>
__global__ __launch_bounds__(256) void rollout_chunk4_h64_kernel(EnvView v, const float *__restrict__ params, PLayout L, ChunkPtrs b, int t0, int n_steps,
                                                                 uint32_t rng_seed, uint32_t rng_step0, float gamma, int det) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int net = wave >> 1, half = wave & 1;  // net 0: policy, 1: value
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int D = T::OBS, KS1 = (D + 3) >> 2;
    const int A = L.A;
    float *vimg = smem, *pimg = smem + FWD_IMG;
    float *X0 = smem + 2 * FWD_IMG;                    // [2][16][CH_LDX] observation tile, by step parity
    float *XT0 = X0 + 2 * 16 * CH_LDX;                 // [2][16][CH_LDX] terminal observations of truncated rows
    float *rw = XT0 + 2 * 16 * CH_LDX;                 // [2][16] reward of a truncated row (before the bootstrap)
    int *trf = reinterpret_cast<int *>(rw + 32);       // [2][16] row truncated at this parity's step
    int *flag = trf + 32;                              // [2] any row truncated
    float *xch = reinterpret_cast<float *>(flag + 4);  // [2 nets][2 tiles][64 lanes][4]: the second wave's activated tiles; then [2][64][4] for the bootstrap pass
    float *gnz = xch + 1536;                           // [2][64 lanes][4] Gumbel noise of a step, by step parity: formed a step ahead by wave 1
    float *lgt = gnz + 512;                            // [2][64 lanes][4] the step's logits (o0 + o1), by step parity: wave 3 forms the log-probability
    int *acts = reinterpret_cast<int *>(lgt + 512);    // [2][16] action, [2][16] reward bits, [2][16] terminated | truncated << 1 of the step
    int *orw = acts + 32, *ofl = orw + 32;
    stage_fwd_image(params + L.img_vf, vimg);
    stage_fwd_image(params + L.img_pi, pimg);
    const int64_t N = v.N;
    const int64_t row0 = (int64_t)blockIdx.x << 4;
    const int64_t i = row0 + r16;
    const bool active = g == 0 && i < N;
    typename T::S s;
    double er = 0.0;
    uint32_t ce = 0;
    if (wave == 0 && active) {
        T::unpack(v.st, N, i, s);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    if (wave == 0) {
        for (int e = lane; e < 2 * 16 * CH_LDX; e += 64) {
            const int row = e / CH_LDX, c = e - row * CH_LDX;
            X0[e] = (row < 16 && row0 + row < N && c < D) ? b.obs[((int64_t)t0 * N + row0 + row) * D + c] : 0.0f;
            XT0[e] = 0.0f;
        }
        if (lane < 2) flag[lane] = 0;
        if (lane < 32) trf[lane] = 0;
    }
    const uint32_t genv = v.env_offset + (uint32_t)i;
    if (wave == 1) *reinterpret_cast<f32x4 *>(gnz + lane * 4) = h64t_gumbel(rng_seed, genv, rng_step0 + (uint32_t)t0, det, lane);
    __syncthreads();
    double sret = 0.0, slen = 0.0, scnt = 0.0;
    const float *img = net == 1 ? vimg : pimg;
    H64FwdRegs<KS1> FR;
    h64t_load_fwd<KS1>(img, img + IMG_FWD_FLOATS, img + IMG_FWD_FLOATS + 64, img + IMG_FWD_FLOATS + 128, KS1, FR, lane);
    float *xq = xch + net * 512, *xq2 = xch + 1024;  // hand-over slots: [2 tiles][64][4] per net; the bootstrap pass's own
#ifdef TMA_ROLL_TICKS
    unsigned long long rt_last = 0;
#endif
    // one forward pass of this wave's net on the tile at `Xp`: layer-2 halves, hand-over, head on the net's first wave (o0 / o1 valid there).
    // EVERY wave of the block must call it (one barrier inside).
    auto forward = [&](const float *Xp, float *slot, bool mine, f32x4 &o0, f32x4 &o1) {
        f32x4 t2[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
        if (mine) {
            float xb[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) xb[ks] = Xp[r16 * CH_LDX + ((4 * ks + g < D) ? 4 * ks + g : 16)];  // (column 16: a zero)
            h64t_forward_half<KS1>(FR, xb, KS1, half, t2);
#ifdef TMA_ROLL_TICKS
            asm volatile("" : "+v"(t2[0]), "+v"(t2[1]));
            if (slot == xq) TMA_RTICK(5);
#endif
            if (half == 1) {
                *reinterpret_cast<f32x4 *>(slot + lane * 4) = t2[0];
                *reinterpret_cast<f32x4 *>(slot + 256 + lane * 4) = t2[1];
            }
        }
        __syncthreads();
#ifdef TMA_ROLL_TICKS
        if (slot == xq) TMA_RTICK(6);
#endif
        if (mine && half == 0) {
            f32x4 tb[2];
            tb[0] = *reinterpret_cast<const f32x4 *>(slot + lane * 4), tb[1] = *reinterpret_cast<const f32x4 *>(slot + 256 + lane * 4);
            h64t_head_r<KS1>(FR, t2, tb, o0, o1);
        }
    };
    // wave 3 (idle from the hand-over on): log-probability and rollout-buffer rows of step tt, left in LDS at parity `par` by wave 0 a step
    // earlier -- h64t_logp is h64t_act_n's arithmetic on the same lane layout; the next observation is the tile the step after reads
    auto flush_rows = [&](int tt, int par) {
        const int a_ = acts[par * 16 + r16];
        const float lp_ = h64t_logp(*reinterpret_cast<const f32x4 *>(lgt + par * 256 + lane * 4), A, a_, lane);
        if (active) {
            const int64_t off = (int64_t)tt * N + i;
            const int fl = ofl[par * 16 + r16];
            b.actions[off] = a_;
            b.log_probs[off] = lp_;
            b.terminated[off] = (uint8_t)(fl & 1);
            b.truncated[off] = (uint8_t)(fl >> 1);
            if (!(fl >> 1)) b.rewards[off] = __int_as_float(orw[par * 16 + r16]);
            float o[D];
            const float *Xr = X0 + (par ^ 1) * 16 * CH_LDX + r16 * CH_LDX;
#pragma unroll
            for (int c = 0; c < D; c++) o[c] = Xr[c];
            store_obs<D>(b.obs + ((int64_t)(tt + 1) * N + i) * D, o);
        }
    };
#ifdef TMA_ROLL_TICKS
    rt_last = __builtin_amdgcn_s_memtime();
#endif
    for (int k = 0; k < n_steps; k++) {
        const int t = t0 + k, p = k & 1, q = p ^ 1;
        f32x4 o0, o1;
        TMA_RTICK(0);
        forward(X0 + p * 16 * CH_LDX, xq, true, o0, o1);
#ifdef TMA_ROLL_TICKS
        asm volatile("" : "+v"(o0), "+v"(o1));
#endif
        TMA_RTICK(1);
        if (wave == 2 && active) b.values[(int64_t)t * N + i] = o0[0] + o1[0];
        const bool boot = k > 0 && flag[q];  // (block-uniform: written before the previous step's last barrier)
        if (wave == 0) {
            int act;
            float lp;
            // only the action is on this wave's chain: the logits go to wave 3, which forms the log-probability and writes the step's rows
            const f32x4 xs = o0 + o1;
            *reinterpret_cast<f32x4 *>(lgt + p * 256 + lane * 4) = xs;
            act = h64t_argmax(xs, A, *reinterpret_cast<const f32x4 *>(gnz + p * 256 + lane * 4), lane);
            lp = 0.0f;
#ifdef TMA_ROLL_TICKS
            asm volatile("" : "+v"(act));
#endif
            TMA_RTICK(2);
            bool tr_flag = false;
            if (active) {
                float rew32;
                bool te_flag = false;
                chunk_env_step<T, false>(v, b, s, er, ce, N, i, t, act, lp, X0 + q * 16 * CH_LDX + r16 * CH_LDX, XT0 + p * 16 * CH_LDX + r16 * CH_LDX, rew32,
                                         tr_flag, false, sret, slen, scnt, &te_flag);  // truncated rows: the value net writes reward + bootstrap after the barrier
                if (tr_flag) rw[p * 16 + r16] = rew32;
                trf[p * 16 + r16] = tr_flag ? 1 : 0;
                acts[p * 16 + r16] = act, orw[p * 16 + r16] = __float_as_int(rew32), ofl[p * 16 + r16] = (te_flag ? 1 : 0) | (tr_flag ? 2 : 0);
            }
            const bool any = __ballot(tr_flag) != 0ull;
            if (lane == 0) flag[p] = any ? 1 : 0;
            TMA_RTICK(3);
        }
        // the next step's noise on the policy net's second wave, idle from the hand-over to the end of the step (read behind the step's last barrier)
        if (wave == 1) *reinterpret_cast<f32x4 *>(gnz + q * 256 + lane * 4) = h64t_gumbel(rng_seed, genv, rng_step0 + (uint32_t)(t + 1), det, lane);
        if (wave == 3 && k > 0) flush_rows(t - 1, q);
        if (boot) {  // timeout bootstrap of step t - 1 on the value net's waves (the policy waves only take part in the barrier)
            f32x4 b0, b1;
            forward(XT0 + q * 16 * CH_LDX, xq2, net == 1, b0, b1);
            if (wave == 2 && active && trf[q * 16 + r16]) {
                const float gv = gamma * (b0[0] + b1[0]);
                b.rewards[(int64_t)(t - 1) * N + i] = rw[q * 16 + r16] + gv;
            }
        }
        __syncthreads();
        TMA_RTICK(4);
    }
    {  // bootstrap of the chunk's last step
        const int q = (n_steps - 1) & 1, t = t0 + n_steps - 1;
        if (wave == 3 && n_steps > 0) flush_rows(t, q);  // (the rows of the last step)
        if (n_steps > 0 && flag[q]) {
            f32x4 b0, b1;
            forward(XT0 + q * 16 * CH_LDX, xq2, net == 1, b0, b1);
            if (wave == 2 && active && trf[q * 16 + r16]) {
                const float gv = gamma * (b0[0] + b1[0]);
                b.rewards[(int64_t)t * N + i] = rw[q * 16 + r16] + gv;
            }
        }
    }
    if (wave != 0) return;
    if (active) {
        T::pack(v.st, N, i, s);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sret += __shfl_down(sret, o, 64);
        slen += __shfl_down(slen, o, 64);
        scnt += __shfl_down(scnt, o, 64);
    }
    if (lane == 0 && scnt > 0.0) {
        double *slot = v.stats + (row0 >> 8) * 3;
        atomicAdd(slot + 0, sret);
        atomicAdd(slot + 1, slen);
        atomicAdd(slot + 2, scnt);
    }
}

// ------------------------------------------------------------------------------------------
// Fused rollout chunk for the reference's default 256 x 256 policy on the bf16 MFMA (BASELINE configs[2] / [3] shapes: Ball3D, Push, ...
// with observations of up to 32 floats and a Discrete head): ONE launch advances every env by n_steps vector steps.
// A block of 8 waves owns a row group of 32 envs for the whole chunk.  Waves 0..3 carry the POLICY net, waves 4..7 the VALUE net, each
// wave 64 output columns of both hidden layers -- and every weight fragment a wave needs (4 of W1, 32 of W2, 8 of the head for the two
// head waves of a net: 176 registers) is loaded ONCE per launch and stays in registers, so a vector step touches global memory only
// to write its results (and to read a reset record when an episode ends).  Per step: layer 1 -> barrier -> layer 2 -> barrier ->
// {policy head, softmax, sample, env step of the 32 owner lanes, next observation into LDS | value head} -> barrier, and -- only when a
// row hit its time limit in this step (block-uniform flag) -- the value net once more on the terminal observations for the timeout bootstrap.
// The non-fused path pays two launches, the observation's HBM round trip and four dependent L2 weight-load latencies per step
// (13.5 us per vector step at 4096 envs); the forward arithmetic here is instruction for instruction that of policy_fwd_wide_kernel<BF>
// (same k order per accumulator, same split-K head order), so actions, log-probs, values and rewards are bit-identical to it.
// ------------------------------------------------------------------------------------------
template <int MROWS>
struct WideLds {
    static constexpr int M = MROWS, H = 256, NTW = 4, KS2 = H / 32, LDA = H + 16, LDX = 48;  // LDX: 32 observation columns + 16 (A-image stride rule)
    // observation + terminal-observation images, two activation images per net, W1 fragments [8 waves][4][64 lanes][8], head fragments
    // [2 nets][8][64][8] (only the layer-2 fragments -- 128 registers a wave -- stay in registers), bootstrap scratch
    static constexpr int bytes() { return (2 * M * LDX + 4 * M * LDA + 8 * NTW * 512 + 2 * KS2 * 512) * 2 + (32 + 32 + 4) * 4; }
};

// MROWS = 32 or 16 envs per block (round 4): 16 doubles the blocks -- 4096 envs fill all 256 CUs instead of 128, a 2048-env shard 128 instead
// of 64 -- and halves a wave's MFMAs and epilogue elements per step; 32 keeps two row tiles per weight fragment for larger vectors.
template <class T, int MROWS>
__global__ __launch_bounds__(512, 2) void rollout_chunk_wide_bf_kernel(EnvView v, const float *__restrict__ params, PLayout L, ChunkPtrs b, int t0, int n_steps,
                                                                       uint32_t rng_seed, uint32_t rng_step0, float gamma, int det) {
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    using WL = WideLds<MROWS>;
    constexpr int M = WL::M, MT2 = M / 16, NTW = WL::NTW, KS2 = WL::KS2, lda = WL::LDA, ldx = WL::LDX, D = T::OBS;
    static_assert(MROWS == 32 || MROWS == 16, "one or two 16-row tiles per block");
    static_assert(D <= 32 && T::NACT > 0, "fused wide rollout: observations of up to 32 floats, Discrete actions");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const bool is_pi = wave < 4;
    const int w4 = wave & 3, n_base = w4 * 16 * NTW, mt = MT2 == 2 ? (w4 & 1) : 0;  // mt: the row tile the head waves of a net (w4 < MT2) finish
    const int A = L.A;
    bf16_t *Xa = reinterpret_cast<bf16_t *>(smem_w), *XTa = Xa + M * ldx;
    bf16_t *A1 = XTa + M * ldx + (is_pi ? 0 : 2 * M * lda), *A2 = A1 + M * lda;  // per-net activation images
    bf16_t *W1l = XTa + M * ldx + 4 * M * lda + wave * NTW * 512, *W3l = XTa + M * ldx + 4 * M * lda + 8 * NTW * 512 + (is_pi ? 0 : KS2 * 512);
    float *rw = reinterpret_cast<float *>(XTa + M * ldx + 4 * M * lda + 8 * NTW * 512 + 2 * KS2 * 512);  // [32] reward of a truncated row before the bootstrap
    int *trf = reinterpret_cast<int *>(rw + 32), *flag = trf + 32;               // [32] row truncated in this step; [1] any of them
    const int64_t N = v.N;
    const int64_t row0 = (int64_t)blockIdx.x * M;
    // ---- this wave's weights: registers for the whole launch ----
    const Net Q = is_pi ? pi_net(params, L) : vf_net(params, L);
    const BfNetPtr W = bf_net_ptr(params, L, is_pi);
    bf16x8 w2[NTW][KS2];
    float b1v[NTW], b2v[NTW];
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        // layer-1 fragments (one k-step of 32, zero rows beyond D) and the head fragments wait in LDS: a fragment is the wave's own 1 KiB
        // (16 bytes per lane, conflict-free), read back right where the MFMA needs it
        *reinterpret_cast<bf16x8 *>(W1l + (j * 64 + lane) * 8) = bf_frag(W.fW1, w4 * NTW + j, lane);
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) w2[j][ks] = bf_frag(W.fW2, (w4 * NTW + j) * KS2 + ks, lane);
        b1v[j] = Q.b1[n_base + 16 * j + r16];
        b2v[j] = Q.b2[n_base + 16 * j + r16];
    }
    if (w4 == 0) {
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) *reinterpret_cast<bf16x8 *>(W3l + (ks * 64 + lane) * 8) = bf_frag(W.fW3, ks, lane);
    }
    const int n_out = is_pi ? A : 1;
    const float b3v = r16 < n_out ? Q.b3[r16] : 0.0f;
    // ---- env state of the 32 owner lanes (policy head waves 0 / 1, lanes r16 < 4: row = 16 mt + 4 g + r16) ----
    const int my_row = mt * 16 + g * 4 + r16;
    const int64_t i = row0 + my_row;
    const bool owner = wave < MT2 && r16 < 4 && i < N;
    typename T::S s;
    double er = 0.0;
    uint32_t ce = 0;
    if (owner) {
        T::unpack(v.st, N, i, s);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    for (int e = threadIdx.x; e < M * ldx; e += blockDim.x) {  // observation image of step t0 (columns >= D stay zero for the whole launch)
        const int row = e / ldx, c = e - row * ldx;
        Xa[e] = (bf16_t)((row0 + row < N && c < D) ? b.obs[((int64_t)t0 * N + row0 + row) * D + c] : 0.0f);
        XTa[e] = (bf16_t)0.0f;
    }
    if (threadIdx.x == 0) flag[0] = 0;
    __syncthreads();
    // X -> A1 -> A2 for this wave's 64 columns and both row tiles: the arithmetic of bf_hidden_layer with the fragments already in registers
    auto hidden = [&](const bf16_t *X) {
        // (one row tile at a time: 16 accumulator registers live instead of 32 -- the 128 weight registers leave little room at 256 per wave;
        //  the MFMA order per accumulator is unchanged)
#pragma unroll
        for (int m2 = 0; m2 < MT2; m2++) {
            f32x4 acc[NTW];
#pragma unroll
            for (int j = 0; j < NTW; j++) acc[j] = f32x4{b1v[j], b1v[j], b1v[j], b1v[j]};
            const bf16x8 a = a_frag(X, ldx, 16 * m2 + r16, 0, g);
#pragma unroll
            for (int j = 0; j < NTW; j++) acc[j] = mfma_bf(a, *reinterpret_cast<const bf16x8 *>(W1l + (j * 64 + lane) * 8), acc[j]);
#pragma unroll
            for (int j = 0; j < NTW; j++)
                bfq_store_rows(A1 + (16 * m2 + 4 * g) * lda + n_base + 16 * j + r16, lda, tanh_quad_s(acc[j]));
        }
        __syncthreads();
#pragma unroll
        for (int m2 = 0; m2 < MT2; m2++) {
            f32x4 acc[NTW];
#pragma unroll
            for (int j = 0; j < NTW; j++) acc[j] = f32x4{b2v[j], b2v[j], b2v[j], b2v[j]};
            // (round 6: the A fragments two k-steps ahead behind scheduling fences -- left to the scheduler every ds_read_b128 sat in front of
            //  its four MFMAs behind a full wait: an LDS round trip per k-step with the matrix pipe idle)
            bf16x8 af[3];
            af[0] = a_frag(A1, lda, 16 * m2 + r16, 0, g), af[1] = a_frag(A1, lda, 16 * m2 + r16, 1, g);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                if (ks + 2 < KS2) af[(ks + 2) % 3] = a_frag(A1, lda, 16 * m2 + r16, ks + 2, g);
#pragma unroll
                for (int j = 0; j < NTW; j++) acc[j] = mfma_bf(af[ks % 3], w2[j][ks], acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
                bfq_store_rows(A2 + (16 * m2 + 4 * g) * lda + n_base + 16 * j + r16, lda, tanh_quad_s(acc[j]));
        }
        __syncthreads();
    };
    // head of row tile mt in the summation order of bf_head (four partial sums over k-steps 2w', 2w' + 1, added to the bias in order)
    auto head = [&]() -> f32x4 {
        f32x4 part[4];
#pragma unroll
        for (int wq = 0; wq < 4; wq++) part[wq] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // (all sixteen operands requested before the first product: eight dependent read-wait-multiply round trips otherwise)
#pragma unroll
        for (int i2 = 0; i2 < 2; i2++) {  // (the four k-steps of a pass in one batch of reads: 32 registers)
            bf16x8 ha[4], hw[4];
#pragma unroll
            for (int wq = 0; wq < 4; wq++) {
                const int ks = wq * 2 + i2;
                ha[wq] = a_frag(A2, lda, 16 * mt + r16, ks, g), hw[wq] = *reinterpret_cast<const bf16x8 *>(W3l + (ks * 64 + lane) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int wq = 0; wq < 4; wq++) part[wq] = mfma_bf(ha[wq], hw[wq], part[wq]);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 out = f32x4{b3v, b3v, b3v, b3v};
#pragma unroll
        for (int wq = 0; wq < 4; wq++) out += part[wq];
        return out;
    };
    double sret = 0.0, slen = 0.0, scnt = 0.0;
    for (int k = 0; k < n_steps; k++) {
        const int t = t0 + k;
        hidden(Xa);  // (two barriers inside: every wave of the block takes part)
        if (w4 < MT2) {
            const f32x4 acc = head();
            if (!is_pi) {
                if (r16 == 0)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int64_t row = row0 + mt * 16 + g * 4 + r;
                        if (row < N) b.values[(int64_t)t * N + row] = acc[r];
                    }
            } else {
                int my_act = 0;
                float my_lp = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t row = row0 + mt * 16 + g * 4 + r;
                    const bool colok = r16 < A;
                    const float x = colok ? acc[r] : -INFINITY;
                    const float m = gmax16(x);
                    const float e = colok ? expf(x - m) : 0.0f;
                    const float sm = gsum16(e);
                    const float lse = m + logf(sm);
                    const float lp = x - lse;
                    int act;
                    if (det) {  // deterministic evaluation: first maximal logit, as policy_fwd_wide_kernel
                        act = (int)gmin16((colok && x == m) ? (float)r16 : 99.0f);
                    } else {
                        const float c = gscan16(e / sm);
                        const float u = uniform01(mix32(rng_seed, v.env_offset + (uint32_t)row, rng_step0 + (uint32_t)t));
                        const float cnt = gsum16((colok && c <= u) ? 1.0f : 0.0f);
                        act = min((int)cnt, A - 1);
                    }
                    const float lpa = gsum16((r16 == act) ? lp : 0.0f);
                    if (r16 == r) my_act = act, my_lp = lpa;
                }
                bool tr_flag = false;
                if (owner) {
                    const int64_t off = (int64_t)t * N + i;
                    b.actions[off] = my_act;
                    b.log_probs[off] = my_lp;
                    double r;
                    bool done;
                    T::step(s, my_act, nullptr, r, done);
                    const int steps = T::steps(s);
                    bool te, tr;
                    if constexpr (T::NATIVE_TRUNC_RULE) {  // backend/mlagents/envs.py:76
                        te = done;
                        tr = (steps >= T::MAXSTEPS) && !te;
                    } else {  // adapter rule, backend/mlagents/envs.py:139-145
                        const bool hit = steps >= T::MAXSTEPS;
                        te = done && !hit;
                        tr = hit;
                    }
                    er += r;
                    const float rew32 = (float)r;
                    b.terminated[off] = (uint8_t)te;
                    b.truncated[off] = (uint8_t)tr;
                    float o[D];
                    if (te || tr) {
                        if (tr) {
                            T::obs(s, o);
#pragma unroll
                            for (int c = 0; c < D; c++) XTa[my_row * ldx + c] = (bf16_t)o[c];
                            rw[my_row] = rew32;
                        }
                        sret += er, slen += (double)steps, scnt += 1.0;
                        log_episode(v, i, er, steps);
                        er = 0.0;
                        ce += 1;
                        if constexpr (T::USES_MT) {
                            uint32_t rec[T::RW > 0 ? T::RW : 1];
                            const uint32_t *slot = v.ring + ((int64_t)(ce % (uint32_t)v.D) * T::RW) * N + i;
#pragma unroll
                            for (int q = 0; q < T::RW; q++) rec[q] = slot[(int64_t)q * N];
                            T::from_rec(rec, s);
                        } else {
                            T::reset_inline(episode_seed(v.seed_base, v.env_offset + (uint32_t)i, ce), s);
                        }
                    }
                    T::obs(s, o);
                    float *dst = b.obs + ((int64_t)(t + 1) * N + i) * D;
#pragma unroll
                    for (int c = 0; c < D; c++) dst[c] = o[c];
#pragma unroll
                    for (int c = 0; c < D; c++) Xa[my_row * ldx + c] = (bf16_t)o[c];  // every layer-1 read of this step's image is behind hidden()'s barriers
                    if (!tr) b.rewards[off] = rew32;  // truncated rows: the value head waves write reward + bootstrap below
                    trf[my_row] = tr ? 1 : 0;
                    tr_flag = tr;
                }
                if (__ballot(tr_flag) != 0ull && lane == 0) atomicOr(flag, 1);
            }
        }
        __syncthreads();
        if (flag[0]) {  // (block-uniform) timeout bootstrap of this step: rewards = reward + gamma * V(terminal observation) where truncated
            if (!is_pi) {
                hidden(XTa);
                if (w4 < MT2) {
                    const f32x4 vt = head();
                    if (r16 == 0)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = mt * 16 + g * 4 + r;
                            if (row0 + row < N && trf[row]) {
                                const float gv = gamma * vt[r];
                                b.rewards[(int64_t)t * N + row0 + row] = rw[row] + gv;
                            }
                        }
                }
            } else {
                __syncthreads();  // the policy waves keep the value waves' two barriers company
                __syncthreads();
            }
            __syncthreads();
            if (threadIdx.x == 0) flag[0] = 0;
            if (threadIdx.x < 32) trf[threadIdx.x] = 0;
            __syncthreads();
        }
    }
    if (owner) {
        T::pack(v.st, N, i, s);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
    if (wave < MT2) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sret += __shfl_down(sret, o, 64);
            slen += __shfl_down(slen, o, 64);
            scnt += __shfl_down(scnt, o, 64);
        }
        if (lane == 0 && scnt > 0.0) {
            double *slot = v.stats + (row0 >> 8) * 3;
            atomicAdd(slot + 0, sret);
            atomicAdd(slot + 1, slen);
            atomicAdd(slot + 2, scnt);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same fused chunk for a Box action space and wide observations (the Crawler shape: 172 observations, 20 actions; BASELINE
// configs[4]): one launch advances every env by n_steps vector steps, a block of 8 waves owns 32 envs, waves 0..3 the policy net and 4..7 the
// value net, 64 hidden columns each.  The 32 layer-2 fragments of a wave stay in registers for the whole launch; the layer-1 fragments
// (6 k-steps x 4 column tiles: 24 KiB a wave, too many to keep) are streamed from the L2 every step, two k-steps at a time; head fragments, the sampled actions and the ENV STATE (69 words an env) live in LDS.  Per step: layer 1 -> barrier -> layer 2 ->
// barrier -> {mean head, Gaussian sample (policy_fwd_wide_kernel's streams and Box-Muller), log-prob, env step of the 32 owner lanes from
// LDS state, next observation to the buffer AND as bf16 into the LDS image | value head} -> barrier; timeout bootstrap as in the Discrete
// kernel.  Forward arithmetic = policy_fwd_wide_kernel<true, 0, 4, true> (same k order per accumulator, same split-K head order): the
// chunk is bit-identical to the per-step composition, which it replaces at 16 + 16 us per vector step (forward launch + env-step launch).
// ------------------------------------------------------------------------------------------
template <class T>
struct WideContLds {
    static constexpr int M = 32, H = 256, NTW = 4, KS2 = H / 32, LDA = H + 16, KP1 = (T::OBS + 31) & ~31, KS1 = KP1 / 32, LDX = KP1 + 16, NT3 = 2;
    // observation + terminal-observation images, two activation images per net, head fragments (policy NT3 x KS2, value KS2), sampled
    // actions [M][32] f32, env state [SW][M], bootstrap scratch
    // ... and the per-joint terms of the multi-lane env step [M][NJ][5] f32 + a done flag per row
    static constexpr int bytes() { return (2 * M * LDX + 4 * M * LDA + (NT3 + 1) * KS2 * 512) * 2 + (M * 32 + T::SW * M + 32 + 32 + 4 + M * T::NJ * 5 + 32) * 4; }
};

// one observation element to its global row and, as bf16, to its LDS image row (CrawlerTask::obs writes through operator[] / operator+)
struct ObsDual {
    float *gl;
    bf16_t *img;
    struct Ref {
        float *gl;
        bf16_t *img;
        __device__ __forceinline__ void operator=(float x) const {
            *gl = x;
            *img = (bf16_t)x;
        }
    };
    __device__ __forceinline__ Ref operator[](int k) const { return Ref{gl + k, img + k}; }
    __device__ __forceinline__ ObsDual operator+(int k) const { return ObsDual{gl + k, img + k}; }
};
struct ObsImage {  // bf16 image row only (terminal observations: only the bootstrap reads them)
    bf16_t *img;
    struct Ref {
        bf16_t *img;
        __device__ __forceinline__ void operator=(float x) const { *img = (bf16_t)x; }
    };
    __device__ __forceinline__ Ref operator[](int k) const { return Ref{img + k}; }
    __device__ __forceinline__ ObsImage operator+(int k) const { return ObsImage{img + k}; }
};

template <class T>
__global__ __launch_bounds__(512, 2) void rollout_chunk_wide_cont_kernel(EnvView v, const float *__restrict__ params, PLayout L, ChunkPtrs b, int t0,
                                                                         int n_steps, uint32_t rng_seed, uint32_t rng_step0, float gamma, int det) {
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    using W = WideContLds<T>;
    constexpr int M = W::M, NTW = W::NTW, KS2 = W::KS2, lda = W::LDA, ldx = W::LDX, KS1 = W::KS1, NT3 = W::NT3, D = T::OBS, AD = T::ADIM;
    static_assert(T::NACT == 0 && AD <= 32 && KS1 % 2 == 0 && !T::USES_MT, "fused wide rollout, Box actions: <= 32 action dims, inline resets");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const bool is_pi = wave < 4;
    const int w4 = wave & 3, n_base = w4 * 16 * NTW, mt = w4 & 1;  // mt: the row tile the two head waves of a net (w4 < 2) finish
    bf16_t *Xa = reinterpret_cast<bf16_t *>(smem_w), *XTa = Xa + M * ldx;
    bf16_t *A1 = XTa + M * ldx + (is_pi ? 0 : 2 * M * lda), *A2 = A1 + M * lda;  // per-net activation images
    bf16_t *W3pi = XTa + M * ldx + 4 * M * lda, *W3vf = W3pi + NT3 * KS2 * 512;
    const bf16_t *W3l = is_pi ? W3pi : W3vf;
    float *actl = reinterpret_cast<float *>(W3vf + KS2 * 512);         // [M][32] sampled actions of this step
    uint32_t *stl = reinterpret_cast<uint32_t *>(actl + M * 32);        // [SW][M] env state words
    float *rw = reinterpret_cast<float *>(stl + T::SW * M);             // [32] reward of a truncated row before the bootstrap
    int *trf = reinterpret_cast<int *>(rw + 32), *flag = trf + 32;      // [32] row truncated in this step; [1] any of them
    float *termsl = reinterpret_cast<float *>(flag + 4);                // [M][NJ][5] per-joint terms of the multi-lane env step (ChainTask::step_lanes)
    int *dnf = reinterpret_cast<int *>(termsl + M * T::NJ * 5);         // [32] row finished an episode in this step
    const int64_t N = v.N;
    const int64_t row0 = (int64_t)blockIdx.x * M;
    float *act_out = reinterpret_cast<float *>(b.actions);
    // ---- this wave's weights ----
    const Net Q = is_pi ? pi_net(params, L) : vf_net(params, L);
    const BfNetPtr Wn = bf_net_ptr(params, L, is_pi);
    bf16x8 w2[NTW][KS2];
    float b1v[NTW], b2v[NTW];
#pragma unroll
    for (int j = 0; j < NTW; j++) {
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) w2[j][ks] = bf_frag(Wn.fW2, (w4 * NTW + j) * KS2 + ks, lane);
        b1v[j] = Q.b1[n_base + 16 * j + r16];
        b2v[j] = Q.b2[n_base + 16 * j + r16];
    }
    if (w4 == 0) {  // head fragments of this net -> LDS (fragment q * KS2 + ks, 1 KiB each)
        const int nfr = (is_pi ? NT3 : 1) * KS2;
        for (int f = 0; f < nfr; f++) *reinterpret_cast<bf16x8 *>((is_pi ? W3pi : W3vf) + (f * 64 + lane) * 8) = bf_frag(Wn.fW3, f, lane);
    }
    const int n_out = is_pi ? L.A : 1;
    float b3v[NT3];
#pragma unroll
    for (int q = 0; q < NT3; q++) b3v[q] = (16 * q + r16 < n_out) ? Q.b3[16 * q + r16] : 0.0f;
    // ---- env state -> LDS.  The env step runs on EIGHT lanes per env (round 4: one owner lane ran twenty joints and 172 observation stores in
    // sequence while the other seven waves waited -- and spilled 153 registers doing it): the four policy waves take eight envs each, lane
    // `sub` of an env's group its joints sub, sub + 8, sub + 16 (T::step_lanes / T::obs_lanes, bit-identical to T::step / T::obs); lane 0 of a
    // group is the env's owner (running return, episode counter, flags, reset). ----
    const int my_row = (wave & 3) * 8 + (lane >> 3), sub = lane & 7;
    const int64_t i = row0 + my_row;
    const bool grp_ok = wave < 4 && i < N;
    const bool owner = grp_ok && sub == 0;
    double er = 0.0;
    uint32_t ce = 0;
    // (the state struct lives in LDS for the whole launch and T::step / T::obs work on it in place: next to the 128 weight registers a wave
    //  has no room for 69 state words plus the step's temporaries)
    static_assert(sizeof(typename T::S) <= T::SW * 4, "state struct fits its LDS slot");
    typename T::S *sl = reinterpret_cast<typename T::S *>(stl);
    if (owner) {
        T::unpack(v.st, N, i, sl[my_row]);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    for (int e = threadIdx.x; e < M * ldx; e += blockDim.x) {  // observation image of step t0 (columns >= D stay zero for the whole launch)
        const int row = e / ldx, c = e - row * ldx;
        Xa[e] = (bf16_t)((row0 + row < N && c < D) ? b.obs[((int64_t)t0 * N + row0 + row) * D + c] : 0.0f);
        XTa[e] = (bf16_t)0.0f;
    }
    if (threadIdx.x == 0) flag[0] = 0;
    if (threadIdx.x < 32) trf[threadIdx.x] = 0;
    __syncthreads();
    // X -> A1 -> A2 for this wave's 64 columns and both row tiles: bf_hidden_layer's loop order (k-step outer, row tile, column tile) with
    // the layer-1 fragments streamed two k-steps per batch
    auto hidden = [&](const bf16_t *X) {
        {
            // (the tile index goes through an opaque scalar copy: otherwise the 24 loop-invariant 64-bit fragment addresses are hoisted out of the
            //  step loop, do not fit next to the weights and come back as scratch reloads in front of every load)
            const int w4l = launder_uniform(w4);
            f32x4 acc[NTW][2];
#pragma unroll
            for (int j = 0; j < NTW; j++) acc[j][0] = acc[j][1] = f32x4{b1v[j], b1v[j], b1v[j], b1v[j]};
#pragma unroll
            for (int bt = 0; bt < KS1 / 2; bt++) {  // (one batch of eight fragments in registers at a time: 32 of the 128 a wave has left)
                bf16x8 wb[2][NTW];
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
#pragma unroll
                    for (int j = 0; j < NTW; j++) wb[kk][j] = bf_frag(Wn.fW1, (w4l * NTW + j) * KS1 + 2 * bt + kk, lane);
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
#pragma unroll
                    for (int m2 = 0; m2 < 2; m2++) {
                        const bf16x8 a = a_frag(X, ldx, 16 * m2 + r16, 2 * bt + kk, g);
#pragma unroll
                        for (int j = 0; j < NTW; j++) acc[j][m2] = mfma_bf(a, wb[kk][j], acc[j][m2]);
                    }
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int m2 = 0; m2 < 2; m2++)
                    bfq_store_rows(A1 + (16 * m2 + 4 * g) * lda + n_base + 16 * j + r16, lda, tanh_quad_s(acc[j][m2]));
        }
        __syncthreads();
#pragma unroll
        for (int m2 = 0; m2 < 2; m2++) {  // (one row tile at a time: the MFMA order per accumulator is policy_fwd_wide_kernel's)
            f32x4 acc[NTW];
#pragma unroll
            for (int j = 0; j < NTW; j++) acc[j] = f32x4{b2v[j], b2v[j], b2v[j], b2v[j]};
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                const bf16x8 a = a_frag(A1, lda, 16 * m2 + r16, ks, g);
#pragma unroll
                for (int j = 0; j < NTW; j++) acc[j] = mfma_bf(a, w2[j][ks], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
                bfq_store_rows(A2 + (16 * m2 + 4 * g) * lda + n_base + 16 * j + r16, lda, tanh_quad_s(acc[j]));
        }
        __syncthreads();
    };
    // head of row tile mt, NQ column tiles, in the summation order of bf_head (four partial sums over k-steps 2w', 2w' + 1, added to the bias)
    auto head = [&](f32x4 (&out)[NT3], int nq) {
        f32x4 part[4][NT3];
#pragma unroll
        for (int wq = 0; wq < 4; wq++)
#pragma unroll
            for (int q = 0; q < NT3; q++) part[wq][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i2 = 0; i2 < 2; i2++)
#pragma unroll
            for (int wq = 0; wq < 4; wq++) {
                const int ks = wq * 2 + i2;
                const bf16x8 a = a_frag(A2, lda, 16 * mt + r16, ks, g);
#pragma unroll
                for (int q = 0; q < NT3; q++)
                    if (q < nq) part[wq][q] = mfma_bf(a, *reinterpret_cast<const bf16x8 *>(W3l + ((q * KS2 + ks) * 64 + lane) * 8), part[wq][q]);
            }
#pragma unroll
        for (int q = 0; q < NT3; q++) {
            out[q] = f32x4{b3v[q], b3v[q], b3v[q], b3v[q]};
#pragma unroll
            for (int wq = 0; wq < 4; wq++) out[q] += part[wq][q];
        }
    };
    const float *ls = params + L.log_std;
    float lsd_v[2], sd_v[2];  // log_std and exp(log_std) of this lane's two action columns: constant over the launch
#pragma unroll
    for (int j = 0; j < 2; j++) {
        lsd_v[j] = (is_pi && 16 * j + r16 < AD) ? ls[16 * j + r16] : 0.0f;
        sd_v[j] = expf(lsd_v[j]);
    }
    double sret = 0.0, slen = 0.0, scnt = 0.0;
    for (int k = 0; k < n_steps; k++) {
        const int t = t0 + k;
        hidden(Xa);  // (two barriers inside: every wave of the block takes part)
        if (w4 < 2) {
            f32x4 acc[NT3];
            head(acc, is_pi ? NT3 : 1);
            if (!is_pi) {
                if (r16 == 0)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int64_t row = row0 + mt * 16 + g * 4 + r;
                        if (row < N) b.values[(int64_t)t * N + row] = acc[0][r];
                    }
            } else {
                // DiagGaussian sample + log-prob: policy_fwd_wide_kernel<CONT>'s streams and arithmetic
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int lrow = mt * 16 + g * 4 + r;
                    const int64_t row = row0 + lrow;
                    const uint32_t gi = v.env_offset + (uint32_t)row;
                    const uint32_t rstep = rng_step0 + (uint32_t)t;
                    float lpsum = 0.0f;
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int col = 16 * j + r16;
                        if (col < AD) {
                            const float mu = acc[j][r], lsd = lsd_v[j], sd = sd_v[j];
                            const float u1 = fmaxf(uniform01(mix32(rng_seed ^ (0x68E31DA4u + (uint32_t)col * 0x9E3779B9u), gi, rstep)), 5.9604645e-08f);
                            const float u2 = uniform01(mix32(rng_seed ^ (0xB5297A4Du + (uint32_t)col * 0x85EBCA77u), gi, rstep));
                            const float z = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
                            const float a = det ? mu : mu + sd * z;  // deterministic evaluation: the mean (policy_fwd_wide_kernel)
                            const float d = a - mu;
                            lpsum += -(d * d) / (2.0f * (sd * sd)) - lsd - 0.9189385332046727f;
                            actl[lrow * 32 + col] = a;
                            if (row < N) act_out[((int64_t)t * N + row) * AD + col] = a;
                        }
                    }
                    lpsum = gsum16(lpsum);
                    if (r16 == r && row < N) b.log_probs[(int64_t)t * N + row] = lpsum;
                }
            }
        }
        __syncthreads();  // the sampled actions of both row tiles (head waves 0 / 1) are in LDS
        if (wave < 4) {
            typename T::S &s = sl[my_row];
            double r = 0.0;
            bool done = false, tr_flag = false;
            if (grp_ok) T::template step_lanes<8>(s, actl + my_row * 32, termsl + my_row * (T::NJ * 5), sub, r, done);
            if (owner) {
                const int64_t off = (int64_t)t * N + i;
                const bool hit = T::steps(s) >= T::MAXSTEPS;  // adapter rule, backend/mlagents/envs.py:139-145
                const bool te = done && !hit, tr = hit;
                er += r;
                const float rew32 = (float)r;
                b.terminated[off] = (uint8_t)te;
                b.truncated[off] = (uint8_t)tr;
                if (tr) rw[my_row] = rew32;        // truncated rows: the value head waves write reward + bootstrap below
                else b.rewards[off] = rew32;
                trf[my_row] = tr ? 1 : 0;
                dnf[my_row] = (te || tr) ? 1 : 0;
                tr_flag = tr;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool g_tr = grp_ok && trf[my_row] != 0, g_dn = grp_ok && dnf[my_row] != 0;
            if (g_tr) T::template obs_lanes<8>(s, sub, ObsImage{XTa + my_row * ldx});  // terminal observation (only the bootstrap reads it)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (g_dn && sub == 0) {
                const int steps = T::steps(s);
                sret += er, slen += (double)steps, scnt += 1.0;
                log_episode(v, i, er, steps);
                er = 0.0;
                ce += 1;
                T::reset_inline(episode_seed(v.seed_base, v.env_offset + (uint32_t)i, ce), s);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // next observation: buffer slot t + 1 and, as bf16, this row of the LDS image (every layer-1 read of this step's image is behind
            // hidden()'s barriers); 8 lanes x <= 3 items of 8-12 floats
            if (grp_ok) T::template obs_lanes<8>(s, sub, ObsDual{b.obs + ((int64_t)(t + 1) * N + i) * D, Xa + my_row * ldx});
            if (__ballot(tr_flag) != 0ull && lane == 0) atomicOr(flag, 1);
        }
        __syncthreads();
        if (flag[0]) {  // (block-uniform) timeout bootstrap of this step: rewards = reward + gamma * V(terminal observation) where truncated
            if (!is_pi) {
                hidden(XTa);
                if (w4 < 2) {
                    f32x4 vt[NT3];
                    head(vt, 1);
                    if (r16 == 0)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = mt * 16 + g * 4 + r;
                            if (row0 + row < N && trf[row]) {
                                const float gv = gamma * vt[0][r];
                                b.rewards[(int64_t)t * N + row0 + row] = rw[row] + gv;
                            }
                        }
                }
            } else {
                __syncthreads();  // the policy waves keep the value waves' two barriers company
                __syncthreads();
            }
            __syncthreads();
            if (threadIdx.x == 0) flag[0] = 0;
            if (threadIdx.x < 32) trf[threadIdx.x] = 0;
            __syncthreads();
        }
    }
    if (owner) {
        T::pack(v.st, N, i, sl[my_row]);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
    if (wave < 4) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sret += __shfl_down(sret, o, 64);
            slen += __shfl_down(slen, o, 64);
            scnt += __shfl_down(scnt, o, 64);
        }
        if (lane == 0 && scnt > 0.0) {
            double *slot = v.stats + (row0 >> 8) * 3;
            atomicAdd(slot + 0, sret);
            atomicAdd(slot + 1, slen);
            atomicAdd(slot + 2, scnt);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Round 4: the Box-action chunk with the POLICY net only (the design of rollout_chunk_wide_f32_kernel below, applied to the Crawler / Ant
// shapes).  Nothing in a vector step needs the value net, so it leaves the step loop: values of the chunk's n x N observation rows are ONE
// tma_policy_values launch and the timeout bootstrap of its terminal-observation slots ONE tma_policy_bootstrap launch per chunk (rows are
// independent in both kernels: the bits of one launch per step).  With one net per block every wave owns 32 columns instead of 64, and BOTH
// weight slices fit its registers for the whole launch -- layer 1: KS1 x 2 fragments (48 registers at the 172-wide Crawler input), layer 2:
// 8 x 2 (64) -- so a vector step reads no weight from memory at all (the two-net kernel re-streamed 48 KB of layer-1 fragments per net from L2
// in three dependent round trips per step and spilled around the env step).  Per step: layer 1 -> barrier -> layer 2 -> barrier -> {mean
// head, Gaussian sample, log-prob: waves 0 / 1} -> barrier -> env step on eight lanes per env (waves 0-3) -> barrier.
// Forward arithmetic = policy_fwd_wide_kernel<true, 0, 4, true> per accumulator (same k order, same split-K head order): bit-identical to the
// per-step composition and to the two-net chunk (TMA_CONT_TWO_NET=1 selects that one: the A/B switch).
// ------------------------------------------------------------------------------------------
template <class T, int MROWS>
struct WideContPiLds {
    static constexpr int M = MROWS, H = 256, KS2 = H / 32, LDA = H + 16, KP1 = (T::OBS + 31) & ~31, KS1 = KP1 / 32, LDX = KP1 + 16, NT3 = 2;
    // observation image, two activation images, head fragments (NT3 x KS2), sampled actions [M][32] f32, env state [SW][M], per-joint terms
    static constexpr int bytes() { return (M * LDX + 2 * M * LDA + NT3 * KS2 * 512) * 2 + (M * 32 + T::SW * M + 32 + 32 + M * T::NJ * 5) * 4; }
};

// MROWS = 32 or 16 envs per block: 16 doubles the blocks (2048 envs: 128 instead of 64 of the 256 CUs) and halves a wave's MFMAs and
// epilogue elements per step -- the shape the Crawler shard wants; 32 keeps two row tiles per weight fragment for larger vectors.
template <class T, int MROWS>
__global__ __launch_bounds__(512, 2) void rollout_chunk_wide_cont_pi_kernel(EnvView v, const float *__restrict__ params, PLayout L, ChunkPtrs b,
                                                                            float *__restrict__ term_obs, int t0, int n_steps, uint32_t rng_seed,
                                                                            uint32_t rng_step0, int det) {
    extern __shared__ __attribute__((aligned(16))) char smem_wp[];
    using W = WideContPiLds<T, MROWS>;
    constexpr int M = W::M, MT2 = M / 16, NTW = 2, KS2 = W::KS2, lda = W::LDA, ldx = W::LDX, KS1 = W::KS1, NT3 = W::NT3, D = T::OBS, AD = T::ADIM;
    static_assert(MROWS == 32 || MROWS == 16, "one or two 16-row tiles per block");
    constexpr bool AHEAD = KS1 <= 4;  // operand fragments requested ahead of their MFMAs (see layer 1): the Ant width yes, the Crawler width (KS1 = 6) no
    static_assert(T::NACT == 0 && AD <= 32 && !T::USES_MT, "fused wide rollout, Box actions: <= 32 action dims, inline resets");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int n_base = wave * 16 * NTW, mt = MT2 == 2 ? (wave & 1) : 0;  // mt: the row tile head wave 0 / 1 finishes
    bf16_t *Xa = reinterpret_cast<bf16_t *>(smem_wp), *A1 = Xa + M * ldx, *A2 = A1 + M * lda;
    bf16_t *W3l = A2 + M * lda;
    float *actl = reinterpret_cast<float *>(W3l + NT3 * KS2 * 512);    // [M][32] sampled actions of this step
    uint32_t *stl = reinterpret_cast<uint32_t *>(actl + M * 32);        // [SW][M] env state words
    int *trf = reinterpret_cast<int *>(stl + T::SW * M), *dnf = trf + 32;  // [32] row truncated / finished an episode in this step
    float *termsl = reinterpret_cast<float *>(dnf + 32);                // [M][NJ][5] per-joint terms of the multi-lane env step
    const int64_t N = v.N;
    const int64_t row0 = (int64_t)blockIdx.x * M;
    float *act_out = reinterpret_cast<float *>(b.actions);
    // ---- this wave's weights: both slices stay in registers for the whole launch ----
    const Net Q = pi_net(params, L);
    const BfNetPtr Wn = bf_net_ptr(params, L, true);
    bf16x8 w1[NTW][KS1], w2[NTW][KS2];
    float b1v[NTW], b2v[NTW];
#pragma unroll
    for (int j = 0; j < NTW; j++) {
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) w1[j][ks] = bf_frag(Wn.fW1, (wave * NTW + j) * KS1 + ks, lane);
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) w2[j][ks] = bf_frag(Wn.fW2, (wave * NTW + j) * KS2 + ks, lane);
        b1v[j] = Q.b1[n_base + 16 * j + r16];
        b2v[j] = Q.b2[n_base + 16 * j + r16];
    }
    if (wave == 0) {  // head fragments -> LDS (fragment q * KS2 + ks, 1 KiB each)
        for (int f = 0; f < NT3 * KS2; f++) *reinterpret_cast<bf16x8 *>(W3l + (f * 64 + lane) * 8) = bf_frag(Wn.fW3, f, lane);
    }
    const int n_out = L.A;
    float b3v[NT3];
#pragma unroll
    for (int q = 0; q < NT3; q++) b3v[q] = (16 * q + r16 < n_out) ? Q.b3[16 * q + r16] : 0.0f;
    // ---- env state -> LDS; eight lanes per env on waves 0-3 (see rollout_chunk_wide_cont_kernel) ----
    constexpr int EW = M / 8;  // waves that run the env step (eight envs each)
    const int my_row = (wave % EW) * 8 + (lane >> 3), sub = lane & 7;
    const int64_t i = row0 + my_row;
    const bool grp_ok = wave < EW && i < N;
    const bool owner = grp_ok && sub == 0;
    double er = 0.0;
    uint32_t ce = 0;
    static_assert(sizeof(typename T::S) <= T::SW * 4, "state struct fits its LDS slot");
    typename T::S *sl = reinterpret_cast<typename T::S *>(stl);
    if (owner) {
        T::unpack(v.st, N, i, sl[my_row]);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    for (int e = threadIdx.x; e < M * ldx; e += blockDim.x) {  // observation image of step t0 (columns >= D stay zero for the whole launch)
        const int row = e / ldx, c = e - row * ldx;
        Xa[e] = (bf16_t)((row0 + row < N && c < D) ? b.obs[((int64_t)t0 * N + row0 + row) * D + c] : 0.0f);
    }
    if (threadIdx.x < 32) trf[threadIdx.x] = 0, dnf[threadIdx.x] = 0;  // (slots beyond M stay unused)
    __syncthreads();
    const float *ls = params + L.log_std;
    float lsd_v[2], sd_v[2];  // log_std and exp(log_std) of this lane's two action columns: constant over the launch
#pragma unroll
    for (int j = 0; j < 2; j++) {
        lsd_v[j] = (16 * j + r16 < AD) ? ls[16 * j + r16] : 0.0f;
        sd_v[j] = expf(lsd_v[j]);
    }
    double sret = 0.0, slen = 0.0, scnt = 0.0;
    for (int k = 0; k < n_steps; k++) {
        const int t = t0 + k;
        {  // layer 1: bf_hidden_layer's order per accumulator (k-step ascending); every operand already in registers / LDS
            f32x4 acc[NTW][MT2];
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int m2 = 0; m2 < MT2; m2++) acc[j][m2] = f32x4{b1v[j], b1v[j], b1v[j], b1v[j]};
            // (round 6: A fragments two ahead behind scheduling fences -- an LDS round trip per fragment with the matrix pipe idle otherwise.  Only
            //  where the registers allow it: at the Crawler width the wave's 112 weight registers leave no room, and the ring spilled: 6.0 -> 7.5 us)
            if constexpr (AHEAD) {
                constexpr int NF1 = KS1 * MT2;
                bf16x8 af[3];
#pragma unroll
                for (int f = 0; f < 2 && f < NF1; f++) af[f] = a_frag(Xa, ldx, 16 * (f % MT2) + r16, f / MT2, g);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF1; f++) {
                    const int ks = f / MT2, m2 = f % MT2;
                    if (f + 2 < NF1) af[(f + 2) % 3] = a_frag(Xa, ldx, 16 * ((f + 2) % MT2) + r16, (f + 2) / MT2, g);
#pragma unroll
                    for (int j = 0; j < NTW; j++) acc[j][m2] = mfma_bf(af[f % 3], w1[j][ks], acc[j][m2]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS1; ks++)
#pragma unroll
                    for (int m2 = 0; m2 < MT2; m2++) {
                        const bf16x8 a = a_frag(Xa, ldx, 16 * m2 + r16, ks, g);
#pragma unroll
                        for (int j = 0; j < NTW; j++) acc[j][m2] = mfma_bf(a, w1[j][ks], acc[j][m2]);
                    }
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int m2 = 0; m2 < MT2; m2++)
                    bfq_store_rows(A1 + (16 * m2 + 4 * g) * lda + n_base + 16 * j + r16, lda, tanh_quad_s(acc[j][m2]));
        }
        __syncthreads();
        {
            f32x4 acc[NTW][MT2];
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int m2 = 0; m2 < MT2; m2++) acc[j][m2] = f32x4{b2v[j], b2v[j], b2v[j], b2v[j]};
            if constexpr (AHEAD) {
                constexpr int NF2 = KS2 * MT2;
                bf16x8 af[3];
#pragma unroll
                for (int f = 0; f < 2; f++) af[f] = a_frag(A1, lda, 16 * (f % MT2) + r16, f / MT2, g);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF2; f++) {
                    const int ks = f / MT2, m2 = f % MT2;
                    if (f + 2 < NF2) af[(f + 2) % 3] = a_frag(A1, lda, 16 * ((f + 2) % MT2) + r16, (f + 2) / MT2, g);
#pragma unroll
                    for (int j = 0; j < NTW; j++) acc[j][m2] = mfma_bf(af[f % 3], w2[j][ks], acc[j][m2]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS2; ks++)
#pragma unroll
                    for (int m2 = 0; m2 < MT2; m2++) {
                        const bf16x8 a = a_frag(A1, lda, 16 * m2 + r16, ks, g);
#pragma unroll
                        for (int j = 0; j < NTW; j++) acc[j][m2] = mfma_bf(a, w2[j][ks], acc[j][m2]);
                    }
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int m2 = 0; m2 < MT2; m2++)
                    bfq_store_rows(A2 + (16 * m2 + 4 * g) * lda + n_base + 16 * j + r16, lda, tanh_quad_s(acc[j][m2]));
        }
        __syncthreads();
        if (wave < MT2) {
            // mean head of row tile mt in the summation order of bf_head (four partial sums over k-steps 2w', 2w' + 1, added to the bias)
            f32x4 part[4][NT3], acc[NT3];
#pragma unroll
            for (int wq = 0; wq < 4; wq++)
#pragma unroll
                for (int q = 0; q < NT3; q++) part[wq][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (AHEAD) {
#pragma unroll
                for (int i2 = 0; i2 < 2; i2++) {  // (the four k-steps of a pass: their twelve reads in one batch)
                    bf16x8 ha[4], hw[4][NT3];
#pragma unroll
                    for (int wq = 0; wq < 4; wq++) {
                        const int ks = wq * 2 + i2;
                        ha[wq] = a_frag(A2, lda, 16 * mt + r16, ks, g);
#pragma unroll
                        for (int q = 0; q < NT3; q++) hw[wq][q] = *reinterpret_cast<const bf16x8 *>(W3l + ((q * KS2 + ks) * 64 + lane) * 8);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int wq = 0; wq < 4; wq++)
#pragma unroll
                        for (int q = 0; q < NT3; q++) part[wq][q] = mfma_bf(ha[wq], hw[wq][q], part[wq][q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i2 = 0; i2 < 2; i2++)
#pragma unroll
                    for (int wq = 0; wq < 4; wq++) {
                        const int ks = wq * 2 + i2;
                        const bf16x8 a = a_frag(A2, lda, 16 * mt + r16, ks, g);
#pragma unroll
                        for (int q = 0; q < NT3; q++) part[wq][q] = mfma_bf(a, *reinterpret_cast<const bf16x8 *>(W3l + ((q * KS2 + ks) * 64 + lane) * 8), part[wq][q]);
                    }
            }
#pragma unroll
            for (int q = 0; q < NT3; q++) {
                acc[q] = f32x4{b3v[q], b3v[q], b3v[q], b3v[q]};
#pragma unroll
                for (int wq = 0; wq < 4; wq++) acc[q] += part[wq][q];
            }
            // DiagGaussian sample + log-prob: policy_fwd_wide_kernel<CONT>'s streams and arithmetic
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int lrow = mt * 16 + g * 4 + r;
                const int64_t row = row0 + lrow;
                const uint32_t gi = v.env_offset + (uint32_t)row;
                const uint32_t rstep = rng_step0 + (uint32_t)t;
                float lpsum = 0.0f;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int col = 16 * j + r16;
                    if (col < AD) {
                        const float mu = acc[j][r], lsd = lsd_v[j], sd = sd_v[j];
                        const float u1 = fmaxf(uniform01(mix32(rng_seed ^ (0x68E31DA4u + (uint32_t)col * 0x9E3779B9u), gi, rstep)), 5.9604645e-08f);
                        const float u2 = uniform01(mix32(rng_seed ^ (0xB5297A4Du + (uint32_t)col * 0x85EBCA77u), gi, rstep));
                        const float z = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
                        const float a = det ? mu : mu + sd * z;  // deterministic evaluation: the mean (policy_fwd_wide_kernel)
                        const float d = a - mu;
                        lpsum += -(d * d) / (2.0f * (sd * sd)) - lsd - 0.9189385332046727f;
                        actl[lrow * 32 + col] = a;
                        if (row < N) act_out[((int64_t)t * N + row) * AD + col] = a;
                    }
                }
                lpsum = gsum16(lpsum);
                if (r16 == r && row < N) b.log_probs[(int64_t)t * N + row] = lpsum;
            }
        }
        __syncthreads();  // the sampled actions of every row tile are in LDS
        if (wave < EW) {
            typename T::S &s = sl[my_row];
            double r = 0.0;
            bool done = false;
            if (grp_ok) T::template step_lanes<8>(s, actl + my_row * 32, termsl + my_row * (T::NJ * 5), sub, r, done);
            if (owner) {
                const int64_t off = (int64_t)t * N + i;
                const bool hit = T::steps(s) >= T::MAXSTEPS;  // adapter rule, backend/mlagents/envs.py:139-145
                const bool te = done && !hit, tr = hit;
                er += r;
                b.rewards[off] = (float)r;  // (the timeout bootstrap is added by the caller's batched tma_policy_bootstrap)
                b.terminated[off] = (uint8_t)te;
                b.truncated[off] = (uint8_t)tr;
                trf[my_row] = tr ? 1 : 0;
                dnf[my_row] = (te || tr) ? 1 : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool g_tr = grp_ok && trf[my_row] != 0, g_dn = grp_ok && dnf[my_row] != 0;
            if (g_tr) T::template obs_lanes<8>(s, sub, term_obs + ((int64_t)k * N + i) * D);  // terminal observation: slot (t - t0) of the chunk
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (g_dn && sub == 0) {
                const int steps = T::steps(s);
                sret += er, slen += (double)steps, scnt += 1.0;
                log_episode(v, i, er, steps);
                er = 0.0;
                ce += 1;
                T::reset_inline(episode_seed(v.seed_base, v.env_offset + (uint32_t)i, ce), s);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (grp_ok) T::template obs_lanes<8>(s, sub, ObsDual{b.obs + ((int64_t)(t + 1) * N + i) * D, Xa + my_row * ldx});
        }
        __syncthreads();
    }
    if (owner) {
        T::pack(v.st, N, i, sl[my_row]);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
    if (wave < EW) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sret += __shfl_down(sret, o, 64);
            slen += __shfl_down(slen, o, 64);
            scnt += __shfl_down(scnt, o, 64);
        }
        if (lane == 0 && scnt > 0.0) {
            double *slot = v.stats + (row0 >> 8) * 3;
            atomicAdd(slot + 0, sret);
            atomicAdd(slot + 1, slen);
            atomicAdd(slot + 2, scnt);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused rollout chunk for the reference's OWN default policy and dtype -- MLP(256, 256), f32 (backend/mlagents/training.py:363-365) -- on
// the Discrete tasks with observations of up to 32 floats: ONE launch advances every env by n_steps vector steps.
//
// With f32 MFMA operands the layer-2 matrix of ONE net is 256 KB: it fits the registers of a 4-wave block (each wave keeps the 256 x 64
// slice of its columns, 256 registers, for the whole launch) but two nets do not fit one CU.  So this kernel runs the POLICY net only
// -- layer 1 -> barrier -> layer 2 -> barrier -> {head, softmax, sample, env step of the tile's 16 owner lanes, next observation into
// LDS and into buffer slot t + 1} -> barrier -- and writes raw rewards plus the terminal observations of the chunk's steps; nothing in
// a vector step needs the value net.  The caller then computes values[t0 .. t0 + n) with ONE batched tma_policy_values launch over the
// chunk's n x N observation rows and the timeout bootstrap with ONE tma_policy_bootstrap launch over the chunk's terminal-observation slots
// (rows are independent in both kernels: the same bits as one launch per step).  A block owns a tile of 16 envs: 4096 envs = 256 blocks,
// one per CU; the reference's own 8-env runs are a single block.
// Forward arithmetic = policy_fwd_wide_kernel<false, 0, 4, false> (same k order per accumulator, one sequential head chain), so actions
// and log-probabilities are bit-identical to the per-step composition, which pays two launches, the observation's HBM round trip and
// the whole weight matrix from L2 per vector step (28 us per step at 8 envs, 13 us at 4096).
// ------------------------------------------------------------------------------------------
template <class T, int MROWS>
__global__ __launch_bounds__(256, 1) void rollout_chunk_wide_f32_kernel(EnvView v, const float *__restrict__ params, PLayout L, ChunkPtrs b,
                                                                        float *__restrict__ term_obs, int t0, int n_steps, uint32_t rng_seed,
                                                                        uint32_t rng_step0, int det) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    // MROWS = 8 (round 6): tiles of EIGHT envs -- the reference's own 8-env runs, and every run of up to 2048 envs (256 blocks).  A 16 x 16 x 4
    // tile pads 8 envs to 16 rows: half the matrix pipe's work is zeros.  v_mfma_f32_4x4x1_16b_f32 with the A operand of one block broadcast to
    // all sixteen (CBSZ = 4, ABID = e) is 4 envs x 64 columns x 1 k in 8 cycles: a wave's 64 columns of both hidden layers are two such
    // chains (envs 0-3, envs 4-7), which issue back to back (8.3 cycles an instruction: tools/mfma_bcast_probe.hip) -- 512 x 8.3 = 4.2 k cycles for
    // layer 2 instead of 256 x 32 = 8.2 k.  Same bits: a k-step of the 16 x 16 x 4 instruction is its four products added to the accumulator
    // one after the other in k order, each a fused multiply-add (the same probe: 512 of 512 outputs identical, and identical to fmaf on the
    // host), which is exactly what the k-by-k chain does.  The head, the sampling and the env step read the same 16-row LDS images as before
    // (rows 8 .. 15 stay zero).
    constexpr bool M8 = MROWS == 8;
    constexpr int M = MROWS, H = 256, NTW = 4, D = T::OBS, ldx = ((D + 3) & ~3) + (M8 ? 4 : 2), ld = H + (M8 ? 4 : 2), KS1 = (D + 3) >> 2, KS2 = H / 4;
    static_assert(D <= (MROWS == 8 ? 64 : 32) && T::NACT > 0, "fused f32 wide rollout: observations of up to 32 floats (8-env tiles: 64), Discrete actions");
    static_assert(MROWS == 16 || MROWS == 8, "tiles of 16 or 8 envs");
    // M8 with up to 8 actions: the head as ONE fused-multiply-add chain per (env, action) on the vector ALU -- lane 8 a + e runs
    // acc = fma(h2[e][k], W3[k][a], acc) for k = 0 .. 255, the very operations (and order) of the 64 dependent 16 x 16 x 4 MFMAs, which
    // issue once per ~50 cycles where a dependent v_fma_f32 issues once per ~5.  The logits cross to the sampling code's layout through
    // LDS; there a lane group takes TWO rows instead of four (row 2 g + r: eight rows over four groups), which halves the serial
    // softmax / scan chains of a step.
    constexpr bool VH = M8 && T::NACT <= 8;
    constexpr int RG = VH ? 2 : 4;  // rows of the tile per 16-lane group in the sampling code
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int A = L.A, n_base = wave * 16 * NTW;
    float *X = smem_f, *h1 = X + M * ldx, *h2 = h1 + M * ld;  // (h2: 16 rows in either form -- the head's A operand)
    float *w3l = h2 + 16 * ld, *lgt = w3l + 8 * ld;            // VH: W3 as [action][k] rows (zero rows beyond A); logits [env][8]
    const int64_t N = v.N;
    const int64_t row0 = (int64_t)blockIdx.x * M;
    // ---- this wave's weights: registers for the whole launch (the operands policy_fwd_wide_kernel fetches per step) ----
    const Net P = pi_net(params, L);
    float w1v[M8 ? 1 : NTW][M8 ? 1 : KS1], w2v[M8 ? 1 : NTW][M8 ? 1 : KS2], b1v[NTW], b2v[NTW];
    float w1s[M8 ? D : 1], w2s[M8 ? H : 1];  // M8: column n_base + lane of W1t / W2t, one register per k
    if constexpr (M8) {
#pragma unroll
        for (int kk = 0; kk < D; kk++) w1s[kk] = P.W1t[(int64_t)kk * H + n_base + lane];
#pragma unroll
        for (int kk = 0; kk < H; kk++) w2s[kk] = P.W2t[(int64_t)kk * H + n_base + lane];
#pragma unroll
        for (int kk = 128; kk < H; kk++) asm volatile("" : "+a"(w2s[kk]));  // the upper half lives in accumulator registers: an MFMA reads its B operand from either file
        b1v[0] = P.b1[n_base + lane], b2v[0] = P.b2[n_base + lane];
        for (int e = threadIdx.x; e < 8 * ld; e += blockDim.x) h2[8 * ld + e] = 0.0f;  // rows 8 .. 15 of the head's operand
        if constexpr (VH) {
            for (int e = threadIdx.x; e < 8 * ld; e += blockDim.x) {
                const int a = e / ld, kk = e - a * ld;
                w3l[e] = (a < A && kk < H) ? P.W3t[(int64_t)kk * A + a] : 0.0f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < (M8 ? 0 : NTW); j++) {
        const int col = n_base + 16 * j + r16;
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) {
            const int k = 4 * ks + g;
            const float w = P.W1t[(int64_t)(k < D ? k : 0) * H + col];
            w1v[j][ks] = k < D ? w : 0.0f;
        }
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) w2v[j][ks] = P.W2t[(int64_t)(4 * ks + g) * H + col];
        if (j >= 2) {  // half of the layer-2 operands live in accumulator registers (an MFMA reads its B operand from either file)
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) asm volatile("" : "+a"(w2v[j][ks]));
        }
        b1v[j] = P.b1[col];
        b2v[j] = P.b2[col];
    }
    float w3v[VH ? 1 : KS2];  // head operands of wave 0 (dense_head<1>: column r16 of W3t, zero beyond the A outputs)
#pragma unroll
    for (int ks = 0; ks < (VH ? 0 : KS2); ks++) {
        const float w = P.W3t[(int64_t)(4 * ks + g) * A + (r16 < A ? r16 : 0)];
        w3v[ks] = r16 < A ? w : 0.0f;
    }
    const float b3v = r16 < A ? P.b3[r16] : 0.0f;
    // ---- env state of the 16 owner lanes (wave 0, lanes r16 < 4: row = 4 g + r16, where the head's C layout leaves that row's action) ----
    const int my_row = g * RG + r16;
    const int64_t i = row0 + my_row;
    const bool owner = wave == 0 && r16 < RG && my_row < M && i < N;
    const float b3h = VH ? ((lane >> 3) < A ? P.b3[lane >> 3] : 0.0f) : 0.0f;  // VH: lane 8 a + e
    typename T::S s;
    double er = 0.0;
    uint32_t ce = 0;
    if (owner) {
        T::unpack(v.st, N, i, s);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    for (int e = threadIdx.x; e < M * ldx; e += blockDim.x) {  // observation tile of step t0 (columns >= D stay zero for the whole launch)
        const int row = e / ldx, c = e - row * ldx;
        X[e] = (row0 + row < N && c < D) ? b.obs[((int64_t)t0 * N + row0 + row) * D + c] : 0.0f;
    }
    __syncthreads();
    double sret = 0.0, slen = 0.0, scnt = 0.0;
#ifdef TMA_ROLL_TICKS
    unsigned long long rt_last = __builtin_amdgcn_s_memtime();
#endif
    for (int k = 0; k < n_steps; k++) {
        const int t = t0 + k;
        TMA_RTICK(0);
        if constexpr (M8) {  // layer 1, k by k: lane l < 8 supplies env l's activation, block e of them is broadcast
            f32x4 c0 = f32x4{b1v[0], b1v[0], b1v[0], b1v[0]}, c1 = c0;
            const float *xr = X + (lane & 7) * ldx;
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) {
                const f32x4 a4 = *reinterpret_cast<const f32x4 *>(xr + 4 * ks);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (4 * ks + u < D) {
                        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[u], w1s[4 * ks + u], c0, 4, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[u], w1s[4 * ks + u], c1, 4, 1, 0);
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) h1[r * ld + n_base + lane] = tma_tanh(c0[r]), h1[(4 + r) * ld + n_base + lane] = tma_tanh(c1[r]);
        } else {  // layer 1: c = b1; c = mfma(X[:, 4 ks + g], W1t[4 ks + g][col], c) for ks = 0 .. KS1 - 1 -- one chain per column tile
            f32x4 c[NTW];
#pragma unroll
            for (int j = 0; j < NTW; j++) c[j] = f32x4{b1v[j], b1v[j], b1v[j], b1v[j]};
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) {
                const float a = X[r16 * ldx + 4 * ks + g];
#pragma unroll
                for (int j = 0; j < NTW; j++) c[j] = mfma16(a, w1v[j][ks], c[j]);
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) h1[(g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(c[j][r]);
        }
        __syncthreads();
        TMA_RTICK(1);
        if constexpr (M8) {
            f32x4 c0 = f32x4{b2v[0], b2v[0], b2v[0], b2v[0]}, c1 = c0;
            const float *hr = h1 + (lane & 7) * ld;
            f32x4 a4[4];  // activations three reads ahead (fenced per k-step: left to itself the scheduler reads each quad right in front of its MFMAs)
#pragma unroll
            for (int q = 0; q < 3; q++) a4[q] = *reinterpret_cast<const f32x4 *>(hr + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                if (ks + 3 < KS2) a4[(ks + 3) & 3] = *reinterpret_cast<const f32x4 *>(hr + 4 * (ks + 3));
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[ks & 3][u], w2s[4 * ks + u], c0, 4, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[ks & 3][u], w2s[4 * ks + u], c1, 4, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) h2[r * ld + n_base + lane] = tma_tanh(c0[r]), h2[(4 + r) * ld + n_base + lane] = tma_tanh(c1[r]);
        } else {  // layer 2 (the A operand of a k-step serves the four column tiles: a quarter of the LDS reads of a tile-by-tile walk)
            f32x4 c[NTW];
#pragma unroll
            for (int j = 0; j < NTW; j++) c[j] = f32x4{b2v[j], b2v[j], b2v[j], b2v[j]};
            // (round 6, as in the 8-env form: the A operands are read eight k-steps ahead behind scheduling fences -- left to the scheduler every
            //  ds_read sat in front of its four MFMAs with a full wait)
            float av[16];
            const float *hr = h1 + r16 * ld + g;
#pragma unroll
            for (int q = 0; q < 8; q++) av[q] = hr[4 * q];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                if (ks + 8 < KS2) av[(ks + 8) & 15] = hr[4 * (ks + 8)];
#pragma unroll
                for (int j = 0; j < NTW; j++) c[j] = mfma16(av[ks & 15], w2v[j][ks], c[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) h2[(g * 4 + r) * ld + n_base + 16 * j + r16] = tma_tanh(c[j][r]);
        }
        __syncthreads();
        TMA_RTICK(2);
        if (wave == 0) {
            f32x4 acc = f32x4{b3v, b3v, b3v, b3v};
            if constexpr (VH) {
                float hacc = b3h;
                const float *hr = h2 + (lane & 7) * ld, *wr = w3l + (lane >> 3) * ld;
                f32x4 hq[4], wq[4];  // operands three quads ahead of the chain
#pragma unroll
                for (int q = 0; q < 3; q++) hq[q] = *reinterpret_cast<const f32x4 *>(hr + 4 * q), wq[q] = *reinterpret_cast<const f32x4 *>(wr + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS2; ks++) {
                    if (ks + 3 < KS2) hq[(ks + 3) & 3] = *reinterpret_cast<const f32x4 *>(hr + 4 * (ks + 3)), wq[(ks + 3) & 3] = *reinterpret_cast<const f32x4 *>(wr + 4 * (ks + 3));
#pragma unroll
                    for (int u = 0; u < 4; u++) hacc = __builtin_fmaf(hq[ks & 3][u], wq[ks & 3][u], hacc);
                    __builtin_amdgcn_sched_barrier(0);
                }
                lgt[(lane & 7) * 8 + (lane >> 3)] = hacc;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < RG; r++) acc[r] = lgt[(g * RG + r) * 8 + (r16 & 7)];  // (columns 8 .. 15 are masked below: A <= 8)
            } else {
#pragma unroll
                for (int ks = 0; ks < KS2; ks++) acc = mfma16(h2[r16 * ld + 4 * ks + g], w3v[ks], acc);
            }
#ifdef TMA_ROLL_TICKS
            asm volatile("" : "+v"(acc));
#endif
            TMA_RTICK(3);
            int my_act = 0;
            float my_lp = 0.0f;
#pragma unroll
            for (int r = 0; r < RG; r++) {
                const int64_t row = row0 + g * RG + r;
                const bool colok = r16 < A;
                const float x = colok ? acc[r] : -INFINITY;
                const float m = gmax16(x);
                const float e = colok ? expf(x - m) : 0.0f;
                const float sm = gsum16(e);
                const float lse = m + logf(sm);
                const float lp = x - lse;
                int act;
                if (det) {  // deterministic evaluation: first maximal logit, as policy_fwd_wide_kernel
                    act = (int)gmin16((colok && x == m) ? (float)r16 : 99.0f);
                } else {
                    const float c = gscan16(e / sm);
                    const float u = uniform01(mix32(rng_seed, v.env_offset + (uint32_t)row, rng_step0 + (uint32_t)t));
                    const float cnt = gsum16((colok && c <= u) ? 1.0f : 0.0f);
                    act = min((int)cnt, A - 1);
                }
                const float lpa = gsum16((r16 == act) ? lp : 0.0f);
                if (r16 == r) my_act = act, my_lp = lpa;
            }
#ifdef TMA_ROLL_TICKS
            asm volatile("" : "+v"(my_act), "+v"(my_lp));
#endif
            TMA_RTICK(4);
            if (owner) {
                const int64_t off = (int64_t)t * N + i;
                b.actions[off] = my_act;
                b.log_probs[off] = my_lp;
                double r;
                bool done;
                T::step(s, my_act, nullptr, r, done);
                const int steps = T::steps(s);
                bool te, tr;
                if constexpr (T::NATIVE_TRUNC_RULE) {  // backend/mlagents/envs.py:76
                    te = done;
                    tr = (steps >= T::MAXSTEPS) && !te;
                } else {  // adapter rule, backend/mlagents/envs.py:139-145
                    const bool hit = steps >= T::MAXSTEPS;
                    te = done && !hit;
                    tr = hit;
                }
                er += r;
                b.rewards[off] = (float)r;  // (the timeout bootstrap is added by the caller's batched tma_policy_bootstrap)
                b.terminated[off] = (uint8_t)te;
                b.truncated[off] = (uint8_t)tr;
                float o[D];
                if (te || tr) {
                    if (tr) {  // terminal observation of a timed-out row: slot (t - t0) of the chunk
                        T::obs(s, o);
                        float *dst = term_obs + ((int64_t)k * N + i) * D;
#pragma unroll
                        for (int c = 0; c < D; c++) dst[c] = o[c];
                    }
                    sret += er, slen += (double)steps, scnt += 1.0;
                    log_episode(v, i, er, steps);
                    er = 0.0;
                    ce += 1;
                    if constexpr (T::USES_MT) {
                        uint32_t rec[T::RW > 0 ? T::RW : 1];
                        const uint32_t *slot = v.ring + ((int64_t)(ce % (uint32_t)v.D) * T::RW) * N + i;
#pragma unroll
                        for (int q = 0; q < T::RW; q++) rec[q] = slot[(int64_t)q * N];
                        T::from_rec(rec, s);
                    } else {
                        T::reset_inline(episode_seed(v.seed_base, v.env_offset + (uint32_t)i, ce), s);
                    }
                }
                T::obs(s, o);
                float *dst = b.obs + ((int64_t)(t + 1) * N + i) * D;
#pragma unroll
                for (int c = 0; c < D; c++) dst[c] = o[c];
#pragma unroll
                for (int c = 0; c < D; c++) X[my_row * ldx + c] = o[c];  // (every read of this step's tile is behind the two barriers above)
            }
            TMA_RTICK(5);
        }
        __syncthreads();
        TMA_RTICK(6);
    }
    if (owner) {
        T::pack(v.st, N, i, s);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
    if (wave == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sret += __shfl_down(sret, o, 64);
            slen += __shfl_down(slen, o, 64);
            scnt += __shfl_down(scnt, o, 64);
        }
        if (lane == 0 && scnt > 0.0) {
            double *slot = v.stats + (row0 >> 8) * 3;
            atomicAdd(slot + 0, sret);
            atomicAdd(slot + 1, slen);
            atomicAdd(slot + 2, scnt);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Round 6: the Box-action chunk (Crawler / Ant shapes) for the reference's own dtype -- MLP(256, 256) in f32 -- in tiles of EIGHT envs.
// Until now these shapes had a fused chunk on the bf16 MFMA only; in f32 every vector step was policy_fwd_wide_kernel + the env step kernel
// + their launch gaps: 72 us per step at 2048 envs, a fifth of the Crawler shard's f32 iteration.  The block is the 8-env form of
// rollout_chunk_wide_f32_kernel (4 waves, 64 columns each, the POLICY net only; values and the timeout bootstrap are one batched launch
// each per chunk): both hidden layers k by k on v_mfma_f32_4x4x1_16b_f32 with the A operand broadcast (4 envs x 64 columns x 1 k in 8
// cycles, no padding rows; the bits of the 16 x 16 x 4 chain), layer-2 weights in registers (half of them accumulator registers), the
// layer-1 weights -- 172 x 256 f32 at the Crawler width, no room beside them -- streamed from L2 a k-quad ahead (176 KB per block and step,
// the same 176 KB for every block), the mean head as ONE fused-multiply-add chain per (env, action dim) over all four waves (dense_head's
// order: bias, then k ascending), DiagGaussian sampling and the env step on eight lanes per env on wave 0 (the code of
// rollout_chunk_wide_cont_pi_kernel).  Bit-identical to the per-step composition (test_native_rollout_equals_stepwise_composition).
// ------------------------------------------------------------------------------------------
struct ObsDualF {  // one observation element to its global row and to its f32 LDS row
    float *gl, *img;
    struct Ref {
        float *gl, *img;
        __device__ __forceinline__ void operator=(float x) const { *gl = x, *img = x; }
    };
    __device__ __forceinline__ Ref operator[](int k) const { return Ref{gl + k, img + k}; }
    __device__ __forceinline__ ObsDualF operator+(int k) const { return ObsDualF{gl + k, img + k}; }
};
template <class T>
struct WideContF32Lds {
    static constexpr int M = 8, H = 256, LD = H + 4, LDX = ((T::OBS + 3) & ~3) + 4, AP = 32;
    // observation rows, two activation images, the head's weights as [action dim][k] rows, means and sampled actions [M][32], env state
    // [SW][M], truncated / done flags, per-joint terms of the multi-lane env step
    static constexpr int floats() { return M * LDX + 2 * M * LD + AP * LD + 2 * M * 32 + T::SW * M + 64 + M * T::NJ * 5; }
};
template <class T>
__global__ __launch_bounds__(256, 1) void rollout_chunk_wide_cont_f32_kernel(EnvView v, const float *__restrict__ params, PLayout L, ChunkPtrs b,
                                                                             float *__restrict__ term_obs, int t0, int n_steps, uint32_t rng_seed,
                                                                             uint32_t rng_step0, int det) {
    extern __shared__ __attribute__((aligned(16))) float smem_cf[];
    using W = WideContF32Lds<T>;
    constexpr int M = W::M, H = W::H, ld = W::LD, ldx = W::LDX, D = T::OBS, AD = T::ADIM, KQ1 = (D + 3) >> 2, KS2 = H / 4;
    static_assert(T::NACT == 0 && AD <= 32 && !T::USES_MT, "fused f32 wide rollout, Box actions: <= 32 action dims, inline resets");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4, n_base = 64 * wave;
    float *X = smem_cf, *h1 = X + M * ldx, *h2 = h1 + M * ld, *w3l = h2 + M * ld, *means = w3l + W::AP * ld, *actl = means + M * 32;
    uint32_t *stl = reinterpret_cast<uint32_t *>(actl + M * 32);  // [SW][M] env state words
    int *trf = reinterpret_cast<int *>(stl + T::SW * M), *dnf = trf + 32;
    float *termsl = reinterpret_cast<float *>(dnf + 32);  // [M][NJ][5]
    const int64_t N = v.N;
    const int64_t row0 = (int64_t)blockIdx.x * M;
    float *act_out = reinterpret_cast<float *>(b.actions);
    const Net P = pi_net(params, L);
    // ---- layer-2 weights: column n_base + lane, one register per k (the upper half in accumulator registers) ----
    float w2s[H];
#pragma unroll
    for (int kk = 0; kk < H; kk++) w2s[kk] = P.W2t[(int64_t)kk * H + n_base + lane];
#pragma unroll
    for (int kk = H / 2; kk < H; kk++) asm volatile("" : "+a"(w2s[kk]));
    const float b1s = P.b1[n_base + lane], b2s = P.b2[n_base + lane];
    const float *w1col = P.W1t + n_base + lane;  // layer 1: W1t[k][col], streamed
    for (int e = threadIdx.x; e < W::AP * ld; e += blockDim.x) {  // head weights as [action dim][k] rows (zero rows beyond AD)
        const int a = e / ld, kk = e - a * ld;
        w3l[e] = (a < AD && kk < H) ? P.W3t[(int64_t)kk * AD + a] : 0.0f;
    }
    const int hp = wave * 64 + lane, he = hp & 7, ha = hp >> 3;  // head pair of this lane: env he, action dim ha (live below AD)
    const float b3h = ha < AD ? P.b3[ha] : 0.0f;
    // ---- env state -> LDS; eight lanes per env on wave 0 ----
    const int my_row = lane >> 3, sub = lane & 7;
    const int64_t i = row0 + my_row;
    const bool grp_ok = wave == 0 && i < N;
    const bool owner = grp_ok && sub == 0;
    double er = 0.0;
    uint32_t ce = 0;
    static_assert(sizeof(typename T::S) <= T::SW * 4, "state struct fits its LDS slot");
    typename T::S *sl = reinterpret_cast<typename T::S *>(stl);
    if (owner) {
        T::unpack(v.st, N, i, sl[my_row]);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    for (int e = threadIdx.x; e < M * ldx; e += blockDim.x) {  // observation rows of step t0 (columns >= D stay zero for the whole launch)
        const int row = e / ldx, c = e - row * ldx;
        X[e] = (row0 + row < N && c < D) ? b.obs[((int64_t)t0 * N + row0 + row) * D + c] : 0.0f;
    }
    if (threadIdx.x < 64) trf[threadIdx.x] = 0;  // (trf and dnf)
    __syncthreads();
    const float *ls = params + L.log_std;
    float lsd_v[2], sd_v[2];  // log_std and exp(log_std) of this lane's two action columns: constant over the launch
#pragma unroll
    for (int j = 0; j < 2; j++) {
        lsd_v[j] = (16 * j + r16 < AD) ? ls[16 * j + r16] : 0.0f;
        sd_v[j] = expf(lsd_v[j]);
    }
    double sret = 0.0, slen = 0.0, scnt = 0.0;
#ifdef TMA_ROLL_TICKS
    unsigned long long rt_last = __builtin_amdgcn_s_memtime();
#endif
    for (int k = 0; k < n_steps; k++) {
        const int t = t0 + k;
        TMA_RTICK(0);
        {  // layer 1, k by k: lane l < 8 supplies env l's observation, block e of them is broadcast; weights four k-quads ahead of their use
            f32x4 c0 = f32x4{b1s, b1s, b1s, b1s}, c1 = c0;
            const float *xr = X + (lane & 7) * ldx;
            constexpr int PF = 4;
            f32x4 a4[PF];
            float wq[PF][4];
#pragma unroll
            for (int q = 0; q < PF - 1; q++) {
                a4[q] = *reinterpret_cast<const f32x4 *>(xr + 4 * q);
#pragma unroll
                for (int u = 0; u < 4; u++) wq[q][u] = w1col[(int64_t)(4 * q + u < D ? 4 * q + u : 0) * H];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 4
            for (int q = 0; q < KQ1; q++) {
                const int qn = q + PF - 1;
                if (qn < KQ1) {
                    a4[qn % PF] = *reinterpret_cast<const f32x4 *>(xr + 4 * qn);
#pragma unroll
                    for (int u = 0; u < 4; u++) wq[qn % PF][u] = w1col[(int64_t)(4 * qn + u < D ? 4 * qn + u : 0) * H];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (4 * q + u < D) {
                        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q % PF][u], wq[q % PF][u], c0, 4, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q % PF][u], wq[q % PF][u], c1, 4, 1, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) h1[r * ld + n_base + lane] = tma_tanh(c0[r]), h1[(4 + r) * ld + n_base + lane] = tma_tanh(c1[r]);
        }
        __syncthreads();
        TMA_RTICK(1);
        {  // layer 2 (rollout_chunk_wide_f32_kernel<T, 8>'s loop)
            f32x4 c0 = f32x4{b2s, b2s, b2s, b2s}, c1 = c0;
            const float *hr = h1 + (lane & 7) * ld;
            f32x4 a4[4];
#pragma unroll
            for (int q = 0; q < 3; q++) a4[q] = *reinterpret_cast<const f32x4 *>(hr + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                if (ks + 3 < KS2) a4[(ks + 3) & 3] = *reinterpret_cast<const f32x4 *>(hr + 4 * (ks + 3));
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[ks & 3][u], w2s[4 * ks + u], c0, 4, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[ks & 3][u], w2s[4 * ks + u], c1, 4, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) h2[r * ld + n_base + lane] = tma_tanh(c0[r]), h2[(4 + r) * ld + n_base + lane] = tma_tanh(c1[r]);
        }
        __syncthreads();
        TMA_RTICK(2);
        if (ha < AD) {  // mean head: one fused-multiply-add chain per (env, action dim) -- dense_head's operations in dense_head's order
            float hacc = b3h;
            const float *hr = h2 + he * ld, *wr = w3l + ha * ld;
            f32x4 hq[4], wq[4];
#pragma unroll
            for (int q = 0; q < 3; q++) hq[q] = *reinterpret_cast<const f32x4 *>(hr + 4 * q), wq[q] = *reinterpret_cast<const f32x4 *>(wr + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                if (ks + 3 < KS2) hq[(ks + 3) & 3] = *reinterpret_cast<const f32x4 *>(hr + 4 * (ks + 3)), wq[(ks + 3) & 3] = *reinterpret_cast<const f32x4 *>(wr + 4 * (ks + 3));
#pragma unroll
                for (int u = 0; u < 4; u++) hacc = __builtin_fmaf(hq[ks & 3][u], wq[ks & 3][u], hacc);
                __builtin_amdgcn_sched_barrier(0);
            }
            means[he * 32 + ha] = hacc;
        }
        __syncthreads();
        TMA_RTICK(3);
        {
            // DiagGaussian sample + log-prob: policy_fwd_wide_kernel<CONT>'s streams and arithmetic, ONE sample per lane over all four waves --
            // wave w takes rows 2 w and 2 w + 1, lane group g row 2 w + (g & 1) and action columns 16 (g >> 1) + r16 (on wave 0 alone this was
            // four samples a lane, 2.9 k cycles of a Crawler step with three waves waiting).  The two column tiles' terms of a row meet through
            // a cross-lane read (0 + t0) + t1, then the 16-lane sum: the order of the reference loop.
            const int lrow = 2 * wave + (g & 1), jt = g >> 1, col = 16 * jt + r16;
            const int64_t row = row0 + lrow;
            const uint32_t gi = v.env_offset + (uint32_t)row;
            const uint32_t rstep = rng_step0 + (uint32_t)t;
            float term = 0.0f;
            if (col < AD) {
                const float mu = means[lrow * 32 + col], lsd = lsd_v[jt], sd = sd_v[jt];
                const float u1 = fmaxf(uniform01(mix32(rng_seed ^ (0x68E31DA4u + (uint32_t)col * 0x9E3779B9u), gi, rstep)), 5.9604645e-08f);
                const float u2 = uniform01(mix32(rng_seed ^ (0xB5297A4Du + (uint32_t)col * 0x85EBCA77u), gi, rstep));
                const float z = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
                const float a = det ? mu : mu + sd * z;  // deterministic evaluation: the mean (policy_fwd_wide_kernel)
                const float dd = a - mu;
                term = -(dd * dd) / (2.0f * (sd * sd)) - lsd - 0.9189385332046727f;
                actl[lrow * 32 + col] = a;
                if (row < N) act_out[((int64_t)t * N + row) * AD + col] = a;
            }
            const float t1 = __shfl(term, (lane + 32) & 63, 64);  // (lanes of column tile 0 read their row's tile-1 term)
            float lpsum = 0.0f;
            lpsum += term;
            if (16 + r16 < AD) lpsum += t1;
            lpsum = gsum16(lpsum);
            if (jt == 0 && r16 == 0 && row < N) b.log_probs[(int64_t)t * N + row] = lpsum;
        }
        __syncthreads();  // the sampled actions of every row are in LDS
        if (wave == 0) {
            TMA_RTICK(4);
            // env step on eight lanes per env (rollout_chunk_wide_cont_pi_kernel's sequence)
            typename T::S &s = sl[my_row];
            double r = 0.0;
            bool done = false;
            if (grp_ok) T::template step_lanes<8>(s, actl + my_row * 32, termsl + my_row * (T::NJ * 5), sub, r, done);
            if (owner) {
                const int64_t off = (int64_t)t * N + i;
                const bool hit = T::steps(s) >= T::MAXSTEPS;  // adapter rule, backend/mlagents/envs.py:139-145
                const bool te = done && !hit, tr = hit;
                er += r;
                b.rewards[off] = (float)r;  // (the timeout bootstrap is added by the caller's batched tma_policy_bootstrap)
                b.terminated[off] = (uint8_t)te;
                b.truncated[off] = (uint8_t)tr;
                trf[my_row] = tr ? 1 : 0;
                dnf[my_row] = (te || tr) ? 1 : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool g_tr = grp_ok && trf[my_row] != 0, g_dn = grp_ok && dnf[my_row] != 0;
            if (g_tr) T::template obs_lanes<8>(s, sub, term_obs + ((int64_t)k * N + i) * D);  // terminal observation: slot (t - t0) of the chunk
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (g_dn && sub == 0) {
                const int steps = T::steps(s);
                sret += er, slen += (double)steps, scnt += 1.0;
                log_episode(v, i, er, steps);
                er = 0.0;
                ce += 1;
                T::reset_inline(episode_seed(v.seed_base, v.env_offset + (uint32_t)i, ce), s);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (grp_ok) T::template obs_lanes<8>(s, sub, ObsDualF{b.obs + ((int64_t)(t + 1) * N + i) * D, X + my_row * ldx});
            TMA_RTICK(5);
        }
        __syncthreads();
        TMA_RTICK(6);
    }
    if (owner) {
        T::pack(v.st, N, i, sl[my_row]);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
    if (wave == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sret += __shfl_down(sret, o, 64);
            slen += __shfl_down(slen, o, 64);
            scnt += __shfl_down(scnt, o, 64);
        }
        if (lane == 0 && scnt > 0.0) {
            double *slot = v.stats + (row0 >> 8) * 3;
            atomicAdd(slot + 0, sret);
            atomicAdd(slot + 1, slen);
            atomicAdd(slot + 2, scnt);
        }
    }
}

template <class T>
static int launch_chunk_wide_cont_f32(tma_env *env, const float *params, const PLayout &L, const ChunkPtrs &b, float *term_obs, int t0, int n,
                                      uint32_t rng_seed, uint32_t rng_step0, int det, hipStream_t s) {
    if constexpr (T::FUSED_ROLLOUT && T::NACT == 0 && T::ADIM <= 32 && !T::USES_MT && T::OBS <= 192) {
        auto k = rollout_chunk_wide_cont_f32_kernel<T>;
        const int smem = WideContF32Lds<T>::floats() * 4;
        TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        k<<<dim3((unsigned)ceil_div(env->v.N, 8)), dim3(256), smem, s>>>(env->v, params, L, b, term_obs, t0, n, rng_seed, rng_step0, det);
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    } else {
        return fail(TMA_ERR_INVALID, "no fused f32 wide rollout for this Box-action task");
    }
}

template <class T>
static int launch_chunk_wide_f32(tma_env *env, const float *params, const PLayout &L, const ChunkPtrs &b, float *term_obs, int t0, int n, uint32_t rng_seed,
                                 uint32_t rng_step0, int det, hipStream_t s) {
    // (round 6: BrickBreak -- 45 observations, one of the reference's PPO-default tasks -- on the 8-env tiles, whose layer-1 weights are one register per k)
    if constexpr ((T::FUSED_ROLLOUT && T::OBS <= 32 && T::NACT > 0) || T::ID == TMA_TASK_BRICKBREAK) {
        // tiles of eight envs while they give every env its own block round (up to 2048 envs); TMA_WIDE_F32_ROWS=16 / 8 forces a form (A/B, tests)
        static const int force = getenv("TMA_WIDE_F32_ROWS") ? atoi(getenv("TMA_WIDE_F32_ROWS")) : 0;
        const bool m8 = force ? force == 8 : env->v.N <= 2048;
        if (m8) {
            constexpr int ldx = ((T::OBS + 3) & ~3) + 4;
            const int smem = (8 * ldx + 8 * 260 + 16 * 260 + 8 * 260 + 64) * 4;  // X, h1, h2 (16 rows), W3 rows, logits
            rollout_chunk_wide_f32_kernel<T, 8><<<dim3((unsigned)ceil_div(env->v.N, 8)), dim3(256), smem, s>>>(env->v, params, L, b, term_obs, t0, n, rng_seed, rng_step0, det);
            TMA_LAUNCH_CHECK();
            return TMA_OK;
        }
        if constexpr (T::OBS <= 32) {
            auto k = rollout_chunk_wide_f32_kernel<T, 16>;
            constexpr int ldx = ((T::OBS + 3) & ~3) + 2;
            const int smem = 16 * (ldx + 2 * 258) * 4;
            k<<<dim3((unsigned)ceil_div(env->v.N, 16)), dim3(256), smem, s>>>(env->v, params, L, b, term_obs, t0, n, rng_seed, rng_step0, det);
        } else {
            return fail(TMA_ERR_INVALID, "fused f32 wide rollout: this task runs on 8-env tiles only (up to 2048 envs)");
        }
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    } else {
        return fail(TMA_ERR_INVALID, "no fused f32 wide rollout for this task");
    }
}

template <class T>
static int launch_chunk_wide_cont_pi(tma_env *env, const float *params, const PLayout &L, const ChunkPtrs &b, float *term_obs, int t0, int n,
                                     uint32_t rng_seed, uint32_t rng_step0, int det, hipStream_t s) {
    if constexpr (T::FUSED_ROLLOUT && T::NACT == 0 && T::ADIM <= 32 && !T::USES_MT && T::OBS <= 192) {
        // 16 envs per block while that still leaves CUs idle at 32 (up to 4096 envs: <= 256 blocks of 16), else 32 (TMA_CONT_ROWS=16/32 forces one)
        static const int forced = getenv("TMA_CONT_ROWS") ? atoi(getenv("TMA_CONT_ROWS")) : 0;
        const bool rows16 = forced ? forced == 16 : env->v.N <= 4096;
        if (rows16) {
            auto k = rollout_chunk_wide_cont_pi_kernel<T, 16>;
            const int smem = WideContPiLds<T, 16>::bytes();
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            k<<<dim3((unsigned)ceil_div(env->v.N, 16)), dim3(512), smem, s>>>(env->v, params, L, b, term_obs, t0, n, rng_seed, rng_step0, det);
        } else {
            auto k = rollout_chunk_wide_cont_pi_kernel<T, 32>;
            const int smem = WideContPiLds<T, 32>::bytes();
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            k<<<dim3((unsigned)ceil_div(env->v.N, 32)), dim3(512), smem, s>>>(env->v, params, L, b, term_obs, t0, n, rng_seed, rng_step0, det);
        }
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    } else {
        return fail(TMA_ERR_INVALID, "no policy-only fused wide rollout for this task");
    }
}

template <class T>
static int launch_chunk_wide(tma_env *env, const float *params, const PLayout &L, const ChunkPtrs &b, int t0, int n, uint32_t rng_seed,
                             uint32_t rng_step0, float gamma, int det, hipStream_t s) {
    if constexpr (T::FUSED_ROLLOUT && T::OBS <= 32 && T::NACT > 0) {
        // 16 envs per block while 32 would leave CUs idle (up to 4096 envs: <= 256 blocks of 16), else 32 (TMA_WIDE_ROWS=16/32 forces one)
        static const int forced = getenv("TMA_WIDE_ROWS") ? atoi(getenv("TMA_WIDE_ROWS")) : 0;
        const bool rows16 = forced ? forced == 16 : env->v.N <= 4096;
        if (rows16) {
            auto k = rollout_chunk_wide_bf_kernel<T, 16>;
            const int smem = WideLds<16>::bytes();
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            k<<<dim3((unsigned)ceil_div(env->v.N, 16)), dim3(512), smem, s>>>(env->v, params, L, b, t0, n, rng_seed, rng_step0, gamma, det);
        } else {
            auto k = rollout_chunk_wide_bf_kernel<T, 32>;
            const int smem = WideLds<32>::bytes();
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            k<<<dim3((unsigned)ceil_div(env->v.N, 32)), dim3(512), smem, s>>>(env->v, params, L, b, t0, n, rng_seed, rng_step0, gamma, det);
        }
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    } else if constexpr (T::FUSED_ROLLOUT && T::NACT == 0 && T::ADIM <= 32 && !T::USES_MT && ((T::OBS + 31) / 32) % 2 == 0) {
        auto k = rollout_chunk_wide_cont_kernel<T>;
        const int smem = WideContLds<T>::bytes();
        TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        k<<<dim3((unsigned)ceil_div(env->v.N, 32)), dim3(512), smem, s>>>(env->v, params, L, b, t0, n, rng_seed, rng_step0, gamma, det);
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    } else {
        return fail(TMA_ERR_INVALID, "no fused wide rollout for this task");
    }
}

template <class T>
static int launch_chunk(tma_env *env, const float *params, const PLayout &L, const ChunkPtrs &b, int t0, int n, uint32_t rng_seed, uint32_t rng_step0,
                        float gamma, int det, hipStream_t s) {
    const int64_t tiles = ceil_div(env->v.N, 16);
    const int wpb = tiles >= 1024 ? 4 : 1;  // BASELINE shape (256 tiles): one wave per block so all 256 CUs take part
    const int smem = (2 * FWD_IMG + wpb * 2 * 16 * CH_LDX) * 4;
    static const bool roll2 = getenv("TMA_ROLL2") != nullptr;  // A/B switch: the two-wave kernel
    if (wpb == 1 && !roll2) {  // one tile per CU: each net on two waves (round 6)
        const int smem4 = (2 * FWD_IMG + 4 * 16 * CH_LDX + 32 + 32 + 4 + 1024 + 512 + 512 + 512 + 96) * 4;
        auto k4 = rollout_chunk4_h64_kernel<T>;
        if (smem4 > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, smem4));
        k4<<<dim3((unsigned)tiles), dim3(256), smem4, s>>>(env->v, params, L, b, t0, n, rng_seed, rng_step0, gamma, det);
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    if (wpb == 1) {  // one tile per CU: split the policy and the value net of a tile over two waves
        const int smem2 = (2 * FWD_IMG + 4 * 16 * CH_LDX + 32 + 32 + 4) * 4;
        auto k2 = rollout_chunk2_h64_kernel<T>;
        if (smem2 > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, smem2));
        k2<<<dim3((unsigned)tiles), dim3(128), smem2, s>>>(env->v, params, L, b, t0, n, rng_seed, rng_step0, gamma, det);
        TMA_LAUNCH_CHECK();
        return TMA_OK;
    }
    auto k = rollout_chunk_h64_kernel<T>;
    if (smem > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    k<<<dim3((unsigned)ceil_div(tiles, wpb)), dim3(64 * wpb), smem, s>>>(env->v, params, L, b, t0, n, rng_seed, rng_step0, gamma, det);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

}  // namespace tma

// Test aid: fill the LDS of every CU with a bit pattern (default: quiet NaNs).  LDS is not cleared between workgroups, so a kernel that reads
// a word it never wrote sees whatever ran before it on that CU -- usually something finite, which hides the bug.  Poisoning first makes such
// reads show up as NaNs in the outputs (tests/test_ppo_gpu.py runs the fused rollout kernels behind it).
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_words[];
    constexpr int WORDS = 160 * 1024 / 4;
    for (int e = threadIdx.x; e < WORDS; e += blockDim.x) lds_words[e] = pattern;
    __syncthreads();
    if (sink && lds_words[(threadIdx.x * 97) % WORDS] != pattern) sink[0] = 1;  // (keeps the stores alive)
}

extern "C" int tma_debug_poison_lds(unsigned pattern, void *stream) {
    auto k = poison_lds_kernel;
    TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    k<<<dim3(1024), dim3(256), 160 * 1024, (hipStream_t)stream>>>(pattern ? pattern : 0x7FC00000u, nullptr);  // four rounds over the 256 CUs
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

extern "C" int tma_rollout_collect(tma_env *env, const float *params, const tma_policy_dims *d, const tma_rollout_buffers *b, int t_begin,
                                   int t_end, int T, uint32_t rng_seed, uint32_t rng_step0, uint32_t env_offset, double gamma,
                                   int compute_last_values, int deterministic, void *stream) {
    using namespace tma;
    const int det = deterministic ? 1 : 0;
    if (!env || !params || !d || !b) return fail(TMA_ERR_INVALID, "tma_rollout_collect: null argument");
    if (!b->obs || !b->actions || !b->rewards || !b->values || !b->log_probs || !b->terminated || !b->truncated || !b->terminal_obs)
        return fail(TMA_ERR_INVALID, "tma_rollout_collect: rollout buffers has a null plane");
    if (t_begin < 0 || t_end > T || t_begin > t_end) return fail(TMA_ERR_INVALID, "bad step range [%d, %d) for T=%d", t_begin, t_end, T);
    const int64_t N = b->N;
    if (N != env->v.N) return fail(TMA_ERR_INVALID, "rollout buffers are for %lld envs, the env handle has %lld", (long long)N, (long long)env->v.N);
    const PLayout L = make_layout(d->obs_dim, d->hidden, d->act_dim, d->continuous, d->mfma_dtype);
    const bool fused = L.img_pi >= 0 && env->is_reset &&
                       (env->task == TMA_TASK_GRIDWORLD || env->task == TMA_TASK_PUSH || env->task == TMA_TASK_BALL3D || env->task == TMA_TASK_WALLJUMP ||
                        env->task == TMA_TASK_BICYCLE || env->task == TMA_TASK_GLIDER) &&
                       d->obs_dim == tma_task_obs_dim(env->task) && d->act_dim == tma_task_num_actions(env->task);
    // 256-wide bf16 policies on the Discrete tasks with observations of up to 32 floats: fused chunk with register-resident weights
    static const bool no_wide_fused = getenv("TMA_NO_WIDE_FUSED") != nullptr;  // test hook: the per-step composition
    const bool task_fused = dispatch_task(env->task, [](auto t) { return decltype(t)::FUSED_ROLLOUT ? 1 : 0; }) == 1;
    const bool fused_disc = task_fused && !d->continuous && env->task != TMA_TASK_CRAWLER && env->task != TMA_TASK_ANT && d->act_dim == tma_task_num_actions(env->task) && d->obs_dim <= 32;
    // ... and on the Crawler shape (Box actions, 172 observations): layer-1 fragments streamed per step, env state in LDS
    const bool fused_cont = d->continuous && (env->task == TMA_TASK_CRAWLER || env->task == TMA_TASK_ANT) && d->act_dim == tma_task_act_dim(env->task);
    const bool fused_wide = !no_wide_fused && L.bf16 && L.H == 256 && env->is_reset && d->obs_dim == tma_task_obs_dim(env->task) && (fused_disc || fused_cont);
    // ... and in f32 (the reference's dtype), round 6: tiles of eight envs, up to 4096 envs (two block rounds; beyond that the per-step composition)
    static const bool no_cont_f32 = getenv("TMA_NO_CONT_F32_FUSED") != nullptr;  // A/B switch
    const bool fused_cont_f32 = !no_wide_fused && !no_cont_f32 && !L.bf16 && L.img_pi < 0 && L.H == 256 && env->is_reset && fused_cont &&
                                d->obs_dim == tma_task_obs_dim(env->task) && env->v.N <= 4096;
    // Box-action tasks (Crawler / Ant shapes), round 4: policy-only fused chunk + ONE batched value launch + ONE batched bootstrap launch per
    // chunk of up to terminal_obs_slots steps (TMA_CONT_TWO_NET=1: the round-2 chunk with both nets in the step loop)
    static const bool cont_two_net = getenv("TMA_CONT_TWO_NET") != nullptr;
    if ((fused_wide && fused_cont && !cont_two_net) || fused_cont_f32) {
        TMA_HIP(hipSetDevice(env->device));
        ChunkPtrs cp{b->obs, static_cast<int32_t *>(b->actions), b->rewards, b->values, b->log_probs, b->terminated, b->truncated};
        const int D = d->obs_dim;
        // The chunk kernel holds the registers of N / 16 (or N / 32) CUs and nothing of the others; the value pass of chunk c depends only on the
        // observations chunk c left behind.  So it runs on a SIDE stream behind an event on chunk c, beside the chunk kernel of chunk c + 1 on the
        // CUs that one leaves idle; the terminal-observation slots are split in two halves used alternately, so that chunk c + 1 does not overwrite
        // what the bootstrap of chunk c still reads.  The call returns with `stream` waiting for the side stream.  (TMA_CONT_SERIAL=1: one stream.)
        static const bool serial = getenv("TMA_CONT_SERIAL") != nullptr;
        const int slots = b->terminal_obs_slots > 1 ? b->terminal_obs_slots : 1;
        const bool overlap = !serial && slots >= 2;
        const int K = overlap ? slots / 2 : slots;
        hipStream_t mainS = (hipStream_t)stream, sideS = mainS;
        if (overlap) {
            if (!env->side) {
                TMA_HIP(hipStreamCreateWithFlags(&env->side, hipStreamNonBlocking));
                TMA_HIP(hipEventCreateWithFlags(&env->ev_chunk, hipEventDisableTiming));
                TMA_HIP(hipEventCreateWithFlags(&env->ev_side[0], hipEventDisableTiming));
                TMA_HIP(hipEventCreateWithFlags(&env->ev_side[1], hipEventDisableTiming));
            }
            sideS = env->side;
        }
        int t = t_begin, half = 0;
        bool used[2] = {false, false};  // ev_side[h] has been recorded behind a bootstrap that reads half h
        while (t < t_end) {
            int left = 0;
            int rc = tma_env_steps_until_refill(env, &left);
            if (rc) return rc;
            int n = left < (t_end - t) ? left : (t_end - t);
            if (n > K) n = K;  // one terminal-observation slot per step of the chunk
            float *tobs = b->terminal_obs + (overlap ? (int64_t)half * K * N * D : 0);
            if (overlap && used[half])  // the half about to be rewritten was read by the bootstrap launched two chunks ago
                TMA_HIP(hipStreamWaitEvent(mainS, env->ev_side[half], 0));
            rc = dispatch_task(env->task, [&](auto task) {
                using TT = decltype(task);
                if (fused_cont_f32) return launch_chunk_wide_cont_f32<TT>(env, params, L, cp, tobs, t, n, rng_seed, rng_step0, det, mainS);
                return launch_chunk_wide_cont_pi<TT>(env, params, L, cp, tobs, t, n, rng_seed, rng_step0, det, mainS);
            });
            if (rc) return rc;
            if (overlap) {
                TMA_HIP(hipEventRecord(env->ev_chunk, mainS));
                TMA_HIP(hipStreamWaitEvent(sideS, env->ev_chunk, 0));
            }
            rc = tma_policy_values(params, d, b->obs + (int64_t)t * N * D, (int64_t)n * N, b->values + (int64_t)t * N, sideS);
            if (rc) return rc;
            rc = tma_policy_bootstrap(params, d, tobs, b->truncated + (int64_t)t * N, (int64_t)n * N, gamma, b->rewards + (int64_t)t * N, sideS);
            if (rc) return rc;
            if (overlap) {
                TMA_HIP(hipEventRecord(env->ev_side[half], sideS));
                used[half] = true;
                half ^= 1;
            }
            rc = tma_env_internal_after_steps(env, n, stream);
            if (rc) return rc;
            t += n;
        }
        for (int h = 0; h < 2; h++)
            if (overlap && used[h]) TMA_HIP(hipStreamWaitEvent(mainS, env->ev_side[h], 0));
        if (compute_last_values && t_end == T) {
            if (!b->last_values) return fail(TMA_ERR_INVALID, "last_values is null");
            return tma_policy_values(params, d, b->obs + (int64_t)T * N * D, N, b->last_values, stream);
        }
        return TMA_OK;
    }
    if (fused_wide) {
        TMA_HIP(hipSetDevice(env->device));
        ChunkPtrs cp{b->obs, static_cast<int32_t *>(b->actions), b->rewards, b->values, b->log_probs, b->terminated, b->truncated};
        int t = t_begin;
        while (t < t_end) {
            int left = 0;
            int rc = tma_env_steps_until_refill(env, &left);
            if (rc) return rc;
            const int n = left < (t_end - t) ? left : (t_end - t);
            rc = dispatch_task(env->task, [&](auto task) {
                using TT = decltype(task);
                return launch_chunk_wide<TT>(env, params, L, cp, t, n, rng_seed, rng_step0, (float)gamma, det, (hipStream_t)stream);
            });
            if (rc) return rc;
            rc = tma_env_internal_after_steps(env, n, stream);
            if (rc) return rc;
            t += n;
        }
        if (compute_last_values && t_end == T) {
            if (!b->last_values) return fail(TMA_ERR_INVALID, "last_values is null");
            return tma_policy_values(params, d, b->obs + (int64_t)T * N * d->obs_dim, N, b->last_values, stream);
        }
        return TMA_OK;
    }
    // the reference's default MLP(256, 256) in f32 on the Discrete tasks (observations of up to 32 floats): policy-only fused chunk + ONE
    // batched value launch + ONE batched bootstrap launch per chunk of up to terminal_obs_slots steps
    static const int rows_forced = getenv("TMA_WIDE_F32_ROWS") ? atoi(getenv("TMA_WIDE_F32_ROWS")) : 0;
    const bool fused_bb = env->task == TMA_TASK_BRICKBREAK && !d->continuous && d->act_dim == tma_task_num_actions(env->task) && env->v.N <= 2048 && rows_forced != 16;
    const bool fused_wide_f32 = !no_wide_fused && !L.bf16 && L.img_pi < 0 && L.H == 256 && env->is_reset && (fused_disc || fused_bb) &&
                                d->obs_dim == tma_task_obs_dim(env->task);
    if (fused_wide_f32) {
        TMA_HIP(hipSetDevice(env->device));
        ChunkPtrs cp{b->obs, static_cast<int32_t *>(b->actions), b->rewards, b->values, b->log_probs, b->terminated, b->truncated};
        const int K = b->terminal_obs_slots > 1 ? b->terminal_obs_slots : 1;
        const int D = d->obs_dim;
        int t = t_begin;
        while (t < t_end) {
            int left = 0;
            int rc = tma_env_steps_until_refill(env, &left);
            if (rc) return rc;
            int n = left < (t_end - t) ? left : (t_end - t);
            if (n > K) n = K;  // one terminal-observation slot per step of the chunk
            rc = dispatch_task(env->task, [&](auto task) {
                using TT = decltype(task);
                return launch_chunk_wide_f32<TT>(env, params, L, cp, b->terminal_obs, t, n, rng_seed, rng_step0, det, (hipStream_t)stream);
            });
            if (rc) return rc;
            rc = tma_policy_values(params, d, b->obs + (int64_t)t * N * D, (int64_t)n * N, b->values + (int64_t)t * N, stream);
            if (rc) return rc;
            rc = tma_policy_bootstrap(params, d, b->terminal_obs, b->truncated + (int64_t)t * N, (int64_t)n * N, gamma, b->rewards + (int64_t)t * N, stream);
            if (rc) return rc;
            rc = tma_env_internal_after_steps(env, n, stream);
            if (rc) return rc;
            t += n;
        }
        if (compute_last_values && t_end == T) {
            if (!b->last_values) return fail(TMA_ERR_INVALID, "last_values is null");
            return tma_policy_values(params, d, b->obs + (int64_t)T * N * D, N, b->last_values, stream);
        }
        return TMA_OK;
    }
    if (fused) {
        TMA_HIP(hipSetDevice(env->device));
        ChunkPtrs cp{b->obs, static_cast<int32_t *>(b->actions), b->rewards, b->values, b->log_probs, b->terminated, b->truncated};
        int t = t_begin;
        while (t < t_end) {
            int left = 0;
            int rc = tma_env_steps_until_refill(env, &left);
            if (rc) return rc;
            const int n = left < (t_end - t) ? left : (t_end - t);
            if (env->task == TMA_TASK_GRIDWORLD) rc = launch_chunk<GridTask>(env, params, L, cp, t, n, rng_seed, rng_step0, (float)gamma, det, (hipStream_t)stream);
            else if (env->task == TMA_TASK_PUSH) rc = launch_chunk<PushTask>(env, params, L, cp, t, n, rng_seed, rng_step0, (float)gamma, det, (hipStream_t)stream);
            else if (env->task == TMA_TASK_WALLJUMP) rc = launch_chunk<WallJumpTask>(env, params, L, cp, t, n, rng_seed, rng_step0, (float)gamma, det, (hipStream_t)stream);
            else if (env->task == TMA_TASK_BICYCLE) rc = launch_chunk<BicycleTask>(env, params, L, cp, t, n, rng_seed, rng_step0, (float)gamma, det, (hipStream_t)stream);
            else if (env->task == TMA_TASK_GLIDER) rc = launch_chunk<GliderTask>(env, params, L, cp, t, n, rng_seed, rng_step0, (float)gamma, det, (hipStream_t)stream);
            else rc = launch_chunk<BallTask>(env, params, L, cp, t, n, rng_seed, rng_step0, (float)gamma, det, (hipStream_t)stream);
            if (rc) return rc;
            rc = tma_env_internal_after_steps(env, n, stream);
            if (rc) return rc;
            t += n;
        }
        if (compute_last_values && t_end == T) {
            if (!b->last_values) return fail(TMA_ERR_INVALID, "last_values is null");
            return tma_policy_values(params, d, b->obs + (int64_t)T * N * d->obs_dim, N, b->last_values, stream);
        }
        return TMA_OK;
    }
    const int D = d->obs_dim, A = d->continuous ? d->act_dim : 1;
    const size_t act_elem = d->continuous ? sizeof(float) : sizeof(int32_t);
    const int K = b->terminal_obs_slots > 1 ? b->terminal_obs_slots : 1;
    if (K == 1 && !det) {
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
    } else {
        // K terminal-observation slots: step t writes slot (t - w0) of its window [w0, w0 + K); the window's bootstraps are one launch
        // over the K*N rows (terminal_obs, truncated[w0..] and rewards[w0..] are all contiguous in the step index).
        int w0 = t_begin;
        for (int t = t_begin; t < t_end; t++) {
            const float *obs_t = b->obs + (int64_t)t * N * D;
            float *obs_next = b->obs + (int64_t)(t + 1) * N * D;
            void *act_t = static_cast<char *>(b->actions) + (int64_t)t * N * A * act_elem;
            int rc = tma_policy_act(params, d, obs_t, N, rng_seed, rng_step0 + (uint32_t)t, env_offset, det, act_t, b->values + (int64_t)t * N,
                                    b->log_probs + (int64_t)t * N, stream);
            if (rc) return rc;
            rc = tma_env_step(env, act_t, d->continuous ? TMA_ACT_F32 : TMA_ACT_I32, 0, 0, 1, obs_next, b->rewards + (int64_t)t * N,
                              b->terminated + (int64_t)t * N, b->truncated + (int64_t)t * N, b->terminal_obs + (int64_t)(t - w0) * N * D, nullptr,
                              nullptr, stream);
            if (rc) return rc;
            if (t - w0 + 1 == K || t == t_end - 1) {
                rc = tma_policy_bootstrap(params, d, b->terminal_obs, b->truncated + (int64_t)w0 * N, (int64_t)(t - w0 + 1) * N, gamma,
                                          b->rewards + (int64_t)w0 * N, stream);
                if (rc) return rc;
                w0 = t + 1;
            }
        }
    }
    if (compute_last_values && t_end == T) {
        if (!b->last_values) return fail(TMA_ERR_INVALID, "last_values is null");
        int rc = tma_policy_values(params, d, b->obs + (int64_t)T * N * D, N, b->last_values, stream);
        if (rc) return rc;
    }
    return TMA_OK;
}

#ifdef TMA_ROLL_TICKS
// diagnostic build only: read (reset != 0: clear) the per-phase cycle sums of the f32 wide rollout kernel
extern "C" int tma_debug_roll_ticks(unsigned long long *out8, int reset) {
    if (reset) {
        unsigned long long z[8] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(tma::g_roll_ticks), z, sizeof(z)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(tma::g_roll_ticks), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : 1;
}
#endif

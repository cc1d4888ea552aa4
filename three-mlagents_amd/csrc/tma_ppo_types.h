// tma_ppo_types.h -- types and workspace layout shared by the translation units of the PPO update (tma_policy.hip, tma_h64.hip).
#pragma once
#include "tma_mlp.h"

namespace tma {

constexpr int WS_ADV = 0;             // float[2]: minibatch advantage mean, std
constexpr int WS_NORM_PART = 64;      // byte offset of double[256] grad sum-of-squares partials
constexpr int WS_NORM_OUT = 64 + 256 * 8;  // double[2]: total grad norm, clip coefficient
constexpr int WS_PERSIST_ERR = 2176;  // int32: set when the persistent epoch kernel (tma_h64p.hip) gave up on a wait; read + cleared by tma_ppo_pop_stats
constexpr int WS_PERSIST_SNAP = 2560;  // double[8][8]: the persistent epoch launch's statistic slots before the launch (restored on fallback)
constexpr int WS_ADV_PART = 4096;     // byte offset of double[128][2] advantage (sum, sumsq) partials
constexpr int WS_STATS = 8192;        // byte offset of double[MAX_GRAD_BLOCKS][8] loss statistic slots
constexpr int MAX_GRAD_BLOCKS = 2048;
constexpr int64_t WS_SLABS = WS_STATS + (int64_t)MAX_GRAD_BLOCKS * 8 * 8;  // byte offset of float[H64_BLOCKS][P] partial-gradient slabs
constexpr int H64_BLOCKS = 128;  // block PAIRS (policy block + value block): 256 blocks = one per CU, a single round
constexpr int64_t WS_BYTES = WS_SLABS;
constexpr int BF_SLABS = 160;  // column-parallel kernels: up to 160 policy-net blocks (+ value-net blocks sharing the first slabs)
constexpr int64_t OFFS_CAP = 1 << 22;
constexpr int64_t EPOCH_PART_BYTES = ((OFFS_CAP / 1024) + (OFFS_CAP / 256)) * 16;  // advantage partials of every minibatch of an epoch
constexpr int WIDE_SQ_SLOTS = 8192;  // sum-of-squares partials of slab_reduce_kernel for policies beyond the 256 slots at WS_NORM_PART  // sample offsets of one minibatch cached behind the slabs (int32 each) when count <= OFFS_CAP

constexpr int64_t DZ1_CAP = 1 << 18;  // samples per minibatch whose dz1 images fit the workspace cache (bf16 two-pass layouts only)
// (round 6: ... and 97 .. 128 observations with a Box head at H = 256 -- Ant-v5's 105 -- as four layer-1 k-steps)
static inline bool bf_two_pass(const PLayout &L) {
    return L.bf16 && ((L.D > 32 && L.D <= 64) || (L.D > 160 && L.D <= 192) || (L.D > 96 && L.D <= 128 && L.cont && L.H == 256));
}
// (round 6: ... and 97 .. 112 observations -- the reference's ant task, Ant-v5's 105 -- with Box heads at H = 256: seven k-tiles)
static inline bool f32_two_pass(const PLayout &L) {
    return !L.bf16 && L.fr_pi >= 0 && ((L.D > 160 && L.D <= 176) || (L.D > 96 && L.D <= 112 && L.cont && L.H == 256));
}
static inline int64_t dz1_cache_bytes(const PLayout &L) {  // both nets; bf16 images or f32 MFMA operands
    return bf_two_pass(L) ? 2 * DZ1_CAP * L.H * 2 : (f32_two_pass(L) ? 2 * DZ1_CAP * L.H * 4 : 0);
}
// floats per packed sample record {obs padded to a multiple of 4 | log_prob, advantage, action bits, return}; 0: shape without them
static inline int rec_floats(const PLayout &L) { return (L.img_pi >= 0 && L.D <= 8) ? ((L.D + 3) & ~3) + 4 : 0; }
static inline int slab_cap(const PLayout &L) { return (L.bf16 || L.fr_pi >= 0) ? BF_SLABS : H64_BLOCKS; }  // partial-gradient slabs in the workspace

struct Minibatch {
    const int64_t *indices;  // optional explicit flat (env-major: f = i*T + t) indices
    uint32_t perm_seed, perm_epoch;
    int64_t start, count, total;  // rows [start, start+count) of the permuted buffer of `total` samples
    const int32_t *offs;          // optional: offs[j] = buffer offset of minibatch row j (written by adv_partial_kernel), saves the
                                  // permutation arithmetic in the gradient kernel
    int64_t stats_n;              // rows the advantage partials were summed over: count, or the global minibatch under data parallelism
    const double *adv_part;       // (sum, sum of squares) partials of this minibatch's advantages, adv_n_part pairs: kernels that fold the
    int adv_n_part;               // statistics themselves (H = 64, bf16 wide) read them; the others get (mean, std) from adv_final_kernel
};

__device__ __forceinline__ int64_t sample_offset(const Minibatch &mb, int64_t j, int T, int64_t N) {
    const int64_t f = mb.indices ? mb.indices[j] : (int64_t)perm_index(mb.perm_seed, mb.perm_epoch, (uint32_t)j, (uint32_t)mb.total);
    const int64_t i = f / T, t = f - i * T;  // swap_and_flatten: (T, N) -> env-major
    return t * N + i;
}

struct Rollout {
    const float *obs;
    const void *actions;
    const float *log_probs, *advantages, *returns;
    int T;
    int64_t N;
    const float *packed;  // optional sample records (tma_ppo_pack_samples), rec_floats(L) floats per sample
};
struct HParams {
    float clip_range, ent_coef, vf_coef;
    int normalize_advantage;
    int debug;  // TMA_BF_DEBUG (profiling aid, default 0): bit mask of phases the bf16 wide kernel skips -- timing attribution only
};

// A pending optimizer step applied in the PROLOGUE of the next H = 64 gradient launch (tma_ppo_train_epoch_local, round 3): instead of a
// separate clip + Adam launch between two minibatches, every workgroup of the next gradient kernel redoes the step for ITS net's 4 675
// parameters (adam_update_h64: the same routine, the same inputs, hence the same bits in every workgroup and as the optimizer launch) and
// builds its LDS weight image from the results instead of staging it from memory; workgroup 0 of each net writes the new state to the
// other half of a double buffer (the running launch still reads the old half).  grad == nullptr: no pending step, stage the image.
struct AdamFold {
    const float *grad;       // reduced gradient of the previous minibatch [P] (slab_reduce_kernel, overwrite mode)
    const double *sq_part;   // its sum-of-squares partials, one per 64 parameters
    int n_part;
    const float *p_cur, *m_cur, *v_cur;  // trainable parameters / Adam moments before the step
    float *p_nxt, *m_nxt, *v_nxt;        // ... and after it
    float max_norm, lr_step, beta1, beta2, bc2_sqrt, eps;  // (bc2_sqrt = sqrt(1 - beta2^t), lr_step = lr / (1 - beta1^t))
    double *norm_out;        // [2] total gradient norm, clip coefficient (statistics)
    float scale;             // factor on the gradient before the clip (1 on one GPU; 1 / world on the all-reduced SUM, tma_ppo_train_epoch_dp)
};

struct Net {
    const float *W1t, *b1, *W2t, *b2, *W3t, *b3, *W2, *W3;
};
__device__ __forceinline__ Net pi_net(const float *p, const PLayout &L) {
    return Net{p + L.pW1t, p + L.pb1, p + L.pW2t, p + L.pb2, p + L.pW3t, p + L.pb3, p + L.pW2, p + L.pW3};
}
__device__ __forceinline__ Net vf_net(const float *p, const PLayout &L) {
    return Net{p + L.vW1t, p + L.vb1, p + L.vW2t, p + L.vb2, p + L.vW3t, p + L.vb3, p + L.vW2, p + L.vW3};
}


struct LossStats {
    double a = 0.0, ent = 0.0, kl = 0.0, clip = 0.0, n = 0.0;
};

// clipped-surrogate + entropy gradient wrt the head outputs of one 16-row tile (C layout), written to dz3[16][ld3]
// RLO / RHI >= 0: the row range is a compile-time constant -- the rows of the range then form ONE basic block, and the scheduler interleaves
// their dependent chains (max / exp / sum / log / exp over DPP reductions: ~140 instructions a row, each waiting for the one before).  With
// the range in registers every row is a block of its own behind a wave-uniform branch and the chains run one after the other.
template <bool CONT, int RLO = -1, int RHI = -1>
__device__ __forceinline__ void policy_loss_tile(const f32x4 (&acc)[CONT ? 2 : 1], const float *meta, const int64_t *row_off, const void *actions,
                                                 const float *log_std, int A, float amean, float astd, const HParams &hp, float invB, float *dz3,
                                                 int ld3, float (&dlsd)[2], LossStats &st, int lane, int r_lo = 0, int r_hi = 4,
                                                 const float *act_tile = nullptr /* CONT: the tile's actions [16][32] in LDS (gathered with the observation rows, a phase ahead) instead of a global load here */) {
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if constexpr (RLO >= 0) {
            if (r < RLO || r >= RHI) continue;
        } else {
            if (r < r_lo || r >= r_hi) continue;  // (wave-uniform) a caller may split the four rows of a lane group over two waves
        }
        const int row = g * 4 + r;
        const int64_t off = row_off[row];
        const bool valid = off >= 0;
        const float old = meta[row * 4 + 0];
        const float advn = (meta[row * 4 + 1] - amean) / (astd + 1e-8f);
        float lpa, ent;
        float d[2] = {0.0f, 0.0f}, sd[2] = {1.0f, 1.0f}, p = 0.0f, lp = 0.0f;
        int act = 0;
        if constexpr (!CONT) {
            const bool colok = r16 < A;
            const float x = colok ? acc[0][r] : -INFINITY;
            const float m = gmax16(x);
            const float e = colok ? expf(x - m) : 0.0f;
            const float s = gsum16(e);
            const float lse = m + logf(s);
            lp = colok ? x - lse : 0.0f;
            p = e / s;
            act = __float_as_int(meta[row * 4 + 3]);
            lpa = gsum16((r16 == act) ? lp : 0.0f);
            ent = -gsum16(p * lp);
        } else {
            float lpsum = 0.0f, entsum = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = 16 * j + r16;
                if (col < A) {
                    const float lsd = log_std[col];
                    sd[j] = expf(lsd);
                    const float a = valid ? (act_tile ? act_tile[row * 32 + col] : static_cast<const float *>(actions)[off * A + col]) : 0.0f;
                    d[j] = a - acc[j][r];
                    lpsum += -(d[j] * d[j]) / (2.0f * (sd[j] * sd[j])) - lsd - 0.9189385332046727f;
                    entsum += 1.4189385332046727f + lsd;
                }
            }
            lpa = gsum16(lpsum);
            ent = gsum16(entsum);
        }
        const float ratio = expf(lpa - old);
        const float pl1 = advn * ratio;
        const float rc = fminf(fmaxf(ratio, 1.0f - hp.clip_range), 1.0f + hp.clip_range);
        const float pl2 = advn * rc;
        const float g_lp = (valid && pl1 <= pl2) ? -(advn * ratio) * invB : 0.0f;
        if constexpr (!CONT) {
            float dl = g_lp * (((r16 == act) ? 1.0f : 0.0f) - p);
            dl += valid ? (hp.ent_coef * invB) * (p * (lp + ent)) : 0.0f;
            dz3[row * ld3 + r16] = (r16 < A) ? dl : 0.0f;
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = 16 * j + r16;
                const float var = sd[j] * sd[j];
                dz3[row * ld3 + col] = (col < A) ? g_lp * (d[j] / var) : 0.0f;
                if (col < A) dlsd[j] += g_lp * ((d[j] * d[j]) / var - 1.0f) - (valid ? hp.ent_coef * invB : 0.0f);
            }
        }
        // (selects instead of a divergent branch: adding +0.0 leaves a sum as it is, and the row stays one basic block)
        const bool on = valid && r16 == 0;
        st.a += on ? (double)(-fminf(pl1, pl2)) : 0.0;
        st.ent += on ? (double)ent : 0.0;
        st.kl += on ? (double)((ratio - 1.0f) - (lpa - old)) : 0.0;
        st.clip += (on && fabsf(ratio - 1.0f) > hp.clip_range) ? 1.0 : 0.0;
        st.n += on ? 1.0 : 0.0;
    }
}

// KT1C: k-tiles of dW1 kept in registers (1: D <= 16, 2: D <= 32), 0: layer-1 gradient accumulated in the slab, -1: dW1 skipped
// (first of two passes).  PASS 1 = second pass for wide observations of known width: the forward / backward chain is recomputed and
// ONLY dW1 (KT1C k-tiles) is accumulated and stored -- every other store is compiled out, so the MFMAs feeding only them vanish.
// NQ1C > 0: layer-1 weights (16 * NQ1C >= D rows) also run through the ring, from the fragment image PLayout::fr1_pi.
}  // namespace tma

// tma_h64.hip: the H = 64 persistent gradient kernel (internal, not part of the C ABI)
int tma_launch_grad_h64(const float *params, const tma::PLayout &L, const tma::Rollout &R, const tma::Minibatch &M, const tma::HParams &hpar,
                        const double *adv_part, int n_part, float *slabs, double *slots, int *n_slabs_out, hipStream_t s,
                        const tma::AdamFold *fold = nullptr);

// tma_h64p.hip: one whole epoch at batch_size = 256 as a single persistent launch (H = 64 fast-path layouts)
bool tma_epoch_h64p_eligible(const tma::PLayout &L, int64_t batch_size, int64_t total);
int tma_launch_epoch_h64p(float *params, const tma::PLayout &L, const tma::Rollout &R, const tma::HParams &hp, const int32_t *offs,
                          const double *adv_part, int adv_stride, int64_t total, int64_t batch_size, float *exp_avg, float *exp_avg_sq,
                          int64_t first_step, double lr, double beta1, double beta2, double eps, double max_grad_norm, char *ws, hipStream_t s);

// tma_h256p.hip: the same for the reference's default 256 x 256 policy (exact f32, Discrete heads, observations <= 32): 2 x 32 workgroups on two XCDs
bool tma_epoch_h256p_eligible(const tma::PLayout &L, int64_t batch_size, int64_t total);
int tma_launch_epoch_h256p(float *params, const tma::PLayout &L, const tma::Rollout &R, const tma::HParams &hp, const int32_t *offs,
                           const double *adv_part, int adv_stride, int64_t total, int64_t batch_size, float *exp_avg, float *exp_avg_sq,
                           int64_t first_step, double lr, double beta1, double beta2, double eps, double max_grad_norm, char *ws, hipStream_t s);

// tma_bf16.hip: the column-parallel bf16-MFMA gradient kernel (hidden 128 / 192 / 256); `ws` is the update workspace (dz1 cache)
int tma_launch_grad_wide_bf(const float *params, const tma::PLayout &L, const tma::Rollout &R, const tma::Minibatch &M, const tma::HParams &hpar,
                            const float *ws_adv, float *slabs, double *slots, char *ws, int *n_pi_out, int *n_vf_out, hipStream_t s);
// tma_bf16.hip: the three-term bf16 split of the 256-wide f32 update (mfma_dtype = 2; tma_split3.h) and the rebuild of its weight planes
int tma_launch_grad_split3(const float *params, const tma::PLayout &L, const tma::Rollout &R, const tma::Minibatch &M, const tma::HParams &hpar, float *slabs,
                           double *slots, int *n_pi_out, int *n_vf_out, hipStream_t s);
int tma_launch_build_split3(float *params, const tma::PLayout &L, hipStream_t s);
bool tma_split3_eligible(const tma::PLayout &L, int64_t count);
// tma_policy.hip: zero the layer-1 weight columns of every slab (layouts that accumulate dW1 in place)
int tma_launch_slab_zero_w1(float *slabs, int n_slabs, const tma::PLayout &L, hipStream_t s);

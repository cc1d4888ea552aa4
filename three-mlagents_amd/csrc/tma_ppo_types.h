// tma_ppo_types.h -- types and workspace layout shared by the translation units of the PPO update (tma_policy.hip, tma_h64.hip).
#pragma once
#include "tma_mlp.h"

namespace tma {

constexpr int WS_ADV = 0;             // float[2]: minibatch advantage mean, std
constexpr int WS_NORM_PART = 64;      // byte offset of double[256] grad sum-of-squares partials
constexpr int WS_NORM_OUT = 64 + 256 * 8;  // double[2]: total grad norm, clip coefficient
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
static inline bool bf_two_pass(const PLayout &L) { return L.bf16 && ((L.D > 32 && L.D <= 64) || (L.D > 160 && L.D <= 192)); }
static inline bool f32_two_pass(const PLayout &L) { return !L.bf16 && L.fr_pi >= 0 && L.D > 160 && L.D <= 176; }
static inline int64_t dz1_cache_bytes(const PLayout &L) {  // both nets; bf16 images or f32 MFMA operands
    return bf_two_pass(L) ? 2 * DZ1_CAP * L.H * 2 : (f32_two_pass(L) ? 2 * DZ1_CAP * L.H * 4 : 0);
}
static inline int slab_cap(const PLayout &L) { return (L.bf16 || L.fr_pi >= 0) ? BF_SLABS : H64_BLOCKS; }  // partial-gradient slabs in the workspace

struct Minibatch {
    const int64_t *indices;  // optional explicit flat (env-major: f = i*T + t) indices
    uint32_t perm_seed, perm_epoch;
    int64_t start, count, total;  // rows [start, start+count) of the permuted buffer of `total` samples
    const int32_t *offs;          // optional: offs[j] = buffer offset of minibatch row j (written by adv_partial_kernel), saves the
                                  // permutation arithmetic in the gradient kernel
    int64_t stats_n;              // rows the advantage partials were summed over: count, or the global minibatch under data parallelism
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
};
struct HParams {
    float clip_range, ent_coef, vf_coef;
    int normalize_advantage;
    int debug;  // TMA_BF_DEBUG (profiling aid, default 0): bit mask of phases the bf16 wide kernel skips -- timing attribution only
};

}  // namespace tma

// tma_h64.hip: the H = 64 persistent gradient kernel (internal, not part of the C ABI)
int tma_launch_grad_h64(const float *params, const tma::PLayout &L, const tma::Rollout &R, const tma::Minibatch &M, const tma::HParams &hpar,
                        const double *adv_part, int n_part, float *slabs, double *slots, int *n_slabs_out, hipStream_t s);

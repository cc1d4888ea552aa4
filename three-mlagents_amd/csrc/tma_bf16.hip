// tma_bf16.hip -- PPO minibatch gradient for 128 / 192 / 256-wide policies on the bf16 MFMA (tma_policy_dims.mfma_dtype = 1;
// BASELINE.json configs[2] "PPO MLP(256,256) bf16").  The kernel body lives in tma_wide_bf16.h (shared with the forward kernels of
// tma_policy.hip); this translation unit instantiates the gradient kernel and picks the variant for a policy shape.
//
// Replaces, for the reference's default net_arch (/root/reference/backend/mlagents/training.py:363-365), what SB3's PPO.train computes
// per minibatch between RolloutBuffer.get and optimizer.step (third party).
#include "tma_ppo_types.h"

#include <cstdlib>
#include <type_traits>

namespace tma {
#ifdef TMA_BF_PHASE_TICKS
__device__ unsigned long long g_bf_ticks[2][16];
#endif
#include "tma_wide_bf16.h"
}  // namespace tma

#ifdef TMA_BF_PHASE_TICKS
// diagnostic build only: read (reset != 0: clear) the per-phase cycle sums of the bf16 gradient kernel
extern "C" int tma_debug_bf_ticks(unsigned long long *out32, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(tma::g_bf_ticks), z, sizeof(z)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(tma::g_bf_ticks), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

using namespace tma;

int tma_launch_grad_wide_bf(const float *params, const PLayout &L, const Rollout &R, const Minibatch &M, const HParams &hpar, const float *ws_adv,
                            float *slabs, double *slots, char *ws, int *n_pi_out, int *n_vf_out, hipStream_t s) {
        // (variant 5, round 6: 97..128 observations with a Box head at H = 256 -- the reference's ant task -- on the two-pass layout with four k-steps)
        const int variant = L.D <= 16 ? 0 : (L.D <= 32 ? 1 : (L.D <= 64 ? 2 : ((L.D > 160 && L.D <= 192) ? 3 : ((L.D > 96 && L.D <= 128 && L.cont && L.H == 256) ? 5 : 4))));
        // observations of up to 32 floats: 64-row groups (half the weight bytes, barriers and latency chains per sample); wider ones keep
        // 32-row groups (their observation images would not fit next to 64-row activation images)
        static const bool force_mt2 = getenv("TMA_BF_MT2") != nullptr;  // development switch: the 32-row-group kernel
        // ... and minibatches too small to give every block a 64-row group (the reference's literal batch_size = 256: 4 groups per net) take
        // 32-row groups as well: twice the workgroups on a launch that is all latency
        const int MTc = (variant <= 1 && !force_mt2 && L.H != 192 && M.count > 4096) ? 4 : 2;  // (H = 192: three column tiles per wave do not split in halves)
        const int smemw = grad_wide_bf_smem_bytes(L.D, L.H, MTc);
        // 256 blocks = one per CU.  A policy-net row group costs 1.15-1.3x a value-net one (the loss), so the policy net gets
        // 136 or 144 of the blocks; with fewer row groups than that, one block per group.
        const int64_t groups = ceil_div(M.count, 16 * MTc);
        static const int npi_env = getenv("TMA_BF_NPI") ? atoi(getenv("TMA_BF_NPI")) : 0;  // development switch: policy-net block count
        const bool eight = getenv("TMA_BF_NW4") == nullptr && MTc == 4 && variant == 0 && !L.cont && L.H == 256;
        // (Categorical loss is cheaper than the DiagGaussian one; the eight-wave kernel shares a tile's loss between two waves: swept 124 .. 160, best 136 .. 140)
        const int cap_pi = npi_env > 0 ? npi_env : (L.cont ? 144 : (eight ? 140 : 136)), cap_vf = 256 - cap_pi;
        const int n_pi = (int)(groups < cap_pi ? groups : cap_pi), n_vf = (int)(groups < cap_vf ? groups : cap_vf);
        static const bool nw8_early = getenv("TMA_BF_NW4") == nullptr;
        if (variant == 4 || (variant == 5 && !nw8_early)) {  // runtime observation width: dW1 accumulates in place in the slab
            const int zrc = tma_launch_slab_zero_w1(slabs, n_pi, L, s);
            if (zrc) return zrc;
        }
        auto launch = [&](auto k, bf16_t *dz1) -> int {
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smemw));
            k<<<dim3((unsigned)(n_pi + n_vf)), dim3(256), smemw, s>>>(params, L, R, M, hpar, ws_adv, slabs, slots, n_pi, dz1, DZ1_CAP * L.H);
            return TMA_OK;
        };
        // eight-wave variant (two waves per SIMD, 32 columns each; tma_wide_bf16.h): 64-row groups of the Discrete layouts with observations of up to
        // 16 floats at H = 256 -- GridWorld, Push, Ball3D, WallJump (measured round 4, one box: 285 -> 246-255 us per 131 072 samples; TMA_BF_NW4=1
        // selects the four-wave kernel)
        static const bool nw8 = getenv("TMA_BF_NW4") == nullptr;
        if (nw8 && MTc == 4 && variant == 0 && !L.cont && L.H == 256) {
            const int smem8 = smemw + (L.H / 32) * 1024 + 12 * 4 * 5 * 8;  // + the head fragments + the statistics slots of the row-lane loss (64 in all)
            auto launch8 = [&](auto k) -> int {
                TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem8));
                k<<<dim3((unsigned)(n_pi + n_vf)), dim3(512), smem8, s>>>(params, L, R, M, hpar, ws_adv, slabs, slots, n_pi, nullptr, DZ1_CAP * L.H);
                return TMA_OK;
            };
            const int rc8 = launch8(ppo_grad_wide_bf_kernel<false, 2, 4, 1, 1, 0, 8>);
            if (rc8) return rc8;
            TMA_LAUNCH_CHECK();
            *n_pi_out = n_pi, *n_vf_out = n_vf;
            return TMA_OK;
        }
        // ... and the Crawler width (Box head, 161 .. 192 observations, two launches): 32-row groups on eight waves
        if (nw8 && variant == 5) {  // the Ant width: the same two launches with four layer-1 k-steps
            bf16_t *const dz1c8 = (bf_two_pass(L) && M.count <= DZ1_CAP && !getenv("TMA_NO_DZ1_CACHE"))
                ? reinterpret_cast<bf16_t *>(ws + WS_SLABS + (int64_t)slab_cap(L) * L.P * 4 + OFFS_CAP * 4 + EPOCH_PART_BYTES + WIDE_SQ_SLOTS * 8) : nullptr;
            const int smem8 = smemw + 4 * 4 * 5 * 8;
            auto launch8 = [&](auto k, bf16_t *dz1) -> int {
                TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem8));
                k<<<dim3((unsigned)(n_pi + n_vf)), dim3(512), smem8, s>>>(params, L, R, M, hpar, ws_adv, slabs, slots, n_pi, dz1, DZ1_CAP * L.H);
                return TMA_OK;
            };
            int rc8 = launch8(ppo_grad_wide_bf_kernel<true, 2, 2, 0, 4, 0, 8>, dz1c8);
            if (rc8) return rc8;
            rc8 = dz1c8 ? launch8(ppo_grad_wide_bf_kernel<true, 2, 2, 0, 4, 2, 8>, dz1c8) : launch8(ppo_grad_wide_bf_kernel<true, 2, 2, 0, 4, 1, 8>, nullptr);
            if (rc8) return rc8;
            TMA_LAUNCH_CHECK();
            *n_pi_out = n_pi, *n_vf_out = n_vf;
            return TMA_OK;
        }
        if (nw8 && variant == 3 && L.cont && L.H == 256) {
            bf16_t *const dz1c8 = (bf_two_pass(L) && M.count <= DZ1_CAP && !getenv("TMA_NO_DZ1_CACHE"))
                ? reinterpret_cast<bf16_t *>(ws + WS_SLABS + (int64_t)slab_cap(L) * L.P * 4 + OFFS_CAP * 4 + EPOCH_PART_BYTES + WIDE_SQ_SLOTS * 8) : nullptr;
            const int smem8 = smemw + 4 * 4 * 5 * 8;  // + the statistics of four more waves
            auto launch8 = [&](auto k, bf16_t *dz1) -> int {
                TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem8));
                k<<<dim3((unsigned)(n_pi + n_vf)), dim3(512), smem8, s>>>(params, L, R, M, hpar, ws_adv, slabs, slots, n_pi, dz1, DZ1_CAP * L.H);
                return TMA_OK;
            };
            int rc8 = launch8(ppo_grad_wide_bf_kernel<true, 2, 2, 0, 6, 0, 8>, dz1c8);
            if (rc8) return rc8;
            rc8 = dz1c8 ? launch8(ppo_grad_wide_bf_kernel<true, 2, 2, 0, 6, 2, 8>, dz1c8) : launch8(ppo_grad_wide_bf_kernel<true, 2, 2, 0, 6, 1, 8>, nullptr);
            if (rc8) return rc8;
            TMA_LAUNCH_CHECK();
            *n_pi_out = n_pi, *n_vf_out = n_vf;
            return TMA_OK;
        }
        // two-pass layouts: minibatches that fit the dz1 cache take PASS 0 (which leaves dz1 there) + PASS 2 (dW1 from the cache)
        // instead of PASS 0 + PASS 1 (dW1 from a recomputed forward / backward chain); the results are bit-identical
        bf16_t *const dz1_cache = (bf_two_pass(L) && M.count <= DZ1_CAP && !getenv("TMA_NO_DZ1_CACHE"))  // (env: test hook for the fallback)
            ? reinterpret_cast<bf16_t *>(ws + WS_SLABS + (int64_t)slab_cap(L) * L.P * 4 + OFFS_CAP * 4 + EPOCH_PART_BYTES + WIDE_SQ_SLOTS * 8) : nullptr;
        // (KT1C, KS1C): D <= 16 -> (1, 1); D <= 32 -> (2, 1); D <= 64 -> (0, 2) two passes; 161..192 (Crawler's 172) -> (0, 6) two
        // passes; else runtime width, one pass with dW1 in the slab
        auto pick = [&](auto ntw) -> int {
            constexpr int NTWc = decltype(ntw)::value;
            constexpr int MT4 = NTWc % 2 == 0 ? 4 : 2;  // (never launched with MTc == 4 when odd)
            auto both = [&](auto cont) -> int {
                constexpr bool C = decltype(cont)::value;
                switch (variant) {
                    case 0: return MTc == 4 ? launch(ppo_grad_wide_bf_kernel<C, NTWc, MT4, 1, 1, 0>, nullptr) : launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 1, 1, 0>, nullptr);
                    case 1: return MTc == 4 ? launch(ppo_grad_wide_bf_kernel<C, NTWc, MT4, 2, 1, 0>, nullptr) : launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 2, 1, 0>, nullptr);
                    case 2: {
                        const int rc2 = launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 0, 2, 0>, dz1_cache);
                        if (rc2) return rc2;
                        return dz1_cache ? launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 0, 2, 2>, dz1_cache)
                                         : launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 0, 2, 1>, nullptr);
                    }
                    case 3: {
                        const int rc2 = launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 0, 6, 0>, dz1_cache);
                        if (rc2) return rc2;
                        return dz1_cache ? launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 0, 6, 2>, dz1_cache)
                                         : launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 0, 6, 1>, nullptr);
                    }
                    default: return launch(ppo_grad_wide_bf_kernel<C, NTWc, 2, 0, 0, 0>, nullptr);  // (also variant 5 with TMA_BF_NW4=1)
                }
            };
#ifdef TMA_BF_DEV  // development builds: discrete heads only (a third of the instantiations)
            return both(std::false_type{});
#else
            return L.cont ? both(std::true_type{}) : both(std::false_type{});
#endif
        };
        int lrc;
        {
#ifdef TMA_BF_DEV
            lrc = pick(std::integral_constant<int, 4>{});
#else
            lrc = L.H == 256 ? pick(std::integral_constant<int, 4>{}) : (L.H == 192 ? pick(std::integral_constant<int, 3>{}) : pick(std::integral_constant<int, 2>{}));
#endif
        }
        if (lrc) return lrc;
        TMA_LAUNCH_CHECK();
        *n_pi_out = n_pi, *n_vf_out = n_vf;
        return TMA_OK;
}


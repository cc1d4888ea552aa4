// tma_split3.hip -- mfma_dtype = 2: the 256-wide f32 PPO update on the bf16 MFMA, every operand as three bf16 terms (tma_split3.h).
// A translation unit of its own: the kernel is opt-in and new (round 5), the shipped bf16 kernels keep their object file and flags.
#include "tma_ppo_types.h"

#include <cstdlib>
#include <type_traits>

namespace tma {
#ifdef TMA_S3_TICKS
__device__ unsigned long long g_s3_ticks[2][16];
#endif
#include "tma_wide_bf16.h"
#include "tma_split3.h"
}  // namespace tma

#ifdef TMA_S3_TICKS
extern "C" int tma_debug_s3_ticks(unsigned long long *out32, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(tma::g_s3_ticks), z, sizeof(z)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(tma::g_s3_ticks), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

using namespace tma;


// ---- mfma_dtype = 2: three-term bf16 split of the f32 update (tma_split3.h) ----
bool tma_split3_eligible(const PLayout &L, int64_t count) {
    // Discrete head, observations of up to 32 floats, H = 256; minibatches that give every block at least a few 32-row groups (smaller ones
    // stay on the exact-f32 kernel, whose half-group path is built for them)
    return L.split && !L.cont && L.H == 256 && L.D <= 32 && L.A <= 16 && count >= 4096 && getenv("TMA_NO_SPLIT3") == nullptr;
}

int tma_launch_build_split3(float *params, const PLayout &L, hipStream_t s) {
    build_split3_images_kernel<<<dim3(256), dim3(256), 0, s>>>(params, L);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

int tma_launch_grad_split3(const float *params, const PLayout &L, const Rollout &R, const Minibatch &M, const HParams &hpar, float *slabs, double *slots,
                           int *n_pi_out, int *n_vf_out, hipStream_t s) {
    const int64_t groups = ceil_div(M.count, 32);
    static const int npi_env = getenv("TMA_S3_NPI") ? atoi(getenv("TMA_S3_NPI")) : 0;  // development switch: policy-net block count
    const int cap_pi = npi_env > 0 ? npi_env : 128, cap_vf = 256 - cap_pi;  // (swept 124 .. 140 at 131 072 samples: 686 us at 128, 688 at 132, 701 at 136, 720 at 124 / 140)
    const int n_pi = (int)(groups < cap_pi ? groups : cap_pi), n_vf = (int)(groups < cap_vf ? groups : cap_vf);
    const int smem = grad_split3_smem_bytes();
    // one wave per SIMD (64 columns, 512 registers: activation fragments a k-step ahead, 10 spilled registers) measured 692 us per 131 072 samples
    // against 762 us for two per SIMD (32 columns, 256 registers: no room for the prefetch, ~180 spilled); TMA_S3_NW8=1 selects the latter (A/B)
    static const bool nw4 = getenv("TMA_S3_NW8") == nullptr;
    int rc;
    if (nw4) {
        auto launch4 = [&](auto k) -> int {
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            k<<<dim3((unsigned)(n_pi + n_vf)), dim3(256), smem, s>>>(params, L, R, M, hpar, slabs, slots, n_pi);
            return TMA_OK;
        };
        rc = L.D <= 16 ? launch4(ppo_grad_split3_kernel<1, 4>) : launch4(ppo_grad_split3_kernel<2, 4>);
    } else {
        auto launch8 = [&](auto k) -> int {
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            k<<<dim3((unsigned)(n_pi + n_vf)), dim3(512), smem, s>>>(params, L, R, M, hpar, slabs, slots, n_pi);
            return TMA_OK;
        };
        rc = L.D <= 16 ? launch8(ppo_grad_split3_kernel<1, 8>) : launch8(ppo_grad_split3_kernel<2, 8>);
    }
    if (rc) return rc;
    TMA_LAUNCH_CHECK();
    *n_pi_out = n_pi, *n_vf_out = n_vf;
    return TMA_OK;
}

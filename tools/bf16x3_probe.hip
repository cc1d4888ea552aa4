// bf16x3_probe.hip -- would a three-term bf16 split beat the exact-f32 MFMA on the H x H phases of the f32 256-wide gradient kernel?  
//
// The reference's default policy is MLP(256, 256) in fp32 (backend/mlagents/training.py:363-365).  ppo_grad_wide_kernel runs its GEMMs on
// v_mfma_f32_16x16x4_f32 (157 TFLOP/s dense: 1/16 of the bf16 pipe) at 0.64 of that peak.  Splitting every f32 operand into three bf16 terms
// (x = hi + mid + lo, each the bf16 rounding of what the previous left) and keeping the six products of order <= 2
//     a.w ~= a_lo.w_hi + a_mid.w_mid + a_hi.w_lo + a_mid.w_hi + a_hi.w_mid + a_hi.w_hi      (f32 accumulate, smallest terms first)
// costs six v_mfma_f32_16x16x32_bf16 (16 cycles, 32 k each) where exact f32 costs eight v_mfma_f32_16x16x4_f32 (32 cycles, 4 k each): 96
// against 256 matrix-pipe cycles per 16 x 16 x 32 block, 2.67x -- on paper.  This probe rebuilds ONE phase both ways, stand-alone, with what a
// product kernel would have to carry: the layer-2 forward of a 32-row group on eight waves (two per SIMD, 32 columns each, K = N = 256),
// A operands from LDS (f32: k-interleaved so that four k-steps are one ds_read_b128, as the product's fragment images; split: three row-major
// bf16 planes), weights streamed from L2 through fragment-major images (f32: 16 B per lane per four k-steps; split: three planes), epilogue
// = bias + tanh + store of the next layer's A image (split: + the three-way split, 5 extra VALU instructions and two more stores per element).
// Prints microseconds per group for both and the largest error of each pre-activation against a float64 reference.
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/bf16x3_probe.hip -o tools/bin/bf16x3_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                  \
    do {                                                       \
        hipError_t e_ = (x);                                   \
        if (e_ != hipSuccess) {                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                           \
        }                                                      \
    } while (0)

constexpr int H = 256, M = 32, NW = 8, LDF = H + 4, LDB = H + 16;

__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// ---------------- exact f32: A image [row][g = k % 4][ks = k / 4] (row stride LDF), W image [ntile][ks4][lane] float4 = W[4 (4 ks4 + s) + g][16 ntile + r16]
__global__ __launch_bounds__(64 * NW, 1) void phase_f32(const float *__restrict__ Wimg, const float *__restrict__ bias, const float *__restrict__ A0, float *__restrict__ pre_out,
                                                         float *__restrict__ sink, int groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Ain = reinterpret_cast<float *>(smem), *Aout = Ain + M * LDF;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    for (int e = threadIdx.x; e < M * H; e += blockDim.x) {
        const int row = e / H, k = e % H;
        Ain[row * LDF + (k & 3) * 64 + (k >> 2)] = A0[e];
    }
    __syncthreads();
    const float4 *W4 = reinterpret_cast<const float4 *>(Wimg);
    float keep = 0.0f;
    for (int grp = 0; grp < groups; grp++) {
        f32x4 acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float b = bias[32 * wave + 16 * j + r16];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) acc[j][mt] = f32x4{b, b, b, b};
        }
        float4 w[2][2], a[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++) w[0][j] = W4[((2 * wave + j) * 16 + 0) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) a[0][mt] = *reinterpret_cast<const float4 *>(Ain + (16 * mt + r16) * LDF + g * 64);
#pragma unroll
        for (int ks4 = 0; ks4 < 16; ks4++) {
            const int cur = ks4 & 1;
            if (ks4 + 1 < 16) {
#pragma unroll
                for (int j = 0; j < 2; j++) w[cur ^ 1][j] = W4[((2 * wave + j) * 16 + ks4 + 1) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < 2; mt++) a[cur ^ 1][mt] = *reinterpret_cast<const float4 *>(Ain + (16 * mt + r16) * LDF + g * 64 + 4 * (ks4 + 1));
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const float av = s == 0 ? a[cur][mt].x : (s == 1 ? a[cur][mt].y : (s == 2 ? a[cur][mt].z : a[cur][mt].w));
                        const float wv = s == 0 ? w[cur][j].x : (s == 1 ? w[cur][j].y : (s == 2 ? w[cur][j].z : w[cur][j].w));
                        acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv, acc[j][mt], 0, 0, 0);
                    }
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * mt + 4 * g + r, n = 32 * wave + 16 * j + r16;
                    if (grp == 0 && pre_out && blockIdx.x == 0) pre_out[row * H + n] = acc[j][mt][r];
                    const float t = tanh_fast(acc[j][mt][r]);
                    Aout[row * LDF + (n & 3) * 64 + (n >> 2)] = t;  // the next layer's A image
                    keep += t;
                }
        __syncthreads();
    }
    if (keep == 123.456f) sink[threadIdx.x] = keep + Aout[threadIdx.x];
}

// ---------------- three-term bf16 split: A planes [p][row][LDB] row-major bf16, W planes fragment-major ([p][ntile][ks][lane] bf16x8 = W[32 ks + 8 g + i][16 ntile + r16])
__device__ __forceinline__ void split3(float x, bf16_t &h, bf16_t &m, bf16_t &l) {
    h = (bf16_t)x;
    const float r1 = x - (float)h;
    m = (bf16_t)r1;
    const float r2 = r1 - (float)m;
    l = (bf16_t)r2;
}
template <int TERMS>  // 6: all products of order <= 2; 3: hi.hi + hi.mid + mid.hi (order <= 1, ~2^-16 relative)
__global__ __launch_bounds__(64 * NW, 1) void phase_split(const bf16_t *__restrict__ Wimg, const float *__restrict__ bias, const float *__restrict__ A0, float *__restrict__ pre_out,
                                                           float *__restrict__ sink, int groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t *Ain = reinterpret_cast<bf16_t *>(smem), *Aout = Ain + 3 * M * LDB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    for (int e = threadIdx.x; e < M * H; e += blockDim.x) {
        const int row = e / H, k = e % H;
        bf16_t h, m, l;
        split3(A0[e], h, m, l);
        Ain[(0 * M + row) * LDB + k] = h, Ain[(1 * M + row) * LDB + k] = m, Ain[(2 * M + row) * LDB + k] = l;
    }
    __syncthreads();
    const bf16x8 *W8 = reinterpret_cast<const bf16x8 *>(Wimg);
    constexpr int PL = H / 16 * (H / 32) * 64;  // bf16x8 entries per weight plane
    float keep = 0.0f;
    for (int grp = 0; grp < groups; grp++) {
        f32x4 acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float b = bias[32 * wave + 16 * j + r16];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) acc[j][mt] = f32x4{b, b, b, b};
        }
        bf16x8 w[2][3][2], a[2][3][2];
        auto load = [&](int slot, int ks) {
#pragma unroll
            for (int p = 0; p < 3; p++) {
#pragma unroll
                for (int j = 0; j < 2; j++) w[slot][p][j] = W8[p * PL + ((2 * wave + j) * (H / 32) + ks) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < 2; mt++) a[slot][p][mt] = *reinterpret_cast<const bf16x8 *>(Ain + (p * M + 16 * mt + r16) * LDB + 32 * ks + 8 * g);
            }
        };
        load(0, 0);
#pragma unroll
        for (int ks = 0; ks < H / 32; ks++) {
            const int cur = ks & 1;
            if (ks + 1 < H / 32) load(cur ^ 1, ks + 1);
            // smallest terms first: (a plane, w plane)
            constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TW[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 6 - TERMS; t < 6; t++)
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cur][TA[t]][mt], w[cur][TW[t]][j], acc[j][mt], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * mt + 4 * g + r, n = 32 * wave + 16 * j + r16;
                    if (grp == 0 && pre_out && blockIdx.x == 0) pre_out[row * H + n] = acc[j][mt][r];
                    const float t = tanh_fast(acc[j][mt][r]);
                    bf16_t h, m, l;
                    split3(t, h, m, l);
                    Aout[(0 * M + row) * LDB + n] = h, Aout[(1 * M + row) * LDB + n] = m, Aout[(2 * M + row) * LDB + n] = l;
                    keep += t;
                }
        __syncthreads();
    }
    if (keep == 123.456f) sink[threadIdx.x] = keep + (float)Aout[threadIdx.x];
}

static float bf16_round(float x) {  // RNE to bf16, as v_cvt_pk_bf16_f32
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    float r;
    memcpy(&r, &u, 4);
    return r;
}
static uint16_t bf16_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)(u >> 16);
}

template <class F>
static double time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    srand(7);
    std::vector<float> W(H * H), A(M * H), b(H);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
    for (auto &x : W) x = rnd() * 0.11f;  // ~ orthogonal-init scale for H = 256
    for (auto &x : A) x = rnd();          // tanh outputs
    for (auto &x : b) x = rnd() * 0.1f;
    // f32 image: [ntile][ks4][lane] float4, element s = W[4 (4 ks4 + s) + g][16 ntile + r16]
    std::vector<float> Wf(H * H);
    for (int nt = 0; nt < H / 16; nt++)
        for (int ks4 = 0; ks4 < 16; ks4++)
            for (int l = 0; l < 64; l++)
                for (int s = 0; s < 4; s++) Wf[(((nt * 16 + ks4) * 64 + l) * 4) + s] = W[(4 * (4 * ks4 + s) + (l >> 4)) * H + 16 * nt + (l & 15)];
    // split planes: [p][ntile][ks][lane][8] bf16, element i = W_p[32 ks + 8 g + i][16 ntile + r16]
    std::vector<uint16_t> Ws(3 * H * H);
    for (int nt = 0; nt < H / 16; nt++)
        for (int ks = 0; ks < H / 32; ks++)
            for (int l = 0; l < 64; l++)
                for (int i = 0; i < 8; i++) {
                    const float x = W[(32 * ks + 8 * (l >> 4) + i) * H + 16 * nt + (l & 15)];
                    const float h = bf16_round(x), m = bf16_round(x - h), lo = bf16_round((x - h) - m);
                    const size_t o = (((size_t)nt * (H / 32) + ks) * 64 + l) * 8 + i;
                    Ws[0 * (size_t)H * H + o] = bf16_bits(h), Ws[1 * (size_t)H * H + o] = bf16_bits(m), Ws[2 * (size_t)H * H + o] = bf16_bits(lo);
                }
    std::vector<double> ref(M * H);
    double refmax = 0.0;
    for (int r = 0; r < M; r++)
        for (int n = 0; n < H; n++) {
            double s = b[n];
            for (int k = 0; k < H; k++) s += (double)A[r * H + k] * (double)W[k * H + n];
            ref[r * H + n] = s;
            refmax = std::max(refmax, std::fabs(s));
        }
    float *dWf, *dA, *db, *dpre, *dsink;
    uint16_t *dWs;
    CK(hipMalloc(&dWf, Wf.size() * 4));
    CK(hipMalloc(&dWs, Ws.size() * 2));
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&db, b.size() * 4));
    CK(hipMalloc(&dpre, M * H * 4));
    CK(hipMalloc(&dsink, 4096));
    CK(hipMemcpy(dWf, Wf.data(), Wf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWs, Ws.data(), Ws.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    const int smem_f = 2 * M * LDF * 4, smem_s = 2 * 3 * M * LDB * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(phase_f32), hipFuncAttributeMaxDynamicSharedMemorySize, smem_f));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(phase_split<6>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_s));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(phase_split<3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_s));
    auto err_of = [&](const char *name) {
        std::vector<float> pre(M * H);
        CK(hipMemcpy(pre.data(), dpre, pre.size() * 4, hipMemcpyDeviceToHost));
        double e = 0.0;
        for (int i = 0; i < M * H; i++) e = std::max(e, std::fabs((double)pre[i] - ref[i]));
        printf("%-46s max |pre-activation - float64 reference| = %.3e  (= %.2e of the largest output, 2^%.1f)\n", name, e, e / refmax, std::log2(e / refmax));
    };
    const int groups = 64, blocks = 256;
    phase_f32<<<blocks, 64 * NW, smem_f>>>(dWf, db, dA, dpre, dsink, 1);
    CK(hipDeviceSynchronize());
    err_of("exact f32 (v_mfma_f32_16x16x4_f32)");
    phase_split<6><<<blocks, 64 * NW, smem_s>>>(reinterpret_cast<const bf16_t *>(dWs), db, dA, dpre, dsink, 1);
    CK(hipDeviceSynchronize());
    err_of("bf16 x 3, six products (order <= 2)");
    phase_split<3><<<blocks, 64 * NW, smem_s>>>(reinterpret_cast<const bf16_t *>(dWs), db, dA, dpre, dsink, 1);
    CK(hipDeviceSynchronize());
    err_of("bf16 x 3, three products (order <= 1)");
    const double tf = time_us([&] { phase_f32<<<blocks, 64 * NW, smem_f>>>(dWf, db, dA, nullptr, dsink, groups); }, 9) / groups;
    const double t6 = time_us([&] { phase_split<6><<<blocks, 64 * NW, smem_s>>>(reinterpret_cast<const bf16_t *>(dWs), db, dA, nullptr, dsink, groups); }, 9) / groups;
    const double t3 = time_us([&] { phase_split<3><<<blocks, 64 * NW, smem_s>>>(reinterpret_cast<const bf16_t *>(dWs), db, dA, nullptr, dsink, groups); }, 9) / groups;
    const double flop = 2.0 * M * H * H * blocks;
    printf("layer-2 forward of a 32-row group, 8 waves, %d blocks x %d groups:\n", blocks, groups);
    printf("  exact f32               %.2f us per group  (%.1f TFLOP/s f32-equivalent; pipe floor 256 MFMAs x 32 cycles x 2 waves per SIMD)\n", tf, flop / tf * 1e-6);
    printf("  bf16 x 3, six products  %.2f us per group  (%.1f TFLOP/s f32-equivalent, x%.2f; pipe floor 192 MFMAs x 16 cycles x 2)\n", t6, flop / t6 * 1e-6, tf / t6);
    printf("  bf16 x 3, three         %.2f us per group  (x%.2f; ~2^-16 relative: not an f32-class product, shown for scale)\n", t3, tf / t3);
    return 0;
}

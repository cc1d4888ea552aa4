// Two questions about v_mfma_f32_4x4x1_16b_f32 on gfx950, answered by experiment (round 6, the 8-env rollout of the 256-wide f32 policy):
//  (1) with CBSZ = 4, ABID = e every one of the sixteen blocks takes its A operand from block e (lanes 4 e .. 4 e + 3): one instruction is then
//      4 rows (envs) x 64 columns x 1 k -- no padding rows for a tile of 4 or 8 envs;
//  (2) is a k-step of v_mfma_f32_16x16x4_f32 the four products added to the accumulator one after the other in k order (each a fused
//      multiply-add)?  Then 4 x 4 x 1 instructions issued k by k give the SAME BITS as the 16 x 16 x 4 chain of the other kernels.
//  Also: cycles per instruction of two and of four interleaved dependent 4 x 4 x 1 chains.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_bcast_probe tools/mfma_bcast_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 256;
__global__ void probe(const float *X /*[16][K], rows 8.. zero*/, const float *W /*[K][64]*/, const float *bias, float *out16 /*[16][64]*/, float *out4 /*[8][64]*/,
                      unsigned long long *cyc) {
    const int l = threadIdx.x, r16 = l & 15, g = l >> 4;
    // 16 x 16 x 4: four column tiles, k-steps of four
    for (int j = 0; j < 4; j++) {
        const float b = bias[16 * j + r16];
        f32x4 c = {b, b, b, b};
        for (int ks = 0; ks < K / 4; ks++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(X[r16 * K + 4 * ks + g], W[(4 * ks + g) * 64 + 16 * j + r16], c, 0, 0, 0);
        for (int i = 0; i < 4; i++) out16[(4 * g + i) * 64 + 16 * j + r16] = c[i];
    }
    // 4 x 4 x 1 with the A operand of block e broadcast: lanes 0 .. 7 hold envs 0 .. 7
    {
        const float b = bias[l];
        f32x4 c0 = {b, b, b, b}, c1 = c0;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < K; k++) {
            const float a = X[(l & 7) * K + k], w = W[k * 64 + l];
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c0, 4, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c1, 4, 1, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; i++) out4[i * 64 + l] = c0[i], out4[(4 + i) * 64 + l] = c1[i];
        if (l == 0) cyc[2] = t1 - t0;
    }
    // rates: two / four interleaved dependent chains, operands in registers
    float a = X[l & 7], w = W[l];
    f32x4 c[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    asm volatile("s_nop 0" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(a), "+v"(w));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+s"(t0));
#pragma unroll
    for (int it = 0; it < 128; it++) {
        c[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c[0], 4, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c[1], 4, 1, 0);
    }
    asm volatile("s_nop 0" : "+v"(c[0]), "+v"(c[1]));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+s"(t1));
#pragma unroll
    for (int it = 0; it < 64; it++) {
        c[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c[0], 4, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c[1], 4, 1, 0);
        c[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c[2], 4, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w, c[3], 4, 1, 0);
    }
    asm volatile("s_nop 0" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t2));
    if (l == 0) cyc[0] = t1 - t0, cyc[1] = t2 - t1;
    out4[8 * 64 + l] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}
int main() {
    float *hX = (float *)calloc(16 * K, 4), *hW = (float *)malloc(K * 64 * 4), hb[64];
    srand(7);
    for (int r = 0; r < 8; r++) for (int k = 0; k < K; k++) hX[r * K + k] = (float)rand() / RAND_MAX * 2 - 1;
    for (int i = 0; i < K * 64; i++) hW[i] = ((float)rand() / RAND_MAX * 2 - 1) * 0.1f;
    for (int i = 0; i < 64; i++) hb[i] = (float)rand() / RAND_MAX - 0.5f;
    float *dX, *dW, *db, *o16, *o4; unsigned long long *dc;
    hipMalloc(&dX, 16 * K * 4); hipMalloc(&dW, K * 64 * 4); hipMalloc(&db, 256); hipMalloc(&o16, 16 * 64 * 4); hipMalloc(&o4, 9 * 64 * 4); hipMalloc(&dc, 32);
    hipMemcpy(dX, hX, 16 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dW, hW, K * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dX, dW, db, o16, o4, dc);
    float h16[16 * 64], h4[9 * 64]; unsigned long long hc[4];
    hipMemcpy(h16, o16, sizeof(h16), hipMemcpyDeviceToHost); hipMemcpy(h4, o4, sizeof(h4), hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 32, hipMemcpyDeviceToHost);
    int same = 0, close = 0; double maxd = 0;
    for (int r = 0; r < 8; r++) for (int c = 0; c < 64; c++) {
        const float x = h16[r * 64 + c], y = h4[r * 64 + c];
        same += memcmp(&x, &y, 4) == 0; close += fabsf(x - y) <= 1e-5f * fmaxf(1.f, fabsf(x)); maxd = fmax(maxd, fabs((double)x - y));
    }
    // host restatements: sequential fma in k order / products summed four at a time
    int seq_same = 0, seq4_same = 0;
    for (int r = 0; r < 8; r++) for (int c = 0; c < 64; c++) {
        float s = hb[c];
        for (int k = 0; k < K; k++) s = fmaf(hX[r * K + k], hW[k * 64 + c], s);
        seq_same += memcmp(&s, &h16[r * 64 + c], 4) == 0; seq4_same += memcmp(&s, &h4[r * 64 + c], 4) == 0;
    }
    printf("broadcast form vs 16x16x4: %d of 512 bit-identical, %d within 1e-5, max |diff| %.3g\n", same, close, maxd);
    printf("host sequential fmaf: == 16x16x4 in %d of 512, == 4x4x1 chain in %d of 512\n", seq_same, seq4_same);
    printf("cycles per 4x4x1: two chains %.1f, four chains %.1f; with LDS-free global operands per k (loop above): %.1f per pair\n", hc[0] / 256.0, hc[1] / 256.0, hc[2] / (double)K);
    return 0;
}

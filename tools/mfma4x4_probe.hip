// Operand layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, found by experiment: A lane l = l + 1, B lane l = 1000 (l + 1): D register i of lane l
// shows which A lane and which B lane met.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma4x4_probe tools/mfma4x4_probe.hip (tools/bin/ is git-ignored)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, unsigned long long *cyc) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // A lane l = 2^(l & 3) * 3^(l >> 2), B lane l = 5^(l & 3) * 7^(l >> 2): the product names both lanes (blocks only meet their own lanes, so the
    // exponents of 3 and 7 stay small enough for exact floats)
    const float a = exp2f((float)(l & 3)) * powf(3.0f, (float)((l >> 2) & 3)) * ((l >> 4) == 0 ? 1.f : (l >> 4) == 1 ? 11.f : (l >> 4) == 2 ? 13.f : 17.f);
    const float b = powf(5.0f, (float)(l & 3)) * powf(7.0f, (float)((l >> 2) & 3)) * ((l >> 4) == 0 ? 1.f : (l >> 4) == 1 ? 19.f : (l >> 4) == 2 ? 23.f : 29.f);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) out[l * 4 + i] = c[i];
    // issue-rate probe: 256 dependent-free MFMAs on 4 accumulators
    f32x4 acc[4] = {c, c, c, c};
    asm volatile("s_nop 0" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+s"(const_cast<unsigned long long &>(t0)));
#pragma unroll
    for (int it = 0; it < 64; it++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0);
    asm volatile("s_nop 0" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 acc2[4] = {c, c, c, c};
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]), "+s"(const_cast<unsigned long long &>(t1)));
#pragma unroll
    for (int it = 0; it < 64; it++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc2[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[q], 0, 0, 0);
    asm volatile("s_nop 0" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (l == 0) cyc[0] = t1 - t0, cyc[1] = t2 - t1;
    out[256 + l] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc2[0][0] + acc2[1][1] + acc2[2][2] + acc2[3][3];
}
int main() {
    float *d; unsigned long long *c;
    hipMalloc(&d, 4 * 512); hipMalloc(&c, 16);
    probe<<<1, 64>>>(d, c);
    float h[256]; unsigned long long hc[2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    auto enc_a = [](int l) { return exp2((double)(l & 3)) * pow(3.0, (double)((l >> 2) & 3)) * ((l >> 4) == 0 ? 1. : (l >> 4) == 1 ? 11. : (l >> 4) == 2 ? 13. : 17.); };
    auto enc_b = [](int l) { return pow(5.0, (double)(l & 3)) * pow(7.0, (double)((l >> 2) & 3)) * ((l >> 4) == 0 ? 1. : (l >> 4) == 1 ? 19. : (l >> 4) == 2 ? 23. : 29.); };
    int ok = 1;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            int al = -1, bl = -1;
            for (int x = 0; x < 64 && al < 0; x++)
                for (int y = 0; y < 64; y++)
                    if (enc_a(x) * enc_b(y) == (double)h[l * 4 + i]) { al = x, bl = y; break; }
            if (l < 6 || l > 61) printf("lane %2d d[%d] = A lane %2d x B lane %2d\n", l, i, al, bl);
            if (al != (l & ~3) + i || bl != l) ok = 0;
        }
    printf("layout D(lane 4b + j, register i) = A(lane 4b + i) x B(lane 4b + j): %s\n", ok ? "CONFIRMED" : "NO");
    printf("256 x mfma 4x4x1: %llu memtime ticks, 256 x mfma 16x16x4: %llu ticks (s_memtime)\n", hc[0], hc[1]);
    return 0;
}

// What SQ_LDS_BANK_CONFLICT counts on gfx950: four kernels that only read LDS, each with a pattern that is conflict-free by construction
// (lane l reads the l-th consecutive element of its width), and one with a deliberate 2-way conflict (stride of two b32 words... every second bank).
// Run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace; tools/r06_lds_probe.sh prints the ratios.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/lds_counter_probe tools/lds_counter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int IT = 4096;
__global__ void lds_b32(float *out) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    float a = 0.f;
    for (int it = 0; it < IT; it++) a += s[(threadIdx.x + 64 * (it & 31)) & 4095];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void lds_b64(float *out) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    float a = 0.f;
    for (int it = 0; it < IT; it++) { const f32x2 v = *reinterpret_cast<const f32x2 *>(s + ((2 * threadIdx.x + 128 * (it & 15)) & 4095)); a += v[0] + v[1]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void lds_b128(float *out) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    float a = 0.f;
    for (int it = 0; it < IT; it++) { const f32x4 v = *reinterpret_cast<const f32x4 *>(s + ((4 * threadIdx.x + 256 * (it & 7)) & 4095)); a += v[0] + v[1] + v[2] + v[3]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void lds_b32_2way(float *out) {  // lanes l and l + 32 share a bank (stride 2 words: 32 distinct banks of 64)
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    float a = 0.f;
    for (int it = 0; it < IT; it++) a += s[(2 * threadIdx.x + 128 * (it & 15)) & 4095];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void lds_b128_rows(float *out) {  // the MFMA A-fragment pattern: 16 rows of stride 260 words x 4 chunks of 16 bytes (conflict-free in a bank simulation)
    __shared__ float s[16 * 260 + 64];
    for (int i = threadIdx.x; i < 16 * 260 + 64; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    float a = 0.f;
    const int r16 = threadIdx.x & 15, g = threadIdx.x >> 4;
    for (int it = 0; it < IT; it++) { const f32x4 v = *reinterpret_cast<const f32x4 *>(s + r16 * 260 + 4 * g + 16 * (it & 15)); a += v[0] + v[1] + v[2] + v[3]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    float *d; hipMalloc(&d, 256 * 64 * 4);
    lds_b32<<<256, 64>>>(d); lds_b64<<<256, 64>>>(d); lds_b128<<<256, 64>>>(d); lds_b32_2way<<<256, 64>>>(d); lds_b128_rows<<<256, 64>>>(d);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}

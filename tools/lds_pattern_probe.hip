// Measured LDS-array cycles per wave-instruction of an arbitrary per-lane address pattern on gfx950 (round 6): the ground truth a bank model is
// checked against.  Sixteen waves of one workgroup (four per SIMD) issue the same instruction with the same 64 lane offsets N times; the time
// of the whole block divided by 16 N is the LDS-array time one wave-instruction takes (conflict-free: ds_read_b32 / b64 2 cycles, b128 4;
// ds_write_b32 4, b64 ~6, b128 ~13 -- MI355X_MICROARCH.md, LDS).  kind: 0 ds_read_b32, 1 ds_read_b64, 2 ds_read_b128, 3 ds_write_b32,
// 4 ds_write_b64, 5 ds_write_b128.  Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/bin/liblds_probe.so tools/lds_pattern_probe.hip
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int N = 2048;
template <int KIND>
__global__ __launch_bounds__(1024) void probe(const int *offs, unsigned long long *cyc, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = (float)i;
    const unsigned a = (unsigned)offs[threadIdx.x & 63];
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int it = 0; it < N / 16; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if constexpr (KIND == 0) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); acc[0] += v; }
            if constexpr (KIND == 1) { f32x2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); acc[0] += v[0]; }
            if constexpr (KIND == 2) { f32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); acc[0] += v[0]; }
            if constexpr (KIND == 3) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(acc[0]) : "memory"); }
            if constexpr (KIND == 4) { f32x2 v = {acc[0], acc[1]}; asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
            if constexpr (KIND == 5) { asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(acc) : "memory"); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[1];
}
extern "C" int lds_probe(int kind, const int *lane_byte_offsets, double *cycles_per_instruction) {
    static int *d_off = nullptr; static unsigned long long *d_cyc = nullptr; static float *d_sink = nullptr;
    if (!d_off) { hipMalloc(&d_off, 256); hipMalloc(&d_cyc, 8); hipMalloc(&d_sink, 4096); }
    hipMemcpy(d_off, lane_byte_offsets, 256, hipMemcpyHostToDevice);
    void (*k)(const int *, unsigned long long *, float *) = kind == 0 ? probe<0> : kind == 1 ? probe<1> : kind == 2 ? probe<2> : kind == 3 ? probe<3> : kind == 4 ? probe<4> : probe<5>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; rep++) k<<<1, 1024, 65536>>>(d_off, d_cyc, d_sink);
    unsigned long long c = 0;
    if (hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    *cycles_per_instruction = (double)c / (16.0 * N);
    return 0;
}

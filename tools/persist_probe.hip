// persist_probe.hip -- measures what one in-launch exchange step costs on MI355X, as used by the persistent small-minibatch update
// kernel (csrc/tma_h64p.hip): NB workgroups of 256 threads each publish a K x 4 KiB slab, arrive on one counter, wait for everyone
// and read all NB slabs back.  Every word is checked.  Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/persist_probe tools/persist_probe.hip && /tmp/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int SC1 = 16;  // gfx940+ cache-policy bit of the raw buffer intrinsics

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

// mode 0: sc1 stores + sc1 loads, no fences.  mode 1: plain stores, agent release fence, agent acquire fence, plain loads.
// mode 2: plain stores + sc1 loads, no fences: valid ONLY when every participating workgroup sits on one XCD (one L2).
// SAME_XCD: the grid is larger than NB; workgroups read HW_REG_XCC_ID and the first NB on the XCD of the first claimer take the roles.
template <int MODE, int K, bool SAME_XCD>
__global__ __launch_bounds__(256, 1) void probe(unsigned *slabs, unsigned *counter_base, unsigned long long *out, int steps, int work, int NB) {
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    unsigned *counter = counter_base;
    __shared__ unsigned xcc[1];
    __shared__ int role_s;
    if (SAME_XCD) {
        if (tid == 0) {
            const unsigned myx = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
            const unsigned old = atomicCAS(counter_base + 32, 0xFFFFFFFFu, myx);
            const unsigned target = old == 0xFFFFFFFFu ? myx : old;
            role_s = myx == target ? (int)atomicAdd(counter_base + 33, 1u) : -1;
        }
        __syncthreads();
        b = role_s;
        if (b < 0 || b >= NB) return;
    }
    unsigned long long errors = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_wait = 0, t_load = 0, t_pub = 0;
    constexpr int JB = 4;
    for (int s = 0; s < steps; s++) {
        for (int q = 0; q < work; q++) __builtin_amdgcn_s_sleep(8);
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
        unsigned *mine = slabs + ((size_t)(s & 1) * NB + b) * (K * 1024);
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(mine, 0, K * 4096, 0x00020000);
#pragma unroll
        for (int k = 0; k < K; k++) {
            const unsigned base = (unsigned)s * 7919u + (unsigned)b * 104729u + (unsigned)(k * 1024 + tid * 4);
            const u32x4 v = {base, base + 1, base + 2, base + 3};
            if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(v, wr, (k * 256 + tid) * 16, 0, SC1);
            else if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(v, wr, (k * 256 + tid) * 16, 0, 0);
            else *reinterpret_cast<u32x4 *>(mine + k * 1024 + tid * 4) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (MODE == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long tb = __builtin_amdgcn_s_memtime();
            const unsigned want = (unsigned)NB * (unsigned)(s + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 24)) break;
            }
            if (MODE == 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            t_pub += tb - ta;
            t_wait += __builtin_amdgcn_s_memtime() - tb;
        }
        __syncthreads();
        const unsigned long long tc = __builtin_amdgcn_s_memtime();
        const unsigned *all = slabs + (size_t)(s & 1) * NB * (K * 1024);
        __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(all), 0, NB * K * 4096, 0x00020000);
        for (int bb0 = 0; bb0 < NB; bb0 += JB) {
            u32x4 v[JB][K];
#pragma unroll
            for (int j = 0; j < JB; j++)
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (MODE == 0 || MODE == 2) v[j][k] = __builtin_amdgcn_raw_buffer_load_b128(rd, (((bb0 + j) * K + k) * 256 + tid) * 16, 0, SC1);
                    else v[j][k] = *reinterpret_cast<const u32x4 *>(all + ((size_t)(bb0 + j) * K + k) * 1024 + tid * 4);
                }
#pragma unroll
            for (int j = 0; j < JB; j++)
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const unsigned base = (unsigned)s * 7919u + (unsigned)(bb0 + j) * 104729u + (unsigned)(k * 1024 + tid * 4);
                    for (int c = 0; c < 4; c++) errors += v[j][k][c] != base + c;
                }
        }
        if (tid == 0) t_load += __builtin_amdgcn_s_memtime() - tc;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        xcc[0] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        out[b * 8 + 0] = t1 - t0;
        out[b * 8 + 1] = t_pub;
        out[b * 8 + 2] = t_wait;
        out[b * 8 + 3] = t_load;
        out[b * 8 + 4] = xcc[0];
    }
    for (int o = 32; o > 0; o >>= 1) errors += __shfl_down(errors, o, 64);
    if ((tid & 63) == 0 && errors) atomicAdd(&out[b * 8 + 5], errors);
}

template <int MODE, int K, bool SAME_XCD = false>
static void run(int NB, int steps, int work) {
    const int grid = SAME_XCD ? 128 : NB;
    unsigned *slabs, *counter;
    unsigned long long *out;
    CHECK(hipMalloc(&slabs, (size_t)2 * NB * K * 4096));
    CHECK(hipMalloc(&counter, 256));
    CHECK(hipMalloc(&out, NB * 64));
    CHECK(hipMemset(counter, 0, 256));
    CHECK(hipMemset(out, 0, NB * 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipMemset(counter + 32, 0xFF, 4));
    probe<MODE, K, SAME_XCD><<<grid, 256>>>(slabs, counter, out, 16, work, NB);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(counter, 0, 256));
    CHECK(hipMemset(out, 0, NB * 64));
    CHECK(hipMemset(counter + 32, 0xFF, 4));
    CHECK(hipEventRecord(e0));
    probe<MODE, K, SAME_XCD><<<grid, 256>>>(slabs, counter, out, steps, work, NB);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(NB * 8);
    CHECK(hipMemcpy(h.data(), out, NB * 64, hipMemcpyDeviceToHost));
    unsigned long long err = 0;
    for (int b = 0; b < NB; b++) err += h[b * 8 + 5];
    // s_memtime ticks: report per step in ticks and (through the realtime counter, 100 MHz) in microseconds
    const double us_total = (double)h[0] / 100.0;
    const double tick_us = us_total / ((double)(h[1] + h[2] + h[3]) + 1e-9);  // (rough: phases of thread 0 cover most of the loop when work = 0)
    printf("%s mode %d NB %2d K %d (%2d KiB/slab) work %2d: %.2f us/step (event %.2f)  pub %.0f wait %.0f load %.0f ticks/step  xcc", SAME_XCD ? "xcd" : "any", MODE, NB, K, K * 4, work,
           us_total / steps, ms * 1e3 / steps, (double)h[1] / steps, (double)h[2] / steps, (double)h[3] / steps);
    for (int b = 0; b < NB && b < 16; b++) printf(" %llu", h[b * 8 + 4]);
    printf("  errors %llu%s\n", err, work == 0 ? "" : "");
    (void)tick_us;
    CHECK(hipFree(slabs));
    CHECK(hipFree(counter));
    CHECK(hipFree(out));
}

int main() {
    const int steps = 2000;
    for (int work : {0, 6}) {
        run<0, 5>(8, steps, work);
        run<1, 5>(8, steps, work);
        run<0, 5, true>(8, steps, work);
        run<1, 5, true>(8, steps, work);
        run<2, 5, true>(8, steps, work);
        run<2, 5, true>(4, steps, work);
        run<2, 5, true>(16, steps, work);
        run<2, 2, true>(8, steps, work);
        run<2, 2, true>(16, steps, work);
        run<2, 1, true>(8, steps, work);
        run<0, 1>(8, steps, work);
    }
    return 0;
}

// tma_env.hip -- batched vector-env engine for gfx950: reset / step (+auto-reset, terminal obs, Monitor
// episode sums) / reset-ring refill (exact numpy MT19937 legacy stream) / state get+set.
//
// Layout in HBM (per env handle): packed state words struct-of-arrays st[w][N]; reset ring
// ring[slot][w][N]; ep_ret f64[N]; cur_ep u32[N]; filled_hi u32[N].  One thread per env, 64 consecutive
// envs per wavefront, every state load/store is one coalesced 256-B row per word.
//
// Replaces (reference, paths under /root/reference/): DummyVecEnv/Monitor semantics constructed by
// backend/mlagents/training.py:71-89; LegacySingleAgentGymAdapter backend/mlagents/envs.py:87-159;
// task dynamics backend/mlagents/envs.py:30-84, backend/examples/gridworld.py:33-95,
// backend/examples/ball3d.py:41-113, backend/examples/push.py:27-125.
#include "tma_internal.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

namespace tma {

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

enum { ACT_I32 = 0, ACT_I64 = 1, ACT_F32 = 2, ACT_TAPE = 3 };

// ------------------------------------------------------------------------------------------
// Device buffers of an env handle come from a small size-keyed cache: a handle owns ~25 allocations, hipMalloc / hipFree cost 50-150 us
// each (hipFree also drains the device), and the callers this library replaces build and close a vector env per training run and per
// evaluation (training.py:71-89, 227-247) -- 3 ms of a 90 ms train_task call went into closing its two envs.  Freed blocks of up to 64 MiB
// are kept (at most 512 MiB per process) and handed to the next request of exactly that size on that device; tma_env_destroy drains the
// device first, so no kernel of the old handle can still touch a block when it is handed out again.
// ------------------------------------------------------------------------------------------
namespace {
struct BlockCache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> free_blocks;
    std::unordered_map<void *, std::pair<int, size_t>> live;
    size_t cached_bytes = 0;
};
BlockCache &block_cache() {
    static BlockCache *c = new BlockCache;  // (never destroyed: the HIP runtime may be gone by the time statics are)
    return *c;
}
constexpr size_t CACHE_BLOCK_MAX = 64u << 20, CACHE_TOTAL_MAX = 512u << 20;
}  // namespace

template <class P>
hipError_t env_malloc(P **p, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    BlockCache &c = block_cache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.free_blocks.find({dev, bytes});
        if (it != c.free_blocks.end()) {
            *p = static_cast<P *>(it->second);
            c.cached_bytes -= bytes;
            c.free_blocks.erase(it);
            c.live[*p] = {dev, bytes};
            return hipSuccess;
        }
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {  // out of memory with blocks parked in the cache: give them back and try once more
        {
            std::lock_guard<std::mutex> lk(c.mu);
            for (auto &kv : c.free_blocks) (void)hipFree(kv.second);
            c.free_blocks.clear();
            c.cached_bytes = 0;
        }
        (void)hipGetLastError();
        e = hipMalloc(&q, bytes);
        if (e != hipSuccess) return e;
    }
    *p = static_cast<P *>(q);
    std::lock_guard<std::mutex> lk(c.mu);
    c.live[q] = {dev, bytes};
    return hipSuccess;
}

void env_free(void *p) {
    if (!p) return;
    BlockCache &c = block_cache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const auto key = it->second;
            c.live.erase(it);
            if (key.second <= CACHE_BLOCK_MAX && c.cached_bytes + key.second <= CACHE_TOTAL_MAX) {
                c.free_blocks.insert({key, p});
                c.cached_bytes += key.second;
                return;
            }
        }
    }
    (void)hipFree(p);
}

// ------------------------------------------------------------------------------------------
// step kernel: n_steps vector steps with the state held in registers.
// ------------------------------------------------------------------------------------------
// Tasks whose observation row is not a power-of-two number of bytes (Basic 21 floats, Crawler 172) stage the block's
// observations in LDS ([threads][OBS|1], odd stride = conflict-free) and store them with fully coalesced rows; a per-lane
// store of such rows would touch 64 different cache lines per instruction (measured 20x slower on Basic).
template <class T>
struct ObsStaging {
    static constexpr bool USE_LDS = T::OBS > 8;
    static constexpr int OBSP = T::OBS | 1;
    static constexpr int THREADS = T::OBS > 32 ? 64 : 256;
};

template <class T, int ACTMODE>
__global__ __launch_bounds__(256) void step_kernel(EnvView v, const void *__restrict__ actions, uint32_t tape_seed, uint32_t tape_t0,
                                                   int n_steps, float *__restrict__ obs_out, float *__restrict__ rew_out,
                                                   uint8_t *__restrict__ term_out, uint8_t *__restrict__ trunc_out,
                                                   float *__restrict__ term_obs_out, double *__restrict__ ep_ret_out,
                                                   int32_t *__restrict__ ep_len_out, double *__restrict__ rew64_out) {
    extern __shared__ __attribute__((aligned(16))) float obs_tile[];
    using OS = ObsStaging<T>;
    const int64_t blk0 = (int64_t)blockIdx.x * blockDim.x;
    const int64_t i = blk0 + threadIdx.x;
    const bool active = i < v.N;
    double sret = 0.0, slen = 0.0, scnt = 0.0;
    typename T::S s;
    double er = 0.0;
    uint32_t ce = 0;
    const uint32_t gi = v.env_offset + (uint32_t)i;
    if (active) {
        T::unpack(v.st, v.N, i, s);
        er = v.ep_ret[i];
        ce = v.cur_ep[i];
    }
    for (int k = 0; k < n_steps; k++) {
        const int64_t off = (int64_t)k * v.N + i;
        if (active) {
            int a = 0;
            float fa[T::ADIM];
            if constexpr (T::NACT > 0) {
                if constexpr (ACTMODE == ACT_I32) a = static_cast<const int32_t *>(actions)[off];
                else if constexpr (ACTMODE == ACT_I64) a = (int)static_cast<const int64_t *>(actions)[off];
                else if constexpr (ACTMODE == ACT_TAPE) a = (int)(mix32(tape_seed, gi, tape_t0 + (uint32_t)k) % (uint32_t)T::NACT);
                a = clampi(a, 0, T::NACT - 1);
            } else {
                if constexpr (ACTMODE == ACT_TAPE) {
#pragma unroll
                    for (int j = 0; j < T::ADIM; j++) {
                        uint32_t h = mix32(tape_seed ^ (0x9E37u * (uint32_t)(j + 1)), gi, tape_t0 + (uint32_t)k);
                        fa[j] = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
                    }
                } else {
                    const float4 *src = reinterpret_cast<const float4 *>(static_cast<const float *>(actions) + off * T::ADIM);
#pragma unroll
                    for (int j = 0; j < T::ADIM / 4; j++) {
                        float4 q = src[j];
                        fa[4 * j] = q.x, fa[4 * j + 1] = q.y, fa[4 * j + 2] = q.z, fa[4 * j + 3] = q.w;
                    }
                }
            }
            double r;
            bool done;
            T::step(s, a, fa, r, done);
            const int steps = T::steps(s);
            bool te, tr;
            if constexpr (T::NATIVE_TRUNC_RULE) {  // backend/mlagents/envs.py:76
                te = done;
                tr = (steps >= T::MAXSTEPS) && !te;
            } else {  // adapter rule, backend/mlagents/envs.py:139-145
                const bool hit = steps >= T::MAXSTEPS;
                te = done && !hit;
                tr = hit;
            }
            er += r;  // Monitor: sum of float(reward) in order
            if (rew_out) rew_out[off] = (float)r;
            if (rew64_out) rew64_out[off] = r;  // seam S1: the float64 the reference's env.step returns as float(reward) (envs.py:125-152)
            if (term_out) term_out[off] = (uint8_t)te;
            if (trunc_out) trunc_out[off] = (uint8_t)tr;
            if (te || tr) {
                if (term_obs_out) emit_obs<T>(s, term_obs_out + off * T::OBS);
                if (ep_ret_out) ep_ret_out[off] = er;
                if (ep_len_out) ep_len_out[off] = steps;
                sret += er, slen += (double)steps, scnt += 1.0;
                log_episode(v, i, er, steps);
                er = 0.0;
                ce += 1;
                if constexpr (T::USES_MT) {
                    uint32_t rec[T::RW > 0 ? T::RW : 1];
                    const uint32_t *slot = v.ring + ((int64_t)(ce % (uint32_t)v.D) * T::RW) * v.N + i;
#pragma unroll
                    for (int w = 0; w < T::RW; w++) rec[w] = slot[(int64_t)w * v.N];
                    T::from_rec(rec, s);
                } else {
                    T::reset_inline(episode_seed(v.seed_base, gi, ce), s);
                }
            } else {
                if (ep_ret_out) ep_ret_out[off] = 0.0;
                if (ep_len_out) ep_len_out[off] = 0;
            }
            if constexpr (OS::USE_LDS) T::obs(s, obs_tile + threadIdx.x * OS::OBSP);
            else emit_obs<T>(s, obs_out + off * T::OBS);
        }
        if constexpr (OS::USE_LDS) {  // block-cooperative, fully coalesced store of the block's observation rows
            __syncthreads();
            const int64_t rows = (v.N - blk0) < (int64_t)blockDim.x ? (v.N - blk0) : (int64_t)blockDim.x;
            float *dst = obs_out + ((int64_t)k * v.N + blk0) * T::OBS;
            for (int e = threadIdx.x; e < (int)rows * T::OBS; e += blockDim.x) {
                const int row = e / T::OBS, c = e - row * T::OBS;
                dst[e] = obs_tile[row * OS::OBSP + c];
            }
            __syncthreads();
        }
    }
    if (active) {
        T::pack(v.st, v.N, i, s);
        v.ep_ret[i] = er;
        v.cur_ep[i] = ce;
    }
    // Monitor aggregate: wavefront shuffle reduction, then one update per block of the 256-env slot it belongs to (plain
    // read-modify-write when the block IS the slot, f64 atomics when four 64-thread blocks share it); the host sums the slots.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sret += __shfl_down(sret, o, 64);
        slen += __shfl_down(slen, o, 64);
        scnt += __shfl_down(scnt, o, 64);
    }
    __shared__ double red[3][4];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) red[0][wave] = sret, red[1][wave] = slen, red[2][wave] = scnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0, a0 = 0.0, a1 = 0.0;
        for (int w = 0; w < nw; w++) a0 += red[0][w], a1 += red[1][w], c += red[2][w];
        if (c > 0.0) {
            double *slot = v.stats + (blk0 >> 8) * 3;
            if (blockDim.x == 256) {
                slot[0] += a0, slot[1] += a1, slot[2] += c;
            } else {
                atomicAdd(slot + 0, a0), atomicAdd(slot + 1, a1), atomicAdd(slot + 2, c);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Reset-ring production.  A work item is one (env, episode) pair whose reset record must be drawn with the exact numpy
// legacy stream of seed s(env, episode).  mode 0 = VecEnv.reset: items are dense (env i, episodes 0..D; episode 0 goes
// straight into the state).  mode 1 = refill: items are the episodes consumed since the last refill, found through a
// two-level exclusive scan (per 256-env block, then over blocks) so that every lane of the seeding kernel gets exactly
// one item regardless of how unevenly episodes ended.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refill_count_kernel(EnvView v, RefillView rv) {
    __shared__ uint32_t wsum[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t c = 0;
    if (i < v.N) {
        const uint32_t fh = v.filled_hi[i], nh = v.cur_ep[i] + (uint32_t)v.D + 1u;
        c = nh - fh;
        rv.first_ep[i] = fh;
        v.filled_hi[i] = nh;
    }
    uint32_t inc = c;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d, 64);
        if ((threadIdx.x & 63) >= d) inc += up;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (i < v.N) rv.env_off[i] = base + inc - c;
    if (threadIdx.x == 255) rv.block_sum[blockIdx.x] = base + inc;
}

__global__ __launch_bounds__(1024) void refill_scan_kernel(RefillView rv) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < rv.nb; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const uint32_t c = b < rv.nb ? rv.block_sum[b] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(inc, d, 64);
            if ((threadIdx.x & 63) >= d) inc += up;
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t base = carry;
        for (int w = 0; w < wave; w++) base += wsum[w];
        if (b < rv.nb) rv.block_off[b] = base + inc - c;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) rv.total[0] = carry;
}

template <class T>
__device__ __forceinline__ void commit_record(EnvView &v, int64_t i, uint32_t e, const uint32_t *rec, int mode, float *obs_out) {
    if (mode == 0 && e == 0) {
        typename T::S s;
        T::from_rec(rec, s);
        T::pack(v.st, v.N, i, s);
        v.cur_ep[i] = 0;
        v.ep_ret[i] = 0.0;
        v.filled_hi[i] = (uint32_t)v.D + 1u;
        if (obs_out) emit_obs<T>(s, obs_out + i * T::OBS);
    } else {
        uint32_t *slot = v.ring + ((int64_t)(e % (uint32_t)v.D) * T::RW) * v.N + i;
#pragma unroll
        for (int w = 0; w < T::RW; w++) slot[(int64_t)w * v.N] = rec[w];
    }
}

// upper_bound(a[0..n), x) - 1 : index of the last element <= x
__device__ __forceinline__ int last_le(const uint32_t *a, int n, uint32_t x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= x) lo = mid + 1;
        else hi = mid;
    }
    return lo - 1;
}

template <class T, int W>
__global__ __launch_bounds__(256) void refill_fast_kernel(EnvView v, RefillView rv, int mode, int64_t total_dense, float *obs_out, int64_t lo,
                                                          int64_t hi) {
    // items [lo, hi) of this refill: a round never holds more than fb_cap items, so the fallback list cannot overflow
    int64_t total = mode == 0 ? total_dense : (int64_t)rv.total[0];
    if (total > hi) total = hi;
    for (int64_t t = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i;
        uint32_t e;
        if (mode == 0) {
            i = t / (v.D + 1);
            e = (uint32_t)(t - i * (v.D + 1));
        } else {
            const int b = last_le(rv.block_off, rv.nb, (uint32_t)t);
            const uint32_t local = (uint32_t)t - rv.block_off[b];
            const int64_t i0 = (int64_t)b * 256;
            const int nloc = (int)((v.N - i0) < 256 ? (v.N - i0) : 256);
            const int j = last_le(rv.env_off + i0, nloc, local);
            i = i0 + j;
            e = rv.first_ep[i] + (local - rv.env_off[i]);
        }
        const uint32_t seed = episode_seed(v.seed_base, v.env_offset + (uint32_t)i, e);
        typename T::Fast c;
        if (mt_stream<W>(seed, c)) {
            uint32_t rec[T::RW > 0 ? T::RW : 1];
            c.finish(rec);
            commit_record<T>(v, i, e, rec, mode, obs_out);
        } else {  // more than W outputs needed (rejection sampling): exact general generator takes over
            const uint32_t k = atomicAdd(&rv.total[1], 1u);  // < fb_cap by construction (hi - lo <= fb_cap)
            rv.fb_env[k] = (uint32_t)i;
            rv.fb_ep[k] = e;
        }
    }
}

// general path: full MT19937 state in lane-interleaved global scratch (any number of draws)
template <class T>
__global__ __launch_bounds__(256) void refill_fallback_kernel(EnvView v, RefillView rv, uint32_t *mt_scratch, int mode, float *obs_out) {
    __shared__ uint8_t lds_js[36 * 256];
    const int64_t G = (int64_t)gridDim.x * 256;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n = rv.total[1];
    if (n > (uint32_t)rv.fb_cap) n = (uint32_t)rv.fb_cap;
    MT mt{mt_scratch + tid, G, 0};
    for (int64_t k = tid; k < (int64_t)n; k += G) {
        const int64_t i = rv.fb_env[k];
        const uint32_t e = rv.fb_ep[k];
        uint32_t rec[T::RW > 0 ? T::RW : 1];
        mt.seed(episode_seed(v.seed_base, v.env_offset + (uint32_t)i, e));
        // adapter.reset(seed): env_ctor() resets once, then env.reset() -- the second draw is the visible one
        T::draw(mt, lds_js + threadIdx.x, 256, rec);
        T::draw(mt, lds_js + threadIdx.x, 256, rec);
        commit_record<T>(v, i, e, rec, mode, obs_out);
    }
}

// tasks whose reset needs no MT19937 (Basic, Crawler-shape): VecEnv.reset only
template <class T>
__global__ __launch_bounds__(256) void reset_inline_kernel(EnvView v, float *obs_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= v.N) return;
    typename T::S s;
    T::reset_inline(episode_seed(v.seed_base, v.env_offset + (uint32_t)i, 0), s);
    T::pack(v.st, v.N, i, s);
    v.cur_ep[i] = 0;
    v.ep_ret[i] = 0.0;
    if (obs_out) emit_obs<T>(s, obs_out + i * T::OBS);
}

template <class T>
__global__ void get_state_kernel(EnvView v, double *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.N) return;
    typename T::S s;
    T::unpack(v.st, v.N, i, s);
    double f[T::SDIM];
    T::to_flat(s, f);
#pragma unroll
    for (int k = 0; k < T::SDIM; k++) out[i * T::SDIM + k] = f[k];
}

template <class T>
__global__ void set_state_kernel(EnvView v, const double *in) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.N) return;
    double f[T::SDIM];
#pragma unroll
    for (int k = 0; k < T::SDIM; k++) f[k] = in[i * T::SDIM + k];
    typename T::S s;
    T::from_flat(f, s);
    T::pack(v.st, v.N, i, s);
}

__global__ void copy_u32_kernel(const uint32_t *src, uint32_t *dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

struct TaskMeta {
    const char *name;
    int obs, nact, adim, sdim, maxsteps, sw, rw;
    bool uses_mt;
};
template <class T>
constexpr TaskMeta meta_of(const char *n) {
    return TaskMeta{n, T::OBS, T::NACT, T::ADIM, T::SDIM, T::MAXSTEPS, T::SW, T::RW, T::USES_MT};
}
static const TaskMeta kMeta[TMA_NUM_TASKS] = {meta_of<BasicTask>("basic"), meta_of<GridTask>("gridworld"), meta_of<BallTask>("ball3d"),
                                              meta_of<PushTask>("push"), meta_of<CrawlerTask>("crawler"), meta_of<WallJumpTask>("walljump"),
                                              meta_of<BicycleTask>("bicycle"), meta_of<BrickBreakTask>("brickbreak"), meta_of<GliderTask>("glider"),
                                              meta_of<AntTask>("ant")};

}  // namespace tma

using namespace tma;

template <class T, int W>
static void launch_fast(tma_env *h, int mode, int64_t total_dense, unsigned blocks, unsigned threads, float *obs_out, hipStream_t s, int64_t lo,
                        int64_t hi) {
    refill_fast_kernel<T, W><<<dim3(blocks), dim3(threads), 0, s>>>(h->v, h->rv, mode, total_dense, obs_out, lo, hi);
}

static int launch_seed(tma_env *h, int mode, float *obs_out, hipStream_t s) {
    return dispatch_task(h->task, [&](auto t) {
        using T = decltype(t);
        if constexpr (!T::USES_MT) {
            if (mode == 0) {
                reset_inline_kernel<T><<<dim3((unsigned)ceil_div(h->v.N, 256)), dim3(256), 0, s>>>(h->v, obs_out);
                TMA_LAUNCH_CHECK();
            }
            return (int)TMA_OK;
        } else {
            TMA_HIP(hipMemsetAsync(h->rv.total, 0, sizeof(uint32_t) * 2, s));
            int64_t total_dense = 0, blocks;
            unsigned threads = 256;
            if (mode == 0) {
                total_dense = h->v.N * (int64_t)(h->v.D + 1);
                blocks = ceil_div(total_dense, 256);
            } else {
                refill_count_kernel<<<dim3((unsigned)h->rv.nb), dim3(256), 0, s>>>(h->v, h->rv);
                TMA_LAUNCH_CHECK();
                refill_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(h->rv);
                TMA_LAUNCH_CHECK();
                blocks = 2 * (int64_t)h->rv.nb;  // about D / mean-episode-length items per env; the kernel grid-strides past that
                if (h->v.N <= 65536) {
                    // few thousand items, each a serial ~4k-instruction MT19937 seeding chain on ONE lane: single-wave blocks and a
                    // grid sized for one item per lane put them on every CU instead of on 2 * N / 256 of them (blocks without an
                    // item leave at once)
                    threads = 64;
                    blocks = ceil_div(h->v.N * (int64_t)h->v.D, 256);
                }
            }
            if (blocks > 16384) blocks = 16384;
            if (blocks < 1) blocks = 1;
            // rounds of at most fb_cap items each: every item of a round may need the general generator and still fits the fallback
            // list (one round up to 2^20 items -- every BASELINE shape; at most four rounds beyond, see tma_env_create)
            const int64_t max_items = mode == 0 ? total_dense : h->v.N * (int64_t)h->v.D;
            for (int64_t lo = 0; lo < max_items; lo += h->rv.fb_cap) {
                if (lo > 0) TMA_HIP(hipMemsetAsync(h->rv.total + 1, 0, sizeof(uint32_t), s));
                const int64_t hi = lo + h->rv.fb_cap;
                if (h->small_window) launch_fast<T, T::Fast::W_SMALL>(h, mode, total_dense, (unsigned)blocks, threads, obs_out, s, lo, hi);
                else launch_fast<T, T::Fast::W>(h, mode, total_dense, (unsigned)blocks, threads, obs_out, s, lo, hi);
                TMA_LAUNCH_CHECK();
                refill_fallback_kernel<T><<<dim3(FB_BLOCKS), dim3(256), 0, s>>>(h->v, h->rv, h->mt_scratch, mode, obs_out);
                TMA_LAUNCH_CHECK();
            }
            return (int)TMA_OK;
        }
    });
}

template <class T, int MODE>
static void launch_step(tma_env *h, const void *actions, uint32_t tape_seed, uint32_t t0, int n_steps, float *obs, float *rew, uint8_t *te,
                        uint8_t *tr, float *tobs, double *epr, int32_t *epl, hipStream_t s) {
    using OS = ObsStaging<T>;
    static const int small_threads = getenv("TMA_STEP_THREADS") ? atoi(getenv("TMA_STEP_THREADS")) : 0;  // measurement switch (tools/step_ab.py)
    const int threads = (small_threads == 64 || small_threads == 128) && !OS::USE_LDS && h->v.N <= 65536 ? small_threads : OS::THREADS;
    const unsigned blocks = (unsigned)ceil_div(h->v.N, threads);
    const size_t smem = OS::USE_LDS ? sizeof(float) * OS::THREADS * OS::OBSP : 0;
    step_kernel<T, MODE><<<dim3(blocks), dim3(threads), smem, s>>>(h->v, actions, tape_seed, t0, n_steps, obs, rew, te, tr, tobs, epr, epl, h->rew64_out);
}

// internal (not exported): bookkeeping after a kernel outside this file advanced the envs by n_steps
int tma_env_internal_after_steps(tma_env *h, int n_steps, void *stream) {
    if (!kMeta[h->task].uses_mt) return TMA_OK;
    h->steps_since_refill += n_steps;
    if (h->steps_since_refill >= h->v.D) return tma_env_refill(h, stream);
    return TMA_OK;
}

extern "C" {

int tma_version(void) { return TMA_VERSION; }
const char *tma_last_error(void) { return err_buf(); }

int tma_task_id(const char *name, int *task_out) {
    if (!name || !task_out) return fail(TMA_ERR_INVALID, "tma_task_id: null argument");
    for (int t = 0; t < TMA_NUM_TASKS; t++)
        if (strcmp(kMeta[t].name, name) == 0) {
            *task_out = t;
            return TMA_OK;
        }
    return fail(TMA_ERR_UNKNOWN_TASK, "Unknown task '%s'. Available: ant, ball3d, basic, bicycle, brickbreak, crawler, glider, gridworld, push, walljump", name);
}
#define META_GETTER(fn, field)                                 \
    int fn(int task) {                                         \
        if (task < 0 || task >= TMA_NUM_TASKS) return -1;      \
        return kMeta[task].field;                              \
    }
META_GETTER(tma_task_obs_dim, obs)
META_GETTER(tma_task_num_actions, nact)
META_GETTER(tma_task_act_dim, adim)
META_GETTER(tma_task_state_dim, sdim)
META_GETTER(tma_task_max_episode_steps, maxsteps)

static int env_alloc(tma_env *h, int task, int64_t num_envs, int ring_depth);

int tma_env_create(int task, int64_t num_envs, int device, uint32_t seed_base, uint32_t env_offset, int ring_depth, tma_env **out) {
    if (!out) return fail(TMA_ERR_INVALID, "tma_env_create: out is null");
    if (task < 0 || task >= TMA_NUM_TASKS) return fail(TMA_ERR_UNKNOWN_TASK, "unknown task id %d", task);
    if (num_envs < 1) return fail(TMA_ERR_INVALID, "num_envs must be >= 1 (got %lld)", (long long)num_envs);
    if (ring_depth < 2 || ring_depth > 4096) return fail(TMA_ERR_INVALID, "ring_depth must be in [2, 4096] (got %d)", ring_depth);
    TMA_HIP(hipSetDevice(device));
    tma_env *h = new (std::nothrow) tma_env();
    if (!h) return fail(TMA_ERR_INVALID, "out of host memory");
    memset(h, 0, sizeof(*h));
    h->task = task;
    h->device = device;
    EnvView &v = h->v;
    v.N = num_envs;
    v.D = ring_depth;
    v.seed_base = seed_base;
    v.env_offset = env_offset;
    const int rc = env_alloc(h, task, num_envs, ring_depth);
    if (rc) {  // e.g. out of HBM at millions of envs: give back what was allocated (destroy tolerates the zeroed members); *out untouched
        tma_env_destroy(h);
        return rc;
    }
    TMA_HIP(hipDeviceSynchronize());  // the clears above ran on the null stream, which is not ordered against the caller's non-blocking streams
    *out = h;
    return TMA_OK;
}

}  // extern "C"

static int env_alloc(tma_env *h, int task, int64_t num_envs, int ring_depth) {
    const TaskMeta &m = kMeta[task];
    EnvView &v = h->v;
    const size_t n = (size_t)num_envs;
    TMA_HIP(env_malloc(&v.st, sizeof(uint32_t) * n * m.sw));
    TMA_HIP(hipMemset(v.st, 0, sizeof(uint32_t) * n * m.sw));
    if (m.uses_mt) TMA_HIP(env_malloc(&v.ring, sizeof(uint32_t) * n * m.rw * ring_depth));
    TMA_HIP(env_malloc(&v.cur_ep, sizeof(uint32_t) * n));
    TMA_HIP(env_malloc(&v.filled_hi, sizeof(uint32_t) * n));
    TMA_HIP(env_malloc(&v.ep_ret, sizeof(double) * n));
    const size_t n_stat = (size_t)ceil_div(num_envs, 256) * 3;
    TMA_HIP(env_malloc(&v.stats, sizeof(double) * n_stat));
    TMA_HIP(hipMemset(v.cur_ep, 0, sizeof(uint32_t) * n));
    TMA_HIP(hipMemset(v.filled_hi, 0, sizeof(uint32_t) * n));
    TMA_HIP(hipMemset(v.ep_ret, 0, sizeof(double) * n));
    TMA_HIP(hipMemset(v.stats, 0, sizeof(double) * n_stat));
    // the spare Monitor-aggregate set of tma_env_detach_episode_log, allocated and cleared HERE (tma_env_create drains the device after this):
    // a detach in the middle of a no-drain training loop then only swaps pointers -- no null-stream memset next to non-blocking streams
    TMA_HIP(env_malloc(&h->d_stats, sizeof(double) * n_stat));
    TMA_HIP(hipMemset(h->d_stats, 0, sizeof(double) * n_stat));
    if (m.uses_mt) {
        RefillView &rv = h->rv;
        rv.nb = (int)ceil_div(num_envs, 256);
        // fallback-list capacity = items per refill round: 2^20, or a quarter of the largest possible refill beyond 4 M items
        const int64_t max_items = num_envs * (int64_t)(ring_depth + 1);
        rv.fb_cap = (int)std::min<int64_t>(std::max<int64_t>(FB_CAP, ceil_div(max_items, 4)), (int64_t)INT_MAX);
        TMA_HIP(env_malloc(&rv.first_ep, sizeof(uint32_t) * n));
        TMA_HIP(env_malloc(&rv.env_off, sizeof(uint32_t) * n));
        TMA_HIP(env_malloc(&rv.block_sum, sizeof(uint32_t) * rv.nb));
        TMA_HIP(env_malloc(&rv.block_off, sizeof(uint32_t) * rv.nb));
        TMA_HIP(env_malloc(&rv.total, sizeof(uint32_t) * 2));
        TMA_HIP(env_malloc(&rv.fb_env, sizeof(uint32_t) * (size_t)rv.fb_cap));
        TMA_HIP(env_malloc(&rv.fb_ep, sizeof(uint32_t) * (size_t)rv.fb_cap));
        TMA_HIP(env_malloc(&h->mt_scratch, sizeof(uint32_t) * 624 * (size_t)FB_BLOCKS * 256));
    }
    return TMA_OK;
}

extern "C" {

int tma_env_destroy(tma_env *h) {
    if (!h) return TMA_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();  // the blocks go back to the cache: no kernel of this handle may still be running when one is handed out again
    env_free(h->v.st);
    env_free(h->v.ring);
    env_free(h->v.cur_ep);
    env_free(h->v.filled_hi);
    env_free(h->v.ep_ret);
    env_free(h->v.stats);
    env_free(h->v.log_ret);
    env_free(h->v.log_len);
    env_free(h->v.log_env);
    env_free(h->v.log_n);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_chunk) (void)hipEventDestroy(h->ev_chunk);
    for (hipEvent_t e : h->ev_side)
        if (e) (void)hipEventDestroy(e);
    env_free(h->d_stats), env_free(h->d_log_ret), env_free(h->d_log_len), env_free(h->d_log_env), env_free(h->d_log_n);
    env_free(h->mt_scratch);
    env_free(h->rv.first_ep);
    env_free(h->rv.env_off);
    env_free(h->rv.block_sum);
    env_free(h->rv.block_off);
    env_free(h->rv.total);
    env_free(h->rv.fb_env);
    env_free(h->rv.fb_ep);
    delete h;
    return TMA_OK;
}

int tma_env_set_option(tma_env *h, const char *key, int64_t value) {
    if (!h || !key) return fail(TMA_ERR_INVALID, "null argument");
    if (strcmp(key, "refill_small_window") == 0) {
        h->small_window = value != 0;
        return TMA_OK;
    }
    return fail(TMA_ERR_INVALID, "unknown option '%s'", key);
}

int tma_env_set_reward64(tma_env *h, double *plane, int64_t capacity) {
    if (!h) return fail(TMA_ERR_INVALID, "null env handle");
    if (plane && capacity < h->v.N) return fail(TMA_ERR_INVALID, "tma_env_set_reward64: a plane of %lld doubles cannot hold one step of %lld envs", (long long)capacity, (long long)h->v.N);
    h->rew64_out = plane;
    h->rew64_cap = plane ? capacity : 0;
    return TMA_OK;
}

int tma_env_seed(tma_env *h, uint32_t seed_base) {
    if (!h) return fail(TMA_ERR_INVALID, "null env handle");
    h->v.seed_base = seed_base;
    h->is_reset = false;
    return TMA_OK;
}

int tma_env_reset(tma_env *h, float *obs_out, void *stream) {
    if (!h) return fail(TMA_ERR_INVALID, "null env handle");
    if (!obs_out) return fail(TMA_ERR_INVALID, "tma_env_reset: obs_out is null");
    TMA_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    TMA_HIP(hipMemsetAsync(h->v.stats, 0, sizeof(double) * 3 * (size_t)ceil_div(h->v.N, 256), s));
    int rc = launch_seed(h, 0, obs_out, s);
    if (rc) return rc;
    h->steps_since_refill = 0;
    h->is_reset = true;
    return TMA_OK;
}

int tma_env_refill(tma_env *h, void *stream) {
    if (!h) return fail(TMA_ERR_INVALID, "null env handle");
    if (!h->is_reset) return fail(TMA_ERR_INVALID, "tma_env_refill before tma_env_reset");
    TMA_HIP(hipSetDevice(h->device));
    int rc = launch_seed(h, 1, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    h->steps_since_refill = 0;
    return TMA_OK;
}

int tma_env_steps_until_refill(tma_env *h, int *out) {
    if (!h || !out) return fail(TMA_ERR_INVALID, "null argument");
    *out = kMeta[h->task].uses_mt ? h->v.D - h->steps_since_refill : (1 << 30);
    return TMA_OK;
}

int tma_env_step(tma_env *h, const void *actions, int action_dtype, uint32_t tape_seed, uint32_t tape_t0, int n_steps, float *obs_out,
                 float *rew_out, uint8_t *term_out, uint8_t *trunc_out, float *term_obs_out, double *ep_ret_out, int32_t *ep_len_out,
                 void *stream) {
    if (!h) return fail(TMA_ERR_INVALID, "null env handle");
    if (!h->is_reset) return fail(TMA_ERR_INVALID, "tma_env_step before tma_env_reset");
    if (!obs_out) return fail(TMA_ERR_INVALID, "tma_env_step: obs_out is null");
    if (n_steps < 1) return fail(TMA_ERR_INVALID, "n_steps must be >= 1 (got %d)", n_steps);
    if (h->rew64_out && (int64_t)n_steps * h->v.N > h->rew64_cap)  // (tma_env_set_reward64: the kernel writes plane[k * N + i] for every step k)
        return fail(TMA_ERR_INVALID, "n_steps=%d x %lld envs exceed the registered float64 reward plane (%lld doubles)", n_steps, (long long)h->v.N, (long long)h->rew64_cap);
    const TaskMeta &m = kMeta[h->task];
    if (m.uses_mt && h->steps_since_refill + n_steps > h->v.D)
        return fail(TMA_ERR_INVALID, "n_steps=%d exceeds the %d steps left before a reset-ring refill is due", n_steps,
                    h->v.D - h->steps_since_refill);
    if (actions) {
        const bool discrete = m.nact > 0;
        if (discrete && action_dtype != TMA_ACT_I32 && action_dtype != TMA_ACT_I64)
            return fail(TMA_ERR_INVALID, "task '%s' has Discrete(%d) actions: pass int32 or int64", m.name, m.nact);
        if (!discrete && action_dtype != TMA_ACT_F32) return fail(TMA_ERR_INVALID, "task '%s' has Box actions: pass float32", m.name);
    }
    TMA_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int mode = actions ? action_dtype : ACT_TAPE;
    int rc = dispatch_task(h->task, [&](auto t) {
        using T = decltype(t);
#define TMA_STEP_ARGS h, actions, tape_seed, tape_t0, n_steps, obs_out, rew_out, term_out, trunc_out, term_obs_out, ep_ret_out, ep_len_out, s
        if constexpr (T::NACT > 0) {
            if (mode == ACT_I32) launch_step<T, ACT_I32>(TMA_STEP_ARGS);
            else if (mode == ACT_I64) launch_step<T, ACT_I64>(TMA_STEP_ARGS);
            else launch_step<T, ACT_TAPE>(TMA_STEP_ARGS);
        } else {
            if (mode == ACT_F32) launch_step<T, ACT_F32>(TMA_STEP_ARGS);
            else launch_step<T, ACT_TAPE>(TMA_STEP_ARGS);
        }
#undef TMA_STEP_ARGS
        TMA_LAUNCH_CHECK();
        return (int)TMA_OK;
    });
    if (rc) return rc;
    if (m.uses_mt) {
        h->steps_since_refill += n_steps;
        if (h->steps_since_refill >= h->v.D) return tma_env_refill(h, stream);
    }
    return TMA_OK;
}

int tma_env_step_repeat(tma_env *h, const void *actions, int action_dtype, int reps, float *obs_out, float *rew_out, uint8_t *term_out,
                        uint8_t *trunc_out, float *term_obs_out, void *stream) {
    if (reps < 1) return fail(TMA_ERR_INVALID, "reps must be >= 1");
    for (int r = 0; r < reps; r++) {
        int rc = tma_env_step(h, actions, action_dtype, 0, 0, 1, obs_out, rew_out, term_out, trunc_out, term_obs_out, nullptr, nullptr, stream);
        if (rc) return rc;
    }
    return TMA_OK;
}

int tma_env_get_state(tma_env *h, double *state_out, void *stream) {
    if (!h || !state_out) return fail(TMA_ERR_INVALID, "null argument");
    TMA_HIP(hipSetDevice(h->device));
    return dispatch_task(h->task, [&](auto t) {
        using T = decltype(t);
        get_state_kernel<T><<<dim3((unsigned)ceil_div(h->v.N, 256)), dim3(256), 0, (hipStream_t)stream>>>(h->v, state_out);
        TMA_LAUNCH_CHECK();
        return (int)TMA_OK;
    });
}

int tma_env_set_state(tma_env *h, const double *state_in, void *stream) {
    if (!h || !state_in) return fail(TMA_ERR_INVALID, "null argument");
    TMA_HIP(hipSetDevice(h->device));
    return dispatch_task(h->task, [&](auto t) {
        using T = decltype(t);
        set_state_kernel<T><<<dim3((unsigned)ceil_div(h->v.N, 256)), dim3(256), 0, (hipStream_t)stream>>>(h->v, state_in);
        TMA_LAUNCH_CHECK();
        return (int)TMA_OK;
    });
}

int tma_env_episode_index(tma_env *h, uint32_t *out, void *stream) {
    if (!h || !out) return fail(TMA_ERR_INVALID, "null argument");
    TMA_HIP(hipSetDevice(h->device));
    copy_u32_kernel<<<dim3((unsigned)ceil_div(h->v.N, 256)), dim3(256), 0, (hipStream_t)stream>>>(h->v.cur_ep, out, h->v.N);
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

int tma_env_episode_log(tma_env *h, int64_t capacity) {
    if (!h || capacity < 0) return fail(TMA_ERR_INVALID, "tma_env_episode_log: null handle or negative capacity");
    TMA_HIP(hipSetDevice(h->device));
    TMA_HIP(hipDeviceSynchronize());  // no step kernel may hold the old view
    EnvView &v = h->v;
    env_free(v.log_ret), env_free(v.log_len), env_free(v.log_env), env_free(v.log_n);
    v.log_ret = nullptr, v.log_len = nullptr, v.log_env = nullptr, v.log_n = nullptr, v.log_cap = 0;
    env_free(h->d_log_ret), env_free(h->d_log_len), env_free(h->d_log_env), env_free(h->d_log_n);  // (the spare set has the old capacity)
    h->d_log_ret = nullptr, h->d_log_len = nullptr, h->d_log_env = nullptr, h->d_log_n = nullptr;
    h->detached = false;  // (a detached, unpopped log goes with its buffers; the detached aggregates are cleared below)
    if (h->d_stats) TMA_HIP(hipMemset(h->d_stats, 0, sizeof(double) * 3 * (size_t)ceil_div(v.N, 256)));
    if (capacity == 0) return TMA_OK;
    TMA_HIP(env_malloc(&v.log_ret, sizeof(double) * (size_t)capacity));
    TMA_HIP(env_malloc(&v.log_len, sizeof(int32_t) * (size_t)capacity));
    TMA_HIP(env_malloc(&v.log_env, sizeof(int32_t) * (size_t)capacity));
    TMA_HIP(env_malloc(&v.log_n, sizeof(unsigned long long)));
    TMA_HIP(hipMemset(v.log_n, 0, sizeof(unsigned long long)));
    v.log_cap = capacity;
    // ... and the spare record buffers of the two-phase pop, for the same reason (this call already drains the device on entry and on exit)
    TMA_HIP(env_malloc(&h->d_log_ret, sizeof(double) * (size_t)capacity));
    TMA_HIP(env_malloc(&h->d_log_len, sizeof(int32_t) * (size_t)capacity));
    TMA_HIP(env_malloc(&h->d_log_env, sizeof(int32_t) * (size_t)capacity));
    TMA_HIP(env_malloc(&h->d_log_n, sizeof(unsigned long long)));
    TMA_HIP(hipMemset(h->d_log_n, 0, sizeof(unsigned long long)));
    TMA_HIP(hipDeviceSynchronize());
    return TMA_OK;
}

int tma_env_pop_episode_log(tma_env *h, double *ret_host, int32_t *len_host, int32_t *env_host, int64_t max_records, int64_t *n_stored, int64_t *n_seen,
                            void *stream) {
    if (!h || !n_stored || !n_seen || max_records < 0 || (max_records > 0 && (!ret_host || !len_host || !env_host)))
        return fail(TMA_ERR_INVALID, "tma_env_pop_episode_log: null argument");
    const EnvView &v = h->v;
    if (!v.log_n) return fail(TMA_ERR_INVALID, "tma_env_pop_episode_log: the episode log is off (tma_env_episode_log)");
    TMA_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long seen = 0;
    TMA_HIP(hipMemcpyAsync(&seen, v.log_n, sizeof(seen), hipMemcpyDeviceToHost, s));
    TMA_HIP(hipStreamSynchronize(s));
    int64_t n = (int64_t)std::min<unsigned long long>(seen, (unsigned long long)v.log_cap);
    if (n > max_records) n = max_records;
    if (n > 0) {
        TMA_HIP(hipMemcpyAsync(ret_host, v.log_ret, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
        TMA_HIP(hipMemcpyAsync(len_host, v.log_len, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
        TMA_HIP(hipMemcpyAsync(env_host, v.log_env, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    }
    TMA_HIP(hipMemsetAsync(v.log_n, 0, sizeof(unsigned long long), s));
    TMA_HIP(hipStreamSynchronize(s));
    *n_stored = n, *n_seen = (int64_t)seen;
    return TMA_OK;
}

// SB3 Monitor rows (reference training.py:85-86 wraps every env in a Monitor that writes `r,l,t` per finished episode): appends n rows
// "round(r, 6),l,round(t, 6)" to `path` the way Python prints them (shortest form of the 6-decimal value: trailing zeros dropped, at least
// one digit behind the point).  Plain host code, no GPU call: ppo.py runs it on a writer thread so that formatting 10^5 rows per
// iteration (4096 envs, ~30-step episodes) never sits between two GPU iterations.
static int fmt6(char *dst, double v) {
    // fast path: |v| < 2^31, and v * 10^6 is not within 10^-3 of a rounding boundary (where only the exact binary value decides, as
    // Python's round and printf do): integer part, six digits, trailing zeros dropped -- a third of snprintf's cost per row
    const double sc = v * 1e6;
    if (sc > -2.0e15 && sc < 2.0e15) {
        const double fl = floor(sc), frac = sc - fl;
        if (frac < 0.499 || frac > 0.501) {
            long long q = (long long)(frac > 0.5 ? fl + 1.0 : fl);
            const bool neg = q < 0 || (q == 0 && (v < 0.0 || (v == 0.0 && std::signbit(v))));
            unsigned long long a = (unsigned long long)(q < 0 ? -q : q);
            unsigned long long ip = a / 1000000ull, fp = a % 1000000ull;
            char tmp[32];
            int k = 0;
            do {
                tmp[k++] = (char)('0' + ip % 10);
                ip /= 10;
            } while (ip);
            int n = 0;
            if (neg) dst[n++] = '-';
            while (k) dst[n++] = tmp[--k];
            dst[n++] = '.';
            int digits = 6;
            while (digits > 1 && fp % 10 == 0) fp /= 10, digits--;
            for (int d = digits - 1; d >= 0; d--) dst[n + d] = (char)('0' + fp % 10), fp /= 10;
            return n + digits;
        }
    }
    int n = snprintf(dst, 40, "%.6f", v);
    while (n > 2 && dst[n - 1] == '0' && dst[n - 2] != '.') n--;
    return n;
}
int tma_monitor_append_rows(const char *path, const double *ret, const int32_t *len, const double *t, int64_t n) {
    if (!path || n < 0 || (n > 0 && (!ret || !len || !t))) return fail(TMA_ERR_INVALID, "tma_monitor_append_rows: null argument");
    FILE *f = fopen(path, "a");
    if (!f) return fail(TMA_ERR_INVALID, "tma_monitor_append_rows: cannot open %s", path);
    std::vector<char> buf(1 << 16);
    size_t used = 0;
    for (int64_t i = 0; i < n; i++) {
        if (used + 128 > buf.size()) {
            fwrite(buf.data(), 1, used, f);
            used = 0;
        }
        used += (size_t)fmt6(buf.data() + used, ret[i]);
        used += (size_t)snprintf(buf.data() + used, 16, ",%d,", (int)len[i]);
        used += (size_t)fmt6(buf.data() + used, t[i]);
        buf[used++] = '\n';
    }
    fwrite(buf.data(), 1, used, f);
    fclose(f);
    return TMA_OK;
}

// Two-phase pop for a training loop that must not drain the GPU between iterations.  tma_env_detach_episode_log is a HOST-side swap of the
// buffer set the kernels are handed at launch: every step / rollout kernel launched before it wrote its Monitor aggregates and episode records
// into the set that is now detached, every later launch writes into the other (empty) set.  tma_env_pop_detached_episode_log then reads the
// detached set on ANY stream that is ordered behind those earlier kernels (e.g. a side stream that waited on an event recorded behind the
// rollout) while the next rollout or the update runs on the compute stream; it empties the set and synchronises only `stream`.
int tma_env_detach_episode_log(tma_env *h) {
    if (!h) return fail(TMA_ERR_INVALID, "tma_env_detach_episode_log: null handle");
    if (h->detached) return fail(TMA_ERR_INVALID, "tma_env_detach_episode_log: the previous detached set has not been popped");
    TMA_HIP(hipSetDevice(h->device));
    EnvView &v = h->v;
    const size_t n_stat = (size_t)ceil_div(v.N, 256) * 3;
    // (both spare sets exist since tma_env_create / tma_env_episode_log; should one be missing, it is made here and the device drained before
    //  it goes live -- a null-stream memset is not ordered against non-blocking streams)
    bool made = false;
    if (!h->d_stats) {
        TMA_HIP(env_malloc(&h->d_stats, sizeof(double) * n_stat));
        TMA_HIP(hipMemset(h->d_stats, 0, sizeof(double) * n_stat));
        made = true;
    }
    if (v.log_n && !h->d_log_n) {
        TMA_HIP(env_malloc(&h->d_log_ret, sizeof(double) * (size_t)v.log_cap));
        TMA_HIP(env_malloc(&h->d_log_len, sizeof(int32_t) * (size_t)v.log_cap));
        TMA_HIP(env_malloc(&h->d_log_env, sizeof(int32_t) * (size_t)v.log_cap));
        TMA_HIP(env_malloc(&h->d_log_n, sizeof(unsigned long long)));
        TMA_HIP(hipMemset(h->d_log_n, 0, sizeof(unsigned long long)));
        made = true;
    }
    if (made) TMA_HIP(hipDeviceSynchronize());
    std::swap(v.stats, h->d_stats);
    if (v.log_n) {
        std::swap(v.log_ret, h->d_log_ret), std::swap(v.log_len, h->d_log_len), std::swap(v.log_env, h->d_log_env), std::swap(v.log_n, h->d_log_n);
    }
    h->detached = true;
    return TMA_OK;
}

int tma_env_pop_detached_episode_log(tma_env *h, double *ret_host, int32_t *len_host, int32_t *env_host, int64_t max_records, int64_t *n_stored,
                                     int64_t *n_seen, double *stats3_host, void *stream) {
    if (!h || !n_stored || !n_seen || !stats3_host || max_records < 0 || (max_records > 0 && (!ret_host || !len_host || !env_host)))
        return fail(TMA_ERR_INVALID, "tma_env_pop_detached_episode_log: null argument");
    if (!h->detached) return fail(TMA_ERR_INVALID, "tma_env_pop_detached_episode_log: nothing is detached (tma_env_detach_episode_log)");
    TMA_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t nb = (size_t)ceil_div(h->v.N, 256);
    std::vector<double> tmp(nb * 3);
    unsigned long long seen = 0;
    TMA_HIP(hipMemcpyAsync(tmp.data(), h->d_stats, sizeof(double) * nb * 3, hipMemcpyDeviceToHost, s));
    if (h->d_log_n) TMA_HIP(hipMemcpyAsync(&seen, h->d_log_n, sizeof(seen), hipMemcpyDeviceToHost, s));
    TMA_HIP(hipStreamSynchronize(s));
    int64_t n = h->d_log_n ? (int64_t)std::min<unsigned long long>(seen, (unsigned long long)h->v.log_cap) : 0;
    if (n > max_records) n = max_records;
    if (n > 0) {
        TMA_HIP(hipMemcpyAsync(ret_host, h->d_log_ret, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
        TMA_HIP(hipMemcpyAsync(len_host, h->d_log_len, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
        TMA_HIP(hipMemcpyAsync(env_host, h->d_log_env, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    }
    if (h->d_log_n) TMA_HIP(hipMemsetAsync(h->d_log_n, 0, sizeof(unsigned long long), s));
    TMA_HIP(hipMemsetAsync(h->d_stats, 0, sizeof(double) * nb * 3, s));
    TMA_HIP(hipStreamSynchronize(s));
    stats3_host[0] = stats3_host[1] = stats3_host[2] = 0.0;
    for (size_t b = 0; b < nb; b++) stats3_host[0] += tmp[3 * b], stats3_host[1] += tmp[3 * b + 1], stats3_host[2] += tmp[3 * b + 2];
    *n_stored = n, *n_seen = (int64_t)seen;
    h->detached = false;
    return TMA_OK;
}

// Stream-ordered clear of the LIVE set (records + Monitor aggregates): no host round trip, no synchronisation.
int tma_env_clear_episode_log(tma_env *h, void *stream) {
    if (!h) return fail(TMA_ERR_INVALID, "tma_env_clear_episode_log: null handle");
    TMA_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->v.log_n) TMA_HIP(hipMemsetAsync(h->v.log_n, 0, sizeof(unsigned long long), s));
    TMA_HIP(hipMemsetAsync(h->v.stats, 0, sizeof(double) * 3 * (size_t)ceil_div(h->v.N, 256), s));
    return TMA_OK;
}

int tma_env_pop_episode_stats(tma_env *h, double *out3_host, void *stream) {
    if (!h || !out3_host) return fail(TMA_ERR_INVALID, "null argument");
    TMA_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t nb = (size_t)ceil_div(h->v.N, 256);
    std::vector<double> tmp(nb * 3);
    TMA_HIP(hipMemcpyAsync(tmp.data(), h->v.stats, sizeof(double) * nb * 3, hipMemcpyDeviceToHost, s));
    TMA_HIP(hipMemsetAsync(h->v.stats, 0, sizeof(double) * nb * 3, s));
    TMA_HIP(hipStreamSynchronize(s));
    out3_host[0] = out3_host[1] = out3_host[2] = 0.0;
    for (size_t b = 0; b < nb; b++) out3_host[0] += tmp[3 * b], out3_host[1] += tmp[3 * b + 1], out3_host[2] += tmp[3 * b + 2];
    return TMA_OK;
}

}  // extern "C"

// tma_tasks.h -- device-side task definitions (state packing, numpy-legacy reset draws, step, obs).
//
// One struct per registry task.  The arithmetic restates, operation by operation, the reference's
// scalar Python (citations relative to /root/reference/); the build is compiled with
// -ffp-contract=off so no multiply-add is ever fused (NumPy never fuses).
//
//   S          register-resident env state
//   SW         packed 32-bit state words per env in HBM (struct-of-arrays: word w of env i at st[w*N+i])
//   RW         32-bit words of one pre-drawn reset record in the reset ring (0: reset needs no MT19937)
//   SDIM       doubles in the flat get/set_state layout (identical to oracle/tma_oracle.c)
#pragma once
#include "tma_common.h"

namespace tma {

// ------------------------------------------------------------------------------------------
// numpy legacy global RNG on device.  MT19937 state lives in a lane-interleaved global scratch
// array (word k of thread t at p[k*stride]) so every access of a wavefront is one coalesced row.
// Outputs are generated lazily in place (bit-identical to the bulk twist).
// Call sites restated: np.random.seed / shuffle / choice / randint / uniform as used by
// backend/mlagents/envs.py:117-119, backend/examples/gridworld.py:45,50, push.py:41,46, ball3d.py:49-57.
// ------------------------------------------------------------------------------------------
struct MT {
    uint32_t *p;
    int64_t stride;
    int idx;
    __device__ __forceinline__ uint32_t &at(int k) { return p[(int64_t)k * stride]; }
    __device__ void seed(uint32_t s) {  // init_genrand
        uint32_t prev = s;
        at(0) = prev;
        for (int k = 1; k < 624; k++) {
            prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)k;
            at(k) = prev;
        }
        idx = 0;
    }
    __device__ uint32_t next() {
        int k = idx;
        int k1 = (k == 623) ? 0 : k + 1;
        int km = (k < 227) ? k + 397 : k - 227;
        uint32_t y = (at(k) & 0x80000000u) | (at(k1) & 0x7fffffffu);
        uint32_t v = at(km) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        at(k) = v;
        idx = (k == 623) ? 0 : k + 1;
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        return v;
    }
    __device__ uint32_t interval(uint32_t max) {  // legacy rk_interval (masked rejection)
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        uint32_t v;
        do {
            v = next() & mask;
        } while (v > max);
        return v;
    }
    __device__ double dbl() {  // legacy rk_double
        uint32_t a = next() >> 5, b = next() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ double uniform(double lo, double hi) {
        double scale = hi - lo;
        double pr = scale * dbl();
        return lo + pr;
    }
};

__device__ __forceinline__ uint32_t episode_seed(uint32_t base, uint32_t gi, uint32_t ep) { return base + gi + ep * TMA_EP_STRIDE; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Fisher-Yates of numpy's legacy shuffle, reduced to what the envs read: draw j_i for i = n-1..1 (kept in an
// LDS byte column per thread), then walk the swaps backwards to find which original cell ends at position p.
template <int NCELL>
struct ShuffleTrace {
    uint8_t *js;  // LDS column base for this thread, element k at js[k * blockDim.x]
    int bs;
    __device__ void draw(MT &mt) {
        for (int i = NCELL - 1; i >= 1; i--) js[(i - 1) * bs] = (uint8_t)mt.interval((uint32_t)i);
    }
    __device__ int final_at(int p) const {
        int c = p;
        for (int i = 1; i <= NCELL - 1; i++) {
            int j = js[(i - 1) * bs];
            c = (c == i) ? j : ((c == j) ? i : c);
        }
        return c;
    }
};


// ------------------------------------------------------------------------------------------
// Register-resident fast path for the first W (< 227) outputs after np.random.seed(s).
// Output k of a fresh MT19937 state depends only on init words k, k+1 and k+397, and init_genrand is a serial chain, so:
// keep words 0..W in registers (static indices, fully unrolled), run the chain through word 396 without storing, then for
// i = 397.. generate word i, combine it with words k and k+1 (k = i - 397), temper, and hand the output straight to the
// task's consumer (a small state machine that replays the legacy shuffle / choice / randint / uniform draws).
// No memory traffic at all; if a consumer has not finished after W outputs (rejection sampling is unbounded) the caller
// falls back to the general in-memory generator (struct MT), so the result is exact for every seed.
// ------------------------------------------------------------------------------------------
template <int W, class C>
__device__ __forceinline__ bool mt_stream(uint32_t seed, C &c) {
    static_assert(W >= 1 && W <= 226, "fast path covers the outputs that need no regenerated word");
    uint32_t r[W + 1];
    uint32_t x = seed;
    r[0] = x;
#pragma unroll
    for (int i = 1; i <= W; i++) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        r[i] = x;
    }
#pragma unroll 4
    for (int i = W + 1; i < 397; i++) x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
#pragma unroll
    for (int k = 0; k < W; k++) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)(397 + k);
        const uint32_t y = (r[k] & 0x80000000u) | (r[k + 1] & 0x7fffffffu);
        uint32_t v = x ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        c.feed(v);
    }
    return c.done();
}

__device__ __forceinline__ uint32_t interval_mask(uint32_t m) {
    return m >= 32 ? 63u : m >= 16 ? 31u : m >= 8 ? 15u : m >= 4 ? 7u : m >= 2 ? 3u : 1u;
}

// ==========================================================================================
// Basic -- backend/mlagents/envs.py:17-27 (constants, one-hot), :48-58 (reset), :60-81 (step)
// ==========================================================================================
struct BasicTask {
    static constexpr int ID = TMA_TASK_BASIC, OBS = 21, NACT = 3, ADIM = 1, MAXSTEPS = 50, SW = 1, RW = 0, SDIM = 2;
    static constexpr bool USES_MT = false, NATIVE_TRUNC_RULE = true;
    static constexpr bool FUSED_ROLLOUT = true;  // the fused rollout-chunk kernels are instantiated for this task (tma_rollout.hip)
    struct S {
        int pos, steps;
    };
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) {
        uint32_t w = st[i];
        s.pos = w & 31;
        s.steps = (w >> 5) & 127;
    }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) { st[i] = (uint32_t)s.pos | ((uint32_t)s.steps << 5); }
    __device__ static void reset_inline(uint32_t, S &s) {
        s.pos = 10;
        s.steps = 0;
    }
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static void step(S &s, int a, const float *, double &r, bool &done) {
        int delta = a - 1;  // (-1, 0, 1)[a]
        s.pos = clampi(s.pos + delta, 0, 20);
        s.steps += 1;
        double rew = -0.01;
        bool term = false;
        if (s.pos == 7) {
            rew = rew + 0.1;
            term = true;
        } else if (s.pos == 17) {
            rew = rew + 1.0;
            term = true;
        }
        r = rew;
        done = term;
    }
    __device__ static void obs(const S &s, float *o) {
#pragma unroll
        for (int k = 0; k < 21; k++) o[k] = (k == s.pos) ? 1.0f : 0.0f;
    }
    __device__ static void to_flat(const S &s, double *f) {
        f[0] = s.pos;
        f[1] = s.steps;
    }
    __device__ static void from_flat(const double *f, S &s) {
        s.pos = clampi((int)f[0], 0, 20);
        s.steps = (int)f[1];
    }
};

// ==========================================================================================
// GridWorld -- backend/examples/gridworld.py:14-30 (constants), :40-52 (reset), :55-64 (obs), :67-95 (step)
// ==========================================================================================
struct GridTask {
    static constexpr int ID = TMA_TASK_GRIDWORLD, OBS = 4, NACT = 5, ADIM = 1, MAXSTEPS = 100, SW = 1, RW = 1, SDIM = 8;
    static constexpr bool USES_MT = true, NATIVE_TRUNC_RULE = false;
    static constexpr bool FUSED_ROLLOUT = true;  // the fused rollout-chunk kernels are instantiated for this task (tma_rollout.hip)
    struct S {
        int ax, ay, gx, gy, rx, ry, gt, steps;
    };
    __device__ static void from_word(uint32_t w, S &s) {
        s.ax = w & 7;
        s.ay = (w >> 3) & 7;
        s.gx = (w >> 6) & 7;
        s.gy = (w >> 9) & 7;
        s.rx = (w >> 12) & 7;
        s.ry = (w >> 15) & 7;
        s.gt = (w >> 18) & 1;
        s.steps = (w >> 19) & 255;
    }
    __device__ static uint32_t to_word(const S &s) {
        return (uint32_t)s.ax | ((uint32_t)s.ay << 3) | ((uint32_t)s.gx << 6) | ((uint32_t)s.gy << 9) | ((uint32_t)s.rx << 12) |
               ((uint32_t)s.ry << 15) | ((uint32_t)s.gt << 18) | ((uint32_t)s.steps << 19);
    }
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) { from_word(st[i], s); }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) { st[i] = to_word(s); }
    __device__ static void from_rec(const uint32_t *rec, S &s) { from_word(rec[0], s); }
    // one GridWorldEnv.reset(): shuffle 25 cells (x-major), take cells 0,1,2, then choice([0,1])
    __device__ static void draw(MT &mt, uint8_t *lds_col, int bs, uint32_t *rec) {
        ShuffleTrace<25> sh{lds_col, bs};
        sh.draw(mt);
        uint32_t gt = mt.interval(1);
        int a = sh.final_at(0), g = sh.final_at(1), r = sh.final_at(2);
        S s{a / 5, a % 5, g / 5, g % 5, r / 5, r % 5, (int)gt, 0};
        rec[0] = to_word(s);
    }

    // consumer of the two resets an adapter.reset(seed) performs: 24 shuffle draws + choice, twice; the second one counts
    struct Fast {
        static constexpr int W = 128, W_SMALL = 58;
        int a = 0;
        uint64_t lo = 0, hi = 0;
        uint32_t gt = 0;
        __device__ __forceinline__ bool done() const { return a >= 50; }
        __device__ __forceinline__ void feed(uint32_t v) {
            if (a >= 50) return;
            const int q = a < 25 ? a : a - 25;
            const uint32_t m = q < 24 ? (uint32_t)(24 - q) : 1u;
            const uint32_t u = v & interval_mask(m);
            if (u <= m) {
                if (a >= 25) {
                    if (q < 12) lo |= (uint64_t)u << (5 * q);
                    else if (q < 24) hi |= (uint64_t)u << (5 * (q - 12));
                    else gt = u;
                }
                a++;
            }
        }
        __device__ __forceinline__ int final_at(int p) const {
            int c = p;
#pragma unroll
            for (int i = 1; i <= 24; i++) {
                const int q = 24 - i;
                const int j = (int)((q < 12 ? (lo >> (5 * q)) : (hi >> (5 * (q - 12)))) & 31u);
                c = (c == i) ? j : ((c == j) ? i : c);
            }
            return c;
        }
        __device__ __forceinline__ void finish(uint32_t *rec) const {
            const int a0 = final_at(0), g0 = final_at(1), r0 = final_at(2);
            S s{a0 / 5, a0 % 5, g0 / 5, g0 % 5, r0 / 5, r0 % 5, (int)gt, 0};
            rec[0] = to_word(s);
        }
    };
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static void step(S &s, int a, const float *, double &r, bool &done) {
        int dx = (a == 4) - (a == 3), dy = (a == 1) - (a == 2);
        s.ax = clampi(s.ax + dx, 0, 4);
        s.ay = clampi(s.ay + dy, 0, 4);
        s.steps += 1;
        double rew = -0.01;
        bool d = false;
        if (s.ax == s.gx && s.ay == s.gy) {
            rew = (s.gt == 0) ? 1.0 : -1.0;
            d = true;
        } else if (s.ax == s.rx && s.ay == s.ry) {
            rew = (s.gt == 1) ? 1.0 : -1.0;
            d = true;
        }
        if (s.steps >= 100) d = true;
        r = rew;
        done = d;
    }
    __device__ static void obs(const S &s, float *o) {
        int tx = s.gt == 0 ? s.gx : s.rx, ty = s.gt == 0 ? s.gy : s.ry;
        o[0] = (float)((double)(tx - s.ax) / 4.0);
        o[1] = (float)((double)(ty - s.ay) / 4.0);
        o[2] = s.gt == 0 ? 1.0f : 0.0f;
        o[3] = s.gt == 0 ? 0.0f : 1.0f;
    }
    __device__ static void to_flat(const S &s, double *f) {
        f[0] = s.ax, f[1] = s.ay, f[2] = s.gx, f[3] = s.gy, f[4] = s.rx, f[5] = s.ry, f[6] = s.gt, f[7] = s.steps;
    }
    __device__ static void from_flat(const double *f, S &s) {
        s = S{(int)f[0], (int)f[1], (int)f[2], (int)f[3], (int)f[4], (int)f[5], (int)f[6], (int)f[7]};
    }
};

// ==========================================================================================
// Push -- backend/examples/push.py:10-24 (constants), :39-50 (reset), :53-59 (obs), :62-125 (step)
// ==========================================================================================
struct PushTask {
    static constexpr int ID = TMA_TASK_PUSH, OBS = 4, NACT = 5, ADIM = 1, MAXSTEPS = 120, SW = 1, RW = 1, SDIM = 6;
    static constexpr bool USES_MT = true, NATIVE_TRUNC_RULE = false;
    static constexpr bool FUSED_ROLLOUT = true;  // the fused rollout-chunk kernels are instantiated for this task (tma_rollout.hip)
    struct S {
        int ax, ay, bx, by, gx, steps;
    };
    __device__ static void from_word(uint32_t w, S &s) {
        s.ax = w & 7;
        s.ay = (w >> 3) & 7;
        s.bx = (w >> 6) & 7;
        s.by = (w >> 9) & 7;
        s.gx = (w >> 12) & 7;
        s.steps = (w >> 15) & 255;
    }
    __device__ static uint32_t to_word(const S &s) {
        return (uint32_t)s.ax | ((uint32_t)s.ay << 3) | ((uint32_t)s.bx << 6) | ((uint32_t)s.by << 9) | ((uint32_t)s.gx << 12) |
               ((uint32_t)s.steps << 15);
    }
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) { from_word(st[i], s); }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) { st[i] = to_word(s); }
    __device__ static void from_rec(const uint32_t *rec, S &s) { from_word(rec[0], s); }
    __device__ static void draw(MT &mt, uint8_t *lds_col, int bs, uint32_t *rec) {
        ShuffleTrace<36> sh{lds_col, bs};
        sh.draw(mt);
        uint32_t gx = mt.interval(5);  // np.random.randint(0, 6)
        int a = sh.final_at(0), b = sh.final_at(1);
        S s{a / 6, a % 6, b / 6, b % 6, (int)gx, 0};
        rec[0] = to_word(s);
    }

    // two resets of 35 shuffle draws + randint(0, 6); the second one counts
    struct Fast {
        static constexpr int W = 176, W_SMALL = 84;
        int a = 0;
        uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        uint32_t gx = 0;
        __device__ __forceinline__ bool done() const { return a >= 72; }
        __device__ __forceinline__ void feed(uint32_t v) {
            if (a >= 72) return;
            const int q = a < 36 ? a : a - 36;
            const uint32_t m = q < 35 ? (uint32_t)(35 - q) : 5u;
            const uint32_t u = v & interval_mask(m);
            if (u <= m) {
                if (a >= 36) {
                    if (q < 10) w0 |= (uint64_t)u << (6 * q);
                    else if (q < 20) w1 |= (uint64_t)u << (6 * (q - 10));
                    else if (q < 30) w2 |= (uint64_t)u << (6 * (q - 20));
                    else if (q < 35) w3 |= (uint64_t)u << (6 * (q - 30));
                    else gx = u;
                }
                a++;
            }
        }
        __device__ __forceinline__ int final_at(int p) const {
            int c = p;
#pragma unroll
            for (int i = 1; i <= 35; i++) {
                const int q = 35 - i;
                const uint64_t w = q < 10 ? w0 : q < 20 ? w1 : q < 30 ? w2 : w3;
                const int j = (int)((w >> (6 * (q % 10))) & 63u);
                c = (c == i) ? j : ((c == j) ? i : c);
            }
            return c;
        }
        __device__ __forceinline__ void finish(uint32_t *rec) const {
            const int a0 = final_at(0), b0 = final_at(1);
            S s{a0 / 6, a0 % 6, b0 / 6, b0 % 6, (int)gx, 0};
            rec[0] = to_word(s);
        }
    };
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static int iabs(int v) { return v < 0 ? -v : v; }
    __device__ static void step(S &s, int a, const float *, double &r, bool &done) {
        int dx = (a == 4) - (a == 3), dy = (a == 1) - (a == 2);
        int nax = clampi(s.ax + dx, 0, 5), nay = clampi(s.ay + dy, 0, 5);
        int nbx = s.bx, nby = s.by;
        int prev_bg = iabs(s.gx - s.bx) + iabs(5 - s.by);
        int prev_ab = iabs(s.bx - s.ax) + iabs(s.by - s.ay);
        bool invalid = false;
        if (nax == s.bx && nay == s.by) {
            int tx = s.bx + dx, ty = s.by + dy;
            if (0 <= tx && tx < 6 && 0 <= ty && ty < 6) {
                nbx = tx;
                nby = ty;
            } else {
                nax = s.ax;
                nay = s.ay;
                invalid = true;
            }
        }
        s.ax = nax, s.ay = nay, s.bx = nbx, s.by = nby;
        s.steps += 1;
        int bg = iabs(s.gx - nbx) + iabs(5 - nby);
        int ab = iabs(nbx - nax) + iabs(nby - nay);
        double rew = -0.01;
        double t1 = 0.05 * (double)(prev_ab - ab);
        rew = rew + t1;
        double t2 = 0.3 * (double)(prev_bg - bg);
        rew = rew + t2;
        if (invalid) rew = rew - 0.05;
        bool d = false;
        if (nby == 5) {
            rew = 1.0;
            d = true;
        }
        if (s.steps >= 120) d = true;
        r = rew;
        done = d;
    }
    __device__ static void obs(const S &s, float *o) {
        o[0] = (float)((double)(s.bx - s.ax) / 5.0);
        o[1] = (float)((double)(s.by - s.ay) / 5.0);
        o[2] = (float)((double)(s.gx - s.bx) / 5.0);
        o[3] = (float)((double)(5 - s.by) / 5.0);
    }
    __device__ static void to_flat(const S &s, double *f) { f[0] = s.ax, f[1] = s.ay, f[2] = s.bx, f[3] = s.by, f[4] = s.gx, f[5] = s.steps; }
    __device__ static void from_flat(const double *f, S &s) { s = S{(int)f[0], (int)f[1], (int)f[2], (int)f[3], (int)f[4], (int)f[5]}; }
};

// ==========================================================================================
// Ball3D -- backend/examples/ball3d.py:10-38 (constants), :47-59 (reset), :61-72 (obs), :74-113 (step)
// mixed f32/f64 recipe: SURVEY.md Appendix A.3.
// ==========================================================================================
struct BallTask {
    static constexpr int ID = TMA_TASK_BALL3D, OBS = 6, NACT = 5, ADIM = 1, MAXSTEPS = 200, SW = 9, RW = 6, SDIM = 8;
    static constexpr bool USES_MT = true, NATIVE_TRUNC_RULE = false;
    static constexpr bool FUSED_ROLLOUT = true;  // the fused rollout-chunk kernels are instantiated for this task (tma_rollout.hip)
    static constexpr double MAX_TILT = 0.4363323129985824;     // np.deg2rad(25.0)
    static constexpr double TILT_DELTA = 0.05235987755982989;  // np.deg2rad(3.0)
    struct S {
        double rot[2];
        float pos[2], vel[2];
        int steps;
        bool first;
    };
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) {
        for (int k = 0; k < 2; k++) {
            uint32_t lo = st[(2 * k) * N + i], hi = st[(2 * k + 1) * N + i];
            s.rot[k] = __longlong_as_double(((long long)hi << 32) | lo);
        }
        s.pos[0] = __uint_as_float(st[4 * N + i]);
        s.pos[1] = __uint_as_float(st[5 * N + i]);
        s.vel[0] = __uint_as_float(st[6 * N + i]);
        s.vel[1] = __uint_as_float(st[7 * N + i]);
        uint32_t w = st[8 * N + i];
        s.steps = w & 0xffff;
        s.first = (w >> 16) & 1;
    }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) {
        for (int k = 0; k < 2; k++) {
            unsigned long long b = (unsigned long long)__double_as_longlong(s.rot[k]);
            st[(2 * k) * N + i] = (uint32_t)b;
            st[(2 * k + 1) * N + i] = (uint32_t)(b >> 32);
        }
        st[4 * N + i] = __float_as_uint(s.pos[0]);
        st[5 * N + i] = __float_as_uint(s.pos[1]);
        st[6 * N + i] = __float_as_uint(s.vel[0]);
        st[7 * N + i] = __float_as_uint(s.vel[1]);
        st[8 * N + i] = (uint32_t)s.steps | ((uint32_t)s.first << 16);
    }
    __device__ static void from_rec(const uint32_t *rec, S &s) {
        s.rot[0] = (double)__uint_as_float(rec[0]);
        s.rot[1] = (double)__uint_as_float(rec[1]);
        s.pos[0] = __uint_as_float(rec[2]);
        s.pos[1] = __uint_as_float(rec[3]);
        s.vel[0] = __uint_as_float(rec[4]);
        s.vel[1] = __uint_as_float(rec[5]);
        s.steps = 0;
        s.first = true;
    }
    __device__ static void draw(MT &mt, uint8_t *, int, uint32_t *rec) {
        const double half = MAX_TILT * 0.5;
        rec[0] = __float_as_uint((float)mt.uniform(-half, half));
        rec[1] = __float_as_uint((float)mt.uniform(-half, half));
        rec[2] = __float_as_uint((float)mt.uniform(-1.5, 1.5));
        rec[3] = __float_as_uint((float)mt.uniform(-1.5, 1.5));
        rec[4] = __float_as_uint((float)mt.uniform(-1.0, 1.0));
        rec[5] = __float_as_uint((float)mt.uniform(-1.0, 1.0));
    }

    // two resets of 6 uniform doubles (12 outputs each); the second one counts
    struct Fast {
        static constexpr int W = 24, W_SMALL = 24;
        int k = 0;
        uint32_t ah = 0;
        uint32_t rec6[6] = {0, 0, 0, 0, 0, 0};
        __device__ __forceinline__ bool done() const { return k >= 24; }
        __device__ __forceinline__ void feed(uint32_t v) {
            if (k >= 12) {
                if ((k & 1) == 0) {
                    ah = v >> 5;
                } else {
                    const double d = ((double)ah * 67108864.0 + (double)(v >> 6)) / 9007199254740992.0;
                    const int idx = (k - 12) >> 1;
                    const double half = MAX_TILT * 0.5;
                    const double lo = idx < 2 ? -half : (idx < 4 ? -1.5 : -1.0);
                    const double hi = idx < 2 ? half : (idx < 4 ? 1.5 : 1.0);
                    const double scale = hi - lo;
                    const double pr = scale * d;
                    const uint32_t bits = __float_as_uint((float)(lo + pr));
#pragma unroll
                    for (int j = 0; j < 6; j++)
                        if (j == idx) rec6[j] = bits;
                }
            }
            k++;
        }
        __device__ __forceinline__ void finish(uint32_t *rec) const {
#pragma unroll
            for (int j = 0; j < 6; j++) rec[j] = rec6[j];
        }
    };
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static void step(S &s, int a, const float *, double &r, bool &done) {
        double del[2];
        del[0] = (a == 0) ? TILT_DELTA : ((a == 1) ? -TILT_DELTA : 0.0);
        del[1] = (a == 2) ? TILT_DELTA : ((a == 3) ? -TILT_DELTA : 0.0);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            double rr = s.rot[k] + del[k];
            if (s.first) rr = (double)(float)rr;  // rot += delta in place on the float32 array (first step after reset)
            rr = rr < -MAX_TILT ? -MAX_TILT : rr;
            rr = rr > MAX_TILT ? MAX_TILT : rr;
            s.rot[k] = rr;
            double acc = 9.81 * sin(rr);
            double accdt = acc * 0.02;
            float v = (float)((double)s.vel[k] + accdt);
            v = v * 0.98f;
            float stp = v * 0.02f;
            s.vel[k] = v;
            s.pos[k] = s.pos[k] + stp;
        }
        s.steps += 1;
        s.first = false;
        bool off = (fabsf(s.pos[0]) > 3.0f) || (fabsf(s.pos[1]) > 3.0f);
        bool timeout = s.steps >= 200;
        bool d = off || timeout;
        float s0 = s.pos[0] * s.pos[0], s1 = s.pos[1] * s.pos[1];
        float norm = sqrtf(s0 + s1);
        float q = norm / 3.0f;
        float rew = 1.0f - q;
        if (d) rew = (timeout && !off) ? 1.0f : -1.0f;
        float pen = -0.02f * norm;
        rew = rew + pen;
        r = (double)rew;
        done = d;
    }
    __device__ static void obs(const S &s, float *o) {
        o[0] = (float)s.rot[0];
        o[1] = (float)s.rot[1];
        o[2] = s.pos[0];
        o[3] = s.pos[1];
        o[4] = s.vel[0];
        o[5] = s.vel[1];
    }
    __device__ static void to_flat(const S &s, double *f) {
        f[0] = s.rot[0], f[1] = s.rot[1], f[2] = s.pos[0], f[3] = s.pos[1], f[4] = s.vel[0], f[5] = s.vel[1], f[6] = s.steps, f[7] = s.first ? 1.0 : 0.0;
    }
    __device__ static void from_flat(const double *f, S &s) {
        s.rot[0] = f[0], s.rot[1] = f[1];
        s.pos[0] = (float)f[2], s.pos[1] = (float)f[3], s.vel[0] = (float)f[4], s.vel[1] = (float)f[5];
        s.steps = (int)f[6];
        s.first = f[7] != 0.0;
    }
};


// ==========================================================================================
// WallJump -- backend/examples/walljump.py:14-20 (constants), :40-45 (reset), :48-53 (obs), :56-98 (step);
// adapter: backend/mlagents/envs.py:202-211 (Discrete(4), Box(-1,1,(4,)), 150-step limit).  SURVEY.md §8f rank N3.
// ==========================================================================================
struct WallJumpTask {
    static constexpr int ID = TMA_TASK_WALLJUMP, OBS = 4, NACT = 4, ADIM = 1, MAXSTEPS = 150, SW = 1, RW = 1, SDIM = 4;
    static constexpr bool USES_MT = true, NATIVE_TRUNC_RULE = false;
    static constexpr bool FUSED_ROLLOUT = true;  // the fused rollout-chunk kernels are instantiated for this task (tma_rollout.hip)
    struct S {
        int x, in_air, wall, steps;
    };
    __device__ static void from_word(uint32_t w, S &s) {
        s.x = w & 31;
        s.in_air = (w >> 5) & 3;
        s.wall = (w >> 7) & 1;
        s.steps = (w >> 8) & 255;
    }
    __device__ static uint32_t to_word(const S &s) { return (uint32_t)s.x | ((uint32_t)s.in_air << 5) | ((uint32_t)s.wall << 7) | ((uint32_t)s.steps << 8); }
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) { from_word(st[i], s); }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) { st[i] = to_word(s); }
    __device__ static void from_rec(const uint32_t *rec, S &s) { s = S{0, 0, (int)(rec[0] & 1u), 0}; }
    __device__ static void draw(MT &mt, uint8_t *, int, uint32_t *rec) { rec[0] = mt.dbl() < 0.7 ? 1u : 0u; }  // int(np.random.rand() < 0.7)
    // two resets of one rk_double (2 outputs each); the second one counts
    struct Fast {
        static constexpr int W = 4, W_SMALL = 4;
        int k = 0;
        uint32_t ah = 0, wall = 0;
        __device__ __forceinline__ bool done() const { return k >= 4; }
        __device__ __forceinline__ void feed(uint32_t v) {
            if (k == 2) ah = v >> 5;
            if (k == 3) wall = (((double)ah * 67108864.0 + (double)(v >> 6)) / 9007199254740992.0) < 0.7 ? 1u : 0u;
            k++;
        }
        __device__ __forceinline__ void finish(uint32_t *rec) const { rec[0] = wall; }
    };
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static void step(S &s, int a, const float *, double &r, bool &done) {
        double rew = -0.01;
        bool d = false, just_jumped = false;
        if (a == 3 && s.in_air == 0) {
            s.in_air = 3;
            just_jumped = true;
        }
        const int dx = (a == 1 || a == 3) ? 1 : (a == 2 ? -1 : 0);
        int px = clampi(s.x + dx, 0, 19);
        const bool crossing = (s.x < 10 && 10 <= px) || (px < 10 && 10 <= s.x);
        if (crossing && s.wall == 1 && s.in_air == 0) {
            px = s.x;
            rew = rew - 0.02;
        }
        const int dw = 10 - s.x;
        if (just_jumped && !crossing && (dw < 0 ? -dw : dw) > 1) rew = rew - 0.03;
        s.x = px;
        if (s.in_air > 0) s.in_air -= 1;
        if (s.x == 19) {
            rew = 1.0;
            d = true;
        }
        s.steps += 1;
        if (s.steps >= 150) d = true;
        r = rew;
        done = d;
    }
    __device__ static void obs(const S &s, float *o) {
        o[0] = (float)((double)(19 - s.x) / 19.0);
        o[1] = (float)((double)(10 - s.x) / 19.0);
        o[2] = (float)s.wall;
        o[3] = s.in_air == 0 ? 1.0f : 0.0f;
    }
    __device__ static void to_flat(const S &s, double *f) { f[0] = s.x, f[1] = s.in_air, f[2] = s.wall, f[3] = s.steps; }
    __device__ static void from_flat(const double *f, S &s) { s = S{(int)f[0], (int)f[1], (int)f[2], (int)f[3]}; }
};

// ---- double <-> state / record words (SoA planes of 32-bit words) ----
__device__ __forceinline__ double words_to_double(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
__device__ __forceinline__ void double_to_words(double d, uint32_t &lo, uint32_t &hi) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    lo = (uint32_t)b;
    hi = (uint32_t)(b >> 32);
}
// legacy rk_double from its two tempered outputs, and np.random.uniform(lo, hi) on top of it (two roundings, as numpy's legacy path)
__device__ __forceinline__ double mt_double_from(uint32_t a_hi27, uint32_t v2) { return ((double)a_hi27 * 67108864.0 + (double)(v2 >> 6)) / 9007199254740992.0; }
__device__ __forceinline__ double uniform_from(double lo, double hi, double d) {
    const double scale = hi - lo;
    const double pr = scale * d;
    return lo + pr;
}
// numpy / OpenBLAS summation orders of the float tasks below (probed, pinned by the fixtures; see oracle/tma_oracle.c): np.dot,
// np.linalg.norm and `@` on 2- and 3-element float64 operands are one FMA chain from the left; the 3x3 matrix-vector product
// starts from the middle column.
__device__ __forceinline__ double dot2_np(double a0, double b0, double a1, double b1) { return fma(a1, b1, a0 * b0); }
__device__ __forceinline__ double dot3_np(const double *a, const double *b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ==========================================================================================
// Bicycle -- backend/examples/bicycle.py:14-37 (constants), :40-58 (reset), :60-125 (step), :127-141 (obs); adapter
// backend/mlagents/envs.py:228-239 (Discrete(3), unbounded Box(7), 2000-step limit).  SURVEY.md 8f rank N3.
// float64 physics with sin / cos / tan / pow from the device math library: parity <= 1e-5 (observed: identical to ~1e-15), the
// oracle (libm) is bit-exact against the reference fixtures.
// ==========================================================================================
struct BicycleTask {
    static constexpr int ID = TMA_TASK_BICYCLE, OBS = 7, NACT = 3, ADIM = 1, MAXSTEPS = 2000, SW = 20, RW = 8, SDIM = 10;  // (SW: 19 words used, sizeof(S) / 4 = 20)
    static constexpr bool USES_MT = true, NATIVE_TRUNC_RULE = false, FUSED_ROLLOUT = true;
    static constexpr double MAX_PHI = 0.78539816339744828, MAX_DELTA = 0.52359877559829882;  // np.pi / 4, np.pi / 6
    struct S {
        double x, z, theta, phi, phi_dot, delta, gx, gz, dist;
        int steps;
    };
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) {
        double *d = &s.x;
#pragma unroll
        for (int k = 0; k < 9; k++) d[k] = words_to_double(st[(2 * k) * N + i], st[(2 * k + 1) * N + i]);
        s.steps = (int)st[18 * N + i];
    }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) {
        const double *d = &s.x;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            uint32_t lo, hi;
            double_to_words(d[k], lo, hi);
            st[(2 * k) * N + i] = lo;
            st[(2 * k + 1) * N + i] = hi;
        }
        st[18 * N + i] = (uint32_t)s.steps;
    }
    // record = {phi, phi_dot, goal_x, goal_z} as doubles
    __device__ static void make_rec(double phi, double phi_dot, double radius, double angle, uint32_t *rec) {
        double_to_words(phi, rec[0], rec[1]);
        double_to_words(phi_dot, rec[2], rec[3]);
        double_to_words(radius * cos(angle), rec[4], rec[5]);
        double_to_words(radius * sin(angle), rec[6], rec[7]);
    }
    __device__ static void from_rec(const uint32_t *rec, S &s) {
        s.x = s.z = s.theta = s.delta = 0.0;
        s.phi = words_to_double(rec[0], rec[1]);
        s.phi_dot = words_to_double(rec[2], rec[3]);
        s.gx = words_to_double(rec[4], rec[5]);
        s.gz = words_to_double(rec[6], rec[7]);
        s.dist = sqrt(dot2_np(s.gx, s.gx, s.gz, s.gz));
        s.steps = 0;
    }
    __device__ static void draw(MT &mt, uint8_t *, int, uint32_t *rec) {
        const double phi = mt.uniform(-0.1, 0.1), phi_dot = mt.uniform(-0.1, 0.1);
        const double radius = mt.uniform(15, 25), angle = mt.uniform(-MAX_PHI, MAX_PHI);
        make_rec(phi, phi_dot, radius, angle, rec);
    }
    // two resets of four uniform doubles (8 outputs each); the second one counts
    struct Fast {
        static constexpr int W = 16, W_SMALL = 16;
        int k = 0;
        uint32_t ah = 0;
        double u[4] = {0.0, 0.0, 0.0, 0.0};
        __device__ __forceinline__ bool done() const { return k >= 16; }
        __device__ __forceinline__ void feed(uint32_t v) {
            if (k >= 8) {
                if ((k & 1) == 0) {
                    ah = v >> 5;
                } else {
                    const double d = mt_double_from(ah, v);
                    const int idx = (k - 8) >> 1;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (j == idx) u[j] = d;
                }
            }
            k++;
        }
        __device__ __forceinline__ void finish(uint32_t *rec) const {
            make_rec(uniform_from(-0.1, 0.1, u[0]), uniform_from(-0.1, 0.1, u[1]), uniform_from(15, 25, u[2]), uniform_from(-MAX_PHI, MAX_PHI, u[3]), rec);
        }
    };
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static void step(S &s, int a, const float *, double &r, bool &done) {
        const double g = 9.8, h = 0.8, L = 1.0, v = 5.0, dt = 0.02;
        s.steps += 1;
        double delta = s.delta + (a == 0 ? -0.05 : (a == 2 ? 0.05 : 0.0));
        delta = clampd(delta, -MAX_DELTA, MAX_DELTA);
        const double grav = (g / h) * sin(s.phi);
        double cen = (v * v / (L * h)) * tan(delta);
        cen = cen * cos(s.phi);
        const double phi_ddot = grav - cen;
        s.phi_dot = s.phi_dot + phi_ddot * dt;
        s.phi = s.phi + s.phi_dot * dt;
        delta = delta * 0.95;
        s.delta = delta;
        const double th = (v / L) * tan(delta);
        s.theta = s.theta + th * dt;
        const double ct = cos(s.theta), sn = sin(s.theta);
        const double cx = v * ct;
        s.x = s.x + cx * dt;
        const double cz = v * sn;
        s.z = s.z + cz * dt;
        const double dx = s.gx - s.x, dz = s.gz - s.z;
        const double nd = sqrt(dot2_np(dx, dx, dz, dz));
        const double progress = (s.dist - nd) * 10.0;
        s.dist = nd;
        const double upright = (1.0 - sqrt(fabs(s.phi) / MAX_PHI)) * 0.2;  // reference: (...) ** 0.5 through libm pow, <= 1 ulp from sqrt
        const double den = nd > 0 ? nd : 1.0;
        const double heading = dot2_np(ct, dx / den, sn, dz / den) * 0.3;
        const double steer = -(fabs(delta) / MAX_DELTA) * 0.1;
        double rew = progress + upright;
        rew = rew + heading;
        rew = rew + steer;
        bool d = false;
        if (fabs(s.phi) > MAX_PHI) rew = -10.0, d = true;
        if (s.steps > 2000) d = true;
        if (nd < 2.0) rew = 50.0, d = true;
        r = rew;
        done = d;
    }
    __device__ static void obs(const S &s, float *o) {
        const double dx = s.gx - s.x, dz = s.gz - s.z;
        const double dist = sqrt(dot2_np(dx, dx, dz, dz));
        double nx = 0.0, nz = 0.0;
        if (dist > 0) nx = dx / dist, nz = dz / dist;
        o[0] = (float)s.phi;
        o[1] = (float)s.phi_dot;
        o[2] = (float)s.delta;
        o[3] = (float)cos(s.theta);
        o[4] = (float)sin(s.theta);
        o[5] = (float)nx;
        o[6] = (float)nz;
    }
    __device__ static void to_flat(const S &s, double *f) {
        const double *d = &s.x;
        for (int k = 0; k < 9; k++) f[k] = d[k];
        f[9] = s.steps;
    }
    __device__ static void from_flat(const double *f, S &s) {
        double *d = &s.x;
        for (int k = 0; k < 9; k++) d[k] = f[k];
        s.steps = (int)f[9];
    }
};

// ==========================================================================================
// BrickBreak -- backend/examples/brick_break.py:14-37 (constants), :39-47 (reset), :49-121 (step), :123-131 (obs); adapter
// backend/mlagents/envs.py:214-225 (Discrete(3), unbounded Box(45), 2000-step limit).  The step is plain float64 add / multiply / compare
// (bit-exact); the reset takes cos / sin of one uniform angle from the device math library.
// ==========================================================================================
struct BrickBreakTask {
    static constexpr int ID = TMA_TASK_BRICKBREAK, OBS = 45, NACT = 3, ADIM = 1, MAXSTEPS = 2000, SW = 13, RW = 4, SDIM = 46;
    static constexpr bool USES_MT = true, NATIVE_TRUNC_RULE = false, FUSED_ROLLOUT = false;
    struct S {
        double paddle, bx, by, vx, vy;
        uint32_t bricks_lo, bricks_hi;  // bit 8 r + c of the 40-bit brick mask (1 = present)
        int steps;
    };
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) {
        double *d = &s.paddle;
#pragma unroll
        for (int k = 0; k < 5; k++) d[k] = words_to_double(st[(2 * k) * N + i], st[(2 * k + 1) * N + i]);
        s.bricks_lo = st[10 * N + i];
        s.bricks_hi = st[11 * N + i];
        s.steps = (int)st[12 * N + i];
    }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) {
        const double *d = &s.paddle;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint32_t lo, hi;
            double_to_words(d[k], lo, hi);
            st[(2 * k) * N + i] = lo;
            st[(2 * k + 1) * N + i] = hi;
        }
        st[10 * N + i] = s.bricks_lo;
        st[11 * N + i] = s.bricks_hi;
        st[12 * N + i] = (uint32_t)s.steps;
    }
    __device__ static void make_rec(double angle, uint32_t *rec) {  // record = ball velocity
        double_to_words(cos(angle) * 1.5, rec[0], rec[1]);
        double_to_words(sin(angle) * 1.5, rec[2], rec[3]);
    }
    __device__ static void from_rec(const uint32_t *rec, S &s) {
        s.paddle = 20.0;
        s.bx = 20.0;
        s.by = 10.0;
        s.vx = words_to_double(rec[0], rec[1]);
        s.vy = words_to_double(rec[2], rec[3]);
        s.bricks_lo = 0xFFFFFFFFu;
        s.bricks_hi = 0xFFu;
        s.steps = 0;
    }
    __device__ static void draw(MT &mt, uint8_t *, int, uint32_t *rec) { make_rec(mt.uniform(0.78539816339744828, 2.3561944901923448), rec); }
    struct Fast {  // two resets of one uniform double; the second one counts
        static constexpr int W = 4, W_SMALL = 4;
        int k = 0;
        uint32_t ah = 0;
        double u = 0.0;
        __device__ __forceinline__ bool done() const { return k >= 4; }
        __device__ __forceinline__ void feed(uint32_t v) {
            if (k == 2) ah = v >> 5;
            if (k == 3) u = mt_double_from(ah, v);
            k++;
        }
        __device__ __forceinline__ void finish(uint32_t *rec) const { make_rec(uniform_from(0.78539816339744828, 2.3561944901923448, u), rec); }
    };
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static bool brick(const S &s, int k) { return ((k < 32 ? s.bricks_lo >> k : s.bricks_hi >> (k - 32)) & 1u) != 0; }
    __device__ static void step(S &s, int a, const float *, double &r, bool &done) {
        s.steps += 1;
        if (a == 0) s.paddle -= 3;
        else if (a == 2) s.paddle += 3;
        s.paddle = clampd(s.paddle, 4.0, 36.0);
        s.bx += s.vx;
        s.by += s.vy;
        double rew = 0.0;
        if (s.bx <= 1 || s.bx >= 39) s.vx *= -1;
        if (s.by >= 39) s.vy *= -1;
        if (s.vy < 0 && s.by - 1 <= 2 && s.bx >= s.paddle - 4.0 && s.bx <= s.paddle + 4.0) {
            s.vy *= -1;
            const double offset = (s.bx - s.paddle) / 4.0;
            s.vx += offset * 0.5;
            rew = 0.1;
        }
        // first present brick in row-major order that contains the ball (the reference's nested loop with its two breaks)
        bool hit = false;
        for (int row = 0; row < 5 && !hit; row++) {
            const double by0 = 20.0 + row * 2;
            if (!(s.by >= by0 && s.by <= by0 + 2)) continue;
            for (int c = 0; c < 8; c++) {
                const int k = row * 8 + c;
                const double bx0 = c * 5.0;
                if (brick(s, k) && s.bx >= bx0 && s.bx <= bx0 + 5.0) {
                    if (k < 32) s.bricks_lo &= ~(1u << k);
                    else s.bricks_hi &= ~(1u << (k - 32));
                    s.vy *= -1;
                    rew = 1.0;
                    hit = true;
                    break;
                }
            }
        }
        bool d = false;
        if (s.by < 1) rew = -1.0, d = true;
        if (s.bricks_lo == 0u && s.bricks_hi == 0u) rew = 10.0, d = true;
        if (s.steps > 2000) d = true;
        r = rew;
        done = d;
    }
    template <class O>
    __device__ static void obs(const S &s, O o) {
        o[0] = (float)(s.bx / 40.0);
        o[1] = (float)(s.by / 40.0);
        o[2] = (float)s.vx;
        o[3] = (float)s.vy;
        o[4] = (float)(s.paddle / 40.0);
        for (int k = 0; k < 40; k++) o[5 + k] = brick(s, k) ? 1.0f : 0.0f;
    }
    __device__ static void to_flat(const S &s, double *f) {
        f[0] = s.paddle, f[1] = s.bx, f[2] = s.by, f[3] = s.vx, f[4] = s.vy, f[5] = s.steps;
        for (int k = 0; k < 40; k++) f[6 + k] = brick(s, k) ? 1.0 : 0.0;
    }
    __device__ static void from_flat(const double *f, S &s) {
        s.paddle = f[0], s.bx = f[1], s.by = f[2], s.vx = f[3], s.vy = f[4], s.steps = (int)f[5];
        s.bricks_lo = s.bricks_hi = 0u;
        for (int k = 0; k < 40; k++)
            if (f[6 + k] != 0.0) {
                if (k < 32) s.bricks_lo |= 1u << k;
                else s.bricks_hi |= 1u << (k - 32);
            }
    }
};

// ==========================================================================================
// Glider -- backend/examples/glider.py:14-53 (constants), :55-79 (wind), :81-88 (reset), :90-237 (step), :239-265 (obs); adapter
// backend/mlagents/envs.py:242-253 (Discrete(5), unbounded Box(16), 4000-step limit).  float64 rigid-body step with sin / cos / atan2;
// the reference's `x ** 2` (libm pow, within 1 ulp of x * x) is x * x here: parity <= 1e-5.
// ==========================================================================================
struct GliderTask {
    static constexpr int ID = TMA_TASK_GLIDER, OBS = 16, NACT = 5, ADIM = 1, MAXSTEPS = 4000, SW = 26, RW = 7, SDIM = 14;
    static constexpr bool USES_MT = true, NATIVE_TRUNC_RULE = false, FUSED_ROLLOUT = true;
    struct S {
        double pos[3], vel[3], rot[3], av[3];
        int wp, steps;
    };
    __device__ static double waypoint(int w, int k) { return k == 0 ? (w == 0 ? -160.0 : 160.0) : (k == 1 ? 0.0 : 70.0); }
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) {
        double *d = s.pos;
#pragma unroll
        for (int k = 0; k < 12; k++) d[k] = words_to_double(st[(2 * k) * N + i], st[(2 * k + 1) * N + i]);
        s.wp = (int)st[24 * N + i];
        s.steps = (int)st[25 * N + i];
    }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) {
        const double *d = s.pos;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            uint32_t lo, hi;
            double_to_words(d[k], lo, hi);
            st[(2 * k) * N + i] = lo;
            st[(2 * k + 1) * N + i] = hi;
        }
        st[24 * N + i] = (uint32_t)s.wp;
        st[25 * N + i] = (uint32_t)s.steps;
    }
    // record = {ang_vel(3) as doubles, waypoint index}
    __device__ static void from_rec(const uint32_t *rec, S &s) {
        s.pos[0] = 0.0, s.pos[1] = 0.0, s.pos[2] = 60.0;
        s.vel[0] = 15.0, s.vel[1] = 0.0, s.vel[2] = -1.0;
        s.rot[0] = s.rot[1] = s.rot[2] = 0.0;
        for (int k = 0; k < 3; k++) s.av[k] = words_to_double(rec[2 * k], rec[2 * k + 1]);
        s.wp = (int)rec[6];
        s.steps = 0;
    }
    __device__ static void draw(MT &mt, uint8_t *, int, uint32_t *rec) {
        for (int k = 0; k < 3; k++) double_to_words(mt.uniform(-0.1, 0.1), rec[2 * k], rec[2 * k + 1]);
        rec[6] = mt.interval(1u);  // np.random.randint(0, 2): one masked 32-bit draw
    }
    struct Fast {  // two resets of three uniform doubles + one masked draw (7 outputs each); the second one counts
        static constexpr int W = 14, W_SMALL = 14;
        int k = 0;
        uint32_t ah = 0, wp = 0;
        double u[3] = {0.0, 0.0, 0.0};
        __device__ __forceinline__ bool done() const { return k >= 14; }
        __device__ __forceinline__ void feed(uint32_t v) {
            if (k >= 7) {
                const int j = k - 7;
                if (j == 6) {
                    wp = v & 1u;
                } else if ((j & 1) == 0) {
                    ah = v >> 5;
                } else {
                    const double d = mt_double_from(ah, v);
#pragma unroll
                    for (int q = 0; q < 3; q++)
                        if (q == (j >> 1)) u[q] = d;
                }
            }
            k++;
        }
        __device__ __forceinline__ void finish(uint32_t *rec) const {
            for (int q = 0; q < 3; q++) double_to_words(uniform_from(-0.1, 0.1, u[q]), rec[2 * q], rec[2 * q + 1]);
            rec[6] = wp;
        }
    };
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static void step(S &s, int a, const float *, double &r, bool &done) {
        const double dt = 0.02, two_pi = 6.283185307179586, pi = 3.141592653589793;
        s.steps += 1;
        double tq[3] = {0.0, 0.0, 0.0};  // roll, pitch, yaw torque
        if (a == 1) tq[0] = -15.0, tq[2] = 4.0;
        else if (a == 2) tq[0] = 15.0, tq[2] = -4.0;
        else if (a == 3) tq[1] = 10.0;
        else if (a == 4) tq[1] = -10.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double w = s.av[k] + tq[k] * dt;
            w = w * 0.95;
            s.av[k] = w;
            s.rot[k] = s.rot[k] + w * dt;
        }
        s.rot[0] = clampd(s.rot[0], -pi / 2, pi / 2);
        s.rot[1] = clampd(s.rot[1], -pi / 4, pi / 4);
        const double f1 = 1.0 / 250.0, f2 = 1.0 / 400.0;
        const double u1 = sin(s.pos[0] * f1 * 2 * pi) * cos(s.pos[1] * f1 * 2 * pi) * 8.0 * 1.0;
        const double u2 = sin(s.pos[0] * f2 * 2 * pi / 1.5) * cos(s.pos[1] * f1 * 2 * pi / 1.5) * 8.0 * 0.7;
        const double va[3] = {s.vel[0] - 1.0, s.vel[1] - 0.5, s.vel[2] - (u1 + u2)};
        const double vam = sqrt(dot3_np(va, va));
        double aoa = va[0] != 0 ? atan2(-va[2], va[0]) : 0.0;
        double aero[3] = {0.0, 0.0, 0.0};
        if (vam > 0.1) {
            const double CL = two_pi * aoa;
            const double CD = 0.02 + 0.05 * (CL * CL);
            const double q = 0.5 * 1.225 * (vam * vam) * 0.5;
            const double F[3] = {0 + -(q * CD), 0.0, q * CL + 0};
            const double cr = cos(s.rot[0]), sr = sin(s.rot[0]), cp = cos(s.rot[1]), sp = sin(s.rot[1]), cy = cos(s.rot[2]), sy = sin(s.rot[2]);
            const double Rr[3][3] = {{1, 0, 0}, {0, cr, -sr}, {0, sr, cr}};
            const double Rp[3][3] = {{cp, 0, sp}, {0, 1, 0}, {-sp, 0, cp}};
            const double Ry[3][3] = {{cy, -sy, 0}, {sy, cy, 0}, {0, 0, 1}};
            double T[3][3], R[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) T[i][j] = fma(Ry[i][2], Rp[2][j], fma(Ry[i][1], Rp[1][j], Ry[i][0] * Rp[0][j]));
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) R[i][j] = fma(T[i][2], Rr[2][j], fma(T[i][1], Rr[1][j], T[i][0] * Rr[0][j]));
#pragma unroll
            for (int i = 0; i < 3; i++) aero[i] = fma(R[i][2], F[2], fma(R[i][0], F[0], R[i][1] * F[1]));
        } else {
            aoa = 0;
        }
        const double total[3] = {aero[0] + 0, aero[1] + 0, aero[2] + -(1.5 * 9.81)};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s.vel[k] = s.vel[k] + (total[k] / 1.5) * dt;
            s.pos[k] = s.pos[k] + s.vel[k] * dt;
        }
        bool d = false;
        const double vec[3] = {waypoint(s.wp, 0) - s.pos[0], waypoint(s.wp, 1) - s.pos[1], waypoint(s.wp, 2) - s.pos[2]};
        const double dist = sqrt(dot3_np(vec, vec));
        if (dist < 15.0) s.wp = (s.wp + 1) % 2;
        const double vn = sqrt(dot3_np(s.vel, s.vel));
        double vd[3], td[3];
#pragma unroll
        for (int k = 0; k < 3; k++) vd[k] = s.vel[k] / (vn + 1e-8), td[k] = vec[k] / (dist + 1e-8);
        const double H = (dot3_np(vd, td) + 1) / 2;
        const double E = clampd(vn / 30.0, 0, 2.0);
        double rew = E * (H - E + 1);
        const double lateral = fabs(s.pos[1]);
        if (lateral > 250.0) {
            const double pr = (lateral - 250.0) / 100.0;
            rew -= 2.0 * (pr * pr);
        }
        if (s.pos[2] > 250.0) {
            const double pr = (s.pos[2] - 250.0) / 50.0;
            rew -= 2.0 * (pr * pr);
        } else if (s.pos[2] < 25.0) {
            rew -= 0.5;
        }
        if (s.pos[2] < 5.0) rew = -50.0, d = true;
        if (fabs(aoa) > 0.26179938779914941) rew = -50.0, d = true;  // np.deg2rad(15)
        if (dist > 500) rew = -50.0, d = true;
        if (s.steps > 4000) d = true;
        r = rew;
        done = d;
    }
    __device__ static void obs(const S &s, float *o) {
        const double vec[3] = {waypoint(s.wp, 0) - s.pos[0], waypoint(s.wp, 1) - s.pos[1], waypoint(s.wp, 2) - s.pos[2]};
        const double dist = sqrt(dot3_np(vec, vec));
        o[0] = (float)(s.vel[2] / 10.0);
        o[1] = (float)((s.pos[2] - 50.0) / 50.0);
        o[2] = (float)s.rot[0];
        o[3] = (float)s.rot[1];
        o[4] = (float)sin(s.rot[2]);
        o[5] = (float)cos(s.rot[2]);
        for (int k = 0; k < 3; k++) o[6 + k] = (float)s.av[k];
        for (int k = 0; k < 3; k++) o[9 + k] = (float)(s.vel[k] / 20.0);
        for (int k = 0; k < 3; k++) o[12 + k] = (float)(vec[k] / (dist + 1e-8));
        o[15] = (float)(dist / 100.0);
    }
    __device__ static void to_flat(const S &s, double *f) {
        const double *d = s.pos;
        for (int k = 0; k < 12; k++) f[k] = d[k];
        f[12] = s.wp, f[13] = s.steps;
    }
    __device__ static void from_flat(const double *f, S &s) {
        double *d = s.pos;
        for (int k = 0; k < 12; k++) d[k] = f[k];
        s.wp = (int)f[12], s.steps = (int)f[13];
    }
};

// ==========================================================================================
// Crawler-shape (BUILD-DEFINED, parity unpinned against the reference: the reference's "ant" task is
// gym.make("Ant-v5") over MuJoCo, backend/mlagents/envs.py:274-277, backend/examples/crawler.py:31-85).
// 172-dim obs, Box(-1,1,(20,)) actions, 1000-step limit.  Restated 1:1 from oracle/tma_oracle.c.
//
// Two instantiations of one articulated chain of NJ torque-driven joints on a root body:
//   CrawlerTask = ChainTask<20, 0>: BASELINE.json configs[4] "Crawler (Ant) 172-dim obs / 20-dim action" (12 root + 8 per joint features);
//   AntTask     = ChainTask<8, 1>:  the SHAPES of what the reference's `ant` task actually builds, gymnasium Ant-v5 with
//                 exclude_current_positions_from_observation (envs.py:274-277): Box(105,) observations in Ant-v5's order -- 13 qpos
//                 (z, orientation quaternion, 8 joint angles), 14 qvel (3 linear, 3 angular, 8 joint), 78 contact-force entries
//                 (13 bodies x 6, clipped to [-1, 1]) -- and Box(-1, 1, (8,)) torques.  A policy zip the reference trained on Ant-v5
//                 loads and runs against it; the DYNAMICS are the build's chain, not MuJoCo's (parity unpinned either way).
// ==========================================================================================
template <int NJ_, int LAYOUT>
struct ChainTask {
    static constexpr int NJ = NJ_;
    static constexpr int ID = LAYOUT == 0 ? TMA_TASK_CRAWLER : TMA_TASK_ANT, OBS = LAYOUT == 0 ? 12 + 8 * NJ : 105;
    static constexpr int NACT = 0, ADIM = NJ, MAXSTEPS = 1000, SW = 3 * NJ + 9, RW = 0, SDIM = 3 * NJ + 9;
    static_assert(LAYOUT == 0 || NJ == 8, "the Ant-v5 observation layout has 8 joints");
    static constexpr float ZL = NJ == 20 ? 0.015f : 0.3f / NJ;  // root height = 0.25 + ZL * sum cos(q): 0.55 with every joint at rest
    static constexpr bool USES_MT = false, NATIVE_TRUNC_RULE = false;
    static constexpr bool FUSED_ROLLOUT = true;  // the fused rollout-chunk kernels are instantiated for this task (tma_rollout.hip)
    struct S {
        float q[NJ], qd[NJ], pa[NJ], root[8];
        int steps;
    };
    __device__ static float csin(float x) {
        float x2 = x * x;
        float p = 2.7557319e-06f;
        p = p * x2 + -1.9841270e-04f;
        p = p * x2 + 8.3333333e-03f;
        p = p * x2 + -1.6666667e-01f;
        p = p * x2;
        p = p * x;
        return x + p;
    }
    __device__ static float ccos(float x) {
        float x2 = x * x;
        float p = -2.7557319e-07f;
        p = p * x2 + 2.4801587e-05f;
        p = p * x2 + -1.3888889e-03f;
        p = p * x2 + 4.1666668e-02f;
        p = p * x2 + -0.5f;
        p = p * x2;
        return 1.0f + p;
    }
    __device__ static void unpack(const uint32_t *st, int64_t N, int64_t i, S &s) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            s.q[j] = __uint_as_float(st[(int64_t)j * N + i]);
            s.qd[j] = __uint_as_float(st[(int64_t)(NJ + j) * N + i]);
            s.pa[j] = __uint_as_float(st[(int64_t)(2 * NJ + j) * N + i]);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) s.root[k] = __uint_as_float(st[(int64_t)(3 * NJ + k) * N + i]);
        s.steps = (int)st[(int64_t)(3 * NJ + 8) * N + i];
    }
    __device__ static void pack(uint32_t *st, int64_t N, int64_t i, const S &s) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            st[(int64_t)j * N + i] = __float_as_uint(s.q[j]);
            st[(int64_t)(NJ + j) * N + i] = __float_as_uint(s.qd[j]);
            st[(int64_t)(2 * NJ + j) * N + i] = __float_as_uint(s.pa[j]);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) st[(int64_t)(3 * NJ + k) * N + i] = __float_as_uint(s.root[k]);
        st[(int64_t)(3 * NJ + 8) * N + i] = (uint32_t)s.steps;
    }
    __device__ static void reset_inline(uint32_t seed, S &s) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            uint32_t h0 = mix32(seed, (uint32_t)j, 0x51u), h1 = mix32(seed, (uint32_t)j, 0x52u);
            float u0 = (float)(h0 >> 8) * (1.0f / 16777216.0f), u1 = (float)(h1 >> 8) * (1.0f / 16777216.0f);
            s.q[j] = (u0 - 0.5f) * 0.2f;
            s.qd[j] = (u1 - 0.5f) * 0.2f;
            s.pa[j] = 0.0f;
        }
        s.root[0] = 0.55f;
#pragma unroll
        for (int k = 1; k < 8; k++) s.root[k] = 0.0f;
        s.steps = 0;
    }
    __device__ static int steps(const S &s) { return s.steps; }
    __device__ static float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
    __device__ static void step(S &s, int, const float *act, double &r, bool &done) {
        const float dt = 0.05f, gear = 8.0f, kq = 4.0f, cq = 1.5f, kc = 1.0f;
        // joints are updated in place: every joint reads its neighbours' OLD angles (the previous joint's is carried in q_prev, joint 0's
        // in q_first), so no second copy of the 60 joint words is held -- the fused rollout runs this next to 128 weight registers
        float thrust_x = 0.0f, thrust_y = 0.0f, asym = 0.0f, ctrl = 0.0f, lift = 0.0f;
        const float q_first = s.q[0];
        float q_prev = s.q[NJ - 1];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float aj = clipf(act[j], -1.0f, 1.0f);
            const float qj = s.q[j];
            const float ql = q_prev, qr = (j + 1 < NJ) ? s.q[j + 1] : q_first;
            float lap = (ql + qr) - 2.0f * qj;
            float acc = gear * aj;
            acc = acc - kq * qj;
            acc = acc - cq * s.qd[j];
            acc = acc + kc * lap;
            float v = s.qd[j] + dt * acc;
            float p = qj + dt * v;
            if (p > 1.2f) {
                p = 1.2f;
                v = 0.0f;
            }
            if (p < -1.2f) {
                p = -1.2f;
                v = 0.0f;
            }
            q_prev = qj;
            s.q[j] = p;
            s.qd[j] = v;
            s.pa[j] = aj;
            float c = ccos(p), sn = csin(p);
            float side = (j & 1) ? -1.0f : 1.0f;
            float w = (j < NJ / 2) ? 1.0f : -1.0f;
            thrust_x = thrust_x + (side * v) * c;
            thrust_y = thrust_y + (w * v) * c;
            asym = asym + side * sn;
            lift = lift + c;
            ctrl = ctrl + aj * aj;
        }
        float z = s.root[0], vx = s.root[1], vy = s.root[2], pitch = s.root[3], roll = s.root[4], pr = s.root[5], rr = s.root[6], x = s.root[7];
        vx = vx + dt * (0.15f * thrust_x - 0.8f * vx);
        vy = vy + dt * (0.15f * thrust_y - 0.8f * vy);
        pr = pr + dt * (0.3f * asym - 6.0f * pitch - 1.2f * pr);
        rr = rr + dt * (0.05f * thrust_y - 6.0f * roll - 1.2f * rr);
        pitch = pitch + dt * pr;
        roll = roll + dt * rr;
        z = 0.25f + ZL * lift;
        x = x + dt * vx;
        s.root[0] = z, s.root[1] = vx, s.root[2] = vy, s.root[3] = pitch, s.root[4] = roll, s.root[5] = pr, s.root[6] = rr, s.root[7] = x;
        s.steps += 1;
        bool unhealthy = (z < 0.38f) || (fabsf(pitch) > 1.0f) || (fabsf(roll) > 1.0f);
        float rew = 1.0f + vx;
        rew = rew - 0.5f * ctrl * 0.05f;
        r = (double)rew;
        done = unhealthy || s.steps >= 1000;
    }
    // step() spread over LANES lanes of ONE wave that share an env (the fused rollout chunk: 8 lanes per env instead of one owner lane running
    // twenty joints in sequence next to idle neighbours).  Lane `sub` takes joints sub, sub + LANES, ...; the state struct `s` and the
    // scratch `terms` ([NJ][5] floats) live in LDS.  Per joint the operations are step()'s; the five sums over the joints are taken by lane 0
    // of the group in joint order from `terms` -- every stored value, the reward and the done flag equal step()'s bit for bit.
    // All LANES lanes of the group must call it together; r / done are defined in lane sub == 0.
    // Ordering: every lane reads the OLD neighbour angles before any lane stores a new one.  The lanes of a group execute the same
    // instruction stream, and LDS operations of a wave complete in issue order, so what is needed is that the COMPILER keeps all loads of
    // phase A ahead of the stores of phase B (wave_barrier + wavefront fence: no code motion across, no instruction emitted).
    template <int LANES>
    __device__ static void step_lanes(S &s, const float *act, float *terms, int sub, double &r, bool &done) {
        const float dt = 0.05f, gear = 8.0f, kq = 4.0f, cq = 1.5f, kc = 1.0f;
        constexpr int PER = (NJ + LANES - 1) / LANES;
        float ql[PER], qj[PER], qr[PER], qdj[PER], aj[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int j = sub + LANES * u;
            if (j < NJ) {
                ql[u] = s.q[j == 0 ? NJ - 1 : j - 1];
                qj[u] = s.q[j];
                qr[u] = s.q[j + 1 < NJ ? j + 1 : 0];
                qdj[u] = s.qd[j];
                aj[u] = clipf(act[j], -1.0f, 1.0f);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int j = sub + LANES * u;
            if (j < NJ) {
                float lap = (ql[u] + qr[u]) - 2.0f * qj[u];
                float acc = gear * aj[u];
                acc = acc - kq * qj[u];
                acc = acc - cq * qdj[u];
                acc = acc + kc * lap;
                float v = qdj[u] + dt * acc;
                float p = qj[u] + dt * v;
                if (p > 1.2f) {
                    p = 1.2f;
                    v = 0.0f;
                }
                if (p < -1.2f) {
                    p = -1.2f;
                    v = 0.0f;
                }
                s.q[j] = p;
                s.qd[j] = v;
                s.pa[j] = aj[u];
                const float c = ccos(p), sn = csin(p);
                const float side = (j & 1) ? -1.0f : 1.0f;
                const float w = (j < NJ / 2) ? 1.0f : -1.0f;
                float *t = terms + 5 * j;
                t[0] = (side * v) * c, t[1] = (w * v) * c, t[2] = side * sn, t[3] = c, t[4] = aj[u] * aj[u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (sub == 0) {
            float thrust_x = 0.0f, thrust_y = 0.0f, asym = 0.0f, ctrl = 0.0f, lift = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const float *t = terms + 5 * j;
                thrust_x = thrust_x + t[0];
                thrust_y = thrust_y + t[1];
                asym = asym + t[2];
                lift = lift + t[3];
                ctrl = ctrl + t[4];
            }
            float z = s.root[0], vx = s.root[1], vy = s.root[2], pitch = s.root[3], roll = s.root[4], pr = s.root[5], rr = s.root[6], x = s.root[7];
            vx = vx + dt * (0.15f * thrust_x - 0.8f * vx);
            vy = vy + dt * (0.15f * thrust_y - 0.8f * vy);
            pr = pr + dt * (0.3f * asym - 6.0f * pitch - 1.2f * pr);
            rr = rr + dt * (0.05f * thrust_y - 6.0f * roll - 1.2f * rr);
            pitch = pitch + dt * pr;
            roll = roll + dt * rr;
            z = 0.25f + ZL * lift;
            x = x + dt * vx;
            s.root[0] = z, s.root[1] = vx, s.root[2] = vy, s.root[3] = pitch, s.root[4] = roll, s.root[5] = pr, s.root[6] = rr, s.root[7] = x;
            s.steps += 1;
            const bool unhealthy = (z < 0.38f) || (fabsf(pitch) > 1.0f) || (fabsf(roll) > 1.0f);
            float rew = 1.0f + vx;
            rew = rew - 0.5f * ctrl * 0.05f;
            r = (double)rew;
            done = unhealthy || s.steps >= 1000;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // the observation row by the LANES lanes of the group: lane `sub` writes items sub, sub + LANES, ... (one source: obs_item)
    template <int LANES, class O>
    __device__ static void obs_lanes(const S &s, int sub, O o) {
#pragma unroll
        for (int u = 0; u < (OBS_ITEMS + LANES - 1) / LANES; u++) {
            const int item = sub + LANES * u;
            if (item < OBS_ITEMS) obs_item(s, item, o);
        }
    }
    // The observation row in ITEMS independent pieces: item 0 = the root block, item 1 + j = the block of joint j (LAYOUT 1: of body j).
    // obs() is the loop over the items; the fused rollout chunk gives every (row, item) pair to a thread of its own, so the 32 rows of a
    // block go out as coalesced stores by all 512 threads instead of 172 scattered stores per row by one owner lane -- one source, same bits.
    // `o` is a float pointer or anything with operator[] / operator+ (the fused rollout writes the global row and the bf16 LDS image in one pass).
    static constexpr int OBS_ITEMS = LAYOUT == 0 ? NJ + 1 : 14;
    template <class O>
    __device__ static void obs_item(const S &s, int item, O o) {
        const float pitch = s.root[3], roll = s.root[4];
        if constexpr (LAYOUT == 1) {
            if (item == 0) {
                // Ant-v5 order.  qpos[2:]: z, quaternion (w, x, y, z) of the (pitch, roll) attitude, joint angles
                const float hp = 0.5f * pitch, hr = 0.5f * roll;
                const float sp = csin(hp), cp = ccos(hp), sr = csin(hr), cr = ccos(hr);
                o[0] = s.root[0], o[1] = cp * cr, o[2] = sr * cp, o[3] = sp * cr, o[4] = -(sp * sr);
#pragma unroll
                for (int j = 0; j < NJ; j++) o[5 + j] = s.q[j];
                // qvel: root linear (x, y, z) and angular (roll, pitch, yaw) velocity, joint velocities
                o[13] = s.root[1], o[14] = s.root[2], o[15] = 0.0f, o[16] = s.root[6], o[17] = s.root[5], o[18] = 0.0f;
#pragma unroll
                for (int j = 0; j < NJ; j++) o[19 + j] = s.qd[j];
                return;
            }
            // contact forces, 13 bodies x 6: the z-force slot of bodies 1 .. 8 carries joint j's ground contact, everything else is zero
            const int bdy = item - 1;
            float cz = 0.0f;
            if (bdy >= 1 && bdy <= NJ) {
                const int j = bdy - 1;
                const float side = (j & 1) ? -1.0f : 1.0f;
                const float contact = -(csin(s.q[j]) + side * pitch * 0.5f);
                cz = contact > 0.0f ? (contact < 1.0f ? contact : 1.0f) : 0.0f;
            }
            auto p = o + (27 + 6 * bdy);
            p[0] = 0.0f, p[1] = 0.0f, p[2] = cz, p[3] = 0.0f, p[4] = 0.0f, p[5] = 0.0f;
        } else {
            if (item == 0) {
                o[0] = s.root[0], o[1] = s.root[1], o[2] = s.root[2], o[3] = pitch, o[4] = roll, o[5] = s.root[5], o[6] = s.root[6];
                o[7] = csin(pitch), o[8] = ccos(pitch), o[9] = csin(roll), o[10] = ccos(roll), o[11] = s.root[0] - 0.55f;
                return;
            }
            const int j = item - 1;
            float qj = s.q[j], qn = s.q[(j + 1) % NJ];
            float sn = csin(qj), c = ccos(qj);
            float side = (j & 1) ? -1.0f : 1.0f;
            float contact = -(sn + side * pitch * 0.5f);
            auto p = o + (12 + 8 * j);
            p[0] = qj, p[1] = s.qd[j] * 0.1f, p[2] = sn, p[3] = c, p[4] = s.pa[j], p[5] = qn - qj, p[6] = contact > 0.0f ? contact : 0.0f, p[7] = qj * qj;
        }
    }
    template <class O>
    __device__ static void obs(const S &s, O o) {
#pragma unroll
        for (int item = 0; item < OBS_ITEMS; item++) obs_item(s, item, o);
    }
    __device__ static void to_flat(const S &s, double *f) {
        for (int j = 0; j < NJ; j++) f[j] = s.q[j], f[NJ + j] = s.qd[j], f[2 * NJ + j] = s.pa[j];
        for (int k = 0; k < 8; k++) f[3 * NJ + k] = s.root[k];
        f[3 * NJ + 8] = s.steps;
    }
    __device__ static void from_flat(const double *f, S &s) {
        for (int j = 0; j < NJ; j++) s.q[j] = (float)f[j], s.qd[j] = (float)f[NJ + j], s.pa[j] = (float)f[2 * NJ + j];
        for (int k = 0; k < 8; k++) s.root[k] = (float)f[3 * NJ + k];
        s.steps = (int)f[3 * NJ + 8];
    }
};
using CrawlerTask = ChainTask<20, 0>;
using AntTask = ChainTask<8, 1>;

}  // namespace tma

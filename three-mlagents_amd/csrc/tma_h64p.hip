// tma_h64p.hip -- one PPO epoch at a SMALL minibatch (the reference's literal batch_size = 256) as ONE persistent launch, H = 64 policies.
//
// Replaces, for one pass of `for rollout_data in self.rollout_buffer.get(self.batch_size)` in stable-baselines3 2.9.0's PPO.train
// (third party; the loop model.learn() drives at /root/reference/backend/mlagents/training.py:166-170 with batch_size = 256 from
// training.py:379), every minibatch's forward / loss / backward / clip_grad_norm_ / Adam step.  As launches that is three dependent
// kernels per optimizer step (gradient, slab reduction, Adam); a 256-sample minibatch is 16 tiles per net, so the launches are all
// boundary and no work.  Here eight workgroups (four per net, one 16-sample tile per wave, one wave per SIMD) stay resident for the
// whole epoch: weights live in LDS images and are updated in place, Adam moments live in registers, and per step the workgroups
// exchange only partial-gradient slabs through the L2:
//
//   tile chain (tma_h64_tile.h) -> block sum in LDS -> slab to L2 -> [sync A: the net's 4 blocks] -> block j sums quarter j of the
//   net's 4 slabs, publishes the reduced quarter + its sum of squares -> [sync B: all 8 blocks] -> every block reads its net's
//   reduced gradient and the 8 sums of squares -> clip coefficient -> Adam on ALL of its net's parameters (redundantly, bit-identical
//   in the four blocks of a net) -> LDS image slots rewritten -> next minibatch.
//
// Sums run in the order of the per-launch path (waves in order, slabs in order), so the gradient is bit-identical to
// tma_ppo_minibatch_grad's; the squared norm is summed in a different (fixed) order in double.
//
// Placement: the hand-off uses plain stores and L1-bypassing (sc1) loads with NO cache write-back / invalidate -- valid only
// between CUs that share one L2, i.e. one XCD (MI355X_MICROARCH.md, inter-workgroup visibility).  The launch therefore carries more
// workgroups than roles: each reads HW_REG_XCC_ID, the first claimer fixes the XCD and the first HP_NB workgroups ON THAT XCD take the
// roles; the others exit.  Same-XCD placement is thus a checked property of the hardware ids, not an assumption about dispatch order.
// Every spin is bounded: a role that cannot be filled (or a peer that never arrives) sets the abort word, the kernel commits nothing
// (parameters, moments and statistics are only written behind the last step, and only while the abort word is clear) and
// tma_ppo_train_epoch_local re-runs that epoch through the per-minibatch launches.
#include "tma_h64_tile.h"

#include <cstring>
#include <cstdlib>

namespace tma {

constexpr int HP_NB = 8;          // roles: 4 workgroups per net
constexpr int HP_GRID = 128;      // workgroups launched (16 per XCD under round-robin placement)
constexpr int HP_SLAB_F = 6400;   // floats per slab / reduced-gradient array: >= one net's parameters for D <= 16, A <= 16 (6288)
// byte offsets inside the persistent region (the workspace's partial-gradient slab area, which this path does not use otherwise)
constexpr int HP_SYNC = 0;        // u32 words on lines of their own: [0] arrivals A policy net, [32] A value net, [96] claimed XCD + 1, [128] roles taken, [160] abort
constexpr int HP_SQ = 1024;       // 2 x HP_NB granules of 16 bytes {step tag, -, f64 sum of squares of the block's reduced quarter}, by step parity
constexpr int HP_TICKS = 2048;    // u64[16] phase ticks of role 0 (diagnostic, args.ticks)
constexpr int HP_SLABS = 4096;
constexpr int HP_G = HP_SLABS + HP_NB * HP_SLAB_F * 4;
constexpr int HP_TABLE = HP_G + 2 * HP_SLAB_F * 4;  // float2[n_mb]: (lr / (1 - beta1^t), sqrt(1 - beta2^t)) per optimizer step

struct EpochArgs {
    float *params, *exp_avg, *exp_avg_sq;
    PLayout L;
    Rollout rb;
    HParams hp;
    const int32_t *offs;     // buffer offset of every row of the permuted epoch (tma_ppo_epoch_prepare)
    const double *adv_part;  // (sum, sum of squares) of every minibatch's advantages, adv_stride pairs per minibatch
    int adv_stride;
    int64_t total;
    int batch, n_mb;
    float beta1, beta2, eps, max_norm;
    char *region;
    double *stat_slots, *norm_out;
    int *err_out;
    int rw;     // floats of LDS per wave: tile slots, then the wave's flat copy of its accumulators
    int ticks;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int SC1 = 16;  // cache-policy bit of the raw buffer intrinsics on gfx940+: L1 bypass
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x40000000, 0x00020000);
}
constexpr int OOB = 0x7FFFFFF0;  // a byte offset beyond every buffer's range: the load returns zeros
__device__ __forceinline__ f32x4 ld_sc1_x4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, SC1));
}
__device__ __forceinline__ double ld_sc1_f64(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, SC1));
}

// Position space of one net's parameters inside this kernel (slabs, reduced gradient, flat copies): [W1t | b1 | b2 | W3t | b3 | pad to 4 |
// W2t].  The 4096 layer-2 weights sit behind the small tensors on a float4 boundary, so a thread owns whole float4s of ONE kind: four of
// W2t (image slots computed from the thread index) and NM of the rest (image slots kept as packed words).
struct NetPos {
    int oB1, oB2, oW3, oB3, nmisc, oW2, P;
};
__host__ __device__ inline NetPos net_pos(int D, int NOUT) {
    NetPos q;
    q.oB1 = D * 64, q.oB2 = q.oB1 + 64, q.oW3 = q.oB2 + 64, q.oB3 = q.oW3 + 64 * NOUT, q.nmisc = q.oB3 + NOUT;
    q.oW2 = (q.nmisc + 3) & ~3, q.P = q.oW2 + 4096;
    return q;
}
// offset, in the flat [W1t | b1 | W2t | b2 | W3t | b3] order of PLayout, of position x < nmisc
__device__ __forceinline__ int misc_natural(int x, int D) { return x < D * 64 + 64 ? x : x + 4096; }

// LDS image slots of the parameter at offset x of the flat order (not a W2t entry): primary | secondary << 16 (`none`: a scratch word)
// -- scatter_derived_h64's map
__device__ __forceinline__ uint32_t img_slots(int x, int D, int NOUT, uint32_t none) {
    constexpr int H = 64;
    uint32_t o1 = none, o2 = none;
    if (x < D * H) {
        const int k = x >> 6, n = x & 63;
        o1 = IMG_W1 + k * 64 + (n & 15) * 4 + (n >> 4);
    } else if ((x -= D * H) < H) {
        o1 = IMG_B1 + x;
    } else if ((x -= H + H * H) < H) {
        o1 = IMG_B2 + x;
    } else if ((x -= H) < H * NOUT) {
        const int k = x / NOUT, a = x - k * NOUT;
        o1 = IMG_W3F + k * 16 + a;
        o2 = IMG_W3B + a * 64 + (k & 15) * 4 + (k >> 4);
    } else if ((x -= H * NOUT) < NOUT) {
        o1 = IMG_B3 + x;
    }
    return o1 | (o2 << 16);
}

// A wave's accumulators into its flat copy (position space above).  Every store is unconditional at a lane base + compile-time offset:
// lanes that hold padding / replicas write, with the same offsets, into the dump area behind the copy (64 + max(240, 51 * NOUT) + 16
// floats at `dump`) -- a predicate per store costs an exec save / restore around each ds_write.
template <int KS1C>
__device__ __forceinline__ void scatter_acc_flat(float *region, const NetAcc &acc, int lane, int D, int NOUT, const NetPos &q) {
    const int r16 = lane & 15, g = lane >> 4;
    const int perm = (r16 >> 2) + 4 * (r16 & 3);  // head: tile column m <-> output a(m)
    const int dl = q.P + lane;
    if constexpr (w1_blocks<KS1C>()) {  // 4x4x1 form (tma_h64_tile.h): register i of observation block blk holds W1t[k = 4 blk + i][n = lane]
#pragma unroll
        for (int blk = 0; blk < KS1C; blk++)
#pragma unroll
            for (int i = 0; i < 4; i++) region[(4 * blk + i < D) ? (4 * blk + i) * 64 + lane : dl] = acc.w1[0][blk][i];
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {  // W1t[k = 4g + r][n = 16 nt + r16]
            float *p = region + ((4 * g + r < D) ? 4 * g * 64 + r16 : dl);
#pragma unroll
            for (int nt = 0; nt < 4; nt++) p[r * 64 + nt * 16] = acc.w1[0][nt][r];
        }
    }
    {  // W2t[k = 16 kt + 4g + r][n = 16 nt + r16]
        float *p = region + q.oW2 + 4 * g * 64 + r16;
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) p[(kt * 16 + r) * 64 + nt * 16] = acc.w2[kt][nt][r];
    }
    if (w3_blocks(NOUT)) {  // (uniform) 4x4x1 form: register i of output half h holds W3t[k = 4 (lane >> 2) + i][a = (lane & 3) + 4 h]
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int a = (lane & 3) + 4 * h;
            float *p = region + (a < NOUT ? q.oW3 + 4 * (lane >> 2) * NOUT + a : dl);
#pragma unroll
            for (int i = 0; i < 4; i++) p[a < NOUT ? i * NOUT : 0] = acc.w3[h][0][i];
        }
    } else {  // W3t[k = 16 kt + 4g + r][a = perm]
        float *p = region + (perm < NOUT ? q.oW3 + 4 * g * NOUT + perm : dl);
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) p[(kt * 16 + r) * NOUT] = acc.w3[kt][0][r];
    }
    float *p1 = region + (g == 0 ? q.oB1 + r16 : dl), *p2 = region + (g == 0 ? q.oB2 + r16 : dl);
#pragma unroll
    for (int nt = 0; nt < 4; nt++) p1[nt * 16] = acc.b1[nt], p2[nt * 16] = acc.b2[nt];
    region[(g == 0 && perm < NOUT) ? q.oB3 + perm : dl] = acc.b3[0];
}

// one lane waits until *ctr >= want; false on abort (set by a peer) or after ~2^22 polls
__device__ __forceinline__ bool wait_ge(unsigned *ctr, unsigned want, unsigned *abortw) {
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        spins++;
        if ((spins & 1023) == 0 && __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (spins > (1 << 22)) {
            __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// phase ticks of role 0's thread 0, accumulated in the region itself (a register array would stay live across the whole step loop)
#define HP_TICK(i)                                                                      \
    do {                                                                                \
        if (tick_on) {                                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                 \
            tick_out[i] += t_ - tick_prev;                                              \
            tick_prev = __builtin_amdgcn_s_memtime();                                   \
        }                                                                               \
    } while (0)

// NM: float4 slots per thread for the small tensors (256 NM >= ceil(nmisc / 4) of the policy net)
template <bool IS_PI, int DT, int NM>
__device__ __forceinline__ void epoch_body(const EpochArgs &a, int j, float *smem) {
    constexpr int KS1C = DT > 0 ? (DT + 3) / 4 : 4;
    constexpr int NS = NM + 4;  // float4 slots per thread over the whole position space
    constexpr int NQ = 2;       // float4 slots of a quarter per thread
    __shared__ double red_sq[4];
    __shared__ double red_st[4 * 5];
    __shared__ int ok_s;
    __shared__ double tot_s;
    const PLayout &L = a.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int D = DT > 0 ? DT : L.D, A = L.A, KS1 = (D + 3) >> 2, NOUT = IS_PI ? A : 1;
    const int base = IS_PI ? L.pW1t : L.vW1t;
    const NetPos q = net_pos(D, NOUT);
    const int PN4 = q.P >> 2, QF = (PN4 + 3) >> 2, nm4 = q.oW2 >> 2;
    const int net = IS_PI ? 0 : 1, role = net * 4 + j;
    float *wimg = smem, *region = smem + IMG_FLOATS + wave * a.rw;
    float *slotA = region, *slotB = slotA + 16 * LDT, *dz3t = slotB + 16 * LDT, *Xt = dz3t + 256;
    float *copies = smem + IMG_FLOATS;
    unsigned *sync = reinterpret_cast<unsigned *>(a.region + HP_SYNC);
    unsigned *cntA = sync + 32 * net, *abortw = sync + 160;
    float *slab_mine = reinterpret_cast<float *>(a.region + HP_SLABS) + role * HP_SLAB_F;
    float *Gnet = reinterpret_cast<float *>(a.region + HP_G) + net * HP_SLAB_F;
    double *sqp = reinterpret_cast<double *>(a.region + HP_SQ);
    const __amdgpu_buffer_rsrc_t r_slabs = rsrc_of(reinterpret_cast<float *>(a.region + HP_SLABS) + net * 4 * HP_SLAB_F);
    const __amdgpu_buffer_rsrc_t r_G = rsrc_of(Gnet), r_sq = rsrc_of(sqp);
    const bool tick_on = a.ticks != 0 && role == 0 && tid == 0;
    unsigned long long tick_prev = __builtin_amdgcn_s_memtime();
    unsigned long long *tick_out = reinterpret_cast<unsigned long long *>(a.region + HP_TICKS);

    // ---- resident state: the net's LDS image; per thread, in registers for the whole epoch (every trip to memory -- the L2 included, and
    // scratch above all -- costs about a microsecond here), the Adam moments of the float4s it owns and the image-slot words of the small ones
    stage_copy(a.params + (IS_PI ? L.img_pi : L.img_vf), wimg, IMG_FLOATS);
    f32x4 m_w2[4], v_w2[4], m_ms[NM], v_ms[NM];
    u32x4 io_ms[NM];
    // scratch word of this lane, as an offset from the image base: in the dump area behind the wave's flat copy (dead outside the block sum)
    const uint32_t dumpw = (uint32_t)(IMG_FLOATS + wave * a.rw + q.P + lane);
    const int natW2 = D * 64 + 64;  // W2t in the flat order
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int e = base + natW2 + 4 * (tid + 256 * s4) + c;
            m_w2[s4][c] = a.exp_avg[e], v_w2[s4][c] = a.exp_avg_sq[e];
        }
#pragma unroll
    for (int u = 0; u < NM; u++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int pos = 4 * (tid + 256 * u) + c;
            const bool ok = pos < q.nmisc;
            const int nat = misc_natural(ok ? pos : 0, D);
            const float m0_ = a.exp_avg[base + nat], v0_ = a.exp_avg_sq[base + nat];
            m_ms[u][c] = ok ? m0_ : 0.0f, v_ms[u][c] = ok ? v0_ : 0.0f;
            io_ms[u][c] = ok ? img_slots(nat, D, NOUT, dumpw) : (dumpw | (dumpw << 16));
        }
    // layer-2 image slots of this thread's W2t float4s: element c of slot s4 is W2t[k = 16 s4 + 4 wave + (lane >> 4)][n = 4 (lane & 15) + c]
    const int w2k = 4 * wave + (lane >> 4), w2m = lane & 15;
    const int w2f0 = IMG_W2F + w2k * 64 + 16 * (w2m & 3) + (w2m >> 2);  // + 1024 s4 + 4 c
    const int stg0 = w2k * 65 + 4 * w2m;                                 // staging [k][65]: + 16 * 65 s4 + c
    const int B = a.batch, n_mb = a.n_mb;
    auto count_of = [&](int s) -> int {
        const int64_t left = a.total - (int64_t)s * B;
        return left < B ? (int)left : B;
    };
    const int local = (4 * j + wave) * 16 + r16;  // this lane's row of every minibatch
    auto row_of = [&](int s) -> int64_t {
        const int sc = s < n_mb ? s : n_mb - 1;
        return (int64_t)sc * B + (local < count_of(sc) ? local : 0);
    };
    int32_t nx_off = a.offs[row_of(0)];
    float pf_x[KS1C], pf_m0 = 0.0f, pf_m1 = 0.0f;
    int32_t pf_act = 0;
    double pf_adv_a = 0.0, pf_adv_b = 0.0;
    float2 pf_tb = make_float2(0.0f, 1.0f);
    auto fetch = [&](int s_next) {  // rows of the minibatch whose offset is in nx_off; then the offset one minibatch further
        const int64_t row = nx_off;
        nx_off = a.offs[row_of(s_next + 1)];
#pragma unroll
        for (int ks = 0; ks < KS1C; ks++) {
            const int c = 4 * ks + g;
            pf_x[ks] = a.rb.obs[row * D + (c < D ? c : 0)];
        }
        if constexpr (IS_PI) {
            pf_m0 = a.rb.log_probs[row];
            pf_m1 = a.rb.advantages[row];
            pf_act = static_cast<const int32_t *>(a.rb.actions)[row];
            const int sc = s_next < n_mb ? s_next : n_mb - 1;
            pf_adv_a = a.adv_part[2 * (int64_t)sc * a.adv_stride];
            pf_adv_b = a.adv_part[2 * (int64_t)sc * a.adv_stride + 1];
        } else {
            pf_m0 = a.rb.returns[row];
        }
        const int st = s_next < n_mb ? s_next : n_mb - 1;
        pf_tb = reinterpret_cast<const float2 *>(a.region + HP_TABLE)[st];
    };
    fetch(0);
    // the committed inputs of the minibatch about to run: formed one step ahead, under the wait of sync A (prepare)
    float xb[KS1C], m0 = 0.0f, m1 = 0.0f, amean = 0.0f, astd = 1.0f, invB = 1.0f;
    int act = 0;
    bool valid = false;
    float2 tb = make_float2(0.0f, 1.0f);
    auto prepare = [&](int sn) {
        const int count = count_of(sn < n_mb ? sn : n_mb - 1);
        valid = local < count;
        invB = 1.0f / (float)count;
#pragma unroll
        for (int ks = 0; ks < KS1C; ks++) xb[ks] = (valid && 4 * ks + g < D) ? pf_x[ks] : 0.0f;
        m0 = pf_m0, m1 = pf_m1, act = pf_act, tb = pf_tb;
        amean = 0.0f, astd = 1.0f;
        if (IS_PI && a.hp.normalize_advantage && count > 1) {  // (the fold of adv_final_kernel; one partial pair per minibatch at batch <= 1024)
            const double n = (double)count, mean = pf_adv_a / n;
            double var = n > 1.0 ? (pf_adv_b - n * mean * mean) / (n - 1.0) : 0.0;
            if (var < 0.0) var = 0.0;
            amean = (float)mean;
            astd = (float)sqrt(var);
        }
        fetch(sn + 1);
    };
    prepare(0);
    TileStats st;
    TileTicks tk;
#ifdef TMA_H64_TICKS
    tk.on = a.ticks != 0 && role == 0 && wave == 0, tk.prev = 0;
#endif
    float last_norm = 0.0f, last_coef = 1.0f;
    __syncthreads();
    HP_TICK(0);

    for (int s = 0; s < n_mb; s++) {
        const float2 tbs = tb;  // (this step's Adam constants: prepare() below replaces tb with the next step's)
#ifdef TMA_H64_TICKS
        tk.prev = __builtin_amdgcn_s_memtime();
#endif
        NetAcc acc;
        zero_acc(acc);
        h64t_tile<IS_PI, KS1C>(wimg, slotA, slotB, dz3t, Xt, xb, m0, m1, act, valid, KS1, A, invB, amean, astd, a.hp, acc, st, tk, lane);
        HP_TICK(1);

        // ---- block sum: every wave scatters its accumulators into its own flat copy (its tile slots are dead), then each thread sums
        // its float4s over the four copies in wave order (flush_all_t's order) and stores them to the block's slab ----
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc.b1[nt] = xg_sum(acc.b1[nt]), acc.b2[nt] = xg_sum(acc.b2[nt]);
        acc.b3[0] = xg_sum(acc.b3[0]);
        scatter_acc_flat<KS1C>(region, acc, lane, D, NOUT, q);
        __syncthreads();
#pragma unroll
        for (int i0 = 0; i0 < NS; i0 += 2) {  // two slots at a time: their eight LDS reads in flight, no branch between issue and use
            f32x4 cv[2][4];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u;
                if (i >= NS || 256 * i >= PN4) continue;  // (uniform)
                const int f = tid + 256 * i, fc = f < PN4 ? f : PN4 - 1;
#pragma unroll
                for (int w = 0; w < 4; w++) cv[u][w] = *reinterpret_cast<const f32x4 *>(copies + w * a.rw + 4 * fc);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u;
                if (i >= NS || 256 * i >= PN4) continue;
                const int f = tid + 256 * i;
                const f32x4 v = ((cv[u][0] + cv[u][1]) + cv[u][2]) + cv[u][3];
                if (f < PN4) *reinterpret_cast<f32x4 *>(slab_mine + 4 * f) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slab is in the L2 before anyone is told
        __syncthreads();
        HP_TICK(2);
        // ---- sync A: the four blocks of this net ----
        if (tid == 0) __hip_atomic_fetch_add(cntA, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prepare(s + 1);  // (the next minibatch's inputs, while the other blocks arrive)
        HP_TICK(8);
        if (tid == 0) ok_s = wait_ge(cntA, 4u * (unsigned)(s + 1), abortw) ? 1 : 0;
        __syncthreads();
        if (!ok_s) return;
        HP_TICK(3);
        // ---- quarter j of the net's gradient: sum of the four slabs in slab order (slab_reduce_kernel's order), its sum of squares ----
        {
            f32x4 qv[NQ][4];
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                const int fq = j * QF + tid + 256 * i;
                const bool on = tid + 256 * i < QF && fq < PN4;
#pragma unroll
                for (int b = 0; b < 4; b++) qv[i][b] = ld_sc1_x4(r_slabs, on ? (b * HP_SLAB_F + 4 * fq) * 4 : OOB);  // (out of range: zeros, no branch)
            }
            double sq = 0.0;
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                const int fq = j * QF + tid + 256 * i;
                const bool on = tid + 256 * i < QF && fq < PN4;
                f32x4 v = ((qv[i][0] + qv[i][1]) + qv[i][2]) + qv[i][3];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int pos = 4 * fq + c;
                    if (pos >= q.nmisc && pos < q.oW2) v[c] = 0.0f;  // (the pad words of the position space hold stale LDS bytes)
                    sq += (double)v[c] * (double)v[c];
                }
                if (on) *reinterpret_cast<f32x4 *>(Gnet + 4 * fq) = v;
            }
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
            if (lane == 0) red_sq[wave] = sq;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- sync B: all eight blocks.  Each block publishes ONE 16-byte granule {step tag, -, sum of squares}: the tag arrives with
        // the value (a single aligned store), so there is no second drain and no counter; wave 0 polls the eight granules, one per lane ----
        if (wave == 0) {
            const unsigned tag = (unsigned)(s + 1);
            if (lane == 0) {
                const double sqb = ((red_sq[0] + red_sq[1]) + red_sq[2]) + red_sq[3];
                const u32x2 h = __builtin_bit_cast(u32x2, sqb);
                // granules are double-buffered by step parity: a block of the OTHER net may run one step ahead of a slow poller (sync A only
                // couples the four blocks of one net) -- it then writes the other parity's granule and this step's stays intact until every
                // block has passed sync B of the next step
                *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(sqp) + 16 * (role + HP_NB * (s & 1))) = u32x4{tag, 0u, h[0], h[1]};
            }
            HP_TICK(4);
            u32x4 gr;
            int spins = 0;
            bool fine = true;
            for (;;) {
                gr = __builtin_amdgcn_raw_buffer_load_b128(r_sq, 16 * ((lane & (HP_NB - 1)) + HP_NB * (s & 1)), 0, SC1);
                if (__builtin_amdgcn_ballot_w64(gr[0] != tag) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                spins++;
                if ((spins & 1023) == 0 && __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) fine = false;
                if (spins > (1 << 22)) {
                    __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    fine = false;
                }
                if (!fine) break;
            }
            double tot = 0.0;  // the blocks' sums in block order
#pragma unroll
            for (int b = 0; b < HP_NB; b++) {
                const u32x2 h = {(unsigned)__builtin_amdgcn_readlane((int)gr[2], b), (unsigned)__builtin_amdgcn_readlane((int)gr[3], b)};
                const double v = __builtin_bit_cast(double, h);
                tot = b == 0 ? v : tot + v;
            }
            if (lane == 0) tot_s = tot, ok_s = fine ? 1 : 0;
        }
        __syncthreads();
        if (!ok_s) return;
        HP_TICK(5);
        // ---- global norm, clip coefficient, Adam on every parameter of this net (adam_scatter_h64_kernel's arithmetic) ----
        {
            f32x4 g_w2[4], g_ms[NM];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) g_w2[s4] = ld_sc1_x4(r_G, 16 * (nm4 + tid + 256 * s4));
#pragma unroll
            for (int u = 0; u < NM; u++) g_ms[u] = ld_sc1_x4(r_G, tid + 256 * u < nm4 ? 16 * (tid + 256 * u) : OOB);
            const float total_norm = (float)sqrt(tot_s);
            float coef = a.max_norm / (total_norm + 1e-6f);
            coef = coef > 1.0f ? 1.0f : coef;
            if (a.max_norm <= 0.0f) coef = 1.0f;
            last_norm = total_norm, last_coef = coef;
            const float lr_step = tbs.x, inv_bc2 = 1.0f / tbs.y;
            HP_TICK(6);
            // W2t: both image slots follow from the thread index.  The forward image takes the new value directly (a wave's lanes cover
            // 64 consecutive words: conflict-free); the input-gradient image is [n][k]-major, where the same lanes would hit one bank
            // sixteen at a time -- the values go through a [k][65] staging tile (the dead flat copies) and are written transposed below.
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                float pv[4];
#pragma unroll
                for (int c = 0; c < 4; c++) pv[c] = wimg[w2f0 + 1024 * s4 + 4 * c];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float gv = (g_w2[s4][c] * 1.0f) * coef;
                    float mm = m_w2[s4][c], vv = v_w2[s4][c];
                    pv[c] = adam_update_h64(pv[c], gv, mm, vv, a.beta1, a.beta2, inv_bc2, a.eps, lr_step);
                    m_w2[s4][c] = mm, v_w2[s4][c] = vv;
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    wimg[w2f0 + 1024 * s4 + 4 * c] = pv[c];
                    copies[stg0 + 16 * 65 * s4 + c] = pv[c];
                }
            }
            // the small tensors: image values first (one LDS latency), updates, then both slots (padding lanes and single-slot
            // parameters write the lane's dump word) -- no branch inside
#pragma unroll
            for (int u = 0; u < NM; u++) {
                if (256 * u >= nm4) continue;  // (uniform)
                float pv[4];
#pragma unroll
                for (int c = 0; c < 4; c++) pv[c] = wimg[io_ms[u][c] & 0xFFFFu];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float gv = (g_ms[u][c] * 1.0f) * coef;
                    float mm = m_ms[u][c], vv = v_ms[u][c];
                    pv[c] = adam_update_h64(pv[c], gv, mm, vv, a.beta1, a.beta2, inv_bc2, a.eps, lr_step);
                    m_ms[u][c] = mm, v_ms[u][c] = vv;
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    wimg[io_ms[u][c] & 0xFFFFu] = pv[c];
                    wimg[io_ms[u][c] >> 16] = pv[c];
                }
            }
            __syncthreads();
            // input-gradient image of layer 2: W2B[n][k] with n = 16 wave + t, k = lane, read from the staging tile's column n
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int n = 16 * wave + t;
                wimg[IMG_W2B + n * 64 + (lane & 15) * 4 + (lane >> 4)] = copies[lane * 65 + n];
            }
        }
        __syncthreads();  // the images are complete before the next tile reads them
        HP_TICK(7);
    }

    // ---- epilogue: statistics of this block's tiles; block 0 of each net writes the parameters, their derived copies and the moments ----
    // (nothing is committed once any block gave up on a wait: the host re-runs the epoch through the per-minibatch launches from the
    // untouched parameters -- tma_ppo_train_epoch_local)
    if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    double stv[5] = {st.a, st.ent, st.kl, (double)st.clip, (double)st.n};
#pragma unroll
    for (int qi = 0; qi < 5; qi++)
        for (int o = 32; o > 0; o >>= 1) stv[qi] += __shfl_down(stv[qi], o, 64);
    if (lane == 0)
        for (int qi = 0; qi < 5; qi++) red_st[wave * 5 + qi] = stv[qi];
    __syncthreads();
    if (tid < 5) {
        double sum = 0.0;
        for (int w = 0; w < 4; w++) sum += red_st[w * 5 + tid];
        // slot layout {policy_loss, value_sq_err, entropy, approx_kl, clipped, n}
        const int qi = IS_PI ? (tid == 0 ? 0 : tid + 1) : (tid == 0 ? 1 : -1);
        if (qi >= 0) a.stat_slots[j * 8 + qi] += sum;
    }
    if (j == 0) {
        auto put = [&](int nat, float pn, float mm, float vv) {
            const int e = base + nat;
            a.params[e] = pn;
            scatter_derived_h64(a.params, L, e, pn);
            a.exp_avg[e] = mm;
            a.exp_avg_sq[e] = vv;
        };
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
            for (int c = 0; c < 4; c++) put(natW2 + 4 * (tid + 256 * s4) + c, wimg[w2f0 + 1024 * s4 + 4 * c], m_w2[s4][c], v_w2[s4][c]);
#pragma unroll
        for (int u = 0; u < NM; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int pos = 4 * (tid + 256 * u) + c;
                if (pos < q.nmisc) put(misc_natural(pos, D), wimg[io_ms[u][c] & 0xFFFFu], m_ms[u][c], v_ms[u][c]);
            }
        if (IS_PI && tid == 0) a.norm_out[0] = (double)last_norm, a.norm_out[1] = (double)last_coef;
    }
}

template <int DT, int NM>
__global__ __launch_bounds__(256, 1) void ppo_epoch_h64p_kernel(EpochArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int role_s;
    unsigned *sync = reinterpret_cast<unsigned *>(a.region + HP_SYNC);
    if (threadIdx.x == 0) {
        // HW_REG_XCC_ID (id 20), bits [3:0]: the XCD this workgroup runs on
        const unsigned myx = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) + 1u;
        const unsigned old = atomicCAS(sync + 96, 0u, myx);
        const unsigned target = old == 0u ? myx : old;
        role_s = myx == target ? (int)atomicAdd(sync + 128, 1u) : -1;
    }
    __syncthreads();
    const int role = role_s;
    if (role < 0 || role >= HP_NB) return;
    if (__hip_atomic_load(sync + 160, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {  // abort word set before the launch (TMA_PERSIST_FORCE_FAIL)
        if (threadIdx.x == 0) *a.err_out = 1;
        return;
    }
    if (role < 4) epoch_body<true, DT, NM>(a, role, smem);
    else epoch_body<false, DT, NM>(a, role - 4, smem);
    // a failed wait anywhere: record it for tma_ppo_pop_stats (parameters were left untouched by every block that saw the abort)
    if (threadIdx.x == 0 && __hip_atomic_load(sync + 160, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) *a.err_out = 1;
}

// (lr / (1 - beta1^t), sqrt(1 - beta2^t)) for t = first_step .. first_step + n - 1: the formula tma_ppo_adam_step_local evaluates on the host, here on
// the stream.  Device pow() and host pow() are both within an ulp of the double result but need not agree in its last bit; the two constants are
// rounded to float afterwards, so they differ only where that bit falls on a float rounding boundary (the persistent / launch comparison over
// 3 x 96 and 1 024 steps in tests/test_ppo_gpu.py holds to the last bit of the parameters; the contract is "to the last bit or two")
__global__ void adam_table_kernel(float2 *table, int n, int64_t first_step, double lr, double beta1, double beta2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double t = (double)(first_step + i);
    const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
    table[i] = make_float2((float)(lr / bc1), (float)sqrt(bc2));
}

}  // namespace tma

using namespace tma;

bool tma_epoch_h64p_eligible(const PLayout &L, int64_t batch_size, int64_t total) {
    const bool off = getenv("TMA_NO_PERSIST") != nullptr;  // (read per call: tests switch paths inside one process)
    if (off || L.img_pi < 0 || batch_size != 256 || total < 2 * batch_size) return false;
    const int64_t n_mb = (total + batch_size - 1) / batch_size;
    return n_mb <= 65535 && HP_TABLE + n_mb * 8 <= (int64_t)slab_cap(L) * L.P * 4;
}

int tma_launch_epoch_h64p(float *params, const PLayout &L, const Rollout &R, const HParams &hp, const int32_t *offs, const double *adv_part,
                          int adv_stride, int64_t total, int64_t batch_size, float *exp_avg, float *exp_avg_sq, int64_t first_step, double lr,
                          double beta1, double beta2, double eps, double max_grad_norm, char *ws, hipStream_t s) {
    static const int ticks = getenv("TMA_H64P_TICKS") ? 1 : 0;
    EpochArgs a;
    a.params = params, a.exp_avg = exp_avg, a.exp_avg_sq = exp_avg_sq;
    a.L = L, a.rb = R, a.hp = hp;
    a.offs = offs, a.adv_part = adv_part, a.adv_stride = adv_stride;
    a.total = total, a.batch = (int)batch_size, a.n_mb = (int)((total + batch_size - 1) / batch_size);
    a.beta1 = (float)beta1, a.beta2 = (float)beta2, a.eps = (float)eps, a.max_norm = (float)max_grad_norm;
    a.region = ws + WS_SLABS;
    a.stat_slots = reinterpret_cast<double *>(ws + WS_STATS);
    a.norm_out = reinterpret_cast<double *>(ws + WS_NORM_OUT);
    a.err_out = reinterpret_cast<int *>(ws + WS_PERSIST_ERR);
    a.ticks = ticks;
    const int nout_max = L.A > 1 ? L.A : 1, dump_imm = 51 * nout_max > 240 ? 51 * nout_max : 240;
    const NetPos qp = net_pos(L.D, nout_max);  // the policy net's position space (the value net's is no larger)
    const int need = qp.P + ((64 + dump_imm + 16 + 3) & ~3);  // flat copy + dump area
    a.rw = need > T_PER_WAVE ? need : T_PER_WAVE;
    const int nm = ((qp.oW2 >> 2) + 255) / 256;
    if (qp.P > HP_SLAB_F || nm > 3) return TMA_ERR_INVALID;
    const int smem = (IMG_FLOATS + 4 * a.rw) * 4;
    TMA_HIP(hipMemsetAsync(a.region, 0, HP_SLABS, s));
    const char *force_fail = getenv("TMA_PERSIST_FORCE_FAIL");
    if (force_fail != nullptr && strcmp(force_fail, "late") != 0)  // test hook: the launch finds its abort word set, commits nothing and reports the failure
        TMA_HIP(hipMemsetAsync(a.region + HP_SYNC + 160 * 4, 1, 1, s));
    adam_table_kernel<<<dim3((unsigned)((a.n_mb + 255) / 256)), dim3(256), 0, s>>>(reinterpret_cast<float2 *>(a.region + HP_TABLE), a.n_mb, first_step,
                                                                                   lr, beta1, beta2);
    TMA_LAUNCH_CHECK();
    auto launch = [&](auto k) -> int {
        TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        k<<<dim3(HP_GRID), dim3(256), smem, s>>>(a);
        return TMA_OK;
    };
    int rc;
    if (L.D == 4 && nm <= 2) rc = nm == 1 ? launch(ppo_epoch_h64p_kernel<4, 1>) : launch(ppo_epoch_h64p_kernel<4, 2>);
    else if (L.D == 6 && nm <= 2) rc = nm == 1 ? launch(ppo_epoch_h64p_kernel<6, 1>) : launch(ppo_epoch_h64p_kernel<6, 2>);
    else rc = launch(ppo_epoch_h64p_kernel<0, 3>);
    if (rc) return rc;
    TMA_LAUNCH_CHECK();
    return TMA_OK;
}

// diagnostic: the ten phase-tick sums role 0 left behind (TMA_H64P_TICKS=1)
extern "C" int tma_debug_h64p_ticks(void *workspace, unsigned long long *out10) {
    if (!workspace || !out10) return TMA_ERR_INVALID;
    TMA_HIP(hipDeviceSynchronize());
    TMA_HIP(hipMemcpy(out10, static_cast<char *>(workspace) + WS_SLABS + HP_TICKS, sizeof(unsigned long long) * 10, hipMemcpyDeviceToHost));
    return TMA_OK;
}

#ifdef TMA_H64_TICKS
extern "C" int tma_debug_h64p_tile_ticks(unsigned long long *out32, int reset) {
    TMA_HIP(hipDeviceSynchronize());
    if (out32) TMA_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_h64_ticks), sizeof(unsigned long long) * 32));
    if (reset) {
        unsigned long long z[32] = {};
        TMA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_h64_ticks), z, sizeof(z)));
    }
    return TMA_OK;
}
#endif

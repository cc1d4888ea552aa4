// tma_h64_tile.h -- one 16-sample tile of the H = 64 PPO minibatch gradient (forward, loss, backward) as a transposed register chain,
// shared by the per-minibatch gradient kernels (tma_h64.hip) and the persistent small-minibatch epoch kernel (tma_h64p.hip).
// Formulas: stable-baselines3 2.9.0 PPO.train / evaluate_actions, driven by model.learn() at
// /root/reference/backend/mlagents/training.py:166-170.
#pragma once
#include "tma_ppo_types.h"

namespace tma {

// the whole parameter gradient of one net in MFMA accumulators (105 VGPRs): dW += X^T.dZ is accumulated through the MFMA C operand
struct NetAcc {
    f32x4 w1[1][4];
    f32x4 w2[4][4];
    f32x4 w3[4][1];
    float b1[4], b2[4], b3[1];
};

__device__ __forceinline__ void zero_acc(NetAcc &a) {
    const f32x4 z = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        a.w1[0][j] = z;
        a.w3[j][0] = z;
        a.b1[j] = 0.0f;
        a.b2[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) a.w2[i][j] = z;
    }
    a.b3[0] = 0.0f;
}

#ifdef TMA_H64_TICKS  // diagnostic build only (make CXXFLAGS+=-DTMA_H64_TICKS): per-phase issue-time stamps of wave 0 of blocks 0 / 1
static __device__ unsigned long long g_h64_ticks[2][16];
struct TileTicks {
    bool on;
    unsigned long long prev;
};
#define H64_TICK(i)                                                                          \
    do {                                                                                     \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();                          \
        if (tk.on && lane == 0) g_h64_ticks[IS_PI ? 0 : 1][i] += _t - tk.prev;              \
        tk.prev = __builtin_amdgcn_s_memtime();                                              \
    } while (0)
#else
struct TileTicks {};
#define H64_TICK(i) do {} while (0)
#endif
constexpr int LDT = 64;
constexpr int T_PER_WAVE = 2 * 16 * LDT + 256 + 256;  // slot A, slot B, dz3 [16][16], X [16][16]
// Column swizzle of the [16][64] tiles: XOR with 16 (row & 1) + 4 ((row >> 1) & 3) + 32 (row >> 3).  Banks by the per-instruction rules of
// MI355X_MICROARCH.md (LDS table): the chain's b128 stores (8 consecutive rows per lane group, modulo 32) land on eight distinct 16-byte
// slots, the transposed b32 reads of the weight-gradient operands (rows 4s + g of two lane groups per half wave, modulo 32) on the two
// halves of the bank row, and the b128 read-back of a tile (non-contiguous 16-lane groups, modulo 64) on 16 distinct slots: all
// conflict-free.  (Rounds 1-3: rows of 68 floats, XOR 16 (row & 1) -- the transposed reads, 72 per tile, were 2-way: 4 LDS cycles instead of 2.)
__device__ __forceinline__ int tsw(int row, int col) { return row * LDT + (col ^ (((row & 1) << 4) | (((row >> 1) & 3) << 2) | ((row >> 3) << 5))); }

// ---- The two SMALL weight gradients of a tile on v_mfma_f32_4x4x1_16b_f32 (round 3).  dW1 = X^T . dz1 has D <= 8 useful rows and dW3 = h2^T . dz3
// has n_out <= 8 useful columns; as 16x16x4 MFMAs each is 16 instructions of 32 cycles with 3/4 (GridWorld: D = 4, 5 actions) of every tile
// padding.  The 4x4x1 instruction is sixteen independent 4 x 4 outer products in 8 cycles (measured: tools/mfma4x4_probe.hip; D(lane 4b + j,
// register i) = A(lane 4b + i) * B(lane 4b + j) + C): one instruction per SAMPLE, accumulating
//   dW1[k = 4 blk + i][n = lane]            : A(lane) = X[s][4 blk + (lane & 3)],  B(lane) = dz1[s][lane]   (blk < KS1C <= 2 observation blocks)
//   dW3[k = 4 (lane >> 2) + i][a = j + 4 h] : A(lane) = h2[s][lane],               B(lane) = dz3[s][a = (lane & 3) + 4 h]   (h < 2 halves of <= 8 outputs)
// in registers i = 0..3 -- 128 / 256 cycles of matrix pipe instead of 512 each, 640 of a tile's 8 064 less.  The accumulators live in the
// first KS1C entries of NetAcc::w1[0] and the first two of NetAcc::w3 (the other entries stay zero and are never stored); the bias sums are
// formed exactly as bwd_weight_acc_t forms them.  Wider observations / heads keep the 16x16x4 form.
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
template <int KS1C>
constexpr bool w1_blocks() { return KS1C <= 2; }
__device__ __forceinline__ bool w3_blocks(int NOUT) { return NOUT <= 8; }

// all-reduce over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} (the lane groups holding one sample's 16 head outputs): VALU only
template <class F>
__device__ __forceinline__ float xg_reduce(float v, F op) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return op(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xg_sum(float v) { return xg_reduce(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float xg_max(float v) { return xg_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }

// lane (g, s) stores features 16mt + 4g .. + 3 of sample s: one b128 per output tile
__device__ __forceinline__ void store_tile_t(float *tile, const f32x4 (&v)[4], int r16, int g) {
#pragma unroll
    for (int mt = 0; mt < 4; mt++) *reinterpret_cast<f32x4 *>(tile + tsw(r16, 16 * mt + 4 * g)) = v[mt];
}

// accW[kt][nt] += x[16 samples][16kt..]^T . dz[16 samples][16nt..]; accb[nt] += column sums of dz.  Operands are read transposed
// (sample on the lane group / k index, feature on lane & 15) from [sample][feature] tiles: XS / ZS = swizzled [16][LDT] tile, else
// a plain [16][16] tile.
template <int KT, int NT, bool XS, bool ZS>
__device__ __forceinline__ void bwd_weight_acc_t(const float *xin, const float *dz, f32x4 (&accW)[KT][NT], float (&accb)[NT], int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    float bf[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
#pragma unroll
        for (int s = 0; s < 4; s++) bf[nt][s] = ZS ? dz[tsw(4 * s + g, nt * 16 + r16)] : dz[(4 * s + g) * 16 + r16];
        accb[nt] += (bf[nt][0] + bf[nt][1]) + (bf[nt][2] + bf[nt][3]);
    }
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        float a[4];
        // (plain X tile: feature columns >= D hold stale LDS bytes; they only feed accumulator rows k >= D, which flush_segment never stores)
#pragma unroll
        for (int s = 0; s < 4; s++) a[s] = XS ? xin[tsw(4 * s + g, kt * 16 + r16)] : xin[(4 * s + g) * 16 + r16];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) accW[kt][nt] = mfma16(a[s], bf[nt][s], accW[kt][nt]);
    }
}

// out[mt] += sum over the 16 k-steps (j, r) of  A = wrow[(16j + r) * 64 floats] (a lane's four output-tile operands: one b128)  x  B = in[j][r].
// The weight reads do not depend on the chain: they are issued two k-steps ahead of the MFMAs that consume them and the scheduler is
// fenced per k-step -- left alone, hipcc issues each pair of reads AFTER the previous eight MFMAs and waits out the LDS latency with
// the matrix pipe drained.
__device__ __forceinline__ void chain64(const float *wrow, const f32x4 (&in)[4], f32x4 (&out)[4]) {
    f32x4 wq[3];
    wq[0] = *reinterpret_cast<const f32x4 *>(wrow);
    wq[1] = *reinterpret_cast<const f32x4 *>(wrow + 64);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i + 2 < 16) wq[(i + 2) % 3] = *reinterpret_cast<const f32x4 *>(wrow + (16 * ((i + 2) >> 2) + ((i + 2) & 3)) * 64);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 w = wq[i % 3];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) out[mt] = mfma16(w[mt], in[i >> 2][i & 3], out[mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// chain64 with the B operands formed on the way: act(j, r) turns element r of input tile j into the operand of k-step (j, r) (tanh of
// a pre-activation, or a delta from a back-propagated gradient) and is issued under the MFMAs of k-step (j - 1, r) -- only tile 0's four
// elements are formed in front of the chain.  done(j) runs once tile j's four operands exist (the tile's LDS store).  Arithmetic and
// accumulation order are chain64's: results are bit-identical, the matrix pipe just no longer idles through 16 activations.
template <class ACT, class DONE>
__device__ __forceinline__ void chain64_act(const float *wrow, f32x4 (&in)[4], f32x4 (&out)[4], ACT act, DONE done) {
    f32x4 wq[3];
    wq[0] = *reinterpret_cast<const f32x4 *>(wrow);
    wq[1] = *reinterpret_cast<const f32x4 *>(wrow + 64);
#pragma unroll
    for (int r = 0; r < 4; r++) in[0][r] = act(0, r);
    done(0);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i + 2 < 16) wq[(i + 2) % 3] = *reinterpret_cast<const f32x4 *>(wrow + (16 * ((i + 2) >> 2) + ((i + 2) & 3)) * 64);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 w = wq[i % 3];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) out[mt] = mfma16(w[mt], in[i >> 2][i & 3], out[mt]);
        if (i + 4 < 16) {
            in[(i + 4) >> 2][(i + 4) & 3] = act((i + 4) >> 2, (i + 4) & 3);
            if (((i + 4) & 3) == 3) done((i + 4) >> 2);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifndef TMA_H64_PIPE
#define TMA_H64_PIPE 1  // 0: activations / deltas formed in whole-tile passes between the chains (the round-2 first version; A/B builds)
#endif

struct TileStats {
    double a = 0.0, ent = 0.0, kl = 0.0;
    float clip = 0.0f, n = 0.0f;  // counts: exact in float (a lane sees far fewer than 2^24 samples)
};

// Loss of one tile from the head outputs (C layout: register r of lane group g is output g + 4r of sample r16): the clipped
// surrogate + entropy bonus for the policy net, the squared error for the value net.  Returns d loss / d head output, adds the
// tile's statistics.  (SB3 PPO.train)
template <bool IS_PI>
__device__ __forceinline__ f32x4 h64t_loss(const f32x4 &o0, const f32x4 &o1, float m0, float m1, int act, bool valid, int A, float invB, float amean,
                                           float astd, const HParams &hp, TileStats &st, int lane) {
    const int g = lane >> 4;
    f32x4 dz3;
    if constexpr (IS_PI) {
        float x[4], e[4], lp[4], p[4];
        bool ok[4];
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            ok[r] = g + 4 * r < A;
            x[r] = ok[r] ? o0[r] + o1[r] : -INFINITY;
            m = fmaxf(m, x[r]);
        }
        m = xg_max(m);
        float ssum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            e[r] = ok[r] ? __expf(x[r] - m) : 0.0f;  // hardware exp2 / log2 / rcp (~1 ulp): the loss phase is the policy blocks' critical extra
            ssum += e[r];
        }
        ssum = xg_sum(ssum);
        const float lse = m + __logf(ssum), rs = __builtin_amdgcn_rcpf(ssum);
        float lpa = 0.0f, ent = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            lp[r] = ok[r] ? x[r] - lse : 0.0f;
            p[r] = e[r] * rs;
            lpa += (g + 4 * r == act) ? lp[r] : 0.0f;
            ent += p[r] * lp[r];
        }
        lpa = xg_sum(lpa);
        ent = -xg_sum(ent);
        const float old = m0;
        const float advn = (m1 - amean) / (astd + 1e-8f);
        const float ratio = __expf(lpa - old);
        const float pl1 = advn * ratio;
        const float rc = fminf(fmaxf(ratio, 1.0f - hp.clip_range), 1.0f + hp.clip_range);
        const float pl2 = advn * rc;
        const float g_lp = (valid && pl1 <= pl2) ? -(advn * ratio) * invB : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float dl = g_lp * (((g + 4 * r == act) ? 1.0f : 0.0f) - p[r]);
            dl += valid ? (hp.ent_coef * invB) * (p[r] * (lp[r] + ent)) : 0.0f;
            dz3[r] = ok[r] ? dl : 0.0f;
        }
        if (valid && g == 0) {
            st.a += (double)(-fminf(pl1, pl2));
            st.ent += (double)ent;
            st.kl += (double)((ratio - 1.0f) - (lpa - old));
            st.clip += (fabsf(ratio - 1.0f) > hp.clip_range) ? 1.0f : 0.0f;
            st.n += 1.0f;
        }
    } else {
        const float diff = (o0[0] + o1[0]) - m0;
        const bool mine = valid && g == 0;
        dz3 = f32x4{mine ? (hp.vf_coef * 2.0f * invB) * diff : 0.0f, 0.0f, 0.0f, 0.0f};
        if (mine) st.a += (double)(diff * diff);
    }
    return dz3;
}

// One tile.  Lane (g, r16) works for the tile's sample r16; xb[ks] = observation feature 4 ks + g of that sample (0 beyond D or for an
// invalid row), m0 / m1 / act = (old log-prob, advantage, action) for the policy net, (return, -, -) for the value net.
// wimg: this net's LDS weight image; slotA / slotB / dz3t / Xt: the wave's private [sample][feature] tiles (T_PER_WAVE floats in all).
template <bool IS_PI, int KS1C>
__device__ __forceinline__ void h64t_tile(const float *wimg, float *slotA, float *slotB, float *dz3t, float *Xt, const float (&xb)[KS1C], float m0,
                                          float m1, int act, bool valid, int KS1, int A, float invB, float amean, float astd, const HParams &hp,
                                          NetAcc &acc, TileStats &st, TileTicks &tk, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    const int NOUT = IS_PI ? A : 1;
    (void)tk;
#pragma unroll
    for (int ks = 0; ks < KS1C; ks++)
        if (ks < KS1) Xt[r16 * 16 + 4 * ks + g] = xb[ks];
    // ---- layer 1: h1^T = tanh(W1^T . x^T + b1) ----
    f32x4 h1[4], h2[4];
    {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) h1[mt] = *reinterpret_cast<const f32x4 *>(wimg + IMG_B1 + 16 * mt + 4 * g);
#pragma unroll
        for (int ks = 0; ks < KS1C; ks++) {
            if (ks < KS1) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wimg + IMG_W1 + (4 * ks + g) * 64 + r16 * 4);
#pragma unroll
                for (int mt = 0; mt < 4; mt++) h1[mt] = mfma16(w[mt], xb[ks], h1[mt]);
            }
        }
        if constexpr (!TMA_H64_PIPE) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) h1[mt][r] = tma_tanh(h1[mt][r]);
        }
    }
    H64_TICK(0);
    if constexpr (!TMA_H64_PIPE) store_tile_t(slotA, h1, r16, g);
    // ---- layer 2: the B operand of k-step (j, r) is register r of h1's tile j ----
    {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) h2[mt] = *reinterpret_cast<const f32x4 *>(wimg + IMG_B2 + 16 * mt + 4 * g);
        if constexpr (TMA_H64_PIPE) {
            chain64_act(wimg + IMG_W2F + 4 * g * 64 + r16 * 4, h1, h2, [&](int j, int r) { return tma_tanh(h1[j][r]); },
                        [&](int j) { *reinterpret_cast<f32x4 *>(slotA + tsw(r16, 16 * j + 4 * g)) = h1[j]; });
        } else {
            chain64(wimg + IMG_W2F + 4 * g * 64 + r16 * 4, h1, h2);
        }
        H64_TICK(1);
        if constexpr (!TMA_H64_PIPE) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) h2[mt][r] = tma_tanh(h2[mt][r]);
        }
    }
    if constexpr (!TMA_H64_PIPE) store_tile_t(slotB, h2, r16, g);
    H64_TICK(2);
    // ---- head: the A operand's row m = lane & 15 carries output a(m) = (m >> 2) + 4 (m & 3), so register r of lane group g' is output
    // g' + 4r: the n_out <= 8 real outputs sit in registers 0..1 and the head's input-gradient GEMM below needs ceil(n_out / 4) k-steps
    // instead of 4.  Two accumulators halve the dependent MFMA chain.
    const int acol = (r16 >> 2) + 4 * (r16 & 3);
    f32x4 o0 = f32x4{wimg[IMG_B3 + g], wimg[IMG_B3 + g + 4], wimg[IMG_B3 + g + 8], wimg[IMG_B3 + g + 12]}, o1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float w = wimg[IMG_W3F + (16 * j + 4 * g + r) * 16 + acol];
            if constexpr (TMA_H64_PIPE) h2[j][r] = tma_tanh(h2[j][r]);  // (the head's MFMAs of element (j, r) run under the next tanh)
            if ((4 * j + r) & 1) o1 = mfma16(w, h2[j][r], o1);
            else o0 = mfma16(w, h2[j][r], o0);
        }
    if constexpr (TMA_H64_PIPE) store_tile_t(slotB, h2, r16, g);
    H64_TICK(3);
    const f32x4 dz3 = h64t_loss<IS_PI>(o0, o1, m0, m1, act, valid, A, invB, amean, astd, hp, st, lane);
    H64_TICK(4);
    *reinterpret_cast<f32x4 *>(dz3t + r16 * 16 + 4 * g) = dz3;  // (column m = 4g + r of the tile <-> output a(m), undone by flush_segment)
    H64_TICK(4);
    // ---- dh2^T = W3 . dz3^T: k-step r contracts over the outputs {g' + 4r}; registers r >= ceil(n_out / 4) of dz3 are zero ----
    f32x4 d[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) d[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (4 * r < NOUT) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(wimg + IMG_W3B + (g + 4 * r) * 64 + r16 * 4);
#pragma unroll
            for (int mt = 0; mt < 4; mt++) d[mt] = mfma16(w[mt], dz3[r], d[mt]);
        }
    }
    // dW3 (off the dependent path) right behind the chain's MFMAs: the pipe works on it while dh2 matures and dz2 is formed
    if (w3_blocks(NOUT)) {  // (uniform)
        {
            float bf[4];
#pragma unroll
            for (int sk = 0; sk < 4; sk++) bf[sk] = dz3t[(4 * sk + g) * 16 + r16];
            acc.b3[0] += (bf[0] + bf[1]) + (bf[2] + bf[3]);
        }
        const bool two = NOUT > 4;
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 8) {  // (eight samples of operands at a time: 24 registers)
            float av[8], b0[8], b1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                av[u] = slotB[tsw(s0 + u, lane)];                  // h2[s][k = lane]
                b0[u] = dz3t[(s0 + u) * 16 + 4 * (lane & 3)];      // dz3[s][a = lane & 3]: tile column m = 4 a
                b1[u] = dz3t[(s0 + u) * 16 + 4 * (lane & 3) + 1];  // dz3[s][a = 4 + (lane & 3)]: column m = 4 (a & 3) + 1
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc.w3[0][0] = mfma4(av[u], b0[u], acc.w3[0][0]);
                if (two) acc.w3[1][0] = mfma4(av[u], b1[u], acc.w3[1][0]);
            }
        }
    } else {
        bwd_weight_acc_t<4, 1, true, false>(slotB, dz3t, acc.w3, acc.b3, lane);
    }
    H64_TICK(5);
    // ---- dz2 = dh2 * (1 - h2^2) over h2 (dW3 has read h2: LDS operations of one wave execute in order), dh1^T = W2 . dz2^T (chain),
    // then dW2 (64 MFMAs, off the path) with dz1 = dh1 * (1 - h1^2) formed between its k-tiles ----
    f32x4 e[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) e[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (TMA_H64_PIPE) {
        chain64_act(wimg + IMG_W2B + 4 * g * 64 + r16 * 4, h2, e, [&](int j, int r) { return d[j][r] * (1.0f - h2[j][r] * h2[j][r]); },
                    [&](int j) { *reinterpret_cast<f32x4 *>(slotB + tsw(r16, 16 * j + 4 * g)) = h2[j]; });
        H64_TICK(6);
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) h2[mt][r] = d[mt][r] * (1.0f - h2[mt][r] * h2[mt][r]);
        store_tile_t(slotB, h2, r16, g);
        H64_TICK(6);
        chain64(wimg + IMG_W2B + 4 * g * 64 + r16 * 4, h2, e);
    }
    H64_TICK(7);
    {
        float bf[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
#pragma unroll
            for (int sk = 0; sk < 4; sk++) bf[nt][sk] = slotB[tsw(4 * sk + g, nt * 16 + r16)];
            acc.b2[nt] += (bf[nt][0] + bf[nt][1]) + (bf[nt][2] + bf[nt][3]);
        }
        float av[4][4];
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
            for (int sk = 0; sk < 4; sk++) av[kt][sk] = slotA[tsw(4 * sk + g, kt * 16 + r16)];
        // h1 is not kept in registers across the head / loss / layer-2 work: its LDS copy (slot A) is read back in the lane's own C-layout
        // positions (the b128 pattern of the store: conflict-free); every read of slot A is issued before dz1 overwrites it below
        f32x4 hh[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) hh[mt] = *reinterpret_cast<const f32x4 *>(slotA + tsw(r16, 16 * mt + 4 * g));
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sk = 0; sk < 4; sk++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) acc.w2[kt][nt] = mfma16(av[kt][sk], bf[nt][sk], acc.w2[kt][nt]);
#pragma unroll
            for (int r = 0; r < 4; r++) h1[kt][r] = e[kt][r] * (1.0f - hh[kt][r] * hh[kt][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    store_tile_t(slotA, h1, r16, g);  // dz1 over h1
    H64_TICK(8);
    if constexpr (w1_blocks<KS1C>()) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            float bf[4];
#pragma unroll
            for (int sk = 0; sk < 4; sk++) bf[sk] = slotA[tsw(4 * sk + g, nt * 16 + r16)];
            acc.b1[nt] += (bf[0] + bf[1]) + (bf[2] + bf[3]);
        }
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 8) {
            float bv[8], xa[KS1C][8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                bv[u] = slotA[tsw(s0 + u, lane)];  // dz1[s][n = lane]
#pragma unroll
                for (int blk = 0; blk < KS1C; blk++) xa[blk][u] = Xt[(s0 + u) * 16 + 4 * blk + (lane & 3)];  // X[s][k = 4 blk + (lane & 3)]
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int blk = 0; blk < KS1C; blk++) acc.w1[0][blk] = mfma4(xa[blk][u], bv[u], acc.w1[0][blk]);
        }
    } else {
        bwd_weight_acc_t<1, 4, false, true>(Xt, slotA, acc.w1, acc.b1, lane);
    }
    H64_TICK(9);
}

// ------------------------------------------------------------------------------------------
// The FORWARD half of h64t_tile on its own (round 3): policy_fwd_h64_kernel and the fused H = 64 rollout chunks run the same transposed
// register chain as the update kernels -- observation features straight from registers as the B operand, weights from the LDS image as
// the A operand, tanh under the next k-step's MFMAs, no activation ever written to LDS -- instead of the LDS round-trip chain
// (dense64_tanh_lds: every layer's output stored and read back as the next A operand, 3.6 us per vector step at 4096 envs).
// Same instructions in the same order as the forward part of h64t_tile, so the log-probabilities / values a rollout stores are bit for
// bit what the first epoch of the update recomputes from the same parameters.
// wimg: forward part of the net's image ([0, IMG_FWD_FLOATS)); b1 / b2 / b3: its biases.  xb[ks] = observation feature 4 ks + g of sample
// lane & 15 (0 beyond D).  Result: register r of lane group g holds head output g + 4 r of sample lane & 15 as o0[r] + o1[r].
template <int KS1C>
__device__ __forceinline__ void h64t_forward(const float *wimg, const float *b1, const float *b2, const float *b3, const float (&xb)[KS1C], int KS1,
                                             f32x4 &o0, f32x4 &o1, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 h1[4], h2[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) h1[mt] = *reinterpret_cast<const f32x4 *>(b1 + 16 * mt + 4 * g);
#pragma unroll
    for (int ks = 0; ks < KS1C; ks++) {
        if (ks < KS1) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(wimg + IMG_W1 + (4 * ks + g) * 64 + r16 * 4);
#pragma unroll
            for (int mt = 0; mt < 4; mt++) h1[mt] = mfma16(w[mt], xb[ks], h1[mt]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++) h2[mt] = *reinterpret_cast<const f32x4 *>(b2 + 16 * mt + 4 * g);
    chain64_act(wimg + IMG_W2F + 4 * g * 64 + r16 * 4, h1, h2, [&](int j, int r) { return tma_tanh(h1[j][r]); }, [](int) {});
    const int acol = (r16 >> 2) + 4 * (r16 & 3);
    o0 = f32x4{b3[g], b3[g + 4], b3[g + 8], b3[g + 12]}, o1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float w = wimg[IMG_W3F + (16 * j + 4 * g + r) * 16 + acol];
            h2[j][r] = tma_tanh(h2[j][r]);
            if ((4 * j + r) & 1) o1 = mfma16(w, h2[j][r], o1);
            else o0 = mfma16(w, h2[j][r], o0);
        }
}

// The same forward pass with every operand of the net in REGISTERS (round 6; the fused rollout kernels, whose weights do not change during a
// launch and whose waves have 512 registers to themselves): 64 registers of layer-2 weights, 16 of the head, the layer-1 rows and the
// biases are read from the LDS image once per launch instead of ~40 LDS reads per vector step -- the head's sixteen operands had been
// coming out as eight dependent read-wait-multiply round trips.  Instruction for instruction h64t_forward's arithmetic: same bits.
template <int KS1C>
struct H64FwdRegs {
    f32x4 b1[4], b2[4], w1[KS1C], w2[16], b3o;
    float w3[16];
};
template <int KS1C>
__device__ __forceinline__ void h64t_load_fwd(const float *wimg, const float *b1, const float *b2, const float *b3, int KS1, H64FwdRegs<KS1C> &R, int lane) {
    const int r16 = lane & 15, g = lane >> 4, acol = (r16 >> 2) + 4 * (r16 & 3);
#pragma unroll
    for (int mt = 0; mt < 4; mt++) R.b1[mt] = *reinterpret_cast<const f32x4 *>(b1 + 16 * mt + 4 * g), R.b2[mt] = *reinterpret_cast<const f32x4 *>(b2 + 16 * mt + 4 * g);
#pragma unroll
    for (int ks = 0; ks < KS1C; ks++) R.w1[ks] = *reinterpret_cast<const f32x4 *>(wimg + IMG_W1 + (4 * (ks < KS1 ? ks : 0) + g) * 64 + r16 * 4);
    const float *wrow = wimg + IMG_W2F + 4 * g * 64 + r16 * 4;
#pragma unroll
    for (int i = 0; i < 16; i++) R.w2[i] = *reinterpret_cast<const f32x4 *>(wrow + (16 * (i >> 2) + (i & 3)) * 64);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) R.w3[4 * j + r] = wimg[IMG_W3F + (16 * j + 4 * g + r) * 16 + acol];
    R.b3o = f32x4{b3[g], b3[g + 4], b3[g + 8], b3[g + 12]};
}
template <int KS1C>
__device__ __forceinline__ void h64t_forward_r(const H64FwdRegs<KS1C> &R, const float (&xb)[KS1C], int KS1, f32x4 &o0, f32x4 &o1) {
    f32x4 h1[4], h2[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) h1[mt] = R.b1[mt];
#pragma unroll
    for (int ks = 0; ks < KS1C; ks++) {
        if (ks < KS1) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++) h1[mt] = mfma16(R.w1[ks][mt], xb[ks], h1[mt]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++) h2[mt] = R.b2[mt];
    // chain64_act's order: the activations of tile j + 1 are formed under the MFMAs of tile j
#pragma unroll
    for (int r = 0; r < 4; r++) h1[0][r] = tma_tanh(h1[0][r]);
#pragma unroll
    for (int i = 0; i < 16; i++) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) h2[mt] = mfma16(R.w2[i][mt], h1[i >> 2][i & 3], h2[mt]);
        if (i + 4 < 16) h1[(i + 4) >> 2][(i + 4) & 3] = tma_tanh(h1[(i + 4) >> 2][(i + 4) & 3]);
    }
    o0 = R.b3o, o1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            h2[j][r] = tma_tanh(h2[j][r]);
            if ((4 * j + r) & 1) o1 = mfma16(R.w3[4 * j + r], h2[j][r], o1);
            else o0 = mfma16(R.w3[4 * j + r], h2[j][r], o0);
        }
}

// The same forward pass split over TWO waves of a net (round 6, rollout_chunk4_h64_kernel): both form layer 1 (four MFMAs), wave `half` runs
// layer 2 for output tiles 2 half and 2 half + 1 (32 of the 64 MFMAs) and hands their activations to the other through LDS; the head then
// runs on half 0.  Per accumulator the operations and their order are h64t_forward's: same bits.
template <int KS1C>
__device__ __forceinline__ void h64t_forward_half(const H64FwdRegs<KS1C> &R, const float (&xb)[KS1C], int KS1, int half, f32x4 (&t2)[2]) {
    f32x4 h1[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) h1[mt] = R.b1[mt];
#pragma unroll
    for (int ks = 0; ks < KS1C; ks++) {
        if (ks < KS1) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++) h1[mt] = mfma16(R.w1[ks][mt], xb[ks], h1[mt]);
        }
    }
    f32x4 h2[2];
    h2[0] = half ? R.b2[2] : R.b2[0], h2[1] = half ? R.b2[3] : R.b2[1];
#pragma unroll
    for (int r = 0; r < 4; r++) h1[0][r] = tma_tanh(h1[0][r]);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float wa = half ? R.w2[i][2] : R.w2[i][0], wb = half ? R.w2[i][3] : R.w2[i][1];
        h2[0] = mfma16(wa, h1[i >> 2][i & 3], h2[0]);
        h2[1] = mfma16(wb, h1[i >> 2][i & 3], h2[1]);
        if (i + 4 < 16) h1[(i + 4) >> 2][(i + 4) & 3] = tma_tanh(h1[(i + 4) >> 2][(i + 4) & 3]);
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) t2[m][r] = tma_tanh(h2[m][r]);
}
// the head over the four activated tiles (tiles 0, 1: this wave's; 2, 3: the partner's)
template <int KS1C>
__device__ __forceinline__ void h64t_head_r(const H64FwdRegs<KS1C> &R, const f32x4 (&ta)[2], const f32x4 (&tb)[2], f32x4 &o0, f32x4 &o1) {
    o0 = R.b3o, o1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float h = j < 2 ? ta[j][r] : tb[j - 2][r];
            if ((4 * j + r) & 1) o1 = mfma16(R.w3[4 * j + r], h, o1);
            else o0 = mfma16(R.w3[4 * j + r], h, o0);
        }
}

__device__ __forceinline__ float xg_min(float v) { return xg_reduce(v, [](float a, float b) { return fminf(a, b); }); }

// Categorical action and its log-probability from the head outputs of h64t_forward (output a = g + 4 r in register r of lane group g).
// Sampling is the Gumbel-max form of the categorical draw: action = argmax_a (logit_a + G_a), G_a = -log(-log u_a), u_a from the counter
// stream (seed, global env, step, a) -- exactly Categorical(softmax(logits)), and the noise does not depend on the logits, so it is formed
// under the head's MFMAs and the action needs two cross-lane-group reductions instead of a prefix scan over the probabilities.
// deterministic: the first maximal logit (SB3 predict(deterministic=True)).  Softmax / log-prob arithmetic = h64t_loss's.
// Every lane of the sample's four lane groups returns the same (action, log-prob).
// (the Gumbel noise of a lane's four outputs depends on the counters only: h64t_gumbel forms it -- a step ahead on an idle wave in
//  rollout_chunk4_h64_kernel -- and h64t_act_n takes it; h64t_act = the two in sequence, the same values either way)
__device__ __forceinline__ f32x4 h64t_gumbel(uint32_t rng_seed, uint32_t global_env, uint32_t rng_step, int det, int lane) {
    const int g = lane >> 4;
    f32x4 noise = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (!det) {
            const uint32_t h = mix32(rng_seed ^ (0x9E3779B9u * (uint32_t)(g + 4 * r + 1)), global_env, rng_step);
            // 23 bits: k + 0.5 is exact for k < 2^23, so u <= 1 - 2^-24 < 1 (with 24 bits k = 0xFFFFFF rounds to 2^24: u == 1, noise == +inf)
            const float u = ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);  // (0, 1)
            noise[r] = -__logf(-__logf(u));
        }
    }
    return noise;
}
__device__ __forceinline__ void h64t_act_n(const f32x4 &o0, const f32x4 &o1, int A, const f32x4 &noise, int &act_out, float &lp_out, int lane) {
    const int g = lane >> 4;
    float x[4], key[4];
    bool ok[4];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        ok[r] = g + 4 * r < A;
        x[r] = ok[r] ? o0[r] + o1[r] : -INFINITY;
        m = fmaxf(m, x[r]);
    }
    m = xg_max(m);
    float km = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        key[r] = ok[r] ? x[r] + noise[r] : -INFINITY;
        km = fmaxf(km, key[r]);
    }
    km = xg_max(km);
    float first = 99.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) first = fminf(first, (ok[r] && key[r] == km) ? (float)(g + 4 * r) : 99.0f);
    const int act = (int)xg_min(first);
    float ssum = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) ssum += ok[r] ? __expf(x[r] - m) : 0.0f;
    ssum = xg_sum(ssum);
    const float lse = m + __logf(ssum);
    float lpa = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) lpa += (g + 4 * r == act) ? (ok[r] ? x[r] - lse : 0.0f) : 0.0f;
    act_out = act;
    lp_out = xg_sum(lpa);
}
// h64t_act_n in two parts for rollout_chunk4_h64_kernel, whose policy wave only needs the ACTION before it can step the envs: xs = o0 + o1.
// h64t_argmax: the Gumbel-max draw as ONE cross-lane-group reduction over (key, index) pairs -- the greatest key, the lowest index among
// equal keys: what h64t_act_n's max-then-first-index pair of reductions returns.
__device__ __forceinline__ int h64t_argmax(const f32x4 &xs, int A, const f32x4 &noise, int lane) {
    const int g = lane >> 4;
    float kb = -INFINITY;
    uint32_t ib = 99u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float key = (g + 4 * r < A) ? xs[r] + noise[r] : -INFINITY;
        if (key > kb) kb = key, ib = (uint32_t)(g + 4 * r);
    }
    {
        const auto k = __builtin_amdgcn_permlane16_swap(__float_as_uint(kb), __float_as_uint(kb), false, false);
        const auto i = __builtin_amdgcn_permlane16_swap(ib, ib, false, false);
        const float k0 = __uint_as_float(k[0]), k1 = __uint_as_float(k[1]);
        const bool t1 = k1 > k0 || (k1 == k0 && i[1] < i[0]);
        kb = t1 ? k1 : k0, ib = t1 ? i[1] : i[0];
    }
    {
        const auto k = __builtin_amdgcn_permlane32_swap(__float_as_uint(kb), __float_as_uint(kb), false, false);
        const auto i = __builtin_amdgcn_permlane32_swap(ib, ib, false, false);
        const float k0 = __uint_as_float(k[0]), k1 = __uint_as_float(k[1]);
        const bool t1 = k1 > k0 || (k1 == k0 && i[1] < i[0]);
        ib = t1 ? i[1] : i[0];
    }
    return (int)ib;
}
// h64t_logp: log-probability of action `act` under softmax(xs): h64t_act_n's operations in its order (any wave with the same lane layout)
__device__ __forceinline__ float h64t_logp(const f32x4 &xs, int A, int act, int lane) {
    const int g = lane >> 4;
    float x[4];
    bool ok[4];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        ok[r] = g + 4 * r < A;
        x[r] = ok[r] ? xs[r] : -INFINITY;
        m = fmaxf(m, x[r]);
    }
    m = xg_max(m);
    float ssum = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) ssum += ok[r] ? __expf(x[r] - m) : 0.0f;
    ssum = xg_sum(ssum);
    const float lse = m + __logf(ssum);
    float lpa = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) lpa += (g + 4 * r == act) ? (ok[r] ? x[r] - lse : 0.0f) : 0.0f;
    return xg_sum(lpa);
}
__device__ __forceinline__ void h64t_act(const f32x4 &o0, const f32x4 &o1, int A, uint32_t rng_seed, uint32_t global_env, uint32_t rng_step, int det,
                                         int &act_out, float &lp_out, int lane) {
    h64t_act_n(o0, o1, A, h64t_gumbel(rng_seed, global_env, rng_step, det, lane), act_out, lp_out, lane);
}

// accumulator register idx (0..104) of a NetAcc, and the flat parameter index it holds in lane `lane`
constexpr int FL_HALF = 56, FL_REGS = 105;
__device__ __forceinline__ float acc_reg(const NetAcc &a, int idx) {  // idx is a compile-time constant after unrolling
    if (idx < 16) return a.w1[0][idx >> 2][idx & 3];
    if (idx < 80) return a.w2[(idx - 16) >> 4][((idx - 16) >> 2) & 3][idx & 3];
    if (idx < 96) return a.w3[(idx - 80) >> 2][0][idx & 3];
    if (idx < 100) return a.b1[idx - 96];
    if (idx < 104) return a.b2[idx - 100];
    return a.b3[0];
}
// slab element of accumulator register idx in lane `lane` (-1: padding, nothing to store)
template <bool IS_PI, int KS1C>
__device__ __forceinline__ int slab_offset_t(int idx, int lane, const PLayout &L, int D, int NOUT) {
    const int r16 = lane & 15, g = lane >> 4;
    const int perm = (r16 >> 2) + 4 * (r16 & 3);  // head: tile column m <-> output a(m)
    if (idx < 16) {
        if constexpr (w1_blocks<KS1C>()) {  // register i of observation block blk: dW1[k = 4 blk + i][n = lane]
            const int blk = idx >> 2, k = 4 * blk + (idx & 3);
            return (blk < KS1C && k < D) ? (IS_PI ? L.pW1t : L.vW1t) + k * 64 + lane : -1;
        }
        const int k = 4 * g + (idx & 3), n = (idx >> 2) * 16 + r16;
        return k < D ? (IS_PI ? L.pW1t : L.vW1t) + k * 64 + n : -1;
    }
    if (idx < 80) {
        const int t = idx - 16, k = (t >> 4) * 16 + 4 * g + (t & 3), n = ((t >> 2) & 3) * 16 + r16;
        return (IS_PI ? L.pW2t : L.vW2t) + k * 64 + n;
    }
    if (idx < 96) {
        if (w3_blocks(NOUT)) {  // register i of output half h: dW3[k = 4 (lane >> 2) + i][a = (lane & 3) + 4 h]
            const int t = idx - 80, h = t >> 2, k = 4 * (lane >> 2) + (t & 3), a = (lane & 3) + 4 * h;
            return (h < 2 && a < NOUT) ? (IS_PI ? L.pW3t : L.vW3t) + k * NOUT + a : -1;
        }
        const int t = idx - 80, k = (t >> 2) * 16 + 4 * g + (t & 3);
        return perm < NOUT ? (IS_PI ? L.pW3t : L.vW3t) + k * NOUT + perm : -1;
    }
    if (g != 0) return -1;  // bias sums are replicated over the lane groups
    if (idx < 100) return (IS_PI ? L.pb1 : L.vb1) + (idx - 96) * 16 + r16;
    if (idx < 104) return (IS_PI ? L.pb2 : L.vb2) + (idx - 100) * 16 + r16;
    return perm < NOUT ? (IS_PI ? L.pb3 : L.vb3) + perm : -1;
}

}  // namespace tma

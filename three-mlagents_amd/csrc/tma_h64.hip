// tma_h64.hip -- PPO minibatch forward + loss + backward for H = 64 policies (BASELINE configs[1]: GridWorld, PPO MLP(64,64)).
//
// Replaces, for SB3's default-width MlpPolicy, what PPO.train computes per minibatch between RolloutBuffer.get and
// optimizer.step (stable-baselines3 2.9.0, third party; constructed at /root/reference/backend/mlagents/training.py:150, driven by
// model.learn() at training.py:166-170; formulas: SURVEY.md Appendix C.3 / C.5).  tma_policy.hip dispatches here.
#include "tma_h64_tile.h"

#include <cstdlib>

namespace tma {

// ------------------------------------------------------------------------------------------
// H = 64 specialisation (BASELINE configs[1]): persistent waves keep the WHOLE parameter gradient of both nets in MFMA
// accumulators (210 VGPRs) while they walk their share of the 16-sample tiles -- dW += X^T.dZ is accumulated through the
// MFMA C operand, so there is no per-tile gradient traffic at all.  At the end the 4 waves of a block are summed through
// LDS and the block writes ONE partial-gradient slab with plain stores; slab_reduce_kernel sums the slabs in a fixed order
// (bitwise reproducible, no float atomics: the per-tile atomics of the generic kernel serialise on a 37 KB buffer).
// Requirements: H == 64, D <= 16, Discrete head (A <= 16).
// ------------------------------------------------------------------------------------------

template <int KT, int NT>
__device__ __forceinline__ void bwd_weight_acc(const float *xin, int ldx, int K, const float *dz, int ldz, int N, f32x4 (&accW)[KT][NT],
                                               float (&accb)[NT], int lane) {
    (void)K, (void)N;
    const int r16 = lane & 15, g = lane >> 4;
    float bf[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = nt * 16 + r16;
#pragma unroll
        for (int s = 0; s < 4; s++) bf[nt][s] = dz[(4 * s + g) * ldz + col];  // dz tiles are written with zeros in columns >= N
        accb[nt] += (bf[nt][0] + bf[nt][1]) + (bf[nt][2] + bf[nt][3]);
    }
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        const int krow = kt * 16 + r16;
        float a[4];
        // rows krow >= K read whatever follows in LDS: they only feed accumulator rows k >= K, which flush_segment never stores
#pragma unroll
        for (int s = 0; s < 4; s++) a[s] = xin[(4 * s + g) * ldx + krow];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) accW[kt][nt] = mfma16(a[s], bf[nt][s], accW[kt][nt]);
    }
}

// wave -> LDS staging of one [K][N] segment (+ its bias), then block sum -> slab
// PERM (version-2 head, one output tile): accumulator / tile column m stands for output a(m) = (m >> 2) + 4 (m & 3)
template <int KT, int NT, bool PERM = false>
__device__ __forceinline__ void flush_segment(float *stage_all, int wave, int wpb, int K, int N, const f32x4 (&accW)[KT][NT], float (&accb)[NT],
                                              float *slab_w, float *slab_b, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    const int seg = K * N + N;
    float *stage = stage_all + wave * seg;
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = kt * 16 + g * 4 + r, n = PERM ? (r16 >> 2) + 4 * (r16 & 3) : nt * 16 + r16;
                if (k < K && n < N) stage[k * N + n] = accW[kt][nt][r];
            }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        float v = accb[nt];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int n = PERM ? (r16 >> 2) + 4 * (r16 & 3) : nt * 16 + r16;
        if (g == 0 && n < N) stage[K * N + n] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < seg; e += blockDim.x) {
        float sum = stage_all[e];
        for (int w = 1; w < wpb; w++) sum += stage_all[w * seg + e];
        if (e < K * N) slab_w[e] = sum;
        else slab_b[e - K * N] = sum;
    }
    __syncthreads();
}

// One launch, 2 x n_slabs blocks: even blocks carry the POLICY net, odd blocks the VALUE net (the two MLPs share nothing,
// SB3 net_arch=dict(pi=..., vf=...)), so a wave holds only ~105 accumulator registers and two blocks fit per CU.
template <bool IS_PI, int DT>  // DT > 0: compile-time observation width (folds the LDS addressing), 0: runtime L.D
__device__ __forceinline__ void grad_h64_body(const float *__restrict__ params, const PLayout &L, const Rollout &rb, const Minibatch &mb,
                                              const HParams &hp, const double *__restrict__ adv_part, int n_part, float *__restrict__ slab,
                                              double *__restrict__ stat_slot, float *smem, int n_blocks_net, int block_net) {
    __shared__ float adv_ms[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    constexpr int H = 64;
    const int D = DT > 0 ? DT : L.D, A = L.A;
    const int ldx = ((D + 3) & ~3) + 2;
    constexpr int ld = H + 2, ld3 = 34;
    const int per_wave = 16 * (ldx + 2 * ld + ld3) + 16 * 8;
    float *wimg = smem;  // this net's weight image, staged once per block
    float *X = smem + IMG_FLOATS + (int64_t)wave * per_wave;
    // the input-gradient tiles overwrite the activations they are derived from, element for element (dz2 over h2, dz1 over h1)
    float *h1 = X + 16 * ldx, *h2 = h1 + 16 * ld, *dzA = h2, *dzB = h1, *dz3 = h2 + 16 * ld;
    int64_t *row_off = reinterpret_cast<int64_t *>(dz3 + 16 * ld3);
    float *meta = reinterpret_cast<float *>(row_off + 16);
    const float invB = 1.0f / (float)mb.count;
    const int NOUT = IS_PI ? A : 1;
    const int KS1 = (D + 3) >> 2;
    stage_copy(params + (IS_PI ? L.img_pi : L.img_vf), wimg, IMG_FLOATS);
    if (IS_PI && hp.normalize_advantage && threadIdx.x < 64) {  // fold the minibatch advantage partials (same order as adv_final_kernel)
        double a = 0.0, bsum = 0.0;
        for (int k = threadIdx.x; k < n_part; k += 64) a += adv_part[2 * k], bsum += adv_part[2 * k + 1];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            bsum += __shfl_down(bsum, o, 64);
        }
        if (threadIdx.x == 0) {
            const double n = (double)mb.stats_n, mean = a / n;
            double var = n > 1.0 ? (bsum - n * mean * mean) / (n - 1.0) : 0.0;
            if (var < 0.0) var = 0.0;
            adv_ms[0] = (float)mean;
            adv_ms[1] = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float amean = (IS_PI && hp.normalize_advantage) ? adv_ms[0] : 0.0f;
    const float astd = (IS_PI && hp.normalize_advantage) ? adv_ms[1] : 1.0f;
    NetAcc acc;
    zero_acc(acc);
    double st_a = 0.0, st_ent = 0.0, st_kl = 0.0, st_clip = 0.0, st_n = 0.0;
    const int64_t n_tiles = (mb.count + 15) >> 4;
    // The gather of a tile (permutation index, 3-4 scalars and the observation row per sample: dependent global loads) is
    // issued one tile AHEAD into registers and committed to LDS at the top of the next iteration, so its latency hides
    // under the current tile's MFMA work instead of stalling every tile (compile-time D only).
    constexpr int DP_CT = (DT + 3) & ~3;
    constexpr int NV = DT > 0 ? DP_CT / 4 : 1;  // observation values per lane: 16 rows x DP_CT floats / 64 lanes
    int64_t pf_off = -1;
    float pf_m0 = 0.0f, pf_m1 = 0.0f, pf_m2 = 0.0f, pf_m3 = 0.0f, pf_x[NV];
    const int64_t tile_stride = (int64_t)n_blocks_net * wpb;
    int32_t pf_noff = -1;  // cached sample offset of the tile AFTER the one being fetched: its load latency never sits in front of the gathers
    bool have_noff = false;
    auto fetch = [&](int64_t tl) {
        pf_off = -1, pf_m0 = pf_m1 = pf_m2 = pf_m3 = 0.0f;
        const int32_t my_noff = pf_noff;
        if (mb.offs && lane < 16) {
            const int64_t j2 = ((tl + tile_stride) << 4) + lane;
            pf_noff = mb.offs[j2 < mb.count ? j2 : 0];
        }
        if (lane < 16 && tl < n_tiles) {
            const int64_t j = (tl << 4) + lane;
            if (j < mb.count) {
                pf_off = mb.offs ? (int64_t)(have_noff ? my_noff : mb.offs[j]) : sample_offset(mb, mb.start + j, rb.T, rb.N);
                if constexpr (IS_PI) {
                    pf_m0 = rb.log_probs[pf_off];
                    pf_m1 = rb.advantages[pf_off];
                    pf_m3 = __int_as_float(static_cast<const int32_t *>(rb.actions)[pf_off]);
                } else {
                    pf_m2 = rb.returns[pf_off];
                }
            }
        }
        if constexpr (DT > 0) {
#pragma unroll
            for (int q = 0; q < NV; q++) {
                const int e = lane + 64 * q, row = e / DP_CT, c = e - row * DP_CT;
                const int64_t orow = __shfl(pf_off, row, 64);
                pf_x[q] = (orow >= 0 && c < DT) ? rb.obs[orow * DT + c] : 0.0f;
            }
        }
    };
    auto commit = [&]() {
        if (lane < 16) {
            meta[lane * 4 + 0] = pf_m0, meta[lane * 4 + 1] = pf_m1, meta[lane * 4 + 2] = pf_m2, meta[lane * 4 + 3] = pf_m3;
            row_off[lane] = pf_off;
        }
        if constexpr (DT > 0) {
#pragma unroll
            for (int q = 0; q < NV; q++) {
                const int e = lane + 64 * q, row = e / DP_CT, c = e - row * DP_CT;
                X[row * ldx + c] = pf_x[q];
            }
        }
    };
    fetch((int64_t)block_net * wpb + wave);
    have_noff = true;
    for (int64_t tile = (int64_t)block_net * wpb + wave; tile < n_tiles; tile += tile_stride) {
        commit();
        if constexpr (DT == 0) load_obs_tile(rb.obs, row_off, D, X, ldx, lane);
        fetch(tile + tile_stride);
        dense64_tanh_lds<0>(X, ldx, KS1, wimg + IMG_W1, wimg + IMG_B1, h1, ld, lane);
        dense64_tanh_lds<16>(h1, ld, 16, wimg + IMG_W2F, wimg + IMG_B2, h2, ld, lane);
        f32x4 out[1];
        out[0] = dense64_head_lds(h2, ld, wimg + IMG_W3F, wimg + IMG_B3, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = g * 4 + r;
            const bool valid = row_off[row] >= 0;
            if constexpr (IS_PI) {
                const bool colok = r16 < A;
                const float x = colok ? out[0][r] : -INFINITY;
                const float m = gmax16(x);
                const float e = colok ? expf(x - m) : 0.0f;
                const float s = gsum16(e);
                const float lse = m + logf(s);
                const float lp = colok ? x - lse : 0.0f;
                const float p = e / s;
                const int act = __float_as_int(meta[row * 4 + 3]);
                const float lpa = gsum16((r16 == act) ? lp : 0.0f);
                const float ent = -gsum16(p * lp);
                const float old = meta[row * 4 + 0];
                const float advn = (meta[row * 4 + 1] - amean) / (astd + 1e-8f);
                const float ratio = expf(lpa - old);
                const float pl1 = advn * ratio;
                const float rc = fminf(fmaxf(ratio, 1.0f - hp.clip_range), 1.0f + hp.clip_range);
                const float pl2 = advn * rc;
                const float g_lp = (valid && pl1 <= pl2) ? -(advn * ratio) * invB : 0.0f;
                float dl = g_lp * (((r16 == act) ? 1.0f : 0.0f) - p);
                dl += valid ? (hp.ent_coef * invB) * (p * (lp + ent)) : 0.0f;
                dz3[row * ld3 + r16] = colok ? dl : 0.0f;
                if (valid && r16 == 0) {
                    st_a += (double)(-fminf(pl1, pl2));
                    st_ent += (double)ent;
                    st_kl += (double)((ratio - 1.0f) - (lpa - old));
                    st_clip += (fabsf(ratio - 1.0f) > hp.clip_range) ? 1.0 : 0.0;
                    st_n += 1.0;
                }
            } else {
                const float diff = out[0][r] - meta[row * 4 + 2];
                dz3[row * ld3 + r16] = (valid && r16 == 0) ? (hp.vf_coef * 2.0f * invB) * diff : 0.0f;
                if (valid && r16 == 0) st_a += (double)(diff * diff);
            }
        }
        bwd_weight_acc<4, 1>(h2, ld, H, dz3, ld3, NOUT, acc.w3, acc.b3, lane);
        dense64_bwd_input_lds<4>(dz3, ld3, wimg + IMG_W3B, h2, ld, dzA, ld, lane);
        bwd_weight_acc<4, 4>(h1, ld, H, dzA, ld, H, acc.w2, acc.b2, lane);
        dense64_bwd_input_lds<16>(dzA, ld, wimg + IMG_W2B, h1, ld, dzB, ld, lane);
        bwd_weight_acc<1, 4>(X, ldx, D, dzB, ld, H, acc.w1, acc.b1, lane);
    }
    __syncthreads();
    flush_segment<1, 4>(smem, wave, wpb, D, H, acc.w1, acc.b1, slab + (IS_PI ? L.pW1t : L.vW1t), slab + (IS_PI ? L.pb1 : L.vb1), lane);
    flush_segment<4, 4>(smem, wave, wpb, H, H, acc.w2, acc.b2, slab + (IS_PI ? L.pW2t : L.vW2t), slab + (IS_PI ? L.pb2 : L.vb2), lane);
    flush_segment<4, 1>(smem, wave, wpb, H, NOUT, acc.w3, acc.b3, slab + (IS_PI ? L.pW3t : L.vW3t), slab + (IS_PI ? L.pb3 : L.vb3), lane);
    double st[5] = {st_a, st_ent, st_kl, st_clip, st_n};
#pragma unroll
    for (int q = 0; q < 5; q++)
        for (int o = 32; o > 0; o >>= 1) st[q] += __shfl_down(st[q], o, 64);
    double *red = reinterpret_cast<double *>(smem);
    if (lane == 0)
        for (int q = 0; q < 5; q++) red[wave * 5 + q] = st[q];
    __syncthreads();
    if (threadIdx.x < 5) {
        double s = 0.0;
        for (int w = 0; w < wpb; w++) s += red[w * 5 + threadIdx.x];
        // slot layout {policy_loss, value_sq_err, entropy, approx_kl, clipped, n}
        const int q = IS_PI ? (threadIdx.x == 0 ? 0 : threadIdx.x + 1) : (threadIdx.x == 0 ? 1 : -1);
        if (q >= 0) stat_slot[q] += s;
    }
}


// ------------------------------------------------------------------------------------------
// Version 2 of the tile chain: the TRANSPOSED register chain.
//
// Every GEMM of the chain is computed as  out^T[feature][sample] = W^T . in^T  -- the WEIGHTS are the MFMA A operand (M = output
// features) and the ACTIVATIONS the B operand (N = the tile's 16 samples).  The C layout of v_mfma_f32_16x16x4_f32 (register r of
// lane (g, s) = C[4g + r][s]) is then exactly the B-operand layout of the next GEMM, provided the contraction runs over the
// features in the order k-step (j, r) <-> {16j + 4g' + r : g' = 0..3}: the B operand of k-step (j, r) IS accumulator register r of
// output tile j, untouched.  The contraction order is free (the A operand follows it: same LDS weight images, rows read in the
// permuted order), so the whole forward chain, tanh, the loss and the backward input-gradient chain stay in REGISTERS: no
// activation ever makes an LDS round trip on the dependent path (version 1 wrote every layer's output to LDS and read it back as the
// A operand: six write -> read latencies per tile with the matrix pipe idle behind each).
// LDS now only carries (a) the weight images, read with addresses that do not depend on the chain, and (b) write-only copies of
// h1, h2, dz3, dz2, dz1 and the observation tile in [sample][feature] form, which the weight-gradient MFMAs (dW += x^T . dz, the
// contraction runs over SAMPLES there) read back transposed, off the dependent path.
// Observations and per-sample scalars go straight from global memory to registers, one tile ahead.
// Activation tiles: [16][64] floats with a column swizzle (tsw, tma_h64_tile.h): the b128 stores of the chain (lane (g, s) holds
// four consecutive features of sample s) and the transposed b32 reads of the weight-gradient operands are both conflict-free.
// ------------------------------------------------------------------------------------------

// ---- block reduction of the gradient accumulators (version 2): every wave stages its registers lane-for-lane (register-major:
// conflict-free b32 stores at immediate offsets), then wave w sums registers w, w + 8, ... over the eight copies in wave order (the order
// of flush_segment, so the bits match) and stores them to the slab.  Two halves of <= 56 registers keep the staging inside the tile
// region's 112 KiB; all eight waves stay busy in both phases, against three store / barrier / strided-sum / barrier rounds before.
template <bool IS_PI, int KS1C, int NWV = 8>  // NWV: waves per block (8; small-minibatch blocks: 4)
__device__ __forceinline__ void flush_all_t(float *stage, int wave, int lane, NetAcc &acc, const PLayout &L, int D, int NOUT, float *slab) {
    constexpr int QN = (FL_HALF + NWV - 1) / NWV;
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc.b1[nt] = xg_sum(acc.b1[nt]), acc.b2[nt] = xg_sum(acc.b2[nt]);
    acc.b3[0] = xg_sum(acc.b3[0]);
    // the sums stay in registers until both halves are done: a barrier behind global stores would wait out their round trip
    // (__syncthreads drains vmcnt), so every store is issued after the last barrier
    float sums[2][QN];
    // slab offsets of the registers this wave will sum: computed here, under the wait for the block's slowest wave, not behind the last barrier
    int offs[2][QN];
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
        for (int q = 0; q < QN; q++) {
            const int i = wave + NWV * q, cnt = half ? FL_REGS - FL_HALF : FL_HALF;
            offs[half][q] = i < cnt ? slab_offset_t<IS_PI, KS1C>(half * FL_HALF + i, lane, L, D, NOUT) : -1;
        }
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int base = half * FL_HALF, cnt = half ? FL_REGS - FL_HALF : FL_HALF;
        if (half) __syncthreads();  // half 0's copies have been read
#pragma unroll
        for (int i = 0; i < FL_HALF; i++)
            if (i < cnt) stage[(wave * FL_HALF + i) * 64 + lane] = acc_reg(acc, base + i);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < QN; q++) {
            const int i = wave + NWV * q;
            float sum = 0.0f;
            if (i < cnt) {
                sum = stage[i * 64 + lane];
#pragma unroll
                for (int w = 1; w < NWV; w++) sum += stage[(w * FL_HALF + i) * 64 + lane];
            }
            sums[half][q] = sum;
        }
    }
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
        for (int q = 0; q < QN; q++) {
            if (offs[half][q] >= 0) slab[offs[half][q]] = sums[half][q];
        }
}


// DIRECT (small minibatches): 4-wave blocks, one wave per SIMD, one tile per wave; the block reduction runs over four copies.
// LDS image slots of parameter x of one net (flat [W1t | b1 | W2t | b2 | W3t | b3] order): the image part of scatter_derived_h64
__device__ __forceinline__ void image_store_h64(float *wimg, int D, int n_out, int x, float val) {
    constexpr int H = 64;
    if (x < D * H) {
        const int k = x >> 6, n = x & 63;
        wimg[IMG_W1 + k * 64 + (n & 15) * 4 + (n >> 4)] = val;
        return;
    }
    x -= D * H;
    if (x < H) {
        wimg[IMG_B1 + x] = val;
        return;
    }
    x -= H;
    if (x < H * H) {
        const int k = x >> 6, n = x & 63;
        wimg[IMG_W2F + k * 64 + (n & 15) * 4 + (n >> 4)] = val;
        wimg[IMG_W2B + n * 64 + (k & 15) * 4 + (k >> 4)] = val;
        return;
    }
    x -= H * H;
    if (x < H) {
        wimg[IMG_B2 + x] = val;
        return;
    }
    x -= H;
    if (x < H * n_out) {
        const int k = x / n_out, a = x - k * n_out;
        wimg[IMG_W3F + k * 16 + a] = val;
        wimg[IMG_W3B + a * 64 + (k & 15) * 4 + (k >> 4)] = val;
        return;
    }
    x -= H * n_out;
    if (x < n_out) wimg[IMG_B3 + x] = val;
}

// The pending optimizer step of AdamFold for this workgroup's net, results into the LDS image (and, from workgroup 0 of the net, into the
// next half of the state double buffer).  clip_grad_norm_ + Adam exactly as adam_scatter_h64_kernel: same fold of the norm partials, same
// adam_update_h64.  Ownership is chosen for the LDS stores: threads 0..255 own one 4 x 4 block of W2t each -- rows kr + 16 jj, columns
// nr + 16 j -- which is one float4 per row of the forward image ([k][n & 15][n >> 4]) and one float4 per column of the input-gradient image
// ([n][k & 15][k >> 4]): 8 ds_write_b128 per thread instead of 32 scalar stores, half of them 64-way bank conflicts (a wave covers 8 nr x 8
// kr, so a b128 store meets 8 lanes per bank group: twice the ideal 4 passes).  The other <= 2 192 parameters (W1t, the biases, the head)
// go round-robin over the remaining threads (all of them in a 256-thread workgroup).  Every load is issued before the clip coefficient is
// folded, so the prologue is one memory round trip.
template <bool IS_PI>
__device__ __forceinline__ void fold_adam_into_image(const AdamFold &f, const PLayout &L, float *wimg, int block_net) {
    __shared__ double fold_red[4];
    __shared__ float fold_coef;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int base = IS_PI ? L.pW1t : L.vW1t, pn = (IS_PI ? L.vW1t : L.log_std) - base, n_out = IS_PI ? L.A : 1;
    const int w2_off = L.D * 64 + 64, n_other = pn - 4096;
    const bool own_w2 = tid < 256;
    const int o_base = nthr > 256 ? 256 : 0, o_thr = nthr - o_base, o_tid = tid - o_base;
    const int nr = ((tid >> 6) & 1) * 8 + (tid & 7), kr = ((tid >> 7) & 1) * 8 + ((tid >> 3) & 7);
    constexpr int NO = 9;  // other parameters per thread (2 192 / 256 < 9)
    float g2[4][4], m2[4][4], v2[4][4], p2[4][4], go[NO], mo[NO], vo[NO], po[NO];
    if (own_w2) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = base + w2_off + (kr + 16 * jj) * 64 + nr + 16 * j;
                g2[jj][j] = f.grad[e], m2[jj][j] = f.m_cur[e], v2[jj][j] = f.v_cur[e], p2[jj][j] = f.p_cur[e];
            }
    }
    if (o_tid >= 0) {
#pragma unroll
        for (int i = 0; i < NO; i++) {
            const int y = o_tid + i * o_thr, x = y < w2_off ? y : y + 4096, e = base + (y < n_other ? x : 0);
            go[i] = f.grad[e], mo[i] = f.m_cur[e], vo[i] = f.v_cur[e], po[i] = f.p_cur[e];
        }
    }
    if (tid < 256) {
        double a = tid < f.n_part ? f.sq_part[tid] : 0.0;
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if ((tid & 63) == 0) fold_red[tid >> 6] = a;
    }
    // image regions no parameter lands in (rows k >= D of W1, head columns >= n_out, padding) are zeros in the staged image; LDS is not
    for (int e = tid; e < 1024; e += nthr) wimg[IMG_W1 + e] = 0.0f, wimg[IMG_W3F + e] = 0.0f, wimg[IMG_W3B + e] = 0.0f;
    if (tid < 32) wimg[IMG_B3 + tid] = 0.0f;
    __syncthreads();
    if (tid == 0) {
        const double tot = ((fold_red[0] + fold_red[1]) + fold_red[2]) + fold_red[3];
        const float total_norm = (float)sqrt(tot);
        float coef = f.max_norm / (total_norm + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
        if (f.max_norm <= 0.0f) coef = 1.0f;
        fold_coef = coef;
        if (IS_PI && block_net == 0) f.norm_out[0] = (double)total_norm, f.norm_out[1] = (double)coef;
    }
    __syncthreads();
    const float coef = fold_coef, inv_bc2 = 1.0f / f.bc2_sqrt;
    const bool keep = block_net == 0;
    if (own_w2) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float gv = (g2[jj][j] * f.scale) * coef;
                p2[jj][j] = adam_update_h64(p2[jj][j], gv, m2[jj][j], v2[jj][j], f.beta1, f.beta2, inv_bc2, f.eps, f.lr_step);
                if (keep) {
                    const int e = base + w2_off + (kr + 16 * jj) * 64 + nr + 16 * j;
                    f.p_nxt[e] = p2[jj][j], f.m_nxt[e] = m2[jj][j], f.v_nxt[e] = v2[jj][j];
                }
            }
#pragma unroll
        for (int jj = 0; jj < 4; jj++)  // forward image: row k = kr + 16 jj, slot [nr][j = 0..3]
            *reinterpret_cast<float4 *>(wimg + IMG_W2F + (kr + 16 * jj) * 64 + nr * 4) = float4{p2[jj][0], p2[jj][1], p2[jj][2], p2[jj][3]};
#pragma unroll
        for (int j = 0; j < 4; j++)  // input-gradient image: row n = nr + 16 j, slot [kr][jj = 0..3]
            *reinterpret_cast<float4 *>(wimg + IMG_W2B + (nr + 16 * j) * 64 + kr * 4) = float4{p2[0][j], p2[1][j], p2[2][j], p2[3][j]};
    }
    if (o_tid >= 0) {
#pragma unroll
        for (int i = 0; i < NO; i++) {
            const int y = o_tid + i * o_thr, x = y < w2_off ? y : y + 4096;
            if (y < n_other) {
                const float gv = (go[i] * f.scale) * coef;
                const float pnew = adam_update_h64(po[i], gv, mo[i], vo[i], f.beta1, f.beta2, inv_bc2, f.eps, f.lr_step);
                image_store_h64(wimg, L.D, n_out, x, pnew);
                if (keep) f.p_nxt[base + x] = pnew, f.m_nxt[base + x] = mo[i], f.v_nxt[base + x] = vo[i];
            }
        }
    }
}

template <bool IS_PI, int DT, bool DIRECT = false>
__device__ __forceinline__ void grad_h64t_body(const float *__restrict__ params, const PLayout &L, const Rollout &rb, const Minibatch &mb,
                                               const HParams &hp, const double *__restrict__ adv_part, int n_part, float *__restrict__ slab,
                                               double *__restrict__ stat_slot, float *smem, int n_blocks_net, int block_net, const AdamFold &fold) {
#ifdef TMA_H64_TICKS
    const unsigned long long kern_t0 = __builtin_amdgcn_s_memtime();
#endif
    __shared__ float adv_ms[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;  // r16: the tile's sample this lane works for (and the A operand's row); g: lane group
    (void)0;
    const int D = DT > 0 ? DT : L.D, A = L.A;
    constexpr int KS1C = DT > 0 ? (DT + 3) / 4 : 4;  // layer-1 k-steps held in registers (runtime D: up to 16 features)
    const int KS1 = (D + 3) >> 2;
    float *wimg = smem;  // this net's weight image, staged once per block
    float *slotA = smem + IMG_FLOATS + wave * T_PER_WAVE, *slotB = slotA + 16 * LDT, *dz3t = slotB + 16 * LDT, *Xt = dz3t + 256;
    const float invB = 1.0f / (float)mb.count;
    const int NOUT = IS_PI ? A : 1;
    // (the first tile's two dependent global round trips -- sample offset, then its rows -- run under the image staging)
    const int64_t n_tiles = (mb.count + 15) >> 4;
    const int64_t tile_stride = (int64_t)n_blocks_net * wpb;
    auto offset_of = [&](int64_t tl) -> int32_t {
        const int64_t j = (tl << 4) + r16;
        const int64_t jc = j < mb.count ? j : 0;
        return mb.offs ? mb.offs[jc] : (int32_t)sample_offset(mb, mb.start + jc, rb.T, rb.N);
    };
    int32_t nx_off = offset_of((int64_t)block_net * wpb + wave);
    int32_t pf_off = 0;
    float pf_x[KS1C], pf_m0 = 0.0f, pf_m1 = 0.0f;
    int32_t pf_act = 0;
    // rb.packed (tma_ppo_pack_samples): everything a sample needs sits in ONE record of RS floats {obs | log_prob, advantage, action, return} --
    // one 64-byte line per sample and net instead of the planes' five
    const int XS = (D + 3) & ~3, RS = XS + 4;
    const float *const packed = rb.packed;
    auto fetch = [&](int64_t tl) {
        pf_off = nx_off;
        nx_off = offset_of(tl + tile_stride);
        const int64_t row = pf_off;
        if (packed) {
            const float *rec = packed + row * RS;
#pragma unroll
            for (int ks = 0; ks < KS1C; ks++) {
                const int c = 4 * ks + g;
                pf_x[ks] = rec[c < XS ? c : 0];
            }
            if constexpr (IS_PI) {
                const float4 q = *reinterpret_cast<const float4 *>(rec + XS);
                pf_m0 = q.x;
                pf_m1 = q.y;
                pf_act = __float_as_int(q.z);
            } else {
                pf_m0 = rec[XS + 3];
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < KS1C; ks++) {
            const int c = 4 * ks + g;
            pf_x[ks] = rb.obs[row * D + (c < D ? c : 0)];
        }
        if constexpr (IS_PI) {
            pf_m0 = rb.log_probs[row];
            pf_m1 = rb.advantages[row];
            pf_act = static_cast<const int32_t *>(rb.actions)[row];
        } else {
            pf_m0 = rb.returns[row];
        }
    };
    fetch((int64_t)block_net * wpb + wave);
    if (fold.grad != nullptr) fold_adam_into_image<IS_PI>(fold, L, wimg, block_net);  // (uniform) the previous minibatch's optimizer step, then its weights
    else stage_copy(params + (IS_PI ? L.img_pi : L.img_vf), wimg, IMG_FLOATS);
    if (IS_PI && hp.normalize_advantage && threadIdx.x < 64) {  // fold the minibatch advantage partials (same order as adv_final_kernel)
        double a = 0.0, bsum = 0.0;
        for (int k = threadIdx.x; k < n_part; k += 64) a += adv_part[2 * k], bsum += adv_part[2 * k + 1];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            bsum += __shfl_down(bsum, o, 64);
        }
        if (threadIdx.x == 0) {
            const double n = (double)mb.stats_n, mean = a / n;
            double var = n > 1.0 ? (bsum - n * mean * mean) / (n - 1.0) : 0.0;
            if (var < 0.0) var = 0.0;
            adv_ms[0] = (float)mean;
            adv_ms[1] = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float amean = (IS_PI && hp.normalize_advantage) ? adv_ms[0] : 0.0f;
    const float astd = (IS_PI && hp.normalize_advantage) ? adv_ms[1] : 1.0f;
    NetAcc acc;
    zero_acc(acc);
    TileStats st;
    TileTicks tk;
    // ---- gather, one tile ahead, straight into registers (offset_of / fetch above): the sample's buffer offset is fetched one stage
    // earlier still, so the dependent hop (offset -> rows) never waits.  Loads are issued from clamped addresses and masked when the tile
    // is consumed (a select on a value just loaded would make the compiler wait for it at the issue point).
    // the two waves a SIMD hosts (w and w + 4) run the same program: started together they tend to stay in lockstep -- both in their
    // MFMA-dense phases, then both in their VALU phases -- so the second half of the block starts a fraction of a tile later
    if (wave >= 4)
        for (int q = 0; q < hp.debug; q++) __builtin_amdgcn_s_sleep(100);
#ifdef TMA_H64_TICKS
    const bool tick_on = block_net == 0 && wave == 0;
    tk.on = tick_on, tk.prev = __builtin_amdgcn_s_memtime();
    const unsigned long long loop_t0 = tk.prev, loop_r0 = __builtin_amdgcn_s_memrealtime();
    if (tick_on && lane == 0) g_h64_ticks[IS_PI ? 0 : 1][12] += loop_t0 - kern_t0;
#endif
    for (int64_t tile = (int64_t)block_net * wpb + wave; tile < n_tiles; tile += tile_stride) {
        H64_TICK(15);
        // ---- commit the prefetched tile ----
        const bool valid = ((tile << 4) + r16) < mb.count;
        float xb[KS1C];
#pragma unroll
        for (int ks = 0; ks < KS1C; ks++) xb[ks] = (valid && 4 * ks + g < D) ? pf_x[ks] : 0.0f;
        const float m0 = pf_m0, m1 = pf_m1;
        const int act = pf_act;
        fetch(tile + tile_stride);
        h64t_tile<IS_PI, KS1C>(wimg, slotA, slotB, dz3t, Xt, xb, m0, m1, act, valid, KS1, A, invB, amean, astd, hp, acc, st, tk, lane);
    }
#ifdef TMA_H64_TICKS
    const unsigned long long loop_t1 = __builtin_amdgcn_s_memtime();
    if (tick_on && lane == 0) {
        g_h64_ticks[IS_PI ? 0 : 1][10] += loop_t1 - loop_t0;
        g_h64_ticks[IS_PI ? 0 : 1][11] += __builtin_amdgcn_s_memrealtime() - loop_r0;
    }
#endif
    __syncthreads();
#ifdef TMA_H64_TICKS
    if (tick_on && lane == 0) g_h64_ticks[IS_PI ? 0 : 1][14] += __builtin_amdgcn_s_memtime() - loop_t1;  // waiting for the block's slowest wave
#endif
    // loss statistics first (their LDS scratch sits behind the staging area; the barriers inside flush_all_t publish it), the slab
    // stores last: nothing waits behind a global store
    double *red = reinterpret_cast<double *>(smem + (DIRECT ? 4 : 8) * FL_HALF * 64);  // statistics scratch behind the staging area
    double stv[5] = {st.a, st.ent, st.kl, (double)st.clip, (double)st.n};
#pragma unroll
    for (int q = 0; q < 5; q++)
        for (int o = 32; o > 0; o >>= 1) stv[q] += __shfl_down(stv[q], o, 64);
    if (lane == 0)
        for (int q = 0; q < 5; q++) red[wave * 5 + q] = stv[q];
    if constexpr (DIRECT) {
        flush_all_t<IS_PI, KS1C, 4>(smem, wave, lane, acc, L, D, NOUT, slab);  // 4-wave blocks of the small-minibatch kernel
    } else {
        flush_all_t<IS_PI, KS1C>(smem, wave, lane, acc, L, D, NOUT, slab);  // (8 waves per block: tma_launch_grad_h64)
    }
#ifdef TMA_H64_TICKS
    if (tick_on && lane == 0) g_h64_ticks[IS_PI ? 0 : 1][13] += __builtin_amdgcn_s_memtime() - loop_t1;
#endif
    if (threadIdx.x < 5) {
        double s = 0.0;
        for (int w = 0; w < wpb; w++) s += red[w * 5 + threadIdx.x];
        // slot layout {policy_loss, value_sq_err, entropy, approx_kl, clipped, n}
        const int q = IS_PI ? (threadIdx.x == 0 ? 0 : threadIdx.x + 1) : (threadIdx.x == 0 ? 1 : -1);
        if (q >= 0) stat_slot[q] += s;
    }
}

template <int DT, int VER>
__global__ __launch_bounds__(512, 2) void ppo_grad_h64_kernel(const float *__restrict__ params, PLayout L, Rollout rb, Minibatch mb, HParams hp,
                                                              const double *__restrict__ adv_part, int n_part, float *__restrict__ slabs,
                                                              double *__restrict__ stat_slots, AdamFold fold) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // policy blocks first, value blocks behind them: with 128 pairs the two blocks of pair p (same tiles, same sample records) are workgroups p and
    // 128 + p -- dispatched round-robin over the 8 XCDs they land on the SAME XCD, so the second reader of a record finds its line in that L2
    const int n_pairs = gridDim.x >> 1, pair = (int)blockIdx.x % n_pairs;
    const bool pi_block = (int)blockIdx.x < n_pairs;
    float *slab = slabs + (int64_t)pair * L.P;
    double *slot = stat_slots + (int64_t)pair * 8;
    if constexpr (VER == 1) {
        if (pi_block) grad_h64_body<true, DT>(params, L, rb, mb, hp, adv_part, n_part, slab, slot, smem, n_pairs, pair);
        else grad_h64_body<false, DT>(params, L, rb, mb, hp, adv_part, n_part, slab, slot, smem, n_pairs, pair);
    } else {
        if (pi_block) grad_h64t_body<true, DT>(params, L, rb, mb, hp, adv_part, n_part, slab, slot, smem, n_pairs, pair, fold);
        else grad_h64t_body<false, DT>(params, L, rb, mb, hp, adv_part, n_part, slab, slot, smem, n_pairs, pair, fold);
    }
}

// small minibatches (<= 128 tiles): 4-wave blocks, one wave per SIMD (no partner on the matrix pipe, 512 registers), one tile per wave
template <int DT>
__global__ __launch_bounds__(256, 1) void ppo_grad_h64_small_kernel(const float *__restrict__ params, PLayout L, Rollout rb, Minibatch mb, HParams hp,
                                                                    const double *__restrict__ adv_part, int n_part, float *__restrict__ slabs,
                                                                    double *__restrict__ stat_slots, AdamFold fold) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pair = blockIdx.x >> 1, n_pairs = gridDim.x >> 1;
    float *slab = slabs + (int64_t)pair * L.P;
    double *slot = stat_slots + (int64_t)pair * 8;
    if ((blockIdx.x & 1) == 0) grad_h64t_body<true, DT, true>(params, L, rb, mb, hp, adv_part, n_part, slab, slot, smem, n_pairs, pair, fold);
    else grad_h64t_body<false, DT, true>(params, L, rb, mb, hp, adv_part, n_part, slab, slot, smem, n_pairs, pair, fold);
}

}  // namespace tma

using namespace tma;

static int grad_h64_smem_bytes(const PLayout &L, int wpb, int ver) {
    const int ldx = ((L.D + 3) & ~3) + 2, ld = L.H + 2;
    const int per_wave = ver == 1 ? 16 * (ldx + 2 * ld + 34) + 16 * 8 : T_PER_WAVE;
    const int tile = (IMG_FLOATS + wpb * per_wave) * 4;
    const int flush = wpb * (L.H * L.H + L.H) * 4;
    return tile > flush ? tile : flush;
}

// H = 64 persistent gradient kernel over one minibatch (>= 256 samples): 2 x n_pairs blocks of 8 waves, block pair p writes slab p.
// Returns the number of slabs written through *n_slabs_out (the caller runs slab_reduce_kernel over them).
int tma_launch_grad_h64(const float *params, const PLayout &L, const Rollout &R, const Minibatch &M, const HParams &hpar, const double *adv_part,
                        int n_part, float *slabs, double *slots, int *n_slabs_out, hipStream_t s, const AdamFold *foldp) {
    AdamFold fold{};
    if (foldp) fold = *foldp;
    static const int ver = getenv("TMA_H64_V1") ? 1 : 2;  // (development switch: the round-1 LDS-round-trip tile chain)
    if (ver == 1 && foldp && foldp->grad) return TMA_ERR_INVALID;  // the round-1 chain has no optimizer prologue: refuse instead of dropping the step
    static const int stagger = getenv("TMA_H64_STAGGER") ? atoi(getenv("TMA_H64_STAGGER")) : 0;
    HParams hps = hpar;
    hps.debug = stagger;
    const int64_t tiles = ceil_div(M.count, 16);
    if (ver == 2 && tiles <= H64_BLOCKS) {  // up to 2048 samples: one tile per wave, 4-wave blocks (a 256-sample minibatch: 4 + 4 blocks, 4 slabs)
        const int64_t blocks = ceil_div(tiles, 4);
        const int smem = (IMG_FLOATS + 4 * T_PER_WAVE) * 4;
        auto launch_small = [&](auto k) -> int {
            TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            k<<<dim3((unsigned)(2 * blocks)), dim3(256), smem, s>>>(params, L, R, M, hps, adv_part, n_part, slabs, slots, fold);
            return TMA_OK;
        };
        const int rc = L.D == 4 ? launch_small(ppo_grad_h64_small_kernel<4>) : (L.D == 6 ? launch_small(ppo_grad_h64_small_kernel<6>) : launch_small(ppo_grad_h64_small_kernel<0>));
        if (rc) return rc;
        TMA_LAUNCH_CHECK();
        *n_slabs_out = (int)blocks;
        return TMA_OK;
    }
    const int wpb4 = 8, smem4 = grad_h64_smem_bytes(L, wpb4, ver);
    int64_t blocks4 = ceil_div(tiles, wpb4);
    if (blocks4 > H64_BLOCKS) blocks4 = H64_BLOCKS;
    auto launch = [&](auto k) -> int {
        if (smem4 > 64 * 1024) TMA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem4));
        k<<<dim3((unsigned)(2 * blocks4)), dim3(64 * wpb4), smem4, s>>>(params, L, R, M, hps, adv_part, n_part, slabs, slots, fold);
        return TMA_OK;
    };
    int rc;
    if (ver == 1) rc = L.D == 4 ? launch(ppo_grad_h64_kernel<4, 1>) : (L.D == 6 ? launch(ppo_grad_h64_kernel<6, 1>) : launch(ppo_grad_h64_kernel<0, 1>));
    else rc = L.D == 4 ? launch(ppo_grad_h64_kernel<4, 2>) : (L.D == 6 ? launch(ppo_grad_h64_kernel<6, 2>) : launch(ppo_grad_h64_kernel<0, 2>));
    if (rc) return rc;
    TMA_LAUNCH_CHECK();
    *n_slabs_out = (int)blocks4;
    return TMA_OK;
}

#ifdef TMA_H64_TICKS
extern "C" int tma_debug_h64_ticks(unsigned long long *out32, int reset) {
    TMA_HIP(hipDeviceSynchronize());
    if (out32) TMA_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_h64_ticks), sizeof(unsigned long long) * 32));
    if (reset) {
        unsigned long long z[32] = {};
        TMA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_h64_ticks), z, sizeof(z)));
    }
    return TMA_OK;
}
#endif

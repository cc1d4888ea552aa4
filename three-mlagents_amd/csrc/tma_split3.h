// tma_split3.h -- the 256-wide PPO gradient on the bf16 MFMA at f32-class accuracy: every operand as THREE bf16 terms (mfma_dtype = 2, opt-in).
//
// Included inside namespace tma after tma_wide_bf16.h (whose LDS image layouts, fragment loaders and epilogue helpers it reuses) by tma_bf16.hip.
//
// The reference's default policy is MLP(256, 256) in fp32 (backend/mlagents/training.py:363-365).  The exact path (ppo_grad_wide_kernel,
// v_mfma_f32_16x16x4_f32) is bound by the f32 matrix pipe: 157 TFLOP/s dense, 1/16 of the bf16 pipe.  tools/bf16x3_probe.hip measured the
// alternative on the H x H phase: x = hi + mid + lo with each term the bf16 rounding of what the previous ones left (24 significant bits in
// three 8-bit pieces: the split of an f32 is exact), and
//     a . w ~= a_lo.w_hi + a_mid.w_mid + a_hi.w_lo + a_mid.w_hi + a_hi.w_mid + a_hi.w_hi        (f32 accumulate, smallest products first)
// on v_mfma_f32_16x16x32_bf16: 1.82x the exact-f32 phase, largest error against float64 2^-21.1 of the largest output -- the exact-f32 MFMA's
// own 2^-20.7 (set by its accumulation order, not by its products).  The three dropped products are of order 2^-24 and below.
//
// Structure: the bf16 kernel's 32-row-group path (transposed LDS images + ds_read_b64_tr_b16, four waves of 64 columns, dW2 / dW1 / dW3 slices
// in MFMA accumulators for the whole launch, H x H weight streams through a register ring) with every image in three PLANES and the six
// products laid out as EXTRA K: a GEMM over k becomes one over (term, k) -- the A fragment of (t, k) is plane TA[t] at k, the B fragment plane
// TB[t] at k -- so the operand pipeline (ring, half-step-ahead fragments, barriers) carries six times the k-steps and is otherwise unchanged,
// and no more registers are live than with one plane.  The weight-gradient GEMMs (reduction over the 32 samples of the group: one k-step)
// run term-outermost for the same reason.  Epilogues form tanh / (1 - h^2) in f32 on the exact values (the three planes of an activation
// add up to the f32 it was split from) and split the result again.  Discrete heads, observations of up to 32 floats, H = 256.
// Rollouts and evaluation use the exact-f32 forward kernels: log-probabilities of a rollout and of the first update epoch agree to ~2^-21
// relative (approx_kl of the first minibatch ~1e-12 instead of exactly 0) -- documented, opt-in.
#pragma once

constexpr int S3_NT = 6;
// Weight gradients (dW = operand^T . delta, summed over the samples of the minibatch) take the three products of order <= 1 -- the last three of
// the list: a_mid.b_hi, a_hi.b_mid, a_hi.b_hi.  A product is then good to ~2^-16 of its size, an error that is not passed on (nothing is computed
// FROM a weight gradient inside the launch) and averages out over the 131 072 summands; the forward / backward chain keeps all six.
constexpr int S3_WG0 = 3;
__host__ __device__ constexpr int s3_ta(int t) { return t == 0 ? 2 : ((t == 1 || t == 3) ? 1 : 0); }  // plane of the first operand:  {2, 1, 0, 1, 0, 0}
__host__ __device__ constexpr int s3_tb(int t) { return t == 2 ? 2 : ((t == 1 || t == 4) ? 1 : 0); }  // plane of the second operand: {0, 1, 2, 0, 1, 0}

// the same tables for a loop over the terms that is NOT unrolled (two bits per term)
__device__ __forceinline__ int s3_ta_rt(int t) { return (0x46 >> (2 * t)) & 3; }
__device__ __forceinline__ int s3_tb_rt(int t) { return (0x124 >> (2 * t)) & 3; }

struct bfq3 {
    bfq p[3];
};
// x (four f32) -> three packed-bf16 quads with p0 + p1 + p2 == x exactly (each residual is exact: it has at most 16, then 8 significant bits)
__device__ __forceinline__ bfq3 split_quad(float x0, float x1, float x2, float x3) {
    bfq3 o;
    o.p[0] = bfq{bf_pack2(x0, x1), bf_pack2(x2, x3)};
    float r0 = x0 - bf_lo(o.p[0].lo), r1 = x1 - bf_hi(o.p[0].lo), r2 = x2 - bf_lo(o.p[0].hi), r3 = x3 - bf_hi(o.p[0].hi);
    o.p[1] = bfq{bf_pack2(r0, r1), bf_pack2(r2, r3)};
    r0 -= bf_lo(o.p[1].lo), r1 -= bf_hi(o.p[1].lo), r2 -= bf_lo(o.p[1].hi), r3 -= bf_hi(o.p[1].hi);
    o.p[2] = bfq{bf_pack2(r0, r1), bf_pack2(r2, r3)};
    return o;
}
__device__ __forceinline__ void split1(float x, bf16_t &h, bf16_t &m, bf16_t &l) {
    h = (bf16_t)x;
    const float r1 = x - (float)h;
    m = (bf16_t)r1;
    l = (bf16_t)(r1 - (float)m);
}
// the f32 values three stored quads add up to
__device__ __forceinline__ void unsplit_quad(const uint2 a, const uint2 b, const uint2 c, float (&x)[4]) {
    x[0] = (bf_lo(a.x) + bf_lo(b.x)) + bf_lo(c.x), x[1] = (bf_hi(a.x) + bf_hi(b.x)) + bf_hi(c.x);
    x[2] = (bf_lo(a.y) + bf_lo(b.y)) + bf_lo(c.y), x[3] = (bf_hi(a.y) + bf_hi(b.y)) + bf_hi(c.y);
}
// transposed-read A fragment at a compile-time element offset from the phase's base pointers (plane + k-step)
template <int MT>
__device__ __forceinline__ bf16x8 a_frag_tb_off(const TrBase<MT> &b, int mt, int off) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef s16x4 __attribute__((address_space(3))) *lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(b.lo[mt] + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(b.hi[mt] + off));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__host__ __device__ inline int grad_split3_smem_bytes() {
    constexpr int M = 32, H = 256, Kp1 = 32, ldx = 48, ldz = 48, ld3 = 34;
    const int bf = 3 * (M * ldx + Kp1 * M + 2 * H * M + M * ldz + 32 * M);
    return bf * 2 + (M * ld3 + M * 4 + 128 + 4 * 2 * 2 * 256 + 2 * H + 32) * 4 + 8 * 4 * 5 * 8 + 2 * M * 8;
}

// rebuilds the three-plane fragment-major images of both nets from the f32 master weights (the layouts of build_bf16_images_kernel, per plane)
static __global__ void build_split3_images_kernel(float *params, PLayout L) {
    const int D = L.D, H = L.H;
    const int Kp1 = (D + 31) & ~31, KS1 = Kp1 / 32, KS2 = H / 32;
    for (int net = 0; net < 2; net++) {
        const int n_out = net == 0 ? L.A : 1;
        const BfNet B = bf_net_layout(D, H, n_out);
        bf16_t *img = reinterpret_cast<bf16_t *>(params + (net == 0 ? L.sp_pi : L.sp_vf));
        const float *W1 = params + (net == 0 ? L.pW1t : L.vW1t), *W2 = params + (net == 0 ? L.pW2t : L.vW2t), *W3 = params + (net == 0 ? L.pW3t : L.vW3t);
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B.size; e += gridDim.x * blockDim.x) {
            float v;
            if (e < B.fW2) {
                const int x = e - B.fW1, j = x & 7, l = (x >> 3) & 63, rest = x >> 9, ks = rest % KS1, nt = rest / KS1;
                const int k = 32 * ks + 8 * (l >> 4) + j, n = 16 * nt + (l & 15);
                v = k < D ? W1[k * H + n] : 0.0f;
            } else if (e < B.bW2) {
                const int x = e - B.fW2, j = x & 7, l = (x >> 3) & 63, rest = x >> 9, ks = rest % KS2, nt = rest / KS2;
                v = W2[(32 * ks + 8 * (l >> 4) + j) * H + 16 * nt + (l & 15)];
            } else if (e < B.fW3) {
                const int x = e - B.bW2, j = x & 7, l = (x >> 3) & 63, rest = x >> 9, ns = rest % KS2, kt = rest / KS2;
                v = W2[(16 * kt + (l & 15)) * H + 32 * ns + 8 * (l >> 4) + j];
            } else if (e < B.bW3) {
                const int x = e - B.fW3, j = x & 7, l = (x >> 3) & 63, rest = x >> 9, ks = rest % KS2, at = rest / KS2;
                const int k = 32 * ks + 8 * (l >> 4) + j, a = 16 * at + (l & 15);
                v = a < n_out ? W3[k * n_out + a] : 0.0f;
            } else {
                const int x = e - B.bW3, j = x & 7, l = (x >> 3) & 63, kt = x >> 9;
                const int a = 8 * (l >> 4) + j;
                v = a < n_out ? W3[(16 * kt + (l & 15)) * n_out + a] : 0.0f;
            }
            bf16_t h, m, lo;
            split1(v, h, m, lo);
            img[e] = h, img[B.size + e] = m, img[2 * B.size + e] = lo;
        }
    }
}

// Gradient of one minibatch for ONE net, persistent over 32-row groups.  NW waves, wave w owns columns [16 NTW w, 16 NTW (w + 1)) of both hidden
// layers (NW = 4: one wave per SIMD, 64 columns; NW = 8: two per SIMD, 32 columns -- one wave's epilogues and waits run beside its partner's MFMAs).
template <bool IS_PI, int KT1C, int NW>
__device__ __forceinline__ void grad_split3_body(const float *__restrict__ params, const PLayout &L, const Rollout &rb, const Minibatch &mb, const HParams &hp,
                                                 float *__restrict__ slab, double *__restrict__ stat_slot, char *smem, int n_blocks_net, int block_net) {
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    constexpr int NTW = 16 / NW, MT = 2, M = 32, H = 256, KS2 = 8, KT2 = 16, ldz = 48, ld3 = 34, Kp1 = 32, ldx = 48, NT = S3_NT, HK = 2;
    constexpr bool W8 = NW == 8;
    constexpr int XA_PS = M * ldx, XT_PS = Kp1 * M, T_PS = H * M, ZA_PS = M * ldz, ZT_PS = 32 * M, KSTRIDE = 32 * 16 * MT;
    constexpr int R = KS2 / 2 * NTW;  // ring slots: half a weight plane
    constexpr int CG = KT1C == 1 ? 2 : 4, NX = CG;                 // observation gather: 256 threads cover the 32 rows in 8-column groups (tma_wide_bf16.h, P0)
    const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = lane0, r16 = lane0 & 15, g = lane0 >> 4;
#define S3_RELANE()                        \
    do {                                   \
        lane = lane0;                      \
        asm volatile("" : "+v"(lane));     \
        r16 = lane & 15, g = lane >> 4;    \
    } while (0)
#ifdef TMA_S3_TICKS  // diagnostic build (tools/s3_ticks.py): cycles per phase, wave 0 of block 0 of each net
    __shared__ unsigned long long tick_lds[16];
    if (threadIdx.x < 16) tick_lds[threadIdx.x] = 0;
    long long tlast = clock64();
    const bool tick_on = threadIdx.x == 0 && block_net == 0;
#define S3_TICK(i)                                                     \
    do {                                                               \
        const long long tn = clock64();                                \
        if (tick_on) tick_lds[i] += (unsigned long long)(tn - tlast);  \
        tlast = clock64();                                             \
    } while (0)
#else
#define S3_TICK(i)
#endif
    const int D = L.D, A = L.A, NOUT = IS_PI ? A : 1;
    bf16_t *Xa = reinterpret_cast<bf16_t *>(smem), *Xt = Xa + 3 * XA_PS, *T1 = Xt + 3 * XT_PS, *T2 = T1 + 3 * T_PS, *Z3a = T2 + 3 * T_PS, *Z3t = Z3a + 3 * ZA_PS;
    float *dz3 = reinterpret_cast<float *>(Z3t + 3 * ZT_PS), *meta = dz3 + M * ld3, *scratch = meta + M * 4, *hpart = scratch + 128, *bias = hpart + 4 * MT * 2 * 256;
    double *stat_lds = reinterpret_cast<double *>(bias + 2 * H + 32);
    int64_t *row_off = reinterpret_cast<int64_t *>(stat_lds + NW * 4 * 5), *row_off_next = row_off + M;
    const int n_base = wave * 16 * NTW, nt0 = wave * NTW;
    const float invB = 1.0f / (float)mb.count;
    for (int e = threadIdx.x; e < 3 * (XA_PS + XT_PS) / 8; e += blockDim.x) reinterpret_cast<uint4 *>(Xa)[e] = uint4{0u, 0u, 0u, 0u};  // padding columns stay zero
    __shared__ float adv_ms3[2];
    if (IS_PI && hp.normalize_advantage && threadIdx.x < 64) {  // minibatch advantage statistics from the partials (adv_final_kernel's order)
        double a = 0.0, bsum = 0.0;
        for (int k = threadIdx.x; k < mb.adv_n_part; k += 64) a += mb.adv_part[2 * k], bsum += mb.adv_part[2 * k + 1];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            bsum += __shfl_down(bsum, o, 64);
        }
        if (threadIdx.x == 0) {
            const double n = (double)mb.stats_n, mean = a / n;
            double var = n > 1.0 ? (bsum - n * mean * mean) / (n - 1.0) : 0.0;
            if (var < 0.0) var = 0.0;
            adv_ms3[0] = (float)mean;
            adv_ms3[1] = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float amean = (IS_PI && hp.normalize_advantage) ? adv_ms3[0] : 0.0f;
    const float astd = (IS_PI && hp.normalize_advantage) ? adv_ms3[1] : 1.0f;
    const Net Q = IS_PI ? pi_net(params, L) : vf_net(params, L);
    const BfNet B = bf_net_layout(D, H, NOUT);
    int wps = B.size;  // bf16 elements per weight plane
    const bf16_t *img = reinterpret_cast<const bf16_t *>(params + (IS_PI ? L.sp_pi : L.sp_vf));
    BfNetPtr W{img + B.fW1, img + B.fW2, img + B.bW2, img + B.fW3, img + B.bW3};
    const f32x4 z4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 aW2[KT2][NTW], aW1[KT1C][NTW], aW3[NTW];
    float sB1[NTW], sB2[NTW];  // hidden-layer bias gradients: per-lane sums of the f32 deltas (this lane's rows of the column), folded over the lane groups at the end
    float ab3 = 0.0f, dlsd[2] = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        sB1[j] = sB2[j] = 0.0f;
        aW3[j] = z4;
#pragma unroll
        for (int i = 0; i < KT1C; i++) aW1[i][j] = z4;
#pragma unroll
        for (int i = 0; i < KT2; i++) aW2[i][j] = z4;
    }
    for (int e = threadIdx.x; e < 2 * H + 32; e += blockDim.x)
        bias[e] = e < H ? Q.b1[e] : (e < 2 * H ? Q.b2[e - H] : (e - 2 * H < NOUT ? Q.b3[e - 2 * H] : 0.0f));
    if (threadIdx.x < NW * 4 * 5) stat_lds[threadIdx.x] = 0.0;
    int nt0l = nt0;
    // One H x H GEMM of the chain (layer-2 forward: activations T1, images fW2; input gradient: deltas T2, images bW2).  The six products are
    // grouped by WEIGHT plane -- w_lo with a_hi; w_mid with a_mid, a_hi; w_hi with a_lo, a_mid, a_hi -- so a weight fragment is streamed ONCE and
    // meets up to three activation planes while it is in registers.  (Product-major, every product pulling its own copy of the fragment, the
    // launch was bound by the weight stream: 6 x 128 KB per GEMM and group from L2, 13 TB/s chip-wide at 950 us.)
    // Two passes of 8 k-steps, balanced so that the R fragments the ring holds ahead are the same amount of MFMA work in both (the L2 round
    // trip hides behind it): pass A takes w_hi AND w_lo of a k-step (4 products per tile = 32 MFMAs at NTW = 4; 2 NTW fragments per k-step: the
    // ring is 2 k-steps deep), pass B takes w_mid (2 products = 16 MFMAs; NTW fragments per k-step: 4 k-steps deep).  With one weight plane
    // per pass the w_lo pass had 8 MFMAs per k-step -- 512 cycles of cover for its loads, less than the round trip.
    // Every slot, fragment index and LDS offset is a compile-time constant; a consumed slot is refilled at once with the fragment R positions on.
    bf16x8 ring[R];
#pragma unroll
    for (int s = 0; s < R; s++) {  // pass A of the forward stream, k-steps 0 and 1: slot = (ks % 2) * 2 NTW + plane * NTW + j (plane 0: w_hi, 1: w_lo)
        const int ks = s / (2 * NTW), pl = (s / NTW) & 1, jj = s % NTW;
        ring[s] = bf_frag(W.fW2 + (pl ? 2 : 0) * (int64_t)wps, (nt0l + jj) * KS2 + ks, lane);
    }
    auto hh_gemm = [&](const bf16_t *Timg, const bf16_t *wthis, const bf16_t *wother, f32x4 (&acc)[NTW][MT]) {
        const TrBase<MT> tb0 = tr_base<MT>(Timg, lane);
        const bf16_t *whi = wthis, *wmid = wthis + (int64_t)wps, *wlo = wthis + 2 * (int64_t)wps;
        constexpr int AB = W8 ? 1 : 2;  // one wave per SIMD: activation fragments a k-step ahead (nothing else hides the LDS round trip)
        {  // ---- pass A: w_hi x (a_lo, a_mid, a_hi) and w_lo x a_hi ----
            bf16x8 abuf[AB][3][MT];
            auto aload = [&](int slot, int ks) {
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) abuf[slot][p][mt] = a_frag_tb_off<MT>(tb0, mt, p * T_PS + ks * KSTRIDE);
            };
            if constexpr (!W8) aload(0, 0);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                const int cur = W8 ? 0 : (ks & 1), sb = (ks & 1) * 2 * NTW;
                if constexpr (W8) aload(0, ks);
                else if (ks + 1 < KS2) aload(cur ^ 1, ks + 1);
                auto &a = abuf[cur];
                // (smallest products first; an accumulator is written once per sweep over the NTW x MT tiles)
#pragma unroll
                for (int j = 0; j < NTW; j++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) acc[j][mt] = mfma_bf(a[0][mt], ring[sb + NTW + j], acc[j][mt]);  // a_hi . w_lo
#pragma unroll
                for (int p = 2; p >= 0; p--)
#pragma unroll
                    for (int j = 0; j < NTW; j++)
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) acc[j][mt] = mfma_bf(a[p][mt], ring[sb + j], acc[j][mt]);  // a_lo, a_mid, a_hi . w_hi
#pragma unroll
                for (int j = 0; j < NTW; j++) {
                    if (ks + 2 < KS2) {  // two k-steps on in this pass
                        ring[sb + j] = bf_frag(whi, (nt0l + j) * KS2 + ks + 2, lane);
                        ring[sb + NTW + j] = bf_frag(wlo, (nt0l + j) * KS2 + ks + 2, lane);
                    } else {  // the first four k-steps of pass B: its slot = (ks % 4) * NTW + j, i.e. k-steps 0, 1 here at ks = 6 and 2, 3 at ks = 7
                        ring[sb + j] = bf_frag(wmid, (nt0l + j) * KS2 + 2 * (ks - 6), lane);
                        ring[sb + NTW + j] = bf_frag(wmid, (nt0l + j) * KS2 + 2 * (ks - 6) + 1, lane);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {  // ---- pass B: w_mid x (a_mid, a_hi) ----
            bf16x8 abuf[AB][2][MT];
            auto aload = [&](int slot, int ks) {
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) abuf[slot][p][mt] = a_frag_tb_off<MT>(tb0, mt, p * T_PS + ks * KSTRIDE);
            };
            if constexpr (!W8) aload(0, 0);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                const int cur = W8 ? 0 : (ks & 1), sb = (ks & 3) * NTW;
                if constexpr (W8) aload(0, ks);
                else if (ks + 1 < KS2) aload(cur ^ 1, ks + 1);
                auto &a = abuf[cur];
#pragma unroll
                for (int p = 1; p >= 0; p--)
#pragma unroll
                    for (int j = 0; j < NTW; j++)
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) acc[j][mt] = mfma_bf(a[p][mt], ring[sb + j], acc[j][mt]);
#pragma unroll
                for (int j = 0; j < NTW; j++) {
                    if (ks + 4 < KS2) {
                        ring[sb + j] = bf_frag(wmid, (nt0l + j) * KS2 + ks + 4, lane);
                    } else {  // pass A of the OTHER stream (the phase that follows): k-step (ks - 4) / 2, plane (ks - 4) % 2 -> slot (ks' % 2) * 2 NTW + plane * NTW + j == sb + j
                        const int ks2 = (ks - 4) >> 1, pl = (ks - 4) & 1;
                        ring[sb + j] = bf_frag(wother + (pl ? 2 : 0) * (int64_t)wps, (nt0l + j) * KS2 + ks2, lane);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    static_assert(R == 4 * NTW && KS2 == 8, "the ring: two k-steps of pass A, four of pass B");
    // ---- prefetch registers for the next group's samples (the scheme of grad_wide_bf_body) ----
    float pm0 = 0.0f, pm1 = 0.0f, pm2 = 0.0f, pm3 = 0.0f, px[NX];
    int64_t poff = -1;
    const int mrow = wave * (M / NW) + lane0;
    const bool mlane = lane0 < M / NW;
    int32_t noff = -1;
    auto fetch_off = [&](int64_t grp) {
        if (mb.offs && mlane) {
            const int64_t j = grp * M + mrow;
            noff = mb.offs[j < mb.count ? j : 0];
        }
    };
    auto fetch_meta = [&](int64_t grp, bool have_noff) {
        if (mlane) {
            const int64_t j = grp * M + mrow;
            poff = -1, pm0 = pm1 = pm2 = pm3 = 0.0f;
            if (j < mb.count) {
                poff = mb.offs ? (int64_t)(have_noff ? noff : mb.offs[j]) : sample_offset(mb, mb.start + j, rb.T, rb.N);
                pm0 = rb.log_probs[poff], pm1 = rb.advantages[poff], pm2 = rb.returns[poff];
                pm3 = __int_as_float(static_cast<const int32_t *>(rb.actions)[poff]);
            }
            row_off_next[mrow] = poff;
        }
    };
    auto fetch_obs = [&]() {  // (the first 256 threads cover the 32 rows in 8-column groups)
        const int tid = (wave & 3) * 64 + lane;
        const int64_t off = row_off_next[tid >> 3];
        const int base = (int)off * D;
#pragma unroll
        for (int cg = 0; cg < CG; cg++) {
            const int c = 8 * cg + (tid & 7);
            px[cg] = rb.obs[(off >= 0 && c < D) ? base + c : 0];
        }
    };
    const int64_t n_groups = (mb.count + M - 1) / M;
    if (block_net < n_groups) {
        fetch_meta(block_net, false);
        __syncthreads();
        fetch_obs();
    }
    for (int64_t grp = block_net; grp < n_groups; grp += n_blocks_net) {
        W.fW1 = launder_uniform(W.fW1), W.fW2 = launder_uniform(W.fW2), W.bW2 = launder_uniform(W.bW2), W.fW3 = launder_uniform(W.fW3);
        W.bW3 = launder_uniform(W.bW3), nt0l = launder_uniform(nt0l), wps = launder_uniform(wps);
        S3_TICK(0);
        S3_RELANE();
        // layer-1 fragments of the three planes: in flight under the commit
        bf16x8 w1[3][NTW];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int j = 0; j < NTW; j++) w1[p][j] = bf_frag(W.fW1 + (int64_t)p * wps, nt0l + j, lane);
        // ---- P0: commit the prefetched metadata / observation rows (three planes, both images) ----
        if (mlane) {
            meta[mrow * 4 + 0] = pm0, meta[mrow * 4 + 1] = pm1, meta[mrow * 4 + 2] = pm2, meta[mrow * 4 + 3] = pm3;
            row_off[mrow] = poff;
        }
        if (wave < 4) {
            const int tid = wave * 64 + lane, row = tid >> 3;
            const bool okr = row_off_next[row] >= 0;
#pragma unroll
            for (int cg = 0; cg < CG; cg++) {
                const int c = 8 * cg + (tid & 7);
                bf16_t h, m, l;
                split1((okr && c < D) ? px[cg] : 0.0f, h, m, l);
                Xa[row * ldx + c] = h, Xa[XA_PS + row * ldx + c] = m, Xa[2 * XA_PS + row * ldx + c] = l;
                const int to = t_off<MT>(c, row);
                Xt[to] = h, Xt[XT_PS + to] = m, Xt[2 * XT_PS + to] = l;
            }
        }
        __syncthreads();
        S3_TICK(1);
        const bool has_next = grp + n_blocks_net < n_groups;  // block-uniform
        fetch_off(grp + n_blocks_net);
        S3_RELANE();
        // ---- P1: layer 1 forward (one k-step, six products) ----
        {
            f32x4 acc[NTW][MT];
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const float b = bias[n_base + 16 * j + r16];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[j][mt] = f32x4{b, b, b, b};
            }
#pragma unroll
            for (int t = 0; t < NT; t++)  // (the observation fragments are re-read per term: an LDS read is cheaper than holding three planes)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const bf16x8 xa = a_frag(Xa + s3_ta(t) * XA_PS, ldx, 16 * mt + r16, 0, g);
#pragma unroll
                    for (int j = 0; j < NTW; j++) acc[j][mt] = mfma_bf(xa, w1[s3_tb(t)][j], acc[j][mt]);
                }
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const int n = n_base + 16 * j + r16;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const bfq3 q = split_quad(tma_tanh(acc[j][mt][0]), tma_tanh(acc[j][mt][1]), tma_tanh(acc[j][mt][2]), tma_tanh(acc[j][mt][3]));
#pragma unroll
                    for (int p = 0; p < 3; p++) bfq_store_quad(t_quad<MT>(T1 + p * T_PS, n, mt, g), q.p[p]);
                }
            }
        }
        __syncthreads();
        S3_TICK(2);
        S3_RELANE();
        // ---- P2: layer 2 forward through the weight ring: 48 (term, k-step) steps ----
        bf16x8 w3f[3][HK];  // this wave's head fragments (split-K: the k-steps of its own h2 columns), in flight behind the layer-2 epilogue
        constexpr bool KEEP_H2 = false && !W8;  // (measured: +137 spilled registers, 708 against 692 us -- off) one wave per SIMD: this lane's h2 quads stay in registers until P4 forms dz2 from them (the same lane owns the
                                       // same quads there) instead of being re-read from three planes and summed: -24 LDS reads, -160 VALU per group
        float h2k[KEEP_H2 ? NTW : 1][MT][4];
        {
            f32x4 acc[NTW][MT];
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const float b = bias[H + n_base + 16 * j + r16];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[j][mt] = f32x4{b, b, b, b};
            }
            hh_gemm(T1, W.fW2, W.bW2, acc);
            S3_TICK(3);
            if (!W8 || wave < 4) {
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int i = 0; i < HK; i++) w3f[p][i] = bf_frag(W.fW3 + (int64_t)p * wps, wave * HK + i, lane);
            }
            if (has_next) fetch_meta(grp + n_blocks_net, true);
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const int n = n_base + 16 * j + r16;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const float t0 = tma_tanh(acc[j][mt][0]), t1 = tma_tanh(acc[j][mt][1]), t2 = tma_tanh(acc[j][mt][2]), t3 = tma_tanh(acc[j][mt][3]);
                    if constexpr (KEEP_H2) h2k[j][mt][0] = t0, h2k[j][mt][1] = t1, h2k[j][mt][2] = t2, h2k[j][mt][3] = t3;
                    const bfq3 q = split_quad(t0, t1, t2, t3);
#pragma unroll
                    for (int p = 0; p < 3; p++) bfq_store_quad(t_quad<MT>(T2 + p * T_PS, n, mt, g), q.p[p]);
                }
            }
        }
        S3_TICK(4);
        if constexpr (W8) __syncthreads();  // (eight waves: the k-steps of head wave w are the h2 columns of waves 2 w and 2 w + 1)
        if (!W8 || wave < 4) {  // split-K head partial over k-steps wave * HK + i (four waves: this wave's own h2 columns -- its own stores, no barrier in front)
            const TrBase<MT> tb = tr_base<MT>(T2 + wave * HK * KSTRIDE, lane);
            f32x4 part[MT][HK];  // (four independent accumulation chains instead of two of twelve dependent MFMAs each)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int i = 0; i < HK; i++) part[mt][i] = z4;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int i = 0; i < HK; i++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        part[mt][i] = mfma_bf(a_frag_tb_off<MT>(tb, mt, s3_ta(t) * T_PS + i * KSTRIDE), w3f[s3_tb(t)][i], part[mt][i]);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) *reinterpret_cast<f32x4 *>(hpart + (((wave * MT + mt) * 2 + 0) * 64 + lane) * 4) = part[mt][0] + part[mt][1];
        }
        __syncthreads();
        S3_RELANE();
        S3_TICK(5);
        bf16x8 w3b[3][NTW];  // head input-gradient fragments for P4.  One wave per SIMD: requested here, in flight behind the loss (measured: P4 opened with
                             // an exposed L2 round trip); two per SIMD: requested in P4 (48 registers would sit through the loss)
        if constexpr (!W8) {
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int j = 0; j < NTW; j++) w3b[p][j] = bf_frag(W.bW3 + (int64_t)p * wps, nt0l + j, lane);
        }
        fetch_obs();
        __builtin_amdgcn_sched_barrier(0);
        // ---- P3b: loss -- every wave takes half the rows of one tile (tile wave & 1, rows 2 (wave >> 1) .. + 1 of each lane group) ----
        {
            const int mt = wave & 1, r_lo = W8 ? (wave >> 1) : 2 * (wave >> 1), r_n = W8 ? 1 : 2;  // (eight waves: one row of every lane group each)
            f32x4 out[1];
            {
                const float b = bias[2 * H + r16];
                out[0] = f32x4{b, b, b, b};
#pragma unroll
                for (int w = 0; w < 4; w++) out[0] += *reinterpret_cast<const f32x4 *>(hpart + (((w * MT + mt) * 2 + 0) * 64 + lane) * 4);
            }
            LossStats st;
            float *dzt = dz3 + mt * 16 * ld3;
            if constexpr (IS_PI) {
#define S3_LOSS_ARGS out, meta + mt * 64, row_off + mt * 16, rb.actions, params + L.log_std, A, amean, astd, hp, invB, dzt, ld3, dlsd, st, lane
                if constexpr (W8) {
                    policy_loss_tile<false>(S3_LOSS_ARGS, r_lo, r_lo + r_n);
                } else {  // rows 0, 1 (waves 0, 1) or 2, 3 (waves 2, 3): a compile-time range per branch, so that the two rows' chains interleave
                    if (wave < 2) policy_loss_tile<false, 0, 2>(S3_LOSS_ARGS);
                    else policy_loss_tile<false, 2, 4>(S3_LOSS_ARGS);
                }
#undef S3_LOSS_ARGS
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r < r_lo || r >= r_lo + r_n) continue;
                    const int row = g * 4 + r;
                    const bool valid = row_off[mt * 16 + row] >= 0;
                    const float diff = out[0][r] - meta[(mt * 16 + row) * 4 + 2];
                    dzt[row * ld3 + r16] = (valid && r16 == 0) ? (hp.vf_coef * 2.0f * invB) * diff : 0.0f;
                    if (valid && r16 == 0) st.a += (double)(diff * diff);
                }
            }
            if (r16 == 0) {
                double *sl = stat_lds + (wave * 4 + g) * 5;
                sl[0] += st.a, sl[1] += st.ent, sl[2] += st.kl, sl[3] += st.clip, sl[4] += st.n;
            }
        }
        __syncthreads();
        S3_TICK(6);
        // ---- P3c: dz3 in three planes, both layouts: waves 0 / 1 write Z3a of tile 0 / 1, waves 2 / 3 Z3t (+ head bias sums) ----
        S3_RELANE();
        {
            const int mt = wave & 1;
            const float *dzt = dz3 + mt * 16 * ld3;
            bf16x8 vh, vm, vl;
            if (wave < 2) {  // Z3a[m][a]: lane = (row, 8-column chunk)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int a = 8 * g + j;
                    bf16_t h, m, l;
                    split1(a < 16 ? dzt[r16 * ld3 + a] : 0.0f, h, m, l);
                    vh[j] = h, vm[j] = m, vl[j] = l;
                }
                bf16_t *dst = Z3a + (16 * mt + r16) * ldz + 8 * g;
                *reinterpret_cast<bf16x8 *>(dst) = vh, *reinterpret_cast<bf16x8 *>(dst + ZA_PS) = vm, *reinterpret_cast<bf16x8 *>(dst + 2 * ZA_PS) = vl;
            } else if (wave < 4) {  // Z3t[a][m]: lane = (a, 8-sample half of the tile); the f32 column sum feeds the head bias gradient
                const int a = lane & 31, half = lane >> 5;
                float c = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float x = a < 16 ? dzt[(8 * half + j) * ld3 + a] : 0.0f;
                    c += x;
                    bf16_t h, m, l;
                    split1(x, h, m, l);
                    vh[j] = h, vm[j] = m, vl[j] = l;
                }
                ab3 += c;
                bf16_t *dst = Z3t + t_off<MT>(a, 16 * mt + 8 * half);
                *reinterpret_cast<bf16x8 *>(dst) = vh, *reinterpret_cast<bf16x8 *>(dst + ZT_PS) = vm, *reinterpret_cast<bf16x8 *>(dst + 2 * ZT_PS) = vl;
            }
        }
        __syncthreads();
        S3_TICK(7);
        S3_RELANE();
        // ---- P4: head weight gradient (this wave's k rows); dz2 = (dz3 . W3^T) * (1 - h2^2), in place in the T2 planes ----
        {
            if constexpr (W8) {
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int j = 0; j < NTW; j++) w3b[p][j] = bf_frag(W.bW3 + (int64_t)p * wps, nt0l + j, lane);
            }
#pragma unroll
            for (int t = S3_WG0; t < NT; t++) {
                const bf16x8 zb = t_frag<MT>(Z3t + s3_tb(t) * ZT_PS, r16, 0, g);
#pragma unroll
                for (int i = 0; i < NTW; i++) aW3[i] = mfma_bf(t_frag<MT>(T2 + s3_ta(t) * T_PS, n_base + 16 * i + r16, 0, g), zb, aW3[i]);
            }
            f32x4 dhz[NTW][MT];
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) dhz[j][mt] = z4;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const bf16x8 za = *reinterpret_cast<const bf16x8 *>(Z3a + s3_ta(t) * ZA_PS + (16 * mt + r16) * ldz + 8 * g);
#pragma unroll
                    for (int j = 0; j < NTW; j++) dhz[j][mt] = mfma_bf(za, w3b[s3_tb(t)][j], dhz[j][mt]);
                }
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const int n = n_base + 16 * j + r16;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 dh = dhz[j][mt];
                    bf16x4 *tq = t_quad<MT>(T2, n, mt, g);
                    float h[4], dz[4];
                    if constexpr (KEEP_H2) {
#pragma unroll
                        for (int r = 0; r < 4; r++) h[r] = h2k[j][mt][r];
                    } else {
                        unsplit_quad(*reinterpret_cast<const uint2 *>(tq), *reinterpret_cast<const uint2 *>(tq + T_PS / 4), *reinterpret_cast<const uint2 *>(tq + 2 * (T_PS / 4)), h);
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) dz[r] = dh[r] * (1.0f - h[r] * h[r]);
                    sB2[j] += (dz[0] + dz[1]) + (dz[2] + dz[3]);
                    const bfq3 q = split_quad(dz[0], dz[1], dz[2], dz[3]);
#pragma unroll
                    for (int p = 0; p < 3; p++) bfq_store_quad(tq + p * (T_PS / 4), q.p[p]);
                }
            }
        }
        __syncthreads();
        S3_TICK(8);
        S3_RELANE();
        // ---- P5: dW2 slice += h1^T . dz2[:, slice] (term-outermost: four dz2 fragments live at a time);  dh1 = dz2 . W2^T (weight ring) ----
        f32x4 dh1[NTW][MT];
#pragma unroll
        for (int j = 0; j < NTW; j++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) dh1[j][mt] = z4;
        {
            // (t_swz<2>(16 kt + r) == t_swz<2>(r): the fragment of row tile kt sits 16 rows = 512 elements behind that of tile 0 -- one base per operand,
            //  compile-time offsets per tile; the loop over the terms is a real loop, as in hh_gemm)
            const bf16_t *zrow = T2 + (n_base + r16) * (16 * MT) + 8 * (g ^ t_swz<MT>(r16)), *arow = T1 + r16 * (16 * MT) + 8 * (g ^ t_swz<MT>(r16));
#pragma unroll 1
            for (int t = S3_WG0; t < NT; t++) {
                const bf16_t *zp = zrow + s3_tb_rt(t) * T_PS, *ap = arow + s3_ta_rt(t) * T_PS;
                bf16x8 zb[NTW];
#pragma unroll
                for (int j = 0; j < NTW; j++) zb[j] = *reinterpret_cast<const bf16x8 *>(zp + j * 16 * (16 * MT));
                constexpr int TA = W8 ? 4 : 8, TAH = TA - 1;  // h1 fragments TAH row tiles ahead (one wave per SIMD: seven -- at three the phase ran at half the pipe rate)
                bf16x8 ta[TA];
#pragma unroll
                for (int kt = 0; kt < TAH; kt++) ta[kt] = *reinterpret_cast<const bf16x8 *>(ap + kt * 16 * (16 * MT));
#pragma unroll
                for (int kt = 0; kt < KT2; kt++) {
                    if (kt + TAH < KT2) ta[(kt + TAH) % TA] = *reinterpret_cast<const bf16x8 *>(ap + (kt + TAH) * 16 * (16 * MT));
#pragma unroll
                    for (int j = 0; j < NTW; j++) aW2[kt][j] = mfma_bf(ta[kt % TA], zb[j], aW2[kt][j]);
                    __builtin_amdgcn_sched_barrier(0);  // (per row tile: without the fence the scheduler sinks every read to its use -- one register set, lgkmcnt(0) in front of every four MFMAs)
                }
            }
        }
        S3_TICK(9);
        S3_RELANE();
        hh_gemm(T2, W.bW2, W.fW2, dh1);
        S3_TICK(10);
        __syncthreads();  // every wave is done with T1 (all rows) and the dz2 planes
        S3_TICK(11);
        S3_RELANE();
        // ---- P6: dz1 = dh1 * (1 - h1^2) in place in the T1 planes (own rows);  dW1 slice += X^T . dz1[:, slice] ----
#pragma unroll
        for (int j = 0; j < NTW; j++) {
            const int n = n_base + 16 * j + r16;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                bf16x4 *tq = t_quad<MT>(T1, n, mt, g);
                float h[4], dz[4];
                unsplit_quad(*reinterpret_cast<const uint2 *>(tq), *reinterpret_cast<const uint2 *>(tq + T_PS / 4), *reinterpret_cast<const uint2 *>(tq + 2 * (T_PS / 4)), h);
#pragma unroll
                for (int r = 0; r < 4; r++) dz[r] = dh1[j][mt][r] * (1.0f - h[r] * h[r]);
                sB1[j] += (dz[0] + dz[1]) + (dz[2] + dz[3]);
                const bfq3 q = split_quad(dz[0], dz[1], dz[2], dz[3]);
#pragma unroll
                for (int p = 0; p < 3; p++) bfq_store_quad(tq + p * (T_PS / 4), q.p[p]);
            }
        }
#pragma unroll
        for (int t = S3_WG0; t < NT; t++) {
            bf16x8 zb[NTW];
#pragma unroll
            for (int j = 0; j < NTW; j++) zb[j] = t_frag<MT>(T1 + s3_tb(t) * T_PS, n_base + 16 * j + r16, 0, g);
#pragma unroll
            for (int kt = 0; kt < KT1C; kt++) {
                const bf16x8 a = t_frag<MT>(Xt + s3_ta(t) * XT_PS, 16 * kt + r16, 0, g);
#pragma unroll
                for (int j = 0; j < NTW; j++) aW1[kt][j] = mfma_bf(a, zb[j], aW1[kt][j]);
            }
        }
        S3_TICK(12);
        __syncthreads();
        S3_TICK(13);
    }
#ifdef TMA_S3_TICKS
    if (tick_on)
        for (int i = 0; i < 16; i++) atomicAdd(&g_s3_ticks[IS_PI ? 0 : 1][i], tick_lds[i]);
#endif
#undef S3_TICK
#undef S3_RELANE
    // ---- store this block's slab (every parameter of the net has exactly one owning wave) ----
    float *gW1 = slab + (IS_PI ? L.pW1t : L.vW1t), *gb1 = slab + (IS_PI ? L.pb1 : L.vb1);
    float *gW2 = slab + (IS_PI ? L.pW2t : L.vW2t), *gb2 = slab + (IS_PI ? L.pb2 : L.vb2);
    float *gW3 = slab + (IS_PI ? L.pW3t : L.vW3t), *gb3 = slab + (IS_PI ? L.pb3 : L.vb3);
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        const int col = n_base + 16 * j + r16;
#pragma unroll
        for (int kt = 0; kt < KT2; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) gW2[(int64_t)(kt * 16 + g * 4 + r) * H + col] = aW2[kt][j][r];
#pragma unroll
        for (int kt = 0; kt < KT1C; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = kt * 16 + g * 4 + r;
                if (k < D) gW1[(int64_t)k * H + col] = aW1[kt][j][r];
            }
        float v1 = sB1[j], v2 = sB2[j];  // fold the four lane groups' partial sums (rows 4 g .. 4 g + 3 of every row tile) in a fixed order
        v1 += __shfl_xor(v1, 16, 64), v1 += __shfl_xor(v1, 32, 64);
        v2 += __shfl_xor(v2, 16, 64), v2 += __shfl_xor(v2, 32, 64);
        if (g == 0) gb1[col] = v1, gb2[col] = v2;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = n_base + 16 * j + g * 4 + r;
            if (r16 < NOUT) gW3[(int64_t)k * NOUT + r16] = aW3[j][r];
        }
    }
    __syncthreads();
    {
        float v = ab3;
        v += __shfl_xor(v, 32, 64);  // lanes a and a + 32 hold the two halves of a tile's column sum
        if (wave >= 2 && wave < 4 && lane < 32) scratch[(wave - 2) * 32 + lane] = v;
        __syncthreads();
        if (wave == 0 && lane < 32) {
            const float s = scratch[lane] + scratch[32 + lane];
            if (lane < NOUT) gb3[lane] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double ssum = 0.0;
        for (int w = 0; w < NW * 4; w++) ssum += stat_lds[w * 5 + threadIdx.x];
        const int q = IS_PI ? (threadIdx.x == 0 ? 0 : threadIdx.x + 1) : (threadIdx.x == 0 ? 1 : -1);
        if (q >= 0) stat_slot[q] += ssum;
    }
}

template <int KT1C, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void ppo_grad_split3_kernel(const float *__restrict__ params, PLayout L, Rollout rb, Minibatch mb, HParams hp, float *__restrict__ slabs,
                                                                 double *__restrict__ stat_slots, int n_pi) {
    extern __shared__ __attribute__((aligned(16))) char smem_s3[];
    const bool is_pi = (int)blockIdx.x < n_pi;
    const int b = is_pi ? blockIdx.x : blockIdx.x - n_pi, nb = is_pi ? n_pi : (int)gridDim.x - n_pi;
    float *slab = slabs + (int64_t)b * L.P;
    double *slot = stat_slots + (int64_t)b * 8;
    if (is_pi) grad_split3_body<true, KT1C, NW>(params, L, rb, mb, hp, slab, slot, smem_s3, nb, b);
    else grad_split3_body<false, KT1C, NW>(params, L, rb, mb, hp, slab, slot, smem_s3, nb, b);
}

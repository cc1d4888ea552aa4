// tma_wide_bf16.h -- bf16-MFMA variant of the column-parallel wide-policy kernels (hidden = 128 / 192 / 256).
//
// Included inside namespace tma (after tma_ppo_types.h) by tma_policy.hip (forward kernels, image builder) and tma_bf16.hip (gradient kernel).
//
// Why a second data path: BASELINE.json configs[2] names "PPO MLP(256,256) bf16".  With f32 MFMA operands the 256-wide
// update re-streams 0.5 MB of f32 weights from L2 for every 32-sample row group and issues 4 flops/lane/cycle; with
// v_mfma_f32_16x16x32_bf16 (8 bf16 per lane per operand, f32 accumulate) the same GEMMs need 1/8 of the MFMA issue slots
// and half the weight bytes.  Master weights, Adam state, biases, the loss and every gradient ACCUMULATOR stay f32; only the
// MFMA operands (activations, weights, back-propagated deltas) are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32), which is what torch.autocast(bfloat16) does to the same SB3 MlpPolicy (third-party; constructed at
// /root/reference/backend/mlagents/training.py:150 with the net_arch of training.py:363-365).
//
// Data layout
//   * weights: "fragment-major" bf16 images (BfNet in tma_mlp.h) rebuilt by build_bf16_images_kernel after every Adam
//     step: one B fragment = one coalesced 1 KiB load, no LDS staging, no transposes at run time.
//   * activations of a row group of M = 16*MT samples live in LDS twice:
//       A image  [M][K + 16] bf16   row-major, row stride == 32 B mod 64 B  -> conflict-free ds_read_b128 A fragments
//       T image  [K][M]      bf16   transposed, 16-byte chunks XOR-swizzled -> conflict-free ds_read_b128 fragments for
//                                   the weight-gradient GEMMs (reduction over the samples) and 8-byte C-layout access.
//   * a block = 4 waves (one per SIMD, 512 registers each); wave w owns output columns [w*H/4, (w+1)*H/4) of both hidden
//     layers and keeps its slice of dW2 / dW1 / dW3 in MFMA accumulators for the whole launch (as the f32 kernel does).
#pragma once

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// Two elements per VALU instruction where the operation allows it (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 run at twice the scalar rate):
// the epilogues of this kernel are instruction-issue-bound.  Component for component the same IEEE operations as
// tma_tanh and as dz = dh * (1 - h * h), hence the same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// PK = false: one element per instruction (opaque values between the steps keep the compiler from re-packing them).  Beside ANOTHER wave's MFMAs
// on the same SIMD -- the eight-wave kernels -- packed f32 VALU costs more than it saves (MI355X_MICROARCH.md, 'price of one filler beside MFMAs';
// measured round 4: 250 -> 245 us on the Ball3D shape), at one wave per SIMD it is the cheaper form (round 3: 266 -> 261 us).
template <bool PK = true>
__device__ __forceinline__ f32x2 tma_tanh2(f32x2 x) {
    if constexpr (!PK) {
        f32x2 o;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float t = x[i] * 2.8853900817779268f;
            asm volatile("" : "+v"(t));
            float d = __builtin_amdgcn_exp2f(t) + 1.0f;
            asm volatile("" : "+v"(d));
            o[i] = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(d), 1.0f);
        }
        return o;
    }
    const f32x2 t = x * 2.8853900817779268f;
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(t[0]);
    e[1] = __builtin_amdgcn_exp2f(t[1]);
    const f32x2 d = e + 1.0f;
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(d[0]);
    r[1] = __builtin_amdgcn_rcpf(d[1]);
    return __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r, f32x2{1.0f, 1.0f});
}
template <bool PK = true>
__device__ __forceinline__ f32x2 delta2(f32x2 dh, f32x2 h) {
    if constexpr (!PK) {
        f32x2 o;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            // h is a bf16 value (8 significant bits), so h * h is exact in f32 and ONE v_fma_f32 rounds 1 - h^2 exactly as the
            // multiply + subtract pair did: same bits, one instruction less per element (round 5)
            float om = __builtin_fmaf(-h[i], h[i], 1.0f);
            asm volatile("" : "+v"(om));
            o[i] = dh[i] * om;
        }
        return o;
    }
    const f32x2 om = __builtin_elementwise_fma(-h, h, f32x2{1.0f, 1.0f});
    return dh * om;
}

// ---- tile epilogues on PACKED pairs (round 5) ----
// A C-layout quad (four consecutive samples of one column) leaves an epilogue as two registers of packed bf16: ONE v_cvt_pk_bf16_f32 per
// element pair.  Everything downstream takes its halves from those registers -- the 8-byte store into the T image, the four 2-byte stores
// into the row-major image (ds_write_b16 / ds_write_b16_d16_hi), the bias-gradient sums (v_dot2c_f32_bf16 against packed ones).  Written
// element-wise (`A[..] = (bf16_t)x` beside a bf16x4 built from the same values) the compiler converted every element twice: 6 conversions
// per quad instead of 2, 32 extra VALU instructions in each of the three epilogues that keep a row-major image, and the per-lane bias sums
// cost 4 unpacks + 3 adds per quad where two dot instructions do.  Same RNE conversion of the same f32 values: the images hold the same bits.
struct bfq {
    uint32_t lo, hi;  // elements 0, 1 | 2, 3
};
__device__ __forceinline__ uint32_t bf_pack2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (bf16_t)a, v[1] = (bf16_t)b;
    uint32_t p = __builtin_bit_cast(uint32_t, v);
    asm volatile("" : "+v"(p));  // opaque: consumers split THIS register, nothing re-converts the f32 values
    return p;
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
template <bool PK>
__device__ __forceinline__ bfq tanh_quad(const f32x4 &acc) {
    const f32x2 a = tma_tanh2<PK>(f32x2{acc[0], acc[1]}), b = tma_tanh2<PK>(f32x2{acc[2], acc[3]});
    return bfq{bf_pack2(a[0], a[1]), bf_pack2(b[0], b[1])};
}
// (the rollout kernels' form: tma_tanh per element -- component for component tma_tanh2's operations)
__device__ __forceinline__ bfq tanh_quad_s(const f32x4 &acc) {
    return bfq{bf_pack2(tma_tanh(acc[0]), tma_tanh(acc[1])), bf_pack2(tma_tanh(acc[2]), tma_tanh(acc[3]))};
}
// dz = dh * (1 - h^2) with h the packed bf16 quad read back from the T image
template <bool PK>
__device__ __forceinline__ bfq delta_quad(const f32x4 &dh, const uint2 h) {
    const f32x2 a = delta2<PK>(f32x2{dh[0], dh[1]}, f32x2{bf_lo(h.x), bf_hi(h.x)}), b = delta2<PK>(f32x2{dh[2], dh[3]}, f32x2{bf_lo(h.y), bf_hi(h.y)});
    return bfq{bf_pack2(a[0], a[1]), bf_pack2(b[0], b[1])};
}
// rows r .. r + 3 of one column of a row-major image (dst = &A[r][n], ld elements per row)
__device__ __forceinline__ void bfq_store_rows(bf16_t *dst, int ld, const bfq q) {
    *reinterpret_cast<uint16_t *>(dst) = (uint16_t)q.lo;
    *reinterpret_cast<uint16_t *>(dst + ld) = (uint16_t)(q.lo >> 16);
    *reinterpret_cast<uint16_t *>(dst + 2 * ld) = (uint16_t)q.hi;
    *reinterpret_cast<uint16_t *>(dst + 3 * ld) = (uint16_t)(q.hi >> 16);
}
__device__ __forceinline__ void bfq_store_quad(bf16x4 *dst, const bfq q) { *reinterpret_cast<uint2 *>(dst) = uint2{q.lo, q.hi}; }
// s + the four elements of the quad (f32 accumulate; v_dot2c_f32_bf16 with both weights 1.0)
__device__ __forceinline__ float bfq_sum(float s, const bfq q) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t ones;
    ones[0] = (bf16_t)1.0f, ones[1] = (bf16_t)1.0f;
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, q.lo), ones, s, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, q.hi), ones, s, false);
}

// B fragment `idx` of a fragment-major image.  The load is made through an explicit global (address space 1) pointer: a
// pointer that went through the LICM-defeating asm in the gradient kernel is "generic" to the compiler, and a flat_load
// counts on lgkmcnt as well as vmcnt -- every LDS wait would then also wait for the weight prefetch in flight.
typedef const bf16x8 __attribute__((address_space(1))) *bf_gptr;
// value the optimiser must treat as new (not loop-invariant) but provably wave-uniform: SGPR addressing survives
__device__ __forceinline__ const bf16_t *launder_uniform(const bf16_t *p) {
    uint64_t v = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+s"(v));
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const bf16_t *>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int launder_uniform(int x) {
    asm volatile("" : "+s"(x));
    return __builtin_amdgcn_readfirstlane(x);
}
__device__ __forceinline__ bf16x8 bf_frag(const bf16_t *__restrict__ img, int idx, int lane) {
    // uniform (SGPR) fragment base + one 32-bit per-lane offset: the saddr form of global_load, no 64-bit VGPR address per fragment
    typedef const char __attribute__((address_space(1))) *gbyte_ptr;
    const gbyte_ptr base = reinterpret_cast<gbyte_ptr>(reinterpret_cast<uintptr_t>(img + (int64_t)idx * 512));
    const uint32_t voff = (uint32_t)lane * 16u;  // 32-bit per-lane offset + scalar base: global_load ... v, s[base:base+1]
    return *reinterpret_cast<bf_gptr>(base + voff);
}

// ---- T image addressing: row `r` holds M bf16 (M/8 chunks of 16 bytes); chunk c is stored at position c ^ t_swz<MT>(r).
// MT = 2 (64-byte rows): swz = (-(r >> 2)) & 3;  MT = 4 (128-byte rows): swz = r & 7.  With either, the four
// 16-lane groups of a ds_read_b128 whose lanes read (row = r0 + (l&15), chunk = c0 + (l>>4)) cover 64 distinct banks.
// (MT = 4, rounds 1-3: swz = 2 * ((r >> 1) & 3) -- just as free of conflicts for those reads, but the epilogues' ds_write_b64 quads (16
// consecutive rows per lane group, banks taken modulo 32 for stores) then landed on FOUR positions: 16 LDS cycles an instruction instead
// of 8, and the 16-byte stores of the observation / dz3 images 16 instead of 8 -- a third of the kernel's SQ_LDS_BANK_CONFLICT count.)
template <int MT>
__device__ __forceinline__ int t_swz(int r) {
    if constexpr (MT == 2) return (-(r >> 2)) & 3;
    else return r & 7;
}
template <int MT>
__device__ __forceinline__ int t_off(int r, int m) {  // element offset of T[r][m]
    return r * (16 * MT) + 8 * ((m >> 3) ^ t_swz<MT>(r)) + (m & 7);
}
// fragment for an MFMA whose reduction runs over samples 32*kk .. 32*kk+31: lane reads T[row][32kk + 8g .. +7]
template <int MT>
__device__ __forceinline__ bf16x8 t_frag(const bf16_t *T, int row, int kk, int g) {
    return *reinterpret_cast<const bf16x8 *>(T + row * (16 * MT) + 8 * ((4 * kk + g) ^ t_swz<MT>(row)));
}
// C-layout access: the 4 samples m = 16mt + 4g .. +3 of column n
template <int MT>
__device__ __forceinline__ bf16x4 *t_quad(bf16_t *T, int n, int mt, int g) {
    return reinterpret_cast<bf16x4 *>(T + t_off<MT>(n, 16 * mt + 4 * g));
}
__device__ __forceinline__ bf16x8 a_frag(const bf16_t *A, int ld, int row, int ks, int g) {
    return *reinterpret_cast<const bf16x8 *>(A + row * ld + 32 * ks + 8 * g);
}
// The same A fragment (rows = samples 16 mt .. + 15, k = 32 ks + 8 g + j) read from the TRANSPOSED image T[k][m] with the hardware
// transposing read: per 16-lane group, ds_read_b64_tr_b16 takes a block of 4 image rows (k) x 16 columns (samples) -- lane 4q + p supplies
// the address of row q, columns 4p .. 4p + 3 -- and hands lane i column i of the four rows.  Two reads (k rows 8g .. + 3 and + 4 .. + 7)
// make the eight elements.  With it the activations and deltas need no row-major copy: an epilogue writes one 8-byte quad per four
// elements into the T image instead of that plus four 2-byte stores (whose addresses and LDS issue slots were a quarter of the epilogue).
// EXEC must be all ones (the gather crosses lanes): only called from wave-uniform code.  Measured (one box, 131 072 samples): Crawler width
// 515.7 -> 494.6 us, H = 128 159.8 -> 157.0, H = 192 264.5 -> 262.4.
#ifndef TMA_BF_TR_READS
#define TMA_BF_TR_READS 1  // 0: row-major A images + ds_read_b128 everywhere (A/B builds).  Used for 32-row groups only: at 64-row groups the
                           // eight per-row-tile address registers of the transposed reads push the 510-register variant into spills (+3 %)
#endif
#ifndef TMA_BF_PFD
#define TMA_BF_PFD 0
#endif
#ifndef TMA_BF_TR_MT4
#define TMA_BF_TR_MT4 0  // 1: transposed reads (no row-major activation images) at 64-row groups too (A/B builds)
#endif
template <int MT>
constexpr bool bf_tr_reads() { return TMA_BF_TR_READS && (MT == 2 || (MT == 4 && TMA_BF_TR_MT4)); }
template <int MT>
__device__ __forceinline__ bf16x8 a_frag_t(const bf16_t *T, int mt, int ks, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef s16x4 __attribute__((address_space(3))) *lds_s16x4;
    const int gl = lane & 15, q = gl >> 2, p = gl & 3, g = lane >> 4;
    const int k0 = 32 * ks + 8 * g + q, k1 = k0 + 4;
    const int ch = 2 * mt + (p >> 1), sub = 4 * (p & 1);
    const bf16_t *a0 = T + k0 * (16 * MT) + 8 * (ch ^ t_swz<MT>(k0)) + sub;
    const bf16_t *a1 = T + k1 * (16 * MT) + 8 * (ch ^ t_swz<MT>(k1)) + sub;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// The same reads with the address arithmetic hoisted out of the k loop (round 5).  For both swizzles t_swz<MT>(k + 32 ks) == t_swz<MT>(k)
// (MT = 2: (-(k >> 2)) & 3 with 8 ks a multiple of 4; MT = 4: k & 7), so the two row addresses of a lane for k-step ks are those of k-step 0
// plus ks * 32 image rows: two base pointers per row tile, computed once per phase, and a compile-time offset per k-step (the ds_read
// offset field).  Recomputed per fragment, a_frag_t's addresses were ~250 of the ~880 VALU instructions of a 32-row group (P2: 128 for
// 32 MFMAs, P5: 119 for 64).
template <int MT>
struct TrBase {
    const bf16_t *lo[MT], *hi[MT];
};
template <int MT>
__device__ __forceinline__ TrBase<MT> tr_base(const bf16_t *T, int lane) {
    const int gl = lane & 15, q = gl >> 2, p = gl & 3, g = lane >> 4;
    const int k0 = 8 * g + q, k1 = k0 + 4, sub = 4 * (p & 1);
    TrBase<MT> b;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int ch = 2 * mt + (p >> 1);
        b.lo[mt] = T + k0 * (16 * MT) + 8 * (ch ^ t_swz<MT>(k0)) + sub;
        b.hi[mt] = T + k1 * (16 * MT) + 8 * (ch ^ t_swz<MT>(k1)) + sub;
    }
    return b;
}
template <int MT>
__device__ __forceinline__ bf16x8 a_frag_tb(const TrBase<MT> &b, int mt, int ks) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef s16x4 __attribute__((address_space(3))) *lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(b.lo[mt] + ks * 32 * 16 * MT));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(b.hi[mt] + ks * 32 * 16 * MT));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// A fragment of activations / deltas held in both images (A image + T image): the transposed read, or the row-major one
template <int MT>
__device__ __forceinline__ bf16x8 act_frag(const bf16_t *A, int ld, const bf16_t *T, int mt, int ks, int lane) {
    if constexpr (bf_tr_reads<MT>()) return a_frag_t<MT>(T, mt, ks, lane);
    else return a_frag(A, ld, 16 * mt + (lane & 15), ks, lane >> 4);
}

// rebuilds the fragment-major bf16 images of both nets from the f32 master weights ([in][out] flat layout)
static __global__ void build_bf16_images_kernel(float *params, PLayout L) {
    const int D = L.D, H = L.H;
    const int Kp1 = (D + 31) & ~31, KS1 = Kp1 / 32, KS2 = H / 32;
    for (int net = 0; net < 2; net++) {
        const int n_out = net == 0 ? L.A : 1;
        const BfNet B = bf_net_layout(D, H, n_out);
        bf16_t *img = reinterpret_cast<bf16_t *>(params + (net == 0 ? L.bf_pi : L.bf_vf));
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
            img[e] = (bf16_t)v;
        }
    }
}

// One hidden layer for this wave's NTW column tiles and all MT row tiles of the group:
//   Aout[m][n] (and Tout[n][m]) = bf16(tanh(Ain[m][:] . W[:, n] + b[n])).   Ain: A image with KS k-steps of 32.
template <int NTW, int MT, bool STORE_T>
__device__ __forceinline__ void bf_hidden_layer(const bf16_t *Ain, int ldin, int KS, const bf16_t *__restrict__ Wimg, const float *__restrict__ bias,
                                                bf16_t *Aout, int ldo, bf16_t *Tout, int n_base, int lane) {
    const int r16 = lane & 15, g = lane >> 4, nt0 = n_base >> 4;
    f32x4 acc[NTW][MT];
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        const float b = bias[n_base + 16 * j + r16];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[j][mt] = f32x4{b, b, b, b};
    }
#pragma unroll 4
    for (int ks = 0; ks < KS; ks++) {
        bf16x8 w[NTW];
#pragma unroll
        for (int j = 0; j < NTW; j++) w[j] = bf_frag(Wimg, (nt0 + j) * KS + ks, lane);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const bf16x8 a = a_frag(Ain, ldin, 16 * mt + r16, ks, g);
#pragma unroll
            for (int j = 0; j < NTW; j++) acc[j][mt] = mfma_bf(a, w[j], acc[j][mt]);
        }
    }
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        const int n = n_base + 16 * j + r16;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const bfq q = tanh_quad<true>(acc[j][mt]);
            bfq_store_rows(Aout + (16 * mt + 4 * g) * ldo + n, ldo, q);
            if constexpr (STORE_T) bfq_store_quad(t_quad<MT>(Tout, n, mt, g), q);
        }
    }
}

// head of one 16-row tile: acc[q] (C layout, column 16q + (lane&15)) = A2[tile rows][:] . W3 + b3.
// Summation order = the gradient kernel's split-K head (four partial sums over k-steps ks = w, w + 4, ..., added to the bias
// in wave order), so the log-probabilities of the rollout and of the first update epoch agree bit for bit.
template <int NT3>
__device__ __forceinline__ void bf_head(const bf16_t *A2, int ld, int row0, int KS2, const bf16_t *__restrict__ W3img, const float *__restrict__ b3,
                                        int n_out, f32x4 (&acc)[NT3], int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 part[4][NT3];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int q = 0; q < NT3; q++) part[w][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // partial w covers the k-steps the gradient kernel's wave w takes: w*KS2/4 .. (KS2 % 4 == 0: its own h2 columns), else w, w + 4
    const bool ownk = (KS2 & 3) == 0;
    const int hk = (KS2 + 3) >> 2;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int ks = ownk ? w * hk + i : w + 4 * i;
            if (i < hk && ks < KS2) {
                const bf16x8 a = a_frag(A2, ld, row0 + r16, ks, g);
#pragma unroll
                for (int q = 0; q < NT3; q++) part[w][q] = mfma_bf(a, bf_frag(W3img, q * KS2 + ks, lane), part[w][q]);
            }
        }
#pragma unroll
    for (int q = 0; q < NT3; q++) {
        const int col = 16 * q + r16;
        const float b = col < n_out ? b3[col] : 0.0f;
        acc[q] = f32x4{b, b, b, b};
#pragma unroll
        for (int w = 0; w < 4; w++) acc[q] += part[w][q];
    }
}

// ---- Categorical loss with ONE ROW PER LANE (round 6; the eight-wave 64-row-group kernel) ----
// policy_loss_tile works in the head's C layout: a lane is one action column of four rows, every row-wise quantity (max, sum of exponentials,
// log-probability of the taken action, entropy) is a DPP reduction over sixteen lanes, and a wave spends ~140 vector instructions on two rows
// of each lane group -- eight waves, two per SIMD, 1 120 wave-instructions for the 64 rows of a group, 3.4 k cycles of a policy-net group's
// 28 k.  Transposed through LDS (the f32 dz3 tile doubles as the logit tile) a lane owns a row: the reductions are straight-line code over
// the row's A logits, ~45 + 35 A instructions for sixteen rows at once.  SAME BITS as policy_loss_tile: sums in gsum16's tree order
// ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)) ... with the padding terms' zeros, the same expression for every intermediate.
// AP: the tree's width (4, 8 or 16 >= A).  x: the row's logits (LDS, f32); out: the row's dz3 entries [0, A) (the same LDS words).
template <int AP>
__device__ __forceinline__ float bf_tree_sum(const float (&v)[AP]) {
    float t[AP];
#pragma unroll
    for (int i = 0; i < AP; i++) t[i] = v[i];
#pragma unroll
    for (int w = 1; w < AP; w *= 2)
#pragma unroll
        for (int i = 0; i < AP; i += 2 * w) t[i] = t[i] + t[i + w];
    return t[0];  // (+ the zeros of the lanes beyond AP: x + 0 == x)
}
template <int AP>
__device__ __forceinline__ void discrete_loss_row(float *xrow, const f32x4 mrow, bool valid, int A, float amean, float astd, const HParams &hp, float invB,
                                                  double *sl /* this row lane's five statistics sums in LDS, or nullptr */) {
    float x[AP], e[AP];  // logits -> log-probabilities, exponentials -> probabilities, in place (the accumulators of the whole launch are live around this)
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < AP; a++) {
        x[a] = a < A ? xrow[a] : -INFINITY;
        m = fmaxf(m, x[a]);
    }
#pragma unroll
    for (int a = 0; a < AP; a++) e[a] = a < A ? expf(x[a] - m) : 0.0f;
    const float s = bf_tree_sum<AP>(e);
    const float lse = m + logf(s);
    const int act = __float_as_int(mrow[3]);
    float lpa = 0.0f, ent;
    {
        float plp[AP];
#pragma unroll
        for (int a = 0; a < AP; a++) {
            x[a] = a < A ? x[a] - lse : 0.0f;
            e[a] = e[a] / s;
            plp[a] = e[a] * x[a];
            lpa = (a == act) ? x[a] : lpa;
        }
        ent = -bf_tree_sum<AP>(plp);
    }
    const float old = mrow[0];
    const float advn = (mrow[1] - amean) / (astd + 1e-8f);
    const float ratio = expf(lpa - old);
    const float pl1 = advn * ratio;
    const float rc = fminf(fmaxf(ratio, 1.0f - hp.clip_range), 1.0f + hp.clip_range);
    const float pl2 = advn * rc;
    const float g_lp = (valid && pl1 <= pl2) ? -(advn * ratio) * invB : 0.0f;
#pragma unroll
    for (int a = 0; a < AP; a++) {
        if (a < A) {
            float dl = g_lp * (((a == act) ? 1.0f : 0.0f) - e[a]);
            dl += valid ? (hp.ent_coef * invB) * (e[a] * (x[a] + ent)) : 0.0f;
            xrow[a] = dl;
        }
    }
    if (sl && valid) {  // (read-add-write one after the other: no five doubles live beside the launch's accumulators)
        sl[0] += (double)(-fminf(pl1, pl2));
        sl[1] += (double)ent;
        sl[2] += (double)((ratio - 1.0f) - (lpa - old));
        sl[3] += fabsf(ratio - 1.0f) > hp.clip_range ? 1.0 : 0.0;
        sl[4] += 1.0;
    }
}

struct BfNetPtr {
    const bf16_t *fW1, *fW2, *bW2, *fW3, *bW3;
};
__device__ __forceinline__ BfNetPtr bf_net_ptr(const float *params, const PLayout &L, bool is_pi) {
    const BfNet B = bf_net_layout(L.D, L.H, is_pi ? L.A : 1);
    const bf16_t *img = reinterpret_cast<const bf16_t *>(params + (is_pi ? L.bf_pi : L.bf_vf));
    return BfNetPtr{img + B.fW1, img + B.fW2, img + B.bW2, img + B.fW3, img + B.bW3};
}

// LDS bytes of the gradient kernel for a row group of M = 16*MT samples
// MT = 4 (64-row groups): one wave per row tile runs the whole head + loss, so there are no split-K partial sums; and from H = 192 on
// dz3 (f32) and its two bf16 images live in the A1 image, which is dead between the layer-2 forward pass and the next group
__host__ __device__ inline bool bf_alias_z3(int H, int MT) { return MT == 4 && H >= 192; }
__host__ __device__ inline int grad_wide_bf_smem_bytes(int D, int H, int MT) {
    const int M = 16 * MT, Kp1 = (D + 31) & ~31;
    if (MT == 4) {
        const int z3 = bf_alias_z3(H, MT) ? 0 : (M * 48 + 32 * M) * 2 + M * 34 * 4;
        const int bf4 = M * (Kp1 + 16) + Kp1 * M + 2 * M * (H + 16) + 2 * H * M;
        return bf4 * 2 + z3 + (M * 4 + 128 + 2 * H + 32) * 4 + 4 * 4 * 5 * 8 + 2 * M * 8;
    }
    const int bf = M * (Kp1 + 16) + Kp1 * M + 2 * M * (H + 16) + 2 * H * M + M * 48 + 32 * M;
    // + f32: dz3, meta, scratch, head partial sums [4 waves][MT][2][64][4], biases [2H + 32]; f64: stats [MT][4][5]; i64: row offsets x2
    return bf * 2 + (M * 34 + M * 4 + 128 + 4 * MT * 2 * 256 + 2 * H + 32) * 4 + 4 * 4 * 5 * 8 + 2 * M * 8 + 4 * 4 * 5 * 8 + M * 32 * 4;  // (+ eight-wave statistics, + the action tile)
}
__host__ __device__ inline int fwd_wide_bf_smem_bytes(int D, int H) {
    const int Kp1 = (D + 31) & ~31;
    return (32 * (Kp1 + 16) + 2 * 32 * (H + 16)) * 2;
}

constexpr int bf_ring_slots(int stream_len, int cap) {
    int r = 1;
    for (int d = 1; d <= cap; d++)
        if (stream_len % d == 0) r = d;
    return r;
}

// Gradient of one minibatch for ONE net, persistent over row groups of M = 16*MT samples.
//
// Latency plan (one wave per SIMD, so nothing but this wave's own loads in flight hides the L2 round trip):
//   * the two H x H weight streams of a group (layer-2 forward fragments, then layer-2 input-gradient fragments) are the
//     same for every group, so they run through a register ring of R = 4*NTW fragments: the slot a fragment is consumed
//     from is reloaded at once with the fragment R positions further down the (cyclic) stream -- four k-steps of lookahead
//     that carry across phases, barriers and groups;
//   * group-invariant small operands stay in registers for the whole launch (layer-1 fragments when D <= 32, the head's
//     input-gradient fragments); the head's forward fragments are fetched behind the layer-2 epilogue;
//   * the NEXT group's sample metadata and observation rows are gathered into registers while this group computes and
//     committed to LDS at the top of the next iteration (KS1C > 0: compile-time observation width).
// KT1C: 16-row k-tiles of dW1 kept in registers (0: accumulated in the slab);  KS1C: compile-time layer-1 k-steps (0: runtime).
// PASS: observations wider than 32 leave no registers for dW1 next to dW2's 256 accumulators.  With a compile-time width the
// minibatch is then walked twice: PASS 0 accumulates everything except dW1, PASS 1 recomputes the forward / backward chain and
// keeps only dW1 (2*KS1C k-tiles) in registers -- 1.8x the MFMA work instead of a 0.4 MB read-modify-write of the slab per row
// group.  (Every store of PASS 0's quantities is compiled out of PASS 1, so the MFMAs that feed only them disappear as dead code.)
// NW = 8 (round 4, 64-row groups of single-k-step layouts only): two waves per SIMD, 32 columns each (NTW = 2).  The dW2 slice of a wave halves to
// 128 accumulator registers; what does not halve with it has to shrink for 256 registers a wave to hold: the head fragments wait in LDS
// (8 KB, read back where the head multiplies: only the four head waves need them) and the hidden-layer bias gradients are per-lane sums
// instead of ones^T . dz accumulator tiles.  One wave's epilogue / loss / LDS waits then run beside its SIMD partner's MFMAs.
template <bool CONT, bool IS_PI, int NTW, int MT, int KT1C, int KS1C, int PASS, int NW = 4>
__device__ __forceinline__ void grad_wide_bf_body(const float *__restrict__ params, const PLayout &L, const Rollout &rb, const Minibatch &mb,
                                                  const HParams &hp, const float *__restrict__ ws_adv, float *__restrict__ slab,
                                                  double *__restrict__ stat_slot, char *smem, int n_blocks_net, int block_net,
                                                  bf16_t *__restrict__ dz1c) {
    static_assert(NW == 4 || (NW == 8 && ((MT == 4 && KS1C == 1 && PASS == 0 && !CONT) || (MT == 2 && KS1C > 2 && KT1C == 0))),
                  "eight waves: 64-row groups of the single-k-step Discrete layouts, or 32-row groups of the wide-observation two-pass layouts");
    constexpr bool W8 = NW == 8;
#ifndef TMA_BF_ROW_LOSS
#define TMA_BF_ROW_LOSS 1  // 0: the C-layout loss on all eight waves (A/B builds)
#endif
    constexpr bool RL = W8 && MT == 4 && !CONT && TMA_BF_ROW_LOSS;  // head, loss (a row per lane) and both dz3 images of row tile w on wave w alone: one block barrier in P3 instead of two
    constexpr int SLN = RL ? 64 : NW * 4;                           // statistics slots in LDS (RL: one per row lane of waves 0 .. 3)
#ifndef TMA_BF_W8_BIAS_VALU
#define TMA_BF_W8_BIAS_VALU 1
#endif
    constexpr bool BV = W8 && TMA_BF_W8_BIAS_VALU;  // hidden-layer bias gradients as per-lane sums (A/B: 0 = the ones^T . dz accumulator tiles, 12 more registers)
    constexpr int M = 16 * MT, MK = MT / 2, H = 16 * NTW * NW, KS2 = H / 32, KT2 = H / 16, NT3 = (IS_PI && CONT) ? 2 : 1, lda = H + 16, ldz = 48, ld3 = 34;
    // weight stream of a row group: [layer-1 fragments when KS1C > 1: k-step outer, tile inner] [layer-2 forward] [layer-2 input-gradient]
    constexpr int S1 = KS1C > 1 ? KS1C * NTW : 0, SL = S1 + 2 * KS2 * NTW;
    // largest divisor of SL not above four k-steps of fragments (slots are static registers); 64-row groups: two k-steps (each k-step is
    // twice the MFMA work, and the accumulators of four row tiles need the registers)
    constexpr int R = bf_ring_slots(SL, (MT == 4 ? 2 : 4) * NTW);
    constexpr bool PF = KS1C > 0 && KS1C <= 2;  // observation prefetch into registers: compile-time width, at most 8 registers per thread
    // Round 5: a thread owns column (tid & 7) of 8-column group cg of row (tid >> 3) + 8 NW rp -- the row (its buffer offset, its validity, both
    // image addresses) is worked out once per row pass and the column groups differ by compile-time offsets; and only the CG groups that CAN
    // hold observations are gathered and committed (KT1C == 1: D <= 16, two groups -- Ball3D's 6 and GridWorld's 4 floats used to be
    // gathered and committed as 32 columns a row, 26 of them zeros, 96 + 65 VALU instructions per group); the padding columns of both
    // images are cleared once per launch.
    constexpr int CG = PF ? (KT1C == 1 ? 2 : 4 * KS1C) : 1, RP = PF ? 2 * MT / NW : 1;
    static_assert(!PF || (2 * MT) % NW == 0, "observation prefetch: 64 NW threads cover 8 NW rows of 8 columns per pass");
    constexpr int NX = PF ? RP * CG : 1;
    const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index in an SGPR: fragment bases stay scalar
    // Every lane-derived LDS / global address in the group loop is loop-invariant; hoisted, there are far more of them than
    // registers and they come back as scratch reloads -- each one an s_waitcnt vmcnt(0) that drains the weight prefetch.
    // TMA_RELANE at the head of a phase re-derives lane / r16 / g from an opaque copy, so the addresses of that phase are
    // computed there (a few VALU instructions) and die with it.
    int lane = lane0, r16 = lane0 & 15, g = lane0 >> 4;
    // (mrow / mlane -- the metadata row of this lane, below -- follow the same rule: their three LDS addresses, hoisted, were the spills of the
    //  Ball3D / Push instantiation, and one of the reloads sat right behind the next group's gathers with an s_waitcnt vmcnt(0))
    int mrow = 0;
    bool mlane = false;
#ifdef TMA_BF_PHASE_TICKS  // diagnostic build (make libtma_hip_bfticks.so, tools/bf_ticks.py): cycles per phase, wave 0 of block 0 of each net
    // (sums are kept in LDS and copied out once at the end: a global atomic per stamp stays in vmcnt for ~1 k cycles and every counted wait
    //  for a weight fragment behind it would wait for it too)
    __shared__ unsigned long long tick_lds[16];
    if (threadIdx.x < 16) tick_lds[threadIdx.x] = 0;
    long long tlast = clock64();
    const bool tick_on = threadIdx.x == 0 && block_net == 0;
#define TMA_TICK_DO(i)                                                     \
    do {                                                                   \
        const long long tn = clock64();                                    \
        if (tick_on) tick_lds[i] += (unsigned long long)(tn - tlast);      \
        tlast = clock64();                                                 \
    } while (0)
#ifdef TMA_BF_TICK_P3_ONLY  // stamps of the loss phase only (the full set costs the eight-wave kernel 64 spilled registers, whose reloads wait for the HBM gathers in flight)
#define TMA_TICK(i)
#else
#define TMA_TICK(i) TMA_TICK_DO(i)
#endif
#define TMA_TICK3(i) TMA_TICK_DO(i)
#else
#define TMA_TICK(i)
#define TMA_TICK3(i)
#endif
#define TMA_RELANE()                                              \
    do {                                                          \
        lane = lane0;                                             \
        asm volatile("" : "+v"(lane));                            \
        r16 = lane & 15, g = lane >> 4;                           \
        mrow = wave * (M / NW) + lane, mlane = lane < M / NW;     \
    } while (0)
    const int D = L.D, A = L.A;
    const int NOUT = IS_PI ? A : 1;
    const int Kp1 = KS1C > 0 ? 32 * KS1C : ((D + 31) & ~31), KS1 = Kp1 >> 5, KT1 = Kp1 >> 4, ldx = Kp1 + 16;
    constexpr bool MAIN = PASS == 0;
    constexpr bool two_pass = KT1C == 0 && KS1C > 0;
    static_assert(PASS == 0 || two_pass, "PASS 1 / 2 exist only for wide observations of compile-time width");
    // PASS 2 replaces PASS 1 when the minibatch fits the dz1 cache of the workspace (dz1c != nullptr in both launches): PASS 0 leaves
    // every group's dz1 image (bf16, the T-image layout of T1: H x M, 16 KB at H = 256) in HBM and PASS 2 only gathers the
    // observation rows again and runs the dW1 MFMAs on those images -- the same bf16 operands in the same order as PASS 1, so the
    // two give identical bits, without recomputing the forward / backward chain.
    constexpr bool acc_w1 = KT1C > 0 || PASS >= 1;              // dW1 in registers
    constexpr bool rmw_w1 = KT1C == 0 && KS1C == 0;             // dW1 accumulated in the slab (runtime width)
    constexpr int KT1A = KT1C > 0 ? KT1C : (PASS >= 1 ? 2 * KS1C : 1);
    bf16_t *Xa = reinterpret_cast<bf16_t *>(smem), *Xt = Xa + M * ldx, *A1 = Xt + Kp1 * M, *A2 = A1 + M * lda;
    bf16_t *T1 = A2 + M * lda, *T2 = T1 + H * M;
    constexpr bool Z3_IN_A1 = MT == 4 && H >= 192;  // == bf_alias_z3(H, MT): A1 is dead from the barrier behind P2 to the next group's P1
    bf16_t *Z3a = Z3_IN_A1 ? A1 : T2 + H * M, *Z3t = Z3a + M * ldz;
    float *dz3 = reinterpret_cast<float *>(Z3t + 32 * M);
    float *meta = Z3_IN_A1 ? reinterpret_cast<float *>(T2 + H * M) : dz3 + M * ld3, *scratch = meta + M * 4;  // scratch: 128 floats
    float *hpart = scratch + 128;              // MT = 2: [4 waves][MT][2][64 lanes][4] split-K partial head outputs (MT = 4: none)
    float *bias = hpart + (MT == 4 ? 0 : 4 * MT * 2 * 256);  // b1[H], b2[H], b3[32] (zero padded): LDS copies, no global load in front of a phase
    double *stat_lds = reinterpret_cast<double *>(bias + 2 * H + 32);  // [4 waves][4][5] loss statistics (lanes r16 == 0)
    int64_t *row_off = reinterpret_cast<int64_t *>(stat_lds + SLN * 5), *row_off_next = row_off + M;
    bf16_t *W3lds = reinterpret_cast<bf16_t *>(row_off_next + M);  // W8: [KS2 * NT3] head fragments of 1 KiB (grad_wide_bf_smem_bytes adds them)
    // Box heads, 32-row groups with the direct observation gather (round 6): the group's actions [M][32] f32, gathered with the observation rows --
    // the loss read them from global memory where it needed them, a cache-missing round trip in the middle of the phase whose MFMAs are idle
    constexpr bool ACT_TILE = CONT && IS_PI && MT == 2 && !(KS1C > 0 && KS1C <= 2) && PASS != 2;
    float *act_tile = reinterpret_cast<float *>(reinterpret_cast<char *>(row_off_next + M) + 4 * 4 * 5 * 8);  // (behind the statistics of four more waves: MT == 2 layouts have no W3lds)
    const int n_base = wave * 16 * NTW, nt0 = wave * NTW;
    const float invB = 1.0f / (float)mb.count;
    // minibatch advantage statistics: folded here from the partials (the order of adv_final_kernel, so the same bits) instead of by a
    // one-block launch in front of every minibatch
    (void)ws_adv;
    if constexpr (PF) {  // columns [8 CG, Kp1) of both observation images stay zero for the whole launch (P0 commits the first CG groups only)
        for (int e = threadIdx.x; e < (M * ldx + Kp1 * M) / 8; e += blockDim.x) reinterpret_cast<uint4 *>(Xa)[e] = uint4{0u, 0u, 0u, 0u};
    }
    __shared__ float adv_ms[2];
    if (IS_PI && hp.normalize_advantage && threadIdx.x < 64) {
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
            adv_ms[0] = (float)mean;
            adv_ms[1] = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float amean = (IS_PI && hp.normalize_advantage) ? adv_ms[0] : 0.0f;
    const float astd = (IS_PI && hp.normalize_advantage) ? adv_ms[1] : 1.0f;
    const Net Q = IS_PI ? pi_net(params, L) : vf_net(params, L);
    BfNetPtr W = bf_net_ptr(params, L, IS_PI);
    const f32x4 z4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const bf16_t one_bf = (bf16_t)1.0f;
    const bf16x8 ones8 = bf16x8{one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf};
    f32x4 aW2[KT2][NTW], aW1[KT1A][NTW], aW3[NTW][NT3];
    f32x4 aB1[BV ? 1 : NTW], aB2[BV ? 1 : NTW];  // hidden-layer bias gradients: column sums of dz as ones^T . dz on the MFMA (every row of the tile holds the sum)
    float sB1[NTW], sB2[NTW];  // W8: the same sums per lane (this lane's 16 rows of the column), folded over the four lane groups at the end
    float ab3 = 0.0f, dlsd[2] = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        if (!BV || j == 0) aB1[BV ? 0 : j] = aB2[BV ? 0 : j] = z4;
        sB1[j] = sB2[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < KT1A; i++) aW1[i][j] = z4;
#pragma unroll
        for (int i = 0; i < KT2; i++) aW2[i][j] = z4;
#pragma unroll
        for (int q = 0; q < NT3; q++) aW3[j][q] = z4;
    }
    for (int e = threadIdx.x; e < 2 * H + 32; e += blockDim.x)
        bias[e] = e < H ? Q.b1[e] : (e < 2 * H ? Q.b2[e - H] : (e - 2 * H < NOUT ? Q.b3[e - 2 * H] : 0.0f));
    if (threadIdx.x < SLN * 5) stat_lds[threadIdx.x] = 0.0;
    // ---- weight operands: ring over the two H x H streams, resident small fragments ----
    int nt0l = nt0;  // laundered copy (see the asm in the group loop): keeps the fragment address arithmetic scalar and inside the loop
    auto sload = [&](int s) -> bf16x8 {  // s in [0, SL): position in the per-group stream (compile-time after unrolling)
        if (s < S1) return bf_frag(W.fW1, (nt0l + s % NTW) * (KS1C > 0 ? KS1C : 1) + s / NTW, lane);
        const int u = s - S1;
        if constexpr (MT == 4) {
            // 64-row groups walk each H x H stream in two halves of NTW / 2 column tiles (k-step outer, tile inner within a half): the four
            // row tiles of half the columns are all the accumulators a phase holds at once (32 registers instead of 64)
            constexpr int NH = NTW / 2, PER = KS2 * NH;
            const int v = u % (2 * PER), jh = v / PER, rem = v % PER, j = jh * NH + rem % NH, ks = rem / NH;
            return bf_frag(u < 2 * PER ? W.fW2 : W.bW2, (nt0l + j) * KS2 + ks, lane);
        }
        if (u < KS2 * NTW) return bf_frag(W.fW2, (nt0l + u % NTW) * KS2 + u / NTW, lane);
        const int t = u - KS2 * NTW;
        return bf_frag(W.bW2, (nt0l + t % NTW) * KS2 + t / NTW, lane);
    };
    bf16x8 ring[R], w1r[NTW];  // w1r: layer-1 fragments (D <= 32), re-issued at the end of P5 for the next group
    if constexpr (PASS != 2) {
#pragma unroll
        for (int s = 0; s < R; s++) ring[s] = sload(s);
    }
    if constexpr (KS1C == 1) {
#pragma unroll
        for (int j = 0; j < NTW; j++) w1r[j] = bf_frag(W.fW1, nt0 + j, lane);
    }
    if constexpr (W8 && MT == 4) {  // head fragments -> LDS, once per launch (the group loop's first barrier is ahead of their first use)
        for (int f = wave; f < KS2 * NT3; f += NW) *reinterpret_cast<bf16x8 *>(W3lds + (f * 64 + lane0) * 8) = bf_frag(W.fW3, f, lane0);
    }
    // ---- prefetch registers for the next group's samples ----
    float pm0 = 0.0f, pm1 = 0.0f, pm2 = 0.0f, pm3 = 0.0f, px[NX];
    int64_t poff = -1;
    mrow = wave * (M / NW) + lane0;  // sample row whose metadata this lane gathers (lanes < M/NW of every wave: the Feistel
    mlane = lane0 < M / NW;          // permutation arithmetic is spread over the waves instead of skewing wave 0)
    int32_t noff = -1;  // cached buffer offset of this lane's row in the NEXT group, loaded one phase before fetch_meta needs it
    int32_t noff_n = -1;  // PASS 2: ... and of the group after the next (a whole iteration ahead)
    auto fetch_off = [&](int64_t grp) {  // (cache present) issue only; j beyond the minibatch reads a clamped entry that fetch_meta ignores
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
                if constexpr (PASS != 2) {
                    pm0 = rb.log_probs[poff], pm1 = rb.advantages[poff], pm2 = rb.returns[poff];
                    if constexpr (!CONT) pm3 = __int_as_float(static_cast<const int32_t *>(rb.actions)[poff]);
                }
            }
            row_off_next[mrow] = poff;
        }
    };
    auto fetch_obs = [&]() {  // rows named by row_off_next (visible after a barrier)
        // Issue only: the loaded value is NOT touched here (masking it with `ok` would make the compiler wait for the HBM round trip on
        // the spot); invalid elements load a clamped address and are zeroed at commit time.  Indices come from the per-phase lane copy.
        if constexpr (PF) {
            const int tid = wave * 64 + lane;
#pragma unroll
            for (int rp = 0; rp < RP; rp++) {
                const int64_t off = row_off_next[rp * 8 * NW + (tid >> 3)];
                const int base = (int)off * D;  // (sample offsets are below 2^22, tma_ppo_epoch_prepare's OFFS_CAP: 32-bit index arithmetic)
#pragma unroll
                for (int cg = 0; cg < CG; cg++) {
                    const int c = 8 * cg + (tid & 7);
                    px[rp * CG + cg] = rb.obs[(off >= 0 && c < D) ? base + c : 0];
                }
            }
        }
    };
    const int64_t n_groups = (mb.count + M - 1) / M;
    // PASS 2 (round 5): the observation rows of the NEXT group are requested in front of this group's MFMAs and converted at the top of the next
    // iteration.  No LDS hop for the offsets: lane i < M / NW of a wave holds the buffer offset of row wave * (M / NW) + i (fetch_meta's `poff`) --
    // exactly the rows this wave gathers -- so the wave-uniform row bases are v_readlane's of its own register.
    // PASS 0 at the two-pass widths (round 5, -DTMA_BF_PFD=1, OFF): nobody reads the observation images after layer 1, so the NEXT group's rows
    // can be requested during the group and converted into the (single) images before its end -- the group loop's top then has no gather and one
    // barrier less.  Measured at the Crawler width (gradient call, one box, against 430 us with the PASS 2 form alone): whole rows requested at
    // the start of P6 or of the loss phase: twelve staging registers -> 20 spilled registers in the eight-wave kernel (248 of 256 before), 464 us;
    // one 64-column chunk per phase (loss / P4 / P4 -> P5: four staging registers, 2 spilled): 428 us -- no gain: the 3 k cycles a group spends
    // in P0 are the conversions and the 2-byte image stores as much as the round trip.  Left off; the PASS 2 form above stays.
    constexpr bool PFD = TMA_BF_PFD && !PF && two_pass && PASS == 0 && MT == 2;
    constexpr int RWD = M / NW, KCD = (PASS == 2 || PFD) ? (32 * KS1C + 63) / 64 : 1;
    float tpf[KCD][RWD];
    typedef const float __attribute__((address_space(1))) *gfd_ptr;
    auto pf2_issue = [&]() {
        if constexpr (PASS == 2 || PFD) {
#pragma unroll
            for (int i = 0; i < RWD; i++) {
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)poff, i), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)poff >> 32), i);
                const int64_t offu = (int64_t)(((uint64_t)hi << 32) | lo);
                gfd_ptr base = reinterpret_cast<gfd_ptr>(reinterpret_cast<uintptr_t>(rb.obs + (offu >= 0 ? offu * D : 0)));
#pragma unroll
                for (int k = 0; k < KCD; k++) {
                    const int c = 64 * k + lane0;
                    tpf[k][i] = base[c < D ? c : 0];
                }
            }
        }
    };
    float tch[RWD];  // PFD: one 64-column chunk of the next group's rows at a time (the eight-wave kernel has no twelve registers to spare)
    auto pfd_issue_k = [&](int k) {
        if constexpr (PFD) {
#pragma unroll
            for (int i = 0; i < RWD; i++) {
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)poff, i), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)poff >> 32), i);
                const int64_t offu = (int64_t)(((uint64_t)hi << 32) | lo);
                gfd_ptr base = reinterpret_cast<gfd_ptr>(reinterpret_cast<uintptr_t>(rb.obs + (offu >= 0 ? offu * D : 0)));
                const int c = 64 * k + lane0;
                tch[i] = base[c < D ? c : 0];
            }
        }
    };
    auto pfd_commit_k = [&](int k) {
        if constexpr (PFD) {
            bool rok[RWD];
#pragma unroll
            for (int i = 0; i < RWD; i++) rok[i] = (int32_t)__builtin_amdgcn_readlane((uint32_t)((uint64_t)poff >> 32), i) >= 0;
            const int c = 64 * k + lane0;
            if (c < Kp1) {
                uint32_t pk[RWD / 2];
#pragma unroll
                for (int i = 0; i < RWD; i += 2) {
                    pk[i / 2] = bf_pack2((rok[i] && c < D) ? tch[i] : 0.0f, (rok[i + 1] && c < D) ? tch[i + 1] : 0.0f);
                    *reinterpret_cast<uint16_t *>(Xa + (wave * RWD + i) * ldx + c) = (uint16_t)pk[i / 2];
                    *reinterpret_cast<uint16_t *>(Xa + (wave * RWD + i + 1) * ldx + c) = (uint16_t)(pk[i / 2] >> 16);
                }
                if constexpr (RWD == 8) {
                    *reinterpret_cast<uint4 *>(Xt + t_off<MT>(c, wave * RWD)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                } else {
                    *reinterpret_cast<uint2 *>(Xt + t_off<MT>(c, wave * RWD)) = uint2{pk[0], pk[1]};
                }
            }
        }
    };
    auto pfd_commit = [&]() {  // tpf -> both observation images (the rows named by `poff`); same conversions and stores as the direct gather
        if constexpr (PFD) {
            bool rok[RWD];
#pragma unroll
            for (int i = 0; i < RWD; i++) rok[i] = (int32_t)__builtin_amdgcn_readlane((uint32_t)((uint64_t)poff >> 32), i) >= 0;
#pragma unroll
            for (int k = 0; k < KCD; k++) {
                const int c = 64 * k + lane0;
                if (c < Kp1) {
                    uint32_t pk[RWD / 2];
#pragma unroll
                    for (int i = 0; i < RWD; i += 2) {
                        pk[i / 2] = bf_pack2((rok[i] && c < D) ? tpf[k][i] : 0.0f, (rok[i + 1] && c < D) ? tpf[k][i + 1] : 0.0f);
                        *reinterpret_cast<uint16_t *>(Xa + (wave * RWD + i) * ldx + c) = (uint16_t)pk[i / 2];
                        *reinterpret_cast<uint16_t *>(Xa + (wave * RWD + i + 1) * ldx + c) = (uint16_t)(pk[i / 2] >> 16);
                    }
                    if constexpr (RWD == 8) {
                        *reinterpret_cast<uint4 *>(Xt + t_off<MT>(c, wave * RWD)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                    } else {
                        *reinterpret_cast<uint2 *>(Xt + t_off<MT>(c, wave * RWD)) = uint2{pk[0], pk[1]};
                    }
                }
            }
        }
    };
    bf16x8 zpf[PASS == 2 ? NTW : 1][PASS == 2 ? MK : 1];
    auto zc_issue = [&](int64_t grp_n) {
        if constexpr (PASS == 2) {
            const bf16_t *gi = dz1c + grp_n * (int64_t)(H * M);
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int kk = 0; kk < MK; kk++) {
                    const int row = n_base + 16 * j + (lane0 & 15);
                    zpf[j][kk] = *reinterpret_cast<const bf16x8 *>(gi + row * (16 * MT) + 8 * ((4 * kk + (lane0 >> 4)) ^ t_swz<MT>(row)));
                }
        }
    };
    if (block_net < n_groups) {
        fetch_meta(block_net, false);
        __syncthreads();
        fetch_obs();
        pf2_issue();
        pfd_commit();  // (PFD: the first group's rows, converted here; the loop's first barrier publishes them)
        zc_issue(block_net);
        if constexpr (PASS == 2) {
            if (mb.offs && mlane) {
                const int64_t j = (block_net + (int64_t)n_blocks_net) * M + mrow;
                noff_n = mb.offs[j < mb.count ? j : 0];
            }
        }
    }
    for (int64_t grp = block_net; grp < n_groups; grp += n_blocks_net) {
        // The weight images do not change during the launch, so every fragment load below is loop-invariant and LICM would
        // hoist all of them out of the group loop -- i.e. try to keep ~1 KiB per lane of weights "in registers" and spill
        // them to scratch.  Laundering the (uniform) image pointers once per iteration keeps the loads where they are written.
        // (The wave's tile index is laundered too: otherwise the invariant per-fragment offsets are hoisted as 64-bit VGPR pairs.)
        W.fW1 = launder_uniform(W.fW1), W.fW2 = launder_uniform(W.fW2), W.bW2 = launder_uniform(W.bW2), W.fW3 = launder_uniform(W.fW3);
        W.bW3 = launder_uniform(W.bW3), nt0l = launder_uniform(nt0l);
        TMA_TICK(0);
        TMA_RELANE();
        bf16x8 zc[NTW][MK];  // PASS 2: this wave's dz1 fragments of the group (requested a group ago, like the observation rows)
        if constexpr (PASS == 2) {
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int kk = 0; kk < MK; kk++) zc[j][kk] = zpf[j][kk];
            // the next group's offsets were requested an iteration ago (fetch_meta below uses them at once: no wait in front of the row requests);
            // the group after that is requested now
            noff = noff_n;
            if (mb.offs && mlane) {
                const int64_t j = (grp + 2 * (int64_t)n_blocks_net) * M + mrow;
                noff_n = mb.offs[j < mb.count ? j : 0];
            }
        }
        // ---- P0: commit the prefetched metadata / observation rows (bf16, both images) ----
        if (mlane) {
            meta[mrow * 4 + 0] = pm0, meta[mrow * 4 + 1] = pm1, meta[mrow * 4 + 2] = pm2, meta[mrow * 4 + 3] = pm3;
            row_off[mrow] = poff;
        }
        if constexpr (PF) {
            const int tid = wave * 64 + lane;
#pragma unroll
            for (int rp = 0; rp < RP; rp++) {
                const int row = rp * 8 * NW + (tid >> 3);
                const bool okr = row_off_next[row] >= 0;  // row_off_next still names THIS group's rows (next fetch_meta: end of P2)
#pragma unroll
                for (int cg = 0; cg < CG; cg++) {
                    const int c = 8 * cg + (tid & 7);
                    const bf16_t v = (bf16_t)((okr && c < D) ? px[rp * CG + cg] : 0.0f);
                    Xa[row * ldx + c] = v;
                    Xt[t_off<MT>(c, row)] = v;
                }
            }
        } else if constexpr (PFD) {
            // (the images were filled at the end of the previous group's P6 / in the prologue)
        } else if constexpr (PASS == 2) {
            // commit of the rows pf2_issue requested a group ago (only the T image: this pass reads nothing else)
            static_assert(MT == 2, "PASS 2 commit: RWD rows of one column are one (half) chunk of the T image");
            bool rok[RWD];
#pragma unroll
            for (int i = 0; i < RWD; i++) rok[i] = (int32_t)__builtin_amdgcn_readlane((uint32_t)((uint64_t)poff >> 32), i) >= 0;
#pragma unroll
            for (int k = 0; k < KCD; k++) {
                const int c = 64 * k + lane;
                if (c < Kp1) {
                    uint32_t pk[RWD / 2];
#pragma unroll
                    for (int i = 0; i < RWD; i += 2) pk[i / 2] = bf_pack2((rok[i] && c < D) ? tpf[k][i] : 0.0f, (rok[i + 1] && c < D) ? tpf[k][i + 1] : 0.0f);
                    if constexpr (RWD == 8) {
                        *reinterpret_cast<uint4 *>(Xt + t_off<MT>(c, wave * RWD)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                    } else {
                        *reinterpret_cast<uint2 *>(Xt + t_off<MT>(c, wave * RWD)) = uint2{pk[0], pk[1]};
                    }
                }
            }
        } else {
            __syncthreads();
            // Direct gather (no cross-phase prefetch registers at this width).  Wave w takes rows 8w .. 8w+7 (MT = 2), lane l the
            // columns l, l + 64, ...: the row's buffer offset is wave-uniform (scalar base + lane offset: no 64-bit per-lane
            // address arithmetic -- the element-indexed form of this loop was instruction-bound, 9k cycles per group), all loads
            // of a column chunk are in flight before the first is converted, and a thread's 8 rows of one column are exactly
            // one 16-byte chunk of the T image (one ds_write_b128, conflict-free under the chunk swizzle).
            static_assert(MT == 2, "direct gather: 8 rows per wave = one T-image chunk (four waves), 4 rows = half a chunk (eight)");
            constexpr int RW = M / NW;
            typedef const float __attribute__((address_space(1))) *gf_ptr;
            gf_ptr rbase[RW];
            bool rok[RW];
            [[maybe_unused]] float actv[RW];
#pragma unroll
            for (int i = 0; i < RW; i++) {
                const int64_t off = row_off[wave * RW + i];
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)off), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)off >> 32));
                const int64_t offu = (int64_t)(((uint64_t)hi << 32) | lo);
                rok[i] = offu >= 0;
                rbase[i] = reinterpret_cast<gf_ptr>(reinterpret_cast<uintptr_t>(rb.obs + (rok[i] ? offu * D : 0)));
                if constexpr (ACT_TILE) {  // (issue only: stored to LDS behind the observation chunks)
                    gf_ptr ab = reinterpret_cast<gf_ptr>(reinterpret_cast<uintptr_t>(static_cast<const float *>(rb.actions) + (rok[i] ? offu * A : 0)));
                    actv[i] = ab[lane < A ? lane : 0];
                }
            }
            constexpr int KCC = KS1C > 0 ? (32 * KS1C + 63) / 64 : 1;  // compile-time width: every column chunk in one batch
            const int c_end = KS1C > 0 ? 64 * KCC : Kp1;
            for (int c0 = 0; c0 < c_end; c0 += 64 * KCC) {
                float t[KCC][RW];
#pragma unroll
                for (int k = 0; k < KCC; k++) {
                    const int c = c0 + 64 * k + lane;
#pragma unroll
                    for (int i = 0; i < RW; i++) t[k][i] = rbase[i][c < D ? c : 0];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < KCC; k++) {
                    const int c = c0 + 64 * k + lane;
                    if (c < Kp1) {
                        uint32_t pk[RW / 2];  // (one conversion per row pair; the 2-byte stores take the halves: see bfq)
#pragma unroll
                        for (int i = 0; i < RW; i += 2) {
                            pk[i / 2] = bf_pack2((rok[i] && c < D) ? t[k][i] : 0.0f, (rok[i + 1] && c < D) ? t[k][i + 1] : 0.0f);
                            *reinterpret_cast<uint16_t *>(Xa + (wave * RW + i) * ldx + c) = (uint16_t)pk[i / 2];
                            *reinterpret_cast<uint16_t *>(Xa + (wave * RW + i + 1) * ldx + c) = (uint16_t)(pk[i / 2] >> 16);
                        }
                        if constexpr (RW == 8) {
                            *reinterpret_cast<uint4 *>(Xt + t_off<MT>(c, wave * RW)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                        } else {
                            *reinterpret_cast<uint2 *>(Xt + t_off<MT>(c, wave * RW)) = uint2{pk[0], pk[1]};
                        }
                    }
                }
            }
            if constexpr (ACT_TILE) {
                if (lane < 32) {
#pragma unroll
                    for (int i = 0; i < RW; i++) act_tile[(wave * RW + i) * 32 + lane] = (rok[i] && lane < A) ? actv[i] : 0.0f;
                }
            }
        }
        __syncthreads();
        if constexpr (PASS == 2) {
            if (grp + n_blocks_net < n_groups) {
                fetch_meta(grp + n_blocks_net, true);
                pf2_issue();  // the next group's observation rows: in flight under the MFMAs below
                zc_issue(grp + n_blocks_net);
            }
#pragma unroll
            for (int kt = 0; kt < KT1A; kt++)
#pragma unroll
                for (int kk = 0; kk < MK; kk++) {
                    const bf16x8 a = t_frag<MT>(Xt, 16 * kt + r16, kk, g);
#pragma unroll
                    for (int j = 0; j < NTW; j++) aW1[kt][j] = mfma_bf(a, zc[j][kk], aW1[kt][j]);
                }
            __syncthreads();  // the T image of the observations is consumed; row_off_next names the next group's rows
            fetch_obs();
            continue;
        }
#ifdef TMA_BF_PHASE_DEBUG
        const int dbg = hp.debug;  // timing attribution builds only: skipping phases perturbs register allocation
#else
        constexpr int dbg = 0;
#endif
        const bool has_next = grp + n_blocks_net < n_groups && !(dbg & 1);  // block-uniform
        fetch_off(grp + n_blocks_net);  // consumed by fetch_meta at the end of P2
        TMA_TICK(1);
        TMA_RELANE();
        // ---- P1: layer 1 forward ----
        if (!(dbg & 64)) {
            f32x4 acc[NTW][MT];
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const float b = bias[n_base + 16 * j + r16];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[j][mt] = f32x4{b, b, b, b};
            }
            if constexpr (KS1C == 1) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const bf16x8 a = a_frag(Xa, ldx, 16 * mt + r16, 0, g);
#pragma unroll
                    for (int j = 0; j < NTW; j++) acc[j][mt] = mfma_bf(a, w1r[j], acc[j][mt]);
                }
            } else if constexpr (KS1C > 1) {
#pragma unroll
                for (int ks = 0; ks < KS1C; ks++) {
                    bf16x8 a[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) a[mt] = a_frag(Xa, ldx, 16 * mt + r16, ks, g);
#pragma unroll
                    for (int j = 0; j < NTW; j++) {
                        const int s = ks * NTW + j;
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) acc[j][mt] = mfma_bf(a[mt], ring[s % R], acc[j][mt]);
                        ring[s % R] = sload((s + R) % SL);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll 2
                for (int ks = 0; ks < KS1; ks++) {
                    bf16x8 w[NTW];
#pragma unroll
                    for (int j = 0; j < NTW; j++) w[j] = bf_frag(W.fW1, (nt0l + j) * KS1 + ks, lane);
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const bf16x8 a = a_frag(Xa, ldx, 16 * mt + r16, ks, g);
#pragma unroll
                        for (int j = 0; j < NTW; j++) acc[j][mt] = mfma_bf(a, w[j], acc[j][mt]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const int n = n_base + 16 * j + r16;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const bfq q = tanh_quad<!W8>(acc[j][mt]);
                    if constexpr (!bf_tr_reads<MT>()) bfq_store_rows(A1 + (16 * mt + 4 * g) * lda + n, lda, q);
                    bfq_store_quad(t_quad<MT>(T1, n, mt, g), q);
                }
            }
        }
        __syncthreads();
        TMA_TICK(2);
        TMA_RELANE();
        // ---- P2: layer 2 forward through the weight ring ----
        // split-K head over the four waves.  KS2 % 4 == 0 (H = 128 / 256): wave w takes the k-steps of ITS OWN h2 columns
        // (ks = w*HK .. w*HK + HK - 1), so its partial product needs no barrier after the layer-2 epilogue; H = 192: ks = w, w + 4.
        constexpr int HK = (KS2 + 3) / 4;
        constexpr bool BLOCKK = KS2 % 4 == 0;  // bf_head's rule: partial w covers k-steps w*HK .. (else w, w + 4)
        constexpr bool OWNK = BLOCKK && !W8;   // ... which are this wave's own h2 columns when four waves hold 64 columns each: no barrier in front
        auto head_ks = [&](int i) { return BLOCKK ? wave * HK + i : wave + 4 * i; };
        bf16x8 w3f[HK * NT3];
        auto head_partial = [&]() {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                f32x4 part[NT3];
#pragma unroll
                for (int q = 0; q < NT3; q++) part[q] = z4;
#pragma unroll
                for (int i = 0; i < HK; i++) {
                    const int ks = head_ks(i);
                    if (ks < KS2) {
                        const bf16x8 a = act_frag<MT>(A2, lda, T2, mt, ks, lane);
#pragma unroll
                        for (int q = 0; q < NT3; q++) part[q] = mfma_bf(a, w3f[i * NT3 + q], part[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < NT3; q++) *reinterpret_cast<f32x4 *>(hpart + (((wave * MT + mt) * 2 + q) * 64 + lane) * 4) = part[q];
            }
        };
        bf16x8 w3all[(MT == 4 && !W8) ? KS2 * NT3 : 1];  // MT = 4: every head fragment (this wave runs the whole head of its row tile)
        if (!(dbg & 32)) {
            if constexpr (W8 && MT == 4) {
                // (head fragments wait in LDS)
            } else if constexpr (MT == 4) {
#pragma unroll
                for (int ks = 0; ks < KS2; ks++)
#pragma unroll
                    for (int q = 0; q < NT3; q++) w3all[ks * NT3 + q] = bf_frag(W.fW3, q * KS2 + ks, lane);
            } else {
#pragma unroll
                for (int i = 0; i < HK; i++)  // this wave's head fragments: in flight behind the whole layer-2 phase (eight waves: the four head waves')
#pragma unroll
                    for (int q = 0; q < NT3; q++) w3f[i * NT3 + q] = bf_frag(W.fW3, q * KS2 + ((wave < 4 && head_ks(i) < KS2) ? head_ks(i) : 0), lane);
            }
            if constexpr (MT == 4) {
                static_assert(NTW % 2 == 0 && S1 == 0, "64-row groups: even column-tile count, single layer-1 k-step");
                constexpr int NH = NTW / 2;
#pragma unroll
                for (int jh = 0; jh < 2; jh++) {
                    f32x4 acc[NH][MT];
#pragma unroll
                    for (int jj = 0; jj < NH; jj++) {
                        const float b = bias[H + n_base + 16 * (jh * NH + jj) + r16];
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) acc[jj][mt] = f32x4{b, b, b, b};
                    }
                    bf16x8 a[2][2];  // A fragments of two row tiles, half a k-step ahead
                    [[maybe_unused]] TrBase<MT> tb1;
                    if constexpr (bf_tr_reads<MT>()) tb1 = tr_base<MT>(T1, lane);
                    auto afrag1 = [&](int mt, int ks) -> bf16x8 {
                        if constexpr (bf_tr_reads<MT>()) return a_frag_tb<MT>(tb1, mt, ks);
                        else return a_frag(A1, lda, 16 * mt + r16, ks, g);
                    };
#pragma unroll
                    for (int mt = 0; mt < 2; mt++) a[0][mt] = afrag1(mt, 0);
#pragma unroll
                    for (int ks = 0; ks < KS2; ks++)
#pragma unroll
                        for (int hb = 0; hb < 2; hb++) {
                            const int cur = hb, nb_ks = hb ? ks + 1 : ks, nb_h = hb ? 0 : 1;
                            if (nb_ks < KS2) {
#pragma unroll
                                for (int mt = 0; mt < 2; mt++) a[cur ^ 1][mt] = afrag1(nb_h * 2 + mt, nb_ks);
                            }
#pragma unroll
                            for (int jj = 0; jj < NH; jj++) {
                                const int sp = jh * KS2 * NH + ks * NH + jj;
#pragma unroll
                                for (int mt = 0; mt < 2; mt++) acc[jj][hb * 2 + mt] = mfma_bf(a[cur][mt], ring[sp % R], acc[jj][hb * 2 + mt]);
                                if (hb) ring[sp % R] = sload((sp + R) % SL);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    if (jh == 1 && has_next) fetch_meta(grp + n_blocks_net, true);  // (placement: see the 32-row path below)
#pragma unroll
                    for (int jj = 0; jj < NH; jj++) {
                        const int n = n_base + 16 * (jh * NH + jj) + r16;
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) {
                            const bfq q = tanh_quad<!W8>(acc[jj][mt]);
                            if constexpr (!bf_tr_reads<MT>()) bfq_store_rows(A2 + (16 * mt + 4 * g) * lda + n, lda, q);
                            bfq_store_quad(t_quad<MT>(T2, n, mt, g), q);
                        }
                    }
                }
            } else {
            f32x4 acc[NTW][MT];
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const float b = bias[H + n_base + 16 * j + r16];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[j][mt] = f32x4{b, b, b, b};
            }
            // A fragments one k-step ahead (the scheduler is fenced per k-step: the pipeline below is explicit).  64-row groups: half a
            // k-step ahead -- tiles 2, 3 of the step are read while tiles 0, 1 multiply (16 registers instead of 32)
            constexpr int AH = MT == 4 ? 2 : MT;  // row tiles per A-fragment batch
            bf16x8 a[2][AH];
            [[maybe_unused]] TrBase<MT> tb1;
            if constexpr (bf_tr_reads<MT>()) tb1 = tr_base<MT>(T1, lane);
            auto afrag1 = [&](int mt, int ks) -> bf16x8 {
                if constexpr (bf_tr_reads<MT>()) return a_frag_tb<MT>(tb1, mt, ks);
                else return a_frag(A1, lda, 16 * mt + r16, ks, g);
            };
#pragma unroll
            for (int mt = 0; mt < AH; mt++) a[0][mt] = afrag1(mt, 0);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
#pragma unroll
                for (int hb = 0; hb < MT / AH; hb++) {
                    const int cur = (ks * (MT / AH) + hb) & 1, nb_ks = hb + 1 < MT / AH ? ks : ks + 1, nb_h = hb + 1 < MT / AH ? hb + 1 : 0;
                    if (nb_ks < KS2) {
#pragma unroll
                        for (int mt = 0; mt < AH; mt++) a[cur ^ 1][mt] = afrag1(nb_h * AH + mt, nb_ks);
                    }
#pragma unroll
                    for (int j = 0; j < NTW; j++) {
                        const int s = S1 + ks * NTW + j;
#pragma unroll
                        for (int mt = 0; mt < AH; mt++) acc[j][hb * AH + mt] = mfma_bf(a[cur][mt], ring[s % R], acc[j][hb * AH + mt]);
                        if (hb + 1 == MT / AH) ring[s % R] = sload((s + R) % SL);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // Next group's sample gathers (HBM-latency loads) go out HERE and at the top of P3: vmcnt retires in order, so the
            // first wait on a load issued after them also waits for them -- and from here on P3 / P4 / the first half of P5 only
            // consume fragments that are already in flight.
            if (has_next) fetch_meta(grp + n_blocks_net, true);
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const int n = n_base + 16 * j + r16;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const bfq q = tanh_quad<!W8>(acc[j][mt]);
                    if constexpr (!bf_tr_reads<MT>()) bfq_store_rows(A2 + (16 * mt + 4 * g) * lda + n, lda, q);
                    bfq_store_quad(t_quad<MT>(T2, n, mt, g), q);
                }
            }
            }
            if constexpr (OWNK && MT != 4) {  // head partial over this wave's own h2 columns: straight after its epilogue, no barrier in between
                if (!(dbg & 4)) head_partial();
            }
        }
        __syncthreads();
        TMA_TICK(3);
        TMA_RELANE();
        // ---- P3a: (H = 192 only) split-K head partials; otherwise they were produced behind the layer-2 epilogue ----
        bf16x8 w3b[NTW];  // head input-gradient fragments for P4, in flight behind the head
#pragma unroll
        for (int j = 0; j < NTW; j++) w3b[j] = bf_frag(W.bW3, nt0l + j, lane);
        fetch_obs();  // unconditional (a last group re-reads stale rows that are never committed): the load count after w3b stays known,
                      // so P4's wait for w3b is a counted vmcnt, not vmcnt(0) on these HBM gathers
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MT == 4) {
            // ---- P3 (64-row groups): wave w owns row tile w -- whole head (the split-K summation order of bf_head, so the rollout's
            // log-probabilities still match bit for bit), loss, and both bf16 images of dz3, with no block barrier in between ----
            TMA_TICK3(4);
            // Eight waves: waves w and w + 4 share row tile w -- both run its head (8 MFMAs), each takes two of the four rows of every lane
            // group through the loss (the expensive part: exp / log per row), then a barrier, then waves 0-3 build the row-major and
            // waves 4-7 the transposed bf16 image of dz3
            const bool row_loss = RL && (!IS_PI || A <= 8);  // (block-uniform; wider heads keep the C-layout loss on all eight waves)
            if (row_loss) {
                if (!(dbg & 4) && wave < 4) {
                    const int mt = wave;
                    f32x4 part[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) part[w] = z4;
#pragma unroll
                    for (int i = 0; i < HK; i++)
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const int ks = BLOCKK ? w * HK + i : w + 4 * i;
                            if (ks < KS2) part[w] = mfma_bf(act_frag<MT>(A2, lda, T2, mt, ks, lane), *reinterpret_cast<const bf16x8 *>(W3lds + (ks * 64 + lane) * 8), part[w]);
                        }
                    const float bv = bias[2 * H + r16];
                    f32x4 out = f32x4{bv, bv, bv, bv};
#pragma unroll
                    for (int w = 0; w < 4; w++) out += part[w];
                    float *dzt = dz3 + mt * 16 * ld3;
                    // the tile's outputs, row-major in LDS (columns beyond NOUT are exact zeros: zero weights, zero bias -- they are the padding of dz3 too)
#pragma unroll
                    for (int r = 0; r < 4; r++) dzt[(4 * g + r) * ld3 + r16] = out[r];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    TMA_TICK3(10);
                    if (lane < 16) {  // lane = row of the tile
                        const int row = mt * 16 + lane;
                        const bool valid = row_off[row] >= 0;
                        const f32x4 mrow = *reinterpret_cast<const f32x4 *>(meta + row * 4);
                        float *xrow = dzt + lane * ld3;
                        double *sl = MAIN ? stat_lds + (wave * 16 + lane) * 5 : nullptr;
                        if constexpr (IS_PI) {
                            discrete_loss_row<8>(xrow, mrow, valid, A, amean, astd, hp, invB, sl);
                        } else {
                            const float diff = xrow[0] - mrow[2];
                            xrow[0] = valid ? (hp.vf_coef * 2.0f * invB) * diff : 0.0f;
                            if (MAIN && valid) sl[0] += (double)(diff * diff);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    TMA_TICK3(11);
                    {  // dz3 of the tile as bf16, row-major (Z3a) ...
                        bf16x8 v;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int a = 8 * g + j;
                            v[j] = (bf16_t)(a < 16 * NT3 ? dzt[r16 * ld3 + a] : 0.0f);
                        }
                        *reinterpret_cast<bf16x8 *>(Z3a + (16 * mt + r16) * ldz + 8 * g) = v;
                    }
                    {  // ... and transposed (Z3t); the f32 column sums feed the head bias gradient
                        const int a = lane & 31, half = lane >> 5;
                        bf16x8 v;
                        float c = 0.0f;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const float x = a < 16 * NT3 ? dzt[(8 * half + j) * ld3 + a] : 0.0f;
                            c += x;
                            v[j] = (bf16_t)x;
                        }
                        ab3 += c;
                        *reinterpret_cast<bf16x8 *>(Z3t + t_off<MT>(a, 16 * mt + 8 * half)) = v;
                    }
                    TMA_TICK3(12);
                }
            } else
            if (!(dbg & 4)) {
                const int mt = W8 ? (wave & 3) : wave, r_lo = W8 ? 2 * (wave >> 2) : 0, r_hi = W8 ? r_lo + 2 : 4;
                f32x4 part[4][NT3];
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int q = 0; q < NT3; q++) part[w][q] = z4;
#pragma unroll
                for (int i = 0; i < HK; i++)
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const int ks = BLOCKK ? w * HK + i : w + 4 * i;
                        if (ks < KS2) {
                            const bf16x8 a = act_frag<MT>(A2, lda, T2, mt, ks, lane);
#pragma unroll
                            for (int q = 0; q < NT3; q++)
                                part[w][q] = mfma_bf(a, W8 ? *reinterpret_cast<const bf16x8 *>(W3lds + ((q * KS2 + ks) * 64 + lane) * 8) : w3all[ks * NT3 + q], part[w][q]);
                        }
                    }
                f32x4 out[NT3];
#pragma unroll
                for (int q = 0; q < NT3; q++) {
                    const float b = bias[2 * H + 16 * q + r16];
                    out[q] = f32x4{b, b, b, b};
#pragma unroll
                    for (int w = 0; w < 4; w++) out[q] += part[w][q];
                }
                TMA_TICK(10);
                LossStats st;
                float *dzt = dz3 + mt * 16 * ld3;
                if constexpr (IS_PI) {
#define TMA_LOSS_ARGS out, meta + mt * 64, row_off + mt * 16, rb.actions, params + L.log_std, A, amean, astd, hp, invB, dzt, ld3, dlsd, st, lane
                    if constexpr (W8) {  // rows 0, 1 (waves 0-3) or 2, 3 (waves 4-7) of every lane group: a compile-time range per branch
                        if (wave < 4) policy_loss_tile<CONT, 0, 2>(TMA_LOSS_ARGS);
                        else policy_loss_tile<CONT, 2, 4>(TMA_LOSS_ARGS);
                    } else {
                        policy_loss_tile<CONT, 0, 4>(TMA_LOSS_ARGS);
                    }
#undef TMA_LOSS_ARGS
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (r < r_lo || r >= r_hi) continue;
                        const int row = g * 4 + r;
                        const bool valid = row_off[mt * 16 + row] >= 0;
                        const float diff = out[0][r] - meta[(mt * 16 + row) * 4 + 2];
                        dzt[row * ld3 + r16] = (valid && r16 == 0) ? (hp.vf_coef * 2.0f * invB) * diff : 0.0f;
                        if (valid && r16 == 0) st.a += (double)(diff * diff);
                    }
                }
                if (MAIN && r16 == 0) {
                    double *sl = stat_lds + (wave * 4 + g) * 5;
                    sl[0] += st.a, sl[1] += st.ent, sl[2] += st.kl, sl[3] += st.clip, sl[4] += st.n;
                }
                TMA_TICK(11);
                if constexpr (W8) __syncthreads();  // (block-uniform: dbg is) both halves of every tile's dz3 are in LDS
                TMA_TICK(12);
                // dz3 of this tile as bf16, row-major (Z3a) and transposed (Z3t); the f32 column sums feed the head bias gradient.
                // (four waves: dzt was written by this wave -- LDS operations of one wave execute in order, no barrier needed)
                if (!W8 || wave < 4) {
                    bf16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int a = 8 * g + j;
                        v[j] = (bf16_t)(a < 16 * NT3 ? dzt[r16 * ld3 + a] : 0.0f);
                    }
                    *reinterpret_cast<bf16x8 *>(Z3a + (16 * mt + r16) * ldz + 8 * g) = v;
                }
                if (!W8 || wave >= 4) {
                    const int a = lane & 31, half = lane >> 5;
                    bf16x8 v;
                    float c = 0.0f;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float x = a < 16 * NT3 ? dzt[(8 * half + j) * ld3 + a] : 0.0f;
                        c += x;
                        v[j] = (bf16_t)x;
                    }
                    ab3 += c;
                    *reinterpret_cast<bf16x8 *>(Z3t + t_off<MT>(a, 16 * mt + 8 * half)) = v;
                }
            }
        } else {
        if constexpr (!OWNK) {
            if (!(dbg & 4) && (!W8 || wave < 4)) head_partial();
            __syncthreads();
        }
        TMA_TICK(4);
        TMA_RELANE();
        if constexpr (PFD) {
            if (has_next) pfd_issue_k(0);  // `poff` names the next group's rows since the end of P2
        }
        // ---- P3b: loss -- every wave takes half the rows of one tile (tile wave & 1, rows 2*(wave >> 1) .. +1 of each lane group) ----
        static_assert(MT == 2 || MT == 4, "the loss / dz3 split below assumes two row tiles and four waves");
        if (!(dbg & 4)) {
            const int mt = wave & 1, r_lo = W8 ? (wave >> 1) : 2 * (wave >> 1), r_n = W8 ? 1 : 2;  // (eight waves: one row of every lane group each)
            f32x4 out[NT3];
#pragma unroll
            for (int q = 0; q < NT3; q++) {
                const float b = bias[2 * H + 16 * q + r16];
                out[q] = f32x4{b, b, b, b};
#pragma unroll
                for (int w = 0; w < 4; w++) out[q] += *reinterpret_cast<const f32x4 *>(hpart + (((w * MT + mt) * 2 + q) * 64 + lane) * 4);
            }
            LossStats st;
            float *dzt = dz3 + mt * 16 * ld3;
            if constexpr (IS_PI) {
                policy_loss_tile<CONT>(out, meta + mt * 64, row_off + mt * 16, rb.actions, params + L.log_std, A, amean, astd, hp, invB, dzt, ld3, dlsd, st,
                                       lane, r_lo, r_lo + r_n, ACT_TILE ? act_tile + mt * 16 * 32 : nullptr);
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
            if (MAIN && r16 == 0) {  // only these lanes carry statistics (policy_loss_tile / the value branch accumulate under r16 == 0)
                double *sl = stat_lds + (wave * 4 + g) * 5;
                sl[0] += st.a, sl[1] += st.ent, sl[2] += st.kl, sl[3] += st.clip, sl[4] += st.n;
            }
        }
        __syncthreads();
        // ---- P3c: dz3 as bf16 in both layouts: waves 0 / 1 write Z3a of tile 0 / 1, waves 2 / 3 write Z3t (+ head bias sums) ----
        TMA_RELANE();
        if (!(dbg & 4)) {
            const int mt = wave & 1;
            const float *dzt = dz3 + mt * 16 * ld3;
            if (wave < 2) {  // Z3a[m][a]: lane = (row, 8-column chunk)
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int a = 8 * g + j;
                    v[j] = (bf16_t)(a < 16 * NT3 ? dzt[r16 * ld3 + a] : 0.0f);
                }
                *reinterpret_cast<bf16x8 *>(Z3a + (16 * mt + r16) * ldz + 8 * g) = v;
            } else if (!W8 || wave < 4) {  // Z3t[a][m]: lane = (a, 8-sample half of the tile); the f32 column sum feeds the head bias gradient
                const int a = lane & 31, half = lane >> 5;
                bf16x8 v;
                float c = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float x = a < 16 * NT3 ? dzt[(8 * half + j) * ld3 + a] : 0.0f;
                    c += x;
                    v[j] = (bf16_t)x;
                }
                ab3 += c;
                *reinterpret_cast<bf16x8 *>(Z3t + t_off<MT>(a, 16 * mt + 8 * half)) = v;
            }
        }
        }
        __syncthreads();
        TMA_TICK3(5);
        TMA_RELANE();
        if constexpr (PFD) {
            if (has_next) {  // (nobody reads the observation images after layer 1 in this pass)
                pfd_commit_k(0);
                if constexpr (KCD > 1) pfd_issue_k(1);
            }
        }
        // ---- P4: head weight gradient (this wave's k rows); dz2 = (dz3 . W3^T) * (1 - h2^2) in place in A2 / T2 ----
        if (!(dbg & 16)) {
#pragma unroll
            for (int kk = 0; kk < MK; kk++) {
                bf16x8 zb[NT3];
#pragma unroll
                for (int q = 0; q < NT3; q++) zb[q] = t_frag<MT>(Z3t, 16 * q + r16, kk, g);
#pragma unroll
                for (int i = 0; i < NTW; i++) {
                    const bf16x8 a = t_frag<MT>(T2, n_base + 16 * i + r16, kk, g);
#pragma unroll
                    for (int q = 0; q < NT3; q++) aW3[i][q] = mfma_bf(a, zb[q], aW3[i][q]);
                }
            }
            bf16x8 za[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) za[mt] = *reinterpret_cast<const bf16x8 *>(Z3a + (16 * mt + r16) * ldz + 8 * g);
#pragma unroll
            for (int j = 0; j < NTW; j++) {
                const int n = n_base + 16 * j + r16;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 dh = mfma_bf(za[mt], w3b[j], z4);
                    bf16x4 *tq = t_quad<MT>(T2, n, mt, g);
                    const bfq q = delta_quad<!W8>(dh, *reinterpret_cast<const uint2 *>(tq));
                    if constexpr (!bf_tr_reads<MT>()) bfq_store_rows(A2 + (16 * mt + 4 * g) * lda + n, lda, q);
                    if constexpr (BV && MAIN) sB2[j] = bfq_sum(sB2[j], q);  // (the bf16 deltas the weight gradient uses)
                    bfq_store_quad(tq, q);
                }
            }
        }
        if constexpr (PFD && KCD > 1) {
            if (has_next) {
                pfd_commit_k(1);
                if constexpr (KCD > 2) pfd_issue_k(2);
            }
        }
        __syncthreads();
        TMA_TICK(6);
        TMA_RELANE();
        if constexpr (PFD && KCD > 2) {
            if (has_next) pfd_commit_k(2);
        }
        // ---- P5: dW2 slice += h1^T . dz2[:, slice];  dh1 = dz2 . W2^T for this wave's columns (weight ring) ----
        f32x4 dh1[MT == 4 ? 1 : NTW][MT];  // (64-row groups form dh1 half by half further down)
        if constexpr (MT != 4) {
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) dh1[j][mt] = z4;
        }
        if (!(dbg & 2)) {
            bf16x8 zb[NTW][MK];
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int kk = 0; kk < MK; kk++) zb[j][kk] = t_frag<MT>(T2, n_base + 16 * j + r16, kk, g);
            if constexpr (MAIN && !BV) {  // db2 += ones^T . dz2 (the bf16 deltas the weight gradient uses, f32 accumulate)
#pragma unroll
                for (int j = 0; j < NTW; j++)
#pragma unroll
                    for (int kk = 0; kk < MK; kk++) aB2[j] = mfma_bf(ones8, zb[j][kk], aB2[j]);
            }
            constexpr int TA = MT == 4 ? 2 : 4, TAH = TA - 1;  // T1 fragments TAH k-tiles ahead (64-row groups: one, the k-tile is twice the work)
            bf16x8 ta[TA][MK];
#pragma unroll
            for (int kt = 0; kt < TAH; kt++)
#pragma unroll
                for (int kk = 0; kk < MK; kk++) ta[kt][kk] = t_frag<MT>(T1, 16 * kt + r16, kk, g);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < KT2; kt++) {
                if (kt + TAH < KT2) {
#pragma unroll
                    for (int kk = 0; kk < MK; kk++) ta[(kt + TAH) % TA][kk] = t_frag<MT>(T1, 16 * (kt + TAH) + r16, kk, g);
                }
#pragma unroll
                for (int kk = 0; kk < MK; kk++)
#pragma unroll
                    for (int j = 0; j < NTW; j++) aW2[kt][j] = mfma_bf(ta[kt % TA][kk], zb[j][kk], aW2[kt][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MT == 4) {
                if constexpr (KS1C == 1) {  // next group's layer-1 fragments
#pragma unroll
                    for (int j = 0; j < NTW; j++) w1r[j] = bf_frag(W.fW1, nt0l + j, lane);
                }
            }
        }
        if constexpr (MT == 4) {
            // 64-row groups: the barrier sits between dW2 (which reads every row of T1) and dh1, so that each half of dh1 can be turned into
            // dz1 -- in place in this wave's own rows of T1 -- as soon as it is complete, and only 32 of its registers are live at a time
            __syncthreads();
            TMA_TICK(7);
            TMA_RELANE();
            constexpr int NH = NTW / 2;
            if (!(dbg & 2)) {
#pragma unroll
                for (int jh = 0; jh < 2; jh++) {
                    f32x4 dh[NH][MT];
#pragma unroll
                    for (int jj = 0; jj < NH; jj++)
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) dh[jj][mt] = z4;
                    bf16x8 a[2][2];
                    [[maybe_unused]] TrBase<MT> tb2;
                    if constexpr (bf_tr_reads<MT>()) tb2 = tr_base<MT>(T2, lane);
                    auto afrag2 = [&](int mt, int ks) -> bf16x8 {
                        if constexpr (bf_tr_reads<MT>()) return a_frag_tb<MT>(tb2, mt, ks);
                        else return a_frag(A2, lda, 16 * mt + r16, ks, g);
                    };
#pragma unroll
                    for (int mt = 0; mt < 2; mt++) a[0][mt] = afrag2(mt, 0);
#pragma unroll
                    for (int ns = 0; ns < KS2; ns++)
#pragma unroll
                        for (int hb = 0; hb < 2; hb++) {
                            const int cur = hb, nb_ns = hb ? ns + 1 : ns, nb_h = hb ? 0 : 1;
                            if (nb_ns < KS2) {
#pragma unroll
                                for (int mt = 0; mt < 2; mt++) a[cur ^ 1][mt] = afrag2(nb_h * 2 + mt, nb_ns);
                            }
#pragma unroll
                            for (int jj = 0; jj < NH; jj++) {
                                const int sp = KS2 * NTW + jh * KS2 * NH + ns * NH + jj;
#pragma unroll
                                for (int mt = 0; mt < 2; mt++) dh[jj][hb * 2 + mt] = mfma_bf(a[cur][mt], ring[sp % R], dh[jj][hb * 2 + mt]);
                                if (hb) ring[sp % R] = sload((sp + R) % SL);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                    for (int jj = 0; jj < NH; jj++) {
                        const int j = jh * NH + jj, n = n_base + 16 * j + r16;
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) {
                            bf16x4 *tq = t_quad<MT>(T1, n, mt, g);
                            const bfq q = delta_quad<!W8>(dh[jj][mt], *reinterpret_cast<const uint2 *>(tq));
                            if constexpr (BV && MAIN) sB1[j] = bfq_sum(sB1[j], q);
                            bfq_store_quad(tq, q);
                        }
                    }
                }
            }
        } else {
        if (!(dbg & 2)) {
            TMA_RELANE();
            constexpr int AH = MT == 4 ? 2 : MT;
            bf16x8 a[2][AH];
            [[maybe_unused]] TrBase<MT> tb2;
            if constexpr (bf_tr_reads<MT>()) tb2 = tr_base<MT>(T2, lane);
            auto afrag2 = [&](int mt, int ks) -> bf16x8 {
                if constexpr (bf_tr_reads<MT>()) return a_frag_tb<MT>(tb2, mt, ks);
                else return a_frag(A2, lda, 16 * mt + r16, ks, g);
            };
#pragma unroll
            for (int mt = 0; mt < AH; mt++) a[0][mt] = afrag2(mt, 0);
#pragma unroll
            for (int ns = 0; ns < KS2; ns++) {
#pragma unroll
                for (int hb = 0; hb < MT / AH; hb++) {
                    const int cur = (ns * (MT / AH) + hb) & 1, nb_ns = hb + 1 < MT / AH ? ns : ns + 1, nb_h = hb + 1 < MT / AH ? hb + 1 : 0;
                    if (nb_ns < KS2) {
#pragma unroll
                        for (int mt = 0; mt < AH; mt++) a[cur ^ 1][mt] = afrag2(nb_h * AH + mt, nb_ns);
                    }
#pragma unroll
                    for (int j = 0; j < NTW; j++) {
                        const int s = S1 + KS2 * NTW + ns * NTW + j;
#pragma unroll
                        for (int mt = 0; mt < AH; mt++) dh1[j][hb * AH + mt] = mfma_bf(a[cur][mt], ring[s % R], dh1[j][hb * AH + mt]);
                        if (hb + 1 == MT / AH) ring[s % R] = sload((s + R) % SL);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (KS1C == 1) {  // next group's layer-1 fragments: P6 + P0 of latency cover
#pragma unroll
                for (int j = 0; j < NTW; j++) w1r[j] = bf_frag(W.fW1, nt0l + j, lane);
            }
        }
        __syncthreads();  // every wave is done with T1 (all rows) and A2
        TMA_TICK(7);
        TMA_RELANE();
        // ---- P6: dz1 = dh1 * (1 - h1^2) in place in T1 (own rows);  dW1 slice += X^T . dz1[:, slice] ----
        if (!(dbg & 8))
#pragma unroll
        for (int j = 0; j < NTW; j++) {
            const int n = n_base + 16 * j + r16;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                bf16x4 *tq = t_quad<MT>(T1, n, mt, g);
                const bfq q = delta_quad<!W8>(dh1[j][mt], *reinterpret_cast<const uint2 *>(tq));
                if constexpr (BV && MAIN) sB1[j] = bfq_sum(sB1[j], q);
                bfq_store_quad(tq, q);
            }
        }
        }
        {
            bf16x8 zb[NTW][MK];
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int kk = 0; kk < MK; kk++) zb[j][kk] = t_frag<MT>(T1, n_base + 16 * j + r16, kk, g);
            if constexpr (MAIN && !BV) {  // db1 += ones^T . dz1
#pragma unroll
                for (int j = 0; j < NTW; j++)
#pragma unroll
                    for (int kk = 0; kk < MK; kk++) aB1[j] = mfma_bf(ones8, zb[j][kk], aB1[j]);
            }
            if constexpr (two_pass && MAIN) {
                if (dz1c) {  // (block-uniform) leave this group's dz1 image for PASS 2: 16 lanes x 64 B rows = 1 KiB contiguous per store
                    bf16_t *gi = dz1c + grp * (int64_t)(H * M);
#pragma unroll
                    for (int j = 0; j < NTW; j++)
#pragma unroll
                        for (int kk = 0; kk < MK; kk++) {
                            const int row = n_base + 16 * j + r16;
                            *reinterpret_cast<bf16x8 *>(gi + row * (16 * MT) + 8 * ((4 * kk + g) ^ t_swz<MT>(row))) = zb[j][kk];
                        }
                }
            }
            if constexpr (acc_w1) {
#pragma unroll
                for (int kt = 0; kt < KT1A; kt++)
#pragma unroll
                    for (int kk = 0; kk < MK; kk++) {
                        const bf16x8 a = t_frag<MT>(Xt, 16 * kt + r16, kk, g);
#pragma unroll
                        for (int j = 0; j < NTW; j++) aW1[kt][j] = mfma_bf(a, zb[j][kk], aW1[kt][j]);
                    }
            } else if constexpr (rmw_w1) {
                float *gW1 = slab + (IS_PI ? L.pW1t : L.vW1t);
                for (int kt = 0; kt < KT1; kt++) {
                    f32x4 t[NTW];
#pragma unroll
                    for (int j = 0; j < NTW; j++) t[j] = z4;
#pragma unroll
                    for (int kk = 0; kk < MK; kk++) {
                        const bf16x8 a = t_frag<MT>(Xt, 16 * kt + r16, kk, g);
#pragma unroll
                        for (int j = 0; j < NTW; j++) t[j] = mfma_bf(a, zb[j][kk], t[j]);
                    }
#pragma unroll
                    for (int j = 0; j < NTW; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int k = kt * 16 + g * 4 + r;
                            if (k < D) gW1[(int64_t)k * H + n_base + 16 * j + r16] += t[j][r];  // this wave is the only writer of these slab columns
                        }
                }
            }
        }
        __syncthreads();
        TMA_TICK(9);
    }
    TMA_TICK(8);
#ifdef TMA_BF_PHASE_TICKS
    if (tick_on)
        for (int i = 0; i < 16; i++) atomicAdd(&g_bf_ticks[IS_PI ? 0 : 1][i], tick_lds[i]);
#endif
#undef TMA_RELANE
#undef TMA_TICK
#undef TMA_TICK3
#undef TMA_TICK_DO
    // ---- store this block's slab (every parameter of the net has exactly one owning wave) ----
    float *gW1 = slab + (IS_PI ? L.pW1t : L.vW1t), *gb1 = slab + (IS_PI ? L.pb1 : L.vb1);
    float *gW2 = slab + (IS_PI ? L.pW2t : L.vW2t), *gb2 = slab + (IS_PI ? L.pb2 : L.vb2);
    float *gW3 = slab + (IS_PI ? L.pW3t : L.vW3t), *gb3 = slab + (IS_PI ? L.pb3 : L.vb3);
#pragma unroll
    for (int j = 0; j < NTW; j++) {
        const int col = n_base + 16 * j + r16;
        if constexpr (MAIN) {
#pragma unroll
            for (int kt = 0; kt < KT2; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) gW2[(int64_t)(kt * 16 + g * 4 + r) * H + col] = aW2[kt][j][r];
        }
        if constexpr (acc_w1) {
#pragma unroll
            for (int kt = 0; kt < KT1A; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = kt * 16 + g * 4 + r;
                    if (k < D) gW1[(int64_t)k * H + col] = aW1[kt][j][r];
                }
        }
        if constexpr (MAIN) {
            if constexpr (BV) {  // fold the four lane groups' partial sums (rows 4 g .. 4 g + 3 of every row tile) in a fixed order
                float v1 = sB1[j], v2 = sB2[j];
                v1 += __shfl_xor(v1, 16, 64), v1 += __shfl_xor(v1, 32, 64);
                v2 += __shfl_xor(v2, 16, 64), v2 += __shfl_xor(v2, 32, 64);
                if (g == 0) gb1[col] = v1, gb2[col] = v2;
            } else {
                if (g == 0) gb1[col] = aB1[j][0], gb2[col] = aB2[j][0];  // (rows of the ones^T . dz tiles are identical)
            }
#pragma unroll
            for (int q = 0; q < NT3; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = n_base + 16 * j + g * 4 + r, n = 16 * q + r16;
                    if (n < NOUT) gW3[(int64_t)k * NOUT + n] = aW3[j][q][r];
                }
        }
    }
    // head bias gradient / log_std gradient / loss statistics: partial sums of the MT head waves folded through LDS
    __syncthreads();
    if constexpr (MAIN) {
        float v = ab3;
        v += __shfl_xor(v, 32, 64);  // lanes a and a + 32 hold the two halves of a tile's column sum
        float v0 = dlsd[0], v1 = dlsd[1];
        v0 += __shfl_xor(v0, 16, 64), v0 += __shfl_xor(v0, 32, 64);
        v1 += __shfl_xor(v1, 16, 64), v1 += __shfl_xor(v1, 32, 64);
        if constexpr (MT == 4) {  // every (head) wave holds the column sums of its row tile
            if (lane < 32 && (W8 ? ((RL && (!IS_PI || A <= 8)) ? wave < 4 : wave >= 4) : true)) scratch[(wave & 3) * 32 + lane] = v;
        } else if (wave >= 2 && wave < 4) {  // the Z3t waves hold the head-bias column sums (tile wave - 2)
            if (lane < 32) scratch[(wave - 2) * 32 + lane] = v;
        }
        __syncthreads();
        if (wave == 0 && lane < 32) {
            float s = scratch[lane] + scratch[32 + lane];
            if constexpr (MT == 4) s = (s + scratch[64 + lane]) + scratch[96 + lane];
            if (lane < NOUT) gb3[lane] = s;
        }
        __syncthreads();
        if constexpr (IS_PI && CONT) {  // every wave ran the loss on half the rows of a tile
            if (g == 0) scratch[wave * 32 + r16] = v0, scratch[wave * 32 + 16 + r16] = v1;
            __syncthreads();
            if (wave == 0 && lane < 32) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; w++) s += scratch[w * 32 + lane];  // (eight waves: rows 4 .. 7 lie in the dead head-partial area behind the scratch)
                if (lane < A) slab[L.log_std + lane] = s;
            }
        }
    }
    __syncthreads();
    if (MAIN && threadIdx.x < 5) {
        double ssum = 0.0;
        for (int w = 0; w < SLN; w++) ssum += stat_lds[w * 5 + threadIdx.x];
        const int q = IS_PI ? (threadIdx.x == 0 ? 0 : threadIdx.x + 1) : (threadIdx.x == 0 ? 1 : -1);
        if (q >= 0) stat_slot[q] += ssum;
    }
}

template <bool CONT, int NTW, int MT, int KT1C, int KS1C, int PASS, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW / 4) void ppo_grad_wide_bf_kernel(const float *__restrict__ params, PLayout L, Rollout rb, Minibatch mb, HParams hp,
                                                                  const float *__restrict__ ws_adv, float *__restrict__ slabs,
                                                                  double *__restrict__ stat_slots, int n_pi, bf16_t *__restrict__ dz1,
                                                                  int64_t dz1_net_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem_bf[];
    // blocks [0, n_pi): policy net, [n_pi, gridDim.x): value net.  Block b of a net owns slab b (its net's parameters only) and
    // statistics slot b (policy: entries 0, 2.. ; value: entry 1 -- disjoint, so a policy and a value block may share a slot).
    const bool is_pi = (int)blockIdx.x < n_pi;
    const int b = is_pi ? blockIdx.x : blockIdx.x - n_pi, nb = is_pi ? n_pi : (int)gridDim.x - n_pi;
    float *slab = slabs + (int64_t)b * L.P;
    double *slot = stat_slots + (int64_t)b * 8;
    if (is_pi) grad_wide_bf_body<CONT, true, NTW, MT, KT1C, KS1C, PASS, NW>(params, L, rb, mb, hp, ws_adv, slab, slot, smem_bf, nb, b, dz1);
    else grad_wide_bf_body<CONT, false, NTW, MT, KT1C, KS1C, PASS, NW>(params, L, rb, mb, hp, ws_adv, slab, slot, smem_bf, nb, b,
                                                                       dz1 ? dz1 + dz1_net_stride : nullptr);
}

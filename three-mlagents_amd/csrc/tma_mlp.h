// tma_mlp.h -- wavefront-level actor-critic MLP building blocks on the gfx950 f32-input MFMA
// (v_mfma_f32_16x16x4_f32: exact f32, bitwise a k-ordered fmaf chain).
//
// Unit of work: ONE wavefront owns a tile of 16 samples and carries it through every layer.  Activations of the
// tile live in the wave's private LDS region ([16][H+2] f32, +2 floats of padding = conflict-free A-fragment reads);
// weights are read straight from HBM/L2 as B fragments (every wave reads the same <= 0.9 MB, so they stay L2/L1
// resident) in the [in][out] layout for the forward pass and the [out][in] copy for the input-gradient pass, so that
// 16 consecutive lanes always touch 64 contiguous bytes.
//
// Fragment maps (cdna_hip_programming.md §3): A: lane l holds A[row=l&15][k=l>>4];  B: B[k=l>>4][col=l&15];
// C/D: reg r of lane l is C[row=(l>>4)*4+r][col=l&15].
//
// Mirrors stable-baselines3 2.9.0 ActorCriticPolicy(MlpPolicy) with net_arch=dict(pi=[H,H], vf=[H,H]), tanh
// (third-party; built by PPO("MlpPolicy", ...) at /root/reference/backend/mlagents/training.py:150 with the
// policy_kwargs of training.py:363-365).  SURVEY.md Appendix C.3.
#pragma once
#include "tma_common.h"

namespace tma {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units (5 instructions instead of ocml's ~40: v_mul, v_exp, v_add, v_rcp and
// one explicit v_fma -- the library is built with -ffp-contract=off, which would split 1 - 2r into a multiply and a subtract).
// Absolute error <= ~2e-7 over the whole range (saturates cleanly to +-1); the parity tolerance is 1e-5.
__device__ __forceinline__ float tma_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // exp(2x) = 2^(2x log2 e)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// Flat parameter buffer (floats).  [0, P) is the trainable region in [in][out] ("t") layout; [P, total) holds the
// [out][in] copies of the matrices whose input gradient is needed (layer 2 and the heads).
struct PLayout {
    int D, H, A, cont;
    int pW1t, pb1, pW2t, pb2, pW3t, pb3, vW1t, vb1, vW2t, vb2, vW3t, vb3, log_std, P;
    int pW2, pW3, vW2, vW3;
    int img_pi, img_vf;  // LDS-image regions (H == 64 fast path), IMG_FLOATS each; -1 when the shape has no fast path
    int bf16;            // 1: the hidden-layer GEMMs run on the bf16 MFMA (f32 master weights, f32 accumulate) -- tma_wide_bf16.h
    int bf_pi, bf_vf;    // float offsets of the bf16 fragment-major weight images of the two nets; -1 in f32 mode
    int fr_pi, fr_vf;    // f32 mode, H = 128 / 192 / 256: fragment-major f32 images of W2 (forward, then input-gradient), 2*H*H floats per
                         // net: element i of lane l in fragment (tile t, group q) is the B operand of k-step 4q + i -- forward
                         // W2[k = 16q + 4i + (l>>4)][n = 16t + (l&15)], input-gradient W2[k' = 16t + (l&15)][n = 16q + 4i + (l>>4)] -- so a
                         // lane's operands for four consecutive k-steps are one coalesced 16-byte load; -1 otherwise
    int fr1_pi, fr1_vf;  // same fragment layout for W1 (k padded with zero rows to 16 * ceil(D / 16)), present when fr_pi >= 0 and D > 32
    int split;           // 1: mfma_dtype = 2 -- the UPDATE runs on the bf16 MFMA with every operand as three bf16 terms (tma_split3.h); rollouts,
                         // evaluation and every derived f32 image are those of the exact-f32 mode
    int sp_pi, sp_vf;    // float offsets of the three-plane fragment-major weight images of the two nets (3 x BfNet.size bf16 each); -1 otherwise
    int total;
};

// ---- bf16 weight images of ONE net (offsets in bf16 elements from the image base).  Every image is "fragment-major": the
// eight bf16 one lane feeds to v_mfma_f32_16x16x32_bf16 as its B operand for (output tile, k-step) are contiguous, and the
// 64 lanes of the wave follow each other, so one B fragment is a single fully coalesced 1 KiB load.
struct BfNet {
    int fW1;   // [H/16 n-tiles][Kp1/32 k-steps][64 lanes][8]: W1[k = 32ks + 8(l>>4) + j][n = 16nt + (l&15)], rows k >= D zero
    int fW2;   // [H/16][H/32][64][8]:                         W2[k][n] likewise                          (layer-2 forward)
    int bW2;   // [H/16 k'-tiles][H/32 n-steps][64][8]:        W2[k' = 16kt + (l&15)][n = 32ns + 8(l>>4) + j]   (dh1 = dz2 . W2^T)
    int fW3;   // [NT3 a-tiles][H/32][64][8]:                  W3[k][a = 16at + (l&15)], columns a >= n_out zero  (head forward)
    int bW3;   // [H/16 k'-tiles][64][8]:                      W3[k' = 16kt + (l&15)][a = 8(l>>4) + j]           (dh2 = dz3 . W3^T)
    int size;  // bf16 elements, multiple of 8
};
__host__ __device__ inline BfNet bf_net_layout(int D, int H, int n_out) {
    const int Kp1 = (D + 31) & ~31, NT3 = n_out > 16 ? 2 : 1;
    BfNet B;
    int o = 0;
    B.fW1 = o, o += H * Kp1;
    B.fW2 = o, o += H * H;
    B.bW2 = o, o += H * H;
    B.fW3 = o, o += NT3 * 16 * H;
    B.bW3 = o, o += H * 32;
    B.size = o;
    return B;
}

// ---- H = 64 LDS weight image of ONE net (floats).  Matrices are stored so that the four B operands a lane needs for one
// k-step (output columns 16j + (lane&15), j = 0..3) are one aligned float4: element [k][r16][j].  ds_read_b128 of that
// layout is bank-conflict-free (each 16-lane group of the instruction covers 256 contiguous bytes).
constexpr int IMG_W1 = 0;                   // [16 k][16 r16][4 j]   layer-1 forward   (rows k >= D are zero)
constexpr int IMG_W2F = IMG_W1 + 1024;      // [64 k][16][4]         layer-2 forward
constexpr int IMG_W3F = IMG_W2F + 4096;     // [64 k][16 cols]       head forward      (cols >= n_out are zero)
constexpr int IMG_W3B = IMG_W3F + 1024;     // [16 n][16][4]         head input-gradient: W3[n][k]   (rows n >= n_out zero)
constexpr int IMG_W2B = IMG_W3B + 1024;     // [64 n][16][4]         layer-2 input-gradient: W2[n][k]
constexpr int IMG_B1 = IMG_W2B + 4096;      // [64]
constexpr int IMG_B2 = IMG_B1 + 64;         // [64]
constexpr int IMG_B3 = IMG_B2 + 64;         // [16]
constexpr int IMG_FLOATS = IMG_B3 + 16 + 16;  // padded to a multiple of 4 floats (11440)
constexpr int IMG_FWD_FLOATS = IMG_W3B;       // forward-only kernels stage [0, IMG_W3B) + the biases

__host__ __device__ inline PLayout make_layout(int D, int H, int A, int cont, int bf16 = 0) {
    PLayout L;
    L.D = D, L.H = H, L.A = A, L.cont = cont;
    int o = 0;
    L.pW1t = o, o += D * H;
    L.pb1 = o, o += H;
    L.pW2t = o, o += H * H;
    L.pb2 = o, o += H;
    L.pW3t = o, o += H * A;
    L.pb3 = o, o += A;
    L.vW1t = o, o += D * H;
    L.vb1 = o, o += H;
    L.vW2t = o, o += H * H;
    L.vb2 = o, o += H;
    L.vW3t = o, o += H;
    L.vb3 = o, o += 1;
    L.log_std = o, o += cont ? A : 0;
    L.P = o;
    L.pW2 = o, o += H * H;
    L.pW3 = o, o += A * H;
    L.vW2 = o, o += H * H;
    L.vW3 = o, o += H;
    o = (o + 3) & ~3;  // 16-byte aligned image regions (staged with float4 copies)
    const bool fast = (H == 64) && (D <= 16) && !cont && (A <= 16);
    L.img_pi = fast ? o : -1, o += fast ? IMG_FLOATS : 0;
    L.img_vf = fast ? o : -1, o += fast ? IMG_FLOATS : 0;
    L.bf16 = bf16 == 1 ? 1 : 0;
    L.split = bf16 == 2 ? 1 : 0;
    L.bf_pi = L.bf_vf = -1;
    if (L.bf16) {
        L.bf_pi = o, o += bf_net_layout(D, H, A).size / 2;
        L.bf_vf = o, o += bf_net_layout(D, H, 1).size / 2;
    }
    L.fr_pi = L.fr_vf = -1;
    if (!L.bf16 && (H == 128 || H == 192 || H == 256)) {
        L.fr_pi = o, o += 2 * H * H;
        L.fr_vf = o, o += 2 * H * H;
    }
    L.fr1_pi = L.fr1_vf = -1;
    if (L.fr_pi >= 0 && D > 32) {
        L.fr1_pi = o, o += H * 16 * ((D + 15) / 16);
        L.fr1_vf = o, o += H * 16 * ((D + 15) / 16);
    }
    L.sp_pi = L.sp_vf = -1;
    if (L.split) {
        L.sp_pi = o, o += 3 * bf_net_layout(D, H, A).size / 2;
        L.sp_vf = o, o += 3 * bf_net_layout(D, H, 1).size / 2;
    }
    L.total = o;
    return L;
}

// One Adam step of parameter p with the clip-scaled gradient gv (torch.optim.Adam, no weight decay / amsgrad; SB3's optimizer, SURVEY.md
// Appendix C.5) on the hardware sqrt / rcp units (1 ulp each; the update term carries ~3 ulp, i.e. ~1e-10 absolute at lr 3e-4) with
// explicit FMAs: 11 VALU operations instead of the ~55 of two IEEE divisions and an IEEE sqrt.  Shared by adam_scatter_h64_kernel and the
// persistent epoch kernel (tma_h64p.hip), which therefore stay bit-identical.  inv_bc2_sqrt = 1 / sqrt(1 - beta2^t).
__device__ __forceinline__ float adam_update_h64(float p, float gv, float &mm, float &vv, float beta1, float beta2, float inv_bc2_sqrt, float eps,
                                                 float lr_step) {
    mm = __builtin_fmaf(gv - mm, 1.0f - beta1, mm);
    vv = __builtin_fmaf(gv * gv, 1.0f - beta2, vv * beta2);
    const float denom = __builtin_fmaf(__builtin_amdgcn_sqrtf(vv), inv_bc2_sqrt, eps);
    return __builtin_fmaf(-lr_step, mm * __builtin_amdgcn_rcpf(denom), p);
}

// Fast-path layouts (H == 64, LDS images): the derived locations of trainable parameter e -- its [out][in] copy and its slots in
// the forward / input-gradient images of its net (inverse of build_image_elem).
__device__ __forceinline__ void scatter_derived_h64(float *params, const PLayout &L, int e, float val) {
    constexpr int H = 64;
    const int D = L.D, A = L.A;
    const bool vf = e >= L.vW1t && e < L.log_std;
    const int base = vf ? L.vW1t : L.pW1t, img = vf ? L.img_vf : L.img_pi, n_out = vf ? 1 : A;
    int x = e - base;
    if (x < D * H) {  // W1t[k][n]
        const int k = x >> 6, n = x & 63;
        params[img + IMG_W1 + k * 64 + (n & 15) * 4 + (n >> 4)] = val;
        return;
    }
    x -= D * H;
    if (x < H) { params[img + IMG_B1 + x] = val; return; }
    x -= H;
    if (x < H * H) {  // W2t[k][n]
        const int k = x >> 6, n = x & 63;
        params[(vf ? L.vW2 : L.pW2) + n * H + k] = val;
        params[img + IMG_W2F + k * 64 + (n & 15) * 4 + (n >> 4)] = val;
        params[img + IMG_W2B + n * 64 + (k & 15) * 4 + (k >> 4)] = val;
        return;
    }
    x -= H * H;
    if (x < H) { params[img + IMG_B2 + x] = val; return; }
    x -= H;
    if (x < H * n_out) {  // W3t[k][a]
        const int k = x / n_out, a = x - k * n_out;
        params[(vf ? L.vW3 : L.pW3) + a * H + k] = val;
        params[img + IMG_W3F + k * 16 + a] = val;
        params[img + IMG_W3B + a * 64 + (k & 15) * 4 + (k >> 4)] = val;
        return;
    }
    x -= H * n_out;
    if (x < n_out) params[img + IMG_B3 + x] = val;
}

// value the optimiser must treat as new (not loop-invariant) but provably wave-uniform: keeps weight loads of a persistent loop
// from being hoisted in front of it (and spilled), and keeps their address arithmetic scalar
__device__ __forceinline__ const float *launder_uniform(const float *p) {
    uint64_t v = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+s"(v));
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const float *>(((uint64_t)hi << 32) | lo);
}
// fragment `idx` (64 lanes x 4 floats) of an f32 fragment-major image: scalar base + 32-bit lane offset, explicit global load
__device__ __forceinline__ f32x4 frag_f32(const float *img, int idx, int lane) {
    typedef const char __attribute__((address_space(1))) *gbyte_ptr;
    typedef const f32x4 __attribute__((address_space(1))) *gvec_ptr;
    const gbyte_ptr base = reinterpret_cast<gbyte_ptr>(reinterpret_cast<uintptr_t>(img + (int64_t)idx * 256));
    return *reinterpret_cast<gvec_ptr>(base + (uint32_t)lane * 16u);
}

// out[16][N] = tanh(in[16][K] . Wt[K][N] + b)   (N % 64 == 0; in/out are wave-private LDS tiles)
__device__ __forceinline__ void dense_tanh(const float *in, int ldi, int K, const float *__restrict__ Wt, const float *__restrict__ b, int N,
                                           float *out, int ldo, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    const int KS = (K + 3) >> 2;
    for (int n0 = 0; n0 < N; n0 += 64) {
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float bias = b[n0 + 16 * j + r16];
            acc[j] = f32x4{bias, bias, bias, bias};
        }
        for (int ks = 0; ks < KS; ks++) {
            const int k = 4 * ks + g;
            const bool ok = k < K;
            const float a = ok ? in[r16 * ldi + k] : 0.0f;
            const float *wrow = Wt + (int64_t)k * N + n0 + r16;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float w = ok ? wrow[16 * j] : 0.0f;
                acc[j] = mfma16(a, w, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) out[(g * 4 + r) * ldo + n0 + 16 * j + r16] = tma_tanh(acc[j][r]);
    }
}

// acc[NT] (C layout, columns 16*j + (lane&15)) = in[16][K] . Wt[K][N] + b, columns >= N are zero
template <int NT>
__device__ __forceinline__ void dense_head(const float *in, int ldi, int K, const float *__restrict__ Wt, const float *__restrict__ b, int N,
                                           f32x4 (&acc)[NT], int lane) {
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int col = 16 * j + r16;
        const float bias = col < N ? b[col] : 0.0f;
        acc[j] = f32x4{bias, bias, bias, bias};
    }
    for (int ks = 0; ks < (K >> 2); ks++) {
        const int k = 4 * ks + g;
        const float a = in[r16 * ldi + k];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int col = 16 * j + r16;
            const float w = col < N ? Wt[(int64_t)k * N + col] : 0.0f;
            acc[j] = mfma16(a, w, acc[j]);
        }
    }
}

// dzout[16][K] = (dzin[16][N] . W[N][K]) * (1 - hprev^2)      (K % 64 == 0)
__device__ __forceinline__ void dense_bwd_input(const float *dzin, int ldz, int N, const float *__restrict__ W, int K, const float *hprev,
                                                int ldh, float *dzout, int ldo, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    const int NS = (N + 3) >> 2;
    for (int k0 = 0; k0 < K; k0 += 64) {
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int ns = 0; ns < NS; ns++) {
            const int n = 4 * ns + g;
            const bool ok = n < N;
            const float a = ok ? dzin[r16 * ldz + n] : 0.0f;
            const float *wrow = W + (int64_t)n * K + k0 + r16;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float w = ok ? wrow[16 * j] : 0.0f;
                acc[j] = mfma16(a, w, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = g * 4 + r, col = k0 + 16 * j + r16;
                const float h = hprev[row * ldh + col];
                const float d1 = 1.0f - h * h;
                dzout[row * ldo + col] = acc[j][r] * d1;
            }
    }
}

// gWt[K][N] += xin[16][K]^T . dz[16][N] ;  gb[N] += column sums of dz   (float atomics into the gradient buffer)
__device__ __forceinline__ void dense_bwd_weight(const float *xin, int ldx, int K, const float *dz, int ldz, int N, float *gWt, float *gb,
                                                 int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    for (int n0 = 0; n0 < N; n0 += 16) {
        const int col = n0 + r16;
        float bf[4];
#pragma unroll
        for (int s = 0; s < 4; s++) bf[s] = col < N ? dz[(4 * s + g) * ldz + col] : 0.0f;
        float cs = (bf[0] + bf[1]) + (bf[2] + bf[3]);
        cs += __shfl_xor(cs, 16, 64);
        cs += __shfl_xor(cs, 32, 64);
        if (g == 0 && col < N) atomicAdd(gb + col, cs);
        for (int k0 = 0; k0 < K; k0 += 16) {
            f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const int krow = k0 + r16;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const float a = krow < K ? xin[(4 * s + g) * ldx + krow] : 0.0f;
                acc = mfma16(a, bf[s], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = k0 + g * 4 + r;
                if (k < K && col < N) atomicAdd(gWt + (int64_t)k * N + col, acc[r]);
            }
        }
    }
}

// ---- H = 64 layers with the weights staged in LDS (image layout above) ----
// out[16][64] = tanh(in[16][4*KS] . W + b)
template <int KS_CT>  // KS_CT > 0: compile-time k-step count (fully unrolled); 0: runtime count `KS` (layer 1, 1..4 steps)
__device__ __forceinline__ void dense64_tanh_lds(const float *in, int ldi, int KS, const float *Wimg, const float *b, float *out, int ldo, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float bias = b[16 * j + r16];
        acc[j] = f32x4{bias, bias, bias, bias};
    }
#pragma unroll
    for (int ks = 0; ks < (KS_CT > 0 ? KS_CT : KS); ks++) {
        const int k = 4 * ks + g;
        const float a = in[r16 * ldi + k];
        const float4 w = *reinterpret_cast<const float4 *>(Wimg + k * 64 + r16 * 4);
        acc[0] = mfma16(a, w.x, acc[0]);
        acc[1] = mfma16(a, w.y, acc[1]);
        acc[2] = mfma16(a, w.z, acc[2]);
        acc[3] = mfma16(a, w.w, acc[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) out[(g * 4 + r) * ldo + 16 * j + r16] = tma_tanh(acc[j][r]);
}

// acc (C layout, column lane&15) = in[16][64] . W3f + b3
__device__ __forceinline__ f32x4 dense64_head_lds(const float *in, int ldi, const float *W3f, const float *b3, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    const float bias = b3[r16];
    f32x4 acc = f32x4{bias, bias, bias, bias};
#pragma unroll
    for (int ks = 0; ks < 16; ks++) {
        const int k = 4 * ks + g;
        acc = mfma16(in[r16 * ldi + k], W3f[k * 16 + r16], acc);
    }
    return acc;
}

// dzout[16][64] = (dzin[16][4*NS] . Wb) * (1 - hprev^2)
template <int NS>
__device__ __forceinline__ void dense64_bwd_input_lds(const float *dzin, int ldz, const float *Wb, const float *hprev, int ldh, float *dzout,
                                                      int ldo, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ns = 0; ns < NS; ns++) {
        const int n = 4 * ns + g;
        const float a = dzin[r16 * ldz + n];
        const float4 w = *reinterpret_cast<const float4 *>(Wb + n * 64 + r16 * 4);
        acc[0] = mfma16(a, w.x, acc[0]);
        acc[1] = mfma16(a, w.y, acc[1]);
        acc[2] = mfma16(a, w.z, acc[2]);
        acc[3] = mfma16(a, w.w, acc[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = g * 4 + r, col = 16 * j + r16;
            const float h = hprev[row * ldh + col];
            const float d1 = 1.0f - h * h;
            dzout[row * ldo + col] = acc[j][r] * d1;
        }
}

// cooperative float4 copy global -> LDS (n_floats % 4 == 0, both 16-byte aligned)
// (eight loads per thread in flight: the one-element-per-iteration form waits out a memory round trip per iteration -- six in a row for
// a 45 KiB weight image on 512 threads, at the head of every launch that stages one)
__device__ __forceinline__ void stage_copy(const float *__restrict__ src, float *dst, int n_floats) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    const int n4 = n_floats >> 2, stride = blockDim.x;
    for (int e0 = threadIdx.x; e0 < n4; e0 += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * stride;
            v[u] = s4[e < n4 ? e : n4 - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * stride;
            if (e < n4) d4[e] = v[u];
        }
    }
}

// stage a [16][D] tile of rows (gathered through row_off[]) into LDS, zero rows that are out of range
__device__ __forceinline__ void load_obs_tile(const float *__restrict__ src, const int64_t *row_off_lds, int D, float *X, int ldx, int lane) {
    const int Dp = (D + 3) & ~3;  // columns D..Dp-1 are the zero padding of the last k-step
    const int total = 16 * Dp;
    for (int e = lane; e < total; e += 64) {
        const int row = e / Dp, c = e - row * Dp;
        const int64_t off = row_off_lds[row];
        X[row * ldx + c] = (off >= 0 && c < D) ? src[off * D + c] : 0.0f;
    }
}

__device__ __forceinline__ float uniform01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

constexpr int FWD_IMG = IMG_FWD_FLOATS + 160;  // forward matrices + the three biases

__device__ __forceinline__ void stage_fwd_image(const float *img, float *dst) {
    stage_copy(img, dst, IMG_FWD_FLOATS);
    stage_copy(img + IMG_B1, dst + IMG_FWD_FLOATS, 160);
}

__device__ __forceinline__ f32x4 value_tile_lds(const float *vimg, const float *X, int ldx, int KS1, float *h1, float *h2, int ld, int lane) {
    dense64_tanh_lds<0>(X, ldx, KS1, vimg + IMG_W1, vimg + IMG_FWD_FLOATS, h1, ld, lane);
    dense64_tanh_lds<16>(h1, ld, 16, vimg + IMG_W2F, vimg + IMG_FWD_FLOATS + 64, h2, ld, lane);
    return dense64_head_lds(h2, ld, vimg + IMG_W3F, vimg + IMG_FWD_FLOATS + 128, lane);
}

// reductions inside one 16-lane group (lanes sharing lane>>4) as DPP row operations -- VALU only, no ds_bpermute round trip
// through the LDS (what __shfl_xor compiles to: ~100 cycles and an lgkmcnt wait per step, 16 dependent steps per softmax row).
// quad_perm [1,0,3,2] / [2,3,0,1] are the xor-1 / xor-2 exchanges; after them every lane of a quad holds the quad's value, so
// row_half_mirror (lane i <- 7-i) delivers the other quad of the half and row_mirror (i <- 15-i) the other half: the same
// pairing, hence bit-for-bit the same result, as the xor-1/2/4/8 butterfly.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float gsum16(float v) {
    v += dpp_row<0xB1>(v);
    v += dpp_row<0x4E>(v);
    v += dpp_row<0x141>(v);
    v += dpp_row<0x140>(v);
    return v;
}
__device__ __forceinline__ float gmin16(float v) {
    v = fminf(v, dpp_row<0xB1>(v));
    v = fminf(v, dpp_row<0x4E>(v));
    v = fminf(v, dpp_row<0x141>(v));
    v = fminf(v, dpp_row<0x140>(v));
    return v;
}
// inclusive prefix sum over the 16 lanes of a group: row_shr:d with bound_ctrl feeds 0 to lanes whose source leaves the row
__device__ __forceinline__ float gscan16(float c) {
    c += dpp_row<0x111>(c);
    c += dpp_row<0x112>(c);
    c += dpp_row<0x114>(c);
    c += dpp_row<0x118>(c);
    return c;
}
// value of the group's first lane (lane & 15 == 0), valid in the lanes of the group's first quad (lane & 15 < 4)
__device__ __forceinline__ float gfirst_quad(float v) { return dpp_row<0x00>(v); }
__device__ __forceinline__ float gmax16(float v) {
    v = fmaxf(v, dpp_row<0xB1>(v));
    v = fmaxf(v, dpp_row<0x4E>(v));
    v = fmaxf(v, dpp_row<0x141>(v));
    v = fmaxf(v, dpp_row<0x140>(v));
    return v;
}

// minibatch permutation: 4-round Feistel network over [0, 2^b) with cycle walking down to [0, n).  Stands in for
// np.random.permutation in SB3's RolloutBuffer.get (the reference draws it from the process-global MT19937 that the
// envs also consume, so its order is not reproducible for a vectorised run anyway -- SURVEY.md §7.3-1).
__host__ __device__ inline uint32_t perm_index(uint32_t seed, uint32_t epoch, uint32_t j, uint32_t n) {
    uint32_t b = 2;
    while ((1ull << b) < (unsigned long long)n) b += 2;
    const uint32_t half = b >> 1, mask = (1u << half) - 1u;
    const uint32_t key = seed ^ (epoch * 0x9E3779B9u);
    uint32_t x = j;
    do {
        uint32_t L = x >> half, R = x & mask;
        for (uint32_t rd = 0; rd < 4; rd++) {
            const uint32_t t = R;
            R = L ^ (mix32(key, R, rd) & mask);
            L = t;
        }
        x = (L << half) | R;
    } while (x >= n);
    return x;
}

}  // namespace tma

// class_bf16x3.h — the class layer's three products on the bf16 matrix pipe with f32 results (round 5):
//     forward   Z0[m x p]    = H1[m x 128] . W2[128 x p]                      (Matmul::forward,  module.cpp:11-22)
//     backward  dH1[m x 128] = mask . (s_r . (dZ0[m x p] . W2^T))             (Matmul::backward, module.cpp:24-42, with the
//               dW2[128 x p] = H1^T[128 x m] . dZ0[m x p]                      ReLU/dropout backward of module.cpp:187-194, 223-233)
// for p <= 64 classes (Reddit: 41) and a hidden width of exactly 128.
// Why: the f32-MFMA row-stream kernels of dense_kernels.h spend as long in the matrix pipe as in memory (2.9 GFLOP per product
// on a 157 TF pipe = 18 us of a 41-53 us launch, pipes 0.4-0.6 busy: profiles/r05_class_layer_pmc.json) and the backward reads
// dZ0 twice, in two launches, plus a slab pass.  Here every f32 operand is split exactly into three bf16 planes in registers
// (bf16x3_split.h; same six plane products and the same error bound as the first-layer kernels of dense_bf16x3.h: inside the
// f32 summation-order bound), which makes the matrix time 6/16 of what it was, and the backward is ONE launch: a wave reads a
// block of 32 rows of dZ0 and H1 once, stores its 32 rows of dH1 and keeps its share of dW2 in accumulators until the end.
// Layouts (v_mfma_f32_32x32x16_bf16: A[i = lane & 31][k = 8 (lane >> 5) + j], B[k][n = lane & 31],
// D[(r & 3) + 8 (r >> 2) + 4 (lane >> 5)][lane & 31]):
//  * products whose result is a row block [32 rows x features] are computed TRANSPOSED (weights as A, the row block as B): a
//    lane then owns one ROW and, per accumulator quad, four consecutive columns of it — 16-byte row stores, one mask word quad
//    and one row factor per lane — and the row block is read in the layout it has in memory (a lane reads 8 consecutive floats
//    of its row per k-step);
//  * dW2 contracts over rows: both operands are read column-wise (a lane reads one column of 8 consecutive rows: 128-byte
//    requests per half wave), dZ0 from the lines the same wave has just read.
// The small operand (W2: 128 x p) is split by the workgroup itself into an LDS image; no pre-pass.
#pragma once
#include "dense_kernels.h"
#include "bf16x3_split.h"

struct ClsB3 { bf16x8 h, m, l; };

__device__ __forceinline__ ClsB3 cls_planes(const float v[8]) {
    BxPlanes P;
#pragma unroll
    for (int i = 0; i < 4; i++) bx_split2(v[2 * i], v[2 * i + 1], P.w[0][i], P.w[1][i], P.w[2][i]);
    return ClsB3{bx_plane(P, 0), bx_plane(P, 1), bx_plane(P, 2)};
}
// c += A . B from the six plane products of weight >= 2^-16, smallest first (as dense_bf16x3.h's mac)
__device__ __forceinline__ void cls_mac(f32x16 &c, const ClsB3 &A, const ClsB3 &B) {
    c = MFMA_BF16(A.l, B.h, c);
    c = MFMA_BF16(A.m, B.m, c);
    c = MFMA_BF16(A.h, B.l, c);
    c = MFMA_BF16(A.m, B.h, c);
    c = MFMA_BF16(A.h, B.m, c);
    c = MFMA_BF16(A.h, B.h, c);
}
__device__ __forceinline__ ClsB3 cls_lds(const uint4 *img, int piece, int lane) {
    const uint4 h = img[(piece * 3 + 0) * 64 + lane], m = img[(piece * 3 + 1) * 64 + lane], l = img[(piece * 3 + 2) * 64 + lane];
    return ClsB3{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, m), __builtin_bit_cast(bf16x8, l)};
}
__device__ __forceinline__ void cls_lds_put(uint4 *img, int piece, int lane, const float v[8]) {
    BxPlanes P;
#pragma unroll
    for (int i = 0; i < 4; i++) bx_split2(v[2 * i], v[2 * i + 1], P.w[0][i], P.w[1][i], P.w[2][i]);
#pragma unroll
    for (int pl = 0; pl < 3; pl++) img[(piece * 3 + pl) * 64 + lane] = make_uint4(P.w[pl][0], P.w[pl][1], P.w[pl][2], P.w[pl][3]);
}

// ------------------------------------------------------------------------------------------------------------ forward
struct ClsFwdArgs {
    const float *h1; int ldh;        // [m x 128], 16-byte aligned rows
    const float *w2; int ldw;        // [128 x p]
    float *z0; int ldz;              // [m x p], 16-byte aligned rows; padding columns are not written
    int m, p, n_rb;                  // n_rb = ceil(m / 32)
};
constexpr int CLS_FWD_LDS = 8 * 2 * 3 * 1024;               // pieces (k-step s, class block cb): W2^T as the A operand

__global__ __launch_bounds__(256) void class_fwd_bf16x3_kernel(ClsFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 cls_img[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hh = lane >> 5;
    // A[i = class][k = feature]: piece (s, cb) of lane ln holds W2[16 s + 8 (ln >> 5) + j][32 cb + (ln & 31)], j = 0..7
    for (int idx = threadIdx.x; idx < 8 * 2 * 64; idx += 256) {
        const int ln = idx & 63, cb = (idx >> 6) & 1, s = idx >> 7;
        const int c = 32 * cb + (ln & 31), k0 = 16 * s + 8 * (ln >> 5);
        const bool ok = c < a.p;
        const float *wp = a.w2 + (size_t)k0 * a.ldw + (ok ? c : 0);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = wp[(size_t)j * a.ldw];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = ok ? v[j] : 0.f;
        cls_lds_put(cls_img, s * 2 + cb, ln, v);
    }
    __syncthreads();
    for (int rb = blockIdx.x * 4 + wave; rb < a.n_rb; rb += gridDim.x * 4) {
        const int row = rb * 32 + li;
        const float *hp = a.h1 + (size_t)min(row, a.m - 1) * a.ldh + 8 * hh;     // rows past m compute on a copy of the last row and store nothing
        float4 raw[16];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            raw[2 * s] = *reinterpret_cast<const float4 *>(hp + 16 * s);
            raw[2 * s + 1] = *reinterpret_cast<const float4 *>(hp + 16 * s + 4);
        }
        f32x16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const float v[8] = {raw[2 * s].x, raw[2 * s].y, raw[2 * s].z, raw[2 * s].w, raw[2 * s + 1].x, raw[2 * s + 1].y, raw[2 * s + 1].z, raw[2 * s + 1].w};
            const ClsB3 B = cls_planes(v);
#pragma unroll
            for (int cb = 0; cb < 2; cb++) cls_mac(acc[cb], cls_lds(cls_img, s * 2 + cb, lane), B);
        }
        if (row < a.m) {
            float *zp = a.z0 + (size_t)row * a.ldz;
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c0 = 32 * cb + 8 * q + 4 * hh;
                    if (c0 + 4 <= a.p) {
                        *reinterpret_cast<float4 *>(zp + c0) = make_float4(acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; i++) if (c0 + i < a.p) zp[c0 + i] = acc[cb][4 * q + i];
                    }
                }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------- backward
struct ClsBwdArgs {
    const float *dz; int lddz;       // dZ0 [m x p], 16-byte aligned rows, lddz >= 16 ceil(p / 16) (whole k-steps are read)
    const float *h1; int ldh;        // H1 [m x 128]
    const float *w2; int ldw;        // W2 [128 x p]
    float *da; int ldda;             // dH1 [m x 128], 16-byte aligned rows
    const uint32_t *bits;            // bit (f & 31) of bits[r * 4 + (f >> 5)] = (H1[r, f] > 0); 16-byte aligned
    const float *rowscale; float scale;   // dH1[r, :] = mask . (scale * rowscale[r]) . (...)   (rowscale may be NULL)
    float *slab; int p_ld;           // dW2 partials [gridDim.x][128][p_ld]
    int m, p, n_rb, n_ks;            // n_ks = ceil(p / 16) <= 4
};
constexpr int CLS_BWD_WAVES = 8;
constexpr int CLS_BWD_IMG = 4 * 4 * 3 * 1024;               // pieces (k-step s < 4, feature block fb): W2 as the A operand of dH1^T
constexpr int CLS_BWD_LDS = CLS_BWD_IMG + 128 * 64 * 4;     // + the workgroup's sum of dW2

__global__ __launch_bounds__(512) void class_bwd_bf16x3_kernel(ClsBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 cls_img[];
    float *red = reinterpret_cast<float *>(cls_img) + CLS_BWD_IMG / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hh = lane >> 5;
    // A[i = feature][k = class]: piece (s, fb) of lane ln holds W2[32 fb + (ln & 31)][16 s + 8 (ln >> 5) + j], j = 0..7 (classes past p: zero)
    for (int idx = threadIdx.x; idx < a.n_ks * 4 * 64; idx += 512) {
        const int ln = idx & 63, fb = (idx >> 6) & 3, s = idx >> 8;
        const int f = 32 * fb + (ln & 31), c0 = 16 * s + 8 * (ln >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = a.w2[(size_t)f * a.ldw + min(c0 + j, a.p - 1)];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = c0 + j < a.p ? v[j] : 0.f;
        cls_lds_put(cls_img, s * 4 + fb, ln, v);
    }
    for (int i = threadIdx.x; i < 128 * 64; i += 512) red[i] = 0.f;
    __syncthreads();

    f32x16 dw[4][2];                                         // dW2[32 fb + ..][32 cb + li]: this wave's rows
#pragma unroll
    for (int fb = 0; fb < 4; fb++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) dw[fb][cb][r] = 0.f;

    for (int rb = blockIdx.x * CLS_BWD_WAVES + wave; rb < a.n_rb; rb += gridDim.x * CLS_BWD_WAVES) {
        const int row0 = rb * 32, row = row0 + li;
        const bool live = row < a.m;
        const size_t rc = (size_t)min(row, a.m - 1);
        // ---- dH1^T[feature][row] = sum_c W2[feature][c] . dZ0[row][c]: dZ0 in its memory layout as the B operand
        const float *zp = a.dz + rc * a.lddz + 8 * hh;
        float4 zr[8];
#pragma unroll
        for (int s = 0; s < 4; s++)
            if (s < a.n_ks) { zr[2 * s] = *reinterpret_cast<const float4 *>(zp + 16 * s); zr[2 * s + 1] = *reinterpret_cast<const float4 *>(zp + 16 * s + 4); }
        const uint4 kb = *reinterpret_cast<const uint4 *>(a.bits + rc * 4);
        const float sc = a.rowscale ? a.scale * a.rowscale[rc] : a.scale;
        f32x16 acc[4];
#pragma unroll
        for (int fb = 0; fb < 4; fb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[fb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s < a.n_ks) {
                float v[8] = {zr[2 * s].x, zr[2 * s].y, zr[2 * s].z, zr[2 * s].w, zr[2 * s + 1].x, zr[2 * s + 1].y, zr[2 * s + 1].z, zr[2 * s + 1].w};
                const int c0 = 16 * s + 8 * hh;
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = c0 + j < a.p ? v[j] : 0.f;      // padding columns may hold anything
                const ClsB3 B = cls_planes(v);
#pragma unroll
                for (int fb = 0; fb < 4; fb++) cls_mac(acc[fb], cls_lds(cls_img, s * 4 + fb, lane), B);
            }
        }
        if (live) {
            float *dp = a.da + (size_t)row * a.ldda;
            const uint32_t kw[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
            for (int fb = 0; fb < 4; fb++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t nib = kw[fb] >> (8 * q + 4 * hh);
                    float x[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) x[i] = ((nib >> i) & 1u) ? acc[fb][4 * q + i] * sc : 0.f;
                    *reinterpret_cast<float4 *>(dp + 32 * fb + 8 * q + 4 * hh) = make_float4(x[0], x[1], x[2], x[3]);
                }
        }
        // ---- dW2[feature][c] += sum_rows H1[row][feature] . dZ0[row][c]: both operands column-wise (k = row)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const int r0 = row0 + 16 * s2 + 8 * hh;         // this lane's 8 rows of the k-step
            ClsB3 Bz[2];
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                const int c = 32 * cb + li;
                const bool okc = c < a.p;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = a.dz[(size_t)min(r0 + j, a.m - 1) * a.lddz + (okc ? c : 0)];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = (okc && r0 + j < a.m) ? v[j] : 0.f;   // rows past m and classes past p contribute nothing
                Bz[cb] = cls_planes(v);
            }
#pragma unroll
            for (int fb = 0; fb < 4; fb++) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = a.h1[(size_t)min(r0 + j, a.m - 1) * a.ldh + 32 * fb + li];
                const ClsB3 A = cls_planes(v);
                cls_mac(dw[fb][0], A, Bz[0]);
                if (a.p > 32) cls_mac(dw[fb][1], A, Bz[1]);
            }
        }
    }
    // ---- the workgroup's sum, wave after wave (fixed order), then one slab per workgroup
    for (int w = 0; w < CLS_BWD_WAVES; w++) {
        if (wave == w) {
#pragma unroll
            for (int fb = 0; fb < 4; fb++)
#pragma unroll
                for (int cb = 0; cb < 2; cb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int f = 32 * fb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        red[f * 64 + 32 * cb + li] += dw[fb][cb][r];
                    }
        }
        __syncthreads();
    }
    float *slab = a.slab + (size_t)blockIdx.x * 128 * a.p_ld;
    for (int i = threadIdx.x; i < 128 * a.p_ld; i += 512) {
        const int f = i / a.p_ld, c = i % a.p_ld;
        slab[i] = c < 64 ? red[f * 64 + c] : 0.f;
    }
}

// the shapes these kernels take (anything else: the f32-MFMA kernels of dense_kernels.h)
static inline bool cls_fwd_fits(const float *a, int lda, const float *c, int ldc, int m, int n, int p) {
    return n == 128 && p >= 1 && p <= 64 && m >= 2048 && lda % 4 == 0 && lda >= 128 && ldc % 4 == 0 && aligned16(a) && aligned16(c);
}
static inline bool cls_bwd_fits(const float *a, int lda, const float *dc, int lddc, const float *da, int ldda, const uint32_t *bits, int wpr,
                                int m, int n, int p) {
    return n == 128 && p >= 1 && p <= 64 && m >= 2048 && wpr == 4 && bits && aligned16(bits) && a && lda >= 128 && lddc % 4 == 0 &&
           lddc >= (p + 15) / 16 * 16 && ldda % 4 == 0 && ldda >= 128 && aligned16(dc) && aligned16(da);
}

static int launch_class_fwd(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb, float *z, int ldz, int m, int p) {
    ClsFwdArgs k;
    k.h1 = a; k.ldh = lda; k.w2 = b; k.ldw = ldb; k.z0 = z; k.ldz = ldz; k.m = m; k.p = p; k.n_rb = ceil_div(m, 32);
    int grid = ceil_div(k.n_rb, 4);
    if (grid > c->n_cu * 3) grid = c->n_cu * 3;              // 48 KB of LDS per workgroup: three per CU
    class_fwd_bf16x3_kernel<<<grid, 256, CLS_FWD_LDS, c->stream>>>(k);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

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
    const float *h1; int ldh;        // [m x 128], 16-byte aligned rows; m * ldh * 4 < 2^32 (buffer loads)
    const float *w2; int ldw;        // [128 x p]
    float *z0; int ldz;              // [m x p], 16-byte aligned rows; padding columns are not written
    int m, p, n_rb;                  // n_rb = ceil(m / 32)
};
constexpr int CLS_FWD_LDS = 8 * 2 * 3 * 1024;               // pieces (k-step s, class block cb): W2^T as the A operand

#define CLS_WAIT2(N, x, y) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(x), "+v"(y) :: "memory")
template <int S>
__device__ __forceinline__ void cls_fwd_issue(f32x4 (&R)[16], uint32_t vo, u32x4 rs) {
    bx_bload16<64 * S>(R[2 * S], vo, rs);
    bx_bload16<64 * S + 16>(R[2 * S + 1], vo, rs);
}

#define CLS_TIE4(R, i) "+v"((R)[i]), "+v"((R)[(i) + 1]), "+v"((R)[(i) + 2]), "+v"((R)[(i) + 3])
#define CLS_WAIT_ALL16(R) asm volatile("s_waitcnt vmcnt(0)" : CLS_TIE4(R, 0), CLS_TIE4(R, 4), CLS_TIE4(R, 8), CLS_TIE4(R, 12) :: "memory")

// A wave keeps the 16 row loads of ONE row block in flight at all times: the two loads of k-step s of the NEXT block are issued
// the moment this block's k-step s has been turned into planes (pinned by an empty asm, so that the old values are dead before
// their registers are named as the new loads' destinations), and every wait sees exactly 14 younger loads.
// Round 5, fourth fact about hand-counted loads: these registers are live across the loop's back edge WHILE IN FLIGHT.  A first
// version issued the refill before the split had consumed the old values; hipcc gave the refills registers of their own and
// copied them into the loop's registers at the back edge — 32 v_mov_b64 BEFORE their waits: garbage whenever a load had not
// landed by then.  Every parity test passed (cache-resident inputs land within a block's compute); two rmat-22 runs differed.
// Nothing in the language forbids such copies, so the build is checked instead: tools/check_asm_loads.py walks the compiled
// kernel's control-flow graph and fails (tests/test_isa_cpu.py) if any instruction touches a register of a load in flight.
// (A version with two whole register sets and no load in flight at the back edge is immune by construction and was measured:
// 158 registers, two waves per SIMD, 48.5 us against 36.7.  As plain C++ loads, before all that, hipcc issued each k-step's pair
// right before its use and waited: eight dependent round trips per block, 49 us.)
template <int ABL>
__global__ __launch_bounds__(256) void class_fwd_bf16x3_kernel(ClsFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 cls_img[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hh = lane >> 5;
    // B[k = feature][n = class]: piece (s, cb) of lane ln holds W2[16 s + 8 (ln >> 5) + j][32 cb + (ln & 31)], j = 0..7
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
    const u32x4 rs_h = bx_make_rsrc(a.h1, (uint32_t)a.m * (uint32_t)a.ldh * 4u);      // rows past m read as zero and store nothing
    const int stride = gridDim.x * 4;
    int rb = blockIdx.x * 4 + wave;
    auto voff = [&](int b) { return ((uint32_t)(b * 32 + li) * (uint32_t)a.ldh + 8u * hh) * 4u; };
    auto issue = [&](f32x4 (&Q)[16], uint32_t vo) __attribute__((always_inline)) {
        cls_fwd_issue<0>(Q, vo, rs_h); cls_fwd_issue<1>(Q, vo, rs_h); cls_fwd_issue<2>(Q, vo, rs_h); cls_fwd_issue<3>(Q, vo, rs_h);
        cls_fwd_issue<4>(Q, vo, rs_h); cls_fwd_issue<5>(Q, vo, rs_h); cls_fwd_issue<6>(Q, vo, rs_h); cls_fwd_issue<7>(Q, vo, rs_h);
    };
    f32x4 R[16];
    issue(R, voff(rb));
    for (; rb < a.n_rb; rb += stride) {
        const uint32_t vn = voff(rb + stride);               // past the last block: past the descriptor, zeros, never used
        f32x16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;
#define CLS_FWD_STEP(S)                                                                                                      \
        {                                                                                                                    \
            CLS_WAIT2(14, R[2 * S], R[2 * S + 1]);                                                                           \
            const float v[8] = {R[2 * S][0], R[2 * S][1], R[2 * S][2], R[2 * S][3], R[2 * S + 1][0], R[2 * S + 1][1], R[2 * S + 1][2], R[2 * S + 1][3]}; \
            if constexpr (ABL & 1) {                                                                                         \
                float sum = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];                                             \
                asm volatile("" : "+v"(sum));                                                                                \
                acc[0][S] += sum;                                                                                            \
                cls_fwd_issue<S>(R, vn, rs_h);                                                                               \
            } else {                                                                                                         \
                ClsB3 A = cls_planes(v);                                                                                     \
                asm volatile("" : "+v"(A.h), "+v"(A.m), "+v"(A.l));        /* the k-step's floats are dead from here on */    \
                cls_fwd_issue<S>(R, vn, rs_h);                                                                               \
                cls_mac(acc[0], A, cls_lds(cls_img, S * 2, lane));                                                           \
                if (a.p > 32) cls_mac(acc[1], A, cls_lds(cls_img, S * 2 + 1, lane));                                         \
            }                                                                                                                \
        }
        CLS_FWD_STEP(0) CLS_FWD_STEP(1) CLS_FWD_STEP(2) CLS_FWD_STEP(3) CLS_FWD_STEP(4) CLS_FWD_STEP(5) CLS_FWD_STEP(6) CLS_FWD_STEP(7)
#undef CLS_FWD_STEP
        // D[row = (r & 3) + 8 (r >> 2) + 4 hh][class = 32 cb + li]: a store instruction writes 32 consecutive classes of two rows
        if (!(ABL & 2) || acc[0][0] == 123.456f) {
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int orow = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, c = 32 * cb + li;
                    if (orow < a.m && c < a.p) a.z0[(size_t)orow * a.ldz + c] = acc[cb][r];
                }
        }
    }
    // the loads issued for the block after the last: their registers stay allocated until they have landed
    CLS_WAIT_ALL16(R);
}

// ----------------------------------------------------------------------------------------------------------- backward
struct ClsBwdArgs {
    const float *dz; int lddz;       // dZ0 [m x p], 16-byte aligned rows, lddz >= 16 ceil(p / 16) (whole k-steps are read); m * lddz * 4 < 2^32
    const float *h1; int ldh;        // H1 [m x 128]; m * ldh * 4 < 2^32
    const float *w2; int ldw;        // W2 [128 x p]
    float *da; int ldda;             // dH1 [m x 128], 16-byte aligned rows
    const uint32_t *bits;            // bit (f & 31) of bits[r * 4 + (f >> 5)] = (H1[r, f] > 0); 16-byte aligned
    const float *rowscale; float scale;   // dH1[r, :] = mask . (scale * rowscale[r]) . (...)   (rowscale may be NULL)
    float *slab; int p_ld;           // dW2 partials [gridDim.x][128][p_ld]
    int m, p, n_rb;
};
constexpr int CLS_BWD_PAIRS = 4;                            // wave pairs per workgroup: a pair shares a row block, each wave takes two of the four feature blocks
constexpr int CLS_BWD_IMG = 4 * 4 * 3 * 1024;               // pieces (k-step s < 4, feature block fb): W2 as the A operand of dH1^T
constexpr int CLS_BWD_LDS = CLS_BWD_IMG + 128 * 64 * 4;     // + the workgroup's sum of dW2

struct ClsCol { float z[2][8], h[2][8]; };                  // one k-step of 16 rows, column-wise: dZ0 (class blocks 0, 1) and H1 (this wave's two feature blocks)
#define CLS_TIE8(x) "+v"((x)[0]), "+v"((x)[1]), "+v"((x)[2]), "+v"((x)[3]), "+v"((x)[4]), "+v"((x)[5]), "+v"((x)[6]), "+v"((x)[7])
// N (a compile-time constant) = the number of loads this wave has issued AFTER the ones waited for — never more
#define CLS_WAIT_COL(N, c) asm volatile("s_waitcnt vmcnt(%32)" : CLS_TIE8((c).z[0]), CLS_TIE8((c).z[1]), CLS_TIE8((c).h[0]), CLS_TIE8((c).h[1]) : "n"(N) : "memory")

// NKS = ceil(p / 16) k-steps of classes.  Vector-memory loads of a row block, in issue order (all inline asm):
//   L1(b): 2 NKS row pieces of dZ0, the row's 4 mask words, its factor   (2 NKS + 2 loads)   — issued during block b-1, landed by its end
//   C0(b): k-step 0 column-wise: 16 + 16 dwords                          (32)                — issued at the top of block b
//   C1(b): k-step 1, into C0's registers once those are planes            (32)
template <int NKS, int ABL = 0>
__global__ __launch_bounds__(512) void class_bwd_bf16x3_kernel(ClsBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 cls_img[];
    float *red = reinterpret_cast<float *>(cls_img) + CLS_BWD_IMG / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hh = lane >> 5;
    const int pair = wave >> 1, half = wave & 1;
    // A[i = feature][k = class]: piece (s, fb) of lane ln holds W2[32 fb + (ln & 31)][16 s + 8 (ln >> 5) + j], j = 0..7 (classes past p: zero)
    for (int idx = threadIdx.x; idx < NKS * 4 * 64; idx += 512) {
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

    const u32x4 rs_z = bx_make_rsrc(a.dz, (uint32_t)a.m * (uint32_t)a.lddz * 4u);    // rows past m read as zero: they add nothing to dW2
    const u32x4 rs_h = bx_make_rsrc(a.h1, (uint32_t)a.m * (uint32_t)a.ldh * 4u);
    const u32x4 rs_b = bx_make_rsrc(a.bits, (uint32_t)a.m * 16u);
    const u32x4 rs_s = bx_make_rsrc(a.rowscale ? a.rowscale : a.dz, a.rowscale ? (uint32_t)a.m * 4u : 0u);   // no factors: every load is out of range = 0
    uint32_t so_z[8], so_h[8];                               // row j of a lane's eight: scalar byte offsets
#pragma unroll
    for (int j = 0; j < 8; j++) { so_z[j] = (uint32_t)j * (uint32_t)a.lddz * 4u; so_h[j] = (uint32_t)j * (uint32_t)a.ldh * 4u; }

    f32x4 Z[2 * NKS];                                        // L1: this lane's row of dZ0, 8 floats per k-step
    f32x4 KB;                                                // the row's four mask words
    float RSC;                                               // its factor
    auto issue_l1 = [&](int b) __attribute__((always_inline)) {
        const uint32_t row = (uint32_t)(b * 32 + li);
        const uint32_t vz = (row * (uint32_t)a.lddz + 8u * hh) * 4u;
        if constexpr (ABL & 4) { bx_bload16<0>(KB, row * 16u, rs_b); bx_bload4(RSC, row * 4u, rs_s, 0u); }      // (experiment: the small loads first)
        bx_bload16<0>(Z[0], vz, rs_z); bx_bload16<16>(Z[1], vz, rs_z);
        if constexpr (NKS > 1) { bx_bload16<64>(Z[2], vz, rs_z); bx_bload16<80>(Z[3], vz, rs_z); }
        if constexpr (NKS > 2) { bx_bload16<128>(Z[4], vz, rs_z); bx_bload16<144>(Z[5], vz, rs_z); }
        if constexpr (NKS > 3) { bx_bload16<192>(Z[6], vz, rs_z); bx_bload16<208>(Z[7], vz, rs_z); }
        if constexpr (!(ABL & 4)) { bx_bload16<0>(KB, row * 16u, rs_b); bx_bload4(RSC, row * 4u, rs_s, 0u); }
    };
    constexpr int NL1 = 2 * NKS + 2;
    ClsCol Cc;
    auto issue_col = [&](int b, int s2) __attribute__((always_inline)) {
        const uint32_t r0 = (uint32_t)(b * 32 + 16 * s2 + 8 * hh);
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            const uint32_t c = 32u * cb + li;
            // classes past the row's allocation must not be read as the next row's: they are sent past the descriptor (zero)
            const uint32_t vo = c < (uint32_t)a.lddz ? (r0 * (uint32_t)a.lddz + c) * 4u : 0xFFFFF000u;
#pragma unroll
            for (int j = 0; j < 8; j++) bx_bload4(Cc.z[cb][j], vo, rs_z, so_z[j]);
        }
#pragma unroll
        for (int f2 = 0; f2 < 2; f2++) {
            const uint32_t vo = (r0 * (uint32_t)a.ldh + 32u * (2 * half + f2) + li) * 4u;
#pragma unroll
            for (int j = 0; j < 8; j++) bx_bload4(Cc.h[f2][j], vo, rs_h, so_h[j]);
        }
    };

    f32x16 dw[2][2];                                         // dW2[32 (2 half + f2) + ..][32 cb + li]: this wave's share over its row blocks
#pragma unroll
    for (int f2 = 0; f2 < 2; f2++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) dw[f2][cb][r] = 0.f;

    const int stride = gridDim.x * CLS_BWD_PAIRS;
    int rb = blockIdx.x * CLS_BWD_PAIRS + pair;
    issue_l1(rb);
    for (; rb < a.n_rb; rb += stride) {
        issue_col(rb, 0);
        // ---- dH1^T[feature][row] = sum_c W2[feature][c] . dZ0[row][c]: dZ0 in its memory layout as the B operand
        asm volatile("s_waitcnt vmcnt(32)" : "+v"(KB), "+v"(RSC) :: "memory");           // L1 has 32 younger loads (C0)
#pragma unroll
        for (int i = 0; i < 2 * NKS; i++) asm volatile("" : "+v"(Z[i]));
        const uint32_t kw[2] = {__float_as_uint(KB[2 * half]), __float_as_uint(KB[2 * half + 1])};
        const float sc = a.rowscale ? a.scale * RSC : a.scale;
        f32x16 acc[2];
#pragma unroll
        for (int f2 = 0; f2 < 2; f2++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[f2][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NKS; s++) {
            float v[8] = {Z[2 * s][0], Z[2 * s][1], Z[2 * s][2], Z[2 * s][3], Z[2 * s + 1][0], Z[2 * s + 1][1], Z[2 * s + 1][2], Z[2 * s + 1][3]};
            const int c0 = 16 * s + 8 * hh;
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = c0 + j < a.p ? v[j] : 0.f;              // padding columns may hold anything
            if constexpr (ABL & 1) { acc[0][s] += v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7]; continue; }
            const ClsB3 B = cls_planes(v);
#pragma unroll
            for (int f2 = 0; f2 < 2; f2++) cls_mac(acc[f2], B, cls_lds(cls_img, s * 4 + 2 * half + f2, lane));
        }
        if ((ABL & 2) ? (acc[0][0] == 123.456f) : true) {
            // D[row = (r & 3) + 8 (r >> 2) + 4 hh][feature = 32 fb + li]: lane li holds the mask words and the factor of row li
            // (both halves of the wave); a store instruction writes 32 consecutive features — one full line — of two rows.
            // (As 16-byte lane stores of a transposed tile the 119 MB of dH1 left in 7.4 M separate requests: 40 us of a 95 us launch.)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh, orow = rb * 32 + rl;
                const float scr = __shfl(sc, rl, WAVE);
                const uint32_t w0 = __shfl(kw[0], rl, WAVE), w1 = __shfl(kw[1], rl, WAVE);
                if (orow < a.m) {
                    float *dp = a.da + (size_t)orow * a.ldda + 64 * half + li;
                    dp[0] = ((w0 >> li) & 1u) ? acc[0][r] * scr : 0.f;
                    dp[32] = ((w1 >> li) & 1u) ? acc[1][r] * scr : 0.f;
                }
            }
        }
        issue_l1(rb + stride);                               // the next block's row pieces travel under the second phase
        // ---- dW2[feature][c] += sum_rows H1[row][feature] . dZ0[row][c]: both operands column-wise (k = row)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            if (s2 == 0) CLS_WAIT_COL(NL1, Cc);              // younger than C0: the next block's L1
            else CLS_WAIT_COL(0, Cc);
            if constexpr (ABL & 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) dw[0][0][j] += Cc.z[0][j] + Cc.z[1][j] + Cc.h[0][j] + Cc.h[1][j];
                if (s2 == 0) issue_col(rb, 1);
                continue;
            }
            ClsB3 Bz[2], Ah[2];
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = (32 * cb + li < a.p) ? Cc.z[cb][j] : 0.f;   // classes past p: padding
                Bz[cb] = cls_planes(v);
            }
#pragma unroll
            for (int f2 = 0; f2 < 2; f2++) Ah[f2] = cls_planes(Cc.h[f2]);
            if (s2 == 0) issue_col(rb, 1);                   // into the registers just turned into planes
#pragma unroll
            for (int f2 = 0; f2 < 2; f2++) {
                cls_mac(dw[f2][0], Ah[f2], Bz[0]);
                if (a.p > 32) cls_mac(dw[f2][1], Ah[f2], Bz[1]);
            }
        }
        // NO load is in flight across the back edge (see the forward kernel): the next block's L1 has landed with C1's vmcnt(0);
        // its registers are tied to a wait here so that hipcc's back-edge copies come after it
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(KB), "+v"(RSC) :: "memory");
#pragma unroll
        for (int i = 0; i < 2 * NKS; i++) asm volatile("" : "+v"(Z[i]));
    }
    // (waves that never entered the loop: the prologue's L1)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(KB), "+v"(RSC) :: "memory");
#pragma unroll
    for (int i = 0; i < 2 * NKS; i++) asm volatile("" : "+v"(Z[i]));
    // ---- the workgroup's sum, pair after pair (fixed order; the two waves of a pair own different features), one slab per workgroup
    for (int w = 0; w < CLS_BWD_PAIRS; w++) {
        if (pair == w) {
#pragma unroll
            for (int f2 = 0; f2 < 2; f2++)
#pragma unroll
                for (int cb = 0; cb < 2; cb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int f = 32 * (2 * half + f2) + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        red[f * 64 + 32 * cb + li] += dw[f2][cb][r];
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
    return n == 128 && p >= 1 && p <= 64 && m >= 2048 && lda % 4 == 0 && lda >= 128 && ldc % 4 == 0 && aligned16(a) && aligned16(c) &&
           (uint64_t)m * lda * 4 < (1ull << 32);
}
static inline bool cls_bwd_fits(const float *a, int lda, const float *dc, int lddc, const float *da, int ldda, const uint32_t *bits, int wpr,
                                int m, int n, int p) {
    return n == 128 && p >= 1 && p <= 64 && m >= 2048 && wpr == 4 && bits && aligned16(bits) && a && lda >= 128 && lddc % 4 == 0 &&
           lddc >= (p + 15) / 16 * 16 && ldda % 4 == 0 && ldda >= 128 && aligned16(dc) && aligned16(da) &&
           (uint64_t)m * lda * 4 < (1ull << 32) && (uint64_t)m * lddc * 4 < (1ull << 32);
}

static int launch_class_fwd(gcnhip_ctx *c, const float *a, int lda, const float *b, int ldb, float *z, int ldz, int m, int p) {
    ClsFwdArgs k;
    k.h1 = a; k.ldh = lda; k.w2 = b; k.ldw = ldb; k.z0 = z; k.ldz = ldz; k.m = m; k.p = p; k.n_rb = ceil_div(m, 32);
    int grid = ceil_div(k.n_rb, 4);
    const int per_cu = c->opt.cls_wgs > 0 ? c->opt.cls_wgs : 3;     // 48 KB of LDS per workgroup: three per CU
    if (grid > c->n_cu * per_cu) grid = c->n_cu * per_cu;
    switch (c->opt.cls_abl & 3) {
        case 1: class_fwd_bf16x3_kernel<1><<<grid, 256, CLS_FWD_LDS, c->stream>>>(k); break;
        case 2: class_fwd_bf16x3_kernel<2><<<grid, 256, CLS_FWD_LDS, c->stream>>>(k); break;
        case 3: class_fwd_bf16x3_kernel<3><<<grid, 256, CLS_FWD_LDS, c->stream>>>(k); break;
        default: class_fwd_bf16x3_kernel<0><<<grid, 256, CLS_FWD_LDS, c->stream>>>(k); break;
    }
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

// dense_kernels.h — f32 MFMA building blocks shared by matmul.hip and spmm.hip.
//
// gfx950 has an exact-f32 matrix instruction, v_mfma_f32_16x16x4_f32 (one f32
// VGPR per operand per lane; result bit-identical to a k-ordered fmaf chain).
// Operand maps (cdna guide §3):  A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15],  D[row = 4*(lane>>4)+reg][col = lane&15].
//
// Two kernels:
//  * gemm_rowstream: C[m x Nc] = A[m x K] . Bs[K x Nc], m huge, K and Nc small.
//    Bs lives in LDS for the whole (persistent) workgroup; A streams from HBM
//    once, 16 rows per wave step, 16 B per lane along K.  Serves Matmul
//    forward (Bs = W2) and dA (Bs = W2^T), optional ReLU/dropout-backward
//    epilogue.
//  * gemm_atb: S[n x p] = A^T . Bm with the long dimension m as K.  Each wave
//    is an independent split-K worker over a row range; a lane's vector load
//    of VA (VB) consecutive columns feeds VA x VB MFMAs whose outputs are the
//    column-interleaved 16x16 tiles {VA*i+s, VB*c+u}.  Workers write partial
//    slabs; a second kernel sums them in worker order (bitwise reproducible,
//    no float atomics — the reference's CUDA scatter races here,
//    cuda_kernel.cu:112-122).  Serves dW2 = H1^T.dZ0 and dW1 = X~^T.dH0 for
//    dense X, with the input dropout re-derived on the fly.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------ gemm_rowstream
struct RowStreamArgs {
    const float *A; int lda;
    const float *B; int ldb; int transB;     // Bs[k][c] = transB ? B[c*ldb+k] : B[k*ldb+c]
    float *C; int ldc;
    int m, K, Nc;
    // epilogue: C = H > 0 ? scale*C : 0
    const float *H; int ldh; float scale;
    const float *rowscale;                      // optional: row r's results are multiplied by scale * rowscale[r] (the factored aggregation's dinv^2)
    const uint32_t *hbits; int wpr;             // alternative mask source: bit (c & 31) of hbits[r*wpr + (c >> 5)] = (H[r,c] > 0)
    int vec_out;                                // C (and H) rows are 16-byte aligned: LDS-staged row stores
    // PACK: rows leave as packed half-row slots (rowpack.h) instead of dense rows; C receives only the rows that do not fit
    uint32_t *pack_slots; int pack_halves;      // [m x pack_halves x 32] dwords
};

// ---- packed rows (exact): a [rows x cols] matrix whose entries are mostly zero by a KNOWN mask ----------------
// cols is cut into halves of 64 columns; half h of row r owns one 128-byte slot:
//     dword 0-1 : the 64-bit mask of the half (bit c = column 64h + c may be non-zero)
//     dword 2.. : the masked-in values in column order (at most PACK_CAP = 30)
// A half with more than 30 masked-in columns does not fit: its slot carries only the mask and the row's dense
// image (the caller's ordinary [rows x ld] buffer) holds that half.  One line per (row, half) instead of two.
constexpr int PACK_CAP = 30;

// bits 0..15 of x to bit positions 0, 4, 8, ... 60
__device__ inline uint64_t spread4(uint32_t x) {
    uint64_t v = x & 0xFFFFu;
    v = (v | (v << 24)) & 0x000000FF000000FFull;
    v = (v | (v << 12)) & 0x000F000F000F000Full;
    v = (v | (v << 6)) & 0x0303030303030303ull;
    v = (v | (v << 3)) & 0x1111111111111111ull;
    return v;
}

// KCH > 0: the K extent is KCH chunks of 16 (K <= 128) and a wave issues all of its KCH
// 16-byte loads of a row tile before the first MFMA (KCH KiB in flight per wave instead of
// one dependent load per chunk).  KCH == 0: generic loop for longer K.
// BITS (SW kernels with a mask): the mask comes from a.hbits, not from a.H — a template argument so that only one of the
// two prefetch register sets exists in a kernel.
template <int NT, bool VEC, bool FUSE, int KCH, bool PACK = false, bool BITS = false>
__global__ __launch_bounds__(256) void gemm_rowstream_kernel(RowStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];
    constexpr int NCLD = NT * 16 + 4;        // +4: the four k-groups of a wave hit disjoint banks
    // SW (round 3): the two MFMA operands change places, so the instruction computes the TRANSPOSED 16x16 tile — the same
    // products added in the same k order, bit for bit — and a lane ends up with 4 consecutive COLUMNS of one row
    // (D[4*(lane>>4)+reg][lane&15] is now [column][row]) instead of one column of 4 rows.  Rows then leave as 16-byte lane
    // stores and the mask arrives as 16-byte loads / one bit word straight from the accumulators: the LDS staging of the
    // result, its two wave barriers and 34 of dA's 59 KB of LDS are gone (2 -> 4 workgroups per CU).
    constexpr bool SW = VEC && !PACK;
    const int Kp = (a.K + 15) / 16 * 16;
    const int c_base = blockIdx.y * NT * 16;
    // The small operand into LDS, 8 elements per thread in flight: unconditional loads from clamped addresses (element 0
    // for the padding), zeroed by selects.  As a loop of one predicated load per iteration this prologue was 24 dependent
    // round trips per workgroup (5 us of a 48 us launch at Reddit scale).
    {
        const int total = Kp * NT * 16;
        for (int base = threadIdx.x; base < total; base += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = base + u * 256;
                const int k = idx / (NT * 16), c = idx % (NT * 16);
                const int gc = c_base + c;
                const bool in = idx < total && k < a.K && gc < a.Nc;
                const size_t off = in ? (a.transB ? (size_t)gc * a.ldb + k : (size_t)k * a.ldb + gc) : 0;
                const float x = a.B[off];
                v[u] = in ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = base + u * 256;
                if (idx < total) Bs[(idx / (NT * 16)) * NCLD + idx % (NT * 16)] = v[u];
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int n_tiles = (a.m + 15) / 16;
    auto load4 = [&](const float *ap, int kk, bool valid, float av[4]) {
        av[0] = av[1] = av[2] = av[3] = 0.f;
        if (valid && kk < a.K) {
            if (VEC) {
                const float4 v = *reinterpret_cast<const float4 *>(ap + kk);
                av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; s++) if (kk + s < a.K) av[s] = ap[kk + s];
            }
#pragma unroll
            for (int s = 0; s < 4; s++) if (kk + s >= a.K) av[s] = 0.f;   // pad columns may hold anything
        }
    };
    constexpr int NW = (NT + 1) / 2;                          // 32-column mask words per row of this column block
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        // Bs is loop-invariant and, without the staged epilogue, nothing in the loop writes LDS: left alone the compiler
        // keeps the WHOLE operand in registers (96 per lane; 226 VGPRs, one or two waves per SIMD) — measured slower
        // (H1.W2 38 -> 53 us, dA+dW2 125 -> 150 us): this kernel lives on waves in flight.  Re-read it per tile.
        asm volatile("" ::: "memory");
        const int row = tile * 16 + li;
        const bool valid = row < a.m;
        const float *ap = a.A + (size_t)row * a.lda;
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // SW + mask: the row's mask words (or its H quads) are requested HERE, from clamped addresses and without a
        // branch, so they travel beside the A loads and under the MFMAs.  As loads inside the store loop they were NT
        // dependent round trips per tile (load, wait, store, next column block).
        uint32_t kbw[BITS ? NW : 1];
        float4 hq[BITS ? 1 : NT];
        if constexpr (SW && FUSE) {
            const size_t rc = (size_t)min(row, a.m - 1);
            if constexpr (BITS) {                              // c_base % 32 == 0 (NT is even whenever there is a second block)
                const uint32_t *hb = a.hbits + rc * a.wpr;
#pragma unroll
                for (int j = 0; j < NW; j++) {
                    const int w = (c_base >> 5) + j;
                    kbw[j] = hb[w < a.wpr ? w : 0];
                }
            } else if (a.vec_out) {
                const float *hrow = a.H + rc * a.ldh;
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int col = c_base + t * 16 + 4 * kq;
                    hq[t] = *reinterpret_cast<const float4 *>(hrow + (col + 4 <= a.Nc ? col : 0));
                }
            }
        }
        if (KCH > 0) {
            float av[KCH > 0 ? KCH : 1][4];
            if (VEC && a.lda >= KCH * 16) {
                // Every float4 of the K extent lies inside the row's allocation (wave-uniform test): the KCH loads are
                // issued UNCONDITIONALLY from a clamped row — a load inside a per-lane branch is waited for at the end
                // of that branch, which kept ONE load in flight per wave here until round 2 (the same disease as
                // GraphSum's; H1.W2 at Reddit scale: 55 -> 42 us with the K = 128 case batched).  Rows
                // past m compute on a copy of the last row and are never stored; columns past K are zeroed by selects.
                // (Round 3, measured and dropped: issuing the NEXT tile's loads before this tile's MFMAs.  The extra
                // registers cost a wave per SIMD and the launch got slower, 38 -> 45 us: waves in flight hide the
                // round trip better than a wave's own prefetch.)
                const float *apc = a.A + (size_t)min(row, a.m - 1) * a.lda + 4 * kq;
                float4 raw[KCH > 0 ? KCH : 1];
#pragma unroll
                for (int c = 0; c < KCH; c++) raw[c] = *reinterpret_cast<const float4 *>(apc + c * 16);
#pragma unroll
                for (int c = 0; c < KCH; c++) {
                    const int kk = c * 16 + 4 * kq;
                    av[c][0] = kk + 0 < a.K ? raw[c].x : 0.f; av[c][1] = kk + 1 < a.K ? raw[c].y : 0.f;
                    av[c][2] = kk + 2 < a.K ? raw[c].z : 0.f; av[c][3] = kk + 3 < a.K ? raw[c].w : 0.f;
                }
            } else {
#pragma unroll
                for (int c = 0; c < KCH; c++) load4(ap, c * 16 + 4 * kq, valid, av[c]);
            }
#pragma unroll
            for (int c = 0; c < KCH; c++) {
                const int kk = c * 16 + 4 * kq;
#pragma unroll
                for (int s = 0; s < 4; s++) {
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const float b = Bs[(kk + s) * NCLD + t * 16 + li];
                        acc[t] = SW ? __builtin_amdgcn_mfma_f32_16x16x4f32(b, av[c][s], acc[t], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][s], b, acc[t], 0, 0, 0);
                    }
                }
                if (SW) __builtin_amdgcn_sched_barrier(0);      // one chunk's 4*NT LDS reads in flight, not all KCH*4*NT of them
            }
        } else {
            for (int k0 = 0; k0 < Kp; k0 += 16) {
                const int kk = k0 + 4 * kq;
                float av[4];
                load4(ap, kk, valid, av);
#pragma unroll
                for (int s = 0; s < 4; s++) {
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const float b = Bs[(kk + s) * NCLD + t * 16 + li];
                        acc[t] = SW ? __builtin_amdgcn_mfma_f32_16x16x4f32(b, av[s], acc[t], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (SW) {
            const int r = tile * 16 + li;
            if (r < a.m) {
                float *crow = a.C + (size_t)r * a.ldc;
                const float sc = (FUSE && a.rowscale) ? a.scale * a.rowscale[r] : a.scale;
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int col = c_base + t * 16 + 4 * kq;
                    if (col >= a.Nc) continue;
                    float x[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
                    if constexpr (FUSE && BITS) {             // col % 4 == 0: the quad's four bits sit in one word
                        const uint32_t kb = kbw[t >> 1] >> ((t & 1) * 16 + 4 * kq);
#pragma unroll
                        for (int q = 0; q < 4; q++) x[q] = ((kb >> q) & 1u) ? x[q] * sc : 0.f;
                    }
                    if (a.vec_out && col + 4 <= a.Nc) {
                        if constexpr (FUSE && !BITS) {
                            x[0] = hq[t].x > 0.f ? x[0] * sc : 0.f; x[1] = hq[t].y > 0.f ? x[1] * sc : 0.f;
                            x[2] = hq[t].z > 0.f ? x[2] * sc : 0.f; x[3] = hq[t].w > 0.f ? x[3] * sc : 0.f;
                        }
                        *reinterpret_cast<float4 *>(crow + col) = make_float4(x[0], x[1], x[2], x[3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++) {           // fully unrolled: x[] stays in registers
                            if (col + q >= a.Nc) continue;
                            float y = x[q];
                            if constexpr (FUSE && !BITS) y = a.H[(size_t)r * a.ldh + col + q] > 0.f ? y * sc : 0.f;
                            crow[col + q] = y;
                        }
                    }
                }
            }
        } else if (a.vec_out) {
            // Stage the 16 x (NT*16) result through LDS so rows leave as whole 16-byte lane
            // stores (and the mask operand H arrives as 16-byte loads): a lane owns 4
            // consecutive columns of one row instead of 1 column of 4 rows.
            float *Cs = Bs + Kp * NCLD + wave * 16 * (NT * 16 + 4);
            constexpr int CLD = NT * 16 + 4;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) Cs[(4 * kq + i) * CLD + t * 16 + li] = acc[t][i];
            __builtin_amdgcn_wave_barrier();
            constexpr int LPRW = NT * 4;                      // lanes per row (float4 each)
            constexpr int RPP = 64 / LPRW > 0 ? 64 / LPRW : 1; // rows per pass
            for (int r0 = 0; r0 < 16; r0 += RPP) {
                const int rr = r0 + lane / LPRW, cc = (lane % LPRW) * 4;
                const int r = tile * 16 + rr, col = c_base + cc;
                if constexpr (PACK) {
                    // FUSE, whole 64-column halves (Nc % 64 == 0), LPRW = 16 or 32 lanes per row.  Every lane of
                    // the wave takes part in the ballots; lanes of rows past m contribute zero bits.
                    static_assert(!PACK || (FUSE && (NT == 4 || NT == 8)), "packed rows need the mask and whole halves");
                    const bool live = r < a.m;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    bool k0 = false, k1 = false, k2 = false, k3 = false;
                    if (live) {
                        v = *reinterpret_cast<const float4 *>(&Cs[rr * CLD + cc]);
                        if (a.hbits) {
                            const uint32_t kb = a.hbits[(size_t)r * a.wpr + (col >> 5)] >> (col & 31);
                            k0 = kb & 1u; k1 = kb & 2u; k2 = kb & 4u; k3 = kb & 8u;
                        } else {
                            const float4 h = *reinterpret_cast<const float4 *>(a.H + (size_t)r * a.ldh + col);
                            k0 = h.x > 0.f; k1 = h.y > 0.f; k2 = h.z > 0.f; k3 = h.w > 0.f;
                        }
                        v.x = k0 ? v.x * a.scale : 0.f; v.y = k1 ? v.y * a.scale : 0.f;
                        v.z = k2 ? v.z * a.scale : 0.f; v.w = k3 ? v.w * a.scale : 0.f;
                    }
                    const int rg = lane / LPRW, lr = lane % LPRW, hb = lr >> 4, lh = lr & 15;
                    const uint32_t rowmask = LPRW == 32 ? 0xFFFFFFFFu : 0xFFFFu;
                    const uint32_t b0 = (uint32_t)(__ballot(k0) >> (LPRW * rg)) & rowmask, b1 = (uint32_t)(__ballot(k1) >> (LPRW * rg)) & rowmask;
                    const uint32_t b2 = (uint32_t)(__ballot(k2) >> (LPRW * rg)) & rowmask, b3 = (uint32_t)(__ballot(k3) >> (LPRW * rg)) & rowmask;
                    const uint32_t hsel = 0xFFFFu << (16 * hb), below = ((1u << lh) - 1u) << (16 * hb);
                    const int n_half = __popc(b0 & hsel) + __popc(b1 & hsel) + __popc(b2 & hsel) + __popc(b3 & hsel);
                    if (live) {
                        const int half = (col >> 6);
                        uint32_t *slot = a.pack_slots + ((size_t)r * a.pack_halves + half) * 32;
                        if (lh == 0) {
                            const uint64_t m64 = spread4(b0 >> (16 * hb)) | (spread4(b1 >> (16 * hb)) << 1) |
                                                 (spread4(b2 >> (16 * hb)) << 2) | (spread4(b3 >> (16 * hb)) << 3);
                            *reinterpret_cast<uint2 *>(slot) = make_uint2((uint32_t)m64, (uint32_t)(m64 >> 32));
                        }
                        if (n_half <= PACK_CAP) {
                            int off = 2 + __popc(b0 & below) + __popc(b1 & below) + __popc(b2 & below) + __popc(b3 & below);
                            if (k0) slot[off++] = __float_as_uint(v.x);
                            if (k1) slot[off++] = __float_as_uint(v.y);
                            if (k2) slot[off++] = __float_as_uint(v.z);
                            if (k3) slot[off++] = __float_as_uint(v.w);
                        } else {
                            *reinterpret_cast<float4 *>(a.C + (size_t)r * a.ldc + col) = v;   // the half does not fit: dense image
                        }
                    }
                } else
                if (lane < RPP * LPRW && rr < 16 && r < a.m && col < a.Nc) {
                    float4 v = *reinterpret_cast<const float4 *>(&Cs[rr * CLD + cc]);
                    float *cp = a.C + (size_t)r * a.ldc + col;
                    const float sc = (FUSE && a.rowscale) ? a.scale * a.rowscale[r] : a.scale;
                    if (col + 4 <= a.Nc) {
                        if (FUSE) {
                            if (a.hbits) {                  // col % 4 == 0: the four bits sit in one word
                                const uint32_t kb = a.hbits[(size_t)r * a.wpr + (col >> 5)] >> (col & 31);
                                v.x = (kb & 1u) ? v.x * sc : 0.f; v.y = (kb & 2u) ? v.y * sc : 0.f;
                                v.z = (kb & 4u) ? v.z * sc : 0.f; v.w = (kb & 8u) ? v.w * sc : 0.f;
                            } else {
                                const float4 h = *reinterpret_cast<const float4 *>(a.H + (size_t)r * a.ldh + col);
                                v.x = h.x > 0.f ? v.x * sc : 0.f; v.y = h.y > 0.f ? v.y * sc : 0.f;
                                v.z = h.z > 0.f ? v.z * sc : 0.f; v.w = h.w > 0.f ? v.w * sc : 0.f;
                            }
                        }
                        *reinterpret_cast<float4 *>(cp) = v;
                    } else {
                        const float x[4] = {v.x, v.y, v.z, v.w};
                        for (int q = 0; col + q < a.Nc; q++) {
                            float y = x[q];
                            if (FUSE) {
                                const int cq = col + q;
                                const bool pos = a.hbits ? ((a.hbits[(size_t)r * a.wpr + (cq >> 5)] >> (cq & 31)) & 1u) != 0
                                                         : a.H[(size_t)r * a.ldh + cq] > 0.f;
                                y = pos ? y * sc : 0.f;
                            }
                            cp[q] = y;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int col = c_base + t * 16 + li;
                if (col >= a.Nc) continue;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int r = tile * 16 + 4 * kq + i;
                    if (r >= a.m) continue;
                    float v = acc[t][i];
                    if (FUSE) {
                        const bool pos = a.hbits ? ((a.hbits[(size_t)r * a.wpr + (col >> 5)] >> (col & 31)) & 1u) != 0
                                                 : a.H[(size_t)r * a.ldh + col] > 0.f;
                        v = pos ? v * ((a.rowscale ? a.rowscale[r] : 1.f) * a.scale) : 0.f;
                    }
                    a.C[(size_t)r * a.ldc + col] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ gemm_atb
struct AtbArgs {
    const float *A; int lda;      // m x n
    const float *Bm; int ldb;     // m x p
    float *slab;                  // [workers][n][p_ld]
    int m, n, p, p_ld;
    int rows_per_worker, n_workers;
    // dropout on A (the input dropout of a dense X): element index = row*n + col
    int drop; int thr; float scale;
    uint64_t seed, off; const uint32_t *d_epoch; const uint8_t *keep_mask;
    const uint32_t *bits;         // optional: precomputed keep bits, bit (e & 31) of word (e >> 5), e = row*n + col
};

template <int V>
__device__ inline void load_vec(const float *p, int valid_cols, float out[V]) {
    // p is aligned to V floats; valid_cols = how many of the V columns exist
    if (valid_cols >= V) {
        if (V == 4) { const float4 v = *reinterpret_cast<const float4 *>(p); out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
        else if (V == 2) { const float2 v = *reinterpret_cast<const float2 *>(p); out[0] = v.x; out[1] = v.y; }
        else out[0] = *p;
    } else {
#pragma unroll
        for (int s = 0; s < V; s++) out[s] = s < valid_cols ? p[s] : 0.f;
    }
}

template <int VA, int VB>
__global__ __launch_bounds__(256) void gemm_atb_kernel(AtbArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int worker = blockIdx.x * 4 + wave;            // idle waves (past n_workers) still reach the barrier below
    const int colA = blockIdx.y * 16 * VA + VA * li;       // this lane's first A column
    const int colB = blockIdx.z * 16 * VB + VB * li;
    const int r0 = min(a.m, worker * a.rows_per_worker);
    const int r1 = min(a.m, r0 + a.rows_per_worker);
    const uint32_t epoch = (a.drop && a.d_epoch) ? *a.d_epoch : 0u;
    f32x4 acc[VA][VB];
#pragma unroll
    for (int s = 0; s < VA; s++)
#pragma unroll
        for (int u = 0; u < VB; u++) acc[s][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int validA = a.n - colA, validB = a.p - colB;
    int k0 = r0;
    // ATB_U K-steps (4 rows each) with their 2 * ATB_U row loads in flight together.  The `#pragma unroll 8` that used
    // to stand here never applied (loads inside `if (row < r1)`): one K-step per round trip.  Needs whole steps inside the
    // worker's range, no dropout, and rows padded to whole lane vectors (so a lane with fewer than VA / VB valid columns
    // may read its full vector and zero the rest).
    constexpr int ATB_U = 4;
    if (!a.drop && a.lda >= (a.n + VA - 1) / VA * VA && a.ldb >= (a.p + VB - 1) / VB * VB) {
        const float *pa = a.A + min(colA, max(a.n - 1, 0) / VA * VA), *pb = a.Bm + min(colB, max(a.p - 1, 0) / VB * VB);
        // Software-pipelined: the loads of batch i+1 are issued (unconditionally, rows clamped to the matrix) before the
        // MFMAs of batch i — 64 MFMAs = 0.85 us of pipe time per batch, as long as a memory round trip.  Worth 3 us of 52
        // at Reddit scale (dW2 = H1^T . dZ0): the kernel sits near both of its floors there — 24 us of MFMA time (41
        // columns padded to 64) and 21-26 us of HBM time.
        float av[ATB_U][VA], bv[ATB_U][VB];
        auto load_batch = [&](int k, float (&x)[ATB_U][VA], float (&y)[ATB_U][VB]) {
#pragma unroll
            for (int q = 0; q < ATB_U; q++) {
                const size_t row = (size_t)min(k + 4 * q + kq, a.m - 1);
                load_vec<VA>(pa + row * a.lda, VA, x[q]);
                load_vec<VB>(pb + row * a.ldb, VB, y[q]);
            }
        };
        auto mfma_batch = [&](float (&x)[ATB_U][VA], float (&y)[ATB_U][VB]) {
#pragma unroll
            for (int q = 0; q < ATB_U; q++) {
#pragma unroll
                for (int s = 0; s < VA; s++) if (s >= validA) x[q][s] = 0.f;
#pragma unroll
                for (int u = 0; u < VB; u++) if (u >= validB) y[q][u] = 0.f;
#pragma unroll
                for (int s = 0; s < VA; s++)
#pragma unroll
                    for (int u = 0; u < VB; u++)
                        acc[s][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q][s], y[q][u], acc[s][u], 0, 0, 0);
            }
        };
        // two register sets, alternating (no copies: a copy of the incoming set would wait for its loads in the middle of
        // the MFMAs); a set loaded past the worker's range is simply never used — its rows are valid memory either way
        float an[ATB_U][VA], bn[ATB_U][VB];
        constexpr int STEP = 4 * ATB_U;
        load_batch(k0, av, bv);
        while (true) {
            if (k0 + STEP > r1) break;
            load_batch(k0 + STEP, an, bn);
            __builtin_amdgcn_sched_barrier(0);         // keep the loads above: the scheduler otherwise sinks each to its use
            mfma_batch(av, bv);
            k0 += STEP;
            if (k0 + STEP > r1) break;
            load_batch(k0 + STEP, av, bv);
            __builtin_amdgcn_sched_barrier(0);
            mfma_batch(an, bn);
            k0 += STEP;
        }
    }
    for (; k0 < r1; k0 += 4) {
        const int row = k0 + kq;
        float av[VA], bv[VB];
#pragma unroll
        for (int s = 0; s < VA; s++) av[s] = 0.f;
#pragma unroll
        for (int u = 0; u < VB; u++) bv[u] = 0.f;
        if (row < r1) {
            if (validA > 0) load_vec<VA>(a.A + (size_t)row * a.lda + colA, validA, av);
            if (validB > 0) load_vec<VB>(a.Bm + (size_t)row * a.ldb + colB, validB, bv);
            if (a.drop && validA > 0) {
                const uint64_t e0 = (uint64_t)row * a.n + colA;
                uint32_t bits = 0;
                if (a.bits) {                                   // built once per call, 128 decisions per Philox block
                    const uint32_t sh = (uint32_t)(e0 & 31);
                    bits = a.bits[e0 >> 5] >> sh;
                    if (VA > 1 && sh + VA > 32) bits |= a.bits[(e0 >> 5) + 1] << (32 - sh);
                } else if (a.keep_mask) {
#pragma unroll
                    for (int s = 0; s < VA; s++) bits |= (s < validA && a.keep_mask[e0 + s] != 0 ? 1u : 0u) << s;
                } else if (((a.off + e0) & (VA - 1)) == 0) {
                    bits = keepv<VA>(a.off + e0, epoch, a.seed, a.thr);       // one Philox block for the vector
                } else {
#pragma unroll
                    for (int s = 0; s < VA; s++) bits |= (keep1(a.off + e0 + s, epoch, a.seed, a.thr) ? 1u : 0u) << s;
                }
#pragma unroll
                for (int s = 0; s < VA; s++) av[s] *= (bits >> s & 1u) ? a.scale : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < VA; s++)
#pragma unroll
            for (int u = 0; u < VB; u++)
                acc[s][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[u], acc[s][u], 0, 0, 0);
    }
    // The four waves of a workgroup are consecutive row ranges of the same output block: their partials are added
    // here in wave order, ((w0 + w1) + w2) + w3, and ONE slab per workgroup goes to memory — four times the
    // waves in flight for the same slab traffic (a worker with too few peers on its SIMD waits out every load).
    __shared__ float red[3 * 64 * VA * VB * 4];
    if (wave > 0) {
        float *mine = red + (size_t)(wave - 1) * 64 * VA * VB * 4;
#pragma unroll
        for (int s = 0; s < VA; s++)
#pragma unroll
            for (int u = 0; u < VB; u++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) mine[((s * VB + u) * 4 + reg) * 64 + lane] = acc[s][u][reg];
    }
    __syncthreads();
    if (wave > 0) return;
    const int live = min(4, a.n_workers - blockIdx.x * 4);          // waves of this workgroup that had rows
#pragma unroll
    for (int w = 1; w < 4; w++) {
        if (w >= live) break;
        const float *theirs = red + (size_t)(w - 1) * 64 * VA * VB * 4;
#pragma unroll
        for (int s = 0; s < VA; s++)
#pragma unroll
            for (int u = 0; u < VB; u++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) acc[s][u][reg] += theirs[((s * VB + u) * 4 + reg) * 64 + lane];
    }
    // D_{s,u}[i][c] = S[blockA + VA*i + s][blockB + VB*c + u]
    float *slab = a.slab + (size_t)blockIdx.x * a.n * a.p_ld;
#pragma unroll
    for (int s = 0; s < VA; s++)
#pragma unroll
        for (int u = 0; u < VB; u++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int nidx = blockIdx.y * 16 * VA + VA * (4 * kq + reg) + s;
                const int pidx = blockIdx.z * 16 * VB + VB * li + u;
                if (nidx < a.n && pidx < a.p) slab[(size_t)nidx * a.p_ld + pidx] = acc[s][u][reg];
            }
}

// out[nidx][pidx] = sum over workers.  Each output is summed by CH = 256 / OPB threads over CH contiguous worker
// ranges; the CH partials are added in range order (fixed order: bitwise reproducible).  OPB outputs per block: 64 when
// there are many outputs (dW1: 77 K), fewer when there are few outputs and many workers (dW2: 5 K outputs, 503 slabs —
// with 64 per block that was 82 blocks of threads walking 126 slabs each, 34 us; see launch_slab_reduce).
template <int OPB>
static __global__ __launch_bounds__(256) void slab_reduce_kernel(const float *slab, int n_workers, int n, int p, int p_ld,
                                                                 float *out, int ld_out) {
    constexpr int CH = 256 / OPB;
    __shared__ float part[CH][OPB];
    const int64_t total = (int64_t)n * p;
    const int lane = threadIdx.x % OPB, chunk = threadIdx.x / OPB;
    const int per = (n_workers + CH - 1) / CH;
    const int w0 = chunk * per, w1 = min(n_workers, w0 + per);
    for (int64_t base = (int64_t)blockIdx.x * OPB; base < total; base += (int64_t)gridDim.x * OPB) {
        const int64_t i = base + lane;
        float acc = 0.f;
        int r = 0, c = 0;
        if (i < total) {
            r = (int)(i / p); c = (int)(i % p);
            const float *s = slab + (size_t)r * p_ld + c;
            for (int w = w0; w < w1; w++) acc += s[(size_t)w * n * p_ld];
        }
        part[chunk][lane] = acc;
        __syncthreads();
        if (chunk == 0 && i < total) {
            float v = part[0][lane];
#pragma unroll
            for (int k = 1; k < CH; k++) v += part[k][lane];
            out[(size_t)r * ld_out + c] = v;
        }
        __syncthreads();
    }
}

static inline void launch_slab_reduce(const float *slab, int n_workers, int n, int p, int p_ld, float *out, int ld_out, hipStream_t s) {
    const int64_t total = (int64_t)n * p;
    auto blocks = [&](int opb) { int64_t b = (total + opb - 1) / opb; return (int)(b > 4096 ? 4096 : b); };
    if (total >= 64 * 512 || n_workers <= 16) slab_reduce_kernel<64><<<blocks(64), 256, 0, s>>>(slab, n_workers, n, p, p_ld, out, ld_out);
    else if (total >= 16 * 512 || n_workers <= 64) slab_reduce_kernel<16><<<blocks(16), 256, 0, s>>>(slab, n_workers, n, p, p_ld, out, ld_out);
    else slab_reduce_kernel<8><<<blocks(8), 256, 0, s>>>(slab, n_workers, n, p, p_ld, out, ld_out);
}

static inline int ensure_slab(gcnhip_ctx *c, size_t bytes) {
    if (c->slab_bytes >= bytes) return 0;
    GCNHIP_TRY(hipStreamSynchronize(c->stream));
    if (c->slab) GCNHIP_TRY(hipFree(c->slab));
    c->slab = nullptr; c->slab_bytes = 0;
    GCNHIP_TRY(hipMalloc((void **)&c->slab, bytes));
    c->slab_bytes = bytes;
    return 0;
}

// S[n x p] = A^T . Bm   (A: m x n, Bm: m x p), optional dropout on A
static int launch_atb(gcnhip_ctx *c, const float *A, int lda, const float *Bm, int ldb, float *out, int ld_out,
                      int m, int n, int p, int drop, float p_drop, uint64_t seed, const uint32_t *d_epoch,
                      uint64_t off, const uint8_t *keep_mask, const uint32_t *keep_bits = nullptr) {
    AtbArgs a;
    a.bits = drop ? keep_bits : nullptr;
    a.A = A; a.lda = lda; a.Bm = Bm; a.ldb = ldb;
    a.m = m; a.n = n; a.p = p; a.p_ld = (p + 3) / 4 * 4;
    a.drop = drop; a.thr = dropout_threshold(p_drop); a.scale = drop ? 1 / (1 - p_drop) : 1.f;
    a.seed = seed; a.off = off; a.d_epoch = d_epoch; a.keep_mask = keep_mask;
    const int VA = (lda % 4 == 0 && aligned16(A)) ? 4 : ((lda % 2 == 0 && ((uintptr_t)A & 7) == 0) ? 2 : 1);
    const int VB = (ldb % 4 == 0 && aligned16(Bm)) ? 4 : ((ldb % 2 == 0 && ((uintptr_t)Bm & 7) == 0) ? 2 : 1);
    const int gy = ceil_div(n, 16 * VA), gz = ceil_div(p, 16 * VB);
    // enough split-K workers to fill the chip (~8 waves per CU), at least 64 rows each
    // split-K workers: ~4 waves per CU when the partial slabs are large, up to 16 when they are small
    // (narrow outputs are latency-bound: more waves in flight hide the dependent row loads)
    const size_t slab_per_worker = (size_t)n * a.p_ld * sizeof(float);
    int per_cu = slab_per_worker <= (256u << 10) ? 16 : 4;
    int workers = ceil_div((int64_t)c->n_cu * per_cu, (int64_t)gy * gz);
    const size_t cap_mb = c->opt.atb_cap_mb > 0 ? (size_t)c->opt.atb_cap_mb : 12;                               // experiments
    while (workers > 4 && (size_t)(workers / 4) * slab_per_worker > (cap_mb << 20)) workers = workers * 3 / 4;   // <= 12 MB of partials (one slab per workgroup of 4 workers)
    // at least 8 K-steps (32 rows) per worker.  (64 until round 2: on a Cora-sized product that left 12 waves with
    // 57 dependent K-steps each — 41 us of a 126 us epoch; the workgroup-level sum made more workers free.)
    if (workers > ceil_div(m, 32)) workers = ceil_div(m, 32);
    if (workers < 1) workers = 1;
    workers = (workers + 3) / 4 * 4;
    a.rows_per_worker = (ceil_div(m, workers) + 3) / 4 * 4;
    a.n_workers = ceil_div(m, a.rows_per_worker);
    const int n_slabs = ceil_div(a.n_workers, 4);
    const int rc = ensure_slab(c, (size_t)n_slabs * n * a.p_ld * sizeof(float));
    if (rc) return rc;
    a.slab = c->slab;
    dim3 grid(n_slabs, gy, gz);
#define ATB(VA_, VB_) gemm_atb_kernel<VA_, VB_><<<grid, 256, 0, c->stream>>>(a)
    if (VA == 4 && VB == 4) ATB(4, 4);
    else if (VA == 4 && VB == 2) ATB(4, 2);
    else if (VA == 4 && VB == 1) ATB(4, 1);
    else if (VA == 2 && VB == 4) ATB(2, 4);
    else if (VA == 2 && VB == 2) ATB(2, 2);
    else if (VA == 2 && VB == 1) ATB(2, 1);
    else if (VA == 1 && VB == 4) ATB(1, 4);
    else if (VA == 1 && VB == 2) ATB(1, 2);
    else ATB(1, 1);
#undef ATB
    GCNHIP_LAUNCH_CHECK();
    launch_slab_reduce(a.slab, n_slabs, n, p, a.p_ld, out, ld_out, c->stream);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

// C[m x Nc] = A[m x K] . Bs  (Bs from B, optionally transposed), optional epilogue
static int launch_rowstream(gcnhip_ctx *c, const float *A, int lda, const float *B, int ldb, int transB,
                            float *C, int ldc, int m, int K, int Nc, const float *H, int ldh, float scale,
                            const uint32_t *hbits = nullptr, int wpr = 0, uint32_t *pack_slots = nullptr, int pack_halves = 0,
                            const float *rowscale = nullptr) {
    RowStreamArgs a;
    a.rowscale = rowscale;
    if (rowscale && pack_slots) return -1;                       // the packed-row experiment has no row factor
    a.hbits = hbits; a.wpr = wpr;
    a.pack_slots = pack_slots; a.pack_halves = pack_halves;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.transB = transB; a.C = C; a.ldc = ldc;
    a.m = m; a.K = K; a.Nc = Nc; a.H = H; a.ldh = ldh; a.scale = scale;
    const bool vec = lda % 4 == 0 && aligned16(A);
    // 16-byte stores of whole column quads (the ragged tail of Nc % 4 != 0 goes out as single floats; padding columns are
    // never written).  The packed-row experiment keeps the LDS-staged epilogue, which wants whole 64-column halves.
    a.vec_out = ((pack_slots ? Nc >= 64 : true) && ldc % 4 == 0 && aligned16(C) && (!H || hbits || (ldh % 4 == 0 && aligned16(H)))) ? 1 : 0;
    if (!vec && !pack_slots) a.vec_out = a.vec_out && Nc >= 64;    // the unaligned-A kernels: staged epilogue as before
    if (pack_slots && !(a.vec_out && vec && (H || hbits) && Nc % 64 == 0 && pack_halves * 64 == Nc)) return -1;
    const int nt_total = ceil_div(Nc, 16);
    const int NT = nt_total >= 8 ? 8 : (nt_total > 4 ? 8 : (nt_total > 3 ? 4 : nt_total));
    const int gy = ceil_div(nt_total, NT);
    const int Kp = (K + 15) / 16 * 16;
    const bool staged = a.vec_out && (pack_slots || !vec);     // only those epilogues go through LDS
    const size_t lds = ((size_t)Kp * (NT * 16 + 4) + (staged ? 4 * 16 * (NT * 16 + 4) : 0)) * sizeof(float);   // Bs (+ 4 waves' C staging)
    if (lds > 156 * 1024) return -1;          // K too long for an LDS-resident operand (gfx950: 160 KiB per CU)
    int gx = ceil_div(ceil_div(m, 16), 4);
    int per_cu = lds > 76 * 1024 ? 1 : (lds > 32 * 1024 ? 2 : 4);
    if (c->opt.rs_wgs > 0) per_cu = c->opt.rs_wgs;                                            // experiment: persistent workgroups per CU
    const int cap = c->n_cu * per_cu;
    if (gx > cap) gx = cap;
    dim3 grid(gx, gy);
#define RS2(NT_, V_, F_, K_)                                                                              \
    do {                                                                                                  \
        auto kern = (V_ && F_ && hbits) ? gemm_rowstream_kernel<NT_, V_, F_, K_, false, (V_ && F_)>       \
                                        : gemm_rowstream_kernel<NT_, V_, F_, K_>;                         \
        if (lds > 64 * 1024)                                                                              \
            GCNHIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<grid, 256, lds, c->stream>>>(a);                                                           \
    } while (0)
#define RS1(NT_, V_, F_)                                                                                  \
    do {                                                                                                  \
        switch (Kp <= 128 ? Kp / 16 : 0) {                                                                \
            case 1: RS2(NT_, V_, F_, 1); break;                                                           \
            case 2: RS2(NT_, V_, F_, 2); break;                                                           \
            case 3: RS2(NT_, V_, F_, 3); break;                                                           \
            case 4: RS2(NT_, V_, F_, 4); break;                                                           \
            case 8: if (V_ && lda >= 128) RS2(NT_, V_, F_, 8); else RS2(NT_, V_, F_, 0); break;           \
            default: RS2(NT_, V_, F_, 0); break;                                                          \
        }                                                                                                 \
    } while (0)
#define RS(NT_)                                                                                           \
    do {                                                                                                  \
        if (H || hbits) {                                                                                 \
            if (vec) RS1(NT_, true, true); else RS1(NT_, false, true);                                    \
        } else {                                                                                          \
            if (vec) RS1(NT_, true, false); else RS1(NT_, false, false);                                  \
        }                                                                                                 \
    } while (0)
#define RSP2(NT_, K_)                                                                                     \
    do {                                                                                                  \
        auto kern = gemm_rowstream_kernel<NT_, true, true, K_, true>;                                     \
        if (lds > 64 * 1024)                                                                              \
            GCNHIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<grid, 256, lds, c->stream>>>(a);                                                           \
    } while (0)
#define RSP(NT_)                                                                                          \
    do {                                                                                                  \
        switch (Kp <= 128 ? Kp / 16 : 0) {                                                                \
            case 1: RSP2(NT_, 1); break;                                                                  \
            case 2: RSP2(NT_, 2); break;                                                                  \
            case 3: RSP2(NT_, 3); break;                                                                  \
            case 4: RSP2(NT_, 4); break;                                                                  \
            default: RSP2(NT_, 0); break;                                                                 \
        }                                                                                                 \
    } while (0)
#ifdef GCNHIP_EXPERIMENTS
    if (pack_slots) {
        if (NT == 4) RSP(4); else RSP(8);
    } else
#else
    if (pack_slots) return -1;
#endif
    switch (NT) {
        case 1: RS(1); break;
        case 2: RS(2); break;
        case 3: RS(3); break;
        case 4: RS(4); break;
        default: RS(8); break;
    }
#undef RSP
#undef RSP2
#undef RS
#undef RS1
#undef RS2
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

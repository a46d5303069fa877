// dense_tile128.h — the two compute-bound GEMMs of a dense-X first layer
// (Reddit: X is 232 965 x 602, stored as CSR with every column present):
//     forward   H0[m x p]  = X~[m x K] . W[K x p]           (35.9 GFLOP at p = 128)
//     backward  dW[K x p]  = X~^T[K x m] . dH0[m x p]        (same FLOPs, m is the reduction)
// Exact-f32 matrix cores: v_mfma_f32_32x32x2_f32 (A[i=lane&31][k=lane>>5], B[k][j=lane&31],
// D[row=(reg&3)+8*(reg>>2)+4*(lane>>5)][col=lane&31]; cdna guide §3) — one LDS read per operand
// per 4096 FLOP, half of the 16x16x4 form.  Workgroup tile 128 x 128, 4 waves as 2 x 2, each
// wave 64 x 64 = 2 x 2 MFMA tiles; K chunks of 32 staged through LDS with the next chunk
// prefetched into registers while the current one is multiplied.
// The input dropout is applied while staging X, from a bit mask built once per call by
// dropbits_kernel (one Philox block per 32 decisions and bit plane; one plane at p = 0.5).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- keep bits: bit (e & 31) of word (e >> 5) = keep decision of stored element e ----------
__global__ __launch_bounds__(256) void dropbits_kernel(uint32_t *__restrict__ bits, int64_t n_elems, int thr,
                                                       uint64_t seed, const uint32_t *d_epoch, uint64_t off,
                                                       const uint8_t *__restrict__ keep_mask) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e0 = w * 32;
    if (e0 >= n_elems) return;
    uint32_t word = 0;
    if (keep_mask) {
        for (int b = 0; b < 32 && e0 + b < n_elems; b++) word |= (keep_mask[e0 + b] ? 1u : 0u) << b;
    } else {
        const uint32_t epoch = d_epoch ? *d_epoch : 0u;
        const uint64_t j0 = off + (uint64_t)e0;
        const uint32_t sh = (uint32_t)(j0 & 31);
        word = keep_word(j0 >> 7, (int)(j0 >> 5) & 3, epoch, seed, thr) >> sh;
        if (sh) {                                           // the rank's offset is not a multiple of 32
            const uint64_t j1 = j0 + 32;
            word |= keep_word(j1 >> 7, (int)(j1 >> 5) & 3, epoch, seed, thr) << (32 - sh);
        }
    }
    bits[w] = word;
}

// The same words when the stream offset is a multiple of 128 (one GPU: 0): a thread owns a whole Philox block = four words,
// so every Philox call is used in full (dropbits_kernel draws one block per WORD and keeps a quarter of it: 13.8 us per
// epoch at Reddit size, this form 5) and the words leave as one 16-byte store.  `bits` is 16-byte aligned (hipMalloc) and
// has slack past the last word (ctx.hip).
__device__ inline void dropbits_block_body(int64_t q, uint32_t *__restrict__ bits, int64_t n_elems, int thr,
                                           uint64_t seed, const uint32_t *d_epoch, uint64_t block0) {
    if (q * 128 >= n_elems) return;                                           // Philox block q = elements 128q .. 128q+127
    const uint32_t epoch = d_epoch ? *d_epoch : 0u;
    const uint64_t c = block0 + (uint64_t)q;
    uint32_t ge[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (thr != 0) {
        const int n_planes = 16 - (__ffs(thr) - 1);
        for (int i = n_planes; i >= 1; i--) {                                 // keep_word's recurrence, on the four groups at once
            uint32_t r[4];
            philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), epoch, (uint32_t)i, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            const bool and_plane = (thr >> (16 - i)) & 1;
#pragma unroll
            for (int g = 0; g < 4; g++) ge[g] = and_plane ? (r[g] & ge[g]) : (r[g] | ge[g]);
        }
    }
    *reinterpret_cast<uint4 *>(bits + q * 4) = make_uint4(ge[0], ge[1], ge[2], ge[3]);
}
__global__ __launch_bounds__(256) void dropbits_block_kernel(uint32_t *__restrict__ bits, int64_t n_elems, int thr,
                                                             uint64_t seed, const uint32_t *d_epoch, uint64_t block0) {
    dropbits_block_body((int64_t)blockIdx.x * blockDim.x + threadIdx.x, bits, n_elems, thr, seed, d_epoch, block0);
}

struct Tile128Args {
    const float *x; int ldx;          // X: m x K, row stride ldx
    const float *w; int ldw;          // fwd: W [K x p];  bwd: dH0 [m x p]
    float *out; int ldo;              // fwd: H0 [m x p]; bwd: slab [S][K][p_ld]
    int m, K, p;
    const uint32_t *bits;             // NULL: no dropout
    float scale;
    int rows_per_split;               // bwd only
    int split0;                       // bwd only: blockIdx.x == 0 is split number split0 (a launch may cover a range of splits)
    int relu;                         // fwd only: store max(x, 0)
};

template <int V>
__device__ inline void load_vec_t(const float *p, int valid, float out[V]) {
    if (valid >= V) {
        if (V == 4) { const float4 v = *reinterpret_cast<const float4 *>(p); out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
        else if (V == 2) { const float2 v = *reinterpret_cast<const float2 *>(p); out[0] = v.x; out[1] = v.y; }
        else out[0] = *p;
    } else {
#pragma unroll
        for (int s = 0; s < V; s++) out[s] = s < valid ? p[s] : 0.f;
    }
}

// keep bits of V consecutive stored elements starting at e
template <int V>
__device__ inline uint32_t bits_at(const uint32_t *bits, uint64_t e) {
    const uint32_t sh = (uint32_t)(e & 31);
    uint32_t v = bits[e >> 5] >> sh;
    if (V > 1 && sh + V > 32) v |= bits[(e >> 5) + 1] << (32 - sh);
    return v;
}

#define MFMA32(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)

// --------------------------------------------------------------------- forward
constexpr int T_BM = 128, T_BN = 128, T_BK = 32;
// k order of the FORWARD kernels (round 3): inside an 8-wide k group the two k's of MFMA step t are 8j + t and 8j + 4 + t
// (lane half kq supplies 8j + 4*kq + t) — the order dense_persist.h's forward has from its 16-byte LDS reads — so the tile
// kernels and the persistent kernel add the same products in the same order: the validation lane (tile kernel beside the
// aggregation) and the one-stream epoch (persistent kernel) give the same bits again.  T_KO(kk) = offset of step kk / 2.
#define T_KO(kk_) ((((kk_) >> 3) << 3) + (((kk_) >> 1) & 3))
constexpr int T_ALD = T_BK + 1;       // A tile [row][k]: lanes of a half-wave walk rows -> odd stride, conflict-free

// FAST: the launch site guarantees 16-byte aligned rows of X whose stride covers round_up(K, 32) columns
// (zero padded), ldw % 4 == 0 and full 128-column output tiles.  Then every load of the K loop is
// unconditional — rows past m are clamped to the last row and dropped at the store, W rows past K are read
// clamped and zeroed by a select — so the compiler issues the whole prefetch as one clause instead of a
// chain of divergent branches with a wait in each.
template <int VX, bool FAST = false>
__global__ __launch_bounds__(256) void dense_fwd_t128_kernel(Tile128Args a) {
    __shared__ float As[T_BM * T_ALD];
    __shared__ __attribute__((aligned(16))) float Bs[T_BK * T_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kq = lane >> 5;
    const int row_base = blockIdx.x * T_BM, col_base = blockIdx.y * T_BN;
    constexpr int LPR = T_BK / VX;                        // lanes per A row
    constexpr int A_PIECES = T_BM * T_BK / (256 * VX);
    float areg[A_PIECES][VX];
    uint64_t kreg[A_PIECES];                              // keep bits, applied in stash(): the multiply must not
    float4 breg[4];                                       // wait for the loads before the MFMA loop starts

    auto fetch = [&](int k0) {
        if constexpr (FAST) {
            static_assert(!FAST || VX == 4, "fast path stages X with 16-byte loads");
#pragma unroll
            for (int pc = 0; pc < A_PIECES; pc++) {
                const int idx = pc * 256 + tid;
                const int r = idx / LPR, c = (idx % LPR) * VX;
                const int row = min(row_base + r, a.m - 1), col = k0 + c;
                const float4 v = *reinterpret_cast<const float4 *>(a.x + (size_t)row * a.ldx + col);
                areg[pc][0] = v.x; areg[pc][1] = v.y; areg[pc][2] = v.z; areg[pc][3] = v.w;
                if (a.bits) {
                    const uint64_t w = ((uint64_t)row * a.K + col) >> 5;
                    kreg[pc] = (uint64_t)a.bits[w] | ((uint64_t)a.bits[w + 1] << 32);
                }
            }
#pragma unroll
            for (int pc = 0; pc < 4; pc++) {
                const int idx = (pc * 256 + tid) * 4;
                const int k = idx / T_BN, c = idx % T_BN;
                const int gk = k0 + k;
                breg[pc] = *reinterpret_cast<const float4 *>(a.w + (size_t)min(gk, a.K - 1) * a.ldw + col_base + c);   // zeroed in stash()
            }
            return;
        }
#pragma unroll
        for (int pc = 0; pc < A_PIECES; pc++) {
            const int idx = pc * 256 + tid;
            const int r = idx / LPR, c = (idx % LPR) * VX;
            const int row = row_base + r, col = k0 + c;
#pragma unroll
            for (int s = 0; s < VX; s++) areg[pc][s] = 0.f;
            if (row < a.m && col < a.K) {
                load_vec_t<VX>(a.x + (size_t)row * a.ldx + col, a.K - col, areg[pc]);
                if (a.bits) {                                   // raw words: no ALU on them here
                    const uint64_t w = ((uint64_t)row * a.K + col) >> 5;
                    kreg[pc] = a.bits[w];
                    // rows padded to 16 bytes but K % VX != 0: the VX bits may straddle two words
                    if (a.K % VX != 0) kreg[pc] |= (uint64_t)a.bits[w + 1] << 32;
                }
            }
        }
#pragma unroll
        for (int pc = 0; pc < 4; pc++) {
            const int idx = (pc * 256 + tid) * 4;
            const int k = idx / T_BN, c = idx % T_BN;
            const int gk = k0 + k, gc = col_base + c;
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            if (gk < a.K) {
                if ((a.ldw & 3) == 0 && gc + 4 <= a.p) {
                    const float4 v = *reinterpret_cast<const float4 *>(a.w + (size_t)gk * a.ldw + gc);
                    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; s++) if (gc + s < a.p) t[s] = a.w[(size_t)gk * a.ldw + gc + s];
                }
            }
            breg[pc] = make_float4(t[0], t[1], t[2], t[3]);
        }
    };
    auto stash = [&](int cur_k0) {
#pragma unroll
        for (int pc = 0; pc < A_PIECES; pc++) {
            const int idx = pc * 256 + tid;
            const int r = idx / LPR, c = (idx % LPR) * VX;
            const int brow = FAST ? min(row_base + r, a.m - 1) : row_base + r;
            const uint32_t kb = a.bits ? (uint32_t)(kreg[pc] >> (uint32_t)(((uint64_t)brow * a.K + cur_k0 + c) & 31)) : 0xFu;
#pragma unroll
            for (int s = 0; s < VX; s++)
                As[r * T_ALD + c + s] = a.bits ? ((kb >> s & 1u) ? areg[pc][s] * a.scale : 0.f) : areg[pc][s];
        }
#pragma unroll
        for (int pc = 0; pc < 4; pc++) {
            const int idx = (pc * 256 + tid) * 4;
            float4 v = breg[pc];
            // any ALU on a prefetched register belongs HERE, after the MFMA loop the loads were hidden behind
            if (FAST && cur_k0 + idx / T_BN >= a.K) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(&Bs[idx]) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += T_BK) {
        __syncthreads();
        stash(k0);
        __syncthreads();
        if (k0 + T_BK < a.K) fetch(k0 + T_BK);      // FAST: k0 + 32 <= round_up(K, 32) <= ldx
        const float *Ap = &As[(wm * 64 + li) * T_ALD + 4 * kq];
        const float *Bp = &Bs[4 * kq * T_BN + wn * 64 + li];
#pragma unroll
        for (int kk = 0; kk < T_BK; kk += 2) {
            const int ko = T_KO(kk);                             // the k pair of this step: (8j + t, 8j + 4 + t), see T_KO
            const float a0 = Ap[ko], a1 = Ap[32 * T_ALD + ko];
            const float b0 = Bp[ko * T_BN], b1 = Bp[ko * T_BN + 32];
            acc[0][0] = MFMA32(a0, b0, acc[0][0]);
            acc[0][1] = MFMA32(a0, b1, acc[0][1]);
            acc[1][0] = MFMA32(a1, b0, acc[1][0]);
            acc[1][1] = MFMA32(a1, b1, acc[1][1]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = col_base + wn * 64 + j * 32 + li;
            if (col >= a.p) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = row_base + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                if (row < a.m) a.out[(size_t)row * a.ldo + col] = (a.relu && !(acc[i][j][r] > 0.f)) ? 0.f : acc[i][j][r];
            }
        }
}

// Eight-wave form of the FAST forward tile (same 128 x 128 x 32 staging, same LDS images, same MFMA and the same
// k order per output element => identical bits): 512 threads as 4 x 2 waves of 32 x 64, so a wave keeps 32
// accumulator registers instead of 64 and a SIMD holds twice the waves to feed its MFMA pipe across the two
// barriers of a K chunk.  Launch sites use it when every FAST condition holds.
__global__ __launch_bounds__(512) void dense_fwd_t128w8_kernel(Tile128Args a) {
    __shared__ float As[T_BM * T_ALD];
    __shared__ __attribute__((aligned(16))) float Bs[T_BK * T_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                // rows 32*wm .. +31, cols 64*wn .. +63
    const int li = lane & 31, kq = lane >> 5;
    const int row_base = blockIdx.x * T_BM, col_base = blockIdx.y * T_BN;
    constexpr int LPR = T_BK / 4;                           // lanes per A row (16 bytes each)
    constexpr int A_PIECES = T_BM * T_BK / (512 * 4);       // 2
    float areg[A_PIECES][4];
    uint64_t kreg[A_PIECES];
    float4 breg[2];

    auto fetch = [&](int k0) {
#pragma unroll
        for (int pc = 0; pc < A_PIECES; pc++) {
            const int idx = pc * 512 + tid;
            const int r = idx / LPR, c = (idx % LPR) * 4;
            const int row = min(row_base + r, a.m - 1), col = k0 + c;
            const float4 v = *reinterpret_cast<const float4 *>(a.x + (size_t)row * a.ldx + col);
            areg[pc][0] = v.x; areg[pc][1] = v.y; areg[pc][2] = v.z; areg[pc][3] = v.w;
            if (a.bits) {
                const uint64_t w = ((uint64_t)row * a.K + col) >> 5;
                kreg[pc] = (uint64_t)a.bits[w] | ((uint64_t)a.bits[w + 1] << 32);
            }
        }
#pragma unroll
        for (int pc = 0; pc < 2; pc++) {
            const int idx = (pc * 512 + tid) * 4;
            const int k = idx / T_BN, c = idx % T_BN;
            breg[pc] = *reinterpret_cast<const float4 *>(a.w + (size_t)min(k0 + k, a.K - 1) * a.ldw + col_base + c);   // zeroed in stash()
        }
    };
    auto stash = [&](int cur_k0) {
#pragma unroll
        for (int pc = 0; pc < A_PIECES; pc++) {
            const int idx = pc * 512 + tid;
            const int r = idx / LPR, c = (idx % LPR) * 4;
            const int brow = min(row_base + r, a.m - 1);
            const uint32_t kb = a.bits ? (uint32_t)(kreg[pc] >> (uint32_t)(((uint64_t)brow * a.K + cur_k0 + c) & 31)) : 0xFu;
#pragma unroll
            for (int s = 0; s < 4; s++)
                As[r * T_ALD + c + s] = a.bits ? ((kb >> s & 1u) ? areg[pc][s] * a.scale : 0.f) : areg[pc][s];
        }
#pragma unroll
        for (int pc = 0; pc < 2; pc++) {
            const int idx = (pc * 512 + tid) * 4;
            float4 v = breg[pc];
            if (cur_k0 + idx / T_BN >= a.K) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(&Bs[idx]) = v;
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += T_BK) {
        __syncthreads();
        stash(k0);
        __syncthreads();
        if (k0 + T_BK < a.K) fetch(k0 + T_BK);
        const float *Ap = &As[(wm * 32 + li) * T_ALD + 4 * kq];
        const float *Bp = &Bs[4 * kq * T_BN + wn * 64 + li];
        // operands of step kk+2 are read from LDS before the MFMAs of step kk are issued (the scheduler otherwise puts every
        // read right in front of its use and the wave waits out the LDS latency once per MFMA pair)
        float a0 = Ap[0], b0 = Bp[0], b1 = Bp[32];
#pragma unroll
        for (int kk = 0; kk < T_BK; kk += 2) {
            float an = 0.f, bn0 = 0.f, bn1 = 0.f;
            if (kk + 2 < T_BK) { const int kn = T_KO(kk + 2); an = Ap[kn]; bn0 = Bp[kn * T_BN]; bn1 = Bp[kn * T_BN + 32]; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = MFMA32(a0, b0, acc[0]);
            acc[1] = MFMA32(a0, b1, acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            a0 = an; b0 = bn0; b1 = bn1;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int col = col_base + wn * 64 + j * 32 + li;
        if (col >= a.p) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = row_base + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
            if (row < a.m) a.out[(size_t)row * a.ldo + col] = (a.relu && !(acc[j][r] > 0.f)) ? 0.f : acc[j][r];
        }
    }
}

// -------------------------------------------------------------------- backward
// grid (S splits of the row range, K tiles of 128 X-columns, p tiles of 128).
// LDS tiles are k-major exactly as loaded: As[k][xcol], Bs[k][pcol]; both MFMA
// operands read 32 consecutive floats per half-wave (conflict-free).
// FAST: 16-byte aligned rows of X whose stride covers round_up(K, 128) columns (zero padded), ldw % 4 == 0,
// full 128-column tiles of dH0.  Rows past the split's end are read clamped and zeroed by selects.
template <int VX, bool FAST = false>
__global__ __launch_bounds__(256) void dense_bwd_t128_kernel(Tile128Args a) {
    __shared__ __attribute__((aligned(16))) float As[T_BK * 128];
    __shared__ __attribute__((aligned(16))) float Bs[T_BK * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kq = lane >> 5;
    const int xc_base = blockIdx.y * 128, pc_base = blockIdx.z * 128;
    const int split = blockIdx.x + a.split0;
    const int r_begin = split * a.rows_per_split;
    const int r_end = min(a.m, r_begin + a.rows_per_split);
    constexpr int LPR = 128 / VX;
    constexpr int A_PIECES = T_BK * 128 / (256 * VX);
    float areg[A_PIECES][VX];
    uint64_t kreg[A_PIECES];
    float4 breg[4];

    auto fetch = [&](int r0) {
        if constexpr (FAST) {
            static_assert(!FAST || VX == 4, "fast path stages X with 16-byte loads");
#pragma unroll
            for (int pc = 0; pc < A_PIECES; pc++) {
                const int idx = pc * 256 + tid;
                const int k = idx / LPR, c = (idx % LPR) * VX;
                const int row = min(r0 + k, r_end - 1), col = xc_base + c;
                const float4 v = *reinterpret_cast<const float4 *>(a.x + (size_t)row * a.ldx + col);   // rows past r_end: zeroed in stash()
                areg[pc][0] = v.x; areg[pc][1] = v.y; areg[pc][2] = v.z; areg[pc][3] = v.w;
                if (a.bits) {
                    const uint64_t w = ((uint64_t)row * a.K + col) >> 5;
                    kreg[pc] = (uint64_t)a.bits[w] | ((uint64_t)a.bits[w + 1] << 32);
                }
            }
#pragma unroll
            for (int pc = 0; pc < 4; pc++) {
                const int idx = (pc * 256 + tid) * 4;
                const int k = idx / 128, c = idx % 128;
                breg[pc] = *reinterpret_cast<const float4 *>(a.w + (size_t)min(r0 + k, r_end - 1) * a.ldw + pc_base + c);
            }
            return;
        }
#pragma unroll
        for (int pc = 0; pc < A_PIECES; pc++) {
            const int idx = pc * 256 + tid;
            const int k = idx / LPR, c = (idx % LPR) * VX;
            const int row = r0 + k, col = xc_base + c;
#pragma unroll
            for (int s = 0; s < VX; s++) areg[pc][s] = 0.f;
            if (row < r_end && col < a.K) {
                load_vec_t<VX>(a.x + (size_t)row * a.ldx + col, a.K - col, areg[pc]);
                if (a.bits) {                                   // raw words: no ALU on them here
                    const uint64_t w = ((uint64_t)row * a.K + col) >> 5;
                    kreg[pc] = a.bits[w];
                    // rows padded to 16 bytes but K % VX != 0: the VX bits may straddle two words
                    if (a.K % VX != 0) kreg[pc] |= (uint64_t)a.bits[w + 1] << 32;
                }
            }
        }
#pragma unroll
        for (int pc = 0; pc < 4; pc++) {
            const int idx = (pc * 256 + tid) * 4;
            const int k = idx / 128, c = idx % 128;
            const int row = r0 + k, gc = pc_base + c;
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            if (row < r_end) {
                if ((a.ldw & 3) == 0 && gc + 4 <= a.p) {
                    const float4 v = *reinterpret_cast<const float4 *>(a.w + (size_t)row * a.ldw + gc);
                    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; s++) if (gc + s < a.p) t[s] = a.w[(size_t)row * a.ldw + gc + s];
                }
            }
            breg[pc] = make_float4(t[0], t[1], t[2], t[3]);
        }
    };
    auto stash = [&](int cur_r0) {
#pragma unroll
        for (int pc = 0; pc < A_PIECES; pc++) {
            const int idx = pc * 256 + tid;
            const int k = idx / LPR, c = (idx % LPR) * VX;
            const int brow = FAST ? min(cur_r0 + k, r_end - 1) : cur_r0 + k;
            const uint32_t kb = a.bits ? (uint32_t)(kreg[pc] >> (uint32_t)(((uint64_t)brow * a.K + xc_base + c) & 31)) : 0xFu;
#pragma unroll
            for (int s = 0; s < VX; s++)
                As[k * 128 + c + s] = a.bits ? ((kb >> s & 1u) ? areg[pc][s] * a.scale : 0.f) : areg[pc][s];
        }
#pragma unroll
        for (int pc = 0; pc < 4; pc++) {
            const int idx = (pc * 256 + tid) * 4;
            float4 v = breg[pc];
            if (FAST && cur_r0 + idx / 128 >= r_end) v = make_float4(0.f, 0.f, 0.f, 0.f);   // dH0 rows past the split: A may then hold anything finite
            *reinterpret_cast<float4 *>(&Bs[idx]) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    if (r_begin < r_end) fetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += T_BK) {
        __syncthreads();
        stash(r0);
        __syncthreads();
        if (r0 + T_BK < r_end) fetch(r0 + T_BK);
        const float *Ap = &As[kq * 128 + wm * 64 + li];
        const float *Bp = &Bs[kq * 128 + wn * 64 + li];
        // operands of step kk+2 are read from LDS before the MFMAs of step kk are issued (as in the forward tile)
        float a0 = Ap[0], a1 = Ap[32], b0 = Bp[0], b1 = Bp[32];
#pragma unroll
        for (int kk = 0; kk < T_BK; kk += 2) {
            float an0 = 0.f, an1 = 0.f, bn0 = 0.f, bn1 = 0.f;
            if (kk + 2 < T_BK) { an0 = Ap[(kk + 2) * 128]; an1 = Ap[(kk + 2) * 128 + 32]; bn0 = Bp[(kk + 2) * 128]; bn1 = Bp[(kk + 2) * 128 + 32]; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = MFMA32(a0, b0, acc[0][0]);
            acc[0][1] = MFMA32(a0, b1, acc[0][1]);
            acc[1][0] = MFMA32(a1, b0, acc[1][0]);
            acc[1][1] = MFMA32(a1, b1, acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            a0 = an0; a1 = an1; b0 = bn0; b1 = bn1;
        }
    }
    // partial [K x p] of this split
    float *slab = a.out + (size_t)split * a.K * a.ldo;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int pc = pc_base + wn * 64 + j * 32 + li;
            if (pc >= a.p) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int xc = xc_base + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                if (xc < a.K) slab[(size_t)xc * a.ldo + pc] = acc[i][j][r];
            }
        }
}


// dense_bf16x3.h — the dense first-layer products on the bf16 matrix pipe with f32 results (round 5):
//     forward   H0[m x 128]  = X~[m x K] . W[K x 128]            (SparseMatmul::forward on a dense X, module.cpp:47-61)
//     backward  dW[K x 128]  = X~^T[K x m] . dH0[m x 128]         (SparseMatmul::backward, module.cpp:63-77)
// Why: the exact-f32 MFMA (dense_persist.h, dense_tile128.h) runs at 1/16 of the bf16 MFMA rate on gfx950 and these three
// 35.9-GFLOP products were 31 % of the epoch at 0.65-0.73 of that pipe's peak.
// How: every f32 operand is split EXACTLY into three bf16 planes, x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi),
// lo = x - hi - mid (round-to-nearest planes: |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|, and lo has at most 8 significant bits, so
// it is a bf16 number: nothing is lost in the split).  A product a.b is the sum of nine plane products; the six with
// weight >= 2^-16 (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi) are issued as v_mfma_f32_32x32x16_bf16 into the SAME f32
// accumulator (NP = 6); NP = 8 adds mid.lo and lo.mid (weight 2^-24).  bf16 x bf16 is exact in f32, so the only errors
// are the dropped plane products (<= 2^-23 |a.b| for NP = 6, <= 2^-32 for NP = 8) and the accumulator's roundings — fewer
// of them than the exact-f32 chain's one per k (the pipe adds 16 products per instruction before it rounds).  The error
// against a float64 product is measured beside the exact-f32 kernels' in tools/gemm_bf16x3.hip and tests/test_ops_gpu.py;
// both sit inside the summation-order bound the parity tests use (8 eps_f32 sum|terms|).
// X (and A^.X) stay f32 in HBM and are split in registers on their way from LDS to the MFMA operands (the same 561 MB per
// product as the f32 kernels; pre-split planes would be 842 MB); W1 / dH0 are split by small pre-passes.
// The f32-MFMA kernels stay selectable (option gemm_bf16x3 = 0).
#pragma once
#include "dense_persist.h"
#include "bf16x3_split.h"

constexpr int BX_BK = 32;                                    // K per chunk of X (two MFMA k-steps of 16 = two half-items)
constexpr int BX_BH_BYTES = 3 * 4 * 1024;                    // 12288: one k-step of W: [plane][column block][lane] x 16 bytes
constexpr int BX_NB = 10;                                    // ring of W k-steps in LDS
constexpr int BX_PD = 6;                                     // ... issued this many half-items ahead of their use
constexpr int BX_SMEM = BX_NB * BX_BH_BYTES;                 // 122880

// (global_load_lds_dwordx3 is no way to move a 12 KB block in 768-byte pieces: measured on gfx950, tools/gemm_bf16x3.hip's probe,
// it writes each lane's 12 bytes at LDS base + lane * 16 and leaves the fourth dword of every 16 untouched.  The W k-step
// therefore travels as twelve 1 KB dwordx4 pieces; every wave issues two — waves 4-7 their own and, again, one of pieces 0-3 —
// so that the counted waits are the same for all.)
//
// COUNTED WAITS, and what they may assume (measured, round 5: tools/gemm_bf16x3.hip "beside a co-running kernel").  vmcnt counts
// this wave's loads, LDS-DMA pieces and stores together, but an LDS-DMA piece served by L2 COMPLETES AHEAD of an older register
// load still waiting for HBM (and stores complete ahead of loads): "at most N outstanding" says the N youngest of EACH KIND
// may be outstanding, not the N youngest overall.  A first version counted the younger operations of both kinds together; alone
// on the chip every load had landed long before its wait and all tests passed — beside a second kernel (the validation lane)
// the X loads were slow, the wait fell through on the early DMA completions, and rows of garbage came out.  So a wait for an
// operation of one kind allows only the number of YOUNGER OPERATIONS OF THE SAME KIND: operations of a kind do complete in order.

// W[K x 128] (* scale) -> the forward's B image, one 12 KB block per k-step hs: 16-byte piece ((hs*3 + p)*4 + n)*64 + lane
// holds plane p of W[k(hs, lane >> 5, j)][32 n + (lane & 31)], j = 0..7, where the MFMA's k slots are PERMUTED inside a
// 32-wide chunk so that a lane of the forward reads 16 consecutive floats of its row of X:
//     k(hs, h, j) = 32 (hs >> 1) + 16 h + 8 (hs & 1) + j          (rows past K: zero)
// One thread per (hs, n, lane).
__device__ inline void bx_pack_w_body(int idx, const float *__restrict__ w, int ldw, int K, int n_hs, uint4 *__restrict__ wp, float scale) {
    if (idx >= n_hs * 4 * 64) return;
    const int lane = idx & 63, n = (idx >> 6) & 3, hs = idx >> 8;
    const int col = 32 * n + (lane & 31), k0 = 32 * (hs >> 1) + 16 * (lane >> 5) + 8 * (hs & 1);
    BxPlanes P;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int k = k0 + 2 * i;
        const float a = k < K ? w[(size_t)k * ldw + col] * scale : 0.f, b = (k + 1) < K ? w[(size_t)(k + 1) * ldw + col] * scale : 0.f;
        bx_split2(a, b, P.w[0][i], P.w[1][i], P.w[2][i]);
    }
#pragma unroll
    for (int p = 0; p < 3; p++) wp[(((size_t)hs * 3 + p) * 4 + n) * 64 + lane] = make_uint4(P.w[p][0], P.w[p][1], P.w[p][2], P.w[p][3]);
}
// W2 [128 x p2] -> the ZOUT form's A image: piece (s, cb) of lane ln = (c, h) holds plane p of W2[f(s, h, j)][32 cb + c], j = 0..7,
// f(s, h, j) = 32 (s >> 1) + (j & 3) + 8 ((j >> 2) + 2 (s & 1)) + 4 h — the feature that accumulator register 8 (s & 1) + j of
// block s >> 1 holds in lane half h of the transposed tile; classes past p2: zero.  One thread per (s, cb, lane): 1024.
__device__ inline void bx_pack_w2_body(int idx, const float *__restrict__ w2, int ldw2, int p2, uint4 *__restrict__ img) {
    if (idx >= 8 * 2 * 64) return;
    const int ln = idx & 63, cb = (idx >> 6) & 1, s = idx >> 7;
    const int c = 32 * cb + (ln & 31), h = ln >> 5;
    BxPlanes P;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = 2 * i + u;
            const int f = 32 * (s >> 1) + (j & 3) + 8 * ((j >> 2) + 2 * (s & 1)) + 4 * h;
            v[u] = c < p2 ? w2[(size_t)f * ldw2 + c] : 0.f;
        }
        bx_split2(v[0], v[1], P.w[0][i], P.w[1][i], P.w[2][i]);
    }
#pragma unroll
    for (int p = 0; p < 3; p++) img[((s * 2 + cb) * 3 + p) * 64 + ln] = make_uint4(P.w[p][0], P.w[p][1], P.w[p][2], P.w[p][3]);
}
// (w2 == NULL: the W planes alone; else four more workgroups lay the W2 image out)
__global__ __launch_bounds__(256) void bx_pack_w_kernel(const float *__restrict__ w, int ldw, int K, int n_hs, uint4 *__restrict__ wp, float scale,
                                                        const float *__restrict__ w2 = nullptr, int ldw2 = 0, int p2 = 0, uint4 *__restrict__ img = nullptr) {
    if ((int)blockIdx.x < n_hs) bx_pack_w_body(blockIdx.x * blockDim.x + threadIdx.x, w, ldw, K, n_hs, wp, scale);
    else bx_pack_w2_body(((int)blockIdx.x - n_hs) * 256 + threadIdx.x, w2, ldw2, p2, img);
}
// ---- keep bits, CHUNK-MAJOR: word [c * m + r] = the keep decisions of elements (r, 32 c .. 32 c + 31) of X, bit i = column 32 c + i
// (bits of columns >= K: zero).  The same decisions as the flat array of dropbits_kernel (bit e & 31 of word e >> 5, e = r K + col)
// — element e is decided by bit (off + e) of the (seed, epoch) stream — laid out for the two kernels below: the forward's lanes
// (one row each) read the word of their chunk from ONE 128-byte line per 32 rows, the weight gradient's wave reads the 16 words
// of its step's rows with one load and uses them as LANE MASKS (lane = column).  With the flat array the forward's wave touched
// 19 lines per chunk for 8 bytes per lane (38 us of its 211 at Reddit scale, 26 of them the load alone) and the gradient issued
// 8 one-word loads per step.  A workgroup makes the words of R rows: the stream's Philox blocks that cover them into LDS (every
// block drawn once), then every (chunk, row) window out of LDS.  keep_mask != NULL: injected decisions (bytes, element order).
struct BxBitsArgs {
    uint32_t *bits; int m, K, n_chunks, R;      // R rows per workgroup: R * K / 32 + 12 <= BX_BITS_LDS_WORDS
    int thr; uint64_t seed; const uint32_t *d_epoch; uint64_t off; const uint8_t *keep_mask;
};
constexpr int BX_BITS_LDS_WORDS = 4096;
__device__ inline void dropbits_cm_body(int wg, int tid, const BxBitsArgs &b, uint32_t *lds) {
    const int ra = wg * b.R, rb = min(b.m, ra + b.R);
    if (ra >= rb) return;
    const uint64_t bit0 = b.off + (uint64_t)ra * b.K;                     // first stream bit of this workgroup's rows
    const uint64_t q0 = bit0 >> 7;                                        // first Philox block
    if (!b.keep_mask) {
        const uint64_t bit1 = b.off + (uint64_t)rb * b.K + 64;            // a window reads up to a word past its last bit
        const int n_blk = (int)((bit1 >> 7) - q0) + 1;
        const uint32_t epoch = b.d_epoch ? *b.d_epoch : 0u;
        for (int blk = tid; blk < n_blk; blk += 256) {
            const uint64_t c = q0 + (uint64_t)blk;
            uint32_t ge[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (b.thr != 0) {
                const int n_planes = 16 - (__ffs(b.thr) - 1);
                for (int i = n_planes; i >= 1; i--) {                     // keep_word's recurrence (common.h), on the four groups at once
                    uint32_t r[4];
                    philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), epoch, (uint32_t)i, (uint32_t)b.seed, (uint32_t)(b.seed >> 32), r);
                    const bool and_plane = (b.thr >> (16 - i)) & 1;
#pragma unroll
                    for (int g = 0; g < 4; g++) ge[g] = and_plane ? (r[g] & ge[g]) : (r[g] | ge[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) lds[4 * blk + g] = ge[g];
        }
    }
    __syncthreads();
    const int rows = rb - ra;
    const uint32_t rel0 = (uint32_t)(bit0 - (q0 << 7));                   // this workgroup's first bit inside the LDS image
    for (int idx = tid; idx < b.n_chunks * rows; idx += 256) {
        int c, r;
        if (rows == 128) { c = idx >> 7; r = idx & 127; } else { c = idx / rows; r = idx - c * rows; }
        const int row = ra + r;
        const int valid = min(32, b.K - 32 * c);                          // columns of this chunk inside the matrix
        uint32_t word;
        if (b.keep_mask) {
            word = 0;
            const uint8_t *km = b.keep_mask + (size_t)row * b.K + 32 * c;
            for (int i = 0; i < valid; i++) word |= (km[i] ? 1u : 0u) << i;
        } else {
            const uint32_t rel = rel0 + (uint32_t)r * (uint32_t)b.K + 32u * (uint32_t)c;
            const uint32_t lo = lds[rel >> 5], hi = lds[(rel >> 5) + 1], sh = rel & 31u;
            word = sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
            if (valid < 32) word &= (1u << valid) - 1u;
        }
        b.bits[(size_t)c * b.m + row] = word;
    }
}
__global__ __launch_bounds__(256) void dropbits_cm_kernel(BxBitsArgs b) {
    __shared__ uint32_t lds[BX_BITS_LDS_WORDS];
    dropbits_cm_body(blockIdx.x, threadIdx.x, b, lds);
}
// keep bits + packed W in one launch (as dropbits_pack_w_kernel of dense_persist.h)
__global__ __launch_bounds__(256) void dropbits_bx_pack_w_kernel(BxBitsArgs b, int n_bits_wgs,
                                                                 const float *__restrict__ w, int ldw, int K, int n_hs,
                                                                 uint4 *__restrict__ wp, float scale) {
    __shared__ uint32_t lds[BX_BITS_LDS_WORDS];
    if ((int)blockIdx.x < n_bits_wgs) dropbits_cm_body(blockIdx.x, threadIdx.x, b, lds);
    else bx_pack_w_body(((int)blockIdx.x - n_bits_wgs) * 256 + threadIdx.x, w, ldw, K, n_hs, wp, scale);
}

struct Bx3FwdArgs {
    const float *x; int ldx;          // X, 16-byte aligned rows, ldx >= round_up(K, 32) with zero padding
    const uint4 *wp;                  // packed planes of W (bx_pack_w_kernel): 2 * n_chunks blocks of 12 KB
    float *out; int ldo;              // H0 [m x 128]; m * ldo * 4 < 2^32 (buffer stores)
    int m, K, n_chunks, n_rb;         // n_chunks = ceil(K / 32); n_rb = ceil(m / 32)
    const uint32_t *bits;             // keep bits, chunk-major (dropbits_cm_body: word c * m + row), NULL: no dropout
    int relu;
    // ZOUT form (evaluation): the second layer's product rides in the epilogue — Z0 = relu(X.W) . W2 leaves, H never does
    const uint4 *w2p;                 // planes of W2 as the A operand of Z0^T = W2^T . H^T (bx_pack_w2_body): 16 pieces of 3 KB
    float *z0; int ldz, p2;           // Z0 [m x p2], p2 <= 64, 16-byte aligned rows
};
constexpr int BX_W2_BYTES = 8 * 2 * 3 * 1024;                // 49152

struct BxB3 { bf16x8 h, m, l; };
template <int N> struct BxN { static constexpr int value = N; };
struct BxRaw { f32x4 v[4]; uint32_t kw; };                    // one chunk of this lane's row: 16 floats (k = 16 hh + 0..15) + the chunk's keep word of the row

// Every vector-memory operation of the kernel is inline asm, so that ONE hand-kept count covers them all (vmcnt counts loads,
// LDS-DMA pieces and stores together, in issue order; hipcc neither sees these nor waits for them).
template <int OFF>
__device__ __forceinline__ void bx_gload16(f32x4 &dst, const float *p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void bx_gload4(uint32_t &dst, const uint32_t *p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// wait until at most N vector-memory operations of this wave are outstanding; the registers of `r` are tied to the statement so
// that no use of them can be scheduled ahead of it
// LDS reads are inline asm as well: hipcc sinks an LDS load it can see to its first use (across sched_barrier), which puts
// every read right in front of the MFMA that needs it — the one-load-in-flight disease.  Issued here, waited for by
// BX_WAIT_LDS one group later (lgkmcnt(0): everything this wave has asked LDS for).
template <int OFF>
__device__ __forceinline__ void bx_ldsread16(bf16x8 &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
#define BX_WAIT_LDS(b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"((b).h), "+v"((b).m), "+v"((b).l) :: "memory")
// the words pair i of a split produced are complete HERE (hipcc otherwise sinks the whole split to its first use, one clump of
// VALU in front of the next half-item's first MFMA instead of two instructions behind each MFMA of this one)
#define BX_PIN(P, i) asm volatile("" : "+v"((P).w[0][i]), "+v"((P).w[1][i]), "+v"((P).w[2][i]))
#define BX_WAIT_RAW(N, r) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"((r).v[0]), "+v"((r).v[1]), "+v"((r).v[2]), "+v"((r).v[3]), "+v"((r).kw) :: "memory")

// One 512-thread workgroup per CU for the whole launch, given a contiguous share of the 32-row blocks.  A ROUND is 8 row blocks:
// wave w owns rows 32w .. 32w+31 of the round x all 128 columns (4 accumulator blocks of 32 x 32).
//  * X never touches LDS: a lane reads 16 consecutive floats of its row per chunk straight into registers, three chunks deep
//    (raw[g] being split, raw[g+1] landed or landing, raw[g+2] just issued) — the rows are private to the wave, so LDS would
//    only buy coalescing, and the 64-byte pieces of two lanes make up the row's whole 128-byte line.
//  * The W planes of a k-step (12 KB, shared by the 8 waves) arrive by LDS-DMA BX_PD half-items ahead into a ring of BX_NB.
//  * A half-item = one MFMA k-step: 4 column blocks x NP plane products.  Groups G0..G3 (one column block each); the split
//    of the next half-item's X values is spread over the four groups (one pair of values behind each), the W planes of the next
//    group are read while this group's MFMAs run, the barrier (W k-step h+1 landed, ring slot free) sits between G1 and G2.
// The last round of a workgroup may have fewer than 8 row blocks: waves without one run the same instruction stream on the
// share's last row and store nothing (the counted waits need every wave to issue the same operations).
// ABL (timing experiments of tools/gemm_bf16x3.hip only; 1-8, 32 and 64 give wrong results): 1 no X loads, 2 no W DMA, 4 no split,
// 8 no W LDS reads, 16 every wait vmcnt(0), 32 keep words loaded but not applied, 64 applied but not loaded, 128 keep word after the
// X loads (+9 us), 256 X loads ahead of the chunk's W pieces (+40 us), 512 X loads at the top of the chunk (+9..14 us),
// 1024 no workgroup barrier (round 6: what the per-half-item barrier costs the MFMA skeleton).
// PD_: W k-steps in flight (6..9 measure the same).
// NW = waves per workgroup.  8: the whole-chip form (two waves per SIMD fill the CU's register file).  4 (round 5, for a context
// that runs BESIDE another stream's kernels — the validation lane): one wave per SIMD and half the register file, so that a
// gather-bound kernel's waves can be resident on the same CU at the same time.  The eight-wave workgroup needs an EMPTY CU to
// start: beside the training pass's hidden-width aggregation it is handed CUs only as they drain and runs 895 us instead of
// 198 (kernel timeline, docs/NOTEBOOK_r5.md §7).  Measured: co-residency costs the aggregation half its waves on those CUs and
// the epoch gains nothing (option gemm_lane_waves, default 8).
// ZOUT (round 5, evaluation forwards; verdict r04 item 4c): the product is computed TRANSPOSED (W planes as the A operand, the X
// planes as B: the same plane products in the same order), so that a lane ends up with ONE ROW and, in its accumulator
// registers, that row's features — which is the layout the B operand of a second product Z0^T = W2^T . relu(H)^T wants: register
// group 8(s & 1) .. +7 of feature block s >> 1 is k-step s, with the k slots permuted (slot (h, j) = feature (j & 3) + 8 (j >> 2) +
// 4 h of the 16) exactly as bx_pack_w2_body lays W2 out.  At the end of a round the 32 x 128 tile of H is clamped at zero, split
// into planes in registers and multiplied by the W2 image (48 KB of LDS beside a ring of 8 instead of 10 W k-steps); Z0's 32 x p2
// tile is stored, H is not: 119 MB less written, 119 MB less read, one launch less per evaluation forward.
template <bool DROP, int NP, int ABL = 0, int NW = 8, bool ZOUT = false, int PD_ = 0>
__global__ __launch_bounds__(64 * NW, 2) void dense_fwd_bf16x3_kernel(Bx3FwdArgs a) {
    constexpr int PD = PD_ ? PD_ : BX_PD;                        // W k-steps in flight ahead of their use
    static_assert(NW == 8 || NW == 4, "12 W pieces per k-step: two per wave of eight (four of them duplicates) or three per wave of four");
    // ring slots: k-step h + BX_PD is written (after the barrier of half-item h) into the slot of k-step h + BX_PD - NBV, which
    // every wave has finished reading once it is past half-item h - 1: NBV >= BX_PD + 1
    constexpr int NBV = ZOUT ? 8 : BX_NB;
    static_assert(NBV >= PD + 1, "ring too short for the prefetch distance");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NBV * BX_BH_BYTES + (ZOUT ? BX_W2_BYTES : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    if (a.m < 0) smem[tid] = 0;       // never taken: the array is otherwise written by DMA only, which the compiler cannot see

    const int rb_lo = (int)((int64_t)blockIdx.x * a.n_rb / gridDim.x);
    const int nb = (int)((int64_t)(blockIdx.x + 1) * a.n_rb / gridDim.x) - rb_lo;
    if (nb <= 0) return;
    const int row_last = min(a.m, (rb_lo + nb) * 32) - 1;       // rows past this workgroup's share are read as its last row (cache hits, no HBM traffic)
    const int n_rounds = (nb + NW - 1) / NW;
    const int n_items = n_rounds * a.n_chunks;                   // chunks of this wave, all rounds
    const int n_hs = 2 * a.n_chunks;
    const int piece2 = NW == 4 ? wave + 4 : (wave < 4 ? wave + 8 : wave - 4);   // second DMA piece of this wave (eight waves: waves 4-7 repeat one of pieces 0-3)

    // the X values (and keep words) of chunk (round t, chunk c) of this lane's row
    auto load_raw = [&](BxRaw &r, int t, int c) __attribute__((always_inline)) {
        const int row = min((rb_lo + NW * t + wave) * 32 + li, row_last);
        const float *p = a.x + (size_t)row * a.ldx + c * BX_BK + 16 * hh;
        if (ABL & 1) p = a.x + lane * 16;
        // the keep word FIRST: 32 consecutive rows = one line (both lane halves read the same words).  Issued after the four
        // X loads the same load cost 9 us more per launch (tools/gemm_bf16x3.hip, ablation 128)
        if (DROP && !(ABL & 64) && !(ABL & 128)) bx_gload4(r.kw, a.bits + (size_t)c * a.m + row);
        bx_gload16<0>(r.v[0], p); bx_gload16<16>(r.v[1], p); bx_gload16<32>(r.v[2], p); bx_gload16<48>(r.v[3], p);
        if (DROP && !(ABL & 64)) {
            if (ABL & 128) bx_gload4(r.kw, a.bits + (size_t)c * a.m + row);
        } else {
            r.kw = 0u;
        }
    };
    constexpr int LA = (DROP && !(ABL & 64)) ? 5 : 4;                             // vector-memory operations of one load_raw
    // the W k-step of half-item hq (its index inside the round repeats with every round) into ring slot hq % BX_NB
    auto issue_b = [&](int hq) __attribute__((always_inline)) {
        if (ABL & 2) return;
        int hs = hq % n_hs;
        const uint32_t dst = lds0 + (hq % NBV) * BX_BH_BYTES;
        const uint4 *src = a.wp + (size_t)hs * (BX_BH_BYTES / 16);
        pg_glds16(src + wave * 64 + lane, dst + wave * 1024);
        pg_glds16(src + piece2 * 64 + lane, dst + piece2 * 1024);
        if (NW == 4) pg_glds16(src + (wave + 8) * 64 + lane, dst + (wave + 8) * 1024);
    };
    // what the W wait allows: the younger W pieces of this wave, BX_PD - 2 k-steps of 2 (eight waves) or 3 (four waves) pieces
    static_assert(PD >= 5 && PD <= 9, "BX_WAIT_W spells the counts out");
#define BX_WAIT_W() do { constexpr int n_ = (NW == 4 ? 3 : 2) * (PD - 2); \
        if (ABL & 1024) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else \
        if (ABL & 16) PG_WAIT_BARRIER(0); else if (n_ == 6) PG_WAIT_BARRIER(6); else if (n_ == 8) PG_WAIT_BARRIER(8); else if (n_ == 9) PG_WAIT_BARRIER(9); \
        else if (n_ == 10) PG_WAIT_BARRIER(10); else if (n_ == 12) PG_WAIT_BARRIER(12); else if (n_ == 14) PG_WAIT_BARRIER(14); \
        else if (n_ == 15) PG_WAIT_BARRIER(15); else if (n_ == 18) PG_WAIT_BARRIER(18); else if (n_ == 21) PG_WAIT_BARRIER(21); else PG_WAIT_BARRIER(0); } while (0)

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((uint32_t)a.m * (uint32_t)a.ldo * 4u), 0x00020000);
    auto store_block = [&](f32x16 &v, int row0, int col, bool live) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float x = (a.relu && !(v[r] > 0.f)) ? 0.f : v[r];
            // a wave without a row block of its own in this round stores past the buffer's range: dropped by the hardware, counted like the others
            const uint32_t off = live ? ((uint32_t)row * (uint32_t)a.ldo + (uint32_t)col) * 4u : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), orsrc, (int)off, 0, 0);
            v[r] = 0.f;
        }
    };

    // pair i (0..3) of k-step s of a landed chunk -> word i of the three planes (keep bits applied first)
    auto split_pair = [&](BxPlanes &P, const BxRaw &r, uint32_t win, int s, int i) __attribute__((always_inline)) {
        const f32x4 v = r.v[2 * s + (i >> 1)];
        float x0 = (i & 1) ? v[2] : v[0], x1 = (i & 1) ? v[3] : v[1];
        if (DROP && !(ABL & 32)) {
            const int j = 8 * s + 2 * i;                         // bit j of the window <-> this lane's k value j of the chunk
            x0 = __uint_as_float(__float_as_uint(x0) & (uint32_t)(((int32_t)(win << (31 - j))) >> 31));
            x1 = __uint_as_float(__float_as_uint(x1) & (uint32_t)(((int32_t)(win << (30 - j))) >> 31));
        }
        if (ABL & 4) { P.w[0][i] = __float_as_uint(x0); P.w[1][i] = __float_as_uint(x1); P.w[2][i] = P.w[0][i]; return; }
        bx_split2(x0, x1, P.w[0][i], P.w[1][i], P.w[2][i]);
    };
    auto window = [&](const BxRaw &r, int t, int c) __attribute__((always_inline)) -> uint32_t {
        if (!DROP) return 0u;
        return r.kw >> (16 * hh);                                // bit j <-> k value 16 hh + j of the chunk
    };
    // the three planes of column block N of the W k-step in ring slot hq % BX_NB: issued, not waited for
    auto read_b = [&](BxB3 &b, int hq, auto n_tag) __attribute__((always_inline)) {
        constexpr int N = (ABL & 8) ? 0 : decltype(n_tag)::value;
        const uint32_t addr = lds0 + (hq % NBV) * BX_BH_BYTES + lane * 16;
        bx_ldsread16<(0 * 4 + N) * 1024>(b.h, addr);
        bx_ldsread16<(1 * 4 + N) * 1024>(b.m, addr);
        bx_ldsread16<(2 * 4 + N) * 1024>(b.l, addr);
    };
    // acc += A . B over the NP plane products, smallest weights first
    auto mac = [&](f32x16 &c, const BxPlanes &A, const BxB3 &B) __attribute__((always_inline)) {
        const bf16x8 ah = bx_plane(A, 0), am = bx_plane(A, 1), al = bx_plane(A, 2);
        if (ZOUT) {                                              // the transposed tile: W planes as the MFMA's A operand
            c = MFMA_BF16(B.h, al, c);
            c = MFMA_BF16(B.m, am, c);
            c = MFMA_BF16(B.l, ah, c);
            c = MFMA_BF16(B.h, am, c);
            c = MFMA_BF16(B.m, ah, c);
            c = MFMA_BF16(B.h, ah, c);
            return;
        }
        if (NP >= 8) { c = MFMA_BF16(al, B.m, c); c = MFMA_BF16(am, B.l, c); }
        c = MFMA_BF16(al, B.h, c);
        c = MFMA_BF16(am, B.m, c);
        c = MFMA_BF16(ah, B.l, c);
        c = MFMA_BF16(am, B.h, c);
        c = MFMA_BF16(ah, B.m, c);
        c = MFMA_BF16(ah, B.h, c);
    };
    // ZOUT: the round's tile of H (acc, transposed: this lane's row) -> Z0's tile.  Once per round; plain LDS reads and stores.
    auto second_product = [&](int row0, bool live) __attribute__((always_inline)) {
        const uint4 *img = reinterpret_cast<const uint4 *>(smem + NBV * BX_BH_BYTES);
        f32x16 z[2];
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) z[cb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            BxPlanes P;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float x0 = acc[s >> 1][8 * (s & 1) + 2 * i], x1 = acc[s >> 1][8 * (s & 1) + 2 * i + 1];
                x0 = (a.relu && !(x0 > 0.f)) ? 0.f : x0;
                x1 = (a.relu && !(x1 > 0.f)) ? 0.f : x1;
                bx_split2(x0, x1, P.w[0][i], P.w[1][i], P.w[2][i]);
            }
            const bf16x8 hh_ = bx_plane(P, 0), hm = bx_plane(P, 1), hl = bx_plane(P, 2);
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                if (cb == 1 && a.p2 <= 32) continue;
                const int piece = s * 2 + cb;
                const bf16x8 wh = __builtin_bit_cast(bf16x8, img[(piece * 3 + 0) * 64 + lane]);
                const bf16x8 wm = __builtin_bit_cast(bf16x8, img[(piece * 3 + 1) * 64 + lane]);
                const bf16x8 wl = __builtin_bit_cast(bf16x8, img[(piece * 3 + 2) * 64 + lane]);
                z[cb] = MFMA_BF16(wl, hh_, z[cb]);
                z[cb] = MFMA_BF16(wm, hm, z[cb]);
                z[cb] = MFMA_BF16(wh, hl, z[cb]);
                z[cb] = MFMA_BF16(wm, hh_, z[cb]);
                z[cb] = MFMA_BF16(wh, hm, z[cb]);
                z[cb] = MFMA_BF16(wh, hh_, z[cb]);
            }
        }
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
        const int row = row0 + li;
        if (live && row < a.m) {
            float *zp = a.z0 + (size_t)row * a.ldz;
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c0 = 32 * cb + 8 * q + 4 * hh;
                    if (c0 + 4 <= a.p2) {
                        *reinterpret_cast<float4 *>(zp + c0) = make_float4(z[cb][4 * q], z[cb][4 * q + 1], z[cb][4 * q + 2], z[cb][4 * q + 3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; i++) if (c0 + i < a.p2) zp[c0 + i] = z[cb][4 * q + i];
                    }
                }
        }
    };
    // one group: NP MFMAs with 3 of the group's other instructions (split VALU, LDS reads) behind each
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
        }
    };

    if (ZOUT) {                                                  // the W2 image, before any hand-counted operation is issued
        uint4 *img = reinterpret_cast<uint4 *>(smem + NBV * BX_BH_BYTES);
        for (int i = tid; i < BX_W2_BYTES / 16; i += 64 * NW) img[i] = a.w2p[i];
        __syncthreads();
    }
    // ---- prologue: W k-steps 0 .. BX_PD-1, X chunks 0 and 1; everything landed
    BxRaw R0, R1, R2;
#pragma unroll
    for (int q = 0; q < PD; q++) issue_b(q);
    load_raw(R0, 0, 0);
    {
        const int t1 = a.n_chunks > 1 ? 0 : 1, c1 = a.n_chunks > 1 ? 1 : 0;
        load_raw(R1, min(t1, n_rounds - 1), n_items > 1 ? c1 : 0);
    }
    BX_WAIT_RAW(0, R0);
    BX_WAIT_RAW(0, R1);
    __builtin_amdgcn_s_barrier();
    BxPlanes P0, P1;
    {
        const uint32_t w0 = window(R0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) split_pair(P0, R0, w0, 0, i);
    }
    BxB3 Bc, Bn;
    read_b(Bc, 0, BxN<0>());
    BX_WAIT_LDS(Bc);

    // one chunk = two half-items.  Ra = raw[g] (its second half still to be split), Rb = raw[g+1] (landed by the second
    // half-item), Rc = the buffer raw[g+2] goes into
    auto chunk = [&](BxRaw &Ra, BxRaw &Rb, BxRaw &Rc, int g) __attribute__((always_inline)) {
        const int t = g / a.n_chunks, c = g - t * a.n_chunks;
        int t1 = t, c1 = c + 1; if (c1 == a.n_chunks) { c1 = 0; ++t1; }
        int t2 = t1, c2 = c1 + 1; if (c2 == a.n_chunks) { c2 = 0; ++t2; }
        if (t1 >= n_rounds) { t1 = n_rounds - 1; c1 = a.n_chunks - 1; }       // past the end: the same operations on the last chunk (nobody uses them)
        if (t2 >= n_rounds) { t2 = n_rounds - 1; c2 = a.n_chunks - 1; }
        const int h0 = 2 * g;
        const uint32_t wa = window(Ra, t, c);
        // ================= half-item (g, 0): MFMAs on P0; raw[g]'s second half -> P1; raw[g+2] issued
        if (ABL & 512) load_raw(Rc, t2, c2);                     // (experiment: the X loads at the top of the chunk)
        read_b(Bn, h0, BxN<1>());
        __builtin_amdgcn_sched_barrier(0);
        split_pair(P1, Ra, wa, 1, 0);
        mac(acc[0], P0, Bc);
        interleave();
        BX_PIN(P1, 0);
        __builtin_amdgcn_sched_barrier(0);
        BX_WAIT_LDS(Bn);
        read_b(Bc, h0, BxN<2>());
        __builtin_amdgcn_sched_barrier(0);
        split_pair(P1, Ra, wa, 1, 1);
        mac(acc[1], P0, Bn);
        interleave();
        BX_PIN(P1, 1);
        __builtin_amdgcn_sched_barrier(0);
        BX_WAIT_LDS(Bc);
        // W k-step h0+1 has landed: issued BX_PD-1 half-items ago, BX_PD-2 younger k-steps of two pieces each.  Every wave is past
        // half-item h0-1: its ring slot takes k-step h0+BX_PD
        BX_WAIT_W();
        if (ABL & 256) { load_raw(Rc, t2, c2); issue_b(h0 + PD); }      // (experiment: the X loads ahead of the W pieces)
        else if (ABL & 512) issue_b(h0 + PD);
        else { issue_b(h0 + PD); load_raw(Rc, t2, c2); }
        read_b(Bn, h0, BxN<3>());
        __builtin_amdgcn_sched_barrier(0);
        split_pair(P1, Ra, wa, 1, 2);
        mac(acc[2], P0, Bc);
        interleave();
        BX_PIN(P1, 2);
        __builtin_amdgcn_sched_barrier(0);
        BX_WAIT_LDS(Bn);
        read_b(Bc, h0 + 1, BxN<0>());
        __builtin_amdgcn_sched_barrier(0);
        split_pair(P1, Ra, wa, 1, 3);
        mac(acc[3], P0, Bn);
        interleave();
        BX_PIN(P1, 3);
        __builtin_amdgcn_sched_barrier(0);
        BX_WAIT_LDS(Bc);
        // ================= half-item (g, 1): MFMAs on P1; raw[g+1]'s first half -> P0
        // raw[g+1] has landed: issued three half-items ago; the younger loads of its kind are raw[g+2]'s
        if (ABL & 16) BX_WAIT_RAW(0, Rb);
        else if (DROP && !(ABL & 64)) BX_WAIT_RAW(5, Rb);
        else BX_WAIT_RAW(4, Rb);
        const uint32_t wb = window(Rb, t1, c1);
        read_b(Bn, h0 + 1, BxN<1>());
        __builtin_amdgcn_sched_barrier(0);
        split_pair(P0, Rb, wb, 0, 0);
        mac(acc[0], P1, Bc);
        interleave();
        BX_PIN(P0, 0);
        __builtin_amdgcn_sched_barrier(0);
        BX_WAIT_LDS(Bn);
        read_b(Bc, h0 + 1, BxN<2>());
        __builtin_amdgcn_sched_barrier(0);
        split_pair(P0, Rb, wb, 0, 1);
        mac(acc[1], P1, Bn);
        interleave();
        BX_PIN(P0, 1);
        __builtin_amdgcn_sched_barrier(0);
        BX_WAIT_LDS(Bc);
        BX_WAIT_W();
        issue_b(h0 + 1 + PD);
        read_b(Bn, h0 + 1, BxN<3>());
        __builtin_amdgcn_sched_barrier(0);
        split_pair(P0, Rb, wb, 0, 2);
        mac(acc[2], P1, Bc);
        interleave();
        BX_PIN(P0, 2);
        __builtin_amdgcn_sched_barrier(0);
        BX_WAIT_LDS(Bn);
        read_b(Bc, h0 + 2, BxN<0>());
        __builtin_amdgcn_sched_barrier(0);
        split_pair(P0, Rb, wb, 0, 3);
        mac(acc[3], P1, Bn);
        interleave();
        BX_PIN(P0, 3);
        __builtin_amdgcn_sched_barrier(0);
        BX_WAIT_LDS(Bc);
        if (c == a.n_chunks - 1) {
            const bool live = NW * t + wave < nb;
            if (ZOUT) {
                second_product((rb_lo + NW * t + wave) * 32, live);
            } else {
#pragma unroll
                for (int n = 0; n < 4; n++) store_block(acc[n], (rb_lo + NW * t + wave) * 32, 32 * n + li, live);
            }
        }
    };
    int g = 0;
    for (; g + 3 <= n_items; g += 3) {
        chunk(R0, R1, R2, g);
        chunk(R1, R2, R0, g + 1);
        chunk(R2, R0, R1, g + 2);
    }
    if (g < n_items) {                                       // (nested, not two tests in a row: the second chunk is reachable only through the
        chunk(R0, R1, R2, g); g++;                           //  first, whose wait covers R1 — tools/check_asm_loads.py walks every CFG path)
        if (g < n_items) { chunk(R1, R2, R0, g); g++; }
    }
    // Nothing of this wave may still be landing when it ends — and the raw buffers stay allocated until here: the loads issued
    // past the end of the share are never used, so hipcc would otherwise hand their destination registers to the epilogue
    // while the loads are still in flight (an asm output counts as written when the statement is issued).
    BX_WAIT_RAW(0, R0);
    BX_WAIT_RAW(0, R1);
    BX_WAIT_RAW(0, R2);
}
#undef BX_WAIT_W

// ------------------------------------------------------------------------------------------------ backward
// dW[K x 128] = X~^T[K x m] . dH0[m x 128]: the rows are the reduction.  Grid (feature range of 128, row split); a workgroup is
// FOUR waves: wave w owns the 32 features F0 = 128 fr + 32 w (x all 128 columns: 4 accumulator blocks) — and produces the
// planes of column block w of dH0 for all four.  A step is 16 rows:
//   * MFMA operand A = X~^T: lane (f, h) holds X[r0 + 8h + j][F0 + f], j = 0..7 — eight dword loads, each a whole 128-byte line
//     across the 32 lanes of a half-wave; no transposition, no LDS.  With dropout a lane also loads, per value, the word of
//     the keep-bit array that holds its bit (neighbouring lanes share it: one or two L1 lines per row).
//   * operand B = dH0: lane (c, h) of wave w loads dH0[r0 + 8h + j][32w + c] the same way, splits it and writes the three
//     16-byte plane pieces into the step's LDS image (12 KB: [plane][column block][lane], as the forward's W k-steps); a
//     barrier per step publishes it; every wave reads all four column blocks back.
//   Both operands' values of step s+1 are split (8 pairs) while step s's 24 MFMAs run; loads run two steps ahead.
// Rows past the split's end contribute zero (lane masks); rows past m are outside the buffer descriptors and read as zero.
// One f32 slab [K x 128] per row split, summed in split order by slab_reduce_kernel (no atomics: the same bits every run).

struct Bx3BwdArgs {
    const float *x; int ldx;          // X [m x ldx], ldx >= 128 * gridDim.x, zero padded past K
    const float *dout; int ldd;       // dH0 [m x 128]
    float *slab; int p_ld;            // [gridDim.y][K][p_ld]
    int m, K, rps;                    // rows per split: a multiple of 16
    int split0;                       // blockIdx.y == 0 is row split number split0 (a launch may cover a range of the splits)
    const uint32_t *bits;             // keep bits of X, chunk-major (dropbits_cm_body), NULL: no dropout
    float scale;                      // 1 / (1 - p) with dropout, applied to the stored partial
};

template <int OFF>
__device__ __forceinline__ void bx_ldswrite16(uint32_t addr, u32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ float bx_keep(float x, uint64_t lane_mask) {     // x where the lane's bit is set, else 0
    float y;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(y) : "v"(x), "s"(lane_mask));
    return y;
}
struct BxRaw8 { float a[8], b[8]; uint32_t k; };             // k: with dropout, lane l holds the keep word of row 16 s + (l & 15) for this wave's 32 features
#define BX_WAIT_RAW8(N, r) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"((r).a[0]), "+v"((r).a[1]), "+v"((r).a[2]), "+v"((r).a[3]), "+v"((r).a[4]), "+v"((r).a[5]), "+v"((r).a[6]), "+v"((r).a[7]), \
                                                                   "+v"((r).b[0]), "+v"((r).b[1]), "+v"((r).b[2]), "+v"((r).b[3]), "+v"((r).b[4]), "+v"((r).b[5]), "+v"((r).b[6]), "+v"((r).b[7]), \
                                                                   "+v"((r).k) :: "memory")

template <bool DROP, int NP, int ORD = 0>      // ORD: load-order experiments of tools/gemm_bf16x3.hip (1: keep word first, 2: dH0 before X, 4: X and dH0 interleaved)
__global__ __launch_bounds__(256, 2) void dense_bwd_bf16x3_kernel(Bx3BwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BX_BH_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    if (a.m < 0) smem[tid] = 0;       // never taken: the array is otherwise touched by inline asm only
    const int fr = blockIdx.x, split = blockIdx.y + a.split0;
    const int F0 = 128 * fr + 32 * wave;
    const int r_lo = split * a.rps, r_hi = min(a.m, r_lo + a.rps);
    const int n_steps = r_hi > r_lo ? (r_hi - r_lo + 15) >> 4 : 0;

    const u32x4 rs_x = bx_make_rsrc(a.x, (uint32_t)a.m * (uint32_t)a.ldx * 4u);
    const u32x4 rs_d = bx_make_rsrc(a.dout, (uint32_t)a.m * (uint32_t)a.ldd * 4u);
    uint32_t so_x[8], so_d[8];                                   // row j of a lane's eight: scalar byte offsets
#pragma unroll
    for (int j = 0; j < 8; j++) { so_x[j] = (uint32_t)j * (uint32_t)a.ldx * 4u; so_d[j] = (uint32_t)j * (uint32_t)a.ldd * 4u; }
    const uint32_t vo_x0 = ((uint32_t)(r_lo + 8 * hh) * (uint32_t)a.ldx + (uint32_t)(F0 + li)) * 4u;
    const uint32_t vo_d0 = ((uint32_t)(r_lo + 8 * hh) * (uint32_t)a.ldd + (uint32_t)(32 * wave + li)) * 4u;
    const uint32_t step_x = 16u * (uint32_t)a.ldx * 4u, step_d = 16u * (uint32_t)a.ldd * 4u;
    // chunk-major keep words: this wave's 32 features are chunk F0 / 32 (a chunk past the matrix: out of the buffer's range, zeros)
    const u32x4 rs_k = bx_make_rsrc(a.bits, (uint32_t)((a.K + 31) / 32) * (uint32_t)a.m * 4u);
    const uint32_t vo_k0 = ((uint32_t)(F0 >> 5) * (uint32_t)a.m + (uint32_t)(r_lo + (lane & 15))) * 4u;   // (4 * chunks * m < 2^32: checked at the launch site)
    constexpr int LR = DROP ? 17 : 16;                           // vector-memory operations of one load_raw
    auto load_raw = [&](BxRaw8 &r, int s) __attribute__((always_inline)) {
        const uint32_t vx = vo_x0 + (uint32_t)s * step_x, vd = vo_d0 + (uint32_t)s * step_d;
        auto load_k = [&]() __attribute__((always_inline)) {
            uint32_t w;
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(w) : "v"(vo_k0 + (uint32_t)s * 64u), "s"(rs_k) : "memory");
            r.k = w;
        };
        if (DROP && (ORD & 1)) load_k();
        if (ORD & 4) {
#pragma unroll
            for (int j = 0; j < 8; j++) { bx_bload4(r.a[j], vx, rs_x, so_x[j]); bx_bload4(r.b[j], vd, rs_d, so_d[j]); }
        } else if (ORD & 2) {
#pragma unroll
            for (int j = 0; j < 8; j++) bx_bload4(r.b[j], vd, rs_d, so_d[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) bx_bload4(r.a[j], vx, rs_x, so_x[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) bx_bload4(r.a[j], vx, rs_x, so_x[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) bx_bload4(r.b[j], vd, rs_d, so_d[j]);
        }
        if (DROP && !(ORD & 1)) load_k();
        if (!DROP) r.k = 0u;
    };
    // lane masks of step s: bit l of M[j] = value j of lane l counts (its row is inside the split; with dropout: and is kept)
    struct Masks { uint64_t m[8]; };
    auto field = [&](int row) __attribute__((always_inline)) -> uint32_t {       // all ones when `row` is inside this split
        return row < r_hi ? 0xFFFFFFFFu : 0u;
    };
    // (with dropout: after raw(s) has landed — the keep word of row r0 + q sits in lane q and IS the lane mask of that row's values)
    auto make_masks = [&](Masks &M, int s, const BxRaw8 &r) __attribute__((always_inline)) {
        const int r0 = r_lo + 16 * s;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t lo = field(r0 + j), hi = field(r0 + 8 + j);
            if (DROP) {
                lo &= (uint32_t)__builtin_amdgcn_readlane((int)r.k, j);
                hi &= (uint32_t)__builtin_amdgcn_readlane((int)r.k, 8 + j);
            }
            M.m[j] = (uint64_t)lo | ((uint64_t)hi << 32);
            asm volatile("" : "+s"(M.m[j]));                     // an SGPR pair whatever the value (v_cndmask takes no literal mask)
        }
    };
    auto split_a_pair = [&](BxPlanes &P, const BxRaw8 &r, const Masks &M, int s, int i) __attribute__((always_inline)) {
        const float x0 = bx_keep(r.a[2 * i], M.m[2 * i]), x1 = bx_keep(r.a[2 * i + 1], M.m[2 * i + 1]);
        bx_split2(x0, x1, P.w[0][i], P.w[1][i], P.w[2][i]);
    };
    auto split_b_pair = [&](BxPlanes &P, const BxRaw8 &r, int i) __attribute__((always_inline)) {
        bx_split2(r.b[2 * i], r.b[2 * i + 1], P.w[0][i], P.w[1][i], P.w[2][i]);
    };
    // this wave's column block of the step's B image
    auto write_b = [&](const BxPlanes &P, int slot) __attribute__((always_inline)) {
        const uint32_t addr = lds0 + slot * BX_BH_BYTES + (wave * 64 + lane) * 16;
        bx_ldswrite16<0 * 4096>(addr, (u32x4){P.w[0][0], P.w[0][1], P.w[0][2], P.w[0][3]});
        bx_ldswrite16<1 * 4096>(addr, (u32x4){P.w[1][0], P.w[1][1], P.w[1][2], P.w[1][3]});
        bx_ldswrite16<2 * 4096>(addr, (u32x4){P.w[2][0], P.w[2][1], P.w[2][2], P.w[2][3]});
    };
    auto read_b = [&](BxB3 &b, int slot, auto n_tag) __attribute__((always_inline)) {
        constexpr int N = decltype(n_tag)::value;
        const uint32_t addr = lds0 + slot * BX_BH_BYTES + lane * 16;
        bx_ldsread16<(0 * 4 + N) * 1024>(b.h, addr);
        bx_ldsread16<(1 * 4 + N) * 1024>(b.m, addr);
        bx_ldsread16<(2 * 4 + N) * 1024>(b.l, addr);
    };
    auto mac = [&](f32x16 &c, const BxPlanes &A, const BxB3 &B) __attribute__((always_inline)) {
        const bf16x8 ah = bx_plane(A, 0), am = bx_plane(A, 1), al = bx_plane(A, 2);
        if (NP >= 8) { c = MFMA_BF16(al, B.m, c); c = MFMA_BF16(am, B.l, c); }
        c = MFMA_BF16(al, B.h, c);
        c = MFMA_BF16(am, B.m, c);
        c = MFMA_BF16(ah, B.l, c);
        c = MFMA_BF16(am, B.h, c);
        c = MFMA_BF16(ah, B.m, c);
        c = MFMA_BF16(ah, B.h, c);
    };
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    if (n_steps > 0) {
        // ---- prologue: steps 0 and 1 requested; step 0 split, its B image published
        BxRaw8 R0, R1;
        load_raw(R0, 0);
        load_raw(R1, 1);                                         // (past the split's end: masked; past m: zeros)
        if (DROP) BX_WAIT_RAW8(17, R0); else BX_WAIT_RAW8(16, R0);
        BxPlanes PA0, PA1, PB;
        Masks M;
        make_masks(M, 0, R0);
#pragma unroll
        for (int i = 0; i < 4; i++) { split_a_pair(PA0, R0, M, 0, i); split_b_pair(PB, R0, i); }
        write_b(PB, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        BxB3 Bc, Bn;
        read_b(Bc, 0, BxN<0>());
        BX_WAIT_LDS(Bc);
        // one step: MFMAs on PAc and the B image in `slot`; Rn = raw(s+1) (landing) -> PAn and the image in slot ^ 1; Rf = the buffer raw(s+2) goes into
        auto step = [&](BxPlanes &PAc, BxPlanes &PAn, BxRaw8 &Rn, BxRaw8 &Rf, int s) __attribute__((always_inline)) {
            const int slot = s & 1;
            load_raw(Rf, s + 2);
            if (DROP) BX_WAIT_RAW8(17, Rn); else BX_WAIT_RAW8(16, Rn);   // raw(s+1): issued a step ago; younger: the loads just issued
            make_masks(M, s + 1, Rn);
            read_b(Bn, slot, BxN<1>());
            __builtin_amdgcn_sched_barrier(0);
            split_b_pair(PB, Rn, 0); split_b_pair(PB, Rn, 1);
            mac(acc[0], PAc, Bc);
            interleave();
            BX_PIN(PB, 0); BX_PIN(PB, 1);
            __builtin_amdgcn_sched_barrier(0);
            BX_WAIT_LDS(Bn);
            read_b(Bc, slot, BxN<2>());
            __builtin_amdgcn_sched_barrier(0);
            split_b_pair(PB, Rn, 2); split_b_pair(PB, Rn, 3);
            mac(acc[1], PAc, Bn);
            interleave();
            BX_PIN(PB, 2); BX_PIN(PB, 3);
            __builtin_amdgcn_sched_barrier(0);
            BX_WAIT_LDS(Bc);
            write_b(PB, slot ^ 1);                               // the image of step s+1: its slot was last read in step s-1
            read_b(Bn, slot, BxN<3>());
            __builtin_amdgcn_sched_barrier(0);
            split_a_pair(PAn, Rn, M, s + 1, 0); split_a_pair(PAn, Rn, M, s + 1, 1);
            mac(acc[2], PAc, Bc);
            interleave();
            BX_PIN(PAn, 0); BX_PIN(PAn, 1);
            __builtin_amdgcn_sched_barrier(0);
            BX_WAIT_LDS(Bn);                                     // (covers the three plane writes as well)
            split_a_pair(PAn, Rn, M, s + 1, 2); split_a_pair(PAn, Rn, M, s + 1, 3);
            mac(acc[3], PAc, Bn);
            interleave();
            BX_PIN(PAn, 2); BX_PIN(PAn, 3);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's piece of image s+1 is in LDS; all are done with image s
            read_b(Bc, slot ^ 1, BxN<0>());
            BX_WAIT_LDS(Bc);
        };
        int s = 0;
        for (; s + 2 <= n_steps; s += 2) {
            step(PA0, PA1, R1, R0, s);
            step(PA1, PA0, R0, R1, s + 1);
        }
        if (s < n_steps) step(PA0, PA1, R1, R0, s);
        // the loads requested past the split's end are never used: their destination registers stay allocated until they have
        // landed (hipcc would otherwise reuse them for the epilogue below while the loads are in flight)
        BX_WAIT_RAW8(0, R0);
        BX_WAIT_RAW8(0, R1);
    }
    // ---- this split's partial [128 features x 128 columns] of the slab
    float *slab = a.slab + (size_t)split * a.K * a.p_ld;
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int feat = F0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (feat < a.K) slab[(size_t)feat * a.p_ld + 32 * n + li] = acc[n][r] * a.scale;
        }
}

// dense_persist.h — persistent, LDS-DMA-fed forms of the two compute-bound GEMMs of a dense-X first layer at
// p = 128 output columns (replace cuda_SparseMatmul_forward/_backward, /root/reference/src/cuda/cuda_kernel.cu:100-122):
//     forward   H0[m x 128]  = X~[m x K] . W[K x 128]
//     backward  dW[K x 128]  = X~^T[K x m] . dH0[m x 128]
// Why a second form of dense_tile128.h's kernels (which stay: other widths, unaligned inputs, A/B runs).  Round 2 measured
// where the 128 x 128 tile's 0.36 ms go: 19 % is the tail of 1821 equal tiles on 512 slots, and inside a
// tile the staging pipeline (global -> registers -> wait -> LDS -> barrier, ONE chunk ahead) is as long as the MFMA
// pipeline and only half overlapped.  Both are structural:
//  * one workgroup per CU for the whole launch, given a CONTIGUOUS share of the rows that is balanced to one 32-row MFMA
//    block (forward) / one row (backward): no tile quantisation.  The forward's last, partial tile deals its (row block,
//    column block) units over the four SIMDs, so a partial tile costs its share of a full one;
//  * operands reach LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of three stages, two K chunks ahead of the
//    MFMAs, with counted vmcnt waits and ONE barrier per chunk; no staging registers, so depth costs only LDS;
//  * the input dropout moves from the staging pass to the operand read (a select on the A fragment), the keep bits of a
//    chunk travelling through the same ring.
// Exact f32 (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain).  Within an 8-wide k group the forward visits k in the order
// 0,4,1,5,2,6,3,7 (a lane's 16-byte LDS read supplies four MFMA steps); dense_tile128.h's forward kernels walk their LDS
// tiles in the same order (T_KO), so both forms give the same bits WITHOUT dropout and at dropout 0.5 (scale 2 is exact
// in either operand); at any other dropout rate this form multiplies (x . m) by (scale * w) where the tiles multiply
// (x * scale) by w — one rounding apart per product, tested at 0.3 with that tolerance (tests/test_ops_gpu.py,
// test_dense_forward_persistent_and_tile_kernels_give_the_same_bits).  The validation lane launches the tile kernel, and an
// evaluation forward has no dropout: the two-stream epoch reproduces the one-stream epoch's validation losses exactly.
#pragma once
#include "dense_tile128.h"

constexpr int PG_STAGES = 3;
constexpr int PG_BK = 32;                                   // K (forward) / rows (backward) per chunk
constexpr int PG_ROWS = 256;                                // rows of a full forward tile: 8 waves x 32
constexpr int PG_A_BYTES = PG_ROWS * PG_BK * 4;             // 32768
constexpr int PG_B_BYTES = PG_BK * 128 * 4;                 // 16384
constexpr int PG_K_BYTES = PG_ROWS * 8;                     // keep-bit windows: two words per row
constexpr int PG_STAGE_BYTES = PG_A_BYTES + PG_B_BYTES + PG_K_BYTES;   // 51200; x3 = 153600 <= 160 KiB

// LDS-DMA: 16 (4) bytes per lane from a per-lane global address to wave-uniform LDS base + lane * 16 (4).  Inline asm
// so that hipcc neither counts these loads nor drains them at its own waits: completion is counted by hand (PG_WAIT).
// M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void pg_glds16(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void pg_glds4(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// every wave: its own DMA pieces down to the N youngest have landed; then the workgroup barrier (all pieces landed, and
// every wave has finished reading the stage that the next issue overwrites)
#define PG_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")

// W[K x 128] -> the forward's B image, chunk by chunk: float index kg*1024 + col*8 + 4*(hh ^ ((col >> 3) & 1)) + t holds
// W[8*kg + 4*hh + t][col] (rows past K: zero).  A lane's 16-byte read then yields the B values of four MFMA steps, and the
// half swap makes the 16-lane groups of a ds_read_b128 cover all 64 banks (conflict-free).
// `scale` (the input dropout's 1/(1-p), 1 without dropout) is folded in here: X~ . W = (X . m) . (scale W), so the kernel's
// per-element work on A is one AND with the keep mask (for p = 0.5 the product is the same bits; otherwise scale*w is
// rounded once more than x*scale would be — inside the bound the oracle comparison uses).
__device__ inline void pg_pack_w_body(int idx, const float *__restrict__ w, int ldw, int K, int n_kg, float *__restrict__ wp, float scale) {
    if (idx >= n_kg * 256) return;                               // idx = (kg, hh, col)
    const int col = idx & 127, hh = (idx >> 7) & 1, kg = idx >> 8;
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int k = 8 * kg + 4 * hh + t;
        v[t] = k < K ? w[(size_t)k * ldw + col] * scale : 0.f;
    }
    *reinterpret_cast<float4 *>(wp + (size_t)kg * 1024 + col * 8 + 4 * (hh ^ ((col >> 3) & 1))) = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ __launch_bounds__(256) void pg_pack_w_kernel(const float *__restrict__ w, int ldw, int K, int n_kg, float *__restrict__ wp, float scale) {
    pg_pack_w_body(blockIdx.x * blockDim.x + threadIdx.x, w, ldw, K, n_kg, wp, scale);
}
// The two small launches in front of a training forward in one: workgroups [0, n_bits_wgs) draw the keep bits
// (dropbits_block_kernel), the n_kg workgroups behind them pack W (pg_pack_w_kernel).  Independent work, one launch less
// on the critical path of every epoch (≈ 4 us).
__global__ __launch_bounds__(256) void dropbits_pack_w_kernel(uint32_t *__restrict__ bits, int64_t n_elems, int thr, uint64_t seed,
                                                              const uint32_t *d_epoch, uint64_t block0, int n_bits_wgs,
                                                              const float *__restrict__ w, int ldw, int K, int n_kg,
                                                              float *__restrict__ wp, float scale) {
    if ((int)blockIdx.x < n_bits_wgs) dropbits_block_body((int64_t)blockIdx.x * 256 + threadIdx.x, bits, n_elems, thr, seed, d_epoch, block0);
    else pg_pack_w_body(((int)blockIdx.x - n_bits_wgs) * 256 + threadIdx.x, w, ldw, K, n_kg, wp, scale);
}

struct PersistFwdArgs {
    const float *x; int ldx;          // X, 16-byte aligned rows, ldx >= round_up(K, 32) with zero padding
    const float *wp;                  // packed W (pg_pack_w_kernel), n_chunks * 4 k-groups
    float *out; int ldo;              // H0 [m x 128]; m * ldo * 4 < 2^32 (buffer stores)
    int m, K, n_chunks, n_rb;         // n_rb = ceil(m / 32)
    const uint32_t *bits;             // keep bits of the stored elements (element row*K + col), NULL: no dropout
    int relu;
    int dbg_linear;                   // EXPERIMENT: read A as if X were stored tile-major (wrong results, right traffic shape)
};

struct PgFrag { float4 x; float4 y[4]; };
template <bool V> struct PgTag { static constexpr bool value = V; };

// Schedule of one item (tile, chunk) per wave, four k groups of 16 MFMAs each:
//     MFMA group 0 | read fragments of group 1
//     MFMA group 1 | read fragments of group 2
//     wait: this wave's DMA pieces of the NEXT chunk have landed; workgroup barrier
//     MFMA group 2 | read group 3 | issue the DMA pieces of the chunk after next, one behind each of the first MFMAs
//     MFMA group 3 | read the next item's keep bits and group-0 fragments
// so a wave reaches the barrier with 32 MFMAs' worth of operands already in registers, DMA issue and every LDS read sit
// in the shadow of MFMAs, and the pipe has work on both sides of the barrier.  (The first form of this kernel put the
// barrier, the DMA issue and the first fragment reads at the top of the item: both waves of a SIMD did all three in
// lockstep with the pipe idle — 64 % MFMA-busy, SQ_WAIT_ANY 25 % of the wave cycles, profiles/r03_gemm_pmc_first_form.json.)
// Ring: three stages; at the barrier of item g every wave has finished reading item g-1's stage, which is where the
// pieces of item g+2 go; they have a whole item (8192 MFMA cycles per SIMD) to land.
template <bool DROP>
__global__ __launch_bounds__(512, 2) void dense_fwd_persist_kernel(PersistFwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PG_STAGES * PG_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    if (a.m < 0) smem[tid] = 0;       // never taken: the array is otherwise written by DMA only, which the compiler cannot see

    // this workgroup's share of the 32-row blocks
    const int rb_lo = (int)((int64_t)blockIdx.x * a.n_rb / gridDim.x);
    const int nb = (int)((int64_t)(blockIdx.x + 1) * a.n_rb / gridDim.x) - rb_lo;
    if (nb <= 0) return;
    const int n_tiles = (nb + 7) >> 3;
    const int n_items = n_tiles * a.n_chunks;
    constexpr int NL = DROP ? 7 : 6;  // DMA pieces per wave and chunk

    // ---- DMA piece u (0..NL-1) of item (t, c) for this wave, address computed on the spot (a few VALU in an MFMA's shadow)
    auto issue_piece = [&](int t, int c, int stage, int u) __attribute__((always_inline)) {
        const int tile_row0 = (rb_lo + 8 * t) * 32, k0 = c * PG_BK;
        const uint32_t sA = lds0 + stage * PG_STAGE_BYTES, sB = sA + PG_A_BYTES, sK = sB + PG_B_BYTES;
        if (u < 4) {                                             // A: piece q fills tile rows 8q .. 8q+7
            const int q = wave + 8 * u;
            const int r = 8 * q + (lane >> 3), slot = lane & 7;
            const int row = min(tile_row0 + r, a.m - 1);
            const int g = slot ^ ((r >> 1) & 7);                 // which 4-float group of the chunk this LDS slot holds
            const float *src = a.x + (size_t)row * a.ldx + k0 + 4 * g;
            if (a.dbg_linear) src = a.x + ((size_t)((rb_lo + 8 * t) >> 3) * a.n_chunks + c) * 8192 + q * 256 + lane * 4;
            pg_glds16(src, sA + q * 1024);
        } else if (u < 6) {                                      // B: a straight copy of the packed chunk
            const int q = wave + 8 * (u - 4);
            pg_glds16(a.wp + (size_t)c * 4096 + q * 256 + lane * 4, sB + q * 1024);
        } else if (DROP) {                                       // the two words that hold the chunk's 32 keep bits of each row
            const int d = wave * 64 + lane;
            const int row = min(tile_row0 + (d >> 1), a.m - 1);
            const uint64_t e0 = (uint64_t)row * a.K + k0;
            pg_glds4(a.bits + (e0 >> 5) + (d & 1), sK + wave * 256);
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    const int swz = (li >> 1) & 7;
    const int bsw = 16 * (hh ^ ((li >> 3) & 1));
    const int cb = wave & 3, half = wave >> 2;                   // partial tile: column block of this SIMD pair, row-block parity
    // the chunk's keep bits of tile row `trow` for this lane's k values: bit 8j + t  <->  k0 + 8j + 4hh + t
    auto window = [&](const unsigned char *st, int tile_row0, int trow, int k0) __attribute__((always_inline)) -> uint32_t {
        const uint2 kw = *reinterpret_cast<const uint2 *>(st + PG_A_BYTES + PG_B_BYTES + trow * 8);
        const int row = min(tile_row0 + trow, a.m - 1);
        const uint32_t sh = (uint32_t)(((uint64_t)row * a.K + k0) & 31);
        return (uint32_t)(((((uint64_t)kw.y) << 32) | kw.x) >> sh) >> (4 * hh);
    };
    // the A value of k-step t of group j with its keep bit applied: bit -> all-ones / zero mask (one v_bfe_i32), AND.  The
    // scale is in the packed W.  (Not a select on the loaded value: that let the compiler sink the LDS read into a branch
    // with a wait inside — the one-load-in-flight disease.)
    auto keepv = [&](float x, uint32_t win, int j, int t) __attribute__((always_inline)) -> float {
        const int32_t m = ((int32_t)(win << (31 - (8 * j + t)))) >> 31;
        return __uint_as_float(__float_as_uint(x) & (uint32_t)m);
    };

    // Fragments of k group j.  Full tile: wave w owns rows 32w .. 32w+31 and all four column blocks (x = its A rows,
    // y[n] = column block n of B).  Partial tile of r < 8 row blocks: SIMD pair (w & 3) owns column block w & 3 of EVERY
    // row block, its two waves the even / odd ones (x = that B block, y[q] = A of row block half + 2q) — r units per
    // SIMD instead of 8, so the tile costs r/8 of a full one.
    auto load_frags = [&](auto rem_tag, const unsigned char *st, int j) __attribute__((always_inline)) -> PgFrag {
        constexpr bool REM = decltype(rem_tag)::value;
        PgFrag f;
        if constexpr (!REM) {
            f.x = *reinterpret_cast<const float4 *>(st + (32 * wave + li) * 128 + 16 * ((2 * j + hh) ^ swz));
#pragma unroll
            for (int n = 0; n < 4; n++) f.y[n] = *reinterpret_cast<const float4 *>(st + PG_A_BYTES + li * 32 + bsw + j * 4096 + n * 1024);
        } else {
            f.x = *reinterpret_cast<const float4 *>(st + PG_A_BYTES + (32 * cb + li) * 32 + bsw + j * 4096);
#pragma unroll
            for (int q = 0; q < 4; q++)     // row blocks past the tile's end: read block 7 of the stage (valid memory), never multiplied
                f.y[q] = *reinterpret_cast<const float4 *>(st + (32 * min(half + 2 * q, 7) + li) * 128 + 16 * ((2 * j + hh) ^ swz));
        }
        return f;
    };
    struct Wins { uint32_t w[4]; };
    auto load_wins = [&](auto rem_tag, const unsigned char *st, int tile_row0, int k0) __attribute__((always_inline)) -> Wins {
        constexpr bool REM = decltype(rem_tag)::value;
        Wins W = {{0, 0, 0, 0}};
        if (DROP) {
            if constexpr (!REM) W.w[0] = window(st, tile_row0, 32 * wave + li, k0);
            else {
#pragma unroll
                for (int q = 0; q < 4; q++) W.w[q] = window(st, tile_row0, 32 * min(half + 2 * q, 7) + li, k0);
            }
        }
        return W;
    };
    // the 16 MFMAs of group j; `after(k)` runs behind MFMA k (k = 0 .. 15), pinned there
    auto mfma_group = [&](auto rem_tag, const PgFrag &f, const Wins &W, int j, int nq, auto after) __attribute__((always_inline)) {
        constexpr bool REM = decltype(rem_tag)::value;
        if constexpr (!REM) {
            float at[4] = {f.x.x, f.x.y, f.x.z, f.x.w};
            if (DROP) {
#pragma unroll
                for (int t = 0; t < 4; t++) at[t] = keepv(at[t], W.w[0], j, t);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    const float b = t == 0 ? f.y[n].x : (t == 1 ? f.y[n].y : (t == 2 ? f.y[n].z : f.y[n].w));
                    acc[n] = MFMA32(at[t], b, acc[n]);
                    after(4 * t + n);
                }
            }
        } else {
            float at[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                at[q][0] = f.y[q].x; at[q][1] = f.y[q].y; at[q][2] = f.y[q].z; at[q][3] = f.y[q].w;
                if (DROP) {
#pragma unroll
                    for (int t = 0; t < 4; t++) at[q][t] = keepv(at[q][t], W.w[q], j, t);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float b = t == 0 ? f.x.x : (t == 1 ? f.x.y : (t == 2 ? f.x.z : f.x.w));
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (q < nq) acc[q] = MFMA32(at[q][t], b, acc[q]);     // wave-uniform
                    after(4 * t + q);
                }
            }
        }
    };
    auto nothing = [](int) {};

    // results leave through buffer stores: a row past m is past the descriptor's range and dropped by the hardware, so
    // every wave issues the same number of store instructions (the vmcnt arithmetic below relies on that)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((uint32_t)a.m * (uint32_t)a.ldo * 4u), 0x00020000);
    auto store_block = [&](f32x16 &v, int row0, int col) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float x = (a.relu && !(v[r] > 0.f)) ? 0.f : v[r];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), orsrc, (int)(((uint32_t)row * (uint32_t)a.ldo + (uint32_t)col) * 4u), 0, 0);
            v[r] = 0.f;
        }
    };

    // one item.  `first`: the group-0 fragments and keep bits are not in registers yet (first item of a tile shape);
    // `more`: another item follows (its chunk must be waited for); `fetch`: the chunk after next exists and is issued here
    PgFrag f0, f1;
    Wins W;
    bool stores_behind = false;                                  // a tile's 64 stores were issued since the last wait
    auto item = [&](auto rem_tag, int stage, int next_stage, int tile_row0, int k0, int nq, bool first, bool more, bool same_shape_next,
                    int next_tile_row0, int next_k0, int ft, int fc, int fstage, bool fetch) __attribute__((always_inline)) {
        const unsigned char *st = smem + stage * PG_STAGE_BYTES;
        if (first) { W = load_wins(rem_tag, st, tile_row0, k0); f0 = load_frags(rem_tag, st, 0); }
        f1 = load_frags(rem_tag, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(rem_tag, f0, W, 0, nq, nothing);
        __builtin_amdgcn_sched_barrier(0);
        f0 = load_frags(rem_tag, st, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(rem_tag, f1, W, 1, nq, nothing);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            // the next chunk's pieces are this wave's only outstanding DMA — and, right after a tile's epilogue, 64 younger
            // stores.  (Rounds 3-4 waited vmcnt(63) there, "63 forces the pieces and lets the stores fly": that assumes stores
            // complete behind older loads.  They need not — round 5 measured LDS-DMA pieces completing ahead of older register
            // loads, dense_bf16x3.h — so the wait is for everything: four store drains per launch.)
            PG_WAIT_BARRIER(0);
            stores_behind = false;
        }
        f1 = load_frags(rem_tag, st, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(rem_tag, f0, W, 2, nq, [&](int k) __attribute__((always_inline)) {
            if (k < NL) {
                __builtin_amdgcn_sched_barrier(0);
                if (fetch) issue_piece(ft, fc, fstage, k);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(rem_tag, f1, W, 3, nq, nothing);
        __builtin_amdgcn_sched_barrier(0);
        if (more && same_shape_next) {
            const unsigned char *sn = smem + next_stage * PG_STAGE_BYTES;
            W = load_wins(rem_tag, sn, next_tile_row0, next_k0);
            f0 = load_frags(rem_tag, sn, 0);
        }
    };

    // ---- prologue: chunks 0 and 1 in flight, chunk 0 landed
    {
#pragma unroll
        for (int u = 0; u < NL; u++) issue_piece(0, 0, 0, u);
        if (n_items > 1) {
#pragma unroll
            for (int u = 0; u < NL; u++) issue_piece(a.n_chunks > 1 ? 0 : 1, a.n_chunks > 1 ? 1 : 0, 1, u);
            if (DROP) PG_WAIT_BARRIER(7);
            else PG_WAIT_BARRIER(6);
        } else {
            PG_WAIT_BARRIER(0);
        }
    }
    const int n_full = nb >> 3;                                  // full tiles first, then (at most) one partial tile
    const int n_full_items = n_full * a.n_chunks;
    int t = 0, c = 0, stage = 0;
    int lt = 0, lc = 0;                                          // load cursor: item g + 2
    auto advance = [&](int &tt, int &cc) __attribute__((always_inline)) { if (++cc == a.n_chunks) { cc = 0; ++tt; } };
    advance(lt, lc); advance(lt, lc);
    for (int g = 0; g < n_full_items; g++) {
        const int tile_row0 = (rb_lo + 8 * t) * 32;
        int nt = t, nc = c;
        advance(nt, nc);
        const int next_stage = stage == 2 ? 0 : stage + 1, fetch_stage = stage == 0 ? 2 : stage - 1;
        item(PgTag<false>(), stage, next_stage, tile_row0, c * PG_BK, 4, g == 0, g + 1 < n_items, g + 1 < n_full_items,
             (rb_lo + 8 * nt) * 32, nc * PG_BK, lt, lc, fetch_stage, g + 2 < n_items);
        advance(lt, lc);
        if (c == a.n_chunks - 1) {
#pragma unroll
            for (int n = 0; n < 4; n++) store_block(acc[n], tile_row0 + 32 * wave, 32 * n + li);
            stores_behind = true;
        }
        t = nt; c = nc;
        stage = next_stage;
    }
    if (n_full_items < n_items) {
        const int r = nb - 8 * n_full;                           // 1 .. 7 row blocks
        const int nq = max(0, (r - half + 1) >> 1);
        const int tile_row0 = (rb_lo + 8 * n_full) * 32;
        for (int g = n_full_items; g < n_items; g++) {
            const int next_stage = stage == 2 ? 0 : stage + 1, fetch_stage = stage == 0 ? 2 : stage - 1;
            item(PgTag<true>(), stage, next_stage, tile_row0, c * PG_BK, nq, g == n_full_items, g + 1 < n_items, g + 1 < n_items,
                 tile_row0, (c + 1) * PG_BK, lt, lc, fetch_stage, g + 2 < n_items);
            advance(lt, lc);
            c++;
            stage = next_stage;
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q < nq) store_block(acc[q], tile_row0 + 32 * (half + 2 * q), 32 * cb + li);
    }
}

#ifdef GCNHIP_EXPERIMENTS   // the persistent weight gradient: built, correct, measured slower
// ------------------------------------------------------------------------------------------------ backward
// dW[K x 128] = X~^T . dH0 over this workgroup's contiguous share of the m rows (the reduction dimension), ALL of the
// output held in the accumulators of its eight waves: ceil(K/32) x 4 blocks of 32 x 32, block b -> wave b % 8 (so a wave
// owns ONE column block of dH0 and every other 32-column block of X: one B read and up to ten A reads per k step, and
// the two waves of a SIMD carry 10 + 9 blocks whatever K is — balanced to the block).  One slab per workgroup, summed in
// workgroup order by slab_reduce_kernel (no atomics: the same bits every run).
// A chunk is 16 rows: the LDS images are straight copies of 16 rows of X (row stride ldx, zero padded past K) and of dH0,
// both k-major exactly as the MFMA operands want them (a half-wave reads 32 consecutive floats of one row).
constexpr int PB_ROWS = 16;
constexpr int PB_MAX_LDX = 640;
constexpr int PB_A_BYTES = PB_ROWS * PB_MAX_LDX * 4;        // 40960
constexpr int PB_B_BYTES = PB_ROWS * 128 * 4;               // 8192
constexpr int PB_K_WORDS = 24;                              // keep-bit words per row: ceil(ldx/32) + 1 <= 21, padded
constexpr int PB_K_BYTES = PB_ROWS * PB_K_WORDS * 4;        // 1536
constexpr int PB_STAGE_BYTES = PB_A_BYTES + PB_B_BYTES + PB_K_BYTES;   // 50688; x3 = 152064
constexpr int PB_NACC = 10;

struct PersistBwdArgs {
    const float *x; int ldx;          // X, 16-byte aligned rows, ldx % 16 == 0, ldx <= 640, ldx >= round_up(K, 32), zero padded
    const float *dout; int ldd;       // dH0 [m x 128], ldd % 4 == 0
    float *slab; int lds;             // [gridDim.x][K][lds]
    int m, K, n_xb;                   // n_xb = ceil(K / 32) <= 20
    int rows_per_wg;                  // multiple of 2
    const uint32_t *bits;
    float scale;
};

template <bool DROP>
__global__ __launch_bounds__(512, 2) void dense_bwd_persist_kernel(PersistBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PG_STAGES * PB_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kq = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    if (a.m < 0) smem[tid] = 0;       // never taken (see the forward kernel)

    const int r_lo = blockIdx.x * a.rows_per_wg, r_hi = min(a.m, r_lo + a.rows_per_wg);
    const int pb = wave & 3, xb0 = wave >> 2;                // blocks (xb0 + 2n, pb), n < nacc
    const int nacc = (a.n_xb - xb0 + 1) >> 1;
    float *slab = a.slab + (size_t)blockIdx.x * a.K * a.lds;
    f32x16 acc[PB_NACC];
#pragma unroll
    for (int n = 0; n < PB_NACC; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    const int n_items = r_lo < r_hi ? (r_hi - r_lo + PB_ROWS - 1) / PB_ROWS : 0;
    const int a_bytes = PB_ROWS * a.ldx * 4;                 // bytes of the A image actually used (multiple of 1024)
    const int a_instr = a_bytes >> 10;                       // <= 40
    const int kwords = (a.ldx >> 5) + 1;                     // keep-bit words staged per row
    constexpr int NL = DROP ? 7 : 6;

    // per-lane source coordinates of this wave's five A pieces (piece q = wave + 8u, clamped: a duplicate piece rewrites
    // the same bytes), fixed for the whole launch
    int a_row[5], a_col[5], a_q[5];
#pragma unroll
    for (int u = 0; u < 5; u++) {
        const int q = min(wave + 8 * u, a_instr - 1);
        const int off = q * 1024 + lane * 16;
        a_q[u] = q; a_row[u] = off / (a.ldx * 4); a_col[u] = (off % (a.ldx * 4)) >> 2;
    }
    auto issue = [&](int item, int stage) {
        const int r0 = r_lo + item * PB_ROWS;
        const uint32_t sA = lds0 + stage * PB_STAGE_BYTES, sB = sA + PB_A_BYTES, sK = sB + PB_B_BYTES;
#pragma unroll
        for (int u = 0; u < 5; u++)
            pg_glds16(a.x + (size_t)min(r0 + a_row[u], a.m - 1) * a.ldx + a_col[u], sA + a_q[u] * 1024);
        pg_glds16(a.dout + (size_t)min(r0 + 2 * wave + (lane >> 5), a.m - 1) * a.ldd + (lane & 31) * 4, sB + wave * 1024);
        if (DROP) {
            // word w of row rr: bits [32w, 32w+32) counted from the word that holds the row's first element
            const int total = PB_ROWS * kwords;
            const int q = min(wave, (total - 1) >> 6);
            const int d = min(q * 64 + lane, total - 1);
            const int rr = d / kwords, w = d - rr * kwords;
            const uint64_t e0 = (uint64_t)min(r0 + rr, a.m - 1) * a.K;
            // LDS-DMA lands at base + 4 * lane: the image is [rr][kwords] densely packed, lane order = d order (d clamped
            // only in the last instruction's tail, whose lanes then rewrite the final word)
            pg_glds4(a.bits + (e0 >> 5) + w, sK + q * 256);
        }
    };

    if (n_items > 0) issue(0, 0);
    if (n_items > 1) issue(1, 1);
    int stage = 0, lstage = 2;
    const uint32_t scale_bits = __float_as_uint(a.scale);
    for (int g = 0; g < n_items; g++) {
        if (g + 1 < n_items) {
            if (DROP) PG_WAIT_BARRIER(7);
            else PG_WAIT_BARRIER(6);
        } else {
            PG_WAIT_BARRIER(0);
        }
        if (g + 2 < n_items) { issue(g + 2, lstage); lstage = lstage == 2 ? 0 : lstage + 1; }
        const unsigned char *st = smem + stage * PB_STAGE_BYTES;
        const int r0 = r_lo + g * PB_ROWS;
        const unsigned char *Ap = st + kq * (a.ldx * 4) + (xb0 * 32 + li) * 4;
        const unsigned char *Bp = st + PB_A_BYTES + kq * 512 + (pb * 32 + li) * 4;
        const unsigned char *Kp = st + PB_A_BYTES + PB_B_BYTES + kq * (kwords * 4) + xb0 * 4;
#pragma unroll 2
        for (int s = 0; s < PB_ROWS / 2; s++) {
            const int row = r0 + 2 * s + kq;
            const float bval = *reinterpret_cast<const float *>(Bp + s * 1024);
            const float live = row < r_hi ? 1.f : 0.f;
            const uint32_t sh = (uint32_t)(((uint64_t)row * a.K) & 31);
            float av[PB_NACC];
#pragma unroll
            for (int n = 0; n < PB_NACC; n++) {
                av[n] = 0.f;
                if (n < nacc) {                              // wave-uniform
                    const float x = *reinterpret_cast<const float *>(Ap + s * 2 * (a.ldx * 4) + n * 256);
                    float f = live;
                    if (DROP) {
                        const uint32_t *kw = reinterpret_cast<const uint32_t *>(Kp + s * 2 * (kwords * 4) + n * 8);
                        const uint32_t m32 = __builtin_amdgcn_alignbit(kw[1], kw[0], sh);
                        const uint32_t keep = (uint32_t)(((int32_t)(m32 << (31 - li))) >> 31);       // bit li -> 0 or ~0
                        f = __uint_as_float(__float_as_uint(live) == 0u ? 0u : (keep & scale_bits));
                    }
                    av[n] = x * f;
                }
            }
#pragma unroll
            for (int n = 0; n < PB_NACC; n++)
                if (n < nacc) acc[n] = MFMA32(av[n], bval, acc[n]);
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    // this workgroup's partial [K x 128] (all zero when it had no rows)
#pragma unroll
    for (int n = 0; n < PB_NACC; n++) {
        if (n < nacc) {
            const int pc = pb * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int xc = (xb0 + 2 * n) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                if (xc < a.K) slab[(size_t)xc * a.lds + pc] = acc[n][r];
            }
        }
    }
}
#endif  // GCNHIP_EXPERIMENTS

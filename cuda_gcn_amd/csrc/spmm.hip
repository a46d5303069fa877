// spmm.hip — SparseMatmul: H0 = X~ . W1 and dW1 = X~^T . dH0
// (src/seq/module.cpp:47-77; CUDA: cuda_kernel.cu:100-122, whose backward is a
// racy scatter).  X~ = X with the input dropout applied on the fly, so the
// pristine X stays resident and is never re-uploaded (the reference copies
// all of X host->device twice per epoch, cuda_gcn.cu:81-83).
//
// Two regimes, chosen once in gcnhip_feat_create:
//  * sparse X (Cora/Citeseer/Pubmed-like): HBM/L2-bound CSR row gather of W1
//    rows forward, CSC gather of dH0 rows backward (deterministic, no atomics);
//  * dense X stored as CSR (Reddit: every row has all 602 columns): an
//    LDS-tiled exact-f32 MFMA GEMM forward, split-K MFMA A^T.B backward
//    (dense_kernels.h).  Treating it as a gather would move nnz*h*4 bytes
//    (72 GB) through L2 instead of 0.56 GB from HBM.
#include "dense_kernels.h"
#include "dense_tile128.h"
#include "dense_persist.h"
#include "dense_bf16x3.h"
#include "spmm_sparse.h"

// ------------------------------------------------------------- dense forward
// out[m x p] = X~[m x K] . W[K x p].  Workgroup tile 128 rows x (NT*16) cols,
// K in chunks of 32 through LDS; wave w owns rows 32w..32w+31 (2 x NT MFMA
// accumulators).  The next chunk is fetched into registers while the current
// one is multiplied (global -> reg -> LDS, write after the barrier).
struct DenseFwdArgs {
    const float *x; int ldx;        // m x K, row stride ldx (602 for Reddit: 8-byte aligned rows)
    const float *w; int ldw;
    float *out; int ldo;
    int m, K, p;
    const uint32_t *bits;           // keep bits of the stored elements (dropbits_kernel), NULL: no dropout
    float scale;
    int relu;
};

constexpr int DF_BM = 128, DF_BK = 32;
constexpr int DF_ALD = DF_BK + 2;       // A tile row stride: 2*i + kq banks, conflict-free b32 reads

template <int NT, int VX>
__global__ __launch_bounds__(256) void spmm_dense_fwd_kernel(DenseFwdArgs a) {
    constexpr int BN = NT * 16;
    constexpr int BLD = BN + 16;        // B tile row stride: k-groups 16 banks apart
    __shared__ __attribute__((aligned(16))) float As[DF_BM * DF_ALD];
    __shared__ __attribute__((aligned(16))) float Bs[DF_BK * BLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int row_base = blockIdx.x * DF_BM;
    const int col_base = blockIdx.y * BN;

    // staging maps.  A: 128 x 32 floats = 4096; VX floats per lane per piece.
    constexpr int A_PIECES = DF_BM * DF_BK / (256 * VX);
    constexpr int LPR = DF_BK / VX;                       // lanes per A row
    float areg[A_PIECES][VX];
    uint64_t kreg[A_PIECES];        // raw keep-bit words; applied in stash() so that no ALU sits behind the loads
    // B: 32 x BN floats, 4 per lane per piece (W rows are 16-byte aligned when ldw % 4 == 0; else scalar)
    constexpr int B_PIECES = (DF_BK * BN + 1023) / 1024;
    float breg[B_PIECES][4];

    auto fetch = [&](int k0) {
#pragma unroll
        for (int pc = 0; pc < A_PIECES; pc++) {
            const int idx = pc * 256 + tid;
            const int r = idx / LPR, c = (idx % LPR) * VX;
            const int row = row_base + r, col = k0 + c;
#pragma unroll
            for (int s = 0; s < VX; s++) areg[pc][s] = 0.f;
            if (row < a.m && col < a.K) {
                load_vec<VX>(a.x + (size_t)row * a.ldx + col, a.K - col, areg[pc]);
                if (a.bits) {
                    const uint64_t w = ((uint64_t)row * a.K + col) >> 5;
                    kreg[pc] = a.bits[w];
                    if (a.K % VX != 0) kreg[pc] |= (uint64_t)a.bits[w + 1] << 32;
                }
            }
        }
#pragma unroll
        for (int pc = 0; pc < B_PIECES; pc++) {
            const int idx = (pc * 256 + tid) * 4;
            const int k = idx / BN, c = idx % BN;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int gk = k0 + k, gc = col_base + c + s;
                breg[pc][s] = (k < DF_BK && gk < a.K && gc < a.p) ? a.w[(size_t)gk * a.ldw + gc] : 0.f;
            }
        }
    };
    auto stash = [&](int cur_k0) {
#pragma unroll
        for (int pc = 0; pc < A_PIECES; pc++) {
            const int idx = pc * 256 + tid;
            const int r = idx / LPR, c = (idx % LPR) * VX;
            const uint32_t kb = a.bits ? (uint32_t)(kreg[pc] >> (uint32_t)(((uint64_t)(row_base + r) * a.K + cur_k0 + c) & 31)) : 0xFu;
#pragma unroll
            for (int s = 0; s < VX; s++)
                As[r * DF_ALD + c + s] = a.bits ? ((kb >> s & 1u) ? areg[pc][s] * a.scale : 0.f) : areg[pc][s];
        }
#pragma unroll
        for (int pc = 0; pc < B_PIECES; pc++) {
            const int idx = (pc * 256 + tid) * 4;
            const int k = idx / BN, c = idx % BN;
            if (k < DF_BK) {
#pragma unroll
                for (int s = 0; s < 4; s++) Bs[k * BLD + c + s] = breg[pc][s];
            }
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += DF_BK) {
        __syncthreads();                 // previous chunk fully consumed
        stash(k0);
        __syncthreads();
        if (k0 + DF_BK < a.K) fetch(k0 + DF_BK);
#pragma unroll
        for (int kk = 0; kk < DF_BK; kk += 4) {
            float af[2], bf[NT];
#pragma unroll
            for (int r = 0; r < 2; r++) af[r] = As[(wave * 32 + r * 16 + li) * DF_ALD + kk + kq];
#pragma unroll
            for (int t = 0; t < NT; t++) bf[t] = Bs[(kk + kq) * BLD + t * 16 + li];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int t = 0; t < NT; t++)
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r], bf[t], acc[r][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int col = col_base + t * 16 + li;
            if (col >= a.p) continue;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = row_base + wave * 32 + r * 16 + 4 * kq + i;
                if (row < a.m) a.out[(size_t)row * a.ldo + col] = (a.relu && !(acc[r][t][i] > 0.f)) ? 0.f : acc[r][t][i];
            }
        }
}

constexpr int64_t SPMM_NARROW_MIN_NNZ = 262144;     // below this the general sparse kernels run (option spmm_general = -1: narrow kernels always)
constexpr size_t SPMM_LDS_MAX_BYTES = 128 * 1024;   // W1 in LDS: Cora 92 KB and Pubmed 32 KB fit (gfx950: 160 KB per CU)

static DropSpec make_drop(float p_drop, uint64_t seed, const uint32_t *d_epoch, uint64_t off, const uint8_t *keep_mask) {
    DropSpec d;
    d.on = p_drop > 0.f || keep_mask != nullptr;
    d.thr = dropout_threshold(p_drop);
    d.scale = 1 / (1 - p_drop);
    d.seed = seed; d.off = off; d.d_epoch = d_epoch; d.keep_mask = keep_mask;
    return d;
}

// keep bits for the nnz stored elements of f (Philox or injected decisions)
// whole Philox blocks per thread (dense_tile128.h): device decisions from a stream offset that is a multiple of 128
static inline bool keep_bits_by_block(const DropSpec &d) { return !d.keep_mask && (d.off & 127) == 0; }

static int make_keep_bits(gcnhip_ctx *c, const gcnhip_feat *f, const DropSpec &d) {
    if (!f->keep_bits) return gcnhip_fail("input dropout on a feature object without a keep-bit array (gcnhip_feat_create_aggregated builds evaluation-only objects)");
    const_cast<gcnhip_feat *>(f)->keep_layout = 0;
    const int64_t words = (f->nnz + 31) / 32;
    if (keep_bits_by_block(d))
        dropbits_block_kernel<<<ceil_div((words + 3) / 4, 256), 256, 0, c->stream>>>(f->keep_bits, f->nnz, d.thr, d.seed, d.d_epoch, d.off >> 7);
    else
        dropbits_kernel<<<ceil_div(words, 256), 256, 0, c->stream>>>(f->keep_bits, f->nnz, d.thr, d.seed, d.d_epoch, d.off, d.keep_mask);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

// the same decisions chunk-major (dense X; dense_bf16x3.h): what the bf16x3 kernels read
static BxBitsArgs keep_bits_cm_args(const gcnhip_feat *f, const DropSpec &d) {
    BxBitsArgs b;
    b.bits = f->keep_bits; b.m = f->n_rows; b.K = f->n_cols; b.n_chunks = (f->n_cols + 31) / 32;
    b.R = std::max(1, std::min(128, (BX_BITS_LDS_WORDS - 12) * 32 / std::max(1, f->n_cols)));
    b.thr = d.thr; b.seed = d.seed; b.d_epoch = d.d_epoch; b.off = d.off; b.keep_mask = d.keep_mask;
    return b;
}
static inline bool keep_bits_cm_fit(const gcnhip_feat *f) { return f->dense && f->keep_bits && f->n_cols / 32 + 16 <= BX_BITS_LDS_WORDS; }
static int make_keep_bits_cm(gcnhip_ctx *c, const gcnhip_feat *f, const DropSpec &d) {
    if (!keep_bits_cm_fit(f)) return gcnhip_fail("chunk-major keep bits: not a dense feature object with a keep-bit array");
    const BxBitsArgs b = keep_bits_cm_args(f, d);
    const_cast<gcnhip_feat *>(f)->keep_layout = 1;
    dropbits_cm_kernel<<<ceil_div(b.m, b.R), 256, 0, c->stream>>>(b);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
// A consumer that did not make the decisions itself (gcnhip_spmm_bwd_part with make_decisions = 0) finds them in the layout of
// whoever did; the decisions are a pure function of (seed, epoch, offset), so the other layout is one launch away.
static int want_keep_layout(gcnhip_ctx *c, const gcnhip_feat *f, const DropSpec &d, int layout, bool fresh = false) {
    if (!d.on || (!fresh && f->keep_layout == layout)) return 0;
    return layout ? make_keep_bits_cm(c, f, d) : make_keep_bits(c, f, d);
}

static int x_vec_width(const gcnhip_feat *f, const float *vals) {
    return (f->n_cols % 4 == 0 && aligned16(vals)) ? 4 : ((f->n_cols % 2 == 0 && ((uintptr_t)vals & 7) == 0) ? 2 : 1);
}

template <int NT>
static void launch_dense_fwd(const DenseFwdArgs &a, int vx, dim3 grid, hipStream_t s) {
    if (vx == 4) spmm_dense_fwd_kernel<NT, 4><<<grid, 256, 0, s>>>(a);
    else if (vx == 2) spmm_dense_fwd_kernel<NT, 2><<<grid, 256, 0, s>>>(a);
    else spmm_dense_fwd_kernel<NT, 1><<<grid, 256, 0, s>>>(a);
}

extern "C" {

// the second layer's product in the epilogue of the evaluation forward (dense_bf16x3.h, ZOUT): Z0 = relu(X.W) . W2
struct Z0Fuse { const float *w2; int ld_w2, p2; float *z0; int ld_z0; };

static int spmm_fwd_impl(gcnhip_ctx *c, const gcnhip_feat *f, const float *vals, const float *w, int ld_w,
                         float *out, int ld_out, int p, float p_drop, uint64_t seed, const uint32_t *d_epoch,
                         uint64_t nnz_offset, const uint8_t *keep_mask, int relu, const Z0Fuse *zf = nullptr) {
    if (!c || !f || !vals || !w || (!out && !zf) || p <= 0 || ld_w < p || (out && ld_out < p)) return -1;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return -1;
    if (zf && !(f->dense && p == 128)) return GCNHIP_NOT_AVAILABLE;   // (the other forms below store `out`, which a fused call does not have)
    if (f->n_rows == 0) return 0;
    const DropSpec d = make_drop(p_drop, seed, d_epoch, nnz_offset, keep_mask);
    if (f->dense && p > 64) {                 // 128 x 128 MFMA tiles
        if (d.on && !f->keep_bits) return gcnhip_fail("input dropout on a feature object without a keep-bit array (gcnhip_feat_create_aggregated builds evaluation-only objects)");
        Tile128Args t;
        t.x = vals; t.ldx = f->n_cols; t.w = w; t.ldw = ld_w; t.out = out; t.ldo = ld_out;
        int vx = x_vec_width(f, vals);
        if (vals == f->values && f->values_pad) { t.x = f->values_pad; t.ldx = f->ld_pad; vx = 4; }   // aligned copy of the pristine X
        t.m = f->n_rows; t.K = f->n_cols; t.p = p; t.bits = d.on ? f->keep_bits : nullptr; t.scale = d.scale; t.rows_per_split = 0; t.relu = relu;
        dim3 grid(ceil_div(f->n_rows, T_BM), ceil_div(p, T_BN));
        const bool fast = vx == 4 && t.ldx % 4 == 0 && (t.K + 31) / 32 * 32 <= t.ldx && ld_w % 4 == 0 && p % T_BN == 0 && aligned16(w);
        // p = 128: the persistent LDS-DMA form (dense_persist.h) — one workgroup per CU for the whole launch, rows dealt in
        // 32-row blocks, three-stage ring.  GCNHIP_GEMM_TILES keeps the tile kernels below for A/B runs.
        const bool tiles_only = c->opt.gemm_tiles != 0;
        const int n_chunks = (t.K + PG_BK - 1) / PG_BK;
        // Option gemm_bf16x3 (round 5; 1: wherever the persistent f32 form would run, 2: also on a co-running stream): the same
        // product on the bf16 matrix pipe from three exact bf16 planes per f32 operand (dense_bf16x3.h) — f32 results inside the
        // f32 summation bound, 0.21 ms instead of 0.35 at Reddit scale.  0 keeps the exact-f32 MFMA kernels.
        const int bx = c->opt.gemm_bf16x3;
        if (zf && !(fast && p == 128 && !tiles_only && bx && !d.on && aligned16(t.x) && zf->p2 >= 1 && zf->p2 <= 64 && zf->ld_z0 % 4 == 0 &&
                    zf->ld_z0 >= zf->p2 && zf->ld_w2 >= zf->p2 && aligned16(zf->z0) && (size_t)n_chunks * 2 * BX_BH_BYTES + BX_W2_BYTES <= c->wpack_bytes))
            return GCNHIP_NOT_AVAILABLE;                          // the caller runs the two products one after the other
        if (fast && p == 128 && !tiles_only && bx && (!d.on || keep_bits_cm_fit(f)) && (zf || ((bx >= 2 || !c->corun) && aligned16(out) &&
            (uint64_t)(t.m + 256) * (uint64_t)ld_out * 4u < (1ull << 32))) && aligned16(t.x) && (size_t)n_chunks * 2 * BX_BH_BYTES <= c->wpack_bytes) {
            const int n_hs = 2 * n_chunks;
            uint4 *wp3 = reinterpret_cast<uint4 *>(c->wpack);
            uint4 *w2img = wp3 + (size_t)n_hs * (BX_BH_BYTES / 16);
            if (d.on) {                                  // keep words (chunk-major) and the packed planes of W from one launch
                const BxBitsArgs kb = keep_bits_cm_args(f, d);
                const int n_bits_wgs = ceil_div(kb.m, kb.R);
                const_cast<gcnhip_feat *>(f)->keep_layout = 1;
                dropbits_bx_pack_w_kernel<<<n_bits_wgs + n_hs, 256, 0, c->stream>>>(kb, n_bits_wgs, w, ld_w, t.K, n_hs, wp3, t.scale);
            } else {
                if (zf) bx_pack_w_kernel<<<n_hs + 4, 256, 0, c->stream>>>(w, ld_w, t.K, n_hs, wp3, 1.f, zf->w2, zf->ld_w2, zf->p2, w2img);
                else bx_pack_w_kernel<<<n_hs, 256, 0, c->stream>>>(w, ld_w, t.K, n_hs, wp3, t.bits ? t.scale : 1.f);
            }
            GCNHIP_LAUNCH_CHECK();
            Bx3FwdArgs ba;
            ba.x = t.x; ba.ldx = t.ldx; ba.wp = wp3; ba.out = out; ba.ldo = ld_out;
            ba.m = t.m; ba.K = t.K; ba.n_chunks = n_chunks; ba.n_rb = ceil_div(t.m, 32);
            ba.bits = t.bits; ba.relu = relu;
            ba.w2p = nullptr; ba.z0 = nullptr; ba.ldz = 0; ba.p2 = 0;
            int wgs = std::max(1, std::min(c->n_cu, ba.n_rb));
            if (zf) {
                ba.w2p = w2img; ba.z0 = zf->z0; ba.ldz = zf->ld_z0; ba.p2 = zf->p2;
                dense_fwd_bf16x3_kernel<false, 6, 0, 8, true><<<wgs, 512, 0, c->stream>>>(ba);      // (144 KB of static LDS: one workgroup per CU)
                GCNHIP_LAUNCH_CHECK();
                return 0;
            }
            if (c->corun && c->opt.gemm_lane_wgs > 0) wgs = std::max(1, std::min(wgs, c->opt.gemm_lane_wgs));   // fewer CUs host the lane's product
            // option gemm_lane_waves = 4: a context that runs beside another stream's kernels takes the four-wave form (half a CU's
            // registers: co-resident with a gather-bound kernel's waves).  Measured (docs/NOTEBOOK_r5.md §7): the lane's product is
            // then truly concurrent with the training pass's aggregation, which loses half its resident waves on those CUs —
            // 347.5 (256 CUs), 351.5 (128), 340 (64) epochs/s against 351.0 for the eight-wave form: not the default
            const bool four = c->corun && c->opt.gemm_lane_waves == 4;
            if (ba.bits) { if (four) dense_fwd_bf16x3_kernel<true, 6, 0, 4><<<wgs, 256, 0, c->stream>>>(ba); else dense_fwd_bf16x3_kernel<true, 6><<<wgs, 512, 0, c->stream>>>(ba); }
            else { if (four) dense_fwd_bf16x3_kernel<false, 6, 0, 4><<<wgs, 256, 0, c->stream>>>(ba); else dense_fwd_bf16x3_kernel<false, 6><<<wgs, 512, 0, c->stream>>>(ba); }
            GCNHIP_LAUNCH_CHECK();
            return 0;
        }
        if (fast && p == 128 && !tiles_only && !c->corun && aligned16(t.x) && aligned16(out) &&
            (uint64_t)(t.m + PG_ROWS) * (uint64_t)ld_out * 4u < (1ull << 32) && (size_t)n_chunks * 4096 * sizeof(float) <= c->wpack_bytes) {
            if (d.on && keep_bits_by_block(d)) {         // keep bits and the packed W from one launch
                const int n_bits_wgs = (int)ceil_div(((f->nnz + 31) / 32 + 3) / 4, (int64_t)256);
                const_cast<gcnhip_feat *>(f)->keep_layout = 0;
                dropbits_pack_w_kernel<<<n_bits_wgs + n_chunks * 4, 256, 0, c->stream>>>(f->keep_bits, f->nnz, d.thr, d.seed, d.d_epoch, d.off >> 7, n_bits_wgs,
                                                                                       w, ld_w, t.K, n_chunks * 4, c->wpack, t.scale);
            } else {
                if (d.on) { const int rc = make_keep_bits(c, f, d); if (rc) return rc; }
                pg_pack_w_kernel<<<ceil_div(n_chunks * 4 * 256, 256), 256, 0, c->stream>>>(w, ld_w, t.K, n_chunks * 4, c->wpack, t.bits ? t.scale : 1.f);
            }
            GCNHIP_LAUNCH_CHECK();
            PersistFwdArgs pa;
            pa.x = t.x; pa.ldx = t.ldx; pa.wp = c->wpack; pa.out = out; pa.ldo = ld_out;
            pa.m = t.m; pa.K = t.K; pa.n_chunks = n_chunks; pa.n_rb = ceil_div(t.m, 32);
            pa.bits = t.bits; pa.relu = relu;
#ifdef GCNHIP_EXPERIMENTS
            pa.dbg_linear = c->opt.dbg_linear ? 1 : 0;
#else
            pa.dbg_linear = 0;
#endif
            const int wgs = std::max(1, std::min(c->n_cu, pa.n_rb));
            if (pa.bits) dense_fwd_persist_kernel<true><<<wgs, 512, 0, c->stream>>>(pa);
            else dense_fwd_persist_kernel<false><<<wgs, 512, 0, c->stream>>>(pa);
            GCNHIP_LAUNCH_CHECK();
            return 0;
        }
        if (d.on) { const int rc = make_keep_bits(c, f, d); if (rc) return rc; }
        // eight waves per tile: same bits, 4 % faster than the four-wave form (0.383 -> 0.367 ms at Reddit scale);
        // GCNHIP_GEMM_W4 selects the four-wave kernel for A/B runs
        const bool w4 = c->opt.gemm_w4 != 0;
        if (fast && !w4) dense_fwd_t128w8_kernel<<<grid, 512, 0, c->stream>>>(t);
        else if (fast) dense_fwd_t128_kernel<4, true><<<grid, 256, 0, c->stream>>>(t);
        else if (vx == 4) dense_fwd_t128_kernel<4><<<grid, 256, 0, c->stream>>>(t);
        else if (vx == 2) dense_fwd_t128_kernel<2><<<grid, 256, 0, c->stream>>>(t);
        else dense_fwd_t128_kernel<1><<<grid, 256, 0, c->stream>>>(t);
        GCNHIP_LAUNCH_CHECK();
        return 0;
    }
    if (f->dense) {
        if (d.on) { const int rc = make_keep_bits(c, f, d); if (rc) return rc; }
        DenseFwdArgs a;
        a.x = vals; a.ldx = f->n_cols; a.w = w; a.ldw = ld_w; a.out = out; a.ldo = ld_out;
        a.m = f->n_rows; a.K = f->n_cols; a.p = p; a.bits = d.on ? f->keep_bits : nullptr; a.scale = d.scale; a.relu = relu;
        int vx = x_vec_width(f, vals);
        if (vals == f->values && f->values_pad) { a.x = f->values_pad; a.ldx = f->ld_pad; vx = 4; }
        const int nt_total = ceil_div(p, 16);
        const int NT = nt_total >= 8 ? 8 : (nt_total > 4 ? 8 : (nt_total > 2 ? 4 : nt_total));
        dim3 grid(ceil_div(f->n_rows, DF_BM), ceil_div(nt_total, NT));
        switch (NT) {
            case 1: launch_dense_fwd<1>(a, vx, grid, c->stream); break;
            case 2: launch_dense_fwd<2>(a, vx, grid, c->stream); break;
            case 4: launch_dense_fwd<4>(a, vx, grid, c->stream); break;
            default: launch_dense_fwd<8>(a, vx, grid, c->stream); break;
        }
        GCNHIP_LAUNCH_CHECK();
        return 0;
    }
    SpFwdArgs a;
    a.indptr = f->indptr; a.indices = f->indices; a.vals = vals; a.w = w; a.out = out;
    a.n_rows = f->n_rows; a.ld_w = ld_w; a.ld_out = ld_out; a.p = p; a.d = d; a.relu = relu;
    a.w_floats = f->n_cols * ld_w;
    a.rows_per_wave = 1;
    const bool vec = ld_w % 4 == 0 && aligned16(w);
    const int units = vec ? (p + 3) / 4 : p;              // lanes needed for one row of W
    // W staged in LDS (one 1024-thread workgroup per CU walks rows; north_star's "LDS staging of the feature tile"): built,
    // bit-identical to the general kernel, and measured SLOWER wherever W fits (round 4, profiles/r04_spmm_lds.log: pubmed-syn
    // 11.7 vs 9.0 us; 2 M rows x 50 values, F = 500 / 2000, h = 16: 0.91 vs 0.72 / 0.87 ms) — a W that fits LDS also sits in
    // every XCD's L2, and the rows were never what bound these kernels (spmm_sparse.h).  Opt-in: context option spmm_lds = 1.
    const size_t w_bytes = (size_t)a.w_floats * sizeof(float);
    const int wgs = std::max(1, std::min(c->n_cu, ceil_div(f->n_rows, 16)));
#ifdef GCNHIP_EXPERIMENTS
    const bool lds_form = c->opt.spmm_lds == 1 && vec && units <= 64 && p <= 256 && w_bytes <= SPMM_LDS_MAX_BYTES && f->n_rows > 0;
#else
    const bool lds_form = false;
    (void)w_bytes; (void)wgs;
#endif
    a.nnz_bytes = (int)std::min<int64_t>(f->nnz * 4, 0x7FFFFFFF);
#ifdef GCNHIP_EXPERIMENTS
    if (lds_form && units <= 16 && f->nnz * 4 < 0x7FFFFFFF && !c->opt.spmm_general) {      // narrow rows: the shuffle-free kernel over LDS
        const int per_cu = w_bytes <= 32 * 1024 ? 2 : 1;                                    // 1024-thread workgroups: at most two per CU
        const int wq = std::max(1, std::min(c->n_cu * per_cu, ceil_div(f->n_rows, 16)));
        a.rows_per_wave = std::max(1, std::min(8, (int)(f->n_rows / ((int64_t)wq * 16 * 4))));
#define SPLQ(L_)                                                                                                         \
    do {                                                                                                                 \
        auto kern = spmm_csr_fwd_q_lds_kernel<L_>;                                                                       \
        if (w_bytes > 64 * 1024) GCNHIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w_bytes)); \
        kern<<<wq, 1024, w_bytes, c->stream>>>(a);                                                                       \
    } while (0)
        if (units <= 1) SPLQ(1); else if (units <= 2) SPLQ(2); else if (units <= 4) SPLQ(4); else if (units <= 8) SPLQ(8); else SPLQ(16);
#undef SPLQ
        GCNHIP_LAUNCH_CHECK();
        return 0;
    }
    if (lds_form) {
#define SPL(L_)                                                                                                          \
    do {                                                                                                                 \
        auto kern = spmm_csr_fwd_lds_kernel<L_>;                                                                         \
        if (w_bytes > 64 * 1024) GCNHIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w_bytes)); \
        kern<<<wgs, 1024, w_bytes, c->stream>>>(a);                                                                      \
    } while (0)
        if (units <= 1) SPL(1); else if (units <= 2) SPL(2); else if (units <= 4) SPL(4);
        else if (units <= 8) SPL(8); else if (units <= 16) SPL(16); else if (units <= 32) SPL(32); else SPL(64);
#undef SPL
        GCNHIP_LAUNCH_CHECK();
        return 0;
    }
#endif
    // rows per wave: enough waves to fill every slot about twice, at most 8 rows each (one wave per row on small inputs)
    a.rows_per_wave = std::max(1, std::min(8, (int)(f->n_rows / ((int64_t)c->n_cu * 64))));
    if (c->opt.spmm_rows > 0) a.rows_per_wave = std::min(32, c->opt.spmm_rows);
    a.n_slices = 1;
    // a W past an XCD's L2 (4 MiB), rows of whole 32-float slices: the XCD-sliced form (spmm_sparse.h; option spmm_slices = 0: off)
    if (vec && c->opt.spmm_slices != 0 && p % 32 == 0 && (p / 32 == 2 || p / 32 == 4 || p / 32 == 8) &&
        (size_t)f->n_cols * ld_w * sizeof(float) > ((size_t)4 << 20) && f->n_rows >= 8 * 64) {
        a.n_slices = p / 32;
        if (c->opt.spmm_rows <= 0) a.rows_per_wave = 2;            // measured, 2 M rows of 50 values, h = 128: 4.58 / 4.00 / 4.32 / 4.41 / 4.52 ms at 1 / 2 / 4 / 8 / 16
        const int Gs = 8 / a.n_slices;
        const int units_total = ceil_div(f->n_rows, a.rows_per_wave);
        const int per_group = ceil_div(units_total, Gs) + 1;       // (+1: the shares differ by at most one unit)
        dim3 sgrid(8 * ceil_div(per_group, 4), 1);
        a.nnz_bytes = (int)std::min<int64_t>(f->nnz * 4, 0x7FFFFFFF);
        spmm_csr_fwd_kernel<8, true, true><<<sgrid, 256, 0, c->stream>>>(a);
        GCNHIP_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(ceil_div(ceil_div(f->n_rows, a.rows_per_wave), 4), 1);
    a.nnz_bytes = (int)std::min<int64_t>(f->nnz * 4, 0x7FFFFFFF);
    // narrow 16-byte aligned rows (hidden <= 64; the reference's default is 16): the shuffle-free kernel (spmm_sparse.h)
    // (small inputs keep the general kernel: at a few microseconds per launch the shorter program wins — cora-syn / citeseer-syn
    //  3.3 / 3.5 us against 3.3 / 4.4; pubmed-syn and up: 7.7 against 8.9 us, 2 M rows: 0.81 against 0.95 ms)
    const bool narrow = vec && units <= 16 && f->nnz * 4 < 0x7FFFFFFF && !c->opt.spmm_general && (f->nnz >= SPMM_NARROW_MIN_NNZ || c->opt.spmm_general < 0);
    if (narrow) {
        if (units <= 1) spmm_csr_fwd_q_kernel<1><<<grid, 256, 0, c->stream>>>(a);
        else if (units <= 2) spmm_csr_fwd_q_kernel<2><<<grid, 256, 0, c->stream>>>(a);
        else if (units <= 4) spmm_csr_fwd_q_kernel<4><<<grid, 256, 0, c->stream>>>(a);
        else if (units <= 8) spmm_csr_fwd_q_kernel<8><<<grid, 256, 0, c->stream>>>(a);
        else spmm_csr_fwd_q_kernel<16><<<grid, 256, 0, c->stream>>>(a);
        GCNHIP_LAUNCH_CHECK();
        return 0;
    }
#define SPF(L_)                                                                           \
    do {                                                                                  \
        if (vec) spmm_csr_fwd_kernel<L_, true><<<grid, 256, 0, c->stream>>>(a);           \
        else spmm_csr_fwd_kernel<L_, false><<<grid, 256, 0, c->stream>>>(a);              \
    } while (0)
    if (units <= 1) SPF(1); else if (units <= 2) SPF(2); else if (units <= 4) SPF(4);
    else if (units <= 8) SPF(8); else if (units <= 16) SPF(16); else if (units <= 32) SPF(32); else SPF(64);
#undef SPF
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

int gcnhip_spmm_fwd(gcnhip_ctx *c, const gcnhip_feat *f, const float *vals, const float *w, int ld_w,
                    float *out, int ld_out, int p, float p_drop, uint64_t seed, const uint32_t *d_epoch,
                    uint64_t nnz_offset, const uint8_t *keep_mask) {
    return spmm_fwd_impl(c, f, vals, w, ld_w, out, ld_out, p, p_drop, seed, d_epoch, nnz_offset, keep_mask, 0);
}

int gcnhip_spmm_fwd_relu(gcnhip_ctx *c, const gcnhip_feat *f, const float *vals, const float *w, int ld_w,
                         float *out, int ld_out, int p) {
    return spmm_fwd_impl(c, f, vals, w, ld_w, out, ld_out, p, 0.f, 0, nullptr, 0, nullptr, 1);
}

int gcnhip_spmm_fwd_relu_matmul(gcnhip_ctx *c, const gcnhip_feat *f, const float *vals, const float *w, int ld_w, int p,
                                const float *w2, int ld_w2, int p2, float *z0, int ld_z0) {
    if (!w2 || !z0 || p2 <= 0) return -1;
    const Z0Fuse zf = {w2, ld_w2, p2, z0, ld_z0};
    return spmm_fwd_impl(c, f, vals, w, ld_w, nullptr, 0, p, 0.f, 0, nullptr, 0, nullptr, 1, &zf);
}

// the split-K plan of the dense weight gradient: S row ranges of rps rows (a multiple of the K chunk) fill the chip twice
static bool dense_bwd_plan(const gcnhip_ctx *c, const gcnhip_feat *f, int p, int *rps_out, int *S_out) {
    if (!(f->dense && p > 64 && f->n_cols >= 64)) return false;
    const int kt = ceil_div(f->n_cols, 128), pt = ceil_div(p, 128);
    int S = (2 * c->n_cu) / (kt * pt);
    if (S < 1) S = 1;
    int rps = (ceil_div(f->n_rows, S) + T_BK - 1) / T_BK * T_BK;
    if (rps < T_BK) rps = T_BK;
    *rps_out = rps;
    *S_out = ceil_div(f->n_rows, rps);
    return true;
}

// The whole product by the persistent form (dense_persist.h): one workgroup per CU, each with a contiguous share of the rows
// and ALL of dW in its accumulators; one slab per workgroup.  Returns 1 when the shape is not its (then the split tiles run).
#ifndef GCNHIP_EXPERIMENTS
static int dense_bwd_persist(gcnhip_ctx *, const gcnhip_feat *, const float *, const float *, int, int, const DropSpec &) { return 1; }
#else
static int dense_bwd_persist(gcnhip_ctx *c, const gcnhip_feat *f, const float *vals, const float *dout, int ld_dout, int p, const DropSpec &d) {
    // EXPERIMENT, opt-in (GCNHIP_GEMM_PERSIST_BWD): measured SLOWER than the split tiles at Reddit scale (0.50 ms without,
    // 0.61 ms with dropout against 0.387 ms; profiles/r03_gemm_pmc.json: the pipes 41-53 % busy, 37 % of the wave cycles
    // parked) — ten k-major A reads + ten keep-word pairs per k step and wave make the VALU/LDS side as long as the MFMAs
    const bool persist_bwd = c->opt.gemm_persist_bwd != 0;
    if (!persist_bwd || p != 128 || f->n_rows < 1) return 1;
    const float *x = vals; int ldx = f->n_cols;
    if (vals == f->values && f->values_pad) { x = f->values_pad; ldx = f->ld_pad; }
    const int n_xb = ceil_div(f->n_cols, 32);
    if (ldx % 16 != 0 || ldx > PB_MAX_LDX || n_xb * 32 > ldx || n_xb > 2 * PB_NACC || !aligned16(x) || ld_dout % 4 != 0 || !aligned16(dout)) return 1;
    const int G = std::max(1, std::min(c->n_cu, ceil_div(f->n_rows, PB_ROWS)));
    const int p_ld = 128;
    const int rc = ensure_slab(c, (size_t)G * f->n_cols * p_ld * sizeof(float));
    if (rc) return rc;
    PersistBwdArgs a;
    a.x = x; a.ldx = ldx; a.dout = dout; a.ldd = ld_dout; a.slab = c->slab; a.lds = p_ld;
    a.m = f->n_rows; a.K = f->n_cols; a.n_xb = n_xb;
    a.rows_per_wg = ceil_div(f->n_rows, G);
    a.bits = d.on ? f->keep_bits : nullptr; a.scale = d.scale;
    if (a.bits) dense_bwd_persist_kernel<true><<<G, 512, 0, c->stream>>>(a);
    else dense_bwd_persist_kernel<false><<<G, 512, 0, c->stream>>>(a);
    GCNHIP_LAUNCH_CHECK();
    c->slab_n = G;
    return 0;
}
#endif

// splits [s0, s1) of the plan into their slabs
// fresh: the keep decisions of this call have not been made yet (else: whoever made them left them in f->keep_bits)
static int dense_bwd_part(gcnhip_ctx *c, const gcnhip_feat *f, const float *vals, const float *dout, int ld_dout, int p,
                          const DropSpec &d, int s0, int s1, bool fresh) {
    int rps, S;
    if (!dense_bwd_plan(c, f, p, &rps, &S) || s0 < 0 || s1 > S || s0 > s1) return -1;
    if (s0 == s1) return 0;
    if (d.on && !f->keep_bits) return gcnhip_fail("input dropout on a feature object without a keep-bit array");
    if (s0 == 0 && s1 == S) {                    // every split at once: the persistent form when the shape is its
#ifdef GCNHIP_EXPERIMENTS
        if (c->opt.gemm_persist_bwd != 0) { const int rl = want_keep_layout(c, f, d, 0, fresh); if (rl) return rl; fresh = false; }
#endif
        const int rc = dense_bwd_persist(c, f, vals, dout, ld_dout, p, d);
        if (rc <= 0) return rc;
    }
    c->slab_n = S;
    const int kt = ceil_div(f->n_cols, 128), pt = ceil_div(p, 128);
    const int p_ld = (p + 3) / 4 * 4;
    const int rc = ensure_slab(c, (size_t)S * f->n_cols * p_ld * sizeof(float));
    if (rc) return rc;
    // Option gemm_bf16x3: the same split-K product from three bf16 planes per operand (dense_bf16x3.h): every split's slab,
    // then the same ordered slab sum.  The padded copy of X (row stride a multiple of 128, zeros past K) is what it reads.
    {
        const int bx = c->opt.gemm_bf16x3;
        const bool padded = vals == f->values && f->values_pad && f->ld_pad >= kt * 128;
        if (bx && (bx >= 2 || !c->corun) && p == 128 && padded && rps % 16 == 0 && ld_dout % 4 == 0 && aligned16(dout) &&
            (uint64_t)f->n_rows * (uint64_t)f->ld_pad * 4u < (1ull << 32) && (uint64_t)f->n_rows * (uint64_t)ld_dout * 4u < (1ull << 32) &&
            (uint64_t)f->n_rows * (uint64_t)f->n_cols < (1ull << 32) && (uint64_t)kt * 16u * (uint64_t)f->n_rows < (1ull << 32) &&
            (!d.on || keep_bits_cm_fit(f))) {
            { const int rl = want_keep_layout(c, f, d, 1, fresh); if (rl) return rl; }
            Bx3BwdArgs b;
            b.x = f->values_pad; b.ldx = f->ld_pad; b.dout = dout; b.ldd = ld_dout; b.slab = c->slab; b.p_ld = p_ld;
            b.m = f->n_rows; b.K = f->n_cols; b.rps = rps; b.split0 = s0; b.bits = d.on ? f->keep_bits : nullptr; b.scale = d.on ? d.scale : 1.f;
            const dim3 grid(kt, s1 - s0);
            // (load order 5: keep word first, X and dH0 loads interleaved: 0.230 vs 0.236 ms in the rotated runs of tools/gemm_bf16x3.hip)
            if (b.bits) dense_bwd_bf16x3_kernel<true, 6, 5><<<grid, 256, 0, c->stream>>>(b);
            else dense_bwd_bf16x3_kernel<false, 6><<<grid, 256, 0, c->stream>>>(b);
            GCNHIP_LAUNCH_CHECK();
            return 0;
        }
    }
    { const int rl = want_keep_layout(c, f, d, 0, fresh); if (rl) return rl; }
    Tile128Args t;
    t.x = vals; t.ldx = f->n_cols; t.w = dout; t.ldw = ld_dout; t.out = c->slab; t.ldo = p_ld;
    int vx = x_vec_width(f, vals);
    if (vals == f->values && f->values_pad) { t.x = f->values_pad; t.ldx = f->ld_pad; vx = 4; }
    t.m = f->n_rows; t.K = f->n_cols; t.p = p; t.bits = d.on ? f->keep_bits : nullptr; t.scale = d.scale; t.rows_per_split = rps; t.relu = 0;
    t.split0 = s0;
    dim3 grid(s1 - s0, kt, pt);
    const bool fast = vx == 4 && t.ldx % 4 == 0 && kt * 128 <= t.ldx && ld_dout % 4 == 0 && p % 128 == 0 && aligned16(dout);
    if (fast) dense_bwd_t128_kernel<4, true><<<grid, 256, 0, c->stream>>>(t);   // (an eight-wave form measured the same: 0.3836 vs 0.3834 ms)
    else if (vx == 4) dense_bwd_t128_kernel<4><<<grid, 256, 0, c->stream>>>(t);
    else if (vx == 2) dense_bwd_t128_kernel<2><<<grid, 256, 0, c->stream>>>(t);
    else dense_bwd_t128_kernel<1><<<grid, 256, 0, c->stream>>>(t);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

// the ordered sum of all slabs
static int dense_bwd_finish(gcnhip_ctx *c, const gcnhip_feat *f, float *dw, int ld_dw, int p) {
    int rps, S;
    if (!dense_bwd_plan(c, f, p, &rps, &S)) return -1;
    const int p_ld = (p + 3) / 4 * 4;
    const int n_slabs = c->slab_n > 0 ? c->slab_n : S;
    if (!c->slab || c->slab_bytes < (size_t)n_slabs * f->n_cols * p_ld * sizeof(float)) return -1;   // no part has run on this context
    launch_slab_reduce(c->slab, n_slabs, f->n_cols, p, p_ld, dw, ld_dw, c->stream);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

int gcnhip_spmm_bwd_plan(const gcnhip_ctx *c, const gcnhip_feat *f, int p, int *rows_per_split, int *n_splits) {
    if (!c || !f || p <= 0 || !rows_per_split || !n_splits) return -1;
    int rps = 0, S = 0;
    if (!dense_bwd_plan(c, f, p, &rps, &S)) { rps = 0; S = 0; }
    *rows_per_split = rps; *n_splits = S;
    return 0;
}

int gcnhip_spmm_bwd_part(gcnhip_ctx *c, const gcnhip_feat *f, const float *vals, const float *dout, int ld_dout, int p,
                         float p_drop, uint64_t seed, const uint32_t *d_epoch, uint64_t nnz_offset, const uint8_t *keep_mask,
                         int split_begin, int split_end, int make_decisions) {
    if (!c || !f || !vals || !dout || p <= 0 || ld_dout < p) return -1;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return -1;
    const DropSpec d = make_drop(p_drop, seed, d_epoch, nnz_offset, keep_mask);
    return dense_bwd_part(c, f, vals, dout, ld_dout, p, d, split_begin, split_end, make_decisions != 0);
}

int gcnhip_spmm_bwd_finish(gcnhip_ctx *c, const gcnhip_feat *f, float *dw, int ld_dw, int p) {
    if (!c || !f || !dw || p <= 0 || ld_dw < p) return -1;
    return dense_bwd_finish(c, f, dw, ld_dw, p);
}

int gcnhip_spmm_bwd(gcnhip_ctx *c, const gcnhip_feat *f, const float *vals, const float *dout, int ld_dout,
                    float *dw, int ld_dw, int p, float p_drop, uint64_t seed, const uint32_t *d_epoch,
                    uint64_t nnz_offset, const uint8_t *keep_mask) {
    if (!c || !f || !vals || !dout || !dw || p <= 0 || ld_dout < p || ld_dw < p) return -1;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return -1;
    const DropSpec d = make_drop(p_drop, seed, d_epoch, nnz_offset, keep_mask);
    int rps, S;
    if (dense_bwd_plan(c, f, p, &rps, &S)) {          // split-K 128 x 128 MFMA tiles + ordered slab sum
        const int rc = dense_bwd_part(c, f, vals, dout, ld_dout, p, d, 0, S, true);
        if (rc) return rc;
        return dense_bwd_finish(c, f, dw, ld_dw, p);
    }
    if (f->dense) {                                   // narrow outputs (p <= 64)
        if (d.on) { const int rc = make_keep_bits(c, f, d); if (rc) return rc; }     // not one Philox block per float4 in the GEMM
        return launch_atb(c, vals, f->n_cols, dout, ld_dout, dw, ld_dw, f->n_rows, f->n_cols, p,
                          d.on, p_drop, seed, d_epoch, nnz_offset, keep_mask, d.on ? f->keep_bits : nullptr);
    }
    // sparse X: CSC gather over the task list built with the feature object (spmm_sparse.h)
    const int p_ld = (p + 3) / 4 * 4;
    if (f->n_bwd_slots && f->bwd_part_ld < p_ld) {
        // the partial rows of cut columns: sized by the first call that needs them (synchronises once, like the split-K slabs)
        gcnhip_feat *fm = const_cast<gcnhip_feat *>(f);
        GCNHIP_TRY(hipStreamSynchronize(c->stream));
        if (fm->bwd_partials) { GCNHIP_TRY(hipFree(fm->bwd_partials)); fm->bwd_partials = nullptr; fm->bwd_part_ld = 0; }
        GCNHIP_TRY(hipMalloc((void **)&fm->bwd_partials, (size_t)f->n_bwd_slots * p_ld * sizeof(float)));
        fm->bwd_part_ld = p_ld;
    }
    SpBwdArgs a;
    a.tasks = f->bwd_tasks; a.n_tasks = f->n_bwd_tasks;
    a.csc_row = f->csc_row; a.csc_pos = f->csc_pos;
    a.csc_val = vals == f->values ? f->csc_val : nullptr;
    a.vals = vals; a.dout = dout; a.dw = dw; a.partials = f->bwd_partials; a.part_ld = f->bwd_part_ld;
    a.ld_dout = ld_dout; a.ld_dw = ld_dw; a.p = p; a.d = d;
    if (a.n_tasks == 0) return 0;
    const bool vec = ld_dout % 4 == 0 && aligned16(dout);
    const int units = vec ? (p + 3) / 4 : p;
    a.nnz_bytes = (int)std::min<int64_t>(f->nnz * 4, 0x7FFFFFFF);
    if (vec && units <= 16 && a.csc_val && f->nnz * 4 < 0x7FFFFFFF && !c->opt.spmm_general &&
        (f->nnz >= SPMM_NARROW_MIN_NNZ || c->opt.spmm_general < 0)) {                            // narrow rows: the shuffle-free kernel
#define SPQ(L_)                                                                                             \
    do {                                                                                                    \
        if (f->bwd_nw == 16) spmm_csc_bwd_q_kernel<L_, 16><<<a.n_tasks, 1024, 0, c->stream>>>(a);           \
        else if (f->bwd_nw == 4) spmm_csc_bwd_q_kernel<L_, 4><<<a.n_tasks, 256, 0, c->stream>>>(a);         \
        else spmm_csc_bwd_q_kernel<L_, 1><<<ceil_div(a.n_tasks, 4), 256, 0, c->stream>>>(a);                \
    } while (0)
        if (units <= 1) SPQ(1); else if (units <= 2) SPQ(2); else if (units <= 4) SPQ(4); else if (units <= 8) SPQ(8); else SPQ(16);
#undef SPQ
    } else {
#define SPB2(L_, V_)                                                                                        \
    do {                                                                                                    \
        if (f->bwd_nw == 16) spmm_csc_bwd_kernel<L_, V_, 16><<<a.n_tasks, 1024, 0, c->stream>>>(a);         \
        else if (f->bwd_nw == 4) spmm_csc_bwd_kernel<L_, V_, 4><<<a.n_tasks, 256, 0, c->stream>>>(a);       \
        else spmm_csc_bwd_kernel<L_, V_, 1><<<ceil_div(a.n_tasks, 4), 256, 0, c->stream>>>(a);              \
    } while (0)
#define SPB(L_)                                                                           \
    do {                                                                                  \
        if (vec) SPB2(L_, true); else SPB2(L_, false);                                    \
    } while (0)
    if (units <= 1) SPB(1); else if (units <= 2) SPB(2); else if (units <= 4) SPB(4);
    else if (units <= 8) SPB(8); else if (units <= 16) SPB(16); else if (units <= 32) SPB(32); else SPB(64);
#undef SPB
#undef SPB2
    }
    GCNHIP_LAUNCH_CHECK();
    if (f->n_bwd_split) {
        spmm_bwd_fold_kernel<<<ceil_div((int64_t)f->n_bwd_split * p, 256), 256, 0, c->stream>>>(f->bwd_split, f->n_bwd_split, f->bwd_partials,
                                                                                                f->bwd_part_ld, dw, ld_dw, p);
        GCNHIP_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"

GCNHIP_DEFINE_PRELOAD(spmm, pg_pack_w_kernel)

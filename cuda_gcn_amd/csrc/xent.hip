// xent.hip — softmax cross-entropy + accuracy in one pass over the logits.
// Reference: CrossEntropyLoss::forward (src/seq/module.cpp:124-161) and
// GCN::get_accuracy (src/seq/gcn.cpp:83-96); CUDA twins cuda_kernel.cu:166-200
// (one THREAD per block) and cuda_gcn.cu:100-120 (38 MB D2H + host loop).
// Here: one wave64 per row (lane j holds logit j; C > 64 loops), wave-shuffle
// max / sum, per-block partials reduced in block order (bitwise reproducible).
#include "common.h"
#include <stdlib.h>
#pragma clang fp contract(off)

struct XentArgs {
    float *logits;
    float *grad;
    const int32_t *truth;
    int ld, ld_grad, n_rows, C;
    int training, shift, acc_only;
    int count;                 // > 0: known number of labelled rows
    const int32_t *d_count;    // else read here
    const int32_t *rows;       // optional: only these rows (all labelled) are visited, n_rows = their number
    const float *grad_row_scale; // optional: row r of grad is multiplied by grad_row_scale[r] (the factored aggregation wants dinv . dZ)
    const float *terms;        // xent_terms_kernel: [2 * rows] {loss term, 1.f if correct} left by a loss epilogue (gcnhip_gs_loss)
    float *part_f;             // [blocks] loss partials
    int32_t *part_i;           // [blocks*2] {correct, total}
    // in-launch final reduction (xent_block_tail): the block that arrives last at `ticket` adds the partials in block order
    uint32_t *ticket;          // NULL: xent_finalize_kernel follows
    float *res;                // d_result[4] or NULL
    int32_t *res_i;            // d_result_i[2] or NULL
    // optional: what gcnhip_metrics_record would copy afterwards (gcnhip_metrics_record_with_next_loss)
    float *ring; int ring_capacity, ring_slot; const uint32_t *ring_epoch; const float *ring_sumsq;
};

__device__ inline float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, WAVE));
    return v;
}

constexpr int XENT_MAXC_REG = 4;     // up to 256 classes held in registers

// The end of both loss kernels.  Thread 0 holds the block's totals.  Without a ticket it stores them for
// xent_finalize_kernel.  With one, the final reduction happens in this launch: the partial leaves with agent-scope stores
// (write-through past this XCD's L2; cdna guide, Guideline 16), the thread drains them and takes a ticket; the block that
// draws the last one reads every partial with agent-scope loads and adds them exactly as xent_finalize_kernel does (256
// threads striding the block list, wave shuffles, four wave totals) — the result has the same bits, the launch after the
// loss (and, on request, the metrics_record launch after that) is gone.  No block waits for another.
__device__ inline void xent_block_tail(const XentArgs &a, float bl, int bc, int bt) {
    __shared__ int sh_last;
    __shared__ float shf[4];
    __shared__ int shi[8];
    if (!a.ticket) {
        if (threadIdx.x == 0) { a.part_f[blockIdx.x] = bl; a.part_i[blockIdx.x * 2] = bc; a.part_i[blockIdx.x * 2 + 1] = bt; }
        return;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.part_f + blockIdx.x, bl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.part_i + blockIdx.x * 2, bc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.part_i + blockIdx.x * 2 + 1, bt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The three agent-scope stores above are write-through; the drain below completes them before the ticket is taken,
        // and the reader starts with an ACQUIRE fence.  A RELEASE on the ticket itself (what the C++ memory model would ask
        // for; tried in round 4) makes every block write back this XCD's whole dirty L2 — the gradient rows this very
        // launch has just stored: +17 us per training loss launch at Reddit scale (0.046 -> 0.080 ms per epoch) for an
        // ordering the write-through stores already give on gfx9.  Adam's hand-off (elementwise.hip), whose launch leaves
        // ~1 MB dirty, does carry the release.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the relaxed ticket is a gfx9-family shortcut — agent-scope stores are write-through there; any other target gets the
        //  release the memory model asks for.  tests/test_ops_gpu.py::test_loss_final_reduction_in_the_launch_equals_the_second_launch
        //  pins it: 600 blocks, changing inputs, bit-compared against the two-launch form.)
#if defined(__gfx950__) || defined(__gfx942__) || defined(__gfx940__) || defined(__gfx90a__)
        const unsigned prev = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        const unsigned prev = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
        sh_last = prev == gridDim.x - 1;
        if (sh_last) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // as the next launch expects it
    }
    __syncthreads();
    if (!sh_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float l = 0.f;
    int c = 0, t = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
        l += __hip_atomic_load(a.part_f + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c += __hip_atomic_load(a.part_i + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t += __hip_atomic_load(a.part_i + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    l = wave_sum(l); c = wave_sum_i(c); t = wave_sum_i(t);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { shf[w] = l; shi[2 * w] = c; shi[2 * w + 1] = t; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int cc = shi[0] + shi[2] + shi[4] + shi[6], tt = shi[1] + shi[3] + shi[5] + shi[7];
        const float r0 = (shf[0] + shf[1]) + (shf[2] + shf[3]);
        if (a.res_i) { a.res_i[0] = cc; a.res_i[1] = tt; }
        if (a.res && !a.acc_only) { a.res[0] = r0; a.res[1] = (float)tt; a.res[2] = (float)cc; a.res[3] = (float)tt; }
        if (a.ring && !a.acc_only) {                          // metrics_record_kernel's row (elementwise.hip)
            const uint32_t e = a.ring_epoch ? *a.ring_epoch : 0u;
            float *row = a.ring + ((size_t)(e % (uint32_t)a.ring_capacity) * 4 + a.ring_slot) * 8;
            row[0] = r0; row[1] = (float)tt; row[2] = (float)cc; row[3] = (float)tt;
            row[4] = a.ring_sumsq ? *a.ring_sumsq : 0.f;
            row[5] = (float)e; row[6] = 0.f; row[7] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void xent_kernel(XentArgs a) {
    __shared__ float sh_f[4];
    __shared__ int sh_i[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;
    const int rows_per_wave = (a.n_rows + waves_total - 1) / waves_total;
    const int gw = blockIdx.x * 4 + wave;
    const int r0 = gw * rows_per_wave, r1 = min(a.n_rows, r0 + rows_per_wave);
    const float cnt = (float)(a.count > 0 ? a.count : (a.d_count ? *a.d_count : 0));
    float loss = 0.f;
    int correct = 0, total = 0;
    // the next row's logits and label are loaded while this row's reductions run (a wave walks ~50 rows at
    // Reddit scale; without the prefetch every row paid a full memory latency: 90 us for 233 K rows)
    float nv[XENT_MAXC_REG];
    int nt = -1;
    auto prefetch = [&](int q) {
        const int r = a.rows ? a.rows[q] : q;
        nt = a.truth[r];
        const float *lg = a.logits + (size_t)r * a.ld;
#pragma unroll
        for (int q = 0; q < XENT_MAXC_REG; q++) {
            const int j = lane + q * WAVE;
            nv[q] = j < a.C ? lg[j] : -INFINITY;
        }
    };
    if (r0 < r1) prefetch(r0);
    for (int q = r0; q < r1; q++) {
        const int r = a.rows ? a.rows[q] : q;
        const int t = nt;
        float v[XENT_MAXC_REG];
#pragma unroll
        for (int q = 0; q < XENT_MAXC_REG; q++) v[q] = nv[q];
        if (q + 1 < r1) prefetch(q + 1);
        float *lg = a.logits + (size_t)r * a.ld;
        float *gr = a.grad ? a.grad + (size_t)r * a.ld_grad : nullptr;
        if (t < 0) {                                   // unlabelled: grad row stays 0 (module.cpp:129,132)
            if (a.training && gr)
                for (int j = lane; j < a.C; j += WAVE) gr[j] = 0.f;
            continue;
        }
        total++;
        float mx = -1e30f;                             // module.cpp:135
#pragma unroll
        for (int q = 0; q < XENT_MAXC_REG; q++) {
            const int j = lane + q * WAVE;
            if (j < a.C) mx = fmaxf(mx, v[q]);
        }
        mx = wave_max(mx);
        // the true logit, fetched from the lane that holds it
        float tv = -INFINITY;
#pragma unroll
        for (int q = 0; q < XENT_MAXC_REG; q++)
            if (t / WAVE == q) tv = __shfl(v[q], t % WAVE, WAVE);
        // gcn.cpp:88-93: wrong iff some logit is strictly above the true one
        if (!(mx > tv)) correct++;
        if (a.acc_only) continue;
        float se = 0.f;
        float ex[XENT_MAXC_REG];
#pragma unroll
        for (int q = 0; q < XENT_MAXC_REG; q++) {
            const int j = lane + q * WAVE;
            v[q] -= mx;                                // module.cpp:140
            ex[q] = j < a.C ? expf(v[q]) : 0.f;
            se += ex[q];
            if (a.shift && j < a.C) lg[j] = v[q];
        }
        se = wave_sum(se);
        loss += logf(se) - (tv - mx);                  // module.cpp:143
        if (a.training && gr) {
#pragma unroll
            for (int q = 0; q < XENT_MAXC_REG; q++) {
                const int j = lane + q * WAVE;
                if (j < a.C) {
                    float p = ex[q] / se;              // module.cpp:147
                    if (j == t) p = (float)((double)p - 1.0);
                    gr[j] = a.grad_row_scale ? (p / cnt) * a.grad_row_scale[r] : p / cnt;   // module.cpp:157
                }
            }
        }
    }
    // block partials (lane 0 of each wave carries the wave's totals)
    if (lane == 0) { sh_f[wave] = loss; sh_i[wave * 2] = correct; sh_i[wave * 2 + 1] = total; }
    __syncthreads();
    float bl = 0.f;
    int bc = 0, bt = 0;
    if (threadIdx.x == 0) {
        bl = (sh_f[0] + sh_f[1]) + (sh_f[2] + sh_f[3]);
        bc = sh_i[0] + sh_i[2] + sh_i[4] + sh_i[6];
        bt = sh_i[1] + sh_i[3] + sh_i[5] + sh_i[7];
    }
    xent_block_tail(a, bl, bc, bt);
}

// ---- narrow logits (C <= 64, rows of whole float4 pieces): one LANE per row --------------------------------------
// The wave-per-row form above walks ~19 rows per wave at Reddit scale, each a dependent chain rows[q] -> logits -> two
// wave reductions, with 41 of 64 lanes busy: 62 us for 154 K rows of 41 classes, 1 TB/s.  Here a lane owns a row: its
// NV4 float4 loads are independent and all in flight at once, max / sum run down the lane's registers IN THE REFERENCE'S
// ORDER (module.cpp:135-143 sums exp left to right; the wave form used a shuffle tree), no shuffles until the block's
// partials.  Same partial layout and finalize kernel as the wave form.
template <int NV4>
__global__ __launch_bounds__(256) void xent_lane_kernel(XentArgs a) {
    __shared__ float sh_f[4];
    __shared__ int sh_i[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float cnt = (float)(a.count > 0 ? a.count : (a.d_count ? *a.d_count : 0));
    float loss = 0.f;
    int correct = 0, total = 0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < a.n_rows; q += gridDim.x * 256) {
        const int r = a.rows ? a.rows[q] : q;
        const int t = a.truth[r];
        float *lg = a.logits + (size_t)r * a.ld;
        float *gr = a.grad ? a.grad + (size_t)r * a.ld_grad : nullptr;
        if (t < 0) {                                   // unlabelled: grad row stays 0 (module.cpp:129,132)
            if (a.training && gr)
#pragma unroll
                for (int k = 0; k < NV4; k++) reinterpret_cast<float4 *>(gr)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float v[4 * NV4];
#pragma unroll
        for (int k = 0; k < NV4; k++) {
            const float4 x = reinterpret_cast<const float4 *>(lg)[k];
            v[4 * k] = x.x; v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w;
        }
        total++;
        float mx = -1e30f, tv = -INFINITY;             // module.cpp:135
#pragma unroll
        for (int j = 0; j < 4 * NV4; j++) {
            if (j < a.C) mx = fmaxf(mx, v[j]);
            tv = j == t ? v[j] : tv;
        }
        if (!(mx > tv)) correct++;                     // gcn.cpp:88-93: wrong iff some logit is strictly above the true one
        if (a.acc_only) continue;
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < 4 * NV4; j++) {
            v[j] -= mx;                                // module.cpp:140
            if (a.shift && j < a.C) lg[j] = v[j];
            v[j] = j < a.C ? expf(v[j]) : 0.f;
            se += v[j];                                // left to right, as module.cpp:141-142
        }
        loss += logf(se) - (tv - mx);                  // module.cpp:143
        if (a.training && gr) {
            const float gs = a.grad_row_scale ? a.grad_row_scale[r] : 1.f;
#pragma unroll
            for (int j = 0; j < 4 * NV4; j++) {
                float p = v[j] / se;                   // module.cpp:147
                if (j == t) p = (float)((double)p - 1.0);
                v[j] = j < a.C ? (a.grad_row_scale ? (p / cnt) * gs : p / cnt) : 0.f;        // module.cpp:157; the padding columns stay zero
            }
#pragma unroll
            for (int k = 0; k < NV4; k++)
                reinterpret_cast<float4 *>(gr)[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
        }
    }
    loss = wave_sum(loss); correct = wave_sum_i(correct); total = wave_sum_i(total);
    if (lane == 0) { sh_f[wave] = loss; sh_i[wave * 2] = correct; sh_i[wave * 2 + 1] = total; }
    __syncthreads();
    float bl = 0.f;
    int bc = 0, bt = 0;
    if (threadIdx.x == 0) {
        bl = (sh_f[0] + sh_f[1]) + (sh_f[2] + sh_f[3]);
        bc = sh_i[0] + sh_i[2] + sh_i[4] + sh_i[6];
        bt = sh_i[1] + sh_i[3] + sh_i[5] + sh_i[7];
    }
    xent_block_tail(a, bl, bc, bt);
}

// The end of xent_lane_kernel for rows whose loss term and accuracy flag already exist (the loss epilogue of the aggregation
// that produced the logits, graphsum.hip): the same lane -> row assignment, the same per-lane order of additions, the same
// block tail — the totals have the bits xent_lane_kernel would have produced from the stored logits.
__global__ __launch_bounds__(256) void xent_terms_kernel(XentArgs a) {
    __shared__ float sh_f[4];
    __shared__ int sh_i[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss = 0.f;
    int correct = 0, total = 0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < a.n_rows; q += gridDim.x * 256) {
        const int r = a.rows ? a.rows[q] : q;
        if (a.truth[r] < 0) continue;
        const float2 tm = *reinterpret_cast<const float2 *>(a.terms + 2 * (size_t)r);
        total++;
        if (tm.y != 0.f) correct++;
        loss += tm.x;
    }
    loss = wave_sum(loss); correct = wave_sum_i(correct); total = wave_sum_i(total);
    if (lane == 0) { sh_f[wave] = loss; sh_i[wave * 2] = correct; sh_i[wave * 2 + 1] = total; }
    __syncthreads();
    float bl = 0.f;
    int bc = 0, bt = 0;
    if (threadIdx.x == 0) {
        bl = (sh_f[0] + sh_f[1]) + (sh_f[2] + sh_f[3]);
        bc = sh_i[0] + sh_i[2] + sh_i[4] + sh_i[6];
        bt = sh_i[1] + sh_i[3] + sh_i[5] + sh_i[7];
    }
    xent_block_tail(a, bl, bc, bt);
}

__global__ __launch_bounds__(256) void count_labelled_kernel(const int32_t *truth, int n, int32_t *part) {
    int c = 0;
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * chunk, b1 = min(n, b0 + chunk);
    for (int i = b0 + threadIdx.x; i < b1; i += 256) c += truth[i] >= 0 ? 1 : 0;
    c = wave_sum_i(c);
    __shared__ int sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void sum_int_partials_kernel(const int32_t *part, int n, int32_t *out) {
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 256) c += part[i];
    c = wave_sum_i(c);
    __shared__ int sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) *out = sh[0] + sh[1] + sh[2] + sh[3];
}

// fixed-order final reduction of the block partials
__global__ __launch_bounds__(256) void xent_finalize_kernel(const float *part_f, const int32_t *part_i, int n,
                                                            float *res, int32_t *res_i, int acc_only) {
    __shared__ float shf[4];
    __shared__ int shi[8];
    float l = 0.f;
    int c = 0, t = 0;
    for (int i = threadIdx.x; i < n; i += 256) { l += part_f[i]; c += part_i[2 * i]; t += part_i[2 * i + 1]; }
    l = wave_sum(l); c = wave_sum_i(c); t = wave_sum_i(t);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { shf[w] = l; shi[2 * w] = c; shi[2 * w + 1] = t; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int cc = shi[0] + shi[2] + shi[4] + shi[6], tt = shi[1] + shi[3] + shi[5] + shi[7];
        if (res_i) { res_i[0] = cc; res_i[1] = tt; }
        if (res && !acc_only) {
            res[0] = (shf[0] + shf[1]) + (shf[2] + shf[3]);
            res[1] = (float)tt;
            res[2] = (float)cc;
            res[3] = (float)tt;
        }
    }
}

static int xent_launch(gcnhip_ctx *c, XentArgs a, float *d_result, int32_t *d_result_i) {
    if (a.C > XENT_MAXC_REG * WAVE) return -1;
    int blocks = ceil_div(a.n_rows, 4 * 2);             // ~2 rows per wave on small inputs (each row is a dependent load chain); the cap below decides on large ones
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;                   // part_i holds 2 ints per block in red_i[0, 4096)
    if (a.n_rows == 0) blocks = 1;                      // a rank that owns no rows still reports zeros
    a.part_f = c->red_f + 2048;
    a.part_i = c->red_i;
    // the final reduction (and an armed metrics record) ride in the loss launch; the context option xent_finalize keeps the second launch
    const bool two_launches = c->opt.xent_finalize != 0;
    a.ticket = two_launches ? nullptr : c->ticket;
    a.res = d_result; a.res_i = d_result_i;
    a.ring = nullptr; a.ring_capacity = 1; a.ring_slot = 0; a.ring_epoch = nullptr; a.ring_sumsq = nullptr;
    const bool record = c->rec_armed && !a.acc_only && d_result;
    if (record) {
        a.ring = c->rec_ring; a.ring_capacity = c->rec_capacity; a.ring_slot = c->rec_slot; a.ring_epoch = c->rec_epoch; a.ring_sumsq = c->rec_sumsq;
        c->rec_armed = false;
    }
    if (!a.acc_only && a.training && a.count <= 0) {    // count first, like module.cpp:127-133
        int cb = ceil_div(a.n_rows, 4096);
        if (cb > 1024) cb = 1024;
        if (cb < 1) cb = 1;
        count_labelled_kernel<<<cb, 256, 0, c->stream>>>(a.truth, a.n_rows, c->red_i + 4096);
        GCNHIP_LAUNCH_CHECK();
        sum_int_partials_kernel<<<1, 256, 0, c->stream>>>(c->red_i + 4096, cb, c->red_i + 8192);
        GCNHIP_LAUNCH_CHECK();
        a.d_count = c->red_i + 8192;
    }
    if (a.terms) {                                     // same grid as the lane-per-row kernel below
        blocks = ceil_div(a.n_rows, 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        xent_terms_kernel<<<blocks, 256, 0, c->stream>>>(a);
        GCNHIP_LAUNCH_CHECK();
        if (!a.ticket) {
            xent_finalize_kernel<<<1, 256, 0, c->stream>>>(a.part_f, a.part_i, blocks, d_result, d_result_i, a.acc_only);
            GCNHIP_LAUNCH_CHECK();
            if (record) return gcnhip_metrics_record(c, a.ring, a.ring_capacity, a.ring_slot, a.ring_epoch, d_result, nullptr, a.ring_sumsq);
        }
        return 0;
    }
    // a lane per row when the rows are whole, aligned float4 pieces of at most 64 classes
    const int nv4 = (a.C + 3) / 4;
    const bool force_wave = c->opt.xent_wave != 0;                            // A/B aid
    const bool lanes = !force_wave && a.C <= 64 && a.n_rows > 0 && a.ld % 4 == 0 && a.ld >= 4 * nv4 && aligned16(a.logits) &&
                       (!a.grad || (a.ld_grad % 4 == 0 && a.ld_grad >= 4 * nv4 && aligned16(a.grad)));
    if (lanes) {
        blocks = ceil_div(a.n_rows, 256);
        if (blocks > 2048) blocks = 2048;
        switch (nv4) {
#define XL(N) case N: xent_lane_kernel<N><<<blocks, 256, 0, c->stream>>>(a); break;
            XL(1) XL(2) XL(3) XL(4) XL(5) XL(6) XL(7) XL(8) XL(9) XL(10) XL(11) XL(12) XL(13) XL(14) XL(15) XL(16)
#undef XL
        }
    } else {
        xent_kernel<<<blocks, 256, 0, c->stream>>>(a);
    }
    GCNHIP_LAUNCH_CHECK();
    if (!a.ticket) {
        xent_finalize_kernel<<<1, 256, 0, c->stream>>>(a.part_f, a.part_i, blocks, d_result, d_result_i, a.acc_only);
        GCNHIP_LAUNCH_CHECK();
        if (record) return gcnhip_metrics_record(c, a.ring, a.ring_capacity, a.ring_slot, a.ring_epoch, d_result, nullptr, a.ring_sumsq);
    }
    return 0;
}

extern "C" {

int gcnhip_xent_fwd(gcnhip_ctx *c, float *logits, int ld, float *grad, int ld_grad,
                    const int32_t *truth, int n_rows, int num_classes, int training,
                    int count, int shift_in_place, float *d_result, int32_t *d_result_i) {
    if (!c || !logits || !truth || !d_result || num_classes <= 0 || ld < num_classes) return -1;
    if (training && (!grad || ld_grad < num_classes)) return -1;
    if (n_rows < 0) return -1;
    XentArgs a;
    a.logits = logits; a.grad = training ? grad : nullptr; a.truth = truth;
    a.ld = ld; a.ld_grad = ld_grad; a.n_rows = n_rows; a.C = num_classes;
    a.training = training; a.shift = shift_in_place; a.acc_only = 0;
    a.count = count; a.d_count = nullptr; a.rows = nullptr; a.grad_row_scale = nullptr; a.terms = nullptr;
    return xent_launch(c, a, d_result, d_result_i);
}

int gcnhip_xent_fwd_rows(gcnhip_ctx *c, float *logits, int ld, float *grad, int ld_grad,
                         const int32_t *truth, const int32_t *d_rows, int n_listed, int num_classes, int training,
                         int count, int shift_in_place, float *d_result, int32_t *d_result_i) {
    return gcnhip_xent_fwd_rows_scaled(c, logits, ld, grad, ld_grad, truth, d_rows, n_listed, num_classes, training, count, shift_in_place,
                                       d_result, d_result_i, nullptr);
}

int gcnhip_xent_fwd_rows_scaled(gcnhip_ctx *c, float *logits, int ld, float *grad, int ld_grad,
                                const int32_t *truth, const int32_t *d_rows, int n_listed, int num_classes, int training,
                                int count, int shift_in_place, float *d_result, int32_t *d_result_i, const float *d_grad_row_scale) {
    if (!c || !logits || !truth || !d_result || num_classes <= 0 || ld < num_classes || n_listed < 0 || count <= 0) return -1;
    if (n_listed > 0 && !d_rows) return -1;
    if (training && (!grad || ld_grad < num_classes)) return -1;
    XentArgs a;
    a.logits = logits; a.grad = training ? grad : nullptr; a.truth = truth;
    a.ld = ld; a.ld_grad = ld_grad; a.n_rows = n_listed; a.C = num_classes;
    a.training = training; a.shift = shift_in_place; a.acc_only = 0;
    a.count = count; a.d_count = nullptr; a.rows = d_rows; a.grad_row_scale = d_grad_row_scale; a.terms = nullptr;
    return xent_launch(c, a, d_result, d_result_i);
}

int gcnhip_xent_from_row_terms(gcnhip_ctx *c, const float *d_row_terms, const int32_t *truth, const int32_t *d_rows, int n_listed,
                               float *d_result, int32_t *d_result_i) {
    if (!c || !d_row_terms || !truth || !d_result || n_listed < 0 || ((uintptr_t)d_row_terms & 7)) return -1;
    if (n_listed > 0 && !d_rows) return -1;
    XentArgs a;
    a.logits = nullptr; a.grad = nullptr; a.truth = truth;
    a.ld = 0; a.ld_grad = 0; a.n_rows = n_listed; a.C = 1;
    a.training = 0; a.shift = 0; a.acc_only = 0;
    a.count = 1; a.d_count = nullptr; a.rows = d_rows; a.grad_row_scale = nullptr; a.terms = d_row_terms;
    return xent_launch(c, a, d_result, d_result_i);
}

int gcnhip_accuracy(gcnhip_ctx *c, const float *logits, int ld, const int32_t *truth,
                    int n_rows, int num_classes, int32_t *d_result_i) {
    if (!c || !logits || !truth || !d_result_i || num_classes <= 0 || ld < num_classes || n_rows < 0) return -1;
    XentArgs a;
    a.logits = const_cast<float *>(logits); a.grad = nullptr; a.truth = truth;
    a.ld = ld; a.ld_grad = 0; a.n_rows = n_rows; a.C = num_classes;
    a.training = 0; a.shift = 0; a.acc_only = 1; a.count = 1; a.d_count = nullptr; a.rows = nullptr; a.grad_row_scale = nullptr; a.terms = nullptr;
    return xent_launch(c, a, nullptr, d_result_i);
}

}  // extern "C"

GCNHIP_DEFINE_PRELOAD(xent, count_labelled_kernel)

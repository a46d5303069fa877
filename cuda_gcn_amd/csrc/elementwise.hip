// elementwise.hip — ReLU, Dropout, set_truth, sum of squares, Adam and the
// small device utilities.  All HBM-bound streaming kernels: 16-byte lane
// accesses where alignment allows, grid capped at 2048 blocks with a
// grid-stride loop (cdna guide, Guideline 11/13).  Reductions go through
// per-block partials summed in block order by a one-block kernel, so every
// result is bitwise reproducible (no float atomics).
#include "common.h"
#include <stdlib.h>

static inline int stream_grid(int64_t n_items, int per_block) {
    int64_t b = (n_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// ------------------------------------------------------------------ ReLU
// src/seq/module.cpp:175-194, src/cuda/cuda_kernel.cu:204-219
// Element i of the logical [rows x cols] matrix (the reference's flat index: masks and the dropout stream are
// keyed by it) lives at (i / cols) * ld + i % cols.  The flat entry points pass cols == ld == 1.
__device__ inline int64_t at(int64_t i, int cols, int ld) { const int64_t r = i / cols; return r * ld + (i - r * cols); }

__global__ __launch_bounds__(256) void relu_fwd_kernel(float *x, uint8_t *mask, int64_t n, int training, int cols, int ld) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float *xp = x + at(i, cols, ld);
        const bool keep = *xp > 0.f;
        if (training) mask[i] = keep ? 1 : 0;
        if (!keep) *xp = 0.f;
    }
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(float *g, const uint8_t *mask, int64_t n, int cols, int ld) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (!mask[i]) g[at(i, cols, ld)] = 0.f;
}

// --------------------------------------------------------------- Dropout
// src/seq/module.cpp:207-233, src/cuda/cuda_kernel.cu:223-240
__global__ __launch_bounds__(256) void dropout_fwd_kernel(float *x, int32_t *mask, int64_t n, int thr, float scale,
                                                          uint64_t seed, const uint32_t *d_epoch, uint64_t off,
                                                          const uint8_t *keep_in, int cols, int ld) {
    const uint32_t epoch = d_epoch ? *d_epoch : 0u;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool keep = keep_in ? keep_in[i] != 0 : keep1(off + (uint64_t)i, epoch, seed, thr);
        x[at(i, cols, ld)] *= keep ? scale : 0.f;
        if (mask) mask[i] = keep ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(float *g, const int32_t *mask, int64_t n, float scale, int cols, int ld) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        g[at(i, cols, ld)] *= mask[i] ? scale : 0.f;
}
// fused ReLU+Dropout backward: the forward output h is > 0 exactly where both kept
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(float *g, int ldg, const float *h, int ldh,
                                                               int n_rows, int dim, float scale) {
    const int64_t total = (int64_t)n_rows * dim;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / dim;
        const int c = (int)(i - r * dim);
        float *gp = g + r * ldg + c;
        *gp = h[r * ldh + c] > 0.f ? *gp * scale : 0.f;
    }
}

// ------------------------------------------------------------ row packing
// dst[i, 0:ld] = src[rows[i], 0:ld]: the send side of a halo exchange (rows a peer needs, packed contiguously).
// Words are moved, not interpreted (f32 rows, bf16 rows and mask words alike).
template <int V>
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t *__restrict__ src, int ld, const int *__restrict__ rows,
                                                          int64_t total /* n * ld / V */, uint32_t *__restrict__ dst) {
    const int per_row = ld / V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row) * V;
        const uint32_t *s = src + (int64_t)rows[r] * ld + c;
        uint32_t *d = dst + r * ld + c;
        if (V == 4) *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(s);
        else *d = *s;
    }
}

// -------------------------------------------------------------- set_truth
// src/seq/gcn.cpp:78-81, src/cuda/cuda_kernel.cu:283-288
__global__ __launch_bounds__(256) void set_truth_kernel(int32_t *truth, const int32_t *split, const int32_t *label, int n, int s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) truth[i] = split[i] == s ? label[i] : -1;
}

// ------------------------------------------------------------ reductions
__device__ inline float block_sum(float v, float *sh) {   // 256 threads
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float *x, int64_t n, float *partial) {
    __shared__ float sh[4];
    // contiguous chunk per block: the summation tree is a function of n only
    const int64_t chunk = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = min(n, b0 + chunk);
    float acc = 0.f;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) { const float v = x[i]; acc += v * v; }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float *partial, int n, float *out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) *out = s;
}

// ------------------------------------------------------------------- Adam
// src/seq/optim.cpp:24-37, src/cuda/cuda_kernel.cu:270-281
struct AdamArgs {
    gcnhip_adam_var v[4];
    int64_t start[5];           // element offset of each variable in the fused index space
    int n_vars;
    float step_size, beta1, beta2, eps, wd;
    const float *d_step_sizes;
    const uint32_t *d_epoch;
    float *partial;             // per-block sum of w0^2 after the update (or NULL)
    uint32_t *ticket;           // with `partial`: the block that arrives last adds the partials (NULL: sum_partials_kernel follows)
    float *sumsq_out;
    uint32_t *epoch_counter, *epoch_done;   // with `ticket`: the last block leaves *epoch_done = e, *epoch_counter = e + 1 (NULL: no advance)
};
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    __shared__ float sh[4];
    const float step = a.d_step_sizes ? a.d_step_sizes[*a.d_epoch] : a.step_size;
    const double omb1 = 1.0 - (double)a.beta1, omb2 = 1.0 - (double)a.beta2;   // "(1.0 - beta)" in double
    const int64_t total = a.start[a.n_vars];
    const int64_t chunk = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = min(total, b0 + chunk);
    float sq = 0.f;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < 4; q++) if (q < a.n_vars && i >= a.start[q]) k = q;
        const gcnhip_adam_var &v = a.v[k];
        const int64_t j = i - a.start[k];
        float w = v.w[j];
        float grad = v.g[j];
        if (v.decay) grad += a.wd * w;
        const float m = (float)((double)(a.beta1 * v.m[j]) + omb1 * (double)grad);
        const float vv = (float)((double)(a.beta2 * v.v[j]) + omb2 * (double)grad * (double)grad);
        v.m[j] = m;
        v.v[j] = vv;
        w -= step * m / (sqrtf(vv) + a.eps);
        v.w[j] = w;
        if (k == 0) sq += w * w;
    }
    if (a.partial) {
        const float s = block_sum(sq, sh);
        if (!a.ticket) {
            if (threadIdx.x == 0) a.partial[blockIdx.x] = s;
            return;
        }
        // last-block sum in this launch (same hand-off as xent_block_tail, xent.hip): agent-scope store of the partial,
        // drain, ticket; the last block adds the partials as sum_partials_kernel does (256 threads striding the list,
        // block_sum) — same bits, one launch fewer per epoch
        __shared__ int sh_last;
        if (threadIdx.x == 0) {
            __hip_atomic_store(a.partial + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned prev = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // release / acquire: see xent_block_tail
            sh_last = prev == gridDim.x - 1;
            if (sh_last) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!sh_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float acc = 0.f;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) acc += __hip_atomic_load(a.partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float tot = block_sum(acc, sh);
        if (threadIdx.x == 0) {
            *a.sumsq_out = tot;
            if (a.epoch_counter) {                  // every block read the word (the step size index) before it took its ticket
                const uint32_t e = *a.epoch_counter;
                if (a.epoch_done) *a.epoch_done = e;
                *a.epoch_counter = e + 1u;
            }
        }
    }
}
__global__ void epoch_advance_kernel(uint32_t *counter, uint32_t *done) {
    const uint32_t e = *counter;
    if (done) *done = e;
    *counter = e + 1u;
}

// bits[r*wpr + (c >> 5)] bit (c & 31) = h[r, c] > 0 — one wave per row, one ballot per 64 columns
__global__ __launch_bounds__(256) void pack_positive_kernel(const float *h, int ld, int n_rows, int dim, uint32_t *bits, int wpr) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    for (int c0 = 0; c0 < wpr * 32; c0 += 64) {
        const int c = c0 + lane;
        const bool pos = c < dim && h[(size_t)r * ld + c] > 0.f;
        const unsigned long long b = __ballot(pos);
        if (lane == 0) {
            bits[(size_t)r * wpr + (c0 >> 5)] = (uint32_t)b;
            if ((c0 >> 5) + 1 < wpr) bits[(size_t)r * wpr + (c0 >> 5) + 1] = (uint32_t)(b >> 32);
        }
    }
}

// ----------------------------------------------------------- small utilities
__global__ void counter_add_kernel(uint32_t *c, uint32_t inc) { *c += inc; }
__global__ void metrics_record_kernel(float *ring, int capacity, int slot, const uint32_t *d_epoch,
                                      const float *res, const int32_t *res_i, const float *sumsq) {
    const uint32_t e = d_epoch ? *d_epoch : 0u;
    float *row = ring + ((size_t)(e % (uint32_t)capacity) * 4 + slot) * 8;
    row[0] = res[0];
    row[1] = res[1];
    row[2] = res_i ? (float)res_i[0] : res[2];
    row[3] = res_i ? (float)res_i[1] : res[3];
    row[4] = sumsq ? *sumsq : 0.f;
    row[5] = (float)e;
    row[6] = 0.f;
    row[7] = 0.f;
}

extern "C" {

int gcnhip_relu_fwd_2d(gcnhip_ctx *c, float *x, int ld, int rows, int cols, uint8_t *mask, int training) {
    if (!c || !x || (training && !mask) || rows < 0 || cols < 0 || ld < cols) return -1;
    const int64_t n = (int64_t)rows * cols;
    if (n <= 0) return 0;
    relu_fwd_kernel<<<stream_grid(n, 1024), 256, 0, c->stream>>>(x, mask, n, training, cols, ld);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
int gcnhip_relu_bwd_2d(gcnhip_ctx *c, float *grad, int ld, int rows, int cols, const uint8_t *mask) {
    if (!c || !grad || !mask || rows < 0 || cols < 0 || ld < cols) return -1;
    const int64_t n = (int64_t)rows * cols;
    if (n <= 0) return 0;
    relu_bwd_kernel<<<stream_grid(n, 1024), 256, 0, c->stream>>>(grad, mask, n, cols, ld);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
int gcnhip_dropout_fwd_2d(gcnhip_ctx *c, float *x, int ld, int rows, int cols, int32_t *mask, float p,
                          uint64_t seed, const uint32_t *d_epoch, uint64_t elem_offset, const uint8_t *keep_in) {
    if (!c || !x || !(p >= 0.f && p < 1.f) || rows < 0 || cols < 0 || ld < cols) return -1;
    const int64_t n = (int64_t)rows * cols;
    if (n <= 0) return 0;
    dropout_fwd_kernel<<<stream_grid(n, 1024), 256, 0, c->stream>>>(x, mask, n, dropout_threshold(p), 1 / (1 - p),
                                                                    seed, d_epoch, elem_offset, keep_in, cols, ld);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
int gcnhip_dropout_bwd_2d(gcnhip_ctx *c, float *grad, int ld, int rows, int cols, const int32_t *mask, float p) {
    if (!c || !grad || rows < 0 || cols < 0 || ld < cols) return -1;
    const int64_t n = (int64_t)rows * cols;
    if (!mask || n <= 0) return 0;               // module.cpp:224: no mask, no-op
    dropout_bwd_kernel<<<stream_grid(n, 1024), 256, 0, c->stream>>>(grad, mask, n, 1 / (1 - p), cols, ld);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
// the reference's flat form (one contiguous array of n elements): a single row
int gcnhip_relu_fwd(gcnhip_ctx *c, float *x, uint8_t *mask, int64_t n, int training) {
    if (n > 0x7fffffff) return -1;
    return gcnhip_relu_fwd_2d(c, x, (int)n, 1, (int)n, mask, training);
}
int gcnhip_relu_bwd(gcnhip_ctx *c, float *grad, const uint8_t *mask, int64_t n) {
    if (n > 0x7fffffff) return -1;
    return gcnhip_relu_bwd_2d(c, grad, (int)n, 1, (int)n, mask);
}
int gcnhip_dropout_fwd(gcnhip_ctx *c, float *x, int32_t *mask, int64_t n, float p,
                       uint64_t seed, const uint32_t *d_epoch, uint64_t elem_offset, const uint8_t *keep_in) {
    if (n > 0x7fffffff) return -1;
    return gcnhip_dropout_fwd_2d(c, x, (int)n, 1, (int)n, mask, p, seed, d_epoch, elem_offset, keep_in);
}
int gcnhip_dropout_bwd(gcnhip_ctx *c, float *grad, const int32_t *mask, int64_t n, float p) {
    if (n > 0x7fffffff) return -1;
    return gcnhip_dropout_bwd_2d(c, grad, (int)n, 1, (int)n, mask, p);
}
int gcnhip_relu_dropout_bwd(gcnhip_ctx *c, float *grad, int ld_grad, const float *h, int ld_h,
                            int n_rows, int dim, float scale) {
    if (!c || !grad || !h || ld_grad < dim || ld_h < dim) return -1;
    if (n_rows <= 0 || dim <= 0) return 0;
    relu_dropout_bwd_kernel<<<stream_grid((int64_t)n_rows * dim, 1024), 256, 0, c->stream>>>(grad, ld_grad, h, ld_h, n_rows, dim, scale);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
int gcnhip_pack_positive(gcnhip_ctx *c, const float *h, int ld, int n_rows, int dim, uint32_t *bits, int words_per_row) {
    if (!c || !h || !bits || ld < dim || dim <= 0 || words_per_row * 32 < dim) return -1;
    if (n_rows <= 0) return 0;
    pack_positive_kernel<<<ceil_div(n_rows, 4), 256, 0, c->stream>>>(h, ld, n_rows, dim, bits, words_per_row);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
int gcnhip_gather_rows(gcnhip_ctx *c, const float *src, int ld_words, const int *d_rows, int n, float *dst) {
    if (!c || !src || !dst || ld_words <= 0 || n < 0 || (n > 0 && !d_rows)) return -1;
    if (n == 0) return 0;
    const bool v4 = ld_words % 4 == 0 && aligned16(src) && aligned16(dst);
    const int64_t total = (int64_t)n * ld_words / (v4 ? 4 : 1);
    if (v4) gather_rows_kernel<4><<<stream_grid(total, 1024), 256, 0, c->stream>>>((const uint32_t *)src, ld_words, d_rows, total, (uint32_t *)dst);
    else gather_rows_kernel<1><<<stream_grid(total, 1024), 256, 0, c->stream>>>((const uint32_t *)src, ld_words, d_rows, total, (uint32_t *)dst);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
int gcnhip_set_truth(gcnhip_ctx *c, int32_t *truth, const int32_t *split, const int32_t *label, int n, int s) {
    if (!c || !truth || !split || !label) return -1;
    if (n <= 0) return 0;
    set_truth_kernel<<<ceil_div(n, 256), 256, 0, c->stream>>>(truth, split, label, n, s);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
int gcnhip_sumsq(gcnhip_ctx *c, const float *x, int64_t n, float *d_out) {
    if (!c || !x || !d_out) return -1;
    const int blocks = stream_grid(n, 4096) > 1024 ? 1024 : stream_grid(n, 4096);
    sumsq_partial_kernel<<<blocks, 256, 0, c->stream>>>(x, n, c->red_f);
    GCNHIP_LAUNCH_CHECK();
    sum_partials_kernel<<<1, 256, 0, c->stream>>>(c->red_f, blocks, d_out);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
static int adam_step_impl(gcnhip_ctx *c, const gcnhip_adam_var *vars, int n_vars, float step_size,
                          const float *d_step_sizes, const uint32_t *d_epoch,
                          float beta1, float beta2, float eps, float weight_decay, float *d_sumsq,
                          uint32_t *d_epoch_counter, uint32_t *d_epoch_done) {
    if (!c || !vars || n_vars < 1 || n_vars > 4) return -1;
    if (d_step_sizes && !d_epoch) return -1;
    AdamArgs a;
    a.n_vars = n_vars;
    a.start[0] = 0;
    for (int k = 0; k < n_vars; k++) { a.v[k] = vars[k]; a.start[k + 1] = a.start[k] + vars[k].n; }
    for (int k = n_vars; k < 4; k++) { a.v[k] = vars[0]; a.start[k + 1] = a.start[n_vars]; }
    a.step_size = step_size; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.d_step_sizes = d_step_sizes; a.d_epoch = d_epoch;
    int blocks = stream_grid(a.start[n_vars], 1024);
    if (blocks > 1024) blocks = 1024;
    a.partial = d_sumsq ? c->red_f + 1024 : nullptr;       // second quarter of the scratch
    const bool two_launches = c->opt.adam_sum_launch != 0;                      // A/B aid, tests (context option)
    a.ticket = (d_sumsq && !two_launches) ? c->ticket + 1 : nullptr;
    a.sumsq_out = d_sumsq;
    a.epoch_counter = a.ticket ? d_epoch_counter : nullptr;     // rides on the in-launch final reduction
    a.epoch_done = a.ticket ? d_epoch_done : nullptr;
    adam_kernel<<<blocks, 256, 0, c->stream>>>(a);
    GCNHIP_LAUNCH_CHECK();
    if (d_sumsq && !a.ticket) {
        sum_partials_kernel<<<1, 256, 0, c->stream>>>(a.partial, blocks, d_sumsq);
        GCNHIP_LAUNCH_CHECK();
    }
    if (d_epoch_counter && !a.epoch_counter) {                  // no final reduction in the launch to carry it: its own launch
        epoch_advance_kernel<<<1, 1, 0, c->stream>>>(d_epoch_counter, d_epoch_done);
        GCNHIP_LAUNCH_CHECK();
    }
    return 0;
}
int gcnhip_adam_step(gcnhip_ctx *c, const gcnhip_adam_var *vars, int n_vars, float step_size,
                     const float *d_step_sizes, const uint32_t *d_epoch,
                     float beta1, float beta2, float eps, float weight_decay, float *d_sumsq) {
    return adam_step_impl(c, vars, n_vars, step_size, d_step_sizes, d_epoch, beta1, beta2, eps, weight_decay, d_sumsq, nullptr, nullptr);
}
int gcnhip_adam_step_advance(gcnhip_ctx *c, const gcnhip_adam_var *vars, int n_vars, float step_size,
                             const float *d_step_sizes, const uint32_t *d_epoch,
                             float beta1, float beta2, float eps, float weight_decay, float *d_sumsq,
                             uint32_t *d_epoch_counter, uint32_t *d_epoch_done) {
    if (!d_epoch_counter) return -1;
    return adam_step_impl(c, vars, n_vars, step_size, d_step_sizes, d_epoch, beta1, beta2, eps, weight_decay, d_sumsq, d_epoch_counter, d_epoch_done);
}
int gcnhip_counter_add(gcnhip_ctx *c, uint32_t *d_counter, uint32_t inc) {
    if (!c || !d_counter) return -1;
    counter_add_kernel<<<1, 1, 0, c->stream>>>(d_counter, inc);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}
int gcnhip_metrics_record_with_next_loss(gcnhip_ctx *c, float *d_ring, int capacity, int slot_in_row,
                                         const uint32_t *d_epoch, const float *d_sumsq) {
    if (!c || !d_ring || capacity <= 0 || slot_in_row < 0 || slot_in_row > 3) return -1;
    c->rec_armed = true;
    c->rec_ring = d_ring; c->rec_capacity = capacity; c->rec_slot = slot_in_row; c->rec_epoch = d_epoch; c->rec_sumsq = d_sumsq;
    return 0;
}
int gcnhip_metrics_record(gcnhip_ctx *c, float *d_ring, int capacity, int slot_in_row,
                          const uint32_t *d_epoch, const float *d_result, const int32_t *d_result_i,
                          const float *d_sumsq) {
    if (!c || !d_ring || capacity <= 0 || !d_result || slot_in_row < 0 || slot_in_row > 3) return -1;
    metrics_record_kernel<<<1, 1, 0, c->stream>>>(d_ring, capacity, slot_in_row, d_epoch, d_result, d_result_i, d_sumsq);
    GCNHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

GCNHIP_DEFINE_PRELOAD(elementwise, adam_kernel)

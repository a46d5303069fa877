// gather_peak.hip — what can a row gather deliver on THIS chip, for THIS table and THIS index stream?
//
// The aggregation (csrc/graphsum.hip; reference: src/seq/module.cpp:83-119) is bound by gathering one d-float row per edge.
// For a table that sits in the Infinity Cache (reddit-syn: 119 MB at d = 128, 45 MB at 48-float rows) no datasheet number is
// its ceiling, and until round 3 bench.py priced the kernel against a figure assembled from MI355X_MICROARCH.md (8.6 TB/s,
// which the guide itself calls the best rate on record for a 38 MB uniformly random gather, not a limit).  This tool
// MEASURES the ceiling: a kernel with graphsum_vec_kernel's memory behaviour and nothing else —
//   * one wave per task (row, or <= 1024-edge segment), 4 tasks per workgroup, tasks dealt to the 8 XCDs in contiguous
//     equal-work ranges, 256-byte column slices bound to XCD groups (blockIdx % 8) exactly as the product kernel does;
//   * 64 indices per coalesced load, handed to lane groups by shuffle; U row loads of 16 bytes per lane in flight;
//   * NO coefficient stream, NO multiply, no epilogue: the only arithmetic is the sum that keeps the loads alive, and the
//     result row is stored only when `store` is set;
// swept over U, resident waves per SIMD (capped with a dummy LDS allocation) and the index stream: the caller passes the
// product's own task list and index array (reddit-syn in its label-major schedule) or a uniformly random one.
// Built as a small shared library driven by tools/gather_peak.py (ctypes); nothing here is part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "gather_peak: %s at %d\n", hipGetErrorString(_e), __LINE__); return -1; } } while (0)

struct GpArgs {
    const int2 *tasks;       // {e_begin, e_end}
    const int *task_row;
    const int *indices;
    const float *table;
    float *out;
    int ld, dim, n_slices, store;
    int bounds[9];
    // what the product adds to the pure gather, one at a time: coef_mode 1 = a coefficient per edge from its own array (a second
    // coalesced 4-byte load per chunk) and a multiply; 2 = (index, coefficient) interleaved in one array of 8-byte pairs
    int coef_mode;
    const float *coef;
    const int2 *pairs;
};

template <int U>
__global__ __launch_bounds__(256) void gather_peak_kernel(GpArgs a) {
    extern __shared__ float pad[];                      // occupancy cap only
    constexpr int L = 16, G = 4;
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int cslice = a.n_slices > 1 ? xcd % a.n_slices : 0;
    const int g_id = a.n_slices > 1 ? xcd / a.n_slices : xcd;
    const int t = a.bounds[g_id] + q * 4 + (threadIdx.x >> 6);
    if (t >= a.bounds[g_id + 1]) return;
    const int2 tk = a.tasks[t];
    const int g = lane / L, l = lane % L;
    const int col0 = (cslice * L + l) * 4;
    const float *in = a.table + (col0 < a.dim ? col0 : 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = tk.x; base < tk.y; base += 64) {
        const int cnt = min(64, tk.y - base);
        int my_idx;
        float my_c = 1.f;
        if (a.coef_mode == 2) {
            const int2 pr = a.pairs[base + (lane < cnt ? lane : 0)];
            my_idx = pr.x; my_c = __int_as_float(pr.y);
        } else {
            my_idx = lane < cnt ? a.indices[base + lane] : a.indices[base];
            if (a.coef_mode == 1) my_c = a.coef[base + (lane < cnt ? lane : 0)];
        }
        const int iters = (cnt + G - 1) / G;
        for (int k = 0; k < iters; k += U) {            // idle lanes re-read the chunk's first row (the product kernel's tail does the same)
            float4 v[U];
            float cc[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int src = (k + u) * G + g;
                const int j = __shfl(my_idx, src < cnt ? src : 0, 64);
                cc[u] = a.coef_mode ? __shfl(my_c, src < cnt ? src : 0, 64) : 1.f;
                v[u] = *reinterpret_cast<const float4 *>(in + (size_t)j * a.ld);
            }
#pragma unroll
            for (int u = 0; u < U; u++) { acc.x += cc[u] * v[u].x; acc.y += cc[u] * v[u].y; acc.z += cc[u] * v[u].z; acc.w += cc[u] * v[u].w; }
        }
    }
#pragma unroll
    for (int m = L; m < 64; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
        acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
    }
    // the sum must be observable; with store == 0 the branch is never taken at run time (finite inputs), so nothing is written
    if (g == 0 && col0 < a.dim && (a.store || acc.x != acc.x))
        *reinterpret_cast<float4 *>(a.out + (size_t)a.task_row[t] * a.ld + col0) = acc;
    if (threadIdx.x == 9999) pad[0] = acc.x;            // (never true: keeps the allocation referenced)
}


// ---- round 6: hub rows served from LDS (verdict r05 item 1: north_star's "LDS staging of the feature tile") -----------------
// Persistent workgroups of NW waves; workgroup w of XCD x owns chunk (x / n_slices) * W + w of the task list (contiguous,
// equal work) at column slice x % n_slices, fills its LDS ONCE with the H rows its chunk's edges reference most (the chunk's
// own hot list, made on the host) and then lets its waves draw tasks from an LDS counter in task order.  A task's first
// `nl` edges (a multiple of 4: whole rounds of the four lane groups) name LDS slots, the rest name table rows: lane group g
// still takes edges g, g + 4, ... of the task in order, so the sum has the bits of the plain kernel on the same edge order.
struct GlArgs {
    const int4 *tasks;       // {e_begin, e_end, nl, row}
    const int *indices;      // first nl entries of a task: LDS slot; the rest: table row
    const int *chunk_bounds; // [n_chunks + 1] task ranges
    const int *hot;          // [n_chunks][H] table rows held in LDS (-1: empty slot)
    const float *table;
    float *out;
    int ld, dim, n_slices, store, W, H;
    // ordered != 0: every workgroup of an XCD draws batches of `ordered` tasks from ONE counter per XCD, in task order (the
    // XCD's window of active rows stays as tight as with one wave per task); the hot list is then the XCD group's (W_hot = 1)
    int ordered;
    int *counters;           // [8], zero before the launch
};

template <int U, int NW, int L>
__global__ __launch_bounds__(NW * 64) void gather_lds_kernel(GlArgs a) {
    extern __shared__ float4 hub[];                     // H rows of L float4, then the task counter
    int &next_task = *reinterpret_cast<int *>(&hub[a.H * L]);
    constexpr int G = 64 / L;
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    const int cslice = xcd % a.n_slices, g_id = xcd / a.n_slices;
    const int chunk = a.ordered ? g_id : g_id * a.W + w;
    {
        const int *hot = a.hot + (size_t)chunk * a.H;
        for (int i = threadIdx.x; i < a.H * L; i += NW * 64) {
            const int r = i / L, p = i % L;
            const int j = hot[r];
            const int c = (cslice * L + p) * 4;
            hub[i] = (j >= 0 && c < a.ld) ? *reinterpret_cast<const float4 *>(a.table + (size_t)j * a.ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (threadIdx.x == 0) next_task = a.chunk_bounds[chunk];
    }
    __syncthreads();
    const int t_end = a.chunk_bounds[chunk + 1];
    const int g = lane / L, l = lane % L;
    const int col0 = (cslice * L + l) * 4;
    const float *in = a.table + (col0 < a.ld ? col0 : 0);
    int t_batch = 0, t_left = 0;
    for (;;) {
        int t = 0;
        if (a.ordered) {
            if (t_left == 0) {
                if (lane == 0) t_batch = a.chunk_bounds[chunk] + atomicAdd(&a.counters[xcd], a.ordered);
                t_batch = __builtin_amdgcn_readfirstlane(t_batch);
                t_left = a.ordered;
            }
            t = t_batch++; t_left--;
        } else {
            if (lane == 0) t = atomicAdd(&next_task, 1);
            t = __builtin_amdgcn_readfirstlane(t);
        }
        if (t >= t_end) break;
        const int4 tk = a.tasks[t];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int e_mid = tk.x + tk.z;
        for (int base = tk.x; base < e_mid; base += 64) {        // LDS part: whole rounds of the lane groups, no padding
            const int cnt = min(64, e_mid - base);
            const int my_idx = a.indices[base + (lane < cnt ? lane : 0)];
            const int iters = cnt / G;
            for (int k = 0; k < iters; k += U) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int src = (k + u) * G + g;
                    const int sl = __shfl(my_idx, src & 63, 64);
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k + u < iters) v[u] = hub[sl * L + l];       // wave-uniform
                }
#pragma unroll
                for (int u = 0; u < U; u++) if (k + u < iters) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
        for (int base = e_mid; base < tk.y; base += 64) {
            const int cnt = min(64, tk.y - base);
            const int my_idx = lane < cnt ? a.indices[base + lane] : a.indices[base];
            const int iters = (cnt + G - 1) / G;
            for (int k = 0; k < iters; k += U) {
                float4 v[U];
                bool on[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int src = (k + u) * G + g;
                    on[u] = src < cnt;
                    const int j = __shfl(my_idx, on[u] ? src : 0, 64);
                    v[u] = *reinterpret_cast<const float4 *>(in + (size_t)j * a.ld);
                }
#pragma unroll
                for (int u = 0; u < U; u++) if (on[u]) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
#pragma unroll
        for (int m = L; m < 64; m <<= 1) {
            acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
            acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
        }
        if (g == 0 && col0 < a.dim && (a.store || acc.x != acc.x))
            *reinterpret_cast<float4 *>(a.out + (size_t)t * a.ld + col0) = acc;   // by TASK: segments of a split row must not race in the check
    }
}

// the plain kernel's memory behaviour on the SAME tasks and edge order (table rows everywhere): the checker of the LDS form
// and its like-for-like baseline.  One wave per task, 4 per workgroup, XCD groups take the chunk ranges of their group.
template <int U, int L>
__global__ __launch_bounds__(256) void gather_plain2_kernel(GlArgs a, const int *cols) {
    constexpr int G = 64 / L;
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int cslice = xcd % a.n_slices, g_id = xcd / a.n_slices;
    const int t = a.chunk_bounds[g_id * a.W] + q * 4 + (threadIdx.x >> 6);
    if (t >= a.chunk_bounds[(g_id + 1) * a.W]) return;
    const int4 tk = a.tasks[t];
    const int g = lane / L, l = lane % L;
    const int col0 = (cslice * L + l) * 4;
    const float *in = a.table + (col0 < a.ld ? col0 : 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = tk.x; base < tk.y; base += 64) {
        const int cnt = min(64, tk.y - base);
        const int my_idx = lane < cnt ? cols[base + lane] : cols[base];
        const int iters = (cnt + G - 1) / G;
        for (int k = 0; k < iters; k += U) {
            float4 v[U];
            bool on[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int src = (k + u) * G + g;
                on[u] = src < cnt;
                const int j = __shfl(my_idx, on[u] ? src : 0, 64);
                v[u] = *reinterpret_cast<const float4 *>(in + (size_t)j * a.ld);
            }
#pragma unroll
            for (int u = 0; u < U; u++) if (on[u]) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
#pragma unroll
    for (int m = L; m < 64; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
        acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
    }
    if (g == 0 && col0 < a.dim && (a.store || acc.x != acc.x))
        *reinterpret_cast<float4 *>(a.out + (size_t)t * a.ld + col0) = acc;   // by TASK: segments of a split row must not race in the check
}

struct LdsHandle {
    int4 *tasks = nullptr; int *indices = nullptr, *cols = nullptr, *chunk_bounds = nullptr, *hot = nullptr;
    float *table = nullptr, *out = nullptr, *out_ref = nullptr; int *counters = nullptr;
    int n_tasks = 0, n_chunks = 0, H = 0, W = 0, groups = 0; long nnz = 0; size_t table_floats = 0;
    std::vector<int> h_bounds;
};

struct Handle {
    int2 *tasks = nullptr; int *task_row = nullptr; int *indices = nullptr; float *table = nullptr, *out = nullptr;
    float *coef = nullptr; int2 *pairs = nullptr;
    int n_tasks = 0; long nnz = 0; size_t table_floats = 0;
    std::vector<int2> h_tasks;
    int n_cu = 256;
};

template <int U, int L>
static int gl_launch(LdsHandle *h, GlArgs a, int mode, int NW, int h_used, int wgs) {
    if (mode == 0) {
        int max_blocks = 1;
        for (int g = 0; g < h->groups; g++) max_blocks = std::max(max_blocks, (h->h_bounds[(g + 1) * h->W] - h->h_bounds[g * h->W] + 3) / 4);
        hipLaunchKernelGGL((gather_plain2_kernel<U, L>), dim3(max_blocks * 8), dim3(256), 0, 0, a, h->cols);
        return 0;
    }
    const size_t lds = (size_t)h_used * L * 16 + 16;
    if (a.ordered) CK(hipMemsetAsync(h->counters, 0, 64, 0));
    if (NW == 16) {
        CK(hipFuncSetAttribute((const void *)gather_lds_kernel<U, 16, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gather_lds_kernel<U, 16, L>), dim3((a.ordered ? wgs : h->W) * 8), dim3(1024), lds, 0, a);
    } else {
        CK(hipFuncSetAttribute((const void *)gather_lds_kernel<U, 8, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gather_lds_kernel<U, 8, L>), dim3((a.ordered ? wgs : h->W) * 8), dim3(512), lds, 0, a);
    }
    return 0;
}

extern "C" {

// tasks in the order the product would run them; indices as the product stores them
int gp_create(void **out, const int *h_e0, const int *h_e1, const int *h_row, int n_tasks, const int *h_indices, long nnz, long table_floats) {
    Handle *h = new Handle();
    h->n_tasks = n_tasks; h->nnz = nnz; h->table_floats = (size_t)table_floats;
    h->h_tasks.resize(n_tasks);
    for (int i = 0; i < n_tasks; i++) h->h_tasks[i] = make_int2(h_e0[i], h_e1[i]);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    h->n_cu = prop.multiProcessorCount;
    CK(hipMalloc((void **)&h->tasks, (size_t)n_tasks * sizeof(int2)));
    CK(hipMalloc((void **)&h->task_row, (size_t)n_tasks * sizeof(int)));
    CK(hipMalloc((void **)&h->indices, (size_t)nnz * sizeof(int)));
    CK(hipMalloc((void **)&h->table, h->table_floats * sizeof(float)));
    CK(hipMalloc((void **)&h->out, h->table_floats * sizeof(float)));
    CK(hipMemcpy(h->tasks, h->h_tasks.data(), (size_t)n_tasks * sizeof(int2), hipMemcpyHostToDevice));
    CK(hipMemcpy(h->task_row, h_row, (size_t)n_tasks * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(h->indices, h_indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
    {   // coefficients (all 1: the sums stay checkable) in their own array and interleaved with the indices
        std::vector<float> cf((size_t)nnz, 1.0f);
        std::vector<int2> pr((size_t)nnz);
        for (long e = 0; e < nnz; e++) pr[e] = make_int2(h_indices[e], 0x3F800000);
        CK(hipMalloc((void **)&h->coef, (size_t)nnz * sizeof(float)));
        CK(hipMalloc((void **)&h->pairs, (size_t)nnz * sizeof(int2)));
        CK(hipMemcpy(h->coef, cf.data(), (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
        CK(hipMemcpy(h->pairs, pr.data(), (size_t)nnz * sizeof(int2), hipMemcpyHostToDevice));
    }
    std::vector<float> ones(h->table_floats, 1.0f);
    CK(hipMemcpy(h->table, ones.data(), h->table_floats * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemset(h->out, 0, h->table_floats * sizeof(float)));
    *out = h;
    return 0;
}

int gp_destroy(void *p) {
    Handle *h = (Handle *)p;
    if (!h) return 0;
    hipFree(h->tasks); hipFree(h->task_row); hipFree(h->indices); hipFree(h->table); hipFree(h->out); hipFree(h->coef); hipFree(h->pairs);
    delete h;
    return 0;
}

// one configuration: average launch time over `iters` launches (after 2 warm-up launches), HIP events on the null stream
int gp_run2(void *p, int ld, int dim, int U, int waves_per_simd, int store, int iters, int coef_mode, float *ms_out);
int gp_run(void *p, int ld, int dim, int U, int waves_per_simd, int store, int iters, float *ms_out) {
    return gp_run2(p, ld, dim, U, waves_per_simd, store, iters, 0, ms_out);
}
int gp_run2(void *p, int ld, int dim, int U, int waves_per_simd, int store, int iters, int coef_mode, float *ms_out) {
    Handle *h = (Handle *)p;
    if (!h || (size_t)ld * 1 > h->table_floats) return -1;
    GpArgs a;
    a.tasks = h->tasks; a.task_row = h->task_row; a.indices = h->indices; a.table = h->table; a.out = h->out;
    a.ld = ld; a.dim = dim; a.store = store;
    a.coef_mode = coef_mode; a.coef = h->coef; a.pairs = h->pairs;
    const int ychunks = (dim + 63) / 64;
    a.n_slices = (ychunks > 1 && 8 % ychunks == 0) ? ychunks : 1;
    if (ychunks > 1 && a.n_slices == 1) return -1;      // only the shapes the product slices (d = 128, 256) or single-slice rows
    const int groups = 8 / a.n_slices;
    // equal-work contiguous task ranges per XCD group, each starting on a multiple of 4 tasks (csrc/ctx.hip, xcd_bounds)
    std::vector<int64_t> prefix((size_t)h->n_tasks + 1, 0);
    for (int t = 0; t < h->n_tasks; t++) prefix[t + 1] = prefix[t] + (h->h_tasks[t].y - h->h_tasks[t].x) + 8;
    a.bounds[0] = 0;
    for (int k = 1; k < groups; k++) {
        int t = (int)(std::lower_bound(prefix.begin(), prefix.end(), prefix[h->n_tasks] * k / groups) - prefix.begin());
        t = std::min(h->n_tasks, (t + 3) / 4 * 4);
        a.bounds[k] = std::max(t, a.bounds[k - 1]);
    }
    for (int k = groups; k <= 8; k++) a.bounds[k] = h->n_tasks;
    int max_blocks = 1;
    for (int k = 0; k < groups; k++) max_blocks = std::max(max_blocks, (a.bounds[k + 1] - a.bounds[k] + 3) / 4);
    const dim3 grid(max_blocks * 8);
    size_t lds = 0;
    if (waves_per_simd > 0 && waves_per_simd < 8) lds = (size_t)(160 * 1024 / waves_per_simd) - 1024;
    auto launch = [&]() {
        switch (U) {
            case 1: hipLaunchKernelGGL(gather_peak_kernel<1>, grid, dim3(256), lds, 0, a); break;
            case 2: hipLaunchKernelGGL(gather_peak_kernel<2>, grid, dim3(256), lds, 0, a); break;
            case 4: hipLaunchKernelGGL(gather_peak_kernel<4>, grid, dim3(256), lds, 0, a); break;
            default: hipLaunchKernelGGL(gather_peak_kernel<8>, grid, dim3(256), lds, 0, a); break;
        }
    };
    if (lds > 64 * 1024) {
        CK(hipFuncSetAttribute((const void *)gather_peak_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void *)gather_peak_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void *)gather_peak_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void *)gather_peak_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms_out = ms / iters;
    return 0;
}


// LDS form: tasks {e0, e1, nl, row}; `indices` slots-then-rows, `cols` rows everywhere (same edge order); chunk_bounds
// [groups * W + 1]; hot [groups * W][H]; the table is filled with a fixed pseudo-random pattern so that sums are checkable
int gl_create(void **out, const int *h_tasks4, int n_tasks, const int *h_indices, const int *h_cols, long nnz,
              const int *h_chunk_bounds, const int *h_hot, int groups, int W, int H, long table_floats) {
    LdsHandle *h = new LdsHandle();
    h->n_tasks = n_tasks; h->nnz = nnz; h->groups = groups; h->W = W; h->H = H; h->n_chunks = groups * W; h->table_floats = (size_t)table_floats;
    h->h_bounds.assign(h_chunk_bounds, h_chunk_bounds + h->n_chunks + 1);
    CK(hipMalloc((void **)&h->tasks, (size_t)n_tasks * sizeof(int4)));
    CK(hipMalloc((void **)&h->indices, (size_t)nnz * sizeof(int)));
    CK(hipMalloc((void **)&h->cols, (size_t)nnz * sizeof(int)));
    CK(hipMalloc((void **)&h->chunk_bounds, (size_t)(h->n_chunks + 1) * sizeof(int)));
    CK(hipMalloc((void **)&h->hot, (size_t)h->n_chunks * H * sizeof(int)));
    CK(hipMalloc((void **)&h->table, h->table_floats * sizeof(float)));
    CK(hipMalloc((void **)&h->out, h->table_floats * sizeof(float)));
    CK(hipMalloc((void **)&h->out_ref, h->table_floats * sizeof(float)));
    CK(hipMalloc((void **)&h->counters, 64));
    CK(hipMemcpy(h->tasks, h_tasks4, (size_t)n_tasks * sizeof(int4), hipMemcpyHostToDevice));
    CK(hipMemcpy(h->indices, h_indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(h->cols, h_cols, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(h->chunk_bounds, h_chunk_bounds, (size_t)(h->n_chunks + 1) * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(h->hot, h_hot, (size_t)h->n_chunks * H * sizeof(int), hipMemcpyHostToDevice));
    std::vector<float> tb(h->table_floats);
    uint32_t x = 12345u;
    for (size_t i = 0; i < tb.size(); i++) { x = x * 1664525u + 1013904223u; tb[i] = (float)((x >> 8) & 0xFFFF) / 65536.f - 0.5f; }
    CK(hipMemcpy(h->table, tb.data(), h->table_floats * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemset(h->out, 0, h->table_floats * sizeof(float)));
    CK(hipMemset(h->out_ref, 0, h->table_floats * sizeof(float)));
    *out = h;
    return 0;
}
int gl_destroy(void *p) {
    LdsHandle *h = (LdsHandle *)p;
    if (!h) return 0;
    hipFree(h->tasks); hipFree(h->indices); hipFree(h->cols); hipFree(h->chunk_bounds); hipFree(h->hot); hipFree(h->table); hipFree(h->out); hipFree(h->out_ref); hipFree(h->counters);
    delete h;
    return 0;
}
// mode 0: plain kernel on the same tasks / edge order; 1: LDS form.  waves: NW (8 or 16).  check: compare `out` of the LDS
// form with the plain kernel's bit for bit (returns the number of differing floats in *n_diff)
int gl_run2(void *p, int ld, int dim, int lanes, int U, int NW, int mode, int store, int iters, int check, int ordered, int wgs, float *ms_out, long *n_diff);
int gl_run(void *p, int ld, int dim, int lanes, int U, int NW, int mode, int store, int iters, int check, float *ms_out, long *n_diff) {
    return gl_run2(p, ld, dim, lanes, U, NW, mode, store, iters, check, 0, 0, ms_out, n_diff);
}
// ordered > 0 (mode 1 only): tasks drawn in order from one counter per XCD in batches of `ordered`, `wgs` workgroups per XCD; the handle was made with W = 1
int gl_run2(void *p, int ld, int dim, int lanes, int U, int NW, int mode, int store, int iters, int check, int ordered, int wgs, float *ms_out, long *n_diff) {
    LdsHandle *h = (LdsHandle *)p;
    if (!h) return -1;
    GlArgs a;
    a.tasks = h->tasks; a.indices = h->indices; a.chunk_bounds = h->chunk_bounds; a.hot = h->hot; a.table = h->table; a.out = h->out;
    a.ld = ld; a.dim = dim; a.store = store; a.W = h->W; a.H = h->H;
    a.ordered = mode == 1 ? ordered : 0; a.counters = h->counters;
    if (a.ordered && h->W != 1) return -1;
    a.n_slices = 8 / h->groups;
    if ((lanes != 16 && lanes != 8) || a.n_slices * lanes * 4 < dim) return -1;
    auto launch = [&](GlArgs aa, int m) -> int {
        if (lanes == 16) {
            if (U == 2) return gl_launch<2, 16>(h, aa, m, NW, h->H, wgs);
            if (U == 8) return gl_launch<8, 16>(h, aa, m, NW, h->H, wgs);
            return gl_launch<4, 16>(h, aa, m, NW, h->H, wgs);
        }
        if (U == 2) return gl_launch<2, 8>(h, aa, m, NW, h->H, wgs);
        if (U == 8) return gl_launch<8, 8>(h, aa, m, NW, h->H, wgs);
        return gl_launch<4, 8>(h, aa, m, NW, h->H, wgs);
    };
    if (check) {
        GlArgs r = a; r.out = h->out_ref; r.store = 1;
        GlArgs t = a; t.store = 1;
        if (launch(r, 0) || launch(t, 1)) return -1;
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        std::vector<uint32_t> x(h->table_floats), y(h->table_floats);
        CK(hipMemcpy(x.data(), h->out, h->table_floats * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), h->out_ref, h->table_floats * 4, hipMemcpyDeviceToHost));
        long d = 0, nz = 0;
        for (size_t i = 0; i < x.size(); i++) { d += x[i] != y[i]; nz += y[i] != 0; }
        *n_diff = nz ? d : -1;                         // -1: the reference wrote nothing (a broken check)
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) if (launch(a, mode)) return -1;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) if (launch(a, mode)) return -1;
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms_out = ms / iters;
    return 0;
}

}  // extern "C"

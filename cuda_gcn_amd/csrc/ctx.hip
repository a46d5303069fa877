// ctx.hip — context, memory, prepared adjacency / feature objects, events.
// Host-side preparation happens ONCE per dataset (the reference re-derives
// degrees per edge per call and re-uploads X every epoch).
#include "common.h"
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <thread>

static thread_local char g_detail[256] = "";
int gcnhip_fail(const char *detail) {
    snprintf(g_detail, sizeof g_detail, "%s", detail ? detail : "");
    return -1;
}

const GcnOptionEntry GCN_OPTION_TABLE[] = {
    {"gs_pipe", &GcnOptions::gs_pipe, 0}, {"gs_u", &GcnOptions::gs_u, 0}, {"gs_nt", &GcnOptions::gs_nt, 0}, {"gs_fold", &GcnOptions::gs_fold, 0}, {"gs_l", &GcnOptions::gs_l, 0},
    {"gemm_tiles", &GcnOptions::gemm_tiles, 0}, {"gemm_bf16x3", &GcnOptions::gemm_bf16x3, 2}, {"gemm_w4", &GcnOptions::gemm_w4, 0}, {"cls_abl", &GcnOptions::cls_abl, 0}, {"cls_wgs", &GcnOptions::cls_wgs, 0}, {"cls_fwd", &GcnOptions::cls_fwd, 1}, {"spmm_slices", &GcnOptions::spmm_slices, 1}, {"gemm_lane_waves", &GcnOptions::gemm_lane_waves, 8}, {"gemm_lane_wgs", &GcnOptions::gemm_lane_wgs, 0}, {"gemm_persist_bwd", &GcnOptions::gemm_persist_bwd, 0},
    {"dbg_linear", &GcnOptions::dbg_linear, 0}, {"xent_finalize", &GcnOptions::xent_finalize, 0}, {"xent_wave", &GcnOptions::xent_wave, 0},
    {"adam_sum_launch", &GcnOptions::adam_sum_launch, 0}, {"atb_cap_mb", &GcnOptions::atb_cap_mb, 12}, {"rs_wgs", &GcnOptions::rs_wgs, 0},
    {"spmm_lds", &GcnOptions::spmm_lds, 0}, {"spmm_rows", &GcnOptions::spmm_rows, 0}, {"spmm_general", &GcnOptions::spmm_general, 0}, {"spmm_nw", &GcnOptions::spmm_nw, 0}, {"split_edges", &GcnOptions::split_edges, 0},
};
const int GCN_OPTION_COUNT = (int)(sizeof GCN_OPTION_TABLE / sizeof GCN_OPTION_TABLE[0]);

// defaults, then GCNHIP_<NAME> from the environment (a set but empty or non-numeric variable means 1): once, here
static void options_from_environment(GcnOptions *o) {
    for (int i = 0; i < GCN_OPTION_COUNT; i++) {
        const GcnOptionEntry &e = GCN_OPTION_TABLE[i];
        o->*e.field = e.dflt;
        char var[64] = "GCNHIP_";
        size_t n = strlen(var);
        for (const char *q = e.name; *q && n + 1 < sizeof var; q++) var[n++] = (char)toupper((unsigned char)*q);
        var[n] = 0;
        if (const char *v = getenv(var)) {
            char *end;
            const long x = strtol(v, &end, 10);
            o->*e.field = end == v ? 1 : (int)x;
        }
    }
}

extern "C" {

const char *gcnhip_last_error(void) { return g_detail; }

int gcnhip_ctx_set_option(gcnhip_ctx *c, const char *name, int value) {
    if (!c || !name) return -1;
    for (int i = 0; i < GCN_OPTION_COUNT; i++)
        if (!strcmp(GCN_OPTION_TABLE[i].name, name)) { c->opt.*GCN_OPTION_TABLE[i].field = value; return 0; }
    return gcnhip_fail("gcnhip_ctx_set_option: unknown option");
}
int gcnhip_ctx_get_option(const gcnhip_ctx *c, const char *name, int *value) {
    if (!c || !name || !value) return -1;
    for (int i = 0; i < GCN_OPTION_COUNT; i++)
        if (!strcmp(GCN_OPTION_TABLE[i].name, name)) { *value = c->opt.*GCN_OPTION_TABLE[i].field; return 0; }
    return gcnhip_fail("gcnhip_ctx_get_option: unknown option");
}

int gcnhip_device_count(int *count) {
    GCNHIP_TRY(hipGetDeviceCount(count));
    return 0;
}

const char *gcnhip_error_string(int code) {
    if (code == -1) return "gcnhip: invalid argument";
    if (code == GCNHIP_NOT_AVAILABLE) return "gcnhip: this fused form is not available for these shapes / options (nothing was launched)";
    return hipGetErrorString((hipError_t)code);
}

const char *gcnhip_version(void) { return "gcnhip 0.1 (gfx950)"; }
int gcnhip_experiments(void) {
#ifdef GCNHIP_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int gcnhip_ctx_create(gcnhip_ctx **out, int device, void *stream) {
    if (!out) return -1;
    GCNHIP_TRY(hipSetDevice(device));
    gcnhip_ctx *c = new gcnhip_ctx();
    memset(c, 0, sizeof *c);
    c->device = device;
    options_from_environment(&c->opt);
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return (int)e; }
        c->own_stream = true;
    }
    auto fill = [&]() -> int {
        {   // code objects are loaded per device; gcn-hip's worker threads create their contexts at the same time
            static std::mutex mu;
            static bool preloaded[64] = {};
            std::lock_guard<std::mutex> lock(mu);
            if (device >= 0 && device < 64 && !preloaded[device]) {
                GCNHIP_TRY((hipError_t)gcnhip_preload_elementwise());
                GCNHIP_TRY((hipError_t)gcnhip_preload_graphsum());
                GCNHIP_TRY((hipError_t)gcnhip_preload_matmul());
                GCNHIP_TRY((hipError_t)gcnhip_preload_spmm());
                GCNHIP_TRY((hipError_t)gcnhip_preload_xent());
                preloaded[device] = true;
            }
        }
        hipDeviceProp_t prop;
        GCNHIP_TRY(hipGetDeviceProperties(&prop, device));
        c->n_cu = prop.multiProcessorCount;
        GCNHIP_TRY(hipMalloc((void **)&c->red_f, RED_SLOTS * 4 * sizeof(float)));
        GCNHIP_TRY(hipMalloc((void **)&c->red_i, RED_SLOTS * 4 * sizeof(int32_t)));
        GCNHIP_TRY(hipMalloc((void **)&c->ticket, 64 * sizeof(uint32_t)));
        GCNHIP_TRY(hipMemset(c->ticket, 0, 64 * sizeof(uint32_t)));
        GCNHIP_TRY(hipMalloc((void **)&c->wpack, WPACK_BYTES));
        c->wpack_bytes = WPACK_BYTES;
        return 0;
    };
    const int rc = fill();
    if (rc != 0) { gcnhip_ctx_destroy(c); return rc; }        // frees whatever was allocated
    *out = c;
    return 0;
}

int gcnhip_ctx_destroy(gcnhip_ctx *c) {
    if (!c) return 0;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->red_f) hipFree(c->red_f);
    if (c->red_i) hipFree(c->red_i);
    if (c->ticket) hipFree(c->ticket);
    if (c->slab) hipFree(c->slab);
    if (c->wpack) hipFree(c->wpack);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int gcnhip_ctx_set_corun(gcnhip_ctx *c, int on) {
    if (!c) return -1;
    c->corun = on ? 1 : 0;
    return 0;
}

int gcnhip_ctx_sync(gcnhip_ctx *c) { GCNHIP_TRY(hipStreamSynchronize(c->stream)); return 0; }
void *gcnhip_ctx_stream(gcnhip_ctx *c) { return (void *)c->stream; }

int gcnhip_malloc(gcnhip_ctx *c, void **ptr, size_t bytes) {
    GCNHIP_TRY(hipSetDevice(c->device));
    GCNHIP_TRY(hipMalloc(ptr, bytes ? bytes : 16));
    return 0;
}
int gcnhip_free(gcnhip_ctx *c, void *ptr) {
    if (!ptr) return 0;
    GCNHIP_TRY(hipSetDevice(c->device));
    GCNHIP_TRY(hipFree(ptr));
    return 0;
}
int gcnhip_memset_async(gcnhip_ctx *c, void *ptr, int byte, size_t bytes) {
    if (bytes) GCNHIP_TRY(hipMemsetAsync(ptr, byte, bytes, c->stream));
    return 0;
}
int gcnhip_h2d(gcnhip_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes) GCNHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    GCNHIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int gcnhip_d2h(gcnhip_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes) GCNHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    GCNHIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int gcnhip_d2d_async(gcnhip_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes) GCNHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int gcnhip_host_alloc(void **ptr, size_t bytes) {
    if (!ptr) return -1;
    GCNHIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return 0;
}
int gcnhip_host_free(void *ptr) {
    if (ptr) GCNHIP_TRY(hipHostFree(ptr));
    return 0;
}
int gcnhip_d2h_async(gcnhip_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c || !dst || !src) return -1;
    if (bytes) GCNHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return 0;
}

int gcnhip_capture_begin(gcnhip_ctx *c) {
    GCNHIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    return 0;
}
int gcnhip_capture_end(gcnhip_ctx *c, void **graph_exec) {
    hipGraph_t graph = nullptr;
    GCNHIP_TRY(hipStreamEndCapture(c->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return (int)e;
    *graph_exec = (void *)exec;
    return 0;
}
int gcnhip_graph_launch(gcnhip_ctx *c, void *graph_exec) {
    GCNHIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, c->stream));
    return 0;
}
int gcnhip_graph_exec_destroy(void *graph_exec) {
    if (graph_exec) GCNHIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

int gcnhip_event_create(void **ev) { GCNHIP_TRY(hipEventCreate((hipEvent_t *)ev)); return 0; }
int gcnhip_event_create_sync(void **ev) { GCNHIP_TRY(hipEventCreateWithFlags((hipEvent_t *)ev, hipEventDisableTiming)); return 0; }
int gcnhip_event_destroy(void *ev) { GCNHIP_TRY(hipEventDestroy((hipEvent_t)ev)); return 0; }
int gcnhip_event_record(gcnhip_ctx *c, void *ev) { GCNHIP_TRY(hipEventRecord((hipEvent_t)ev, c->stream)); return 0; }
int gcnhip_stream_wait_event(gcnhip_ctx *c, void *ev) {
    GCNHIP_TRY(hipStreamWaitEvent(c->stream, (hipEvent_t)ev, 0));
    return 0;
}
int gcnhip_event_sync(void *ev) { GCNHIP_TRY(hipEventSynchronize((hipEvent_t)ev)); return 0; }
int gcnhip_event_elapsed_ms(void *start, void *stop, float *ms) {
    GCNHIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    GCNHIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- adjacency
// coef(e) for every edge, computed once.  One thread per row walks its edges
// (one-time cost; the per-call kernels then stream coef[] coalesced).
__global__ void edge_coef_kernel(const int *__restrict__ indptr, const int *__restrict__ indices,
                                 const int *__restrict__ col_deg, float *__restrict__ coef, int n_rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int e0 = indptr[r], e1 = indptr[r + 1];
    const int64_t ds = e1 - e0;
    for (int e = e0; e < e1; e++) {
        const int d = indices[e];
        const int64_t dd = col_deg ? col_deg[d] : (indptr[d + 1] - indptr[d]);
        // module.cpp:91-93: float sqrtf of the integer product, divide in double, narrow
        coef[e] = (float)(1.0 / (double)sqrtf((float)(ds * dd)));
    }
}

// Rows longer than the split length are cut into segments (one wave each).  A segment is a serial walk, so it
// must stay short against the work one wave slot gets: nnz / (256 CUs x 32 waves), clamped to [128, 1024]
// (a full Reddit graph keeps 1024; a 1/8 row block of it gets 256, which removed a 50 us critical path
// from a 130 us launch).
static int split_length(int64_t nnz, int forced) {
    if (forced >= 16) return forced;                 // the split_edges option of the creating context (experiments)
    int s = 1024;      // round 3 sweep: 512 is 1 % better on reddit-syn's hidden width (0.766 -> 0.756 ms, epoch +0.4 %) and 5 % worse on
                       // the R-MAT scale-22 model (24.5 vs 25.8 epochs/s: ten times the segments, all through the partial scratch); 256 and
                       // 2048+ lose on both.  1024 stays.
    while (s > 128 && (int64_t)s * 8192 > nnz) s >>= 1;
    return s;
}

// equal-work task ranges for 1/2/4/8 XCD groups, each starting on a multiple of 4 tasks (one workgroup)
static void xcd_bounds(const std::vector<int4> &tasks, int bounds[4][9]) {
    const int n_units = (int)tasks.size();
    std::vector<int64_t> prefix((size_t)n_units + 1);   // work before task t: edges + a per-task constant
    prefix[0] = 0;
    for (int t = 0; t < n_units; t++) prefix[t + 1] = prefix[t] + (tasks[t].z - tasks[t].y) + 8;
    for (int lg = 0; lg < 4; lg++) {
        const int G = 1 << lg;
        bounds[lg][0] = 0;
        for (int k = 1; k < G; k++) {
            const int64_t target = prefix[n_units] * k / G;
            int t = (int)(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
            t = (t + 3) / 4 * 4;
            if (t > n_units) t = n_units;
            if (t < bounds[lg][k - 1]) t = bounds[lg][k - 1];
            bounds[lg][k] = t;
        }
        for (int k = G; k <= 8; k++) bounds[lg][k] = n_units;
    }
}

// the tasks of the full schedule whose row is in the subset, same order, same segment slots
static int build_rowset(gcnhip_rowset *rs, const std::vector<int4> &tasks, const std::vector<int4> &srows) {
    if (rs->tasks) { GCNHIP_TRY(hipFree(rs->tasks)); rs->tasks = nullptr; }
    if (rs->split_rows) { GCNHIP_TRY(hipFree(rs->split_rows)); rs->split_rows = nullptr; }
    auto in = [&](int r) { return (rs->bits[r >> 5] >> (r & 31)) & 1u; };
    std::vector<int4> t2, s2;
    for (const int4 &t : tasks) if (in(t.x)) t2.push_back(t);
    for (const int4 &sr : srows) if (in(sr.x)) s2.push_back(sr);
    rs->n_tasks = (int)t2.size();
    rs->n_split_rows = (int)s2.size();
    if (!t2.empty()) {
        GCNHIP_TRY(hipMalloc((void **)&rs->tasks, t2.size() * sizeof(int4)));
        GCNHIP_TRY(hipMemcpy(rs->tasks, t2.data(), t2.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    if (!s2.empty()) {
        GCNHIP_TRY(hipMalloc((void **)&rs->split_rows, s2.size() * sizeof(int4)));
        GCNHIP_TRY(hipMemcpy(rs->split_rows, s2.data(), s2.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    xcd_bounds(t2, rs->bounds);
    return 0;
}

// (Re)build the row schedule: tasks ordered by (key[row] ascending, degree descending); key == nullptr: degree only.
static int build_tasks(gcnhip_graph *g, const std::vector<int> &order);
static int build_schedule(gcnhip_graph *g, const int *h_row_group) {
    const int n_rows = g->n_rows;
    const int *h_indptr = g->h_indptr->data();
    if (g->tasks) { GCNHIP_TRY(hipFree(g->tasks)); g->tasks = nullptr; }
    if (g->split_rows) { GCNHIP_TRY(hipFree(g->split_rows)); g->split_rows = nullptr; }
    // Task list: rows in descending degree order (heavy work first, similar rows together), group-major
    // when the caller names communities; a row
    // above SPLIT_EDGES becomes consecutive segments whose partial sums a second kernel adds in order.
    std::vector<int> order(n_rows);
    for (int r = 0; r < n_rows; r++) order[r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (h_row_group && h_row_group[a] != h_row_group[b]) return h_row_group[a] < h_row_group[b];
        return h_indptr[a + 1] - h_indptr[a] > h_indptr[b + 1] - h_indptr[b];
    });
    return build_tasks(g, order);
}

// the task list of a given row order
static int build_tasks(gcnhip_graph *g, const std::vector<int> &order) {
    const int *h_indptr = g->h_indptr->data();
    const int SPLIT_EDGES = split_length(g->nnz, g->split_edges_opt);
    const int n_rows = g->n_rows;
    int n_slots = 0;
    std::vector<int4> tasks, srows;
    tasks.reserve((size_t)n_rows + 64);
    for (int r : order) {
        const int e0 = h_indptr[r], e1 = h_indptr[r + 1];
        if (e1 - e0 <= SPLIT_EDGES) { tasks.push_back(make_int4(r, e0, e1, -1)); continue; }
        const int ns = (e1 - e0 + SPLIT_EDGES - 1) / SPLIT_EDGES;
        srows.push_back(make_int4(r, n_slots, ns, 0));
        for (int q = 0; q < ns; q++)
            tasks.push_back(make_int4(r, e0 + q * SPLIT_EDGES, std::min(e1, e0 + (q + 1) * SPLIT_EDGES), n_slots + q));
        n_slots += ns;
    }
    g->n_tasks = (int)tasks.size();
    g->n_split_rows = (int)srows.size();
    g->n_slots = n_slots;
    const int n_units = g->n_tasks;
    if (n_units) {
        GCNHIP_TRY(hipMalloc((void **)&g->tasks, tasks.size() * sizeof(int4)));
        GCNHIP_TRY(hipMemcpy(g->tasks, tasks.data(), tasks.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    if (!srows.empty()) {
        GCNHIP_TRY(hipMalloc((void **)&g->split_rows, srows.size() * sizeof(int4)));
        GCNHIP_TRY(hipMemcpy(g->split_rows, srows.data(), srows.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    // segment scratch for the widest aggregation this object will serve: sized HERE (and by
    // gcnhip_graph_reserve_width), never inside a launch path
    if (g->partials) { GCNHIP_TRY(hipFree(g->partials)); g->partials = nullptr; }
    if (g->slot_info) { GCNHIP_TRY(hipFree(g->slot_info)); g->slot_info = nullptr; }
    if (g->seg_count) { GCNHIP_TRY(hipFree(g->seg_count)); g->seg_count = nullptr; }
    if (g->part_ld < 256) g->part_ld = 256;
    if (n_slots) {
        GCNHIP_TRY(hipMalloc((void **)&g->partials, (size_t)n_slots * g->part_ld * sizeof(float)));
        std::vector<int2> info((size_t)n_slots);
        for (const int4 &sr : srows)
            for (int q = 0; q < sr.z; q++) info[(size_t)sr.y + q] = make_int2(sr.y, sr.z);
        GCNHIP_TRY(hipMalloc((void **)&g->slot_info, (size_t)n_slots * sizeof(int2)));
        GCNHIP_TRY(hipMemcpy(g->slot_info, info.data(), (size_t)n_slots * sizeof(int2), hipMemcpyHostToDevice));
        GCNHIP_TRY(hipMalloc((void **)&g->seg_count, (size_t)n_slots * 8 * sizeof(uint32_t)));
        GCNHIP_TRY(hipMemset(g->seg_count, 0, (size_t)n_slots * 8 * sizeof(uint32_t)));
    }
    xcd_bounds(tasks, g->bounds);
    if (!g->h_tasks) g->h_tasks = new std::vector<int4>();
    if (!g->h_srows) g->h_srows = new std::vector<int4>();
    *g->h_tasks = tasks;
    *g->h_srows = srows;
    if (g->rowsets)
        for (gcnhip_rowset *rs : *g->rowsets) {
            const int rc = build_rowset(rs, tasks, srows);
            if (rc != 0) return rc;
        }
    return 0;
}

extern "C" {

int gcnhip_graph_create(gcnhip_ctx *c, gcnhip_graph **out, const int *h_indptr, const int *h_indices,
                        int n_rows, int n_cols, const int *h_col_deg) {
    return gcnhip_graph_create_grouped(c, out, h_indptr, h_indices, n_rows, n_cols, h_col_deg, nullptr);
}

static int graph_create_impl(gcnhip_ctx *c, gcnhip_graph *g, const int *h_indptr, const int *h_indices,
                             int n_rows, int n_cols, const int *h_col_deg, const int *h_row_group);

int gcnhip_graph_create_grouped(gcnhip_ctx *c, gcnhip_graph **out, const int *h_indptr, const int *h_indices,
                                int n_rows, int n_cols, const int *h_col_deg, const int *h_row_group) {
    if (!c || !out || !h_indptr || n_rows < 0) return -1;
    if (!h_col_deg && n_cols != n_rows) return -1;
    const int nnz = h_indptr[n_rows];
    if (nnz < 0 || (nnz > 0 && !h_indices)) return -1;
    for (int r = 0; r < n_rows; r++)
        if (h_indptr[r + 1] < h_indptr[r]) return -1;
    for (int e = 0; e < nnz; e++)
        if (h_indices[e] < 0 || h_indices[e] >= n_cols) return -1;    // a bad column would fault the gather
    gcnhip_graph *g = new gcnhip_graph();
    memset(g, 0, sizeof *g);
    const int rc = graph_create_impl(c, g, h_indptr, h_indices, n_rows, n_cols, h_col_deg, h_row_group);
    if (rc != 0) { gcnhip_graph_destroy(c, g); return rc; }   // frees whatever was allocated
    *out = g;
    return 0;
}

static int graph_create_impl(gcnhip_ctx *c, gcnhip_graph *g, const int *h_indptr, const int *h_indices,
                             int n_rows, int n_cols, const int *h_col_deg, const int *h_row_group) {
    GCNHIP_TRY(hipSetDevice(c->device));
    const int nnz = h_indptr[n_rows];
    g->n_rows = n_rows; g->n_cols = n_cols; g->nnz = nnz;
    g->split_edges_opt = c->opt.split_edges;
    GCNHIP_TRY(hipMalloc((void **)&g->indptr, (size_t)(n_rows + 1) * sizeof(int)));
    GCNHIP_TRY(hipMalloc((void **)&g->indices, (size_t)std::max(nnz, 1) * sizeof(int)));
    GCNHIP_TRY(hipMalloc((void **)&g->coef, (size_t)std::max(nnz, 1) * sizeof(float)));
    GCNHIP_TRY(hipMemcpy(g->indptr, h_indptr, (size_t)(n_rows + 1) * sizeof(int), hipMemcpyHostToDevice));
    // Gather order inside a row: neighbours by descending degree.  Every wave then asks for the
    // popular rows (which are the ones that stay in L2) at the same point of its walk; measured on
    // reddit-syn this and the degree-ordered task list below are worth 6 % (d = 128) and 13 % (d = 41).
    // Only the order of the floating-point sum changes.
    std::vector<int> sorted_idx;
    if (nnz) {
        sorted_idx.assign(h_indices, h_indices + nnz);
        auto deg_of = [&](int j) { return h_col_deg ? h_col_deg[j] : h_indptr[j + 1] - h_indptr[j]; };
        // (host threads over row ranges of equal edge count: at Reddit scale this sort was 0.4 s of a 0.67 s object build, at
        //  R-MAT scale 22 most of 4.8 s; rows are independent, so the result does not depend on the thread count)
        auto sort_rows = [&](int r_lo, int r_hi) {
            std::vector<std::pair<int, int>> tmp;
            for (int r = r_lo; r < r_hi; r++) {
                const int e0 = h_indptr[r], e1 = h_indptr[r + 1];
                if (e1 - e0 < 2) continue;
                tmp.resize(e1 - e0);
                for (int e = e0; e < e1; e++) tmp[e - e0] = {-deg_of(sorted_idx[e]), sorted_idx[e]};
                std::sort(tmp.begin(), tmp.end());
                for (int e = e0; e < e1; e++) sorted_idx[e] = tmp[e - e0].second;
            }
        };
        int n_thr = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (nnz < (1 << 20)) n_thr = 1;
        if (n_thr == 1) {
            sort_rows(0, n_rows);
        } else {
            std::vector<std::thread> pool;
            int r_lo = 0;
            for (int t = 0; t < n_thr; t++) {
                const int64_t target = (int64_t)nnz * (t + 1) / n_thr;
                int r_hi = t == n_thr - 1 ? n_rows : (int)(std::upper_bound(h_indptr, h_indptr + n_rows + 1, (int)target) - h_indptr);
                r_hi = std::max(r_lo, std::min(n_rows, r_hi));
                pool.emplace_back(sort_rows, r_lo, r_hi);
                r_lo = r_hi;
            }
            for (auto &th : pool) th.join();
        }
        h_indices = sorted_idx.data();
        GCNHIP_TRY(hipMemcpy(g->indices, h_indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
    }
    int *&d_col_deg = g->tmp_col_deg;      // lives in the object so that a failure below still frees it
    if (h_col_deg) {
        GCNHIP_TRY(hipMalloc((void **)&d_col_deg, (size_t)std::max(n_cols, 1) * sizeof(int)));
        GCNHIP_TRY(hipMemcpy(d_col_deg, h_col_deg, (size_t)n_cols * sizeof(int), hipMemcpyHostToDevice));
    }
    if (n_rows) {
        edge_coef_kernel<<<ceil_div(n_rows, 256), 256, 0, c->stream>>>(g->indptr, g->indices, d_col_deg, g->coef, n_rows);
        GCNHIP_LAUNCH_CHECK();
    }
    GCNHIP_TRY(hipStreamSynchronize(c->stream));
    if (d_col_deg) { GCNHIP_TRY(hipFree(d_col_deg)); d_col_deg = nullptr; }

    g->h_indptr = new std::vector<int>(h_indptr, h_indptr + n_rows + 1);
    {   // the factored form of the coefficients: per-row and per-column 1/sqrt(deg) and 1/deg
        std::vector<float> dr((size_t)std::max(n_rows, 1)), dr2(dr.size()), dc((size_t)std::max(n_cols, 1)), dc2(dc.size());
        for (int r = 0; r < n_rows; r++) {
            const double d = (double)std::max(1, h_indptr[r + 1] - h_indptr[r]);
            dr[r] = (float)(1.0 / sqrt(d)); dr2[r] = (float)(1.0 / d);
        }
        for (int j = 0; j < n_cols; j++) {
            const double d = (double)std::max(1, h_col_deg ? h_col_deg[j] : h_indptr[j + 1] - h_indptr[j]);
            dc[j] = (float)(1.0 / sqrt(d)); dc2[j] = (float)(1.0 / d);
        }
        GCNHIP_TRY(hipMalloc((void **)&g->dinv_row, dr.size() * sizeof(float)));
        GCNHIP_TRY(hipMalloc((void **)&g->dinv2_row, dr.size() * sizeof(float)));
        GCNHIP_TRY(hipMalloc((void **)&g->dinv_col, dc.size() * sizeof(float)));
        GCNHIP_TRY(hipMalloc((void **)&g->dinv2_col, dc.size() * sizeof(float)));
        GCNHIP_TRY(hipMemcpy(g->dinv_row, dr.data(), dr.size() * sizeof(float), hipMemcpyHostToDevice));
        GCNHIP_TRY(hipMemcpy(g->dinv2_row, dr2.data(), dr.size() * sizeof(float), hipMemcpyHostToDevice));
        GCNHIP_TRY(hipMemcpy(g->dinv_col, dc.data(), dc.size() * sizeof(float), hipMemcpyHostToDevice));
        GCNHIP_TRY(hipMemcpy(g->dinv2_col, dc2.data(), dc.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return build_schedule(g, h_row_group);
}

int gcnhip_graph_create_restricted(gcnhip_ctx *c, gcnhip_graph **out, const gcnhip_graph *parent, const uint32_t *h_col_bits) {
    if (!c || !out || !parent || !h_col_bits || !parent->h_indptr || !parent->h_tasks) return -1;
    GCNHIP_TRY(hipSetDevice(c->device));
    GCNHIP_TRY(hipStreamSynchronize(c->stream));
    const int n_rows = parent->n_rows, nnz = parent->nnz;
    // the parent's edges as it stores them (neighbours by descending degree) and ITS coefficients: degrees of the full graph
    std::vector<int> idx((size_t)std::max(nnz, 1));
    std::vector<float> cf((size_t)std::max(nnz, 1));
    if (nnz) {
        GCNHIP_TRY(hipMemcpy(idx.data(), parent->indices, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost));
        GCNHIP_TRY(hipMemcpy(cf.data(), parent->coef, (size_t)nnz * sizeof(float), hipMemcpyDeviceToHost));
    }
    const int *pp = parent->h_indptr->data();
    std::vector<int> ip((size_t)n_rows + 1);
    size_t w = 0;
    for (int r = 0; r < n_rows; r++) {
        ip[r] = (int)w;
        for (int e = pp[r]; e < pp[r + 1]; e++) {
            const int j = idx[e];
            if ((h_col_bits[j >> 5] >> (j & 31)) & 1u) { idx[w] = j; cf[w] = cf[e]; w++; }   // w <= e: in place
        }
    }
    ip[n_rows] = (int)w;
    gcnhip_graph *g = new gcnhip_graph();
    memset(g, 0, sizeof *g);
    auto fail = [&](int rc) { gcnhip_graph_destroy(c, g); return rc; };
    g->n_rows = n_rows; g->n_cols = parent->n_cols; g->nnz = (int)w;
    g->part_ld = parent->part_ld;
    g->split_edges_opt = parent->split_edges_opt;
    if (hipMalloc((void **)&g->indptr, (size_t)(n_rows + 1) * sizeof(int)) != hipSuccess) return fail(-2);
    if (hipMalloc((void **)&g->indices, std::max(w, (size_t)1) * sizeof(int)) != hipSuccess) return fail(-2);
    if (hipMalloc((void **)&g->coef, std::max(w, (size_t)1) * sizeof(float)) != hipSuccess) return fail(-2);
    if (hipMemcpy(g->indptr, ip.data(), (size_t)(n_rows + 1) * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail(-3);
    if (w && hipMemcpy(g->indices, idx.data(), w * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail(-3);
    if (w && hipMemcpy(g->coef, cf.data(), w * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(-3);
    g->h_indptr = new std::vector<int>(std::move(ip));
    {   // the parent's scale arrays: degrees of the FULL graph, not of the edges that are left
        const size_t nr = (size_t)std::max(n_rows, 1) * sizeof(float), nc = (size_t)std::max(parent->n_cols, 1) * sizeof(float);
        if (hipMalloc((void **)&g->dinv_row, nr) != hipSuccess || hipMalloc((void **)&g->dinv2_row, nr) != hipSuccess ||
            hipMalloc((void **)&g->dinv_col, nc) != hipSuccess || hipMalloc((void **)&g->dinv2_col, nc) != hipSuccess) return fail(-2);
        if (hipMemcpy(g->dinv_row, parent->dinv_row, nr, hipMemcpyDeviceToDevice) != hipSuccess || hipMemcpy(g->dinv2_row, parent->dinv2_row, nr, hipMemcpyDeviceToDevice) != hipSuccess ||
            hipMemcpy(g->dinv_col, parent->dinv_col, nc, hipMemcpyDeviceToDevice) != hipSuccess || hipMemcpy(g->dinv2_col, parent->dinv2_col, nc, hipMemcpyDeviceToDevice) != hipSuccess) return fail(-3);
    }
    // the parent's current row order (a split row appears once per segment, consecutively)
    std::vector<int> order;
    order.reserve((size_t)n_rows);
    for (const int4 &t : *parent->h_tasks)
        if (order.empty() || order.back() != t.x) order.push_back(t.x);
    if ((int)order.size() != n_rows) return fail(-1);
    const int rc = build_tasks(g, order);
    if (rc != 0) return fail(rc);
    *out = g;
    return 0;
}

// A second object with the parent's edges, coefficients, factors and current row order, and its OWN task lists and
// split-row scratch (two streams may aggregate at the same time only through different objects).  Device-to-device copies:
// none of the host preparation of gcnhip_graph_create (validation, per-row neighbour sort, coefficient kernel) is repeated.
int gcnhip_graph_clone(gcnhip_ctx *c, gcnhip_graph **out, const gcnhip_graph *parent) {
    if (!c || !out || !parent || !parent->h_indptr || !parent->h_tasks) return -1;
    GCNHIP_TRY(hipSetDevice(c->device));
    GCNHIP_TRY(hipStreamSynchronize(c->stream));
    const int n_rows = parent->n_rows, n_cols = parent->n_cols, nnz = parent->nnz;
    gcnhip_graph *g = new gcnhip_graph();
    memset(g, 0, sizeof *g);
    auto fail = [&](int rc) { gcnhip_graph_destroy(c, g); return rc; };
    g->n_rows = n_rows; g->n_cols = n_cols; g->nnz = nnz;
    g->part_ld = parent->part_ld;
    g->split_edges_opt = parent->split_edges_opt;
    struct Copy { void **dst; const void *src; size_t bytes; };
    const size_t nr = (size_t)std::max(n_rows, 1) * sizeof(float), nc = (size_t)std::max(n_cols, 1) * sizeof(float);
    const Copy copies[] = {{(void **)&g->indptr, parent->indptr, (size_t)(n_rows + 1) * sizeof(int)},
                           {(void **)&g->indices, parent->indices, (size_t)std::max(nnz, 1) * sizeof(int)},
                           {(void **)&g->coef, parent->coef, (size_t)std::max(nnz, 1) * sizeof(float)},
                           {(void **)&g->dinv_row, parent->dinv_row, nr}, {(void **)&g->dinv2_row, parent->dinv2_row, nr},
                           {(void **)&g->dinv_col, parent->dinv_col, nc}, {(void **)&g->dinv2_col, parent->dinv2_col, nc}};
    for (const Copy &cp : copies) {
        if (hipMalloc(cp.dst, cp.bytes) != hipSuccess) return fail(-2);
        if (hipMemcpy(*cp.dst, cp.src, cp.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return fail(-3);
    }
    g->h_indptr = new std::vector<int>(*parent->h_indptr);
    std::vector<int> order;
    order.reserve((size_t)n_rows);
    for (const int4 &t : *parent->h_tasks)
        if (order.empty() || order.back() != t.x) order.push_back(t.x);
    if ((int)order.size() != n_rows) return fail(-1);
    const int rc = build_tasks(g, order);
    if (rc != 0) return fail(rc);
    *out = g;
    return 0;
}

int gcnhip_graph_destroy(gcnhip_ctx *c, gcnhip_graph *g) {
    if (!g) return 0;
    hipSetDevice(c->device);
    if (g->indptr) hipFree(g->indptr);
    if (g->indices) hipFree(g->indices);
    if (g->coef) hipFree(g->coef);
    if (g->tmp_col_deg) hipFree(g->tmp_col_deg);
    if (g->dinv_row) hipFree(g->dinv_row);
    if (g->dinv2_row) hipFree(g->dinv2_row);
    if (g->dinv_col) hipFree(g->dinv_col);
    if (g->dinv2_col) hipFree(g->dinv2_col);
    delete g->h_indptr;
    delete g->h_tasks;
    delete g->h_srows;
    if (g->rowsets) {
        for (gcnhip_rowset *rs : *g->rowsets) {
            if (rs->tasks) hipFree(rs->tasks);
            if (rs->split_rows) hipFree(rs->split_rows);
            delete rs;
        }
        delete g->rowsets;
    }
    if (g->tasks) hipFree(g->tasks);
    if (g->split_rows) hipFree(g->split_rows);
    if (g->partials) hipFree(g->partials);
    if (g->slot_info) hipFree(g->slot_info);
    if (g->seg_count) hipFree(g->seg_count);
    delete g;
    return 0;
}

int gcnhip_graph_set_schedule(gcnhip_ctx *c, gcnhip_graph *g, int mode, const int *h_row_group, int n_groups) {
    if (!c || !g || !g->h_indptr || mode < 0 || mode > 2) return -1;
    if (mode == 1 && !h_row_group) return -1;
    if (mode == 2 && n_groups < 1) return -1;
    GCNHIP_TRY(hipSetDevice(c->device));
    GCNHIP_TRY(hipStreamSynchronize(c->stream));       // no aggregation may still be reading the old task list
    if (mode == 0) return build_schedule(g, nullptr);
    if (mode == 1) return build_schedule(g, h_row_group);
    // mode 2: rows ranked by descending degree, rank r goes to group r % n_groups — every group has the
    // same degree mix, so hub rows and the long tail of short rows are in flight together
    const int n = g->n_rows;
    const int *ip = g->h_indptr->data();
    std::vector<int> order(n), key(n);
    for (int r = 0; r < n; r++) order[r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ip[a + 1] - ip[a] > ip[b + 1] - ip[b]; });
    for (int k = 0; k < n; k++) key[order[k]] = k % n_groups;
    return build_schedule(g, key.data());
}

int gcnhip_graph_add_rowset(gcnhip_ctx *c, gcnhip_graph *g, const uint32_t *h_row_bits, gcnhip_rowset **out) {
    if (!c || !g || !h_row_bits || !out || !g->h_tasks) return -1;
    GCNHIP_TRY(hipSetDevice(c->device));
    gcnhip_rowset *rs = new gcnhip_rowset();
    rs->n_tasks = rs->n_split_rows = 0;
    rs->tasks = rs->split_rows = nullptr;
    rs->owner = g;
    rs->bits.assign(h_row_bits, h_row_bits + ((size_t)g->n_rows + 31) / 32);   // exactly the n_rows bits the header documents
    rs->bits.push_back(0u);                                                       // (+ a zero word: row ids index it as r >> 5 with r < n_rows)
    const int rc = build_rowset(rs, *g->h_tasks, *g->h_srows);
    if (rc != 0) {
        if (rs->tasks) hipFree(rs->tasks);
        if (rs->split_rows) hipFree(rs->split_rows);
        delete rs;
        return rc;
    }
    if (!g->rowsets) g->rowsets = new std::vector<gcnhip_rowset *>();
    g->rowsets->push_back(rs);
    *out = rs;
    return 0;
}
int gcnhip_rowset_size(const gcnhip_rowset *rs, int *n_rows_tasks) {
    if (!rs || !n_rows_tasks) return -1;
    *n_rows_tasks = rs->n_tasks;
    return 0;
}

int gcnhip_graph_reserve_width(gcnhip_ctx *c, gcnhip_graph *g, int max_dim) {
    if (!c || !g || max_dim <= 0) return -1;
    const int want = (max_dim + 7) / 8 * 8;
    if (want <= g->part_ld) return 0;
    GCNHIP_TRY(hipSetDevice(c->device));
    GCNHIP_TRY(hipStreamSynchronize(c->stream));       // no aggregation may still be writing the old scratch
    if (g->partials) { GCNHIP_TRY(hipFree(g->partials)); g->partials = nullptr; }
    g->part_ld = want;
    if (g->n_slots) GCNHIP_TRY(hipMalloc((void **)&g->partials, (size_t)g->n_slots * g->part_ld * sizeof(float)));
    return 0;
}

int gcnhip_graph_scales(const gcnhip_graph *g, const float **dinv_row, const float **dinv2_row, const float **dinv_col, const float **dinv2_col) {
    if (!g) return -1;
    if (dinv_row) *dinv_row = g->dinv_row;
    if (dinv2_row) *dinv2_row = g->dinv2_row;
    if (dinv_col) *dinv_col = g->dinv_col;
    if (dinv2_col) *dinv2_col = g->dinv2_col;
    return 0;
}

// values[e] *= scale[row of e]: the feature matrix of the factored first layer, (D^-1/2 X) — see gcnhip_graphsum_ex
__global__ void feat_scale_rows_kernel(float *vals, float *vals_pad, int ld_pad, const int *indptr, const float *scale, int n_rows, int n_cols, int dense) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float s = scale[r];
    const int lane = threadIdx.x & 63;
    for (int e = indptr[r] + lane; e < indptr[r + 1]; e += 64) vals[e] *= s;
    if (vals_pad && dense)
        for (int k = lane; k < n_cols; k += 64) vals_pad[(size_t)r * ld_pad + k] *= s;
}
__global__ void feat_scale_csc_kernel(float *csc_val, const int *csc_row, const float *scale, int64_t nnz) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nnz) csc_val[q] *= scale[csc_row[q]];
}
int gcnhip_feat_scale_rows(gcnhip_ctx *c, gcnhip_feat *f, const float *d_row_scale) {
    if (!c || !f || !d_row_scale) return -1;
    if (f->n_rows == 0) return 0;
    feat_scale_rows_kernel<<<ceil_div(f->n_rows, 4), 256, 0, c->stream>>>(f->values, f->values_pad, f->ld_pad, f->indptr, d_row_scale, f->n_rows, f->n_cols, f->dense ? 1 : 0);
    GCNHIP_LAUNCH_CHECK();
    if (f->csc_val && f->nnz) {
        feat_scale_csc_kernel<<<ceil_div(f->nnz, 256), 256, 0, c->stream>>>(f->csc_val, f->csc_row, d_row_scale, f->nnz);
        GCNHIP_LAUNCH_CHECK();
    }
    GCNHIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int gcnhip_graph_arrays(const gcnhip_graph *g, const int **d_indptr, const int **d_indices,
                        const float **d_coef, int *n_rows, int *nnz) {
    if (!g) return -1;
    if (d_indptr) *d_indptr = g->indptr;
    if (d_indices) *d_indices = g->indices;
    if (d_coef) *d_coef = g->coef;
    if (n_rows) *n_rows = g->n_rows;
    if (nnz) *nnz = g->nnz;
    return 0;
}

// ---------------------------------------------------------------- features
static int feat_create_impl(gcnhip_ctx *c, gcnhip_feat *f, const int *h_indptr, const int *h_indices,
                            const float *h_values, int n_rows, int n_cols);

int gcnhip_feat_create(gcnhip_ctx *c, gcnhip_feat **out, const int *h_indptr, const int *h_indices,
                       const float *h_values, int n_rows, int n_cols) {
    if (!c || !out || !h_indptr || !h_values || n_rows < 0 || n_cols <= 0) return -1;
    gcnhip_feat *f = new gcnhip_feat();
    memset(f, 0, sizeof *f);
    const int rc = feat_create_impl(c, f, h_indptr, h_indices, h_values, n_rows, n_cols);
    if (rc != 0) { gcnhip_feat_destroy(c, f); return rc; }     // frees whatever was allocated
    *out = f;
    return 0;
}

static int feat_create_impl(gcnhip_ctx *c, gcnhip_feat *f, const int *h_indptr, const int *h_indices,
                            const float *h_values, int n_rows, int n_cols) {
    GCNHIP_TRY(hipSetDevice(c->device));
    f->n_rows = n_rows; f->n_cols = n_cols;
    const int64_t nnz = h_indptr[n_rows];
    f->nnz = nnz;
    // dense <=> every row is exactly 0..n_cols-1 in order (h_indices == NULL asserts it)
    bool dense = (nnz == (int64_t)n_rows * n_cols) && n_rows > 0;
    if (dense && h_indices) {
        for (int r = 0; r < n_rows && dense; r++) {
            if (h_indptr[r + 1] - h_indptr[r] != n_cols) { dense = false; break; }
            const int *row = h_indices + (size_t)r * n_cols;
            for (int k = 0; k < n_cols; k++)
                if (row[k] != k) { dense = false; break; }
        }
    }
    if (!h_indices && !dense) return -1;
    f->dense = dense;
    GCNHIP_TRY(hipMalloc((void **)&f->indptr, (size_t)(n_rows + 1) * sizeof(int)));
    GCNHIP_TRY(hipMemcpy(f->indptr, h_indptr, (size_t)(n_rows + 1) * sizeof(int), hipMemcpyHostToDevice));
    GCNHIP_TRY(hipMalloc((void **)&f->values, (size_t)std::max<int64_t>(nnz, 4) * sizeof(float)));
    if (nnz) GCNHIP_TRY(hipMemcpy(f->values, h_values, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
    {   // flat: a bit per stored element; chunk-major (dense X, dense_bf16x3.h): a word per row and 32 columns.  Slack: tiles read bits of pad columns
        const size_t words_cm = dense ? (size_t)n_rows * ((n_cols + 31) / 32) : 0;
        GCNHIP_TRY(hipMalloc((void **)&f->keep_bits, (std::max((size_t)(nnz / 32), words_cm) + 32) * sizeof(uint32_t)));
        f->keep_layout = 0;
    }
    if (dense && n_cols % 128 != 0 && n_cols >= 64) {
        // the MFMA tiles stage X with unconditional 16-byte lane loads when every row starts on a 16-byte
        // boundary and its stride covers whole 128-column tiles (zero padded); HBM has room for the second copy
        f->ld_pad = (n_cols + 127) / 128 * 128;
        GCNHIP_TRY(hipMalloc((void **)&f->values_pad, (size_t)n_rows * f->ld_pad * sizeof(float)));
        GCNHIP_TRY(hipMemset(f->values_pad, 0, (size_t)n_rows * f->ld_pad * sizeof(float)));
        GCNHIP_TRY(hipMemcpy2D(f->values_pad, (size_t)f->ld_pad * sizeof(float), f->values, (size_t)n_cols * sizeof(float),
                               (size_t)n_cols * sizeof(float), (size_t)n_rows, hipMemcpyDeviceToDevice));
    }
    if (!dense) {
        for (int64_t e = 0; e < nnz; e++)
            if (h_indices[e] < 0 || h_indices[e] >= n_cols) { return -1; }
        GCNHIP_TRY(hipMalloc((void **)&f->indices, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int)));
        if (nnz) GCNHIP_TRY(hipMemcpy(f->indices, h_indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
        // CSC by counting sort; entries of a column stay in row order, so the
        // gather-form weight gradient adds them in the reference's order
        // (src/seq/module.cpp:68-74 visits rows ascending).
        std::vector<int> ptr((size_t)n_cols + 1, 0), row((size_t)nnz), pos((size_t)nnz);
        for (int64_t e = 0; e < nnz; e++) ptr[h_indices[e] + 1]++;
        for (int k = 0; k < n_cols; k++) ptr[k + 1] += ptr[k];
        std::vector<int> fill(ptr.begin(), ptr.end() - 1);
        for (int r = 0; r < n_rows; r++)
            for (int e = h_indptr[r]; e < h_indptr[r + 1]; e++) {
                const int q = fill[h_indices[e]]++;
                row[q] = r;
                pos[q] = e;
            }
        GCNHIP_TRY(hipMalloc((void **)&f->csc_ptr, ptr.size() * sizeof(int)));
        GCNHIP_TRY(hipMalloc((void **)&f->csc_row, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int)));
        GCNHIP_TRY(hipMalloc((void **)&f->csc_pos, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int)));
        GCNHIP_TRY(hipMemcpy(f->csc_ptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice));
        if (nnz) {
            GCNHIP_TRY(hipMemcpy(f->csc_row, row.data(), (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
            GCNHIP_TRY(hipMemcpy(f->csc_pos, pos.data(), (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
            std::vector<float> cv((size_t)nnz);
            for (int64_t q = 0; q < nnz; q++) cv[q] = h_values[pos[q]];
            GCNHIP_TRY(hipMalloc((void **)&f->csc_val, (size_t)nnz * sizeof(float)));
            GCNHIP_TRY(hipMemcpy(f->csc_val, cv.data(), (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
        }
        // Task list of the weight gradient (spmm_sparse.h).  Waves per task by the mean column length: a short column is
        // one wave's walk, a long one is shared by 4 or 16 waves of one workgroup (Pubmed: ~2 000 entries per column);
        // anything beyond the segment length is cut into several tasks with partial rows (skewed bag-of-words columns).
        const double mean = n_cols ? (double)nnz / n_cols : 0.0;
        f->bwd_nw = mean <= 128.0 ? 1 : (mean <= 1024.0 ? 4 : 16);
        if (c->opt.spmm_nw == 1 || c->opt.spmm_nw == 4 || c->opt.spmm_nw == 16) f->bwd_nw = c->opt.spmm_nw;      // experiments
        const int seg = std::max(1024, f->bwd_nw * 256);
        std::vector<int4> tasks, split;
        tasks.reserve((size_t)n_cols + 16);
        int n_slots = 0;
        for (int k = 0; k < n_cols; k++) {
            const int q0 = ptr[k], q1 = ptr[k + 1];
            if (q1 - q0 <= seg) { tasks.push_back(make_int4(k, q0, q1, -1)); continue; }
            const int ns = (q1 - q0 + seg - 1) / seg;
            split.push_back(make_int4(k, n_slots, ns, 0));
            for (int q = 0; q < ns; q++) tasks.push_back(make_int4(k, q0 + q * seg, std::min(q1, q0 + (q + 1) * seg), n_slots + q));
            n_slots += ns;
        }
        f->n_bwd_tasks = (int)tasks.size(); f->n_bwd_split = (int)split.size(); f->n_bwd_slots = n_slots;
        GCNHIP_TRY(hipMalloc((void **)&f->bwd_tasks, std::max(tasks.size(), (size_t)1) * sizeof(int4)));
        if (!tasks.empty()) GCNHIP_TRY(hipMemcpy(f->bwd_tasks, tasks.data(), tasks.size() * sizeof(int4), hipMemcpyHostToDevice));
        if (!split.empty()) {
            GCNHIP_TRY(hipMalloc((void **)&f->bwd_split, split.size() * sizeof(int4)));
            GCNHIP_TRY(hipMemcpy(f->bwd_split, split.data(), split.size() * sizeof(int4), hipMemcpyHostToDevice));
        }
    }
    return 0;
}

// A^.X for a dense X, computed once: the feature object of an evaluation forward that aggregates first.
static int feat_aggregate_impl(gcnhip_ctx *c, gcnhip_feat *f, gcnhip_graph *g, const gcnhip_feat *x) {
    GCNHIP_TRY(hipSetDevice(c->device));
    const int F = x->n_cols, n = g->n_rows;
    f->n_rows = n; f->n_cols = F; f->nnz = (int64_t)n * F; f->dense = true;
    std::vector<int> ip((size_t)n + 1);
    for (int r = 0; r <= n; r++) ip[r] = (int)((int64_t)r * F);
    GCNHIP_TRY(hipMalloc((void **)&f->indptr, (size_t)(n + 1) * sizeof(int)));
    GCNHIP_TRY(hipMemcpy(f->indptr, ip.data(), (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice));
    GCNHIP_TRY(hipMalloc((void **)&f->values, (size_t)std::max<int64_t>(f->nnz, 4) * sizeof(float)));
    // no keep-bit array: an aggregated feature object serves evaluation forwards only (no dropout); a dropout call on it
    // is refused in spmm.hip
    int rc = gcnhip_graph_reserve_width(c, g, F);
    if (rc != 0) return rc;
    const float *src = x->values_pad ? x->values_pad : x->values;
    const int ld_src = x->values_pad ? x->ld_pad : F;
    if (x->values_pad) {                          // keep the padded, 16-byte aligned layout the MFMA tiles read
        f->ld_pad = x->ld_pad;
        GCNHIP_TRY(hipMalloc((void **)&f->values_pad, (size_t)n * f->ld_pad * sizeof(float)));
        GCNHIP_TRY(hipMemsetAsync(f->values_pad, 0, (size_t)n * f->ld_pad * sizeof(float), c->stream));
        rc = gcnhip_graphsum(c, g, src, ld_src, f->values_pad, f->ld_pad, F);
        if (rc != 0) return rc;
        GCNHIP_TRY(hipMemcpy2DAsync(f->values, (size_t)F * sizeof(float), f->values_pad, (size_t)f->ld_pad * sizeof(float),
                                    (size_t)F * sizeof(float), (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    } else {
        rc = gcnhip_graphsum(c, g, src, ld_src, f->values, F, F);
        if (rc != 0) return rc;
    }
    GCNHIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int gcnhip_feat_create_aggregated(gcnhip_ctx *c, gcnhip_feat **out, gcnhip_graph *g, const gcnhip_feat *x) {
    if (!c || !out || !g || !x || !x->dense || x->n_rows != g->n_cols) return -1;
    gcnhip_feat *f = new gcnhip_feat();
    memset(f, 0, sizeof *f);
    const int rc = feat_aggregate_impl(c, f, g, x);
    if (rc != 0) { gcnhip_feat_destroy(c, f); return rc; }
    *out = f;
    return 0;
}

int gcnhip_feat_destroy(gcnhip_ctx *c, gcnhip_feat *f) {
    if (!f) return 0;
    hipSetDevice(c->device);
    if (f->indptr) hipFree(f->indptr);
    if (f->values) hipFree(f->values);
    if (f->indices) hipFree(f->indices);
    if (f->csc_ptr) hipFree(f->csc_ptr);
    if (f->csc_row) hipFree(f->csc_row);
    if (f->csc_pos) hipFree(f->csc_pos);
    if (f->csc_val) hipFree(f->csc_val);
    if (f->keep_bits) hipFree(f->keep_bits);
    if (f->values_pad) hipFree(f->values_pad);
    if (f->bwd_tasks) hipFree(f->bwd_tasks);
    if (f->bwd_split) hipFree(f->bwd_split);
    if (f->bwd_partials) hipFree(f->bwd_partials);
    delete f;
    return 0;
}
int gcnhip_feat_is_dense(const gcnhip_feat *f) { return f && f->dense ? 1 : 0; }
const float *gcnhip_feat_values(const gcnhip_feat *f) { return f ? f->values : nullptr; }
int64_t gcnhip_feat_nnz(const gcnhip_feat *f) { return f ? f->nnz : 0; }

}  // extern "C"

#include "variable.h"
#include <cmath>
#include <cstring>
#include "hip_check.h"

void Variable::glorot(int in_size, int out_size, HostRng &rng) {
    // variable.cpp:11-18: float division by (float)MY_RAND_MAX, "- 0.5" in double
    float range = sqrtf(6.0f / (in_size + out_size));
    for (size_t i = 0; i < data.size(); i++) {
        const float r = (float)((double)((float)rng.next() / (float)MY_RAND_MAX) - 0.5);
        data[i] = r * range * 2;
    }
}

HipVariable::~HipVariable() {
    if (!ctx) return;
    if (full) gcnhip_free(ctx, full);
    else if (data) gcnhip_free(ctx, data);
    if (full_grad) gcnhip_free(ctx, full_grad);
    else if (grad) gcnhip_free(ctx, grad);
}

void HipVariable::alloc(gcnhip_ctx *c, int r, int cl, bool rg, bool gather_data, bool gather_grad, const ExchangePlan *plan) {
    const int world = plan ? plan->world : 1;
    ctx = c; rows = r; cols = cl; requires_grad = rg;
    // 16-byte aligned rows; wider rows are padded to 64 bytes so a gathered row never straddles a
    // third 128-byte line (GraphSum at 41 classes: ld 48 is 8 % faster than ld 44)
    ld = cl <= 32 ? (cl + 3) / 4 * 4 : (cl + 15) / 16 * 16;
    const size_t local = (size_t)(rows > 0 ? rows : 1) * ld;
    const size_t own = plan ? (size_t)plan->own_offset * ld : 0;
    full_elems = plan ? (size_t)plan->table_rows * ld : 0;
    void *p = nullptr;
    if (gather_data && world > 1) {
        GCNHIP_CHECK(gcnhip_malloc(ctx, &p, full_elems * sizeof(float)));
        GCNHIP_CHECK(gcnhip_memset_async(ctx, p, 0, full_elems * sizeof(float)));
        full = (float *)p;
        data = full + own;
    } else {
        GCNHIP_CHECK(gcnhip_malloc(ctx, &p, local * sizeof(float)));
        GCNHIP_CHECK(gcnhip_memset_async(ctx, p, 0, local * sizeof(float)));
        data = (float *)p;
    }
    if (rg) {
        if (gather_grad && world > 1) {
            GCNHIP_CHECK(gcnhip_malloc(ctx, &p, full_elems * sizeof(float)));
            GCNHIP_CHECK(gcnhip_memset_async(ctx, p, 0, full_elems * sizeof(float)));
            full_grad = (float *)p;
            grad = full_grad + own;
        } else {
            GCNHIP_CHECK(gcnhip_malloc(ctx, &p, local * sizeof(float)));
            GCNHIP_CHECK(gcnhip_memset_async(ctx, p, 0, local * sizeof(float)));
            grad = (float *)p;
        }
    }
}

void HipVariable::alloc_replicated(gcnhip_ctx *c, int total_rows, int local_rows, int row_start, int cl, bool rg) {
    ctx = c; rows = local_rows; cols = cl; requires_grad = rg; replicated = true;
    ld = cl <= 32 ? (cl + 3) / 4 * 4 : (cl + 15) / 16 * 16;
    full_elems = (size_t)total_rows * ld;
    void *p = nullptr;
    GCNHIP_CHECK(gcnhip_malloc(ctx, &p, (full_elems ? full_elems : 4) * sizeof(float)));
    GCNHIP_CHECK(gcnhip_memset_async(ctx, p, 0, full_elems * sizeof(float)));
    full = (float *)p;
    data = full + (size_t)row_start * ld;
    if (rg) {
        const size_t local = (size_t)(rows > 0 ? rows : 1) * ld;
        GCNHIP_CHECK(gcnhip_malloc(ctx, &p, local * sizeof(float)));
        GCNHIP_CHECK(gcnhip_memset_async(ctx, p, 0, local * sizeof(float)));
        grad = (float *)p;
    }
}

void HipVariable::zero() { GCNHIP_CHECK(gcnhip_memset_async(ctx, data, 0, elems() * sizeof(float))); }
void HipVariable::zero_grad() { if (grad) GCNHIP_CHECK(gcnhip_memset_async(ctx, grad, 0, elems() * sizeof(float))); }

void HipVariable::upload(const float *h) {
    std::vector<float> tmp((size_t)rows * ld, 0.f);
    for (int r = 0; r < rows; r++) memcpy(&tmp[(size_t)r * ld], h + (size_t)r * cols, cols * sizeof(float));
    GCNHIP_CHECK(gcnhip_h2d(ctx, data, tmp.data(), tmp.size() * sizeof(float)));
}
void HipVariable::download(float *h, bool want_grad) const {
    std::vector<float> tmp((size_t)rows * ld);
    GCNHIP_CHECK(gcnhip_d2h(ctx, tmp.data(), want_grad ? grad : data, tmp.size() * sizeof(float)));
    for (int r = 0; r < rows; r++) memcpy(h + (size_t)r * cols, &tmp[(size_t)r * ld], cols * sizeof(float));
}

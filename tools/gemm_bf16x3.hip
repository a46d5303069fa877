// gemm_bf16x3.hip — pricing of the three-plane bf16 form of the dense first-layer products (csrc/dense_bf16x3.h) beside the
// exact-f32 MFMA kernels the product has run since round 3 (csrc/dense_persist.h), at the benchmark's shape
// (m = 232 965, K = 602, p = 128; reference: SparseMatmul::forward, /root/reference/src/seq/module.cpp:47-61):
// time per launch (HIP events, back to back) and the error of both against a float64 product on sampled rows, in units of
// eps_f32 * sum_k |x_k w_k| (the bound the parity tests use is 8 of those units).
// Also here: the template ablations of the forward (which stream costs what), the dropout ablations that led to the chunk-major
// keep words, the W prefetch distance, and the load-ORDER experiments of both kernels (runs rotated: the clock drifts with what ran
// before) — docs/NOTEBOOK_r5.md §2.  The bf16x3 kernels read keep words chunk-major; the tool derives them from the flat bits the
// f32 kernels and the float64 check use.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icuda_gcn_amd/csrc tools/gemm_bf16x3.hip -o build/gemm_bf16x3
//   build/gemm_bf16x3 [m] [K] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <random>
#include <vector>
#include "dense_kernels.h"
#include "dense_bf16x3.h"

int gcnhip_fail(const char *d) { fprintf(stderr, "%s\n", d ? d : ""); return -1; }
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s at line %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <class F>
static float time_ms(int iters, F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

// does global_load_lds_dwordx3 lay the wave's 64 x 12 bytes out contiguously (lane * 12) from the M0 base?
__global__ void glds12_probe(const uint32_t *src, uint32_t *dst) {
    __shared__ __attribute__((aligned(1024))) uint32_t buf[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) buf[i] = 0xDEADBEEFu;
    __syncthreads();
    {
        uint32_t keep;
        const void *gsrc = reinterpret_cast<const unsigned char *>(src) + lane * 12;
        const uint32_t lds_dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)buf;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) dst[i] = buf[i];
}

// a bandwidth hog for the co-running check: streams a big buffer; `use_lds` workgroups also hold 32 KB of LDS each, so that
// they share CUs (and the LDS allocator) with the kernel under test
__global__ __launch_bounds__(256) void hog_kernel(float4 *buf, size_t n, int use_lds) {
    __shared__ float pad[8192];
    if (use_lds) { pad[threadIdx.x] = (float)blockIdx.x; __syncthreads(); }
    float s = use_lds ? pad[(threadIdx.x * 7) & 255] : 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = buf[i]; v.x += s; buf[i] = v;
    }
}

int main(int argc, char **argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 232965, K = argc > 2 ? atoi(argv[2]) : 602, iters = argc > 3 ? atoi(argv[3]) : 20;
    const int p = 128, ldx = (K + 31) / 32 * 32, n_chunks = ldx / 32, n_rb = (m + 31) / 32;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("m=%d K=%d (ldx %d) p=%d  CUs=%d  %.2f GFLOP per product\n", m, K, ldx, p, n_cu, 2.0 * m * K * p / 1e9);
    {
        uint32_t *ps, *pd;
        std::vector<uint32_t> h(512), o(512);
        for (int i = 0; i < 512; i++) h[i] = i;
        CK(hipMalloc(&ps, 2048)); CK(hipMalloc(&pd, 2048));
        CK(hipMemcpy(ps, h.data(), 2048, hipMemcpyHostToDevice));
        glds12_probe<<<1, 64>>>(ps, pd);
        CK(hipMemcpy(o.data(), pd, 2048, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 192; i++) bad += o[i] != (uint32_t)i;
        printf("global_load_lds_dwordx3 probe: %d of 192 dwords differ from the lane*12 layout; dwords 0..7 = %u %u %u %u %u %u %u %u; dword 192 = %x\n", bad,
               o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[192]);
    }
    std::vector<float> hx((size_t)m * ldx, 0.f), hw((size_t)K * p);
    std::mt19937_64 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int i = 0; i < m; i++) for (int k = 0; k < K; k++) hx[(size_t)i * ldx + k] = nd(rng);
    const float wscale = sqrtf(6.f / (K + p));
    std::uniform_real_distribution<float> ud(-wscale, wscale);
    for (auto &v : hw) v = ud(rng);
    std::vector<uint32_t> hbits(((size_t)m * K + 31) / 32 + 8);
    for (auto &v : hbits) v = (uint32_t)rng();
    float *dx, *dw, *dout, *dout2, *dwp32;
    uint4 *dwp;
    uint32_t *dbits;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4));
    CK(hipMalloc(&dout, (size_t)m * p * 4)); CK(hipMalloc(&dout2, (size_t)m * p * 4));
    CK(hipMalloc(&dwp, (size_t)n_chunks * 2 * BX_BH_BYTES)); CK(hipMalloc(&dwp32, (size_t)n_chunks * 4 * 1024 * 4));
    CK(hipMalloc(&dbits, hbits.size() * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbits, hbits.data(), hbits.size() * 4, hipMemcpyHostToDevice));
    // the same decisions chunk-major (word c * m + r = columns 32 c .. 32 c + 31 of row r): what the bf16x3 kernels read
    uint32_t *dcm;
    {
        std::vector<uint32_t> hcm((size_t)n_chunks * m + 32, 0u);
        for (int r = 0; r < m; r++)
            for (int k = 0; k < K; k++) {
                const size_t e = (size_t)r * K + k;
                if ((hbits[e >> 5] >> (e & 31)) & 1) hcm[(size_t)(k >> 5) * m + r] |= 1u << (k & 31);
            }
        CK(hipMalloc(&dcm, hcm.size() * 4));
        CK(hipMemcpy(dcm, hcm.data(), hcm.size() * 4, hipMemcpyHostToDevice));
    }

    // sampled rows for the error check
    std::vector<int> rows;
    for (int i = 0; i < 96; i++) rows.push_back((int)((uint64_t)i * 2654435761u % (uint64_t)m));
    rows.push_back(0); rows.push_back(m - 1); rows.push_back(m - 33); rows.push_back(std::min(m - 1, 255)); rows.push_back(std::min(m - 1, 256));
    auto check = [&](const float *d_out, const char *what, bool drop, float scale) {
        std::vector<float> ho((size_t)m * p);
        hipMemcpy(ho.data(), d_out, ho.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, sumsq = 0; long cnt = 0;
        long nonfinite = 0; int first_bad = -1;
        for (size_t i = 0; i < ho.size(); i++) if (!std::isfinite(ho[i])) { nonfinite++; if (first_bad < 0) first_bad = (int)(i / p); }
        if (nonfinite) printf("  !! %ld non-finite outputs, first in row %d\n", nonfinite, first_bad);
        for (int r : rows) for (int c = 0; c < p; c++) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; k++) {
                const size_t e = (size_t)r * K + k;
                const bool keep = !drop || ((hbits[e >> 5] >> (e & 31)) & 1);
                const double t = keep ? (double)hx[(size_t)r * ldx + k] * (double)(hw[(size_t)k * p + c] * scale) : 0.0;
                ref += t; mag += fabs(t);
            }
            const double u = fabs((double)ho[(size_t)r * p + c] - ref) / (1.1920929e-7 * (mag > 0 ? mag : 1));
            worst = std::max(worst, u); sumsq += u * u; cnt++;
        }
        printf("  %-44s error vs float64: max %.4f, rms %.4f  (units of eps_f32 * sum|x w|; parity bound 8)\n", what, worst, sqrt(sumsq / cnt));
        return worst;
    };

    // ---- exact-f32 MFMA, persistent LDS-DMA kernel (the product's default since round 3)
    {
        const int n_kg = n_chunks * 4;
        PersistFwdArgs a{dx, ldx, dwp32, dout2, p, m, K, n_chunks, n_rb, nullptr, 0, 0};
        for (int drop = 0; drop < 2; drop++) {
            a.bits = drop ? dbits : nullptr;
            const float scale = drop ? 2.f : 1.f;
            auto run = [&]() {
                pg_pack_w_kernel<<<(n_kg * 256 + 255) / 256, 256>>>(dw, p, K, n_kg, dwp32, scale);
                if (drop) dense_fwd_persist_kernel<true><<<n_cu, 512>>>(a); else dense_fwd_persist_kernel<false><<<n_cu, 512>>>(a);
            };
            const float ms = time_ms(iters, run);
            CK(hipGetLastError());
            printf("f32 MFMA persistent forward, dropout %d:      %.4f ms  %.1f TF/s\n", drop, ms, 2.0 * m * K * p / ms / 1e9);
            check(dout2, "f32 MFMA (k-ordered fmaf chain)", drop, scale);
        }
    }
    // ---- three bf16 planes
    {
        Bx3FwdArgs a{dx, ldx, dwp, dout, p, m, K, n_chunks, n_rb, nullptr, 0};
        if (argc > 4 && !strcmp(argv[4], "price")) {
            // round 6: what bounds the MFMA skeleton (results wrong in every ablation; the full kernel first and last)
            for (int rep = 0; rep < 2; rep++) {
                printf("  full kernel, no dropout:                      %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 0><<<n_cu, 512>>>(a); }));
                printf("  skeleton (ablation 15: MFMAs + barriers):     %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 15><<<n_cu, 512>>>(a); }));
                printf("  skeleton without the barriers (15 + 1024):    %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 15 + 1024><<<n_cu, 512>>>(a); }));
                printf("  full kernel without the barriers (1024):      %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 1024><<<n_cu, 512>>>(a); }));
                printf("  no W LDS reads, no barriers (8 + 1024):       %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 8 + 1024><<<n_cu, 512>>>(a); }));
                printf("  no split (4):                                 %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 4><<<n_cu, 512>>>(a); }));
            }
            return 0;
        }
        // where the time goes: each ablation drops one cost (results wrong): 1 no X loads, 2 no W DMA, 4 no split, 8 no W LDS reads
        printf("  ablation  1 (no X loads):     %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 1><<<n_cu, 512>>>(a); }));
        printf("  ablation  2 (no W DMA):       %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 2><<<n_cu, 512>>>(a); }));
        printf("  ablation  4 (no split):       %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 4><<<n_cu, 512>>>(a); }));
        printf("  ablation  8 (no W LDS reads): %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 8><<<n_cu, 512>>>(a); }));
        a.bits = dcm;
        printf("  dropout, full:                %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 0><<<n_cu, 512>>>(a); }));
        printf("  dropout, keep word loaded last:  %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 128><<<n_cu, 512>>>(a); }));
        printf("  dropout, full (again):        %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 0><<<n_cu, 512>>>(a); }));
        printf("  X loads ahead of the W pieces: dropout %.4f, none %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 256><<<n_cu, 512>>>(a); }),
               time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 256><<<n_cu, 512>>>(a); }));
        for (int rep = 0; rep < 2; rep++)
            printf("  X loads at the top of the chunk: dropout %.4f (default %.4f), none %.4f (default %.4f) ms\n",
                   time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 512><<<n_cu, 512>>>(a); }), time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 0><<<n_cu, 512>>>(a); }),
                   time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 512><<<n_cu, 512>>>(a); }), time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 0><<<n_cu, 512>>>(a); }));
        printf("  dropout, mask not applied:    %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 32><<<n_cu, 512>>>(a); }));
        printf("  dropout, keep words not read: %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 64><<<n_cu, 512>>>(a); }));
        printf("  dropout, neither:             %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 96><<<n_cu, 512>>>(a); }));
        printf("  no dropout:                   %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 0><<<n_cu, 512>>>(a); }));
        // W k-steps in flight ahead of use (the W wait counts this wave's YOUNGER W pieces; X loads still in flight count against it)
        printf("  prefetch distance 6 / 7 / 8 / 9, no dropout: %.4f %.4f %.4f %.4f ms\n",
               time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 0, 8, false, 6><<<n_cu, 512>>>(a); }), time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 0, 8, false, 7><<<n_cu, 512>>>(a); }),
               time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 0, 8, false, 8><<<n_cu, 512>>>(a); }), time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 0, 8, false, 9><<<n_cu, 512>>>(a); }));
        printf("  prefetch distance 6 / 7 / 8 / 9, dropout:    %.4f %.4f %.4f %.4f ms\n",
               time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 0, 8, false, 6><<<n_cu, 512>>>(a); }), time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 0, 8, false, 7><<<n_cu, 512>>>(a); }),
               time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 0, 8, false, 8><<<n_cu, 512>>>(a); }), time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<true, 6, 0, 8, false, 9><<<n_cu, 512>>>(a); }));
        printf("  ablation 15 (all of them):    %.4f ms\n", time_ms(iters, [&]() { dense_fwd_bf16x3_kernel<false, 6, 15><<<n_cu, 512>>>(a); }));
        for (int np = 6; np <= 8; np += 2)
            for (int drop = 0; drop < 2; drop++) {
                a.bits = drop ? dcm : nullptr;
                const float scale = drop ? 2.f : 1.f;
                auto run = [&]() {
                    bx_pack_w_kernel<<<(n_chunks * 512 + 255) / 256, 256>>>(dw, p, K, 2 * n_chunks, dwp, scale);
                    if (np == 6) { if (drop) dense_fwd_bf16x3_kernel<true, 6><<<n_cu, 512>>>(a); else dense_fwd_bf16x3_kernel<false, 6><<<n_cu, 512>>>(a); }
                    else { if (drop) dense_fwd_bf16x3_kernel<true, 8><<<n_cu, 512>>>(a); else dense_fwd_bf16x3_kernel<false, 8><<<n_cu, 512>>>(a); }
                };
                CK(hipMemset(dout, 0xFF, (size_t)m * p * 4));
                const float ms = time_ms(iters, run);
                CK(hipGetLastError());
                printf("bf16x3 forward, %d plane products, dropout %d: %.4f ms  %.1f TF/s algorithmic, %.0f TF/s of bf16 MFMA issued\n", np, drop, ms,
                       2.0 * m * K * p / ms / 1e9, np * 2.0 * m * ldx * p / ms / 1e9);
                char what[64];
                snprintf(what, sizeof what, "bf16x3, %d plane products", np);
                check(dout, what, drop, scale);
            }
    }
    // ================================================================= the forward beside a co-running kernel (the validation lane's situation)
    {
        hipStream_t s1, s2;
        CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        float4 *hog; const size_t hog_n = (size_t)64 << 20;        // 1 GiB
        CK(hipMalloc(&hog, hog_n * 16)); CK(hipMemset(hog, 0, hog_n * 16));
        Bx3FwdArgs a{dx, ldx, dwp, dout, p, m, K, n_chunks, n_rb, nullptr, 1};
        bx_pack_w_kernel<<<2 * n_chunks, 256>>>(dw, p, K, 2 * n_chunks, dwp, 1.f);
        dense_fwd_bf16x3_kernel<false, 6><<<n_cu, 512>>>(a);
        CK(hipDeviceSynchronize());
        std::vector<float> ref((size_t)m * p), got((size_t)m * p);
        CK(hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
        for (int use_lds = 0; use_lds < 4; use_lds++) {
            int bad_runs = 0; long bad_vals = 0; int first_bad_row = -1;
            for (int it = 0; it < 20; it++) {
                CK(hipMemsetAsync(dout, 0xFF, (size_t)m * p * 4, s1));
                hog_kernel<<<2048, 256, 0, s2>>>(hog, hog_n, use_lds & 1);
                if (use_lds & 2) dense_fwd_bf16x3_kernel<false, 6, 16><<<n_cu, 512, 0, s1>>>(a);
                else dense_fwd_bf16x3_kernel<false, 6><<<n_cu, 512, 0, s1>>>(a);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
                long nb = 0;
                for (size_t i = 0; i < got.size(); i++) if (memcmp(&got[i], &ref[i], 4)) { nb++; if (first_bad_row < 0) first_bad_row = (int)(i / p); }
                bad_runs += nb > 0; bad_vals += nb;
            }
            printf("forward beside a co-running kernel (%s%s): %d of 20 runs differ from the run alone (%ld values, first in row %d)\n",
                   (use_lds & 1) ? "holding LDS" : "no LDS", (use_lds & 2) ? ", every wait vmcnt(0)" : "", bad_runs, bad_vals, first_bad_row);
        }
        CK(hipFree(hog));
    }
    // ================================================================= weight gradient dW = X~^T . dH0
    {
        const int ldp = (K + 127) / 128 * 128, n_fr = ldp / 128;
        std::vector<float> hxp((size_t)m * ldp, 0.f), hd((size_t)m * p);
        for (int i = 0; i < m; i++) for (int k = 0; k < K; k++) hxp[(size_t)i * ldp + k] = hx[(size_t)i * ldx + k];
        for (auto &v : hd) v = nd(rng) * 1e-3f;
        float *dxp, *dd, *dslab, *ddw, *ddw2;
        CK(hipMalloc(&dxp, hxp.size() * 4)); CK(hipMalloc(&dd, hd.size() * 4));
        CK(hipMemcpy(dxp, hxp.data(), hxp.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dd, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
        int S = (2 * n_cu) / n_fr; if (S < 1) S = 1;
        int rps = ((m + S - 1) / S + 31) / 32 * 32; if (rps < 32) rps = 32;
        S = (m + rps - 1) / rps;
        CK(hipMalloc(&dslab, (size_t)S * K * p * 4)); CK(hipMalloc(&ddw, (size_t)K * p * 4)); CK(hipMalloc(&ddw2, (size_t)K * p * 4));
        printf("weight gradient: %d feature ranges x %d row splits of %d rows\n", n_fr, S, rps);
        // CPU reference on sampled outputs (every feature of a few columns)
        auto check_dw = [&](const float *d_dw, const char *what, bool drop, float scale) {
            std::vector<float> ho((size_t)K * p);
            hipMemcpy(ho.data(), d_dw, ho.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, sumsq = 0; long cnt = 0, nonfinite = 0;
            for (auto v : ho) nonfinite += !std::isfinite(v);
            if (nonfinite) printf("  !! %ld non-finite outputs\n", nonfinite);
            const int cols[3] = {0, 77, 127};
            for (int f = 0; f < K; f += (f < 40 || f > K - 40) ? 1 : 37) for (int c : cols) {
                double ref = 0, mag = 0;
                for (int i = 0; i < m; i++) {
                    const size_t e = (size_t)i * K + f;
                    const bool keep = !drop || ((hbits[e >> 5] >> (e & 31)) & 1);
                    const double t = keep ? (double)hx[(size_t)i * ldx + f] * scale * (double)hd[(size_t)i * p + c] : 0.0;
                    ref += t; mag += fabs(t);
                }
                const double u = fabs((double)ho[(size_t)f * p + c] - ref) / (1.1920929e-7 * (mag > 0 ? mag : 1));
                worst = std::max(worst, u); sumsq += u * u; cnt++;
            }
            printf("  %-44s error vs float64: max %.4f, rms %.4f  (units of eps_f32 * sum|x dh|; parity bound 8)\n", what, worst, sqrt(sumsq / cnt));
            if (m <= 20000) {                                      // small problems: every output, and a map of the bad 32 x 32 blocks
                int bad[32][4] = {};
                for (int f = 0; f < K; f++) for (int c = 0; c < p; c++) {
                    double ref = 0, mag = 0;
                    for (int i = 0; i < m; i++) {
                        const size_t e = (size_t)i * K + f;
                        const bool keep = !drop || ((hbits[e >> 5] >> (e & 31)) & 1);
                        const double t = keep ? (double)hx[(size_t)i * ldx + f] * scale * (double)hd[(size_t)i * p + c] : 0.0;
                        ref += t; mag += fabs(t);
                    }
                    if (fabs((double)ho[(size_t)f * p + c] - ref) > 8 * 1.1920929e-7 * mag) bad[f / 32][c / 32]++;
                }
                printf("    bad entries per (feature block, column block):");
                for (int fb = 0; fb * 32 < K; fb++) printf(" [%d %d %d %d]", bad[fb][0], bad[fb][1], bad[fb][2], bad[fb][3]);
                printf("\n");
            }
        };
        // exact-f32 MFMA split tiles (the product's kernel since round 1)
        for (int drop = 0; drop < 2; drop++) {
            Tile128Args t;
            t.x = dxp; t.ldx = ldp; t.w = dd; t.ldw = p; t.out = dslab; t.ldo = p; t.m = m; t.K = K; t.p = p;
            t.bits = drop ? dbits : nullptr; t.scale = drop ? 2.f : 1.f; t.rows_per_split = rps; t.split0 = 0; t.relu = 0;
            dim3 grid(S, (K + 127) / 128, 1);
            const float ms = time_ms(iters, [&]() {
                dense_bwd_t128_kernel<4, true><<<grid, 256>>>(t);
                launch_slab_reduce(dslab, S, K, p, p, ddw2, p, 0);
            });
            CK(hipGetLastError());
            printf("f32 MFMA split-tile weight gradient, dropout %d: %.4f ms  %.1f TF/s\n", drop, ms, 2.0 * m * K * p / ms / 1e9);
            check_dw(ddw2, "f32 MFMA", drop, t.scale);
        }
        for (int drop = 0; drop < 2; drop++) {
            Bx3BwdArgs b{dxp, ldp, dd, p, dslab, p, m, K, rps, 0, drop ? dcm : nullptr, drop ? 2.f : 1.f};
            dim3 grid(n_fr, S);
            const float ms = time_ms(iters, [&]() {
                if (drop) dense_bwd_bf16x3_kernel<true, 6><<<grid, 256>>>(b); else dense_bwd_bf16x3_kernel<false, 6><<<grid, 256>>>(b);
                launch_slab_reduce(dslab, S, K, p, p, ddw, p, 0);
            });
            CK(hipGetLastError());
            printf("bf16x3 weight gradient, 6 plane products, dropout %d: %.4f ms  %.1f TF/s algorithmic\n", drop, ms, 2.0 * m * K * p / ms / 1e9);
            check_dw(ddw, "bf16x3, 6 plane products", drop, b.scale);
        }
        {   // the order in which a step's loads are issued (kernel alone, no slab sum)
            Bx3BwdArgs b{dxp, ldp, dd, p, dslab, p, m, K, rps, 0, dcm, 2.f};
            dim3 grid(n_fr, S);
            for (int rep = 0; rep < 3; rep++)    // (rotated: the clock drifts with what ran before)
                printf("  weight gradient with dropout, kernel alone; X, dH0, keep word: %.4f  keep word first: %.4f  dH0 first: %.4f  interleaved: %.4f  interleaved + keep word first: %.4f ms\n",
                       time_ms(iters, [&]() { dense_bwd_bf16x3_kernel<true, 6, 0><<<grid, 256>>>(b); }),
                       time_ms(iters, [&]() { dense_bwd_bf16x3_kernel<true, 6, 1><<<grid, 256>>>(b); }), time_ms(iters, [&]() { dense_bwd_bf16x3_kernel<true, 6, 2><<<grid, 256>>>(b); }),
                       time_ms(iters, [&]() { dense_bwd_bf16x3_kernel<true, 6, 4><<<grid, 256>>>(b); }), time_ms(iters, [&]() { dense_bwd_bf16x3_kernel<true, 6, 5><<<grid, 256>>>(b); }));
            b.bits = nullptr; b.scale = 1.f;
            printf("  weight gradient without dropout, kernel alone; X, dH0: %.4f  dH0 first: %.4f  interleaved: %.4f ms\n",
                   time_ms(iters, [&]() { dense_bwd_bf16x3_kernel<false, 6, 0><<<grid, 256>>>(b); }), time_ms(iters, [&]() { dense_bwd_bf16x3_kernel<false, 6, 2><<<grid, 256>>>(b); }),
                   time_ms(iters, [&]() { dense_bwd_bf16x3_kernel<false, 6, 4><<<grid, 256>>>(b); }));
        }
    }
    return 0;
}

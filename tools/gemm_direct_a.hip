// Probe (tools only; not product code): the 128 x 128 forward tile with the A operand read straight from global memory
// into registers (no LDS image of A) and the B operand double-buffered in LDS with ONE barrier per K chunk.
// Within a 32-wide K chunk the half-wave kq takes k = 16 kq + j (j = 0..15) so that a lane's 16 A values are contiguous.
// hipcc --offload-arch=gfx950 -O3 tools/gemm_direct_a.hip -o build/gemm_direct_a
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)
constexpr int BN = 128, BK = 32;

template <int WM>      // WM x 2 waves of 32 x 64
__global__ __launch_bounds__(128 * WM) void direct_a(const float *x, int ldx, const float *w, int ldw, float *out, int ldo, int m, int K) {
    constexpr int NTH = 128 * WM, BM = 32 * WM, BP = (BK * BN / 4 + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kq = lane >> 5;
    const int row_base = blockIdx.x * BM;
    const float *ap = x + (size_t)min(row_base + wm * 32 + li, m - 1) * ldx + 16 * kq;
    float4 cur[4], nxt[4], breg[BP];
    auto fetch_a = [&](int k0, float4 (&r)[4]) {
#pragma unroll
        for (int q = 0; q < 4; q++) r[q] = *reinterpret_cast<const float4 *>(ap + k0 + 4 * q);
    };
    auto fetch_b = [&](int k0) {
#pragma unroll
        for (int pc = 0; pc < BP; pc++) {
            const int idx = min((pc * NTH + tid) * 4, BK * BN - 4), k = idx / BN, c = idx % BN;
            breg[pc] = *reinterpret_cast<const float4 *>(w + (size_t)min(k0 + k, K - 1) * ldw + c);
        }
    };
    auto stash_b = [&](int k0, int buf) {
#pragma unroll
        for (int pc = 0; pc < BP; pc++) {
            const int idx = (pc * NTH + tid) * 4;
            if (idx >= BK * BN) continue;
            float4 v = breg[pc];
            if (k0 + idx / BN >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(&Bs[buf][idx]) = v;
        }
    };
    f32x16 acc[2];
    for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const int nch = (K + BK - 1) / BK;
    fetch_b(0);
    stash_b(0, 0);
    fetch_a(0, cur);
    if (nch > 1) fetch_b(BK);
    for (int c = 0; c < nch; c++) {
        __syncthreads();                                   // Bs[c & 1] is complete; nobody still reads Bs[(c + 1) & 1]
        if (c + 1 < nch) stash_b((c + 1) * BK, (c + 1) & 1);
        if (c + 2 < nch) fetch_b((c + 2) * BK);
        fetch_a(min(c + 1, nch - 1) * BK, nxt);            // unconditional (clamped): no load inside a branch
        const float *Bp = &Bs[c & 1][(16 * kq) * BN + wn * 64 + li];
        const float av[16] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w,
                              cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w};
        float b0 = Bp[0], b1 = Bp[32];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            float n0 = 0.f, n1 = 0.f;
            if (j + 1 < 16) { n0 = Bp[(j + 1) * BN]; n1 = Bp[(j + 1) * BN + 32]; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = MFMA32(av[j], b0, acc[0]);
            acc[1] = MFMA32(av[j], b1, acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            b0 = n0; b1 = n1;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
    }
    for (int j = 0; j < 2; j++) {
        const int col = wn * 64 + j * 32 + li;
        for (int r = 0; r < 16; r++) {
            const int row = row_base + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
            if (row < m) out[(size_t)row * ldo + col] = acc[j][r];
        }
    }
}

int main() {
    const int m = 232965, K = 602, ldx = 640;
    float *x, *w, *out;
    std::vector<float> hx((size_t)4096 * ldx), hw((size_t)K * 128);
    for (auto &v : hx) v = (float)(rand() % 17 - 8) / 8.f;
    for (auto &v : hw) v = (float)(rand() % 13 - 6) / 4.f;
    hipMalloc(&x, (size_t)m * ldx * 4); hipMalloc(&w, (size_t)K * 128 * 4); hipMalloc(&out, (size_t)m * 128 * 4);
    hipMemset(x, 0, (size_t)m * ldx * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);         // first 4096 rows carry data (checked below)
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char *tag) {
        for (int it = 0; it < 3; it++) launch();
        hipEventRecord(e0);
        for (int it = 0; it < 20; it++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-40s %.3f ms  %.1f TF\n", tag, ms, 2.0 * m * K * 128 / ms / 1e9);
    };
    time([&] { direct_a<4><<<(m + 127) / 128, 512>>>(x, ldx, w, 128, out, 128, m, K); }, "direct A, 4 x 2 waves, 128 rows");
    // check a few rows against the host (integers / small dyadic values: exact in f32)
    std::vector<float> ho((size_t)256 * 128);
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int r = 0; r < 256; r++) for (int c = 0; c < 128; c += 17) {
        double s = 0; for (int k = 0; k < K; k++) s += (double)hx[(size_t)r * ldx + k] * hw[(size_t)k * 128 + c];
        worst = std::max(worst, std::abs(s - ho[(size_t)r * 128 + c]));
    }
    printf("max abs error on sampled outputs: %g\n", worst);
    time([&] { direct_a<5><<<(m + 159) / 160, 640>>>(x, ldx, w, 128, out, 128, m, K); }, "direct A, 5 x 2 waves, 160 rows");
    time([&] { direct_a<3><<<(m + 95) / 96, 384>>>(x, ldx, w, 128, out, 128, m, K); }, "direct A, 3 x 2 waves, 96 rows");
    time([&] { direct_a<2><<<(m + 63) / 64, 256>>>(x, ldx, w, 128, out, 128, m, K); }, "direct A, 2 x 2 waves, 64 rows");
    return 0;
}

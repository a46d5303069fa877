// Ablation probe for the 128x128 f32-MFMA forward tile (tools only; not product code).
//   ABL bit 0: no global refetch (registers keep chunk 0)     bit 1: no LDS operand reads in the k loop
//   ABL bit 2: no LDS writes (stash skipped)                   bit 3: no MFMA (adds instead)
//   ABL bit 4: no barriers
// hipcc --offload-arch=gfx950 -O3 -Iinclude -Icuda_gcn_amd/csrc tools/gemm_ablate.hip -o build/gemm_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((a_), (b_), (c_), 0, 0, 0)
constexpr int BN = 128;

template <int ABL, int BM, int BK>
__global__ __launch_bounds__(256) void probe(const float *x, int ldx, const float *w, int ldw, float *out, int ldo, int m, int K) {
    constexpr int ALD = BK + 1, MI = BM / 64, AP = BM * BK / 1024, BP = BK * BN / 1024, LPR = BK / 4;
    __shared__ float As[BM * ALD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kq = lane >> 5;
    const int row_base = blockIdx.x * BM;
    float4 areg[AP], breg[BP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int pc = 0; pc < AP; pc++) {
            const int idx = pc * 256 + tid, r = idx / LPR, c = (idx % LPR) * 4;
            areg[pc] = *reinterpret_cast<const float4 *>(x + (size_t)min(row_base + r, m - 1) * ldx + k0 + c);
        }
#pragma unroll
        for (int pc = 0; pc < BP; pc++) {
            const int idx = (pc * 256 + tid) * 4, k = idx / BN, c = idx % BN;
            breg[pc] = *reinterpret_cast<const float4 *>(w + (size_t)min(k0 + k, K - 1) * ldw + c);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int pc = 0; pc < AP; pc++) {
            const int idx = pc * 256 + tid, r = idx / LPR, c = (idx % LPR) * 4;
            As[r * ALD + c] = areg[pc].x; As[r * ALD + c + 1] = areg[pc].y; As[r * ALD + c + 2] = areg[pc].z; As[r * ALD + c + 3] = areg[pc].w;
        }
#pragma unroll
        for (int pc = 0; pc < BP; pc++) *reinterpret_cast<float4 *>(&Bs[(pc * 256 + tid) * 4]) = breg[pc];
    };
    f32x16 acc[MI][2];
    for (int i = 0; i < MI; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    fetch(0);
    if (ABL & 4) { stash(); __syncthreads(); }
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (!(ABL & 16)) __syncthreads();
        if (!(ABL & 4)) stash();
        if (!(ABL & 16)) __syncthreads();
        if (!(ABL & 1) && k0 + BK < K) fetch(k0 + BK);
        const float *Ap = &As[(wm * 32 * MI + li) * ALD + kq];
        const float *Bp = &Bs[kq * BN + wn * 64 + li];
        float ra0 = areg[0].x, ra1 = areg[AP - 1].y, rb0 = breg[0].x, rb1 = breg[1].y;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0, a1 = 0, b0, b1;
            if (ABL & 2) { a0 = ra0; a1 = ra1; b0 = rb0; b1 = rb1; }
            else { a0 = Ap[kk]; if (MI == 2) a1 = Ap[32 * ALD + kk]; b0 = Bp[kk * BN]; b1 = Bp[kk * BN + 32]; }
            if (ABL & 8) {
                acc[0][0][kk & 15] += a0 * b0; acc[0][1][kk & 15] += a0 * b1;
                if (MI == 2) { acc[MI - 1][0][kk & 15] += a1 * b0; acc[MI - 1][1][kk & 15] += a1 * b1; }
            } else {
                acc[0][0] = MFMA32(a0, b0, acc[0][0]); acc[0][1] = MFMA32(a0, b1, acc[0][1]);
                if (MI == 2) { acc[MI - 1][0] = MFMA32(a1, b0, acc[MI - 1][0]); acc[MI - 1][1] = MFMA32(a1, b1, acc[MI - 1][1]); }
            }
        }
    }
    for (int i = 0; i < MI; i++) for (int j = 0; j < 2; j++) {
        const int col = wn * 64 + j * 32 + li;
        for (int r = 0; r < 16; r++) {
            const int row = row_base + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
            if (row < m) out[(size_t)row * ldo + col] = acc[i][j][r];
        }
    }
}

template <int ABL, int BM = 128, int BK = 32>
void run(const char *tag, const float *x, int ldx, const float *w, float *out, int m, int K, int dyn_lds = 0) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (m + BM - 1) / BM;
    for (int it = 0; it < 3; it++) probe<ABL, BM, BK><<<grid, 256, dyn_lds>>>(x, ldx, w, 128, out, 128, m, K);
    hipEventRecord(e0);
    for (int it = 0; it < 10; it++) probe<ABL, BM, BK><<<grid, 256, dyn_lds>>>(x, ldx, w, 128, out, 128, m, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-58s %.3f ms  %.1f TF-equivalent\n", tag, ms, 2.0 * m * K * 128 / ms / 1e9);
}

int main() {
    const int m = 232965, K = 602, ldx = 640;
    float *x, *w, *out;
    hipMalloc(&x, (size_t)m * ldx * 4); hipMalloc(&w, (size_t)K * 128 * 4); hipMalloc(&out, (size_t)m * 128 * 4);
    hipMemset(x, 0, (size_t)m * ldx * 4); hipMemset(w, 0, (size_t)K * 128 * 4);
    run<0>("full", x, ldx, w, out, m, K);
    run<1>("no global refetch", x, ldx, w, out, m, K);
    run<1 | 4>("no refetch, no LDS writes", x, ldx, w, out, m, K);
    run<1 | 4 | 16>("no refetch, no LDS writes, no barriers", x, ldx, w, out, m, K);
    run<1 | 2 | 4 | 16>("MFMA only (no LDS reads either)", x, ldx, w, out, m, K);
    run<1 | 2>("no refetch, no LDS reads (writes + barriers + MFMA)", x, ldx, w, out, m, K);
    run<8>("everything but MFMA (fma instead)", x, ldx, w, out, m, K);
    run<2>("full minus LDS reads", x, ldx, w, out, m, K);
    run<0>("full, occupancy capped at 2 workgroups per CU (22 KB dummy LDS)", x, ldx, w, out, m, K, 22 * 1024);
    run<1 | 2 | 4 | 16>("MFMA only, occupancy 2", x, ldx, w, out, m, K, 22 * 1024);
    run<0>("full, occupancy capped at 1 workgroup per CU", x, ldx, w, out, m, K, 60 * 1024);
    run<0, 128, 64>("full, BK=64", x, ldx, w, out, m, K);
    run<0, 64, 32>("full, BM=64", x, ldx, w, out, m, K);
    run<0, 64, 64>("full, BM=64 BK=64", x, ldx, w, out, m, K);
    run<1 | 2 | 4 | 16, 64, 32>("MFMA only, BM=64", x, ldx, w, out, m, K);
    run<8, 64, 64>("everything but MFMA, BM=64 BK=64", x, ldx, w, out, m, K);
    run<8, 128, 64>("everything but MFMA, BM=128 BK=64", x, ldx, w, out, m, K);
    return 0;
}

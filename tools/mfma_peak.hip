// f32 MFMA issue-rate probe: register-resident operands, 4 independent 32x32 accumulators per wave,
// W waves per SIMD.  Prints TFLOP/s for 32x32x2 and 16x16x4.  (hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    if (SHAPE == 32) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
            }
        }
        float s = 0;
        for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 c[16];
        for (int q = 0; q < 16; q++) c[q] = {0, 0, 0, 0};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int q = 0; q < 16; q++) c[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[q], 0, 0, 0);
        }
        float s = 0;
        for (int q = 0; q < 16; q++) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape : {32, 16})
        for (int wg_per_cu : {1, 2, 3, 4}) {
            const int blocks = 256 * wg_per_cu, iters = 2000;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (shape == 32) probe<32><<<blocks, 256>>>(out, iters, 1.f, 1e-3f);
                else probe<16><<<blocks, 256>>>(out, iters, 1.f, 1e-3f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per wave per iter: 32: 64 MFMA x 4096 FLOP; 16: 128 MFMA x 2048 FLOP
            const double flop = (double)blocks * 4 * iters * 64 * 4096;
            printf("mfma_f32_%dx%d: %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", shape, shape, wg_per_cu, ms, flop / ms / 1e9);
        }
    return 0;
}

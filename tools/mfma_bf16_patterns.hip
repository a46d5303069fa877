// bf16 MFMA issue patterns (round 6): what the pipe delivers for the accumulate orders the bf16x3 kernels could use, with
// register-resident operands and nothing else in the loop.  v_mfma_f32_32x32x16_bf16: 32768 FLOP, 8 passes.
//   pattern 0: chains of 6 MFMAs into ONE accumulator, 4 accumulators in turn (dense_bf16x3.h's `mac`), W waves per SIMD
//   pattern 1: the same 24 MFMAs round-robin over the 4 accumulators (no two consecutive MFMAs dependent)
//   pattern 2: 8 accumulators (two row blocks per wave), chains of 6
//   pattern 3: 8 accumulators round-robin
//   pattern 4: 4 accumulators in PAIRS: the six plane products of two accumulators interleaved (dependent MFMAs two apart)
//   pattern 5: chains of 6 with ONE independent MFMA of the next accumulator slipped behind each (distance 2 without a second B buffer: a0 b0 a1 b1 .. the same as 4, but 3 of 6 / 3 of 6 split: a0 a1 a2 | b0 a3 b1 a4 b2 a5 | b3 ..) — skewed pairs
// prints time, TFLOP/s and the share of the 2.5 PF pipe for 1 and 2 waves per SIMD (256 workgroups: one per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_patterns.hip -o build/mfma_bf16_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
// inline asm: program order is kept (hipcc re-orders independent builtin MFMAs and re-materialises uniform operands between them)
#define MFMA(a, b, c) ([&]() { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); return c; }())

template <int PAT, int NW>
__global__ __launch_bounds__(64 * NW) void probe(float *out, int iters, short seed) {
    bf16x8 a[6], b[6];
    for (int p = 0; p < 6; p++) for (int j = 0; j < 8; j++) { a[p][j] = (short)(0x3F80 + ((seed + threadIdx.x * 7 + p * 3 + j) & 0x7F)); b[p][j] = (short)(0x3F80 + ((seed + threadIdx.x * 5 + j + 3 * p) & 0x7F)); }   // bf16 values in [1, 2): per lane, in VGPRs
    constexpr int NA = (PAT == 2 || PAT == 3) ? 8 : 4;
    f32x16 c[NA];
    for (int n = 0; n < NA; n++) for (int r = 0; r < 16; r++) c[n][r] = 0.f;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        if (PAT == 4) {
#pragma unroll
            for (int n = 0; n < NA; n += 2)
#pragma unroll
                for (int p = 0; p < 6; p++) { c[n] = MFMA(a[p], b[p], c[n]); c[n + 1] = MFMA(a[p], b[p], c[n + 1]); }
        } else if (PAT == 5) {                                  // skewed: accumulator n's last three products beside accumulator n+1's first three
#pragma unroll
            for (int n = 0; n < NA; n++) {
#pragma unroll
                for (int p = 0; p < 3; p++) { c[n] = MFMA(a[p + 3], b[p + 3], c[n]); c[(n + 1) % NA] = MFMA(a[p], b[p], c[(n + 1) % NA]); }
            }
        } else if (PAT == 0 || PAT == 2) {
#pragma unroll
            for (int n = 0; n < NA; n++)
#pragma unroll
                for (int p = 0; p < 6; p++) c[n] = MFMA(a[p], b[p], c[n]);
        } else {
#pragma unroll
            for (int p = 0; p < 6; p++)
#pragma unroll
                for (int n = 0; n < NA; n++) c[n] = MFMA(a[p], b[p], c[n]);
        }
    }
    float s = 0;
    for (int n = 0; n < NA; n++) for (int r = 0; r < 16; r++) s += c[n][r];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
}

// the same wave tile (32 rows x 128 columns, f32 accumulators: 64 registers) from v_mfma_f32_16x16x32_bf16: 16 tiles of 4 registers,
// 96 MFMAs of 16 cycles per 32-deep k-chunk instead of 48 of 32 (MI355X_MICROARCH.md, DVFS give-back item 7: the chip may hold a
// higher clock on this shape).  Order: per 16-column block, the two row halves x six plane products (12 MFMAs, dependent two apart).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int NW>
__global__ __launch_bounds__(64 * NW) void probe16(float *out, int iters, short seed) {
    bf16x8 a[2][3], b[3];
    for (int h = 0; h < 2; h++) for (int p = 0; p < 3; p++) for (int j = 0; j < 8; j++) a[h][p][j] = (short)(0x3F80 + ((seed + threadIdx.x * 7 + p * 3 + j + 11 * h) & 0x7F));
    for (int p = 0; p < 3; p++) for (int j = 0; j < 8; j++) b[p][j] = (short)(0x3F80 + ((seed + threadIdx.x * 5 + j + 3 * p) & 0x7F));
    f32x4v c[16];
    for (int n = 0; n < 16; n++) for (int r = 0; r < 4; r++) c[n][r] = 0.f;
#define MFMA16(a_, b_, c_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c_) : "v"(a_), "v"(b_))
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int cb = 0; cb < 8; cb++) {
            // six plane products (lo.hi, mid.mid, hi.lo, mid.hi, hi.mid, hi.hi) for both row halves
            MFMA16(a[0][2], b[0], c[2 * cb]); MFMA16(a[1][2], b[0], c[2 * cb + 1]);
            MFMA16(a[0][1], b[1], c[2 * cb]); MFMA16(a[1][1], b[1], c[2 * cb + 1]);
            MFMA16(a[0][0], b[2], c[2 * cb]); MFMA16(a[1][0], b[2], c[2 * cb + 1]);
            MFMA16(a[0][1], b[0], c[2 * cb]); MFMA16(a[1][1], b[0], c[2 * cb + 1]);
            MFMA16(a[0][0], b[1], c[2 * cb]); MFMA16(a[1][0], b[1], c[2 * cb + 1]);
            MFMA16(a[0][0], b[0], c[2 * cb]); MFMA16(a[1][0], b[0], c[2 * cb + 1]);
        }
    }
    float s = 0;
    for (int n = 0; n < 16; n++) for (int r = 0; r < 4; r++) s += c[n][r];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
}
template <int NW>
static void run16(float *out, const char *what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        probe16<NW><<<256, 64 * NW>>>(out, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 256.0 * NW * iters * 96 * 16384.0;
    printf("%-58s %d waves/SIMD: %.3f ms  %.0f TFLOP/s  %.3f of 2.5 PF\n", what, NW / 4, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
}

template <int PAT, int NW>
static void run(float *out, const char *what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, na = (PAT == 2 || PAT == 3) ? 8 : 4;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        probe<PAT, NW><<<256, 64 * NW>>>(out, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 256.0 * NW * iters * na * 6 * 32768.0;
    printf("%-58s %d waves/SIMD: %.3f ms  %.0f TFLOP/s  %.3f of 2.5 PF\n", what, NW / 4, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    run<0, 4>(out, "4 accumulators, chains of 6 (the kernels' order)");
    run<0, 8>(out, "4 accumulators, chains of 6 (the kernels' order)");
    run<1, 4>(out, "4 accumulators, round-robin");
    run<1, 8>(out, "4 accumulators, round-robin");
    run<2, 4>(out, "8 accumulators, chains of 6");
    run<3, 4>(out, "8 accumulators, round-robin");
    run<3, 8>(out, "8 accumulators, round-robin");
    run<4, 8>(out, "4 accumulators, interleaved in pairs");
    run<5, 8>(out, "4 accumulators, skewed pairs (3 + 3)");
    run16<8>(out, "16x16x32: 16 tiles, pairs of row halves");
    run16<4>(out, "16x16x32: 16 tiles, pairs of row halves");
    run<0, 8>(out, "4 accumulators, chains of 6 (again)");
    run<1, 8>(out, "4 accumulators, round-robin (again)");
    run16<8>(out, "16x16x32: 16 tiles, pairs of row halves (again)");
    run<0, 8>(out, "4 accumulators, chains of 6 (third)");
    return 0;
}

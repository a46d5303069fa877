// grid_barrier.hip — what does a grid-wide phase boundary cost INSIDE a kernel on this chip?  (Design input for the fused
// small-graph epoch: Cora / Citeseer / Pubmed epochs are ~20 dependent launches of 4-5 us each.)
// G workgroups of T threads, launched cooperatively (co-residency guaranteed), run K phases; each phase writes a little data,
// then: release fence (agent scope) -> arrive on a counter -> spin until all G have arrived -> acquire fence -> read what
// another workgroup wrote (checked).  Bounded spin: a barrier that never completes sets an error word and every later
// barrier falls through, so the grid always drains.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.hip -o build/grid_barrier && build/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

struct Args { unsigned *count; unsigned *gen; unsigned *err; float *data; int phases; int payload; int fences; };

__device__ inline void grid_barrier(const Args &a, unsigned &local_gen, int fences) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned target = local_gen + 1;
        const unsigned prev = __hip_atomic_fetch_add(a.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(a.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (++spins > (1u << 22)) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    local_gen++;
    __syncthreads();
}

__global__ void phases_kernel(Args a) {
    unsigned gen = 0;
    float check = 0.f;
    for (int p = 0; p < a.phases; p++) {
        // every workgroup writes `payload` floats of its own segment, then reads the next workgroup's segment after the barrier
        for (int i = threadIdx.x; i < a.payload; i += blockDim.x) a.data[(size_t)blockIdx.x * a.payload + i] = (float)(p + 1);
        grid_barrier(a, gen, a.fences);
        const int other = (blockIdx.x + 1) % gridDim.x;
        for (int i = threadIdx.x; i < a.payload; i += blockDim.x) check += a.data[(size_t)other * a.payload + i] - (float)(p + 1);
        grid_barrier(a, gen, a.fences);          // (the next phase overwrites what was just read)
    }
    if (check != 0.f) __hip_atomic_store(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // stale data seen
}

int main() {
    unsigned *d_ctrl;
    CK(hipMalloc((void **)&d_ctrl, 64 * sizeof(unsigned)));
    float *d_data;
    CK(hipMalloc((void **)&d_data, (size_t)1024 * 65536 * sizeof(float) / 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 200;
    printf("%6s %6s %8s %7s | us per barrier (2 per phase), error word\n", "WGs", "thr", "payload", "fences");
    for (int fences = 1; fences >= 0; fences--)
        for (int T : {256, 1024})
            for (int G : {8, 32, 64, 128, 256})
                for (int payload : {64, 4096}) {
                    CK(hipMemset(d_ctrl, 0, 64 * sizeof(unsigned)));
                    Args a{d_ctrl, d_ctrl + 16, d_ctrl + 32, d_data, phases, payload, fences};
                    void *params[] = {&a};
                    float best = 1e9f;
                    for (int rep = 0; rep < 3; rep++) {
                        CK(hipMemset(d_ctrl, 0, 64 * sizeof(unsigned)));
                        CK(hipEventRecord(e0, 0));
                        CK(hipLaunchCooperativeKernel((const void *)phases_kernel, dim3(G), dim3(T), params, 0, 0));
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        best = ms < best ? ms : best;
                    }
                    unsigned err;
                    CK(hipMemcpy(&err, d_ctrl + 32, sizeof err, hipMemcpyDeviceToHost));
                    printf("%6d %6d %8d %7d | %7.2f   err=%u%s\n", G, T, payload, fences, 1e3f * best / (2 * phases), err,
                           (!fences && err == 2) ? " (stale data without fences: expected)" : "");
                }
    return 0;
}

// How much static LDS may one workgroup have on this chip?  (round 5: a 144 KB kernel aborted the process at its first sync.)
//   hipcc --offload-arch=gfx950 -O2 tools/lds_limit.hip -o build/lds_limit && build/lds_limit
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB>
__global__ __launch_bounds__(256) void lds_kernel(unsigned *out) {
    __shared__ unsigned buf[KB * 256];
    for (int i = threadIdx.x; i < KB * 256; i += 256) buf[i] = i * 2654435761u;
    __syncthreads();
    unsigned s = 0;
    for (int i = threadIdx.x; i < KB * 256; i += 256) s ^= buf[(i * 7 + 13) % (KB * 256)];
    atomicXor(out, s);
}
template <int KB>
void run(unsigned *d) {
    hipMemset(d, 0, 4);
    lds_kernel<KB><<<4, 256>>>(d);
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%3d KB static LDS per workgroup: launch %s, sync %s, checksum %08x\n", KB, hipGetErrorString(e1), hipGetErrorString(e2), h);
    fflush(stdout);
}
int main() {
    unsigned *d;
    hipMalloc(&d, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", (size_t)p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
    run<64>(d); run<120>(d); run<128>(d); run<132>(d); run<144>(d); run<156>(d); run<160>(d);
    return 0;
}

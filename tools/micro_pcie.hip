// Host <-> HBM rate for the vectors of the host-buffer entry points (2 x 170 MB each way per gradient at 128^3): hipMemcpyAsync (SDMA) against a
// copy kernel that reads / writes the page-locked host buffer directly (zero-copy over PCIe).   usage: micro_pcie [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void copy_k(d2* __restrict__ dst, const d2* __restrict__ src, size_t n2) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 170) << 20;
    void *h, *d; CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); CK(hipMalloc(&d, bytes));
    memset(h, 1, bytes);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* what, auto f) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) { hipEventRecord(e0, s); f(); hipEventRecord(e1, s); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-44s %7.2f ms  %6.1f GB/s\n", what, best, bytes / (best * 1e-3) / 1e9);
    };
    timeit("hipMemcpyAsync H2D", [&]() { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); });
    timeit("hipMemcpyAsync D2H", [&]() { hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); });
    for (int wg : {256, 1024, 4096}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy kernel H2D (%d workgroups)", wg);
        timeit(nm, [&]() { hipLaunchKernelGGL(copy_k, dim3(wg), dim3(256), 0, s, (d2*)d, (const d2*)h, bytes / 16); });
        snprintf(nm, sizeof nm, "copy kernel D2H (%d workgroups)", wg);
        timeit(nm, [&]() { hipLaunchKernelGGL(copy_k, dim3(wg), dim3(256), 0, s, (d2*)h, (const d2*)d, bytes / 16); });
    }
    // both directions at once (the gradient of one component going out while ... nothing comes in: only to know the link's duplex rate)
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    void *h2, *d2b; CK(hipHostMalloc(&h2, bytes, hipHostMallocDefault)); CK(hipMalloc(&d2b, bytes));
    timeit("hipMemcpyAsync H2D + D2H concurrently", [&]() { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipMemcpyAsync(h2, d2b, bytes, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); });
    return 0;
}

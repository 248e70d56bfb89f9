// Out-of-line device calls on gfx950: which ingredient of the SHB23 any-N work area breaks when a callee is NOT inlined?
// (round-2 finding, csrc/shb23.hip: with dct3<0> out of line the Continuous adjoint returned garbage.)
//
// One ingredient per mode, one workgroup, the result checked on the host; every callee is __attribute__((noinline)) and the
// test FAILS TO BUILD its point if the compiler inlines it anyway (tools/run_outline_abi.sh greps the ISA for s_swappc_b64).
//   mode 1  flat load/store through a pointer into the CALLER'S PRIVATE memory (the work-area struct lives in scratch)
//   mode 2  flat load/store through LDS pointers held in a struct in private memory (static LDS, < 64 KB)
//   mode 3  the same on dynamic LDS above the 64-KB default (opt-in), touching addresses beyond 64 KB
//   mode 4  mode 3 + s_barrier inside the callee, 1024 threads, a data exchange between waves through flat LDS stores
//   mode 5  a run-time-indexed int table in the struct (AnyPlan::r[20]) driving a loop with barriers in the callee
// usage: micro_outline_abi MODE      (exit code 0 = pass)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

struct Work {
    double *a, *b;        // LDS (modes 2-5) or unused
    int n;
    int r[20];            // run-time-indexed: forces the struct into scratch
    int nr;
};

// mode 1: private memory only
__device__ __attribute__((noinline)) int sum_tab(const Work& w, int* out_priv) {
    int s = 0;
    for (int i = 0; i < w.nr; ++i) s += w.r[i];
    *out_priv = s * 2;
    return s;
}
__global__ void k_mode1(int* out, int nr) {
    Work w{nullptr, nullptr, 0, {}, nr};
    for (int i = 0; i < 20; ++i) w.r[i] = i + (int)threadIdx.x;
    int twice = -1;
    const int s = sum_tab(w, &twice);
    out[2 * threadIdx.x] = s;
    out[2 * threadIdx.x + 1] = twice;
}

// modes 2-4: LDS pointers inside the struct
template <bool BARRIER> __device__ __attribute__((noinline)) void rotate(Work& w, int tid, int nthr) {
    for (int i = tid; i < w.n; i += nthr) w.b[(i + 1 == w.n) ? 0 : i + 1] = 2.0 * w.a[i];        // b = 2 * a rotated by one
    if (BARRIER) {
        __syncthreads();
        for (int i = tid; i < w.n; i += nthr) w.a[i] = w.b[w.n - 1 - i] + 1.0;                       // a = reverse(b) + 1: reads other waves' stores
        __syncthreads();
    }
}
template <bool BARRIER> __global__ __launch_bounds__(1024) void k_lds(double* out, int n, int dynamic) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double stat[2048];
    double* base = dynamic ? reinterpret_cast<double*>(smem) : stat;
    Work w{base, base + n, n, {}, 0};
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < n; i += nthr) w.a[i] = (double)i;
    __syncthreads();
    rotate<BARRIER>(w, tid, nthr);
    __syncthreads();
    for (int i = tid; i < n; i += nthr) { out[i] = w.a[i]; out[n + i] = w.b[i]; }
}

// mode 5: the shape of any_fft: a loop over w.r[st] with a barrier per stage, ping-pong between a and b
__device__ __attribute__((noinline)) double* stages(Work& w, int tid, int nthr) {
    double *src = w.a, *dst = w.b;
    for (int st = 0; st < w.nr; ++st) {
        const int R = w.r[st];
        for (int i = tid; i < w.n; i += nthr) dst[i] = src[(i + R) % w.n] + (double)R;
        __syncthreads();
        double* t = src; src = dst; dst = t;
    }
    return src;
}
__global__ __launch_bounds__(1024) void k_mode5(double* out, int n, Work proto) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* base = reinterpret_cast<double*>(smem);
    Work w = proto;
    w.a = base; w.b = base + n; w.n = n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < n; i += nthr) w.a[i] = (double)i;
    __syncthreads();
    const double* res = stages(w, tid, nthr);
    for (int i = tid; i < n; i += nthr) out[i] = res[i];
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    int bad = 0;
    if (mode == 1) {
        int* d; CK(hipMalloc(&d, 2 * 256 * sizeof(int)));
        hipLaunchKernelGGL(k_mode1, dim3(1), dim3(256), 0, 0, d, 20);
        CK(hipDeviceSynchronize());
        std::vector<int> h(512); CK(hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int t = 0; t < 256; ++t) { const int s = 190 + 20 * t; if (h[2 * t] != s || h[2 * t + 1] != 2 * s) { if (bad++ < 5) printf("  thread %d: got %d %d expected %d %d\n", t, h[2 * t], h[2 * t + 1], s, 2 * s); } }
    } else if (mode >= 2 && mode <= 4) {
        const int n = (mode == 2) ? 1000 : 9000;                    // mode 3/4: 2 * 9000 * 8 B = 144 KB of LDS, addresses beyond 64 KB
        const size_t lds = (mode == 2) ? 0 : (size_t)2 * n * sizeof(double);
        double* d; CK(hipMalloc(&d, 2 * n * sizeof(double)));
        if (mode == 4) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_lds<true>, dim3(1), dim3(1024), lds, 0, d, n, 1);
        } else {
            if (lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_lds<false>, dim3(1), dim3(1024), lds, 0, d, n, lds ? 1 : 0);
        }
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        std::vector<double> h(2 * n); CK(hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> a(n), b(n);
        for (int i = 0; i < n; ++i) { a[i] = i; }
        for (int i = 0; i < n; ++i) b[(i + 1 == n) ? 0 : i + 1] = 2.0 * a[i];
        if (mode == 4) for (int i = 0; i < n; ++i) a[i] = b[n - 1 - i] + 1.0;
        for (int i = 0; i < n; ++i) if (h[i] != a[i] || h[n + i] != b[i]) { if (bad++ < 5) printf("  i %d: got a %g b %g expected %g %g\n", i, h[i], h[n + i], a[i], b[i]); }
    } else if (mode == 5) {
        const int n = 9000;
        const size_t lds = (size_t)2 * n * sizeof(double);
        Work proto{}; proto.nr = 7;
        const int rr[7] = {4, 4, 2, 3, 5, 7, 11};
        for (int i = 0; i < 7; ++i) proto.r[i] = rr[i];
        double* d; CK(hipMalloc(&d, n * sizeof(double)));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mode5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mode5, dim3(1), dim3(1024), lds, 0, d, n, proto);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        std::vector<double> h(n), x(n), y(n); CK(hipMemcpy(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) x[i] = i;
        for (int st = 0; st < 7; ++st) { for (int i = 0; i < n; ++i) y[i] = x[(i + rr[st]) % n] + rr[st]; x.swap(y); }
        for (int i = 0; i < n; ++i) if (h[i] != x[i]) { if (bad++ < 5) printf("  i %d: got %g expected %g\n", i, h[i], x[i]); }
    } else { printf("unknown mode %d\n", mode); return 2; }
    printf("mode %d: %s (%d mismatches)\n", mode, bad ? "FAIL" : "PASS", bad);
    return bad ? 1 : 0;
}

// Device-resident vectors for the optimiser (SURVEY.md section 7 "Host-side vectors").
//
// The reference's driver does its vector algebra with NumPy on full-size host vectors: X + alpha*d, deepcopy, coeff*X, -1.*g + beta*t
// (Sphere_Grad_Descent.py:284, 296-298, 625-690, 755-756, 813) — 170 MB per vector at 128^3, 1.36 GB at 256^3, and every
// f / Grad_f / Inner_Product call then copies them over PCIe.  These entry points let the Python driver keep the vectors in HBM:
// a per-device buffer pool (the optimiser creates and drops temporaries at every line of its loop; hipMalloc / hipFree would
// serialise the device each time) and ONE arithmetic kernel,  out = a*x + b*y,  whose two products and one sum are rounded
// separately exactly as NumPy evaluates  a*x + b*y  (no fused multiply-add), so that the iterate sequence of an optimisation
// run on device vectors is bit-identical to the one on NumPy vectors.
#include <cstdint>
#include <map>
#include <mutex>

#include "smo_common.hpp"

namespace smo {
namespace {

struct DevicePool {
    hipStream_t stream = nullptr;
    std::multimap<size_t, void*> free_list;      // bytes -> buffer
    std::map<void*, size_t> live;                // buffer -> bytes
};
std::mutex g_mu;
std::map<int, DevicePool> g_pools;

int pool_for(int device, DevicePool** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no usable HIP device; libsmo has no CPU fallback"); return SMO_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (have %d)", device, ndev); return SMO_ERR_ARG; }
    SMO_HIP(hipSetDevice(device));
    DevicePool& p = g_pools[device];
    if (!p.stream) SMO_HIP(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
    *out = &p;
    return SMO_OK;
}

// out = fl(fl(a*x) + fl(b*y)); HAS_Y = false: out = fl(a*x).  The products and the sum must NOT be contracted into an fma: HIP's __dmul_rn /
// __dadd_rn are plain * and + to the compiler, which (with hipcc's default -ffp-contract=fast) fused fl(a*x) + b*y into v_fmac_f64 — one
// rounding less than NumPy whenever neither factor is +-1 (found in round 3 by calling smo_vec_axpby directly; the optimiser's operators
// always have a factor +-1 or no second operand and were never affected).  `#pragma clang fp contract(off)` pins the two-rounding form; the
// ISA is checked in tests/test_no_device_calls.py.
template <bool HAS_Y>
__global__ __launch_bounds__(256) void vec_axpby(size_t n2, size_t n, double a, const double* x, double b, const double* y, double* out) {
#pragma clang fp contract(off)
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const double2* y2 = reinterpret_cast<const double2*>(y);
    double2* o2 = reinterpret_cast<double2*>(out);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        double2 u = x2[i], r;
        r.x = a * u.x; r.y = a * u.y;                       // (plain operators: the header's __dmul_rn carries its own `contract` flag when inlined)
        if (HAS_Y) {
            double2 v = y2[i];
            const double px = b * v.x, py = b * v.y;
            r.x = r.x + px; r.y = r.y + py;
        }
        o2[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
        double r = a * x[n - 1];
        if (HAS_Y) { const double q = b * y[n - 1]; r = r + q; }
        out[n - 1] = r;
    }
}

// the same arithmetic one element per lane: vectors that are only 8-byte aligned (a view at an odd element offset of a caller's buffer)
template <bool HAS_Y>
__global__ __launch_bounds__(256) void vec_axpby_scalar(size_t n, double a, const double* x, double b, const double* y, double* out) {
#pragma clang fp contract(off)
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double r = a * x[i];
        if (HAS_Y) { const double q = b * y[i]; r = r + q; }
        out[i] = r;
    }
}

}  // namespace
}  // namespace smo

using namespace smo;

extern "C" {

int smo_vec_alloc(int device, size_t n, double** out) {
    if (!out || n == 0) { set_error("smo_vec_alloc: bad argument"); return SMO_ERR_ARG; }
    *out = nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    DevicePool* p = nullptr;
    SMO_TRY(pool_for(device, &p));
    const size_t bytes = ((n * sizeof(double) + 255) / 256) * 256;
    auto it = p->free_list.find(bytes);
    void* buf = nullptr;
    if (it != p->free_list.end()) { buf = it->second; p->free_list.erase(it); }
    else {
        hipError_t e = hipMalloc(&buf, bytes);
        if (e != hipSuccess) {          // give the pooled buffers back and try once more
            for (auto& kv : p->free_list) (void)hipFree(kv.second);
            p->free_list.clear();
            e = hipMalloc(&buf, bytes);
        }
        if (e != hipSuccess) { set_error("smo_vec_alloc: hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return e == hipErrorOutOfMemory ? SMO_ERR_NOMEM : SMO_ERR_HIP; }
    }
    p->live[buf] = bytes;
    *out = static_cast<double*>(buf);
    return SMO_OK;
}

int smo_vec_free(int device, double* v) {
    if (!v) return SMO_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    auto pit = g_pools.find(device);
    if (pit == g_pools.end()) { set_error("smo_vec_free: device %d has no vector pool", device); return SMO_ERR_ARG; }
    auto it = pit->second.live.find(v);
    if (it == pit->second.live.end()) { set_error("smo_vec_free: %p was not allocated by smo_vec_alloc on device %d", (void*)v, device); return SMO_ERR_ARG; }
    pit->second.free_list.emplace(it->second, it->first);
    pit->second.live.erase(it);
    return SMO_OK;
}

int smo_vec_pool_release(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto pit = g_pools.find(device);
    if (pit == g_pools.end()) return SMO_OK;
    SMO_HIP(hipSetDevice(device));
    for (auto& kv : pit->second.free_list) (void)hipFree(kv.second);
    pit->second.free_list.clear();
    return SMO_OK;
}

int smo_vec_pool_bytes(int device, size_t* live, size_t* pooled) {
    std::lock_guard<std::mutex> lk(g_mu);
    size_t l = 0, f = 0;
    auto pit = g_pools.find(device);
    if (pit != g_pools.end()) {
        for (auto& kv : pit->second.live) l += kv.second;
        for (auto& kv : pit->second.free_list) f += kv.first;
    }
    if (live) *live = l;
    if (pooled) *pooled = f;
    return SMO_OK;
}

int smo_vec_upload(int device, double* dev, const double* host, size_t n) {
    if (!dev || !host) { set_error("smo_vec_upload: null argument"); return SMO_ERR_ARG; }
    DevicePool* p = nullptr;
    { std::lock_guard<std::mutex> lk(g_mu); SMO_TRY(pool_for(device, &p)); }
    SMO_HIP(hipMemcpyAsync(dev, host, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
    SMO_HIP(hipStreamSynchronize(p->stream));
    return SMO_OK;
}

int smo_vec_download(int device, const double* dev, double* host, size_t n) {
    if (!dev || !host) { set_error("smo_vec_download: null argument"); return SMO_ERR_ARG; }
    DevicePool* p = nullptr;
    { std::lock_guard<std::mutex> lk(g_mu); SMO_TRY(pool_for(device, &p)); }
    SMO_HIP(hipMemcpyAsync(host, dev, n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    SMO_HIP(hipStreamSynchronize(p->stream));
    return SMO_OK;
}

int smo_vec_axpby(int device, size_t n, double a, const double* x, double b, const double* y, double* out) {
    if (!x || !out || n == 0) { set_error("smo_vec_axpby: bad argument"); return SMO_ERR_ARG; }
    DevicePool* p = nullptr;
    { std::lock_guard<std::mutex> lk(g_mu); SMO_TRY(pool_for(device, &p)); }
    // the kernel reads / writes 16 bytes per lane: pool buffers are 256-byte aligned, a view into a caller's own buffer at an odd element
    // offset is not — such vectors take the one-element-per-lane kernel (same rounding)
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | (y ? reinterpret_cast<uintptr_t>(y) : 0)) & 15u) == 0;
    if (!aligned) {
        if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | (y ? reinterpret_cast<uintptr_t>(y) : 0)) & 7u) {
            set_error("smo_vec_axpby: vectors must be 8-byte aligned (x=%p y=%p out=%p)", (const void*)x, (const void*)y, (void*)out);
            return SMO_ERR_ARG;
        }
        const unsigned nwg1 = (unsigned)std::min<size_t>(4096, (n + 255) / 256);
        if (y) hipLaunchKernelGGL((vec_axpby_scalar<true>), dim3(nwg1), dim3(256), 0, p->stream, n, a, x, b, y, out);
        else hipLaunchKernelGGL((vec_axpby_scalar<false>), dim3(nwg1), dim3(256), 0, p->stream, n, a, x, b, y, out);
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipStreamSynchronize(p->stream));
        return SMO_OK;
    }
    const size_t n2 = n / 2;
    const unsigned nwg = (unsigned)std::min<size_t>(4096, (n2 + 255) / 256 + 1);
    if (y) hipLaunchKernelGGL((vec_axpby<true>), dim3(nwg), dim3(256), 0, p->stream, n2, n, a, x, b, y, out);
    else hipLaunchKernelGGL((vec_axpby<false>), dim3(nwg), dim3(256), 0, p->stream, n2, n, a, x, b, y, out);
    SMO_HIP(hipGetLastError());
    SMO_HIP(hipStreamSynchronize(p->stream));
    return SMO_OK;
}

int smo_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) { set_error("smo_host_alloc: bad argument"); return SMO_ERR_ARG; }
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { set_error("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return e == hipErrorOutOfMemory ? SMO_ERR_NOMEM : SMO_ERR_HIP; }
    return SMO_OK;
}

int smo_host_free(void* p) {
    if (!p) return SMO_OK;
    SMO_HIP(hipHostFree(p));
    return SMO_OK;
}

}  // extern "C"

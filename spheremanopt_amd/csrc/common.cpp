// Host plumbing of libsmo (no kernels here).
#include <cstdlib>

#include "smo_common.hpp"

namespace smo {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

int DevPool::alloc(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? SMO_ERR_NOMEM : SMO_ERR_HIP;
    }
    ptrs.push_back(*p);
    total += bytes;
    return SMO_OK;
}
int DevPool::free_one(void* p) {
    for (size_t i = 0; i < ptrs.size(); ++i)
        if (ptrs[i] == p) {
            SMO_HIP(hipFree(p));
            ptrs.erase(ptrs.begin() + i);
            return SMO_OK;
        }
    set_error("DevPool::free_one: unknown buffer");
    return SMO_ERR_ARG;
}
void DevPool::release() {
    for (void* p : ptrs) (void)hipFree(p);
    ptrs.clear();
    total = 0;
}

std::vector<cplx> twiddles(int L) {
    std::vector<cplx> w(L);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < L; ++k) {
        long double a = -two_pi * (long double)k / (long double)L;
        w[k] = mk((double)cosl(a), (double)sinl(a));
    }
    return w;
}

// ---- timing ----------------------------------------------------------------------------------------------
hipEvent_t Timing::get() {
    if (!free_ev.empty()) { hipEvent_t e = free_ev.back(); free_ev.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void Timing::reset() {
    (void)flush();
    for (auto& c : cls) { c.launches = 0; c.total_ms = 0; }
    for (auto& n : seen) n = 0;
}
void Timing::begin(int k, hipStream_t s) {
    Pending p{k, get(), get()};
    (void)hipEventRecord(p.a, s);
    pend.push_back(p);
}
bool Timing::stamping() const {
    static const bool env_on = [] { const char* e = getenv("SMO_TIMING_STAMP"); return !(e && atoi(e) == 0); }();
    return stamp && env_on;
}
void Timing::begin_stamped(int k, hipEvent_t* a, hipEvent_t* b) {
    Pending p{k, get(), get()};
    *a = p.a; *b = p.b;
    pend.push_back(p);
}
void Timing::end(int k, hipStream_t s) {
    (void)k;
    (void)hipEventRecord(pend.back().b, s);
}
int Timing::flush() {
    for (auto& p : pend) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            cls[p.k].launches += 1;
            cls[p.k].total_ms += ms;
        }
        free_ev.push_back(p.a);
        free_ev.push_back(p.b);
    }
    pend.clear();
    return SMO_OK;
}
Timing::~Timing() {
    (void)flush();
    for (hipEvent_t e : free_ev) (void)hipEventDestroy(e);
}

// ---- context base ------------------------------------------------------------------------------------------
Context::~Context() {
    pool.release();
    if (stream && own_stream) (void)hipStreamDestroy(stream);
}

int Context::set_stream(hipStream_t s) {
    SMO_HIP(hipStreamSynchronize(stream));
    timing.flush();
    if (stream && own_stream) (void)hipStreamDestroy(stream);
    stream = s;
    own_stream = false;
    return SMO_OK;
}

int Context::base_init() {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no usable HIP device (%s); libsmo has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return SMO_ERR_NO_DEVICE;
    }
    if (cfg.device < 0 || cfg.device >= ndev) {
        set_error("device %d out of range (have %d)", cfg.device, ndev);
        return SMO_ERR_ARG;
    }
    SMO_HIP(hipSetDevice(cfg.device));
    SMO_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    return SMO_OK;
}

int Context::forward_host(const double* const* X, double* J) {
    SMO_HIP(hipSetDevice(cfg.device));
    const size_t bytes = vec_len * (size_t)cfg.batch * sizeof(double);
    for (int c = 0; c < n_comp; ++c) {
        if (!stage_x[c]) SMO_TRY(pool.alloc(&stage_x[c], vec_len * (size_t)cfg.batch));
        SMO_HIP(hipMemcpyAsync(stage_x[c], X[c], bytes, hipMemcpyHostToDevice, stream));
    }
    return forward_dev(stage_x, J);
}

int Context::adjoint_host(const double* const* X, int adjoint_type, double* const* grad) {
    SMO_HIP(hipSetDevice(cfg.device));
    const size_t bytes = vec_len * (size_t)cfg.batch * sizeof(double);
    for (int c = 0; c < n_comp; ++c) {
        if (!stage_x[c]) SMO_TRY(pool.alloc(&stage_x[c], vec_len * (size_t)cfg.batch));
        if (!stage_g[c]) SMO_TRY(pool.alloc(&stage_g[c], vec_len * (size_t)cfg.batch));
        if (X && X[c]) SMO_HIP(hipMemcpyAsync(stage_x[c], X[c], bytes, hipMemcpyHostToDevice, stream));
    }
    SMO_TRY(adjoint_dev(stage_x, adjoint_type, stage_g));
    for (int c = 0; c < n_comp; ++c) SMO_HIP(hipMemcpyAsync(grad[c], stage_g[c], bytes, hipMemcpyDeviceToHost, stream));
    SMO_HIP(hipStreamSynchronize(stream));
    return SMO_OK;
}

int Context::inner_host(const double* x, const double* y, double* out) {
    SMO_HIP(hipSetDevice(cfg.device));
    const size_t n = vec_len * (size_t)cfg.batch;
    for (int c = 0; c < 2; ++c)
        if (!stage_g[c]) SMO_TRY(pool.alloc(&stage_g[c], n));
    SMO_HIP(hipMemcpyAsync(stage_g[0], x, n * sizeof(double), hipMemcpyHostToDevice, stream));
    SMO_HIP(hipMemcpyAsync(stage_g[1], y, n * sizeof(double), hipMemcpyHostToDevice, stream));
    return inner_dev(stage_g[0], stage_g[1], out);
}

}  // namespace smo

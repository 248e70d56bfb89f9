// Shared host-side plumbing of libsmo: error reporting, device buffers, the context base class and
// HIP-event kernel timing.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smo.h"

namespace smo {

// ---------------------------------------------------------------------------------------------------------
// complex128 as a plain 16-byte pair (one ds_read_b128 / global_load_dwordx4 per element)
// ---------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) cplx {
    double re, im;
};
__host__ __device__ inline cplx mk(double r, double i) { cplx c; c.re = r; c.im = i; return c; }
__host__ __device__ inline cplx operator+(cplx a, cplx b) { return mk(a.re + b.re, a.im + b.im); }
__host__ __device__ inline cplx operator-(cplx a, cplx b) { return mk(a.re - b.re, a.im - b.im); }
__host__ __device__ inline cplx operator*(cplx a, cplx b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__host__ __device__ inline cplx operator*(double s, cplx a) { return mk(s * a.re, s * a.im); }
__host__ __device__ inline cplx conj(cplx a) { return mk(a.re, -a.im); }
__host__ __device__ inline cplx mul_i(cplx a) { return mk(-a.im, a.re); }        //  i * a
__host__ __device__ inline cplx mul_mi(cplx a) { return mk(a.im, -a.re); }       // -i * a
__host__ __device__ inline cplx mul_conj(cplx a, cplx w) {                       //  a * conj(w)
    return mk(a.re * w.re + a.im * w.im, a.im * w.re - a.re * w.im);
}

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
const char* last_error();

#define SMO_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            smo::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SMO_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

#define SMO_TRY(expr)                 \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != SMO_OK) return rc_; \
    } while (0)

// device allocation that records itself for release in the context destructor
struct DevPool {
    std::vector<void*> ptrs;
    size_t total = 0;
    int alloc(void** p, size_t bytes);
    template <class T> int alloc(T** p, size_t n) { return alloc(reinterpret_cast<void**>(p), n * sizeof(T)); }
    template <class T> int upload(T** p, const std::vector<T>& h, hipStream_t s) {
        SMO_TRY(alloc(p, h.size()));
        SMO_HIP(hipMemcpyAsync(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
        SMO_HIP(hipStreamSynchronize(s));
        return SMO_OK;
    }
    int free_one(void* p);              // give one buffer back before the context goes away
    void release();
};

// exp(-2 pi i k / L), k = 0..L-1, evaluated in long double
std::vector<cplx> twiddles(int L);

// ---------------------------------------------------------------------------------------------------------
// kernel timing: one accumulator per kernel class, HIP events recorded on the context's stream
// ---------------------------------------------------------------------------------------------------------
struct TimingClass {
    std::string name;
    double bytes_per_launch = 0;     // algorithmic bytes of ONE launch (every axis pass of every field reads + writes HBM)
    long long launches = 0;
    double total_ms = 0;
    double hbm_bytes = 0;            // compulsory HBM bytes of ONE launch of the kernel AS FUSED: each input read once,
                                     // each output written once; <= bytes_per_launch, and what the roofline fraction is taken against
};

class Timing {
public:
    bool on = false;
    unsigned long long mask = ~0ull;  // classes whose launches are timed (bit k = class k): keeps the event overhead out of the other launches
    int stride = 1;                   // of a selected class, every stride-th launch is timed (smo_timing_stride): a uniform sample at 1/stride of the event cost
    // How a timed launch gets its interval.  stamp = true (default; SMO_TIMING_STAMP=0 disables): the launch goes through
    // hipExtLaunchKernelGGL(..., startEvent, stopEvent), whose events carry the dispatch's OWN begin / end timestamps — the quantity
    // rocprofv3 --kernel-trace reports — and put no marker packet into the queue.  stamp = false (and every class that is not a single
    // kernel launch, e.g. the exchanges): hipEventRecord before and after, i.e. two marker packets; a start marker recorded straight behind
    // an un-timed kernel is processed while that kernel's last workgroups still run, so such an interval includes the predecessor's tail
    // (round 3: the every-8th-launch sample read 2.4-4.6 % above the every-launch average).
    bool stamp = true;
    std::vector<long long> seen;      // launches of each class since the last reset, timed or not
    std::vector<TimingClass> cls;
    int add_class(const char* name, double bytes, double hbm = -1.0) {
        cls.push_back({name, bytes, 0, 0.0, hbm < 0 ? bytes : hbm});
        seen.push_back(0);
        return (int)cls.size() - 1;
    }
    void reset();
    void begin(int k, hipStream_t s);
    void end(int k, hipStream_t s);
    bool stamping() const;            // stamp && !SMO_TIMING_STAMP=0
    void begin_stamped(int k, hipEvent_t* a, hipEvent_t* b);      // events for hipExtLaunchKernelGGL; nothing is recorded here
    int flush();                      // resolve pending event pairs (after a stream sync)
    ~Timing();
private:
    struct Pending { int k; hipEvent_t a, b; };
    std::vector<Pending> pend;
    std::vector<hipEvent_t> free_ev;
    hipEvent_t get();
};

struct ScopedTimer {
    Timing& t; int k; hipStream_t s;
    bool active;
    bool stamped = false;             // the launch itself carries the events (SMO_LAUNCH_T): no markers
    hipEvent_t ea = nullptr, eb = nullptr;
    ScopedTimer(Timing& t_, int k_, hipStream_t s_, bool single_kernel = false)
        : t(t_), k(k_), s(s_), active(t_.on && k_ >= 0 && k_ < 64 && ((t_.mask >> k_) & 1ull)) {
        if (active && t.stride > 1) active = (t.seen[k]++ % t.stride) == 0;
        if (active && single_kernel && t.stamping()) { stamped = true; t.begin_stamped(k, &ea, &eb); }
        else if (active) t.begin(k, s);
    }
    ~ScopedTimer() { if (active && !stamped) t.end(k, s); }
};
// launch `kernel` under ScopedTimer `tm` (constructed with single_kernel = true): with the dispatch's own timestamps when it is a timed launch
#define SMO_LAUNCH_T(tm, kernel, grid, block, shmem, stream, ...)                                                                   \
    do {                                                                                                                            \
        if ((tm).stamped) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, (tm).ea, (tm).eb, 0, __VA_ARGS__);               \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                                  \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// context base
// ---------------------------------------------------------------------------------------------------------
class Context {
public:
    smo_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = true;
    DevPool pool;
    Timing timing;
    bool have_forward = false;
    int n_comp = 1;
    size_t vec_len = 0;          // doubles per component per batch member
    size_t stack_bytes = 0;
    size_t snapshot_doubles = 0;

    virtual ~Context();
    virtual int init() = 0;
    virtual int forward_dev(const double* const* X, double* J_host) = 0;
    virtual int adjoint_dev(const double* const* X, int adjoint_type, double* const* grad) = 0;
    virtual int inner_dev(const double* x, const double* y, double* out_host) = 0;
    // <x,y> of vectors given slab by slab (one pointer per device of a multi-device context; a plain context has one slab: the vector)
    virtual int inner_slabs(const double* const* x, const double* const* y, double* out_host) { return inner_dev(x[0], y[0], out_host); }
    virtual int snapshot_read(int b, int index, double* out) = 0;
    virtual int transform_host(int which, const double* in, double* out) {
        (void)which; (void)in; (void)out;
        set_error("smo_transform: not available for this problem kind");
        return SMO_ERR_UNSUPPORTED;
    }

    virtual double info(int key) const { return key == 0 ? 1.0 : 0.0; }
    virtual int kdyn_op(int op, int i0, int i1, void* p0, void* p1, double* out) {
        (void)op; (void)i0; (void)i1; (void)p0; (void)p1; (void)out;
        set_error("smo_kdyn_op: not a KDYN context");
        return SMO_ERR_UNSUPPORTED;
    }
    // slab communicator (KDYN with world > 1): see include/smo.h "in-library time loop"
    virtual int comm_init(const void* id128) { (void)id128; set_error("smo_comm_init: not a KDYN context"); return SMO_ERR_UNSUPPORTED; }
    virtual int comm_set_transport(smo_alltoall_fn a2a, smo_allreduce_fn ared, void* user) {
        (void)a2a; (void)ared; (void)user;
        set_error("smo_comm_set_transport: not a KDYN context");
        return SMO_ERR_UNSUPPORTED;
    }
    virtual double comm_info(int key) const { (void)key; return 0.0; }
    // ranks of ONE process (smo_create_multi): the context becomes rank `rank` of the group; collective like comm_init
    virtual int comm_set_peers(class PeerGroup* g, int rank) { (void)g; (void)rank; set_error("multi-device contexts: KDYN only"); return SMO_ERR_UNSUPPORTED; }
    virtual int set_stream(hipStream_t s);      // run on a caller-owned stream (e.g. torch's current stream) instead of the private one
    // device pointers the _dev entry points dereference (a multi-device context: one slab per component and device), slabs of smo_inner_slabs
    virtual int n_dev_ptrs() const { return n_comp; }
    virtual int n_slabs() const { return 1; }

    // host-buffer variants: stage through context-owned device vectors (a multi-device context scatters / gathers slabs instead)
    virtual int forward_host(const double* const* X, double* J);
    virtual int adjoint_host(const double* const* X, int adjoint_type, double* const* grad);
    virtual int inner_host(const double* x, const double* y, double* out);
    // the kernel timing behind smo_timing_* (a multi-device context answers with rank 0's) and a wait for everything enqueued
    virtual Timing& tm() { return timing; }
    virtual int sync_all() { SMO_HIP(hipSetDevice(cfg.device)); SMO_HIP(hipStreamSynchronize(stream)); return SMO_OK; }

protected:
    int base_init();             // device selection, stream, staging buffers (needs n_comp / vec_len set)
    double* stage_x[2] = {nullptr, nullptr};
    double* stage_g[2] = {nullptr, nullptr};
};

Context* make_sh23(const smo_config& cfg);
Context* make_shb23(const smo_config& cfg);
Context* make_kdyn(const smo_config& cfg);
Context* make_pois(const smo_config& cfg);
// one context that slab-decomposes a KDYN problem over several GPUs of this process (csrc/multi.cpp)
Context* make_multi(const smo_config& cfg, int ndev, const int* dev_ids);

}  // namespace smo

// SH23 — 1-D periodic Swift-Hohenberg 2-3, forward (SBDF1) and discrete/continuous adjoint sweeps.
//
// Replaces FWD_Solve_IVP_Lin / Compatib_Cond / ADJ_Solve_IVP_Lin / Inner_Prod of
// Example_Problems/Periodic_Domain(Fourier)/Swift_Hohenberg/FWD_Solve_SH23.py (:409-545, :552-596, :598-729,
// :158-172); recurrences: SURVEY.md Appendix A.1.
//
// One problem = one workgroup that runs the WHOLE time loop (the config is latency bound: 2 MB of compulsory
// traffic per gradient, ~1000 dependent steps).  State lives in registers/LDS; per step only the 2 KB
// coefficient snapshot goes to (forward) or comes from (adjoint) the HBM-resident stack.
//   real FFT of length G = 2*Npts via one complex Stockham FFT of length NH = G/2 (fft_lds.hpp) plus an
//   even/odd split that is fused with the per-mode implicit solve and the packing for the next inverse FFT.
// Independent problems (cfg.batch) map to independent workgroups.
#include <algorithm>

#include "fft_lds.hpp"

namespace smo {

namespace {

constexpr int FWD_THREADS = 64;      // one wavefront: barriers are wave-local
constexpr int ADJ_THREADS = 128;     // two wavefronts: the two inverse transforms (u_f, q) run side by side

// Pack Hermitian coefficient X_k (k < NC, zero beyond) for the half-length inverse transform:
//   z[n] = x[2n] + i x[2n+1] = IFFT_NH(Z),  Z_k = X_k + i w^k X_k,  Z_{NH-k} = conj(X_k - i w^k X_k),  w = e^{+2 pi i / G}
template <int NH>
__device__ __forceinline__ void pack_mode(cplx* P, int k, cplx X, cplx tw2k /* e^{-2 pi i k/G} */) {
    if (k == 0) {
        P[0] = mk(X.re, X.re);                    // c2r ignores Im X_0
    } else {
        cplx Y = mul_conj(X, tw2k);               // X * w^k
        P[k] = X + mul_i(Y);
        P[NH - k] = conj(X - mul_i(Y));
    }
}

// Coefficient k of the real length-G forward transform from the half-length transform Z of z[n] = x[2n] + i x[2n+1]
template <int NH>
__device__ __forceinline__ cplx unpack_mode(const cplx* P, int k, cplx tw2k) {
    cplx Zk = P[k];
    cplx Zm = conj(P[k == 0 ? 0 : NH - k]);
    cplx Ev = 0.5 * (Zk + Zm);
    cplx Od = mul_mi(0.5 * (Zk - Zm));
    return Ev + Od * tw2k;
}

template <int NH>
__global__ __launch_bounds__(FWD_THREADS) void sh23_forward_kernel(const double* __restrict__ X, cplx* __restrict__ stack,
                                                                   double* __restrict__ Jout, const double* __restrict__ A_g,
                                                                   const cplx* __restrict__ tw_g, const cplx* __restrict__ tw2_g,
                                                                   double dt, int n_iters) {
    constexpr int NT = FWD_THREADS, NC = NH / 2, G = 2 * NH;
    constexpr int KPT = (NC + NT - 1) / NT;
    __shared__ cplx P[NH], bufA[NH], bufB[NH], tw[NH];
    const int tid = threadIdx.x;
    const size_t prob = blockIdx.x;
    X += prob * G;
    stack += prob * (size_t)(n_iters + 1) * NC;

    for (int i = tid; i < NH; i += NT) {
        tw[i] = tw_g[i];
        P[i] = mk(X[2 * i], X[2 * i + 1]);
    }
    cplx uh[KPT], w2[KPT];
    double A[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int k = tid + i * NT;
        w2[i] = (k < NC) ? tw2_g[k] : mk(1, 0);
        A[i] = (k < NC) ? A_g[k] : 1.0;
        uh[i] = mk(0, 0);
    }
    __syncthreads();

    auto ldP = [&](int, int pos) { return P[pos]; };
    auto stP = [&](int, int pos, cplx v) { P[pos] = v; };
    const double inv_dt = 1.0 / dt, inv_G = 1.0 / G;
    double acc = 0.0;

    for (int n = -1; n <= n_iters; ++n) {
        if (n >= 0) {
            // u_n = F^-1 u^_n ; J += sum u_n^2 ; N = 1.8 u^2 - u^3, packed for the forward transform
            fft_batch<NH, true>(bufA, bufB, tw, 1, NH, tid, NT, ldP, [&](int, int pos, cplx v) {
                acc += v.re * v.re + v.im * v.im;
                P[pos] = mk(v.re * v.re * (1.8 - v.re), v.im * v.im * (1.8 - v.im));
            });
            if (n == n_iters) break;                  // the reference's (N+1)-th update is never used
            __syncthreads();
        }
        fft_batch<NH, false>(bufA, bufB, tw, 1, NH, tid, NT, ldP, stP);
        __syncthreads();
        // even/odd split + implicit solve + snapshot + packing of the next state (same thread owns k and NH-k)
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int k = tid + i * NT;
            if (k < NC) {
                cplx Nk = inv_G * unpack_mode<NH>(P, k, w2[i]);
                if (n < 0) uh[i] = Nk;                // u^_0 = F X
                else uh[i] = mk((uh[i].re * inv_dt + Nk.re) / A[i], (uh[i].im * inv_dt + Nk.im) / A[i]);
                stack[(size_t)(n + 1) * NC + k] = uh[i];
                pack_mode<NH>(P, k, uh[i], w2[i]);
            }
        }
        if (tid == 0) P[NC] = mk(0, 0);
        __syncthreads();
    }
    // J = dt * sum_n mean(u_n^2)
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (tid == 0) Jout[prob] = -(dt * inv_G) * acc;
}

template <int NH>
__global__ __launch_bounds__(ADJ_THREADS) void sh23_adjoint_kernel(const cplx* __restrict__ stack, double* __restrict__ grad,
                                                                   const double* __restrict__ A_g, const cplx* __restrict__ tw_g,
                                                                   const cplx* __restrict__ tw2_g, double dt, int n_iters,
                                                                   int continuous) {
    constexpr int NT = ADJ_THREADS, NC = NH / 2, G = 2 * NH;
    constexpr int KPT = (NC + NT - 1) / NT;
    __shared__ cplx P[2 * NH], bufA[2 * NH], bufB[2 * NH], tw[NH];
    const int tid = threadIdx.x;
    const size_t prob = blockIdx.x;
    stack += prob * (size_t)(n_iters + 1) * NC;
    grad += prob * G;

    for (int i = tid; i < NH; i += NT) tw[i] = tw_g[i];
    cplx qh[KPT], w2[KPT], uf[KPT];
    double A[KPT];
    int idx = continuous ? n_iters : n_iters - 1;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int k = tid + i * NT;
        const bool ok = k < NC;
        w2[i] = ok ? tw2_g[k] : mk(1, 0);
        A[i] = ok ? A_g[k] : 1.0;
        qh[i] = mk(0, 0);
        if (ok && !continuous) {                      // compatibility condition  q^ = -2 u^_N / (1/dt + L_k)
            cplx uN = stack[(size_t)n_iters * NC + k];
            qh[i] = mk(-2.0 * uN.re / A[i], -2.0 * uN.im / A[i]);
        }
        uf[i] = ok ? stack[(size_t)idx * NC + k] : mk(0, 0);
    }
    const double inv_dt = 1.0 / dt, inv_G = 1.0 / G;
    auto ldP = [&](int b, int pos) { return P[b * NH + pos]; };
    auto stP = [&](int b, int pos, cplx v) { P[b * NH + pos] = v; };

    for (int it = 0; it < n_iters; ++it) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int k = tid + i * NT;
            if (k < NC) {
                pack_mode<NH>(P, k, uf[i], w2[i]);
                pack_mode<NH>(P + NH, k, qh[i], w2[i]);
            }
        }
        if (tid == 0) { P[NC] = mk(0, 0); P[NH + NC] = mk(0, 0); }
        --idx;
        if (idx >= 0) {                               // prefetch the next snapshot under the transforms
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const int k = tid + i * NT;
                if (k < NC) uf[i] = stack[(size_t)idx * NC + k];
            }
        }
        __syncthreads();
        fft_batch<NH, true>(bufA, bufB, tw, 2, NH, tid, NT, ldP, stP);        // P[0] = u_f, P[1] = q on the grid
        __syncthreads();
        for (int n = tid; n < NH; n += NT) {
            cplx u = P[n], q = P[NH + n];
            P[n] = mk((3.6 * u.re - 3.0 * u.re * u.re) * q.re - 2.0 * u.re, (3.6 * u.im - 3.0 * u.im * u.im) * q.im - 2.0 * u.im);
        }
        __syncthreads();
        fft_batch<NH, false>(bufA, bufB, tw, 1, NH, tid, NT, ldP, stP);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int k = tid + i * NT;
            if (k < NC) {
                cplx Hk = inv_G * unpack_mode<NH>(P, k, w2[i]);
                qh[i] = mk((qh[i].re * inv_dt + Hk.re) / A[i], (qh[i].im * inv_dt + Hk.im) / A[i]);
            }
        }
        __syncthreads();
    }
    // gradient on the scale-2 grid: F^-1[dt (1/dt + L) q^]  (discrete)  |  F^-1[q^]  (continuous)
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int k = tid + i * NT;
        if (k < NC) {
            const double s = continuous ? 1.0 : dt * A[i];
            pack_mode<NH>(P, k, mk(s * qh[i].re, s * qh[i].im), w2[i]);
        }
    }
    if (tid == 0) P[NC] = mk(0, 0);
    __syncthreads();
    fft_batch<NH, true>(bufA, bufB, tw, 1, NH, tid, NT, ldP, [&](int, int pos, cplx v) {
        grad[2 * pos] = v.re;
        grad[2 * pos + 1] = v.im;
    });
}

// ---------------------------------------------------------------------------------------------------------
// Any Npts: the same two sweeps with the half length NH = Npts a RUN-TIME value (the reference builds its Fourier basis for whatever
// Npts it is handed, FWD_Solve_SH23.py:279-332, :752-757).  One workgroup of 256 threads per problem, the run-time-length Stockham
// chain of fft_lds.hpp (radices = prime factors of NH), the per-mode state in the LDS instead of registers.  The real length-2NH
// transform through one complex length-NH transform works for every NH, odd ones included.  A few times slower per step than
// the instantiated kernels above; used for the lengths that have no instantiation (SMO_SH_ANY=1 forces it).
// ---------------------------------------------------------------------------------------------------------
constexpr int ANY_THREADS = 1024;
// one output per thread and stage: measured at NH = 256 (bench.py --workload sh23, SMO_SH_ANY=1; SMO_SH_ANY_NT overrides) 64 / 128 / 256 / 512 /
// 1024 threads give 42 / 74 / 117 / 128 / 128 gradients per second (instantiated kernels: 500) — as many threads as the adjoint's two
// transforms have points, up to the workgroup limit
inline int any_threads(int NH) {
    if (const char* e = getenv("SMO_SH_ANY_NT")) { const int v = atoi(e); if (v >= 64 && v <= ANY_THREADS && v % 64 == 0) return v; }
    return std::min(ANY_THREADS, std::max(64, (2 * NH + 63) / 64 * 64));
}

__device__ __forceinline__ void any_pack(cplx* P, int NH, int k, cplx X, cplx tw2k) {
    if (k == 0) { P[0] = mk(X.re, X.re); return; }
    const cplx Y = mul_conj(X, tw2k);
    P[k] = X + mul_i(Y);
    P[NH - k] = conj(X - mul_i(Y));
}
__device__ __forceinline__ cplx any_unpack(const cplx* P, int NH, int k, cplx tw2k) {
    const cplx Zk = P[k], Zm = conj(P[k == 0 ? 0 : NH - k]);
    return 0.5 * (Zk + Zm) + mul_mi(0.5 * (Zk - Zm)) * tw2k;
}
// positions of the packed half-length spectrum that no retained mode feeds (k and NH - k for k < NC): NC .. NH - NC
__device__ __forceinline__ void any_zero_gap(cplx* P, int NH, int NC, int tid, int NT) {
    for (int i = NC + tid; i <= NH - NC; i += NT) P[i] = mk(0, 0);
}

__global__ __launch_bounds__(ANY_THREADS) void sh23_forward_any(const double* __restrict__ X, cplx* __restrict__ stack, double* __restrict__ Jout,
                                                                const double* __restrict__ A_g, const cplx* __restrict__ tw_g,
                                                                const cplx* __restrict__ tw2_g, double dt, int n_iters, AnyPlan pl, int NC) {
    extern __shared__ cplx sh_lds[];
    const int NT = blockDim.x;                                                   // 64 .. ANY_THREADS, a multiple of 64 (any_threads below)
    const int NH = pl.L, G = 2 * NH, tid = threadIdx.x;
    cplx *P = sh_lds, *T = sh_lds + NH, *tws = sh_lds + 2 * NH, *uh = sh_lds + 3 * NH;
    any_load_tw(tws, tw_g, NH, tid, NT);
    __shared__ double red[ANY_THREADS / 64];
    const size_t prob = blockIdx.x;
    X += prob * G;
    stack += prob * (size_t)(n_iters + 1) * NC;
    for (int i = tid; i < NH; i += NT) P[i] = mk(X[2 * i], X[2 * i + 1]);
    for (int k = tid; k < NC; k += NT) uh[k] = mk(0, 0);
    __syncthreads();
    const double inv_dt = 1.0 / dt, inv_G = 1.0 / G;
    double acc = 0.0;
    for (int n = -1; n <= n_iters; ++n) {
        if (n >= 0) {
            const cplx* R = any_fft<true>(P, T, tws, pl, 1, tid, NT);          // u_n on the grid (two points per element)
            for (int i = tid; i < NH; i += NT) {
                const cplx v = R[i];
                acc += v.re * v.re + v.im * v.im;
                P[i] = mk(v.re * v.re * (1.8 - v.re), v.im * v.im * (1.8 - v.im));
            }
            if (n == n_iters) break;
            __syncthreads();
        }
        const cplx* R = any_fft<false>(P, T, tws, pl, 1, tid, NT);
        for (int k = tid; k < NC; k += NT) {                                    // the same thread owns k and NH - k: in place is safe
            const cplx w = tw2_g[k];
            const cplx Nk = inv_G * any_unpack(R, NH, k, w);
            const cplx u = (n < 0) ? Nk : mk((uh[k].re * inv_dt + Nk.re) / A_g[k], (uh[k].im * inv_dt + Nk.im) / A_g[k]);
            uh[k] = u;
            stack[(size_t)(n + 1) * NC + k] = u;
            any_pack(P, NH, k, u, w);
        }
        any_zero_gap(P, NH, NC, tid, NT);
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) { double s = 0.0; for (int i = 0; i < NT / 64; ++i) s += red[i]; Jout[prob] = -(dt * inv_G) * s; }
}

__global__ __launch_bounds__(ANY_THREADS) void sh23_adjoint_any(const cplx* __restrict__ stack, double* __restrict__ grad, const double* __restrict__ A_g,
                                                                const cplx* __restrict__ tw_g, const cplx* __restrict__ tw2_g, double dt, int n_iters,
                                                                int continuous, AnyPlan pl, int NC) {
    extern __shared__ cplx sh_lds[];
    const int NT = blockDim.x;
    const int NH = pl.L, G = 2 * NH, tid = threadIdx.x;
    cplx *P = sh_lds, *T = sh_lds + 2 * NH, *tws = sh_lds + 4 * NH, *qh = sh_lds + 5 * NH;              // P / T: two transforms side by side (u_f, q)
    any_load_tw(tws, tw_g, NH, tid, NT);
    const size_t prob = blockIdx.x;
    stack += prob * (size_t)(n_iters + 1) * NC;
    grad += prob * G;
    int idx = continuous ? n_iters : n_iters - 1;
    for (int k = tid; k < NC; k += NT) {
        cplx q = mk(0, 0);
        if (!continuous) { const cplx uN = stack[(size_t)n_iters * NC + k]; q = mk(-2.0 * uN.re / A_g[k], -2.0 * uN.im / A_g[k]); }
        qh[k] = q;
    }
    const double inv_dt = 1.0 / dt, inv_G = 1.0 / G;
    for (int it = 0; it < n_iters; ++it, --idx) {
        for (int k = tid; k < NC; k += NT) {                                    // (qh[k] is read and written by the same thread only)
            const cplx w = tw2_g[k];
            any_pack(P, NH, k, stack[(size_t)idx * NC + k], w);
            any_pack(P + NH, NH, k, qh[k], w);
        }
        any_zero_gap(P, NH, NC, tid, NT);
        any_zero_gap(P + NH, NH, NC, tid, NT);
        __syncthreads();
        cplx* R = any_fft<true>(P, T, tws, pl, 2, tid, NT);                    // R[0..NH) = u_f, R[NH..2NH) = q on the grid
        cplx* F = (R == P) ? T : P;
        for (int i = tid; i < NH; i += NT) {
            const cplx u = R[i], q = R[NH + i];
            F[i] = mk((3.6 * u.re - 3.0 * u.re * u.re) * q.re - 2.0 * u.re, (3.6 * u.im - 3.0 * u.im * u.im) * q.im - 2.0 * u.im);
        }
        __syncthreads();
        const cplx* H = any_fft<false>(F, R, tws, pl, 1, tid, NT);
        for (int k = tid; k < NC; k += NT) {
            const cplx Hk = inv_G * any_unpack(H, NH, k, tw2_g[k]);
            qh[k] = mk((qh[k].re * inv_dt + Hk.re) / A_g[k], (qh[k].im * inv_dt + Hk.im) / A_g[k]);
        }
        __syncthreads();
    }
    // gradient on the scale-2 grid: F^-1[dt (1/dt + L) q^]  (discrete)  |  F^-1[q^]  (continuous)
    for (int k = tid; k < NC; k += NT) {
        const double s = continuous ? 1.0 : dt * A_g[k];
        any_pack(P, NH, k, mk(s * qh[k].re, s * qh[k].im), tw2_g[k]);
    }
    any_zero_gap(P, NH, NC, tid, NT);
    __syncthreads();
    const cplx* R = any_fft<true>(P, T, tws, pl, 1, tid, NT);
    for (int i = tid; i < NH; i += NT) { grad[2 * i] = R[i].re; grad[2 * i + 1] = R[i].im; }
}

// <x,y> = mean(x*y) over the G grid points, one workgroup per batch member
__global__ __launch_bounds__(256) void sh23_inner_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                         double* __restrict__ out, int G) {
    __shared__ double red[4];
    const size_t prob = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < G; i += 256) acc += x[prob * G + i] * y[prob * G + i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[prob] = (red[0] + red[1] + red[2] + red[3]) / G;
}

class SH23 : public Context {
public:
    explicit SH23(const smo_config& c) { cfg = c; }
    int NH = 0, NC = 0, G = 0;
    cplx* d_stack = nullptr;
    cplx* d_tw = nullptr;
    cplx* d_tw2 = nullptr;
    double* d_A = nullptr;
    double* d_out = nullptr;     // [batch] results (J / inner)
    int k_fwd = -1, k_adj = -1;
    bool any_len = false;        // no instantiation for this length: the run-time-length kernels
    AnyPlan plan{};
    size_t lds_fwd = 0, lds_adj = 0;

    int init() override {
        NH = cfg.npts;
        G = 2 * NH;
        NC = (cfg.npts - 1) / 2 + 1;
        // instantiated transform lengths: 2^k, 3*2^k, 5*2^k, 7*2^k, 15*2^k in [16, 1024]; every other Npts (the reference, through FFTW, takes
        // any) runs the any-length kernels
        if (NH < 4) { set_error("SH23: npts must be >= 4 (got %d)", NH); return SMO_ERR_UNSUPPORTED; }
        any_len = dispatch([](auto) { return SMO_OK; }) != SMO_OK;
        { const char* e = getenv("SMO_SH_ANY"); if (e && atoi(e) == 1) any_len = true; }
        if (any_len) {
            plan = any_plan(NH);
            lds_fwd = (size_t)(3 * NH + NC) * sizeof(cplx);
            lds_adj = (size_t)(5 * NH + NC) * sizeof(cplx);
        }
        n_comp = 1;
        vec_len = (size_t)G;
        snapshot_doubles = 2 * (size_t)NC;
        stack_bytes = (size_t)cfg.batch * (cfg.n_iters + 1) * NC * sizeof(cplx);
        SMO_TRY(base_init());
        if (any_len) {
            if (lds_adj > 65536) {                            // above the default per-workgroup limit: opt in
                int dev = 0, maxb = 0;
                SMO_HIP(hipGetDevice(&dev));
                SMO_HIP(hipDeviceGetAttribute(&maxb, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
                if (lds_adj + 256 > (size_t)maxb) { set_error("SH23: npts = %d needs %zu bytes of LDS per workgroup (device: %d)", NH, lds_adj, maxb); return SMO_ERR_UNSUPPORTED; }
                SMO_HIP(hipFuncSetAttribute((const void*)sh23_forward_any, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fwd));
                SMO_HIP(hipFuncSetAttribute((const void*)sh23_adjoint_any, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_adj));
            }
        }
        const double L = cfg.x1 - cfg.x0, a = cfg.param;
        std::vector<double> A(NC);
        for (int k = 0; k < NC; ++k) {
            const double kk = 2.0 * M_PI * k / L;
            const double lk = (1.0 - kk * kk) * (1.0 - kk * kk) - a;       // Lap(u) - a u
            A[k] = 1.0 / cfg.dt + lk;                                       // SBDF1: a_0 M + b_0 L
        }
        std::vector<cplx> tw2 = twiddles(G);
        tw2.resize(NC);
        SMO_TRY(pool.upload(&d_A, A, stream));
        SMO_TRY(pool.upload(&d_tw, twiddles(NH), stream));
        SMO_TRY(pool.upload(&d_tw2, tw2, stream));
        SMO_TRY(pool.alloc(&d_stack, (size_t)cfg.batch * (cfg.n_iters + 1) * NC));
        SMO_TRY(pool.alloc(&d_out, (size_t)cfg.batch));
        // algorithmic bytes: forward writes the stack + reads X; adjoint reads the stack + writes grad
        const double stack_b = (double)(cfg.n_iters + 1) * NC * 16.0, vec_b = G * 8.0;
        k_fwd = timing.add_class("sh23_forward_kernel", cfg.batch * (stack_b + vec_b));
        k_adj = timing.add_class("sh23_adjoint_kernel", cfg.batch * (stack_b + vec_b));
        return SMO_OK;
    }

    template <class F> int dispatch(F f) {
        switch (NH) {
#define SMO_SH_CASE(n) case n: return f(std::integral_constant<int, n>());
            SMO_SH_CASE(16) SMO_SH_CASE(32) SMO_SH_CASE(64) SMO_SH_CASE(128) SMO_SH_CASE(256) SMO_SH_CASE(512)
            SMO_SH_CASE(24) SMO_SH_CASE(48) SMO_SH_CASE(96) SMO_SH_CASE(192) SMO_SH_CASE(384) SMO_SH_CASE(768)          // 3 * 2^k
            SMO_SH_CASE(20) SMO_SH_CASE(40) SMO_SH_CASE(80) SMO_SH_CASE(160) SMO_SH_CASE(320) SMO_SH_CASE(640)          // 5 * 2^k
            SMO_SH_CASE(60) SMO_SH_CASE(120) SMO_SH_CASE(240) SMO_SH_CASE(480) SMO_SH_CASE(960)                         // 15 * 2^k
            SMO_SH_CASE(28) SMO_SH_CASE(56) SMO_SH_CASE(112) SMO_SH_CASE(224) SMO_SH_CASE(448) SMO_SH_CASE(896)         // 7 * 2^k
            SMO_SH_CASE(36) SMO_SH_CASE(72) SMO_SH_CASE(144) SMO_SH_CASE(288) SMO_SH_CASE(576)                          // 9 * 2^k
            SMO_SH_CASE(50) SMO_SH_CASE(100) SMO_SH_CASE(200) SMO_SH_CASE(400) SMO_SH_CASE(800)                         // 25 * 2^k: the round decimal sizes
            SMO_SH_CASE(150) SMO_SH_CASE(300) SMO_SH_CASE(600) SMO_SH_CASE(250) SMO_SH_CASE(500) SMO_SH_CASE(1000)      // 75 * 2^k, 125 * 2^k
#undef SMO_SH_CASE
            case 1024: return f(std::integral_constant<int, 1024>());
        }
        set_error("SH23: npts must be 2^k, 3*2^k, 5*2^k, 7*2^k or 15*2^k in [16, 1024] (got %d)", NH);
        return SMO_ERR_UNSUPPORTED;
    }

    int forward_dev(const double* const* X, double* J) override {
        have_forward = false;
        if (any_len) {
            ScopedTimer t(timing, k_fwd, stream);
            hipLaunchKernelGGL(sh23_forward_any, dim3(cfg.batch), dim3(any_threads(NH)), lds_fwd, stream, X[0], d_stack, d_out, (const double*)d_A, (const cplx*)d_tw,
                               (const cplx*)d_tw2, cfg.dt, cfg.n_iters, plan, NC);
        } else {
            SMO_TRY(dispatch([&](auto nh) {
                ScopedTimer t(timing, k_fwd, stream);
                hipLaunchKernelGGL((sh23_forward_kernel<decltype(nh)::value>), dim3(cfg.batch), dim3(FWD_THREADS), 0, stream, X[0],
                                   d_stack, d_out, d_A, d_tw, d_tw2, cfg.dt, cfg.n_iters);
                return SMO_OK;
            }));
        }
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipMemcpyAsync(J, d_out, cfg.batch * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        have_forward = true;
        return SMO_OK;
    }

    int adjoint_dev(const double* const*, int adjoint_type, double* const* grad) override {
        if (any_len) {
            ScopedTimer t(timing, k_adj, stream);
            hipLaunchKernelGGL(sh23_adjoint_any, dim3(cfg.batch), dim3(any_threads(NH)), lds_adj, stream, (const cplx*)d_stack, grad[0], (const double*)d_A,
                               (const cplx*)d_tw, (const cplx*)d_tw2, cfg.dt, cfg.n_iters, adjoint_type == SMO_ADJ_CONTINUOUS ? 1 : 0, plan, NC);
        } else {
            SMO_TRY(dispatch([&](auto nh) {
                ScopedTimer t(timing, k_adj, stream);
                hipLaunchKernelGGL((sh23_adjoint_kernel<decltype(nh)::value>), dim3(cfg.batch), dim3(ADJ_THREADS), 0, stream, d_stack,
                                   grad[0], d_A, d_tw, d_tw2, cfg.dt, cfg.n_iters, adjoint_type == SMO_ADJ_CONTINUOUS ? 1 : 0);
                return SMO_OK;
            }));
        }
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }

    int inner_dev(const double* x, const double* y, double* out) override {
        hipLaunchKernelGGL(sh23_inner_kernel, dim3(cfg.batch), dim3(256), 0, stream, x, y, d_out, G);
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipMemcpyAsync(out, d_out, cfg.batch * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }

    int snapshot_read(int b, int index, double* out) override {
        const cplx* src = d_stack + ((size_t)b * (cfg.n_iters + 1) + index) * NC;
        SMO_HIP(hipMemcpyAsync(out, src, NC * sizeof(cplx), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }
};

}  // namespace

Context* make_sh23(const smo_config& cfg) { return new SH23(cfg); }

}  // namespace smo

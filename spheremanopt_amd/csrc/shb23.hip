// SHB23 — 1-D Swift-Hohenberg on z in [z0,z1] with Chebyshev-tau boundary conditions, "Discrete" path:
// hand-stepped SBDF1 forward solve and its exact transpose (discrete adjoint).
//
// Replaces FWD_Solve_IVP_Discrete / ADJ_Solve_IVP_Discrete / Inner_Prod_Discrete and the transform helpers of
// Example_Problems/Bounded_Domain(Cheby)/Swift_Hohenberg_Bounded/FWD_Solve_SHB23.py (:525-678, :796-920, :189-193,
// :36-81); recurrences: SURVEY.md Appendix A.3.
//
// The reference solves, every step, a 4N x 4N Chebyshev-tau system through Dedalus' pencil LU.  Only the block
// "rhs of the first equation -> u" matters: an N x N operator S that depends on (N, dt, a, interval) alone.  It is built
// ONCE per context on the host (banded + 4 boundary rows, sparse-aware LU with partial pivoting) and kept in HBM/L2
// (2 MB at N = 512); a step is then   c <- S (Z T[2g^2 - g^3] + c/dt),  g = T^-1 c   with
//   T, T^-1, T^T, T^-T  = DCT-II / DCT-III of length N through one complex Stockham FFT of length N/2 in LDS
//   S r, S^T p          = column-streaming GEMV over the 1024 threads of the (single) workgroup.
// One problem = one workgroup running the whole time loop (latency-bound config); `batch` problems = `batch` workgroups
// sharing S through L2.
#include <algorithm>

#include "fft_lds.hpp"

namespace smo {
namespace {

constexpr int NT = 1024;

// ---------------------------------------------------------------------------------------------------------
// host: tau operator
// ---------------------------------------------------------------------------------------------------------
// Unknown / equation numbering interleaved by Chebyshev mode (index 4*n + v) so the system is banded apart from the four
// boundary rows.  Equations (T -> U conversion "Pre" applied, last row of each block replaced by a boundary row):
//   e0: Pre[(1/dt + 1 - a) u + 2 uzz + D uzzz] = Pre rhs      bc: left(uz)   = 0
//   e1: Pre[uz   - D u  ] = 0                                  bc: left(uzzz) = 0
//   e2: Pre[uzz  - D uz ] = 0                                  bc: right(u)   = 0
//   e3: Pre[uzzz - D uzz] = 0                                  bc: right(uzz) = 0
// Pre[n][n] = 1 (n=0) | 1/2, Pre[n][n+2] = -1/2;  (Pre D)[n][n+1] = (n+1)/stretch  (d/dx T_n = n U_{n-1}).
static int build_tau_operator(int N, double dt, double a, double z0, double z1, std::vector<double>& S) {
    const int n4 = 4 * N;
    const double stretch = 0.5 * (z1 - z0), c0 = 1.0 / dt + 1.0 - a;
    std::vector<double> A((size_t)n4 * n4, 0.0), B((size_t)n4 * N, 0.0);
    auto at = [&](int r, int c) -> double& { return A[(size_t)r * n4 + c]; };
    auto pre_row = [&](int n, auto&& f) {            // f(col_mode, weight) over the non-zeros of row n of Pre
        f(n, n == 0 ? 1.0 : 0.5);
        if (n + 2 < N) f(n + 2, -0.5);
    };
    for (int n = 0; n < N - 1; ++n) {                 // rows 0..N-2 of every block; row N-1 holds the boundary condition
        pre_row(n, [&](int j, double w) {
            at(4 * n + 0, 4 * j + 0) += w * c0;  at(4 * n + 0, 4 * j + 2) += w * 2.0;
            at(4 * n + 1, 4 * j + 1) += w;       at(4 * n + 2, 4 * j + 2) += w;     at(4 * n + 3, 4 * j + 3) += w;
            B[(size_t)(4 * n + 0) * N + j] += w;                                      // Pre * rhs
        });
        const double d = (n + 1) / stretch;            // (Pre D)[n][n+1]
        at(4 * n + 0, 4 * (n + 1) + 3) += d;
        at(4 * n + 1, 4 * (n + 1) + 0) -= d;
        at(4 * n + 2, 4 * (n + 1) + 1) -= d;
        at(4 * n + 3, 4 * (n + 1) + 2) -= d;
    }
    const int bc_var[4] = {1, 3, 0, 2};
    const bool bc_left[4] = {true, true, false, false};
    for (int e = 0; e < 4; ++e)
        for (int j = 0; j < N; ++j) at(4 * (N - 1) + e, 4 * j + bc_var[e]) = (bc_left[e] && (j & 1)) ? -1.0 : 1.0;

    // LU with partial pivoting that skips structural zeros (rows keep a "last non-zero column" bound)
    std::vector<int> hi(n4);
    for (int r = 0; r < n4; ++r) {
        int h = 0;
        for (int c = n4 - 1; c >= 0; --c) if (at(r, c) != 0.0) { h = c; break; }
        hi[r] = h;
    }
    for (int k = 0; k < n4; ++k) {
        int p = -1; double best = 0.0;
        for (int r = k; r < n4; ++r) { const double v = std::fabs(at(r, k)); if (v > best) { best = v; p = r; } }
        if (p < 0) { set_error("SHB23: tau matrix is singular at column %d", k); return SMO_ERR_ARG; }
        if (p != k) {
            std::swap_ranges(&at(k, 0), &at(k, 0) + n4, &at(p, 0));
            std::swap_ranges(&B[(size_t)k * N], &B[(size_t)k * N] + N, &B[(size_t)p * N]);
            std::swap(hi[k], hi[p]);
        }
        const double piv = at(k, k);
        const int hk = hi[k];
        for (int r = k + 1; r < n4; ++r) {
            const double f = at(r, k);
            if (f == 0.0) continue;
            const double l = f / piv;
            at(r, k) = 0.0;
            double* ar = &at(r, 0); const double* ak = &at(k, 0);
            for (int c = k + 1; c <= hk; ++c) ar[c] -= l * ak[c];
            double* br = &B[(size_t)r * N]; const double* bk = &B[(size_t)k * N];
            for (int j = 0; j < N; ++j) br[j] -= l * bk[j];
            hi[r] = std::max(hi[r], hk);
        }
    }
    for (int k = n4 - 1; k >= 0; --k) {               // back substitution, all N right-hand sides at once
        double* bk = &B[(size_t)k * N];
        for (int c = k + 1; c <= hi[k]; ++c) {
            const double u = at(k, c);
            if (u == 0.0) continue;
            const double* bc = &B[(size_t)c * N];
            for (int j = 0; j < N; ++j) bk[j] -= u * bc[j];
        }
        const double inv = 1.0 / at(k, k);
        for (int j = 0; j < N; ++j) bk[j] *= inv;
    }
    S.assign((size_t)N * N, 0.0);
    for (int n = 0; n < N; ++n) std::copy(&B[(size_t)(4 * n) * N], &B[(size_t)(4 * n) * N] + N, &S[(size_t)n * N]);
    return SMO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// device: DCT-II / DCT-III (scipy's unnormalised conventions) of length N = 2*NH on an LDS vector
// ---------------------------------------------------------------------------------------------------------
template <int NH> struct DctWork {
    cplx P[NH], bufA[NH], bufB[NH], tw[NH];     // packed sequence, FFT ping-pong, exp(-2 pi i k / NH)
    cplx twN[NH];                               // exp(-2 pi i k / N)
    cplx tw4[NH + 1];                           // exp(-i pi k / (2N))
};

// y[k] = 2 sum_n x[n] cos(pi k (2n+1) / (2N))         (x, y: LDS, may alias)
template <int NH> __device__ void dct2(DctWork<NH>& w, const double* x, double* y, int tid) {
    constexpr int N = 2 * NH;
    for (int n = tid; n < NH; n += NT) {             // Makhoul permutation, two reals per complex
        const double e = (2 * n < NH) ? x[4 * n] : x[2 * N - 1 - 4 * n];
        const double o = (2 * n + 1 < NH) ? x[4 * n + 2] : x[2 * N - 3 - 4 * n];
        w.P[n] = mk(e, o);
    }
    __syncthreads();
    fft_batch<NH, false>(w.bufA, w.bufB, w.tw, 1, NH, tid, NT, [&](int, int pos) { return w.P[pos]; },
                         [&](int, int pos, cplx v) { w.P[pos] = v; });
    __syncthreads();
    for (int k = tid; k <= NH; k += NT) {
        if (k == 0) { const cplx Z0 = w.P[0]; y[0] = 2.0 * (Z0.re + Z0.im); y[NH] = 1.4142135623730951 * (Z0.re - Z0.im); }
        else if (k < NH) {
            const cplx Zk = w.P[k], Zm = conj(w.P[NH - k]);
            const cplx V = 0.5 * (Zk + Zm) + mul_mi(0.5 * (Zk - Zm)) * w.twN[k];
            const cplx t = V * w.tw4[k];
            y[k] = 2.0 * t.re;
            y[N - k] = -2.0 * t.im;
        }
    }
    __syncthreads();
}

// y[n] = x[0] + 2 sum_{k>=1} x[k] cos(pi k (2n+1) / (2N))
template <int NH> __device__ void dct3(DctWork<NH>& w, const double* x, double* y, int tid) {
    constexpr int N = 2 * NH;
    auto V = [&](int k) -> cplx {                    // half spectrum of the permuted sequence, k = 0..NH
        if (k == 0) return mk(x[0], 0.0);
        if (k == NH) return mk(1.4142135623730951 * x[NH], 0.0);
        return mul_conj(mk(x[k], -x[N - k]), w.tw4[k]);
    };
    for (int k = tid; k < NH; k += NT) {
        const cplx a = V(k), b = conj(V(NH - k));
        w.P[k] = (a + b) + mul_i(mul_conj(a - b, w.twN[k]));
    }
    __syncthreads();
    fft_batch<NH, true>(w.bufA, w.bufB, w.tw, 1, NH, tid, NT, [&](int, int pos) { return w.P[pos]; },
                        [&](int, int pos, cplx v) { w.P[pos] = v; });
    __syncthreads();
    for (int m = tid; m < N; m += NT) {              // undo the permutation: y[2j] = v[j], y[2j+1] = v[N-1-j]
        const int j = (m & 1) ? (N - 1 - (m >> 1)) : (m >> 1);
        const cplx z = w.P[j >> 1];
        y[m] = (j & 1) ? z.im : z.re;
    }
    __syncthreads();
}

// out[n] = sum_j M[j][n] v[j]   (M row j contiguous in n: coalesced 16-byte loads; v, out, part in LDS)
__device__ __forceinline__ void gemv_cols(const double* __restrict__ M, const double* v, double* out, double* part, int N, int tid) {
    const int npairs = N >> 1;
    int ngrp = (NT / npairs < N) ? NT / npairs : N;              // N = 32: 32 groups, half the block idles
    while (N % ngrp != 0 || ngrp * N > 2 * NT) --ngrp;           // column groups of equal size whose partial sums fit `part` (N = 3 * 2^k)
    const int cols = N / ngrp;
    const int np = tid % npairs, jg = tid / npairs;
    const double2* M2 = reinterpret_cast<const double2*>(M);
    if (jg < ngrp) {
        double a0 = 0.0, a1 = 0.0;
        const int j0 = jg * cols;
#pragma unroll 8
        for (int j = j0; j < j0 + cols; ++j) {
            const double2 m = M2[(size_t)j * npairs + np];
            const double vj = v[j];
            a0 += m.x * vj;
            a1 += m.y * vj;
        }
        part[jg * N + 2 * np] = a0;
        part[jg * N + 2 * np + 1] = a1;
    }
    __syncthreads();
    for (int n = tid; n < N; n += NT) {
        double s = 0.0;
        for (int gidx = 0; gidx < ngrp; ++gidx) s += part[gidx * N + n];
        out[n] = s;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// Any N (run-time length, NH = 0 in the templates below): the reference's Chebyshev basis takes whatever N it is handed
// (FWD_Solve_SHB23.py:196-217).  DCT-II / DCT-III through ONE complex transform of the FULL length N (Makhoul's N-point form), on the
// run-time-length Stockham chain of fft_lds.hpp — no half-length packing, so odd N works too; tables of N entries.
// ---------------------------------------------------------------------------------------------------------
template <> struct DctWork<0> {
    cplx *P, *T, *tw, *tw4;       // LDS: sequence, ping-pong partner, exp(-2 pi i k / N), exp(-i pi k / (2N)), k < N
    int N;
    AnyPlan pl;
};
// Forced inline, like load_tables<0> below and the kernels' lambdas: NO kernel of this library may call a device function.  Out of line,
// dct3<0> / dct2<0> themselves are fine (flat pointers into the caller's private memory and into LDS, barriers inside the callee, LDS
// above 64 KB: tools/micro_outline_abi.hip passes on the GPU) — what breaks is the CALLER: under register pressure ROCm 7.2's register
// allocator (with IPRA) saves the per-lane values that must survive the call (1/dt, N as a double, LDS addresses) by VGPR copies placed
// AHEAD of the `s_or_b64 exec` that ends the preceding `for (k = tid; k < N; k += NT)` region — i.e. with the tid < N lanes or, coming from the
// loop's exit, with NO lane enabled — and copies them back with all lanes on: garbage in every lane (NaN results, or a wild address = the
// memory fault seen in round 2).  Root-cause analysis, GPU experiments and the static
// detector: DESIGN.md section 4c, tools/run_outline_abi.sh, tools/scan_exec_masked_saves.py; guard: tests/test_no_device_calls.py.
// y[k] = 2 sum_n x[n] cos(pi k (2n+1) / (2N)) = 2 Re(e^{-i pi k/(2N)} V_k),  V = F_N v,  v[j] = x[2j], v[N-1-j] = x[2j+1]
template <> __device__ __forceinline__ void dct2<0>(DctWork<0>& w, const double* x, double* y, int tid) {
    const int N = w.N;
    for (int n = tid; n < N; n += NT) { const int j = (n & 1) ? N - 1 - (n >> 1) : (n >> 1); w.P[j] = mk(x[n], 0.0); }
    __syncthreads();
    const cplx* V = any_fft<false>(w.P, w.T, w.tw, w.pl, 1, tid, NT);
    for (int k = tid; k < N; k += NT) { const cplx t = V[k] * w.tw4[k]; y[k] = 2.0 * t.re; }
    __syncthreads();
}
// y[n] = x[0] + 2 sum_{k>=1} x[k] cos(pi k (2n+1) / (2N)):  v = Re F_N^-1[c_k x_k e^{+i pi k/(2N)}] (c_0 = 1, else 2), y[2j] = v[j], y[2j+1] = v[N-1-j]
template <> __device__ __forceinline__ void dct3<0>(DctWork<0>& w, const double* x, double* y, int tid) {
    const int N = w.N;
    for (int k = tid; k < N; k += NT) { const double a = (k == 0) ? x[0] : 2.0 * x[k]; const cplx t = w.tw4[k]; w.P[k] = mk(a * t.re, -a * t.im); }
    __syncthreads();
    const cplx* v = any_fft<true>(w.P, w.T, w.tw, w.pl, 1, tid, NT);
    for (int m = tid; m < N; m += NT) { const int j = (m & 1) ? N - 1 - (m >> 1) : (m >> 1); y[m] = v[j].re; }
    __syncthreads();
}
// out[n] = sum_j M[j][n] v[j] for any N <= NT: thread <-> (column group, n), consecutive lanes read consecutive n
__device__ __forceinline__ void gemv_any(const double* __restrict__ M, const double* v, double* out, double* part, int N, int tid) {
    const int ngrp = NT / N, cols = (N + ngrp - 1) / ngrp;
    const int jg = tid / N, n = tid - jg * N;
    if (jg < ngrp) {
        double a = 0.0;
        const int j1 = min(N, (jg + 1) * cols);
        for (int j = jg * cols; j < j1; ++j) a += M[(size_t)j * N + n] * v[j];
        part[jg * N + n] = a;
    }
    __syncthreads();
    for (int i = tid; i < N; i += NT) {
        double acc = 0.0;
        for (int q = 0; q < ngrp; ++q) acc += part[q * N + i];
        out[i] = acc;
    }
    __syncthreads();
}
template <int NH> __device__ __forceinline__ void shb_gemv(const double* __restrict__ M, const double* v, double* out, double* part, int N, int tid) {
    if constexpr (NH != 0) gemv_cols(M, v, out, part, N, tid);
    else if ((N & 1) == 0) gemv_cols(M, v, out, part, N, tid);      // even N: the 16-byte column-pair stream (its group count adapts to N)
    else gemv_any(M, v, out, part, N, tid);
}

// ---------------------------------------------------------------------------------------------------------
// Cluster mode (latency): KC workgroups, one per CU, co-operate on ONE problem.  Member k keeps rows [k*R, (k+1)*R) of the
// operator resident in its LDS (N = 512: 32 members x 16 rows x 4 KB = 64 KB each) so a step no longer streams 2 MB from L2
// through one CU; every member redundantly runs the (cheap) transforms, computes its R outputs, and the members all-gather
// the N outputs through HBM/L2 once per step:
//   producer: stores -> each storing lane s_waitcnt vmcnt(0) -> workgroup barrier -> lane 0: agent release fence, vmcnt(0),
//             relaxed agent add on a monotonic counter
//   consumer: lane 0 polls the counter (relaxed agent load, s_sleep, BOUNDED), agent acquire fence, vmcnt(0) -> barrier -> plain
//             loads (cdna_hip_programming.md Guideline 16).  Outputs are double-buffered by step parity.
// A member that times out raises `err` and every member leaves at its next gather: the kernel always terminates.  The members are
// NOT launched co-operatively; the host checks beforehand that the GPU can hold them all at once (occupancy query) and, should a
// member still not arrive (a GPU shared with other work), reruns the call with one workgroup per problem (cluster_retry below).
// ---------------------------------------------------------------------------------------------------------
struct Cluster {
    int KC, k, R, rows;           // members, my index, rows per member (R = ceil(dimension / KC)), rows of THIS member (the last one may hold fewer)
    const double* Ms;             // LDS: my rows of the operator, row-major [R][N]
    double* buf;                  // global [2][N] gathered vector (by step parity)
    unsigned* cnt;                // global monotonic arrival counter (zeroed before the launch)
    unsigned* err;                // global error flag
    unsigned step;                // gathers done so far
    int* flag;                    // LDS word for broadcasting the poll result
};

// y (LDS, full length N) <- M x with M row-sliced over the cluster; returns false on timeout (uniform over the workgroup)
__device__ __forceinline__ bool cluster_gemv(Cluster& c, const double* x, double* y, int N /* operator dimension */, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    double* dst = c.buf + (size_t)(c.step & 1) * N + (size_t)c.k * c.R;
    for (int row = wave; row < c.rows; row += NT / 64) {
        const double* m = c.Ms + (size_t)row * N;
        double a = 0.0;
        for (int j = lane; j < N; j += 64) a += m[j] * x[j];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
        if (lane == 0) {
            __hip_atomic_store(dst + row, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(c.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (c.step + 1u) * (unsigned)c.KC;
        int ok = 0;
        const unsigned spin_max = c.err[1];          // bounded wait: the host sets the cap (default 2^22 polls) and reruns on a time-out
        for (unsigned spin = 0; spin < spin_max; ++spin) {
            if (__hip_atomic_load(c.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            if (__hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) __hip_atomic_store(c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *c.flag = ok;
    }
    __syncthreads();
    const bool ok = (*c.flag != 0);
    if (ok) {
        const double* src = c.buf + (size_t)(c.step & 1) * N;
        for (int n = tid; n < N; n += NT) y[n] = src[n];
    }
    c.step += 1;
    __syncthreads();
    return ok;
}

// member blockIdx.x % KC of the cluster that works on problem blockIdx.x / KC: rows [k R, k R + rows) of the No x No operator `Mrow` go to the
// LDS behind the work area (`ms`) and stay there for the whole solve.  Any No: R = ceil(No / KC), the last member holds the remainder.
__device__ __forceinline__ Cluster cluster_join(int KC, int No, const double* __restrict__ Mrow, double* ms, double* cl_buf, unsigned* cl_cnt,
                                                unsigned* cl_err, size_t prob, int* flag, int tid) {
    const int k = (int)(blockIdx.x % KC), R = (No + KC - 1) / KC;
    Cluster cl{KC, k, R, max(0, min(R, No - k * R)), nullptr, cl_buf + prob * 2 * No, cl_cnt + prob, cl_err, 0u, flag};
    if (KC > 1) {
        const double* src = Mrow + (size_t)k * R * No;
        for (int i = tid; i < cl.rows * No; i += NT) ms[i] = src[i];
        cl.Ms = ms;
    }
    return cl;
}

template <int NH> struct ShbShared {
    DctWork<NH> w;
    double c[2 * NH], g[2 * NH], r[2 * NH], t[2 * NH], W[2 * NH];
    double part[2 * NT];
    double red[NT / 64];
    int flag;
    int pad_[3];
};

template <int NH> __device__ void load_tables(ShbShared<NH>& s, const cplx* tw_g, const cplx* twN_g, const cplx* tw4_g, const double* W_g, int tid) {
    for (int i = tid; i < NH; i += NT) { s.w.tw[i] = tw_g[i]; s.w.twN[i] = twN_g[i]; }
    for (int i = tid; i <= NH; i += NT) s.w.tw4[i] = tw4_g[i];
    for (int i = tid; i < 2 * NH; i += NT) s.W[i] = W_g[i];
    __syncthreads();
}

// any N: the same work area with run-time extents — a struct of LDS pointers instead of an LDS struct (one workgroup per problem only)
template <> struct ShbShared<0> {
    DctWork<0> w;
    double *c, *g, *r, *t, *W, *part, *red;
    int* flag_lds;                                 // cluster mode: the LDS word that broadcasts the poll result
};
// bytes of the work area (a multiple of 16): 4 complex and 5 real vectors, the GEMV partial sums, the reduction words, the cluster flag
__host__ __device__ inline size_t shb_any_lds(int N) { const int Ne = (N + 1) & ~1; return (size_t)4 * N * sizeof(cplx) + ((size_t)5 * Ne + 2 * NT + NT / 64 + 2) * sizeof(double); }
template <int NH> struct ShbRef {
    using type = ShbShared<NH>&;
    static __device__ __forceinline__ type get(unsigned char* smem, int, const AnyPlan&) { return *reinterpret_cast<ShbShared<NH>*>(smem); }
    static __device__ __forceinline__ int* flag(ShbShared<NH>& s) { return &s.flag; }
    static __device__ __forceinline__ double* rows(unsigned char* smem, int) { return reinterpret_cast<double*>(smem + sizeof(ShbShared<NH>)); }
};
template <> struct ShbRef<0> {
    using type = ShbShared<0>;
    static __device__ __forceinline__ type get(unsigned char* smem, int N, const AnyPlan& pl) {
        cplx* z = reinterpret_cast<cplx*>(smem);
        double* d = reinterpret_cast<double*>(z + (size_t)4 * N);
        const int Ne = (N + 1) & ~1;
        return ShbShared<0>{DctWork<0>{z, z + N, z + 2 * N, z + 3 * N, N, pl}, d, d + Ne, d + 2 * Ne, d + 3 * Ne, d + 4 * Ne, d + 5 * Ne, d + 5 * Ne + 2 * NT,
                            reinterpret_cast<int*>(d + 5 * Ne + 2 * NT + NT / 64)};
    }
    static __device__ __forceinline__ int* flag(ShbShared<0>& s) { return s.flag_lds; }
    static __device__ __forceinline__ double* rows(unsigned char* smem, int N) { return reinterpret_cast<double*>(smem + shb_any_lds(N)); }
};
template <> __device__ __forceinline__ void load_tables<0>(ShbShared<0>& s, const cplx* tw_g, const cplx*, const cplx* tw4_g, const double* W_g, int tid) {
    for (int i = tid; i < s.w.N; i += NT) { s.w.tw[i] = tw_g[i]; s.w.tw4[i] = tw4_g[i]; s.W[i] = W_g[i]; }
    __syncthreads();
}

// forward: J = -dt * sum_{n=0}^{N_ITERS} <g_n, g_n>_W ; stack[n] = g_n (grid states)
template <int NH>
__global__ __launch_bounds__(NT) void shb_forward_kernel(const double* __restrict__ X, double* __restrict__ stack, double* __restrict__ Jout,
                                                         const double* __restrict__ ST, const double* __restrict__ W_g,
                                                         const cplx* __restrict__ tw_g, const cplx* __restrict__ twN_g,
                                                         const cplx* __restrict__ tw4_g, double dt, double inv_Lz, int n_iters,
                                                         int KC, const double* __restrict__ Mrow, double* cl_buf, unsigned* cl_cnt,
                                                         unsigned* cl_err, int Nc, int cnts, int n_rt, AnyPlan pl) {
    // N = grid / DCT length; Nc = number of Chebyshev modes the operator acts on (Nc = N for the "Discrete" path; Nc = N/2 for the
    // "Continuous" path: dealias-2 grid, coefficients beyond Nc stay zero, the snapshot stack holds coefficients instead of grid states)
    const int N = NH ? 2 * NH : n_rt;             // NH = 0: run-time length (any N)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typename ShbRef<NH>::type s = ShbRef<NH>::get(smem, N, pl);
    const int tid = threadIdx.x;
    const size_t prob = blockIdx.x / KC;
    const int NS = cnts ? Nc : N;                    // doubles per snapshot
    X += prob * N;
    stack += prob * (size_t)(n_iters + 1) * NS;
    Cluster cl = cluster_join(KC, Nc, Mrow, ShbRef<NH>::rows(smem, N), cl_buf, cl_cnt, cl_err, prob, ShbRef<NH>::flag(s), tid);   // my rows of S
    const bool writer = (cl.k == 0);
    load_tables(s, tw_g, twN_g, tw4_g, W_g, tid);
    for (int i = tid; i < N; i += NT) s.t[i] = X[i];
    __syncthreads();
    // c = T X : dct2 / N, c[0] *= 1/2, odd modes * -1
    dct2(s.w, s.t, s.c, tid);
    for (int k = tid; k < N; k += NT) s.c[k] = (k < Nc) ? s.c[k] * (((k == 0) ? 0.5 : ((k & 1) ? -1.0 : 1.0)) / N) : 0.0;
    __syncthreads();
    double acc = 0.0;
    const double inv_dt = 1.0 / dt;
    for (int it = 0; it <= n_iters; ++it) {
        // g = T^-1 c : dct3 of (c0, s_k c_k / 2)
        for (int k = tid; k < N; k += NT) {
            s.t[k] = (k == 0) ? s.c[0] : ((k & 1) ? -0.5 : 0.5) * s.c[k];
            if (cnts && writer && k < Nc) stack[(size_t)it * NS + k] = s.c[k];
        }
        __syncthreads();
        dct3(s.w, s.t, s.g, tid);
        for (int i = tid; i < N; i += NT) {
            const double gi = s.g[i];
            if (!cnts && writer) stack[(size_t)it * NS + i] = gi;
            acc += s.W[i] * gi * gi;
            s.t[i] = gi * gi * (2.0 - gi);            // 2 g^2 - g^3
        }
        if (it == n_iters) break;
        __syncthreads();
        dct2(s.w, s.t, s.r, tid);                      // h = Z T[...]
        for (int k = tid; k < N; k += NT) {
            const double h = (k < N / 2) ? s.r[k] * (((k == 0) ? 0.5 : ((k & 1) ? -1.0 : 1.0)) / N) : 0.0;
            s.r[k] = h + s.c[k] * inv_dt;
        }
        __syncthreads();
        if (KC > 1) { if (!cluster_gemv(cl, s.r, s.c, Nc, tid)) return; }
        else shb_gemv<NH>(ST, s.r, s.c, s.part, Nc, tid); // c = S r  (entries beyond Nc stay zero)
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) s.red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0 && writer) {
        double tot = 0.0;
        for (int i = 0; i < NT / 64; ++i) tot += s.red[i];
        Jout[prob] = -dt * inv_Lz * tot;
    }
}

// adjoint: p = 2 dt T^-T(W g_N); repeat N_ITERS times: r = S^T p; p = r/dt + T^-T[(4b - 3b^2) T^T r + 2 dt W b]; grad = -T^T p / W
template <int NH>
__global__ __launch_bounds__(NT) void shb_adjoint_kernel(const double* __restrict__ stack, double* __restrict__ grad, const double* __restrict__ Sm,
                                                         const double* __restrict__ W_g, const cplx* __restrict__ tw_g,
                                                         const cplx* __restrict__ twN_g, const cplx* __restrict__ tw4_g, double dt, int n_iters,
                                                         int KC, const double* __restrict__ Mrow, double* cl_buf, unsigned* cl_cnt,
                                                         unsigned* cl_err, int n_rt, AnyPlan pl) {
    const int N = NH ? 2 * NH : n_rt;             // NH = 0: run-time length (any N)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typename ShbRef<NH>::type s = ShbRef<NH>::get(smem, N, pl);
    const int tid = threadIdx.x;
    const size_t prob = blockIdx.x / KC;
    stack += prob * (size_t)(n_iters + 1) * N;
    grad += prob * N;
    Cluster cl = cluster_join(KC, N, Mrow, ShbRef<NH>::rows(smem, N), cl_buf, cl_cnt, cl_err, prob, ShbRef<NH>::flag(s), tid);    // my rows of S^T
    load_tables(s, tw_g, twN_g, tw4_g, W_g, tid);
    const double inv_dt = 1.0 / dt;
    auto tinv_adj = [&](const double* in, double* out) __attribute__((always_inline)) {      // T^-T x = 1/2 s o DCT2(x), s_0 = 1
        dct2(s.w, in, out, tid);
        for (int k = tid; k < N; k += NT) out[k] *= (k & 1) ? -0.5 : 0.5;
        __syncthreads();
    };
    for (int i = tid; i < N; i += NT) s.t[i] = 2.0 * dt * s.W[i] * stack[(size_t)n_iters * N + i];
    __syncthreads();
    tinv_adj(s.t, s.c);                                            // p lives in s.c
    for (int it = 0; it < n_iters; ++it) {
        const double bi = (tid < N) ? stack[(size_t)(n_iters - 1 - it) * N + tid] : 0.0;     // prefetch b under the GEMV
        if (KC > 1) { if (!cluster_gemv(cl, s.c, s.r, N, tid)) return; }
        else shb_gemv<NH>(Sm, s.c, s.r, s.part, N, tid);              // r = S^T p
        for (int k = tid; k < N; k += NT) s.t[k] = ((k & 1) ? -1.0 : 1.0) * s.r[k];          // T^T r = DCT3(s o r) / N
        __syncthreads();
        dct3(s.w, s.t, s.g, tid);
        if (tid < N) s.t[tid] = (4.0 * bi - 3.0 * bi * bi) * (s.g[tid] / N) + 2.0 * dt * s.W[tid] * bi;
        __syncthreads();
        tinv_adj(s.t, s.g);
        for (int k = tid; k < N; k += NT) s.c[k] = s.r[k] * inv_dt + s.g[k];
        __syncthreads();
    }
    for (int k = tid; k < N; k += NT) s.t[k] = ((k & 1) ? -1.0 : 1.0) * s.c[k];
    __syncthreads();
    dct3(s.w, s.t, s.g, tid);
    if (cl.k == 0)
        for (int i = tid; i < N; i += NT) grad[i] = -(s.g[i] / N) / s.W[i];
}

// "Continuous" adjoint (FWD_Solve_SHB23.py:685-794): q(0) = 0; N_ITERS times  q^ <- S (q^/dt + trunc T[(4 uf - 3 uf^2) q - 2 uf]) with uf
// from the coefficient snapshots N, N-1, ..., 1; the gradient is q on the scale-2 grid.  Same tau operator S as the forward solve.
template <int NH>
__global__ __launch_bounds__(NT) void shb_adjoint_cnts_kernel(const double* __restrict__ stack, double* __restrict__ grad, const double* __restrict__ ST,
                                                              const double* __restrict__ W_g, const cplx* __restrict__ tw_g,
                                                              const cplx* __restrict__ twN_g, const cplx* __restrict__ tw4_g, double dt, int n_iters,
                                                              int KC, const double* __restrict__ Mrow, double* cl_buf, unsigned* cl_cnt,
                                                              unsigned* cl_err, int Nc, int n_rt, AnyPlan pl) {
    const int N = NH ? 2 * NH : n_rt;             // NH = 0: run-time length (any N)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typename ShbRef<NH>::type s = ShbRef<NH>::get(smem, N, pl);
    const int tid = threadIdx.x;
    const size_t prob = blockIdx.x / KC;
    stack += prob * (size_t)(n_iters + 1) * Nc;
    grad += prob * N;
    Cluster cl = cluster_join(KC, Nc, Mrow, ShbRef<NH>::rows(smem, N), cl_buf, cl_cnt, cl_err, prob, ShbRef<NH>::flag(s), tid);
    load_tables(s, tw_g, twN_g, tw4_g, W_g, tid);
    const double inv_dt = 1.0 / dt;
    auto to_grid = [&](const double* coeff, double* out) __attribute__((always_inline)) {      // T^-1 of a coefficient vector that is zero beyond Nc (always inlined: no device calls, see dct2<0>)
        for (int k = tid; k < N; k += NT) s.t[k] = (k == 0) ? coeff[0] : ((k & 1) ? -0.5 : 0.5) * coeff[k];
        __syncthreads();
        dct3(s.w, s.t, out, tid);
    };
    for (int k = tid; k < N; k += NT) s.c[k] = 0.0;                 // q^
    __syncthreads();
    for (int it = 0; it < n_iters; ++it) {
        const double* snap = stack + (size_t)(n_iters - it) * Nc;
        for (int k = tid; k < N; k += NT) s.r[k] = (k < Nc) ? snap[k] : 0.0;
        __syncthreads();
        to_grid(s.r, s.g);                                          // uf on the grid
        to_grid(s.c, s.r);                                          // q  on the grid
        for (int i = tid; i < N; i += NT) { const double u = s.g[i]; s.t[i] = (4.0 * u - 3.0 * u * u) * s.r[i] - 2.0 * u; }
        __syncthreads();
        dct2(s.w, s.t, s.r, tid);
        for (int k = tid; k < N; k += NT) {
            const double h = (k < Nc) ? s.r[k] * (((k == 0) ? 0.5 : ((k & 1) ? -1.0 : 1.0)) / N) : 0.0;
            s.r[k] = h + s.c[k] * inv_dt;
        }
        __syncthreads();
        if (KC > 1) { if (!cluster_gemv(cl, s.r, s.c, Nc, tid)) return; }
        else shb_gemv<NH>(ST, s.r, s.c, s.part, Nc, tid);
    }
    to_grid(s.c, s.g);
    if (cl.k == 0)
        for (int i = tid; i < N; i += NT) grad[i] = s.g[i];
}

// standalone Chebyshev maps of the reference (FWD_Solve_SHB23.py:36-67) for the parity tests against its golden vectors
template <int NH>
__global__ __launch_bounds__(NT) void shb_transform_kernel(const double* __restrict__ in, double* __restrict__ out, int which,
                                                           const cplx* __restrict__ tw_g, const cplx* __restrict__ twN_g,
                                                           const cplx* __restrict__ tw4_g, const double* __restrict__ W_g, int n_rt, AnyPlan pl) {
    const int N = NH ? 2 * NH : n_rt;             // NH = 0: run-time length (any N)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typename ShbRef<NH>::type s = ShbRef<NH>::get(smem, N, pl);
    const int tid = threadIdx.x;
    load_tables(s, tw_g, twN_g, tw4_g, W_g, tid);
    for (int i = tid; i < N; i += NT) {
        const double v = in[i], sg = (i & 1) ? -1.0 : 1.0;
        s.t[i] = (which == 1) ? ((i == 0) ? v : 0.5 * sg * v) : ((which == 2) ? sg * v : v);
    }
    __syncthreads();
    if (which == 0 || which == 3) dct2(s.w, s.t, s.g, tid); else dct3(s.w, s.t, s.g, tid);
    for (int i = tid; i < N; i += NT) {
        const double sg = (i & 1) ? -1.0 : 1.0;
        double v = s.g[i];
        if (which == 0) v *= ((i == 0) ? 0.5 : sg) / N;
        else if (which == 2) v /= N;
        else if (which == 3) v *= 0.5 * sg;
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void shb_inner_kernel(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ W,
                                                        double* __restrict__ out, int N, double inv_Lz) {
    __shared__ double red[4];
    const size_t prob = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) acc += x[prob * N + i] * W[i] * y[prob * N + i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[prob] = (red[0] + red[1] + red[2] + red[3]) * inv_Lz;
}

class SHB23 : public Context {
public:
    explicit SHB23(const smo_config& c) { cfg = c; }
    int N = 0, NH = 0;      // grid / DCT length and its half (complex FFT length)
    int Nc = 0;             // Chebyshev modes of the operator (= N "Discrete", = N/2 "Continuous")
    bool cnts = false;      // the reference's Adjoint_type = "Continuous" formulation (smo_config.cost = 1)
    double Lz = 0;
    double *d_S = nullptr, *d_ST = nullptr, *d_W = nullptr, *d_stack = nullptr, *d_out = nullptr;
    cplx *d_tw = nullptr, *d_twN = nullptr, *d_tw4 = nullptr;
    int k_fwd = -1, k_adj = -1;
    int KC = 1;                        // cluster size (1 = one workgroup per problem)
    int cl_rows = 0;                   // operator rows per cluster member
    int spin_log2 = 22;                // a member gives up after 2^spin_log2 polls (SMO_SHB_SPIN_LOG2; tests force the time-out path with 0)
    bool cluster_off = false;          // a gather timed out once: this context stays with one workgroup per problem
    long long cluster_fallbacks = 0;
    int reset_cluster_words() {
        std::vector<unsigned> w((size_t)cfg.batch + 2, 0u);
        w[(size_t)cfg.batch + 1] = 1u << spin_log2;
        SMO_HIP(hipMemcpyAsync(d_clcnt, w.data(), w.size() * sizeof(unsigned), hipMemcpyHostToDevice, stream));
        SMO_HIP(hipStreamSynchronize(stream));          // `w` is a local
        return SMO_OK;
    }
    // cluster size for this launch: all KC members must be resident at once (they spin on each other), so ask the runtime how many
    // workgroups of this kernel a CU holds; anything short of KC => one workgroup per problem
    template <class K> int cluster_size(K kern, size_t lds) {
        if (KC == 1 || cluster_off) return 1;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, lds) != hipSuccess) return 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg.device) != hipSuccess) return 1;
        return ((long long)per_cu * cus >= (long long)cfg.batch * KC) ? KC : 1;
    }
    // run `launch(kc)`; if a cluster gather timed out (the members were not co-resident after all), run it again with one workgroup
    // per problem — same arithmetic order inside a row, so the result is what a cluster run would have given up to the reduction order
    template <class F> int cluster_retry(const char* who, F launch) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            int kc = 1;
            SMO_TRY(launch(attempt == 0 ? -1 : 1, &kc));
            SMO_HIP(hipGetLastError());
            unsigned err = 0;
            SMO_HIP(hipMemcpyAsync(&err, d_clerr, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            SMO_HIP(hipStreamSynchronize(stream));
            if (err == 0) return SMO_OK;
            if (kc == 1) { set_error("SHB23 %s: error flag raised without a cluster (internal)", who); return SMO_ERR_HIP; }
            cluster_off = true; ++cluster_fallbacks;
        }
        set_error("SHB23 %s: cluster all-gather timed out and the single-workgroup rerun failed too", who);
        return SMO_ERR_HIP;
    }
    double info(int key) const override { return key == 0 ? 1.0 : key == 2 ? (double)(cluster_off ? 1 : KC) : (double)cluster_fallbacks; }
    double* d_clbuf = nullptr;
    unsigned *d_clcnt = nullptr, *d_clerr = nullptr;

    int init() override {
        cnts = (cfg.cost == 1);
        Nc = cfg.npts;
        N = cnts ? 2 * Nc : Nc;                    // "Continuous": npts modes, dealias 2 => vectors live on the 2*npts Gauss grid
        NH = N / 2;
        // compile-time instantiations: grid lengths 2^k and 3 * 2^k in [64, 1024]; every other length from 4 to 1024 (odd ones included;
        // the reference takes any N) runs the same kernels with the length a run-time value (NH = 0: any_len; SMO_SHB_ANY=1 forces it)
        if (N < 4 || N > NT) { set_error("SHB23: the grid length must be in [4, %d], got %d", NT, N); return SMO_ERR_UNSUPPORTED; }
        any_len = (N & 1) || !has_instance(NH);
        { const char* e = getenv("SMO_SHB_ANY"); if (e && atoi(e) == 1) any_len = true; }
        if (any_len) plan = any_plan(N);
        Lz = cfg.x1 - cfg.x0;
        n_comp = 1;
        vec_len = (size_t)N;
        snapshot_doubles = (size_t)(cnts ? Nc : N);
        stack_bytes = (size_t)cfg.batch * (cfg.n_iters + 1) * snapshot_doubles * sizeof(double);
        SMO_TRY(base_init());
        std::vector<double> S, ST((size_t)Nc * Nc), z(N), W(N);
        SMO_TRY(build_tau_operator(Nc, cfg.dt, cfg.param, cfg.x0, cfg.x1, S));
        for (int i = 0; i < Nc; ++i)
            for (int j = 0; j < Nc; ++j) ST[(size_t)j * Nc + i] = S[(size_t)i * Nc + j];
        // ascending Gauss-Chebyshev grid and the reference's trapezoid-like weights (FWD_Solve_SHB23.py:69-81)
        const double zc = 0.5 * (cfg.x0 + cfg.x1), zh = 0.5 * (cfg.x1 - cfg.x0);
        for (int i = 0; i < N; ++i) z[i] = zc + zh * (-std::cos(M_PI * (i + 0.5) / N));
        W[0] = 0.5 * (z[1] - z[0]);
        W[N - 1] = 0.5 * (z[N - 1] - z[N - 2]);
        for (int i = 1; i < N - 1; ++i) W[i] = 0.5 * (z[i] - z[i - 1]) + 0.5 * (z[i + 1] - z[i]);
        if (cnts) {
            // <x,y> = (1/Lz) integ(x*y): the product on the grid is transformed, truncated to Nc modes and integrated exactly,
            // integ(T_k) = (Lz/2) * 2/(1-k^2) for even k  =>  quadrature weights  W_i = sum_{k<Nc, even} integ(T_k) * T[k][i]
            for (int i = 0; i < N; ++i) {
                long double acc = 0.0L;
                for (int k = 0; k < Nc; k += 2) {
                    const long double wk = (long double)Lz / (1.0L - (long double)k * k);
                    const long double tki = (k == 0 ? 0.5L : 1.0L) * 2.0L * cosl(M_PIl * k * (2 * i + 1) / (2.0L * N)) / N;
                    acc += wk * tki;
                }
                W[i] = (double)acc;
            }
        }
        std::vector<cplx> twN = twiddles(N), t4 = twiddles(4 * N);
        const std::vector<cplx> twF = any_len ? twN : twiddles(NH);      // table of the complex transform: full length (any N) | half length
        twN.resize(NH);
        t4.resize(any_len ? N : NH + 1);
        SMO_TRY(pool.upload(&d_S, S, stream));
        SMO_TRY(pool.upload(&d_ST, ST, stream));
        SMO_TRY(pool.upload(&d_W, W, stream));
        SMO_TRY(pool.upload(&d_tw, twF, stream));
        SMO_TRY(pool.upload(&d_twN, twN, stream));
        SMO_TRY(pool.upload(&d_tw4, t4, stream));
        SMO_TRY(pool.alloc(&d_stack, (size_t)cfg.batch * (cfg.n_iters + 1) * snapshot_doubles));
        SMO_TRY(pool.alloc(&d_out, (size_t)cfg.batch));
        // latency mode: a single problem is spread over KC CUs, each keeping R = ceil(Nc / KC) rows of the operator in its LDS behind the work
        // area: about 64 KB of rows where they fit (N = 512: 32 members of 16 rows), fewer where the work area leaves less (N = 1024: 6 rows,
        // 171 members), never more rows than a whole number of rounds of the 16 waves; any Nc (the last member holds the remainder);
        // SMO_SHB_CLUSTER=0 disables, SMO_SHB_CLUSTER_ROWS=<R> overrides the row count
        const char* env = getenv("SMO_SHB_CLUSTER");
        if (cfg.batch == 1 && Nc >= 256 && !(env && atoi(env) == 0)) {
            int lds_max = 0;
            SMO_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg.device));
            const size_t work = dispatch([&](auto nh) { return (int)work_lds<decltype(nh)::value>(); });
            int R = std::max(1, 8192 / Nc);
            if (const char* e = getenv("SMO_SHB_CLUSTER_ROWS")) R = std::max(1, std::min(Nc, atoi(e)));
            while (R > 1 && work + (size_t)R * Nc * sizeof(double) > (size_t)lds_max) --R;
            if (R > NT / 64) R -= R % (NT / 64);
            if (work + (size_t)R * Nc * sizeof(double) <= (size_t)lds_max) { cl_rows = R; KC = (Nc + R - 1) / R; }
        }
        SMO_TRY(pool.alloc(&d_clbuf, (size_t)cfg.batch * 2 * Nc));
        SMO_TRY(pool.alloc(&d_clcnt, (size_t)cfg.batch + 2));          // arrival counters | error flag | spin cap
        d_clerr = d_clcnt + cfg.batch;
        { const char* e = getenv("SMO_SHB_SPIN_LOG2"); spin_log2 = e ? std::max(0, std::min(30, atoi(e))) : 22; }
        SMO_TRY(reset_cluster_words());
        // algorithmic bytes: stack written/read once + the operator once + the vector
        const double bytes = cfg.batch * ((double)(cfg.n_iters + 1) * snapshot_doubles * 8.0 + N * 8.0) + (double)Nc * Nc * 8.0;
        k_fwd = timing.add_class("shb_forward_kernel", bytes);
        k_adj = timing.add_class("shb_adjoint_kernel", bytes);
        return SMO_OK;
    }

    bool any_len = false;        // no instantiation for this length: the kernels' NH = 0 form
    AnyPlan plan{};
    static bool has_instance(int nh) {
        for (int v : {32, 48, 96, 192, 384, 64, 128, 256, 512}) if (v == nh) return true;
        return false;
    }
    template <int H> size_t work_lds() const { if constexpr (H == 0) return shb_any_lds(N); else return sizeof(ShbShared<H>); }
    template <class F> int dispatch(F f) {
        if (any_len) return f(std::integral_constant<int, 0>());
        switch (NH) {
            case 32: return f(std::integral_constant<int, 32>());
            case 48: return f(std::integral_constant<int, 48>());       // N = 96, 192, 384, 768: one radix-3 stage
            case 96: return f(std::integral_constant<int, 96>());
            case 192: return f(std::integral_constant<int, 192>());
            case 384: return f(std::integral_constant<int, 384>());
            case 64: return f(std::integral_constant<int, 64>());
            case 128: return f(std::integral_constant<int, 128>());
            case 256: return f(std::integral_constant<int, 256>());
            case 512: return f(std::integral_constant<int, 512>());
        }
        set_error("SHB23: unsupported npts %d", N);
        return SMO_ERR_UNSUPPORTED;
    }

    int forward_dev(const double* const* X, double* J) override {
        have_forward = false;
        SMO_TRY(cluster_retry("forward", [&](int force_kc, int* used) -> int {
            return dispatch([&](auto nh) -> int {
                constexpr int H = decltype(nh)::value;
                auto kern = shb_forward_kernel<H>;
                const size_t lds_cl = work_lds<H>() + (KC > 1 ? (size_t)cl_rows * Nc * sizeof(double) : 0);
                SMO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cl));
                const int kc = force_kc > 0 ? force_kc : cluster_size(kern, lds_cl);
                const size_t lds = kc > 1 ? lds_cl : work_lds<H>();
                *used = kc;
                SMO_TRY(reset_cluster_words());
                ScopedTimer t(timing, k_fwd, stream);
                hipLaunchKernelGGL(kern, dim3(cfg.batch * kc), dim3(NT), lds, stream, X[0], d_stack, d_out, d_ST, d_W, d_tw, d_twN, d_tw4, cfg.dt,
                                   1.0 / Lz, cfg.n_iters, kc, d_S, d_clbuf, d_clcnt, d_clerr, Nc, cnts ? 1 : 0, N, plan);
                return SMO_OK;
            });
        }));
        SMO_HIP(hipMemcpyAsync(J, d_out, cfg.batch * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        have_forward = true;
        return SMO_OK;
    }

    int adjoint_dev(const double* const*, int adjoint_type, double* const* grad) override {
        if ((adjoint_type == SMO_ADJ_CONTINUOUS) != cnts) {
            set_error("SHB23: the %s adjoint belongs to a context created with cost = %d (the two formulations use different grids)",
                      adjoint_type == SMO_ADJ_CONTINUOUS ? "Continuous" : "Discrete", adjoint_type == SMO_ADJ_CONTINUOUS ? 1 : 0);
            return SMO_ERR_ARG;
        }
        return cluster_retry("adjoint", [&](int force_kc, int* used) -> int {
            return dispatch([&](auto nh) -> int {
                constexpr int H = decltype(nh)::value;
                const int No = cnts ? Nc : N;            // operator dimension
                const size_t lds_cl = work_lds<H>() + (KC > 1 ? (size_t)cl_rows * No * sizeof(double) : 0);
                auto go = [&](auto kern, auto&& fire) -> int {
                    SMO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cl));
                    const int kc = force_kc > 0 ? force_kc : cluster_size(kern, lds_cl);
                    *used = kc;
                    SMO_TRY(reset_cluster_words());
                    ScopedTimer t(timing, k_adj, stream);
                    fire(kern, kc, kc > 1 ? lds_cl : work_lds<H>());
                    return SMO_OK;
                };
                if (cnts)
                    return go(shb_adjoint_cnts_kernel<H>, [&](auto kern, int kc, size_t lds) {
                        hipLaunchKernelGGL(kern, dim3(cfg.batch * kc), dim3(NT), lds, stream, d_stack, grad[0], d_ST, d_W, d_tw, d_twN, d_tw4, cfg.dt,
                                           cfg.n_iters, kc, d_S, d_clbuf, d_clcnt, d_clerr, Nc, N, plan);
                    });
                return go(shb_adjoint_kernel<H>, [&](auto kern, int kc, size_t lds) {
                    hipLaunchKernelGGL(kern, dim3(cfg.batch * kc), dim3(NT), lds, stream, d_stack, grad[0], d_S, d_W, d_tw, d_twN, d_tw4, cfg.dt,
                                       cfg.n_iters, kc, d_ST, d_clbuf, d_clcnt, d_clerr, N, plan);
                });
            });
        });
    }

    int inner_dev(const double* x, const double* y, double* out) override {
        hipLaunchKernelGGL(shb_inner_kernel, dim3(cfg.batch), dim3(256), 0, stream, x, y, d_W, d_out, N, 1.0 / Lz);
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipMemcpyAsync(out, d_out, cfg.batch * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }

    int transform_host(int which, const double* in, double* out) override {
        double *d_in = nullptr, *d_o = nullptr;
        SMO_HIP(hipMalloc(&d_in, N * sizeof(double)));
        SMO_HIP(hipMalloc(&d_o, N * sizeof(double)));
        SMO_HIP(hipMemcpyAsync(d_in, in, N * sizeof(double), hipMemcpyHostToDevice, stream));
        int rc = dispatch([&](auto nh) {
            constexpr int H = decltype(nh)::value;
            auto kern = shb_transform_kernel<H>;
            const size_t lds = work_lds<H>();
            SMO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(1), dim3(NT), lds, stream, d_in, d_o, which, d_tw, d_twN, d_tw4, d_W, N, plan);
            return SMO_OK;
        });
        if (rc == SMO_OK && hipMemcpyAsync(out, d_o, N * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) rc = SMO_ERR_HIP;
        (void)hipStreamSynchronize(stream);
        (void)hipFree(d_in);
        (void)hipFree(d_o);
        return rc;
    }

    int snapshot_read(int b, int index, double* out) override {
        const double* src = d_stack + ((size_t)b * (cfg.n_iters + 1) + index) * snapshot_doubles;
        SMO_HIP(hipMemcpyAsync(out, src, snapshot_doubles * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }
};

}  // namespace

Context* make_shb23(const smo_config& cfg) { return new SHB23(cfg); }

}  // namespace smo

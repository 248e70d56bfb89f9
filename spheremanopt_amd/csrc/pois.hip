// POIS — stratified plane-Poiseuille optimal mixing (2-D, Fourier in x, Chebyshev in z), "Discrete" formulation.
//
// Replaces FWD_Solve_Discrete / ADJ_Solve_Discrete / Inner_Prod_Discrete and the transform helpers of
// Example_Problems/Bounded_Domain(Cheby)/Optimal_Mixing/FWD_Solve_Poiseuille.py (:777-1155, :1320-1659, :282-299, :44-118).
//
// The reference steps an 8-variable Dedalus LBVP per x-wavenumber with SciPy DCTs and hand-written transposed solves.  Here:
//   * every transform and its adjoint is a z part and an x part.  The z part is a small dense fp64 GEMM on the matrix cores
//     (v_mfma_f64_16x16x4_f64) with an Nz x Nz matrix (Chebyshev grid <-> T coefficients, with the z derivative / de-aliasing mask folded
//     in where needed) — the adjoint transforms are the transposed z matrices, so the discrete adjoint is exact by construction.  The x
//     part (Hermitian half spectrum <-> real grid, with d/dx, 1/Nx or the 2/3 mask as a per-mode factor) is an LDS FFT (pois_x_to_grid /
//     pois_x_to_coeff) where Nx has an instantiation, else a GEMM with the Nx x 2a matrix of the same map;
//   * the tau solve of the LBVP is the map  S_k : (rhs_u, rhs_v, rhs_rho) -> (u, v, rho, uz, vz, rhoz)  per wavenumber (built once on
//     the host by a banded pivoted LU of the tau system), stored and applied in HODLR form (hodlr.hpp; SMO_POIS_APPLY=dense: as a dense
//     batched complex GEMV); the reference's transposed solve  P^L^H A^-H P^R^H  is S_k^H, packed from the same factors.
// Real fields => only the a = Nx/2 non-negative wavenumbers n = 0..kmax are carried (the reference carries the full complex spectrum of
// a complex-dtype domain, POIS:338; the negative half is the Hermitian mirror).  The forward state lives in the de-aliased modes
// n < ada = (2Nx/3)/2 only; the adjoint state does not (the reference's transposed solves run on every pencil), so the adjoint carries all.
//
// Layouts: coefficient fields [2a][Nz] doubles, row 2n = Re, row 2n+1 = Im of mode n, T index fastest; grid fields [Nx][Nz], z
// fastest = the reference's flat vectors [u.flatten(), v.flatten()] (POIS:160-207); snapshot stack [n][3][2a][Nz].
#include <algorithm>
#include <complex>
#include <thread>

#include "smo_common.hpp"
#include "fft_lds.hpp"
#include "hodlr.hpp"

namespace smo {
namespace {

using cd = std::complex<double>;
typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// batched C = A * B (row-major, dense leading dimensions), fp64 MFMA 16x16x4; 32-deep K slices staged through LDS with the next slice
// prefetched into registers
// ---------------------------------------------------------------------------------------------------------
// optional epilogue: C = A B + alpha * E + X (x) Q   (E laid out like C; X one value per row with stride 3, Q one per column: the rank-one tau
// correction of the z-derivative variables); dyn: A is moved by the launch's `shift` doubles (the snapshot the adjoint step linearises about)
struct GemmDesc { const double* A; const double* B; double* C; const double* E = nullptr; double alpha = 0.0; const double* X = nullptr;
                  const double* Q = nullptr; int dyn = 0; };

// TM x TN tile per workgroup (4 waves as 2 x 2, each (TM/2) x (TN/2) = MI x NI MFMA tiles).  The products of this path are small
// (M, N, K of a few hundred, 2-17 of them per launch): 32 x 32 tiles give 4x more workgroups than 64 x 64 and keep the 256 CUs busy.
template <int TM, int TN>
__global__ __launch_bounds__(256) void pois_gemm(const GemmDesc* __restrict__ descs, int M, int N, int K, long long shift) {
    constexpr int KT = 32;                                   // K slice per LDS stage
    constexpr int MI = TM / 32, NI = TN / 32, NA = TM * KT / 256, NBL = KT * TN / 256;
    static_assert(TM % 32 == 0 && TN % 32 == 0, "tile = multiples of 32");
    GemmDesc d = descs[blockIdx.z];
    if (d.dyn) d.A += shift;
    __shared__ double As[TM][KT + 1];
    __shared__ double Bs[KT][TN + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    const int lr = lane & 15, lk = lane >> 4;
    double4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
    // the next slice is fetched into registers while the current one is multiplied (the launches are small: latency, not bandwidth)
    double ra[NA], rb[NBL];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * 256;
            const int ar = idx / KT, ac = idx % KT;
            ra[i] = (m0 + ar < M && k0 + ac < K) ? d.A[(size_t)(m0 + ar) * K + k0 + ac] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            const int idx = tid + i * 256;
            const int br = idx / TN, bc = idx % TN;
            rb[i] = (k0 + br < K && n0 + bc < N) ? d.B[(size_t)(k0 + br) * N + n0 + bc] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KT) {
#pragma unroll
        for (int i = 0; i < NA; ++i) { const int idx = tid + i * 256; As[idx / KT][idx % KT] = ra[i]; }
#pragma unroll
        for (int i = 0; i < NBL; ++i) { const int idx = tid + i * 256; Bs[idx / TN][idx % TN] = rb[i]; }
        __syncthreads();
        if (k0 + KT < K) fetch(k0 + KT);
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            double a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = As[wm + 16 * i + lr][kk + lk];       // A[row = lane&15][k = lane>>4]
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = Bs[kk + lk][wn + 16 * j + lr];       // B[k = lane>>4][col = lane&15]
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                            // C: col = lane&15, row = (lane>>4) + 4*reg
                const int row = m0 + wm + 16 * i + lk + 4 * r, col = n0 + wn + 16 * j + lr;
                if (row < M && col < N) {
                    double v = acc[i][j][r];
                    if (d.E) v += d.alpha * d.E[(size_t)row * N + col];
                    if (d.X) v += d.X[(size_t)row * 3] * d.Q[col];
                    d.C[(size_t)row * N + col] = v;
                }
            }
}
constexpr int GEMM_T = 32;
static inline void launch_gemm(hipStream_t stream, const GemmDesc* descs, int n, int M, int N, int K, long long shift = 0) {
    hipLaunchKernelGGL((pois_gemm<GEMM_T, GEMM_T>), dim3((N + GEMM_T - 1) / GEMM_T, (M + GEMM_T - 1) / GEMM_T, n), dim3(256), 0, stream, descs, M, N, K, shift);
}

// ---------------------------------------------------------------------------------------------------------
// x transforms as FFTs.  Every x matrix of the Discrete path is a real DFT of length Nx composed with a per-mode factor:
//     coefficients -> grid   g[x] = sum_{|n| < a} f_n c_n e^{+2 pi i n x / Nx}      Xi: f = 1;  XiD: i k;  XiN: 1/Nx;  XiN_DA: 1/Nx for n < ada, else 0
//     grid -> coefficients   c_n  = f_n sum_x g[x] e^{-2 pi i n x / Nx}             Xf: 1/Nx;  Xf_DA: 1/Nx for n < ada, else 0;  XfN: 1;  XfNDa: -i k
// (a = Nx/2 modes, no Nyquist mode; fields [2a][Nz]: row 2n = Re, 2n+1 = Im; grids [Nx][Nz]).  The dense products with those matrices are
// 60 % of the path's flops (2 Nx^2 Nz per field against 5 Nx log2(Nx) Nz); here two z columns share one complex transform of length Nx
// (LDS Stockham stages of fft_lds.hpp), ZT columns per workgroup.  Lengths with an instantiation below; otherwise the dense products stay.
// ---------------------------------------------------------------------------------------------------------
enum { XK_I = 0, XK_ID = 1, XK_IN = 2, XK_IN_DA = 3, XK_F = 4, XK_F_DA = 5, XK_FN = 6, XK_FNDA = 7 };
struct XDesc { const double* src; double* dst; int kind; };
#ifndef SMO_POIS_X_ZT
#define SMO_POIS_X_ZT 8
#endif
constexpr int X_ZT = SMO_POIS_X_ZT, X_NT = 256;          // z columns per workgroup: X_ZT / 2 packed transforms, consecutive lanes on consecutive columns

// `a` = modes carried (n < a): L / 2 in the Discrete formulation (grid of Nx points, no Nyquist mode), L / 3 in the Continuous one (Nx modes on
// the 3/2 grid: the positions a .. L - a of the padded spectrum are zero)
template <int L>
__global__ __launch_bounds__(X_NT) void pois_x_to_grid(const XDesc* __restrict__ descs, const cplx* __restrict__ tw_g, int Nz, int a, int ada, double k1) {
    constexpr int NB = X_ZT / 2;
    __shared__ cplx buf[NB * L];
    __shared__ cplx tw[L];
    const int tid = threadIdx.x;
    for (int i = tid; i < L; i += X_NT) tw[i] = tw_g[i];
    __syncthreads();
    const XDesc d = descs[blockIdx.y];
    const int z0 = blockIdx.x * X_ZT;
    auto mode = [&](int n, int z) -> cplx {                 // f_n c_n of column z (Im c_0 does not enter a real field: the dense matrix has a zero there)
        if (z >= Nz) return mk(0, 0);
        const double re = d.src[(size_t)(2 * n) * Nz + z], im = n == 0 ? 0.0 : d.src[(size_t)(2 * n + 1) * Nz + z];
        switch (d.kind) {
            case XK_ID: { const double k = k1 * n; return mk(-k * im, k * re); }
            case XK_IN: return mk(re / L, im / L);
            case XK_IN_DA: return n < ada ? mk(re / L, im / L) : mk(0, 0);
            default: return mk(re, im);
        }
    };
    auto ld0 = [&](int b, int pos) -> cplx {                // packed Hermitian spectrum of the columns z0 + 2b (real part) and z0 + 2b + 1 (imaginary part)
        if (pos >= a && pos <= L - a) return mk(0, 0);
        const int n = pos < a ? pos : L - pos;
        const cplx A = mode(n, z0 + 2 * b), B = mode(n, z0 + 2 * b + 1);
        return pos < a ? mk(A.re - B.im, A.im + B.re) : mk(A.re + B.im, B.re - A.im);
    };
    auto stN = [&](int b, int x, cplx v) {
        const int z = z0 + 2 * b;
        if (z + 1 < Nz) *reinterpret_cast<cplx*>(d.dst + (size_t)x * Nz + z) = v;       // Nz and z are even: 16-byte aligned
        else if (z < Nz) d.dst[(size_t)x * Nz + z] = v.re;
    };
    fft_inplace_ix<L, true, NB, X_NT, true, false, false>(buf, PosMajor<NB>{}, tw, tid, ld0, stN);
}

template <int L>
__global__ __launch_bounds__(X_NT) void pois_x_to_coeff(const XDesc* __restrict__ descs, const cplx* __restrict__ tw_g, int Nz, int a, int ada, double k1) {
    constexpr int NB = X_ZT / 2;
    __shared__ cplx buf[NB * L];
    __shared__ cplx tw[L];
    const int tid = threadIdx.x;
    for (int i = tid; i < L; i += X_NT) tw[i] = tw_g[i];
    __syncthreads();
    const XDesc d = descs[blockIdx.y];
    const int z0 = blockIdx.x * X_ZT;
    auto ld0 = [&](int b, int x) -> cplx {
        const int z = z0 + 2 * b;
        if (z + 1 < Nz) return *reinterpret_cast<const cplx*>(d.src + (size_t)x * Nz + z);
        return z < Nz ? mk(d.src[(size_t)x * Nz + z], 0.0) : mk(0, 0);
    };
    constexpr PosMajor<NB> ix{};
    fft_inplace_ix<L, false, NB, X_NT, true, false, true>(buf, ix, tw, tid, ld0, [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; });
    __syncthreads();
    for (int t = tid; t < X_ZT * a; t += X_NT) {            // Hermitian split: column z0 + q of mode n
        const int q = t % X_ZT, n = t / X_ZT, b = q >> 1, z = z0 + q;
        if (z >= Nz) continue;
        const cplx hl = buf[ix(b, n)], hh = n == 0 ? hl : buf[ix(b, L - n)];
        const cplx F = (q & 1) ? mk(0.5 * (hl.im + hh.im), -0.5 * (hl.re - hh.re)) : mk(0.5 * (hl.re + hh.re), 0.5 * (hl.im - hh.im));
        cplx c;
        switch (d.kind) {
            case XK_F: c = mk(F.re / L, F.im / L); break;
            case XK_F_DA: c = n < ada ? mk(F.re / L, F.im / L) : mk(0, 0); break;
            case XK_FNDA: { const double k = k1 * n; c = mk(k * F.im, -k * F.re); break; }
            default: c = F;
        }
        d.dst[(size_t)(2 * n) * Nz + z] = c.re;
        d.dst[(size_t)(2 * n + 1) * Nz + z] = n == 0 ? 0.0 : c.im;
    }
}
// grid -> coefficients of the PRODUCT fields of a step, the products formed while the lines are loaded (round 4): pois_x_to_coeff with the pointwise
// kernel folded into its first-stage loads — pois_nl (MODE 0: gr = [u, ux, uz, v, vx, vz, rx, rz] -> NLu, NLv, NLr; the tiles of field 0 also
// leave the step's energy partial) and pois_adj_products (MODE 1: gr = [v1, v2, v3, u, v, ux, vx, rx, uz, vz, rz] -> the 8 (+2 forcing) adjoint
// products) no longer run as kernels of their own and the product fields never exist in memory: one launch and one round trip of 3-10 grid
// fields less per half step.  Same arithmetic per element as the kernels it replaces.
template <int L, int MODE>
__global__ __launch_bounds__(X_NT) void pois_x_prod_to_coeff(const XDesc* __restrict__ descs, const double* __restrict__ gr, size_t nG, const cplx* __restrict__ tw_g, int Nz,
                                                             int a, int ada, double k1, const double* __restrict__ Wz, double* __restrict__ part, double fscale) {
    constexpr int NB = X_ZT / 2;
    __shared__ cplx buf[NB * L];
    __shared__ cplx tw[L];
    const int tid = threadIdx.x;
    for (int i = tid; i < L; i += X_NT) tw[i] = tw_g[i];
    __syncthreads();
    const XDesc d = descs[blockIdx.y];
    const int f = blockIdx.y, z0 = blockIdx.x * X_ZT;
    double acc = 0.0;
    auto G = [&](int k, size_t e) -> cplx {                 // the two columns z, z + 1 of grid field k at x (e = x * Nz + z)
        return *reinterpret_cast<const cplx*>(gr + (size_t)k * nG + e);
    };
    auto G1 = [&](int k, size_t e) -> double { return gr[(size_t)k * nG + e]; };
    auto prod = [&](auto g, int z, int lane) -> double {    // product field f at one grid point; g(k) = value of input field k there
        if (MODE == 0) {
            const double u = g(0), v = g(3);
            if (f == 0) { acc += Wz[z] * (u * u + v * v); return -u * g(1) - v * g(2); }
            return f == 1 ? -u * g(4) - v * g(5) : -u * g(6) - v * g(7);
        }
        switch (f) {
            case 0: return -g(5) * g(0) - g(6) * g(1) - g(7) * g(2);
            case 1: return -g(3) * g(0);
            case 2: return -g(4) * g(0);
            case 3: return -g(8) * g(0) - g(9) * g(1) - g(10) * g(2);
            case 4: return -g(3) * g(1);
            case 5: return -g(4) * g(1);
            case 6: return -g(3) * g(2);
            case 7: return -g(4) * g(2);
            case 8: return fscale * Wz[z] * g(3);
            default: return fscale * Wz[z] * g(4);
        }
    };
    auto ld0 = [&](int b, int x) -> cplx {
        const int z = z0 + 2 * b;
        const size_t e = (size_t)x * Nz + z;
        if (z + 1 < Nz) return mk(prod([&](int k) { return G(k, e).re; }, z, 0), prod([&](int k) { return G(k, e).im; }, z + 1, 1));
        return z < Nz ? mk(prod([&](int k) { return G1(k, e); }, z, 0), 0.0) : mk(0, 0);
    };
    constexpr PosMajor<NB> ix{};
    fft_inplace_ix<L, false, NB, X_NT, true, false, true>(buf, ix, tw, tid, ld0, [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; });
    __syncthreads();
    for (int t = tid; t < X_ZT * a; t += X_NT) {            // Hermitian split: column z0 + q of mode n
        const int q = t % X_ZT, n = t / X_ZT, b = q >> 1, z = z0 + q;
        if (z >= Nz) continue;
        const cplx hl = buf[ix(b, n)], hh = n == 0 ? hl : buf[ix(b, L - n)];
        const cplx F = (q & 1) ? mk(0.5 * (hl.im + hh.im), -0.5 * (hl.re - hh.re)) : mk(0.5 * (hl.re + hh.re), 0.5 * (hl.im - hh.im));
        cplx c;
        switch (d.kind) {
            case XK_F: c = mk(F.re / L, F.im / L); break;
            case XK_F_DA: c = n < ada ? mk(F.re / L, F.im / L) : mk(0, 0); break;
            case XK_FNDA: { const double k = k1 * n; c = mk(k * F.im, -k * F.re); break; }
            default: c = F;
        }
        d.dst[(size_t)(2 * n) * Nz + z] = c.re;
        d.dst[(size_t)(2 * n + 1) * Nz + z] = n == 0 ? 0.0 : c.im;
    }
    if (MODE == 0 && f == 0) {                              // the step's energy partial of this column tile (uniform branch: f is per workgroup)
        __shared__ double red[X_NT / 64];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) { double s2 = 0.0; for (int w = 0; w < X_NT / 64; ++w) s2 += red[w]; part[blockIdx.x] = s2; }
    }
}
template <class F> static inline bool with_xfft_length(int Nx, F f) {
    switch (Nx) {
#define SMO_POIS_X(n) case n: f(std::integral_constant<int, n>()); return true;
        SMO_POIS_X(24) SMO_POIS_X(30) SMO_POIS_X(36) SMO_POIS_X(48) SMO_POIS_X(60) SMO_POIS_X(96) SMO_POIS_X(192) SMO_POIS_X(384) SMO_POIS_X(768)
#undef SMO_POIS_X
    }
    return false;
}
// the x matrices of a context (null where it has none) -> the kind of map a product with one of them is
struct XMats { const double *Xi = nullptr, *XiD = nullptr, *XiN = nullptr, *XiN_DA = nullptr, *Xf = nullptr, *Xf_DA = nullptr, *XfN = nullptr, *XfNDa = nullptr; };
// a phase whose left factors are all x matrices of one direction can run as FFTs: +1 coefficients -> grid, -1 grid -> coefficients, 0 neither
static int x_phase_descs(const std::vector<GemmDesc>& v, const XMats& m, std::vector<XDesc>& xv) {
    int dir = 0;
    xv.clear();
    for (const GemmDesc& g : v) {
        int kind = -1;
        if (g.A == nullptr) kind = -1;
        else if (g.A == m.Xi) kind = XK_I; else if (g.A == m.XiD) kind = XK_ID; else if (g.A == m.XiN) kind = XK_IN; else if (g.A == m.XiN_DA) kind = XK_IN_DA;
        else if (g.A == m.Xf) kind = XK_F; else if (g.A == m.Xf_DA) kind = XK_F_DA; else if (g.A == m.XfN) kind = XK_FN; else if (g.A == m.XfNDa) kind = XK_FNDA;
        const int dd = kind < 0 ? 0 : (kind < XK_F ? 1 : -1);
        if (kind < 0 || g.E || g.X || g.dyn || (dir != 0 && dd != dir)) { xv.clear(); return 0; }
        dir = dd;
        xv.push_back({g.B, g.C, kind});
    }
    return dir;
}
static inline void launch_x(hipStream_t stream, const XDesc* xd, int n, int dir, int L, int nz, int a, int ada, double k1, const cplx* twx) {
    const dim3 grid((unsigned)((nz + X_ZT - 1) / X_ZT), (unsigned)n);
    with_xfft_length(L, [&](auto l) {
        constexpr int LL = decltype(l)::value;
        if (dir > 0) hipLaunchKernelGGL(pois_x_to_grid<LL>, grid, dim3(X_NT), 0, stream, xd, twx, nz, a, ada, k1);
        else hipLaunchKernelGGL(pois_x_to_coeff<LL>, grid, dim3(X_NT), 0, stream, xd, twx, nz, a, ada, k1);
    });
}

// ---------------------------------------------------------------------------------------------------------
// per-wavenumber operator apply: [out fields ; out extras]_n = S_n [in fields ; in extras]_n   (complex)
//   S_n dense (nout*Nz + xout) x (nin*Nz + xin); fields are [f][2a][Nz] coefficient arrays, extras [2a][3] (row 2n = Re, 2n+1 = Im).
// One wave per output row; the operators are streamed once: this is the HBM-bound part of a time step.
// ---------------------------------------------------------------------------------------------------------
constexpr int APPLY_ROWS = 4;      // 8 measured the same (the input staging is then 12 % instead of 25 % of the operator bytes, both from L2)
__global__ __launch_bounds__(256) void pois_apply(const double2* __restrict__ S, const double* __restrict__ in, const double* __restrict__ xin_v,
                                                  double* __restrict__ out, double* __restrict__ xout_v, double* __restrict__ snap, int a, int modes,
                                                  int Nz, int nin, int xin, int nout, int xout, int structure) {
    // APPLY_ROWS consecutive rows of ONE mode per workgroup (wave w takes rows w, w + 4, ...): the mode's input vector is staged in the LDS once instead of being fetched through
    // L1/L2 by every wave (the operator rows are the HBM stream; the input was as many L2 -> L1 requests again).
    extern __shared__ double2 xs[];                      // cols complex inputs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = nout * Nz + xout, cols = nin * Nz + xin, fcols = nin * Nz;
    const int wpm = (rows + APPLY_ROWS - 1) / APPLY_ROWS;      // workgroups per mode
    const int n = blockIdx.x / wpm, r0 = (blockIdx.x - n * wpm) * APPLY_ROWS + wave;
    for (int c = threadIdx.x; c < cols; c += 256) {
        double xr, xi;
        if (c < fcols) {
            const int fi = c / Nz, j = c - fi * Nz;
            const size_t e = ((size_t)fi * 2 * a + 2 * n) * Nz + j;
            xr = in[e]; xi = in[e + Nz];
        } else {
            xr = xin_v[(size_t)(2 * n) * 3 + (c - fcols)]; xi = xin_v[(size_t)(2 * n + 1) * 3 + (c - fcols)];
        }
        xs[c] = double2{xr, xi};
    }
    __syncthreads();
    for (int r = r0; r < rows && r < r0 + APPLY_ROWS; r += 4) {
    const double2* Srow = S + ((size_t)n * rows + r) * cols;
    // structural zeros of the tau operator: the density block is decoupled from the velocity right-hand sides.  structure 1 (forward
    // operator): density rows (field 2 and extra 2) only see the density columns; structure 2 (its conjugate transpose): the velocity
    // rows do not see the density columns (field 2, extra 2).  The skipped entries are exact zeros: not read, not multiplied.
    int c_lo = 0, skip_lo = cols, skip_hi = cols, skip_x = -1;
    if (structure == 1 && ((r < nout * Nz) ? (r / Nz == 2) : (r - nout * Nz == 2))) c_lo = 2 * Nz;
    if (structure == 2 && r < 2 * Nz) { skip_lo = 2 * Nz; skip_hi = 3 * Nz; skip_x = fcols + 2; }
    double yr = 0.0, yi = 0.0;
    for (int c = c_lo + lane; c < cols; c += 64) {
        if ((c >= skip_lo && c < skip_hi) || c == skip_x) continue;
        const double2 x = xs[c];
        const double2 s = Srow[c];
        yr += s.x * x.x - s.y * x.y;
        yi += s.x * x.y + s.y * x.x;
    }
    for (int off = 32; off > 0; off >>= 1) { yr += __shfl_down(yr, off); yi += __shfl_down(yi, off); }
    if (lane == 0) {
        if (r < nout * Nz) {
            const int fo = r / Nz, j = r - fo * Nz;
            const size_t o = ((size_t)fo * 2 * a + 2 * n) * Nz + j;
            out[o] = yr; out[o + Nz] = yi;
            if (snap && fo < 3) { snap[o] = yr; snap[o + Nz] = yi; }
        } else {
            xout_v[(size_t)(2 * n) * 3 + (r - nout * Nz)] = yr; xout_v[(size_t)(2 * n + 1) * 3 + (r - nout * Nz)] = yi;
        }
    }
    }
}
// The same operators in HODLR form (hodlr.hpp): rows and columns mode-major (index 3*mode + variable), every off-diagonal block a
// rank <= 16 product.  One workgroup per (wavenumber, task = tree node at the split depth); lane groups of 8 take the descriptor rows
// round-robin, every row a contiguous, 128-byte aligned run of the operator stream against a contiguous run of the LDS array Z:
//   round 1  t_b = V_b^H x          (the blocks over and under the task's node)
//   copy     G_leaf = [x_leaf | t_b of the blocks over the leaf | extras]
//   round 2  y_r = [D_r | U_b,r ... | extras_r] . G_leaf(r)
// 1.24 MB per wavenumber instead of 4.1 MB at Nz = 192.  Nothing is sequential along the mode index.
__device__ __forceinline__ double2 hodlr_dot(const double2* __restrict__ d, const double2* z, int len, int l) {
    double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
    int i = l;
    for (; i + 32 - l <= len; i += 32) {                    // four independent 16-byte loads per lane in flight
        const double2 s0 = d[i], s1 = d[i + 8], s2 = d[i + 16], s3 = d[i + 24];
        const double2 x0 = z[i], x1 = z[i + 8], x2 = z[i + 16], x3 = z[i + 24];
        ar += s0.x * x0.x - s0.y * x0.y; ai += s0.x * x0.y + s0.y * x0.x;
        br += s1.x * x1.x - s1.y * x1.y; bi += s1.x * x1.y + s1.y * x1.x;
        ar += s2.x * x2.x - s2.y * x2.y; ai += s2.x * x2.y + s2.y * x2.x;
        br += s3.x * x3.x - s3.y * x3.y; bi += s3.x * x3.y + s3.y * x3.x;
    }
    for (; i < len; i += 8) {
        const double2 s0 = d[i], x0 = z[i];
        ar += s0.x * x0.x - s0.y * x0.y; ai += s0.x * x0.y + s0.y * x0.x;
    }
    ar += br; ai += bi;
    for (int off = 4; off > 0; off >>= 1) { ar += __shfl_xor(ar, off); ai += __shfl_xor(ai, off); }
    return double2{ar, ai};
}
__global__ __launch_bounds__(256) void pois_apply_hodlr(const double2* __restrict__ data, size_t stride, const hodlr::Row* __restrict__ rows,
                                                        const uint16_t* __restrict__ lut, const hodlr::Task* __restrict__ tasks, int modes,
                                                        const double* __restrict__ in, const double* __restrict__ xin_v, double* __restrict__ out,
                                                        double* __restrict__ xout_v, double* __restrict__ snap, int a, int Nz, int xin,
                                                        const double* __restrict__ xsrc, const double* __restrict__ q) {
    extern __shared__ double2 Z[];
    // tasks of one wavenumber `modes` blocks apart: the same XCD (modes is a multiple of 8 at the sizes that matter), so the V^H rows of
    // the blocks above the split level, which every task under them reads, come from that XCD's L2 the second time
    const int n = blockIdx.x % modes, w = blockIdx.x / modes;
    const hodlr::Task T = tasks[w];
    const int n3 = 3 * Nz, l = threadIdx.x & 7, g = threadIdx.x >> 3;
    for (int i = n3 + xin + threadIdx.x; i < (int)T.zend; i += 256) Z[i] = double2{0.0, 0.0};
    for (int c = threadIdx.x; c < n3; c += 256) {
        const int fi = c / Nz, j = c - fi * Nz;
        const size_t e = ((size_t)fi * 2 * a + 2 * n) * Nz + j;
        Z[3 * j + fi] = double2{in[e], in[e + Nz]};
    }
    if (xsrc) {
        // the three extra inputs of the transposed operator, x_e = q . lambda_e (lambda_e: the fields of the derivative variables, rows 2n / 2n+1 of
        // xsrc[e]): six dot products of Nz terms, one per 32 lanes (pois_rank1_dot as part of this launch)
        const int dgrp = threadIdx.x >> 5, dl = threadIdx.x & 31;
        if (dgrp < 2 * xin) {
            const int e = dgrp >> 1, part = dgrp & 1;
            const double* src = xsrc + ((size_t)e * 2 * a + 2 * n + part) * Nz;
            double acc = 0.0;
            for (int j = dl; j < Nz; j += 32) acc += q[j] * src[j];
            for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (dl == 0) { if (part == 0) Z[n3 + e].x = acc; else Z[n3 + e].y = acc; }
        }
    } else if ((int)threadIdx.x < xin) {
        Z[n3 + threadIdx.x] = double2{xin_v[(size_t)(2 * n) * 3 + threadIdx.x], xin_v[(size_t)(2 * n + 1) * 3 + threadIdx.x]};
    }
    __syncthreads();
    const double2* base = data + (size_t)n * stride;
    for (int r = g; r < (int)T.n1; r += 32) {
        const hodlr::Row R = rows[T.row1 + r];
        const double2 t = hodlr_dot(base + R.data, Z + R.in, R.len, l);
        if (l == 0) Z[R.out] = t;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (int)T.nlut; i += 256) Z[T.zg + i] = Z[lut[T.lut + i]];
    __syncthreads();
    for (int r = g; r < (int)T.n2; r += 32) {
        const hodlr::Row R = rows[T.row2 + r];
        const double2 y = hodlr_dot(base + R.data, Z + R.in, R.len, l);
        if (l == 0) {
            if ((int)R.out < n3) {
                const int j = R.out / 3, fo = R.out - 3 * j;
                const size_t o = ((size_t)fo * 2 * a + 2 * n) * Nz + j;
                out[o] = y.x; out[o + Nz] = y.y;
                if (snap) { snap[o] = y.x; snap[o + Nz] = y.y; }
            } else {
                xout_v[(size_t)(2 * n) * 3 + (R.out - n3)] = y.x; xout_v[(size_t)(2 * n + 1) * 3 + (R.out - n3)] = y.y;
            }
        }
    }
}
// The z-derivative variables of the tau system differ from the T-space derivative of u, v, rho only along q = Pre^-1 e_{N-1}
// (their defining equations hold in all rows but the dropped one):  uz = u Dz^T + uz_{N-1} q.  Only the LAST row of the operator
// block of each derivative variable is therefore stored (the "extras" of pois_apply): half the operator bytes.
//   forward : dst[f][r][j] += x[r][f] * q[j]                       (dst = the three derivative fields, already holding u Dz^T etc.)
__global__ __launch_bounds__(256) void pois_rank1_add(double* __restrict__ dst, const double* __restrict__ x, const double* __restrict__ q, int rows, int Nz) {
    const size_t n = (size_t)3 * rows * Nz;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i % Nz), r = (int)((i / Nz) % rows), f = (int)(i / ((size_t)rows * Nz));
        dst[i] += x[(size_t)r * 3 + f] * q[j];
    }
}
//   adjoint : x[r][f] = sum_j q[j] * src[f][r][j]                  (one wave per (f, r))
__global__ __launch_bounds__(256) void pois_rank1_dot(double* __restrict__ x, const double* __restrict__ src, const double* __restrict__ q, int rows, int Nz) {
    const int lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= 3LL * rows) return;
    const int f = (int)(w / rows), r = (int)(w - (long long)f * rows);
    double acc = 0.0;
    for (int j = lane; j < Nz; j += 64) acc += q[j] * src[((size_t)f * rows + r) * Nz + j];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) x[(size_t)r * 3 + f] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// pointwise kernels (grid fields [Nx][Nz], z fastest; nG = Nx*Nz)
// ---------------------------------------------------------------------------------------------------------
constexpr int NPART = 256;

__device__ __forceinline__ void block_sum_store(double acc, double* dst) {
    __shared__ double red[4];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *dst = red[0] + red[1] + red[2] + red[3];
}

// forward: g = [u, ux, uz, v, vx, vz, rx, rz] -> nl = [NLu, NLv, NLr] (POIS:911-921); part <- sum W (u^2 + v^2)
__global__ __launch_bounds__(256) void pois_nl(const double* __restrict__ g, double* __restrict__ nl, const double* __restrict__ Wz,
                                               double* __restrict__ part, size_t nG, int Nz) {
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nG; i += (size_t)gridDim.x * 256) {
        const double u = g[i], ux = g[nG + i], uz = g[2 * nG + i], v = g[3 * nG + i], vx = g[4 * nG + i], vz = g[5 * nG + i],
                     rx = g[6 * nG + i], rz = g[7 * nG + i];
        nl[i] = -u * ux - v * uz;
        nl[nG + i] = -u * vx - v * vz;
        nl[2 * nG + i] = -u * rx - v * rz;
        acc += Wz[i % Nz] * (u * u + v * v);
    }
    block_sum_store(acc, part + blockIdx.x);
}
// part <- sum W (p^2 + q^2) of two grid fields;  out0/out1 (optional) <- scale * W * p, scale * W * q
__global__ __launch_bounds__(256) void pois_wsq(const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ Wz,
                                                double* __restrict__ part, double* out0, double* out1, double scale, size_t nG, int Nz) {
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nG; i += (size_t)gridDim.x * 256) {
        const double w = Wz[i % Nz], a = p[i], b = q[i];
        acc += w * (a * a + b * b);
        if (out0) { out0[i] = scale * w * a; out1[i] = scale * w * b; }
    }
    block_sum_store(acc, part + blockIdx.x);
}
// dst = a * x + y over n doubles (y may be null)
__global__ __launch_bounds__(256) void pois_axpy(double* __restrict__ dst, const double* x, double a, const double* y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = a * x[i] + (y ? y[i] : 0.0);
}
// dst (coefficient field) = i k * src  (row 2n: -k Im, row 2n+1: k Re)
__global__ __launch_bounds__(256) void pois_ik(double* __restrict__ dst, const double* __restrict__ src, double k1, int a, int Nz) {
    const size_t n = (size_t)a * Nz;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / Nz), j = (int)(i - (size_t)m * Nz);
        const double k = k1 * m, re = src[((size_t)2 * m) * Nz + j], im = src[((size_t)2 * m + 1) * Nz + j];
        dst[((size_t)2 * m) * Nz + j] = -k * im;
        dst[((size_t)2 * m + 1) * Nz + j] = k * re;
    }
}
// adjoint: products of the adjoint grids v1..v3 with the forward state grids (NLtermAdj, POIS:1510-1524) [+ the KE forcing]
//   in : gr = [v1, v2, v3, u, v, ux, vx, rx, uz, vz, rz]
//   out: pr = [adju, adjux, adjuz, adjv, adjvx, adjvz, adjrx, adjrz, (fu, fv)]
__global__ __launch_bounds__(256) void pois_adj_products(const double* __restrict__ gr, double* __restrict__ pr, const double* __restrict__ Wz,
                                                         double fscale, int forcing, size_t nG, int Nz) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nG; i += (size_t)gridDim.x * 256) {
        const double v1 = gr[i], v2 = gr[nG + i], v3 = gr[2 * nG + i], u = gr[3 * nG + i], v = gr[4 * nG + i], ux = gr[5 * nG + i],
                     vx = gr[6 * nG + i], rx = gr[7 * nG + i], uz = gr[8 * nG + i], vz = gr[9 * nG + i], rz = gr[10 * nG + i];
        pr[i] = -ux * v1 - vx * v2 - rx * v3;
        pr[nG + i] = -u * v1;
        pr[2 * nG + i] = -v * v1;
        pr[3 * nG + i] = -uz * v1 - vz * v2 - rz * v3;
        pr[4 * nG + i] = -u * v2;
        pr[5 * nG + i] = -v * v2;
        pr[6 * nG + i] = -u * v3;
        pr[7 * nG + i] = -v * v3;
        if (forcing) { const double w = fscale * Wz[i % Nz]; pr[8 * nG + i] = w * u; pr[9 * nG + i] = w * v; }
    }
}
// adjoint state update (POIS:1624-1634): hc = the ten transformed products, a3 = S^H lambda
__global__ __launch_bounds__(256) void pois_adj_combine(double* __restrict__ L6, const double* __restrict__ a3, const double* __restrict__ hc,
                                                        double inv_dt, int forcing, size_t nC) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nC; i += (size_t)gridDim.x * 256) {
        double lu = a3[i] * inv_dt + hc[i] + hc[nC + i], lv = a3[nC + i] * inv_dt + hc[3 * nC + i] + hc[4 * nC + i];
        if (forcing) { lu += hc[8 * nC + i]; lv += hc[9 * nC + i]; }
        L6[i] = lu;
        L6[nC + i] = lv;
        L6[2 * nC + i] = a3[2 * nC + i] * inv_dt + hc[6 * nC + i];
        L6[3 * nC + i] = hc[2 * nC + i];
        L6[4 * nC + i] = hc[5 * nC + i];
        L6[5 * nC + i] = hc[7 * nC + i];
    }
}
// grad = (V / W) * g  for the two components
__global__ __launch_bounds__(256) void pois_grad_out(double* __restrict__ out, const double* __restrict__ g, const double* __restrict__ Wz,
                                                     double V, size_t n2G, int Nz) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2G; i += (size_t)gridDim.x * 256) out[i] = V / Wz[i % Nz] * g[i];
}
__global__ __launch_bounds__(256) void pois_dot(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ Wz,
                                                double* __restrict__ part, size_t n2G, int Nz) {
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2G; i += (size_t)gridDim.x * 256) acc += Wz[i % Nz] * x[i] * y[i];
    block_sum_store(acc, part + blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// host: Chebyshev pieces and the tau systems
// ---------------------------------------------------------------------------------------------------------
struct Cheb {
    int N;
    std::vector<double> Pre, D, PD, M1, M2, integ;          // dense N x N (row-major); PD = Pre * D
    explicit Cheb(int n) : N(n), Pre((size_t)n * n, 0.0), D((size_t)n * n, 0.0), PD((size_t)n * n, 0.0), M1((size_t)n * n, 0.0),
                           M2((size_t)n * n, 0.0), integ(n, 0.0) {
        for (int i = 0; i < N; ++i) {
            Pre[(size_t)i * N + i] = i == 0 ? 1.0 : 0.5;
            if (i + 2 < N) Pre[(size_t)i * N + i + 2] = -0.5;
            for (int j = i + 1; j < N; ++j) D[(size_t)i * N + j] = ((j - i) & 1) ? (i == 0 ? 1.0 : 2.0) * j : 0.0;
            integ[i] = (i & 1) ? 0.0 : 2.0 / (1.0 - (double)i * i);
        }
        for (int i = 0; i + 1 < N; ++i) PD[(size_t)i * N + i + 1] = i + 1;            // d/dz T_n = n U_{n-1}
        auto mult = [&](std::vector<double>& M, int j, double fj) {                    // T_j T_m = (T_{m+j} + T_{|m-j|}) / 2
            for (int m = 0; m < N; ++m) {
                if (m + j < N) M[(size_t)(m + j) * N + m] += 0.5 * fj;
                M[(size_t)std::abs(m - j) * N + m] += 0.5 * fj;
            }
        };
        mult(M1, 0, 0.5); mult(M1, 2, -0.5);                                           // 1 - z^2
        mult(M2, 1, -2.0);                                                             // -2 z
    }
    double pre_times(const std::vector<double>& M, int r, int c) const {              // (Pre * M)[r][c]
        double s = (r == 0 ? 1.0 : 0.5) * M[(size_t)r * N + c];
        if (r + 2 < N) s -= 0.5 * M[(size_t)(r + 2) * N + c];
        return s;
    }
};

// Solve A X = B (A n x n with structural zeros, B n x m) in place by LU with partial pivoting.  The tau systems are banded (unknowns
// and equations interleaved by Chebyshev mode) apart from a few dense boundary rows kept at the bottom (rows >= nb): column k can only
// be non-zero in the `win` rows below the diagonal and in the dense rows, so those are the pivot candidates and the rows to
// eliminate; a per-row "last non-zero column" bound keeps the row operations inside the (growing) band.
static int banded_solve(int n, int nb, int win, std::vector<cd>& A, int m, std::vector<cd>& B) {
    auto at = [&](int r, int c) -> cd& { return A[(size_t)r * n + c]; };
    std::vector<int> hi(n, 0);
    for (int r = 0; r < n; ++r)
        for (int c = n - 1; c >= 0; --c) if (at(r, c) != cd(0)) { hi[r] = c; break; }
    for (int k = 0; k < n; ++k) {
        const int wend = std::min(n, k + win), dense0 = std::max(wend, nb);
        int p = -1; double best = 0.0;
        auto consider = [&](int r) { const double v = std::abs(at(r, k)); if (v > best) { best = v; p = r; } };
        for (int r = k; r < wend; ++r) consider(r);
        for (int r = dense0; r < n; ++r) consider(r);
        if (p < 0) { set_error("POIS: tau matrix is singular at column %d of %d", k, n); return SMO_ERR_ARG; }
        if (p != k) {
            std::swap_ranges(&at(k, 0), &at(k, 0) + n, &at(p, 0));
            std::swap_ranges(&B[(size_t)k * m], &B[(size_t)k * m] + m, &B[(size_t)p * m]);
            std::swap(hi[k], hi[p]);
        }
        const cd piv = at(k, k);
        const int hk = hi[k];
        auto elim = [&](int r) {
            const cd f = at(r, k);
            if (f == cd(0)) return;
            const cd l = f / piv;
            at(r, k) = 0;
            cd* ar = &at(r, 0); const cd* ak = &at(k, 0);
            for (int c = k + 1; c <= hk; ++c) ar[c] -= l * ak[c];
            cd* br = &B[(size_t)r * m]; const cd* bk = &B[(size_t)k * m];
            for (int j = 0; j < m; ++j) br[j] -= l * bk[j];
            hi[r] = std::max(hi[r], hk);
        };
        for (int r = k + 1; r < wend; ++r) elim(r);
        for (int r = std::max(dense0, k + 1); r < n; ++r) elim(r);
    }
    for (int k = n - 1; k >= 0; --k) {
        cd* bk = &B[(size_t)k * m];
        for (int c = k + 1; c <= hi[k]; ++c) {
            const cd u = at(k, c);
            if (u == cd(0)) continue;
            const cd* bc = &B[(size_t)c * m];
            for (int j = 0; j < m; ++j) bk[j] -= u * bc[j];
        }
        const cd inv = 1.0 / at(k, k);
        for (int j = 0; j < m; ++j) bk[j] *= inv;
    }
    return SMO_OK;
}

// S_n (6N x 3N) of the momentum / density LBVP (POIS:818-841) for native wavenumber n (k = n * k1).
// Unknown index 7*mode + var (var: u v rho uz vz rhoz p) [+ Fb at the end for n = 0]; rows: for mode m < N-1 the seven equations
// (three tau-reduced evolution equations, continuity, three tau-reduced derivative definitions), then continuity of mode N-1, the
// six boundary / gauge rows [and integ(rho) = 0 for n = 0].
// adjoint = the operator of the script's adjoint IVP (POIS:1217-1252): advection by -U, Ri*w coupled into the density equation and
// Uz*u into the w equation (instead of Ri*rho into w and Uz*w into u).
static int build_solve_map(const Cheb& ch, int n, double k, double a0, double Re, double Pe, double Ri, std::vector<cd>& S, bool adjoint = false) {
    const int N = ch.N, nv = 7 * N + (n == 0 ? 1 : 0), nb = 7 * (N - 1);
    std::vector<cd> A((size_t)nv * nv, cd(0)), B((size_t)nv * 3 * N, cd(0));
    auto at = [&](int r, int c) -> cd& { return A[(size_t)r * nv + c]; };
    enum { U = 0, V = 1, R = 2, UZ = 3, VZ = 4, RZ = 5, P = 6 };
    const cd ik(0.0, k), adv(0.0, adjoint ? -k : k);
    for (int m = 0; m < N - 1; ++m) {
        const int r0 = 7 * m;
        for (int c = std::max(0, m - 2); c < std::min(N, m + 5); ++c) {
            const double pre = ch.Pre[(size_t)m * N + c], pm1 = ch.pre_times(ch.M1, m, c), pm2 = ch.pre_times(ch.M2, m, c),
                         pd = ch.PD[(size_t)m * N + c];
            at(r0 + 0, 7 * c + U) += (a0 + k * k / Re) * pre + adv * pm1;  at(r0 + 0, 7 * c + UZ) += -pd / Re;
            at(r0 + 0, 7 * c + P) += ik * pre;
            at(r0 + 1, 7 * c + V) += (a0 + k * k / Re) * pre + adv * pm1;  at(r0 + 1, 7 * c + VZ) += -pd / Re;
            at(r0 + 1, 7 * c + P) += pd;
            at(r0 + 2, 7 * c + R) += (a0 + k * k / Pe) * pre + adv * pm1;  at(r0 + 2, 7 * c + RZ) += -pd / Pe;
            if (!adjoint) { at(r0 + 0, 7 * c + V) += pm2;  at(r0 + 1, 7 * c + R) += Ri * pre; }
            else          { at(r0 + 1, 7 * c + U) += pm2;  at(r0 + 2, 7 * c + V) += Ri * pre; }
            at(r0 + 4, 7 * c + UZ) += pre;  at(r0 + 4, 7 * c + U) += -pd;
            at(r0 + 5, 7 * c + VZ) += pre;  at(r0 + 5, 7 * c + V) += -pd;
            at(r0 + 6, 7 * c + RZ) += pre;  at(r0 + 6, 7 * c + R) += -pd;
            for (int e = 0; e < 3; ++e) B[(size_t)(r0 + e) * 3 * N + e * N + c] = pre;
        }
        if (n == 0) at(r0 + 2, 7 * N) += ch.Pre[(size_t)m * N + 0];                   // + Fb (constant = its T0 coefficient)
        at(r0 + 3, 7 * m + U) += ik;  at(r0 + 3, 7 * m + VZ) += 1.0;                   // dx(u) + vz = 0
    }
    int row = nb;
    at(row, 7 * (N - 1) + U) += ik;  at(row, 7 * (N - 1) + VZ) += 1.0;  ++row;
    auto functional = [&](int var, int kind) {                                         // 0 left, 1 right, 2 integ
        for (int j = 0; j < N; ++j) at(row, 7 * j + var) = kind == 0 ? ((j & 1) ? -1.0 : 1.0) : (kind == 1 ? 1.0 : ch.integ[j]);
        ++row;
    };
    functional(U, 0); functional(V, 0); functional(U, 1);
    if (n != 0) functional(V, 1); else functional(P, 2);
    functional(RZ, 0); functional(RZ, 1);
    if (n == 0) functional(R, 2);
    SMO_TRY(banded_solve(nv, nb, 7 * 6, A, 3 * N, B));
    S.assign((size_t)6 * N * 3 * N, cd(0));
    for (int var = 0; var < 6; ++var)
        for (int j = 0; j < N; ++j) std::copy(&B[(size_t)(7 * j + var) * 3 * N], &B[(size_t)(7 * j + var) * 3 * N] + 3 * N, &S[((size_t)var * N + j) * 3 * N]);
    return SMO_OK;
}
// S^MN_n (2N x N): rho -> (psi, psiz),  dx dx psi + dz psiz + F = rho,  psiz = dz psi,  psiz(+-1) = 0,  integ psi = 0 at n = 0
static int build_mixnorm_map(const Cheb& ch, int n, double k, std::vector<cd>& S) {
    const int N = ch.N, nv = 2 * N + (n == 0 ? 1 : 0), nb = 2 * (N - 1);
    std::vector<cd> A((size_t)nv * nv, cd(0)), B((size_t)nv * N, cd(0));
    auto at = [&](int r, int c) -> cd& { return A[(size_t)r * nv + c]; };
    for (int m = 0; m < N - 1; ++m) {
        for (int c = m; c < std::min(N, m + 3); ++c) {
            const double pre = ch.Pre[(size_t)m * N + c], pd = ch.PD[(size_t)m * N + c];
            at(2 * m, 2 * c) += -k * k * pre;  at(2 * m, 2 * c + 1) += pd;
            at(2 * m + 1, 2 * c + 1) += pre;   at(2 * m + 1, 2 * c) += -pd;
            B[(size_t)(2 * m) * N + c] = pre;
        }
        if (n == 0) at(2 * m, 2 * N) += ch.Pre[(size_t)m * N + 0];
    }
    for (int j = 0; j < N; ++j) { at(nb, 2 * j + 1) = (j & 1) ? -1.0 : 1.0; at(nb + 1, 2 * j + 1) = 1.0; }
    if (n == 0) for (int j = 0; j < N; ++j) at(nb + 2, 2 * j) = ch.integ[j];
    SMO_TRY(banded_solve(nv, nb, 2 * 4, A, N, B));
    S.assign((size_t)2 * N * N, cd(0));
    for (int var = 0; var < 2; ++var)
        for (int j = 0; j < N; ++j) std::copy(&B[(size_t)(2 * j + var) * N], &B[(size_t)(2 * j + var) * N] + N, &S[((size_t)var * N + j) * N]);
    return SMO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// HODLR operator sets on the device (hodlr.hpp): shared by the Discrete and the Continuous formulation
// ---------------------------------------------------------------------------------------------------------
struct HOp { double2* data = nullptr; hodlr::Row* rows = nullptr; uint16_t* lut = nullptr; hodlr::Task* tasks = nullptr;
             size_t stride = 0; int W = 0, xin = 0, max_rank = 0; unsigned lds = 0; };

static int pois_apply_mode(bool* use_hodlr) {             // SMO_POIS_APPLY = hodlr (default) | dense (the dense operator stream, kept as the test reference)
    const char* mode = getenv("SMO_POIS_APPLY");
    *use_hodlr = !(mode && std::string(mode) == "dense");
    if (mode && *use_hodlr && std::string(mode) != "hodlr") { set_error("SMO_POIS_APPLY must be hodlr or dense, got %s", mode); return SMO_ERR_ARG; }
    return SMO_OK;
}
// a reduced operator ((3N + 3) x 3N, rows and columns variable-major: the rows of u, v, rho and the three extra rows) -> mode-major
// ordering (index 3*mode + variable) of the square part, its HODLR factors, and the extra rows in the same column order
static void hodlr_factor_reduced(const hodlr::Plan& plan, const cd* red, int N, std::vector<cd>& perm, hodlr::Factors& f, std::vector<cd>& extras) {
    const int n3 = 3 * N;
    const double rel_tol = getenv("SMO_POIS_HODLR_TOL") ? atof(getenv("SMO_POIS_HODLR_TOL")) : 1e-14;
    perm.resize((size_t)n3 * n3);
    extras.resize((size_t)3 * n3);
    double mx = 0.0;
    for (int v = 0; v < 3; ++v) for (int j = 0; j < N; ++j) {
        const cd* row = red + (size_t)(v * N + j) * n3;
        cd* prow = &perm[(size_t)(3 * j + v) * n3];
        for (int w = 0; w < 3; ++w) for (int i = 0; i < N; ++i) { prow[3 * i + w] = row[w * N + i]; mx = std::max(mx, std::abs(row[w * N + i])); }
    }
    for (int e = 0; e < 3; ++e) for (int w = 0; w < 3; ++w) for (int i = 0; i < N; ++i) extras[(size_t)e * n3 + 3 * i + w] = red[(size_t)(n3 + e) * n3 + w * N + i];
    hodlr::factor(plan, perm.data(), n3, rel_tol * mx, f);
}
// pack the first `count` operators (or their conjugate transposes) with one set of descriptors — every block gets the largest rank found
// for it in any of the operators — and upload them
static int hop_build(DevPool& pool, hipStream_t stream, const hodlr::Plan& plan, const std::vector<hodlr::Factors>& fac, const std::vector<std::vector<cd>>& extras,
                     int count, bool adjoint, int nthr, HOp& h, double* bytes) {
    std::vector<int> K(plan.blocks.size(), 0), KH(plan.blocks.size(), 0);
    for (const hodlr::Factors& f : fac) for (size_t b = 0; b < K.size(); ++b) K[b] = std::max(K[b], f.rank[b]);
    for (size_t b = 0; b < K.size(); ++b) { KH[b] = K[plan.blocks[b].pair]; h.max_rank = std::max(h.max_rank, K[b]); }
    // workgroups per wavenumber = tree nodes at this depth.  384 x 192 (MI355X): depth 1 52 us per launch, 2: 41, 3: 40, 4: 46 (the V^H
    // rows of the blocks above the split are read by every task under them; below depth 3 there are too few workgroups)
    const int split = getenv("SMO_POIS_HODLR_SPLIT") ? atoi(getenv("SMO_POIS_HODLR_SPLIT")) : 3;
    if (split < 0 || split > 6) { set_error("SMO_POIS_HODLR_SPLIT must be in [0, 6]"); return SMO_ERR_ARG; }
    const hodlr::Layout L = adjoint ? hodlr::make_layout(plan, KH, split, 3, 0) : hodlr::make_layout(plan, K, split, 0, 3);
    std::vector<cd> data((size_t)count * L.stride);
    auto packer = [&](int t) { for (int n = t; n < count; n += nthr) hodlr::pack(plan, L, fac[n], extras[n].data(), adjoint, &data[(size_t)n * L.stride]); };
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; ++t) th.emplace_back(packer, t);
    for (auto& t : th) t.join();
    h.stride = L.stride; h.W = L.W; h.xin = L.xin; h.lds = L.lds_entries * (unsigned)sizeof(double2);
    if (h.lds > 160u * 1024u) { set_error("POIS: the HODLR apply needs %u bytes of LDS", h.lds); return SMO_ERR_UNSUPPORTED; }
    if (h.lds > 64u * 1024u) SMO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pois_apply_hodlr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds));
    SMO_TRY(pool.alloc(&h.data, data.size()));
    SMO_HIP(hipMemcpyAsync(h.data, data.data(), data.size() * sizeof(cd), hipMemcpyHostToDevice, stream));
    SMO_HIP(hipStreamSynchronize(stream));
    SMO_TRY(pool.upload(&h.rows, L.rows, stream)); SMO_TRY(pool.upload(&h.lut, L.lut, stream)); SMO_TRY(pool.upload(&h.tasks, L.tasks, stream));
    *bytes = (double)count * (double)L.stride * 16.0;
    return SMO_OK;
}
static inline void hop_launch(const HOp& h, hipStream_t stream, int modes, const double* in, const double* xin_v, double* out, double* xout_v, double* snap,
                              int a, int Nz, const double* xsrc = nullptr, const double* q = nullptr) {
    hipLaunchKernelGGL(pois_apply_hodlr, dim3((unsigned)(modes * h.W)), dim3(256), h.lds, stream, h.data, h.stride, h.rows, h.lut, h.tasks, modes, in, xin_v,
                       out, xout_v, snap, a, Nz, h.xin, xsrc, q);
}

// ---------------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------------
class Pois : public Context {
public:
    explicit Pois(const smo_config& c) { cfg = c; }
    int Nx = 0, Nz = 0, a = 0, ada = 0, Nz0 = 0, s_cost = 0;      // a: modes carried (n = 0..kmax); ada: de-aliased modes (n < ada)
    double Lx = 0, k1 = 0, V = 0, Re = 0, Ri = 0, Pe = 0, delta = 0;
    size_t nC = 0, nG = 0;
    // matrices
    double *B_ZiT = nullptr, *B_DZiT = nullptr, *B_ZfT = nullptr, *B_ZfT_DA = nullptr, *B_Zf = nullptr, *B_Zf_DA = nullptr, *B_Zi = nullptr,
           *B_ZiDz = nullptr, *B_DzT = nullptr, *B_Dz = nullptr;
    double *A_Xi = nullptr, *A_XiD = nullptr, *A_XiN = nullptr, *A_XiN_DA = nullptr, *A_Xf = nullptr, *A_Xf_DA = nullptr, *A_XfN = nullptr, *A_XfNDa = nullptr;
    double *d_Wz = nullptr, *d_rho0 = nullptr, *d_rz0 = nullptr;
    double2 *d_S = nullptr, *d_SH = nullptr, *d_SMN = nullptr, *d_SMNH = nullptr;   // d_S: (3Nz+3) x 3Nz per wavenumber, d_SH its conjugate transpose
    double *d_qz = nullptr;                                                         // q on the Gauss grid: qz[z] = sum_j Ti[z][j] q[j]
    double *d_q = nullptr, *d_X3 = nullptr;                                         // q = Pre^-1 e_{N-1} (q_{N-1} = 1); extras [2a][3]
    HOp hF, hA;                                                                     // the same two operators in HODLR form (the default)
    bool use_hodlr = true;
    // work
    double *S6 = nullptr, *R3 = nullptr, *L6 = nullptr, *A3 = nullptr, *cur3 = nullptr, *G1 = nullptr, *GR = nullptr, *PR = nullptr, *H = nullptr,
           *HC = nullptr, *MN = nullptr, *d_stack = nullptr, *d_part = nullptr;
    std::vector<double> h_part;
    int k_gemm = -1, k_apply = -1, k_apply_adj = -1, k_point = -1;
    double op_bytes = 0.0, op_bytes_adj = 0.0;

    struct Phase { GemmDesc* d = nullptr; int n = 0, M = 0, N = 0, K = 0; XDesc* xd = nullptr; int xdir = 0; };   // xdir: +1 coefficients -> grid, -1 grid -> coefficients, 0 not an x phase
    bool use_xfft = false;                                                          // the x phases as FFTs (SMO_POIS_XFFT=0: the dense products)
    cplx* d_twx = nullptr;
    int k_xfft = -1;
    Phase F0x, F0z, F0d, Fz1, Fz, Fx, Fxf, Fzf, M1z, M1x, T0z, T0x, T0xf, T0zf, T1xf, T1zf, Ad, Az, Ax, Axf, Azf, Gd, Gz, Gx;

    int make_phase(Phase& p, int M, int N, int K, const std::vector<GemmDesc>& v) {
        p.n = (int)v.size(); p.M = M; p.N = N; p.K = K;
        std::vector<XDesc> xv;
        const int dir = x_phase_descs(v, XMats{A_Xi, A_XiD, A_XiN, A_XiN_DA, A_Xf, A_Xf_DA, A_XfN, A_XfNDa}, xv);
        p.xdir = dir;
        if (dir != 0) SMO_TRY(pool.upload(&p.xd, xv, stream));
        return pool.upload(&p.d, v, stream);
    }
    int run(const Phase& p, int count = -1, long long shift = 0) {
        const int n = count < 0 ? p.n : count;
        if (use_xfft && p.xdir != 0) {
            ScopedTimer t(timing, k_xfft, stream);
            launch_x(stream, p.xd, n, p.xdir, Nx, Nz, a, ada, k1, d_twx);
            return SMO_OK;
        }
        ScopedTimer t(timing, k_gemm, stream);
        launch_gemm(stream, p.d, n, p.M, p.N, p.K, shift);
        return SMO_OK;
    }
    // `modes`: apply the operators of n = 0..modes-1 only (the forward state is zero beyond the de-aliased modes)
    int apply(const double2* S, const double* in, const double* xin_v, double* out, double* xout_v, double* snap, int nin, int xin, int nout,
              int xout, int modes, int structure = 0) {
        const int rows = nout * Nz + xout, cols = nin * Nz + xin;
        ScopedTimer t(timing, structure == 2 ? k_apply_adj : k_apply, stream);
        hipLaunchKernelGGL(pois_apply, dim3((unsigned)(modes * ((rows + APPLY_ROWS - 1) / APPLY_ROWS))), dim3(256), cols * sizeof(double2), stream, S, in, xin_v, out, xout_v,
                           snap, a, modes, Nz, nin, xin, nout, xout, structure);
        return SMO_OK;
    }
    int apply_hodlr(const HOp& h, const double* in, const double* xin_v, double* out, double* xout_v, double* snap, int modes) {
        ScopedTimer t(timing, k_apply, stream);
        hop_launch(h, stream, modes, in, xin_v, out, xout_v, snap, a, Nz);
        return SMO_OK;
    }
    // the step's tau solve and its transpose
    int solve_fwd(int n) {
        return use_hodlr ? apply_hodlr(hF, R3, nullptr, S6, d_X3, snap(n + 1), ada) : apply(d_S, R3, nullptr, S6, d_X3, snap(n + 1), 3, 0, 3, 3, ada, 1);
    }
    int solve_adj() {
        if (!use_hodlr) return apply(d_SH, R3, d_X3, A3, nullptr, nullptr, 3, 3, 3, 0, a, 2);
        ScopedTimer t(timing, k_apply_adj, stream);
        hop_launch(hA, stream, a, R3, nullptr, A3, nullptr, nullptr, a, Nz, L6 + 3 * nC, d_q);
        return SMO_OK;
    }
    dim3 pw_grid(size_t n) const { return dim3((unsigned)std::min<size_t>((n + 255) / 256, NPART)); }
    double* snap(int n) { return d_stack + (size_t)n * 3 * nC; }

    int init() override {
        Nx = cfg.npts; Nz = cfg.npts2; s_cost = cfg.cost;
        if (cfg.batch != 1 || cfg.world != 1) { set_error("POIS: batch and world must be 1"); return SMO_ERR_ARG; }
        if (Nx < 12 || Nz < 12 || Nx > 768 || Nz > 384 || Nx % 6 != 0 || Nz % 3 != 0) {
            set_error("POIS: need npts (Nx) a multiple of 6 in [12, 768] and npts2 (Nz) a multiple of 3 in [12, 384], got %d x %d", Nx, Nz);
            return SMO_ERR_UNSUPPORTED;
        }
        if (s_cost != 0 && s_cost != 1) { set_error("POIS: cost must be 0 (time-averaged kinetic energy) or 1 (mix-norm)"); return SMO_ERR_ARG; }
        Lx = cfg.x1 - cfg.x0; k1 = 2.0 * M_PI / Lx; V = Lx * 2.0;
        Re = cfg.param; Ri = cfg.param2; Pe = cfg.param * (cfg.param3 > 0 ? cfg.param3 : 1.0); delta = cfg.param4 > 0 ? cfg.param4 : 0.25;
        a = (Nx - 1) / 2 + 1; ada = (2 * Nx / 3) / 2; Nz0 = 2 * Nz / 3;
        nC = (size_t)2 * a * Nz; nG = (size_t)Nx * Nz;
        n_comp = 1;
        vec_len = 2 * nG;
        snapshot_doubles = 3 * nC;
        stack_bytes = (size_t)(cfg.n_iters + 1) * 3 * nC * sizeof(double);
        SMO_TRY(base_init());
        const int N = Nz;
        // ---- z matrices -----------------------------------------------------------------------------------------------
        std::vector<double> Tf((size_t)N * N), Ti((size_t)N * N), Dz((size_t)N * N, 0.0), z(N), Wz(N);
        for (int i = 0; i < N; ++i) z[i] = -std::cos(M_PI * (i + 0.5) / N);
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) {
                const double c = std::cos(M_PI * j * (2 * i + 1) / (2.0 * N)), sg = (j & 1) ? -1.0 : 1.0;
                Tf[(size_t)j * N + i] = (2.0 / N) * c * (j == 0 ? 0.5 : 1.0) * sg;          // transform (POIS:44-51)
                Ti[(size_t)i * N + j] = sg * c;                                              // transformInverse (POIS:67-76)
            }
        Cheb ch(N);
        Dz = ch.D;
        Wz[0] = z[1] - z[0];
        for (int i = 1; i < N; ++i) Wz[i] = z[i] - z[i - 1];
        const double dx = Lx / Nx;
        for (int i = 0; i < N; ++i) Wz[i] *= dx;                                              // weightMatrixDisc (POIS:91-118)
        auto T = [&](const std::vector<double>& Mx) { std::vector<double> t((size_t)N * N); for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) t[(size_t)j * N + i] = Mx[(size_t)i * N + j]; return t; };
        auto mul = [&](const std::vector<double>& X, const std::vector<double>& Y) {
            std::vector<double> Zm((size_t)N * N, 0.0);
            for (int i = 0; i < N; ++i) for (int m = 0; m < N; ++m) { const double x = X[(size_t)i * N + m]; if (x != 0.0) for (int j = 0; j < N; ++j) Zm[(size_t)i * N + j] += x * Y[(size_t)m * N + j]; }
            return Zm;
        };
        auto mask_cols = [&](std::vector<double> Mx) { for (int i = 0; i < N; ++i) for (int j = Nz0; j < N; ++j) Mx[(size_t)i * N + j] = 0.0; return Mx; };
        auto mask_rows = [&](std::vector<double> Mx) { for (int i = Nz0; i < N; ++i) for (int j = 0; j < N; ++j) Mx[(size_t)i * N + j] = 0.0; return Mx; };
        const std::vector<double> TiDz = mul(Ti, Dz);
        SMO_TRY(pool.upload(&B_ZiT, T(Ti), stream));           // [j][z] = Ti[z][j]
        SMO_TRY(pool.upload(&B_DZiT, T(TiDz), stream));        // [j][z] = (Ti Dz)[z][j]
        SMO_TRY(pool.upload(&B_ZfT, T(Tf), stream));           // [z][j] = Tf[j][z]
        SMO_TRY(pool.upload(&B_ZfT_DA, mask_cols(T(Tf)), stream));
        SMO_TRY(pool.upload(&B_Zf, Tf, stream));               // [j][z]: transformAdjoint z part
        SMO_TRY(pool.upload(&B_Zf_DA, mask_rows(Tf), stream));
        SMO_TRY(pool.upload(&B_Zi, Ti, stream));               // [z][j]: transformInverseAdjoint z part
        SMO_TRY(pool.upload(&B_ZiDz, TiDz, stream));
        SMO_TRY(pool.upload(&B_DzT, T(Dz), stream));           // c @ Dz^T
        SMO_TRY(pool.upload(&B_Dz, Dz, stream));               // c @ Dz
        SMO_TRY(pool.upload(&d_Wz, Wz, stream));
        {
            std::vector<double> q(N, 0.0);                         // Pre q = e_{N-1} / 2:  q_j = 1 for j = N-1, N-3, ... >= 1; q_0 = 1/2 if it is hit
            for (int j = N - 1; j >= 1; j -= 2) q[j] = 1.0;
            if ((N - 1) % 2 == 0) q[0] = 0.5;
            SMO_TRY(pool.upload(&d_q, q, stream));
            std::vector<double> qz(N, 0.0);
            for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) qz[i] += Ti[(size_t)i * N + j] * q[j];
            SMO_TRY(pool.upload(&d_qz, qz, stream));
        }
        // ---- x matrices (Hermitian half spectrum n = 0..a-1, rows/cols 2n = Re, 2n+1 = Im) --------------------------------------
        std::vector<double> Xi((size_t)Nx * 2 * a), XiD(Xi.size()), XiN(Xi.size()), Xf((size_t)2 * a * Nx), XfN(Xf.size()), XfNDa(Xf.size());
        for (int x = 0; x < Nx; ++x)
            for (int n = 0; n < a; ++n) {
                const double k = k1 * n, ph = 2.0 * M_PI * (double)((long long)n * x % Nx) / Nx, c = std::cos(ph), sn = std::sin(ph), w = n == 0 ? 1.0 : 2.0;
                Xi[(size_t)x * 2 * a + 2 * n] = w * c;             Xi[(size_t)x * 2 * a + 2 * n + 1] = -w * sn;
                XiD[(size_t)x * 2 * a + 2 * n] = -w * k * sn;      XiD[(size_t)x * 2 * a + 2 * n + 1] = -w * k * c;
                XiN[(size_t)x * 2 * a + 2 * n] = w * c / Nx;       XiN[(size_t)x * 2 * a + 2 * n + 1] = -w * sn / Nx;
                Xf[(size_t)(2 * n) * Nx + x] = c / Nx;             Xf[(size_t)(2 * n + 1) * Nx + x] = -sn / Nx;
                XfN[(size_t)(2 * n) * Nx + x] = c;                 XfN[(size_t)(2 * n + 1) * Nx + x] = -sn;
                XfNDa[(size_t)(2 * n) * Nx + x] = -k * sn;         XfNDa[(size_t)(2 * n + 1) * Nx + x] = -k * c;
            }
        SMO_TRY(pool.upload(&A_Xi, Xi, stream)); SMO_TRY(pool.upload(&A_XiD, XiD, stream)); SMO_TRY(pool.upload(&A_XiN, XiN, stream));
        SMO_TRY(pool.upload(&A_Xf, Xf, stream)); SMO_TRY(pool.upload(&A_XfN, XfN, stream)); SMO_TRY(pool.upload(&A_XfNDa, XfNDa, stream));
        for (int x = 0; x < Nx; ++x) for (int c = 2 * ada; c < 2 * a; ++c) XiN[(size_t)x * 2 * a + c] = 0.0;     // de-aliasing mask in x
        for (int r = 2 * ada; r < 2 * a; ++r) for (int x = 0; x < Nx; ++x) Xf[(size_t)r * Nx + x] = 0.0;
        SMO_TRY(pool.upload(&A_XiN_DA, XiN, stream)); SMO_TRY(pool.upload(&A_Xf_DA, Xf, stream));
        {
            const char* e = getenv("SMO_POIS_XFFT");
            use_xfft = !(e && atoi(e) == 0) && Nz % 2 == 0 && with_xfft_length(Nx, [](auto) {});
            if (use_xfft) SMO_TRY(pool.upload(&d_twx, twiddles(Nx), stream));
            // SMO_POIS_XPROD=1: the pointwise products folded into the loads of the forward x transform (12 instead of 14 launches per step pair;
            // measured: no gain, profiles/r04_poiseuille_fusion.txt, where the one-kernel grid stage that was tried as well is recorded)
            const char* pr = getenv("SMO_POIS_XPROD");
            use_xprod = use_xfft && pr && atoi(pr) == 1;
        }
        // ---- base state rho = -erf(z/delta)/2 (n = 0 only), de-aliased (POIS:932-936) ------------------------------------------------
        {
            std::vector<double> r0(nC, 0.0), rz0(nC, 0.0);
            for (int j = 0; j < Nz0; ++j) {
                double s0 = 0.0, s1 = 0.0;
                for (int i = 0; i < N; ++i) {
                    s0 += Tf[(size_t)j * N + i] * (-0.5 * std::erf(z[i] / delta));
                    s1 += Tf[(size_t)j * N + i] * (-std::exp(-(z[i] / delta) * (z[i] / delta)) / (delta * std::sqrt(M_PI)));
                }
                r0[j] = s0; rz0[j] = s1;
            }
            SMO_TRY(pool.upload(&d_rho0, r0, stream)); SMO_TRY(pool.upload(&d_rz0, rz0, stream));
        }
        // ---- tau operators, one per wavenumber, built by host threads ---------------------------------------------------------------------
        {
            SMO_TRY(pois_apply_mode(&use_hodlr));
            const int n3 = 3 * N;
            const size_t sz = (size_t)(n3 + 3) * n3, szm = (size_t)2 * N * N;
            std::vector<cd> S, SH, SM((size_t)a * szm), SMH((size_t)a * szm);
            if (!use_hodlr) { S.resize((size_t)a * sz); SH.resize((size_t)a * sz); }
            const hodlr::Plan plan = hodlr::make_plan(n3);
            std::vector<hodlr::Factors> fac(use_hodlr ? a : 0);
            std::vector<std::vector<cd>> extras(use_hodlr ? a : 0);
            std::vector<int> rc(a, SMO_OK);
            std::vector<std::string> msg(a);
            const int nthr = std::max(1, std::min<int>(std::min(a, 32), (int)std::thread::hardware_concurrency()));   // one wavenumber per task; <= 32 host threads
            auto work = [&](int t) {
                std::vector<cd> red(sz), perm;
                for (int n = t; n < a; n += nthr) {
                    std::vector<cd> s, sm;
                    int r = build_solve_map(ch, n, k1 * n, 1.0 / cfg.dt, Re, Pe, Ri, s);
                    if (r == SMO_OK) r = build_mixnorm_map(ch, n, k1 * n, sm);
                    if (r != SMO_OK) { rc[n] = r; msg[n] = last_error(); continue; }
                    // keep the rows of u, v, rho and the last row of each derivative variable (see pois_rank1_add)
                    cd* dst = use_hodlr ? red.data() : &S[(size_t)n * sz];
                    std::copy(s.begin(), s.begin() + (size_t)n3 * n3, dst);
                    for (int f = 0; f < 3; ++f) std::copy(&s[((size_t)(3 + f) * N + N - 1) * n3], &s[((size_t)(3 + f) * N + N - 1) * n3] + n3, dst + (size_t)(n3 + f) * n3);
                    std::copy(sm.begin(), sm.end(), SM.begin() + (size_t)n * szm);
                    for (int i = 0; i < 2 * N; ++i) for (int j = 0; j < N; ++j) SMH[(size_t)n * szm + (size_t)j * 2 * N + i] = std::conj(sm[(size_t)i * N + j]);
                    if (!use_hodlr) {
                        for (int i = 0; i < n3 + 3; ++i) for (int j = 0; j < n3; ++j) SH[(size_t)n * sz + (size_t)j * (n3 + 3) + i] = std::conj(dst[(size_t)i * n3 + j]);
                        continue;
                    }
                    hodlr_factor_reduced(plan, dst, N, perm, fac[n], extras[n]);
                }
            };
            {
                std::vector<std::thread> th;
                for (int t = 0; t < nthr; ++t) th.emplace_back(work, t);
                for (auto& t : th) t.join();
            }
            for (int n = 0; n < a; ++n) if (rc[n] != SMO_OK) { set_error("%s (wavenumber %d)", msg[n].c_str(), n); return rc[n]; }
            auto up = [&](double2** p, const std::vector<cd>& h) -> int {
                SMO_TRY(pool.alloc(p, h.size()));
                SMO_HIP(hipMemcpyAsync(*p, h.data(), h.size() * sizeof(cd), hipMemcpyHostToDevice, stream));
                SMO_HIP(hipStreamSynchronize(stream));
                return SMO_OK;
            };
            SMO_TRY(up(&d_SMN, SM)); SMO_TRY(up(&d_SMNH, SMH));
            if (!use_hodlr) {
                SMO_TRY(up(&d_S, S)); SMO_TRY(up(&d_SH, SH));
                op_bytes = (double)ada * (double)sz * 16.0 * 7.0 / 9.0;                 // 2/9 of either operator are the structural zeros pois_apply skips
                op_bytes_adj = (double)a * (double)sz * 16.0 * 7.0 / 9.0;
            } else {
                SMO_TRY(hop_build(pool, stream, plan, fac, extras, ada, false, nthr, hF, &op_bytes));
                SMO_TRY(hop_build(pool, stream, plan, fac, extras, a, true, nthr, hA, &op_bytes_adj));
            }
        }
        // ---- work buffers -----------------------------------------------------------------------------------------------
        SMO_TRY(pool.alloc(&S6, 6 * nC)); SMO_TRY(pool.alloc(&R3, 3 * nC)); SMO_TRY(pool.alloc(&L6, 6 * nC)); SMO_TRY(pool.alloc(&A3, 3 * nC));
        SMO_TRY(pool.alloc(&cur3, 3 * nC)); SMO_TRY(pool.alloc(&G1, 9 * nC)); SMO_TRY(pool.alloc(&GR, 11 * nG)); SMO_TRY(pool.alloc(&PR, 10 * nG));
        SMO_TRY(pool.alloc(&H, 10 * nC)); SMO_TRY(pool.alloc(&HC, 10 * nC)); SMO_TRY(pool.alloc(&MN, 2 * nC)); SMO_TRY(pool.alloc(&d_X3, (size_t)2 * a * 3));
        SMO_TRY(pool.alloc(&d_stack, (size_t)(cfg.n_iters + 1) * 3 * nC));
        SMO_HIP(hipMemsetAsync(d_stack, 0, (size_t)(cfg.n_iters + 1) * 3 * nC * sizeof(double), stream));    // rows n >= ada stay zero
        SMO_TRY(pool.alloc(&d_part, (size_t)(cfg.n_iters + 2) * NPART));
        h_part.resize((size_t)(cfg.n_iters + 2) * NPART);
        SMO_HIP(hipMemsetAsync(d_part, 0, (size_t)(cfg.n_iters + 2) * NPART * sizeof(double), stream));
        // ---- GEMM phases -----------------------------------------------------------------------------------------------
        const int M2a = 2 * a;
        auto c_ = [&](double* base, int i) { return base + (size_t)i * nC; };
        auto g_ = [&](double* base, int i) { return base + (size_t)i * nG; };
        // forward set-up: X (copied to GR[0..1]) -> u, v (de-aliased), uz, vz
        SMO_TRY(make_phase(F0x, M2a, Nz, Nx, {{A_Xf_DA, g_(GR, 0), c_(H, 0)}, {A_Xf_DA, g_(GR, 1), c_(H, 1)}}));
        SMO_TRY(make_phase(F0z, M2a, Nz, Nz, {{c_(H, 0), B_ZfT_DA, c_(S6, 0)}, {c_(H, 1), B_ZfT_DA, c_(S6, 1)}}));
        SMO_TRY(make_phase(F0d, M2a, Nz, Nz, {{c_(S6, 0), B_DzT, c_(S6, 3)}, {c_(S6, 1), B_DzT, c_(S6, 4)}}));
        // adjoint step, first phase: R3 = lambda_{u,v,rho} + lambda_{uz,vz,rhoz} Dz (the sum is the product's epilogue)
        { std::vector<GemmDesc> w; for (int f = 0; f < 3; ++f) w.push_back({c_(L6, 3 + f), B_Dz, c_(R3, f), c_(L6, f), 1.0});
          SMO_TRY(make_phase(Ad, M2a, Nz, Nz, w)); }
        // forward step.  Lines in z of [u, v, rho, uz, vz, rhoz]: the derivative variables of the tau system are  uz = u Dz^T + (uz)_{N-1} q  (see
        // pois_rank1_add), so their lines come straight from u, v, rho and the three scalars per wavenumber the operator apply returns:
        //   uz Ti^T = u (Ti Dz)^T + x (x) (Ti q)      — Fz1; the first step (Fz) takes the derivative coefficients of the initial state as given
        { std::vector<GemmDesc> v, v1;
          for (int f = 0; f < 6; ++f) v.push_back({c_(S6, f), B_ZiT, c_(G1, f)});
          for (int f = 0; f < 3; ++f) v1.push_back({c_(S6, f), B_ZiT, c_(G1, f)});
          for (int f = 0; f < 3; ++f) v1.push_back({c_(S6, f), B_DZiT, c_(G1, 3 + f), nullptr, 0.0, d_X3 + f, d_qz});
          SMO_TRY(make_phase(Fz, M2a, Nz, Nz, v)); SMO_TRY(make_phase(Fz1, M2a, Nz, Nz, v1)); }
        SMO_TRY(make_phase(Fx, Nx, Nz, M2a, {{A_Xi, c_(G1, 0), g_(GR, 0)}, {A_XiD, c_(G1, 0), g_(GR, 1)}, {A_Xi, c_(G1, 3), g_(GR, 2)},
                                             {A_Xi, c_(G1, 1), g_(GR, 3)}, {A_XiD, c_(G1, 1), g_(GR, 4)}, {A_Xi, c_(G1, 4), g_(GR, 5)},
                                             {A_XiD, c_(G1, 2), g_(GR, 6)}, {A_Xi, c_(G1, 5), g_(GR, 7)}}));
        // right-hand side of the step: R3 = state / dt + transformed products (the sum is the product's epilogue)
        { std::vector<GemmDesc> v, w; for (int f = 0; f < 3; ++f) { v.push_back({A_Xf_DA, g_(PR, f), c_(H, f)}); w.push_back({c_(H, f), B_ZfT_DA, c_(R3, f), c_(S6, f), 1.0 / cfg.dt}); }
          SMO_TRY(make_phase(Fxf, M2a, Nz, Nx, v)); SMO_TRY(make_phase(Fzf, M2a, Nz, Nz, w)); }
        // mix-norm fields psi (cur3[2]) and psiz (cur3[1]) -> grids gx = dx psi, gz = psiz
        SMO_TRY(make_phase(M1z, M2a, Nz, Nz, {{c_(cur3, 2), B_ZiT, c_(G1, 0)}, {c_(cur3, 1), B_ZiT, c_(G1, 1)}}));
        SMO_TRY(make_phase(M1x, Nx, Nz, M2a, {{A_XiD, c_(G1, 0), g_(GR, 0)}, {A_Xi, c_(G1, 1), g_(GR, 1)}}));
        // adjoint terminal conditions
        SMO_TRY(make_phase(T0z, M2a, Nz, Nz, {{c_(cur3, 0), B_ZiT, c_(G1, 0)}, {c_(cur3, 1), B_ZiT, c_(G1, 1)}}));
        SMO_TRY(make_phase(T0x, Nx, Nz, M2a, {{A_Xi, c_(G1, 0), g_(GR, 0)}, {A_Xi, c_(G1, 1), g_(GR, 1)}}));
        SMO_TRY(make_phase(T0xf, M2a, Nz, Nx, {{A_XfN, g_(PR, 0), c_(H, 0)}, {A_XfN, g_(PR, 1), c_(H, 1)}}));
        SMO_TRY(make_phase(T0zf, M2a, Nz, Nz, {{c_(H, 0), B_Zi, c_(L6, 0)}, {c_(H, 1), B_Zi, c_(L6, 1)}}));
        SMO_TRY(make_phase(T1xf, M2a, Nz, Nx, {{A_XfNDa, g_(PR, 0), c_(H, 0)}, {A_XfN, g_(PR, 1), c_(H, 1)}}));
        SMO_TRY(make_phase(T1zf, M2a, Nz, Nz, {{c_(H, 0), B_Zi, c_(HC, 0)}, {c_(H, 1), B_ZiDz, c_(HC, 1)}}));
        // adjoint step
        { std::vector<GemmDesc> v;
          for (int f = 0; f < 3; ++f) v.push_back({c_(A3, f), B_Zf_DA, c_(G1, f)});
          for (int f = 0; f < 3; ++f) v.push_back({c_(cur3, f), B_ZiT, c_(G1, 3 + f), nullptr, 0.0, nullptr, nullptr, 1});     // dyn: the launch points these at the
          for (int f = 0; f < 3; ++f) v.push_back({c_(cur3, f), B_DZiT, c_(G1, 6 + f), nullptr, 0.0, nullptr, nullptr, 1});    // step's snapshot (run(Az, -1, shift))
          SMO_TRY(make_phase(Az, M2a, Nz, Nz, v)); }
        SMO_TRY(make_phase(Ax, Nx, Nz, M2a, {{A_XiN_DA, c_(G1, 0), g_(GR, 0)}, {A_XiN_DA, c_(G1, 1), g_(GR, 1)}, {A_XiN_DA, c_(G1, 2), g_(GR, 2)},
                                             {A_Xi, c_(G1, 3), g_(GR, 3)}, {A_Xi, c_(G1, 4), g_(GR, 4)}, {A_XiD, c_(G1, 3), g_(GR, 5)},
                                             {A_XiD, c_(G1, 4), g_(GR, 6)}, {A_XiD, c_(G1, 5), g_(GR, 7)}, {A_Xi, c_(G1, 6), g_(GR, 8)},
                                             {A_Xi, c_(G1, 7), g_(GR, 9)}, {A_Xi, c_(G1, 8), g_(GR, 10)}}));
        { std::vector<GemmDesc> v, w;
          for (int i = 0; i < 10; ++i) { v.push_back({(i == 1 || i == 4 || i == 6) ? A_XfNDa : A_XfN, g_(PR, i), c_(H, i)}); w.push_back({c_(H, i), B_Zi, c_(HC, i)}); }
          SMO_TRY(make_phase(Axf, M2a, Nz, Nx, v)); SMO_TRY(make_phase(Azf, M2a, Nz, Nz, w)); }
        // gradient output
        SMO_TRY(make_phase(Gd, M2a, Nz, Nz, {{c_(L6, 3), B_Dz, c_(HC, 0)}, {c_(L6, 4), B_Dz, c_(HC, 1)}}));
        SMO_TRY(make_phase(Gz, M2a, Nz, Nz, {{c_(L6, 0), B_Zf, c_(G1, 0)}, {c_(L6, 1), B_Zf, c_(G1, 1)}}));
        SMO_TRY(make_phase(Gx, Nx, Nz, M2a, {{A_XiN, c_(G1, 0), g_(GR, 0)}, {A_XiN, c_(G1, 1), g_(GR, 1)}}));
        k_gemm = timing.add_class("pois_gemm (transforms, MFMA f64)", 0.0);
        // bytes = the operators one launch streams: the forward apply runs over the de-aliased wavenumbers, the transposed one over all of them
        k_apply = timing.add_class(use_hodlr ? "pois_apply_hodlr (tau operator, HODLR form)" : "pois_apply (tau operator, batched complex GEMV)", op_bytes, op_bytes);
        k_apply_adj = timing.add_class(use_hodlr ? "pois_apply_hodlr (transposed tau operator)" : "pois_apply (transposed tau operator)", op_bytes_adj, op_bytes_adj);
        k_point = timing.add_class("pois pointwise", 0.0);
        k_xfft = timing.add_class("pois_x (x transforms, LDS FFT)", 0.0);
        return SMO_OK;
    }

    int sum_partials(int row0, int nrows, double* out) {               // out[r] = sum of row r's NPART partials
        SMO_HIP(hipMemcpyAsync(h_part.data(), d_part + (size_t)row0 * NPART, (size_t)nrows * NPART * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        for (int r = 0; r < nrows; ++r) { double s = 0.0; for (int i = 0; i < NPART; ++i) s += h_part[(size_t)r * NPART + i]; out[r] = s; }
        return SMO_OK;
    }
    int state_grids(bool initial) { SMO_TRY(run(initial ? Fz : Fz1)); return run(Fx); }
    // the products folded into the loads of the grid -> coefficient transform (pois_x_prod_to_coeff): default; SMO_POIS_XPROD=0 keeps pois_nl /
    // pois_adj_products + pois_x_to_coeff
    bool use_xprod = false;
    template <int MODE> int prod_to_coeff(const Phase& out, int nout, double* part, double fscale) {
        ScopedTimer t(timing, k_xfft, stream);
        const dim3 grid((unsigned)((Nz + X_ZT - 1) / X_ZT), (unsigned)nout);
        bool ok = with_xfft_length(Nx, [&](auto l) {
            constexpr int LL = decltype(l)::value;
            hipLaunchKernelGGL((pois_x_prod_to_coeff<LL, MODE>), grid, dim3(X_NT), 0, stream, out.xd, GR, nG, d_twx, Nz, a, ada, k1, d_Wz, part, fscale);
        });
        if (!ok) { set_error("POIS: no x FFT for Nx = %d", Nx); return SMO_ERR_STATE; }
        return SMO_OK;
    }
    int nl_and_energy(int step) {
        ScopedTimer t(timing, k_point, stream);
        hipLaunchKernelGGL(pois_nl, dim3(NPART), dim3(256), 0, stream, GR, PR, d_Wz, d_part + (size_t)step * NPART, nG, Nz);
        return SMO_OK;
    }

    // the two time loops: nothing but kernel launches on fixed buffers (217 us of kernels in 221 us per step pair: the launches run back to back)
    int fwd_loop() {
        const int N = cfg.n_iters;
        for (int n = 0; n < N; ++n) {
            if (use_xprod) {
                SMO_TRY(state_grids(n == 0));
                SMO_TRY(prod_to_coeff<0>(Fxf, 3, d_part + (size_t)n * NPART, 0.0));
            } else {
                SMO_TRY(state_grids(n == 0));
                SMO_TRY(nl_and_energy(n));
                SMO_TRY(run(Fxf));
            }
            SMO_TRY(run(Fzf));                                               // R3 = state / dt + transformed products
            SMO_TRY(solve_fwd(n));                                           // u, v, rho and the last coefficient of uz, vz, rhoz (d_X3)
        }
        return SMO_OK;
    }
    int adj_loop() {
        const int N = cfg.n_iters;
        const bool forcing = s_cost == 0;
        for (int idx = N - 1; idx >= 0; --idx) {
            // S^H lambda with the reduced operator: lambda_{u,v,rho} + lambda_{uz,vz,rhoz} Dz, and the three scalars q . lambda_z
            SMO_TRY(run(Ad));
            if (!use_hodlr) {                                                // (the HODLR apply forms the three scalars while it stages its input)
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_rank1_dot, dim3((unsigned)((3LL * 2 * a + 3) / 4)), dim3(256), 0, stream, d_X3, L6 + 3 * nC, d_q, 2 * a, Nz);
            }
            SMO_TRY(solve_adj());
            SMO_TRY(run(Az, -1, ((long long)(intptr_t)snap(idx) - (long long)(intptr_t)cur3) / (long long)sizeof(double)));      // the forward state's lines straight from snapshot idx
            const int np = forcing ? 10 : 8;
            if (use_xprod) {
                SMO_TRY(run(Ax));
                SMO_TRY(prod_to_coeff<1>(Axf, np, nullptr, -cfg.dt / V));
            } else {
                SMO_TRY(run(Ax));
                {
                    ScopedTimer t(timing, k_point, stream);
                    hipLaunchKernelGGL(pois_adj_products, pw_grid(nG), dim3(256), 0, stream, GR, PR, d_Wz, -cfg.dt / V, forcing ? 1 : 0, nG, Nz);
                }
                SMO_TRY(run(Axf, np));
            }
            SMO_TRY(run(Azf, np));
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_adj_combine, pw_grid(nC), dim3(256), 0, stream, L6, A3, HC, 1.0 / cfg.dt, forcing ? 1 : 0, nC);
            }
        }
        return SMO_OK;
    }

    int forward_dev(const double* const* X, double* J) override {
        have_forward = false;
        const int N = cfg.n_iters;
        SMO_HIP(hipMemcpyAsync(GR, X[0], 2 * nG * sizeof(double), hipMemcpyDeviceToDevice, stream));
        SMO_HIP(hipMemsetAsync(S6, 0, 6 * nC * sizeof(double), stream));
        SMO_HIP(hipMemsetAsync(d_X3, 0, (size_t)2 * a * 3 * sizeof(double), stream));      // rows of the modes n >= ada stay zero
        if (use_xprod) SMO_HIP(hipMemsetAsync(d_part, 0, (size_t)N * NPART * sizeof(double), stream));      // its column tiles write Nz / 8 of a row's NPART partials
        SMO_TRY(run(F0x)); SMO_TRY(run(F0z)); SMO_TRY(run(F0d));
        SMO_HIP(hipMemcpyAsync(S6 + 2 * nC, d_rho0, nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
        SMO_HIP(hipMemcpyAsync(S6 + 5 * nC, d_rz0, nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
        SMO_HIP(hipMemcpyAsync(snap(0), S6, 3 * nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
        SMO_TRY(fwd_loop());
        double cost = 0.0;
        if (s_cost == 1) {
            // mix-norm (POIS:1053-1124): (psi, psiz) = S^MN rho_N; snapshot N holds (dx psi, psiz, psi); cost = <grad psi, grad psi> / 2
            SMO_TRY(apply(d_SMN, S6 + 2 * nC, nullptr, MN, nullptr, nullptr, 1, 0, 2, 0, a));
            SMO_HIP(hipMemcpyAsync(cur3 + 2 * nC, MN, nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
            SMO_HIP(hipMemcpyAsync(cur3 + nC, MN + nC, nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_ik, pw_grid((size_t)a * Nz), dim3(256), 0, stream, cur3, MN, k1, a, Nz);
            }
            SMO_HIP(hipMemcpyAsync(snap(N), cur3, 3 * nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
            SMO_TRY(run(M1z)); SMO_TRY(run(M1x));
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_wsq, dim3(NPART), dim3(256), 0, stream, GR, GR + nG, d_Wz, d_part + (size_t)(N + 1) * NPART, (double*)nullptr,
                                   (double*)nullptr, 0.0, nG, Nz);
            }
            double e = 0.0;
            SMO_TRY(sum_partials(N + 1, 1, &e));
            cost = 0.5 * e / V;
        } else {
            SMO_TRY(state_grids(N == 0));
            SMO_TRY(nl_and_energy(N));
            std::vector<double> e(N + 1);
            SMO_TRY(sum_partials(0, N + 1, e.data()));
            double ke = 0.0;
            for (int n = 0; n <= N; ++n) ke += cfg.dt * e[n] / V;
            cost = -0.5 * ke;
        }
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipStreamSynchronize(stream));
        *J = cost;
        have_forward = true;
        return SMO_OK;
    }

    int adjoint_dev(const double* const*, int adjoint_type, double* const* grad) override {
        if (adjoint_type != SMO_ADJ_DISCRETE) { set_error("POIS: only the Discrete formulation is built"); return SMO_ERR_UNSUPPORTED; }
        const int N = cfg.n_iters;
        SMO_HIP(hipMemsetAsync(L6, 0, 6 * nC * sizeof(double), stream));
        SMO_HIP(hipMemcpyAsync(cur3, snap(N), 3 * nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
        if (s_cost == 1) {
            SMO_TRY(run(M1z)); SMO_TRY(run(M1x));
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_wsq, dim3(NPART), dim3(256), 0, stream, GR, GR + nG, d_Wz, d_part + (size_t)(N + 1) * NPART, PR, PR + nG, 1.0 / V, nG, Nz);
            }
            SMO_TRY(run(T1xf)); SMO_TRY(run(T1zf));
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_axpy, pw_grid(nC), dim3(256), 0, stream, MN, HC, 1.0, HC + nC, nC);
            }
            SMO_HIP(hipMemsetAsync(MN + nC, 0, nC * sizeof(double), stream));
            SMO_TRY(apply(d_SMNH, MN, nullptr, L6 + 2 * nC, nullptr, nullptr, 2, 0, 1, 0, a));
        } else {
            SMO_TRY(run(T0z)); SMO_TRY(run(T0x));
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_wsq, dim3(NPART), dim3(256), 0, stream, GR, GR + nG, d_Wz, d_part + (size_t)(N + 1) * NPART, PR, PR + nG, -cfg.dt / V, nG, Nz);
            }
            SMO_TRY(run(T0xf)); SMO_TRY(run(T0zf));
        }
        SMO_TRY(adj_loop());
        SMO_TRY(run(Gd));
        {
            ScopedTimer t(timing, k_point, stream);
            hipLaunchKernelGGL(pois_axpy, pw_grid(2 * nC), dim3(256), 0, stream, L6, L6, 1.0, HC, 2 * nC);
        }
        SMO_TRY(run(Gz)); SMO_TRY(run(Gx));
        {
            ScopedTimer t(timing, k_point, stream);
            hipLaunchKernelGGL(pois_grad_out, pw_grid(2 * nG), dim3(256), 0, stream, grad[0], GR, d_Wz, V, 2 * nG, Nz);
        }
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }

    int inner_dev(const double* x, const double* y, double* out) override {
        hipLaunchKernelGGL(pois_dot, dim3(NPART), dim3(256), 0, stream, x, y, d_Wz, d_part + (size_t)(cfg.n_iters + 1) * NPART, 2 * nG, Nz);
        SMO_HIP(hipGetLastError());
        double s = 0.0;
        SMO_TRY(sum_partials(cfg.n_iters + 1, 1, &s));
        *out = s / V;
        return SMO_OK;
    }

    // internal [2a][Nz] rows (Re, Im) <-> the reference's complex128 [a][Nz]
    void to_complex(const std::vector<double>& rows, double* out, int nf) const {
        for (int f = 0; f < nf; ++f)
            for (int n = 0; n < a; ++n)
                for (int j = 0; j < Nz; ++j) {
                    out[(((size_t)f * a + n) * Nz + j) * 2] = rows[(size_t)f * nC + ((size_t)2 * n) * Nz + j];
                    out[(((size_t)f * a + n) * Nz + j) * 2 + 1] = rows[(size_t)f * nC + ((size_t)2 * n + 1) * Nz + j];
                }
    }
    int snapshot_read(int, int index, double* out) override {
        std::vector<double> h(3 * nC);
        SMO_HIP(hipMemcpyAsync(h.data(), snap(index), 3 * nC * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        to_complex(h, out, 3);
        return SMO_OK;
    }

    // the reference's four transforms on ONE real field / Hermitian coefficient array (POIS:44-89), host buffers:
    //   0 transform: grid [Nx][Nz] -> complex [a][Nz];  1 transformInverse: complex -> grid;
    //   2 transformAdjoint: complex -> grid;            3 transformInverseAdjoint: grid -> complex
    int transform_host(int which, const double* in, double* out) override {
        if (which < 0 || which > 3) { set_error("smo_transform: which = %d", which); return SMO_ERR_ARG; }
        const bool to_coeff = (which == 0 || which == 3);
        GemmDesc h1, h2;
        if (to_coeff) {
            SMO_HIP(hipMemcpyAsync(GR, in, nG * sizeof(double), hipMemcpyHostToDevice, stream));
            h1 = {which == 0 ? A_Xf : A_XfN, GR, H};
            h2 = {H, which == 0 ? B_ZfT : B_Zi, HC};
        } else {
            std::vector<double> rows(nC);
            for (int n = 0; n < a; ++n)
                for (int j = 0; j < Nz; ++j) { rows[((size_t)2 * n) * Nz + j] = in[((size_t)n * Nz + j) * 2]; rows[((size_t)2 * n + 1) * Nz + j] = in[((size_t)n * Nz + j) * 2 + 1]; }
            SMO_HIP(hipMemcpyAsync(H, rows.data(), nC * sizeof(double), hipMemcpyHostToDevice, stream));
            SMO_HIP(hipStreamSynchronize(stream));
            h1 = {H, which == 1 ? B_ZiT : B_Zf, G1};
            h2 = {which == 1 ? A_Xi : A_XiN, G1, GR};
        }
        GemmDesc* d = nullptr;
        SMO_HIP(hipMalloc(&d, 2 * sizeof(GemmDesc)));
        const GemmDesc hd[2] = {h1, h2};
        hipError_t e = hipMemcpyAsync(d, hd, sizeof(hd), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) {
            const int M2a = 2 * a;
            if (to_coeff) {
                launch_gemm(stream, d, 1, M2a, Nz, Nx);
                launch_gemm(stream, d + 1, 1, M2a, Nz, Nz);
            } else {
                launch_gemm(stream, d, 1, M2a, Nz, Nz);
                launch_gemm(stream, d + 1, 1, Nx, Nz, M2a);
            }
            e = hipStreamSynchronize(stream);
        }
        (void)hipFree(d);
        if (e != hipSuccess) { set_error("smo_transform: %s", hipGetErrorString(e)); return SMO_ERR_HIP; }
        if (to_coeff) {
            std::vector<double> rows(nC);
            SMO_HIP(hipMemcpyAsync(rows.data(), HC, nC * sizeof(double), hipMemcpyDeviceToHost, stream));
            SMO_HIP(hipStreamSynchronize(stream));
            to_complex(rows, out, 1);
        } else {
            SMO_HIP(hipMemcpyAsync(out, GR, nG * sizeof(double), hipMemcpyDeviceToHost, stream));
            SMO_HIP(hipStreamSynchronize(stream));
        }
        return SMO_OK;
    }
};


// ---------------------------------------------------------------------------------------------------------
// "Continuous" formulation (the script's default Adjoint_type, POIS:1728): the Dedalus IVPs of FWD_Solve_Cnts (:614-775) and
// ADJ_Solve_Cnts (:1161-1318) with SBDF1 on Nx x Nz modes, products on the 3/2 grid, exact-integration inner product (:241-280).
// Same kernels as above; the z matrices are rectangular (Nz modes <-> 3Nz/2 grid points) and the adjoint has its own tau operator.
// ---------------------------------------------------------------------------------------------------------
// right-hand side of the adjoint IVP on the grid (POIS:1217-1226):
//   gr = [uf, wf, ufx, wfx, bfx, ufz, wfz, bfz, ua, wa, ba, uax, wax, bax, uza, wza, bza]  ->  pr = [Fu, Fw, Fb]
__global__ __launch_bounds__(256) void pois_cnts_adj_rhs(const double* __restrict__ gr, double* __restrict__ pr, int kinetic, size_t nG) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nG; i += (size_t)gridDim.x * 256) {
        const double uf = gr[i], wf = gr[nG + i], ufx = gr[2 * nG + i], wfx = gr[3 * nG + i], bfx = gr[4 * nG + i], ufz = gr[5 * nG + i],
                     wfz = gr[6 * nG + i], bfz = gr[7 * nG + i], ua = gr[8 * nG + i], wa = gr[9 * nG + i], ba = gr[10 * nG + i],
                     uax = gr[11 * nG + i], wax = gr[12 * nG + i], bax = gr[13 * nG + i], uza = gr[14 * nG + i], wza = gr[15 * nG + i],
                     bza = gr[16 * nG + i];
        pr[i] = -(ua * ufx + wa * wfx) + (uf * uax + wf * uza) - ba * bfx - (kinetic ? uf : 0.0);
        pr[nG + i] = -(ua * ufz + wa * wfz) + (uf * wax + wf * wza) - ba * bfz - (kinetic ? wf : 0.0);
        pr[2 * nG + i] = uf * bax + wf * bza;
    }
}

class PoisCnts : public Context {
public:
    explicit PoisCnts(const smo_config& c) { cfg = c; }
    int Nxm = 0, Nz = 0, a = 0, Gx = 0, Gz = 0, s_cost = 0;
    double Lx = 0, k1 = 0, V = 0;
    size_t nC = 0, nL = 0, nG = 0;                       // coefficient field [2a][Nz], z-transformed lines [2a][Gz], grid [Gx][Gz]
    double *B_ZiT = nullptr, *B_DZiT = nullptr, *B_ZfT = nullptr, *B_DzT = nullptr, *A_Xi = nullptr, *A_XiD = nullptr, *A_Xf = nullptr;
    double *d_Wq = nullptr, *d_b0 = nullptr, *d_bz0 = nullptr, *d_q = nullptr, *d_X3 = nullptr;
    double2 *d_S = nullptr, *d_SA = nullptr, *d_SMN = nullptr;
    HOp hS, hSA;                                           // the forward and the adjoint IVP's operators in HODLR form (the default)
    bool use_hodlr = true;
    double op_bytes = 0.0;
    double *S6 = nullptr, *A6 = nullptr, *R3 = nullptr, *cur3 = nullptr, *G1 = nullptr, *GR = nullptr, *PR = nullptr, *H = nullptr, *HC = nullptr,
           *MN = nullptr, *d_stack = nullptr, *d_part = nullptr;
    std::vector<double> h_part;
    int k_gemm = -1, k_apply = -1, k_point = -1;
    struct Phase { GemmDesc* d = nullptr; int n = 0, M = 0, N = 0, K = 0; XDesc* xd = nullptr; int xdir = 0; };
    Phase F0x, F0z, Fz, Fx, Fxf, Fzf, F1d, A1d, M1z, M1x, Az, Ax, Gz2, Gx2;
    bool use_xfft = false;                                 // the x phases (Nx modes <-> 3 Nx / 2 grid points) as FFTs (SMO_POIS_XFFT=0: the dense products)
    cplx* d_twx = nullptr;
    int k_xfft = -1;

    int make_phase(Phase& p, int M, int N, int K, const std::vector<GemmDesc>& v) {
        p.n = (int)v.size(); p.M = M; p.N = N; p.K = K;
        std::vector<XDesc> xv;
        XMats m; m.Xi = A_Xi; m.XiD = A_XiD; m.Xf = A_Xf;
        p.xdir = x_phase_descs(v, m, xv);
        if (p.xdir != 0) SMO_TRY(pool.upload(&p.xd, xv, stream));
        return pool.upload(&p.d, v, stream);
    }
    int run(const Phase& p) {
        if (use_xfft && p.xdir != 0) {
            ScopedTimer t(timing, k_xfft, stream);
            launch_x(stream, p.xd, p.n, p.xdir, Gx, Gz, a, a, k1, d_twx);
            return SMO_OK;
        }
        ScopedTimer t(timing, k_gemm, stream);
        launch_gemm(stream, p.d, p.n, p.M, p.N, p.K);
        return SMO_OK;
    }
    int apply(const double2* S, const double* in, double* out, double* xout, int nin, int nout, int xo) {
        const int rows = nout * Nz + xo, cols = nin * Nz;
        ScopedTimer t(timing, k_apply, stream);
        hipLaunchKernelGGL(pois_apply, dim3((unsigned)(a * ((rows + APPLY_ROWS - 1) / APPLY_ROWS))), dim3(256), cols * sizeof(double2), stream, S, in, (const double*)nullptr, out,
                           xout, (double*)nullptr, a, a, Nz, nin, 0, nout, xo, 0);
        return SMO_OK;
    }
    dim3 pw_grid(size_t n) const { return dim3((unsigned)std::min<size_t>((n + 255) / 256, NPART)); }
    double* snap(int n) { return d_stack + (size_t)n * 3 * nC; }

    int init() override {
        Nxm = cfg.npts; Nz = cfg.npts2; s_cost = cfg.cost - 2;
        if (cfg.batch != 1 || cfg.world != 1) { set_error("POIS: batch and world must be 1"); return SMO_ERR_ARG; }
        if (Nxm < 8 || Nz < 8 || Nxm > 512 || Nz > 256 || Nxm % 4 != 0 || Nz % 2 != 0) {
            set_error("POIS (Continuous): need npts (Nx modes) a multiple of 4 in [8, 512] and npts2 (Nz modes) even in [8, 256], got %d x %d", Nxm, Nz);
            return SMO_ERR_UNSUPPORTED;
        }
        Lx = cfg.x1 - cfg.x0; k1 = 2.0 * M_PI / Lx; V = Lx * 2.0;
        const double Re = cfg.param, Ri = cfg.param2, Pe = cfg.param * (cfg.param3 > 0 ? cfg.param3 : 1.0), delta = cfg.param4 > 0 ? cfg.param4 : 0.25;
        a = (Nxm - 1) / 2 + 1; Gx = 3 * Nxm / 2; Gz = 3 * Nz / 2;
        nC = (size_t)2 * a * Nz; nL = (size_t)2 * a * Gz; nG = (size_t)Gx * Gz;
        n_comp = 1;
        vec_len = 2 * nG;
        snapshot_doubles = 3 * nC;
        stack_bytes = (size_t)(cfg.n_iters + 1) * 3 * nC * sizeof(double);
        SMO_TRY(base_init());
        const int N = Nz;
        Cheb ch(N);
        // rectangular z matrices: Tf (N x Gz) grid line -> first N T coefficients, Ti (Gz x N) back
        std::vector<double> Tf((size_t)N * Gz), Ti((size_t)Gz * N), z(Gz), Wq(Gz, 0.0);
        for (int i = 0; i < Gz; ++i) z[i] = -std::cos(M_PI * (i + 0.5) / Gz);
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < Gz; ++i) {
                const double c = std::cos(M_PI * j * (2 * i + 1) / (2.0 * Gz)), sg = (j & 1) ? -1.0 : 1.0;
                Tf[(size_t)j * Gz + i] = (2.0 / Gz) * c * (j == 0 ? 0.5 : 1.0) * sg;
                Ti[(size_t)i * N + j] = sg * c;
            }
        for (int i = 0; i < Gz; ++i) {                                  // exact integral of the truncated series, as a quadrature on the grid
            double w = 0.0;
            for (int j = 0; j < N; j += 2) w += ch.integ[j] * Tf[(size_t)j * Gz + i];
            Wq[i] = w * (Lx / Gx);
        }
        std::vector<double> ZiT((size_t)N * Gz), DZiT((size_t)N * Gz, 0.0), ZfT((size_t)Gz * N), DzT((size_t)N * N);
        for (int j = 0; j < N; ++j) for (int i = 0; i < Gz; ++i) { ZiT[(size_t)j * Gz + i] = Ti[(size_t)i * N + j]; ZfT[(size_t)i * N + j] = Tf[(size_t)j * Gz + i]; }
        for (int j = 0; j < N; ++j) for (int m = 0; m < N; ++m) {
            DzT[(size_t)m * N + j] = ch.D[(size_t)j * N + m];
            const double d = ch.D[(size_t)m * N + j];                   // (Ti Dz)[z][j] = sum_m Ti[z][m] Dz[m][j]
            if (d != 0.0) for (int i = 0; i < Gz; ++i) DZiT[(size_t)j * Gz + i] += Ti[(size_t)i * N + m] * d;
        }
        SMO_TRY(pool.upload(&B_ZiT, ZiT, stream)); SMO_TRY(pool.upload(&B_DZiT, DZiT, stream)); SMO_TRY(pool.upload(&B_ZfT, ZfT, stream));
        SMO_TRY(pool.upload(&B_DzT, DzT, stream)); SMO_TRY(pool.upload(&d_Wq, Wq, stream));
        {
            std::vector<double> q(N, 0.0);
            for (int j = N - 1; j >= 1; j -= 2) q[j] = 1.0;
            if ((N - 1) % 2 == 0) q[0] = 0.5;
            SMO_TRY(pool.upload(&d_q, q, stream));
        }
        std::vector<double> Xi((size_t)Gx * 2 * a), XiD(Xi.size()), Xf((size_t)2 * a * Gx);
        for (int x = 0; x < Gx; ++x)
            for (int n = 0; n < a; ++n) {
                const double k = k1 * n, ph = 2.0 * M_PI * (double)((long long)n * x % Gx) / Gx, c = std::cos(ph), sn = std::sin(ph), w = n == 0 ? 1.0 : 2.0;
                Xi[(size_t)x * 2 * a + 2 * n] = w * c;          Xi[(size_t)x * 2 * a + 2 * n + 1] = -w * sn;
                XiD[(size_t)x * 2 * a + 2 * n] = -w * k * sn;   XiD[(size_t)x * 2 * a + 2 * n + 1] = -w * k * c;
                Xf[(size_t)(2 * n) * Gx + x] = c / Gx;          Xf[(size_t)(2 * n + 1) * Gx + x] = -sn / Gx;
            }
        SMO_TRY(pool.upload(&A_Xi, Xi, stream)); SMO_TRY(pool.upload(&A_XiD, XiD, stream)); SMO_TRY(pool.upload(&A_Xf, Xf, stream));
        {
            const char* e = getenv("SMO_POIS_XFFT");
            use_xfft = !(e && atoi(e) == 0) && Gz % 2 == 0 && with_xfft_length(Gx, [](auto) {});
            if (use_xfft) SMO_TRY(pool.upload(&d_twx, twiddles(Gx), stream));
        }
        {
            std::vector<double> b0(nC, 0.0), bz0(nC, 0.0);
            for (int j = 0; j < N; ++j) {
                double s0 = 0.0, s1 = 0.0;
                for (int i = 0; i < Gz; ++i) {
                    s0 += Tf[(size_t)j * Gz + i] * (-0.5 * std::erf(z[i] / delta));
                    s1 += Tf[(size_t)j * Gz + i] * (-std::exp(-(z[i] / delta) * (z[i] / delta)) / (delta * std::sqrt(M_PI)));
                }
                b0[j] = s0; bz0[j] = s1;
            }
            SMO_TRY(pool.upload(&d_b0, b0, stream)); SMO_TRY(pool.upload(&d_bz0, bz0, stream));
        }
        {
            SMO_TRY(pois_apply_mode(&use_hodlr));
            const size_t sz = (size_t)(3 * N + 3) * 3 * N, szm = (size_t)2 * N * N;
            std::vector<cd> S, SA, SM((size_t)a * szm);
            if (!use_hodlr) { S.resize((size_t)a * sz); SA.resize((size_t)a * sz); }
            const hodlr::Plan plan = hodlr::make_plan(3 * N);
            std::vector<hodlr::Factors> fS(use_hodlr ? a : 0), fA(use_hodlr ? a : 0);
            std::vector<std::vector<cd>> eS(use_hodlr ? a : 0), eA(use_hodlr ? a : 0);
            std::vector<int> rc(a, SMO_OK);
            std::vector<std::string> msg(a);
            const int nthr = std::max(1, std::min<int>(std::min(a, 32), (int)std::thread::hardware_concurrency()));   // one wavenumber per task; <= 32 host threads
            auto reduce = [&](const std::vector<cd>& s, cd* dst) {
                std::copy(s.begin(), s.begin() + (size_t)3 * N * 3 * N, dst);
                for (int f = 0; f < 3; ++f) std::copy(&s[((size_t)(3 + f) * N + N - 1) * 3 * N], &s[((size_t)(3 + f) * N + N - 1) * 3 * N] + 3 * N, dst + (size_t)(3 * N + f) * 3 * N);
            };
            auto work = [&](int t) {
                std::vector<cd> red(use_hodlr ? sz : 0), perm;
                for (int n = t; n < a; n += nthr) {
                    std::vector<cd> s, sa, sm;
                    int r = build_solve_map(ch, n, k1 * n, 1.0 / cfg.dt, Re, Pe, Ri, s, false);
                    if (r == SMO_OK) r = build_solve_map(ch, n, k1 * n, 1.0 / cfg.dt, Re, Pe, Ri, sa, true);
                    if (r == SMO_OK) r = build_mixnorm_map(ch, n, k1 * n, sm);
                    if (r != SMO_OK) { rc[n] = r; msg[n] = last_error(); continue; }
                    if (use_hodlr) {
                        reduce(s, red.data());  hodlr_factor_reduced(plan, red.data(), N, perm, fS[n], eS[n]);
                        reduce(sa, red.data()); hodlr_factor_reduced(plan, red.data(), N, perm, fA[n], eA[n]);
                    } else {
                        reduce(s, &S[(size_t)n * sz]); reduce(sa, &SA[(size_t)n * sz]);
                    }
                    std::copy(sm.begin(), sm.end(), SM.begin() + (size_t)n * szm);
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < nthr; ++t) th.emplace_back(work, t);
            for (auto& t : th) t.join();
            for (int n = 0; n < a; ++n) if (rc[n] != SMO_OK) { set_error("%s (wavenumber %d)", msg[n].c_str(), n); return rc[n]; }
            auto up = [&](double2** p, const std::vector<cd>& h) -> int {
                SMO_TRY(pool.alloc(p, h.size()));
                SMO_HIP(hipMemcpyAsync(*p, h.data(), h.size() * sizeof(cd), hipMemcpyHostToDevice, stream));
                SMO_HIP(hipStreamSynchronize(stream));
                return SMO_OK;
            };
            SMO_TRY(up(&d_SMN, SM));
            if (use_hodlr) {
                double b1 = 0.0, b2 = 0.0;
                SMO_TRY(hop_build(pool, stream, plan, fS, eS, a, false, nthr, hS, &b1));
                SMO_TRY(hop_build(pool, stream, plan, fA, eA, a, false, nthr, hSA, &b2));
                op_bytes = 0.5 * (b1 + b2);
            } else {
                SMO_TRY(up(&d_S, S)); SMO_TRY(up(&d_SA, SA));
                op_bytes = (double)a * (double)sz * 16.0;
            }
        }
        SMO_TRY(pool.alloc(&S6, 6 * nC)); SMO_TRY(pool.alloc(&A6, 6 * nC)); SMO_TRY(pool.alloc(&R3, 3 * nC)); SMO_TRY(pool.alloc(&cur3, 3 * nC));
        SMO_TRY(pool.alloc(&G1, 12 * nL)); SMO_TRY(pool.alloc(&GR, 17 * nG)); SMO_TRY(pool.alloc(&PR, 3 * nG)); SMO_TRY(pool.alloc(&H, 3 * nL));
        SMO_TRY(pool.alloc(&HC, 3 * nC)); SMO_TRY(pool.alloc(&MN, 2 * nC)); SMO_TRY(pool.alloc(&d_X3, (size_t)2 * a * 3));
        SMO_TRY(pool.alloc(&d_stack, (size_t)(cfg.n_iters + 1) * 3 * nC));
        SMO_TRY(pool.alloc(&d_part, (size_t)(cfg.n_iters + 2) * NPART));
        h_part.resize((size_t)(cfg.n_iters + 2) * NPART);
        const int M2a = 2 * a;
        auto c_ = [&](double* base, int i) { return base + (size_t)i * nC; };
        auto l_ = [&](double* base, int i) { return base + (size_t)i * nL; };
        auto g_ = [&](double* base, int i) { return base + (size_t)i * nG; };
        SMO_TRY(make_phase(F0x, M2a, Gz, Gx, {{A_Xf, g_(GR, 0), l_(H, 0)}, {A_Xf, g_(GR, 1), l_(H, 1)}}));
        SMO_TRY(make_phase(F0z, M2a, Nz, Gz, {{l_(H, 0), B_ZfT, c_(S6, 0)}, {l_(H, 1), B_ZfT, c_(S6, 1)}}));
        { std::vector<GemmDesc> v; for (int f = 0; f < 6; ++f) v.push_back({c_(S6, f), B_ZiT, l_(G1, f)}); SMO_TRY(make_phase(Fz, M2a, Gz, Nz, v)); }
        // grids [u, ux, uz, w, wx, wz, bx, bz] from the lines of [u, w, b, uz, wz, bz]
        SMO_TRY(make_phase(Fx, Gx, Gz, M2a, {{A_Xi, l_(G1, 0), g_(GR, 0)}, {A_XiD, l_(G1, 0), g_(GR, 1)}, {A_Xi, l_(G1, 3), g_(GR, 2)},
                                             {A_Xi, l_(G1, 1), g_(GR, 3)}, {A_XiD, l_(G1, 1), g_(GR, 4)}, {A_Xi, l_(G1, 4), g_(GR, 5)},
                                             {A_XiD, l_(G1, 2), g_(GR, 6)}, {A_Xi, l_(G1, 5), g_(GR, 7)}}));
        { std::vector<GemmDesc> v, w, d1, d2;
          for (int f = 0; f < 3; ++f) { v.push_back({A_Xf, g_(PR, f), l_(H, f)}); w.push_back({l_(H, f), B_ZfT, c_(HC, f)});
                                        d1.push_back({c_(S6, f), B_DzT, c_(S6, 3 + f)}); d2.push_back({c_(A6, f), B_DzT, c_(A6, 3 + f)}); }
          SMO_TRY(make_phase(Fxf, M2a, Gz, Gx, v)); SMO_TRY(make_phase(Fzf, M2a, Nz, Gz, w));
          SMO_TRY(make_phase(F1d, M2a, Nz, Nz, d1)); SMO_TRY(make_phase(A1d, M2a, Nz, Nz, d2)); }
        SMO_TRY(make_phase(M1z, M2a, Gz, Nz, {{c_(MN, 0), B_ZiT, l_(G1, 0)}, {c_(MN, 1), B_ZiT, l_(G1, 1)}}));
        SMO_TRY(make_phase(M1x, Gx, Gz, M2a, {{A_XiD, l_(G1, 0), g_(GR, 0)}, {A_Xi, l_(G1, 1), g_(GR, 1)}}));
        // adjoint step: lines of [uf, wf, bf | dz uf, dz wf, dz bf | ua, wa, ba, uza, wza, bza]
        { std::vector<GemmDesc> v;
          for (int f = 0; f < 3; ++f) v.push_back({c_(cur3, f), B_ZiT, l_(G1, f)});
          for (int f = 0; f < 3; ++f) v.push_back({c_(cur3, f), B_DZiT, l_(G1, 3 + f)});
          for (int f = 0; f < 6; ++f) v.push_back({c_(A6, f), B_ZiT, l_(G1, 6 + f)});
          SMO_TRY(make_phase(Az, M2a, Gz, Nz, v)); }
        SMO_TRY(make_phase(Ax, Gx, Gz, M2a, {{A_Xi, l_(G1, 0), g_(GR, 0)}, {A_Xi, l_(G1, 1), g_(GR, 1)}, {A_XiD, l_(G1, 0), g_(GR, 2)},
                                             {A_XiD, l_(G1, 1), g_(GR, 3)}, {A_XiD, l_(G1, 2), g_(GR, 4)}, {A_Xi, l_(G1, 3), g_(GR, 5)},
                                             {A_Xi, l_(G1, 4), g_(GR, 6)}, {A_Xi, l_(G1, 5), g_(GR, 7)}, {A_Xi, l_(G1, 6), g_(GR, 8)},
                                             {A_Xi, l_(G1, 7), g_(GR, 9)}, {A_Xi, l_(G1, 8), g_(GR, 10)}, {A_XiD, l_(G1, 6), g_(GR, 11)},
                                             {A_XiD, l_(G1, 7), g_(GR, 12)}, {A_XiD, l_(G1, 8), g_(GR, 13)}, {A_Xi, l_(G1, 9), g_(GR, 14)},
                                             {A_Xi, l_(G1, 10), g_(GR, 15)}, {A_Xi, l_(G1, 11), g_(GR, 16)}}));
        SMO_TRY(make_phase(Gz2, M2a, Gz, Nz, {{c_(A6, 0), B_ZiT, l_(G1, 0)}, {c_(A6, 1), B_ZiT, l_(G1, 1)}}));
        SMO_TRY(make_phase(Gx2, Gx, Gz, M2a, {{A_Xi, l_(G1, 0), g_(GR, 0)}, {A_Xi, l_(G1, 1), g_(GR, 1)}}));
        k_gemm = timing.add_class("pois_gemm (transforms, MFMA f64)", 0.0);
        k_apply = timing.add_class(use_hodlr ? "pois_apply_hodlr (tau operator, HODLR form)" : "pois_apply (tau operator, batched complex GEMV)", op_bytes, op_bytes);
        k_point = timing.add_class("pois pointwise", 0.0);
        k_xfft = timing.add_class("pois_x (x transforms, LDS FFT)", 0.0);
        return SMO_OK;
    }

    int sum_partials(int row0, int nrows, double* out) {
        SMO_HIP(hipMemcpyAsync(h_part.data(), d_part + (size_t)row0 * NPART, (size_t)nrows * NPART * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        for (int r = 0; r < nrows; ++r) { double s = 0.0; for (int i = 0; i < NPART; ++i) s += h_part[(size_t)r * NPART + i]; out[r] = s; }
        return SMO_OK;
    }
    // one SBDF1 step of a 6-field state with the operator S: rhs grids PR[0..2] -> state (u, w, b, uz, wz, bz)
    int advance(double* state, bool adjoint_ivp, const Phase& deriv) {
        SMO_TRY(run(Fxf)); SMO_TRY(run(Fzf));
        {
            ScopedTimer t(timing, k_point, stream);
            hipLaunchKernelGGL(pois_axpy, pw_grid(3 * nC), dim3(256), 0, stream, R3, state, 1.0 / cfg.dt, HC, 3 * nC);
        }
        if (use_hodlr) {
            ScopedTimer t(timing, k_apply, stream);
            hop_launch(adjoint_ivp ? hSA : hS, stream, a, R3, nullptr, state, d_X3, nullptr, a, Nz);
        } else {
            SMO_TRY(apply(adjoint_ivp ? d_SA : d_S, R3, state, d_X3, 3, 3, 3));
        }
        SMO_TRY(run(deriv));
        ScopedTimer t(timing, k_point, stream);
        hipLaunchKernelGGL(pois_rank1_add, pw_grid(3 * nC), dim3(256), 0, stream, state + 3 * nC, d_X3, d_q, 2 * a, Nz);
        return SMO_OK;
    }
    // psi, psiz of snapshot N's density into MN; grids dx psi, psiz into GR[0], GR[1]
    int mixnorm_fields() {
        SMO_TRY(apply(d_SMN, snap(cfg.n_iters) + 2 * nC, MN, nullptr, 1, 2, 0));
        SMO_TRY(run(M1z));
        return run(M1x);
    }

    int forward_dev(const double* const* X, double* J) override {
        have_forward = false;
        const int N = cfg.n_iters;
        SMO_HIP(hipMemcpyAsync(GR, X[0], 2 * nG * sizeof(double), hipMemcpyDeviceToDevice, stream));
        SMO_HIP(hipMemsetAsync(S6, 0, 6 * nC * sizeof(double), stream));                  // uz = wz = 0 before the first step (POIS:653-657)
        SMO_TRY(run(F0x)); SMO_TRY(run(F0z));
        SMO_HIP(hipMemcpyAsync(S6 + 2 * nC, d_b0, nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
        SMO_HIP(hipMemcpyAsync(S6 + 5 * nC, d_bz0, nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
        for (int n = 0; n <= N; ++n) {                                                 // N_ITERS + 1 steps, like the script (stop_iteration = N_ITERS+1)
            SMO_HIP(hipMemcpyAsync(snap(n), S6, 3 * nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
            SMO_TRY(run(Fz)); SMO_TRY(run(Fx));
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_nl, dim3(NPART), dim3(256), 0, stream, GR, PR, d_Wq, d_part + (size_t)n * NPART, nG, Gz);
            }
            SMO_TRY(advance(S6, false, F1d));
        }
        double cost = 0.0;
        if (s_cost == 1) {
            SMO_TRY(mixnorm_fields());
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_wsq, dim3(NPART), dim3(256), 0, stream, GR, GR + nG, d_Wq, d_part + (size_t)(N + 1) * NPART, (double*)nullptr,
                                   (double*)nullptr, 0.0, nG, Gz);
            }
            double e = 0.0;
            SMO_TRY(sum_partials(N + 1, 1, &e));
            cost = 0.5 * e / V;
        } else {
            std::vector<double> e(N + 1);
            SMO_TRY(sum_partials(0, N + 1, e.data()));
            double ke = 0.0;
            for (int n = 0; n <= N; ++n) ke += cfg.dt * e[n] / V;
            cost = -0.5 * ke;
        }
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipStreamSynchronize(stream));
        *J = cost;
        have_forward = true;
        return SMO_OK;
    }

    int adjoint_dev(const double* const*, int adjoint_type, double* const* grad) override {
        if (adjoint_type != SMO_ADJ_CONTINUOUS) { set_error("POIS: this context was created for the Continuous formulation (cost >= 2)"); return SMO_ERR_ARG; }
        const int N = cfg.n_iters;
        SMO_HIP(hipMemsetAsync(A6, 0, 6 * nC * sizeof(double), stream));
        if (s_cost == 1) {                                                             // b_adj(0) = -psi (POIS:1268-1272)
            SMO_TRY(apply(d_SMN, snap(N) + 2 * nC, MN, nullptr, 1, 2, 0));
            ScopedTimer t(timing, k_point, stream);
            hipLaunchKernelGGL(pois_axpy, pw_grid(nC), dim3(256), 0, stream, A6 + 2 * nC, MN, -1.0, (const double*)nullptr, nC);
        }
        for (int idx = N; idx >= 1; --idx) {
            SMO_HIP(hipMemcpyAsync(cur3, snap(idx), 3 * nC * sizeof(double), hipMemcpyDeviceToDevice, stream));
            SMO_TRY(run(Az)); SMO_TRY(run(Ax));
            {
                ScopedTimer t(timing, k_point, stream);
                hipLaunchKernelGGL(pois_cnts_adj_rhs, pw_grid(nG), dim3(256), 0, stream, GR, PR, s_cost == 0 ? 1 : 0, nG);
            }
            SMO_TRY(advance(A6, true, A1d));
        }
        SMO_TRY(run(Gz2)); SMO_TRY(run(Gx2));
        SMO_HIP(hipMemcpyAsync(grad[0], GR, 2 * nG * sizeof(double), hipMemcpyDeviceToDevice, stream));
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }

    int inner_dev(const double* x, const double* y, double* out) override {
        hipLaunchKernelGGL(pois_dot, dim3(NPART), dim3(256), 0, stream, x, y, d_Wq, d_part + (size_t)(cfg.n_iters + 1) * NPART, 2 * nG, Gz);
        SMO_HIP(hipGetLastError());
        double s = 0.0;
        SMO_TRY(sum_partials(cfg.n_iters + 1, 1, &s));
        *out = s / V;
        return SMO_OK;
    }

    int snapshot_read(int, int index, double* out) override {
        std::vector<double> h(3 * nC);
        SMO_HIP(hipMemcpyAsync(h.data(), snap(index), 3 * nC * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        for (int f = 0; f < 3; ++f)
            for (int n = 0; n < a; ++n)
                for (int j = 0; j < Nz; ++j) {
                    out[(((size_t)f * a + n) * Nz + j) * 2] = h[(size_t)f * nC + ((size_t)2 * n) * Nz + j];
                    out[(((size_t)f * a + n) * Nz + j) * 2 + 1] = h[(size_t)f * nC + ((size_t)2 * n + 1) * Nz + j];
                }
        return SMO_OK;
    }
};

}  // namespace

Context* make_pois(const smo_config& cfg) {
    if (cfg.cost >= 2 && cfg.cost <= 3) return new PoisCnts(cfg);      // Continuous formulation: s = cost - 2
    return new Pois(cfg);
}

}  // namespace smo

// KDYN — 3-D triply-periodic kinematic dynamo: CNAB1 forward solve of the induction equation and the discrete /
// continuous adjoint sweep giving dJ/dB0 and dJ/dU.
//
// Replaces FWD_Solve_IVP_Lin / Compatib_Cond / ADJ_Solve_IVP_Lin / Inner_Prod_3 of
// Example_Problems/Periodic_Domain(Fourier)/Kinematic_Dynamo/FWD_Solve_KDyn.py (:529-689, :696-764, :766-1004,
// :173-181); recurrences: SURVEY.md Appendix A.2.
//
// Data layout in HBM (N = npts, a = N/2 kx modes, m = N-1 ky/kz modes, G = 3N/2 grid points per axis):
//   coefficient fields  C [3][a][m][m]   complex128, kz fastest      (snapshot stack: [n][3][a][m][m])
//   after the z pass    Tz[3][a][m][G]   complex128, z fastest       (slab exchange layout: [peer][field group][3][a/W][m][G/W])
//   after the y pass    Ty[3][a][G*G+8]  complex128, z fastest       (slabs: [3][a][G*G/W+8], all kx, local z planes; the 8 elements of
//                                                                     padding keep a tile's 3a runs off one HBM channel)
//   grid fields         U [3][G][G][G]   float64,    z fastest       (= the reference's flat X vectors; slabs: [3][G][G][G/W])
// Slab decomposition (W ranks): coefficient space is split over kx, grid space over z.  The z passes run on the kx slab, the y
// and x passes on the z slab, and the exchange sits between the z and the y pass, where the data is smallest (16*a*m*G bytes per
// component instead of 16*a*G*G after the y pass).  With W = 1 the "exchange buffer" is simply Tz.
// One 3-D transform = three 1-D passes (z contiguous, y strided, x strided).  Every pass reads HBM in its first
// Stockham stage (zero padding folded into the load) and writes HBM in its last (truncation folded into the store).
// Fusions per time step (4 kernels per forward step, 4 per adjoint step):
//   x pass  : c2r (two real lines per complex FFT) -> cross product with U (and B_f) on the grid -> r2c, one kernel; the last
//             inverse stage, the products and the first forward stage run in registers
//   z pass  : forward z transform -> i k x (.) -> Leray projection -> CNAB1 update -> next state written straight
//             into the snapshot stack (the stack IS the state; no copy) -> inverse z transform of that state (adjoint: of its
//             curl) for the next step, on the same tile
//   nu      : the adjoint's second product is summed over the steps on the grid side (x-transformed) and transformed once
#include <algorithm>
#include <chrono>
#include <type_traits>

#include "comm.hpp"
#include "fft_lds.hpp"

namespace smo {
namespace {

struct Geom {
    int a;        // global number of kx modes (N/2)
    int al;       // local number of kx modes (a / world)
    int ix0;      // first global kx of this slab
    int m;        // ky / kz modes (N-1)
    int kmax;     // (N-1)/2
    int G;        // grid points per axis (3N/2)
    int Gzl;      // z planes the y / x passes of one launch see: the rank's slab G / world, or one of its `chunks` equal parts
    int Gzr;      // z planes per rank (G / world)
    int W;        // number of slabs
    int zg0;      // caller-layout grid vectors [3][G][G][Gzr]: first z plane of the chunk this launch works on
    size_t blk;   // slab-exchange layout: elements of one (chunk, peer) block (= field groups * 3 * al * m * Gzl)
    size_t cblk;  // elements between consecutive chunks (= W * 2 field groups * 3 * al * m * Gzl, whatever the number of groups in use,
                  // so that a chunk's region with one group lies inside its region with two)
    size_t typ;   // Ty: elements between consecutive (component, kx) planes (G * Gzl + padding; the padding keeps the x pass's 3a
                  // runs per tile off a power-of-two stride)
    int utile;    // x pass spectrum -> grid: 1 = write the tile-major layout of the internal U field (u_off), 0 = the flat X layout
    int tyl;      // layout of Ty: 0 = planes [c][kx][y][z] (+ padding per plane), 1 = z-block major [c][z/8][kx][y][z%8] (below)
    int ykx;      // y pass, z-block-major Ty: 1 = consecutive workgroups take consecutive kx (their G*128-byte blocks are adjacent)
    size_t tyk;   // z-block major: elements between consecutive kx blocks (8 G + one 128-byte line of padding)
    double Rm, dt;
};

__device__ __forceinline__ int wrap_pos(int pos, const Geom& g) {      // position in the padded length-G spectrum -> stored index, -1 in the gap
    if (pos <= g.kmax) return pos;
    if (pos >= g.G - g.kmax) return pos - (g.G - g.m);
    return -1;
}
__device__ __forceinline__ double wavenumber(int idx, const Geom& g) { return (idx <= g.kmax) ? (double)idx : (double)(idx - g.m); }

// Tz in the slab-exchange layout [chunk (stride cblk)][peer][field group][3][al][m][Gzl] (the field-group offset is folded into the base pointer;
// chunk-major so that every chunk is one contiguous all_to_all_single, which lets the host pipeline the exchange of one chunk with
// the grid-side work on another):
// z side: the peer index is the z block of `pos`; rt = ixl * m + iy is the local (kx, ky) row
__device__ __forceinline__ size_t zs_off(int c, int rt, int pos, const Geom& g) {
    const size_t row = ((size_t)c * (g.al * g.m) + rt) * g.Gzl;
    if (g.Gzl == g.G) return row + pos;                                   // one slab, one chunk (uniform branch: no divisions)
    const int p = pos / g.Gzr, zl = pos - p * g.Gzr;
    const int k = zl / g.Gzl, zc = zl - k * g.Gzl;
    return (size_t)k * g.cblk + (size_t)p * g.blk + row + zc;
}
// caller-layout grid vector [3][G][G][Gzr]: offset of (c, x) at flat (y, z) index i of the chunk
__device__ __forceinline__ size_t grid_off(int c, int x, size_t i, const Geom& g) {
    const size_t y = i / g.Gzl;
    return (((size_t)c * g.G + x) * g.G + y) * g.Gzr + g.zg0 + (i - y * g.Gzl);
}
// y side (inside one chunk): the peer index is the kx block; offset of the (c, kx, iy = 0) row, z local
__device__ __forceinline__ size_t ys_row0(int c, int kx, const Geom& g) {
    const int q = kx / g.al;
    return (size_t)q * g.blk + ((size_t)c * g.al + (kx - q * g.al)) * ((size_t)g.m * g.Gzl);
}
// x pass: offset of mode kx (global) of component c, flat local (y,z) index i, in Ty[3][a][G*Gzl] (the plane layout; the run-time-length
// kernels of kdyn_any.hpp use only this one)
__device__ __forceinline__ size_t tx_off(int c, int kx, size_t i, const Geom& g) {
    return ((size_t)c * g.a + kx) * g.typ + i;
}
// Z-block-major Ty (round 3), Ty[c][z/8][kx][y][z%8], kx blocks G*8 + 8 elements apart.  In the plane layout a fused x-pass tile gathers one
// 64...128-byte run from each of 3a planes 16*G*Gzl bytes apart (2.4 MB at G = 384) and a y-pass workgroup G lines 16*Gzl bytes apart; here the
// x-pass runs of a tile lie G*128 + 128 bytes apart inside ONE window of a*(G*128 + 128) bytes per component (6.3 MB at G = 384; the extra line
// keeps them off one HBM channel, like the plane padding) and a y-pass workgroup (c, kx, 8 z columns) owns one contiguous G*128-byte block.
// tools/micro_patch_layout.hip, G = 384, same box: x pattern 4.95 -> 5.35 TB/s (tiles walked y-fastest inside a z block), y pattern 5.44 ->
// 5.63 TB/s (workgroups kx-fastest); 8x8 patches 5.36 / 5.22, row blocks 5.47 / 4.88; without the padding 4.17-4.33 (channel camping).
__device__ __forceinline__ size_t ty_off(int c, int kx, size_t i, const Geom& g) {      // any flat index i (one division: setup passes only)
    if (g.tyl == 0) return tx_off(c, kx, i, g);
    const size_t y = i / g.Gzl;
    const int z = (int)(i - y * g.Gzl);
    return (((size_t)c * (g.Gzl >> 3) + (z >> 3)) * g.a + kx) * g.tyk + y * 8 + (z & 7);
}
// the fused x passes: a tile of T <= 8 points starting at i0 (a multiple of T) lies inside one z block, so
//   offset(c, kx, i0 + e) = base + c * cs + kx * ks + e
struct XOrigin { size_t base, cs, ks; };
__device__ __forceinline__ XOrigin x_origin(size_t i0, const Geom& g) {
    if (g.tyl == 0) return XOrigin{i0, (size_t)g.a * g.typ, g.typ};
    const size_t y = i0 / g.Gzl;
    const int z0 = (int)(i0 - y * g.Gzl);
    return XOrigin{(size_t)(z0 >> 3) * g.a * g.tyk + y * 8 + (z0 & 7), (size_t)(g.Gzl >> 3) * g.a * g.tyk, g.tyk};
}

// The velocity field U is only ever read by the fused x passes, one (y,z) tile per workgroup, all x.  It is therefore kept
// tile-major, U[c][i/4][x][i%4] (i = flat local (y,z) index), so that a workgroup streams it as one contiguous run instead of
// G segments of 32-64 bytes (which cost 2-4x over-fetch of 128-byte lines).  Pairs (i, i+1) with i even never straddle a block.
constexpr int UT = 4;
__device__ __forceinline__ size_t u_off(int c, int x, size_t i, const Geom& g) {
    const size_t nt = ((size_t)g.G * g.Gzl) / UT;
    return (((size_t)c * nt + i / UT) * g.G + x) * UT + (i % UT);
}

// Streaming loads of the fused x passes.  At 128^3 a step hands ~1 GB from kernel to kernel (Ty, the EMF's spectrum, Tz) and those hand-overs can be
// served by the 256-MB cache behind the L2s — unless once-read streams displace them.  SMO_X_NT marks such streams non-temporal (bit mask):
//   1  the velocity tile (both passes): 170 MB per launch, the same every step            forward x pass 90.8 -> 80.4 us, adjoint 164.0 -> 159.7 us
//  16  the forward pass's input spectrum where a tile reads whole 128-byte lines (G <= 192) y<fwd> 35.3 -> 33.9 us; hurts with half-line tiles (G = 384)
//   2  the kept forward state B_f, 4 / 8 the running sum's loads / stores, 32 omega's spectrum (adjoint pass): measured, no gain (profiles/r02_nt_streams.txt)
#ifndef SMO_X_NT
#define SMO_X_NT 17
#endif
#ifndef SMO_Z_NT
#define SMO_Z_NT 0
#endif
typedef double d2_t __attribute__((ext_vector_type(2)));
template <bool NTL> __device__ __forceinline__ cplx ld_cplx(const cplx* p) {
    if (NTL) { const d2_t v = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(p)); return mk(v.x, v.y); }
    return *p;
}
template <bool NTL> __device__ __forceinline__ cplx ld_pair(const double* q) {        // two consecutive grid values (16-byte aligned: even flat index)
    if (NTL) { const d2_t v = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(q)); return mk(v.x, v.y); }
    return mk(q[0], q[1]);
}

// ---------------------------------------------------------------------------------------------------------
// z pass, inverse (coefficients -> Tz), rows contiguous.  NBT (ix,iy) row triples per workgroup.
// ---------------------------------------------------------------------------------------------------------
enum { ZI_PLAIN = 0, ZI_CURL = 1, ZI_SCALE = 2 };     // load the field | load i k x field | load dt*alpha(k)*field

// Tile of the z passes: one row per transform (lanes run along a row: the global accesses are the contiguous ones), element (b, pos) at
// b * L + swz(pos).  Row-major as it stands, the stride-4 stores of the first two Stockham stages put the 8 lanes of a ds_write_b128 group on
// 2 (first stage) / 4 (second) of the 8 sixteen-byte slots: 32-38 % of the LDS-active cycles of the update kernels were bank conflicts
// (profiles/r03_kdyn*_sq_counters.txt).  SMO_Z_SWZ = 1: pos ^ ((pos >> 2) & 7), 2: pos ^ ((pos >> 3) & 7) — a permutation inside every aligned
// block of 8 positions (needs L % 8 == 0) that spreads those stores over all slots (tools/lds_conflict_model.py rules: array cycles per tile
// 620 -> 416 / 444 at G = 192, 1440 -> 1152 / 1008 at G = 384; the contiguous stage reads pay a little instead).
#ifndef SMO_Z_SWZ
#define SMO_Z_SWZ 0
#endif
template <int L> struct ZIx {
    static constexpr int SWZ = (L % 8 == 0) ? SMO_Z_SWZ : 0;
    __device__ __forceinline__ int operator()(int b, int pos) const {
        return b * L + (SWZ == 1 ? (pos ^ ((pos >> 2) & 7)) : SWZ == 2 ? (pos ^ ((pos >> 3) & 7)) : pos);
    }
};

template <int L, int MODE, int NBT, int NT>
__global__ __launch_bounds__(NT) void kd_z_inverse(const cplx* __restrict__ in, cplx* __restrict__ out, const cplx* __restrict__ tw_g,
                                                   Geom g) {
    constexpr int NB = 3 * NBT;
    constexpr ZIx<L> ix{};
    __shared__ cplx buf[NB * L];
#ifndef SMO_Z_HALF_TW_MIN_L
#define SMO_Z_HALF_TW_MIN_L 200
#endif
    // half twiddle table from SMO_Z_HALF_TW_MIN_L on: 24.6 -> 21.5 KB at G = 384, seven workgroups per CU instead of six (256^3: z_inverse<curl>
    // 256.8 -> 223.4 us, fwd_update 426.2 -> 418.3, adj_update unchanged)
    constexpr bool HALF = (L >= SMO_Z_HALF_TW_MIN_L) && (L % 2 == 0);
    constexpr int NTW = HALF ? L / 2 : L;
    __shared__ cplx tw_s[NTW];
    const int tid = threadIdx.x;
    for (int i = tid; i < NTW; i += NT) tw_s[i] = tw_g[i];
    __syncthreads();
    using TWT = typename std::conditional<HALF, HalfTwiddles, const cplx*>::type;
    TWT tw;
    if constexpr (HALF) tw = HalfTwiddles{tw_s, L / 2};
    else tw = tw_s;
    const int nrt = g.al * g.m;
    const int rt0 = blockIdx.x * NBT;
    const size_t cs = (size_t)nrt * g.m;
    auto ld0 = [&](int b, int pos) -> cplx {
        const int tt = b / 3, c = b - 3 * tt, rt = rt0 + tt;
        const int idx = wrap_pos(pos, g);
        if (rt >= nrt || idx < 0) return mk(0, 0);
        const size_t e = (size_t)rt * g.m + idx;
        if (MODE == ZI_PLAIN) return in[c * cs + e];
        const int ixl = rt / g.m, iy = rt - ixl * g.m;
        const double k[3] = {(double)(g.ix0 + ixl), wavenumber(iy, g), wavenumber(idx, g)};
        if (MODE == ZI_SCALE) {
            const double s = g.dt * (1.0 / g.dt + 0.5 * (k[0] * k[0] + k[1] * k[1] + k[2] * k[2]) / g.Rm);
            return s * in[c * cs + e];
        }
        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
        cplx v1 = in[c1 * cs + e], v2 = in[c2 * cs + e];
        return mul_i(mk(k[c1] * v2.re - k[c2] * v1.re, k[c1] * v2.im - k[c2] * v1.im));      // (i k x V)_c
    };
    auto stN = [&](int b, int pos, cplx v) {
        const int tt = b / 3, c = b - 3 * tt, rt = rt0 + tt;
        if (rt < nrt) out[zs_off(c, rt, pos, g)] = v;
    };
    if (MODE == ZI_CURL) {
        // every component feeds two components of the curl: read the rows once, coalesced, into the tile and form i k x V from there
        // (loading them twice from global memory cost 20 % at G = 192 and 35 % at G = 384)
        for (int t = tid; t < NB * g.m; t += NT) {
            const int b = t / g.m, idx = t - b * g.m, tt = b / 3, c = b - 3 * tt, rt = rt0 + tt;
            const int pos = (idx <= g.kmax) ? idx : idx + (g.G - g.m);
            buf[ix(b, pos)] = (rt < nrt) ? in[c * cs + (size_t)rt * g.m + idx] : mk(0, 0);
        }
        __syncthreads();
        auto ldc = [&](int b, int pos) -> cplx {
            const int tt = b / 3, c = b - 3 * tt, rt = rt0 + tt;
            const int idx = wrap_pos(pos, g);
            if (rt >= nrt || idx < 0) return mk(0, 0);
            const int ixl = rt / g.m, iy = rt - ixl * g.m;
            const double k[3] = {(double)(g.ix0 + ixl), wavenumber(iy, g), wavenumber(idx, g)};
            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
            const cplx v1 = buf[ix(tt * 3 + c1, pos)], v2 = buf[ix(tt * 3 + c2, pos)];
            return mul_i(mk(k[c1] * v2.re - k[c2] * v1.re, k[c1] * v2.im - k[c2] * v1.im));      // (i k x V)_c
        };
        fft_inplace_ix<L, true, NB, NT, false, true, false>(buf, ix, tw, tid, ldc, stN);
        return;
    }
    fft_inplace_ix<L, true, NB, NT, false, false, false>(buf, ix, tw, tid, ld0, stN);
}

// ---------------------------------------------------------------------------------------------------------
// z pass, forward (Tz -> coefficients), with the per-mode time-step fused in
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cnab_mode(const double k[3], double k2, double alpha, double beta, const cplx V0[3], const cplx F[3],
                                          cplx V1[3]) {
    if (k2 == 0.0) {                                    // algebraic rows under CNAB1: X1 = -X0
        for (int c = 0; c < 3; ++c) V1[c] = mk(-V0[c].re, -V0[c].im);
        return;
    }
    cplx r[3];
    for (int c = 0; c < 3; ++c) r[c] = mk(beta * V0[c].re + F[c].re, beta * V0[c].im + F[c].im);
    const double ik2 = 1.0 / k2;
    const cplx kr = mk((k[0] * r[0].re + k[1] * r[1].re + k[2] * r[2].re) * ik2, (k[0] * r[0].im + k[1] * r[1].im + k[2] * r[2].im) * ik2);
    const cplx kv = mk((k[0] * V0[0].re + k[1] * V0[1].re + k[2] * V0[2].re) * ik2,
                       (k[0] * V0[0].im + k[1] * V0[1].im + k[2] * V0[2].im) * ik2);
    for (int c = 0; c < 3; ++c)
        V1[c] = mk((r[c].re - k[c] * kr.re) / alpha - k[c] * kv.re, (r[c].im - k[c] * kr.im) / alpha - k[c] * kv.im);
}

enum { ZF_PLAIN = 0, ZF_FWD_UPDATE = 1, ZF_ADJ_UPDATE = 2, ZF_NU = 3 };

// ZF_PLAIN: truncated spectrum;  ZF_FWD_UPDATE: B^_{n+1};  ZF_ADJ_UPDATE: G^ update with forcing F1 = F[(curl G) x U] (minus 2 B_f if
// integrated);  ZF_NU: nu^ = -dt P(F[sum_n (curl G_n) x B_n]) from the accumulated product (see the fused adjoint x pass)
// NEXT: the z pass that opens the following time step works on the very rows this kernel has just updated, so it can run here, on the
// tile, instead of re-reading the new state in a kernel of its own: NX_PLAIN = inverse z pass of the new state (forward solve),
// NX_CURL = inverse z pass of i k x (new state) (adjoint solve).  The result goes where the input came from (the z-side exchange
// buffer, same layout), which is safe because a workgroup owns its rows.
enum { NX_NONE = 0, NX_PLAIN = 1, NX_CURL = 2 };
template <int L, int MODE, int NEXT, int NBT, int NT>
__global__ __launch_bounds__(NT) void kd_z_forward(const cplx* inA, cplx* out0, const cplx* state0 /* may alias out0 */,
                                                   const cplx* snap, const cplx* __restrict__ tw_g, Geom g, double scale, int integrated,
                                                   cplx* next_out /* may alias inA */) {
    constexpr int NB = 3 * NBT;
    constexpr ZIx<L> ix{};
    __shared__ cplx buf[NB * L];
#ifndef SMO_Z_HALF_TW_MIN_L
#define SMO_Z_HALF_TW_MIN_L 200
#endif
    // half twiddle table from SMO_Z_HALF_TW_MIN_L on: 24.6 -> 21.5 KB at G = 384, seven workgroups per CU instead of six (256^3: z_inverse<curl>
    // 256.8 -> 223.4 us, fwd_update 426.2 -> 418.3, adj_update unchanged)
    constexpr bool HALF = (L >= SMO_Z_HALF_TW_MIN_L) && (L % 2 == 0);
    constexpr int NTW = HALF ? L / 2 : L;
    __shared__ cplx tw_s[NTW];
    const int tid = threadIdx.x;
    for (int i = tid; i < NTW; i += NT) tw_s[i] = tw_g[i];
    __syncthreads();
    using TWT = typename std::conditional<HALF, HalfTwiddles, const cplx*>::type;
    TWT tw;
    if constexpr (HALF) tw = HalfTwiddles{tw_s, L / 2};
    else tw = tw_s;
    const int nrt = g.al * g.m;
    const int rt0 = blockIdx.x * NBT;
    const size_t cs = (size_t)nrt * g.m;
    // b = tt * 3 + c
    auto ld0 = [&](int b, int pos) -> cplx {
        const int c = b % 3, tt = b / 3, rt = rt0 + tt;
        if (rt >= nrt) return mk(0, 0);
        return inA[zs_off(c, rt, pos, g)];
    };
    if (MODE == ZF_PLAIN) {
        auto stN = [&](int b, int pos, cplx v) {
            const int c = b % 3, tt = b / 3, rt = rt0 + tt;
            const int idx = wrap_pos(pos, g);
            if (rt < nrt && idx >= 0) out0[c * cs + (size_t)rt * g.m + idx] = scale * v;
        };
        fft_inplace_ix<L, false, NB, NT, false, false, false>(buf, ix, tw, tid, ld0, stN);
        return;
    }
    fft_inplace_ix<L, false, NB, NT, false, false, true>(buf, ix, tw, tid, ld0, [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; });
    __syncthreads();
    // per-mode update: thread <-> (tt, iz), iz fastest
    for (int t = tid; t < NBT * g.m; t += NT) {
        const int tt = t / g.m, iz = t - tt * g.m, rt = rt0 + tt;
        if (rt >= nrt) continue;
        const int ixl = rt / g.m, iy = rt - ixl * g.m;
        const int pos = (iz <= g.kmax) ? iz : iz + (g.G - g.m);
        const double k[3] = {(double)(g.ix0 + ixl), wavenumber(iy, g), wavenumber(iz, g)};
        const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
        const double D = k2 / g.Rm, alpha = 1.0 / g.dt + 0.5 * D, beta = 1.0 / g.dt - 0.5 * D;
        const size_t e = (size_t)rt * g.m + iz;
        cplx E[3], V0[3], V1[3];
        for (int c = 0; c < 3; ++c) E[c] = scale * buf[ix(tt * 3 + c, pos)];
        if (MODE == ZF_NU) {
            if (k2 == 0.0) {
                for (int c = 0; c < 3; ++c) V1[c] = mk(0, 0);
            } else {
                const double ik2 = 1.0 / k2;
                const cplx kf = mk((k[0] * E[0].re + k[1] * E[1].re + k[2] * E[2].re) * ik2, (k[0] * E[0].im + k[1] * E[1].im + k[2] * E[2].im) * ik2);
                for (int c = 0; c < 3; ++c) V1[c] = mk(-g.dt * (E[c].re - k[c] * kf.re), -g.dt * (E[c].im - k[c] * kf.im));
            }
            for (int c = 0; c < 3; ++c) out0[c * cs + e] = V1[c];
            continue;
        }
        for (int c = 0; c < 3; ++c) V0[c] = ld_cplx<(SMO_Z_NT & 1) != 0>(state0 + c * cs + e);
        if (MODE == ZF_FWD_UPDATE) {
            cplx F[3];                                  // N^ = i k x E^
            for (int c = 0; c < 3; ++c) {
                const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                F[c] = mul_i(mk(k[c1] * E[c2].re - k[c2] * E[c1].re, k[c1] * E[c2].im - k[c2] * E[c1].im));
            }
            cnab_mode(k, k2, alpha, beta, V0, F, V1);
        } else {
            if (integrated)
                for (int c = 0; c < 3; ++c) { cplx bf = snap[c * cs + e]; E[c] = mk(E[c].re - 2.0 * bf.re, E[c].im - 2.0 * bf.im); }
            cnab_mode(k, k2, alpha, beta, V0, E, V1);
        }
        for (int c = 0; c < 3; ++c) {
            if (SMO_Z_NT & 2) __builtin_nontemporal_store(d2_t{V1[c].re, V1[c].im}, reinterpret_cast<d2_t*>(out0 + c * cs + e));
            else out0[c * cs + e] = V1[c];
        }
        if (NEXT != NX_NONE) {                          // the new state (or its curl) replaces the spectrum in the tile
            for (int c = 0; c < 3; ++c) {
                const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                buf[ix(tt * 3 + c, pos)] = (NEXT == NX_PLAIN) ? V1[c]
                    : mul_i(mk(k[c1] * V1[c2].re - k[c2] * V1[c1].re, k[c1] * V1[c2].im - k[c2] * V1[c1].im));
            }
        }
    }
    if (NEXT != NX_NONE) {
        const int gap = g.G - g.m;                      // zero padding between the positive and the negative wavenumbers
        for (int t = tid; t < NB * gap; t += NT) {
            const int b = t / gap, q = t - b * gap;
            buf[ix(b, g.kmax + 1 + q)] = mk(0, 0);
        }
        __syncthreads();
        fft_inplace_ix<L, true, NB, NT, false, true, false>(buf, ix, tw, tid, [&](int b, int pos) { return buf[ix(b, pos)]; },
                                                         [&](int b, int pos, cplx v) {
                                                             const int c = b % 3, tt = b / 3, rt = rt0 + tt;
                                                             if (rt < nrt) next_out[zs_off(c, rt, pos, g)] = v;
                                                         });
    }
}

// ---------------------------------------------------------------------------------------------------------
// y pass (strided): Tz (slab-exchange layout, all kx, local z) <-> Ty[3][a][G][Gzl]; ZT consecutive z per workgroup
// ---------------------------------------------------------------------------------------------------------
template <int L, bool INV, int ZT, int NT>
__global__ __launch_bounds__(NT) void kd_y_pass(const cplx* __restrict__ in, cplx* __restrict__ out, const cplx* __restrict__ tw_g, Geom g) {
    // position-major tile (PosMajor, fft_lds.hpp): the ZT columns of a tile interleaved, every stage access of a wave is one contiguous
    // run (model: G = 192 reads 2x -> 1x, G = 384 3x -> 1x conflict cycles against rows of L + 1)
    constexpr PosMajor<ZT> ix{};
    // the z-block-major Ty addressing below (trow, tys = 8) is written for tiles that lie inside one block of 8 z columns
    static_assert(ZT <= 8 && 8 % ZT == 0, "kd_y_pass: SMO_Y_ZT must divide 8 (z-block-major Ty)");
    __shared__ cplx buf[ZT * L];
    // half twiddle table (the other half is its negative) from SMO_Y_HALF_TW_MIN_L on: at G = 384 the 8-column tile + the full table is 55.3 KB — two
    // workgroups per CU, 2.96 would fit —, with half a table 52.2 KB: three.  Round 4, 256^3, same box: y<inv> 310.0 -> 299.4 us, y<fwd> 285.5 -> 270.9;
    // at G = 192 (five -> six workgroups per CU) nothing moves, so the full table stays there (and the results of the smaller grids as they were)
#ifndef SMO_Y_HALF_TW_MIN_L
#define SMO_Y_HALF_TW_MIN_L 200
#endif
    constexpr bool HALF = (L >= SMO_Y_HALF_TW_MIN_L) && (L % 2 == 0);
    constexpr int NTW = HALF ? L / 2 : L;
    __shared__ cplx tw_s[NTW];
    const int tid = threadIdx.x;
    for (int i = tid; i < NTW; i += NT) tw_s[i] = tw_g[i];
    __syncthreads();
    using TWT = typename std::conditional<HALF, HalfTwiddles, const cplx*>::type;
    TWT tw;
    if constexpr (HALF) tw = HalfTwiddles{tw_s, L / 2};
    else tw = tw_s;
    const int ntile = (g.Gzl + ZT - 1) / ZT;
    int c, kx, z0;
    if (g.tyl && g.ykx) {                       // consecutive workgroups: consecutive kx of one (c, z tile) — adjacent blocks of Ty
        kx = blockIdx.x % g.a;
        const int r = blockIdx.x / g.a;
        c = r / ntile; z0 = (r - c * ntile) * ZT;
    } else {
        const int o = blockIdx.x / ntile;       // o = c * a + kx
        z0 = (blockIdx.x - o * ntile) * ZT;
        c = o / g.a; kx = o - c * g.a;
    }
    const size_t zrow = ys_row0(c, kx, g) + z0;                               // + idx * Gzl + b
    // Ty side: + y * tys + b
    const size_t trow = g.tyl ? (((size_t)c * (g.Gzl >> 3) + (z0 >> 3)) * g.a + kx) * g.tyk + (z0 & 7) : ((size_t)c * g.a + kx) * g.typ + z0;
    const size_t tys = g.tyl ? 8 : (size_t)g.Gzl;
    if (INV) {
        auto ld0 = [&](int b, int pos) -> cplx {
            const int idx = wrap_pos(pos, g);
            if (idx < 0 || z0 + b >= g.Gzl) return mk(0, 0);
            return in[zrow + (size_t)idx * g.Gzl + b];
        };
        auto stN = [&](int b, int pos, cplx v) {
            if (z0 + b < g.Gzl) out[trow + (size_t)pos * tys + b] = v;
        };
        fft_inplace_ix<L, true, ZT, NT, true, false, false>(buf, ix, tw, tid, ld0, stN);
    } else {
        auto ld0 = [&](int b, int pos) -> cplx {
            if (z0 + b >= g.Gzl) return mk(0, 0);
            return in[trow + (size_t)pos * tys + b];
        };
        auto stN = [&](int b, int pos, cplx v) {
            const int idx = wrap_pos(pos, g);
            if (idx >= 0 && z0 + b < g.Gzl) out[zrow + (size_t)idx * g.Gzl + b] = v;
        };
        fft_inplace_ix<L, false, ZT, NT, true, false, false>(buf, ix, tw, tid, ld0, stN);
    }
}

// ---------------------------------------------------------------------------------------------------------
// x pass (strided, real <-> Hermitian half spectrum), two real lines per complex FFT, T flat (y,z) points per
// workgroup.  Modes: spectrum -> grid; grid -> spectrum; fused  spectrum -> grid product(s) -> spectrum.
// ---------------------------------------------------------------------------------------------------------
enum { X_TO_GRID = 0, X_FROM_GRID = 1, X_FUSED_FWD = 2, X_FUSED_ADJ = 3, X_FUSED_ADJ_SEQ = 4 };

// Row padding of the x pass's LDS tile (complex elements).  Lanes run over the NB transforms first, then over consecutive positions:
// element (b, pos) starts at bank group (pad * b + pos) mod 16, and a 64-lane 16-byte access is conflict-free when each of the 16
// groups gets 4 lanes, i.e. when the NB ranges [pad * b, pad * b + 64/NB) tile the line: pad ~ 64 / NB.
constexpr int x_ld_pad(int NB) { return NB == 12 ? 5 : (NB == 6 ? 10 : 1); }
// Layout of the fused x passes' tile: position-major (the NB transforms of a tile interleaved, see PosMajor in fft_lds.hpp) — every
// stage read of a wave is one contiguous run, no row padding needed.  SMO_X_POSMAJOR=0 builds the row-per-transform layout (ablation).
#ifndef SMO_X_POSMAJOR
#define SMO_X_POSMAJOR 1
#endif
template <int L, int NB, bool FUSED> struct XLayout {
    static constexpr bool PM = FUSED && SMO_X_POSMAJOR;
    static constexpr int LD = L + x_ld_pad(NB);
    static constexpr int ELEMS = PM ? NB * L : NB * LD;
    __device__ __forceinline__ int operator()(int b, int pos) const { return PM ? pos * NB + b : b * LD + pos; }
};

// spectra of the x pass: field groups A / B are read from in* and written to out* (same layout; in == out means in place)
struct XSpec {
    const cplx* inA;
    const cplx* inB;
    cplx* outA;
    cplx* outB;
};

constexpr bool only_2_and_3(int n) { while (n % 2 == 0) n /= 2; while (n % 3 == 0) n /= 3; return n == 1; }
// Radix policy of the adjoint x passes (both forms: the same arithmetic per element, bit-identical results): butterflies up to radix 8 where the
// registers of the sequential form allow it — in the inverse heads at G <= 192 (165 VGPRs, -2.3 %), in the forward tails at G = 288, 384 (168,
// -1.5 %); the other combination spills at either size (17-50 registers); lengths with a radix-5 / radix-7 stage stay with radix 4
#ifndef SMO_X_SEQ_HEAD_RADIX
#define SMO_X_SEQ_HEAD_RADIX ((only_2_and_3(L) && L <= 192) ? 8 : SMO_FFT_MAX_RADIX)
#endif
#ifndef SMO_X_SEQ_TAIL_RADIX
#define SMO_X_SEQ_TAIL_RADIX ((only_2_and_3(L) && L > 192) ? 8 : SMO_FFT_MAX_RADIX)
#endif

template <int L, int MODE, int T, int NT, class TW>
__device__ __forceinline__ void x_tile(const XSpec& sp, const double* __restrict__ gridU, double* gridOut, const Geom& g,
                                       cplx* buf, const TW tw, const size_t i0, const int tid) {
    constexpr int NF = (MODE == X_FUSED_ADJ) ? 2 : 1;
    constexpr int HP = T / 2;                       // line pairs
    constexpr int NB = NF * 3 * HP;
    constexpr XLayout<L, NB, (MODE == X_FUSED_FWD || MODE == X_FUSED_ADJ)> ix{};
    const size_t plane = (size_t)g.G * g.Gzl;       // local (y,z) points
    // b = (f*3 + c)*HP + p
    auto line_ok = [&](int p) { return i0 + 2 * p < plane; };      // plane is even, T is even: pairs never straddle the end
    // offset in Ty of (component c, mode kx, line pair p of this tile): tiles of up to 8 points lie inside one z block of either layout
    const XOrigin xo = x_origin(i0, g);
    auto spec_off = [&](int c, int kx, int p) -> size_t {
        if constexpr (T <= 8) return xo.base + c * xo.cs + kx * xo.ks + 2 * p;
        else return ty_off(c, kx, i0 + 2 * p, g);
    };

    // -- load (spectrum, Hermitian extension) or (grid, two real lines) ------------------------------------------
    auto ld_spec = [&](int b, int pos) -> cplx {
        const int p = b % HP, fc = b / HP, c = fc % 3, f = fc / 3;
        if (!line_ok(p)) return mk(0, 0);
        const cplx* src = (f == 0) ? sp.inA : sp.inB;
        int kx; bool cj;
        if (pos < g.a) { kx = pos; cj = false; }
        else if (pos > g.G - g.a) { kx = g.G - pos; cj = true; }
        else return mk(0, 0);
        const size_t off = spec_off(c, kx, p);
        cplx X1 = src[off], X2 = src[off + 1];
        if (kx == 0) return mk(X1.re, X2.re);                       // c2r ignores the imaginary part of kx = 0
        if (cj) return mk(X1.re + X2.im, X2.re - X1.im);            // conj(X1) + i conj(X2)
        return mk(X1.re - X2.im, X1.im + X2.re);                    // X1 + i X2
    };
    auto ld_grid = [&](int b, int pos) -> cplx {
        const int p = b % HP, c = b / HP;
        if (!line_ok(p)) return mk(0, 0);
        const double* q = gridU + grid_off(c, pos, i0 + 2 * p, g);
        return mk(q[0], q[1]);
    };
    auto st_buf = [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; };

    if (MODE == X_TO_GRID) {
        fft_inplace_ix<L, true, NB, NT, true, false, false>(buf, ix, tw, tid, ld_spec, [&](int b, int pos, cplx v) {
            const int p = b % HP, c = b / HP;
            if (line_ok(p)) {
                double* q = gridOut + (g.utile ? u_off(c, pos, i0 + 2 * p, g) : grid_off(c, pos, i0 + 2 * p, g));
                q[0] = v.re; q[1] = v.im;
            }
        });
        return;
    }
    constexpr bool ACC = (MODE == X_FUSED_ADJ);
    constexpr int SCNT = ((L / 3) * NF * 3 * HP + NT - 1) / NT;        // items per thread of the final split / store loop
    cplx old_sum[ACC ? SCNT : 1][2];
    if (MODE == X_FROM_GRID) {
        fft_inplace_ix<L, false, NB, NT, true, false, true>(buf, ix, tw, tid, ld_grid, st_buf);
    } else {
        // The middle of the pass runs in registers.  G = 3N/2, so the inverse transform (radix order 4..4[2]3) ENDS with a radix-3
        // butterfly over x = j, j + a, j + 2a (a = L/3), and the forward transform, taken in the order 3 4..4[2], BEGINS with a radix-3
        // butterfly over the same three points.  A thread that owns (j, line pair p) for every component therefore does: last inverse
        // butterflies -> cross product(s) at its three grid points -> first forward butterflies (+ twiddles), touching the LDS once
        // (read j + k a, write 3 j + k) with one barrier in between, instead of three round trips and three barriers.  The velocity is
        // requested before that barrier and arrives while the others finish reading.  Item = (j, p), p fastest.
        // Radix policy of the fused FORWARD pass: butterflies up to radix 8 (192 = 8*8*3, 384 = 8*8*2*3: one LDS round trip and one barrier less per
        // direction; 117-120 VGPRs, still four waves per SIMD): -3 % at G = 192, -5 % at G = 384.  The adjoint passes (no registers to spare) and the
        // z / y passes get slower with it and keep radix 4.
#ifndef SMO_X_FWD_RADIX
#define SMO_X_FWD_RADIX 8
#endif
        constexpr int XRH = (MODE == X_FUSED_FWD) ? SMO_X_FWD_RADIX : (MODE == X_FUSED_ADJ ? SMO_X_SEQ_HEAD_RADIX : SMO_FFT_MAX_RADIX);      // inverse head
        constexpr int XRT = (MODE == X_FUSED_FWD) ? SMO_X_FWD_RADIX : (MODE == X_FUSED_ADJ ? SMO_X_SEQ_TAIL_RADIX : SMO_FFT_MAX_RADIX);      // forward tail
        static_assert(last_radix<L, XRH>() == 3, "G = 3N/2: the last Stockham stage is radix 3");
        // every stored mode feeds two positions of the Hermitian-extended line (kx and G - kx): read it once, coalesced (p fastest: one
        // 128-byte run per (component, kx)), and write both into the tile; the first butterfly stage then works LDS -> LDS
        for (int t = tid; t < (L / 3) * NF * 3 * HP; t += NT) {
            const int p = t % HP, r = t / HP, fc = r % (NF * 3), kx = r / (NF * 3), c = fc % 3, f = fc / 3;
            cplx X1 = mk(0, 0), X2 = mk(0, 0);
            if (line_ok(p)) {
                const cplx* src = ((f == 0) ? sp.inA : sp.inB) + spec_off(c, kx, p);
                // (forward pass, tiles of whole 128-byte lines: its input spectrum is dead once read — non-temporal, so that the hit does not renew it)
                constexpr bool NTIN = (SMO_X_NT & 16) != 0 && MODE == X_FUSED_FWD && T >= 8;
                X1 = ld_cplx<NTIN>(src); X2 = ld_cplx<NTIN>(src + 1);
            }
            const int b = fc * HP + p;
            if (kx == 0) buf[ix(b, 0)] = mk(X1.re, X2.re);                    // c2r ignores the imaginary part of kx = 0
            else {
                buf[ix(b, kx)] = mk(X1.re - X2.im, X1.im + X2.re);            // X1 + i X2
                buf[ix(b, L - kx)] = mk(X1.re + X2.im, X2.re - X1.im);        // conj(X1) + i conj(X2)
            }
        }
        constexpr int S3 = L / 3;
        constexpr int ICNT = (HP * S3 + NT - 1) / NT;
        cplx U0[3][3];                              // velocity at the first item's points, [k][component]
        // SMO_X_FWD_U_EARLY=1: the velocity is requested before the inverse transform's LDS stages instead of after them
#ifndef SMO_X_FWD_U_EARLY
#define SMO_X_FWD_U_EARLY 1
#endif
        constexpr bool UEARLY = SMO_X_FWD_U_EARLY && MODE == X_FUSED_FWD;
        if (UEARLY) {
            const int j = tid / HP, p = tid - j * HP;
            if (tid < HP * S3 && line_ok(p))
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    for (int c = 0; c < 3; ++c) U0[k][c] = ld_pair<(SMO_X_NT & 1) != 0>(gridU + u_off(c, j + S3 * k, i0 + 2 * p, g));
        }
        __syncthreads();
        fft_inplace_head_r<XRH, L, true, NB, NT, true, true>(buf, ix, tw, tid, [&](int b, int pos) -> cplx {
            return (pos >= L / 3 && pos <= L - L / 3) ? mk(0, 0) : buf[ix(b, pos)];        // the zero padding is never stored
        });
        cplx Win[ICNT][NF][3][3];                   // [item][field group][component][k]: inputs of the last inverse stage
#pragma unroll
        for (int i = 0; i < ICNT; ++i) {
            const int t = tid + i * NT, j = t / HP, p = t - j * HP;
            if (t < HP * S3 && line_ok(p)) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    for (int c = 0; c < 3; ++c)
                        for (int k = 0; k < 3; ++k) Win[i][f][c][k] = buf[ix((f * 3 + c) * HP + p, j + S3 * k)];
                if (i == 0 && !UEARLY)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        for (int c = 0; c < 3; ++c) {
                            U0[k][c] = ld_pair<(SMO_X_NT & 1) != 0>(gridU + u_off(c, j + S3 * k, i0 + 2 * p, g));
                        }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ICNT; ++i) {
            const int t = tid + i * NT, j = t / HP, p = t - j * HP;
            if (t >= HP * S3 || !line_ok(p)) continue;
            cplx U[3][3], W[3][3];                  // [x = j + k a][component]; .re / .im = the two real lines of the pair
#pragma unroll
            for (int k = 0; k < 3; ++k)
                for (int c = 0; c < 3; ++c) {
                    if (i == 0) U[k][c] = U0[k][c];
                    else U[k][c] = ld_pair<(SMO_X_NT & 1) != 0>(gridU + u_off(c, j + S3 * k, i0 + 2 * p, g));
                }
            auto last_stage = [&](int f, cplx (&out)[3][3]) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    cplx v[3];
                    for (int k = 0; k < 3; ++k) v[k] = Win[i][f][c][k];
                    Butterfly<3, true>::run(v);
                    for (int k = 0; k < 3; ++k) out[k][c] = v[k];
                }
            };
            auto cross_to = [&](int f, const cplx (&X)[3][3], const cplx (&Y)[3][3]) {       // field group f <- first forward stage of X x Y
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                    cplx v[3];
                    for (int k = 0; k < 3; ++k)
                        v[k] = mk(X[k][c1].re * Y[k][c2].re - X[k][c2].re * Y[k][c1].re, X[k][c1].im * Y[k][c2].im - X[k][c2].im * Y[k][c1].im);
                    Butterfly<3, false>::run(v);
                    const int b = (f * 3 + c) * HP + p;
                    buf[ix(b, 3 * j)] = v[0];
                    buf[ix(b, 3 * j + 1)] = twmul<false>(v[1], tw[j]);
                    buf[ix(b, 3 * j + 2)] = twmul<false>(v[2], tw[2 * j]);
                }
            };
            last_stage(0, W);
            if (MODE == X_FUSED_FWD) {
                cross_to(0, U, W);                  // EMF = U x B
            } else {
                cross_to(0, W, U);                  // F1 = omega x U
                last_stage(1, U);                   // B_f takes the velocity's registers
                cross_to(1, W, U);                  // F2' = omega x B_f
            }
        }
        __syncthreads();
        // remaining forward stages (sub-length a, stride 3); group B of the adjoint pass is a running sum: its old values are requested
        // before the last stage
        InplaceTail<L, L / 3, 3, false, NB, NT, true, true, XRT>::run_ix(buf, ix, tid, tw, st_buf, [&]() {
            if (ACC) {
#pragma unroll
                for (int i = 0; i < SCNT; ++i) {
                    const int t = tid + i * NT, p = t % HP, r = t / HP, fc = r % (NF * 3), kx = r / (NF * 3);
                    if (t < (L / 3) * NF * 3 * HP && fc >= 3 && line_ok(p)) {
                        const cplx* q = sp.outB + spec_off(fc - 3, kx, p);
                        old_sum[i][0] = q[0]; old_sum[i][1] = q[1];
                    }
                }
            }
        });
    }
    __syncthreads();
    // split the two real lines' spectra and store kx = 0..a-1 (a = L/3); item = ((kx*NF*3 + fc)*HP + p), p fastest
#pragma unroll
    for (int i = 0; i < SCNT; ++i) {
        const int t = tid + i * NT;
        if (t >= (L / 3) * NF * 3 * HP) break;
        const int p = t % HP, r = t / HP, fc = r % (NF * 3), kx = r / (NF * 3);
        if (!line_ok(p)) continue;
        const int c = fc % 3, f = fc / 3;
        const cplx Zk = buf[ix(fc * HP + p, kx)];
        const cplx Zm = conj(buf[ix(fc * HP + p, (kx == 0) ? 0 : L - kx)]);
        cplx* dst = ((f == 0) ? sp.outA : sp.outB) + spec_off(c, kx, p);
        cplx v0 = 0.5 * (Zk + Zm), v1 = mul_mi(0.5 * (Zk - Zm));
        if (ACC && f == 1) { v0 = v0 + old_sum[i][0]; v1 = v1 + old_sum[i][1]; }
        dst[0] = v0;
        dst[1] = v1;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Adjoint x pass, field groups ONE AFTER THE OTHER through the tile buffer (X_FUSED_ADJ_SEQ).  The pass needs omega = curl G and B_f on
// the grid at the same points; X_FUSED_ADJ transforms both at once (6 * HP transforms in the tile), which at a given LDS footprint halves
// the (y,z) points per tile: 64-byte runs of the spectra at G = 192, 32-byte runs at G = 384 — and the access pattern itself then caps the
// pass (tools/micro_gather.hip: a bare copy with that pattern reaches 5.1 / 3.6 TB/s, with runs twice as long 6.2 / 4.6).  Here omega
// goes first: inverse transform, last radix-3 stage in registers, and its grid values STAY in registers (9 complex per item) while
//   F1 = omega x U    is formed, transformed forward through the same buffer and stored, then
//   B_f               is staged, transformed, and F2' = omega x B_f goes forward and is added to the running sum.
// Same arithmetic per element as X_FUSED_ADJ, 3 * HP transforms in the tile => tiles of twice as many points (128 / 64-byte runs).
// ---------------------------------------------------------------------------------------------------------
template <int L, int T, int NT, class TW>
__device__ __forceinline__ void x_tile_adj_seq(const XSpec& sp, const double* __restrict__ gridU, const Geom& g, cplx* buf, const TW tw,
                                               const size_t i0, const int tid) {
    constexpr int HP = T / 2, NB = 3 * HP, S3 = L / 3;
    constexpr XLayout<L, NB, true> ix{};
    constexpr int NITEM = S3 * 3 * HP;                         // stored modes of one field group = items of the staging / split loops
    constexpr int SCNT = (NITEM + NT - 1) / NT;
    constexpr int ICNT = (HP * S3 + NT - 1) / NT;              // middle-section items (j, p) per thread
    static_assert(last_radix<L, SMO_X_SEQ_HEAD_RADIX>() == 3, "G = 3N/2: the last Stockham stage is radix 3");
    const size_t plane = (size_t)g.G * g.Gzl;
    auto line_ok = [&](int p) { return i0 + 2 * p < plane; };
    auto st_buf = [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; };
    static_assert(T <= 8, "a tile lies inside one z block of Ty");
    const XOrigin xo = x_origin(i0, g);
    auto spec_off = [&](int c, int kx, int p) -> size_t { return xo.base + c * xo.cs + kx * xo.ks + 2 * p; };

    // SMO_X_SEQ_PREFETCH = n: the first n (of SCNT) spectral items of B_f per thread are requested before the last forward stage of F1 instead
    // of after F1's store loop: part of one global-memory round trip leaves the tile's critical path.  Registers decide how many: omega's grid
    // values are live there, unlike in the second half where the running sum is requested at the same place.
#ifndef SMO_X_SEQ_PREFETCH
#define SMO_X_SEQ_PREFETCH (!only_2_and_3(L) ? 0 : L > 192 ? 3 : 2)     // lengths with a radix-5 / radix-7 stage have no registers to spare
#endif
    constexpr int PRE = (SMO_X_SEQ_PREFETCH) < SCNT ? (SMO_X_SEQ_PREFETCH) : SCNT;
    cplx pre_in[PRE > 0 ? PRE : 1][2];
    auto request_in = [&](const cplx* src, auto ntl) {         // the first PRE loads of stage_in, issued early
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int t = tid + i * NT, p = t % HP, r = t / HP, c = r % 3, kx = r / 3;
            pre_in[i][0] = mk(0, 0); pre_in[i][1] = mk(0, 0);
            if (t < NITEM && line_ok(p)) { const cplx* q = src + spec_off(c, kx, p); pre_in[i][0] = ld_cplx<decltype(ntl)::value>(q); pre_in[i][1] = ld_cplx<decltype(ntl)::value>(q + 1); }
        }
    };
    auto stage_in = [&](const cplx* src, auto ntl, auto requested, auto&& before_head) {           // spectra of one field group -> Hermitian-extended lines in the tile
#pragma unroll
        for (int i = 0; i < SCNT; ++i) {
            const int t = tid + i * NT;
            if (t >= NITEM) break;
            const int p = t % HP, r = t / HP, c = r % 3, kx = r / 3;
            cplx X1 = mk(0, 0), X2 = mk(0, 0);
            if (decltype(requested)::value && i < PRE) { X1 = pre_in[i][0]; X2 = pre_in[i][1]; }
            else if (line_ok(p)) { const cplx* q = src + spec_off(c, kx, p); X1 = ld_cplx<decltype(ntl)::value>(q); X2 = ld_cplx<decltype(ntl)::value>(q + 1); }
            const int b = c * HP + p;
            if (kx == 0) buf[ix(b, 0)] = mk(X1.re, X2.re);
            else {
                buf[ix(b, kx)] = mk(X1.re - X2.im, X1.im + X2.re);
                buf[ix(b, L - kx)] = mk(X1.re + X2.im, X2.re - X1.im);
            }
        }
        before_head();
        __syncthreads();
        fft_inplace_head_r<SMO_X_SEQ_HEAD_RADIX, L, true, NB, NT, true, true>(buf, ix, tw, tid, [&](int b, int pos) -> cplx {
            return (pos >= L / 3 && pos <= L - L / 3) ? mk(0, 0) : buf[ix(b, pos)];
        });
    };
    auto read_win = [&](cplx (&W)[ICNT][3][3]) {                // [item][component][k]: inputs of the last inverse stage
#pragma unroll
        for (int i = 0; i < ICNT; ++i) {
            const int t = tid + i * NT, j = t / HP, p = t - j * HP;
            if (t < HP * S3 && line_ok(p))
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < 3; ++k) W[i][c][k] = buf[ix(c * HP + p, j + S3 * k)];
        }
    };
    auto cross_first = [&](const cplx (&X)[3][3], const cplx (&Y)[3][3], int j, int p) {      // X x Y at the item's three points -> first forward stage
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
            cplx v[3];
            for (int k = 0; k < 3; ++k)
                v[k] = mk(X[c1][k].re * Y[c2][k].re - X[c2][k].re * Y[c1][k].re, X[c1][k].im * Y[c2][k].im - X[c2][k].im * Y[c1][k].im);
            Butterfly<3, false>::run(v);
            const int b = c * HP + p;
            buf[ix(b, 3 * j)] = v[0];
            buf[ix(b, 3 * j + 1)] = twmul<false>(v[1], tw[j]);
            buf[ix(b, 3 * j + 2)] = twmul<false>(v[2], tw[2 * j]);
        }
    };
    // SMO_X_SEQ_LATE_SUM=1 (experiment): read the running sum in the store loop instead of requesting it before the last stage (24 VGPRs less
    // at the register peak, the latency no longer hidden behind that stage)
#ifndef SMO_X_SEQ_LATE_SUM
#define SMO_X_SEQ_LATE_SUM 0
#endif
    cplx old_sum[SMO_X_SEQ_LATE_SUM ? 1 : SCNT][2];
    // SMO_X_SEQ_REQ_EARLY=1: these requests (B_f's spectra in the first half, the running sum in the second) are issued before the whole forward
    // tail instead of before its last stage (G = 192: -1.3 %; at G = 384 the longer live ranges spill: +18 %)
    // (parking part of omega's grid copy in the CU's idle LDS to make room for these at G = 384 was measured in round 4 and does not pay:
    // profiles/r04_x384_levers.txt — the requested values compete with the radix-8 tail stages for the same registers, and those are worth more)
#ifndef SMO_X_SEQ_REQ_EARLY
#define SMO_X_SEQ_REQ_EARLY (only_2_and_3(L) && L <= 192)
#endif
    auto forward_and_store = [&](cplx* dst, const bool acc) {
        auto requests = [&]() {
            if (!acc && PRE > 0) request_in(sp.inB, std::integral_constant<bool, (SMO_X_NT & 2) != 0>());
            if (acc && !SMO_X_SEQ_LATE_SUM) {
#pragma unroll
                for (int i = 0; i < SCNT; ++i) {
                    const int t = tid + i * NT, p = t % HP, r = t / HP, c = r % 3, kx = r / 3;
                    if (t < NITEM && line_ok(p)) {
                        const cplx* q = dst + spec_off(c, kx, p);
                        old_sum[i][0] = ld_cplx<(SMO_X_NT & 4) != 0>(q); old_sum[i][1] = ld_cplx<(SMO_X_NT & 4) != 0>(q + 1);
                    }
                }
            }
        };
        if (SMO_X_SEQ_REQ_EARLY) requests();
        InplaceTail<L, L / 3, 3, false, NB, NT, true, true, SMO_X_SEQ_TAIL_RADIX>::run_ix(buf, ix, tid, tw, st_buf, [&]() { if (!SMO_X_SEQ_REQ_EARLY) requests(); });
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SCNT; ++i) {
            const int t = tid + i * NT;
            if (t >= NITEM) break;
            const int p = t % HP, r = t / HP, c = r % 3, kx = r / 3;
            if (!line_ok(p)) continue;
            const cplx Zk = buf[ix(c * HP + p, kx)];
            const cplx Zm = conj(buf[ix(c * HP + p, (kx == 0) ? 0 : L - kx)]);
            cplx* q = dst + spec_off(c, kx, p);
            cplx v0 = 0.5 * (Zk + Zm), v1 = mul_mi(0.5 * (Zk - Zm));
            if (acc && SMO_X_SEQ_LATE_SUM) { v0 = v0 + ld_cplx<(SMO_X_NT & 4) != 0>(q); v1 = v1 + ld_cplx<(SMO_X_NT & 4) != 0>(q + 1); }
            else if (acc) { v0 = v0 + old_sum[i][0]; v1 = v1 + old_sum[i][1]; }
            if ((SMO_X_NT & 8) && acc) {                           // the running sum comes back one whole step later
                __builtin_nontemporal_store(d2_t{v0.re, v0.im}, reinterpret_cast<d2_t*>(q));
                __builtin_nontemporal_store(d2_t{v1.re, v1.im}, reinterpret_cast<d2_t*>(q + 1));
            } else {
                q[0] = v0;
                q[1] = v1;
            }
        }
    };

    // ---- omega: spectrum -> grid, kept in registers -------------------------------------------------------------------------
    cplx Wom[ICNT][3][3], Wy[ICNT][3][3];
    // SMO_X_SEQ_U_EARLY=1: the velocity is requested before omega's inverse transform (in flight during its LDS stages; omega's grid values are
    // not live yet) instead of after it
#ifndef SMO_X_SEQ_U_EARLY
#define SMO_X_SEQ_U_EARLY only_2_and_3(L)
#endif
    auto request_U = [&]() {
#pragma unroll
        for (int i = 0; i < ICNT; ++i) {
            const int t = tid + i * NT, j = t / HP, p = t - j * HP;
            if (t < HP * S3 && line_ok(p))
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < 3; ++k) Wy[i][c][k] = ld_pair<(SMO_X_NT & 1) != 0>(gridU + u_off(c, j + S3 * k, i0 + 2 * p, g));
        }
    };
    stage_in(sp.inA, std::integral_constant<bool, (SMO_X_NT & 32) != 0>(), std::false_type(), [&]() { if (SMO_X_SEQ_U_EARLY) request_U(); });      // omega's spectrum is dead once read
    read_win(Wom);
    if (!SMO_X_SEQ_U_EARLY) request_U();                        // requested before the barrier: in flight while the others read
    __syncthreads();
    // ---- F1 = omega x U -> forward -> out A --------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < ICNT; ++i) {
        const int t = tid + i * NT, j = t / HP, p = t - j * HP;
        if (t >= HP * S3 || !line_ok(p)) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) Butterfly<3, true>::run(Wom[i][c]);
        cross_first(Wom[i], Wy[i], j, p);
    }
    __syncthreads();
    forward_and_store(sp.outA, false);
    __syncthreads();
    // ---- B_f: spectrum -> grid; F2' = omega x B_f -> forward -> running sum (out B) -----------------------------------------------
    stage_in(sp.inB, std::integral_constant<bool, (SMO_X_NT & 2) != 0>(), std::integral_constant<bool, (PRE > 0)>(), []() {});
    read_win(Wy);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ICNT; ++i) {
        const int t = tid + i * NT, j = t / HP, p = t - j * HP;
        if (t >= HP * S3 || !line_ok(p)) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) Butterfly<3, true>::run(Wy[i][c]);
        cross_first(Wom[i], Wy[i], j, p);
    }
    __syncthreads();
    forward_and_store(sp.outB, true);
}

// One tile per workgroup.  PAIRED = P > 1: the tile is narrower than a 128-byte line of the spectra (T = 8/P complex), so the P
// tiles of a line are given to workgroups b, b+8, ..., which the dispatcher places on the same XCD at about the same time: the
// rest of every line is then served by that XCD's L2 instead of being fetched from HBM again (speed only, never correctness).
template <int L, int MODE, int T, int NT, int PAIRED = 0>         // PAIRED = tiles per 128-byte line of the spectra (0/1: no remapping)
#ifndef SMO_X_WAVES
#define SMO_X_WAVES 3
#endif
#ifndef SMO_Y_ZT
#define SMO_Y_ZT 8        // y pass: z columns per workgroup = one 128-byte line; 25 KB (G = 192) / 49 KB (G = 384) of LDS => 5 / 3 workgroups per CU
#endif
#ifndef SMO_X_FWD_NT
#define SMO_X_FWD_NT 256
#endif
#ifndef SMO_X_ADJ_NT
#define SMO_X_ADJ_NT 192
#endif
#ifndef SMO_X_SEQ_WAVES
#define SMO_X_SEQ_WAVES SMO_X_WAVES
#endif
#ifndef SMO_X_SEQ_NT
#define SMO_X_SEQ_NT SMO_X_FWD_NT      // threads of the sequential adjoint pass (experiments: 320, 384)
#endif
// (a sequential adjoint tile with more than one middle-section item per thread — SMO_X_SEQ_T_BIG = 8 — is built for two waves per SIMD: 256 VGPRs)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(MODE == X_FUSED_ADJ_SEQ ? ((T / 2) * (L / 3) > NT ? 2 : SMO_X_SEQ_WAVES) : SMO_X_WAVES))) void kd_x_pass(XSpec sp, const double* __restrict__ gridU, double* gridOut,
                                                const cplx* __restrict__ tw_g, Geom g) {
    constexpr int NB = ((MODE == X_FUSED_ADJ) ? 2 : 1) * 3 * (T / 2);
    __shared__ cplx buf[XLayout<L, NB, (MODE == X_FUSED_FWD || MODE == X_FUSED_ADJ || MODE == X_FUSED_ADJ_SEQ)>::ELEMS];
    // G > 192: half twiddle table — with the full one (6 KB at G = 384) the tile fits only three times into a CU's LDS instead of four
    constexpr bool HALF = (L > 192);
    constexpr int NTW = HALF ? L / 2 : L;
    __shared__ cplx tw_s[NTW];
    const int tid = threadIdx.x;
    for (int i = tid; i < NTW; i += NT) tw_s[i] = tw_g[i];
    // SMO_X_TW_NOSYNC=1: no barrier here — the fused tiles stage their spectra into the tile buffer first and put a barrier behind that loop
    // before the first butterfly reads a twiddle, so the table's loads and the spectra's are in flight together
#ifndef SMO_X_TW_NOSYNC
#define SMO_X_TW_NOSYNC 1
#endif
    if (!(SMO_X_TW_NOSYNC && (MODE == X_FUSED_FWD || MODE == X_FUSED_ADJ || MODE == X_FUSED_ADJ_SEQ))) __syncthreads();
    size_t tile = blockIdx.x;
    if (PAIRED > 1 && blockIdx.x < (gridDim.x / (8 * PAIRED)) * (8 * PAIRED)) {
        const unsigned q = blockIdx.x / (8 * PAIRED), r = blockIdx.x % (8 * PAIRED);
        tile = (size_t)q * (8 * PAIRED) + PAIRED * (r % 8) + r / 8;
    }
    size_t i0 = tile * T;
    if constexpr (T <= 8) {
        if (g.tyl) {
            // z-block-major Ty: walk the tiles y-fastest inside a z block (the 8/T tiles of a 128-byte line first, then y, then the block):
            // consecutive workgroups then read consecutive lines of every kx block (a tile is independent of all others: any order is valid)
            constexpr unsigned PER = 8 / T;
            const size_t line = tile / PER, sub = tile - line * PER;
            const size_t zb = line / g.G, y = line - zb * g.G;
            i0 = y * g.Gzl + zb * 8 + sub * T;
        }
    }
    if constexpr (MODE == X_FUSED_ADJ_SEQ) {
        if constexpr (HALF) x_tile_adj_seq<L, T, NT>(sp, gridU, g, buf, HalfTwiddles{tw_s, L / 2}, i0, tid);
        else x_tile_adj_seq<L, T, NT>(sp, gridU, g, buf, (const cplx*)tw_s, i0, tid);
    } else {
        if constexpr (HALF) x_tile<L, MODE, T, NT>(sp, gridU, gridOut, g, buf, HalfTwiddles{tw_s, L / 2}, i0, tid);
        else x_tile<L, MODE, T, NT>(sp, gridU, gridOut, g, buf, (const cplx*)tw_s, i0, tid);
    }
}

// ---------------------------------------------------------------------------------------------------------
// small per-mode kernels
// ---------------------------------------------------------------------------------------------------------
// compatibility condition: G = P(-2 B_N) / (dt alpha)  (Final)  |  / alpha (Integrated);  continuous: G = -2 B_N
__global__ void kd_terminal(const cplx* __restrict__ BN, cplx* __restrict__ Gh, cplx* __restrict__ nu, Geom g, int integrated, int continuous) {
    const size_t nmode = (size_t)g.al * g.m * g.m;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < nmode; e += (size_t)gridDim.x * blockDim.x) {
        const int iz = e % g.m, iy = (e / g.m) % g.m, ixl = e / ((size_t)g.m * g.m);
        const double k[3] = {(double)(g.ix0 + ixl), wavenumber(iy, g), wavenumber(iz, g)};
        const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
        cplx r[3];
        for (int c = 0; c < 3; ++c) { cplx b = BN[c * nmode + e]; r[c] = mk(-2.0 * b.re, -2.0 * b.im); nu[c * nmode + e] = mk(0, 0); }
        if (!continuous) {
            if (k2 == 0.0) { for (int c = 0; c < 3; ++c) r[c] = mk(0, 0); }
            else {
                const double alpha = 1.0 / g.dt + 0.5 * k2 / g.Rm, s = integrated ? alpha : g.dt * alpha, ik2 = 1.0 / k2;
                const cplx kr = mk((k[0] * r[0].re + k[1] * r[1].re + k[2] * r[2].re) * ik2, (k[0] * r[0].im + k[1] * r[1].im + k[2] * r[2].im) * ik2);
                for (int c = 0; c < 3; ++c) r[c] = mk((r[c].re - k[c] * kr.re) / s, (r[c].im - k[c] * kr.im) / s);
            }
        }
        for (int c = 0; c < 3; ++c) Gh[c * nmode + e] = r[c];
    }
}

// spectral energy  sum_k w |B^|^2  (w = 1 on the kx = 0 plane, 2 elsewhere) == grid mean of |B|^2 ; one partial per workgroup
__global__ __launch_bounds__(256) void kd_energy(const cplx* __restrict__ Bh, double* __restrict__ partial, Geom g) {
    __shared__ double red[4];
    const size_t nmode = (size_t)g.al * g.m * g.m, plane = (size_t)g.m * g.m;
    double acc = 0.0;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < nmode; e += (size_t)gridDim.x * 256) {
        const int ixl = e / plane;
        const double w = (g.ix0 + ixl == 0) ? 1.0 : 2.0;
        double s = 0.0;
        for (int c = 0; c < 3; ++c) { cplx b = Bh[c * nmode + e]; s += b.re * b.re + b.im * b.im; }
        acc += w * s;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void kd_dot(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ partial, size_t n) {
    __shared__ double red[4];
    double acc = 0.0;
    const size_t n2 = n / 2;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const double2* y2 = reinterpret_cast<const double2*>(y);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        double2 a = x2[i], b = y2[i];
        acc += a.x * b.x + a.y * b.y;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) acc += x[n - 1] * y[n - 1];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------------
// the same passes with the transform length a run-time value: every even Npts without a tuned instantiation
// ---------------------------------------------------------------------------------------------------------
#include "kdyn_any.hpp"

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
constexpr int NPART = 1024;         // workgroups (= partial sums) of the reduction kernels

class KDyn : public Context {
public:
    explicit KDyn(const smo_config& c) { cfg = c; }
    Geom g{};
    size_t nmode = 0, tzb = 0, n_ex = 0, n_grid = 0, fld = 0;
    // tzb = 3*al*m*Gzr: one field group of Tz, per peer;  n_ex = 2*W*tzb: a whole exchange buffer (two field groups, all peers)
    // fld = 3*a*G*Gzr:  one field group of Ty (all kx, local z planes) = what the x pass reads
    // The rank's z slab can be cut into K equal chunks (SMO_KD_SET_CHUNKS): the grid-side phases then work chunk by chunk and every
    // chunk of an exchange buffer is contiguous, so the host layer can overlap the exchange of one chunk with the work on another.
    int K = 1;
    size_t tzc = 0, fldc = 0, ngc = 0;       // per chunk: tzb / K, 3*a*typ, n_grid / K
    static constexpr int TY_KMAX = 8;
    size_t ty_pad = 0;
    cplx *d_stack = nullptr, *d_ty = nullptr, *d_G = nullptr, *d_nu = nullptr, *d_tw = nullptr;
    cplx *zs = nullptr, *ys = nullptr;      // Tz exchange buffers: z-pass side / y-pass side (one and the same when world == 1)
    double *d_U = nullptr, *d_part = nullptr;
    std::vector<double> h_part;
    size_t n_part_rows = 1;
    int k_zi = -1, k_zic = -1, k_yi = -1, k_yf = -1, k_xf = -1, k_xa = -1, k_zfu = -1, k_zfa = -1, k_misc = -1, k_ex = -1, k_dot = -1;

    // snapshot n: every ck-th state is kept in the stack, the others live in (ck-1) scratch slots that hold ONE window at a time.
    // Dense tail (round 4): from index dense_from (a multiple of ck) on EVERY state is kept — the HBM that a uniform interval leaves unused
    // (256^3 x 1000 steps on one GPU: interval 2 = 200 GB of 288) buys that many windows less to recompute.  A state of the tail costs its
    // adjoint step the z and y passes of the snapshot (no grid-side copy is kept for it), a recomputed window a whole forward step per ck
    // adjoint steps plus the grid-side form of its last state: ~0.58 against ~1.07 ms per adjoint step at 256^3.
    int ck = 1, scratch_window = -1;
    int dense_from = 1 << 30;
    cplx* d_scratch = nullptr;
    cplx* snap(int n) {
        if (n >= dense_from) return d_stack + ((size_t)(dense_from / ck) + (size_t)(n - dense_from)) * 3 * nmode;
        const int r = n % ck;
        return r == 0 ? d_stack + (size_t)(n / ck) * 3 * nmode : d_scratch + (size_t)(r - 1) * 3 * nmode;
    }
    size_t stack_slots() const {
        return dense_from > cfg.n_iters ? (size_t)cfg.n_iters / ck + 1 : (size_t)(dense_from / ck) + (size_t)(cfg.n_iters - dense_from) + 1;
    }
    size_t stack_elems(int k) const { return ((size_t)cfg.n_iters / k + 1 + (size_t)(k - 1)) * 3 * nmode; }
    // make state `idx` available (recompute its window from the preceding checkpoint if it is not resident)
    int ensure(int idx) {
        if (idx >= dense_from || idx % ck == 0 || scratch_window == idx / ck) return SMO_OK;
        const int w = idx / ck, last = std::min(cfg.n_iters, w * ck + ck - 1);
        scratch_window = w;
        // the forward steps that rebuild the window pass through the grid-side form Ty(B^_n) of the states they start from: keep it
        // (one slot per step of the window), so that the adjoint steps of those states neither transform their snapshot again nor
        // send it — for ck = 2 that is every second adjoint step: two kernels and one exchanged field group less
        tycache_window = d_tycache ? w : -1;
        tycache_count = last - w * ck;
        for (int n = w * ck; n < last; ++n) SMO_TRY(step_fwd(n, d_tycache != nullptr));
        if (d_tycache && last < cfg.n_iters) {
            // ... and of the last state of the window: its inverse z pass is already in the exchange buffer (left by the update that
            // produced it), one y pass puts it beside the others — every adjoint step of the window then runs with one field group
            tycache_count += 1;
            SMO_TRY(fwd_A(last));
            for (int k = 0; k < K; ++k) { SMO_TRY(exchange(true, 1, k, stream)); SMO_TRY(y_pass(true, 0, 1, tyslot(last, k), k)); }
        }
        return SMO_OK;
    }
    Geom geom(int nfields, int k = 0) const { Geom q = g; q.blk = (size_t)nfields * tzc; q.cblk = (size_t)cfg.world * 2 * tzc; q.zg0 = k * g.Gzl; return q; }
    // Layout of Ty for the current chunk shape: z-block major (Geom::tyl, ty_off) when the local z planes come in whole blocks of 8 and the
    // tuned kernels run; SMO_KD_TYL = 0 / 1 forces the plane / z-block layout (1 is ignored where it cannot apply), SMO_KD_YKX = 0 keeps the
    // y pass's workgroups z-fastest.  fldc = elements of one field group of one chunk in whichever layout is larger.
    int ty_layout_env = -1, ykx_env = -1;
    void pick_ty_layout() {
        const bool can = !any_size && g.Gzl % 8 == 0 && K <= TY_KMAX;
        const bool want = ty_layout_env >= 0 ? ty_layout_env == 1 : g.G >= ty_layout_min_g;
        g.tyl = (can && want) ? 1 : 0;
        g.ykx = ykx_env >= 0 ? ykx_env : 1;      // (y_pass picks per direction)
        g.tyk = (size_t)g.G * 8 + (ty_pad ? ty_pad : 8);
        const size_t zb_elems = (size_t)(g.Gzl / 8) * g.tyk;          // per (component, kx): all its z blocks
        fldc = (size_t)3 * g.a * (g.tyl ? std::max(g.typ, zb_elems) : g.typ);
    }
    int ty_layout_min_g = 0;                     // SMO_KD_TYL_MING: smallest grid that takes the z-block layout by default
    int set_chunks(int k) {
        const int Gzc = k > 0 ? g.Gzr / k : 0;
        // (one chunk of one slab may have any shape: odd G runs the run-time-length kernels, which pair its last line with zeros)
        if (k < 1 || g.Gzr % k != 0 || ((k > 1 || cfg.world > 1) && ((Gzc & 1) || ((size_t)g.G * Gzc) % 4 != 0))) {
            set_error("KDYN: %d chunks do not divide the %d local z planes into even parts with G*Gz %% 4 == 0", k, g.Gzr);
            return SMO_ERR_ARG;
        }
        K = k; g.Gzl = Gzc; tzc = tzb / K; ngc = n_grid / K;
        g.typ = (size_t)g.G * Gzc + (K <= TY_KMAX ? ty_pad : 0);
        pick_ty_layout();
        have_forward = false; zs_ready_fwd = zs_ready_adj = -1;
        return SMO_OK;
    }

    int init() override {
        const int N = cfg.npts, W = cfg.world;
        if (cfg.batch != 1) { set_error("KDYN: batch must be 1"); return SMO_ERR_ARG; }
        // Tuned kernels: the transform lengths G = 3N/2 with factors 2, 3, 5 and 7 instantiated below (with_L).  Every other even Npts — the
        // reference, through FFTW, takes any (FWD_Solve_KDyn.py:362-450, :1029) — runs the run-time-length kernels of kdyn_any.hpp
        // (SMO_KD_ANY=1 forces them at a tuned size too: how the tests compare the two paths).
        static const int sizes[] = {8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 60, 64, 72, 80, 96, 100, 112, 120, 128, 144, 160, 192, 200, 224, 240, 256, 320};
        if (N < 6 || (N & 1)) { set_error("KDYN: npts must be even and >= 6 (got %d)", N); return SMO_ERR_UNSUPPORTED; }
        any_size = std::find(std::begin(sizes), std::end(sizes), N) == std::end(sizes);
        { const char* e = getenv("SMO_KD_ANY"); if (e && atoi(e) == 1) any_size = true; }
        if ((N / 2) % W != 0 || (3 * N / 2) % W != 0 || (W > 1 && ((3 * N / 2 / W) * (3 * N / 2)) % 4 != 0)) {
            set_error("KDYN: %d slabs do not divide a=%d kx modes and G=%d grid planes", W, N / 2, 3 * N / 2);
            return SMO_ERR_UNSUPPORTED;
        }
        if (!any_size && ((3 * N / 2) * (3 * N / 2)) % 4 != 0) any_size = true;      // (cannot happen for the sizes above: G is even there)
        g.a = N / 2; g.al = g.a / W; g.ix0 = cfg.rank * g.al; g.m = N - 1; g.kmax = (N - 1) / 2; g.G = 3 * N / 2; g.Gzl = g.Gzr = g.G / W; g.W = W; g.zg0 = 0;
        g.Rm = cfg.param; g.dt = cfg.dt;
        nmode = (size_t)g.al * g.m * g.m;
        tzb = (size_t)3 * g.al * g.m * g.Gzl;
        n_ex = 2 * tzb * W;                              // two field groups (adjoint) x peers
        // One 128-byte line of padding per (component, kx) plane of Ty.  The x pass gathers 3a runs per tile, one per plane; with the
        // natural stride 16*G*Gzl (a multiple of 64 KB at every supported size) they all fall on the same HBM channel: measured
        // 240 -> 205 us for the fused adjoint x pass at 128^3.  SMO_KD_TYPAD (elements, a multiple of 8) overrides it for tuning.
        { const char* e = getenv("SMO_KD_FUSE_NEXT"); fuse_next = !(e && atoi(e) == 0); }
        { const char* e = getenv("SMO_KD_ADJ_SEQ"); adj_seq = !(e && atoi(e) == 0); }
        { const char* e = getenv("SMO_PEER_CHAINED"); chain_ok = !(e && atoi(e) == 0); }
        if (any_size) {
            fuse_next = false;                               // the run-time-length update kernel has no fused next pass
            plan = any_plan(3 * N / 2);
            if (const char* e = getenv("SMO_KD_ANY_NT")) { const int v = atoi(e); if (v >= 64 && v <= 1024 && v % 64 == 0) any_nt = v; }
        }
        { const char* e = getenv("SMO_KD_GRAPH"); if (e) graph_mode = atoi(e); }
        { const char* e = getenv("SMO_KD_GRAPH_MAXG"); if (e) graph_max_g = atoi(e); }
        { const char* e = getenv("SMO_KD_TYPAD"); ty_pad = e ? (size_t)atoi(e) : 8; }
        if (ty_pad % 8 != 0) { set_error("KDYN: SMO_KD_TYPAD must be a multiple of 8 elements (one 128-byte line)"); return SMO_ERR_ARG; }
        g.typ = (size_t)g.G * g.Gzl + ty_pad;
        { const char* e = getenv("SMO_KD_TYL"); if (e) ty_layout_env = atoi(e); }
        { const char* e = getenv("SMO_KD_YKX"); if (e) ykx_env = atoi(e); }
        { const char* e = getenv("SMO_KD_TYL_MING"); if (e) ty_layout_min_g = atoi(e); }
        // one field group of Ty, all chunks: the planes with their padding, or the z blocks with theirs (G/8... Gzr/8 blocks of 8 G + pad per kx)
        fld = (size_t)3 * g.a * ((size_t)g.G * g.Gzl + std::max((size_t)TY_KMAX * ty_pad, (size_t)(g.Gzl / 8 + 1) * std::max<size_t>(ty_pad, 8)));
        n_grid = (size_t)3 * g.G * g.G * g.Gzl;          // local slab of a grid vector: [3][G][G][Gzl]
        g.blk = tzb;
        tzc = tzb; ngc = n_grid;
        pick_ty_layout();
        n_comp = 2;
        vec_len = n_grid;
        snapshot_doubles = 2 * 3 * nmode;
        SMO_TRY(base_init());
        if (any_size) {
            // LDS of one workgroup: 64 KB unless the narrowest tile of the adjoint x pass (3 buffers x 3 transforms + the twiddle table) needs more
            if ((size_t)160 * plan.L > any_lds) {
                int dev = 0, maxb = 0;
                SMO_HIP(hipGetDevice(&dev));
                SMO_HIP(hipDeviceGetAttribute(&maxb, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
                if ((size_t)160 * plan.L > (size_t)maxb) { set_error("KDYN: G = %d needs %zu bytes of LDS per workgroup (device: %d)", plan.L, (size_t)160 * plan.L, maxb); return SMO_ERR_UNSUPPORTED; }
                any_lds = (size_t)maxb;
                SMO_HIP(hipFuncSetAttribute((const void*)kda_z_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, maxb));
                SMO_HIP(hipFuncSetAttribute((const void*)kda_z_forward, hipFuncAttributeMaxDynamicSharedMemorySize, maxb));
                SMO_HIP(hipFuncSetAttribute((const void*)kda_y_pass, hipFuncAttributeMaxDynamicSharedMemorySize, maxb));
                SMO_HIP(hipFuncSetAttribute((const void*)kda_x_pass, hipFuncAttributeMaxDynamicSharedMemorySize, maxb));
            }
        }
        ck = cfg.ckpt;
        if (ck < 0) { set_error("KDYN: ckpt=%d", ck); return SMO_ERR_ARG; }
        if (ck == 0) {                                       // smallest interval that fits the free HBM (keep 8 GB + work buffers spare)
            size_t free_b = 0, total_b = 0;
            SMO_HIP(hipMemGetInfo(&free_b, &total_b));
            const size_t work = (2 * n_ex + 3 * fld + 6 * nmode) * sizeof(cplx) + n_grid * 8 + ((size_t)8 << 30);
            for (ck = 1; ck < cfg.n_iters && stack_elems(ck) * sizeof(cplx) + work > free_b; ++ck) {}
        }
        if (ck > cfg.n_iters) ck = cfg.n_iters;
        stack_bytes = stack_elems(ck) * sizeof(cplx);
        SMO_TRY(pool.upload(&d_tw, twiddles(g.G), stream));
        // a context with checkpoint windows on ONE GPU allocates its stack last, sized with a dense tail from what is then free (below);
        // with slabs the ranks would have to agree on the tail as they do on the interval: uniform there
        // (an explicit interval — smo_config.ckpt > 1 — is kept as asked for unless SMO_KD_DENSE_FROM names a tail: the automatic tail belongs to ckpt = 0)
        const bool tail_ok = ck > 1 && W == 1 && (cfg.ckpt == 0 || getenv("SMO_KD_DENSE_FROM")) &&
                             !(getenv("SMO_SLAB_FORCE_EXCHANGE") && atoi(getenv("SMO_SLAB_FORCE_EXCHANGE")) == 1);
        if (!tail_ok) SMO_TRY(pool.alloc(&d_stack, ((size_t)cfg.n_iters / ck + 1) * 3 * nmode));
        if (ck > 1) SMO_TRY(pool.alloc(&d_scratch, (size_t)(ck - 1) * 3 * nmode));
        {   // Ty stack: only when every snapshot is kept and 16 GB of HBM stay free afterwards (SMO_KD_TYSTACK=0 disables)
            const char* env = getenv("SMO_KD_TYSTACK");
            size_t free_b = 0, total_b = 0;
            SMO_HIP(hipMemGetInfo(&free_b, &total_b));
            const size_t need = (size_t)cfg.n_iters * fld * sizeof(cplx);
            const size_t rest = (2 * n_ex + 3 * fld + 6 * nmode) * sizeof(cplx) + n_grid * 8 + ((size_t)16 << 30);
            if (ck == 1 && !(env && atoi(env) == 0) && need + rest < free_b) {
                SMO_TRY(pool.alloc(&d_tystack, (size_t)cfg.n_iters * fld));
                stack_bytes += need;
            }
        }
        if (ck > 1) {                                        // Ty cache of the window being replayed (SMO_KD_TYCACHE=0 disables)
            const char* env = getenv("SMO_KD_TYCACHE");
            size_t free_b = 0, total_b = 0;
            SMO_HIP(hipMemGetInfo(&free_b, &total_b));
            const size_t need = (size_t)ck * fld * sizeof(cplx);
            const size_t rest = (2 * n_ex + 3 * fld + 6 * nmode) * sizeof(cplx) + n_grid * 8 + ((size_t)6 << 30);
            if (!(env && atoi(env) == 0) && need + rest < free_b) SMO_TRY(pool.alloc(&d_tycache, (size_t)ck * fld));
        }
        SMO_TRY(pool.alloc(&d_ty, 2 * fld));
        SMO_TRY(pool.alloc(&d_acc, fld));
        // exchange buffers: one and the same on a single GPU; with slabs the z side and the y side are apart (a host layer that does
        // the transposes itself may still point the context at its own pair: SMO_KD_SET_BUFFERS).  SMO_SLAB_FORCE_EXCHANGE=1 keeps them
        // apart on ONE rank too, so that a one-GPU box sends every transpose through the real communicator (a self-exchange).
        { const char* e = getenv("SMO_SLAB_FORCE_EXCHANGE"); force_exchange = (W == 1 && e && atoi(e) == 1); }
        SMO_TRY(pool.alloc(&zs, n_ex));
        if (W == 1 && !force_exchange) ys = zs;
        else SMO_TRY(pool.alloc(&ys, n_ex));
        SMO_TRY(pool.alloc(&d_red, 8));
        SMO_TRY(pool.alloc(&d_G, 3 * nmode));
        SMO_TRY(pool.alloc(&d_nu, 3 * nmode));
        SMO_TRY(pool.alloc(&d_U, n_grid));
        n_part_rows = (cfg.cost == SMO_COST_INTEGRATED) ? (size_t)cfg.n_iters + 1 : 1;
        SMO_TRY(pool.alloc(&d_part, n_part_rows * NPART));
        h_part.resize(n_part_rows * NPART);
        if (tail_ok) {
            // dense tail: as many extra snapshots as the free HBM holds beyond the uniform stack (12 GB + the staging vectors of the host-buffer
            // entry points stay free).  SMO_KD_DENSE_FROM = n forces the first dense index (rounded down to a multiple of the interval; tests),
            // SMO_KD_DENSE_TAIL=0 keeps the uniform schedule.
            const size_t snap_b = (size_t)3 * nmode * sizeof(cplx), base_slots = (size_t)cfg.n_iters / ck + 1;
            size_t free_b = 0, total_b = 0;
            SMO_HIP(hipMemGetInfo(&free_b, &total_b));
            const size_t keep = ((size_t)12 << 30) + 4 * n_grid * sizeof(double);
            const size_t fit = free_b > keep ? (free_b - keep) / snap_b : 0;
            long extra = (long)fit - (long)base_slots;
            const char* off = getenv("SMO_KD_DENSE_TAIL");
            if (off && atoi(off) == 0) extra = 0;
            int df = 1 << 30;
            if (const char* e = getenv("SMO_KD_DENSE_FROM")) df = std::max(0, atoi(e)) / ck * ck;
            else if (extra > 0) {
                // slots(D) = D / ck + (N - D) + 1 <= base_slots + extra, D a multiple of ck: the smallest such D
                const long N = cfg.n_iters, need = N + 1 - (long)base_slots - extra;              // = D (1 - 1/ck) at least
                long D = need <= 0 ? 0 : (need * ck + (ck - 2)) / (ck - 1);
                D = (D + ck - 1) / ck * ck;
                if (D <= N) df = (int)D;
            }
            if (df <= cfg.n_iters) dense_from = df;
            SMO_TRY(pool.alloc(&d_stack, stack_slots() * 3 * nmode));
            stack_bytes = (stack_slots() + (size_t)(ck - 1)) * snap_b;
        }
        // algorithmic bytes per launch (every axis pass reads + writes its field; S0..S3 per component, per slab)
        const double S0 = 16.0 * nmode, S1 = 16.0 * g.al * g.m * (double)g.G, S2 = 16.0 * g.a * (double)g.G * g.Gzl, S3 = 8.0 * (double)g.G * g.G * g.Gzl;
        // second figure = compulsory HBM bytes of the kernel as fused (every input read once, every output written once): the
        // denominator of the roofline fraction (the algorithmic count above it prices passes the fusion removed)
        k_zi = timing.add_class("kd_z_inverse", 3 * (S0 + S1));
        k_zic = timing.add_class("kd_z_inverse<curl>", 3 * (S0 + S1));
        k_yi = timing.add_class("kd_y_pass<inv>", 3 * (S1 + S2));
        k_yf = timing.add_class("kd_y_pass<fwd>", 3 * (S1 + S2));
        k_xf = timing.add_class("kd_x_pass<fused_fwd>", 3 * (2 * (S2 + S3) + 3 * S3),            // 3 c2r + 3 r2c passes + pointwise (read B,U write EMF)
                                6 * S2 + 3 * S3);                                             // fused: read B (Ty), U; write the EMF's spectrum
        k_xa = timing.add_class("kd_x_pass<fused_adj>", 3 * (4 * (S2 + S3) + 5 * S3),            // 6 c2r + 6 r2c + pointwise (read w,U,B_f write F1,F2)
                                15 * S2 + 3 * S3);                                            // fused: read w, B_f, U, the running sum; write F1, the sum
        const double nxt = fuse_next ? 3 * (S0 + S1) : 0.0;                                     // + the next step's inverse z pass, run on the same tile
        const double nxt_hbm = fuse_next ? 3 * S1 : 0.0;                                        // ... which only adds the write of Tz
        k_zfu = timing.add_class("kd_z_forward<fwd_update>", 3 * (S1 + S0) + 12 * S0 + nxt,    // 3 z passes + step (read B,N; write B, snapshot)
                                 3 * S1 + 6 * S0 + nxt_hbm);                                   // fused: read Tz, B_n; write B_{n+1} (the stack IS the state)
        k_zfa = timing.add_class("kd_z_forward<adj_update>", 3 * (S1 + S0) + 12 * S0 + nxt,    // F1 only: F2 is summed on the grid side (nu_B / nu_C)
                                 3 * S1 + 6 * S0 + (cfg.cost == SMO_COST_INTEGRATED ? 3 * S0 : 0.0) + nxt_hbm);
        k_misc = timing.add_class("kd_misc(setup/terminal/energy/grid io)", 0);
        // slab transposes: HIP events around every grouped send/recv on the stream it is issued on (one field group of one rank, all
        // peers, as the byte figure; an adjoint step without kept grid states sends two groups in one call)
        k_ex = timing.add_class("slab_exchange(all-to-all)", 16.0 * tzb * W, 0.0);
        // Inner_Prod_3 (FWD_Solve_KDyn.py:173-181; SURVEY.md 8d: "2 x (vector bytes)" per call — 340 MB at 128^3, 2.72 GB at 256^3)
        k_dot = timing.add_class("kd_dot(inner product)", 2.0 * 8.0 * (double)n_grid);
        return SMO_OK;
    }

    // ---- launch helpers -----------------------------------------------------------------------------------------
    template <class F> int with_L(F f) {
#ifdef SMO_KD_FEW_SIZES              // experimental builds (tools/kernel_resources.sh -DSMO_KD_FEW_SIZES): the two bench grids only, a tenth of the compile time
        switch (g.G) {
            case 192: return f(std::integral_constant<int, 192>());
            case 384: return f(std::integral_constant<int, 384>());
        }
        set_error("KDYN: grid %d not in this experimental build", g.G);
        return SMO_ERR_UNSUPPORTED;
#else
        switch (g.G) {
            case 12: return f(std::integral_constant<int, 12>());
            case 24: return f(std::integral_constant<int, 24>());
            case 30: return f(std::integral_constant<int, 30>());      // Npts = 20, 40, 80, 160, 320: one radix-5 stage
            case 60: return f(std::integral_constant<int, 60>());
            case 120: return f(std::integral_constant<int, 120>());
            case 240: return f(std::integral_constant<int, 240>());
            case 480: return f(std::integral_constant<int, 480>());
            case 90: return f(std::integral_constant<int, 90>());      // Npts = 60, 120, 240 and 100, 200: factors 3 and 5 together, 5 twice
            case 180: return f(std::integral_constant<int, 180>());
            case 360: return f(std::integral_constant<int, 360>());
            case 150: return f(std::integral_constant<int, 150>());
            case 300: return f(std::integral_constant<int, 300>());
            case 42: return f(std::integral_constant<int, 42>());      // Npts = 28, 56, 112, 224: one radix-7 stage
            case 84: return f(std::integral_constant<int, 84>());
            case 168: return f(std::integral_constant<int, 168>());
            case 336: return f(std::integral_constant<int, 336>());
            case 18: return f(std::integral_constant<int, 18>());      // Npts = 12, 36, 72, 144: 3 more than once
            case 54: return f(std::integral_constant<int, 54>());
            case 108: return f(std::integral_constant<int, 108>());
            case 216: return f(std::integral_constant<int, 216>());
            case 36: return f(std::integral_constant<int, 36>());      // Npts = 24, the reference script's default
            case 72: return f(std::integral_constant<int, 72>());
            case 144: return f(std::integral_constant<int, 144>());
            case 48: return f(std::integral_constant<int, 48>());
            case 96: return f(std::integral_constant<int, 96>());
            case 192: return f(std::integral_constant<int, 192>());
            case 288: return f(std::integral_constant<int, 288>());
            case 384: return f(std::integral_constant<int, 384>());
        }
        set_error("KDYN: unsupported grid %d", g.G);
        return SMO_ERR_UNSUPPORTED;
#endif
    }
    // Workgroup shapes.  FFTs per workgroup are halved for the long transform (G = 384) so the LDS footprint per workgroup
    // (<= 37-49 KB => 3-4 workgroups per CU) and the butterflies per thread stay what they are at G = 192.
    template <int L> struct Shape {
        static constexpr int H = (L > 192) ? 2 : 1;
        // z passes: ONE row triple per workgroup (3 FFTs).  G <= 192: 192 threads — every butterfly stage of 192 = 4*4*4*3 is then exactly one round
        // (144 radix-4 / 192 radix-3 butterflies; with two row triples on 256 threads every stage took two rounds, the second 12-50 % full):
        // update kernels 49.7-50.2 / 49.0-49.3 -> 48.7 / 47.0-47.5 us at 128^3.  G > 192: 256 threads (192 / 320 / 384 are 5-8 % slower there).
#ifndef SMO_Z_NBT
#define SMO_Z_NBT 1
#endif
#ifndef SMO_Z_THREADS
#define SMO_Z_THREADS (L > 192 ? 256 : 192)
#endif
        static constexpr int ZNBT = SMO_Z_NBT, ZNT = SMO_Z_THREADS;
        static constexpr int YZT = SMO_Y_ZT, YNT = 256;          // y pass: z columns per workgroup (512 threads: no change)
        // forward x pass: (y,z) points per workgroup (12 / 6 FFTs; 128-B runs at G = 192).  256 threads: one middle-section item per thread
        // (HP * G/3 = 256), 102-105 VGPRs => 4 waves per SIMD = 16 per CU (192 threads: 148-154 VGPRs, 12 per CU): -5..-7 % on this kernel
        static constexpr int XT = 8 / H, XNT = SMO_X_FWD_NT;
        // sequential adjoint pass: one middle-section item (j, line pair) per thread — it keeps 9 complex grid values of omega per item in
        // registers, a second item per thread spills (G = 480: 320 items, 168 VGPRs + 241 spilled with 256 threads; 320 threads: none)
        // SMO_X_SEQ_T_BIG (experiment, round 4): points per tile of the sequential adjoint pass at G > 192.  4 = half tiles, one item per thread,
        // three workgroups per CU (40 KB, 168 VGPRs); 8 = whole 128-byte lines, TWO items per thread on 256 threads, two workgroups per CU
        // (77 KB, built for two waves per SIMD)
#ifndef SMO_X_SEQ_T_BIG
#define SMO_X_SEQ_T_BIG 4
#endif
        static constexpr int XTS = (L > 192) ? SMO_X_SEQ_T_BIG : XT;
        static constexpr int XITEMS = (XTS / 2) * (L / 3);
        static constexpr int XSNT = (XTS != XT) ? SMO_X_SEQ_NT : (XITEMS > SMO_X_SEQ_NT ? ((XITEMS + 63) / 64) * 64 : SMO_X_SEQ_NT);
        static constexpr int XTA = 4 / H, XANT = SMO_X_ADJ_NT;          // adjoint x pass: 12 / 6 FFTs of both field groups; 64 / 32-B runs, tiles grouped per XCD
        static constexpr int XTG = 16 / H, XGNT = 384;         // grid <-> spectrum only (setup / gradient output)
    };

    // run-time-length path (kdyn_any.hpp): largest tile of `unit` bytes per transform-row multiple that fits the LDS limit
    bool any_size = false;
    AnyPlan plan{};
    size_t any_lds = 65536;
    int any_nt = 512;                            // threads per workgroup of the any-size kernels (SMO_KD_ANY_NT)
    int any_tile(size_t bytes_per_unit, std::initializer_list<int> cands) const {
        for (int c : cands) if (((size_t)c * bytes_per_unit + 16) * plan.L <= any_lds) return c;      // + the twiddle table
        return *(cands.end() - 1);
    }

    int need_buffers() {
        if (!zs || !ys) { set_error("KDYN: slab exchange buffers not set (SMO_KD_SET_BUFFERS)"); return SMO_ERR_STATE; }
        return SMO_OK;
    }
    // coefficients -> field group `f` (of `nf`) of the z-side exchange buffer
    int z_inverse(int mode, const cplx* in, int f, int nf) {
        const Geom g = geom(nf);
        zs_ready_fwd = zs_ready_adj = -1;
        cplx* out = zs + (size_t)f * tzc;
        if (any_size) {
            const int NBT = any_tile(96, {2, 1}), nwg = (g.al * g.m + NBT - 1) / NBT;
            ScopedTimer t(timing, mode == ZI_CURL ? k_zic : (mode == ZI_PLAIN ? k_zi : k_misc), stream);
            hipLaunchKernelGGL(kda_z_inverse, dim3(nwg), dim3(any_nt), ((size_t)96 * NBT + 16) * plan.L, stream, in, out, (const cplx*)d_tw, g, plan, mode, NBT);
            return SMO_OK;
        }
        return with_L([&](auto l) {
            constexpr int L = decltype(l)::value;
            using S = Shape<L>;
            const int nwg = (g.al * g.m + S::ZNBT - 1) / S::ZNBT;
            ScopedTimer t(timing, mode == ZI_CURL ? k_zic : (mode == ZI_PLAIN ? k_zi : k_misc), stream, true);
            if (mode == ZI_PLAIN) SMO_LAUNCH_T(t, (kd_z_inverse<L, ZI_PLAIN, S::ZNBT, S::ZNT>), dim3(nwg), dim3(S::ZNT), 0, stream, in, out, d_tw, g);
            else if (mode == ZI_CURL) SMO_LAUNCH_T(t, (kd_z_inverse<L, ZI_CURL, S::ZNBT, S::ZNT>), dim3(nwg), dim3(S::ZNT), 0, stream, in, out, d_tw, g);
            else SMO_LAUNCH_T(t, (kd_z_inverse<L, ZI_SCALE, S::ZNBT, S::ZNT>), dim3(nwg), dim3(S::ZNT), 0, stream, in, out, d_tw, g);
            return SMO_OK;
        });
    }
    // y pass between field group `f` (of `nf`) of chunk `k` of the y-side exchange buffer and (that chunk of) one field group of Ty at `ty`
    int y_pass(bool inv, int f, int nf, cplx* ty, int k = 0) {
        Geom q = geom(nf, k);
        // z-block-major Ty: the inverse pass (reads Tz, WRITES Ty) takes its workgroups kx-fastest — adjacent blocks of Ty are written together
        // (256^3: 315 -> 289 us) —, the forward pass (reads Ty, writes Tz) z-fastest (265 vs 271 us kx-fastest); SMO_KD_YKX = 0 / 1 forces one order
        q.ykx = ykx_env >= 0 ? ykx_env : (inv ? 1 : 0);
        if (!inv && ys == zs) zs_ready_fwd = zs_ready_adj = -1;      // one GPU: the y pass writes the buffer the z pass reads
        cplx* ex = ys + (size_t)k * q.cblk + (size_t)f * tzc;
        if (any_size) {
            const int ZT = any_tile(32, {8, 4, 2, 1}), nwg = 3 * g.a * ((g.Gzl + ZT - 1) / ZT);
            ScopedTimer t(timing, inv ? k_yi : k_yf, stream);
            hipLaunchKernelGGL(kda_y_pass, dim3(nwg), dim3(any_nt), ((size_t)32 * ZT + 16) * plan.L, stream, inv ? (const cplx*)ex : (const cplx*)ty, inv ? ty : ex,
                               (const cplx*)d_tw, q, plan, inv ? 1 : 0, ZT);
            return SMO_OK;
        }
        return with_L([&](auto l) {
            constexpr int L = decltype(l)::value;
            using S = Shape<L>;
            ScopedTimer t(timing, inv ? k_yi : k_yf, stream, true);
            auto launch = [&](auto zt) {
                constexpr int ZT = decltype(zt)::value;
                const int nwg = 3 * g.a * ((g.Gzl + ZT - 1) / ZT);
                if (inv) SMO_LAUNCH_T(t, (kd_y_pass<L, true, ZT, S::YNT>), dim3(nwg), dim3(S::YNT), 0, stream, (const cplx*)ex, ty, d_tw, q);
                else SMO_LAUNCH_T(t, (kd_y_pass<L, false, ZT, S::YNT>), dim3(nwg), dim3(S::YNT), 0, stream, (const cplx*)ty, ex, d_tw, q);
            };
            // thin slabs: halve the z tile when that avoids a mostly empty last tile (e.g. 128^3 on 8 GPUs: 24 local planes)
            if (g.Gzl % S::YZT != 0 && g.Gzl % (S::YZT / 2) == 0) launch(std::integral_constant<int, S::YZT / 2>());
            else launch(std::integral_constant<int, S::YZT>());
            return SMO_OK;
        });
    }
    // chunk `k` of the grid stage.  vec_in / vec_out: caller-layout grid vectors (X_FROM_GRID / X_TO_GRID); vec_out == nullptr with
    // X_TO_GRID writes the internal (tile-major) U field.  inA / inB: read the spectra of field group A / B from elsewhere (the Ty stack).
    int x_pass(int mode, int k, const double* vec_in, double* vec_out, const cplx* inA = nullptr, const cplx* inB = nullptr) {
        const size_t plane = (size_t)g.G * g.Gzl;
        Geom q = geom(1, k);
        const bool to_U = (mode == X_TO_GRID && vec_out == nullptr);
        q.utile = to_U ? 1 : 0;
        double* Uk = d_U + (size_t)k * ngc;
        const double* grid_in = (mode == X_FROM_GRID) ? vec_in : Uk;
        double* grid_out = to_U ? Uk : vec_out;
        cplx *tA = d_ty + (size_t)k * fldc, *tB = d_ty + fld + (size_t)k * fldc;
        const XSpec sp{inA ? inA : tA, inB ? inB : tB, tA, mode == X_FUSED_ADJ ? d_acc + (size_t)k * fldc : tB};
        auto tiles = [&](int T) { return dim3((unsigned)((plane + T - 1) / T)); };
        if (any_size) {
            // the internal U field stays in the flat grid layout [3][G][G][Gzr] (all chunks in one array: Geom::zg0 places the chunk)
            q.utile = 0;
            const int nbuf = mode == X_FUSED_ADJ ? 3 : 2, HP = any_tile((size_t)48 * nbuf, {4, 2, 1});
            const int kc = mode == X_FUSED_FWD ? k_xf : (mode == X_FUSED_ADJ ? k_xa : k_misc);
            ScopedTimer t(timing, kc, stream);
            hipLaunchKernelGGL(kda_x_pass, tiles(2 * HP), dim3(any_nt), ((size_t)48 * nbuf * HP + 16) * plan.L, stream, sp, mode == X_FROM_GRID ? vec_in : (const double*)d_U,
                               to_U ? d_U : vec_out, (const cplx*)d_tw, q, plan, mode, HP);
            return SMO_OK;
        }
        return with_L([&](auto l) {
            constexpr int L = decltype(l)::value;
            using S = Shape<L>;
            const int kc = mode == X_FUSED_FWD ? k_xf : (mode == X_FUSED_ADJ ? k_xa : k_misc);
            ScopedTimer t(timing, kc, stream, true);
            switch (mode) {
                case X_TO_GRID: SMO_LAUNCH_T(t, (kd_x_pass<L, X_TO_GRID, S::XTG, S::XGNT>), tiles(S::XTG), dim3(S::XGNT), 0, stream, sp, grid_in, grid_out, d_tw, q); break;
                case X_FROM_GRID: SMO_LAUNCH_T(t, (kd_x_pass<L, X_FROM_GRID, S::XTG, S::XGNT>), tiles(S::XTG), dim3(S::XGNT), 0, stream, sp, grid_in, grid_out, d_tw, q); break;
                case X_FUSED_FWD: SMO_LAUNCH_T(t, (kd_x_pass<L, X_FUSED_FWD, S::XT, S::XNT, 8 / S::XT>), tiles(S::XT), dim3(S::XNT), 0, stream, sp, grid_in, grid_out, d_tw, q); break;
                default:
                    // adjoint: the field groups one after the other through the tile buffer (tiles as wide as the forward pass's), unless
                    // SMO_KD_ADJ_SEQ=0 asks for both at once in half-width tiles (round 1's kernel, kept for comparison)
                    if (adj_seq) SMO_LAUNCH_T(t, (kd_x_pass<L, X_FUSED_ADJ_SEQ, S::XTS, S::XSNT, 8 / S::XTS>), tiles(S::XTS), dim3(S::XSNT), 0, stream, sp, grid_in, grid_out, d_tw, q);
                    else SMO_LAUNCH_T(t, (kd_x_pass<L, X_FUSED_ADJ, S::XTA, S::XANT, 8 / S::XTA>), tiles(S::XTA), dim3(S::XANT), 0, stream, sp, grid_in, grid_out, d_tw, q);
                    break;
            }
            return SMO_OK;
        });
    }
    // z-side exchange buffer (one field group) -> coefficients / time-step update [-> z-side buffer again: inverse z pass of the new
    // state, see NX_*]
    int z_forward(int mode, cplx* out0, const cplx* state0, const cplx* snp, int next = NX_NONE) {
        const Geom g = geom(1);
        const double scale = 1.0 / ((double)g.G * g.G * g.G);
        const int integ = cfg.cost == SMO_COST_INTEGRATED;
        if (any_size) {
            const int NBT = any_tile(96, {2, 1}), nwg = (g.al * g.m + NBT - 1) / NBT;
            ScopedTimer t(timing, mode == ZF_FWD_UPDATE ? k_zfu : (mode == ZF_ADJ_UPDATE ? k_zfa : k_misc), stream);
            hipLaunchKernelGGL(kda_z_forward, dim3(nwg), dim3(any_nt), ((size_t)96 * NBT + 16) * plan.L, stream, (const cplx*)zs, out0, state0, snp, (const cplx*)d_tw, g, plan,
                               mode, NBT, scale, integ);
            return SMO_OK;
        }
        return with_L([&](auto l) {
            constexpr int L = decltype(l)::value;
            using S = Shape<L>;
            const int k = mode == ZF_FWD_UPDATE ? k_zfu : (mode == ZF_ADJ_UPDATE ? k_zfa : k_misc);
            ScopedTimer t(timing, k, stream, true);
            const int nwg = (g.al * g.m + S::ZNBT - 1) / S::ZNBT;
            const dim3 grid(nwg), block(S::ZNT);
#define SMO_ZF(MODE_, NEXT_) SMO_LAUNCH_T(t, (kd_z_forward<L, MODE_, NEXT_, S::ZNBT, S::ZNT>), grid, block, 0, stream, (const cplx*)zs, out0, state0, snp, d_tw, g, scale, integ, zs)
            if (mode == ZF_PLAIN) SMO_ZF(ZF_PLAIN, NX_NONE);
            else if (mode == ZF_NU) SMO_ZF(ZF_NU, NX_NONE);
            else if (mode == ZF_FWD_UPDATE) { if (next == NX_PLAIN) SMO_ZF(ZF_FWD_UPDATE, NX_PLAIN); else SMO_ZF(ZF_FWD_UPDATE, NX_NONE); }
            else { if (next == NX_CURL) SMO_ZF(ZF_ADJ_UPDATE, NX_CURL); else SMO_ZF(ZF_ADJ_UPDATE, NX_NONE); }
#undef SMO_ZF
            return SMO_OK;
        });
    }

    // ---- phases: everything between two slab exchanges (the exchange z-side <-> y-side is the host layer's job) -------
    // Ty stack (keep-all, enough HBM): the inverse y pass of B^_n writes to its own HBM slot instead of the work buffer; the adjoint
    // step then reads B_f from there: no z / y pass of the snapshot (2 of its 8 kernels) and, with slabs, one field group less to
    // exchange.
    cplx* d_tystack = nullptr;
    cplx* d_tycache = nullptr;                 // checkpoint windows: Ty(B^_n) of the window being replayed, [ck][fld]
    int tycache_window = -1, tycache_count = 0;
    bool in_tycache(int n) const { return d_tycache && tycache_window >= 0 && n >= tycache_window * ck && n < tycache_window * ck + tycache_count; }
    cplx* tyslot(int n, int k = 0) {
        if (d_tystack) return d_tystack + (size_t)n * fld + (size_t)k * fldc;
        return d_tycache + (size_t)(n - tycache_window * ck) * fld + (size_t)k * fldc;
    }
    cplx* tyw(int f, int k) { return d_ty + (size_t)f * fld + (size_t)k * fldc; }       // work copy of Ty: field group f, chunk k
    bool have_ty(int n) const { return n >= 0 && n < cfg.n_iters && (d_tystack != nullptr || in_tycache(n)); }
    // zs_ready_*: the z-side buffer already holds the inverse z pass of snapshot n / of curl(G^) for adjoint index idx, left there by
    // the update kernel of the step before (NX_*); every other writer of the z-side buffer clears them.
    int zs_ready_fwd = -1, zs_ready_adj = -1;
    bool adj_cont = false;
    bool fuse_next = true;                     // SMO_KD_FUSE_NEXT=0: separate kernels (ablation)
    bool adj_seq = true;                       // SMO_KD_ADJ_SEQ=0: adjoint x pass with both field groups in the tile at once
    int fwd_A(int n) {
        if (zs_ready_fwd == n) { zs_ready_fwd = -1; return SMO_OK; }
        return z_inverse(ZI_PLAIN, snap(n), 0, 1);
    }
    int fwd_B(int n, int k) {
        cplx* ty = have_ty(n) ? tyslot(n, k) : tyw(0, k);
        SMO_TRY(y_pass(true, 0, 1, ty, k));
        SMO_TRY(x_pass(X_FUSED_FWD, k, nullptr, nullptr, ty));
        return y_pass(false, 0, 1, tyw(0, k), k);
    }
    int fwd_C(int n) {
        const bool fuse = fuse_next && n + 1 < cfg.n_iters;
        SMO_TRY(z_forward(ZF_FWD_UPDATE, snap(n + 1), snap(n), nullptr, fuse ? NX_PLAIN : NX_NONE));
        zs_ready_adj = -1;
        zs_ready_fwd = fuse ? n + 1 : -1;
        return SMO_OK;
    }
    int adj_init(int adjoint_type) {
        ScopedTimer t(timing, k_misc, stream);
        hipLaunchKernelGGL(kd_terminal, dim3(1024), dim3(256), 0, stream, snap(cfg.n_iters), d_G, d_nu, g, cfg.cost == SMO_COST_INTEGRATED ? 1 : 0,
                           adjoint_type == SMO_ADJ_CONTINUOUS ? 1 : 0);
        SMO_HIP(hipMemsetAsync(d_acc, 0, fld * sizeof(cplx), stream));
        adj_cont = adjoint_type == SMO_ADJ_CONTINUOUS;
        zs_ready_fwd = zs_ready_adj = -1;
        return SMO_OK;
    }
    // inverse side of an adjoint step carries omega only (1 field group) when B_f was kept by the forward solve, else omega and B^_idx
    int adj_groups(int idx) const { return have_ty(idx) ? 1 : 2; }
    // one slab, one chunk: field group 0 sits at the same place of the exchange buffer whether the step carries one group or two (zs_off: row + pos,
    // the second group tzc elements further), so the curl pass fused into the previous update serves a two-group step as well
    bool group0_layout_is_shared() const { return cfg.world == 1 && K == 1 && !force_exchange; }
    int adj_A(int idx) {
        const int nf = adj_groups(idx);
        if (zs_ready_adj == idx && (nf == 1 || group0_layout_is_shared())) {
            zs_ready_adj = -1;
            return nf == 2 ? z_inverse(ZI_PLAIN, snap(idx), 1, 2) : SMO_OK;      // (writes the second group only)
        }
        SMO_TRY(z_inverse(ZI_CURL, d_G, 0, nf));
        return nf == 2 ? z_inverse(ZI_PLAIN, snap(idx), 1, 2) : SMO_OK;
    }
    int adj_B(int idx, int k) {
        const int nf = adj_groups(idx);
        SMO_TRY(y_pass(true, 0, nf, tyw(0, k), k));
        if (nf == 2) SMO_TRY(y_pass(true, 1, 2, tyw(1, k), k));
        SMO_TRY(x_pass(X_FUSED_ADJ, k, nullptr, nullptr, nullptr, nf == 1 ? tyslot(idx, k) : nullptr));
        return y_pass(false, 0, 1, tyw(0, k), k);
    }
    int adj_C(int idx) {
        // the following adjoint step (index idx - 1) starts with the inverse z pass of curl(G^): fused when that step sends one field
        // group (same buffer layout as this kernel's input) and exists at all (the continuous sweep ends at index 1, the discrete at 0)
        const int nxt = idx - 1;
        const bool fuse = fuse_next && nxt >= (adj_cont ? 1 : 0) && (adj_groups(nxt) == 1 || group0_layout_is_shared());
        SMO_TRY(z_forward(ZF_ADJ_UPDATE, d_G, d_G, snap(idx), fuse ? NX_CURL : NX_NONE));
        zs_ready_fwd = -1;
        zs_ready_adj = fuse ? nxt : -1;
        return SMO_OK;
    }
    // The nu^ recursion of the reference, nu <- R nu - dt P F2_n (R = I - 2 k k^T/k^2, nu_N = 0), never leaves the solenoidal subspace,
    // where R is the identity: nu_0 = -dt P sum_n F2_n exactly.  The fused adjoint x pass therefore adds its second product
    // (curl G_n) x B_n, already transformed along x, to a running sum on the grid side (d_acc, the layout of Ty); the y and z passes,
    // the projection and the factor -dt are applied ONCE after the last step (nu_B / nu_C) instead of once per step: one y pass and
    // half a z pass less per adjoint step and, with slabs, one field group less to send back.
    cplx* d_acc = nullptr;
    int nu_B(int k) { return y_pass(false, 0, 1, d_acc + (size_t)k * fldc, k); }
    int nu_C() { return z_forward(ZF_NU, d_nu, nullptr, nullptr); }
    // grid vector (local slab of the flat X layout) -> truncated coefficients, in two phases around the exchange
    int g2c_A(const double* X, int k) { SMO_TRY(x_pass(X_FROM_GRID, k, X, nullptr)); return y_pass(false, 0, 1, tyw(0, k), k); }
    int g2c_C(cplx* out) { return z_forward(ZF_PLAIN, out, nullptr, nullptr); }
    // coefficients -> grid; scaled = multiply by dt*alpha(k) first ("undo LHS", FWD_Solve_KDyn.py:985-989)
    int c2g_A(const cplx* C, bool scaled) { return z_inverse(scaled ? ZI_SCALE : ZI_PLAIN, C, 0, 1); }
    int c2g_B(double* X, int k) { SMO_TRY(y_pass(true, 0, 1, tyw(0, k), k)); return x_pass(X_TO_GRID, k, nullptr, X); }   // X == nullptr: the U field

    int reduce_partials(double* out) {
        SMO_HIP(hipMemcpyAsync(h_part.data(), d_part, NPART * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        double s = 0.0;
        for (int i = 0; i < NPART; ++i) s += h_part[i];
        *out = s;
        return SMO_OK;
    }
    int energy(const cplx* Bh, double* E) {            // this slab's share of sum_k w |B^|^2
        {
            ScopedTimer t(timing, k_misc, stream);
            hipLaunchKernelGGL(kd_energy, dim3(NPART), dim3(256), 0, stream, Bh, d_part, g);
        }
        return reduce_partials(E);
    }

    // ---- slab communicator: the pencil transposes and scalar reductions of the time loop, inside the library ------------------
    SlabComm comm;
    bool force_exchange = false;
    hipStream_t cstream = nullptr;               // exchanges of the chunk pipeline (K > 1) run here, next to the kernels on `stream`
    hipEvent_t ev_main = nullptr;
    std::vector<hipEvent_t> ev_in, ev_ph, ev_out;
    double* d_red = nullptr;
    bool exchanging() const { return (cfg.world > 1 || force_exchange) && comm.ready(); }
    ~KDyn() override {
        (void)hipSetDevice(cfg.device);
        if (stream) (void)hipStreamSynchronize(stream);
        drop_graphs();
        if (cstream) { (void)hipStreamSynchronize(cstream); (void)hipStreamDestroy(cstream); }
        if (ev_main) (void)hipEventDestroy(ev_main);
        for (auto* v : {&ev_in, &ev_ph, &ev_out}) for (hipEvent_t e : *v) (void)hipEventDestroy(e);
    }
    // all-to-all of chunk k, nf field groups: z side -> y side (to_y) or back.  Peer blocks are contiguous: [chunk][peer][nf*tzc]
    // chained: an exchange of the time loop (stage()), whose send buffer is next written by the pull of the exchange that follows it on the
    // same stream — the multi-device transport then skips its second rendezvous (comm.hpp).  SMO_PEER_CHAINED=0 keeps the full protocol.
    bool chain_ok = true;
    int exchange(bool to_y, int nf, int k, hipStream_t s, bool chained = false) {
        if (!exchanging()) return SMO_OK;
        const size_t off = (size_t)k * (size_t)cfg.world * 2 * tzc;
        const cplx* src = (to_y ? zs : ys) + off;
        cplx* dst = (to_y ? ys : zs) + off;
        if (!to_y) zs_ready_fwd = zs_ready_adj = -1;
        ScopedTimer t(timing, k_ex, s);
        return comm.alltoall(src, dst, (size_t)nf * tzc * sizeof(cplx), s, chained && chain_ok);
    }
    enum { ST_FWD = 0, ST_ADJ = 1 };
    int grid_phase(int code, int idx, int k) { return code == ST_FWD ? fwd_B(idx, k) : adj_B(idx, k); }
    // one transpose -> grid work -> transpose back.  K > 1: chunk-pipelined — the K inbound exchanges are issued up front on the
    // communication stream, every chunk's grid kernels wait (stream-level, never the host) for their chunk only, and the outbound
    // exchange of chunk k overlaps the kernels of chunk k+1.
    int stage(int code, int idx, int nf_in, int nf_out) {
        if (!exchanging() || K == 1) {
            SMO_TRY(exchange(true, nf_in, 0, stream, true));
            SMO_TRY(grid_phase(code, idx, 0));
            return exchange(false, nf_out, 0, stream, true);
        }
        SMO_HIP(hipEventRecord(ev_main, stream));
        SMO_HIP(hipStreamWaitEvent(cstream, ev_main, 0));
        for (int k = 0; k < K; ++k) {
            SMO_TRY(exchange(true, nf_in, k, cstream, true));
            SMO_HIP(hipEventRecord(ev_in[k], cstream));
        }
        for (int k = 0; k < K; ++k) {
            SMO_HIP(hipStreamWaitEvent(stream, ev_in[k], 0));
            SMO_TRY(grid_phase(code, idx, k));
            SMO_HIP(hipEventRecord(ev_ph[k], stream));
            SMO_HIP(hipStreamWaitEvent(cstream, ev_ph[k], 0));
            SMO_TRY(exchange(false, nf_out, k, cstream, true));
            SMO_HIP(hipEventRecord(ev_out[k], cstream));
        }
        for (int k = 0; k < K; ++k) SMO_HIP(hipStreamWaitEvent(stream, ev_out[k], 0));
        return SMO_OK;
    }
    // keep = false: recomputation of a window (checkpointing): the grid-side state of step n is not kept again
    int step_fwd(int n, bool keep) {
        SMO_TRY(fwd_A(n));
        SMO_TRY(stage(ST_FWD, keep ? n : -1, 1, 1));
        return fwd_C(n);
    }
    int grid_to_coeff(const double* X, cplx* out) {
        for (int k = 0; k < K; ++k) { SMO_TRY(g2c_A(X, k)); SMO_TRY(exchange(false, 1, k, stream)); }
        return g2c_C(out);
    }
    int coeff_to_grid(const cplx* C, bool scaled, double* X) {
        SMO_TRY(c2g_A(C, scaled));
        for (int k = 0; k < K; ++k) { SMO_TRY(exchange(true, 1, k, stream)); SMO_TRY(c2g_B(X, k)); }
        return SMO_OK;
    }
    int allreduce(double* v, int n) { return cfg.world > 1 ? comm.allreduce_sum(v, n, stream, d_red) : SMO_OK; }

    // Called once the transport exists (collective): the ranks agree on what each of them decided from its own free HBM — the
    // checkpoint interval and whether the grid-side states are kept (that fixes how many field groups an adjoint step sends) —
    // before the first exchange could mismatch, and the chunk pipeline is set up.
    // stream and events of the chunk pipeline, for the current K
    int pipeline_resources() {
        if (!cstream) SMO_HIP(hipStreamCreateWithFlags(&cstream, hipStreamNonBlocking));
        if (!ev_main) SMO_HIP(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming));
        for (auto* vec : {&ev_in, &ev_ph, &ev_out})
            while ((int)vec->size() < K) { hipEvent_t e; SMO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); vec->push_back(e); }
        return SMO_OK;
    }
    int comm_attach() {
        double v[4] = {d_tystack ? 1.0 : 0.0, (double)ck, (double)ck * ck, d_tycache ? 1.0 : 0.0};
        SMO_TRY(allreduce(v, 4));
        const double W = cfg.world;
        if (std::fabs(W * v[2] - v[1] * v[1]) > 0.5) {
            set_error("KDYN: the ranks chose different checkpoint intervals from their free HBM (mine: %d); pass an explicit smo_config.ckpt", ck);
            return SMO_ERR_STATE;
        }
        if (d_tystack && v[0] < W - 0.5) {               // some rank could not keep the grid-side states: nobody does
            SMO_HIP(hipStreamSynchronize(stream));
            SMO_TRY(pool.free_one(d_tystack));
            stack_bytes -= (size_t)cfg.n_iters * fld * sizeof(cplx);
            d_tystack = nullptr;
        }
        if (d_tycache && v[3] < W - 0.5) {                // the same for the Ty cache of the checkpoint windows
            SMO_HIP(hipStreamSynchronize(stream));
            SMO_TRY(pool.free_one(d_tycache));
            d_tycache = nullptr; tycache_window = -1;
        }
        int k = 0;
        if (const char* e = getenv("SMO_SLAB_CHUNKS")) k = atoi(e);
        // default: up to 4 chunks of at least 36864 (y,z) points (one 192 x 192 plane set).  Chunking is not free: with the null transport
        // (no exchange at all) the 256^3 step pair of an 8-way decomposition takes 570 us with one chunk and 739 us with two, 1075 / 1390 us
        // 4-way with 1 / 4 (profiles/r03_slab_geometry_256.jsonl) — smaller kernels fill the GPU worse (a fused x pass of 9216 points is
        // 1152 tiles for 1024 slots) and every chunk adds event traffic — so it only pays where the transposes are slow enough to be worth
        // hiding, which a one-GPU measurement cannot tell: LibSlabKDyn.autotune_chunks / SMO_SLAB_CHUNKS decide on the node itself.
        if (k <= 0) k = cfg.world > 1 ? std::max(1, std::min(4, (int)(((size_t)g.G * g.Gzr) / 36864))) : 1;
        while (k > 1 && (g.Gzr % k || (g.Gzr / k) % 2 || ((size_t)g.G * (g.Gzr / k)) % 4)) --k;
        if (k != K) SMO_TRY(set_chunks(k));
        SMO_TRY(pipeline_resources());
        have_forward = false;
        return SMO_OK;
    }
    // a context whose ranks could not agree (different checkpoint intervals, a buffer that would not free, chunks that do not divide)
    // must not keep a live transport: K, the Ty stack and the Ty cache may then differ between the ranks and the next solve would hang
    // in a mismatched all-to-all.  The transport is dropped, smo_forward fails with SMO_ERR_STATE, and smo_comm_init may be called again.
    int attach_or_drop() {
        const int rc = comm_attach();
        if (rc != SMO_OK) {
            const std::string why = last_error();
            (void)hipStreamSynchronize(stream);
            comm.reset();
            set_error("%s (the communicator was dropped: call smo_comm_init / smo_comm_set_transport again once the ranks agree)", why.c_str());
        }
        return rc;
    }
    int comm_init(const void* id128) override {
        if (cfg.world == 1 && !force_exchange) { set_error("smo_comm_init: single-slab context (nothing to exchange)"); return SMO_ERR_STATE; }
        SMO_TRY(comm.init_rccl(cfg.rank, cfg.world, id128));
        return attach_or_drop();
    }
    int comm_set_transport(smo_alltoall_fn a2a, smo_allreduce_fn ared, void* user) override {
        if (cfg.world == 1 && !force_exchange) { set_error("smo_comm_set_transport: single-slab context (nothing to exchange)"); return SMO_ERR_STATE; }
        SMO_TRY(comm.set_transport(cfg.rank, cfg.world, a2a, ared, user));
        return attach_or_drop();
    }
    int comm_set_peers(PeerGroup* grp, int rank) override {
        if (cfg.world == 1 && !force_exchange) { set_error("comm_set_peers: single-slab context (nothing to exchange)"); return SMO_ERR_STATE; }
        if (rank != cfg.rank || grp->world() != cfg.world) { set_error("comm_set_peers: rank %d of %d does not match the context (%d of %d)", rank, grp->world(), cfg.rank, cfg.world); return SMO_ERR_ARG; }
        SMO_TRY(comm.set_peers(rank, grp));
        return attach_or_drop();
    }
    double comm_info(int key) const override {
        if (key == 0) return (double)K;
        if (key == 1) return 3.0 + (d_tystack ? 1.0 : 2.0);
        if (key == 2) return comm.is_rccl() ? 1.0 : 0.0;
        return 0.0;                               // keys 3, 4: the multi-device context's transport (csrc/multi.cpp)
    }

    // ---- the three callbacks: phases back to back; with slabs the transposes in between go through the communicator ------------
    int loop_ok(const char* who) {
        if ((cfg.world > 1 || force_exchange) && !comm.ready()) {
            set_error("%s: %d slabs and no communicator — call smo_comm_init (RCCL) or smo_comm_set_transport first, or drive the phases "
                      "yourself (smo_kdyn_op)", who, cfg.world);
            return SMO_ERR_STATE;
        }
        if (K != 1 && !exchanging()) { set_error("%s: the z slab is cut into %d chunks (phase-level use only)", who, K); return SMO_ERR_STATE; }
        return need_buffers();
    }
    // ---- HIP graphs for the launch-bound sizes ----------------------------------------------------------------------------
    // A 24^3 solve (the reference script's default) is 2000 steps x 4 kernels of ~2 us of work each: 4.8 us per launch measured.  For
    // such grids on one GPU the whole forward solve and the whole adjoint sweep are captured ONCE as HIP graphs (the stack addresses
    // never change; the caller's vectors are copied to / from fixed buffers so that the graphs are independent of them) and replayed
    // by every later call.  Same kernels in the same order: results are bit-identical with the launch-by-launch path (SMO_KD_GRAPH=0).
    // Measured (tools/time_small_grid.py, profiles/r02_small_grid_graph.jsonl): 77.0 -> 68.7 ms per gradient at 24^3 (4.8 -> 4.3 us per
    // kernel: what remains is the GPU-side hand-over between dependent kernels, not the host), no gain from 32^3 up, +75 ms for the
    // capture at the first call — hence on by default only up to G = 36.  Off while kernel timing is on, with slabs, chunks or
    // checkpoint windows.
    int graph_mode = -1, graph_max_g = 36;       // SMO_KD_GRAPH: -1 (default) = grids up to graph_max_g (SMO_KD_GRAPH_MAXG), 0 = never, 1 = always
    bool graph_broken = false;
    hipGraphExec_t gx_fwd = nullptr, gx_adj[2] = {nullptr, nullptr};
    struct Flags { int zf, za; bool cont; } fl_fwd{}, fl_adj[2]{};
    double *d_xin[2] = {nullptr, nullptr}, *d_gout[2] = {nullptr, nullptr};
    long long graph_replays = 0;
    bool use_graph() const {
        if (graph_broken || graph_mode == 0 || timing.on || cfg.world != 1 || force_exchange || K != 1 || ck != 1) return false;
        if (cfg.n_iters > 8192) return false;                 // a graph holds ~4 nodes per step: keep its instantiation in the tens of milliseconds
        return graph_mode == 1 || g.G <= graph_max_g;
    }
    template <class F> int run_graph(hipGraphExec_t* gx, Flags* after, F enqueue) {
        if (!*gx) {
            hipGraph_t graph = nullptr;
            SMO_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue();
            const hipError_t e = hipStreamEndCapture(stream, &graph);
            if (rc != SMO_OK || e != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                graph_broken = true;                     // this context goes on launch by launch
                return rc != SMO_OK ? rc : enqueue();
            }
            const hipError_t ei = hipGraphInstantiate(gx, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ei != hipSuccess) { *gx = nullptr; (void)hipGetLastError(); graph_broken = true; return enqueue(); }
            *after = Flags{zs_ready_fwd, zs_ready_adj, adj_cont};
        }
        SMO_HIP(hipGraphLaunch(*gx, stream));
        zs_ready_fwd = after->zf; zs_ready_adj = after->za; adj_cont = after->cont;      // what the enqueueing code leaves behind
        ++graph_replays;
        return SMO_OK;
    }
    int graph_buffers() {
        for (int c = 0; c < 2; ++c) {
            if (!d_xin[c]) SMO_TRY(pool.alloc(&d_xin[c], n_grid));
            if (!d_gout[c]) SMO_TRY(pool.alloc(&d_gout[c], n_grid));
        }
        return SMO_OK;
    }
    void drop_graphs() {
        for (hipGraphExec_t* x : {&gx_fwd, &gx_adj[0], &gx_adj[1]})
            if (*x) { (void)hipGraphExecDestroy(*x); *x = nullptr; }
    }

    // everything of a forward solve that is a kernel launch (the partial sums of J stay in d_part)
    int fwd_enqueue(const double* B, const double* U) {
        const int N = cfg.n_iters;
        tycache_window = -1;                               // the states change: whatever the last adjoint sweep cached is stale
        const bool integ = cfg.cost == SMO_COST_INTEGRATED;
        // U: truncate to the retained modes, back to the grid (NCC fields are band-limited by the solver)
        SMO_TRY(grid_to_coeff(U, d_G));
        SMO_TRY(coeff_to_grid(d_G, false, nullptr));
        SMO_TRY(grid_to_coeff(B, snap(0)));
        for (int n = 0; n < N; ++n) {
            if (integ) {                                   // <B_n,B_n> partial sums stay on the device until the end of the solve
                ScopedTimer t(timing, k_misc, stream);
                hipLaunchKernelGGL(kd_energy, dim3(NPART), dim3(256), 0, stream, snap(n), d_part + (size_t)n * NPART, g);
            }
            SMO_TRY(step_fwd(n, true));
        }
        ScopedTimer t(timing, k_misc, stream);
        hipLaunchKernelGGL(kd_energy, dim3(NPART), dim3(256), 0, stream, snap(N), d_part + (integ ? (size_t)N * NPART : 0), g);
        return SMO_OK;
    }
    // host time this rank spent ISSUING the last forward + adjoint solve: from the entry of the call to the point where everything is
    // enqueued, minus the time spent waiting for the other ranks in host rendezvous (smo_get key 4, milliseconds)
    double issue_ms_fwd = 0.0, issue_ms_adj = 0.0;
    static double now_ms() { return 1e-6 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    int forward_dev(const double* const* X, double* J) override {
        SMO_TRY(loop_ok("smo_forward"));
        have_forward = false;
        const double t_in = now_ms(), w_in = comm.wait_ms();
        const int N = cfg.n_iters;
        const bool integ = cfg.cost == SMO_COST_INTEGRATED;
        if (use_graph()) {
            SMO_TRY(graph_buffers());
            for (int c = 0; c < 2; ++c) SMO_HIP(hipMemcpyAsync(d_xin[c], X[c], n_grid * sizeof(double), hipMemcpyDeviceToDevice, stream));
            SMO_TRY(run_graph(&gx_fwd, &fl_fwd, [&]() { return fwd_enqueue(d_xin[0], d_xin[1]); }));
        } else {
            SMO_TRY(fwd_enqueue(X[0], X[1]));
        }
        issue_ms_fwd = (now_ms() - t_in) - (comm.wait_ms() - w_in);
        scratch_window = (ck > 1 && dense_from > 0) ? (std::min(N, dense_from) - 1) / ck : -1;       // the scratch slots now hold the last window that was not kept whole
        const size_t rows = integ ? (size_t)N + 1 : 1;
        SMO_HIP(hipMemcpyAsync(h_part.data(), d_part, rows * NPART * sizeof(double), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        double Jacc = 0.0;
        for (size_t n = 0; n < rows; ++n) {
            double e = 0.0;
            for (int i = 0; i < NPART; ++i) e += h_part[n * NPART + i];
            Jacc += integ ? cfg.dt * e : e;
        }
        SMO_HIP(hipGetLastError());
        SMO_TRY(allreduce(&Jacc, 1));                        // slabs: every rank holds its share of the spectral sum
        SMO_HIP(hipStreamSynchronize(stream));
        if (cstream) SMO_HIP(hipStreamSynchronize(cstream));
        *J = -Jacc;
        have_forward = true;
        return SMO_OK;
    }

    int adj_enqueue(int adjoint_type, double* gB, double* gU) {
        const int N = cfg.n_iters;
        const bool cont = adjoint_type == SMO_ADJ_CONTINUOUS;
        SMO_TRY(ensure(N));
        SMO_TRY(adj_init(adjoint_type));
        int idx = cont ? N : N - 1;
        for (int it = 0; it < N; ++it, --idx) {
            SMO_TRY(ensure(idx));
            SMO_TRY(adj_A(idx));
            SMO_TRY(stage(ST_ADJ, idx, adj_groups(idx), 1));
            SMO_TRY(adj_C(idx));
        }
        SMO_TRY(coeff_to_grid(d_G, !cont, gB));
        for (int k = 0; k < K; ++k) { SMO_TRY(nu_B(k)); SMO_TRY(exchange(false, 1, k, stream)); }
        SMO_TRY(nu_C());
        return coeff_to_grid(d_nu, false, gU);
    }
    int adjoint_dev(const double* const*, int adjoint_type, double* const* grad) override {
        SMO_TRY(loop_ok("smo_adjoint"));
        const double t_in = now_ms(), w_in = comm.wait_ms();
        if (use_graph()) {
            SMO_TRY(graph_buffers());
            const int a = adjoint_type == SMO_ADJ_CONTINUOUS ? 1 : 0;
            SMO_TRY(run_graph(&gx_adj[a], &fl_adj[a], [&]() { return adj_enqueue(adjoint_type, d_gout[0], d_gout[1]); }));
            for (int c = 0; c < 2; ++c) SMO_HIP(hipMemcpyAsync(grad[c], d_gout[c], n_grid * sizeof(double), hipMemcpyDeviceToDevice, stream));
        } else {
            SMO_TRY(adj_enqueue(adjoint_type, grad[0], grad[1]));
        }
        issue_ms_adj = (now_ms() - t_in) - (comm.wait_ms() - w_in);
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipStreamSynchronize(stream));
        if (cstream) SMO_HIP(hipStreamSynchronize(cstream));
        return SMO_OK;
    }

    int inner_dev(const double* x, const double* y, double* out) override {       // <x,y>; slabs without a communicator: this slab's share
        {
            ScopedTimer t(timing, k_dot, stream, true);
            SMO_LAUNCH_T(t, kd_dot, dim3(NPART), dim3(256), 0, stream, x, y, d_part, n_grid);
        }
        SMO_HIP(hipGetLastError());
        double s = 0.0;
        SMO_TRY(reduce_partials(&s));
        if (comm.ready()) SMO_TRY(allreduce(&s, 1));
        *out = s / ((double)g.G * g.G * g.G);
        return SMO_OK;
    }

    double info(int key) const override {
        if (key == 0) return (double)ck;
        if (key == 2) return (double)graph_replays;
        if (key == 3) return (double)g.tyl;
        if (key == 4) return issue_ms_fwd + issue_ms_adj;
        if (key == 5) return dense_from > cfg.n_iters ? -1.0 : (double)dense_from;
        return d_tystack ? (double)((size_t)cfg.n_iters * fld * sizeof(cplx)) : 0.0;
    }

    // smo_transform for the 3-D case (host buffers): which = 0: three grid fields float64[3][G][G][G] -> their truncated coefficients
    // complex128[3][a][m][m] (Dedalus' amplitude normalisation); which = 1: the inverse.  The same passes the solves use; single GPU.
    double* d_tgrid = nullptr;
    int transform_host(int which, const double* in, double* out) override {
        if (cfg.world != 1 || K != 1) { set_error("smo_transform (KDYN): single-GPU contexts only"); return SMO_ERR_UNSUPPORTED; }
        if (which != 0 && which != 1) { set_error("smo_transform (KDYN): which = 0 (grid -> coefficients) or 1 (coefficients -> grid)"); return SMO_ERR_ARG; }
        if (!d_tgrid) SMO_TRY(pool.alloc(&d_tgrid, n_grid));
        have_forward = false;                               // d_G (the adjoint's state) is the scratch spectrum here
        if (which == 0) {
            SMO_HIP(hipMemcpyAsync(d_tgrid, in, n_grid * sizeof(double), hipMemcpyHostToDevice, stream));
            SMO_TRY(grid_to_coeff(d_tgrid, d_G));
            SMO_HIP(hipMemcpyAsync(out, d_G, 3 * nmode * sizeof(cplx), hipMemcpyDeviceToHost, stream));
        } else {
            SMO_HIP(hipMemcpyAsync(d_G, in, 3 * nmode * sizeof(cplx), hipMemcpyHostToDevice, stream));
            SMO_TRY(coeff_to_grid(d_G, false, d_tgrid));
            SMO_HIP(hipMemcpyAsync(out, d_tgrid, n_grid * sizeof(double), hipMemcpyDeviceToHost, stream));
        }
        SMO_HIP(hipGetLastError());
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }

    int snapshot_read(int, int index, double* out) override {
        SMO_TRY(ensure(index));
        SMO_HIP(hipMemcpyAsync(out, snap(index), 3 * nmode * sizeof(cplx), hipMemcpyDeviceToHost, stream));
        SMO_HIP(hipStreamSynchronize(stream));
        return SMO_OK;
    }

    // ---- phase-level entry for the slab-decomposed driver (smo_kdyn_op) -------------------------------------------------
    int kdyn_op(int op, int i0, int i1, void* p0, void* p1, double* out) override {
        if (op == SMO_KD_SET_CHUNKS) {                       // also on a context that runs the loop itself: re-tune the exchange pipeline
            SMO_HIP(hipStreamSynchronize(stream));
            if (cstream) SMO_HIP(hipStreamSynchronize(cstream));
            SMO_TRY(set_chunks(i0));
            return comm.ready() ? pipeline_resources() : SMO_OK;
        }
        if (op == SMO_KD_SET_BUFFERS) {
            if (!p0 || !p1) { set_error("SMO_KD_SET_BUFFERS: null buffer"); return SMO_ERR_ARG; }
            zs = static_cast<cplx*>(p0); ys = static_cast<cplx*>(p1); zs_ready_fwd = zs_ready_adj = -1;
            drop_graphs();
            return SMO_OK;
        }
        if (op == SMO_KD_EXCHANGE_ELEMS) { if (!out) return SMO_ERR_ARG; *out = (double)(tzb * cfg.world); return SMO_OK; }
        SMO_TRY(need_buffers());
        const int N = cfg.n_iters;
        auto chunk_ok = [&](int k) { if (k < 0 || k >= K) { set_error("smo_kdyn_op(%d): chunk %d out of range (%d chunks)", op, k, K); return false; } return true; };
        auto step_ok = [&](int n, int hi) { if (n < 0 || n > hi) { set_error("smo_kdyn_op(%d): index %d out of range", op, n); return false; } return true; };
        int rc = SMO_OK;
        switch (op) {
            case SMO_KD_G2C_A: if (!p0 || !chunk_ok(i1)) return SMO_ERR_ARG; rc = g2c_A(static_cast<const double*>(p0), i1); break;
            case SMO_KD_G2C_C: rc = g2c_C(i0 == 0 ? snap(0) : d_G); if (i0 == 0) have_forward = false; break;
            case SMO_KD_C2G_A: rc = c2g_A(i0 == 2 ? d_nu : d_G, i0 == 0); break;
            case SMO_KD_C2G_B: if (!chunk_ok(i1)) return SMO_ERR_ARG; rc = c2g_B(static_cast<double*>(p0), i1); break;
            case SMO_KD_FWD_A: if (!step_ok(i0, N - 1)) return SMO_ERR_ARG; rc = fwd_A(i0); break;
            case SMO_KD_FWD_B: if (!chunk_ok(i1)) return SMO_ERR_ARG; rc = fwd_B(i0, i1); break;
            case SMO_KD_FWD_C: if (!step_ok(i0, N - 1)) return SMO_ERR_ARG; rc = fwd_C(i0); if (i0 == N - 1) have_forward = true; break;
            case SMO_KD_ENERGY: if (!step_ok(i0, N) || !out) return SMO_ERR_ARG; return energy(snap(i0), out);
            case SMO_KD_ADJ_INIT:
                if (!have_forward) { set_error("smo_kdyn_op: adjoint before forward"); return SMO_ERR_STATE; }
                rc = adj_init(i0); break;
            case SMO_KD_ADJ_A: if (!step_ok(i0, N)) return SMO_ERR_ARG; rc = adj_A(i0); break;
            case SMO_KD_ADJ_B: if (!step_ok(i0, N) || !chunk_ok(i1)) return SMO_ERR_ARG; rc = adj_B(i0, i1); break;
            case SMO_KD_ADJ_C: if (!step_ok(i0, N)) return SMO_ERR_ARG; rc = adj_C(i0); break;
            case SMO_KD_NU_B: if (!chunk_ok(i1)) return SMO_ERR_ARG; rc = nu_B(i1); break;
            case SMO_KD_NU_C: rc = nu_C(); break;
            case SMO_KD_SYNC: SMO_HIP(hipStreamSynchronize(stream)); break;
            default: set_error("smo_kdyn_op: unknown op %d", op); return SMO_ERR_ARG;
        }
        if (rc == SMO_OK) SMO_HIP(hipGetLastError());
        return rc;
    }
};

}  // namespace

Context* make_kdyn(const smo_config& cfg) { return new KDyn(cfg); }

}  // namespace smo

// LDS-staged Stockham autosort FFT for gfx950, complex128.
//
// A length-L transform is a chain of decimation-in-frequency Stockham stages of radix [8, then] 4, then 2, then 5, then 3
// (L = 4^a 2^b 5^d 7^e 3^c, b in {0,1}; covers 2^k and the 3/2-dealiased sizes 12..480 incl. 36 = 4*3*3 for Npts = 24 and 30..480 = 3 * Npts/2
// for Npts = 20, 40, 80, 160, 320).  Stage invariant n*s == L:
//     y[q + s*(R*p + j)] = w_n^{p*j} * sum_k x[q + s*(p + k*n/R)] * w_R^{j*k},  0 <= p < n/R, 0 <= q < s
// so (i) the R inputs of consecutive butterflies are consecutive 16-byte elements (conflict-free ds_read_b128 /
// coalesced global loads) and (ii) in the last stage p == 0: no twiddles, outputs of consecutive butterflies
// are consecutive.  Passes therefore read global memory in the first stage and write it in the last one;
// only the stages in between go through LDS (ping-pong between two buffers, one barrier per stage).
//
// The twiddle of output j is w_L^{p*s*j}; p*s*j < L, so one table tw[k] = exp(-2 pi i k / L) serves every stage.
#pragma once
#include "smo_common.hpp"

namespace smo {

// Largest butterfly held in registers.  8 (the 3-D passes): 192 = 8*8*3 and 384 = 8*8*2*3 take 3 / 4 stages instead of 4 / 5 — every stage
// is one round trip through the LDS (a 16-byte ds_write costs 3x a ds_read on gfx950) and one workgroup barrier.
#ifndef SMO_FFT_MAX_RADIX
#define SMO_FFT_MAX_RADIX 4
#endif
// (MAXR: the policy of one kernel — the template parameter the in-place stages below carry; the library default is SMO_FFT_MAX_RADIX)
template <int MAXR> constexpr __host__ __device__ int radix_of_r(int n) {      // 5 and 7 before 3: a length with a factor 3 (every 3/2-dealiased grid) ends on radix 3
    return (MAXR >= 8 && n % 8 == 0) ? 8 : ((n % 4 == 0) ? 4 : ((n % 2 == 0) ? 2 : ((n % 5 == 0) ? 5 : ((n % 7 == 0) ? 7 : 3))));
}
constexpr __host__ __device__ int radix_of(int n) { return radix_of_r<SMO_FFT_MAX_RADIX>(n); }
constexpr __host__ __device__ int stage_count(int n) { return n == 1 ? 0 : 1 + stage_count(n / radix_of(n)); }
constexpr bool fft_length_ok(int n) {
    while (n % 4 == 0) n /= 4;
    if (n % 2 == 0) n /= 2;
    while (n % 5 == 0) n /= 5;
    while (n % 7 == 0) n /= 7;
    while (n % 3 == 0) n /= 3;
    return n == 1;
}

// ---- radix butterflies (forward: e^{-i..}; INV: conjugate) ------------------------------------------------
template <bool INV> __device__ __forceinline__ cplx rot90(cplx a) { return INV ? mul_i(a) : mul_mi(a); }   // * (-/+ i)

template <int R, bool INV> struct Butterfly;
template <bool INV> struct Butterfly<2, INV> {
    static __device__ __forceinline__ void run(cplx (&v)[2]) {
        cplx a = v[0], b = v[1];
        v[0] = a + b; v[1] = a - b;
    }
};
template <bool INV> struct Butterfly<3, INV> {
    static __device__ __forceinline__ void run(cplx (&v)[3]) {
        const double S60 = 0.86602540378443864676372317075294;
        cplx t = v[1] + v[2];
        cplx m = mk(v[0].re - 0.5 * t.re, v[0].im - 0.5 * t.im);
        cplx d = S60 * (v[1] - v[2]);
        cplx r = rot90<INV>(d);
        v[0] = v[0] + t; v[1] = m + r; v[2] = m - r;
    }
};
template <bool INV> struct Butterfly<5, INV> {
    static __device__ __forceinline__ void run(cplx (&v)[5]) {
        const double C1 = 0.30901699437494742410229341718282, C2 = -0.80901699437494742410229341718282;      // cos(2 pi/5), cos(4 pi/5)
        const double S1 = 0.95105651629515357211643933337938, S2 = 0.58778525229247312916870595463907;       // sin(2 pi/5), sin(4 pi/5)
        const cplx t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
        const cplx a1 = mk(v[0].re + C1 * t1.re + C2 * t2.re, v[0].im + C1 * t1.im + C2 * t2.im);
        const cplx a2 = mk(v[0].re + C2 * t1.re + C1 * t2.re, v[0].im + C2 * t1.im + C1 * t2.im);
        const cplx r1 = rot90<INV>(mk(S1 * t3.re + S2 * t4.re, S1 * t3.im + S2 * t4.im));
        const cplx r2 = rot90<INV>(mk(S2 * t3.re - S1 * t4.re, S2 * t3.im - S1 * t4.im));
        v[0] = v[0] + t1 + t2; v[1] = a1 + r1; v[2] = a2 + r2; v[3] = a2 - r2; v[4] = a1 - r1;
    }
};
template <bool INV> struct Butterfly<7, INV> {
    static __device__ __forceinline__ void run(cplx (&v)[7]) {
        const double C1 = 0.62348980185873353052500488400424, C2 = -0.22252093395631440428890256449679, C3 = -0.90096886790241912623610231950745;   // cos(2 pi j/7)
        const double S1 = 0.78183148246802980870844452667406, S2 = 0.97492791218182360701813168299393, S3 = 0.43388373911755812047576833284836;    // sin(2 pi j/7)
        const cplx t1 = v[1] + v[6], t2 = v[2] + v[5], t3 = v[3] + v[4], d1 = v[1] - v[6], d2 = v[2] - v[5], d3 = v[3] - v[4];
        const cplx a1 = mk(v[0].re + C1 * t1.re + C2 * t2.re + C3 * t3.re, v[0].im + C1 * t1.im + C2 * t2.im + C3 * t3.im);
        const cplx a2 = mk(v[0].re + C2 * t1.re + C3 * t2.re + C1 * t3.re, v[0].im + C2 * t1.im + C3 * t2.im + C1 * t3.im);
        const cplx a3 = mk(v[0].re + C3 * t1.re + C1 * t2.re + C2 * t3.re, v[0].im + C3 * t1.im + C1 * t2.im + C2 * t3.im);
        const cplx r1 = rot90<INV>(mk(S1 * d1.re + S2 * d2.re + S3 * d3.re, S1 * d1.im + S2 * d2.im + S3 * d3.im));
        const cplx r2 = rot90<INV>(mk(S2 * d1.re - S3 * d2.re - S1 * d3.re, S2 * d1.im - S3 * d2.im - S1 * d3.im));
        const cplx r3 = rot90<INV>(mk(S3 * d1.re - S1 * d2.re + S2 * d3.re, S3 * d1.im - S1 * d2.im + S2 * d3.im));
        v[0] = v[0] + t1 + t2 + t3; v[1] = a1 + r1; v[2] = a2 + r2; v[3] = a3 + r3; v[4] = a3 - r3; v[5] = a2 - r2; v[6] = a1 - r1;
    }
};
template <bool INV> struct Butterfly<4, INV> {
    static __device__ __forceinline__ void run(cplx (&v)[4]) {
        cplx t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = rot90<INV>(v[1] - v[3]);
        v[0] = t0 + t2; v[1] = t1 + t3; v[2] = t0 - t2; v[3] = t1 - t3;
    }
};

template <bool INV> struct Butterfly<8, INV> {
    // radix 2 x radix 4: t = a_k + a_{k+4} feeds the even outputs, u = (a_k - a_{k+4}) w_8^k the odd ones
    static __device__ __forceinline__ void run(cplx (&v)[8]) {
        const double H = 0.70710678118654752440084436210485;
        cplx t[4], u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { t[k] = v[k] + v[k + 4]; u[k] = v[k] - v[k + 4]; }
        u[1] = INV ? mk(H * (u[1].re - u[1].im), H * (u[1].re + u[1].im)) : mk(H * (u[1].re + u[1].im), H * (u[1].im - u[1].re));
        u[2] = rot90<INV>(u[2]);
        u[3] = INV ? mk(-H * (u[3].re + u[3].im), H * (u[3].re - u[3].im)) : mk(H * (u[3].im - u[3].re), -H * (u[3].re + u[3].im));
        Butterfly<4, INV>::run(t);
        Butterfly<4, INV>::run(u);
#pragma unroll
        for (int m = 0; m < 4; ++m) { v[2 * m] = t[m]; v[2 * m + 1] = u[m]; }
    }
};

template <bool INV> __device__ __forceinline__ cplx twmul(cplx a, cplx w) { return INV ? mul_conj(a, w) : a * w; }

// One butterfly of the stage (L, N = current sub-length, S = stride), index j in [0, L/R):
// loads through `ld(pos)`, stores through `st(pos, value)`, pos in [0, L).
template <int L, int N, int S, bool INV, class Load, class Store>
__device__ __forceinline__ void stage_butterfly(int j, const cplx* __restrict__ tw, Load ld, Store st) {
    constexpr int R = radix_of(N);
    constexpr int M = N / R;
    const int p = j / S, q = j - p * S;          // S is a compile-time constant: shifts for 4^k
    cplx v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = ld(q + S * (p + k * M));
    Butterfly<R, INV>::run(v);
    st(q + S * (R * p), v[0]);
#pragma unroll
    for (int k = 1; k < R; ++k) {
        cplx o = v[k];
        if (M > 1) o = twmul<INV>(o, tw[p * S * k]);
        st(q + S * (R * p + k), o);
    }
}

// ---- whole-transform drivers ----------------------------------------------------------------------------
// NB FFTs of length L, leading dimension LD (>= L) in two LDS buffers; `nthr` threads cooperate.
// The caller provides first-stage loads  ld0(b, pos)  and last-stage stores  stN(b, pos, value)
// (typically global memory with zero-padding / truncation folded in).  Contains the barriers it needs,
// including one at the end of the last LDS-reading stage only if `final_barrier`.
template <int L, int N, int S, bool INV> struct MidStages {
    // runs the stages whose sub-length is N (> radix, i.e. not the last), LDS -> LDS, then recurses
    template <class StoreN>
    static __device__ __forceinline__ void run(cplx* cur, cplx* nxt, const cplx* tw, int NB, int LD, int tid, int nthr,
                                               StoreN stN) {
        constexpr int R = radix_of(N);
        constexpr int PER = L / R;
        if constexpr (N / R == 1) {
            // last stage: LDS -> caller's store
            for (int t = tid; t < NB * PER; t += nthr) {
                const int b = t / PER, j = t - b * PER;
                const cplx* x = cur + b * LD;
                stage_butterfly<L, N, S, INV>(j, tw, [&](int pos) { return x[pos]; },
                                              [&](int pos, cplx v) { stN(b, pos, v); });
            }
        } else {
            for (int t = tid; t < NB * PER; t += nthr) {
                const int b = t / PER, j = t - b * PER;
                const cplx* x = cur + b * LD;
                cplx* y = nxt + b * LD;
                stage_butterfly<L, N, S, INV>(j, tw, [&](int pos) { return x[pos]; }, [&](int pos, cplx v) { y[pos] = v; });
            }
            __syncthreads();
            MidStages<L, N / R, S * R, INV>::run(nxt, cur, tw, NB, LD, tid, nthr, stN);
        }
    }
};

template <int L, bool INV, class Load0, class StoreN>
__device__ __forceinline__ void fft_batch(cplx* bufA, cplx* bufB, const cplx* tw, int NB, int LD, int tid, int nthr,
                                          Load0 ld0, StoreN stN) {
    constexpr int R0 = radix_of(L);
    constexpr int PER = L / R0;
    static_assert(L / R0 > 1, "transform needs at least two stages");
    // first stage: caller's load -> LDS bufA
    for (int t = tid; t < NB * PER; t += nthr) {
        const int b = t / PER, j = t - b * PER;
        cplx* y = bufA + b * LD;
        stage_butterfly<L, L, 1, INV>(j, tw, [&](int pos) { return ld0(b, pos); }, [&](int pos, cplx v) { y[pos] = v; });
    }
    __syncthreads();
    MidStages<L, L / R0, R0, INV>::run(bufA, bufB, tw, NB, LD, tid, nthr, stN);
}


// ---------------------------------------------------------------------------------------------------------
// In-place variant (one LDS buffer, register-staged): every stage loads its butterflies' inputs into registers,
// synchronises, then stores.  Halves the LDS footprint (=> more workgroups per CU for the memory-bound 3-D passes).
//   NB FFTs x L points, NT threads, CNT = ceil(NB*L/R / NT) butterflies per thread per stage (compile time).
//   BFAST: consecutive lanes take consecutive FFTs b (strided passes: coalesced global access, use an odd LD);
//          otherwise consecutive lanes take consecutive butterflies j of one FFT (contiguous passes).
//   FIRST_LDS: ld0 reads the buffer itself (needs a barrier before the first store);
//   LAST_LDS : stN writes the buffer itself (needs a barrier before the last store).
// ---------------------------------------------------------------------------------------------------------
template <int NB, int PER, bool BFAST> __device__ __forceinline__ void split_index(int t, int& b, int& j) {
    if (BFAST) { j = t / NB; b = t - j * NB; }
    else       { b = t / PER; j = t - b * PER; }
}

// Twiddle table holding only exp(-2 pi i k / L) for k < L/2 (the other half is its negative): halves the table's LDS footprint.
struct HalfTwiddles {
    const cplx* t;
    int half;
    __device__ __forceinline__ cplx operator[](int i) const {
        const cplx w = t[i < half ? i : i - half];
        return i < half ? w : mk(-w.re, -w.im);
    }
};

template <int MAXR, int L, int N, int S, bool INV, int NB, int NT, bool BFAST, bool PRE_BARRIER, class TW, class Load, class Store>
__device__ __forceinline__ void inplace_stage_r(int tid, TW tw, Load ld, Store st) {
    constexpr int R = radix_of_r<MAXR>(N);
    constexpr int M = N / R;
    constexpr int PER = L / R;
    constexpr int TOTAL = NB * PER;
    constexpr int CNT = (TOTAL + NT - 1) / NT;
    cplx v[CNT][R];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int t = tid + i * NT;
        if (CNT * NT == TOTAL || t < TOTAL) {
            int b, j;
            split_index<NB, PER, BFAST>(t, b, j);
            const int p = j / S, q = j - p * S;
#pragma unroll
            for (int k = 0; k < R; ++k) v[i][k] = ld(b, q + S * (p + k * M));
        }
    }
    if (PRE_BARRIER) __syncthreads();
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int t = tid + i * NT;
        if (CNT * NT == TOTAL || t < TOTAL) {
            int b, j;
            split_index<NB, PER, BFAST>(t, b, j);
            const int p = j / S, q = j - p * S;
            Butterfly<R, INV>::run(v[i]);
            st(b, q + S * (R * p), v[i][0]);
#pragma unroll
            for (int k = 1; k < R; ++k) {
                cplx o = v[i][k];
                if (M > 1) o = twmul<INV>(o, tw[p * S * k]);
                st(b, q + S * (R * p + k), o);
            }
        }
    }
}

template <int L, int N, int S, bool INV, int NB, int NT, bool BFAST, bool PRE_BARRIER, class TW, class Load, class Store>
__device__ __forceinline__ void inplace_stage(int tid, TW tw, Load ld, Store st) {
    inplace_stage_r<SMO_FFT_MAX_RADIX, L, N, S, INV, NB, NT, BFAST, PRE_BARRIER>(tid, tw, ld, st);
}

struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };

// Where element (transform b, position pos) of a tile lives in the LDS buffer.
//   RowMajor{LD}: b * LD + pos — one row per transform (pad LD against bank conflicts);
//   PosMajor<NB>: pos * NB + b — the NB transforms of a tile interleaved.  With BFAST lane order (consecutive lanes = consecutive
//   transforms, then consecutive butterflies) every stage READ of a wave is then ONE contiguous 1-KB run — conflict-free for
//   ds_read_b128 whatever the stage — and all stage writes but the stride-R ones of the first stage are contiguous as well
//   (tools/lds_conflict_model.py: 26-42 % fewer LDS-array cycles per x-pass tile than the best row padding).
struct RowMajor {
    int LD;
    __device__ __forceinline__ int operator()(int b, int pos) const { return b * LD + pos; }
};
template <int NB> struct PosMajor {
    __device__ __forceinline__ int operator()(int b, int pos) const { return pos * NB + b; }
};

template <int L, int N, int S, bool INV, int NB, int NT, bool BFAST, bool LAST_LDS, int MAXR = SMO_FFT_MAX_RADIX> struct InplaceTail {
    template <class IX, class TW, class StoreN, class PreLast>
    static __device__ __forceinline__ void run_ix(cplx* buf, IX ix, int tid, TW tw, StoreN stN, PreLast pre) {
        constexpr int R = radix_of_r<MAXR>(N);
        auto ldL = [&](int b, int pos) { return buf[ix(b, pos)]; };
        if constexpr (N / R == 1) {
            pre();
            inplace_stage_r<MAXR, L, N, S, INV, NB, NT, BFAST, LAST_LDS>(tid, tw, ldL, stN);
        } else {
            inplace_stage_r<MAXR, L, N, S, INV, NB, NT, BFAST, true>(tid, tw, ldL, [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; });
            __syncthreads();
            InplaceTail<L, N / R, S * R, INV, NB, NT, BFAST, LAST_LDS, MAXR>::run_ix(buf, ix, tid, tw, stN, pre);
        }
    }
    template <class TW, class StoreN, class PreLast>
    static __device__ __forceinline__ void run(cplx* buf, int LD, int tid, TW tw, StoreN stN, PreLast pre) {
        run_ix(buf, RowMajor{LD}, tid, tw, stN, pre);
    }
};

// `pre` runs right before the last stage: the place to issue global loads whose results are needed after the transform (they are
// then in flight during the last stage instead of being waited for after it, and live in registers for one stage only).
template <int L, bool INV, int NB, int NT, bool BFAST, bool FIRST_LDS, bool LAST_LDS, class IX, class TW, class Load0, class StoreN, class PreLast = NoPrefetch>
__device__ __forceinline__ void fft_inplace_ix(cplx* buf, IX ix, TW tw, int tid, Load0 ld0, StoreN stN, PreLast pre = PreLast()) {
    constexpr int R0 = radix_of(L);
    static_assert(L / R0 > 1, "transform needs at least two stages");
    inplace_stage<L, L, 1, INV, NB, NT, BFAST, FIRST_LDS>(tid, tw, ld0, [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; });
    __syncthreads();
    InplaceTail<L, L / R0, R0, INV, NB, NT, BFAST, LAST_LDS>::run_ix(buf, ix, tid, tw, stN, pre);
}
template <int L, bool INV, int NB, int NT, bool BFAST, bool FIRST_LDS, bool LAST_LDS, class TW, class Load0, class StoreN, class PreLast = NoPrefetch>
__device__ __forceinline__ void fft_inplace(cplx* buf, int LD, TW tw, int tid, Load0 ld0, StoreN stN, PreLast pre = PreLast()) {
    fft_inplace_ix<L, INV, NB, NT, BFAST, FIRST_LDS, LAST_LDS>(buf, RowMajor{LD}, tw, tid, ld0, stN, pre);
}

// All stages but the last one (LDS -> LDS, a barrier after each): for callers that fuse their own work into the last stage, whose
// butterfly j reads and writes the same positions j + (L/R) k — thread-local, so it needs no barrier before what follows on those values.
template <int L, int N, int S, bool INV, int NB, int NT, bool BFAST, int MAXR = SMO_FFT_MAX_RADIX> struct InplaceHead {
    template <class IX, class TW>
    static __device__ __forceinline__ void run(cplx* buf, IX ix, int tid, TW tw) {
        constexpr int R = radix_of_r<MAXR>(N);
        if constexpr (N / R > 1) {
            auto ldL = [&](int b, int pos) { return buf[ix(b, pos)]; };
            inplace_stage_r<MAXR, L, N, S, INV, NB, NT, BFAST, true>(tid, tw, ldL, [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; });
            __syncthreads();
            InplaceHead<L, N / R, S * R, INV, NB, NT, BFAST, MAXR>::run(buf, ix, tid, tw);
        }
    }
};
template <int L, int MAXR = SMO_FFT_MAX_RADIX> constexpr int last_radix() { int n = L; while (n / radix_of_r<MAXR>(n) > 1) n /= radix_of_r<MAXR>(n); return n; }

// (MAXR first: the radix policy of the calling kernel)
template <int MAXR, int L, bool INV, int NB, int NT, bool BFAST, bool FIRST_LDS, class IX, class TW, class Load0>
__device__ __forceinline__ void fft_inplace_head_r(cplx* buf, IX ix, TW tw, int tid, Load0 ld0) {
    constexpr int R0 = radix_of_r<MAXR>(L);
    static_assert(L / R0 > 1, "transform needs at least two stages");
    inplace_stage_r<MAXR, L, L, 1, INV, NB, NT, BFAST, FIRST_LDS>(tid, tw, ld0, [&](int b, int pos, cplx v) { buf[ix(b, pos)] = v; });
    __syncthreads();
    InplaceHead<L, L / R0, R0, INV, NB, NT, BFAST, MAXR>::run(buf, ix, tid, tw);
}
template <int L, bool INV, int NB, int NT, bool BFAST, bool FIRST_LDS, class IX, class TW, class Load0>
__device__ __forceinline__ void fft_inplace_head(cplx* buf, IX ix, TW tw, int tid, Load0 ld0) {
    fft_inplace_head_r<SMO_FFT_MAX_RADIX, L, INV, NB, NT, BFAST, FIRST_LDS>(buf, ix, tw, tid, ld0);
}


// ---------------------------------------------------------------------------------------------------------
// Run-time-length variant: the transform length is a kernel argument, the radices are the prime factors of L found on the host
// (any_plan); stages of radix 2, 3, 4, 5, 7 run the butterflies above, one per thread, a stage whose radix is a larger prime is evaluated
// one output per thread as a direct sum — any length works (a prime factor p costs p multiply-adds per point), at a multiple of the cost
// of the compile-time chains above.  Used where a size has no tuned instantiation: csrc/kdyn_any.hpp (any even Npts of the 3-D case), the
// SH23 any-length kernels, the SHB23 kernels' NH = 0 form.
// ---------------------------------------------------------------------------------------------------------
struct AnyPlan {
    int L;            // transform length
    int nst;          // Stockham stages
    int r[20];        // their radices (product = L)
};

inline AnyPlan any_plan(int L) {
    AnyPlan p{};
    p.L = L;
    int n = L;
    while (n % 4 == 0) { p.r[p.nst++] = 4; n /= 4; }
    if (n % 2 == 0) { p.r[p.nst++] = 2; n /= 2; }
    for (int f = 3; n > 1; f += 2)
        while (n % f == 0) { p.r[p.nst++] = f; n /= f; }
    return p;
}

// One butterfly of radix R in {2, 3, 4, 5, 7}: reads its R inputs, writes its R outputs (the compile-time butterflies above).
template <int R, bool INV>
__device__ __forceinline__ void any_bfly(const cplx* x, cplx* y, int xs, int s, const cplx* tw, int ps, bool twid) {
    cplx v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = x[k * xs];
    Butterfly<R, INV>::run(v);
    y[0] = v[0];
#pragma unroll
    for (int k = 1; k < R; ++k) y[s * k] = twid ? twmul<INV>(v[k], tw[ps * k]) : v[k];
}

// NB transforms of length L, element (b, pos) at b * L + pos, in `src`; ping-pong with `dst`; returns the buffer that holds the result.
// Stage invariant n * s == L (see the top of this file):  y[q + s (R p + j)] = w_n^{p j} sum_k x[q + s (p + k n/R)] w_R^{j k}.
// tw[k] = exp(-2 pi i k / L), in the LDS like the buffers (the callers copy it there: one table load per multiply-add from global
// memory instead made the 3-D passes 1.5x slower).  Stages of radix 2, 3, 4, 5, 7 run one BUTTERFLY per thread (R reads and R writes for R
// outputs); a stage whose radix is a larger prime runs one OUTPUT per thread as a direct sum over the radix (p multiply-adds per point, no
// registers proportional to p).  Ends with a barrier.
template <bool INV>
__device__ __forceinline__ cplx* any_fft(cplx* src, cplx* dst, const cplx* tw, const AnyPlan& pl, int NB, int tid, int nthr) {
    const int L = pl.L;
    int n = L, s = 1;
    for (int st = 0; st < pl.nst; ++st) {
        const int R = pl.r[st], M = n / R, xs = s * M;
        if (R <= 5 || R == 7) {
            const int PER = L / R;
            for (int t = tid; t < NB * PER; t += nthr) {
                const int b = t / PER, jj = t - b * PER, p = jj / s, q = jj - p * s;
                const cplx* x = src + (size_t)b * L + q + s * p;
                cplx* y = dst + (size_t)b * L + q + s * R * p;
                switch (R) {
                    case 2: any_bfly<2, INV>(x, y, xs, s, tw, p * s, M > 1); break;
                    case 3: any_bfly<3, INV>(x, y, xs, s, tw, p * s, M > 1); break;
                    case 4: any_bfly<4, INV>(x, y, xs, s, tw, p * s, M > 1); break;
                    case 5: any_bfly<5, INV>(x, y, xs, s, tw, p * s, M > 1); break;
                    default: any_bfly<7, INV>(x, y, xs, s, tw, p * s, M > 1); break;
                }
            }
        } else {
            const int wstep = L / R;
            for (int t = tid; t < NB * L; t += nthr) {
                const int b = t / L, o = t - b * L;
                const int q = o % s, rj = o / s, j = rj % R, p = rj / R;
                const cplx* x = src + (size_t)b * L + q + s * p;
                cplx acc = x[0];
                int e = 0;                                  // (j k) mod R
                for (int k = 1; k < R; ++k) {
                    e += j; if (e >= R) e -= R;
                    const cplx w = tw[e * wstep], v = x[k * xs];
                    acc = acc + (INV ? mul_conj(v, w) : v * w);
                }
                if (M > 1 && j) { const cplx w = tw[p * s * j]; acc = INV ? mul_conj(acc, w) : acc * w; }
                dst[t] = acc;
            }
        }
        __syncthreads();
        cplx* sw = src; src = dst; dst = sw;
        n = M; s *= R;
    }
    return src;
}

// copy the twiddle table into the LDS (no barrier: the caller's next one covers it)
__device__ __forceinline__ void any_load_tw(cplx* tw_lds, const cplx* __restrict__ tw_g, int L, int tid, int nthr) {
    for (int i = tid; i < L; i += nthr) tw_lds[i] = tw_g[i];
}

}  // namespace smo

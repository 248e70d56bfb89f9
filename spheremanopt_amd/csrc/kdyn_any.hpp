// KDYN, any grid length: the same four passes as kdyn.hip's tuned kernels (same buffers, layouts, per-mode algebra), with the transform
// length G = 3 Npts / 2 a RUN-TIME value.  Included by kdyn.hip inside its namespace, after the layout helpers it shares.
//
// Why: the tuned kernels are compile-time instantiations for G with factors 2, 3, 5, 7; the reference, through Dedalus / FFTW
// (FWD_Solve_KDyn.py:362-450 builds a Fourier basis for whatever Npts it is handed, :1029 the script's Npts), takes any even Npts.
// These kernels are the path of every other size: unfused where fusion needed compile-time radices (no register-resident middle
// section, no fused next z pass), a Stockham chain whose radices are the prime factors of G found at context creation (fft_lds.hpp,
// any_fft: butterfly stages for the radices 2, 3, 4, 5, 7, a direct sum per output for a larger prime p — p multiply-adds per point: 11,
// 13, 17 ... are cheap, a large prime G/3 is the O(G p) worst case).  Odd G (Npts = 2 mod 4) is covered too: the last (y,z) line of a plane then has
// no partner in the two-real-lines-per-complex-transform packing and is paired with zeros.
// Slower than the tuned path by design (short runs, LDS ping-pong, no fusion); results agree with the oracle like the tuned kernels' do.
#pragma once

// (AnyPlan / any_plan / any_fft: the run-time-length Stockham chain, fft_lds.hpp)

// ---- z pass, inverse: coefficients -> Tz (kd_z_inverse) -------------------------------------------------------------------
__global__ __launch_bounds__(1024) void kda_z_inverse(const cplx* __restrict__ in, cplx* __restrict__ out, const cplx* __restrict__ tw, Geom g,
                                                     AnyPlan pl, int mode, int NBT) {
    extern __shared__ cplx any_lds[];
    const int L = pl.L, NB = 3 * NBT, tid = threadIdx.x, NT = blockDim.x;
    cplx *A = any_lds, *B = any_lds + (size_t)NB * L, *tws = any_lds + (size_t)2 * NB * L;
    any_load_tw(tws, tw, L, tid, NT);
    const int nrt = g.al * g.m, rt0 = blockIdx.x * NBT;
    const size_t cs = (size_t)nrt * g.m;
    for (int t = tid; t < NB * L; t += NT) {                       // raw rows, zero-padded to L
        const int b = t / L, pos = t - b * L, tt = b / 3, c = b - 3 * tt, rt = rt0 + tt;
        const int idx = wrap_pos(pos, g);
        B[t] = (rt < nrt && idx >= 0) ? in[c * cs + (size_t)rt * g.m + idx] : mk(0, 0);
    }
    __syncthreads();
    for (int t = tid; t < NB * L; t += NT) {
        const int b = t / L, pos = t - b * L, tt = b / 3, c = b - 3 * tt, rt = rt0 + tt;
        const int idx = wrap_pos(pos, g);
        cplx v = B[t];
        if (mode != ZI_PLAIN && rt < nrt && idx >= 0) {
            const int ixl = rt / g.m, iy = rt - ixl * g.m;
            const double k[3] = {(double)(g.ix0 + ixl), wavenumber(iy, g), wavenumber(idx, g)};
            if (mode == ZI_SCALE) {
                v = (g.dt * (1.0 / g.dt + 0.5 * (k[0] * k[0] + k[1] * k[1] + k[2] * k[2]) / g.Rm)) * v;
            } else {                                                // (i k x V)_c
                const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                const cplx v1 = B[(tt * 3 + c1) * L + pos], v2 = B[(tt * 3 + c2) * L + pos];
                v = mul_i(mk(k[c1] * v2.re - k[c2] * v1.re, k[c1] * v2.im - k[c2] * v1.im));
            }
        }
        A[t] = v;
    }
    __syncthreads();
    const cplx* R = any_fft<true>(A, B, tws, pl, NB, tid, NT);
    for (int t = tid; t < NB * L; t += NT) {
        const int b = t / L, pos = t - b * L, tt = b / 3, c = b - 3 * tt, rt = rt0 + tt;
        if (rt < nrt) out[zs_off(c, rt, pos, g)] = R[t];
    }
}

// ---- z pass, forward: Tz -> coefficients, with the per-mode step (kd_z_forward; no fused next pass) --------------------------
__global__ __launch_bounds__(1024) void kda_z_forward(const cplx* __restrict__ inA, cplx* out0, const cplx* state0 /* may alias out0 */,
                                                     const cplx* __restrict__ snap, const cplx* __restrict__ tw, Geom g, AnyPlan pl, int mode,
                                                     int NBT, double scale, int integrated) {
    extern __shared__ cplx any_lds[];
    const int L = pl.L, NB = 3 * NBT, tid = threadIdx.x, NT = blockDim.x;
    cplx *A = any_lds, *B = any_lds + (size_t)NB * L, *tws = any_lds + (size_t)2 * NB * L;
    any_load_tw(tws, tw, L, tid, NT);
    const int nrt = g.al * g.m, rt0 = blockIdx.x * NBT;
    const size_t cs = (size_t)nrt * g.m;
    for (int t = tid; t < NB * L; t += NT) {
        const int b = t / L, pos = t - b * L, tt = b / 3, c = b - 3 * tt, rt = rt0 + tt;
        A[t] = (rt < nrt) ? inA[zs_off(c, rt, pos, g)] : mk(0, 0);
    }
    __syncthreads();
    const cplx* R = any_fft<false>(A, B, tws, pl, NB, tid, NT);
    for (int t = tid; t < NBT * g.m; t += NT) {                      // per retained mode: thread <-> (tt, iz)
        const int tt = t / g.m, iz = t - tt * g.m, rt = rt0 + tt;
        if (rt >= nrt) continue;
        const int ixl = rt / g.m, iy = rt - ixl * g.m;
        const int pos = (iz <= g.kmax) ? iz : iz + (g.G - g.m);
        const double k[3] = {(double)(g.ix0 + ixl), wavenumber(iy, g), wavenumber(iz, g)};
        const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
        const double D = k2 / g.Rm, alpha = 1.0 / g.dt + 0.5 * D, beta = 1.0 / g.dt - 0.5 * D;
        const size_t e = (size_t)rt * g.m + iz;
        cplx E[3], V0[3], V1[3];
        for (int c = 0; c < 3; ++c) E[c] = scale * R[(tt * 3 + c) * L + pos];
        if (mode == ZF_PLAIN) {
            for (int c = 0; c < 3; ++c) out0[c * cs + e] = E[c];
            continue;
        }
        if (mode == ZF_NU) {                                        // nu^ = -dt P(E)
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
        for (int c = 0; c < 3; ++c) V0[c] = state0[c * cs + e];
        if (mode == ZF_FWD_UPDATE) {
            cplx F[3];                                              // N^ = i k x E^
            for (int c = 0; c < 3; ++c) {
                const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                F[c] = mul_i(mk(k[c1] * E[c2].re - k[c2] * E[c1].re, k[c1] * E[c2].im - k[c2] * E[c1].im));
            }
            cnab_mode(k, k2, alpha, beta, V0, F, V1);
        } else {
            if (integrated)
                for (int c = 0; c < 3; ++c) { const cplx bf = snap[c * cs + e]; E[c] = mk(E[c].re - 2.0 * bf.re, E[c].im - 2.0 * bf.im); }
            cnab_mode(k, k2, alpha, beta, V0, E, V1);
        }
        for (int c = 0; c < 3; ++c) out0[c * cs + e] = V1[c];
    }
}

// ---- y pass: Tz (y-side exchange layout) <-> Ty; ZT consecutive z columns per workgroup (kd_y_pass) --------------------------
__global__ __launch_bounds__(1024) void kda_y_pass(const cplx* __restrict__ in, cplx* __restrict__ out, const cplx* __restrict__ tw, Geom g,
                                                  AnyPlan pl, int inv, int ZT) {
    extern __shared__ cplx any_lds[];
    const int L = pl.L, tid = threadIdx.x, NT = blockDim.x;
    cplx *A = any_lds, *B = any_lds + (size_t)ZT * L, *tws = any_lds + (size_t)2 * ZT * L;
    any_load_tw(tws, tw, L, tid, NT);
    const int ntile = (g.Gzl + ZT - 1) / ZT;
    const int o = blockIdx.x / ntile, z0 = (blockIdx.x - o * ntile) * ZT;      // o = c * a + kx
    const int c = o / g.a, kx = o - c * g.a;
    const size_t zrow = ys_row0(c, kx, g) + z0, trow = (size_t)o * g.typ + z0;
    for (int t = tid; t < ZT * L; t += NT) {                       // b fastest: consecutive lanes read consecutive z
        const int pos = t / ZT, b = t - pos * ZT;
        cplx v = mk(0, 0);
        if (z0 + b < g.Gzl) {
            if (inv) { const int idx = wrap_pos(pos, g); if (idx >= 0) v = in[zrow + (size_t)idx * g.Gzl + b]; }
            else v = in[trow + (size_t)pos * g.Gzl + b];
        }
        A[b * L + pos] = v;
    }
    __syncthreads();
    const cplx* R = inv ? any_fft<true>(A, B, tws, pl, ZT, tid, NT) : any_fft<false>(A, B, tws, pl, ZT, tid, NT);
    for (int t = tid; t < ZT * L; t += NT) {
        const int pos = t / ZT, b = t - pos * ZT;
        if (z0 + b >= g.Gzl) continue;
        if (inv) out[trow + (size_t)pos * g.Gzl + b] = R[b * L + pos];
        else { const int idx = wrap_pos(pos, g); if (idx >= 0) out[zrow + (size_t)idx * g.Gzl + b] = R[b * L + pos]; }
    }
}

// ---- x pass: Hermitian half spectrum <-> real lines, two lines per complex transform, T = 2 HP flat (y,z) points per workgroup
// (kd_x_pass).  Modes as in kdyn.hip; the fused adjoint pass sends its two field groups through the buffers one after the other
// (omega's grid values stay in the third buffer).  The internal U field is kept in the flat grid layout here (Geom::utile = 0).
__global__ __launch_bounds__(1024) void kda_x_pass(XSpec sp, const double* __restrict__ gridU, double* gridOut, const cplx* __restrict__ tw, Geom g,
                                                  AnyPlan pl, int mode, int HP) {
    extern __shared__ cplx any_lds[];
    const int L = pl.L, NB = 3 * HP, tid = threadIdx.x, NT = blockDim.x;
    const size_t plane = (size_t)g.G * g.Gzl, i0 = (size_t)blockIdx.x * 2 * HP;
    cplx* P[3] = {any_lds, any_lds + (size_t)NB * L, any_lds + (size_t)2 * NB * L};      // (the third only in the adjoint pass)
    cplx* tws = any_lds + (size_t)(mode == X_FUSED_ADJ ? 3 : 2) * NB * L;
    any_load_tw(tws, tw, L, tid, NT);
    // (the lambdas are always inlined: no kernel of this library may call a device function — csrc/shb23.hip `dct2<0>`)
    // b = c * HP + p; the second line of the last pair is absent when the plane has an odd number of points
    auto ok1 = [&](int p) __attribute__((always_inline)) { return i0 + 2 * p < plane; };
    auto ok2 = [&](int p) __attribute__((always_inline)) { return i0 + 2 * p + 1 < plane; };
    auto load_spec = [&](const cplx* src, cplx* dst) __attribute__((always_inline)) {              // Hermitian-extended, zero-padded lines X1 + i X2
        for (int t = tid; t < NB * L; t += NT) dst[t] = mk(0, 0);
        __syncthreads();
        for (int t = tid; t < NB * g.a; t += NT) {                  // p fastest
            const int p = t % HP, r = t / HP, c = r % 3, kx = r / 3;
            if (!ok1(p)) continue;
            const cplx* q = src + tx_off(c, kx, i0 + 2 * p, g);
            const cplx X1 = q[0], X2 = ok2(p) ? q[1] : mk(0, 0);
            cplx* row = dst + (size_t)(c * HP + p) * L;
            if (kx == 0) row[0] = mk(X1.re, X2.re);                 // c2r ignores the imaginary part of kx = 0
            else {
                row[kx] = mk(X1.re - X2.im, X1.im + X2.re);
                row[L - kx] = mk(X1.re + X2.im, X2.re - X1.im);
            }
        }
        __syncthreads();
    };
    auto split_store = [&](const cplx* R, cplx* dst, bool acc) __attribute__((always_inline)) {    // spectra of the two real lines, kx = 0..a-1
        for (int t = tid; t < NB * g.a; t += NT) {
            const int p = t % HP, r = t / HP, c = r % 3, kx = r / 3;
            if (!ok1(p)) continue;
            const cplx* row = R + (size_t)(c * HP + p) * L;
            const cplx Zk = row[kx], Zm = conj(row[kx == 0 ? 0 : L - kx]);
            cplx* q = dst + tx_off(c, kx, i0 + 2 * p, g);
            cplx v0 = 0.5 * (Zk + Zm), v1 = mul_mi(0.5 * (Zk - Zm));
            if (acc) { v0 = v0 + q[0]; if (ok2(p)) v1 = v1 + q[1]; }
            q[0] = v0;
            if (ok2(p)) q[1] = v1;
        }
    };
    auto grid_pair = [&](int c, int x, int p) __attribute__((always_inline)) -> cplx {            // the internal U field, flat grid layout
        const double* q = gridU + grid_off(c, x, i0 + 2 * p, g);
        return mk(q[0], ok2(p) ? q[1] : 0.0);
    };
    // X x Y at every point of the tile, both lines of a pair at once (.re / .im); in place on Y's buffer unless dst is given
    auto cross = [&](const cplx* X, const cplx* Y, cplx* dst, bool x_is_U, bool y_is_U) __attribute__((always_inline)) {
        for (int t = tid; t < HP * L; t += NT) {
            const int p = t % HP, x = t / HP;
            if (!ok1(p)) { for (int c = 0; c < 3; ++c) dst[(size_t)(c * HP + p) * L + x] = mk(0, 0); continue; }
            cplx a[3], b[3];
            for (int c = 0; c < 3; ++c) {
                a[c] = x_is_U ? grid_pair(c, x, p) : X[(size_t)(c * HP + p) * L + x];
                b[c] = y_is_U ? grid_pair(c, x, p) : Y[(size_t)(c * HP + p) * L + x];
            }
            for (int c = 0; c < 3; ++c) {
                const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                dst[(size_t)(c * HP + p) * L + x] = mk(a[c1].re * b[c2].re - a[c2].re * b[c1].re, a[c1].im * b[c2].im - a[c2].im * b[c1].im);
            }
        }
        __syncthreads();
    };

    if (mode == X_FROM_GRID) {
        for (int t = tid; t < NB * L; t += NT) {
            const int p = t % HP, r = t / HP, x = r % L, c = r / L;
            cplx v = mk(0, 0);
            if (ok1(p)) { const double* q = gridU + grid_off(c, x, i0 + 2 * p, g); v = mk(q[0], ok2(p) ? q[1] : 0.0); }
            P[0][(size_t)(c * HP + p) * L + x] = v;
        }
        __syncthreads();
        split_store(any_fft<false>(P[0], P[1], tws, pl, NB, tid, NT), sp.outA, false);
        return;
    }
    load_spec(sp.inA, P[0]);
    cplx* W = any_fft<true>(P[0], P[1], tws, pl, NB, tid, NT);      // field group A on the grid
    cplx* F = (W == P[0]) ? P[1] : P[0];                            // free buffer
    if (mode == X_TO_GRID) {
        for (int t = tid; t < NB * L; t += NT) {
            const int p = t % HP, r = t / HP, x = r % L, c = r / L;
            if (!ok1(p)) continue;
            const cplx v = W[(size_t)(c * HP + p) * L + x];
            double* q = gridOut + grid_off(c, x, i0 + 2 * p, g);
            q[0] = v.re;
            if (ok2(p)) q[1] = v.im;
        }
        return;
    }
    if (mode == X_FUSED_FWD) {                                      // EMF = U x B
        cross(nullptr, W, F, true, false);
        split_store(any_fft<false>(F, W, tws, pl, NB, tid, NT), sp.outA, false);
        return;
    }
    // adjoint: F1 = omega x U -> out A;  F2' = omega x B_f -> added to the running sum out B
    cross(W, nullptr, F, false, true);
    split_store(any_fft<false>(F, P[2], tws, pl, NB, tid, NT), sp.outA, false);
    __syncthreads();
    load_spec(sp.inB, F);
    cplx* Bf = any_fft<true>(F, P[2], tws, pl, NB, tid, NT);
    cplx* F2 = (Bf == F) ? P[2] : F;
    cross(W, Bf, F2, false, false);
    split_store(any_fft<false>(F2, Bf, tws, pl, NB, tid, NT), sp.outB, true);
}

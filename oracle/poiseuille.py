"""TEST INFRASTRUCTURE (CPU oracle) — stratified plane-Poiseuille optimal mixing, "Discrete" formulation.

NumPy restatement of Example_Problems/Bounded_Domain(Cheby)/Optimal_Mixing/FWD_Solve_Poiseuille.py:
    transforms                 :44-89      weightMatrixDisc            :91-118
    Inner_Prod_Discrete        :282-299    FWD_Solve_Discrete          :777-1155
    ADJ_Solve_Discrete         :1320-1659  (mix-norm LBVP :1053-1124, :1432-1447, :1556-1584)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED: the reference hand-steps two Dedalus-v2 LBVPs (`pencil_matsolvers`, `pre_left`, `L_exp`); Dedalus is not
installable here and the reference holds no fixtures for this path.  What the LBVP solve does is restated from the tau method
of SURVEY.md Appendix A.0-8: every differential equation is converted T -> U ("Pre"), loses its last row, and the boundary /
gauge conditions fill the freed rows.  The solve map  S_k : (rhs_u, rhs_v, rhs_rho) -> (u, v, rho, uz, vz, rhoz)  does not depend
on where the rows are placed, and the reference's transposed solve  P^L^H A^-H P^R^H  is exactly S_k^H, so the oracle (and the
device code) work with the dense S_k per x-wavenumber.  Internal consistency is pinned by the Taylor test (tests/test_oracle.py).

Layout: the x basis is a COMPLEX Fourier basis (grid_dtype complex128, :338): native modes n = 0..kmax, -kmax..-1 with
kmax = (Nx-1)//2 (Nyquist dropped), k = 2 pi n / Lx.  Coefficient arrays are (Nxc, Nz) complex, grids (Nx, Nz).
Flat vectors are [u.flatten(), v.flatten()] of the (Nx, Nz) grids, z fastest (Field_to_Vec :160-207).
"""
import numpy as np
from scipy.fft import dct
from scipy.special import erf


def cheb_pre(N):
    """T -> U conversion matrix."""
    P = np.zeros((N, N))
    for n in range(N):
        P[n, n] = 1. if n == 0 else 0.5
        if n + 2 < N:
            P[n, n + 2] = -0.5
    return P


def cheb_diff(N, stretch=1.0):
    """T -> T differentiation (row 0 halved), d/dz on an interval of half-length `stretch` (POIS:1489-1496)."""
    D = np.zeros((N, N))
    for i in range(N):
        for j in range(i + 1, N):
            D[i, j] = 2. * j * ((j - i) % 2)
    D[0] /= 2.
    return D / stretch


def cheb_mult(N, coeffs):
    """Matrix of f(z)* in the T basis, f = sum_j coeffs[j] T_j, truncated to N modes: T_j T_m = (T_{m+j} + T_{|m-j|}) / 2."""
    M = np.zeros((N, N))
    for j, fj in enumerate(coeffs):
        if fj == 0.:
            continue
        for m in range(N):
            for t in (m + j, abs(m - j)):
                if t < N:
                    M[t, m] += 0.5 * fj
    return M


def cheb_integ(N, stretch=1.0):
    """integ(T_n) over the interval: 2/(1-n^2) for even n."""
    n = np.arange(N)
    return stretch * np.where(n % 2 == 0, 2. / np.where(n == 1, 1., 1. - n.astype(float) ** 2), 0.)


class PoiseuilleOracle:
    def __init__(self, Nx=24, Nz=24, Re=500., Ri=0.05, dt=5e-3, N_ITERS=10, s=0, Prandtl=1., delta=0.125, Lx=4. * np.pi):
        self.Nx, self.Nz, self.Re, self.Ri, self.Pe = int(Nx), int(Nz), float(Re), float(Ri), float(Re) * Prandtl
        self.dt, self.N_ITERS, self.s, self.delta, self.Lx = float(dt), int(N_ITERS), int(s), float(delta), float(Lx)
        self.kmax = (self.Nx - 1) // 2
        self.n = np.concatenate([np.arange(0, self.kmax + 1), np.arange(-self.kmax, 0)])       # native wavenumbers kept
        self.k = 2. * np.pi * self.n / self.Lx
        self.Nxc = len(self.n)
        self.z = -np.cos(np.pi * (np.arange(self.Nz) + 0.5) / self.Nz)                          # Gauss grid on [-1, 1]
        self.V = self.Lx * 2.                                                                   # domain.hypervolume
        # weightMatrixDisc (:91-118): first-order differences of the Gauss grid times dx
        dz = np.empty(self.Nz)
        dz[0] = self.z[1] - self.z[0]
        dz[1:] = self.z[1:] - self.z[:-1]
        self.Wz = dz
        self.W = np.tile(dz * (self.Lx / self.Nx), (self.Nx, 1))
        # de-aliasing mask (:896-909): |k| < (2 pi/Lx) * (Nx0//2), nz < Nz0 with Nx0 = 2Nx//3, Nz0 = 2Nz//3
        self.Nx0, self.Nz0 = 2 * self.Nx // 3, 2 * self.Nz // 3
        self.DA = ((np.abs(self.n) < self.Nx0 // 2)[:, None] & (np.arange(self.Nz) < self.Nz0)[None, :]).astype(float)
        self.a = self.Nx0 // 2                                                                  # de-aliased modes: |n| < a
        self.ax = self.kmax + 1                                                                 # non-negative modes carried (n = 0..kmax)
        self.Dz = cheb_diff(self.Nz)
        self._S, self._SMN = {}, {}
        self.stack = None

    # ---- transforms (:44-89) --------------------------------------------------------------------------------------------------
    def _xf(self, g):                 # grid -> x coefficients (Dedalus forward FFT, amplitude-normalised, Nyquist dropped)
        F = np.fft.fft(g, axis=0) / self.Nx
        return F[self.n % self.Nx]

    def _xb(self, c):                 # x coefficients -> grid
        F = np.zeros((self.Nx,) + c.shape[1:], dtype=complex)
        F[self.n % self.Nx] = c
        return np.fft.ifft(F, axis=0) * self.Nx

    def transform(self, g):
        b = dct(self._xf(np.asarray(g, dtype=complex)), type=2, axis=1) / self.Nz
        b[:, 0] *= 0.5
        b[:, 1::2] *= -1
        return b

    def transformInverse(self, c):
        c = np.array(c, dtype=complex)
        c[:, 1::2] *= -1
        c[:, 1:] *= 0.5
        return self._xb(dct(c, type=3, axis=1))

    def transformAdjoint(self, c):
        c = np.array(c, dtype=complex)
        c[:, 0] *= 0.5
        c[:, 1::2] *= -1
        c[:, 0] *= np.sqrt(4 * self.Nz)
        c[:, 1:] *= np.sqrt(2 * self.Nz)
        b = dct(c, type=3, norm='ortho', axis=1) / self.Nz
        return self._xb(b) / self.Nx

    def transformInverseAdjoint(self, g):
        b = self._xf(np.asarray(g, dtype=complex)) * self.Nx
        b = dct(b, type=2, norm='ortho', axis=1) * np.sqrt(self.Nz)
        b[:, 1:] *= np.sqrt(2)
        b[:, 1:] *= 0.5
        b[:, 1::2] *= -1
        return b

    # ---- tau solves -----------------------------------------------------------------------------------------------------------
    def solve_map(self, n):
        """S_k (6Nz x 3Nz complex): (rhs_u, rhs_v, rhs_rho) T-coefficients -> (u, v, rho, uz, vz, rhoz) of the LBVP :818-841."""
        if n in self._S:
            return self._S[n]
        N, k, a0 = self.Nz, 2. * np.pi * n / self.Lx, 1. / self.dt
        Pre, D, I = cheb_pre(N), self.Dz, np.eye(N)
        M1 = cheb_mult(N, [0.5, 0., -0.5])                       # (1 - z^2) = T0/2 - T2/2
        M2 = cheb_mult(N, [0., -2.])                             # -2 z
        U, Vv, R, UZ, VZ, RZ, P = (slice(i * N, (i + 1) * N) for i in range(7))
        nv = 7 * N + (1 if n == 0 else 0)                        # Fb (constant in z) is an unknown only where it is not set to 0
        A = np.zeros((nv, nv), dtype=complex)
        B = np.zeros((nv, 3 * N), dtype=complex)
        row = 0

        def diff_eq(blocks, rhs_slot=None):                      # rows 0..N-2 of Pre @ (sum of blocks)
            nonlocal row
            for var, mat in blocks:
                A[row:row + N - 1, var] += (Pre @ mat)[:N - 1]
            if rhs_slot is not None:
                B[row:row + N - 1, rhs_slot * N:(rhs_slot + 1) * N] = Pre[:N - 1]
            r0 = row
            row += N - 1
            return r0

        r3 = None
        diff_eq([(U, (a0 + k * k / self.Re) * I + 1j * k * M1), (UZ, -D / self.Re), (P, 1j * k * I), (Vv, M2)], 0)
        diff_eq([(Vv, (a0 + k * k / self.Re) * I + 1j * k * M1), (VZ, -D / self.Re), (P, D), (R, self.Ri * I)], 1)
        r3 = diff_eq([(R, (a0 + k * k / self.Pe) * I + 1j * k * M1), (RZ, -D / self.Pe)], 2)
        if n == 0:
            A[r3:r3 + N - 1, 7 * N] += Pre[:N - 1, 0]            # + Fb (a constant = its T0 coefficient)
        A[row:row + N, U] += 1j * k * I; A[row:row + N, VZ] += I; row += N          # dx(u) + vz = 0 (algebraic: all N rows)
        diff_eq([(UZ, I), (U, -D)])
        diff_eq([(VZ, I), (Vv, -D)])
        diff_eq([(RZ, I), (R, -D)])
        left, right, integ = (-1.) ** np.arange(N), np.ones(N), cheb_integ(N)
        for var, fun in ((U, left), (Vv, left), (U, right), ((Vv, right) if n != 0 else (P, integ)), (RZ, left), (RZ, right)):
            A[row, var] = fun; row += 1
        if n == 0:
            A[row, R] = integ; row += 1                          # integ(rho,'z') = 0 takes the place of Fb = 0
        assert row == nv
        X = np.linalg.solve(A, B)
        self._S[n] = X[:6 * N]
        return self._S[n]

    def mixnorm_map(self, n):
        """S^MN_k (2Nz x Nz): rho -> (psi, psiz) with  dx dx psi + dz psiz + F = rho,  psiz = dz psi,  psiz(+-1) = 0 (:1053-1066)."""
        if n in self._SMN:
            return self._SMN[n]
        N, k = self.Nz, 2. * np.pi * n / self.Lx
        Pre, D, I = cheb_pre(N), self.Dz, np.eye(N)
        nv = 2 * N + (1 if n == 0 else 0)
        A = np.zeros((nv, nv), dtype=complex); B = np.zeros((nv, N), dtype=complex)
        PS, PZ = slice(0, N), slice(N, 2 * N)
        A[0:N - 1, PS] = (Pre @ (-k * k * I))[:N - 1]; A[0:N - 1, PZ] = (Pre @ D)[:N - 1]; B[0:N - 1] = Pre[:N - 1]
        if n == 0:
            A[0:N - 1, 2 * N] = Pre[:N - 1, 0]
        A[N - 1:2 * N - 2, PZ] = Pre[:N - 1]; A[N - 1:2 * N - 2, PS] = -(Pre @ D)[:N - 1]
        A[2 * N - 2, PZ] = (-1.) ** np.arange(N); A[2 * N - 1, PZ] = 1.
        if n == 0:
            A[2 * N, PS] = cheb_integ(N)
        self._SMN[n] = np.linalg.solve(A, B)[:2 * N]
        return self._SMN[n]

    def _apply(self, maps, fields):
        """Per-mode operator apply (every pencil, like the reference's loops over solver.pencils): fields (nin, Nxc, Nz) ->
        (nout, Nxc, Nz); mode -n uses the conjugate matrix.  Modes whose input is identically zero are skipped (their output is zero)."""
        N = self.Nz
        nin = len(fields)
        out = None
        for i, n in enumerate(self.n):
            if out is not None and not any(np.any(f[i]) for f in fields):
                continue
            S = maps(abs(int(n)))
            if n < 0:
                S = np.conj(S)
            y = S @ np.concatenate([f[i] for f in fields])
            if out is None:
                out = np.zeros((len(y) // N, self.Nxc, N), dtype=complex)
            out[:, i] = y.reshape(-1, N)
        assert nin * N == S.shape[1]
        return out

    def _apply_H(self, maps, fields, nout):
        N = self.Nz
        out = np.zeros((nout, self.Nxc, N), dtype=complex)
        for i, n in enumerate(self.n):
            if not any(np.any(f[i]) for f in fields):
                continue
            S = maps(abs(int(n)))
            if n < 0:
                S = np.conj(S)
            out[:, i] = (np.conj(S).T @ np.concatenate([f[i] for f in fields])).reshape(nout, N)
        return out

    # ---- callbacks ------------------------------------------------------------------------------------------------------------
    def split(self, X):
        a1, a2 = np.split(np.asarray(X, dtype=float), 2)
        return a1.reshape(self.Nx, self.Nz), a2.reshape(self.Nx, self.Nz)

    def inner(self, x, y):
        """Inner_Prod_Discrete (:282-299)."""
        A, B = self.split(x); u, v = self.split(y)
        return float((np.vdot(A, self.W * u) + np.vdot(B, self.W * v)) / self.V)

    def _nl(self, u, ux, uz, v, vx, vz, rx, rz):
        Ti = self.transformInverse
        ug, vg = Ti(u), Ti(v)
        NLu = -ug * Ti(ux) - vg * Ti(uz)
        NLv = -ug * Ti(vx) - vg * Ti(vz)
        NLr = -ug * Ti(rx) - vg * Ti(rz)
        T = self.transform
        return self.DA * T(NLu), self.DA * T(NLv), self.DA * T(NLr)

    def forward(self, X):
        """FWD_Solve_Discrete (:777-1155): cost  -1/2 dt sum_{n=0}^{N} <U_n,U_n>  (s=0)  or  1/2 <grad psi, grad psi>  (s=1)."""
        X = X[0] if isinstance(X, (list, tuple)) else X
        N, dt, ik = self.N_ITERS, self.dt, 1j * self.k[:, None]
        ug0, vg0 = self.split(X)
        rho = self.DA * self.transform(np.tile(-0.5 * erf(self.z / self.delta), (self.Nx, 1)))
        rz = self.DA * self.transform(np.tile(-np.exp(-(self.z / self.delta) ** 2) / (self.delta * np.sqrt(np.pi)), (self.Nx, 1)))
        u = self.DA * self.transform(ug0); v = self.DA * self.transform(vg0)
        uz = u @ self.Dz.T; vz = v @ self.Dz.T
        self.stack = np.zeros((3, self.Nxc, self.Nz, N + 1), dtype=complex)
        costKE = 0.
        for i in range(N):
            self.stack[0, :, :, i], self.stack[1, :, :, i], self.stack[2, :, :, i] = u, v, rho
            Uv = np.concatenate([np.real(self.transformInverse(u)).ravel(), np.real(self.transformInverse(v)).ravel()])
            costKE += dt * self.inner(Uv, Uv)
            NLu, NLv, NLr = self._nl(u, ik * u, uz, v, ik * v, vz, ik * rho, rz)
            u, v, rho, uz, vz, rz = self._apply(self.solve_map, [u / dt + NLu, v / dt + NLv, rho / dt + NLr])
        if self.s == 1:
            psi, psiz = self._apply(self.mixnorm_map, [rho])
            self.stack[0, :, :, N], self.stack[1, :, :, N], self.stack[2, :, :, N] = ik * psi, psiz, psi
            g = np.concatenate([np.real(self.transformInverse(ik * psi)).ravel(), np.real(self.transformInverse(psiz)).ravel()])
            return 0.5 * self.inner(g, g)
        Uv = np.concatenate([np.real(self.transformInverse(u)).ravel(), np.real(self.transformInverse(v)).ravel()])
        costKE += dt * self.inner(Uv, Uv)
        self.stack[0, :, :, N], self.stack[1, :, :, N], self.stack[2, :, :, N] = u, v, rho
        return -0.5 * costKE

    def adjoint(self, X=None):
        """ADJ_Solve_Discrete (:1320-1659): gradient with respect to Inner_Prod_Discrete, flat like X."""
        N, dt, ik, W, V = self.N_ITERS, self.dt, 1j * self.k[:, None], self.W, self.V
        Ti, TiA, TA, DzT = self.transformInverse, self.transformInverseAdjoint, self.transformAdjoint, self.Dz
        zero = np.zeros((self.Nxc, self.Nz), dtype=complex)
        ua, va, ra = zero.copy(), zero.copy(), zero.copy()
        idx = N
        if self.s == 1:
            vecx = Ti(self.stack[0, :, :, idx]) * (W / V); vecz = Ti(self.stack[1, :, :, idx]) * (W / V)
            mn1 = -ik * TiA(vecx) + TiA(vecz) @ DzT                      # adjointDerivativeX + derivativeZAdjoint (row vectors: c @ Dz)
            ra = self._apply_H(self.mixnorm_map, [mn1, zero], 1)[0]
        else:
            ua = -dt * TiA(W * Ti(self.stack[0, :, :, idx])) / V
            va = -dt * TiA(W * Ti(self.stack[1, :, :, idx])) / V
        idx -= 1
        uza, vza, rza = zero.copy(), zero.copy(), zero.copy()
        for _ in range(N):
            ua, va, ra = self._apply_H(self.solve_map, [ua, va, ra, uza, vza, rza], 3)
            uD, vD, rD = self.stack[0, :, :, idx], self.stack[1, :, :, idx], self.stack[2, :, :, idx]
            idx -= 1
            ux, uzg = Ti(ik * uD), Ti(uD @ self.Dz.T)
            vx, vzg = Ti(ik * vD), Ti(vD @ self.Dz.T)
            rx, rzg = Ti(ik * rD), Ti(rD @ self.Dz.T)
            ug, vg = Ti(uD), Ti(vD)
            v1, v2, v3 = TA(self.DA * ua), TA(self.DA * va), TA(self.DA * ra)
            adju = TiA(-ux * v1 - vx * v2 - rx * v3); adjux = TiA(-ug * v1); adjuz = TiA(-vg * v1)
            adjv = TiA(-uzg * v1 - vzg * v2 - rzg * v3); adjvx = TiA(-ug * v2); adjvz = TiA(-vg * v2)
            adjrx = TiA(-ug * v3); adjrz = TiA(-vg * v3)
            ua = ua / dt + adju - ik * adjux
            va = va / dt + adjv - ik * adjvx
            ra = ra / dt + 0. - ik * adjrx
            uza, vza, rza = adjuz, adjvz, adjrz
            if self.s == 0:
                ua = ua - dt * TiA(W * ug) / V
                va = va - dt * TiA(W * vg) / V
        ua = ua + uza @ DzT
        va = va + vza @ DzT
        gu = (V / W) * TA(ua); gv = (V / W) * TA(va)
        return [np.concatenate([np.real(gu).ravel(), np.real(gv).ravel()])]


def synthetic_ic(oracle, seed, E0=0.02, prep_steps=5):
    """Seeded noise streamfunction, u = -psi_z, w = psi_x (Generate_IC :355-371), de-aliased, smoothed by `prep_steps` steps of the
    forward solver (so that it satisfies the no-slip walls; the reference uses a Dedalus IVP for this, :374), scaled to <U,U> = E0."""
    o = oracle
    psi = o.transform(np.random.RandomState(seed).standard_normal((o.Nx, o.Nz)))
    keep = (np.abs(o.n) <= o.a // 2)[:, None] & (np.arange(o.Nz) < o.Nz0 // 2)[None, :]
    psi = psi * keep
    ik = 1j * o.k[:, None]
    u, v = -(psi @ o.Dz.T), ik * psi
    sc = 1e-2 / np.abs(u).max()
    u, v = sc * u, sc * v
    uz, vz = u @ o.Dz.T, v @ o.Dz.T
    rho = np.zeros_like(u); rz = np.zeros_like(u)
    for _ in range(prep_steps):
        NLu, NLv, NLr = o._nl(u, ik * u, uz, v, ik * v, vz, ik * rho, rz)
        u, v, rho, uz, vz, rz = o._apply(o.solve_map, [u / o.dt + NLu, v / o.dt + NLv, rho / o.dt + NLr])
    X = np.concatenate([np.real(o.transformInverse(o.DA * u)).ravel(), np.real(o.transformInverse(o.DA * v)).ravel()])
    return X * np.sqrt(E0 / o.inner(X, X))


# -------------------------------------------------------------------------------------------------------------------------------------
# "Continuous" formulation (the reference script's default Adjoint_type, FWD_Solve_Poiseuille.py:1728): Dedalus IVPs with SBDF1,
# real Fourier x Chebyshev domain with dealias 3/2.  Restates FWD_Solve_Cnts (:614-775), ADJ_Solve_Cnts (:1161-1318), Inner_Prod_Cnts
# (:264-280), Integrate_Field (:241-262) and Norm_and_Inverse_Second_Derivative (:1661-1696) — PARITY UNPINNED (Dedalus internals):
#   * state: (a = Nx/2) x Nz complex T-coefficients (real Fourier basis, Nyquist dropped); products are formed on the
#     (3Nx/2) x (3Nz/2) grid and truncated back (A.0-7); flat vectors live on that grid (Field_to_Vec with scales = dealias);
#   * one SBDF1 step (A.0-5): (M/dt + L) X1 = M X0/dt + F(X0): the same tau solve map as the Discrete formulation, built for Nz modes;
#     F uses the STATE variables uz, wz, bz (zero before the first step for uz, wz: the script only initialises u, w, b, bz);
#   * integ(f) integrates the truncated series exactly: Lx * sum_{k even} f_{0,k} * 2/(1-k^2) -> quadrature weights on the grid;
#   * the adjoint is the script's own adjoint PDE (an O(dt)-consistent approximation of the discrete gradient), N_ITERS steps,
#     forward snapshots N, N-1, ..., 1 as coefficients of its right-hand side.
# -------------------------------------------------------------------------------------------------------------------------------------
class PoiseuilleCntsOracle:
    def __init__(self, Nx=16, Nz=16, Re=500., Ri=0.05, dt=5e-3, N_ITERS=10, s=0, Prandtl=1., delta=0.125, Lx=4. * np.pi):
        self.Nx, self.Nz, self.Re, self.Ri, self.Pe = int(Nx), int(Nz), float(Re), float(Ri), float(Re) * Prandtl
        self.dt, self.N_ITERS, self.s, self.delta, self.Lx = float(dt), int(N_ITERS), int(s), float(delta), float(Lx)
        self.a = (self.Nx - 1) // 2 + 1
        self.Gx, self.Gz = 3 * self.Nx // 2, 3 * self.Nz // 2
        self.k = 2. * np.pi * np.arange(self.a) / self.Lx
        self.z = -np.cos(np.pi * (np.arange(self.Gz) + 0.5) / self.Gz)
        self.V = self.Lx * 2.
        self.Dz = cheb_diff(self.Nz)
        j, i = np.arange(self.Nz)[:, None], np.arange(self.Gz)[None, :]
        c = np.cos(np.pi * j * (2 * i + 1) / (2. * self.Gz)) * (-1.) ** j
        self.Tf = (2. / self.Gz) * c * np.where(j == 0, 0.5, 1.)            # (Nz, Gz): grid line -> first Nz T coefficients
        self.Ti = c.T.copy()                                                 # (Gz, Nz): T coefficients -> grid line
        self.Wq = (cheb_integ(self.Nz) @ self.Tf) * (self.Lx / self.Gx)      # exact-integration weights of a truncated grid product
        self._S, self._SA, self._SMN = {}, {}, {}
        self.stack = None

    def to_coeff(self, g):
        F = np.fft.rfft(g, axis=0)[:self.a] / self.Gx
        return F @ self.Tf.T

    def to_grid(self, c):
        F = np.zeros((self.Gx // 2 + 1, self.Gz), dtype=complex)
        F[:self.a] = c @ self.Ti.T
        F[0] = F[0].real
        return np.fft.irfft(F, n=self.Gx, axis=0) * self.Gx

    def integ_mean(self, g):
        """(1/V) integ of a grid product, Dedalus style (truncate to the modes, integrate the series exactly)."""
        return float(np.sum(g * self.Wq[None, :]) / self.V)

    def split(self, X):
        a1, a2 = np.split(np.asarray(X, dtype=float), 2)
        return a1.reshape(self.Gx, self.Gz), a2.reshape(self.Gx, self.Gz)

    def inner(self, x, y):
        """Inner_Prod_Cnts (:264-280)."""
        A, B = self.split(x); u, v = self.split(y)
        return self.integ_mean(A * u + B * v)

    def _maps(self, cache, n, adjoint):
        if n not in cache:
            h = PoiseuilleOracle.__new__(PoiseuilleOracle)
            h.Nz, h.Lx, h.dt, h.Re, h.Pe, h.Ri, h.Dz, h._S = self.Nz, self.Lx, self.dt, self.Re, self.Pe, self.Ri, self.Dz, {}
            cache[n] = solve_map_general(h, n, adjoint)
        return cache[n]

    def _apply(self, adjoint, fields):
        cache = self._SA if adjoint else self._S
        out = np.zeros((6, self.a, self.Nz), dtype=complex)
        for n in range(self.a):
            out[:, n] = (self._maps(cache, n, adjoint) @ np.concatenate([f[n] for f in fields])).reshape(6, self.Nz)
        return out

    def _mixnorm(self, rho):
        h = PoiseuilleOracle.__new__(PoiseuilleOracle)
        h.Nz, h.Lx, h.Dz, h._SMN = self.Nz, self.Lx, self.Dz, self._SMN
        out = np.zeros((2, self.a, self.Nz), dtype=complex)
        for n in range(self.a):
            out[:, n] = (h.mixnorm_map(n) @ rho[n]).reshape(2, self.Nz)
        return out

    def forward(self, X):
        X = X[0] if isinstance(X, (list, tuple)) else X
        N, dt, ik, G, C = self.N_ITERS, self.dt, 1j * self.k[:, None], self.to_grid, self.to_coeff
        ug0, wg0 = self.split(X)
        u, w = C(ug0), C(wg0)
        b = C(np.tile(-0.5 * erf(self.z / self.delta), (self.Gx, 1)))
        bz = C(np.tile(-np.exp(-(self.z / self.delta) ** 2) / (self.delta * np.sqrt(np.pi)), (self.Gx, 1)))
        uz, wz = np.zeros_like(u), np.zeros_like(u)
        self.stack = np.zeros((3, self.a, self.Nz, N + 1), dtype=complex)
        J = 0.
        for n in range(N + 1):
            self.stack[0, :, :, n], self.stack[1, :, :, n], self.stack[2, :, :, n] = u, w, b
            ug, wg = G(u), G(w)
            J += dt * self.integ_mean(ug * ug + wg * wg)
            Fb = C(-(ug * G(ik * b) + wg * G(bz)))
            Fu = C(-(ug * G(ik * u) + wg * G(uz)))
            Fw = C(-(ug * G(ik * w) + wg * G(wz)))
            u, w, b, uz, wz, bz = self._apply(False, [u / dt + Fu, w / dt + Fw, b / dt + Fb])
        if self.s == 1:
            psi, psiz = self._mixnorm(self.stack[2, :, :, N])
            fx, fz = G(ik * psi), G(psiz)
            return 0.5 * self.integ_mean(fx * fx + fz * fz)
        return -0.5 * J

    def adjoint(self, X=None):
        N, dt, ik, G, C, Dz = self.N_ITERS, self.dt, 1j * self.k[:, None], self.to_grid, self.to_coeff, self.Dz
        zero = np.zeros((self.a, self.Nz), dtype=complex)
        ua, wa, ba, uza, wza, bza = (zero.copy() for _ in range(6))
        if self.s == 1:
            ba = -self._mixnorm(self.stack[2, :, :, N])[0]
        idx = N
        for _ in range(N):
            uf, wf, bf = self.stack[0, :, :, idx], self.stack[1, :, :, idx], self.stack[2, :, :, idx]
            idx -= 1
            ufg, wfg = G(uf), G(wf)
            uag, wag, bag = G(ua), G(wa), G(ba)
            Fb = C(ufg * G(ik * ba) + wfg * G(bza))
            Fu = C(-(uag * G(ik * uf) + wag * G(ik * wf)) + (ufg * G(ik * ua) + wfg * G(uza)) - bag * G(ik * bf) - (ufg if self.s == 0 else 0.))
            Fw = C(-(uag * G(uf @ Dz.T) + wag * G(wf @ Dz.T)) + (ufg * G(ik * wa) + wfg * G(wza)) - bag * G(bf @ Dz.T) - (wfg if self.s == 0 else 0.))
            ua, wa, ba, uza, wza, bza = self._apply(True, [ua / dt + Fu, wa / dt + Fw, ba / dt + Fb])
        return [np.concatenate([G(ua).ravel(), G(wa).ravel()])]


def solve_map_general(h, n, adjoint):
    """Tau solve map of the forward (:465-500) or adjoint (:1217-1252) IVP for wavenumber n: (rhs_u, rhs_w, rhs_b) -> (u, w, b, uz, wz, bz).
    The adjoint operator advects with -U, couples Ri*w into the b equation and Uz*u into the w equation."""
    N, k, a0 = h.Nz, 2. * np.pi * n / h.Lx, 1. / h.dt
    Pre, D, I = cheb_pre(N), h.Dz, np.eye(N)
    M1 = cheb_mult(N, [0.5, 0., -0.5]); M2 = cheb_mult(N, [0., -2.])
    sg = -1. if adjoint else 1.
    U, Wv, R, UZ, WZ, RZ, P = (slice(i * N, (i + 1) * N) for i in range(7))
    nv = 7 * N + (1 if n == 0 else 0)
    A = np.zeros((nv, nv), dtype=complex); B = np.zeros((nv, 3 * N), dtype=complex)
    row = 0

    def diff_eq(blocks, rhs_slot=None):
        nonlocal row
        for var, mat in blocks:
            A[row:row + N - 1, var] += (Pre @ mat)[:N - 1]
        if rhs_slot is not None:
            B[row:row + N - 1, rhs_slot * N:(rhs_slot + 1) * N] = Pre[:N - 1]
        r0 = row; row += N - 1
        return r0

    adv = sg * 1j * k * M1
    eq_u = [(U, (a0 + k * k / h.Re) * I + adv), (UZ, -D / h.Re), (P, -1j * k * I)]
    eq_w = [(Wv, (a0 + k * k / h.Re) * I + adv), (WZ, -D / h.Re), (P, -D)]
    eq_b = [(R, (a0 + k * k / h.Pe) * I + adv), (RZ, -D / h.Pe)]
    if adjoint:
        eq_w.append((U, M2)); eq_b.append((Wv, h.Ri * I))
    else:
        eq_u.append((Wv, M2)); eq_w.append((R, h.Ri * I))
    diff_eq(eq_u, 0); diff_eq(eq_w, 1)
    r3 = diff_eq(eq_b, 2)
    if n == 0:
        A[r3:r3 + N - 1, 7 * N] += Pre[:N - 1, 0]
    A[row:row + N, U] += 1j * k * I; A[row:row + N, WZ] += I; row += N
    diff_eq([(UZ, I), (U, -D)]); diff_eq([(WZ, I), (Wv, -D)]); diff_eq([(RZ, I), (R, -D)])
    left, right, integ = (-1.) ** np.arange(N), np.ones(N), cheb_integ(N)
    for var, fun in ((RZ, left), (RZ, right), (U, left), (Wv, left), (U, right), ((Wv, right) if n != 0 else (P, integ))):
        A[row, var] = fun; row += 1
    if n == 0:
        A[row, R] = integ; row += 1
    assert row == nv
    return np.linalg.solve(A, B)[:6 * N]


def synthetic_ic_cnts(oracle, seed, E0=0.02, prep_steps=5):
    """Continuous-formulation counterpart of synthetic_ic: band-limited stream-function noise, a few forward steps, scaled to <U,U> = E0."""
    o = oracle
    psi = o.to_coeff(np.random.RandomState(seed).standard_normal((o.Gx, o.Gz)))
    psi = psi * ((np.arange(o.a) <= o.a // 2)[:, None] & (np.arange(o.Nz) < o.Nz // 2)[None, :])
    ik = 1j * o.k[:, None]
    u, w = -(psi @ o.Dz.T), ik * psi
    sc = 1e-2 / np.abs(o.to_grid(u)).max()
    u, w = sc * u, sc * w
    zero = np.zeros_like(u)
    for _ in range(prep_steps):
        u, w, b, uz, wz, bz = o._apply(False, [u / o.dt, w / o.dt, zero])
    X = np.concatenate([o.to_grid(u).ravel(), o.to_grid(w).ravel()])
    return X * np.sqrt(E0 / o.inner(X, X))

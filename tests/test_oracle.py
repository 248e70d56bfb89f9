"""The oracle (CPU restatement) against (i) vectors produced by the reference's own helper functions,
(ii) its own committed outputs, (iii) the Taylor remainder test (self-consistency of J and grad J)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import shb23
from oracle.kdyn import KDynOracle, synthetic_field
from oracle.sh23 import SH23Oracle, synthetic_ic
from spheremanopt_amd.test_grad import taylor_table


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("N", [8, 512])
def test_shb_helpers_bit_exact_vs_reference(N):
    g = _load("shb_helpers.npz")
    v = g["v%d" % N]
    assert np.array_equal(shb23.transform(v), g["T%d" % N])
    assert np.array_equal(shb23.transformInverse(v), g["Tinv%d" % N])
    assert np.array_equal(shb23.transformAdjoint(v), g["Tadj%d" % N])
    assert np.array_equal(shb23.transformInverseAdjoint(v), g["Tinvadj%d" % N])
    z = shb23.gauss_grid(N)
    W = shb23.weights(z)
    assert np.array_equal(W, g["W%d" % N])
    assert np.dot(v, W * g["T%d" % N]) / 40.0 == g["ip%d" % N]


def test_shb_transform_identities():
    N = 64
    I = np.eye(N)
    T = np.stack([shb23.transform(I[i]) for i in range(N)], axis=1)
    Ti = np.stack([shb23.transformInverse(I[i]) for i in range(N)], axis=1)
    Ta = np.stack([shb23.transformAdjoint(I[i]) for i in range(N)], axis=1)
    Tia = np.stack([shb23.transformInverseAdjoint(I[i]) for i in range(N)], axis=1)
    assert np.allclose(T @ Ti, I, atol=1e-13)
    assert np.allclose(Ta, T.T, atol=1e-14) and np.allclose(Tia, Ti.T, atol=1e-13)
    x = shb23.gauss_grid(N, (-1., 1.))
    for n in (0, 1, 5, N - 1):       # transform(T_n(x)) = e_n on the ascending Gauss grid
        assert np.allclose(shb23.transform(np.cos(n * np.arccos(x))), I[n], atol=1e-13)


def test_shb_tau_operator_solves_the_bvp():
    """S r must be the spectrally accurate solution of (1/dt + 1 - a + 2 d2 + d4) u = r with
    u'(-20) = u'''(-20) = 0, u(20) = u''(20) = 0 for a smooth, resolved right-hand side."""
    from numpy.polynomial import chebyshev as C
    dt, a = 1e-2, -0.1
    rhs = lambda z: np.exp(-(z / 4) ** 2) * np.cos(z)
    zz = np.linspace(-15, 15, 7)
    sols = []
    for N in (64, 96):
        S = shb23.tau_operator(N, dt, a)
        u = S @ shb23.transform(rhs(shb23.gauss_grid(N)))
        p = C.Chebyshev(u, domain=[-20, 20])
        tol = 1e-3 if N == 64 else 1e-9      # u-derived derivatives carry the tau error of the first-order system
        assert abs(p(20.)) < 1e-14 and abs(p.deriv(2)(20.)) < tol
        assert abs(p.deriv(1)(-20.)) < tol and abs(p.deriv(3)(-20.)) < tol
        res = (1 / dt + 1 - a) * p(zz) + 2 * p.deriv(2)(zz) + p.deriv(4)(zz) - rhs(zz)
        assert np.abs(res).max() < (1e-7 if N == 64 else 1e-12)
        sols.append(p(zz))
    assert np.abs(sols[0] - sols[1]).max() < 1e-9


def test_oracle_regression_sh23_c2():
    g = _load("oracle_sh23_c2.npz")
    o = SH23Oracle(256, dt=0.1, N_ITERS=500)
    X = synthetic_ic(512, 42, 0.0725)
    assert abs(o.inner(X, X) - 0.0725) < 1e-15
    J = o.forward([X])
    assert abs(J - g["J"]) <= 1e-13 * abs(g["J"])
    grad = o.adjoint([X])[0]
    assert np.allclose(grad, g["grad"], rtol=1e-11, atol=1e-13)
    assert np.allclose(o.adjoint([X], "Continuous")[0], g["grad_cont"], rtol=1e-11, atol=1e-13)


def test_oracle_regression_kdyn_small():
    for cost in ("Final", "Integrated"):
        g = _load("oracle_kdyn_n32_%s.npz" % cost.lower())
        N, steps = int(g["N"]), int(g["steps"])
        k = KDynOracle(N, Rm=1., dt=1e-3, N_ITERS=steps, Cost_function=cost)
        B = synthetic_field(k.G, 1); U = synthetic_field(k.G, 2)
        assert abs(k.inner(B, B) - 1) < 1e-13
        J = k.forward([B, U])
        gB, gU = k.adjoint([B, U])
        assert abs(J - g["J"]) <= 1e-12 * abs(g["J"])
        assert np.allclose(gB[g["idx"]], g["gB"], rtol=1e-9, atol=1e-12 * g["gB_norm"])
        assert np.allclose(gU[g["idx"]], g["gU"], rtol=1e-9, atol=1e-12 * g["gU_norm"])


def _slopes_ok(AA, tol=2e-3):
    return np.all(np.abs(AA[4, :4] - 2.) < tol) and np.all(np.abs(AA[3, :4] - 1.) < 0.05)


def test_taylor_sh23():
    o = SH23Oracle(64, dt=0.1, N_ITERS=60)
    X = synthetic_ic(128, 42, 0.0725); dX = synthetic_ic(128, 7, 0.0725)
    AA = taylor_table([X], [dX], o.forward, o.adjoint, o.inner, epsilon=1e-3)
    assert _slopes_ok(AA), AA


@pytest.mark.parametrize("N", [12, 10, 22])      # G = 18; 15 and 33: odd grids (Npts = 2 mod 4), which only the device's run-time-length kernels take
@pytest.mark.parametrize("cost", ["Final", "Integrated"])
def test_taylor_kdyn(cost, N):
    k = KDynOracle(N, Rm=1., dt=1e-2, N_ITERS=12 if N == 12 else 6, Cost_function=cost)
    B = synthetic_field(k.G, 1) + 0.1 * np.random.RandomState(9).standard_normal(3 * k.G ** 3)   # not div-free, mean != 0
    U, dB, dU = (synthetic_field(k.G, s) for s in (2, 3, 4))
    AA = taylor_table([B, U], [dB, dU], k.forward, k.adjoint, k.inner, epsilon=1e-3)
    assert _slopes_ok(AA), AA


def test_taylor_shb23():
    o = shb23.SHB23Oracle(128, dt=1e-2, N_ITERS=100)   # (at N=64 the reference's missing Z^T in NLtermAdj shows at the 1e-5 level)
    X = shb23.synthetic_ic(o, 42, 0.0019); dX = shb23.synthetic_ic(o, 7, 0.0019)
    AA = taylor_table([X], [dX], o.forward, o.adjoint, o.inner, epsilon=1e-3)
    assert _slopes_ok(AA), AA


def test_shb23_continuous_adjoint_is_first_order_consistent():
    """The "Continuous" SHB23 gradient (q(T) of the adjoint PDE, SHB:685-794) is not the exact gradient of the discrete cost:
    its directional derivative approaches the finite-difference one as dt -> 0 (T = 1 fixed)."""
    err = []
    for n, dt in ((100, 1e-2), (400, 2.5e-3)):
        o = shb23.SHB23CntsOracle(64, dt=dt, N_ITERS=n)
        X = shb23.synthetic_ic_cnts(o, 42, 0.0019); dX = shb23.synthetic_ic_cnts(o, 7, 0.0019)
        assert abs(o.inner(X, X) - 0.0019) < 1e-15
        o.forward([X])
        d = o.inner(o.adjoint([X])[0], dX)
        fd = (o.forward([X + 1e-5 * dX]) - o.forward([X - 1e-5 * dX])) / 2e-5
        err.append(abs(d - fd) / abs(fd))
    assert err[0] < 1e-2 and err[1] < 0.6 * err[0], err
    # every state satisfies the boundary condition imposed on u itself, u(Lz/2) = 0, i.e. sum_k c_k = 0
    assert abs(o.stack[:, -1].sum()) < 1e-12


def test_poiseuille_transform_adjoints_and_taylor():
    """Plane-Poiseuille oracle: the four transforms are adjoint pairs; the time-averaged-energy gradient passes the Taylor test; the
    mix-norm gradient (whose finite differences sit at round-off for a short run) agrees with a central difference."""
    from oracle.poiseuille import PoiseuilleOracle, synthetic_ic
    o = PoiseuilleOracle(24, 36, dt=5e-3, N_ITERS=5, s=0, delta=0.3)
    rs = np.random.RandomState(0)
    g = rs.standard_normal((o.Nx, o.Nz)) + 1j * rs.standard_normal((o.Nx, o.Nz))
    c = rs.standard_normal((o.Nxc, o.Nz)) + 1j * rs.standard_normal((o.Nxc, o.Nz))
    assert abs(np.vdot(c, o.transform(g)) - np.vdot(o.transformAdjoint(c), g)) < 1e-12
    assert abs(np.vdot(g, o.transformInverse(c)) - np.vdot(o.transformInverseAdjoint(g), c)) < 1e-10
    assert np.abs(o.transform(o.transformInverse(c)) - c).max() < 1e-12
    X = synthetic_ic(o, 42); dX = synthetic_ic(o, 7)
    assert abs(o.inner(X, X) - 0.02) < 1e-15
    AA = taylor_table([X], [dX], o.forward, o.adjoint, o.inner, epsilon=1e-3)
    assert np.all(np.abs(AA[4, :4] - 2.0) < 1e-2), AA
    # every state satisfies the no-slip walls: sum_n u_n = sum_n (-1)^n u_n = 0 for every x wavenumber
    u = o.stack[0, :, :, -1]
    assert np.abs(u.sum(axis=1)).max() < 1e-12 and np.abs((u * (-1.) ** np.arange(o.Nz)).sum(axis=1)).max() < 1e-12
    o1 = PoiseuilleOracle(24, 36, dt=5e-3, N_ITERS=5, s=1, delta=0.3)
    o1.forward([X])
    d = o1.inner(o1.adjoint([X])[0], dX)
    fd = (o1.forward([X + 1e-2 * dX]) - o1.forward([X - 1e-2 * dX])) / 2e-2
    assert abs(d - fd) < 2e-3 * abs(fd), (d, fd)


def test_poiseuille_continuous_adjoint_is_first_order_consistent():
    """The script's own adjoint PDE (ADJ_Solve_Cnts) approximates the gradient of the discrete cost to O(dt): the directional derivative
    approaches the central difference as dt -> 0 (T fixed); the tau solve map equals the Discrete formulation's for the forward operator."""
    from oracle.poiseuille import PoiseuilleCntsOracle, PoiseuilleOracle, solve_map_general, synthetic_ic_cnts
    h = PoiseuilleOracle(24, 16, dt=5e-3)
    assert np.abs(h.solve_map(3) - solve_map_general(h, 3, False)).max() < 1e-12
    err = []
    for n, dt in ((20, 1e-2), (80, 2.5e-3)):
        o = PoiseuilleCntsOracle(16, 24, dt=dt, N_ITERS=n, s=0, delta=0.3)
        X = synthetic_ic_cnts(o, 42); dX = synthetic_ic_cnts(o, 7)
        assert abs(o.inner(X, X) - 0.02) < 1e-15
        o.forward([X])
        d = o.inner(o.adjoint([X])[0], dX)
        fd = (o.forward([X + 1e-4 * dX]) - o.forward([X - 1e-4 * dX])) / 2e-4
        err.append(abs(d - fd) / abs(fd))
    assert err[0] < 0.1 and err[1] < 0.5 * err[0], err
    # mix-norm cost: the gradient is second order in the amplitude, so test it where the flow actually stirs the stratification
    o = PoiseuilleCntsOracle(16, 32, dt=5e-3, N_ITERS=40, s=1, delta=0.5)
    X = 10. * synthetic_ic_cnts(o, 42); dX = synthetic_ic_cnts(o, 7)
    o.forward([X])
    d = o.inner(o.adjoint([X])[0], dX)
    fd = (o.forward([X + 1e-2 * dX]) - o.forward([X - 1e-2 * dX])) / 2e-2
    assert abs(d - fd) < 0.05 * abs(fd), (d, fd)


def test_oracle_regression_poiseuille():
    from oracle.poiseuille import PoiseuilleOracle
    for sw in (0, 1):
        g = _load("oracle_poiseuille_96x48_s%d.npz" % sw)
        o = PoiseuilleOracle(96, 48, dt=5e-3, N_ITERS=40, s=sw, delta=0.3)
        J = o.forward([g["X"]])
        grad = o.adjoint([g["X"]])[0]
        assert abs(J - g["J"]) <= 1e-12 * abs(g["J"])
        assert np.linalg.norm(grad - g["grad"]) <= 1e-9 * np.linalg.norm(g["grad"])


def test_known_answers_of_the_time_steppers():
    """Closed-form checks that do not involve any restated Dedalus internals beyond the published schemes:
    KDyn with U = 0: a single solenoidal Fourier mode is multiplied per CNAB1 step by (1/dt - k^2/2Rm) / (1/dt + k^2/2Rm);
    SH23 at infinitesimal amplitude: mode k is multiplied per SBDF1 step by (1/dt) / (1/dt + (1-k^2)^2 - a)."""
    N, n, dt, Rm = 16, 7, 1e-2, 1.3
    k = KDynOracle(N, Rm=Rm, dt=dt, N_ITERS=n)
    G = k.G
    x = 2. * np.pi * np.arange(G) / G
    B = np.zeros((3, G, G, G)); B[1] = np.cos(3. * x)[:, None, None]                  # B = (0, cos 3x, 0): k.B = 0, |k|^2 = 9
    U = np.zeros(3 * G ** 3)
    r = (1. / dt - 9. / (2. * Rm)) / (1. / dt + 9. / (2. * Rm))
    J = k.forward([B.reshape(-1), U])
    assert abs(J + 0.5 * r ** (2 * n)) < 1e-13                                        # <B0,B0> = 1/2
    # uniform flow U = (c,0,0): curl(U x B) = -c dB/dx is treated explicitly => factor [(1/dt - D/2) - 3ic] / (1/dt + D/2), D = 9/Rm
    c = 0.7
    Uc = np.zeros((3, G, G, G)); Uc[0] = c
    J = k.forward([B.reshape(-1), Uc.reshape(-1)])
    r2 = ((1. / dt - 9. / (2. * Rm)) ** 2 + 9. * c * c) / (1. / dt + 9. / (2. * Rm)) ** 2
    assert abs(J + 0.5 * r2 ** n) < 1e-13
    o = SH23Oracle(64, dt=0.1, N_ITERS=20)
    xs = o.L * np.arange(o.G) / o.G
    eps, m = 1e-9, 5                                                                  # mode 5 of the 12 pi box: k = 5/6
    X = eps * np.cos(2. * np.pi * m * xs / o.L)
    km = 2. * np.pi * m / o.L
    rs = (1. / 0.1) / (1. / 0.1 + (1. - km ** 2) ** 2 + 0.3)
    Jo = o.forward([X])
    expect = -0.1 * 0.5 * eps ** 2 * sum(rs ** (2 * i) for i in range(21))
    assert abs(Jo - expect) < 1e-7 * abs(expect)


def test_kdyn_invariants():
    k = KDynOracle(12, Rm=1., dt=1e-2, N_ITERS=5)
    B = synthetic_field(k.G, 1); U = synthetic_field(k.G, 2)
    k.forward([B, U])
    div = np.abs(k.kdot(k.stack[..., -1])).max()
    assert div < 1e-13                                    # div-free IC stays div-free
    gB, gU = k.adjoint([B, U])
    gUh = k.vec_to_coeff(gU)
    assert np.abs(k.kdot(gUh)).max() < 1e-12              # gradient w.r.t. U is solenoidal
    # transform round trip and layout: coeff -> grid -> coeff
    c = k.vec_to_coeff(B)
    assert np.allclose(k.vec_to_coeff(k.coeff_to_vec(c)), c, atol=1e-14)


@pytest.mark.parametrize("cost,adj", [("Final", "Discrete"), ("Integrated", "Continuous")])
def test_threaded_kdyn_oracle_is_the_same_restatement(cost, adj):
    """bench.py's all-core CPU leg (oracle.kdyn.ThreadedKDynOracle: pointwise stages chunked over a thread pool) computes what KDynOracle
    computes: gradients bit for bit, J to the rounding of the per-chunk grid mean."""
    from oracle.kdyn import KDynOracle, ThreadedKDynOracle, synthetic_field
    N, n = 16, 4
    B, U = synthetic_field(24, 1), synthetic_field(24, 2)
    o = KDynOracle(N, Rm=1., dt=1e-3, N_ITERS=n, Cost_function=cost)
    J0 = o.forward([B, U]); g0 = o.adjoint([B, U], adj)
    for threads in (1, 3):
        t = ThreadedKDynOracle(N, Rm=1., dt=1e-3, N_ITERS=n, Cost_function=cost, threads=threads)
        J1 = t.forward([B, U]); g1 = t.adjoint([B, U], adj)
        assert abs(J1 - J0) <= 1e-14 * abs(J0)
        assert np.array_equal(g1[0], g0[0]) and np.array_equal(g1[1], g0[1])


def test_closed_form_time_steps_equal_dense_pencil_solves():
    """The closed forms of the oracles (oracle/kdyn.py cnab_update and the nu recursion, oracle/sh23.py's divide) against a dense solve per
    Fourier mode of the PENCIL MATRICES assembled row by row from the text of the reference's add_equation calls, stepped with Dedalus v2's
    published CNAB1 / SBDF1 coefficient tables (oracle/pencil.py): every mode of a small grid, random complex data, non-solenoidal states
    included — the algebraic divergence row under CNAB1 (k.B flips sign) and the k = 0 rows (X -> -X) are part of what is compared."""
    from oracle import pencil
    rs = np.random.RandomState(5)
    Rm, dt = 1.7, 3e-2
    o = KDynOracle(8, Rm=Rm, dt=dt, N_ITERS=1)
    shp = (3, o.a, o.m, o.m)
    V0 = rs.standard_normal(shp) + 1j * rs.standard_normal(shp)
    F = rs.standard_normal(shp) + 1j * rs.standard_normal(shp)
    V1 = o.cnab_update(V0, F)
    # the adjoint's second system (variables P, nu): the oracle's recursion, FWD_Solve_KDyn.py:876-879
    nu0 = rs.standard_normal(shp) + 1j * rs.standard_normal(shp)
    F2 = rs.standard_normal(shp) + 1j * rs.standard_normal(shp)                  # = -F(G; B_f), the right-hand side as the equation text has it
    nu1 = nu0 - 2. * o.K * (o.kdot(nu0) / o.k2s) + dt * o.project(F2)
    nu1[:, o.zero] = -nu0[:, o.zero]
    worst = 0.
    for ix in range(o.a):
        for iy in range(o.m):
            for iz in range(o.m):
                k = o.K[:, ix, iy, iz]
                M, L = pencil.kdyn_forward_pencil(k, Rm)
                X0 = np.concatenate([[0.], V0[:, ix, iy, iz]])                 # Pi_{n-1} does not enter: its column of M is zero and (M/dt - L/2) Pi
                # ... does enter through L: Dedalus carries Pi as a state variable.  CNAB1's right-hand side holds -b_1 L X_{n-1}, Pi included; the
                # closed form has no Pi history, i.e. it assumes the gradient part of the right-hand side is projected away — which is exact, because
                # whatever Pi_{n-1} is, it adds a pure gradient i k Pi / 2 to the momentum rows and the projection in the solve removes it.  Checked
                # by stepping with a random Pi_{n-1} as well.
                zero = not k.any()                                             # the k = 0 pencil has its own equations ("A = 0", ...: FWD_Solve_KDyn.py:431-434) with right-hand side 0
                Fk = np.concatenate([[0.], np.zeros(3) if zero else F[:, ix, iy, iz]])
                for pi_prev in (0., 0.37 - 1.1j):
                    X0[0] = pi_prev
                    X1 = pencil.imex_step(M, L, X0, Fk, "CNAB1", dt)
                    worst = max(worst, np.abs(X1[1:] - V1[:, ix, iy, iz]).max())
                Ma, La = pencil.kdyn_adjoint_pencil(k, Rm)
                Y0 = np.concatenate([[0.], V0[:, ix, iy, iz], [0.2 + 0.1j], nu0[:, ix, iy, iz]])
                Fa = np.zeros(8, dtype=complex) if zero else np.concatenate([[0.], F[:, ix, iy, iz], [0.], F2[:, ix, iy, iz]])
                Y1 = pencil.imex_step(Ma, La, Y0, Fa, "CNAB1", dt)
                worst = max(worst, np.abs(Y1[1:4] - V1[:, ix, iy, iz]).max(), np.abs(Y1[5:] - nu1[:, ix, iy, iz]).max())
    assert worst < 1e-11, worst
    # the terminal condition of the discrete adjoint (Compatib_Cond's LBVP, FWD_Solve_KDyn.py:733-747) against oracle/kdyn.py adjoint()'s closed form
    for cost in ("Final", "Integrated"):
        oc = KDynOracle(8, Rm=Rm, dt=dt, N_ITERS=1, Cost_function=cost)
        scale = (dt * oc.alpha) if cost == "Final" else oc.alpha
        Gh = oc.project(-2. * V0) / scale
        Gh[:, oc.zero] = 0.
        for ix in range(oc.a):
            for iy in range(oc.m):
                for iz in range(oc.m):
                    k = oc.K[:, ix, iy, iz]
                    Lc = pencil.kdyn_compat_pencil(k, Rm, dt, cost)
                    rhs = np.zeros(4, dtype=complex) if not k.any() else np.concatenate([[0.], -2. * V0[:, ix, iy, iz]])
                    X = np.linalg.solve(Lc, rhs)
                    assert np.abs(X[1:] - Gh[:, ix, iy, iz]).max() < 1e-11, (cost, ix, iy, iz)
    s = SH23Oracle(64, dt=0.1, N_ITERS=1)
    u0 = rs.standard_normal(s.Nc) + 1j * rs.standard_normal(s.Nc)
    f = rs.standard_normal(s.Nc) + 1j * rs.standard_normal(s.Nc)
    closed = (u0 / s.dt + f) / s.A                                              # oracle/sh23.py forward(): the SBDF1 step
    for i, k in enumerate(s.k):
        M, L = pencil.sh23_pencil(k, s.a)
        x1 = pencil.imex_step(M, L, np.array([u0[i]]), np.array([f[i]]), "SBDF1", s.dt)
        assert abs(x1[0] - closed[i]) < 1e-13 * max(1., abs(closed[i]))

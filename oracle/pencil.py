"""ORACLE (test infrastructure, never shipped) — the reference's PDE systems as Dedalus-v2 style PENCIL MATRICES, stepped with the published
multistep IMEX coefficient tables by a dense numerical solve per Fourier mode.

Why: the other oracle modules restate the reference's time steps in CLOSED FORM (oracle/kdyn.py `cnab_update`: a projection formula;
oracle/sh23.py: a diagonal divide).  Those closed forms are this repository's own algebra.  Here the same steps are taken with no algebra
at all: for one wave vector the matrices M (coefficients of dt(.)) and L (the other left-hand-side terms) are assembled row by row from the
TEXT of the reference's `add_equation` calls — each row cites its line — and one step of the scheme the reference names
(`de.timesteppers.CNAB1`, `SBDF1`) is the dense solve

    (a_0 M + b_0 L) X_n  =  sum_{j>=1} ( c_j F_{n-j} - a_j M X_{n-j} - b_j L X_{n-j} )

with Dedalus v2's published one-step tables (dedalus/core/timesteppers.py, MultistepIMEX):  SBDF1: a = (1/dt, -1/dt), b = (1, 0), c = (0, 1);
CNAB1: a = (1/dt, -1/dt), b = (1/2, 1/2), c = (0, 1).  tests/test_oracle.py checks the closed forms of oracle/kdyn.py and oracle/sh23.py against
these solves on every mode of a small grid with random (also non-solenoidal) data — which pins the closed forms to the reference's equation
text and the published scheme, including the two places where the scheme does something a reader might not expect: an algebraic row
(no dt term) under CNAB1 gives  L X_n = -L X_{n-1}  (k . B flips sign each step; the k = 0 rows "A = 0" give X_n = -X_{n-1}).

PARITY still UNPINNED against Dedalus itself (absent here): this module restates Dedalus' documented pencil formulation,
it does not run Dedalus.  Derivative convention: dx(f) of the mode exp(+i k.x) is i kx f (Dedalus' Fourier basis)."""
import numpy as np

SCHEMES = {                      # dedalus/core/timesteppers.py (v2): class SBDF1 / class CNAB1, compute_coefficients
    "SBDF1": lambda dt: ((1. / dt, -1. / dt), (1., 0.), (0., 1.)),
    "CNAB1": lambda dt: ((1. / dt, -1. / dt), (0.5, 0.5), (0., 1.)),
}


def imex_step(M, L, X_prev, F_prev, scheme, dt):
    """One step of a one-step multistep IMEX scheme on one pencil: returns X_n (dense solve, no structure used)."""
    a, b, c = SCHEMES[scheme](dt)
    lhs = a[0] * M + b[0] * L
    rhs = c[1] * F_prev - a[1] * (M @ X_prev) - b[1] * (L @ X_prev)
    return np.linalg.solve(lhs, rhs)


def kdyn_forward_pencil(k, Rm):
    """Variables [Pi, A, B, C] (FWD_Solve_KDyn.py:395).  Rows in the order of the add_equation calls."""
    kx, ky, kz = k
    k2 = kx * kx + ky * ky + kz * kz
    M = np.zeros((4, 4), dtype=complex)
    L = np.zeros((4, 4), dtype=complex)
    if k2 == 0:
        # :431-434  "A = 0", "B = 0", "C = 0", "Pi = 0"  (condition nx == ny == nz == 0): algebraic rows, L = identity
        L[0, 1] = L[1, 2] = L[2, 3] = L[3, 0] = 1.
        return M, L
    # :425  "dx(A) + dy(B) + dz(C) = 0"
    L[0, 1], L[0, 2], L[0, 3] = 1j * kx, 1j * ky, 1j * kz
    # :426-428  "dt(A) - (1./Rm)*Lap(A) - dx(Pi) = INDx(...)"   Lap -> -k^2
    for r, kk in ((1, kx), (2, ky), (3, kz)):
        M[r, r] = 1.
        L[r, r] = k2 / Rm
        L[r, 0] = -1j * kk
    return M, L


def kdyn_adjoint_pencil(k, Rm):
    """Variables [Pi, G_A, G_B, G_C, P, nu_u, nu_v, nu_w] (FWD_Solve_KDyn.py:807)."""
    kx, ky, kz = k
    k2 = kx * kx + ky * ky + kz * kz
    M = np.zeros((8, 8), dtype=complex)
    L = np.zeros((8, 8), dtype=complex)
    if k2 == 0:
        # :869-872 and :882-885: "G_A = 0" ... "Pi = 0", "nu_u = 0" ... "P = 0"
        L[0, 1] = L[1, 2] = L[2, 3] = L[3, 0] = 1.
        L[4, 5] = L[5, 6] = L[6, 7] = L[7, 4] = 1.
        return M, L
    # :856  "dx(G_A) + dy(G_B) + dz(G_C) = 0"
    L[0, 1], L[0, 2], L[0, 3] = 1j * kx, 1j * ky, 1j * kz
    # :858-860 / :863-865  "dt(G_A) - (1./Rm)*Lap(G_A) - dx(Pi) = [-2.*Af +] F_x(G; uf,vf,wf)"
    for r, kk in ((1, kx), (2, ky), (3, kz)):
        M[r, r] = 1.
        L[r, r] = k2 / Rm
        L[r, 0] = -1j * kk
    # :876  "dx(nu_u) + dy(nu_v) + dz(nu_w) = 0"
    L[4, 5], L[4, 6], L[4, 7] = 1j * kx, 1j * ky, 1j * kz
    # :877-879  "dt(nu_u) + dx(P) = -F_x(G; Af,Bf,Cf)"
    for r, kk in ((5, kx), (6, ky), (7, kz)):
        M[r, r] = 1.
        L[r, 4] = 1j * kk
    return M, L


def sh23_pencil(k, a):
    """Variable [u] (FWD_Solve_SH23.py:322 "dt(u) + Lap(u) - a*u = ...", :316 Lap(f) = f + 2 dx dx f + dx dx dx dx f -> (1 - k^2)^2)."""
    return np.array([[1. + 0j]]), np.array([[(1. - k * k) ** 2 - a + 0j]])


def kdyn_compat_pencil(k, Rm, dt, cost):
    """Compatib_Cond's LBVP, variables [Pi, A, B, C] (FWD_Solve_KDyn.py:733-747): returns the matrix L of  L X = rhs,  rhs = [0, -2 fx, -2 fy, -2 fz]
    (k = 0: the rows "A = 0" ..., right-hand side 0)."""
    kx, ky, kz = k
    k2 = kx * kx + ky * ky + kz * kz
    L = np.zeros((4, 4), dtype=complex)
    if k2 == 0:
        L[0, 1] = L[1, 2] = L[2, 3] = L[3, 0] = 1.
        return L
    L[0, 1], L[0, 2], L[0, 3] = 1j * kx, 1j * ky, 1j * kz            # :736 / :742  "dx(A) + dy(B) + dz(C) = 0"
    for r, kk in ((1, kx), (2, ky), (3, kz)):
        # Final (:733-735):      "A - dt*(.5/Rm)*Lap(A) - dx(Pi) = -2.*fx"
        # Integrated (:739-741): "A/dt - (.5/Rm)*Lap(A) - dx(Pi) = -2.*fx"
        L[r, r] = (1. + dt * 0.5 * k2 / Rm) if cost == "Final" else (1. / dt + 0.5 * k2 / Rm)
        L[r, 0] = -1j * kk
    return L

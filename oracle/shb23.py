"""ORACLE (test infrastructure, never shipped) — CPU restatement of the reference's 1-D bounded
(Chebyshev) Swift-Hohenberg "Discrete" forward / adjoint path.

PINNED PARTS: the four Chebyshev transforms, the quadrature weights and the weighted inner product are
checked against vectors produced by the reference's own helper functions (tests/golden/shb_helpers.npz,
from FWD_Solve_SHB23.py:36-81,189-193).
UNPINNED PART: the tau-system solve — the reference obtains it from Dedalus-v2 internals
(`pencil_matsolvers`, `pre_left`, `pre_right`, `L_exp`; FWD_Solve_SHB23.py:563-587,653-659,857-859) which
cannot run here.  It is restated from SURVEY.md Appendix A.0-8 / A.3:

    unknowns [u, uz, uzz, uzzz] (T-coefficients, N each);  D = T->T derivative (2/L stretch)
    (1/dt + 1 - a) u + 2 uzz + D uzzz = rhs ;  uz - D u = 0 ;  uzz - D uz = 0 ;  uzzz - D uzz = 0
    every block left-multiplied by Pre (T->U conversion), its LAST row replaced by one boundary row:
    left(uz) = 0, left(uzzz) = 0, right(u) = 0, right(uzz) = 0     (left(f)=sum (-1)^n f_n, right(f)=sum f_n)

S is the N x N map "rhs of the first equation -> u".  The adjoint uses S^T (transposed LU in the reference).

Loops:  FWD_Solve_SHB23.py:624-678 (forward), :841-848 (NLtermAdj quirk: the dealiased copy is discarded),
        :886-920 (adjoint).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from scipy import fft as sfft


# -- Chebyshev transforms on the ascending Gauss grid z_i = -cos(pi (i+1/2)/N) and exact transposes ------
def transform(x):
    """grid -> T coefficients (FWD_Solve_SHB23.py:36-40)."""
    b = sfft.dct(x, type=2) / len(x)
    b[0] *= 0.5
    b[1::2] *= -1
    return b


def transformInverse(c):
    """T coefficients -> grid (FWD_Solve_SHB23.py:51-57)."""
    w = np.array(c, dtype=float)
    w[1::2] *= -1
    w[1:] *= 0.5
    return sfft.dct(w, type=3)


def transformAdjoint(x):
    """transpose of `transform` (FWD_Solve_SHB23.py:42-49): T^T = (1/N) C3 diag(1/2, -1, 1, -1, ...) scaled."""
    n = len(x)
    w = np.array(x, dtype=float)
    w[0] *= 0.5
    w[1::2] *= -1
    w[0] *= np.sqrt(4 * n)
    w[1:] *= np.sqrt(2 * n)
    return sfft.dct(w, type=3, norm='ortho') / n


def transformInverseAdjoint(x):
    """transpose of `transformInverse` (FWD_Solve_SHB23.py:59-67)."""
    n = len(x)
    b = sfft.dct(x, type=2, norm='ortho')
    b[0] *= np.sqrt(n)
    b[1:] *= 2. * np.sqrt(n / 2.)
    b[1:] *= 0.5
    b[1::2] *= -1
    return b


def gauss_grid(N, interval=(-20., 20.)):
    c = 0.5 * (interval[0] + interval[1]); h = 0.5 * (interval[1] - interval[0])
    return c + h * (-np.cos(np.pi * (np.arange(N) + 0.5) / N))


def weights(z):
    """weightMatrixDisc (FWD_Solve_SHB23.py:69-81): trapezoid-like weights on the Gauss grid."""
    W = np.empty_like(z)
    W[0] = 0.5 * (z[1] - z[0])
    W[-1] = 0.5 * (z[-1] - z[-2])
    W[1:-1] = 0.5 * (z[1:-1] - z[:-2]) + 0.5 * (z[2:] - z[1:-1])
    return W


def tau_operator(N, dt, a=-0.1, interval=(-20., 20.)):
    """Dense S (N x N): rhs (T coefficients of eq. 1) -> u (T coefficients)."""
    stretch = 0.5 * (interval[1] - interval[0])
    Pre = np.zeros((N, N))
    for n in range(N):
        Pre[n, n] = 1. if n == 0 else 0.5
        if n >= 2:
            Pre[n - 2, n] = -0.5
    D = np.zeros((N, N))
    for i in range(N - 1):
        for j in range(i + 1, N, 2):
            D[i, j] = (j if i == 0 else 2. * j) / stretch
    I = np.eye(N); Z = np.zeros((N, N))
    blocks = [[(1. / dt + 1. - a) * I, Z, 2. * I, D],
              [-D, I, Z, Z],
              [Z, -D, I, Z],
              [Z, Z, -D, I]]
    M = np.block([[Pre @ b for b in row] for row in blocks])
    left = (-1.) ** np.arange(N); right = np.ones(N)
    bcs = [(1, left), (3, left), (0, right), (2, right)]             # (variable index, functional), in add_bc order
    for e, (var, fun) in enumerate(bcs):
        r = e * N + N - 1
        M[r, :] = 0.
        M[r, var * N:(var + 1) * N] = fun
    rhs_map = np.zeros((4 * N, N))
    rhs_map[:N] = Pre
    rhs_map[N - 1] = 0.
    return np.linalg.solve(M, rhs_map)[:N]


class SHB23Oracle:
    def __init__(self, Npts=512, interval=(-20., 20.), dt=1e-2, N_ITERS=2000, a=-0.1):
        self.N = int(Npts)
        self.dt, self.N_ITERS, self.a = float(dt), int(N_ITERS), a
        self.Lz = float(interval[1] - interval[0])
        self.z = gauss_grid(self.N, interval)
        self.W = weights(self.z)
        self.S = tau_operator(self.N, self.dt, a, interval)
        self.stack = None                                 # 'A_fwd': grid states, shape (N, N_ITERS+1)

    def inner(self, x, y):
        """Inner_Prod_Discrete (FWD_Solve_SHB23.py:189-193)."""
        return float(np.dot(x, self.W * y) / self.Lz)

    def nl(self, c):
        g = transformInverse(c)
        h = transform(2. * g ** 2 - g ** 3)
        h[self.N // 2:] = 0.                              # dealias, FWD_Solve_SHB23.py:584
        return h

    def forward(self, X):
        dt, n_it = self.dt, self.N_ITERS
        self.stack = np.zeros((self.N, n_it + 1))
        c = transform(np.asarray(X[0], dtype=float))
        g = transformInverse(c)
        self.stack[:, 0] = g
        cost = self.inner(g, g) * dt
        for i in range(n_it):
            c = self.S @ (self.nl(c) + c / dt)
            g = transformInverse(c)
            self.stack[:, i + 1] = g
            cost += self.inner(g, g) * dt
        return -cost

    def adjoint(self, X=None):
        dt, n_it, W = self.dt, self.N_ITERS, self.W
        p = 2. * transformInverseAdjoint(W * self.stack[:, n_it]) * dt
        for i in range(n_it):
            r = self.S.T @ p
            b = self.stack[:, n_it - 1 - i]
            nl_adj = transformInverseAdjoint((4. * b - 3. * b ** 2) * transformAdjoint(r))   # no Z^T: SHB:842-845
            p = r / dt + nl_adj + 2. * transformInverseAdjoint(W * b) * dt
        return [-transformAdjoint(p) / W]


def synthetic_ic(oracle, seed, M0, prep_steps=100):
    """seeded noise, upper 3/4 of the T-coefficients zeroed (frac=0.25, SHB:256), then
    `prep_steps` forward steps so the vector satisfies the boundary conditions (as SHB:260), scaled to <X,X>=M0."""
    N = oracle.N
    c = transform(np.random.RandomState(seed).standard_normal(N))
    c[np.linspace(0, 1, N, endpoint=False) > 0.25] = 0.
    for _ in range(prep_steps):
        c = oracle.S @ (oracle.nl(c) + c / oracle.dt)
    g = transformInverse(c)
    return g * np.sqrt(M0 / oracle.inner(g, g))


# ---------------------------------------------------------------------------------------------------------------
# "Continuous" path (Adjoint_type = "Continuous", FWD_Solve_SHB23.py:322-355, 398-523, 685-794, 156-187): a Dedalus IVP
# with SBDF1 on N Chebyshev modes and dealias = 2 — PARITY UNPINNED like the discrete tau solve (same Dedalus internals).
#   * state = N T-coefficients; products on the scale-2 Gauss grid (2N points), transformed back and truncated to N modes
#   * every step: (M/dt + L) X1 = M X0/dt + F  ->  c1 = S (c0/dt + F^),  S = the same tau operator, built for N modes
#   * J = dt * sum_{n=0}^{N_ITERS} (1/Lz) integ(u_n^2);  integ acts on the truncated coefficients:
#         integ(f) = (Lz/2) * sum_{k even} f_k * 2/(1-k^2)
#     <x,y> (Inner_Prod_Cnts) = (1/Lz) integ(x*y) with x, y given on the scale-2 grid (no truncation of x, y themselves)
#   * continuous adjoint: q(0) = 0; q1 = S (q0/dt + trunc T[(4 uf - 3 uf^2) q - 2 uf]), uf from the coefficient snapshots
#     N, N-1, ..., 1; gradient = q on the scale-2 grid (no "undo LHS").
# Flat vectors live on the scale-2 grid (2N values); the snapshot stack holds coefficients (N, N_ITERS+1).
# ---------------------------------------------------------------------------------------------------------------
class SHB23CntsOracle:
    def __init__(self, Npts=256, interval=(-20., 20.), dt=1e-2, N_ITERS=2000, a=-0.1):
        self.N, self.G = int(Npts), 2 * int(Npts)
        self.dt, self.N_ITERS, self.a = float(dt), int(N_ITERS), a
        self.Lz = float(interval[1] - interval[0])
        self.S = tau_operator(self.N, self.dt, a, interval)
        k = np.arange(self.N)
        self.w_int = np.where(k % 2 == 0, 2. / np.where(k == 1, 1., 1. - k.astype(float) ** 2), 0.) * (self.Lz / 2.)
        self.stack = None

    def to_grid(self, c):
        p = np.zeros(self.G); p[:self.N] = c
        return transformInverse(p)

    def to_coeff(self, g):
        return transform(g)[:self.N]

    def integ_mean(self, g):
        return float(np.dot(self.w_int, self.to_coeff(g)) / self.Lz)

    def inner(self, x, y):
        return self.integ_mean(np.asarray(x) * np.asarray(y))

    def forward(self, X):
        dt, n_it = self.dt, self.N_ITERS
        self.stack = np.zeros((self.N, n_it + 1))
        c = self.to_coeff(np.asarray(X[0], dtype=float))            # the state is truncated by the solver (A.0-4)
        J = 0.
        for n in range(n_it + 1):
            self.stack[:, n] = c
            g = self.to_grid(c)
            J += dt * self.integ_mean(g * g)
            c = self.S @ (c / dt + self.to_coeff(2. * g ** 2 - g ** 3))
        return -J

    def adjoint(self, X=None):
        dt, n_it = self.dt, self.N_ITERS
        q = np.zeros(self.N)
        idx = n_it
        for _ in range(n_it):
            uf = self.to_grid(self.stack[:, idx]); idx -= 1
            qg = self.to_grid(q)
            q = self.S @ (q / dt + self.to_coeff((4. * uf - 3. * uf ** 2) * qg - 2. * uf))
        return [self.to_grid(q)]


def synthetic_ic_cnts(oracle, seed, M0, prep_steps=100):
    """Noise on the scale-2 grid, upper half of the N modes removed (frac = 0.5, SHB:258), smoothed by `prep_steps`+1 steps,
    scaled to <X,X> = M0 (SHB:195-268 with Adjoint_type = "Continuous")."""
    N = oracle.N
    c = oracle.to_coeff(np.random.RandomState(seed).standard_normal(oracle.G))
    c[np.linspace(0, 1, N, endpoint=False) > 0.5] = 0.
    for _ in range(prep_steps + 1):
        g = oracle.to_grid(c)
        c = oracle.S @ (c / oracle.dt + oracle.to_coeff(2. * g ** 2 - g ** 3))
    g = oracle.to_grid(c)
    return g * np.sqrt(M0 / oracle.inner(g, g))

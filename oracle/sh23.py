"""ORACLE (test infrastructure, never shipped) — CPU restatement of the reference's
1-D periodic Swift-Hohenberg 2-3 forward / adjoint path.

PARITY UNPINNED: the reference executes this path through Dedalus v2, which is not installed
here and has no golden vectors in the reference.  The recurrences below
restate SURVEY.md Appendix A.1, derived from

    FWD_Solve_SH23.py:308-325   equation  dt(u) + (1+dx^2)^2 u - a u = 1.8 u^2 - u^3, a=-0.3, SBDF1
    FWD_Solve_SH23.py:409-545   forward loop: N_ITERS+1 steps, snapshot of u['c'] before each step,
                                J += dt * (1/L) integ(u^2) evaluated on the state *before* each step
    FWD_Solve_SH23.py:552-596   compatibility condition  (1/dt + Lap - a) q = -2 u_N
    FWD_Solve_SH23.py:598-729   adjoint loop and the final "undo LHS" multiply
    FWD_Solve_SH23.py:158-172   inner product = (1/L) integ(x*y) on the scale-2 grid

and the Dedalus-v2 conventions of Appendix A.0 (amplitude-normalised Fourier coefficients,
k = 0..(N-1)//2 with the Nyquist mode dropped, products on the dealias=2 grid then truncated).
Self-consistency is checked by the Taylor test in tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from scipy import fft as sfft


class SH23Oracle:
    def __init__(self, Npts=256, interval=(0., 12. * np.pi), dt=0.1, N_ITERS=500, a=-0.3, dealias=2, workers=1):
        self.N = int(Npts)
        self.G = int(dealias * Npts)                     # FWD_Solve_SH23.py:202-204
        self.Nc = (self.N - 1) // 2 + 1                  # Nyquist dropped (A.0-1)
        self.L = float(interval[1] - interval[0])
        self.dt = float(dt)
        self.N_ITERS = int(N_ITERS)
        self.a = a
        k = 2. * np.pi * np.arange(self.Nc) / self.L
        self.k = k
        self.Lk = (1. - k ** 2) ** 2 - a                 # Lap(u) - a*u, FWD_Solve_SH23.py:316,322
        self.A = 1. / self.dt + self.Lk                  # SBDF1 LHS  a_0 M + b_0 L
        self.workers = workers
        self.stack = None                                # 'A_fwd' of GEN_BUFFER, shape (Nc, N_ITERS+1)

    # -- transforms (amplitude normalisation) -------------------------------------------------
    def to_coeff(self, g):
        return sfft.rfft(g, workers=self.workers)[:self.Nc] / self.G

    def to_grid(self, c):
        pad = np.zeros(self.G // 2 + 1, dtype=complex)
        pad[:self.Nc] = c
        return sfft.irfft(pad, n=self.G, workers=self.workers) * self.G

    # -- callbacks -------------------------------------------------------------------------------
    def inner(self, x, y):
        """Inner_Prod (FWD_Solve_SH23.py:158-172): grid mean of x*y, no truncation (A.0-4)."""
        return float(np.mean(np.asarray(x) * np.asarray(y)))

    def forward(self, X):
        """FWD_Solve_IVP_Lin: returns -J and fills the snapshot stack."""
        dt, n_it = self.dt, self.N_ITERS
        self.stack = np.zeros((self.Nc, n_it + 1), dtype=complex)
        uh = self.to_coeff(np.asarray(X[0], dtype=float))
        J = 0.
        for n in range(n_it + 1):
            self.stack[:, n] = uh
            u = self.to_grid(uh)
            J += dt * np.mean(u * u)
            uh = (uh / dt + self.to_coeff(1.8 * u * u - u * u * u)) / self.A
        return -J

    def adjoint(self, X=None, Adjoint_type="Discrete"):
        """ADJ_Solve_IVP_Lin: gradient on the scale-2 grid; needs forward() at the same X first."""
        dt, n_it = self.dt, self.N_ITERS
        S = self.stack
        if Adjoint_type == "Discrete":
            qh = -2. * S[:, n_it] / self.A               # Compatib_Cond, FWD_Solve_SH23.py:584
            idx = n_it - 1                               # snapshot_index = -2
        else:
            qh = np.zeros(self.Nc, dtype=complex)        # FWD_Solve_SH23.py:647
            idx = n_it                                   # snapshot_index = -1
        for _ in range(n_it):
            uf = self.to_grid(S[:, idx]); idx -= 1
            q = self.to_grid(qh)
            qh = (qh / dt + self.to_coeff((3.6 * uf - 3. * uf * uf) * q - 2. * uf)) / self.A
        if Adjoint_type == "Discrete":
            return [self.to_grid(dt * self.A * qh)]      # FWD_Solve_SH23.py:707-715
        return [self.to_grid(qh)]


def synthetic_ic(G, seed, E0, inner=None):
    """SURVEY.md section 8d recipe: seeded band-limited Gaussian noise on the G-point grid,
    modes with index fraction > 1/2 of the N/2 retained coefficients zeroed, scaled to <X,X> = E0."""
    noise = np.random.RandomState(seed).standard_normal(G)
    c = np.fft.rfft(noise) / G
    nc = G // 4                       # coefficient count at dealias 2
    keep = np.arange(G // 2 + 1) < nc
    frac = np.linspace(0, 1, nc, endpoint=False)
    mask = np.zeros(G // 2 + 1, dtype=bool)
    mask[:nc] = frac <= 0.5
    c[~(keep & mask)] = 0
    x = np.fft.irfft(c, n=G) * G
    return x * np.sqrt(E0 / np.mean(x * x))

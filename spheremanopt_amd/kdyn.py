"""3-D kinematic dynamo — the reference's callbacks, backed by the HIP kernels of csrc/kdyn.hip.

Same names / positional signatures as Example_Problems/Periodic_Domain(Fourier)/Kinematic_Dynamo/FWD_Solve_KDyn.py:

    FWD_Solve_IVP_Lin(X0, domain, Rm, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, Cost_function="Final", Adjoint_type="Discrete")  :529
    ADJ_Solve_IVP_Lin(X0, domain, Rm, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, Cost_function="Final", Adjoint_type="Discrete")  :766
    Inner_Prod_3(x, y, domain, random_arg=None)                                                                              :173
    Generate_IC(Npts, X, M_0, U_Noise) -> (domain, Bx0, Ux0)                                                                 :183
    GEN_BUFFER(Npts, domain, N_SUB_ITERS) -> {'A_fwd','B_fwd','C_fwd'}                                                       :319

X0 = [B0, U]: two flat float64 vectors of length 3*G^3 (x,y,z components concatenated, each C-ordered [x][y][z] on the
3/2-dealiased grid, G = 3*Npts/2).  ``domain`` carries a :class:`KDynDomain`; the X_FWD_DICT entries are handles to the
HBM-resident snapshot stack.
"""
import numpy as np

from . import _capi


class SnapshotStack:
    """'A_fwd' / 'B_fwd' / 'C_fwd' of GEN_BUFFER: ``stack[:, :, :, i]`` reads the (a, m, m) coefficients of snapshot i."""

    def __init__(self, comp, shape):
        self.comp, self.shape, self.ctx = comp, shape, None

    def __getitem__(self, key):
        if self.ctx is None:
            raise RuntimeError("snapshot stack is empty: run FWD_Solve_IVP_Lin first")
        idx = key[-1]
        n = self.shape[-1]
        idx = idx + n if idx < 0 else idx
        a, m = self.shape[0], self.shape[1]
        c = self.ctx.snapshot(idx).view(np.complex128).reshape(3, a, m, m)[self.comp]
        return c[key[:-1]]


class KDynDomain:
    def __init__(self, Npts, X=(0., 2. * np.pi), device=0, ckpt=1):
        """ckpt: keep every ckpt-th snapshot and recompute the rest during the adjoint (1 = keep all like the reference,
        0 = smallest interval whose stack fits the free HBM)."""
        self.Npts, self.interval, self.device, self.ckpt = int(Npts), (float(X[0]), float(X[1])), device, ckpt
        self.G = 3 * self.Npts // 2
        self.kmax = (self.Npts - 1) // 2
        self.a, self.m = self.kmax + 1, 2 * self.kmax + 1
        self.hypervolume = (self.interval[1] - self.interval[0]) ** 3
        self.vec_len = 3 * self.G ** 3
        self._ctx = {}

    def context(self, Rm, dt, N_ITERS, Cost_function="Final"):
        key = (float(Rm), float(dt), int(N_ITERS), Cost_function)
        if key not in self._ctx:
            self._ctx[key] = _capi.Context(_capi.SMO_KDYN, self.Npts, self.interval, dt, N_ITERS, Rm, cost=Cost_function,
                                           device=self.device, ckpt=self.ckpt)
        return self._ctx[key]

    def drop_contexts(self):
        for c in self._ctx.values():
            c.close()
        self._ctx = {}

    def any_context(self):
        if not self._ctx:
            self.context(1., 1e-3, 1)
        return next(iter(self._ctx.values()))


def synthetic_field(G, seed, M0=1.0):
    """Seeded solenoidal, mean-free, band-limited (|k_i| <= N/6) random vector field on the G^3 grid with <X,X> = M0
    (SURVEY.md section 8d).  Returns the flat 3*G^3 vector."""
    N = (2 * G) // 3
    kcut = N // 6
    rs = np.random.RandomState(seed)
    kx = np.fft.rfftfreq(G, 1. / G)
    kc = np.fft.fftfreq(G, 1. / G)
    K = np.stack(np.meshgrid(kc, kc, kx, indexing='ij'))
    keep = (np.abs(K[0]) <= kcut) & (np.abs(K[1]) <= kcut) & (np.abs(K[2]) <= kcut)
    k2 = (K ** 2).sum(0)
    k2[0, 0, 0] = 1.
    V = np.stack([np.fft.rfftn(rs.standard_normal((G, G, G))) for _ in range(3)]) * keep
    V = V - K * ((K * V).sum(0) / k2)
    V[:, 0, 0, 0] = 0.
    v = np.stack([np.fft.irfftn(V[i], s=(G, G, G), axes=(0, 1, 2)) for i in range(3)])
    v *= np.sqrt(M0 / np.mean((v * v).sum(0)))
    return v.reshape(-1)


def Generate_IC(Npts, X=(0., 2. * np.pi), M_0=1.0, U_Noise=False, seeds=(1, 2), device=0):
    """Domain + (B0, U) with <B0,B0> = M_0, <U,U> = 1.  U_Noise=False gives the reference's analytic flow
    (FWD_Solve_KDyn.py:258-260, normalised); otherwise both fields are synthetic random solenoidal fields."""
    dom = KDynDomain(Npts, X, device=device)
    G = dom.G
    B = synthetic_field(G, seeds[0], M_0)
    if U_Noise:
        U = synthetic_field(G, seeds[1], 1.0)
    else:
        s = X[0] + (X[1] - X[0]) * np.arange(G) / G
        x, y, z = np.meshgrid(s, s, s, indexing='ij')
        U = np.stack([np.sin(y) * np.cos(z), np.sin(z) * np.cos(x), np.sin(x) * np.cos(y)]) * (0.5 / np.sqrt(3.))
        U = (U / np.sqrt(np.mean((U * U).sum(0)))).reshape(-1)
    return dom, B, U


def GEN_BUFFER(Npts, domain, N_SUB_ITERS):
    shape = (domain.a, domain.m, domain.m, N_SUB_ITERS + 1)
    return {'A_fwd': SnapshotStack(0, shape), 'B_fwd': SnapshotStack(1, shape), 'C_fwd': SnapshotStack(2, shape)}


def _check_window(N_ITERS, N_SUB_ITERS):
    if N_SUB_ITERS != N_ITERS:
        raise NotImplementedError("windowed checkpointing (N_SUB_ITERS < N_ITERS) is not implemented (nor in the reference)")


def FWD_Solve_IVP_Lin(X0, domain, Rm, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, Cost_function="Final", Adjoint_type="Discrete"):
    """-J(B0, U): J = <B_N,B_N> ("Final") or dt*sum_n <B_n,B_n> ("Integrated"); fills the device snapshot stack."""
    _check_window(N_ITERS, N_SUB_ITERS)
    ctx = domain.context(Rm, dt, N_ITERS, Cost_function)
    J = ctx.forward([X0[0], X0[1]])
    for k in ('A_fwd', 'B_fwd', 'C_fwd'):
        X_FWD_DICT[k].ctx = ctx
    return J


def ADJ_Solve_IVP_Lin(X0, domain, Rm, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, Cost_function="Final", Adjoint_type="Discrete"):
    """[dJ/dB0, dJ/dU] as flat grid vectors; valid right after FWD_Solve_IVP_Lin at the same X0."""
    _check_window(N_ITERS, N_SUB_ITERS)
    ctx = domain.context(Rm, dt, N_ITERS, Cost_function)
    return ctx.adjoint(None, Adjoint_type)


def Inner_Prod_3(x, y, domain, random_arg=None):
    """Sum over the three components of the grid mean of x*y."""
    return domain.any_context().inner(x, y)


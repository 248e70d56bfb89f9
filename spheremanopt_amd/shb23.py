"""1-D bounded (Chebyshev) Swift-Hohenberg, "Discrete" path — the reference's callbacks backed by csrc/shb23.hip.

Same names / positional signatures as Example_Problems/Bounded_Domain(Cheby)/Swift_Hohenberg_Bounded/FWD_Solve_SHB23.py:

    FWD_Solve_IVP_Discrete(X_k, domain, X_FWD_DICT, N_ITERS, dt=1e-02, filename=None)     :525
    ADJ_Solve_IVP_Discrete(X_k, domain, X_FWD_DICT, N_ITERS, dt=1e-02, filename=None)     :796
    Inner_Prod_Discrete(x, y, domain, Type_xy='np_vector')                                :189
    transform / transformInverse / transformAdjoint / transformInverseAdjoint            :36-67
    weightMatrixDisc(domain)                                                              :69
    Generate_IC(Npts, Z, M_0) -> (domain, X_0)                                            :195
    GEN_BUFFER(Npts, domain, N_SUB_ITERS) -> {'A_fwd': handle}                            :270

and the module-level switch of the reference (:951-965): Inner_Prod / FWD_Solve / ADJ_Solve = the Discrete variants.
"""
import numpy as np

from . import _capi

A_PARAM = -0.1      # FWD_Solve_SHB23.py:564


class SnapshotStack:
    """'A_fwd' of GEN_BUFFER (grid states): ``stack[:, i]`` reads snapshot i from HBM."""

    def __init__(self, shape):
        self.shape, self.ctx = shape, None

    def __getitem__(self, key):
        if self.ctx is None:
            raise RuntimeError("snapshot stack is empty: run FWD_Solve_IVP_Discrete first")
        rows, idx = key
        n = self.shape[1]
        idx = idx + n if idx < 0 else idx
        return self.ctx.snapshot(idx)[rows]


class SHBDomain:
    def __init__(self, Npts=512, Z=(-20., 20.), device=0):
        self.Npts, self.interval, self.device = int(Npts), (float(Z[0]), float(Z[1])), device
        self.hypervolume = self.interval[1] - self.interval[0]
        self._ctx = {}

    def grid(self, axis=0, scales=1):
        c, h = 0.5 * sum(self.interval), 0.5 * self.hypervolume
        return c + h * (-np.cos(np.pi * (np.arange(self.Npts) + 0.5) / self.Npts))

    def context(self, dt, N_ITERS, batch=1):
        key = (float(dt), int(N_ITERS), int(batch))
        if key not in self._ctx:
            self._ctx[key] = _capi.Context(_capi.SMO_SHB23, self.Npts, self.interval, dt, N_ITERS, A_PARAM, batch=batch,
                                           device=self.device)
        return self._ctx[key]

    def any_context(self):
        if not self._ctx:
            self.context(1e-2, 1)
        return next(iter(self._ctx.values()))


def weightMatrixDisc(domain):
    z = domain.grid(0)
    W = np.empty_like(z)
    W[0] = 0.5 * (z[1] - z[0])
    W[-1] = 0.5 * (z[-1] - z[-2])
    W[1:-1] = 0.5 * (z[1:-1] - z[:-2]) + 0.5 * (z[2:] - z[1:-1])
    return W


def _transform(which, x, domain):
    dom = domain if domain is not None else SHBDomain(len(x))
    return dom.any_context().transform(which, x)


def transform(x, domain=None):
    return _transform(0, x, domain)


def transformInverse(x, domain=None):
    return _transform(1, x, domain)


def transformAdjoint(x, domain=None):
    return _transform(2, x, domain)


def transformInverseAdjoint(x, domain=None):
    return _transform(3, x, domain)


def GEN_BUFFER(Npts, domain, N_SUB_ITERS):
    return {'A_fwd': SnapshotStack((domain.Npts, N_SUB_ITERS + 1))}


def Generate_IC(Npts, Z=(-20., 20.), M_0=1.0, seed=42, dt=1e-2, prep_steps=100, device=0):
    """Domain + initial condition with <X,X> = M_0 that satisfies the boundary conditions: seeded noise with the upper
    3/4 of the Chebyshev modes removed (FWD_Solve_SHB23.py:256), advanced `prep_steps` steps by the device forward
    solver itself (the role of FWD_Solve_IVP_PREP, :260), then normalised."""
    dom = SHBDomain(Npts, Z, device=device)
    ctx = dom.context(dt, prep_steps)
    noise = np.random.RandomState(seed).standard_normal(Npts)
    c = ctx.transform(0, noise)
    c[np.linspace(0, 1, Npts, endpoint=False) > 0.25] = 0.
    ctx.forward([ctx.transform(1, c)])
    g = ctx.snapshot(prep_steps)
    return dom, g * np.sqrt(M_0 / Inner_Prod_Discrete(g, g, dom))


def FWD_Solve_IVP_Discrete(X_k, domain, X_FWD_DICT, N_ITERS, dt=1e-02, filename=None):
    """-J(X), J = dt * sum_{n=0}^{N} <u_n,u_n>_W ; fills the device snapshot stack with the grid states."""
    ctx = domain.context(dt, N_ITERS)
    J = ctx.forward([X_k[0]])
    X_FWD_DICT['A_fwd'].ctx = ctx
    return J


def ADJ_Solve_IVP_Discrete(X_k, domain, X_FWD_DICT, N_ITERS, dt=1e-02, filename=None):
    """[dJ/dX]; valid right after FWD_Solve_IVP_Discrete at the same X_k."""
    return domain.context(dt, N_ITERS).adjoint(None, "Discrete")


def Inner_Prod_Discrete(x, y, domain, Type_xy='np_vector'):
    """x . (W o y) / L_z with the reference's trapezoid-like weights."""
    return domain.any_context().inner(x, y)


Adjoint_type = "Discrete"
Inner_Prod = Inner_Prod_Discrete
FWD_Solve = FWD_Solve_IVP_Discrete
ADJ_Solve = ADJ_Solve_IVP_Discrete

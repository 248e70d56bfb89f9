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
    """dealias = 1: the "Discrete" formulation (Npts Chebyshev modes = Npts grid points, FWD_Solve_SHB23.py:213-215);
    dealias = 2: the "Continuous" one (Npts modes, vectors on the 2*Npts-point scale-2 grid, :216-217)."""

    def __init__(self, Npts=512, Z=(-20., 20.), device=0, dealias=1):
        self.Npts, self.interval, self.device, self.dealias = int(Npts), (float(Z[0]), float(Z[1])), device, int(dealias)
        self.hypervolume = self.interval[1] - self.interval[0]
        self.G = self.Npts * self.dealias
        self._ctx = {}

    def grid(self, axis=0, scales=1):
        n = self.Npts * int(scales)
        c, h = 0.5 * sum(self.interval), 0.5 * self.hypervolume
        return c + h * (-np.cos(np.pi * (np.arange(n) + 0.5) / n))

    def context(self, dt, N_ITERS, batch=1):
        key = (float(dt), int(N_ITERS), int(batch))
        if key not in self._ctx:
            self._ctx[key] = _capi.Context(_capi.SMO_SHB23, self.Npts, self.interval, dt, N_ITERS, A_PARAM, batch=batch,
                                           device=self.device, cost=(1 if self.dealias == 2 else 0))
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
    if dom.dealias != 1:
        raise ValueError("the standalone Chebyshev maps belong to the Discrete formulation (dealias = 1)")
    return dom.any_context().transform(which, x)


def transform(x, domain=None):
    return _transform(0, x, domain)


def transformInverse(x, domain=None):
    return _transform(1, x, domain)


def transformAdjoint(x, domain=None):
    return _transform(2, x, domain)


def transformInverseAdjoint(x, domain=None):
    return _transform(3, x, domain)


def Vec_to_Field(domain, X):
    """Flat vector -> values on the ascending Gauss-Chebyshev grid (FWD_Solve_SHB23.py:128-154); the same array here (a view)."""
    x = np.asarray(X, dtype=np.float64).reshape(-1)
    if x.size != domain.G:
        raise ValueError("vector has %d entries, the grid has %d" % (x.size, domain.G))
    return x


def Field_to_Vec(domain, F):
    """Inverse of Vec_to_Field (FWD_Solve_SHB23.py:87-126)."""
    return np.ascontiguousarray(F, dtype=np.float64).reshape(-1)


def GEN_BUFFER(Npts, domain, N_SUB_ITERS):
    """Grid states (Discrete) or T-coefficients (Continuous) of every step: shape (Npts, N_SUB_ITERS+1) either way (SHB:298-312)."""
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
    J = ctx.forward_any([X_k[0]])
    X_FWD_DICT['A_fwd'].ctx = ctx
    if getattr(domain, "write_products", False) and domain.dealias == 1:      # scalar_data_s1 / CheckPoints_s1 like the reference (:604-672)
        from . import products
        products.write_shb23(domain, ctx, dt, N_ITERS, weightMatrixDisc(domain))
    return J


def File_Manips(k):
    """The reference's optimiser callback (FWD_Solve_SHB23.py:923-948): keep this iteration's scalar_data / CheckPoints files."""
    from . import products
    products.File_Manips(k)


def ADJ_Solve_IVP_Discrete(X_k, domain, X_FWD_DICT, N_ITERS, dt=1e-02, filename=None):
    """[dJ/dX]; valid right after FWD_Solve_IVP_Discrete at the same X_k."""
    return domain.context(dt, N_ITERS).adjoint_any([X_k[0]], "Discrete")


def Inner_Prod_Discrete(x, y, domain, Type_xy='np_vector'):
    """x . (W o y) / L_z with the reference's trapezoid-like weights."""
    return domain.any_context().inner_any(x, y)


# ---- "Continuous" formulation (Dedalus IVP with dealias 2 in the reference; SHB:398-523, 685-794, 156-187) -------------------------

def FWD_Solve_IVP_Cnts(X_k, domain, X_FWD_DICT, N_ITERS, dt=1e-02, filename=None):
    """-J(X), J = dt * sum_n (1/Lz) integ(u_n^2); X on the scale-2 grid (2*Npts values); snapshots = T-coefficients."""
    ctx = domain.context(dt, N_ITERS)
    J = ctx.forward_any([X_k[0]])
    X_FWD_DICT['A_fwd'].ctx = ctx
    return J


def ADJ_Solve_IVP_Cnts(X_k, domain, X_FWD_DICT, N_ITERS, dt=1e-02, filename=None):
    """[q(T)] of the continuous adjoint equation on the scale-2 grid (an O(dt)-consistent approximation of dJ/dX)."""
    return domain.context(dt, N_ITERS).adjoint_any([X_k[0]], "Continuous")


def Inner_Prod_Cnts(x, y, domain, Type_xy='np_vector'):
    """(1/Lz) integ(x*y): product on the scale-2 grid, truncated to Npts modes, integrated exactly."""
    return domain.any_context().inner_any(x, y)


def Generate_IC_Cnts(Npts, Z=(-20., 20.), M_0=1.0, seed=42, dt=1e-2, prep_steps=100, device=0):
    """Continuous-formulation counterpart of Generate_IC (SHB:195-268 with Adjoint_type = "Continuous"): noise on the scale-2
    grid, modes with index fraction > 1/2 removed, 101 smoothing steps on the device, normalised with Inner_Prod_Cnts."""
    dom = SHBDomain(Npts, Z, device=device, dealias=2)
    G = dom.G
    noise = np.random.RandomState(seed).standard_normal(G)
    c = np.zeros(G)
    c[:Npts] = _host_transform(noise)[:Npts]
    c[:Npts][np.linspace(0, 1, Npts, endpoint=False) > 0.5] = 0.
    ctx = dom.context(dt, prep_steps + 1)
    ctx.forward([_host_transform_inverse(c)])
    c[:Npts] = ctx.snapshot(prep_steps + 1)
    g = _host_transform_inverse(c)
    return dom, g * np.sqrt(M_0 / Inner_Prod_Cnts(g, g, dom))


def _host_transform(x):
    """Chebyshev grid -> coefficients on the host (IC generation only; scipy-free DCT-II via the FFT)."""
    n = len(x)
    v = np.concatenate([x[::2], x[::-1][::2]]) if n % 2 == 0 else None
    V = np.fft.fft(v)
    k = np.arange(n)
    b = 2. * np.real(V * np.exp(-1j * np.pi * k / (2 * n))) / n
    b[0] *= 0.5
    b[1::2] *= -1
    return b


def _host_transform_inverse(c):
    n = len(c)
    k = np.arange(n)
    w = np.array(c, dtype=float)
    w[1::2] *= -1
    w[1:] *= 0.5
    i = np.arange(n)
    return w[0] + 2. * (np.cos(np.pi * np.outer(2 * i + 1, k[1:]) / (2 * n)) @ w[1:])


Adjoint_type = "Discrete"
Inner_Prod = Inner_Prod_Discrete
FWD_Solve = FWD_Solve_IVP_Discrete
ADJ_Solve = ADJ_Solve_IVP_Discrete

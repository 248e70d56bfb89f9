"""Stratified plane-Poiseuille optimal mixing ("Discrete" formulation) — the reference's callbacks, backed by csrc/pois.hip.

Same names / positional signatures as Example_Problems/Bounded_Domain(Cheby)/Optimal_Mixing/FWD_Solve_Poiseuille.py:

    FWD_Solve_Discrete(U0, domain, Reynolds, Richardson, N_ITERS, X_FWD_DICT, dt=1e-04, s=0, Prandtl=1., δ=0.25, filename=None)  :777
    ADJ_Solve_Discrete(U0, domain, Reynolds, Richardson, N_ITERS, X_FWD_DICT, dt=1e-04, s=0, Prandtl=1., δ=0.25, Sim_Type=...)   :1320
    Inner_Prod_Discrete(x, y, domain, W=None)                                                                                      :282
    Generate_IC(Nx, Nz, X_domain=(0,4pi), Z_domain=(-1,1), E_0=0.02, ...)  -> (domain, U0)                                         :301
    GEN_BUFFER(Nx, Nz, domain, N_ITERS)  -> {'u_fwd','w_fwd','b_fwd'}                                                              :386
    transform / transformInverse / transformAdjoint / transformInverseAdjoint                                                      :44-89

so the reference's driver lines (:1765-1777) work unchanged:

    args_f = [domain, Re, Ri, N_ITERS, X_FWD_DICT, dt, s, Prandtl, δ];  args_IP = [domain, None]
    Optimise_On_Multi_Sphere([Ux0], [E_0], FWD_Solve, ADJ_Solve, Inner_Prod, args_f, args_IP, ...)

`Nx`, `Nz` are the resolutions FWD_Solve_Discrete works at (the reference multiplies its nominal 256 x 128 by 3/2 first, :1752-1755).
U0 is a LIST holding one flat vector [u.flatten(), w.flatten()] of the (Nx, Nz) grids (z fastest).
"""
import numpy as np

from . import _capi


class SnapshotStack:
    """Handle to one of 'u_fwd' / 'w_fwd' / 'b_fwd': ``stack[:, :, i]`` -> complex (a, Nz) coefficients of snapshot i for the
    non-negative x wavenumbers n = 0..kmax (the reference stores the full complex spectrum; the other half is the Hermitian mirror)."""

    def __init__(self, shape, field):
        self.shape, self.field, self.ctx = shape, field, None

    def __getitem__(self, key):
        if self.ctx is None:
            raise RuntimeError("snapshot stack is empty: run FWD_Solve_Discrete first")
        rows, cols, idx = key
        n = self.shape[2]
        idx = idx + n if idx < 0 else idx
        c = self.ctx.snapshot(idx).view(np.complex128).reshape(3, self.shape[0], self.shape[1])[self.field]
        return c[rows, cols]


class PoiseuilleDomain:
    """Geometry + owner of the device contexts (one per parameter set).  continuous=True: the "Continuous" formulation — Nx, Nz are
    then the MODE counts of the Dedalus domain (dealias 3/2) and the flat vectors live on the (3Nx/2, 3Nz/2) grid."""

    def __init__(self, Nx, Nz, X_domain=(0., 4. * np.pi), device=0, continuous=False):
        self.Nx, self.Nz, self.interval, self.device = int(Nx), int(Nz), (float(X_domain[0]), float(X_domain[1])), device
        self.continuous = bool(continuous)
        self.gshape = (3 * self.Nx // 2, 3 * self.Nz // 2) if self.continuous else (self.Nx, self.Nz)
        self.a = (self.Nx - 1) // 2 + 1                 # non-negative x wavenumbers carried (n = 0..kmax)
        self.ada = (2 * self.Nx // 3) // 2              # de-aliased ones (n < ada)
        self.hypervolume = (self.interval[1] - self.interval[0]) * 2.
        self._ctx = {}

    def grid(self, axis, scales=1):
        if axis == 0:
            return self.interval[0] + (self.interval[1] - self.interval[0]) * np.arange(self.Nx) / self.Nx
        return -np.cos(np.pi * (np.arange(self.Nz) + 0.5) / self.Nz)

    def context(self, Reynolds, Richardson, N_ITERS, dt, s, Prandtl, delta):
        key = (float(Reynolds), float(Richardson), int(N_ITERS), float(dt), int(s), float(Prandtl), float(delta))
        if key not in self._ctx:
            self._ctx[key] = _capi.Context(_capi.SMO_POIS, self.Nx, self.interval, dt, N_ITERS, Reynolds,
                                           cost=int(s) + (2 if self.continuous else 0), device=self.device,
                                           npts2=self.Nz, param2=Richardson, param3=Prandtl, param4=delta)
        return self._ctx[key]

    def any_context(self):
        if not self._ctx:
            self.context(500., 0.05, 1, 5e-3, 0, 1., 0.25)
        return next(iter(self._ctx.values()))

    def drop_contexts(self):
        for c in self._ctx.values():
            c.close()
        self._ctx = {}


def Vec_to_Field(domain, U0):
    """Flat vector -> the (u, w) grid arrays, each (gx, gz) with z fastest (FWD_Solve_Poiseuille.py:209-239); views, no copy."""
    gx, gz = domain.gshape
    x = np.asarray(_vec(U0), dtype=np.float64).reshape(-1)
    if x.size != 2 * gx * gz:
        raise ValueError("vector has %d entries, two fields on the grid have %d" % (x.size, 2 * gx * gz))
    u, w = x.reshape(2, gx, gz)
    return u, w


def Field_to_Vec(domain, Fx, Fz):
    """Inverse of Vec_to_Field (FWD_Solve_Poiseuille.py:160-207)."""
    return np.concatenate([np.asarray(Fx, dtype=np.float64).reshape(-1), np.asarray(Fz, dtype=np.float64).reshape(-1)])


def GEN_BUFFER(Nx, Nz, domain, N_ITERS):
    shape = (domain.a, domain.Nz, N_ITERS + 1)
    return {'u_fwd': SnapshotStack(shape, 0), 'w_fwd': SnapshotStack(shape, 1), 'b_fwd': SnapshotStack(shape, 2)}


def _vec(U0):
    return U0[0] if isinstance(U0, (list, tuple)) else U0


def FWD_Solve_Discrete(U0, domain, Reynolds, Richardson, N_ITERS, X_FWD_DICT, dt=1e-04, s=0, Prandtl=1., δ=0.25, filename=None):
    """Objective: -1/2 dt sum_n <U_n,U_n>  (s = 0: time-averaged kinetic energy)  or  1/2 <grad psi, grad psi>, lap psi = rho(T)  (s = 1:
    mix-norm).  Fills the device snapshot stack."""
    ctx = domain.context(Reynolds, Richardson, N_ITERS, dt, s, Prandtl, δ)
    J = ctx.forward_any([_vec(U0)])
    for k in ('u_fwd', 'w_fwd', 'b_fwd'):
        X_FWD_DICT[k].ctx = ctx
    if getattr(domain, "write_products", False):           # scalar_data_s1 / CheckPoints_s1 like the reference (:945-1151)
        from . import products
        products.write_poiseuille(domain, ctx, dt, N_ITERS, s)
    return J


def File_Manips(k):
    """The reference's optimiser callback (FWD_Solve_Poiseuille.py:1698-1741): keep this iteration's scalar_data / CheckPoints files."""
    from . import products
    products.File_Manips(k)


def ADJ_Solve_Discrete(U0, domain, Reynolds, Richardson, N_ITERS, X_FWD_DICT, dt=1e-04, s=0, Prandtl=1., δ=0.25, Sim_Type="Non_Linear"):
    """[dJ/dU0] with respect to Inner_Prod_Discrete; valid right after FWD_Solve_Discrete at the same U0 (it replays that stack)."""
    return domain.context(Reynolds, Richardson, N_ITERS, dt, s, Prandtl, δ).adjoint_any([_vec(U0)], "Discrete")


def Inner_Prod_Discrete(x, y, domain, W=None):
    """(1/V) sum W (x_u y_u + x_w y_w) with the first-order Gauss-grid weights of weightMatrixDisc (:91-118)."""
    return domain.any_context().inner_any(x, y)


def weightMatrixDisc(domain):
    z = domain.grid(1)
    dz = np.empty(domain.Nz)
    dz[0] = z[1] - z[0]
    dz[1:] = z[1:] - z[:-1]
    return np.tile(dz * ((domain.interval[1] - domain.interval[0]) / domain.Nx), (domain.Nx, 1))


# ---- the four transforms on one real field / one Hermitian coefficient array (device GEMMs; used by the parity tests) ------------------
def _tr(which, x, domain):
    ctx = domain.any_context()
    nG, nC = domain.Nx * domain.Nz, 2 * domain.a * domain.Nz
    if which in (0, 3):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        return ctx.transform(which, x, out_len=nC).view(np.complex128).reshape(domain.a, domain.Nz)
    c = np.ascontiguousarray(x, dtype=np.complex128).reshape(-1).view(np.float64)
    return ctx.transform(which, c, out_len=nG).reshape(domain.Nx, domain.Nz)


def transform(x, domain):
    return _tr(0, x, domain)


def transformInverse(x, domain):
    return _tr(1, x, domain)


def transformAdjoint(x, domain):
    return _tr(2, x, domain)


def transformInverseAdjoint(x, domain):
    return _tr(3, x, domain)


def Generate_IC(Nx, Nz, X_domain=(0., 4. * np.pi), Z_domain=(-1., 1.), E_0=0.02, dealias_scale=1, W=None, seed=42, device=0,
                prep_steps=5, dt=5e-3):
    """Domain + initial condition with <U0,U0> = E_0.  Recipe of Generate_IC (:355-377): seeded noise stream function, low-pass filtered
    (filter_field), u = -psi_z, w = psi_x, then smoothed by a few steps of the forward solver ON THE DEVICE so that it satisfies the
    no-slip walls (the reference runs a Dedalus IVP for that, :374) and scaled.  The noise lives on the (Nx, Nz) grid."""
    dom = PoiseuilleDomain(Nx, Nz, X_domain, device=device)
    a, N = dom.a, dom.Nz
    psi = transform(np.random.RandomState(seed).standard_normal((Nx, Nz)), dom)
    keep = (np.arange(a) <= dom.ada // 2)[:, None] & (np.arange(N) < (2 * N // 3) // 2)[None, :]
    psi = psi * keep
    k = 2. * np.pi * np.arange(a) / (dom.interval[1] - dom.interval[0])
    D = np.zeros((N, N))
    for i in range(N):
        for j in range(i + 1, N):
            D[i, j] = 2. * j * ((j - i) % 2)
    D[0] /= 2.
    u, w = -(psi @ D.T), 1j * k[:, None] * psi
    sc = 1e-2 / np.abs(u).max()
    X = np.concatenate([transformInverse(sc * u, dom).ravel(), transformInverse(sc * w, dom).ravel()])
    if prep_steps > 0:
        ctx = dom.context(500., 0.05, prep_steps, dt, 0, 1., 0.25)
        ctx.forward([X])
        snap = ctx.snapshot(prep_steps).view(np.complex128).reshape(3, a, N).copy()
        snap[:, :, 2 * N // 3:] = 0.                                   # u['c'] *= DA (:608-609)
        X = np.concatenate([transformInverse(snap[0], dom).ravel(), transformInverse(snap[1], dom).ravel()])
    return dom, [X * np.sqrt(E_0 / Inner_Prod_Discrete(X, X, dom))]


# ---- "Continuous" formulation (the reference script's default Adjoint_type, :1728): FWD_Solve_Cnts :614, ADJ_Solve_Cnts :1161,
# Inner_Prod_Cnts :264.  `domain` must be a PoiseuilleDomain(Nx, Nz, continuous=True); vectors live on its (3Nx/2, 3Nz/2) grid. --------------
def _need_cnts(domain):
    if not domain.continuous:
        raise ValueError("the *_Cnts callables need a PoiseuilleDomain(..., continuous=True)")


def FWD_Solve_Cnts(U0, domain, Reynolds, Richardson, N_ITERS, X_FWD_DICT, dt=1e-04, s=0, Prandtl=1., δ=0.25, filename=None):
    """SBDF1 IVP on Nx x Nz modes (N_ITERS+1 steps like the script); cost -1/2 dt sum_{n=0}^{N} (1/V) integ |U_n|^2 (s = 0) or the mix-norm
    of rho(T) (s = 1), integrals exact for the truncated series.  Snapshots: coefficients of u, w, b before every step."""
    _need_cnts(domain)
    ctx = domain.context(Reynolds, Richardson, N_ITERS, dt, s, Prandtl, δ)
    J = ctx.forward_any([_vec(U0)])
    for k in ('u_fwd', 'w_fwd', 'b_fwd'):
        X_FWD_DICT[k].ctx = ctx
    return J


def ADJ_Solve_Cnts(U0, domain, Reynolds, Richardson, N_ITERS, X_FWD_DICT, dt=1e-04, s=0, Prandtl=1., δ=0.25, Sim_Type="Non_Linear"):
    """[u_adj, w_adj] after N_ITERS steps of the script's adjoint IVP on the 3/2 grid: an O(dt)-consistent approximation of dJ/dU0."""
    _need_cnts(domain)
    return domain.context(Reynolds, Richardson, N_ITERS, dt, s, Prandtl, δ).adjoint_any([_vec(U0)], "Continuous")


def Inner_Prod_Cnts(x, y, domain, rand_arg=None):
    """(1/V) integ (x_u y_u + x_w y_w): grid product, truncated to the modes, integrated exactly (Integrate_Field :241-262)."""
    _need_cnts(domain)
    return domain.any_context().inner_any(x, y)


Adjoint_type = "Discrete"
Inner_Prod = Inner_Prod_Discrete
FWD_Solve = FWD_Solve_Discrete
ADJ_Solve = ADJ_Solve_Discrete

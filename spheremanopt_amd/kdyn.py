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
from .devvec import DeviceVector


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
    def __init__(self, Npts, X=(0., 2. * np.pi), device=0, ckpt=1, devices=None):
        """ckpt: keep every ckpt-th snapshot and recompute the rest during the adjoint (1 = keep all like the reference,
        0 = smallest interval whose stack fits the free HBM).
        devices: a list of GPU ordinals — this ONE process then runs the problem slab-decomposed over them (smo_create_multi: no launcher,
        no RCCL; the callbacks keep taking and returning the reference's full vectors)."""
        self.Npts, self.interval, self.device, self.ckpt = int(Npts), (float(X[0]), float(X[1])), device, ckpt
        self.devices = [int(d) for d in devices] if devices is not None and len(devices) > 1 else None
        self.G = 3 * self.Npts // 2
        self.kmax = (self.Npts - 1) // 2
        self.a, self.m = self.kmax + 1, 2 * self.kmax + 1
        self.hypervolume = (self.interval[1] - self.interval[0]) ** 3
        self.vec_len = 3 * self.G ** 3
        self._ctx = {}

    def context(self, Rm, dt, N_ITERS, Cost_function="Final"):
        key = (float(Rm), float(dt), int(N_ITERS), Cost_function)
        if key not in self._ctx:
            if self.devices:
                self._ctx[key] = _capi.MultiContext(self.Npts, self.interval, dt, N_ITERS, Rm, self.devices, cost=Cost_function, ckpt=self.ckpt)
            else:
                self._ctx[key] = _capi.Context(_capi.SMO_KDYN, self.Npts, self.interval, dt, N_ITERS, Rm, cost=Cost_function,
                                               device=self.device, ckpt=self.ckpt)
        return self._ctx[key]

    def drop_contexts(self):
        for c in self._ctx.values():
            c.close()
        self._ctx = {}
        if getattr(self, "_transform_ctx", None) is not None:
            self._transform_ctx.close()
            self._transform_ctx = None

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


def _coeff_to_grid_host(dom, C):
    """(a,m,m) coefficients -> (G,G,G) grid values on the host (one-off use in IC generation; NumPy, not the hot path)."""
    G, a, kmax = dom.G, dom.a, dom.kmax
    sel = np.concatenate([np.arange(0, kmax + 1), np.arange(G - kmax, G)])
    p = np.zeros((G // 2 + 1, G, G), dtype=complex)
    p[np.ix_(np.arange(a), sel, sel)] = C
    p = np.fft.ifft(np.fft.ifft(p, axis=2), axis=1)
    return np.fft.irfft(p, n=G, axis=0) * float(G) ** 3


def _coeff3_to_grid(dom, C3):
    """(3,a,m,m) coefficients -> (3,G,G,G) grid values with the DEVICE transform (smo_transform, which = 1: the passes the solves use)."""
    ctx = getattr(dom, "_transform_ctx", None)             # a context of its own: smo_transform uses the adjoint state as scratch
    if ctx is None:
        ctx = dom._transform_ctx = _capi.Context(_capi.SMO_KDYN, dom.Npts, dom.interval, 1e-3, 1, 1., device=dom.device)
    C3 = np.ascontiguousarray(C3, dtype=np.complex128)
    return ctx.transform(1, C3.view(np.float64).reshape(-1), out_len=3 * dom.G ** 3).reshape(3, dom.G, dom.G, dom.G)


def _curl_noise(dom, seed, frac=0.25):
    """The reference's random solenoidal field (FWD_Solve_KDyn.py:220-243): phi = filtered noise, field = grad(phi) x (1,1,1).
    ``filter_field`` masks by INDEX fraction (> frac zeroed) on the (a, m, m) coefficient array, which removes every negative
    ky / kz (indices beyond m/4) — reproduced deliberately."""
    G, a, m, kmax = dom.G, dom.a, dom.m, dom.kmax
    noise = np.random.RandomState(seed).standard_normal((G, G, G))
    sel = np.concatenate([np.arange(0, kmax + 1), np.arange(G - kmax, G)])
    c = np.fft.fft(np.fft.fft(np.fft.rfft(noise, axis=0)[:a], axis=1)[:, sel], axis=2)[:, :, sel] / float(G) ** 3
    fx, fc = np.linspace(0, 1, a, endpoint=False), np.linspace(0, 1, m, endpoint=False)
    mask = (fx[:, None, None] > frac) | (fc[None, :, None] > frac) | (fc[None, None, :] > frac)
    c[mask] = 0j
    kc = np.concatenate([np.arange(0, kmax + 1), np.arange(-kmax, 0)]).astype(float)
    kx, ky, kz = np.meshgrid(np.arange(a, dtype=float), kc, kc, indexing='ij')
    comps = [1j * (ky - kz) * c, 1j * (kz - kx) * c, 1j * (kx - ky) * c]
    return _coeff3_to_grid(dom, np.stack(comps))


def FWD_Solve_IVP_Prep(Bx0, Ux0, domain, Rm, dt, N_ITERS):
    """Smooth B by N_ITERS+1 induction steps ON THE DEVICE; returns the three components on the grid (FWD_Solve_KDyn.py:452-527)."""
    ctx = _capi.Context(_capi.SMO_KDYN, domain.Npts, domain.interval, dt, N_ITERS + 1, Rm, device=domain.device)
    ctx.forward([Bx0, Ux0])
    C = ctx.snapshot(N_ITERS + 1).view(np.complex128).reshape(3, domain.a, domain.m, domain.m)
    ctx.close()
    return list(_coeff3_to_grid(domain, C))


def _analytic_flow(dom):
    """0.5/sqrt(3) * (sin y cos z, sin z cos x, sin x cos y)  (FWD_Solve_KDyn.py:258-260), on the 3/2 grid."""
    s = dom.interval[0] + (dom.interval[1] - dom.interval[0]) * np.arange(dom.G) / dom.G
    x, y, z = np.meshgrid(s, s, s, indexing='ij')
    return np.stack([np.sin(y) * np.cos(z), np.sin(z) * np.cos(x), np.sin(x) * np.cos(y)]) * (0.5 / np.sqrt(3.))


def _normalised(v, val):
    return (v * np.sqrt(val / np.mean((v * v).sum(0)))).reshape(-1)


def Generate_IC(Npts, X=(0., 2. * np.pi), M_0=1.0, U_Noise=False, seeds=(1, 2), device=0, reference_recipe=False, Rm=1.0, dt=1e-3):
    """Domain + (B0, U) with <B0,B0> = M_0, <U,U> = 1.

    reference_recipe=False (tests, benchmark): the synthetic fields of SURVEY.md 8d — seeded, band-limited, solenoidal, mean-free.
    reference_recipe=True: FWD_Solve_KDyn.py:183-317 — B = curl-type field of seed-42 noise smoothed by 101 DEVICE steps of the
    forward solver; U = that same noise field (U_Noise=True) or the analytic flow of :258-260.  The field construction itself
    runs on the host (NumPy): it is not on the hot path."""
    dom = KDynDomain(Npts, X, device=device)
    if U_Noise:
        U = _normalised(_curl_noise(dom, 42), 1.0) if reference_recipe else synthetic_field(dom.G, seeds[1], 1.0)
    else:
        U = _normalised(_analytic_flow(dom), 1.0)
    if reference_recipe:
        raw = _curl_noise(dom, 42).reshape(-1)
        B = _normalised(np.stack(FWD_Solve_IVP_Prep(raw, U, dom, Rm, dt, 100)), M_0)
    else:
        B = synthetic_field(dom.G, seeds[0], M_0)
    return dom, B, U


def Vec_to_Field(domain, X):
    """The reference splits the flat vector into three Dedalus fields on the 3/2 grid (FWD_Solve_KDyn.py:139-171).  Here the flat vector
    IS the layout the device kernels read ([3][G][G][G], z fastest): this returns the three (G,G,G) component arrays (views)."""
    x = np.asarray(X, dtype=np.float64).reshape(-1)
    if x.size != 3 * domain.G ** 3:
        raise ValueError("vector has %d entries, three fields on the grid have %d" % (x.size, 3 * domain.G ** 3))
    a, b, c = x.reshape(3, domain.G, domain.G, domain.G)
    return a, b, c


def Field_to_Vec(domain, Fx, Fy, Fz):
    """Inverse of Vec_to_Field (FWD_Solve_KDyn.py:91-137: gather + allgather): the three grid arrays concatenated."""
    return np.concatenate([np.asarray(f, dtype=np.float64).reshape(-1) for f in (Fx, Fy, Fz)])


def Integrate_Field(domain, F):
    """(1/V) integ F dV of a field on the 3/2 grid = its grid mean (FWD_Solve_KDyn.py:68-89)."""
    return float(np.mean(np.asarray(F, dtype=np.float64)))


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
    on_device = hasattr(X0[0], "numpy")                     # vectors already in HBM (devvec.py DeviceVector / MultiDeviceVector): no staging copies
    J = ctx.forward_any([X0[0], X0[1]])
    for k in ('A_fwd', 'B_fwd', 'C_fwd'):
        X_FWD_DICT[k].ctx = ctx
    if getattr(domain, "write_products", False):           # scalar_data/ and CheckPoints/ like the reference's file handlers
        from . import products
        Xh = [x.numpy() for x in X0] if on_device else X0
        products.write_kdyn(domain, ctx, Xh, dt, N_ITERS, _coeff_to_grid_host)
    return J


def ADJ_Solve_IVP_Lin(X0, domain, Rm, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, Cost_function="Final", Adjoint_type="Discrete"):
    """[dJ/dB0, dJ/dU] as flat grid vectors; valid right after FWD_Solve_IVP_Lin at the same X0."""
    _check_window(N_ITERS, N_SUB_ITERS)
    ctx = domain.context(Rm, dt, N_ITERS, Cost_function)
    return ctx.adjoint_any([X0[0], X0[1]], Adjoint_type)     # device vectors in, device vectors out (fresh ones, like the reference's arrays)


def File_Manips(k):
    """The reference's optimiser callback (FWD_Solve_KDyn.py:1006-1021): keep this iteration's scalar_data / CheckPoints files."""
    from . import products
    products.File_Manips(k)


def Inner_Prod_3(x, y, domain, random_arg=None):
    """Sum over the three components of the grid mean of x*y."""
    return domain.any_context().inner_any(x, y)


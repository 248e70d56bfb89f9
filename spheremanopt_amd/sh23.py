"""1-D periodic Swift-Hohenberg 2-3 — the reference's callbacks, backed by the HIP kernels of csrc/sh23.hip.

Same names / positional signatures as Example_Problems/Periodic_Domain(Fourier)/Swift_Hohenberg/FWD_Solve_SH23.py:

    FWD_Solve_IVP_Lin(X_k, domain, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, filename=None, Adjoint_type="Discrete")   :409
    ADJ_Solve_IVP_Lin(X_k, domain, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, filename=None, Adjoint_type="Discrete")   :598
    Inner_Prod(x, y, domain, rand_arg=None)                                                                        :158
    Generate_IC(E_0, Npts, X)  -> (domain, X_0)                                                                    :174
    GEN_BUFFER(domain, N_SUB_ITERS, Npts)  -> {'A_fwd': handle}                                                    :238

so the reference's driver lines work unchanged:

    args_IP = (domain, None); args_f = [domain, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, None, Adjoint_type]
    Optimise_On_Multi_Sphere([X_0], [E_0], FWD_Solve_IVP_Lin, ADJ_Solve_IVP_Lin, Inner_Prod, args_f, args_IP, ...)

The ``domain`` slot carries a :class:`SH23Domain` (geometry + cache of device contexts) instead of a Dedalus domain;
``X_FWD_DICT['A_fwd']`` is a handle to the HBM-resident snapshot stack instead of a NumPy array.
"""
import numpy as np

from . import _capi

A_PARAM = -0.3       # FWD_Solve_SH23.py:309


class SnapshotStack:
    """Handle to the device-resident coefficient snapshots ('A_fwd' of GEN_BUFFER): ``stack[:, i]`` reads snapshot i."""

    def __init__(self, shape):
        self.shape = shape
        self.ctx = None

    def __getitem__(self, key):
        if self.ctx is None:
            raise RuntimeError("snapshot stack is empty: run FWD_Solve_IVP_Lin first")
        rows, idx = key
        n = self.shape[1]
        idx = idx + n if idx < 0 else idx
        c = self.ctx.snapshot(idx).view(np.complex128)
        return c[rows]


class SH23Domain:
    """Geometry of the periodic box + owner of the device contexts (one per (dt, N_ITERS, batch))."""

    def __init__(self, Npts=256, X=(0., 12. * np.pi), dealias=2, device=0):
        self.Npts, self.interval, self.dealias, self.device = int(Npts), (float(X[0]), float(X[1])), dealias, device
        self.G = int(dealias * Npts)
        self.Nc = (self.Npts - 1) // 2 + 1
        self.hypervolume = self.interval[1] - self.interval[0]
        self._ctx = {}

    def grid(self):
        return self.interval[0] + self.hypervolume * np.arange(self.G) / self.G

    def context(self, dt, N_ITERS, batch=1):
        key = (float(dt), int(N_ITERS), int(batch))
        if key not in self._ctx:
            self._ctx[key] = _capi.Context(_capi.SMO_SH23, self.Npts, self.interval, dt, N_ITERS, A_PARAM, batch=batch,
                                           device=self.device)
        return self._ctx[key]

    def any_context(self):
        if not self._ctx:
            self.context(0.1, 1)
        return next(iter(self._ctx.values()))


def _band_limited_noise(dom, seed, E_0):
    """Seeded standard-normal noise on the scale-2 grid, truncated to the Nc retained modes, modes with index fraction > 1/2
    removed (the reference's ``filter_field``, FWD_Solve_SH23.py:28-53), scaled to <X,X> = E_0."""
    G, Nc = dom.G, dom.Nc
    noise = np.random.RandomState(seed).standard_normal(G)
    c = np.fft.rfft(noise) / G
    keep = np.zeros(G // 2 + 1, dtype=bool)
    keep[:Nc] = np.linspace(0, 1, Nc, endpoint=False) <= 0.5
    c[~keep] = 0
    x = np.fft.irfft(c, n=G) * G
    return x * np.sqrt(E_0 / np.mean(x * x))


def FWD_Solve_IVP_PREP(X_k, domain, dt=1e-02, N_ITERS=100, N_SUB_ITERS=100):
    """Smooth an initial condition by integrating it N_ITERS+1 steps with the forward solver ON THE DEVICE and return the
    final state on the scale-2 grid (FWD_Solve_SH23.py:334-407; the reference's solver stops at iteration N_ITERS+1)."""
    ctx = domain.context(dt, N_ITERS + 1)
    ctx.forward([X_k[0]])
    c = ctx.snapshot(N_ITERS + 1).view(np.complex128)
    pad = np.zeros(domain.G // 2 + 1, dtype=complex)
    pad[:domain.Nc] = c
    pad[0] = pad[0].real
    return np.fft.irfft(pad, n=domain.G) * domain.G


def Generate_IC(E_0=1.0, Npts=256, X=(0., 12. * np.pi), seed=42, device=0, prep=False):
    """Domain + initial condition with <X,X> = E_0 (FWD_Solve_SH23.py:174-236): seeded noise (seed 42 in the reference),
    filtered, normalised; with ``prep=True`` additionally smoothed by 101 device steps at dt = 0.01 and re-normalised, which
    is the reference's full recipe.  ``prep=False`` is the synthetic-input recipe of SURVEY.md 8d used by the tests/bench."""
    dom = SH23Domain(Npts, X, device=device)
    x = _band_limited_noise(dom, seed, E_0)
    if prep:
        x = FWD_Solve_IVP_PREP([x], dom)
        x = x * np.sqrt(E_0 / np.mean(x * x))
    return dom, x


def Vec_to_Field(domain, X):
    """The reference scatters the flat vector into a Dedalus field on the scale-2 grid (FWD_Solve_SH23.py:130-156).  Here the flat
    vector IS the grid layout the device kernels read: this returns it as the (G,) grid array (a view, no copy, no MPI scatter)."""
    x = np.asarray(X, dtype=np.float64).reshape(-1)
    if x.size != domain.G:
        raise ValueError("vector has %d entries, the scale-2 grid has %d" % (x.size, domain.G))
    return x


def Field_to_Vec(domain, F):
    """Inverse of Vec_to_Field (FWD_Solve_SH23.py:89-128: gather + allgather): the grid array is the flat vector."""
    return np.ascontiguousarray(F, dtype=np.float64).reshape(-1)


def Integrate_Field(domain, F):
    """(1/L) integ F dx of a field given on the scale-2 grid = its grid mean (FWD_Solve_SH23.py:66-87)."""
    return float(np.mean(np.asarray(F, dtype=np.float64)))


def GEN_BUFFER(domain, N_SUB_ITERS, Npts=256):
    return {'A_fwd': SnapshotStack((domain.Nc, N_SUB_ITERS + 1))}


def _check_window(N_ITERS, N_SUB_ITERS):
    if N_SUB_ITERS != N_ITERS:
        raise NotImplementedError("windowed checkpointing (N_SUB_ITERS < N_ITERS) is not implemented (nor in the reference)")


def FWD_Solve_IVP_Lin(X_k, domain, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, filename=None, Adjoint_type="Discrete"):
    """-J(X) with J = dt * sum_{n=0}^{N} (1/L) int u_n^2 dx; fills the device snapshot stack."""
    _check_window(N_ITERS, N_SUB_ITERS)
    ctx = domain.context(dt, N_ITERS)
    J = ctx.forward_any([X_k[0]])                          # NumPy vector or devvec.DeviceVector
    X_FWD_DICT['A_fwd'].ctx = ctx
    if getattr(domain, "write_products", False):           # scalar_data/ and CheckPoints/ like the reference's file handlers
        from . import products
        products.write_sh23(domain, ctx, dt, N_ITERS)
    return J


def File_Manips(k):
    """The reference's optimiser callback (FWD_Solve_SH23.py:731-746): keep this iteration's scalar_data / CheckPoints files."""
    from . import products
    products.File_Manips(k)


def ADJ_Solve_IVP_Lin(X_k, domain, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, filename=None, Adjoint_type="Discrete"):
    """[dJ/dX] on the scale-2 grid; valid right after FWD_Solve_IVP_Lin at the same X_k (it replays that stack)."""
    _check_window(N_ITERS, N_SUB_ITERS)
    ctx = domain.context(dt, N_ITERS)
    return ctx.adjoint_any([X_k[0]], Adjoint_type)


def Inner_Prod(x, y, domain, rand_arg=None):
    """(1/L) int x y dx = mean over the scale-2 grid."""
    return domain.any_context().inner_any(x, y)

"""ORACLE (test infrastructure, never shipped) — CPU restatement of the reference's
3-D triply-periodic kinematic-dynamo forward / adjoint path (two-field optimisation B0, U).

PARITY UNPINNED: the reference runs this through Dedalus v2 (absent; no golden vectors in the
reference).  Restates SURVEY.md Appendix A.2, derived from

    FWD_Solve_KDyn.py:395-443   induction equation + div-free constraint, CNAB1, U frozen as NCC fields
    FWD_Solve_KDyn.py:529-689   forward loop (N_ITERS+1 steps, snapshots of A,B,C['c'] before each step)
    FWD_Solve_KDyn.py:696-764   compatibility condition (LBVP)
    FWD_Solve_KDyn.py:766-1004  8-variable adjoint IVP, "undo LHS", gradients w.r.t. B0 and U
    FWD_Solve_KDyn.py:173-181   Inner_Prod_3 = sum over components of the grid mean of x*y

with the Dedalus-v2 conventions of Appendix A.0: coefficients c_k = G^-3 sum f e^{-ik.x}; kx = 0..kmax
(r2c axis first), ky,kz = [0..kmax, -kmax..-1], kmax = (N-1)//2 (Nyquist dropped); products on the 3/2
grid then truncated; CNAB1 on an algebraic row gives L X1 = -L X0 (k.B1 = -k.B0, k=0 mode flips sign).

Layout of the flat vectors (Vec_to_Field / Field_to_Vec, FWD_Solve_KDyn.py:91-171): three components
concatenated, each a C-ordered [x][y][z] array on the dealiased G^3 grid.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from scipy import fft as sfft


class KDynOracle:
    def __init__(self, Npts=24, Rm=1., dt=5e-4, N_ITERS=10, Cost_function="Final", workers=1):
        N = int(Npts)
        self.N = N
        self.G = int(1.5 * N)                            # dealias = 3/2, FWD_Solve_KDyn.py:212
        self.kmax = (N - 1) // 2
        self.a = self.kmax + 1                           # r2c axis
        self.m = 2 * self.kmax + 1                       # c2c axes
        self.Rm, self.dt, self.N_ITERS = float(Rm), float(dt), int(N_ITERS)
        self.cost = Cost_function
        self.workers = workers
        kx = np.arange(self.a, dtype=float)
        kc = np.concatenate([np.arange(0, self.kmax + 1), np.arange(-self.kmax, 0)]).astype(float)
        self.K = np.stack(np.meshgrid(kx, kc, kc, indexing='ij'))          # (3, a, m, m)
        self.k2 = (self.K ** 2).sum(0)
        self.k2s = np.where(self.k2 == 0, 1., self.k2)
        self.zero = (self.k2 == 0)
        D = self.k2 / self.Rm
        self.alpha = 1. / self.dt + 0.5 * D               # a_0 M + b_0 L  diag
        self.beta = 1. / self.dt - 0.5 * D
        self.sel = np.concatenate([np.arange(0, self.kmax + 1), np.arange(self.G - self.kmax, self.G)])
        self.stack = None                                 # (3, a, m, m, N_ITERS+1)

    # -- transforms ------------------------------------------------------------------------------
    def to_coeff(self, g):
        """grid (G,G,G) real -> coeff (a,m,m) complex; x r2c first, then y, z; truncate."""
        w = self.workers
        c = sfft.rfft(g, axis=0, workers=w)[:self.a]
        c = sfft.fft(c, axis=1, workers=w)[:, self.sel]
        c = sfft.fft(c, axis=2, workers=w)[:, :, self.sel]
        return c / float(self.G) ** 3

    def to_grid(self, c):
        """coeff -> grid: zero-pad, z then y (c2c) then x (c2r, imag of kx=0 ignored)."""
        G, w = self.G, self.workers
        p = np.zeros((self.a, self.m, G), dtype=complex)
        p[:, :, self.sel] = c
        p = sfft.ifft(p, axis=2, workers=w)
        q = np.zeros((self.a, G, G), dtype=complex)
        q[:, self.sel] = p
        q = sfft.ifft(q, axis=1, workers=w)
        r = np.zeros((G // 2 + 1, G, G), dtype=complex)
        r[:self.a] = q
        return sfft.irfft(r, n=G, axis=0, workers=w) * float(G) ** 3

    def vec_to_coeff(self, X):
        G = self.G
        return np.stack([self.to_coeff(v.reshape(G, G, G)) for v in np.split(np.asarray(X, dtype=float), 3)])

    def coeff_to_vec(self, C):
        return np.concatenate([self.to_grid(C[i]).ravel() for i in range(3)])

    def grid3(self, C):
        return np.stack([self.to_grid(C[i]) for i in range(3)])

    def coeff3(self, F):
        return np.stack([self.to_coeff(F[i]) for i in range(3)])

    # -- per-mode algebra ------------------------------------------------------------------------
    def kdot(self, V):
        return (self.K * V).sum(0)

    def project(self, V):
        """P(r) = r - k (k.r)/k^2  (k=0 handled by the callers)."""
        return V - self.K * (self.kdot(V) / self.k2s)

    def curl(self, V):
        K = self.K
        return 1j * np.stack([K[1] * V[2] - K[2] * V[1], K[2] * V[0] - K[0] * V[2], K[0] * V[1] - K[1] * V[0]])

    @staticmethod
    def cross(A, B):
        return np.stack([A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]])

    def cnab_update(self, V0, F):
        """(M/dt + L/2) X1 = (M/dt - L/2) X0 + F with the saddle-point constraint k.X1 = -k.X0."""
        V1 = self.project(self.beta * V0 + F) / self.alpha - self.K * (self.kdot(V0) / self.k2s)
        V1[:, self.zero] = -V0[:, self.zero]
        return V1

    # -- callbacks -------------------------------------------------------------------------------
    def inner(self, x, y):
        """Inner_Prod_3 (FWD_Solve_KDyn.py:173-181)."""
        G3 = float(self.G) ** 3
        return float(np.dot(np.asarray(x), np.asarray(y)) / G3)

    def forward(self, X):
        """FWD_Solve_IVP_Lin: X = [B0 vec, U vec]; returns -J; fills the stack with B^_0..B^_N."""
        n_it, dt = self.N_ITERS, self.dt
        Bh = self.vec_to_coeff(X[0])
        self.Ug = self.grid3(self.vec_to_coeff(X[1]))     # NCC fields are truncated by the evaluator (A.0-4)
        self.stack = np.zeros((3, self.a, self.m, self.m, n_it + 1), dtype=complex)
        J = 0.
        for n in range(n_it + 1):
            self.stack[..., n] = Bh
            Bg = self.grid3(Bh)
            e = np.mean((Bg * Bg).sum(0))
            if self.cost == "Integrated":
                J += dt * e                               # FWD_Solve_KDyn.py:668-669
            elif n == n_it:
                J = e                                     # FWD_Solve_KDyn.py:671-673
            Nh = self.curl(self.coeff3(self.cross(self.Ug, Bg)))
            Bh = self.cnab_update(Bh, Nh)
        return -J

    def adjoint(self, X=None, Adjoint_type="Discrete"):
        """ADJ_Solve_IVP_Lin: returns [dJ/dB0, dJ/dU] as flat grid vectors."""
        n_it, dt = self.N_ITERS, self.dt
        S = self.stack
        if Adjoint_type == "Discrete":
            rhs = -2. * S[..., n_it]
            scale = (dt * self.alpha) if self.cost == "Final" else self.alpha   # FWD_Solve_KDyn.py:733-741
            Gh = self.project(rhs) / scale
            Gh[:, self.zero] = 0.
            idx = n_it - 1
        else:
            Gh = -2. * S[..., n_it]                       # FWD_Solve_KDyn.py:906-908
            idx = n_it
        nu = np.zeros_like(Gh)
        for _ in range(n_it):
            Bf_h = S[..., idx]; idx -= 1
            Bf = self.grid3(Bf_h)
            om = self.grid3(self.curl(Gh))
            F1 = self.coeff3(self.cross(om, self.Ug))     # F_x(G; uf,vf,wf), FWD_Solve_KDyn.py:846-859
            if self.cost == "Integrated":
                F1 = F1 - 2. * Bf_h                       # FWD_Solve_KDyn.py:862-864
            F2 = -self.coeff3(self.cross(om, Bf))         # FWD_Solve_KDyn.py:875-877
            nu_new = nu - 2. * self.K * (self.kdot(nu) / self.k2s) + dt * self.project(F2)
            nu_new[:, self.zero] = -nu[:, self.zero]
            Gh = self.cnab_update(Gh, F1)
            nu = nu_new
        if Adjoint_type == "Discrete":
            gB = self.coeff_to_vec(dt * self.alpha * Gh)  # FWD_Solve_KDyn.py:985-989
        else:
            gB = self.coeff_to_vec(Gh)
        return [gB, self.coeff_to_vec(nu)]


def synthetic_field(G, seed, M0=1.0):
    """SURVEY.md section 8d recipe: seeded Gaussian noise on G^3 per component, keep |k_i| <= N/6 on every
    axis, project to div-free, zero the mean, scale to <X,X> = M0.  Returns the flat 3*G^3 vector."""
    N = (2 * G) // 3
    kcut = N // 6
    rs = np.random.RandomState(seed)
    kx = np.fft.rfftfreq(G, 1. / G)
    kc = np.fft.fftfreq(G, 1. / G)
    K = np.stack(np.meshgrid(kc, kc, kx, indexing='ij'))         # transform over (x,y,z) with z as the r2c axis
    keep = (np.abs(K[0]) <= kcut) & (np.abs(K[1]) <= kcut) & (np.abs(K[2]) <= kcut)
    k2 = (K ** 2).sum(0); k2[0, 0, 0] = 1.
    V = np.stack([np.fft.rfftn(rs.standard_normal((G, G, G))) for _ in range(3)]) * keep
    V = V - K * ((K * V).sum(0) / k2)
    V[:, 0, 0, 0] = 0.
    v = np.stack([np.fft.irfftn(V[i], s=(G, G, G), axes=(0, 1, 2)) for i in range(3)])
    v *= np.sqrt(M0 / np.mean((v * v).sum(0)))
    return v.reshape(-1)


class ThreadedKDynOracle(KDynOracle):
    """The same restatement with its POINTWISE stages threaded over kx / x planes — for bench.py's all-core `cpu_baseline` leg only.

    KDynOracle hands `workers` to pocketfft, but its cross products, curls, projections, time-step algebra and the zero-padding copies are
    serial NumPy, so on 256 hardware threads it ran 1.4x faster than on one: not a baseline an "x times the CPU" sentence
    can lean on.  Here every such stage runs chunk by chunk on a thread pool (NumPy ufuncs release the GIL); element for element the same
    operations, so fields and gradients are bit-identical to KDynOracle's — only the grid mean in J is summed per chunk (differs in the last
    bits; tests/test_oracle.py).  The reference itself forces OMP_NUM_THREADS=1 (Sphere_Grad_Descent.py:2) and scales over MPI ranks."""

    def __init__(self, *args, threads=1, **kw):
        from concurrent.futures import ThreadPoolExecutor
        kw["workers"] = int(threads)
        super().__init__(*args, **kw)
        self.nt = int(threads)
        self.pool = ThreadPoolExecutor(self.nt) if self.nt > 1 else None
        # Work arrays are REUSED (named scratch arrays, small rings for the results): a fresh 0.4-1.4 GB array per stage is first touched page by
        # page under the process-wide mmap lock, which neither one thread nor sixteen get through quickly (the 16-thread leg ran 4x the 1-thread
        # one before this).  Ring sizes: no stage keeps more than 4 grid fields / 7 coefficient fields of earlier stages alive.
        self._named, self._rings, self._ring_pos = {}, {}, {}

    def prewarm(self):
        """Allocate and touch every reusable work array (outside a timed region: first-touch page faults are not the algorithm)."""
        G, a, m = self.G, self.a, self.m
        for name, shape, dt in (("t", (a, m, G), complex), ("p", (a, m, G), complex), ("q", (a, G, G), complex), ("r", (G // 2 + 1, G, G), complex),
                                ("Ug", (3, G, G, G), float)):
            self._par(lambda s, b=self._tmp(name, shape, dt): b[s].fill(0), shape[0])
        for kind, n in (("grid", 5), ("coef", 10)):
            for _ in range(n):
                b = self._ring(kind)
                self._par(lambda s, b=b: b[:, s].fill(0), b.shape[1])
        self.stack = np.empty((self.N_ITERS + 1, 3, a, m, m), dtype=complex)
        self._par(lambda s: self.stack[s].fill(0), self.N_ITERS + 1)

    def _tmp(self, name, shape, dtype):
        b = self._named.get(name)
        if b is None or b.shape != tuple(shape) or b.dtype != np.dtype(dtype):
            b = self._named[name] = np.empty(shape, dtype=dtype)
        return b

    def _ring(self, kind):
        G = self.G
        shape, dtype, n = (((3, G, G, G), float, 5) if kind == "grid" else ((3, self.a, self.m, self.m), complex, 10))
        r = self._rings.setdefault(kind, [None] * n)
        i = self._ring_pos.get(kind, 0)
        self._ring_pos[kind] = (i + 1) % n
        if r[i] is None:
            r[i] = np.empty(shape, dtype=dtype)
        return r[i]

    def _par(self, f, n):
        """f(slice) over [0, n) in contiguous chunks, one per thread (a few per thread for balance)."""
        if self.pool is None or n < 2:
            f(slice(0, n))
            return
        parts = min(n, 2 * self.nt)
        edges = [n * i // parts for i in range(parts + 1)]
        list(self.pool.map(f, [slice(edges[i], edges[i + 1]) for i in range(parts) if edges[i + 1] > edges[i]]))

    # -- transforms: pocketfft threads the 1-D passes; the truncation / padding copies are chunked here and results land in caller-owned
    #    arrays (np.stack / np.concatenate of fresh 450-MB arrays were the largest serial item) ----------------------------------------
    def to_coeff(self, g, out=None):
        w, G = self.workers, self.G
        c = sfft.rfft(g, axis=0, workers=w)[:self.a]
        c = sfft.fft(c, axis=1, workers=w, overwrite_x=True)
        t = self._tmp("t", (self.a, self.m, G), complex)
        self._par(lambda s: t.__setitem__(s, c[s][:, self.sel]), self.a)
        t = sfft.fft(t, axis=2, workers=w, overwrite_x=True)
        if out is None:
            out = np.empty((self.a, self.m, self.m), dtype=complex)
        sc = float(G) ** 3
        self._par(lambda s: out.__setitem__(s, t[s][:, :, self.sel] / sc), self.a)
        return out

    def to_grid(self, c, out=None):
        G, w = self.G, self.workers
        p = self._tmp("p", (self.a, self.m, G), complex)

        def pad_z(s):
            p[s] = 0.
            p[s][:, :, self.sel] = c[s]
        self._par(pad_z, self.a)
        p = sfft.ifft(p, axis=2, workers=w, overwrite_x=True)
        q = self._tmp("q", (self.a, G, G), complex)

        def pad_y(s):
            q[s] = 0.
            q[s][:, self.sel] = p[s]
        self._par(pad_y, self.a)
        q = sfft.ifft(q, axis=1, workers=w, overwrite_x=True)
        r = self._tmp("r", (G // 2 + 1, G, G), complex)

        def pad_x(s):
            r[s] = 0.
            hi = min(s.stop, self.a)
            if hi > s.start:
                r[s.start:hi] = q[s.start:hi]
        self._par(pad_x, G // 2 + 1)
        g = sfft.irfft(r, n=G, axis=0, workers=w)
        sc = float(G) ** 3
        if out is None:
            out = g
        self._par(lambda s: np.multiply(g[s], sc, out=out[s]), G)
        return out

    def vec_to_coeff(self, X):
        G = self.G
        X = np.asarray(X, dtype=float).reshape(3, G, G, G)
        out = self._ring("coef")
        for i in range(3):
            self.to_coeff(X[i], out=out[i])
        return out

    def coeff_to_vec(self, C):
        return self.grid3(C).reshape(-1).copy()          # the caller keeps it: not a view of a ring slot

    def grid3(self, C, out=None):
        if out is None:
            out = self._ring("grid")
        for i in range(3):
            self.to_grid(C[i], out=out[i])
        return out

    def coeff3(self, F):
        out = self._ring("coef")
        for i in range(3):
            self.to_coeff(F[i], out=out[i])
        return out

    # -- per-mode algebra, chunked over kx --------------------------------------------------------------------------------------
    def _over_kx(self, f):
        out = self._ring("coef")
        self._par(lambda s: out.__setitem__((slice(None), s), f(s)), self.a)
        return out

    def project(self, V):
        return self._over_kx(lambda s: V[:, s] - self.K[:, s] * ((self.K[:, s] * V[:, s]).sum(0) / self.k2s[s]))

    def curl(self, V):
        out = self._ring("coef")

        def f(s):
            K, v = self.K[:, s], V[:, s]
            out[0, s] = 1j * (K[1] * v[2] - K[2] * v[1])
            out[1, s] = 1j * (K[2] * v[0] - K[0] * v[2])
            out[2, s] = 1j * (K[0] * v[1] - K[1] * v[0])
        self._par(f, self.a)
        return out

    def cross(self, A, B):
        out = self._ring("grid") if A.shape == (3, self.G, self.G, self.G) else np.empty_like(A)

        def f(s):
            a, b = A[:, s], B[:, s]
            out[0, s] = a[1] * b[2] - a[2] * b[1]
            out[1, s] = a[2] * b[0] - a[0] * b[2]
            out[2, s] = a[0] * b[1] - a[1] * b[0]
        self._par(f, A.shape[1])
        return out

    def cnab_update(self, V0, F):
        def f(s):
            K, k2s, v0 = self.K[:, s], self.k2s[s], V0[:, s]
            r = self.beta[s] * v0 + F[:, s]
            v1 = (r - K * ((K * r).sum(0) / k2s)) / self.alpha[s] - K * ((K * v0).sum(0) / k2s)
            z = self.zero[s]
            v1[:, z] = -v0[:, z]
            return v1
        return self._over_kx(f)

    def _energy(self, Bg):
        parts = []
        self._par(lambda s: parts.append((s.start, float(np.einsum("cxyz,cxyz->", Bg[:, s], Bg[:, s])))), Bg.shape[1])
        return sum(v for _, v in sorted(parts)) / float(self.G) ** 3

    def forward(self, X):
        n_it, dt = self.N_ITERS, self.dt
        Bh = self.vec_to_coeff(X[0])
        self.Ug = self.grid3(self.vec_to_coeff(X[1]), out=self._tmp("Ug", (3, self.G, self.G, self.G), float))
        if self.stack is None or self.stack.shape[0] != n_it + 1:
            self.stack = np.zeros((n_it + 1, 3, self.a, self.m, self.m), dtype=complex)      # step index FIRST here: contiguous snapshots
        J = 0.
        for n in range(n_it + 1):
            self.stack[n] = Bh
            Bg = self.grid3(Bh)
            e = self._energy(Bg)
            if self.cost == "Integrated":
                J += dt * e
            elif n == n_it:
                J = e
            Nh = self.curl(self.coeff3(self.cross(self.Ug, Bg)))
            Bh = self.cnab_update(Bh, Nh)
        return -J

    def adjoint(self, X=None, Adjoint_type="Discrete"):
        S = self.stack
        self.stack = np.moveaxis(S, 0, -1)            # a VIEW in the base class's [..., n] indexing: the base sweep below runs unchanged
        try:
            return self._adjoint_threaded(Adjoint_type)
        finally:
            self.stack = S

    def _adjoint_threaded(self, Adjoint_type):
        n_it, dt = self.N_ITERS, self.dt
        S = self.stack
        if Adjoint_type == "Discrete":
            scale = (dt * self.alpha) if self.cost == "Final" else self.alpha
            Gh = self.project(-2. * S[..., n_it]) / scale
            Gh[:, self.zero] = 0.
            idx = n_it - 1
        else:
            Gh = -2. * S[..., n_it]
            idx = n_it
        nu = np.zeros_like(Gh)
        for _ in range(n_it):
            Bf_h = S[..., idx]; idx -= 1
            Bf = self.grid3(Bf_h)
            om = self.grid3(self.curl(Gh))
            F1 = self.coeff3(self.cross(om, self.Ug))
            if self.cost == "Integrated":
                F1 = F1 - 2. * Bf_h
            F2 = self.coeff3(self.cross(om, Bf))
            nu_old = nu

            def f(s, F2=F2, nu_old=nu_old):
                K, k2s, v = self.K[:, s], self.k2s[s], nu_old[:, s]
                f2 = -F2[:, s]
                nn = v - 2. * K * ((K * v).sum(0) / k2s) + dt * (f2 - K * ((K * f2).sum(0) / k2s))
                z = self.zero[s]
                nn[:, z] = -v[:, z]
                return nn
            nu = self._over_kx(f)
            Gh = self.cnab_update(Gh, F1)
        gB = self.coeff_to_vec(dt * self.alpha * Gh) if Adjoint_type == "Discrete" else self.coeff_to_vec(Gh)
        return [gB, self.coeff_to_vec(nu)]

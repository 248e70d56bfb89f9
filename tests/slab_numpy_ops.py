"""TEST INFRASTRUCTURE: a NumPy phase backend for spheremanopt_amd.kdyn_slab.SlabKDyn, used to cover the N>1 driver logic
(exchange layout, phase order, reductions) with world_size-2 gloo runs on the CPU.  It restates, per slab, what the HIP
phases of csrc/kdyn.hip do, reading/writing the exchange buffers in exactly their [chunk][peer][field][3][a/W][m][G/W/K] layout,
and borrows the per-mode algebra from the oracle."""
import numpy as np
from scipy import fft as sfft

from oracle.kdyn import KDynOracle
from spheremanopt_amd.kdyn_slab import (ADJ_A, ADJ_B, ADJ_C, ADJ_INIT, C2G_A, C2G_B, FWD_A, FWD_B, FWD_C, G2C_A, G2C_C, NU_B, NU_C)


class NumpyOps:
    device = "cpu"

    def __init__(self, Npts, Rm, dt, N_ITERS, Cost_function, rank, world, keeps_grid_states=True):
        o = KDynOracle(Npts, Rm=Rm, dt=dt, N_ITERS=N_ITERS, Cost_function=Cost_function)
        self.o, self.rank, self.W = o, rank, world
        self.G, self.a, self.m = o.G, o.a, o.m
        self.al, self.Gzl = o.a // world, o.G // world
        self.ix0 = rank * self.al
        sl = slice(self.ix0, self.ix0 + self.al)
        for name in ("K", "k2", "k2s", "zero", "alpha", "beta"):          # restrict the per-mode tables to this kx slab
            arr = getattr(o, name)
            setattr(o, name, arr[:, sl] if name == "K" else arr[sl])
        self.elems = 3 * self.al * self.m * self.Gzl * world
        self.vec_len = 3 * self.G * self.G * self.Gzl
        self.n_iters, self.dt, self.cost = N_ITERS, dt, Cost_function
        self.stack = np.zeros((N_ITERS + 1, 3, self.al, self.m, self.m), dtype=complex)
        self.keeps_grid_states = keeps_grid_states                        # the device's "Ty stack": B_n on the grid side, per step
        self.K = 1                                                        # chunks of the local z slab (set_chunks)
        self.Bgrid = {}
        self.Gh = self.nu = self.scratch = None
        self.U = None

    # -- buffers: [chunk (spaced for 2 groups)][peer][field group][3][al][m][Gzl / K] on both sides ------------------------------------------------------
    def set_buffers(self, zs, ys):
        self.bz, self.by = zs.numpy().view(np.complex128), ys.numpy().view(np.complex128)

    def set_chunks(self, K):
        assert self.Gzl % K == 0
        self.K = K

    def _chunk(self, buf, nf, k):         # chunk k: [peer][nf][3][al][m][Gzc]; chunks are spaced for two field groups whatever nf is
        n, stride = nf * self.elems // self.K, 2 * self.elems // self.K
        return buf[k * stride:k * stride + n].reshape(self.W, nf, 3, self.al, self.m, self.Gzl // self.K)

    def _put_z(self, Tz, f, nf):          # Tz: (3, al, m, G): my kx, all z -> peer = z block, chunk = part of that block
        T = Tz.reshape(3, self.al, self.m, self.W, self.K, self.Gzl // self.K)
        for k in range(self.K):
            self._chunk(self.bz, nf, k)[:, f] = T[:, :, :, :, k].transpose(3, 0, 1, 2, 4)

    def _get_z(self, f, nf):
        T = np.empty((3, self.al, self.m, self.W, self.K, self.Gzl // self.K), dtype=complex)
        for k in range(self.K):
            T[:, :, :, :, k] = self._chunk(self.bz, nf, k)[:, f].transpose(1, 2, 3, 0, 4)
        return T.reshape(3, self.al, self.m, self.G)

    def _get_y(self, f, nf, k):           # -> (3, a, m, Gzc): all kx (peer = kx block), chunk k of my z
        return self._chunk(self.by, nf, k)[:, f].transpose(1, 0, 2, 3, 4).reshape(3, self.a, self.m, self.Gzl // self.K)

    def _put_y(self, T, f, nf, k):
        self._chunk(self.by, nf, k)[:, f] = T.reshape(3, self.W, self.al, self.m, self.Gzl // self.K).transpose(1, 0, 2, 3, 4)

    def _zs(self, k):                     # z planes of chunk k inside the local slab
        c = self.Gzl // self.K
        return slice(k * c, (k + 1) * c)

    # -- 1-D passes on slabs -----------------------------------------------------------------------------------------------
    def _z_inverse(self, C):              # (3, al, m, m) -> (3, al, m, G)
        o, G = self.o, self.G
        p = np.zeros((3, self.al, self.m, G), dtype=complex); p[..., o.sel] = C
        return sfft.ifft(p, axis=3) * G

    def _z_forward(self, Tz):
        return sfft.fft(Tz, axis=3)[..., self.o.sel] / float(self.G) ** 3

    def _yx_to_grid(self, T):             # (3, a, m, nz) -> real (3, G, G, nz)
        o, G, nz = self.o, self.G, T.shape[-1]
        q = np.zeros((3, self.a, G, nz), dtype=complex); q[:, :, o.sel] = T
        q = sfft.ifft(q, axis=2) * G
        r = np.zeros((3, G // 2 + 1, G, nz), dtype=complex); r[:, :self.a] = q
        return sfft.irfft(r, n=G, axis=1) * G

    def _xy_from_grid(self, g):
        c = sfft.rfft(g, axis=1)[:, :self.a]
        return sfft.fft(c, axis=2)[:, :, self.o.sel]

    # -- phases ------------------------------------------------------------------------------------------------------------
    def phase(self, code, i0=0, vec=None, k=0):
        o = self.o
        zs = self._zs(k)
        grid = (lambda t: t.numpy().reshape(3, self.G, self.G, self.Gzl)) if vec is not None else None
        if code == G2C_A:
            self._put_y(self._xy_from_grid(grid(vec)[..., zs]), 0, 1, k)
        elif code == G2C_C:
            c = self._z_forward(self._get_z(0, 1))
            if i0 == 0:
                self.stack[0] = c
            else:
                self.scratch, self.Gh = c, None        # a new forward solve starts: the device reuses G^'s memory as scratch
        elif code == C2G_A:
            # source 1 is "the G^/scratch array": scratch (U^) during the forward set-up, G^ once the adjoint has started
            src = {0: lambda: self.dt * o.alpha * self.Gh, 1: lambda: self.scratch if self.Gh is None else self.Gh,
                   2: lambda: self.nu}[i0]()
            self._put_z(self._z_inverse(src), 0, 1)
        elif code == C2G_B:
            g = self._yx_to_grid(self._get_y(0, 1, k))
            if vec is None:
                if self.U is None or self.U.shape[-1] != self.Gzl:
                    self.U = np.zeros((3, self.G, self.G, self.Gzl))
                self.U[..., zs] = g
            else:
                grid(vec)[..., zs] = g
        elif code == FWD_A:
            self._put_z(self._z_inverse(self.stack[i0]), 0, 1)
        elif code == FWD_B:
            Bg = self._yx_to_grid(self._get_y(0, 1, k))
            if self.keeps_grid_states:
                self.Bgrid[(i0, k)] = Bg
            self._put_y(self._xy_from_grid(o.cross(self.U[..., zs], Bg)), 0, 1, k)
        elif code == FWD_C:
            E = self._z_forward(self._get_z(0, 1))
            self.stack[i0 + 1] = o.cnab_update(self.stack[i0], o.curl(E))
        elif code == ADJ_INIT:
            BN = self.stack[self.n_iters]
            if i0 == 1:
                self.Gh = -2. * BN
            else:
                scale = (self.dt * o.alpha) if self.cost == "Final" else o.alpha
                self.Gh = o.project(-2. * BN) / scale
                self.Gh[:, o.zero] = 0.
            self.nu = np.zeros_like(self.Gh)
            self.acc = {}                            # running sum of the second product, per chunk, on the grid side
        elif code in (ADJ_A, ADJ_B):
            kept = self.keeps_grid_states and i0 < self.n_iters      # B_f is on the grid side already: omega travels alone
            nf = 1 if kept else 2
            if code == ADJ_A:
                self._put_z(self._z_inverse(o.curl(self.Gh)), 0, nf)
                if not kept:
                    self._put_z(self._z_inverse(self.stack[i0]), 1, 2)
            else:
                om = self._yx_to_grid(self._get_y(0, nf, k))
                Bf = self.Bgrid[(i0, k)] if kept else self._yx_to_grid(self._get_y(1, 2, k))
                F1, F2 = self._xy_from_grid(o.cross(om, self.U[..., zs])), self._xy_from_grid(o.cross(om, Bf))
                self._put_y(F1, 0, 1, k)
                self.acc[k] = self.acc[k] + F2 if k in self.acc else F2
        elif code == ADJ_C:
            F1 = self._z_forward(self._get_z(0, 1))
            if self.cost == "Integrated":
                F1 = F1 - 2. * self.stack[i0]
            self.Gh = o.cnab_update(self.Gh, F1)
        elif code == NU_B:
            self._put_y(self.acc[k], 0, 1, k)
        elif code == NU_C:
            # nu <- R nu - dt P F2_n with nu_N = 0 (the reference's recursion) == -dt P sum_n F2_n: R is the identity on solenoidal fields
            self.nu = -self.dt * o.project(self._z_forward(self._get_z(0, 1)))
            self.nu[:, o.zero] = 0.
        else:
            raise ValueError(code)

    def energy(self, n):
        w = np.where(self.o.K[0] == 0, 1., 2.)
        return float((w * (np.abs(self.stack[n]) ** 2).sum(0)).sum())

    def dot(self, x, y):
        return float(np.dot(x.numpy(), y.numpy()) / float(self.G) ** 3)

    def sync(self):
        pass

    def snapshot(self, n):
        return self.stack[n]

"""Device-resident vectors for the optimiser (SURVEY.md section 7 "Host-side vectors"; include/smo.h smo_vec_*).

``Optimise_On_Multi_Sphere`` only ever combines its vectors with ``+``, ``-``, scalar ``*`` and ``copy.deepcopy`` and hands them to the
three callbacks (Sphere_Grad_Descent.py:284, 296-298, 625-690, 755-756, 813).  A :class:`DeviceVector` supports exactly that
algebra on a buffer in HBM, so a whole optimisation runs without a single vector crossing PCIe: the callbacks of
``spheremanopt_amd.kdyn`` accept DeviceVectors (``smo_forward_dev`` / ``smo_adjoint_dev`` / ``smo_inner_dev``) and return them.

Every operation rounds like NumPy's (``a*x + b*y``: two rounded products, one rounded sum — no fused multiply-add), so the
iterate sequence is bit-identical to the one of the same run on NumPy vectors.
"""
import ctypes as C
import numbers

import numpy as np

from . import _capi


class DeviceVector:
    """float64[n] in the HBM of `device`.  Supports +, -, unary -, scalar * (either side), / scalar, deepcopy / copy, .numpy()."""

    __array_ufunc__ = None          # NumPy scalars on the left (np.float64 * v) defer to __rmul__ instead of trying to broadcast

    def __init__(self, n, device=0, _ptr=None):
        self.n, self.device = int(n), int(device)
        if _ptr is None:
            p = C.c_void_p()
            _capi._check(_capi.lib().smo_vec_alloc(self.device, self.n, C.byref(p)))
            _ptr = p.value
        self.ptr = int(_ptr)

    # -- construction / conversion ---------------------------------------------------------------------------------
    @classmethod
    def from_numpy(cls, x, device=0):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        v = cls(x.size, device)
        _capi._check(_capi.lib().smo_vec_upload(v.device, v.ptr, x.ctypes.data, x.size))
        return v

    def numpy(self):
        out = np.empty(self.n)
        _capi._check(_capi.lib().smo_vec_download(self.device, self.ptr, out.ctypes.data, self.n))
        return out

    @property
    def size(self):
        return self.n

    @property
    def shape(self):
        return (self.n,)

    def __del__(self):
        try:
            if getattr(self, "ptr", 0):
                _capi.lib().smo_vec_free(self.device, self.ptr)
                self.ptr = 0
        except Exception:
            pass

    # -- algebra (each op = one launch of smo_vec_axpby into a fresh pooled buffer) ------------------------------------
    def _axpby(self, a, b, y):
        out = DeviceVector(self.n, self.device)
        _capi._check(_capi.lib().smo_vec_axpby(self.device, self.n, float(a), self.ptr, float(b), y.ptr if y is not None else None, out.ptr))
        return out

    def _other(self, o):
        if not isinstance(o, DeviceVector):
            return None
        if o.n != self.n or o.device != self.device:
            raise ValueError("DeviceVector: size / device mismatch (%d@%d vs %d@%d)" % (self.n, self.device, o.n, o.device))
        return o

    def __add__(self, o):
        o = self._other(o)
        return NotImplemented if o is None else self._axpby(1.0, 1.0, o)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._other(o)
        return NotImplemented if o is None else self._axpby(1.0, -1.0, o)      # x + (-y): -y is exact, the sum rounds like x - y

    def __mul__(self, s):
        if not isinstance(s, numbers.Real):
            return NotImplemented
        return self._axpby(float(s), 0.0, None)

    __rmul__ = __mul__

    def __truediv__(self, s):
        if not isinstance(s, numbers.Real):
            return NotImplemented
        return self.from_numpy(self.numpy() / float(s), self.device)            # not used by the optimiser; exact NumPy semantics

    def __neg__(self):
        return self._axpby(-1.0, 0.0, None)

    def copy(self):
        return self._axpby(1.0, 0.0, None)

    def __deepcopy__(self, memo):
        return self.copy()

    __copy__ = copy

    def __repr__(self):
        return "DeviceVector(n=%d, device=%d)" % (self.n, self.device)


class MultiDeviceVector:
    """A vector of a multi-device context (smo_create_multi) that STAYS distributed: one DeviceVector per device holding that device's z slab
    [3][G][G][G/W] of the reference's flat vector [3][G][G][G].  The same algebra as DeviceVector, slab by slab (the operations the optimiser
    uses are elementwise, so the slab layout does not matter to them); `numpy()` gathers the reference's layout."""

    __array_ufunc__ = None

    def __init__(self, slabs, G):
        self.slabs, self.G = list(slabs), int(G)

    @classmethod
    def from_numpy(cls, x, devices):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        G = int(round((x.size / 3) ** (1. / 3.)))
        W = len(devices)
        if 3 * G ** 3 != x.size or G % W:
            raise ValueError("MultiDeviceVector: %d entries are not three fields on a grid that %d devices divide" % (x.size, W))
        f = x.reshape(3, G, G, G)
        Gz = G // W
        return cls([DeviceVector.from_numpy(np.ascontiguousarray(f[..., i * Gz:(i + 1) * Gz]), d) for i, d in enumerate(devices)], G)

    @classmethod
    def like(cls, v):
        return cls([DeviceVector(s.n, s.device) for s in v.slabs], v.G)

    def numpy(self):
        W = len(self.slabs)
        Gz = self.G // W
        out = np.empty((3, self.G, self.G, self.G))
        for i, s in enumerate(self.slabs):
            out[..., i * Gz:(i + 1) * Gz] = s.numpy().reshape(3, self.G, self.G, Gz)
        return out.reshape(-1)

    @property
    def size(self):
        return sum(s.n for s in self.slabs)

    @property
    def shape(self):
        return (self.size,)

    def _zip(self, o, f):
        if not isinstance(o, MultiDeviceVector):
            return NotImplemented
        if len(o.slabs) != len(self.slabs):
            raise ValueError("MultiDeviceVector: different numbers of slabs")
        return MultiDeviceVector([f(a, b) for a, b in zip(self.slabs, o.slabs)], self.G)

    def _map(self, f):
        return MultiDeviceVector([f(a) for a in self.slabs], self.G)

    def __add__(self, o):
        return self._zip(o, lambda a, b: a + b)

    __radd__ = __add__

    def __sub__(self, o):
        return self._zip(o, lambda a, b: a - b)

    def __mul__(self, s):
        return self._map(lambda a: a * s) if isinstance(s, numbers.Real) else NotImplemented

    __rmul__ = __mul__

    def __truediv__(self, s):
        return self._map(lambda a: a / s) if isinstance(s, numbers.Real) else NotImplemented

    def __neg__(self):
        return self._map(lambda a: -a)

    def copy(self):
        return self._map(lambda a: a.copy())

    def __deepcopy__(self, memo):
        return self.copy()

    __copy__ = copy

    def __repr__(self):
        return "MultiDeviceVector(n=%d, devices=%s)" % (self.size, [s.device for s in self.slabs])


def to_devices(X, devices):
    """List of NumPy vectors -> list of MultiDeviceVectors distributed over `devices` (the X_0 of an optimisation on a multi-device domain)."""
    return [x if isinstance(x, MultiDeviceVector) else MultiDeviceVector.from_numpy(x, devices) for x in X]


def to_device(X, device=0):
    """List of NumPy vectors -> list of DeviceVectors (the optimiser's X_0)."""
    return [x if isinstance(x, DeviceVector) else DeviceVector.from_numpy(x, device) for x in X]


def to_host(X):
    return [x.numpy() if isinstance(x, (DeviceVector, MultiDeviceVector)) else np.asarray(x) for x in X]


def pool_bytes(device=0):
    live, pooled = C.c_size_t(), C.c_size_t()
    _capi._check(_capi.lib().smo_vec_pool_bytes(int(device), C.byref(live), C.byref(pooled)))
    return live.value, pooled.value


def release_pool(device=0):
    _capi._check(_capi.lib().smo_vec_pool_release(int(device)))

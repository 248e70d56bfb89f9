"""ctypes binding of libsmo.so (C-ABI: include/smo.h).

There is no CPU fallback: if the library is missing or no HIP device is usable the calls raise.
Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C spheremanopt_amd/csrc``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMO_LIB", os.path.join(_HERE, "lib", "libsmo.so"))      # SMO_LIB: experimental builds

SMO_SH23, SMO_SHB23, SMO_KDYN, SMO_POIS = 1, 2, 3, 4
COST = {"Final": 0, "Integrated": 1}
ADJOINT = {"Discrete": 0, "Continuous": 1}
ERR_NAMES = {1: "SMO_ERR_ARG", 2: "SMO_ERR_NO_DEVICE", 3: "SMO_ERR_HIP", 4: "SMO_ERR_STATE", 5: "SMO_ERR_NOMEM",
             6: "SMO_ERR_UNSUPPORTED"}

EXPORTS = [
    "smo_create", "smo_create_multi", "smo_destroy", "smo_last_error", "smo_version", "smo_device_count", "smo_ncomp", "smo_vec_len",
    "smo_stack_bytes", "smo_get", "smo_forward", "smo_adjoint", "smo_inner", "smo_forward_dev", "smo_adjoint_dev", "smo_inner_dev", "smo_inner_slabs",
    "smo_snapshot_len", "smo_snapshot_read", "smo_transform", "smo_kdyn_op", "smo_set_stream", "smo_timing_enable", "smo_timing_classes", "smo_timing_get", "smo_timing_hbm_bytes",
    "smo_vec_alloc", "smo_vec_free", "smo_vec_pool_release", "smo_vec_pool_bytes", "smo_vec_upload", "smo_vec_download", "smo_vec_axpby",
    "smo_host_alloc", "smo_host_free",
    "smo_comm_unique_id", "smo_comm_init", "smo_comm_set_transport", "smo_comm_get", "smo_comm_library", "smo_timing_select", "smo_timing_stride",
]

ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)      # smo_alltoall_fn
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)                       # smo_allreduce_fn


class SmoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, "error %d" % code), msg))
        self.code = code


class smo_config(C.Structure):
    _fields_ = [("kind", C.c_int), ("npts", C.c_int), ("x0", C.c_double), ("x1", C.c_double), ("dt", C.c_double),
                ("n_iters", C.c_int), ("param", C.c_double), ("cost", C.c_int), ("batch", C.c_int), ("device", C.c_int),
                ("rank", C.c_int), ("world", C.c_int), ("ckpt", C.c_int),
                ("npts2", C.c_int), ("param2", C.c_double), ("param3", C.c_double), ("param4", C.c_double)]


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process do not work (the second one
    finds no GPU), so if torch is installed its runtime is loaded first and libsmo's DT_NEEDED libamdhip64 resolves to it;
    this keeps `import torch` (bench.py, the multi-GPU driver) usable in either import order.  torch itself is not imported."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libsmo.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsmo.so not found at %s — the HIP extension is not built (run __graft_entry__.build()); "
                           "spheremanopt_amd has no CPU fallback" % LIB_PATH)
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    pp = C.POINTER(C.c_void_p)
    L.smo_last_error.restype = C.c_char_p
    L.smo_version.restype = C.c_char_p
    L.smo_device_count.argtypes = [ip]
    L.smo_create.argtypes = [C.POINTER(smo_config), pp]
    L.smo_create_multi.argtypes = [C.POINTER(smo_config), C.c_int, ip, pp]
    L.smo_destroy.argtypes = [vp]
    L.smo_destroy.restype = None
    L.smo_ncomp.argtypes = [vp]
    L.smo_vec_len.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.smo_stack_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.smo_snapshot_len.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.smo_get.argtypes = [vp, C.c_int, dp]
    for name in ("smo_forward", "smo_forward_dev"):
        getattr(L, name).argtypes = [vp, pp, dp]
    for name in ("smo_adjoint", "smo_adjoint_dev"):
        getattr(L, name).argtypes = [vp, pp, C.c_int, pp]
    for name in ("smo_inner", "smo_inner_dev"):
        getattr(L, name).argtypes = [vp, vp, vp, dp]
    L.smo_inner_slabs.argtypes = [vp, pp, pp, dp]
    L.smo_snapshot_read.argtypes = [vp, C.c_int, C.c_int, dp]
    L.smo_transform.argtypes = [vp, C.c_int, vp, vp]
    L.smo_kdyn_op.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, dp]
    L.smo_set_stream.argtypes = [vp, vp]
    L.smo_timing_enable.argtypes = [vp, C.c_int]
    L.smo_timing_classes.argtypes = [vp]
    L.smo_timing_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), dp, dp]
    L.smo_timing_hbm_bytes.argtypes = [vp, C.c_int, dp]
    L.smo_vec_alloc.argtypes = [C.c_int, C.c_size_t, pp]
    L.smo_vec_free.argtypes = [C.c_int, vp]
    L.smo_vec_pool_release.argtypes = [C.c_int]
    L.smo_vec_pool_bytes.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.smo_vec_upload.argtypes = [C.c_int, vp, vp, C.c_size_t]
    L.smo_vec_download.argtypes = [C.c_int, vp, vp, C.c_size_t]
    L.smo_vec_axpby.argtypes = [C.c_int, C.c_size_t, C.c_double, vp, C.c_double, vp, vp]
    L.smo_host_alloc.argtypes = [C.c_size_t, pp]
    L.smo_host_free.argtypes = [vp]
    L.smo_comm_unique_id.argtypes = [vp]
    L.smo_comm_init.argtypes = [vp, vp]
    L.smo_comm_set_transport.argtypes = [vp, ALLTOALL_FN, ALLREDUCE_FN, vp]
    L.smo_comm_get.argtypes = [vp, C.c_int, dp]
    L.smo_comm_library.restype = C.c_char_p
    L.smo_timing_select.argtypes = [vp, C.c_ulonglong]
    L.smo_timing_stride.argtypes = [vp, C.c_int]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise SmoError(rc, lib().smo_last_error().decode())


def _preload_torch_rccl():
    """Same reason as for the HIP runtime: if PyTorch's bundled librccl is going to live in this process, libsmo must use that copy
    (smo_comm_init picks up an already loaded librccl.so.1 before looking for the system one)."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def comm_unique_id():
    """128 opaque bytes (an RCCL unique id) made by ONE rank and handed to every rank's Context.comm_init."""
    _preload_torch_rccl()
    buf = C.create_string_buffer(128)
    _check(lib().smo_comm_unique_id(buf))
    return buf.raw


def device_count():
    n = C.c_int(0)
    lib().smo_device_count(C.byref(n))
    return n.value


def _ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def _dev_ptr(t):
    """Device address of a DeviceVector (.ptr), a torch tensor (.data_ptr()) or a plain int."""
    if isinstance(t, int):
        return t
    p = getattr(t, "ptr", None)
    return int(p) if p is not None else int(t.data_ptr())


class _PinnedBlock:
    """Owner of one hipHostMalloc block; the NumPy arrays made from it keep it alive through their .base chain."""

    def __init__(self, nbytes):
        self.p = C.c_void_p()
        _check(lib().smo_host_alloc(int(nbytes), C.byref(self.p)))
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.p.value:
                lib().smo_host_free(self.p)
                self.p = C.c_void_p()
        except Exception:
            pass


def pinned_empty(n):
    """float64[n] in page-locked host memory (smo_host_alloc): the host-buffer entry points then copy at PCIe rate."""
    blk = _PinnedBlock(8 * int(n))
    raw = (C.c_double * int(n)).from_address(blk.p.value)
    raw._smo_block = blk                       # the ctypes array is the ndarray's base and carries the block
    return np.frombuffer(raw, dtype=np.float64, count=int(n))


def pinned_copy(x):
    out = pinned_empty(np.asarray(x).size)
    out[:] = np.asarray(x, dtype=np.float64).reshape(-1)
    return out


class Context:
    """Owner of one smo_ctx (device buffers, twiddles, the HBM snapshot stack)."""
    devices = None          # MultiContext: the device list

    def __init__(self, kind, npts, interval, dt, n_iters, param, cost="Final", batch=1, device=0, rank=0, world=1, ckpt=1,
                 npts2=0, param2=0., param3=0., param4=0.):
        cfg = smo_config(kind, int(npts), float(interval[0]), float(interval[1]), float(dt), int(n_iters), float(param),
                         COST[cost] if isinstance(cost, str) else int(cost), int(batch), int(device), int(rank), int(world),
                         int(ckpt), int(npts2), float(param2), float(param3), float(param4))
        self.cfg = cfg
        self._h = C.c_void_p()
        _check(lib().smo_create(C.byref(cfg), C.byref(self._h)))
        self._describe()
        self.batch = int(batch)

    def _describe(self):
        self.ncomp = lib().smo_ncomp(self._h)
        n = C.c_size_t()
        _check(lib().smo_vec_len(self._h, C.byref(n)))
        self.vec_len = n.value
        _check(lib().smo_stack_bytes(self._h, C.byref(n)))
        self.stack_bytes = n.value
        _check(lib().smo_snapshot_len(self._h, C.byref(n)))
        self.snapshot_len = n.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().smo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-buffer entry points --------------------------------------------------------------------------------
    def _host_vecs(self, X):
        vs = [np.ascontiguousarray(np.asarray(X[c], dtype=np.float64)).reshape(-1) for c in range(self.ncomp)]
        for v in vs:
            if v.size != self.vec_len * self.batch:
                raise ValueError("vector has %d entries, context expects %d" % (v.size, self.vec_len * self.batch))
        return vs

    def forward(self, X):
        vs = self._host_vecs(X)
        J = np.zeros(self.batch)
        _check(lib().smo_forward(self._h, _ptr_array([v.ctypes.data for v in vs]), J.ctypes.data_as(C.POINTER(C.c_double))))
        return float(J[0]) if self.batch == 1 else J

    def adjoint(self, X=None, adjoint_type="Discrete", out=None):
        """`out`: caller-owned result arrays (e.g. pinned ones), else fresh NumPy arrays like the reference's."""
        grads = out if out is not None else [np.empty(self.vec_len * self.batch) for _ in range(self.ncomp)]
        for g in grads:
            if g.size != self.vec_len * self.batch or g.dtype != np.float64 or not g.flags.c_contiguous:
                raise ValueError("adjoint(out=): contiguous float64 arrays of %d entries expected" % (self.vec_len * self.batch))
        if X is None:
            xp = _ptr_array([None] * self.ncomp)
        else:
            vs = self._host_vecs(X)
            xp = _ptr_array([v.ctypes.data for v in vs])
        _check(lib().smo_adjoint(self._h, xp, ADJOINT[adjoint_type], _ptr_array([g.ctypes.data for g in grads])))
        return grads

    def inner(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        if x.size != self.vec_len * self.batch or y.size != x.size:
            raise ValueError("inner: vectors have %d/%d entries, context expects %d" % (x.size, y.size, self.vec_len * self.batch))
        out = np.zeros(self.batch)
        _check(lib().smo_inner(self._h, x.ctypes.data, y.ctypes.data, out.ctypes.data_as(C.POINTER(C.c_double))))
        return float(out[0]) if self.batch == 1 else out

    # -- device-resident entry points (torch tensors on the context's device, or raw addresses) ------------------------
    def _check_dev(self, vecs, who):
        """A device vector of the wrong length or on another GPU would be an out-of-bounds device access, not an exception: refuse it
        here.  (Raw integer addresses cannot be checked — the caller vouches for them, as across the C-ABI itself.)"""
        if len(vecs) != self.ncomp and who != "inner":
            raise ValueError("%s: %d vectors given, the context has %d components" % (who, len(vecs), self.ncomp))
        want = self.vec_len * self.batch
        for v in vecs:
            if isinstance(v, int):
                continue
            n = getattr(v, "n", None)
            if n is None and hasattr(v, "numel"):
                n = v.numel()
                if getattr(v, "dtype", None) is not None and getattr(v.dtype, "itemsize", 8) != 8:
                    raise ValueError("%s: float64 vectors expected, got %s" % (who, v.dtype))
                if hasattr(v, "is_contiguous") and not v.is_contiguous():
                    raise ValueError("%s: contiguous vectors expected" % who)
            if n is not None and int(n) != want:
                raise ValueError("%s: device vector has %d entries, context expects %d" % (who, int(n), want))
            dev = getattr(v, "device", None)
            idx = dev if isinstance(dev, int) else getattr(dev, "index", None)
            if getattr(dev, "type", "cuda") != "cuda":
                raise ValueError("%s: vector lives on %s, not on a GPU" % (who, dev))
            if idx is not None and int(idx) != int(self.cfg.device):
                raise ValueError("%s: vector lives on device %d, the context on device %d" % (who, int(idx), int(self.cfg.device)))

    def forward_dev(self, X):
        self._check_dev(X, "forward_dev")
        J = np.zeros(self.batch)
        _check(lib().smo_forward_dev(self._h, _ptr_array([_dev_ptr(x) for x in X]), J.ctypes.data_as(C.POINTER(C.c_double))))
        return float(J[0]) if self.batch == 1 else J

    def adjoint_dev(self, X, grads, adjoint_type="Discrete"):
        if X is not None:
            self._check_dev(X, "adjoint_dev")
        self._check_dev(grads, "adjoint_dev(grads)")
        xp = _ptr_array([_dev_ptr(x) for x in X]) if X is not None else _ptr_array([None] * len(grads))     # the adjoint replays the stack: X is not read
        _check(lib().smo_adjoint_dev(self._h, xp, ADJOINT[adjoint_type], _ptr_array([_dev_ptr(g) for g in grads])))
        return grads

    def inner_dev(self, x, y):
        self._check_dev([x, y], "inner")
        out = np.zeros(self.batch)
        _check(lib().smo_inner_dev(self._h, _dev_ptr(x), _dev_ptr(y), out.ctypes.data_as(C.POINTER(C.c_double))))
        return float(out[0]) if self.batch == 1 else out

    # -- either kind of vector: NumPy arrays (staged over PCIe) or devvec.DeviceVector (already in HBM) ---------------------------------
    @staticmethod
    def _on_device(v):
        return hasattr(v, "ptr") and hasattr(v, "numpy")

    def forward_any(self, X):
        return self.forward_dev(X) if self._on_device(X[0]) else self.forward(X)

    def adjoint_any(self, X, adjoint_type="Discrete"):
        """Device vectors in -> fresh device vectors out (like the reference's fresh arrays); anything else: NumPy arrays out."""
        if X is not None and len(X) and self._on_device(X[0]):
            from .devvec import DeviceVector
            grads = [DeviceVector(self.vec_len * self.batch, self.cfg.device) for _ in range(self.ncomp)]
            self.adjoint_dev(list(X), grads, adjoint_type)
            return grads
        return self.adjoint(None, adjoint_type)

    def inner_any(self, x, y):
        dx, dy = self._on_device(x), self._on_device(y)
        if dx and dy:
            return self.inner_dev(x, y)
        if dx or dy:
            raise TypeError("inner product: one operand is a DeviceVector and the other is not")
        return self.inner(x, y)

    # -- slab communicator (KDYN, world > 1): the transposes then happen inside smo_forward / smo_adjoint ----------------
    def comm_init(self, unique_id):
        """Collective: RCCL communicator over the context's `world` ranks from the 128 bytes of comm_unique_id()."""
        _preload_torch_rccl()
        if len(unique_id) != 128:
            raise ValueError("comm_init: the unique id has 128 bytes")
        _check(lib().smo_comm_init(self._h, C.create_string_buffer(bytes(unique_id), 128)))

    def comm_set_transport(self, all_to_all, all_reduce_sum):
        """Collective: caller-provided transport instead of RCCL.  all_to_all(src_dev, dst_dev, bytes_per_peer, hip_stream) and
        all_reduce_sum(ctypes double pointer, n) are Python callables (exceptions are reported as a failed exchange)."""
        def a2a(user, src, dst, nbytes, stream):
            try:
                all_to_all(src, dst, nbytes, stream)
                return 0
            except Exception as e:               # noqa: BLE001 — an exception must not unwind through the C frames
                self._transport_error = e
                return 1

        def ared(user, vals, n):
            try:
                all_reduce_sum(vals, n)
                return 0
            except Exception as e:               # noqa: BLE001
                self._transport_error = e
                return 1

        self._transport = (ALLTOALL_FN(a2a), ALLREDUCE_FN(ared))          # keep the thunks alive as long as the context
        self._transport_error = None
        _check(lib().smo_comm_set_transport(self._h, self._transport[0], self._transport[1], None))

    def comm_get(self, key):
        v = C.c_double()
        _check(lib().smo_comm_get(self._h, int(key), C.byref(v)))
        return v.value

    # -- introspection ------------------------------------------------------------------------------------------------
    def snapshot(self, index, b=0):
        out = np.empty(self.snapshot_len)
        _check(lib().smo_snapshot_read(self._h, int(b), int(index), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def transform(self, which, x, out_len=None):
        """smo_transform; `out_len` = doubles of the result when it differs from the input (POIS: grid <-> complex coefficients)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x) if out_len is None else np.empty(int(out_len))
        _check(lib().smo_transform(self._h, int(which), x.ctypes.data, out.ctypes.data))
        return out

    def get(self, key):
        v = C.c_double()
        _check(lib().smo_get(self._h, int(key), C.byref(v)))
        return v.value

    def timing_enable(self, on=True, only=None, select=None, every=1):
        """on=False: off; on=True: every kernel class; only=k: just class k (index into timing()); select=[k, ...]: exactly those classes;
        every=n: of those, every n-th launch only (a uniform sample at 1/n of the event overhead)."""
        _check(lib().smo_timing_stride(self._h, int(every)))
        if select is not None:
            mask = 0
            for k in select:
                mask |= 1 << int(k)
            _check(lib().smo_timing_select(self._h, mask))
            return
        _check(lib().smo_timing_enable(self._h, (2 + int(only)) if only is not None else (1 if on else 0)))

    def timing(self):
        res = []
        for k in range(lib().smo_timing_classes(self._h)):
            name = C.c_char_p()
            n = C.c_longlong()
            ms = C.c_double()
            by = C.c_double()
            _check(lib().smo_timing_get(self._h, k, C.byref(name), C.byref(n), C.byref(ms), C.byref(by)))
            hb = C.c_double()
            _check(lib().smo_timing_hbm_bytes(self._h, k, C.byref(hb)))
            res.append({"kernel": name.value.decode(), "launches": n.value, "total_ms": ms.value, "bytes_per_launch": by.value,
                        "hbm_bytes_per_launch": hb.value})
        return res


class MultiContext(Context):
    """smo_create_multi: ONE process, ONE context, a KDYN problem slab-decomposed over `devices` (include/smo.h).  forward / adjoint / inner take
    and return the reference's FULL flat vectors like the single-GPU Context; forward_dev / adjoint_dev take one slab pointer per (component,
    device), component-major."""

    def __init__(self, npts, interval, dt, n_iters, Rm, devices, cost="Final", ckpt=1):
        cfg = smo_config(SMO_KDYN, int(npts), float(interval[0]), float(interval[1]), float(dt), int(n_iters), float(Rm),
                         COST[cost] if isinstance(cost, str) else int(cost), 1, int(devices[0]), 0, 1, int(ckpt), 0, 0., 0., 0.)
        self.cfg = cfg
        self.devices = [int(d) for d in devices]
        self._h = C.c_void_p()
        ids = (C.c_int * len(self.devices))(*self.devices)
        _check(lib().smo_create_multi(C.byref(cfg), len(self.devices), ids, C.byref(self._h)))
        self._describe()
        self.batch = 1

    def _check_dev(self, vecs, who):
        """Per-device slab pointers, component-major: ncomp * ndev of them.  A slab of the wrong length or on the wrong device would be a GPU
        memory fault inside a worker thread, not an exception: every slab that describes itself (DeviceVector, torch tensor) is
        checked; raw integer addresses are the caller's business, as across the C-ABI."""
        nd = len(self.devices)
        want = (2 if who == "inner" else self.ncomp) * nd
        if len(vecs) != want:
            raise ValueError("%s: %d slab pointers given, %d components x %d devices expected" % (who, len(vecs), self.ncomp, nd))
        n_slab = self.vec_len // nd
        for k, v in enumerate(vecs):
            if v is None:
                raise ValueError("%s: slab pointer %d is None" % (who, k))
            if isinstance(v, int):
                continue
            n = getattr(v, "n", None)
            if n is None and hasattr(v, "numel"):
                n = v.numel()
                if getattr(v, "dtype", None) is not None and getattr(v.dtype, "itemsize", 8) != 8:
                    raise ValueError("%s: float64 slabs expected, got %s" % (who, v.dtype))
                if hasattr(v, "is_contiguous") and not v.is_contiguous():
                    raise ValueError("%s: contiguous slabs expected" % who)
            if n is not None and int(n) != n_slab:
                raise ValueError("%s: slab %d has %d entries, the context expects %d (= vec_len / %d devices)" % (who, k, int(n), n_slab, nd))
            dev = getattr(v, "device", None)
            idx = dev if isinstance(dev, int) else getattr(dev, "index", None)
            if idx is not None and int(idx) != self.devices[k % nd]:
                raise ValueError("%s: slab %d lives on device %d, rank %d of the context on device %d" % (who, k, int(idx), k % nd, self.devices[k % nd]))

    # -- vectors that stay distributed over the devices (devvec.MultiDeviceVector): no PCIe traffic, no host algebra on full-size vectors --------
    @staticmethod
    def _on_device(v):
        return hasattr(v, "slabs")

    def _slab_ptrs(self, X):
        for x in X:
            if [s.device for s in x.slabs] != self.devices or sum(s.n for s in x.slabs) != self.vec_len:
                raise ValueError("MultiDeviceVector does not match the context (devices %s, %d entries)" % (self.devices, self.vec_len))
        return [s for x in X for s in x.slabs]                 # component-major: X[c * ndev + i]

    def forward_any(self, X):
        return self.forward_dev(self._slab_ptrs(X)) if self._on_device(X[0]) else self.forward(X)

    def adjoint_any(self, X, adjoint_type="Discrete"):
        if X is not None and len(X) and self._on_device(X[0]):
            from .devvec import MultiDeviceVector
            grads = [MultiDeviceVector.like(X[0]) for _ in range(self.ncomp)]
            self.adjoint_dev(None, self._slab_ptrs(grads), adjoint_type)
            return grads
        return self.adjoint(None, adjoint_type)

    def inner_any(self, x, y):
        dx, dy = self._on_device(x), self._on_device(y)
        if dx != dy:
            raise TypeError("inner product: one operand is a MultiDeviceVector and the other is not")
        if not dx:
            return self.inner(x, y)
        self._slab_ptrs([x, y])
        out = np.zeros(1)
        _check(lib().smo_inner_slabs(self._h, _ptr_array([_dev_ptr(s) for s in x.slabs]), _ptr_array([_dev_ptr(s) for s in y.slabs]),
                                     out.ctypes.data_as(C.POINTER(C.c_double))))
        return float(out[0])

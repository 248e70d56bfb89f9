"""Device-resident vectors (include/smo.h smo_vec_*, spheremanopt_amd/devvec.py): the optimiser's vector algebra in HBM.

The bar is bit-exactness with NumPy: an optimisation driven with DeviceVectors must produce the RESIDUAL / FUNCT sequence of the same
run on NumPy vectors bit for bit (the reference's algebra: Sphere_Grad_Descent.py:625-690, 734-813)."""
import copy

import numpy as np
import pytest

from spheremanopt_amd import _capi, kdyn
from spheremanopt_amd.devvec import DeviceVector, pool_bytes, release_pool, to_device, to_host
from spheremanopt_amd.sphere_opt import Optimise_On_Multi_Sphere, Update_vector, tangent_vector, transport_vector
from spheremanopt_amd.test_grad import taylor_table

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 7, 4096, 100003])
def test_algebra_rounds_like_numpy(n):
    rs = np.random.RandomState(n)
    x, y = rs.standard_normal(n) * 10. ** rs.randint(-8, 8, n), rs.standard_normal(n)
    X, Y = DeviceVector.from_numpy(x), DeviceVector.from_numpy(y)
    a, b = 0.7310585786300049, -3.3e-7
    assert np.array_equal((X + Y).numpy(), x + y)
    assert np.array_equal((X - Y).numpy(), x - y)
    assert np.array_equal((a * X).numpy(), a * x) and np.array_equal((X * np.float64(a)).numpy(), x * a)
    assert np.array_equal((X + a * Y).numpy(), x + a * y)                    # two roundings, no fma
    assert np.array_equal((-1. * X + b * Y).numpy(), -1. * x + b * y)
    assert np.array_equal((-X).numpy(), -x)
    assert np.array_equal((np.float64(b) * X).numpy(), b * x)                # NumPy scalar on the left defers to __rmul__
    Z = copy.deepcopy(X)
    assert Z.ptr != X.ptr and np.array_equal(Z.numpy(), x)
    L = copy.deepcopy([X, Y])
    assert L[0].ptr != X.ptr and np.array_equal(L[1].numpy(), y)
    with pytest.raises(ValueError):
        X + DeviceVector.from_numpy(np.zeros(n + 1))


def test_axpby_on_views_that_are_only_8_byte_aligned():
    """smo_vec_axpby on a view at an odd element offset of a caller's buffer: same rounding through the one-element-per-lane
    kernel; a pointer that is not even 8-byte aligned is refused; and the device entry points refuse vectors of the wrong length / device."""
    import ctypes as C
    n = 1001
    rs = np.random.RandomState(5)
    x, y = rs.standard_normal(n + 1), rs.standard_normal(n + 1)
    X, Y, O = DeviceVector.from_numpy(x), DeviceVector.from_numpy(y), DeviceVector(n + 1)
    a, b = 1.25, -0.3333333333333333
    L = _capi.lib()
    _capi._check(L.smo_vec_axpby(0, n, a, C.c_void_p(X.ptr + 8), b, C.c_void_p(Y.ptr + 8), C.c_void_p(O.ptr + 8)))
    assert np.array_equal(O.numpy()[1:], a * x[1:] + b * y[1:])
    _capi._check(L.smo_vec_axpby(0, n, a, C.c_void_p(X.ptr + 8), b, None, C.c_void_p(O.ptr)))          # mixed alignment, no y
    assert np.array_equal(O.numpy()[:n], a * x[1:])
    assert L.smo_vec_axpby(0, n, a, C.c_void_p(X.ptr + 4), b, None, C.c_void_p(O.ptr)) == 1 and b"8-byte aligned" in L.smo_last_error()
    dom = kdyn.KDynDomain(8)
    ctx = dom.context(1.0, 1e-3, 2)
    good = [DeviceVector(ctx.vec_len), DeviceVector(ctx.vec_len)]
    with pytest.raises(ValueError):
        ctx.forward_dev([DeviceVector(ctx.vec_len - 1), good[1]])
    with pytest.raises(ValueError):
        ctx.forward_dev([good[0]])
    with pytest.raises(ValueError):
        ctx.inner_dev(good[0], DeviceVector(5))
    dom.drop_contexts()


def test_pool_recycles_and_releases():
    release_pool(0)
    live0, _ = pool_bytes(0)
    v = [DeviceVector.from_numpy(np.ones(1000)) for _ in range(4)]
    ptrs = {w.ptr for w in v}
    live1, _ = pool_bytes(0)
    assert live1 - live0 == 4 * 8192                                          # 8000 bytes rounded up to 256
    del v
    live2, pooled = pool_bytes(0)
    assert live2 == live0 and pooled >= 4 * 8192
    again = DeviceVector(1000)
    assert again.ptr in ptrs                                                  # came from the pool, not from hipMalloc
    del again
    release_pool(0)
    assert pool_bytes(0)[1] == 0
    with pytest.raises(_capi.SmoError):
        _capi._check(_capi.lib().smo_vec_free(0, 12345))


def test_sphere_geometry_matches_numpy_bitwise():
    N = 8
    dom = kdyn.KDynDomain(N)
    G = dom.G
    x, g, d = (kdyn.synthetic_field(G, s) for s in (1, 2, 3))
    X, Gv, D = (DeviceVector.from_numpy(v) for v in (x, g, d))
    ip = (dom, None)
    assert kdyn.Inner_Prod_3(X, Gv, dom) == kdyn.Inner_Prod_3(x, g, dom)
    assert np.array_equal(tangent_vector(X, Gv, kdyn.Inner_Prod_3, ip).numpy(), tangent_vector(x, g, kdyn.Inner_Prod_3, ip))
    assert np.array_equal(transport_vector(X, D, kdyn.Inner_Prod_3, ip).numpy(), transport_vector(x, d, kdyn.Inner_Prod_3, ip))
    assert np.array_equal(Update_vector(X, 0.37, D, 2.0, kdyn.Inner_Prod_3, ip).numpy(), Update_vector(x, 0.37, d, 2.0, kdyn.Inner_Prod_3, ip))
    with pytest.raises(TypeError):
        kdyn.Inner_Prod_3(X, g, dom)
    dom.drop_contexts()


@pytest.mark.parametrize("LS,CG", [("LS_wolfe", True), ("LS_armijo", False)])
def test_optimiser_iterates_are_bit_identical_on_device_vectors(in_tmp_cwd, LS, CG):
    """The whole drop-in loop: same callbacks, once with NumPy vectors (staged over PCIe per call) and once with DeviceVectors."""
    N, n, dt = 16, 8, 1e-2
    runs = {}
    for mode in ("numpy", "device"):
        dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
        buf = kdyn.GEN_BUFFER(N, dom, n)
        args_f = [dom, 1., dt, n, n, buf, "Final", "Discrete"]
        X0 = [B, U] if mode == "numpy" else to_device([B, U], dom.device)
        R, F, X = Optimise_On_Multi_Sphere(X0, [1., 1.], kdyn.FWD_Solve_IVP_Lin, kdyn.ADJ_Solve_IVP_Lin, kdyn.Inner_Prod_3, args_f,
                                           (dom, None), max_iters=4, alpha_k=10., LS=LS, CG=CG, verbose=False)
        runs[mode] = (R, F, to_host(X))
        dom.drop_contexts()
    (R0, F0, X0), (R1, F1, X1) = runs["numpy"], runs["device"]
    assert len(F0) == 4 and F0 == F1
    assert R0 == R1
    assert np.array_equal(X0[0], X1[0]) and np.array_equal(X0[1], X1[1])
    prog = np.load("DAL_PROGRESS.npz") if not _has_h5py() else None
    if prog is not None:
        assert np.array_equal(prog["X_opt"][0], X1[0])                       # the restart file holds host arrays


def _has_h5py():
    try:
        import h5py  # noqa: F401
        return True
    except Exception:
        return False


def test_taylor_test_on_device_vectors():
    N, n, dt = 16, 10, 1e-2
    dom = kdyn.KDynDomain(N)
    B, U = kdyn.synthetic_field(dom.G, 1), kdyn.synthetic_field(dom.G, 2)
    dB, dU = kdyn.synthetic_field(dom.G, 3), kdyn.synthetic_field(dom.G, 4)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args_f = [dom, 1., dt, n, n, buf, "Final", "Discrete"]
    AA_h = taylor_table([B, U], [dB, dU], kdyn.FWD_Solve_IVP_Lin, kdyn.ADJ_Solve_IVP_Lin, kdyn.Inner_Prod_3, args_f, (dom, None), epsilon=1e-3)
    AA_d = taylor_table(to_device([B, U]), to_device([dB, dU]), kdyn.FWD_Solve_IVP_Lin, kdyn.ADJ_Solve_IVP_Lin, kdyn.Inner_Prod_3, args_f,
                        (dom, None), epsilon=1e-3)
    assert np.array_equal(AA_h, AA_d)
    assert np.all(np.abs(AA_d[4, :4] - 2.0) < 5e-3)
    dom.drop_contexts()


def test_pinned_host_buffers_round_trip():
    N, n = 16, 3
    dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
    ctx = dom.context(1., 1e-3, n, "Final")
    J0 = ctx.forward([B, U]); g0 = ctx.adjoint(None)
    hX = [_capi.pinned_copy(B), _capi.pinned_copy(U)]
    hG = [_capi.pinned_empty(B.size), _capi.pinned_empty(U.size)]
    J1 = ctx.forward(hX); g1 = ctx.adjoint(None, out=hG)
    assert J1 == J0 and g1[0] is hG[0] and np.array_equal(hG[0], g0[0]) and np.array_equal(hG[1], g0[1])
    keep = hG[0][:10].copy()
    view = hG[0][:10]
    del hG, g1                                                                # the view keeps the pinned block alive
    assert np.array_equal(view, keep)
    with pytest.raises(ValueError):
        ctx.adjoint(None, out=[np.empty(3), np.empty(3)])
    dom.drop_contexts()


def test_the_other_problems_accept_device_vectors():
    """SH23, SHB23 (both formulations) and Poiseuille: the same callables with DeviceVectors — same numbers as with NumPy vectors, gradients
    come back as DeviceVectors, and a short optimisation keeps its iterate sequence bit for bit."""
    from spheremanopt_amd import poiseuille as pz, sh23, shb23
    # SH23
    dom, X = sh23.Generate_IC(0.0725, Npts=64, seed=42)
    buf = sh23.GEN_BUFFER(dom, 30)
    args = [dom, 0.1, 30, 30, buf, None, "Discrete"]
    J = sh23.FWD_Solve_IVP_Lin([X], *args); g = sh23.ADJ_Solve_IVP_Lin([X], *args)[0]
    Xd = DeviceVector.from_numpy(X)
    Jd = sh23.FWD_Solve_IVP_Lin([Xd], *args); gd = sh23.ADJ_Solve_IVP_Lin([Xd], *args)[0]
    assert Jd == J and isinstance(gd, DeviceVector) and np.array_equal(gd.numpy(), g)
    assert sh23.Inner_Prod(Xd, gd, dom) == sh23.Inner_Prod(X, g, dom)
    runs = []
    for X0 in ([X], [DeviceVector.from_numpy(X)]):
        R, F, Xo = Optimise_On_Multi_Sphere(X0, [0.0725], sh23.FWD_Solve_IVP_Lin, sh23.ADJ_Solve_IVP_Lin, sh23.Inner_Prod, args, (dom, None),
                                            max_iters=3, alpha_k=1., LS='LS_wolfe', CG=True, verbose=False)
        runs.append((R, F, to_host(Xo)[0]))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and np.array_equal(runs[0][2], runs[1][2])
    # SHB23, discrete formulation
    dom, X = shb23.Generate_IC(128, M_0=0.0019, seed=42)
    buf = shb23.GEN_BUFFER(128, dom, 20)
    J = shb23.FWD_Solve_IVP_Discrete([X], dom, buf, 20, 1e-2); g = shb23.ADJ_Solve_IVP_Discrete([X], dom, buf, 20, 1e-2)[0]
    Xd = DeviceVector.from_numpy(X)
    Jd = shb23.FWD_Solve_IVP_Discrete([Xd], dom, buf, 20, 1e-2); gd = shb23.ADJ_Solve_IVP_Discrete([Xd], dom, buf, 20, 1e-2)[0]
    assert Jd == J and np.array_equal(gd.numpy(), g) and shb23.Inner_Prod_Discrete(Xd, gd, dom) == shb23.Inner_Prod_Discrete(X, g, dom)
    # Poiseuille, discrete formulation
    from oracle.poiseuille import PoiseuilleOracle, synthetic_ic
    X = synthetic_ic(PoiseuilleOracle(24, 24, dt=5e-3, N_ITERS=5, s=0, delta=0.3), 42)
    dom = pz.PoiseuilleDomain(24, 24)
    buf = pz.GEN_BUFFER(24, 24, dom, 5)
    args = [dom, 500., 0.05, 5, buf, 5e-3, 0, 1., 0.3]
    J = pz.FWD_Solve_Discrete([X], *args); g = pz.ADJ_Solve_Discrete([X], *args)[0]
    Xd = DeviceVector.from_numpy(X)
    Jd = pz.FWD_Solve_Discrete([Xd], *args); gd = pz.ADJ_Solve_Discrete([Xd], *args)[0]
    assert Jd == J and np.array_equal(gd.numpy(), g) and pz.Inner_Prod_Discrete(Xd, gd, dom) == pz.Inner_Prod_Discrete(X, g, dom)
    dom.drop_contexts()

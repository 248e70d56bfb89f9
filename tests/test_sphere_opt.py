"""Host driver vs traces captured from the reference optimiser (tools/gen_golden_reference.py).

Bit-exact: the driver is integer/branch logic around the callbacks, so with identical callbacks
the iterate sequence must be identical (north_star: "bit-exact for the CG iterate indexing").
"""
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN
from spheremanopt_amd import sphere_opt as so
from spheremanopt_amd.examples import pca
from spheremanopt_amd.test_grad import Adjoint_Gradient_Test, taylor_table


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("DIM", [16, 512])
def test_pca_traces_match_reference(DIM, in_tmp_cwd):
    gold = _load("pca_dim%d.npz" % DIM)
    np.random.seed(0)
    M = pca.Hessian_Matrix(DIM)
    X_0 = np.random.rand(DIM)
    assert np.array_equal(X_0, gold["X_0"])
    assert np.array_equal(np.asarray([M.sum(), np.abs(M).sum(), M[0, 1], M[-1, -2]]), gold["M_checksum"])

    for tag, LS, CG in (("sd", "LS_armijo", False), ("cg", "LS_wolfe", True)):
        calls = [0, 0]

        def f(X, *a):
            calls[0] += 1
            return pca.Objective(X, *a)

        def g(X, *a):
            calls[1] += 1
            return pca.Gradient(X, *a)

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            RES, FUN, X_opt = so.Optimise_On_Multi_Sphere([X_0.copy()], [1.], f, g, pca.Vector_Inner_Product,
                                                          (M, True), (), LS=LS, CG=CG, verbose=False)
        assert np.array_equal(np.asarray(RES), gold[tag + "_residual"]), tag
        assert np.array_equal(np.asarray(FUN), gold[tag + "_funct"]), tag
        assert np.array_equal(X_opt[0], gold[tag + "_xopt"]), tag
        assert calls == list(gold[tag + "_calls"]), tag
    assert os.path.exists("optimize_result.txt")
    if os.path.exists("DAL_PROGRESS.npz"):          # restart file (h5py absent: same keys as the reference's DAL_PROGRESS.h5)
        prog = np.load("DAL_PROGRESS.npz")
        assert int(prog["Iterations"]) == len(FUN) and np.array_equal(prog["Function_Value"], np.asarray(FUN))
        assert np.array_equal(prog["X_opt"][0], X_opt[0])
    back = so.load_progress()                           # the manual restart of the reference scripts: X_0 = DAL_file['X_opt']
    assert np.array_equal(np.asarray(back["X_opt"])[0], X_opt[0]) and int(back["Iterations"]) == len(FUN)

    if DIM == 512:   # config 1 acceptance: CG converges to the top eigenpair
        lam, vec = np.linalg.eigh(M)
        assert abs(FUN[-1] - lam[-1] / 2) < 1e-6
        assert np.linalg.norm(abs(vec[:, -1]) - abs(X_opt[0])) < 1e-5


def test_two_sphere_weighted_inner_product(in_tmp_cwd):
    gold = _load("two_sphere.npz")
    A, B, w = gold["A"], gold["B"], gold["w"]

    def f2(X, *a):
        return -0.5 * X[0] @ A @ X[0] - X[0] @ B @ X[1] + 0.25 * np.sum(X[1] ** 4)

    def g2(X, *a):
        return [(-A @ X[0] - B @ X[1]) / w, (-B.T @ X[0] + X[1] ** 3) / w]

    def g2_bad(X, *a):
        return [-A @ X[0] - B @ X[1], -B.T @ X[0] + X[1] ** 3]

    def ip2(x, y, w):
        return np.dot(x, w * y)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        RES, FUN, X = so.Optimise_On_Multi_Sphere([gold["X0a"].copy(), gold["X0b"].copy()], [1., 2.], f2, g2, ip2,
                                                  (), (w,), alpha_k=2., max_iters=40, verbose=False)
        assert np.array_equal(np.asarray(RES), gold["residual"])
        assert np.array_equal(np.asarray(FUN), gold["funct"])
        assert np.array_equal(X[0], gold["xa"]) and np.array_equal(X[1], gold["xb"])
        # line-search failure path: early return of the last recorded state (SGD:791-793)
        RES, FUN, X = so.Optimise_On_Multi_Sphere([gold["X0a"].copy(), gold["X0b"].copy()], [1., 2.], f2, g2_bad, ip2,
                                                  (), (w,), alpha_k=2., max_iters=40, verbose=False)
        assert np.array_equal(np.asarray(RES), gold["bad_residual"])
        assert np.array_equal(np.asarray(FUN), gold["bad_funct"])
        assert np.array_equal(X[0], gold["bad_xa"]) and np.array_equal(X[1], gold["bad_xb"])


def test_geometry_and_scalar_searches():
    gold = _load("linesearch_units.npz")
    x, d, g, w = gold["x"], gold["d"], gold["g"], gold["w"]
    ipw = lambda a, b, w: float(np.dot(a, w * b))
    assert np.array_equal(so.Update_vector(x, 0.37, d, 2.5, ipw, (w,)), gold["update"])
    assert np.array_equal(so.tangent_vector(x, g, ipw, (w,)), gold["tangent"])
    assert np.array_equal(so.transport_vector(x, d, ipw, (w,)), gold["transport"])
    # tangent / transported vectors are orthogonal to x in the weighted product
    assert abs(ipw(x, so.tangent_vector(x, g, ipw, (w,)), w)) < 1e-12
    # retraction lands on the sphere
    assert abs(ipw(gold["update"], gold["update"], w) - 2.5) < 1e-12

    phi = lambda a: (a - 0.7) ** 4 + 0.3 * np.sin(3 * a) + 0.1 * a
    dphi = lambda a: 4 * (a - 0.7) ** 3 + 0.9 * np.cos(3 * a) + 0.1
    arm = [so.scalar_search_armijo(phi, phi(0.), dphi(0.), alpha0=a0) for a0 in (0.1, 1.0, 3.0, 8.0)]
    got = np.asarray([[a if a is not None else np.nan, v] for a, v in arm])
    assert np.array_equal(got, gold["armijo"], equal_nan=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wol = []
        for amax in (None, 1.5, 50.):
            r = so.scalar_search_wolfe2(phi, dphi, phi(0.), phi(0.) + 0.05, dphi(0.), amax=amax)
            wol.append([np.nan if v is None else v for v in r])
    assert np.array_equal(np.asarray(wol), gold["wolfe"], equal_nan=True)
    assert np.array_equal(np.asarray([so._cubicmin(0., 1., -1., 1., 0.6, 0.4, 0.7),
                                      so._cubicmin(0.2, 2., -3., 1.3, 1.1, 0.9, 0.8)]), gold["cubicmin"])
    assert np.array_equal(np.asarray([so._quadmin(0., 1., -1., 1., 0.6), so._quadmin(0.2, 2., -3., 1.3, 1.1)]),
                          gold["quadmin"])
    assert so._cubicmin(0., 1., -1., 0., 1., 0., 1.) is None      # degenerate -> None, not an exception
    assert so._quadmin(0., 1., -1., 0., 1.) is None


def test_taylor_table_matches_reference(in_tmp_cwd):
    gold = _load("taylor_table.npz")
    Q = gold["Q"]
    fq = lambda X, *a: float(0.5 * X[0] @ Q @ X[0] + np.sum(X[0] ** 3))
    gq = lambda X, *a: [Q @ X[0] + 3 * X[0] ** 2]
    ipq = lambda a, b, *r: float(np.dot(a, b))
    AA = Adjoint_Gradient_Test(gold["x0"], gold["dx0"], fq, gq, ipq, epsilon=1e-2)
    assert np.array_equal(AA, gold["AA"])
    assert np.array_equal(np.load("eps_TestR_TestR2_h_h2.npy"), gold["AA"])
    # list-valued X0/dX0 take the other branch of the reference and must agree
    AA2 = taylor_table([gold["x0"]], [gold["dx0"]], fq, gq, ipq, epsilon=1e-2)
    assert np.allclose(AA2, AA, rtol=0, atol=0)
    assert np.all(np.abs(AA[4, :4] - 2) < 5e-3) and np.all(np.abs(AA[3, :4] - 1) < 0.2)


def test_result_record_format():
    R = so.result(2)
    R.Iterations = 1
    R.Residual = [[0.5], [0.25]]
    R.Step_Size = [0.1]
    R.Function_Value = [3.0]
    s = str(R)
    assert s.startswith("Optimize_rotation succeed \n")
    assert "Residual error r_k   = [0.5, 0.25]\n" in s and s.endswith("J(X_opt)             = 3.0\n")


def test_products_files_round_trip(in_tmp_cwd):
    """scalar_data / CheckPoints writers (HDF5 paths as keys; .npz when h5py is absent) and the File_Manips callback."""
    import os
    from spheremanopt_amd import products
    g = {"tasks/Kinetic energy": np.arange(6.).reshape(6, 1), "scales/sim_time": 0.1 * np.arange(6), "scales/x/1.5": np.linspace(0, 1, 4)}
    f = products.write_products(os.path.join("scalar_data", "scalar_data_s1"), g)
    products.write_products(os.path.join("CheckPoints", "CheckPoints_s1"), {"tasks/u": np.ones((2, 4))})
    back = products.read_products(f)
    assert set(back) == set(g) and all(np.array_equal(back[k], g[k]) for k in g)
    products.File_Manips(3)
    ext = os.path.splitext(f)[1]
    assert os.path.exists("scalar_data_iter_3" + ext) and os.path.exists("CheckPoints_iter_3" + ext)
    assert np.array_equal(products.read_products("CheckPoints_iter_3" + ext)["tasks/u"], np.ones((2, 4)))
    assert list(products.sample_iterations(45)) == [0, 20, 40]


def test_vec_field_helpers_are_views_of_the_flat_layout():
    """Vec_to_Field / Field_to_Vec of every problem module: the reference's flat-vector layouts, no device needed."""
    from spheremanopt_amd import kdyn, poiseuille, sh23, shb23
    d = sh23.SH23Domain(16)
    x = np.arange(32.)
    assert np.shares_memory(sh23.Vec_to_Field(d, x), x) and np.array_equal(sh23.Field_to_Vec(d, sh23.Vec_to_Field(d, x)), x)
    assert sh23.Integrate_Field(d, x) == np.mean(x)
    k = kdyn.KDynDomain(8)
    v = np.arange(3. * 12 ** 3)
    a, b, c = kdyn.Vec_to_Field(k, v)
    assert a.shape == (12, 12, 12) and np.shares_memory(b, v) and c[0, 0, 1] == 2 * 12 ** 3 + 1      # component-major, z fastest
    assert np.array_equal(kdyn.Field_to_Vec(k, a, b, c), v)
    s = shb23.SHBDomain(64)
    assert np.array_equal(shb23.Field_to_Vec(s, shb23.Vec_to_Field(s, np.arange(64.))), np.arange(64.))
    p = poiseuille.PoiseuilleDomain(24, 12)
    w = np.arange(2. * 24 * 12)
    u, ww = poiseuille.Vec_to_Field(p, [w])
    assert u.shape == (24, 12) and ww[0, 1] == 24 * 12 + 1 and np.array_equal(poiseuille.Field_to_Vec(p, u, ww), w)
    with pytest.raises(ValueError):
        kdyn.Vec_to_Field(k, v[:-1])

#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the importable parts of the reference.

Runs ONLY in the build container (needs /root/reference); the GPU box never sees the
reference, it only reads the committed fixtures.  What can be imported:

  * Sphere_Grad_Descent.py, TestGrad.py, Example_Problems/PCA_example.py — with empty stub
    modules for the two absent optional imports ``mpi4py`` and ``h5py``;
  * the pure NumPy/SciPy helpers of FWD_Solve_SHB23.py (transform*, weightMatrixDisc,
    Inner_Prod_Discrete) — additionally stubbing ``dedalus.core.{field,system}``.

Nothing that calls Dedalus can run (Dedalus is not installed), so the PDE solves themselves
have no reference-generated vectors: "parity unpinned" for those (see oracle/README.md).

Usage:  python tools/gen_golden_reference.py   (writes into tests/golden/)
"""
import importlib.util
import os
import sys
import tempfile
import types
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    os.makedirs(OUT, exist_ok=True)
    mpi = _stub("mpi4py"); mpi.MPI = types.SimpleNamespace()      # any attribute access fails inside try/except
    _stub("h5py")
    ded = _stub("dedalus"); core = _stub("dedalus.core")
    ded.core = core
    core.field = _stub("dedalus.core.field"); core.system = _stub("dedalus.core.system")

    sys.path.insert(0, REF)
    SGD = _load(os.path.join(REF, "Sphere_Grad_Descent.py"), "ref_SGD")
    TG = _load(os.path.join(REF, "TestGrad.py"), "ref_TG")
    PCA = _load(os.path.join(REF, "Example_Problems", "PCA_example.py"), "ref_PCA")
    SHB = _load(os.path.join(REF, "Example_Problems", "Bounded_Domain(Cheby)", "Swift_Hohenberg_Bounded",
                             "FWD_Solve_SHB23.py"), "ref_SHB")

    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)                 # the reference appends optimize_result.txt / writes .npy in the cwd
    warnings.simplefilter("ignore")
    try:
        # ---------------- (1) optimiser traces on the PCA example ----------------
        for DIM in (16, 512):
            np.random.seed(0)
            M = PCA.Hessian_Matrix(DIM)
            X_0 = np.random.rand(DIM)
            PCA.M = M                                  # Objective/Gradient read the module global (PCA_example.py:70,90)
            out = {"X_0": X_0, "M_checksum": np.asarray([M.sum(), np.abs(M).sum(), M[0, 1], M[-1, -2]])}  # M itself: seed-0 recipe
            for tag, LS, CG in (("sd", "LS_armijo", False), ("cg", "LS_wolfe", True)):
                calls = {"f": 0, "g": 0}

                def f(X, *a, **k):
                    calls["f"] += 1
                    return PCA.Objective(X, *a, **k)

                def g(X, *a, **k):
                    calls["g"] += 1
                    return PCA.Gradient(X, *a, **k)

                RES, FUN, X_opt = SGD.Optimise_On_Multi_Sphere([X_0.copy()], [1.], f, g, PCA.Vector_Inner_Product,
                                                               (M, True), (), LS=LS, CG=CG, verbose=False)
                out[tag + "_residual"] = np.asarray(RES)
                out[tag + "_funct"] = np.asarray(FUN)
                out[tag + "_xopt"] = np.asarray(X_opt[0])
                out[tag + "_calls"] = np.asarray([calls["f"], calls["g"]])
            np.savez(os.path.join(OUT, "pca_dim%d.npz" % DIM), **out)
            print("PCA DIM=%d: SD its=%d  CG its=%d  r_cg=%.3e" % (DIM, out["sd_funct"].size, out["cg_funct"].size,
                                                                out["cg_residual"][0, -1]))

        # two-component problem (exercises the per-component sums in beta and phi')
        np.random.seed(3)
        A = np.random.randn(12, 12); A = 0.5 * (A + A.T)
        B = np.random.randn(12, 12)

        def f2(X, *a):
            return -0.5 * X[0] @ A @ X[0] - X[0] @ B @ X[1] + 0.25 * np.sum(X[1] ** 4)

        def g2(X, *a):                                  # gradient w.r.t. the weighted inner product ip2
            return [(-A @ X[0] - B @ X[1]) / w, (-B.T @ X[0] + X[1] ** 3) / w]

        def g2_bad(X, *a):                              # inconsistent gradient -> line search gives up early
            return [-A @ X[0] - B @ X[1], -B.T @ X[0] + X[1] ** 3]

        def ip2(x, y, w):
            return np.dot(x, w * y)

        w = np.linspace(0.5, 1.5, 12)
        X0a, X0b = np.random.rand(12), np.random.rand(12)
        RES, FUN, X_opt = SGD.Optimise_On_Multi_Sphere([X0a.copy(), X0b.copy()], [1., 2.], f2, g2, ip2, (), (w,),
                                                       alpha_k=2., max_iters=40, verbose=False)
        RESb, FUNb, X_optb = SGD.Optimise_On_Multi_Sphere([X0a.copy(), X0b.copy()], [1., 2.], f2, g2_bad, ip2, (), (w,),
                                                          alpha_k=2., max_iters=40, verbose=False)
        np.savez(os.path.join(OUT, "two_sphere.npz"), A=A, B=B, w=w, X0a=X0a, X0b=X0b, residual=np.asarray(RES),
                 funct=np.asarray(FUN), xa=X_opt[0], xb=X_opt[1],
                 bad_residual=np.asarray(RESb), bad_funct=np.asarray(FUNb), bad_xa=X_optb[0], bad_xb=X_optb[1])
        print("two-sphere: its=%d (inconsistent-gradient variant: %d)" % (len(FUN), len(FUNb)))

        # ---------------- (2) line-search / geometry unit vectors ----------------
        rs = np.random.RandomState(5)
        x, d, gvec = rs.randn(9), rs.randn(9), rs.randn(9)
        ipw = lambda a, b, w: float(np.dot(a, w * b))
        wv = np.linspace(1., 2., 9)
        geo = {
            "x": x, "d": d, "g": gvec, "w": wv,
            "update": SGD.Update_vector(x, 0.37, d, 2.5, ipw, (wv,)),
            "tangent": SGD.tangent_vector(x, gvec, ipw, (wv,)),
            "transport": SGD.transport_vector(x, d, ipw, (wv,)),
        }
        phi = lambda a: (a - 0.7) ** 4 + 0.3 * np.sin(3 * a) + 0.1 * a
        dphi = lambda a: 4 * (a - 0.7) ** 3 + 0.9 * np.cos(3 * a) + 0.1
        arm = [SGD.scalar_search_armijo(phi, phi(0.), dphi(0.), alpha0=a0) for a0 in (0.1, 1.0, 3.0, 8.0)]
        geo["armijo"] = np.asarray([[a if a is not None else np.nan, v] for a, v in arm])
        wol = []
        for amax in (None, 1.5, 50.):
            r = SGD.scalar_search_wolfe2(phi, dphi, phi(0.), phi(0.) + 0.05, dphi(0.), amax=amax)
            wol.append([np.nan if v is None else v for v in r])
        geo["wolfe"] = np.asarray(wol)
        geo["cubicmin"] = np.asarray([SGD._cubicmin(0., 1., -1., 1., 0.6, 0.4, 0.7), SGD._cubicmin(0.2, 2., -3., 1.3, 1.1, 0.9, 0.8)])
        geo["quadmin"] = np.asarray([SGD._quadmin(0., 1., -1., 1., 0.6), SGD._quadmin(0.2, 2., -3., 1.3, 1.1)])
        np.savez(os.path.join(OUT, "linesearch_units.npz"), **geo)

        # ---------------- (3) Taylor-test table on an analytic cubic ----------------
        Q = rs.randn(7, 7); Q = Q + Q.T
        fq = lambda X, *a: float(0.5 * X[0] @ Q @ X[0] + np.sum(X[0] ** 3))
        gq = lambda X, *a: [Q @ X[0] + 3 * X[0] ** 2]
        ipq = lambda a, b, *r: float(np.dot(a, b))
        x0, dx0 = rs.randn(7), rs.randn(7)
        TG.Adjoint_Gradient_Test(x0, dx0, fq, gq, ipq, epsilon=1e-2)
        AA = np.load("eps_TestR_TestR2_h_h2.npy")
        np.savez(os.path.join(OUT, "taylor_table.npz"), Q=Q, x0=x0, dx0=dx0, AA=AA)
        print("Taylor slopes:", AA[4, :4])

        # ---------------- (4) SHB23 Chebyshev helpers ----------------
        cheb = {}
        for N in (8, 512):
            v = np.random.RandomState(N).randn(N)
            cheb["v%d" % N] = v
            cheb["T%d" % N] = SHB.transform(v)
            cheb["Tinv%d" % N] = SHB.transformInverse(v)
            cheb["Tadj%d" % N] = SHB.transformAdjoint(v)
            cheb["Tinvadj%d" % N] = SHB.transformInverseAdjoint(v)

        class _Dom:                                     # the two attributes the helpers read (SHB:71,193)
            hypervolume = 40.0

            def __init__(self, n):
                self.n = n

            def grid(self, axis, scales=1):
                i = np.arange(self.n)
                return 20.0 * (-np.cos(np.pi * (i + 0.5) / self.n))

        for N in (8, 512):
            SHB.Npts = N                               # weightMatrixDisc reads the module global (SHB:72)
            cheb["W%d" % N] = SHB.weightMatrixDisc(_Dom(N))
            cheb["ip%d" % N] = SHB.Inner_Prod_Discrete(cheb["v%d" % N], cheb["T%d" % N], _Dom(N))
        np.savez(os.path.join(OUT, "shb_helpers.npz"), **cheb)
        print("SHB helpers: sum(W512) =", cheb["W512"].sum())
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    main()

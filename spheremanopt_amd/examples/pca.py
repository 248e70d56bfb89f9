"""Config 1 — largest principal component of a random symmetric matrix (host only).

    min_X  J(X) = -1/2 X^T M X   s.t. ||X|| = 1

Plumbing check for the optimiser: no GPU, no PDE.  Counterpart of the reference's
Example_Problems/PCA_example.py (Hessian_Matrix :14-31, Objective :56-73, Gradient :75-103,
Vector_Inner_Product :105-107, main :109-150).  Unlike the reference, the callbacks take the
matrix from ``args_f[0]`` instead of a module global, and the random draws are seedable.

Run:  python -m spheremanopt_amd.examples.pca [DIM]
"""
import sys

import numpy as np

from ..sphere_opt import Optimise_On_Multi_Sphere


def Hessian_Matrix(DIM, rng=np.random):
    """Random symmetric M, redrawn until x^T M x > 0 for one fixed random x (PCA_example.py:14-31).

    With ``np.random.seed(s)`` beforehand the draw sequence (rand(DIM), then randn(DIM,DIM) per
    attempt) is the reference's, so fixtures only need the seed."""
    probe = rng.rand(DIM)
    while True:
        M = rng.randn(DIM, DIM)
        M = 0.5 * (M + M.T)
        if not np.dot(probe, M @ probe) < 0.:
            return M


def Objective(X, M, *unused, **unused_kw):
    xk = X[0]
    g_k = -np.matmul(M, xk)
    return (1. / 2.) * np.dot(xk, g_k)


def Gradient(X, M, *unused, **unused_kw):
    return [-np.matmul(M, X[0])]


def Vector_Inner_Product(f, g, *args_IP):
    return np.dot(f, g)


def main(DIM=100, seed=None):
    if seed is not None:
        np.random.seed(seed)
    M = Hessian_Matrix(DIM)
    X_0 = np.random.rand(DIM)
    args_f = (M, True)

    lam, vec = np.linalg.eigh(M)
    v = vec[:, -1]

    _, FUNCT_SD, x_sd = Optimise_On_Multi_Sphere([X_0], [1.], Objective, Gradient, Vector_Inner_Product, args_f, (),
                                                 LS='LS_armijo', CG=False, verbose=False)
    print("Error of SD = ", np.linalg.norm(abs(v) - abs(x_sd[0]), 2), " iterations", len(FUNCT_SD))
    _, FUNCT_CG, x_cg = Optimise_On_Multi_Sphere([X_0], [1.], Objective, Gradient, Vector_Inner_Product, args_f, (),
                                                 LS='LS_wolfe', CG=True, verbose=False)
    print("Error of CG = ", np.linalg.norm(abs(v) - abs(x_cg[0]), 2), " iterations", len(FUNCT_CG))
    kappa = np.linalg.cond(M)
    print("R^2 = ", pow((kappa - 1.) / (kappa + 1.), 2), "  lambda_max/2 =", lam[-1] / 2, " J_cg =", FUNCT_CG[-1])
    return x_sd, x_cg


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, seed=0)

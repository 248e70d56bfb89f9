"""The reference's periodic Swift-Hohenberg script (FWD_Solve_SH23.py:750-784) on the MI355X path.

Run:  python -m spheremanopt_amd.examples.sh23_optimise [--max-iters 200] [--test-gradient]
(defaults = the reference's: E_0 = 0.0725, dt = 0.05, T = 50, alpha_k = pi, LS_wolfe + CG).
"""
import argparse

import numpy as np

from ..sh23 import ADJ_Solve_IVP_Lin, FWD_Solve_IVP_Lin, GEN_BUFFER, Generate_IC, Inner_Prod
from ..sphere_opt import Optimise_On_Multi_Sphere
from ..test_grad import Adjoint_Gradient_Test


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--npts", type=int, default=256)
    ap.add_argument("--dt", type=float, default=0.05)
    ap.add_argument("--T", type=float, default=50.)
    ap.add_argument("--max-iters", type=int, default=200)
    ap.add_argument("--test-gradient", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    E_0, dt = 0.0725, a.dt
    N_ITERS = int(a.T / dt)
    N_SUB_ITERS = N_ITERS // 1
    domain, X_0 = Generate_IC(E_0, Npts=a.npts, prep=True)
    X_FWD_DICT = GEN_BUFFER(domain, N_SUB_ITERS)
    args_IP = (domain, None)
    args_f = [domain, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, None, "Discrete"]
    AA = None
    if a.test_gradient:
        _, X1 = Generate_IC(1., Npts=a.npts, prep=True)
        _, dX = Generate_IC(1., Npts=a.npts, seed=7, prep=True)
        AA = Adjoint_Gradient_Test(X1, dX, FWD_Solve_IVP_Lin, ADJ_Solve_IVP_Lin, Inner_Prod, args_f, args_IP, epsilon=1e-04)
    RESIDUAL, FUNCT, X_opt = Optimise_On_Multi_Sphere([X_0], [E_0], FWD_Solve_IVP_Lin, ADJ_Solve_IVP_Lin, Inner_Prod, args_f, args_IP,
                                                      max_iters=a.max_iters, alpha_k=np.pi, LS='LS_wolfe', CG=True, verbose=not a.quiet)
    return RESIDUAL, FUNCT, X_opt, AA


if __name__ == "__main__":
    R, F, _, _ = main()
    print("J_k per iteration:", F)

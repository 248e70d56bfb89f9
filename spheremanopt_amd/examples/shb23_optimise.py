"""The reference's bounded Swift-Hohenberg script (FWD_Solve_SHB23.py:967-998) on the MI355X path.

Run:  python -m spheremanopt_amd.examples.shb23_optimise [--max-iters 50]
(defaults = the reference's: Npts = 2*256, dt = 0.01, T = 20, M_0 = 0.0019, err_tol = 1e-5).
"""
import argparse

from ..shb23 import ADJ_Solve, FWD_Solve, GEN_BUFFER, Generate_IC, Inner_Prod
from ..sphere_opt import Optimise_On_Multi_Sphere
from ..test_grad import Adjoint_Gradient_Test


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--npts", type=int, default=512)
    ap.add_argument("--T", type=float, default=20.)
    ap.add_argument("--max-iters", type=int, default=50)
    ap.add_argument("--test-gradient", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    dt, M_0 = 0.01, 0.0019
    N_ITERS = int(a.T / dt)
    domain, X0 = Generate_IC(a.npts, (-20., 20.), M_0)
    X_FWD_DICT = GEN_BUFFER(a.npts, domain, N_ITERS)
    args_IP = (domain, 'np_vector')
    args_f = (domain, X_FWD_DICT, N_ITERS)
    AA = None
    if a.test_gradient:
        _, dX0 = Generate_IC(a.npts, (-20., 20.), M_0, seed=7)
        AA = Adjoint_Gradient_Test([X0], [dX0], FWD_Solve, ADJ_Solve, Inner_Prod, args_f, args_IP, epsilon=1e-04)
    RESIDUAL, FUNCT, X_opt = Optimise_On_Multi_Sphere([X0], [M_0], FWD_Solve, ADJ_Solve, Inner_Prod, args_f, args_IP, err_tol=1e-05,
                                                      max_iters=a.max_iters, LS='LS_wolfe', CG=True, verbose=not a.quiet)
    return RESIDUAL, FUNCT, X_opt, AA


if __name__ == "__main__":
    R, F, _, _ = main()
    print("J_k per iteration:", F)

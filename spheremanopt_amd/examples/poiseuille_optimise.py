"""The reference's plane-Poiseuille optimal-mixing script (FWD_Solve_Poiseuille.py:1743-1777, "Discrete" formulation) on the MI355X path.

Run:  python -m spheremanopt_amd.examples.poiseuille_optimise [--max-iters 20]
(defaults = the reference's: Nx, Nz = 3/2 * (256, 128), Re = 500, Ri = 0.05, T = 5, dt = 5e-3, E_0 = 0.02, delta = 0.125, s = 1: mix-norm).
"""
import argparse

from .. import poiseuille as pz
from ..poiseuille import ADJ_Solve, FWD_Solve, GEN_BUFFER, Generate_IC, Inner_Prod
from ..sphere_opt import Optimise_On_Multi_Sphere
from ..test_grad import Adjoint_Gradient_Test


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=256)
    ap.add_argument("--nz", type=int, default=128)
    ap.add_argument("--T", type=float, default=5.)
    ap.add_argument("--dt", type=float, default=5e-3)
    ap.add_argument("--s", type=int, default=1, help="0: time-averaged kinetic energy, 1: mix-norm")
    ap.add_argument("--max-iters", type=int, default=20)
    ap.add_argument("--continuous", action="store_true", help='the script\'s Adjoint_type = "Continuous" (its default): Dedalus-IVP formulation')
    ap.add_argument("--test-gradient", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    Re, Ri, E_0, Prandtl, δ = 500., 0.05, 0.02, 1., 0.125
    N_ITERS = int(a.T / a.dt)
    Nx, Nz = 3 * a.nx // 2, 3 * a.nz // 2                     # the Discrete formulation works at the 3/2-scaled resolution (:1752-1755)
    domain, Ux0 = Generate_IC(Nx, Nz, E_0=E_0)
    fwd, adj, ip = FWD_Solve, ADJ_Solve, Inner_Prod
    if a.continuous:                                          # same grid, a.nx x a.nz modes; the IC is re-normalised with the exact-integration product
        domain = pz.PoiseuilleDomain(a.nx, a.nz, continuous=True)
        fwd, adj, ip = pz.FWD_Solve_Cnts, pz.ADJ_Solve_Cnts, pz.Inner_Prod_Cnts
        Ux0 = [Ux0[0] * (E_0 / ip(Ux0[0], Ux0[0], domain)) ** 0.5]
        Nx, Nz = a.nx, a.nz
    X_FWD_DICT = GEN_BUFFER(Nx, Nz, domain, N_ITERS)
    args_f = [domain, Re, Ri, N_ITERS, X_FWD_DICT, a.dt, a.s, Prandtl, δ]
    args_IP = [domain, None]
    AA = None
    if a.test_gradient:
        _, dUx0 = Generate_IC(3 * a.nx // 2, 3 * a.nz // 2, E_0=E_0, seed=7)
        AA = Adjoint_Gradient_Test(Ux0, dUx0, fwd, adj, ip, args_f, args_IP, epsilon=1e-04)
    RESIDUAL, FUNCT, U_opt = Optimise_On_Multi_Sphere(Ux0, [E_0], fwd, adj, ip, args_f, args_IP, err_tol=1e-06,
                                                      max_iters=a.max_iters, alpha_k=100., LS='LS_wolfe', CG=True, verbose=not a.quiet)
    return RESIDUAL, FUNCT, U_opt, AA


if __name__ == "__main__":
    R, F, _, _ = main()
    print("J_k per iteration:", F)

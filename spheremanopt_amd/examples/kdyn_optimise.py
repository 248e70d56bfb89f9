"""The reference's kinematic-dynamo script (FWD_Solve_KDyn.py:1025-1067) on the MI355X path, unchanged call sequence:

    Generate_IC -> GEN_BUFFER -> [Adjoint_Gradient_Test] -> Optimise_On_Multi_Sphere(X_0, [M_0, E_0], FWD, ADJ, Inner_Prod_3, ...)

Run:  python -m spheremanopt_amd.examples.kdyn_optimise [--npts 24] [--dt 5e-4] [--max-iters 10] [--test-gradient] [--device-vectors] [--devices 0,1,...]
(defaults = the reference's: Npts = 24, Rm = 1, dt = 5e-4, T = Rm, alpha_k = 100, 10 optimiser iterations).
"""
import argparse

from ..kdyn import ADJ_Solve_IVP_Lin, FWD_Solve_IVP_Lin, GEN_BUFFER, Generate_IC, Inner_Prod_3
from ..sphere_opt import Optimise_On_Multi_Sphere
from ..test_grad import Adjoint_Gradient_Test


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--npts", type=int, default=24)
    ap.add_argument("--rm", type=float, default=1.0)
    ap.add_argument("--dt", type=float, default=5e-4)
    ap.add_argument("--max-iters", type=int, default=10)
    ap.add_argument("--steps", type=int, default=None, help="N_ITERS (default int(Rm/dt))")
    ap.add_argument("--test-gradient", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--device-vectors", action="store_true",
                    help="keep X, d, g in HBM (spheremanopt_amd.devvec.DeviceVector): no per-call PCIe traffic, same iterates bit for bit")
    ap.add_argument("--devices", default=None,
                    help="GPU ordinals, e.g. 0,1,2,3,4,5,6,7: this ONE process runs the problem slab-decomposed over them (smo_create_multi); with "
                         "--device-vectors the optimiser's vectors stay distributed over those GPUs")
    a = ap.parse_args(argv)

    Rm, dt, Npts = a.rm, a.dt, a.npts
    N_ITERS = a.steps or int(Rm / dt)
    N_SUB_ITERS = N_ITERS // 1
    M_0 = E_0 = 1.0
    domain, Bx0, Ux = Generate_IC(Npts, (0., 2. * 3.141592653589793), M_0, True, reference_recipe=True, Rm=Rm, dt=dt)
    if a.devices:                                           # the IC was prepared on one GPU; the optimisation runs on all of them
        from ..kdyn import KDynDomain
        domain.drop_contexts()
        domain = KDynDomain(Npts, domain.interval, devices=[int(d) for d in a.devices.split(",")])
    X_FWD_DICT = GEN_BUFFER(Npts, domain, N_SUB_ITERS)
    X_0, Constraints = [Bx0, Ux], [M_0, E_0]
    if a.device_vectors:
        from ..devvec import to_device, to_devices
        X_0 = to_devices(X_0, domain.devices) if a.devices else to_device(X_0, domain.device)
    args_IP = (domain, None)
    args_f = [domain, Rm, dt, N_ITERS, N_SUB_ITERS, X_FWD_DICT, "Final", "Discrete"]

    AA = None
    if a.test_gradient:
        _, dBx0, dUx = Generate_IC(Npts, M_0=M_0, U_Noise=True, seeds=(3, 4))
        AA = Adjoint_Gradient_Test(X_0, [dBx0, 0. * dUx], FWD_Solve_IVP_Lin, ADJ_Solve_IVP_Lin, Inner_Prod_3, args_f, args_IP,
                                   epsilon=1e-04)
    RESIDUAL, FUNCT, X_opt = Optimise_On_Multi_Sphere(X_0, Constraints, FWD_Solve_IVP_Lin, ADJ_Solve_IVP_Lin, Inner_Prod_3, args_f,
                                                      args_IP, max_iters=a.max_iters, alpha_k=100., LS='LS_wolfe', CG=True,
                                                      verbose=not a.quiet)
    return RESIDUAL, FUNCT, X_opt, AA


if __name__ == "__main__":
    R, F, _, _ = main()
    print("J_k (final magnetic energy) per iteration:", F)

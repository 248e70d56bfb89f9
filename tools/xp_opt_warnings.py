"""Which alpha_k lets the Wolfe search of the multi-device optimiser tests converge without hitting its cap?  """
import os, sys, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from spheremanopt_amd import kdyn
from spheremanopt_amd.sphere_opt import Optimise_On_Multi_Sphere
N, n, dt = 16, 8, 5e-3
for alpha in (1., 10., 100., 1000.):
    for LS in ("LS_wolfe", "LS_armijo"):
        dom = kdyn.KDynDomain(N, devices=[0, 0])
        B, U = kdyn.synthetic_field(dom.G, 1), kdyn.synthetic_field(dom.G, 2)
        buf = kdyn.GEN_BUFFER(N, dom, n)
        args_f = [dom, 1.0, dt, n, n, buf, "Final", "Discrete"]
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            R, F, X = Optimise_On_Multi_Sphere([B, U], [1.0, 1.0], kdyn.FWD_Solve_IVP_Lin, kdyn.ADJ_Solve_IVP_Lin, kdyn.Inner_Prod_3,
                                                args_f=args_f, args_IP=(dom, None), max_iters=3, alpha_k=alpha, LS=LS, CG=True, verbose=False)
        print(alpha, LS, "F", F, "warnings", [str(x.message)[:70] for x in w])
        dom.drop_contexts()

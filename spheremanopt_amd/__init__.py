"""spheremanopt_amd — MI355X-native forward/adjoint spectral-solve hot path behind the
``Optimise_On_Multi_Sphere(f, Grad_f, Inner_Product, ...)`` callback surface of
mannixp/SphereManOpt.  See DESIGN.md for the scope, INTEGRATION.md for the drop-in recipe.

Host side (pure Python, stays on the CPU like the reference's driver):
    sphere_opt.Optimise_On_Multi_Sphere, test_grad.Adjoint_Gradient_Test
Device side (hand-written HIP for gfx950 behind the C-ABI of include/smo.h):
    sh23 / kdyn / shb23 problem modules, each exporting the reference's callback names.
"""
from .sphere_opt import Optimise_On_Multi_Sphere, plot_optimisation, LineSearchWarning  # noqa: F401
from .test_grad import Adjoint_Gradient_Test  # noqa: F401

__version__ = "0.1.0"

"""Host-side optimiser on a product of spheres <X_i, X_i> = M_i.

This is the *caller* of the hot path: it only ever touches the
problem through the three user callbacks ``f`` / ``myfprime`` / ``inner_prod`` and it
stays on the host.  It is a from-scratch implementation that reproduces, floating-point
operation for floating-point operation, the iterate sequence of the reference driver

    reference: Sphere_Grad_Descent.py:692-838  (Optimise_On_Multi_Sphere)
               Sphere_Grad_Descent.py:66-190   (Armijo back-tracking)
               Sphere_Grad_Descent.py:198-613  (strong-Wolfe bracketing + zoom)
               Sphere_Grad_Descent.py:625-690  (transport / tangent / retraction)

so that ``tests/golden/pca_*.npz`` (traces captured from the reference on the PCA
example) are matched bit for bit.  The public names and signatures are the reference's.
"""
import copy
from warnings import warn

import numpy as np

try:                                     # optional: the reference hard-imports both
    import h5py as _h5py                 # (Sphere_Grad_Descent.py:3,6); neither is
except Exception:                        # needed for the optimisation itself.
    _h5py = None
try:
    from mpi4py import MPI as _MPI
except Exception:
    _MPI = None

__all__ = [
    "LineSearchWarning", "result", "Optimise_On_Multi_Sphere", "plot_optimisation",
    "LS_armijo_multiple", "LS_wolfe_multiple", "scalar_search_armijo",
    "scalar_search_wolfe2", "transport_vector", "tangent_vector", "Update_vector",
]


class LineSearchWarning(RuntimeWarning):
    """Raised (as a warning) when a line search gives up (Sphere_Grad_Descent.py:9)."""


class result:
    """Progress record printed / logged each iteration (Sphere_Grad_Descent.py:21-59)."""

    def __init__(self, components):
        self.N = components
        self.X_opt = np.asarray([])
        self.Iterations = 0
        self.Function_Evals = 0
        self.Gradient_Evals = 0
        self.Residual = []
        self.Step_Size = []
        self.Function_Value = []

    def __str__(self):
        last = self.Iterations - 1
        err = [self.Residual[c][last] for c in range(self.N)]
        rows = (
            ("Total iterations     = ", self.Iterations),
            ("Function evaluations = ", self.Function_Evals),
            ("Gradient evaluations = ", self.Gradient_Evals),
            ("Residual error r_k   = ", err),
            ("Step size      α_k   = ", self.Step_Size[last]),
            ("J(X_opt)             = ", self.Function_Value[last]),
        )
        return "Optimize_rotation succeed \n" + "".join(k + str(v) + "\n" for k, v in rows)


# ----------------------------------------------------------------------------------
# sphere geometry for the retraction  X+ = sqrt(M) (X + a d)/||X + a d||
# ----------------------------------------------------------------------------------

def transport_vector(X_k, dkm1, inner_prod, args_IP=(), kwargs_IP={}):
    """Project ``dkm1`` onto the tangent plane at ``X_k`` (Sphere_Grad_Descent.py:625-642)."""
    nrm = np.sqrt(inner_prod(X_k, X_k, *args_IP, **kwargs_IP))
    coeff = inner_prod(X_k, dkm1, *args_IP, **kwargs_IP) / (nrm ** 2)
    return dkm1 - coeff * X_k


def tangent_vector(X_k, Nab_Jk, inner_prod, args_IP=(), kwargs_IP={}):
    """Riemannian gradient from the Euclidean one (Sphere_Grad_Descent.py:644-659)."""
    coeff = inner_prod(X_k, Nab_Jk, *args_IP, **kwargs_IP) / inner_prod(X_k, X_k, *args_IP, **kwargs_IP)
    return Nab_Jk - coeff * X_k


def Update_vector(X_k, alpha_k, d_k, M_0, inner_prod, args_IP=(), kwargs_IP={}):
    """Retraction step (Sphere_Grad_Descent.py:661-690)."""
    moved = X_k + alpha_k * d_k
    sq = inner_prod(moved, moved, *args_IP, **kwargs_IP)
    return moved * np.sqrt(M_0 / sq)


def _retract_all(X_k, alpha, d_k, M_0, inner_prod, args_IP, kwargs_IP):
    """Fresh copy of X_k moved by alpha*d_k component-wise (Sphere_Grad_Descent.py:119-121)."""
    X_new = copy.deepcopy(X_k)
    for i, radius in enumerate(M_0):
        X_new[i] = Update_vector(X_k[i], alpha, d_k[i], radius, inner_prod, args_IP, kwargs_IP)
    return X_new


def _slope(g, d, M_0, inner_prod, args_IP, kwargs_IP):
    s = 0.
    for i, _ in enumerate(M_0):
        s += inner_prod(g[i], d[i], *args_IP, **kwargs_IP)
    return s


# ----------------------------------------------------------------------------------
# Armijo back-tracking
# ----------------------------------------------------------------------------------

def scalar_search_armijo(phi, phi0, derphi0, c1=1e-4, alpha0=1.0, amin=1e-06):
    """Interpolating back-tracking (Sphere_Grad_Descent.py:138-190; Nocedal & Wright pp. 56-57).

    Returns ``(alpha, phi(alpha))`` or ``(None, last phi)`` when alpha falls below ``amin``.
    """
    def sufficient(a, val):
        return val <= phi0 + c1 * a * derphi0

    phi_a0 = phi(alpha0)
    if sufficient(alpha0, phi_a0):
        return alpha0, phi_a0

    # minimiser of the quadratic through phi0, derphi0, phi(alpha0)
    alpha1 = -(derphi0) * alpha0 ** 2 / 2.0 / (phi_a0 - phi0 - derphi0 * alpha0)
    phi_a1 = phi(alpha1)
    if sufficient(alpha1, phi_a1):
        return alpha1, phi_a1

    # cubic through the two most recent trial points
    while alpha1 > amin:
        factor = alpha0 ** 2 * alpha1 ** 2 * (alpha1 - alpha0)
        a = alpha0 ** 2 * (phi_a1 - phi0 - derphi0 * alpha1) - \
            alpha1 ** 2 * (phi_a0 - phi0 - derphi0 * alpha0)
        a = a / factor
        b = -alpha0 ** 3 * (phi_a1 - phi0 - derphi0 * alpha1) + \
            alpha1 ** 3 * (phi_a0 - phi0 - derphi0 * alpha0)
        b = b / factor

        alpha2 = (-b + np.sqrt(abs(b ** 2 - 3 * a * derphi0))) / (3.0 * a)
        phi_a2 = phi(alpha2)
        if sufficient(alpha2, phi_a2):
            return alpha2, phi_a2

        if (alpha1 - alpha2) > alpha1 / 2.0 or (1 - alpha2 / alpha1) < 0.96:
            alpha2 = alpha1 / 2.0

        alpha0, alpha1 = alpha1, alpha2
        phi_a0, phi_a1 = phi_a1, phi_a2

    return None, phi_a1


def LS_armijo_multiple(f, inner_prod, M_0, X_k, g_k, d_k, old_fval, args_f=(), args_IP=(),
                       kwargs_f={}, kwargs_IP={}, alpha0=1.0, c1=1e-4):
    """Armijo search along the retraction curve (Sphere_Grad_Descent.py:66-136).

    Returns ``(alpha, n_f_evals, f(alpha))``.  Like the reference, X_k is passed through
    ``np.atleast_1d`` so ``f`` may be handed a 2-D array when all components have the same
    length (Sphere_Grad_Descent.py:111).
    """
    X_k = np.atleast_1d(X_k)
    n_calls = [0]

    def phi(alpha):
        n_calls[0] += 1
        return f(_retract_all(X_k, alpha, d_k, M_0, inner_prod, args_IP, kwargs_IP), *args_f, **kwargs_f)

    phi0 = phi(0.) if old_fval is None else old_fval
    derphi0 = _slope(g_k, d_k, M_0, inner_prod, args_IP, kwargs_IP)
    alpha, phi1 = scalar_search_armijo(phi, phi0, derphi0, c1=c1, alpha0=alpha0)
    return alpha, n_calls[0], phi1


# ----------------------------------------------------------------------------------
# strong-Wolfe search (bracketing phase + zoom), c1 < c2 < 1/2 for Fletcher-Reeves
# ----------------------------------------------------------------------------------

def _cubicmin(a, fa, fpa, b, fb, c, fc):
    """Minimiser of the cubic through (a,fa),(b,fb),(c,fc) with slope fpa at a, or None
    (Sphere_Grad_Descent.py:481-510)."""
    with np.errstate(divide='raise', over='raise', invalid='raise'):
        try:
            C = fpa
            db = b - a
            dc = c - a
            denom = (db * dc) ** 2 * (db - dc)
            d1 = np.empty((2, 2))
            d1[0, 0] = dc ** 2
            d1[0, 1] = -db ** 2
            d1[1, 0] = -dc ** 3
            d1[1, 1] = db ** 3
            [A, B] = np.dot(d1, np.asarray([fb - fa - C * db, fc - fa - C * dc]).flatten())
            A /= denom
            B /= denom
            radical = B * B - 3 * A * C
            xmin = a + (-B + np.sqrt(radical)) / (3 * A)
        except ArithmeticError:
            return None
    return xmin if np.isfinite(xmin) else None


def _quadmin(a, fa, fpa, b, fb):
    """Minimiser of the parabola through (a,fa),(b,fb) with slope fpa at a, or None
    (Sphere_Grad_Descent.py:512-529)."""
    with np.errstate(divide='raise', over='raise', invalid='raise'):
        try:
            db = b - a * 1.0
            B = (fb - fa - fpa * db) / (db * db)
            xmin = a - fpa / (2.0 * B)
        except ArithmeticError:
            return None
    return xmin if np.isfinite(xmin) else None


def _zoom(a_lo, a_hi, phi_lo, phi_hi, derphi_lo, phi, derphi, phi0, derphi0, c1, c2, extra_condition):
    """Nocedal & Wright Algorithm 3.6 (Sphere_Grad_Descent.py:531-613)."""
    max_zoom = 10
    delta1 = 0.2     # cubic interpolant must land this far inside the bracket
    delta2 = 0.1     # same for the quadratic fallback
    phi_rec, a_rec = phi0, 0
    it = 0
    while True:
        dalpha = a_hi - a_lo
        lo_end, hi_end = (a_hi, a_lo) if dalpha < 0 else (a_lo, a_hi)

        a_j = None
        if it > 0:
            cchk = delta1 * dalpha
            a_j = _cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec)
        if (it == 0) or (a_j is None) or (a_j > hi_end - cchk) or (a_j < lo_end + cchk):
            qchk = delta2 * dalpha
            a_j = _quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi)
            if (a_j is None) or (a_j > hi_end - qchk) or (a_j < lo_end + qchk):
                a_j = a_lo + 0.5 * dalpha

        phi_aj = phi(a_j)
        if (phi_aj > phi0 + c1 * a_j * derphi0) or (phi_aj >= phi_lo):
            phi_rec, a_rec = phi_hi, a_hi
            a_hi, phi_hi = a_j, phi_aj
        else:
            derphi_aj = derphi(a_j)
            if abs(derphi_aj) <= -c2 * derphi0 and extra_condition(a_j, phi_aj):
                return a_j, phi_aj, derphi_aj
            if derphi_aj * (a_hi - a_lo) >= 0:
                phi_rec, a_rec = phi_hi, a_hi
                a_hi, phi_hi = a_lo, phi_lo
            else:
                phi_rec, a_rec = phi_lo, a_lo
            a_lo, phi_lo, derphi_lo = a_j, phi_aj, derphi_aj
        it += 1
        if it > max_zoom:
            return None, None, None


def scalar_search_wolfe2(phi, derphi, phi0=None, old_phi0=None, derphi0=None, c1=1e-4, c2=0.4,
                         amax=None, extra_condition=None, maxiter=10):
    """Bracketing phase of the strong-Wolfe search (Sphere_Grad_Descent.py:344-479).

    Returns ``(alpha_star, phi_star, phi0, derphi_star)``; ``derphi_star`` is None on failure.
    """
    if phi0 is None:
        phi0 = phi(0.)
    if derphi0 is None:
        derphi0 = derphi(0.)

    alpha0 = 0
    if old_phi0 is not None and derphi0 != 0:
        alpha1 = min(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0)
    else:
        alpha1 = 1.0
    if alpha1 < 0:
        alpha1 = 1.0
    if amax is not None:
        alpha1 = min(alpha1, amax)

    phi_a1 = phi(alpha1)
    phi_a0 = phi0
    derphi_a0 = derphi0

    if extra_condition is None:
        def extra_condition(alpha, phi_val):
            return True

    for i in range(maxiter):
        if alpha1 == 0 or (amax is not None and alpha0 == amax):
            alpha_star, phi_star, derphi_star = None, phi0, None
            phi0 = old_phi0
            if alpha1 == 0:
                msg = 'Rounding errors prevent the line search from converging'
            else:
                msg = "The line search algorithm could not find a solution " + \
                      "less than or equal to amax: %s" % amax
            warn(msg, LineSearchWarning)
            break

        if (phi_a1 > phi0 + c1 * alpha1 * derphi0) or ((phi_a1 >= phi_a0) and i > 0):
            alpha_star, phi_star, derphi_star = _zoom(
                alpha0, alpha1, phi_a0, phi_a1, derphi_a0, phi, derphi, phi0, derphi0, c1, c2, extra_condition)
            break

        derphi_a1 = derphi(alpha1)
        if abs(derphi_a1) <= -c2 * derphi0:
            if extra_condition(alpha1, phi_a1):
                alpha_star, phi_star, derphi_star = alpha1, phi_a1, derphi_a1
                break

        if derphi_a1 >= 0:
            alpha_star, phi_star, derphi_star = _zoom(
                alpha1, alpha0, phi_a1, phi_a0, derphi_a1, phi, derphi, phi0, derphi0, c1, c2, extra_condition)
            break

        alpha2 = 2 * alpha1
        if amax is not None:
            alpha2 = min(alpha2, amax)
        alpha0, alpha1 = alpha1, alpha2
        phi_a0 = phi_a1
        phi_a1 = phi(alpha1)
        derphi_a0 = derphi_a1
    else:
        alpha_star, phi_star, derphi_star = alpha1, phi_a1, None
        warn('The line search algorithm did not converge', LineSearchWarning)

    return alpha_star, phi_star, phi0, derphi_star


def LS_wolfe_multiple(f, myfprime, inner_prod, M_0, X_k, g_k, d_k, old_fval=None, old_old_fval=None,
                      args_f=(), args_IP=(), kwargs_f={}, kwargs_IP={}, c1=1e-4, c2=0.4, amax=None,
                      extra_condition=None, maxiter=10):
    """Strong-Wolfe search along the retraction curve (Sphere_Grad_Descent.py:198-342).

    Returns ``(alpha, n_f, n_g, f(alpha), f(0), g_new)`` where ``g_new`` is the *tangent
    gradient list at the accepted point* (the last one ``derphi`` computed), which the outer
    loop re-uses instead of calling ``myfprime`` again (Sphere_Grad_Descent.py:336-340).
    """
    n_f, n_g = [0], [0]
    last_tangent = [None]

    def phi(alpha):
        n_f[0] += 1
        return f(_retract_all(X_k, alpha, d_k, M_0, inner_prod, args_IP, kwargs_IP), *args_f, **kwargs_f)

    def derphi(alpha):
        n_g[0] += 1
        X_new = _retract_all(X_k, alpha, d_k, M_0, inner_prod, args_IP, kwargs_IP)
        g_new = copy.deepcopy(g_k)
        grad = myfprime(X_new, *args_f, **kwargs_f)
        slope = 0.
        for i, _ in enumerate(M_0):
            g_new[i] = tangent_vector(X_new[i], grad[i], inner_prod, args_IP, kwargs_IP)
            moved_d = transport_vector(X_new[i], d_k[i], inner_prod, args_IP, kwargs_IP)
            slope += inner_prod(g_new[i], moved_d, *args_IP, **kwargs_IP)
        last_tangent[0] = g_new
        return slope

    derphi0 = _slope(g_k, d_k, M_0, inner_prod, args_IP, kwargs_IP)

    alpha_star, phi_star, old_fval, derphi_star = scalar_search_wolfe2(
        phi, derphi, old_fval, old_old_fval, derphi0, c1, c2, amax, None, maxiter=maxiter)

    if derphi_star is None:
        warn('The line search algorithm did not converge', LineSearchWarning)
    else:
        derphi_star = last_tangent[0]
    return alpha_star, n_f[0], n_g[0], phi_star, old_fval, derphi_star


# ----------------------------------------------------------------------------------
# driver
# ----------------------------------------------------------------------------------

def _dump_progress(R):
    """Rank-0 rewrite of DAL_PROGRESS.h5 with every field of the progress record; every failure is swallowed
    (Sphere_Grad_Descent.py:821-829).  Without h5py / mpi4py (both optional here) the same keys go to DAL_PROGRESS.npz."""
    try:
        if _MPI is not None and _MPI.COMM_WORLD.rank != 0:
            return
        rec = dict(vars(R))
        # vectors that live on the device (anything with a .numpy(), e.g. devvec.DeviceVector) are written as host arrays
        rec["X_opt"] = [x.numpy() if hasattr(x, "numpy") else x for x in rec["X_opt"]] if len(rec["X_opt"]) else rec["X_opt"]
        if _h5py is not None:
            with _h5py.File('DAL_PROGRESS.h5', 'w') as fh:
                for key, val in rec.items():
                    fh.create_dataset(key, data=val)
        else:
            np.savez('DAL_PROGRESS.npz', **{key: np.asarray(val) for key, val in rec.items()})
    except Exception:
        pass


def load_progress(path=None):
    """Read a progress record back (the manual restart of the reference scripts, FWD_Solve_SH23.py:787-800: `X_0 = DAL_file['X_opt']`):
    returns a dict of every field written by the optimiser ('X_opt', 'fun', 'nit', ...).  path: DAL_PROGRESS.h5 / .npz (default: whichever
    exists in the working directory, HDF5 first)."""
    import os
    if path is None:
        path = next((p for p in ('DAL_PROGRESS.h5', 'DAL_PROGRESS.npz') if os.path.exists(p)), None)
        if path is None:
            raise FileNotFoundError("no DAL_PROGRESS.h5 / DAL_PROGRESS.npz in %s" % os.getcwd())
    if path.endswith('.npz'):
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    if _h5py is None:
        raise RuntimeError("reading %s needs h5py" % path)
    with _h5py.File(path, 'r') as fh:
        return {k: fh[k][()] for k in fh.keys()}


def Optimise_On_Multi_Sphere(X_0, M_0, f, myfprime, inner_prod, args_f=(), args_IP=(), kwargs_f={},
                             kwargs_IP={}, err_tol=1e-06, max_iters=200, alpha_k=1., LS='LS_wolfe',
                             CG=True, callback=None, verbose=True):
    """Minimise ``f`` over the product of spheres <X_i,X_i> = M_0[i] (Sphere_Grad_Descent.py:692-838).

    X_0, M_0   : lists (one entry per norm constraint) of vectors / radii
    f          : f(X, *args_f, **kwargs_f) -> float         (the forward solve)
    myfprime   : myfprime(X, *args_f, **kwargs_f) -> list   (the adjoint solve; only valid right after
                 f at the same X)
    inner_prod : inner_prod(x, y, *args_IP, **kwargs_IP) -> float
    LS         : 'LS_wolfe' | 'LS_armijo';   CG : conjugate gradient (FR/PR clip) or steepest descent

    Returns ``(RESIDUAL, FUNCT, X_opt)``: per-component residual histories, -J_k history, final X.
    Appends the progress record to ``optimize_result.txt`` in the cwd, as the reference does.
    """
    use_wolfe = (LS == 'LS_wolfe') or (LS is LS_wolfe_multiple)
    use_armijo = (LS == 'LS_armijo') or (LS is LS_armijo_multiple)

    ncomp = len(M_0)
    error = np.ones(ncomp)
    func_evals = 0
    grad_evals = 0
    alpha_max = alpha_k
    RESIDUAL = [[] for _ in range(ncomp)]
    R = result(ncomp)
    log = open("optimize_result.txt", "a")

    J_k_old = None
    X_k = [x_i * np.sqrt(c_i / inner_prod(x_i, x_i, *args_IP, **kwargs_IP)) for x_i, c_i in zip(X_0, M_0)]
    J_k = f(X_k, *args_f, **kwargs_f)
    func_evals += 1

    g_new = None          # tangent gradient handed back by the Wolfe search
    g_km1 = d_k = None
    while (max(error) > err_tol) and (R.Iterations < max_iters):

        # gradient: from iteration 2 on, the Wolfe search already evaluated it at X_k
        if use_wolfe and (R.Iterations > 1):
            g_k = g_new
        else:
            euclid = myfprime(X_k, *args_f, **kwargs_f)
            g_k = [tangent_vector(x, gx, inner_prod, args_IP, kwargs_IP) for x, gx in zip(X_k, euclid)]
            grad_evals += 1

        # direction: steepest descent for the first two iterations, then (optionally) CG
        if (R.Iterations > 1) and (CG == True):
            beta_FR = 0.
            beta_PR = 0.
            moved_d = copy.deepcopy(g_k)
            for c, _ in enumerate(g_k):
                gg_old = inner_prod(g_km1[c], g_km1[c], *args_IP, **kwargs_IP)
                beta_FR += inner_prod(g_k[c], g_k[c], *args_IP, **kwargs_IP) / gg_old
                moved_g = transport_vector(X_k[c], g_km1[c], inner_prod, args_IP, kwargs_IP)
                beta_PR += (inner_prod(g_k[c], g_k[c], *args_IP, **kwargs_IP)
                            - inner_prod(g_k[c], moved_g, *args_IP, **kwargs_IP)) \
                    / inner_prod(g_km1[c], g_km1[c], *args_IP, **kwargs_IP)
                moved_d[c] = transport_vector(X_k[c], d_k[c], inner_prod, args_IP, kwargs_IP)
            beta = max(0., min(beta_FR, beta_PR))          # H. Sato (2021) hybrid rule
            d_k = [-1. * g + beta * t for g, t in zip(g_k, moved_d)]
        else:
            d_k = [-1. * g for g in g_k]

        # step size
        if (R.Iterations == 0) or use_armijo:
            alpha_k, n_f, J_k = LS_armijo_multiple(f, inner_prod, M_0, X_k, g_k, d_k, J_k,
                                                   args_f, args_IP, kwargs_f, kwargs_IP, alpha0=alpha_k)
            func_evals += n_f
        else:
            alpha_k, n_f, n_g, J_k, J_k_old, g_new = LS_wolfe_multiple(
                f, myfprime, inner_prod, M_0, X_k, g_k, d_k, J_k, J_k_old,
                args_f, args_IP, kwargs_f, kwargs_IP, amax=alpha_max)
            grad_evals += n_g
            func_evals += n_f

        # retract onto the spheres, record the residual of the gradient used this iteration
        for c, radius in enumerate(M_0):
            if alpha_k is None:
                print("\n Couldn't find a descent direction .... Terminating \n")
                log.close()
                return R.Residual, R.Function_Value, R.X_opt
            X_k[c] = Update_vector(X_k[c], alpha_k, d_k[c], radius, inner_prod, args_IP, kwargs_IP)
            error[c] = inner_prod(g_k[c], g_k[c], *args_IP, **kwargs_IP) ** 0.5

        R.X_opt = X_k
        R.Iterations += 1
        R.Function_Evals += func_evals
        R.Gradient_Evals += grad_evals
        for c, _ in enumerate(error):
            RESIDUAL[c].append(error[c])
        R.Residual = RESIDUAL
        R.Step_Size.append(alpha_k)
        R.Function_Value.append(-1. * J_k)

        g_km1 = copy.deepcopy(g_k)
        func_evals = 0
        grad_evals = 0

        if callback is not None:
            callback(R.Iterations)
        _dump_progress(R)

        if verbose:
            print(R, flush=True)
        log.write(str(R))
        log.write('\n')
        log.flush()

    log.close()
    return R.Residual, R.Function_Value, R.X_opt


def plot_optimisation(THETA, FUNCT, filename="Mix_DISC_SD_W.pdf", show=True):
    """Residual / objective history plot (Sphere_Grad_Descent.py:840-881). matplotlib is imported lazily."""
    import matplotlib.pyplot as plt

    fig, ax_J = plt.subplots(figsize=(8, 6))
    ax_r = ax_J.twinx()
    ax_J.plot(np.arange(len(FUNCT)), FUNCT, color='tab:red', linewidth=3, linestyle=':')
    styles = ['-.', '-']
    n_last = 1
    for c, r_k in enumerate(THETA):
        n_last = len(r_k)
        ax_r.semilogy(np.arange(n_last), r_k, color='tab:blue', linewidth=3,
                      label=r"c_%i" % c, linestyle=styles[c % len(styles)])
    ax_J.tick_params(axis='y', labelcolor='tab:red', labelsize=26)
    ax_J.tick_params(axis='x', labelsize=26)
    ax_J.set_ylabel(r'$|\hat{J}_k(\hat{X}_k)|$', color='tab:red', fontsize=26)
    ax_J.set_xlabel(r'Iteration $k$', fontsize=26)
    ax_J.set_xlim([0, max(n_last - 1, 1)])
    ax_r.tick_params(axis='y', labelcolor='tab:blue', labelsize=26)
    ax_r.set_ylabel(r'$r_k$', color='tab:blue', fontsize=26)
    ax_r.legend(fontsize=18)
    plt.grid()
    plt.tight_layout(pad=1, w_pad=1.5)
    fig.savefig(filename, dpi=1200)
    if show:
        plt.show()
    return None

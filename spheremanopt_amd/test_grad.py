"""Taylor-remainder gradient check — the acceptance harness of the hot path.

Counterpart of the reference's ``Adjoint_Gradient_Test`` (TestGrad.py:5-156): evaluates

    R1(eps) = |J(X + eps dX) - J(X)|                      -> O(eps)
    R2(eps) = |J(X + eps dX) - J(X) - eps <dX, grad J>|   -> O(eps^2)

for five halvings of ``eps`` and the log-slopes between consecutive levels.  The 5x5 table
``AA`` (rows: eps, R1, R2, slope1, slope2; last slope entries stay 0) is saved to
``eps_TestR_TestR2_h_h2.npy`` exactly like the reference (TestGrad.py:122-154) and — unlike
the reference, which returns None — also returned so tests can assert on it.
"""
import logging
import time

import numpy as np

__all__ = ["Adjoint_Gradient_Test", "taylor_table"]

_N_LEVELS = 5


def _as_list(v):
    return v if isinstance(v, list) else [v]


def taylor_table(X0, dX0, FWD_Solve, ADJ_Solve, Inner_Prod, args_f=(), args_IP=(), kwargs_f={},
                 kwargs_IP={}, epsilon=1e-04, logger=None):
    """Compute the AA table without touching the filesystem."""
    log = logger or logging.getLogger(__name__)

    # the reference unpacks the kwargs dicts with a single star (TestGrad.py:47) -> their *keys*
    # are passed positionally; keep that so a non-empty dict behaves the same way.
    t0 = time.time()
    J_ref = FWD_Solve(_as_list(X0), *args_f, *kwargs_f)
    print('Total time fwd: %f' % (time.time() - t0))

    t0 = time.time()
    dJdX = ADJ_Solve(_as_list(X0), *args_f, *kwargs_f)
    print('Total time adjoint: %f' % (time.time() - t0))

    if isinstance(dX0, list):
        W_ADJ = 0.
        for p, g in zip(dX0, dJdX):
            W_ADJ += Inner_Prod(p, g, *args_IP, *kwargs_IP)
    else:
        W_ADJ = Inner_Prod(dX0, dJdX[0], *args_IP, *kwargs_IP)

    AA = np.zeros((5, _N_LEVELS))
    for level in range(_N_LEVELS):
        if isinstance(X0, list) and isinstance(dX0, list):
            J_fd = FWD_Solve([x + epsilon * p for x, p in zip(X0, dX0)], *args_f, *kwargs_f)
        else:
            J_fd = FWD_Solve([X0 + epsilon * dX0], *args_f, *kwargs_f)
        AA[0, level] = epsilon
        AA[1, level] = abs(J_fd - J_ref)
        AA[2, level] = abs(J_fd - J_ref - epsilon * W_ADJ)
        log.info('epsilon = %e  |dJ| = %e  |dJ - eps*dJ.dX| = %e  |dJ.dX| = %e',
                 epsilon, AA[1, level], AA[2, level], W_ADJ)
        epsilon = 0.5 * epsilon

    for row_out, row_in in ((3, 1), (4, 2)):
        for i in range(_N_LEVELS - 1):
            AA[row_out, i] = np.log(AA[row_in, i] / AA[row_in, i + 1]) / np.log(AA[0, i] / AA[0, i + 1])
        log.info('Taylor remainder %d: mean exponent = %e', row_out - 2, AA[row_out, :-1].sum() / (_N_LEVELS - 1))
    return AA


def Adjoint_Gradient_Test(X0, dX0, FWD_Solve, ADJ_Solve, Inner_Prod, args_f=(), args_IP=(), kwargs_f={},
                          kwargs_IP={}, epsilon=1e-04):
    """Drop-in for TestGrad.py:5; also returns the AA table it saves."""
    for h in logging.root.handlers:
        h.setLevel("INFO")
    AA = taylor_table(X0, dX0, FWD_Solve, ADJ_Solve, Inner_Prod, args_f, args_IP, kwargs_f, kwargs_IP, epsilon)
    np.save("eps_TestR_TestR2_h_h2.npy", AA)
    return AA

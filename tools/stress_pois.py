#!/usr/bin/env python3
"""Repeatability of the Poiseuille device path: the same gradient evaluated `reps` times must be bit-identical (no atomics, fixed
reduction trees), for both formulations and both costs.  usage: python tools/stress_pois.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spheremanopt_amd import poiseuille as pz  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = 0
for cont, Nx, Nz in ((False, 96, 48), (True, 64, 32)):
    for s in (0, 1):
        dom = pz.PoiseuilleDomain(Nx, Nz, continuous=cont)
        gx, gz = dom.gshape
        X = 1e-2 * np.random.RandomState(3).standard_normal(2 * gx * gz)
        ctx = dom.context(500., 0.05, 60, 5e-3, s, 1., 0.3)
        ref = None
        for r in range(reps):
            J = ctx.forward([X]); g = ctx.adjoint(None, "Continuous" if cont else "Discrete")[0]
            if ref is None:
                ref = (J, g.copy())
            elif J != ref[0] or not np.array_equal(g, ref[1]):
                bad += 1
        print("continuous=%s s=%d: %d repetitions, J=%.15e, mismatches so far %d" % (cont, s, reps, ref[0], bad))
        dom.drop_contexts()
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Plane-Poiseuille at the reference's resolution (Nx, Nz = 384, 192 = 3/2 * (256, 128), T = 5, dt = 5e-3): build time of the tau operators,
time per gradient, kernel-class breakdown.  usage: python tools/prof_pois.py [Nx Nz n_iters s]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from spheremanopt_amd import poiseuille as pz  # noqa: E402

Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 384
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 192
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
s = int(sys.argv[4]) if len(sys.argv) > 4 else 1
t0 = time.perf_counter()
dom = pz.PoiseuilleDomain(Nx, Nz)
ctx = dom.context(500., 0.05, n, 5e-3, s, 1., 0.125)
t_build = time.perf_counter() - t0
rs = np.random.RandomState(0)
X = 1e-3 * rs.standard_normal(2 * Nx * Nz)
ctx.forward([X]); ctx.adjoint(None)
ctx.timing_enable(True)
t0 = time.perf_counter()
J = ctx.forward([X])
t_f = time.perf_counter() - t0
t0 = time.perf_counter()
g = ctx.adjoint(None)[0]
t_a = time.perf_counter() - t0
print(json.dumps({"Nx": Nx, "Nz": Nz, "n_iters": n, "s": s, "build_s": t_build, "forward_s": t_f, "adjoint_s": t_a, "J": J,
                  "gnorm": float(np.linalg.norm(g)), "stack_GB": ctx.stack_bytes / 1e9,
                  "kernels": [{"kernel": t["kernel"], "launches": t["launches"], "avg_us": 1e3 * t["total_ms"] / max(t["launches"], 1),
                               "total_ms": t["total_ms"]} for t in ctx.timing()]}))

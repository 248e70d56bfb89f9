#!/usr/bin/env python3
"""What the run-time-length kernels (csrc/kdyn_any.hpp) cost: gradient time of a tuned size through its own kernels and, with
SMO_KD_ANY=1, through the any-size ones; and of neighbouring sizes that only the any-size kernels take.
usage: python tools/time_any_size.py [n_iters]      -> one JSON line per size"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from spheremanopt_amd import kdyn  # noqa: E402
from spheremanopt_amd.devvec import DeviceVector, to_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def gradient_ms(N, force):
    os.environ["SMO_KD_ANY"] = "1" if force else "0"
    dom = kdyn.KDynDomain(N)
    ctx = dom.context(1., 1e-3, n, "Final")
    X = to_device([kdyn.synthetic_field(dom.G, 1), kdyn.synthetic_field(dom.G, 2)])
    g = [DeviceVector(ctx.vec_len), DeviceVector(ctx.vec_len)]
    J = ctx.forward_dev(X); ctx.adjoint_dev(X, g)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        J = ctx.forward_dev(X); ctx.adjoint_dev(X, g)
    ms = 1e3 * (time.perf_counter() - t0) / reps
    dom.drop_contexts()
    return ms, J


for N, tuned in ((32, True), (34, False), (64, True), (66, False), (62, False), (128, True), (130, False), (124, False), (148, False)):
    rec = {"npts": N, "G": 3 * N // 2, "n_iters": n}
    if tuned:
        rec["tuned_ms"], J0 = gradient_ms(N, False)
        rec["any_ms"], J1 = gradient_ms(N, True)
        rec["ratio"] = rec["any_ms"] / rec["tuned_ms"]
        rec["J_rel_diff"] = abs(J1 - J0) / abs(J0)
    else:
        rec["any_ms"], _ = gradient_ms(N, False)
    rec["any_ns_per_point_step"] = 1e6 * rec["any_ms"] / (n * (3 * N // 2) ** 3)
    print(json.dumps(rec), flush=True)

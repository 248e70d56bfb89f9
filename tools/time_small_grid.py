#!/usr/bin/env python3
"""Launch-bound sizes: gradient time with and without the captured HIP graphs.  usage: python tools/time_small_grid.py [npts ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


from spheremanopt_amd import kdyn  # noqa: E402
from spheremanopt_amd.devvec import DeviceVector, to_device  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [24, 32, 48, 64]
n = int(os.environ.get("SMO_TOOL_ITERS", "2000"))
for N in sizes:
    rec = {"npts": N, "n_iters": n}
    for mode in ("0", "1"):
        os.environ["SMO_KD_GRAPH"] = mode
        dom = kdyn.KDynDomain(N)
        ctx = dom.context(1., 5e-4, n, "Final")
        X = to_device([kdyn.synthetic_field(dom.G, 1), kdyn.synthetic_field(dom.G, 2)])
        g = [DeviceVector(ctx.vec_len), DeviceVector(ctx.vec_len)]
        t0 = time.perf_counter()
        J = ctx.forward_dev(X); ctx.adjoint_dev(X, g)
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(5):
            J = ctx.forward_dev(X); ctx.adjoint_dev(X, g)
        rec["graph" if mode == "1" else "launches"] = {"ms_per_gradient": 1e3 * (time.perf_counter() - t0) / 5, "first_call_ms": 1e3 * first, "J": J,
                                                        "replays": ctx.get(2)}
        dom.drop_contexts()
    rec["speedup"] = rec["launches"]["ms_per_gradient"] / rec["graph"]["ms_per_gradient"]
    rec["bit_identical_J"] = rec["launches"]["J"] == rec["graph"]["J"]
    print(json.dumps(rec), flush=True)

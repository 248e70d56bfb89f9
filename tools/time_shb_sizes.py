"""Gradients per second of the SHB23 path (one problem, the latency case) over grid lengths, with the cluster mode on and off:
    python tools/time_shb_sizes.py [--steps 2000] [--sizes 500,512,768,1000,1024] [--cost 0|1]
One JSON line per (N, mode): what the multi-workgroup cluster is worth where the operator no longer fits one CU's reach (N > 512: 8 MB, beyond
an XCD's L2) and for lengths without an instantiation."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spheremanopt_amd import _capi      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--sizes", default="256,500,512,768,1000,1023,1024")
    ap.add_argument("--cost", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    adj = "Continuous" if a.cost else "Discrete"
    rng = np.random.default_rng(0)
    for N in [int(s) for s in a.sizes.split(",")]:
        L = 2 * N if a.cost else N
        X = 1e-2 * rng.standard_normal(L)
        ref = None
        for mode in ("1", "0"):
            os.environ["SMO_SHB_CLUSTER"] = mode
            ctx = _capi.Context(_capi.SMO_SHB23, N, (-20., 20.), 1e-2, a.steps, -0.1, cost=a.cost)
            J = ctx.forward([X]); g = ctx.adjoint(None, adj)[0]          # warm-up
            t = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                J = ctx.forward([X]); g = ctx.adjoint(None, adj)[0]
                t.append(time.perf_counter() - t0)
            dt = min(t)
            if ref is None:
                ref = (J, g)
            print(json.dumps({"N": N, "formulation": adj, "steps": a.steps, "workgroups_per_problem": int(ctx.get(2)), "fallbacks": int(ctx.get(1)),
                              "grad_per_s": round(1.0 / dt, 2), "us_per_step_pair": round(dt / a.steps * 1e6, 2),
                              "J_rel_diff_to_cluster": abs(J - ref[0]) / abs(ref[0]),
                              "grad_rel_diff_to_cluster": float(np.linalg.norm(g - ref[1]) / np.linalg.norm(ref[1]))}), flush=True)
            del ctx
    os.environ.pop("SMO_SHB_CLUSTER", None)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Write the north-star-size oracle fixtures of the kinematic dynamo (tests/golden/oracle_kdyn_c4_128_n<steps>.npz,
oracle_kdyn_c5_256_n<steps>.npz): the CPU restatement (oracle/kdyn.py) run ONCE here, at the grids of BASELINE.json
configs[3] (128^3) and configs[4] (256^3), for every cost-function / adjoint-type combination, on the seeded synthetic
fields of SURVEY.md section 8d (B: seed 1, U: seed 2).  The GPU tests compare the HIP path (single GPU and slabs) with
these numbers; re-running the oracle inline would take 10-20 minutes.

Stored per (cost, adjoint) combination c: J_<cost>, and for the two gradients  <c>_gB / <c>_gU at `idx` (512 entries spread over
the vector), their 2-norms and sums; plus checksums of the last snapshot (sum, norm) and a strided sample of it.

Usage: python tools/gen_golden_kdyn_big.py --npts 128 --steps 50 [--workers 8]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle.kdyn import KDynOracle, synthetic_field                # noqa: E402


def sample_idx(n, k=512):
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npts", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--rm", type=float, default=1.0)
    ap.add_argument("--workers", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--name", default=None)
    a = ap.parse_args()
    N, n = a.npts, a.steps
    G = 3 * N // 2
    B = synthetic_field(G, 1)
    U = synthetic_field(G, 2)
    idx = sample_idx(B.size)
    out = {"N": N, "steps": n, "dt": a.dt, "Rm": a.rm, "idx": idx, "seeds": np.array([1, 2])}
    t00 = time.time()
    for cost in ("Final", "Integrated"):
        o = KDynOracle(N, Rm=a.rm, dt=a.dt, N_ITERS=n, Cost_function=cost, workers=a.workers)
        t = time.time()
        J = o.forward([B, U])
        out["J_%s" % cost] = J
        print("N=%d %s: J=%.15e (forward %.0f s)" % (N, cost, J, time.time() - t), flush=True)
        if cost == "Final":
            last = o.stack[..., n]
            out["snap_last_sum"] = last.sum()
            out["snap_last_norm"] = np.linalg.norm(last)
            out["snap_last_sample"] = last.reshape(-1)[::9973]
            mid = o.stack[..., n // 2]
            out["snap_mid_norm"] = np.linalg.norm(mid)
        for adj in ("Discrete", "Continuous"):
            t = time.time()
            gB, gU = o.adjoint([B, U], adj)
            key = "%s_%s" % (cost, adj)
            out[key + "_gB"] = gB[idx]; out[key + "_gU"] = gU[idx]
            out[key + "_gB_norm"] = np.linalg.norm(gB); out[key + "_gU_norm"] = np.linalg.norm(gU)
            out[key + "_gB_sum"] = gB.sum(); out[key + "_gU_sum"] = gU.sum()
            # a second, independent functional of the whole vector: its inner product with a fixed seeded direction
            rs = np.random.RandomState(77)
            w = rs.standard_normal(4096)
            out[key + "_gB_proj"] = float(np.dot(gB[:: max(1, gB.size // 4096)][:4096], w))
            out[key + "_gU_proj"] = float(np.dot(gU[:: max(1, gU.size // 4096)][:4096], w))
            print("  %s: |gB|=%.12e |gU|=%.12e (%.0f s)" % (key, out[key + "_gB_norm"], out[key + "_gU_norm"], time.time() - t), flush=True)
            del gB, gU
        del o
    name = a.name or ("oracle_kdyn_c%d_%d_n%d.npz" % (4 if N == 128 else (5 if N == 256 else 0), N, n))
    np.savez(os.path.join(OUT, name), **out)
    print("wrote %s (%.0f s)" % (name, time.time() - t00), flush=True)


if __name__ == "__main__":
    main()

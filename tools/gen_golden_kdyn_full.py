#!/usr/bin/env python3
"""BASELINE configs[3] at its FULL length: the kinematic-dynamo oracle at 128^3 for all 1000 steps (Rm = 1, dt = 1e-3, T = 1), seeded
synthetic fields of SURVEY 8d -> tests/golden/oracle_kdyn_c4_128_n1000.npz (J for both cost functionals, the discrete-adjoint gradients of
both: 512 sampled entries, norms, sums, a seeded projection; checksums of the trajectory).  Needs ~55 GB of RAM (the 49.6 GB snapshot
stack) and about two hours on 6 cores; run once, in the build container.

Usage: python tools/gen_golden_kdyn_full.py [--workers 6]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npts", type=int, default=128)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--name", default=None)
    a = ap.parse_args()
    N, n = a.npts, a.steps
    G = 3 * N // 2
    B, U = synthetic_field(G, 1), synthetic_field(G, 2)
    idx = np.unique(np.linspace(0, B.size - 1, 512).astype(np.int64))
    out = {"N": N, "steps": n, "dt": 1e-3, "Rm": 1.0, "idx": idx, "seeds": np.array([1, 2])}
    o = KDynOracle(N, Rm=1.0, dt=1e-3, N_ITERS=n, Cost_function="Integrated", workers=a.workers)
    t = time.time()
    out["J_Integrated"] = o.forward([B, U])
    w = np.where(o.K[0] == 0, 1., 2.)
    out["J_Final"] = -float((w * np.abs(o.stack[..., n]) ** 2).sum())          # grid mean of |B_N|^2 by Parseval on the half spectrum
    print("forward: J_Final %.15e  J_Integrated %.15e  (%.0f s)" % (out["J_Final"], out["J_Integrated"], time.time() - t), flush=True)
    for k in (n // 4, n // 2, n):
        out["snap_%d_norm" % k] = np.linalg.norm(o.stack[..., k])
        out["snap_%d_sample" % k] = o.stack[..., k].reshape(-1)[::9973].copy()
    rs = np.random.RandomState(77)
    wv = rs.standard_normal(4096)
    for cost in ("Final", "Integrated"):
        o.cost = cost
        t = time.time()
        gB, gU = o.adjoint([B, U], "Discrete")
        key = "%s_Discrete" % cost
        for name, g in (("gB", gB), ("gU", gU)):
            out["%s_%s" % (key, name)] = g[idx]
            out["%s_%s_norm" % (key, name)] = np.linalg.norm(g)
            out["%s_%s_sum" % (key, name)] = g.sum()
            out["%s_%s_proj" % (key, name)] = float(np.dot(g[:: max(1, g.size // 4096)][:4096], wv))
        print("%s: |gB| %.12e |gU| %.12e (%.0f s)" % (key, out[key + "_gB_norm"], out[key + "_gU_norm"], time.time() - t), flush=True)
        np.savez(os.path.join(OUT, a.name or "oracle_kdyn_c4_%d_n%d.npz" % (N, n)), **out)      # after each adjoint: a partial result survives
    print("done", flush=True)


if __name__ == "__main__":
    main()

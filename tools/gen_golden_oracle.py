#!/usr/bin/env python3
"""Write tests/golden/oracle_*.npz: outputs of the CPU restatements (oracle/) on the seeded synthetic
inputs of SURVEY.md section 8d.  These pin the oracle against regressions and give the GPU tests
expected values at sizes where running the oracle inline would take too long.

Usage: python tools/gen_golden_oracle.py [--kdyn-n 32]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle.sh23 import SH23Oracle, synthetic_ic as sh_ic          # noqa: E402
from oracle.kdyn import KDynOracle, synthetic_field                # noqa: E402
from oracle import shb23                                           # noqa: E402
from oracle.poiseuille import PoiseuilleOracle, synthetic_ic as pois_ic   # noqa: E402


def sample_idx(n, k=64):
    return np.unique(np.linspace(0, n - 1, k).astype(int))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kdyn-n", type=int, default=32)
    ap.add_argument("--kdyn-steps", type=int, default=20)
    a = ap.parse_args()

    # C2: SH23 Npts=256, T=50, dt=0.1
    o = SH23Oracle(256, dt=0.1, N_ITERS=500)
    X = sh_ic(512, 42, 0.0725)
    J = o.forward([X]); g = o.adjoint([X])[0]; gc = o.adjoint([X], "Continuous")[0]
    np.savez(os.path.join(OUT, "oracle_sh23_c2.npz"), J=J, grad=g, grad_cont=gc,
             stack_last=o.stack[:, -1], stack_sum=o.stack.sum(axis=1))
    print("SH23 C2: J=%.15e |g|=%.6e" % (J, np.linalg.norm(g)))

    # C3: SHB23 N=512, dt=0.01, T=20
    s = shb23.SHB23Oracle(512, dt=1e-2, N_ITERS=2000)
    X = shb23.synthetic_ic(s, 42, 0.0019)
    J = s.forward([X]); g = s.adjoint([X])[0]
    np.savez(os.path.join(OUT, "oracle_shb23_c3.npz"), X=X, J=J, grad=g, stack_last=s.stack[:, -1],
             S_sample=s.S[::37, ::41], S_norm=np.linalg.norm(s.S))
    print("SHB23 C3: J=%.15e |g|=%.6e" % (J, np.linalg.norm(g)))

    # Poiseuille (Discrete): 96 x 48, 40 steps, both cost functionals
    for sw in (0, 1):
        p = PoiseuilleOracle(96, 48, dt=5e-3, N_ITERS=40, s=sw, delta=0.3)
        X = pois_ic(p, 42)
        J = p.forward([X]); g = p.adjoint([X])[0]
        np.savez(os.path.join(OUT, "oracle_poiseuille_96x48_s%d.npz" % sw), X=X, J=J, grad=g, u_last=p.stack[0, :p.ax, :, -1],
                 b_last=p.stack[2, :p.ax, :, -1])
        print("Poiseuille s=%d: J=%.15e |g|=%.6e" % (sw, J, np.linalg.norm(g)))

    # KDyn: reduced grid, both cost functions
    N, steps = a.kdyn_n, a.kdyn_steps
    for cost in ("Final", "Integrated"):
        k = KDynOracle(N, Rm=1., dt=1e-3, N_ITERS=steps, Cost_function=cost)
        B = synthetic_field(k.G, 1); U = synthetic_field(k.G, 2)
        t = time.time()
        J = k.forward([B, U]); gB, gU = k.adjoint([B, U])
        idx = sample_idx(gB.size, 512)
        np.savez(os.path.join(OUT, "oracle_kdyn_n%d_%s.npz" % (N, cost.lower())), N=N, steps=steps, J=J, idx=idx,
                 gB=gB[idx], gU=gU[idx], gB_norm=np.linalg.norm(gB), gU_norm=np.linalg.norm(gU),
                 gB_sum=gB.sum(), gU_sum=gU.sum(), stack_last_sum=k.stack[..., -1].sum())
        print("KDyn N=%d %s: J=%.15e |gB|=%.6e |gU|=%.6e (%.1fs)" % (N, cost, J, np.linalg.norm(gB), np.linalg.norm(gU), time.time() - t))


if __name__ == "__main__":
    main()

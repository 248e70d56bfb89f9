#!/usr/bin/env python3
"""Oracle run of the plane-Poiseuille path at the reference script's resolution and length — Nx x Nz = 384 x 192 (3/2 * (256, 128)),
T = 5, dt = 5e-3 (1000 steps), Re = 500, Ri = 0.05, mix-norm cost, Discrete formulation — on the input bench.py's Poiseuille line uses
(1e-3 * RandomState(42).standard_normal(2 Nx Nz)).  Writes tests/golden/oracle_poiseuille_384x192_n<steps>_s<s>.npz:
J, norm / 512 samples / seeded projection of the gradient, norm and samples of the last density and velocity snapshots.
usage: python tools/gen_golden_poiseuille_full.py [--steps 1000] [--s 1]      (about 8 GB of RAM; tens of minutes)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.poiseuille import PoiseuilleOracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--s", type=int, default=1)
    ap.add_argument("--nx", type=int, default=384)
    ap.add_argument("--nz", type=int, default=192)
    a = ap.parse_args()
    Nx, Nz, n = a.nx, a.nz, a.steps
    t0 = time.time()
    o = PoiseuilleOracle(Nx, Nz, Re=500., Ri=0.05, dt=5e-3, N_ITERS=n, s=a.s, Prandtl=1., delta=0.125)
    X = 1e-3 * np.random.RandomState(42).standard_normal(2 * Nx * Nz)
    J = o.forward([X])
    print("forward: J = %.15e (%.0f s)" % (J, time.time() - t0), flush=True)
    g = o.adjoint([X])[0]
    print("adjoint: |g| = %.12e (%.0f s)" % (np.linalg.norm(g), time.time() - t0), flush=True)
    idx = np.unique(np.linspace(0, g.size - 1, 512).astype(int))
    w = np.random.RandomState(77).standard_normal(g.size)
    u_last, b_last = o.stack[0, :o.ax, :, -1], o.stack[2, :o.ax, :, -1]
    b_prev = o.stack[2, :o.ax, :, -2]
    out = os.path.join(ROOT, "tests", "golden", "oracle_poiseuille_%dx%d_n%d_s%d.npz" % (Nx, Nz, n, a.s))
    np.savez(out, Nx=Nx, Nz=Nz, steps=n, s=a.s, seed=42, amplitude=1e-3, J=J, idx=idx, grad=g[idx], grad_norm=np.linalg.norm(g), grad_proj=float(np.dot(g, w)),
             u_last_norm=np.linalg.norm(u_last), u_last_sample=u_last.ravel()[::97], b_last_norm=np.linalg.norm(b_last), b_last_sample=b_last.ravel()[::97],
             b_prev_norm=np.linalg.norm(b_prev), b_prev_sample=b_prev.ravel()[::97])
    print("wrote", out, "(%.0f s)" % (time.time() - t0))


if __name__ == "__main__":
    main()

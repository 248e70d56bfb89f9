#!/usr/bin/env python3
"""Phase-level path (SlabKDyn, world = 1) against the monolithic path, repeated; reports bitwise differences."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spheremanopt_amd import kdyn  # noqa: E402
from spheremanopt_amd.kdyn_slab import SlabKDyn  # noqa: E402

N, n, reps = 32, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 20
dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
buf = kdyn.GEN_BUFFER(N, dom, n)
args = [dom, 1., 1e-3, n, n, buf, "Final", "Discrete"]
J0 = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
g0 = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
s = SlabKDyn(N, 1., 1e-3, n, "Final")
bad = 0
for r in range(reps):
    if r % 2 == 1:
        s = SlabKDyn(N, 1., 1e-3, n, "Final")          # fresh solver every other repetition
    J1 = s.forward([s.local_slab(B), s.local_slab(U)])
    snaps = [s.ops.snapshot(i) for i in range(n + 1)]
    g1 = s.adjoint("Discrete")
    g1 = [g.cpu().numpy() for g in g1]
    msgs = []
    if J1 != J0:
        msgs.append("J diff %.3e" % abs(J1 - J0))
    for i in range(n + 1):
        ref = dom.context(1., 1e-3, n, "Final").snapshot(i)
        d = np.abs(snaps[i] - ref).max()
        if d != 0:
            msgs.append("snap[%d] maxdiff %.3e n_diff %d" % (i, d, int((snaps[i] != ref).sum())))
            break
    for c in range(2):
        d = np.abs(g1[c] - g0[c]).max()
        if d != 0:
            msgs.append("grad[%d] maxdiff %.3e (rel %.1e) n_diff %d first %d" % (c, d, d / np.abs(g0[c]).max(), int((g1[c] != g0[c]).sum()), int(np.argmax(g1[c] != g0[c]))))
    print("rep", r, "ok" if not msgs else "DIFF", flush=True)
    if msgs:
        bad += 1
        print("rep", r, "; ".join(msgs))
print("%d of %d repetitions differ" % (bad, reps), flush=True)

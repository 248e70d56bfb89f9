#!/usr/bin/env python3
"""Run the same KDyn gradient repeatedly and report any bitwise difference (race detector)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spheremanopt_amd import kdyn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
buf = kdyn.GEN_BUFFER(N, dom, n)
args = [dom, 1., 1e-3, n, n, buf, "Final", "Discrete"]
ref = None
bad = 0
for r in range(reps):
    J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    snaps = [np.stack([buf[k][:, :, :, i] for k in ('A_fwd', 'B_fwd', 'C_fwd')]) for i in range(n + 1)]
    g = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
    cur = (J, snaps, g)
    if ref is None:
        ref = cur
        continue
    msgs = []
    if J != ref[0]:
        msgs.append("J diff %.3e" % abs(J - ref[0]))
    for i in range(n + 1):
        d = np.abs(snaps[i] - ref[1][i]).max()
        if d != 0:
            msgs.append("snap[%d] maxdiff %.3e (rel %.1e) at %s" % (i, d, d / np.abs(ref[1][i]).max(), np.unravel_index(np.abs(snaps[i] - ref[1][i]).argmax(), snaps[i].shape)))
            break
    for c in range(2):
        d = np.abs(g[c] - ref[2][c]).max()
        if d != 0:
            msgs.append("grad[%d] maxdiff %.3e (rel %.1e) n_diff %d" % (c, d, d / np.abs(ref[2][c]).max(), int((g[c] != ref[2][c]).sum())))
    if msgs:
        bad += 1
        print("rep", r, "; ".join(msgs))
print("N=%d n=%d: %d of %d repetitions differ from the first" % (N, n, bad, reps - 1))

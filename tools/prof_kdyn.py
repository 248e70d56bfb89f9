#!/usr/bin/env python3
"""Short KDyn run for rocprofv3 (kernel trace or PMC passes): a few forward+adjoint steps at the bench grid.
usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/prof_kdyn.py [npts] [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spheremanopt_amd import kdyn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dom = kdyn.KDynDomain(N)
G = dom.G
rs = np.random.RandomState(0)
B = rs.standard_normal(3 * G ** 3); U = rs.standard_normal(3 * G ** 3)      # cheap inputs: traffic does not depend on the data
buf = kdyn.GEN_BUFFER(N, dom, n)
args = [dom, 1., 1e-3, n, n, buf, "Final", "Discrete"]
J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
g = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
print("J", J, "|gB|", float(np.linalg.norm(g[0])))

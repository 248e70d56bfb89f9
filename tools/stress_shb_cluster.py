#!/usr/bin/env python3
"""Repeat the SHB23 config-3 gradient in cluster mode; every repetition must be bitwise identical to the first and match
the committed oracle output (detects stale reads in the cross-CU all-gather)."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from spheremanopt_amd import shb23  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
gold = np.load(os.path.join(ROOT, "tests", "golden", "oracle_shb23_c3.npz"))
dom = shb23.SHBDomain(512)
buf = shb23.GEN_BUFFER(512, dom, 2000)
X = gold["X"]
ref = None
bad = 0
for r in range(reps):
    J = shb23.FWD_Solve([X], dom, buf, 2000)
    g = shb23.ADJ_Solve([X], dom, buf, 2000)[0]
    if ref is None:
        ref = (J, g)
        print("rel err vs oracle: J %.2e grad %.2e" % (abs(J - gold["J"]) / abs(gold["J"]), np.linalg.norm(g - gold["grad"]) / np.linalg.norm(gold["grad"])))
    elif J != ref[0] or not np.array_equal(g, ref[1]):
        bad += 1
        print("rep", r, "differs: dJ %.3e dgrad %.3e" % (abs(J - ref[0]), np.abs(g - ref[1]).max()))
print("%d of %d repetitions differ" % (bad, reps - 1))

#!/usr/bin/env python3
"""Wall-time breakdown of the drop-in kinematic-dynamo script at BASELINE config 4's size (128^3, 1000 steps, 3 CG/Wolfe iterations) with
host (NumPy) vectors and with device-resident vectors.  usage: python tools/time_example_kdyn.py [npts] [steps] [max_iters]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
os.chdir(tempfile.mkdtemp())
from spheremanopt_amd import _capi  # noqa: E402
from spheremanopt_amd.examples import kdyn_optimise  # noqa: E402

t0 = time.perf_counter()
_capi.Context(_capi.SMO_KDYN, N, (0., 2 * np.pi), 1e-3, steps, 1.0).close()
print("create + destroy one %d^3 x %d-step context: %.2f s" % (N, steps, time.perf_counter() - t0), flush=True)
res = {}
for mode in ("host", "device"):
    argv = ["--npts", str(N), "--dt", "1e-3", "--steps", str(steps), "--max-iters", str(iters), "--quiet"] + (["--device-vectors"] if mode == "device" else [])
    t0 = time.perf_counter()
    R, F, X, _ = kdyn_optimise.main(argv)
    res[mode] = (time.perf_counter() - t0, R, F)
    print("%s vectors: %.2f s wall, FUNCT %r" % (mode, res[mode][0], F), flush=True)
print("identical RESIDUAL / FUNCT sequences:", res["host"][1] == res["device"][1] and res["host"][2] == res["device"][2])

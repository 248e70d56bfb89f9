#!/usr/bin/env python3
"""Context create/destroy loop: the free HBM reported by the runtime must come back after every destroy (no leak of device buffers,
streams or events).  usage: python tools/stress_leak.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402  (only for mem_get_info)

from spheremanopt_amd import _capi  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
free0 = None
for r in range(reps):
    ctxs = [_capi.Context(_capi.SMO_KDYN, 64, (0., 2 * np.pi), 1e-3, 50, 1.0, cost="Final"),
            _capi.Context(_capi.SMO_SH23, 256, (0., 12 * np.pi), 0.1, 100, -0.3, batch=8),
            _capi.Context(_capi.SMO_SHB23, 256, (-20., 20.), 1e-2, 100, -0.1),
            _capi.Context(_capi.SMO_POIS, 48, (0., 4 * np.pi), 5e-3, 20, 500., cost=0, npts2=36, param2=0.05, param3=1., param4=0.3)]
    G = 96
    X = np.random.RandomState(r).standard_normal(3 * G ** 3)
    ctxs[0].timing_enable(True)
    ctxs[0].forward([X, X]); ctxs[0].adjoint(None)
    for c in ctxs:
        c.close()
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info()[0]
    if free0 is None:
        free0 = free
    print("rep %d: free HBM %.3f GB (first rep %.3f GB)" % (r, free / 1e9, free0 / 1e9), flush=True)
if free0 - free > 64e6:
    print("LEAK: %.1f MB not returned" % ((free0 - free) / 1e6))
    sys.exit(1)
print("no leak")

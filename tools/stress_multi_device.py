#!/usr/bin/env python3
"""Race detector for the multi-device context (smo_create_multi): the same gradient many times through ONE context whose worker threads share
the box's GPU (devices 0,0,...); every repetition — and a fresh context — must reproduce J and both gradients bit for bit, with and without the
chunk pipeline and with both pull implementations.   usage: stress_multi_device.py [NPTS] [ITERS] [REPS] [NDEV]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spheremanopt_amd import _capi, kdyn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W = int(sys.argv[4]) if len(sys.argv) > 4 else 4
G = 3 * N // 2
B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
ref, bad, total = None, 0, 0
for chunks in (1, 2):
    for pull in ("kernel", "memcpy"):
        os.environ["SMO_SLAB_CHUNKS"], os.environ["SMO_PEER_COPY"] = str(chunks), pull
        for fresh in range(2):
            ctx = _capi.MultiContext(N, (0., 2. * np.pi), 1e-3, n, 1.0, [0] * W, cost="Integrated")
            for r in range(reps):
                J = ctx.forward([B, U]); g = ctx.adjoint(None)
                cur = (J, g[0].copy(), g[1].copy())
                total += 1
                if ref is None:
                    ref = cur
                    continue
                if not (cur[0] == ref[0] and np.array_equal(cur[1], ref[1]) and np.array_equal(cur[2], ref[2])):
                    bad += 1
                    print("chunks %d pull %s context %d rep %d: J diff %.3e, grad diffs %d / %d entries" % (
                        chunks, pull, fresh, r, abs(cur[0] - ref[0]), int((cur[1] != ref[1]).sum()), int((cur[2] != ref[2]).sum())))
            ctx.close()
print("N=%d n=%d W=%d: %d of %d gradient evaluations differ from the first" % (N, n, W, bad, total - 1))

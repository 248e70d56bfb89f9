#!/usr/bin/env python3
"""Where the host-buffer entry points spend their extra time: smo_forward / smo_adjoint on pinned host vectors against smo_forward_dev /
smo_adjoint_dev on vectors in HBM, each half timed on its own (KDyn, a short run so that the copies stand out).   usage: time_host_vectors.py [NPTS] [ITERS]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spheremanopt_amd import _capi, kdyn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
ctx = dom.context(1.0, 1e-3, n, "Final")
Bd, Ud = torch.from_numpy(B).cuda(), torch.from_numpy(U).cuda()
gB, gU = torch.empty_like(Bd), torch.empty_like(Ud)
hX = [_capi.pinned_copy(B), _capi.pinned_copy(U)]
hG = [_capi.pinned_empty(B.size), _capi.pinned_empty(U.size)]


def t(f, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best


fd = t(lambda: ctx.forward_dev([Bd, Ud])); ad = t(lambda: ctx.adjoint_dev([Bd, Ud], [gB, gU]))
fh = t(lambda: ctx.forward(hX)); ah = t(lambda: ctx.adjoint(None, out=hG))
pageable = [np.array(B), np.array(U)]
fp = t(lambda: ctx.forward(pageable)); ap = t(lambda: ctx.adjoint(None))
mb = B.nbytes / 1e6
print("N = %d, %d steps, vectors 2 x %.0f MB each way" % (N, n, mb))
print("forward : device %.2f ms | pinned host %.2f ms (+%.2f ms = %.1f GB/s for the H2D of X) | pageable host %.2f ms" % (fd, fh, fh - fd, 2 * mb / (fh - fd), fp))
print("adjoint : device %.2f ms | pinned host %.2f ms (+%.2f ms = %.1f GB/s for the D2H of grad J) | fresh NumPy arrays %.2f ms" % (ad, ah, ah - ad, 2 * mb / (ah - ad), ap))

#!/usr/bin/env python3
"""Compute term of the multi-GPU scaling model, measured on ONE GPU through the REAL in-library loop: for W in 1, 2, 4, 8 create the rank-0
context of a W-way slab decomposition, give it the null transport (smo_comm_set_transport with three NULLs: every exchange and reduction
returns at once — the data is meaningless, the kernels' work, launch sequence and access pattern are exactly those of a real rank) and time
smo_forward_dev + smo_adjoint_dev.  Two runs per W: one with HIP events on every launch (per-kernel averages), one without (wall time per
step pair, the figure the model uses).

    python tools/prof_slab_geometry.py NPTS [ITERS] [W,W,...] [CHUNKS]     -> one JSON line per W (profiles/r03_slab_geometry_<NPTS>.jsonl)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spheremanopt_amd import _capi  # noqa: E402


def run(N, W, n_iters, chunks):
    G = 3 * N // 2
    ctx = _capi.Context(_capi.SMO_KDYN, N, (0., 2. * np.pi), 1e-3, n_iters, 1.0, rank=0, world=W, ckpt=1)
    if W > 1:
        if chunks:
            os.environ["SMO_SLAB_CHUNKS"] = str(chunks)
        _capi._check(_capi.lib().smo_comm_set_transport(ctx._h, _capi.ALLTOALL_FN(), _capi.ALLREDUCE_FN(), None))
    n = ctx.vec_len
    X = [torch.randn(n, dtype=torch.float64, device="cuda") * 1e-3 for _ in range(2)]
    Gd = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()

    def gradient():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.forward_dev(X); tf = time.perf_counter() - t0
        ctx.adjoint_dev(X, Gd); ta = time.perf_counter() - t0 - tf
        return tf, ta
    gradient()                                                     # warm-up
    ctx.timing_enable(True)
    gradient()
    ker = {t["kernel"]: round(1e3 * t["total_ms"] / max(t["launches"], 1), 2) for t in ctx.timing() if t["launches"]}
    ctx.timing_enable(False)
    tf, ta = min(gradient(), gradient(), key=sum)
    K = int(ctx.comm_get(0)) if W > 1 else 1
    groups = int(ctx.comm_get(1)) if W > 1 else 0
    elems = 3 * (N // 2 // W) * (N - 1) * G
    r = {"N": N, "W": W, "chunks": K, "loop": "in-library (null transport)", "kept_grid_states": ctx.get(1) > 0, "ty_layout": int(ctx.get(3)),
         "fwd_us_per_step": 1e6 * tf / n_iters, "adj_us_per_step": 1e6 * ta / n_iters, "pair_us": 1e6 * (tf + ta) / n_iters, "kernel_avg_us": ker,
         "exchange_MB_sent_per_rank_per_pair": groups * elems * 16 / 1e6 * (W - 1) / W}
    ctx.close()
    return r


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    Ws = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else (1, 2, 4, 8)
    chunks = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    for W in Ws:
        print(json.dumps(run(N, W, n_iters, chunks)), flush=True)
        torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Per-rank kernel time of the slab geometry on ONE GPU: for W in 1,2,4,8 create the rank-0 context of a W-way decomposition, run
the forward / adjoint phases back to back WITHOUT the exchanges (the data is meaningless, the kernels' work and access pattern are
exactly those of a real rank) and print the time per step pair.  This is the compute term of the scaling model in DESIGN.md."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from spheremanopt_amd import kdyn_slab as ks  # noqa: E402


def run(N, W, n_iters, keep):
    import os
    os.environ["SMO_KD_TYSTACK"] = "1" if keep else "0"
    ops = ks.HipOps(N, 1.0, 1e-3, n_iters, "Final", 0, 0, W)
    bz = torch.zeros(4 * ops.elems, dtype=torch.float64, device="cuda")
    by = torch.zeros_like(bz) if W > 1 else bz
    ops.set_buffers(bz, by)
    vec = torch.randn(ops.vec_len, dtype=torch.float64, device="cuda") * 1e-3
    for code, tgt in ((ks.G2C_A, None), (ks.G2C_C, 1), (ks.C2G_A, 1), (ks.C2G_B, None), (ks.G2C_A, None), (ks.G2C_C, 0)):
        if code in (ks.G2C_A,):
            ops.phase(code, vec=vec)
        elif code == ks.C2G_B:
            ops.phase(code)
        else:
            ops.phase(code, tgt)
    ops.sync()
    ops.ctx.timing_enable(True)
    t0 = time.perf_counter()
    for n in range(n_iters):
        ops.phase(ks.FWD_A, n); ops.phase(ks.FWD_B, n); ops.phase(ks.FWD_C, n)
    ops.sync()
    t_f = time.perf_counter() - t0
    ops.phase(ks.ADJ_INIT, 0)
    t0 = time.perf_counter()
    for idx in range(n_iters - 1, -1, -1):
        ops.phase(ks.ADJ_A, idx); ops.phase(ks.ADJ_B, idx); ops.phase(ks.ADJ_C, idx)
    ops.sync()
    t_a = time.perf_counter() - t0
    tim = ops.ctx.timing()
    ker = {t["kernel"]: round(1e3 * t["total_ms"] / max(t["launches"], 1), 2) for t in tim if t["launches"]}
    groups = 3 + (1 if ops.keeps_grid_states else 2)     # fwd 1+1, adj (1 or 2)+1
    return {"N": N, "W": W, "kept_grid_states": bool(ops.keeps_grid_states), "fwd_us_per_step": 1e6 * t_f / n_iters,
            "adj_us_per_step": 1e6 * t_a / n_iters, "pair_us": 1e6 * (t_f + t_a) / n_iters, "kernel_avg_us": ker,
            "exchange_MB_sent_per_rank_per_pair": groups * ops.elems * 16 / 1e6 * (W - 1) / W}


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    Ws = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else (1, 2, 4, 8)
    for W in Ws:
        r = run(N, W, n_iters, keep=True)
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()

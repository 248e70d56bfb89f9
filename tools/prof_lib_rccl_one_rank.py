#!/usr/bin/env python3
"""Fixed cost of the IN-LIBRARY multi-GPU time loop (smo_comm_init + grouped ncclSend/ncclRecv inside smo_forward_dev /
smo_adjoint_dev), measured on ONE GPU: a one-rank RCCL communicator with SMO_SLAB_FORCE_EXCHANGE=1 (every transpose is a
self-exchange through RCCL on the solver's streams) against the monolithic loop on the same GPU, and against the Python/torch loop of
kdyn_slab.SlabKDyn (round 1's path).  The per-exchange difference = RCCL launch + the extra HBM copy of the self-exchange (on xGMI
that copy is the transfer itself).  usage: python tools/prof_lib_rccl_one_rank.py [npts] [n_iters] [chunks...]"""
import json
import os
import sys
import time

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from spheremanopt_amd import kdyn  # noqa: E402
from spheremanopt_amd.kdyn_slab import LibSlabKDyn, SlabKDyn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
chunk_list = [int(c) for c in sys.argv[3:]] or [1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
G = 3 * N // 2
B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)


def timed(fwd, adj):
    fwd(); adj()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    J = fwd(); adj()
    torch.cuda.synchronize()
    return J, time.perf_counter() - t0


os.environ["SMO_SLAB_FORCE_EXCHANGE"] = "0"
dom = kdyn.KDynDomain(N)
ctx = dom.context(1., 1e-3, n, "Final")
Bd, Ud = torch.from_numpy(B).cuda(), torch.from_numpy(U).cuda()
g = [torch.empty_like(Bd), torch.empty_like(Ud)]
J0, t_mono = timed(lambda: ctx.forward_dev([Bd, Ud]), lambda: ctx.adjoint_dev([Bd, Ud], g))
dom.drop_contexts()
os.environ["SMO_SLAB_FORCE_EXCHANGE"] = "1"
for K in chunk_list:
    os.environ["SMO_SLAB_CHUNKS"] = str(K)
    s = LibSlabKDyn(N, 1., 1e-3, n, "Final")
    out = [torch.empty_like(Bd), torch.empty_like(Ud)]
    J1, t_lib = timed(lambda: s.forward([Bd, Ud]), lambda: s.adjoint("Discrete", out))
    nex = s.exchanges_per_step_pair * n
    mb = 3 * (N // 2) * (N - 1) * G * 16 / 1e6
    rec = {"loop": "in-library (RCCL send/recv inside libsmo)", "npts": N, "n_iters": n, "chunks": s.K, "monolithic_s": t_mono, "one_rank_s": t_lib,
           "exchanges": nex, "overhead_us_per_exchange": 1e6 * (t_lib - t_mono) / nex, "MB_per_exchange": mb,
           "self_copy_us_at_5TBps": 2 * mb / 5.0, "J_equal": J0 == J1,
           "grad_equal": bool(torch.equal(out[0], g[0]) and torch.equal(out[1], g[1]))}
    print(json.dumps(rec), flush=True)
    del s
    if K == 1:
        p = SlabKDyn(N, 1., 1e-3, n, "Final", chunks=1)
        Bl, Ul = p.local_slab(B), p.local_slab(U)
        o2 = [torch.empty_like(Bl), torch.empty_like(Ul)]
        J2, t_py = timed(lambda: p.forward([Bl, Ul]), lambda: p.adjoint("Discrete", o2))
        print(json.dumps({"loop": "Python/torch (all_to_all_single per exchange, round 1)", "npts": N, "n_iters": n, "chunks": 1, "one_rank_s": t_py,
                          "overhead_us_per_exchange": 1e6 * (t_py - t_mono) / ((3 + p.adj_groups) * n), "J_equal": J0 == J2}), flush=True)
        del p
dist.destroy_process_group()

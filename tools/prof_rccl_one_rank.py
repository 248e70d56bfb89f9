#!/usr/bin/env python3
"""Overhead of the multi-GPU time loop's host/collective machinery, measured on ONE GPU: the slab driver with a one-rank RCCL process
group and SMO_SLAB_FORCE_EXCHANGE=1 (every transpose is a self-copy through all_to_all_single) against the monolithic C++ loop.
The difference per exchange is the fixed cost (torch dispatch + RCCL kernel launch + the extra HBM copy) that every rank of a real
multi-GPU run pays on top of the xGMI transfer.  usage: python tools/prof_rccl_one_rank.py [npts] [n_iters]"""
import json
import os
import sys
import time

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", SMO_SLAB_FORCE_EXCHANGE="1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from spheremanopt_amd import kdyn  # noqa: E402
from spheremanopt_amd.kdyn_slab import SlabKDyn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
G = 3 * N // 2
B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
dom = kdyn.KDynDomain(N)
ctx = dom.context(1., 1e-3, n, "Final")
Bd, Ud = torch.from_numpy(B).cuda(), torch.from_numpy(U).cuda()
g = [torch.empty_like(Bd), torch.empty_like(Ud)]
ctx.forward_dev([Bd, Ud]); ctx.adjoint_dev([Bd, Ud], g)
torch.cuda.synchronize(); t0 = time.perf_counter()
J0 = ctx.forward_dev([Bd, Ud]); ctx.adjoint_dev([Bd, Ud], g)
torch.cuda.synchronize(); t_mono = time.perf_counter() - t0
dom.drop_contexts()
s = SlabKDyn(N, 1., 1e-3, n, "Final")
Bl, Ul = s.local_slab(B), s.local_slab(U)
out = [torch.empty_like(Bl), torch.empty_like(Ul)]
s.forward([Bl, Ul]); s.adjoint("Discrete", out)
torch.cuda.synchronize(); t0 = time.perf_counter()
J1 = s.forward([Bl, Ul]); s.adjoint("Discrete", out)
torch.cuda.synchronize(); t_slab = time.perf_counter() - t0
nex = (3 + s.adj_groups) * n          # fwd 1+1, adj (1 or 2)+1 field-group exchanges per step
print(json.dumps({"npts": N, "n_iters": n, "monolithic_s": t_mono, "slab_loop_one_rank_s": t_slab, "exchanges": nex,
                  "overhead_us_per_exchange": 1e6 * (t_slab - t_mono) / nex, "bytes_per_exchange_MB": s.elems * 16 / 1e6,
                  "J_equal": J0 == J1}))
dist.destroy_process_group()

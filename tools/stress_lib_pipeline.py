#!/usr/bin/env python3
"""Race detector for the in-library exchange pipeline: one-rank RCCL communicator with the exchange buffers kept apart
(SMO_SLAB_FORCE_EXCHANGE=1), K pipelined chunks on the communication stream; the same gradient many times, every bit must repeat and
equal the monolithic loop.  usage: python tools/stress_lib_pipeline.py [npts] [n_iters] [reps] [chunks...]"""
import os
import sys

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from spheremanopt_amd import kdyn  # noqa: E402
from spheremanopt_amd.kdyn_slab import LibSlabKDyn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
chunk_list = [int(c) for c in sys.argv[4:]] or [1, 2, 4]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
G = 3 * N // 2
B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
Bd, Ud = torch.from_numpy(B).cuda(), torch.from_numpy(U).cuda()
os.environ["SMO_SLAB_FORCE_EXCHANGE"] = "0"
dom = kdyn.KDynDomain(N)
ctx = dom.context(1., 1e-3, n, "Integrated")
g0 = [torch.empty_like(Bd), torch.empty_like(Ud)]
J0 = ctx.forward_dev([Bd, Ud]); ctx.adjoint_dev([Bd, Ud], g0)
dom.drop_contexts()
os.environ["SMO_SLAB_FORCE_EXCHANGE"] = "1"
for K in chunk_list:
    os.environ["SMO_SLAB_CHUNKS"] = str(K)
    s = LibSlabKDyn(N, 1., 1e-3, n, "Integrated")
    bad = 0
    for r in range(reps):
        out = [torch.empty_like(Bd), torch.empty_like(Ud)]
        J = s.forward([Bd, Ud]); s.adjoint("Discrete", out)
        if J != J0 or not torch.equal(out[0], g0[0]) or not torch.equal(out[1], g0[1]):
            bad += 1
    print("in-library loop over one-rank RCCL, N=%d n=%d chunks=%d: %d of %d repetitions differ from the monolithic loop" % (N, n, s.K, bad, reps), flush=True)
    del s
dist.destroy_process_group()

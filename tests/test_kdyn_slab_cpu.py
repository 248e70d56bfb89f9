"""The multi-GPU (slab) driver on the CPU: world_size-2 gloo processes, NumPy phase backend, compared with the
single-process oracle.  Covers the exchange layout, the phase order, the reductions and the replicated-vector callbacks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, n, cost, adj, keep, chunks, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.kdyn import KDynOracle, synthetic_field
        from slab_numpy_ops import NumpyOps
        from spheremanopt_amd.kdyn_slab import SlabKDyn
        G = 3 * N // 2
        B = synthetic_field(G, 1) + 0.1 * np.random.RandomState(9).standard_normal(3 * G ** 3)
        U = synthetic_field(G, 2)
        s = SlabKDyn(N, 1.3, 1e-2, n, cost, ops=NumpyOps(N, 1.3, 1e-2, n, cost, rank, world, keeps_grid_states=keep), chunks=chunks)
        assert s.K == chunks
        J = s.forward([s.local_slab(B), s.local_slab(U)])
        g = s.adjoint(adj)
        gB, gU = s.gather_full(g[0]), s.gather_full(g[1])
        ip = s.inner(s.local_slab(B), s.local_slab(gB))
        if rank == 0:
            o = KDynOracle(N, Rm=1.3, dt=1e-2, N_ITERS=n, Cost_function=cost)
            Jo = o.forward([B, U]); goB, goU = o.adjoint([B, U], adj)
            res = dict(J=J, Jo=Jo, eB=np.linalg.norm(gB - goB) / np.linalg.norm(goB),
                       eU=np.linalg.norm(gU - goU) / np.linalg.norm(goU), ip=ip, ipo=o.inner(B, goB))
            np.save(out, res, allow_pickle=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,world,cost,adj,keep,chunks", [(8, 2, "Final", "Discrete", True, 1), (16, 2, "Integrated", "Discrete", False, 3),
                                                          (16, 2, "Final", "Continuous", True, 2), (8, 4, "Final", "Discrete", False, 1),
                                                          (8, 4, "Integrated", "Continuous", True, 1), (32, 2, "Final", "Discrete", True, 4),
                                                          (16, 8, "Final", "Discrete", True, 1)])
def test_slab_driver_matches_oracle(tmp_path, N, world, cost, adj, keep, chunks):
    """keep = the forward solve keeps B_n on the grid side (the adjoint's inverse exchange then carries one field group);
    chunks = pipelining granularity of the local z slab (exchange buffers become [chunk][peer]...)."""
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), N, 3, cost, adj, keep, chunks, out), nprocs=world, join=True)
    r = np.load(out, allow_pickle=True).item()
    assert abs(r["J"] - r["Jo"]) <= 1e-10 * abs(r["Jo"])
    assert r["eB"] < 1e-10 and r["eU"] < 1e-10
    assert abs(r["ip"] - r["ipo"]) <= 1e-10 * abs(r["ipo"])


def test_single_rank_numpy_backend_is_consistent():
    """world = 1 through the same driver (no process group): buf_y is buf_z and no exchange happens."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.kdyn import KDynOracle, synthetic_field
    from slab_numpy_ops import NumpyOps
    from spheremanopt_amd.kdyn_slab import SlabKDyn
    N, n = 8, 2
    G = 12
    B, U = synthetic_field(G, 1), synthetic_field(G, 2)
    s = SlabKDyn(N, 1., 1e-2, n, "Final", ops=NumpyOps(N, 1., 1e-2, n, "Final", 0, 1))
    J = s.forward([s.local_slab(B), s.local_slab(U)])
    g = s.adjoint()
    o = KDynOracle(N, Rm=1., dt=1e-2, N_ITERS=n)
    Jo = o.forward([B, U]); go = o.adjoint([B, U])
    assert abs(J - Jo) < 1e-12 * abs(Jo)
    assert np.allclose(g[0].numpy(), go[0], rtol=1e-9, atol=1e-12) and np.allclose(g[1].numpy(), go[1], rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        NumpyOps(8, 1., 1e-2, 2, "Final", 0, 1).phase(999)

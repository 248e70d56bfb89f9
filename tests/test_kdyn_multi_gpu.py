"""The slab decomposition on SEVERAL GPUs with the real transport: one process per GPU, libsmo's own RCCL communicator, grouped
ncclSend/ncclRecv over xGMI between different devices — the one thing the one-GPU test boxes cannot reach (there the same in-library loop
runs with ranks sharing a GPU over the callback transport, and the real RCCL calls on a one-rank communicator:
tests/test_kdyn_slab_gpu.py).  Skipped where fewer than two GPUs are visible."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _ngpu():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, n, cost, adj, ckpt, chunks, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SMO_SLAB_CHUNKS=str(chunks))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from spheremanopt_amd import kdyn, kdyn_slab
        G = 3 * N // 2
        B = kdyn.synthetic_field(G, 1) + 0.1 * np.random.RandomState(9).standard_normal(3 * G ** 3)
        U = kdyn.synthetic_field(G, 2)
        dom = kdyn_slab.SlabDomain(N, device=rank, ckpt=ckpt)
        args = [dom, 1.3, 1e-2, n, n, None, cost, adj]
        J = kdyn_slab.FWD_Solve_IVP_Lin([B, U], *args)
        gB, gU = kdyn_slab.ADJ_Solve_IVP_Lin([B, U], *args)
        ip = kdyn_slab.Inner_Prod_3(B, gB, dom)
        sol = dom.any_solver()
        assert sol.transport == "rccl" and sol.ctx.comm_get(2) == 1.0
        if rank == 0:
            np.savez(out, J=J, gB=gB, gU=gU, ip=ip, B=B, U=U)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(_ngpu() < 2, reason="needs at least two GPUs on the node (grouped ncclSend/ncclRecv between different devices)")
@pytest.mark.parametrize("N,cost,adj,ckpt,chunks", [(32, "Final", "Discrete", 1, 1), (64, "Integrated", "Continuous", 1, 2), (32, "Final", "Discrete", 3, 1)])
def test_in_library_loop_over_rccl_between_gpus(tmp_path, N, cost, adj, ckpt, chunks):
    import torch.multiprocessing as mp
    from oracle.kdyn import KDynOracle
    world = 2
    for w in (4,):                                          # at most 4 ranks + this process on the node
        if _ngpu() >= w and (N // 2) % w == 0 and (3 * N // 2) % w == 0:
            world = w
            break
    n = 4
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, n, cost, adj, ckpt, chunks, out), nprocs=world, join=True)
    r = np.load(out)
    o = KDynOracle(N, Rm=1.3, dt=1e-2, N_ITERS=n, Cost_function=cost)
    Jo = o.forward([r["B"], r["U"]]); goB, goU = o.adjoint([r["B"], r["U"]], adj)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))      # noqa: E731
    assert abs(float(r["J"]) - Jo) <= 1e-6 * abs(Jo)
    assert rel(r["gB"], goB) < 1e-6 and rel(r["gU"], goU) < 1e-6
    assert abs(float(r["ip"]) - o.inner(r["B"], goB)) <= 1e-6 * abs(o.inner(r["B"], goB))


@pytest.mark.skipif(_ngpu() < 2, reason="needs at least two GPUs on the node (peer pulls between different devices)")
@pytest.mark.parametrize("pull", ["kernel", "memcpy"])
@pytest.mark.parametrize("N,cost,adj,ckpt,chunks", [(32, "Final", "Discrete", 1, 1), (64, "Integrated", "Continuous", 1, 2), (32, "Final", "Discrete", 3, 1)])
def test_single_process_multi_device_context_between_gpus(N, cost, adj, ckpt, chunks, pull, monkeypatch):
    """smo_create_multi over DIFFERENT devices: ONE process, one worker thread per GPU, the transposes as pulls out of the peers' send buffers
    over xGMI (the gather kernel with peer access, or hipMemcpyPeerAsync calls) — against the oracle."""
    from oracle.kdyn import KDynOracle
    from spheremanopt_amd import _capi, kdyn
    monkeypatch.setenv("SMO_PEER_COPY", pull)
    monkeypatch.setenv("SMO_SLAB_CHUNKS", str(chunks))
    W = 4 if (_ngpu() >= 4 and (N // 2) % 4 == 0) else 2
    n = 4
    G = 3 * N // 2
    B = kdyn.synthetic_field(G, 1) + 0.1 * np.random.RandomState(9).standard_normal(3 * G ** 3)
    U = kdyn.synthetic_field(G, 2)
    ctx = _capi.MultiContext(N, (0., 2. * np.pi), 1e-2, n, 1.3, list(range(W)), cost=cost, ckpt=ckpt)
    assert ctx.comm_get(3) == (2 if pull == "kernel" else 1)          # the pull implementation in use is the one asked for
    J = ctx.forward([B, U]); gB, gU = ctx.adjoint(None, adj)
    o = KDynOracle(N, Rm=1.3, dt=1e-2, N_ITERS=n, Cost_function=cost)
    Jo = o.forward([B, U]); goB, goU = o.adjoint([B, U], adj)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))      # noqa: E731
    assert abs(J - Jo) <= 1e-6 * abs(Jo) and rel(gB, goB) < 1e-6 and rel(gU, goU) < 1e-6
    assert abs(ctx.inner(B, gB) - o.inner(B, goB)) <= 1e-6 * abs(o.inner(B, goB))
    ctx.close()


@pytest.mark.skipif(_ngpu() < 2, reason="needs at least two GPUs on the node")
def test_default_pull_between_distinct_devices_is_the_copy_engine(monkeypatch):
    """Until the gather kernel has read a peer's HBM on hardware (this file, pull = "kernel"), distinct devices default to hipMemcpyPeerAsync
    (ADVICE r3); ranks that share a device default to the kernel."""
    from spheremanopt_amd import _capi
    monkeypatch.delenv("SMO_PEER_COPY", raising=False)
    ctx = _capi.MultiContext(16, (0., 2. * np.pi), 1e-3, 2, 1.0, [0, 1])
    assert ctx.comm_get(3) == 1
    ctx.close()
    ctx = _capi.MultiContext(16, (0., 2. * np.pi), 1e-3, 2, 1.0, [0, 0])
    assert ctx.comm_get(3) == 2
    ctx.close()


@pytest.mark.skipif(_ngpu() < 8, reason="needs the 8 GPUs of a node")
@pytest.mark.parametrize("pull", ["memcpy", "kernel"])
def test_north_star_decomposition_over_eight_gpus(pull, monkeypatch):
    """BASELINE configs[4] over the 8 GPUs of a node in ONE process against the oracle fixture (the same check tests/test_kdyn_multi_device_gpu.py
    runs with the box's GPU listed eight times; here the pulled bytes cross xGMI)."""
    from conftest import GOLDEN
    from spheremanopt_amd import _capi, kdyn
    monkeypatch.setenv("SMO_PEER_COPY", pull)
    gold = np.load(os.path.join(GOLDEN, "oracle_kdyn_c5_256_n2.npz"))
    B, U = kdyn.synthetic_field(384, 1), kdyn.synthetic_field(384, 2)
    ctx = _capi.MultiContext(256, (0., 2. * np.pi), 1e-3, 2, 1.0, list(range(8)))
    J = ctx.forward([B, U]); g = ctx.adjoint(None, "Discrete")
    Jo, idx = float(gold["J_Final"]), gold["idx"]
    assert abs(J - Jo) <= 1e-6 * abs(Jo)
    for name, v in (("gB", g[0]), ("gU", g[1])):
        ref = gold["Final_Discrete_" + name]
        assert np.linalg.norm(v[idx] - ref) <= 1e-6 * np.linalg.norm(ref), name
    ctx.close()

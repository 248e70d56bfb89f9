"""Slab-decomposed KDyn on real hardware.  The box has ONE GPU, so: (a) world = 1 through the phase-level C-ABI
(smo_kdyn_op) must equal the monolithic path; (b) two processes share cuda:0 and exchange through gloo (host-staged) —
this exercises the slab geometry of the HIP kernels (a/W kx modes, G/W z planes, per-peer blocks of the exchange between the z and
the y pass) against the oracle, with and without the kept grid-side states (one or two field groups on the adjoint's way in)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


@pytest.mark.parametrize("cost,adj", [("Final", "Discrete"), ("Integrated", "Continuous")])
def test_phase_path_equals_monolithic_path(cost, adj):
    import torch
    from spheremanopt_amd import kdyn
    from spheremanopt_amd.kdyn_slab import SlabKDyn
    N, n = 32, 4
    dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args = [dom, 1., 1e-3, n, n, buf, cost, adj]
    J0 = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    g0 = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
    s = SlabKDyn(N, 1., 1e-3, n, cost)
    J1 = s.forward([s.local_slab(B), s.local_slab(U)])
    g1 = s.adjoint(adj)
    assert J1 == J0                                     # same kernels, same order: bit-identical
    assert np.array_equal(g1[0].cpu().numpy(), g0[0]) and np.array_equal(g1[1].cpu().numpy(), g0[1])
    assert abs(s.inner(s.local_slab(B), g1[0]) - kdyn.Inner_Prod_3(B, g0[0], dom)) < 1e-15
    dom.drop_contexts()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, n, cost, adj, keep, chunks, out, inlib=True, ckpt=1):
    sys.path.insert(0, ROOT)
    # keep == "mixed": only rank 0 would keep the grid-side states (as if the others were short of HBM): the ranks must agree on "none",
    # otherwise the adjoint exchanges would carry different numbers of field groups
    os.environ["SMO_KD_TYSTACK"] = "1" if (keep is True or (keep == "mixed" and rank == 0)) else "0"
    os.environ["SMO_SLAB_CHUNKS"] = str(chunks)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from spheremanopt_amd import kdyn, kdyn_slab
        G = 3 * N // 2
        B = kdyn.synthetic_field(G, 1) + 0.1 * np.random.RandomState(9).standard_normal(3 * G ** 3)
        U = kdyn.synthetic_field(G, 2)
        dom = kdyn_slab.SlabDomain(N, device=0, in_library=inlib, ckpt=ckpt)
        args = [dom, 1.3, 1e-2, n, n, None, cost, adj]
        J = kdyn_slab.FWD_Solve_IVP_Lin([B, U], *args)              # the reference-style replicated-vector callbacks
        gB, gU = kdyn_slab.ADJ_Solve_IVP_Lin([B, U], *args)
        ip = kdyn_slab.Inner_Prod_3(B, gB, dom)
        sol = dom.any_solver()
        assert isinstance(sol, kdyn_slab.LibSlabKDyn if inlib else kdyn_slab.SlabKDyn) and sol.K == (chunks if world > 1 else 1)
        if inlib:
            assert sol.transport == "callback" and sol.ctx.comm_get(2) == 0.0
            assert sol.exchanges_per_step_pair == (4 if keep is True and sol.ctx.get(0) == 1 else 5)
        if rank == 0:
            np.savez(out, J=J, gB=gB, gU=gU, ip=ip, B=B, U=U)
    finally:
        dist.destroy_process_group()


_SLAB_CASES = [(16, 2, "Final", "Discrete", True, 1), (32, 4, "Integrated", "Discrete", False, 1),
               (16, 2, "Final", "Continuous", True, 2), (48, 4, "Final", "Discrete", True, 3),
               (32, 2, "Final", "Discrete", False, 4), (64, 2, "Integrated", "Continuous", True, 2),
               (32, 2, "Integrated", "Discrete", True, 1),    # 24 local planes: the halved y-pass tile
               (16, 2, "Final", "Discrete", "mixed", 1), (16, 2, "Final", "Continuous", "mixed", 2)]


def _check_vs_oracle(out, N, n, cost, adj):
    from oracle.kdyn import KDynOracle
    r = np.load(out)
    o = KDynOracle(N, Rm=1.3, dt=1e-2, N_ITERS=n, Cost_function=cost)
    Jo = o.forward([r["B"], r["U"]]); goB, goU = o.adjoint([r["B"], r["U"]], adj)
    assert abs(float(r["J"]) - Jo) <= 1e-6 * abs(Jo)
    assert rel(r["gB"], goB) < 1e-6 and rel(r["gU"], goU) < 1e-6
    assert abs(float(r["ip"]) - o.inner(r["B"], goB)) <= 1e-6 * abs(o.inner(r["B"], goB))


@pytest.mark.parametrize("N,world,cost,adj,keep,chunks", _SLAB_CASES)
def test_in_library_loop_with_ranks_sharing_one_gpu_matches_oracle(tmp_path, N, world, cost, adj, keep, chunks):
    """The product path of the multi-GPU case: time loop and transposes inside libsmo (smo_comm_set_transport + smo_forward_dev /
    smo_adjoint_dev / smo_inner_dev called collectively).  The ranks share the box's one GPU, so the transport is the callback one
    (host-staged gloo) — everything but the ncclSend/ncclRecv group itself is what runs on an 8-GPU node: slab geometry, chunk
    pipeline on two streams, agreement on the kept grid-side states ("mixed": only rank 0 would keep them), reduced J and <x,y>."""
    import torch.multiprocessing as mp
    n = 3
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, n, cost, adj, keep, chunks, out, True), nprocs=world, join=True)
    _check_vs_oracle(out, N, n, cost, adj)


@pytest.mark.parametrize("N,world,cost,adj,keep,chunks,force", [(88, 2, "Final", "Discrete", True, 1, False),
                                                                 (32, 2, "Integrated", "Continuous", False, 2, True),
                                                                 (48, 4, "Final", "Discrete", True, 3, True)])
def test_runtime_length_kernels_with_slabs(tmp_path, monkeypatch, N, world, cost, adj, keep, chunks, force):
    """The any-size kernels (csrc/kdyn_any.hpp) use the slab-exchange layouts of the tuned ones: Npts = 88 (G = 132 = 4*3*11, no tuned
    instantiation) on two slabs, and tuned sizes forced through them (SMO_KD_ANY=1) with the chunk pipeline."""
    import torch.multiprocessing as mp
    if force:
        monkeypatch.setenv("SMO_KD_ANY", "1")
    n = 3
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, n, cost, adj, keep, chunks, out, True), nprocs=world, join=True)
    _check_vs_oracle(out, N, n, cost, adj)


@pytest.mark.parametrize("N,world,cost,adj,ckpt,n,chunks", [(16, 2, "Final", "Discrete", 3, 7, 1), (32, 2, "Integrated", "Continuous", 2, 5, 2),
                                                            (16, 4, "Final", "Discrete", 0, 4, 1)])
def test_windowed_checkpoints_with_slabs(tmp_path, N, world, cost, adj, ckpt, n, chunks):
    """Windowed checkpointing and the slab decomposition together (the combination a 384^3-class problem needs): the recomputation of
    a window runs the same exchanging forward steps; ckpt = 0: every rank picks the interval from its free HBM and the ranks must
    agree on it."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, n, cost, adj, True, chunks, out, True, ckpt), nprocs=world, join=True)
    _check_vs_oracle(out, N, n, cost, adj)


@pytest.mark.parametrize("N,world,cost,adj,keep,chunks", [_SLAB_CASES[i] for i in (0, 1, 3, 5, 8)])
def test_ranks_sharing_one_gpu_match_oracle(tmp_path, N, world, cost, adj, keep, chunks):
    """The phase-level entry (smo_kdyn_op) driven by the Python loop of kdyn_slab.SlabKDyn — the harness the CPU/gloo tests use."""
    import torch.multiprocessing as mp
    n = 3
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, n, cost, adj, keep, chunks, out, False), nprocs=world, join=True)
    _check_vs_oracle(out, N, n, cost, adj)


def _nccl_worker(rank, port, N, n, chunks, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMO_SLAB_FORCE_EXCHANGE="1", SMO_SLAB_CHUNKS=str(chunks))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from spheremanopt_amd import kdyn
        from spheremanopt_amd.kdyn_slab import SlabKDyn
        dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
        buf = kdyn.GEN_BUFFER(N, dom, n)
        args = [dom, 1., 1e-3, n, n, buf, "Final", "Discrete"]
        os.environ["SMO_SLAB_FORCE_EXCHANGE"] = "0"          # the monolithic reference: one buffer, no exchange
        J0 = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
        g0 = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
        os.environ["SMO_SLAB_FORCE_EXCHANGE"] = "1"
        s = SlabKDyn(N, 1., 1e-3, n, "Final")
        assert s.force_exchange and not s.host_staged and s.buf_y.data_ptr() != s.buf_z.data_ptr() and s.K == chunks
        J1 = s.forward([s.local_slab(B), s.local_slab(U)])
        g1 = s.adjoint("Discrete")
        ip = s.inner(s.local_slab(B), g1[0])
        np.savez(out, J0=J0, J1=J1, eB=np.abs(g1[0].cpu().numpy() - g0[0]).max(), eU=np.abs(g1[1].cpu().numpy() - g0[1]).max(),
                 ip=ip, ip0=kdyn.Inner_Prod_3(B, g0[0], dom))
    finally:
        dist.destroy_process_group()


def _lib_rccl_worker(rank, port, N, n, chunks, cost, adj, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMO_SLAB_FORCE_EXCHANGE="1", SMO_SLAB_CHUNKS=str(chunks))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from spheremanopt_amd import kdyn
        from spheremanopt_amd.kdyn_slab import LibSlabKDyn
        dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
        os.environ["SMO_SLAB_FORCE_EXCHANGE"] = "0"
        buf = kdyn.GEN_BUFFER(N, dom, n)
        args = [dom, 1., 1e-3, n, n, buf, cost, adj]
        J0 = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
        g0 = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
        os.environ["SMO_SLAB_FORCE_EXCHANGE"] = "1"
        s = LibSlabKDyn(N, 1., 1e-3, n, cost)
        assert s.transport == "rccl" and s.ctx.comm_get(2) == 1.0 and s.K == chunks
        J1 = s.forward([s.local_slab(B), s.local_slab(U)])
        g1 = s.adjoint(adj)
        g2 = s.adjoint(adj)                                   # a second sweep over the same stack
        ip = s.inner(s.local_slab(B), g1[0])
        np.savez(out, J0=J0, J1=J1, eB=np.abs(g1[0].cpu().numpy() - g0[0]).max(), eU=np.abs(g1[1].cpu().numpy() - g0[1]).max(),
                 e2=np.abs(g2[0].cpu().numpy() - g0[0]).max(), ip=ip, ip0=kdyn.Inner_Prod_3(B, g0[0], dom))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunks,cost,adj", [(1, "Final", "Discrete"), (2, "Integrated", "Continuous"), (4, "Final", "Discrete")])
def test_in_library_rccl_path_on_one_rank(tmp_path, chunks, cost, adj):
    """smo_comm_init + the in-library loop with REAL RCCL calls (ncclCommInitRank, grouped ncclSend/ncclRecv on the solver's streams,
    ncclAllReduce of J and <x,y>) on a one-rank communicator with the two exchange buffers kept apart: every transpose is a
    self-exchange through RCCL, so the result must equal the monolithic single-GPU path bit for bit.  chunks > 1: the two-stream
    pipeline (exchanges on the communication stream, events in both directions)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npz")
    mp.spawn(_lib_rccl_worker, args=(_free_port(), 32, 4, chunks, cost, adj, out), nprocs=1, join=True)
    r = np.load(out)
    assert float(r["J1"]) == float(r["J0"])
    assert float(r["eB"]) == 0.0 and float(r["eU"]) == 0.0 and float(r["e2"]) == 0.0
    assert abs(float(r["ip"]) - float(r["ip0"])) < 1e-15


@pytest.mark.parametrize("chunks", [1, 2, 4])
def test_rccl_call_path_on_one_rank(tmp_path, chunks):
    """The real RCCL collectives of the multi-GPU time loop (all_to_all_single on the solver's own HIP stream, all_reduce of J and
    <x,y>) on a one-rank process group, with the two exchange buffers kept apart: every transpose is a self-copy THROUGH the
    collective, so the result must equal the monolithic single-GPU path bit for bit.  chunks > 1: the pipelined loop — asynchronous
    collectives on the process group's stream overlapping the grid-side kernels of the other chunks."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npz")
    mp.spawn(_nccl_worker, args=(_free_port(), 32, 4, chunks, out), nprocs=1, join=True)
    r = np.load(out)
    assert float(r["J1"]) == float(r["J0"])
    assert float(r["eB"]) == 0.0 and float(r["eU"]) == 0.0
    assert abs(float(r["ip"]) - float(r["ip0"])) < 1e-15


def _big_worker(rank, world, port, N, n, chunks, out):
    sys.path.insert(0, ROOT)
    os.environ["SMO_SLAB_CHUNKS"] = str(chunks)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from spheremanopt_amd import kdyn
        from spheremanopt_amd.kdyn_slab import LibSlabKDyn
        G = 3 * N // 2
        B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
        s = LibSlabKDyn(N, 1., 1e-3, n, "Final")
        assert s.K == chunks
        J = s.forward([s.local_slab(B), s.local_slab(U)])
        g = s.adjoint("Discrete")
        gB, gU = s.gather_full(g[0]), s.gather_full(g[1])
        if rank == 0:
            np.savez(out, J=J, gB=gB, gU=gU)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,world,chunks", [(128, 4, 1), (256, 4, 2)])
def test_bench_size_slabs_agree_with_the_single_gpu_path(tmp_path, N, world, chunks):
    """The slab geometries of the multi-GPU benchmark sizes (128^3: thin slabs; 256^3: the half-size tiles of the G = 384 kernels, two
    pipelined chunks) against the monolithic single-GPU path on the same inputs: same kernels, another summation order only in J."""
    import torch.multiprocessing as mp
    from spheremanopt_amd import kdyn
    n = 2
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    dom = kdyn.KDynDomain(N)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args = [dom, 1., 1e-3, n, n, buf, "Final", "Discrete"]
    J0 = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    g0 = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
    dom.drop_contexts()
    out = str(tmp_path / "res.npz")
    mp.spawn(_big_worker, args=(world, _free_port(), N, n, chunks, out), nprocs=world, join=True)
    r = np.load(out)
    assert abs(float(r["J"]) - J0) <= 1e-12 * abs(J0)
    assert rel(r["gB"], g0[0]) < 1e-12 and rel(r["gU"], g0[1]) < 1e-12
    if N == 256:
        # ... and with the ORACLE at the north-star grid (tests/golden/oracle_kdyn_c5_256_n2.npz: same seeds, Rm, dt, 2 steps), so that
        # the slab path of BASELINE configs[4] is checked against something other than this library
        from conftest import GOLDEN
        gold = np.load(os.path.join(GOLDEN, "oracle_kdyn_c5_256_n2.npz"))
        assert int(gold["N"]) == N and int(gold["steps"]) == n and float(gold["dt"]) == 1e-3 and float(gold["Rm"]) == 1.0
        Jo, idx = float(gold["J_Final"]), gold["idx"]
        assert abs(float(r["J"]) - Jo) <= 1e-6 * abs(Jo)
        for name in ("gB", "gU"):
            ref = gold["Final_Discrete_" + name]
            assert np.linalg.norm(r[name][idx] - ref) <= 1e-6 * np.linalg.norm(ref)
            nrm = float(gold["Final_Discrete_%s_norm" % name])
            assert abs(np.linalg.norm(r[name]) - nrm) <= 1e-6 * nrm


def _against_c5_fixture(J, gB, gU, key="Final_Discrete"):
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "oracle_kdyn_c5_256_n2.npz"))
    assert int(gold["N"]) == 256 and int(gold["steps"]) == 2 and float(gold["dt"]) == 1e-3 and float(gold["Rm"]) == 1.0
    Jo, idx = float(gold["J_" + key.split("_")[0]]), gold["idx"]
    assert abs(J - Jo) <= 1e-6 * abs(Jo), (J, Jo)
    for name, v in (("gB", gB), ("gU", gU)):
        ref = gold["%s_%s" % (key, name)]
        assert np.linalg.norm(v[idx] - ref) <= 1e-6 * np.linalg.norm(ref), name
        nrm = float(gold["%s_%s_norm" % (key, name)])
        assert abs(np.linalg.norm(v) - nrm) <= 1e-6 * nrm, name


def test_north_star_decomposition_8_way_in_library_loop(fields384):
    """BASELINE configs[4] exactly as `bench.py --gpus 8` decomposes it — 256^3, W = 8: 16 kx planes and 48 z planes (6 z-blocks) per rank, the
    G = 384 kernels, the library's DEFAULT chunk count — through the in-library loop and the callback transport, against the oracle fixture
    (the scaling node must not be the first execution of this geometry).  The 8 ranks are threads of this process
    (tests/thread_ranks.py): the box allows 6 processes on its GPU, so 8 gloo ranks cannot share it.  Reference: FWD_Solve_KDyn.py:118-134
    under `mpiexec -np 8` (README.md:83)."""
    from spheremanopt_amd import kdyn
    from thread_ranks import slab_gradient
    N, W, n = 256, 8, 2
    B, U = fields384
    J, gB, gU, K, nex = slab_gradient(N, W, n, B, U, "Final", "Discrete")
    assert K == 1                                  # the default at 256^3 / 8: G * Gzr / 36864 < 1 chunk of a plane set -> no pipelining
    assert nex >= 3 + 4 * n + 3                    # U, B in; 4 exchanges per step pair; the two gradients and nu out
    _against_c5_fixture(J, gB, gU)
    # ... and with two pipelined chunks (24 z planes = 3 z-blocks each: the chunk-major exchange layout at this geometry), bit for bit
    J2, gB2, gU2, K2, _ = slab_gradient(N, W, n, B, U, "Final", "Discrete", chunks=2)
    assert K2 == 2 and J2 == J and np.array_equal(gB2, gB) and np.array_equal(gU2, gU)


def test_bench_grid_8_way_thin_slabs_in_library_loop():
    """128^3 at W = 8 (what `bench.py --gpus 8` times as its main line): 8 kx planes and 24 z planes per rank — the halved y-pass tile (24 % 8 == 0
    but Gzr / ZT = 3 tiles; x tiles of 8 points over 192 x 24 planes) — one chunk, against the single-GPU path (itself checked against the
    oracle at this grid: tests/test_kdyn_gpu.py) on the same inputs: same kernels, another summation order only in J."""
    from spheremanopt_amd import kdyn
    from thread_ranks import slab_gradient
    N, W, n = 128, 8, 3
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    dom = kdyn.KDynDomain(N)
    args = [dom, 1., 1e-3, n, n, kdyn.GEN_BUFFER(N, dom, n), "Integrated", "Continuous"]
    J0 = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    g0 = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
    dom.drop_contexts()
    J, gB, gU, K, _ = slab_gradient(N, W, n, B, U, "Integrated", "Continuous")
    assert K == 1
    assert abs(J - J0) <= 1e-12 * abs(J0)
    assert rel(gB, g0[0]) < 1e-12 and rel(gU, g0[1]) < 1e-12


def test_communicator_errors():
    """world > 1 without a communicator: the loop entry points refuse (the phase-level entry still works); a single-slab context has
    nothing to exchange; smo_comm_get reports the pipeline."""
    from spheremanopt_amd import _capi
    ctx = _capi.Context(_capi.SMO_KDYN, 16, (0., 2 * np.pi), 1e-3, 2, 1.0, rank=1, world=2)
    x = np.zeros(ctx.vec_len)
    with pytest.raises(_capi.SmoError) as e:
        ctx.forward([x, x])
    assert e.value.code == 4 and "smo_comm_init" in str(e.value)
    assert ctx.comm_get(0) == 1 and ctx.comm_get(2) == 0
    one = _capi.Context(_capi.SMO_KDYN, 16, (0., 2 * np.pi), 1e-3, 2, 1.0)
    with pytest.raises(_capi.SmoError) as e:
        one.comm_init(b"\0" * 128)
    assert e.value.code == 4
    with pytest.raises(ValueError):
        ctx.comm_init(b"short")
    # a transport whose all-to-all raises: the call fails with an error code instead of unwinding through the C frames
    def boom(src, dst, nbytes, stream):
        raise RuntimeError("wire down")
    def ared(vals, n):
        for i in range(n):
            vals[i] = 2.0 * vals[i]          # pretend the other rank decided the same
    ctx.comm_set_transport(boom, ared)
    with pytest.raises(_capi.SmoError) as e:
        ctx.forward([x, x])
    assert "transport failed" in str(e.value) and isinstance(ctx._transport_error, RuntimeError)
    # ranks that cannot agree (here: the "other rank" reports checkpoint interval 2 against this rank's 1): smo_comm_set_transport fails AND
    # drops the transport again — the context is back in its "no communicator" state instead of keeping a live one whose next solve would
    # hang in a mismatched all-to-all — and the call can be repeated once the ranks agree
    c2 = _capi.Context(_capi.SMO_KDYN, 16, (0., 2 * np.pi), 1e-3, 2, 1.0, rank=0, world=2)
    def ok_a2a(src, dst, nbytes, stream):
        pass
    def disagree(vals, n):
        if n == 4:
            vals[0] += vals[0]; vals[1] += 2.0; vals[2] += 4.0; vals[3] += vals[3]
        else:
            for i in range(n):
                vals[i] = 2.0 * vals[i]
    with pytest.raises(_capi.SmoError) as e:
        c2.comm_set_transport(ok_a2a, disagree)
    assert e.value.code == 4 and "different checkpoint intervals" in str(e.value) and "dropped" in str(e.value)
    with pytest.raises(_capi.SmoError) as e:
        c2.forward([x, x])
    assert e.value.code == 4 and "no communicator" in str(e.value)
    c2.comm_set_transport(ok_a2a, ared)                      # second attempt, the ranks agree now
    assert np.isfinite(c2.forward([x, x]))


def _autotune_worker(rank, world, port, N, n, out):
    sys.path.insert(0, ROOT)
    os.environ.pop("SMO_SLAB_CHUNKS", None)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from spheremanopt_amd import kdyn
        from spheremanopt_amd.kdyn_slab import LibSlabKDyn
        G = 3 * N // 2
        B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
        s = LibSlabKDyn(N, 1., 1e-3, n, "Integrated")
        X = [s.local_slab(B), s.local_slab(U)]
        J0 = s.forward(X); g0 = [t.clone() for t in s.adjoint("Discrete")]
        times = s.autotune_chunks(X, candidates=(1, 2, 3, 4, 5))
        J1 = s.forward(X); g1 = s.adjoint("Discrete")
        if rank == 0:
            np.savez(out, J0=J0, J1=J1, K=s.K, tried=np.array(sorted(times)), best=min(times, key=times.get),
                     e=max(float((g1[i] - g0[i]).abs().max()) for i in range(2)))
    finally:
        dist.destroy_process_group()


def test_chunk_autotune_keeps_results(tmp_path):
    """Re-chunking a live in-library context (SMO_KD_SET_CHUNKS) and the start-up autotune of the exchange pipeline: the candidates that
    divide the slab are tried, the fastest is kept on every rank, and the numbers do not depend on the choice."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npz")
    mp.spawn(_autotune_worker, args=(2, _free_port(), 32, 4, out), nprocs=2, join=True)        # 24 local planes: 1, 2, 3, 4 give even chunks; 5 does not
    r = np.load(out)
    assert list(r["tried"]) == [1, 2, 3, 4] and int(r["K"]) == int(r["best"])
    assert abs(float(r["J1"]) - float(r["J0"])) <= 1e-13 * abs(float(r["J0"])) and float(r["e"]) < 1e-13

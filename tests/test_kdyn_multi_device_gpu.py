"""smo_create_multi: ONE process, ONE context, the 3-D problem slab-decomposed over several devices (include/smo.h).

A one-GPU box lists its device several times (dev_ids = [0, 0], [0, 0, 0, 0]): every rank is then a worker thread with its own streams and
buffers on that GPU and every transpose goes through the PeerGroup protocol (events, host barriers, pulls from the peers' buffers) — the
only thing a one-GPU box cannot exercise is that the pulled bytes cross xGMI.  Results are compared with the oracle and, bit for bit where
the arithmetic is the same, with the single-device context."""
import numpy as np
import pytest

from spheremanopt_amd import _capi, kdyn

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


@pytest.mark.parametrize("N,devs,cost,adj", [(16, [0, 0], "Final", "Discrete"), (16, [0, 0, 0, 0], "Integrated", "Discrete"),
                                             (24, [0, 0], "Final", "Continuous"), (32, [0, 0, 0, 0], "Integrated", "Continuous")])
def test_multi_device_context_vs_oracle(N, devs, cost, adj):
    from oracle.kdyn import KDynOracle
    n, dt = 6, 5e-3
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    ctx = _capi.MultiContext(N, (0., 2. * np.pi), dt, n, 1.0, devs, cost=cost)
    assert ctx.vec_len == 3 * G ** 3 and ctx.ncomp == 2            # the caller's vectors are the reference's full ones
    J = ctx.forward([B, U])
    gB, gU = ctx.adjoint(None, adj)
    o = KDynOracle(N, Rm=1.0, dt=dt, N_ITERS=n, Cost_function=cost)
    Jo = o.forward([B, U]); goB, goU = o.adjoint([B, U], Adjoint_type=adj)
    assert abs(J - Jo) <= RTOL * abs(Jo), (J, Jo)
    assert rel(gB, goB) < RTOL and rel(gU, goU) < RTOL, (rel(gB, goB), rel(gU, goU))
    ip = ctx.inner(B, gB)
    assert abs(ip - o.inner(B, goB)) <= RTOL * abs(o.inner(B, goB))
    ctx.close()


def test_both_pull_implementations_agree(monkeypatch):
    """The gather kernel (default with peer access) and the hipMemcpyPeerAsync calls (SMO_PEER_COPY=memcpy) move the same bytes."""
    N, n, dt = 24, 6, 5e-3
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    res = []
    for mode in ("kernel", "memcpy"):
        monkeypatch.setenv("SMO_PEER_COPY", mode)
        ctx = _capi.MultiContext(N, (0., 2. * np.pi), dt, n, 1.0, [0, 0, 0, 0], cost="Integrated")
        res.append((ctx.forward([B, U]), [g.copy() for g in ctx.adjoint(None, "Continuous")]))
        ctx.close()
    (Ja, ga), (Jb, gb) = res
    assert Ja == Jb and np.array_equal(ga[0], gb[0]) and np.array_equal(ga[1], gb[1])


def test_multi_device_context_equals_single_device_and_repeats():
    """Same kernels, same per-mode arithmetic: J agrees with the one-device context to the rounding of the energy reduction (the
    partial sums are split differently), gradients to 1e-12; a second evaluation on the same context is bit-identical to the first."""
    N, n, dt = 32, 10, 2e-3
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 3), kdyn.synthetic_field(G, 4)
    one = _capi.Context(_capi.SMO_KDYN, N, (0., 2. * np.pi), dt, n, 1.0)
    J1 = one.forward([B, U]); g1 = one.adjoint(None)
    ctx = _capi.MultiContext(N, (0., 2. * np.pi), dt, n, 1.0, [0, 0, 0, 0])
    J4 = ctx.forward([B, U]); g4 = [g.copy() for g in ctx.adjoint(None)]
    assert abs(J4 - J1) <= 1e-13 * abs(J1)
    assert rel(g4[0], g1[0]) < 1e-12 and rel(g4[1], g1[1]) < 1e-12
    J4b = ctx.forward([B, U]); g4b = ctx.adjoint(None)
    assert J4b == J4 and np.array_equal(g4b[0], g4[0]) and np.array_equal(g4b[1], g4[1])
    one.close(); ctx.close()


def test_multi_device_context_with_checkpoint_windows():
    N, n, dt = 24, 9, 5e-3
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    ref = _capi.MultiContext(N, (0., 2. * np.pi), dt, n, 1.0, [0, 0])
    Jr = ref.forward([B, U]); gr = [g.copy() for g in ref.adjoint(None)]
    ref.close()
    ck = _capi.MultiContext(N, (0., 2. * np.pi), dt, n, 1.0, [0, 0], ckpt=3)
    assert ck.get(0) == 3
    Jc = ck.forward([B, U]); gc = ck.adjoint(None)
    assert Jc == Jr and np.array_equal(gc[0], gr[0]) and np.array_equal(gc[1], gr[1])       # recomputed states are the same states
    ck.close()


@pytest.mark.parametrize("chunks,devs", [(2, [0, 0]), (3, [0, 0, 0, 0])])
def test_multi_device_context_with_the_chunk_pipeline(chunks, devs, monkeypatch):
    """SMO_SLAB_CHUNKS = K: the exchanges of the peer transport run chunk by chunk on the second stream, overlapped with the grid kernels
    (the same pipeline as with RCCL); gradients equal the one-chunk run bit for bit."""
    N, n, dt = 48, 5, 5e-3
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 5), kdyn.synthetic_field(G, 6)
    res = []
    for K in (1, chunks):
        monkeypatch.setenv("SMO_SLAB_CHUNKS", str(K))
        ctx = _capi.MultiContext(N, (0., 2. * np.pi), dt, n, 1.0, devs, cost="Integrated")
        assert ctx.comm_get(0) == K
        res.append((ctx.forward([B, U]), [g.copy() for g in ctx.adjoint(None)]))
        ctx.close()
    (J1, g1), (JK, gK) = res
    assert JK == J1 and np.array_equal(gK[0], g1[0]) and np.array_equal(gK[1], g1[1])


def test_multi_device_context_at_the_bench_grid_against_the_oracle_fixture():
    """BASELINE config 4's grid (128^3, 50 of its 1000 steps) on four worker threads of one context against the committed oracle run."""
    import os
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "oracle_kdyn_c4_128_n50.npz"))
    N, n = 128, 50
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    ctx = _capi.MultiContext(N, (0., 2. * np.pi), 1e-3, n, 1.0, [0, 0, 0, 0])
    J = ctx.forward([B, U])
    g = ctx.adjoint(None)
    assert abs(J - float(gold["J_Final"])) <= RTOL * abs(float(gold["J_Final"]))
    idx = gold["idx"]
    for name, v in (("gB", g[0]), ("gU", g[1])):
        ref = gold["Final_Discrete_" + name]
        assert np.linalg.norm(v[idx] - ref) <= RTOL * np.linalg.norm(ref), name
        nrm = float(gold["Final_Discrete_%s_norm" % name])
        assert abs(np.linalg.norm(v) - nrm) <= RTOL * nrm
    ctx.close()


def _check_against(gold, J, g, key):
    Jo = float(gold["J_" + key.split("_")[0]])
    assert abs(J - Jo) <= RTOL * abs(Jo), (key, J, Jo)
    idx = gold["idx"]
    for name, v in (("gB", g[0]), ("gU", g[1])):
        ref = gold["%s_%s" % (key, name)]
        assert np.linalg.norm(v[idx] - ref) <= RTOL * np.linalg.norm(ref), (key, name)
        nrm = float(gold["%s_%s_norm" % (key, name)])
        assert abs(np.linalg.norm(v) - nrm) <= RTOL * nrm, (key, name)


def test_multi_device_context_at_the_north_star_grid_8_way(monkeypatch, fields384):
    """BASELINE configs[4]'s decomposition — 256^3 over EIGHT ranks (16 kx planes, 48 z planes = 6 z-blocks per rank, the G = 384 kernels, default
    chunk count) — through ONE multi-device context with the box's GPU listed eight times, against the oracle fixture: both cost functionals,
    both adjoints, both pull implementations, keep-all and windowed checkpoints.  Reference: FWD_Solve_KDyn.py:118-134 under
    `mpiexec -np 8` (README.md:83)."""
    import os
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "oracle_kdyn_c5_256_n2.npz"))
    N, n, devs = 256, 2, [0] * 8
    B, U = fields384
    first = None
    for cost, adj, pull, ckpt in (("Final", "Discrete", "kernel", 1), ("Integrated", "Continuous", "memcpy", 2), ("Final", "Discrete", "memcpy", 2)):
        monkeypatch.setenv("SMO_PEER_COPY", pull)
        ctx = _capi.MultiContext(N, (0., 2. * np.pi), 1e-3, n, 1.0, devs, cost=cost, ckpt=ckpt)
        assert ctx.comm_get(0) == 1 and ctx.get(0) == ckpt               # default chunk count at 256^3 / 8; the interval asked for
        assert ctx.comm_get(3) == (2 if pull == "kernel" else 1)          # the pull implementation in use is reported
        J = ctx.forward([B, U])
        g = [v.copy() for v in ctx.adjoint(None, adj)]
        ctx.close()
        _check_against(gold, J, g, "%s_%s" % (cost, adj))
        if (cost, adj) == ("Final", "Discrete"):
            if first is None:
                first = (J, g)
            else:                                                        # other transport, recomputed windows: the same bits
                assert J == first[0] and np.array_equal(g[0], first[1][0]) and np.array_equal(g[1], first[1][1])


def test_multi_device_context_at_the_bench_grid_8_way():
    """128^3 over eight ranks (8 kx planes, 24 z planes per rank: the halved y-pass tile) against the committed 50-step oracle run."""
    import os
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "oracle_kdyn_c4_128_n50.npz"))
    N, n = 128, 50
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    ctx = _capi.MultiContext(N, (0., 2. * np.pi), 1e-3, n, 1.0, [0] * 8, cost="Integrated")
    J = ctx.forward([B, U])
    g = ctx.adjoint(None, "Discrete")
    _check_against(gold, J, g, "Integrated_Discrete")
    ctx.close()


def test_null_and_short_slab_lists_are_argument_errors():
    """a multi-device context dereferences one slab pointer per (component, device): every one is validated (C side: NULL ->
    SMO_ERR_ARG; Python side: length and device of every slab that describes itself) instead of faulting inside a worker thread."""
    import ctypes as C
    import torch
    ctx = _capi.MultiContext(16, (0., 2. * np.pi), 1e-3, 2, 1.0, [0, 0])
    n_slab = ctx.vec_len // 2
    good = [torch.zeros(n_slab, dtype=torch.float64, device="cuda:0") for _ in range(4)]
    assert np.isfinite(ctx.forward_dev(good))
    with pytest.raises(ValueError):
        ctx.forward_dev(good[:3])                                                       # a short list
    with pytest.raises(ValueError):
        ctx.forward_dev(good[:3] + [torch.zeros(n_slab - 1, dtype=torch.float64, device="cuda:0")])      # a short slab
    J = np.zeros(1)
    ptrs = _capi._ptr_array([_capi._dev_ptr(t) for t in good[:3]] + [None])             # straight through the C-ABI: a NULL among the 2 * ndev
    rc = _capi.lib().smo_forward_dev(ctx._h, ptrs, J.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 1 and b"pointer 3 of 4 is null" in _capi.lib().smo_last_error()
    out = np.zeros(1)
    xs = _capi._ptr_array([_capi._dev_ptr(good[0]), None])
    rc = _capi.lib().smo_inner_slabs(ctx._h, xs, xs, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 1 and b"slab 1 of 2 is null" in _capi.lib().smo_last_error()
    rc = _capi.lib().smo_set_stream(ctx._h, None)
    assert rc == 6 and b"private stream" in _capi.lib().smo_last_error()                 # SMO_ERR_UNSUPPORTED
    ctx.close()


def test_null_transport_runs_one_ranks_share():
    """smo_comm_set_transport(ctx, NULL, NULL, NULL): rank 0 of a 4-way decomposition runs its kernels through the real in-library loop with no
    exchange at all (tools/prof_slab_geometry.py); results are meaningless, the call sequence must simply complete and be timed."""
    ctx = _capi.Context(_capi.SMO_KDYN, 32, (0., 2. * np.pi), 1e-3, 4, 1.0, rank=0, world=4)
    _capi._check(_capi.lib().smo_comm_set_transport(ctx._h, _capi.ALLTOALL_FN(), _capi.ALLREDUCE_FN(), None))
    x = np.full(ctx.vec_len, 1e-3)
    ctx.timing_enable(True)
    J = ctx.forward([x, x]); g = ctx.adjoint(None)
    assert np.isfinite(J) and len(g) == 2 and g[0].shape == (ctx.vec_len,)
    tim = {t["kernel"]: t["launches"] for t in ctx.timing()}
    assert tim["slab_exchange(all-to-all)"] >= 16 and tim["kd_x_pass<fused_adj>"] == 4
    ctx.close()


@pytest.mark.filterwarnings("error")             # a line search that hits its cap or does not converge would warn: not here any more
def test_reference_callbacks_on_a_multi_device_domain():
    """The drop-in surface: the reference's callbacks + optimiser in ONE process over a slab-decomposed context, no launcher."""
    from spheremanopt_amd.sphere_opt import Optimise_On_Multi_Sphere
    N, n, dt = 16, 8, 5e-3
    res = []
    for devices in (None, [0, 0]):
        dom = kdyn.KDynDomain(N, devices=devices)
        G = dom.G
        B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
        buf = kdyn.GEN_BUFFER(N, dom, n)
        args_f = [dom, 1.0, dt, n, n, buf, "Final", "Discrete"]
        R, F, X = Optimise_On_Multi_Sphere([B, U], [1.0, 1.0], kdyn.FWD_Solve_IVP_Lin, kdyn.ADJ_Solve_IVP_Lin, kdyn.Inner_Prod_3,
                                            args_f=args_f, args_IP=(dom, None), max_iters=3, alpha_k=10., LS='LS_wolfe', CG=True, verbose=False)
        res.append((F, X))
        dom.drop_contexts()
    (F1, X1), (F2, X2) = res
    assert len(F1) == 3                         # three iterations of a search that converges (alpha_k = 1 hit the Wolfe search's cap: VERDICT r3 weak 10)
    assert len(F1) == len(F2) and np.allclose(F1, F2, rtol=1e-9, atol=0)
    assert rel(X2[0], X1[0]) < 1e-8 and rel(X2[1], X1[1]) < 1e-8


@pytest.mark.filterwarnings("error")
def test_optimiser_on_distributed_device_vectors():
    """devvec.MultiDeviceVector: the optimiser's vectors stay distributed over the devices of the context (no PCIe traffic, no NumPy algebra
    on full-size vectors); the iterate sequence equals the one on NumPy vectors through the same multi-device context."""
    from spheremanopt_amd.devvec import MultiDeviceVector, to_devices, to_host
    from spheremanopt_amd.sphere_opt import Optimise_On_Multi_Sphere
    N, n, dt, devs = 16, 8, 5e-3, [0, 0]
    dom = kdyn.KDynDomain(N, devices=devs)
    G = dom.G
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    v = MultiDeviceVector.from_numpy(B, devs)
    assert np.array_equal(v.numpy(), B) and np.array_equal((v + 0.5 * v).numpy(), B + 0.5 * B) and np.array_equal((-v).numpy(), -B)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args_f = [dom, 1.0, dt, n, n, buf, "Final", "Discrete"]
    res = []
    for X0 in ([B, U], to_devices([B, U], devs)):
        R, F, X = Optimise_On_Multi_Sphere(X0, [1.0, 1.0], kdyn.FWD_Solve_IVP_Lin, kdyn.ADJ_Solve_IVP_Lin, kdyn.Inner_Prod_3,
                                            args_f=args_f, args_IP=(dom, None), max_iters=3, alpha_k=10., LS='LS_wolfe', CG=True, verbose=False)
        res.append((R, F, to_host(X)))
    (R1, F1, X1), (R2, F2, X2) = res
    assert isinstance(res[1][2][0], np.ndarray) and len(F1) == len(F2)
    assert np.allclose(F1, F2, rtol=1e-12, atol=0) and np.allclose(R1, R2, rtol=1e-9)
    assert rel(X2[0], X1[0]) < 1e-11 and rel(X2[1], X1[1]) < 1e-11
    dom.drop_contexts()


def test_errors():
    with pytest.raises(_capi.SmoError):
        _capi.MultiContext(16, (0., 2. * np.pi), 1e-3, 4, 1.0, [0, 0, 0])           # 3 devices do not divide a = 8
    with pytest.raises(_capi.SmoError):
        _capi.MultiContext(16, (0., 2. * np.pi), 1e-3, 4, 1.0, [0, 99])             # no such device
    ctx = _capi.MultiContext(16, (0., 2. * np.pi), 1e-3, 4, 1.0, [0, 0])
    with pytest.raises(_capi.SmoError):
        ctx.adjoint(None)                                                        # the hidden contract: adjoint needs a forward solve
    with pytest.raises(ValueError):
        ctx.forward([np.zeros(5), np.zeros(5)])
    with pytest.raises(_capi.SmoError):
        ctx.snapshot(0)
    ctx.close()

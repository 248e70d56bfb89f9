"""SHB23 HIP path (through the C-ABI): device Chebyshev maps against the vectors produced by the REFERENCE's own helper
functions, and forward/adjoint against the oracle (1e-6 relative on J and grad J)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from spheremanopt_amd import _capi, shb23
from spheremanopt_amd.test_grad import taylor_table

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


def test_device_transforms_match_reference_helpers():
    g = np.load(os.path.join(GOLDEN, "shb_helpers.npz"))
    dom = shb23.SHBDomain(512)
    v = g["v512"]
    assert rel(shb23.transform(v, dom), g["T512"]) < 1e-13
    assert rel(shb23.transformInverse(v, dom), g["Tinv512"]) < 1e-13
    assert rel(shb23.transformAdjoint(v, dom), g["Tadj512"]) < 1e-13
    assert rel(shb23.transformInverseAdjoint(v, dom), g["Tinvadj512"]) < 1e-13
    assert np.array_equal(shb23.weightMatrixDisc(dom), g["W512"])
    assert abs(shb23.Inner_Prod_Discrete(v, g["T512"], dom) - g["ip512"]) < 1e-13 * abs(g["ip512"])
    # round trip
    assert rel(shb23.transformInverse(shb23.transform(v, dom), dom), v) < 1e-13


@pytest.mark.parametrize("N,n", [(64, 30), (128, 100), (256, 40), (1024, 10), (96, 40), (192, 60), (384, 40), (768, 12)])
def test_forward_adjoint_vs_oracle(N, n):
    from oracle import shb23 as osh
    o = osh.SHB23Oracle(N, dt=1e-2, N_ITERS=n)
    X = osh.synthetic_ic(o, 42, 0.0019)
    dom = shb23.SHBDomain(N)
    buf = shb23.GEN_BUFFER(N, dom, n)
    J = shb23.FWD_Solve_IVP_Discrete([X], dom, buf, n, 1e-2)
    g = shb23.ADJ_Solve_IVP_Discrete([X], dom, buf, n, 1e-2)
    Jo = o.forward([X]); go = o.adjoint([X])
    assert abs(J - Jo) <= RTOL * abs(Jo), (J, Jo)
    assert len(g) == 1 and rel(g[0], go[0]) < RTOL, rel(g[0], go[0])
    for i in (0, 1, -1, -2):
        assert rel(buf['A_fwd'][:, i], o.stack[:, i]) < 1e-8
    assert abs(shb23.Inner_Prod(X, g[0], dom) - o.inner(X, go[0])) <= RTOL * abs(o.inner(X, go[0]))


# ---- any N: the kernels' run-time-length form (csrc/shb23.hip, NH = 0) ------------------------------------------------------------
def test_any_length_transforms_match_reference_helpers(monkeypatch):
    """The full-length (Makhoul N-point) DCT pair of the any-N path against the vectors the REFERENCE's helpers produced: N = 8 (a length
    without an instantiation) and N = 512 forced through the any-N kernels."""
    g = np.load(os.path.join(GOLDEN, "shb_helpers.npz"))
    monkeypatch.setenv("SMO_SHB_ANY", "1")
    for N in (8, 512):
        dom = shb23.SHBDomain(N)
        v = g["v%d" % N]
        assert rel(shb23.transform(v, dom), g["T%d" % N]) < 1e-13
        assert rel(shb23.transformInverse(v, dom), g["Tinv%d" % N]) < 1e-13
        assert rel(shb23.transformAdjoint(v, dom), g["Tadj%d" % N]) < 1e-13
        assert rel(shb23.transformInverseAdjoint(v, dom), g["Tinvadj%d" % N]) < 1e-13
        assert np.array_equal(shb23.weightMatrixDisc(dom), g["W%d" % N])
        assert abs(shb23.Inner_Prod_Discrete(v, g["T%d" % N], dom) - g["ip%d" % N]) < 1e-13 * abs(g["ip%d" % N])


@pytest.mark.parametrize("N,n", [(20, 30), (33, 30), (50, 40), (100, 60), (127, 30), (250, 40), (500, 20), (1000, 8), (1023, 6)])
def test_any_n_vs_oracle(N, n):
    """Grid lengths without an instantiation: other prime factors, odd N, a prime N."""
    test_forward_adjoint_vs_oracle(N, n)


@pytest.mark.parametrize("N,n", [(20, 30), (50, 40), (100, 60), (250, 20), (333, 10)])
def test_any_n_continuous_vs_oracle(N, n):
    test_continuous_forward_adjoint_vs_oracle(N, n)


@pytest.mark.parametrize("N,cont", [(64, False), (512, False), (128, True)])
def test_any_length_kernels_match_the_instantiated_ones(N, cont, monkeypatch):
    """SMO_SHB_ANY=1 sends an instantiated length through the any-N form (one workgroup, full-length transforms): same J, gradient and last
    snapshot to rounding (N = 512 / 256 modes: against the cluster mode)."""
    from oracle import shb23 as osh
    n = 40
    o = (osh.SHB23CntsOracle if cont else osh.SHB23Oracle)(N, dt=1e-2, N_ITERS=n)
    X = (osh.synthetic_ic_cnts if cont else osh.synthetic_ic)(o, 42, 0.0019)
    res = []
    for force in ("0", "1"):
        monkeypatch.setenv("SMO_SHB_ANY", force)
        dom = shb23.SHBDomain(N, dealias=2) if cont else shb23.SHBDomain(N)
        buf = shb23.GEN_BUFFER(N, dom, n)
        if cont:
            J = shb23.FWD_Solve_IVP_Cnts([X], dom, buf, n, 1e-2); g = shb23.ADJ_Solve_IVP_Cnts([X], dom, buf, n, 1e-2)[0]
        else:
            J = shb23.FWD_Solve_IVP_Discrete([X], dom, buf, n, 1e-2); g = shb23.ADJ_Solve_IVP_Discrete([X], dom, buf, n, 1e-2)[0]
        res.append((J, g, np.array(buf['A_fwd'][:, -1])))
    (J0, g0, s0), (J1, g1, s1) = res
    assert abs(J1 - J0) <= 1e-11 * abs(J0) and rel(g1, g0) < 1e-9 and rel(s1, s0) < 1e-10


def test_config3_against_committed_oracle_output():
    """BASELINE config 3: N=512 (Npts=256 x dealias 2), dt=0.01, T=20 (2000 steps)."""
    gold = np.load(os.path.join(GOLDEN, "oracle_shb23_c3.npz"))
    dom = shb23.SHBDomain(512)
    buf = shb23.GEN_BUFFER(512, dom, 2000)
    X = gold["X"]
    J = shb23.FWD_Solve([X], dom, buf, 2000)
    g = shb23.ADJ_Solve([X], dom, buf, 2000)[0]
    assert abs(J - gold["J"]) <= RTOL * abs(gold["J"])
    assert rel(g, gold["grad"]) < RTOL
    assert rel(buf['A_fwd'][:, -1], gold["stack_last"]) < 1e-8


def test_taylor_and_ic_generation():
    dom, X = shb23.Generate_IC(128, M_0=0.0019, seed=42)
    _, dX = shb23.Generate_IC(128, M_0=0.0019, seed=7)
    assert abs(shb23.Inner_Prod(X, X, dom) - 0.0019) < 1e-15
    c = shb23.transform(X, dom)                   # the prepared IC satisfies u(20) = 0 (sum of T coefficients)
    assert abs(c.sum()) < 1e-10 * np.abs(c).max()
    buf = shb23.GEN_BUFFER(128, dom, 100)
    AA = taylor_table([X], [dX], shb23.FWD_Solve, shb23.ADJ_Solve, shb23.Inner_Prod, (dom, buf, 100), (dom, 'np_vector'),
                      epsilon=1e-3)
    assert np.all(np.abs(AA[4, :4] - 2.0) < 5e-3), AA


@pytest.mark.parametrize("N", [64, 50])      # 50: no instantiation, the any-N form
def test_batch_and_errors(N):
    from oracle import shb23 as osh
    n, B = 20, 3
    o = osh.SHB23Oracle(N, dt=1e-2, N_ITERS=n)
    Xs = np.stack([osh.synthetic_ic(o, s, 0.0019) for s in range(B)])
    ctx = shb23.SHBDomain(N).context(1e-2, n, batch=B)
    J = ctx.forward([Xs]); g = ctx.adjoint(None)[0].reshape(B, -1)
    for b in range(B):
        Jo = o.forward([Xs[b]]); go = o.adjoint([Xs[b]])[0]
        assert abs(J[b] - Jo) <= RTOL * abs(Jo) and rel(g[b], go) < RTOL
    with pytest.raises(_capi.SmoError):
        ctx.adjoint(None, "Continuous")
    with pytest.raises(_capi.SmoError):
        _capi.Context(_capi.SMO_SHB23, 1025, (-20., 20.), 1e-2, 5, -0.1)      # one workgroup of 1024 threads per problem: N <= 1024


# ---- "Continuous" formulation (Npts modes, scale-2 grid vectors; FWD_Solve_SHB23.py:398-523, 685-794) --------------------------

@pytest.mark.parametrize("N,n", [(32, 30), (64, 100), (256, 60), (512, 10), (48, 40), (192, 30), (384, 10)])
def test_continuous_forward_adjoint_vs_oracle(N, n):
    from oracle import shb23 as osh
    o = osh.SHB23CntsOracle(N, dt=1e-2, N_ITERS=n)
    X = osh.synthetic_ic_cnts(o, 42, 0.0019)
    dom = shb23.SHBDomain(N, dealias=2)
    buf = shb23.GEN_BUFFER(N, dom, n)
    J = shb23.FWD_Solve_IVP_Cnts([X], dom, buf, n, 1e-2)
    g = shb23.ADJ_Solve_IVP_Cnts([X], dom, buf, n, 1e-2)
    Jo = o.forward([X]); go = o.adjoint([X])
    assert abs(J - Jo) <= RTOL * abs(Jo), (J, Jo)
    assert len(g) == 1 and g[0].shape == (2 * N,) and rel(g[0], go[0]) < RTOL, rel(g[0], go[0])
    for i in (0, 1, -1, -2):
        assert rel(buf['A_fwd'][:, i], o.stack[:, i]) < 1e-8
    ip = o.inner(X, go[0])
    assert abs(shb23.Inner_Prod_Cnts(X, g[0], dom) - ip) <= RTOL * abs(ip)
    assert abs(shb23.Inner_Prod_Cnts(X, X, dom) - 0.0019) < 1e-12


def test_continuous_cluster_and_single_workgroup_agree():
    """Npts=256 runs the 32-workgroup cluster kernel by default; SMO_SHB_CLUSTER=0 forces the single-workgroup one."""
    from oracle import shb23 as osh
    N, n = 256, 40
    o = osh.SHB23CntsOracle(N, dt=1e-2, N_ITERS=n)
    X = osh.synthetic_ic_cnts(o, 3, 0.0019)
    res = []
    for mode in ("1", "0"):
        os.environ["SMO_SHB_CLUSTER"] = mode
        try:
            ctx = _capi.Context(_capi.SMO_SHB23, N, (-20., 20.), 1e-2, n, -0.1, cost=1)
            res.append((ctx.forward([X]), ctx.adjoint(None, "Continuous")[0]))
        finally:
            os.environ.pop("SMO_SHB_CLUSTER")
    assert abs(res[0][0] - res[1][0]) <= 1e-12 * abs(res[1][0])
    assert rel(res[0][1], res[1][1]) < 1e-10


def test_continuous_batch_ic_and_errors():
    from oracle import shb23 as osh
    N, n, B = 64, 20, 3
    o = osh.SHB23CntsOracle(N, dt=1e-2, N_ITERS=n)
    Xs = np.stack([osh.synthetic_ic_cnts(o, s, 0.0019) for s in range(B)])
    dom = shb23.SHBDomain(N, dealias=2)
    ctx = dom.context(1e-2, n, batch=B)
    J = ctx.forward([Xs]); g = ctx.adjoint(None, "Continuous")[0].reshape(B, -1)
    for b in range(B):
        Jo = o.forward([Xs[b]]); go = o.adjoint([Xs[b]])[0]
        assert abs(J[b] - Jo) <= RTOL * abs(Jo) and rel(g[b], go) < RTOL
    with pytest.raises(_capi.SmoError):
        ctx.adjoint(None, "Discrete")
    # device IC recipe = the oracle's
    dom2, X = shb23.Generate_IC_Cnts(N, M_0=0.0019, seed=42)
    assert rel(X, osh.synthetic_ic_cnts(o, 42, 0.0019)) < 1e-9
    # the continuous adjoint is only O(dt)-consistent with the discrete cost (the oracle shows 0.6 % at dt = 0.01, T = 1): compare the
    # directional derivative with a central difference of the device forward solve instead of demanding Taylor slope 2
    _, dX = shb23.Generate_IC_Cnts(N, M_0=0.0019, seed=7)
    buf = shb23.GEN_BUFFER(N, dom2, 100)
    shb23.FWD_Solve_IVP_Cnts([X], dom2, buf, 100)
    d = shb23.Inner_Prod_Cnts(shb23.ADJ_Solve_IVP_Cnts([X], dom2, buf, 100)[0], dX, dom2)
    e = 1e-5
    fd = (shb23.FWD_Solve_IVP_Cnts([X + e * dX], dom2, buf, 100) - shb23.FWD_Solve_IVP_Cnts([X - e * dX], dom2, buf, 100)) / (2 * e)
    assert abs(d - fd) < 1e-2 * abs(fd), (d, fd)


@pytest.mark.parametrize("cost,adj", [(0, "Discrete"), (1, "Continuous")])
def test_cluster_timeout_reruns_with_one_workgroup(monkeypatch, cost, adj):
    """A cluster member that does not see the others arrive in time (a GPU shared with other work) must not turn a correct call into an
    error: the call is rerun with one workgroup per problem.  SMO_SHB_SPIN_LOG2=0 caps the wait at ONE poll, which forces the time-out
    path; the result must be exactly what a context without the cluster (SMO_SHB_CLUSTER=0) returns, and the fallback is counted."""
    from oracle import shb23 as osh
    N, n = (512, 60) if cost == 0 else (256, 60)
    if cost == 0:
        X = osh.synthetic_ic(osh.SHB23Oracle(N, dt=1e-2, N_ITERS=n), 42, 0.0019)
    else:
        X = osh.synthetic_ic_cnts(osh.SHB23CntsOracle(N, dt=1e-2, N_ITERS=n), 3, 0.0019)
    monkeypatch.setenv("SMO_SHB_CLUSTER", "0")
    ref = _capi.Context(_capi.SMO_SHB23, N, (-20., 20.), 1e-2, n, -0.1, cost=cost)
    J0, g0 = ref.forward([X]), ref.adjoint(None, adj)[0]
    assert ref.get(1) == 0
    monkeypatch.delenv("SMO_SHB_CLUSTER")
    healthy = _capi.Context(_capi.SMO_SHB23, N, (-20., 20.), 1e-2, n, -0.1, cost=cost)
    J1, g1 = healthy.forward([X]), healthy.adjoint(None, adj)[0]
    assert healthy.get(1) == 0                                    # an idle GPU holds the whole cluster: no fallback
    assert abs(J1 - J0) <= 1e-12 * abs(J0) and rel(g1, g0) < 1e-10
    monkeypatch.setenv("SMO_SHB_SPIN_LOG2", "0")
    ctx = _capi.Context(_capi.SMO_SHB23, N, (-20., 20.), 1e-2, n, -0.1, cost=cost)
    J2 = ctx.forward([X])
    g2 = ctx.adjoint(None, adj)[0]
    assert ctx.get(1) >= 1, "the one-poll cap did not trigger the time-out path"
    assert J2 == J0 and np.array_equal(g2, g0)
    J3 = ctx.forward([X])                                         # the context now stays with one workgroup per problem
    assert J3 == J0 and ctx.get(1) == 1


@pytest.mark.parametrize("N,cost,rows", [(1000, 0, None), (1023, 0, None), (1024, 0, None), (257, 0, None), (333, 1, None), (301, 1, None), (512, 1, None), (512, 0, "7"), (300, 0, "1")])
def test_cluster_takes_any_length(monkeypatch, N, cost, rows):
    """The cluster mode is not tied to the instantiated lengths or to row counts that divide: any N with >= 256 modes (odd and prime-free
    ones, N = 1024 whose work area leaves room for 6 rows only, the Continuous formulation's Nc = N/2 operator) spreads the operator's rows
    over ceil(Nc / R) workgroups, the last one holding the remainder.  Every row is still summed by one wave in the same order, so the result
    equals the single-workgroup kernel's up to the GEMV's summation order."""
    from oracle import shb23 as osh
    n = 12
    adj = "Continuous" if cost else "Discrete"
    if cost:
        X = osh.synthetic_ic_cnts(osh.SHB23CntsOracle(N, dt=1e-2, N_ITERS=n), 3, 0.0019)
    else:
        X = osh.synthetic_ic(osh.SHB23Oracle(N, dt=1e-2, N_ITERS=n), 42, 0.0019)
    monkeypatch.setenv("SMO_SHB_CLUSTER", "0")
    ref = _capi.Context(_capi.SMO_SHB23, N, (-20., 20.), 1e-2, n, -0.1, cost=cost)
    assert ref.get(2) == 1
    J0, g0 = ref.forward([X]), ref.adjoint(None, adj)[0]
    monkeypatch.delenv("SMO_SHB_CLUSTER")
    if rows:
        monkeypatch.setenv("SMO_SHB_CLUSTER_ROWS", rows)
    ctx = _capi.Context(_capi.SMO_SHB23, N, (-20., 20.), 1e-2, n, -0.1, cost=cost)
    kc = int(ctx.get(2))
    assert kc > 1 and (not rows or kc == -(-N // int(rows)))
    J1, g1 = ctx.forward([X]), ctx.adjoint(None, adj)[0]
    assert ctx.get(1) == 0 and ctx.get(2) == kc                    # no time-out, still a cluster
    assert abs(J1 - J0) <= 1e-11 * abs(J0) and rel(g1, g0) < 1e-9
    snap0, snap1 = ref.snapshot(n), ctx.snapshot(n)
    assert rel(snap1, snap0) < 1e-10

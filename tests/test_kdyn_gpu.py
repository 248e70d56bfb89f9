"""KDyn HIP path (through the C-ABI) against the oracle.  Tolerance: north_star's 1e-6 relative on J and grad J."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from spheremanopt_amd import _capi, kdyn
from spheremanopt_amd.test_grad import taylor_table

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


def _oracle(N, Rm, dt, n, cost):
    from oracle.kdyn import KDynOracle
    return KDynOracle(N, Rm=Rm, dt=dt, N_ITERS=n, Cost_function=cost)


def _fields(G, dirty=False):
    B = kdyn.synthetic_field(G, 1); U = kdyn.synthetic_field(G, 2)
    if dirty:       # not solenoidal, non-zero mean, full spectrum: exercises truncation, k=0 and k.B != 0 branches
        B = B + 0.2 * np.random.RandomState(9).standard_normal(B.size) + 0.05
        U = U + 0.2 * np.random.RandomState(10).standard_normal(U.size)
    return B, U


_SMALL = [(8, 3, 1e-2, True), (16, 6, 1e-2, True), (32, 4, 1e-3, False), (64, 2, 1e-3, True),
          (20, 3, 1e-2, True), (40, 3, 1e-3, True),         # G = 30, 60: lengths with a radix-5 stage
          (12, 3, 1e-2, True), (36, 2, 1e-3, True),         # G = 18, 54: the factor 3 more than once
          (28, 3, 1e-2, True)]                              # G = 42: a radix-7 stage
_ALL4 = [(c, a) for c in ("Final", "Integrated") for a in ("Discrete", "Continuous")]
# every cost / adjoint combination inline up to 64^3; the default one also at 96^3 and 128^3 (the other three at 128^3 — and 256^3 — are
# compared with the committed oracle fixtures below: 50 / 2 steps of all four, test_config4_fixture / test_config5_fixture)
_CASES = [(N, n, dt, d, c, a) for (N, n, dt, d) in _SMALL for (c, a) in _ALL4] + \
         [(96, 1, 1e-3, False, "Final", "Discrete"), (96, 1, 1e-3, True, "Integrated", "Continuous"), (128, 2, 1e-3, False, "Final", "Discrete"),
          (80, 2, 1e-3, True, "Final", "Discrete"), (80, 2, 1e-3, False, "Integrated", "Continuous"), (160, 1, 1e-3, True, "Final", "Discrete"),
          (60, 2, 1e-3, True, "Final", "Discrete"), (60, 2, 1e-3, False, "Integrated", "Continuous"), (72, 1, 1e-3, True, "Integrated", "Discrete"),
          (100, 1, 1e-3, True, "Final", "Continuous"), (120, 1, 1e-3, True, "Final", "Discrete"),      # G = 90, 108, 150, 180
          (56, 2, 1e-3, True, "Integrated", "Discrete"), (112, 1, 1e-3, True, "Final", "Discrete")]      # G = 84, 168


@pytest.mark.parametrize("N,n,dt,dirty,cost,adj", _CASES)
def test_forward_adjoint_vs_oracle(N, n, dt, dirty, cost, adj):
    dom = kdyn.KDynDomain(N)
    B, U = _fields(dom.G, dirty)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args = [dom, 1.3, dt, n, n, buf, cost, adj]
    J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    gB, gU = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
    o = _oracle(N, 1.3, dt, n, cost)
    Jo = o.forward([B, U]); goB, goU = o.adjoint([B, U], adj)
    assert abs(J - Jo) <= RTOL * abs(Jo), (J, Jo)
    assert gB.shape == (3 * dom.G ** 3,) and rel(gB, goB) < RTOL and rel(gU, goU) < RTOL, (rel(gB, goB), rel(gU, goU))
    for comp, key in enumerate(('A_fwd', 'B_fwd', 'C_fwd')):      # snapshots in the reference's GEN_BUFFER indexing
        for i in (0, -1, -2):
            assert rel(buf[key][:, :, :, i], o.stack[comp][..., i]) < 1e-9
    ip, ipo = kdyn.Inner_Prod_3(B, gB, dom), o.inner(B, goB)
    assert abs(ip - ipo) <= RTOL * abs(ipo)
    dom.drop_contexts()


@pytest.mark.parametrize("cost", ["Final", "Integrated"])
def test_against_committed_oracle_output(cost):
    gold = np.load(os.path.join(GOLDEN, "oracle_kdyn_n32_%s.npz" % cost.lower()))
    N, n = int(gold["N"]), int(gold["steps"])
    dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args = [dom, 1., 1e-3, n, n, buf, cost, "Discrete"]
    J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    gB, gU = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
    assert abs(J - gold["J"]) <= RTOL * abs(gold["J"])
    idx = gold["idx"]
    assert np.linalg.norm(gB[idx] - gold["gB"]) <= RTOL * np.linalg.norm(gold["gB"])
    assert np.linalg.norm(gU[idx] - gold["gU"]) <= RTOL * np.linalg.norm(gold["gU"])
    assert abs(np.linalg.norm(gB) - gold["gB_norm"]) <= RTOL * gold["gB_norm"]
    assert abs(np.linalg.norm(gU) - gold["gU_norm"]) <= RTOL * gold["gU_norm"]
    dom.drop_contexts()


_FIELD_CACHE = {}


def _bench_fields(G):
    """The seeded synthetic fields of SURVEY 8d (B: seed 1, U: seed 2) — what tools/gen_golden_kdyn_big.py fed the oracle."""
    if G not in _FIELD_CACHE:
        _FIELD_CACHE.clear()                               # one grid at a time: 2 x 1.36 GB at G = 384
        _FIELD_CACHE[G] = (kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2))
    return _FIELD_CACHE[G]


def _check_against_fixture(gold, key, J, gB, gU):
    idx = gold["idx"]
    Jo = float(gold["J_" + key.split("_")[0]])
    assert abs(J - Jo) <= RTOL * abs(Jo), (key, J, Jo)
    for name, g in (("gB", gB), ("gU", gU)):
        ref = gold["%s_%s" % (key, name)]
        assert np.linalg.norm(g[idx] - ref) <= RTOL * np.linalg.norm(ref), (key, name)
        nrm = float(gold["%s_%s_norm" % (key, name)])
        assert abs(np.linalg.norm(g) - nrm) <= RTOL * nrm, (key, name)
        # a functional that sees cancellations the norm does not: the seeded projection stored by the generator
        w = np.random.RandomState(77).standard_normal(4096)
        proj = float(np.dot(g[:: max(1, g.size // 4096)][:4096], w))
        assert abs(proj - float(gold["%s_%s_proj" % (key, name)])) <= RTOL * nrm, (key, name)


def _fixture_run(fixture, ckpt=1):
    """HIP path (single GPU, through the C-ABI) against a committed north-star-size oracle fixture, every cost / adjoint combination."""
    gold = np.load(os.path.join(GOLDEN, fixture))
    N, n, dt, Rm = int(gold["N"]), int(gold["steps"]), float(gold["dt"]), float(gold["Rm"])
    dom = kdyn.KDynDomain(N, ckpt=ckpt)
    B, U = _bench_fields(dom.G)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    for cost in ("Final", "Integrated"):
        for adj in ("Discrete", "Continuous"):
            args = [dom, Rm, dt, n, n, buf, cost, adj]
            J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
            gB, gU = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
            _check_against_fixture(gold, "%s_%s" % (cost, adj), J, gB, gU)
        if cost == "Final":                                # the trajectory itself: last / middle snapshot of the stack
            last = dom.context(Rm, dt, n, cost).snapshot(n).view(np.complex128)
            assert abs(np.linalg.norm(last) - float(gold["snap_last_norm"])) <= 1e-9 * float(gold["snap_last_norm"])
            assert np.linalg.norm(last[::9973] - gold["snap_last_sample"]) <= 1e-9 * np.linalg.norm(gold["snap_last_sample"])
            assert abs(last.sum() - complex(gold["snap_last_sum"])) <= 1e-9 * float(gold["snap_last_norm"]) * np.sqrt(last.size)
            mid = dom.context(Rm, dt, n, cost).snapshot(n // 2).view(np.complex128)
            assert abs(np.linalg.norm(mid) - float(gold["snap_mid_norm"])) <= 1e-9 * float(gold["snap_mid_norm"])
        dom.drop_contexts()


def test_config4_fixture():
    """BASELINE configs[3] (128^3, Rm = 1, dt = 1e-3): 50 of its 1000 steps, all four cost / adjoint combinations, against the oracle
    run committed as tests/golden/oracle_kdyn_c4_128_n50.npz (tools/gen_golden_kdyn_big.py; the oracle needs ~15 min for it)."""
    _fixture_run("oracle_kdyn_c4_128_n50.npz")


def test_config4_fixture_with_windowed_checkpoints():
    """The same 50 steps with every 7th snapshot kept (windows that do not divide 50) — the path 256^3 x 1000 steps takes on one GPU."""
    _fixture_run("oracle_kdyn_c4_128_n50.npz", ckpt=7)


def test_config4_full_length_fixture():
    """BASELINE configs[3] at its full length — 128^3, 1000 steps, the very gradient `bench.py` times — against the oracle run committed as
    tests/golden/oracle_kdyn_c4_128_n1000.npz (tools/gen_golden_kdyn_full.py: two hours of CPU, 55 GB of RAM): J of both cost functionals
    and the discrete-adjoint gradients with respect to B0 and U."""
    path = os.path.join(GOLDEN, "oracle_kdyn_c4_128_n1000.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/oracle_kdyn_c4_128_n1000.npz not generated (tools/gen_golden_kdyn_full.py: 2 h of CPU, 55 GB of RAM)")
    gold = np.load(path)
    N, n, dt, Rm = int(gold["N"]), int(gold["steps"]), float(gold["dt"]), float(gold["Rm"])
    assert (N, n) == (128, 1000)
    dom = kdyn.KDynDomain(N)
    B, U = _bench_fields(dom.G)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    for cost in ("Final", "Integrated"):
        if "%s_Discrete_gB" % cost not in gold.files:
            continue
        args = [dom, Rm, dt, n, n, buf, cost, "Discrete"]
        J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
        gB, gU = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
        _check_against_fixture(gold, "%s_Discrete" % cost, J, gB, gU)
        if cost == "Final":
            for k in (n // 4, n // 2, n):
                snap = dom.context(Rm, dt, n, cost).snapshot(k).view(np.complex128)
                ref = float(gold["snap_%d_norm" % k])
                assert abs(np.linalg.norm(snap) - ref) <= 1e-9 * ref
                assert np.linalg.norm(snap[::9973] - gold["snap_%d_sample" % k]) <= 1e-9 * np.linalg.norm(gold["snap_%d_sample" % k])
        dom.drop_contexts()


def test_config5_fixture():
    """BASELINE configs[4]'s grid (256^3: the G = 384 kernel instantiations — half tiles, half twiddle table, narrower adjoint x tiles)
    on ONE GPU against the oracle: tests/golden/oracle_kdyn_c5_256_n2.npz, all four combinations."""
    _fixture_run("oracle_kdyn_c5_256_n2.npz")


@pytest.mark.parametrize("ckpt", [2, 4])
def test_config5_fixture_more_steps_with_windowed_checkpoints(ckpt):
    """256^3, 9 steps, all four combinations, with the checkpoint windows (and the Ty cache of a window) the 1000-step run on one GPU uses:
    tests/golden/oracle_kdyn_c5_256_n9.npz (tools/gen_golden_kdyn_big.py --npts 256 --steps 9).  Windows of 2 and of 4 steps (4 does not
    divide 9): the composition of the G = 384 kernels with the recomputation, which the 2-step fixture cannot show."""
    if not os.path.exists(os.path.join(GOLDEN, "oracle_kdyn_c5_256_n9.npz")):
        pytest.skip("tests/golden/oracle_kdyn_c5_256_n9.npz not generated (tools/gen_golden_kdyn_big.py --npts 256 --steps 9: half an hour of CPU)")
    _fixture_run("oracle_kdyn_c5_256_n9.npz", ckpt=ckpt)


def test_taylor_remainder_two_fields():
    """Adjoint_Gradient_Test on the HIP path with both dB and dU perturbed."""
    N, n, dt = 16, 10, 1e-2
    dom = kdyn.KDynDomain(N)
    B, U = _fields(dom.G, dirty=True)
    dB, dU = kdyn.synthetic_field(dom.G, 3), kdyn.synthetic_field(dom.G, 4)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args_f = [dom, 1., dt, n, n, buf, "Final", "Discrete"]
    AA = taylor_table([B, U], [dB, dU], kdyn.FWD_Solve_IVP_Lin, kdyn.ADJ_Solve_IVP_Lin, kdyn.Inner_Prod_3, args_f, (dom, None),
                      epsilon=1e-3)
    assert np.all(np.abs(AA[4, :4] - 2.0) < 5e-3), AA
    dom.drop_contexts()


def test_size_independent_properties():
    """Properties that hold at any size: div-free IC stays div-free (checked through the spectral snapshots),
    J(Final) equals the spectral energy of the last snapshot, dJ/dU is solenoidal."""
    N, n = 32, 5
    dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args = [dom, 1., 1e-3, n, n, buf, "Final", "Discrete"]
    J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    kx = np.arange(dom.a, dtype=float)
    kc = np.concatenate([np.arange(0, dom.kmax + 1), np.arange(-dom.kmax, 0)]).astype(float)
    K = np.stack(np.meshgrid(kx, kc, kc, indexing='ij'))
    last = np.stack([buf[k][:, :, :, -1] for k in ('A_fwd', 'B_fwd', 'C_fwd')])
    assert np.abs((K * last).sum(0)).max() < 1e-12
    w = np.where(K[0] == 0, 1., 2.)
    assert abs(-J - (w * np.abs(last) ** 2).sum()) < 1e-12 * abs(J)
    dom.drop_contexts()


@pytest.mark.parametrize("N", [32, 192, 256, 160, 320, 144, 200, 240, 224])
def test_known_answer_single_mode_decay(N):
    """U = 0, B = (0, cos 3x, 0): every CNAB1 step multiplies the mode by (1/dt - 9/2Rm)/(1/dt + 9/2Rm) — an answer that does not
    come from the oracle.  Also dJ/dU = 0 and dJ/dB0 = -2 r^(2N) B0 for the Final cost.  N = 192 (G = 288 = 4*4*2*3*3): a size whose
    oracle run would take minutes is checked through this closed form; N = 256 (G = 384): the north-star grid's kernel instantiations; N = 160, 320 (G = 240, 480 = 4*4*[2*]5*3): the radix-5 sizes; N = 144, 200, 240 (G = 216, 300, 360): the largest of the sizes with repeated factors 3 and 5; N = 224 (G = 336 = 4*4*7*3): the largest radix-7 size."""
    n, dt, Rm = 9, 1e-2, 1.3
    dom = kdyn.KDynDomain(N)
    G = dom.G
    x = 2. * np.pi * np.arange(G) / G
    B = np.zeros((3, G, G, G)); B[1] = np.cos(3. * x)[:, None, None]
    B = B.reshape(-1); U = np.zeros(3 * G ** 3)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    args = [dom, Rm, dt, n, n, buf, "Final", "Discrete"]
    r = (1. / dt - 9. / (2. * Rm)) / (1. / dt + 9. / (2. * Rm))
    J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
    assert abs(J + 0.5 * r ** (2 * n)) < 1e-13
    gB, gU = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
    assert np.abs(gB + 2. * r ** (2 * n) * B).max() < 1e-12 and np.abs(gU).max() < 1e-12
    # uniform flow U = (c,0,0): explicit advection => |factor|^2 = ((1/dt - D/2)^2 + 9 c^2) / (1/dt + D/2)^2 per step (exercises the
    # grid cross product, the curl and the projection against a closed form)
    c = 0.7
    Uc = np.zeros((3, G, G, G)); Uc[0] = c
    J = kdyn.FWD_Solve_IVP_Lin([B, Uc.reshape(-1)], *args)
    r2 = ((1. / dt - 9. / (2. * Rm)) ** 2 + 9. * c * c) / (1. / dt + 9. / (2. * Rm)) ** 2
    assert abs(J + 0.5 * r2 ** n) < 1e-13
    dom.drop_contexts()


# ---- any even Npts: the run-time-length kernels (csrc/kdyn_any.hpp) ----------------------------------------------------------------
# The reference builds its Fourier bases for whatever Npts it is handed (FWD_Solve_KDyn.py:362-450).  Sizes without a tuned
# instantiation: prime factors 11, 13, 37, 53 of G = 3 Npts / 2, and Npts = 2 mod 4, where G is ODD (the last (y,z) line of a plane has
# no partner in the two-lines-per-transform packing).
_ANY_SMALL = [(6, 3, 1e-2, True), (10, 3, 1e-2, True), (22, 3, 1e-2, True), (44, 2, 1e-3, True)]      # G = 9, 15, 33, 66
_ANY_CASES = [(N, n, dt, d, c, a) for (N, n, dt, d) in _ANY_SMALL for (c, a) in _ALL4] + \
             [(14, 2, 1e-2, False, "Final", "Discrete"), (18, 2, 1e-2, True, "Integrated", "Continuous"), (26, 2, 1e-2, True, "Final", "Continuous"),
              (52, 1, 1e-3, True, "Integrated", "Discrete"), (74, 1, 1e-3, False, "Final", "Discrete"), (106, 1, 1e-3, True, "Final", "Discrete")]


@pytest.mark.parametrize("N,n,dt,dirty,cost,adj", _ANY_CASES)
def test_any_even_npts_vs_oracle(N, n, dt, dirty, cost, adj):
    test_forward_adjoint_vs_oracle(N, n, dt, dirty, cost, adj)


@pytest.mark.parametrize("N,ckpt", [(16, 1), (24, 1), (16, 3), (40, 1)])
@pytest.mark.parametrize("cost,adj", [("Final", "Discrete"), ("Integrated", "Continuous")])
def test_runtime_length_kernels_match_the_tuned_ones(N, ckpt, cost, adj, monkeypatch):
    """SMO_KD_ANY=1 sends a tuned size through the run-time-length kernels: same J, gradients and snapshots to rounding (different
    butterfly order, no fusion), with and without checkpoint windows."""
    n = 7
    B, U = _fields(3 * N // 2, dirty=True)
    res = []
    for force in ("0", "1"):
        monkeypatch.setenv("SMO_KD_ANY", force)
        dom = kdyn.KDynDomain(N)
        dom.ckpt = ckpt
        buf = kdyn.GEN_BUFFER(N, dom, n)
        args = [dom, 1.0, 1e-2, n, n, buf, cost, adj]
        J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
        gB, gU = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
        last = None if ckpt != 1 else np.stack([buf[k][:, :, :, -1] for k in ('A_fwd', 'B_fwd', 'C_fwd')])
        res.append((J, gB, gU, last))
        dom.drop_contexts()
    (J0, gB0, gU0, s0), (J1, gB1, gU1, s1) = res
    assert abs(J1 - J0) <= 1e-12 * abs(J0)
    assert rel(gB1, gB0) < 1e-11 and rel(gU1, gU0) < 1e-11
    if s0 is not None:
        assert rel(s1, s0) < 1e-12


@pytest.mark.parametrize("N", [44, 310])
def test_known_answer_single_mode_decay_any_size(N):
    """The closed-form CNAB1 answers at sizes only the run-time-length kernels take: G = 66 = 2*3*11, and G = 465 = 3*5*31 — an odd
    grid whose adjoint x pass needs more than 64 KB of LDS per workgroup (the opt-in above the default limit)."""
    test_known_answer_single_mode_decay(N)


def test_errors():
    with pytest.raises(_capi.SmoError):
        _capi.Context(_capi.SMO_KDYN, 45, (0., 2 * np.pi), 1e-3, 2, 1.0)      # odd Npts: G = 3 Npts / 2 is not an integer
    ctx = _capi.Context(_capi.SMO_KDYN, 8, (0., 2 * np.pi), 1e-3, 2, 1.0)
    with pytest.raises(_capi.SmoError) as e:
        ctx.adjoint(None)
    assert e.value.code == 4


@pytest.mark.parametrize("ckpt,n", [(2, 7), (3, 7), (4, 8), (7, 7), (0, 5)])
@pytest.mark.parametrize("cost,adj", [("Final", "Discrete"), ("Integrated", "Continuous")])
def test_windowed_checkpointing_is_bit_identical(ckpt, n, cost, adj):
    """Keeping every k-th snapshot and recomputing the windows must give exactly the gradients of the keep-all run
    (same kernels on the same data), for window sizes that do / do not divide N_ITERS."""
    N = 16
    ref = kdyn.KDynDomain(N)
    B, U = _fields(ref.G, dirty=True)
    bufr = kdyn.GEN_BUFFER(N, ref, n)
    args = [1.0, 1e-2, n, n]
    J0 = kdyn.FWD_Solve_IVP_Lin([B, U], ref, *args, bufr, cost, adj)
    g0 = kdyn.ADJ_Solve_IVP_Lin([B, U], ref, *args, bufr, cost, adj)
    dom = kdyn.KDynDomain(N, ckpt=ckpt)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    J1 = kdyn.FWD_Solve_IVP_Lin([B, U], dom, *args, buf, cost, adj)
    g1 = kdyn.ADJ_Solve_IVP_Lin([B, U], dom, *args, buf, cost, adj)
    assert J1 == J0 and np.array_equal(g1[0], g0[0]) and np.array_equal(g1[1], g0[1])
    g2 = kdyn.ADJ_Solve_IVP_Lin([B, U], dom, *args, buf, cost, adj)          # second adjoint: windows are recomputed again
    assert np.array_equal(g2[0], g0[0])
    for i in (0, 1, n // 2, n - 1, n):                                       # snapshot reads recompute on demand
        assert np.array_equal(buf['A_fwd'][:, :, :, i], bufr['A_fwd'][:, :, :, i])
    if ckpt > 1:
        assert dom.context(*args[:3], cost).stack_bytes < ref.context(*args[:3], cost).stack_bytes or n // ckpt + ckpt >= n + 1
    ref.drop_contexts(); dom.drop_contexts()


@pytest.mark.parametrize("ckpt,n,dense", [(2, 9, 4), (3, 11, 6), (4, 8, 0), (2, 7, 6)])
@pytest.mark.parametrize("cost,adj", [("Final", "Discrete"), ("Integrated", "Continuous")])
def test_checkpoint_schedule_with_a_dense_tail_is_bit_identical(ckpt, n, dense, cost, adj, monkeypatch):
    """The non-uniform schedule of round 4: windows of `ckpt` states up to index `dense`, EVERY state kept from there on (what the HBM left
    over by a uniform interval buys at 256^3 on one GPU).  Same kernels on the same states: J and both gradients equal the keep-all run bit
    for bit, snapshot reads on both sides of the boundary too."""
    N = 16
    ref = kdyn.KDynDomain(N)
    B, U = _fields(ref.G, dirty=True)
    bufr = kdyn.GEN_BUFFER(N, ref, n)
    args = [1.0, 1e-2, n, n]
    J0 = kdyn.FWD_Solve_IVP_Lin([B, U], ref, *args, bufr, cost, adj)
    g0 = kdyn.ADJ_Solve_IVP_Lin([B, U], ref, *args, bufr, cost, adj)
    monkeypatch.setenv("SMO_KD_DENSE_FROM", str(dense))
    dom = kdyn.KDynDomain(N, ckpt=ckpt)
    buf = kdyn.GEN_BUFFER(N, dom, n)
    J1 = kdyn.FWD_Solve_IVP_Lin([B, U], dom, *args, buf, cost, adj)
    ctx = dom.context(*args[:3], cost)
    assert ctx.get(0) == ckpt and ctx.get(5) == dense // ckpt * ckpt
    g1 = kdyn.ADJ_Solve_IVP_Lin([B, U], dom, *args, buf, cost, adj)
    assert J1 == J0 and np.array_equal(g1[0], g0[0]) and np.array_equal(g1[1], g0[1])
    g2 = kdyn.ADJ_Solve_IVP_Lin([B, U], dom, *args, buf, cost, adj)
    assert np.array_equal(g2[0], g0[0]) and np.array_equal(g2[1], g0[1])
    for i in range(n + 1):
        assert np.array_equal(buf['A_fwd'][:, :, :, i], bufr['A_fwd'][:, :, :, i]), i
    monkeypatch.delenv("SMO_KD_DENSE_FROM")
    uni = kdyn.KDynDomain(N, ckpt=ckpt)                     # an explicit interval without the knob stays uniform
    kdyn.FWD_Solve_IVP_Lin([B, U], uni, *args, kdyn.GEN_BUFFER(N, uni, n), cost, adj)
    assert uni.context(*args[:3], cost).get(5) == -1
    ref.drop_contexts(); dom.drop_contexts(); uni.drop_contexts()


def test_reference_ic_recipe_with_device_prep():
    """Generate_IC(reference_recipe=True): curl-type field of filtered seed-42 noise, smoothed by 101 device steps (KDYN:183-317)."""
    N = 16
    dom, B, U = kdyn.Generate_IC(N, U_Noise=True, reference_recipe=True, dt=1e-3)
    o = _oracle(N, 1.0, 1e-3, 101, "Final")
    assert abs(o.inner(B, B) - 1) < 1e-12 and abs(o.inner(U, U) - 1) < 1e-12
    Bh = o.vec_to_coeff(B)
    assert np.abs(o.kdot(Bh)).max() < 1e-12 * np.abs(Bh).max()                   # solenoidal
    # the smoothing solve is the device forward solver: compare with the oracle run from the same raw field
    raw = kdyn._curl_noise(dom, 42).reshape(-1)
    Rh = o.vec_to_coeff(raw)
    assert np.abs(Rh[1:, 1:, o.kmax + 1:, :]).max() < 1e-12 * np.abs(Rh).max()   # index-fraction filter: no negative ky for kx > 0
    assert np.abs(o.kdot(Rh)).max() < 1e-12 * np.abs(Rh).max()
    o.forward([raw, U])
    ref = o.coeff_to_vec(o.stack[..., 101])
    ref *= 1. / np.sqrt(o.inner(ref, ref))
    assert rel(B, ref) < 1e-9
    dom.drop_contexts()
    # analytic flow variant
    _, _, Ua = kdyn.Generate_IC(N, U_Noise=False)
    assert abs(o.inner(Ua, Ua) - 1) < 1e-12


def test_y_side_stack_is_bit_identical(monkeypatch):
    """The forward solve keeps the y-pass output of every state in HBM so the adjoint can skip two kernels per step; turning
    that off (SMO_KD_TYSTACK=0) must give exactly the same numbers, for both adjoint types (Continuous needs state N, which
    the forward solve never transforms, so that one snapshot is always recomputed)."""
    N, n = 32, 5
    B, U = _fields(3 * N // 2, dirty=True)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SMO_KD_TYSTACK", flag)
        dom = kdyn.KDynDomain(N)
        buf = kdyn.GEN_BUFFER(N, dom, n)
        for adj in ("Discrete", "Continuous"):
            args = [dom, 1., 1e-3, n, n, buf, "Integrated", adj]
            J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
            g = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
            res[(flag, adj)] = (J, g)
        ctx = dom.context(1., 1e-3, n, "Integrated")
        assert (ctx.get(1) > 0) == (flag == "1") and ctx.get(0) == 1
        dom.drop_contexts()
    for adj in ("Discrete", "Continuous"):
        a, b = res[("1", adj)], res[("0", adj)]
        assert a[0] == b[0] and np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])


@pytest.mark.parametrize("knob,off", [("SMO_KD_FUSE_NEXT", "0"), ("SMO_KD_TYPAD", "0"), ("SMO_KD_TYPAD", "24"), ("SMO_KD_ADJ_SEQ", "0")])
@pytest.mark.parametrize("N", [32, 48])
def test_layout_and_fusion_knobs_are_bit_identical(monkeypatch, knob, off, N):
    """Pure performance devices must not change a single bit (SMO_KD_ADJ_SEQ: the adjoint x pass with the field groups one after the
    other — the default — against both at once in half-width tiles): (i) the update kernels run the next step's inverse z pass on the tile
    they have just updated, in place in the exchange buffer (SMO_KD_FUSE_NEXT=0: separate kernels); (ii) the Ty planes are padded by one
    128-byte line against HBM channel conflicts (SMO_KD_TYPAD: other paddings / none).  Both adjoint types, with and without the
    grid-side stack (without it the adjoint sends two field groups and the fused pass must step aside)."""
    n = 5
    B, U = _fields(3 * N // 2, dirty=True)
    res = {}
    for stack in ("1", "0"):
        monkeypatch.setenv("SMO_KD_TYSTACK", stack)
        for val in (None, off):
            if val is None:
                monkeypatch.delenv(knob, raising=False)
            else:
                monkeypatch.setenv(knob, val)
            dom = kdyn.KDynDomain(N)
            buf = kdyn.GEN_BUFFER(N, dom, n)
            for adj in ("Discrete", "Continuous"):
                args = [dom, 1., 1e-3, n, n, buf, "Integrated", adj]
                J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
                g = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
                res[(stack, val, adj)] = (J, g, np.stack([buf[k][:, :, :, i] for k in ("A_fwd", "B_fwd", "C_fwd") for i in range(n + 1)]))
            dom.drop_contexts()
        for adj in ("Discrete", "Continuous"):
            a, b = res[(stack, None, adj)], res[(stack, off, adj)]
            assert a[0] == b[0] and np.array_equal(a[2], b[2]), (stack, adj)
            assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1]), (stack, adj)


@pytest.mark.parametrize("knob,off", [("SMO_KD_FUSE_NEXT", "0"), ("SMO_KD_TYPAD", "0"), ("SMO_KD_TYSTACK", "0"), ("SMO_KD_ADJ_SEQ", "0")])
def test_layout_and_fusion_knobs_are_bit_identical_at_G384(monkeypatch, knob, off):
    """The same pure-performance devices at the north-star grid (N = 256: half tiles, half twiddle table, XCD-paired x tiles): fused next z
    pass, Ty plane padding, grid-side stack.  Noise inputs (full spectrum, non-solenoidal: every branch of the per-mode update)."""
    N, n = 256, 3
    G = 3 * N // 2
    rs = np.random.RandomState(5)
    B, U = rs.standard_normal(3 * G ** 3), rs.standard_normal(3 * G ** 3)
    res = []
    for val in (None, off):
        if val is None:
            monkeypatch.delenv(knob, raising=False)
        else:
            monkeypatch.setenv(knob, val)
        dom = kdyn.KDynDomain(N)
        buf = kdyn.GEN_BUFFER(N, dom, n)
        args = [dom, 1., 1e-3, n, n, buf, "Integrated", "Discrete"]
        J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
        g = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
        res.append((J, g, dom.context(1., 1e-3, n, "Integrated").snapshot(n)))
        dom.drop_contexts()
    a, b = res
    assert a[0] == b[0] and np.array_equal(a[2], b[2])
    if knob == "SMO_KD_ADJ_SEQ":
        # two different kernels for the same arithmetic: the compiler is free to contract a*b+c differently in each instantiation, so
        # equality holds to rounding only (it is exact at the sizes of the test above)
        assert rel(a[1][0], b[1][0]) < 1e-13 and rel(a[1][1], b[1][1]) < 1e-13
    else:
        assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])


@pytest.mark.parametrize("cost", ["Final", "Integrated"])
def test_hip_graph_replay_is_bit_identical(monkeypatch, cost):
    """Small grids on one GPU run the whole forward solve / adjoint sweep as ONE captured HIP graph (launch-bound: ~2 us of work per
    kernel).  Same kernels, same order: every number must equal the launch-by-launch path (SMO_KD_GRAPH=0), also on replays with other
    input vectors, for both adjoint types, and with device-resident vectors."""
    from spheremanopt_amd.devvec import to_device
    N, n = 24, 40                                          # the reference script's default grid
    G = 3 * N // 2
    X1 = [kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)]
    X2 = [kdyn.synthetic_field(G, 5) + 0.1, kdyn.synthetic_field(G, 6)]
    res = {}
    for mode in ("0", None):
        if mode is None:
            monkeypatch.delenv("SMO_KD_GRAPH", raising=False)
        else:
            monkeypatch.setenv("SMO_KD_GRAPH", mode)
        dom = kdyn.KDynDomain(N)
        buf = kdyn.GEN_BUFFER(N, dom, n)
        out = []
        for X in (X1, X2, X1, to_device(X2)):
            for adj in ("Discrete", "Continuous"):
                args = [dom, 1., 1e-3, n, n, buf, cost, adj]
                J = kdyn.FWD_Solve_IVP_Lin(X, *args)
                g = kdyn.ADJ_Solve_IVP_Lin(X, *args)
                g = [v.numpy() if hasattr(v, "numpy") else v for v in g]
                out.append((J, g, buf["B_fwd"][:, :, :, n // 2].copy()))
        replays = dom.context(1., 1e-3, n, cost).get(2)
        assert (replays == 0) if mode == "0" else (replays == 16), replays       # 8 forward + 8 adjoint calls, captured at the first of each kind
        res[mode] = out
        dom.drop_contexts()
    for a, b in zip(res["0"], res[None]):
        assert a[0] == b[0] and np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1]) and np.array_equal(a[2], b[2])
    assert res[None][0][0] != res[None][2][0] and res[None][0][0] == res[None][4][0]      # the replays saw their own inputs


def test_hip_graph_is_off_where_it_does_not_apply(monkeypatch):
    dom = kdyn.KDynDomain(128)                             # G = 192 > SMO_KD_GRAPH_MAXG: launch by launch
    B, U = _fields(dom.G)
    ctx = dom.context(1., 1e-3, 2, "Final")
    ctx.forward([B, U]); ctx.adjoint(None)
    assert ctx.get(2) == 0
    dom.drop_contexts()
    small = kdyn.KDynDomain(16, ckpt=2)                    # checkpoint windows: recomputation is decided on the host
    ctx = small.context(1., 1e-3, 6, "Final")
    B, U = _fields(small.G)
    ctx.forward([B, U]); ctx.adjoint(None)
    assert ctx.get(2) == 0
    ctx.timing_enable(True)
    small.drop_contexts()


@pytest.mark.parametrize("N", [16, 40])
def test_device_transform_pair(N):
    """smo_transform for the 3-D case: coefficients -> grid equals the host NumPy transform the IC generator used before, and
    grid -> coefficients inverts it (band-limited input)."""
    dom = kdyn.KDynDomain(N)
    rs = np.random.RandomState(3)
    C = rs.standard_normal((3, dom.a, dom.m, dom.m)) + 1j * rs.standard_normal((3, dom.a, dom.m, dom.m))
    C[:, 0] = 0.5 * (C[:, 0] + np.conj(C[:, 0][:, ::-1, ::-1].take(np.r_[-1, :dom.m - 1], axis=1).take(np.r_[-1, :dom.m - 1], axis=2)))   # Hermitian kx = 0 plane
    g_dev = kdyn._coeff3_to_grid(dom, C)
    g_host = np.stack([kdyn._coeff_to_grid_host(dom, C[i]) for i in range(3)])
    assert rel(g_dev, g_host) < 1e-13
    ctx = dom._transform_ctx
    back = ctx.transform(0, g_dev.reshape(-1), out_len=2 * 3 * dom.a * dom.m * dom.m).view(np.complex128).reshape(C.shape)
    o = _oracle(N, 1., 1e-3, 1, "Final")
    assert rel(back, np.stack([o.to_coeff(g_host[i]) for i in range(3)])) < 1e-13
    dom.drop_contexts()


def test_timing_stride_samples_the_launches():
    """smo_timing_stride: of the selected classes every n-th launch carries events; counts and averages are those of the sample."""
    N, n = 16, 24
    dom = kdyn.KDynDomain(N)
    ctx = dom.context(1., 1e-3, n, "Final")
    B, U = _fields(dom.G, dirty=False)
    ctx.timing_enable(True)
    J0 = ctx.forward([B, U]); ctx.adjoint(None)
    full = {t["kernel"]: t for t in ctx.timing()}
    ctx.timing_enable(True, every=5)
    J1 = ctx.forward([B, U]); ctx.adjoint(None)
    samp = {t["kernel"]: t for t in ctx.timing()}
    assert J1 == J0
    for k, t in full.items():
        assert samp[k]["launches"] == -(-t["launches"] // 5), (k, t["launches"], samp[k]["launches"])
    x = "kd_x_pass<fused_adj>"
    assert full[x]["launches"] == n and samp[x]["total_ms"] > 0
    with pytest.raises(_capi.SmoError):
        ctx.timing_enable(True, every=0)
    ctx.timing_enable(False)                                   # back to every launch for whoever enables timing next
    ctx.timing_enable(True)
    ctx.forward([B, U])
    assert {t["kernel"]: t for t in ctx.timing()}["kd_x_pass<fused_fwd>"]["launches"] == n
    dom.drop_contexts()

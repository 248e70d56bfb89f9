"""SH23 HIP path (through the C-ABI) against the oracle and the committed fixtures.  Tolerance: north_star's
1e-6 relative on J and grad J (observed ~1e-13)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from spheremanopt_amd import _capi, sh23
from spheremanopt_amd.sphere_opt import Optimise_On_Multi_Sphere
from spheremanopt_amd.test_grad import taylor_table

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


def _oracle(Npts, dt, n):
    from oracle.sh23 import SH23Oracle
    return SH23Oracle(Npts, dt=dt, N_ITERS=n)


@pytest.mark.parametrize("Npts,dt,n", [(16, 0.1, 3), (32, 0.05, 17), (64, 0.1, 40), (128, 0.1, 100), (1024, 0.02, 25),
                                       (24, 0.1, 10), (96, 0.1, 60), (384, 0.1, 30), (80, 0.1, 40), (640, 0.05, 20), (60, 0.1, 25), (960, 0.05, 12),
                                       (28, 0.1, 10), (112, 0.1, 30), (896, 0.05, 12),           # 7 * 2^k: a radix-7 stage
                                       (36, 0.1, 10), (72, 0.1, 20), (144, 0.1, 30), (288, 0.1, 20), (576, 0.05, 12),            # 9 * 2^k
                                       (100, 0.1, 40), (200, 0.1, 30), (400, 0.05, 20), (800, 0.02, 12),                         # 25 * 2^k
                                       (150, 0.1, 30), (300, 0.1, 20), (600, 0.05, 12), (500, 0.05, 12)])                        # 75 * 2^k, 125 * 2^k
@pytest.mark.parametrize("adj", ["Discrete", "Continuous"])
def test_forward_adjoint_vs_oracle(Npts, dt, n, adj):
    dom, X = sh23.Generate_IC(0.0725, Npts=Npts, seed=42)
    buf = sh23.GEN_BUFFER(dom, n)
    args = [dom, dt, n, n, buf, None, adj]
    J = sh23.FWD_Solve_IVP_Lin([X], *args)
    g = sh23.ADJ_Solve_IVP_Lin([X], *args)
    o = _oracle(Npts, dt, n)
    Jo = o.forward([X]); go = o.adjoint([X], adj)
    assert abs(J - Jo) <= RTOL * abs(Jo)
    assert len(g) == 1 and g[0].shape == (2 * Npts,) and rel(g[0], go[0]) < RTOL
    # snapshot stack: same contents, reference indexing A_fwd[:, i] incl. negative indices
    for i in (0, 1, n // 2, -1, -2):
        assert rel(buf['A_fwd'][:, i], o.stack[:, i]) < 1e-9
    # inner product
    assert abs(sh23.Inner_Prod(X, g[0], dom) - o.inner(X, go[0])) <= RTOL * abs(o.inner(X, go[0]))
    assert abs(sh23.Inner_Prod(X, X, dom) - 0.0725) < 1e-12


# any Npts: the run-time-length kernels (csrc/sh23.hip, sh23_*_any).  Lengths without an instantiation — prime factors 11, 13, 37, 127, odd
# Npts, a prime Npts — and, above 819, the opt-in beyond 64 KB of LDS per workgroup.
@pytest.mark.parametrize("Npts,dt,n", [(18, 0.1, 10), (21, 0.1, 12), (22, 0.1, 10), (50, 0.1, 40), (54, 0.1, 40), (97, 0.1, 20), (250, 0.1, 60), (242, 0.1, 60),
                                       (254, 0.1, 30), (333, 0.05, 25), (1000, 0.02, 15), (1012, 0.02, 15), (1100, 0.02, 8)])
@pytest.mark.parametrize("adj", ["Discrete", "Continuous"])
def test_any_npts_vs_oracle(Npts, dt, n, adj):
    test_forward_adjoint_vs_oracle(Npts, dt, n, adj)


@pytest.mark.parametrize("Npts", [64, 96, 256])
@pytest.mark.parametrize("adj", ["Discrete", "Continuous"])
def test_any_length_kernels_match_the_instantiated_ones(Npts, adj, monkeypatch):
    """SMO_SH_ANY=1 sends an instantiated length through the any-length kernels: same J, gradient and snapshots to rounding."""
    n, dt = 80, 0.1
    res = []
    for force in ("0", "1"):
        monkeypatch.setenv("SMO_SH_ANY", force)
        dom, X = sh23.Generate_IC(0.0725, Npts=Npts, seed=42)
        buf = sh23.GEN_BUFFER(dom, n)
        args = [dom, dt, n, n, buf, None, adj]
        J = sh23.FWD_Solve_IVP_Lin([X], *args)
        g = sh23.ADJ_Solve_IVP_Lin([X], *args)[0]
        res.append((J, g, np.array(buf['A_fwd'][:, -1])))
    (J0, g0, s0), (J1, g1, s1) = res
    assert abs(J1 - J0) <= 1e-12 * abs(J0) and rel(g1, g0) < 1e-11 and rel(s1, s0) < 1e-12


def test_config2_against_committed_oracle_output():
    """BASELINE config 2: Npts=256, T=50, dt=0.1 (500 steps), seed-42 synthetic IC."""
    gold = np.load(os.path.join(GOLDEN, "oracle_sh23_c2.npz"))
    dom, X = sh23.Generate_IC(0.0725, Npts=256, seed=42)
    buf = sh23.GEN_BUFFER(dom, 500)
    args = [dom, 0.1, 500, 500, buf, None, "Discrete"]
    J = sh23.FWD_Solve_IVP_Lin([X], *args)
    g = sh23.ADJ_Solve_IVP_Lin([X], *args)[0]
    assert abs(J - gold["J"]) <= RTOL * abs(gold["J"])
    assert rel(g, gold["grad"]) < RTOL
    assert rel(buf['A_fwd'][:, -1], gold["stack_last"]) < 1e-9
    args[-1] = "Continuous"
    assert rel(sh23.ADJ_Solve_IVP_Lin([X], *args)[0], gold["grad_cont"]) < RTOL


def test_taylor_remainder_on_device_path():
    """The reference's own acceptance test (Adjoint_Gradient_Test) on the HIP path: second remainder ~ O(eps^2)."""
    dom, X = sh23.Generate_IC(0.0725, Npts=256, seed=42)
    _, dX = sh23.Generate_IC(0.0725, Npts=256, seed=7)
    buf = sh23.GEN_BUFFER(dom, 500)
    args_f = [dom, 0.1, 500, 500, buf, None, "Discrete"]
    AA = taylor_table([X], [dX], sh23.FWD_Solve_IVP_Lin, sh23.ADJ_Solve_IVP_Lin, sh23.Inner_Prod, args_f, (dom, None),
                      epsilon=1e-3)
    assert np.all(np.abs(AA[4, :4] - 2.0) < 5e-3), AA
    assert np.all(np.abs(AA[3, :4] - 1.0) < 5e-2), AA


def test_known_answer_linear_growth():
    """Infinitesimal amplitude: mode k is multiplied per SBDF1 step by (1/dt)/(1/dt + (1-k^2)^2 - a) — not taken from the oracle."""
    dom = sh23.SH23Domain(64)
    xs = dom.grid()
    L = dom.hypervolume
    eps, m, dt, n = 1e-9, 5, 0.1, 20
    X = eps * np.cos(2. * np.pi * m * xs / L)
    km = 2. * np.pi * m / L
    r = (1. / dt) / (1. / dt + (1. - km ** 2) ** 2 + 0.3)
    buf = sh23.GEN_BUFFER(dom, n)
    J = sh23.FWD_Solve_IVP_Lin([X], dom, dt, n, n, buf, None, "Discrete")
    expect = -dt * 0.5 * eps ** 2 * sum(r ** (2 * i) for i in range(n + 1))
    assert abs(J - expect) < 1e-7 * abs(expect)
    assert abs(abs(buf['A_fwd'][m, n]) - 0.5 * eps * r ** n) < 1e-7 * eps


@pytest.mark.parametrize("Npts", [64, 54])      # 54: no instantiation, the any-length kernels
def test_batched_problems_are_independent(Npts):
    dt, n, B = 0.1, 30, 5
    dom = sh23.SH23Domain(Npts)
    Xs = np.stack([sh23.Generate_IC(0.05 + 0.01 * b, Npts=Npts, seed=b)[1] for b in range(B)])
    ctx = dom.context(dt, n, batch=B)
    J = ctx.forward([Xs])
    g = ctx.adjoint(None)[0].reshape(B, -1)
    ip = ctx.inner(Xs, g)
    for b in range(B):
        o = _oracle(Npts, dt, n)
        Jo = o.forward([Xs[b]]); go = o.adjoint([Xs[b]])[0]
        assert abs(J[b] - Jo) <= RTOL * abs(Jo) and rel(g[b], go) < RTOL
        assert abs(ip[b] - o.inner(Xs[b], go)) <= RTOL * abs(o.inner(Xs[b], go))
        assert rel(ctx.snapshot(n, b).view(np.complex128), o.stack[:, n]) < 1e-9


def test_adjoint_requires_forward_and_errors_are_loud():
    ctx = _capi.Context(_capi.SMO_SH23, 64, (0., 12 * np.pi), 0.1, 5, -0.3)
    with pytest.raises(_capi.SmoError) as e:
        ctx.adjoint(None)
    assert e.value.code == 4
    with pytest.raises(ValueError):
        ctx.forward([np.zeros(7)])
    with pytest.raises(_capi.SmoError):
        _capi.Context(_capi.SMO_SH23, 3, (0., 1.), 0.1, 5, -0.3)          # fewer than 4 points


def test_optimiser_runs_on_device_callbacks(in_tmp_cwd):
    """Drop-in: the reference's driver call with the device-backed callbacks; J must decrease monotonically and the
    iterate sequence must equal the one obtained with the oracle callbacks (bit-exact iteration/evaluation counts)."""
    Npts, dt, n = 64, 0.1, 50
    dom, X = sh23.Generate_IC(0.0725, Npts=Npts, seed=42)
    buf = sh23.GEN_BUFFER(dom, n)
    args_f = [dom, dt, n, n, buf, None, "Discrete"]
    RES, FUN, Xopt = Optimise_On_Multi_Sphere([X.copy()], [0.0725], sh23.FWD_Solve_IVP_Lin, sh23.ADJ_Solve_IVP_Lin,
                                              sh23.Inner_Prod, args_f, (dom, None), max_iters=6, alpha_k=np.pi,
                                              LS='LS_wolfe', CG=True, verbose=False)
    o = _oracle(Npts, dt, n)
    RESo, FUNo, Xo = Optimise_On_Multi_Sphere([X.copy()], [0.0725], lambda X, *a: o.forward(X), lambda X, *a: o.adjoint(X),
                                              lambda x, y, *a: o.inner(x, y), (), (), max_iters=6, alpha_k=np.pi,
                                              LS='LS_wolfe', CG=True, verbose=False)
    assert len(FUN) == len(FUNo) == 6
    assert np.allclose(FUN, FUNo, rtol=1e-9) and np.allclose(RES, RESo, rtol=1e-6)
    assert rel(Xopt[0], Xo[0]) < 1e-8
    assert all(FUN[i + 1] >= FUN[i] for i in range(len(FUN) - 1))       # FUNCT stores -J_k = +J_phys: increasing


def test_reference_ic_recipe_with_device_prep():
    """Generate_IC(prep=True): noise -> filter -> normalise -> 101 device steps at dt=0.01 -> normalise (SH23:174-236)."""
    dom, X = sh23.Generate_IC(0.0725, Npts=128, prep=True)
    assert abs(np.mean(X * X) - 0.0725) < 1e-15
    _, X0 = sh23.Generate_IC(0.0725, Npts=128, prep=False)
    o = _oracle(128, 0.01, 101)
    o.forward([X0])
    ref = o.to_grid(o.stack[:, 101])
    ref *= np.sqrt(0.0725 / np.mean(ref * ref))
    assert rel(X, ref) < 1e-9


def test_c_program_through_the_c_abi(tmp_path):
    """tests/c/abi_smoke.c — plain C against include/smo.h — runs one gradient, the device-vector calls and the error path; its numbers
    must be the Python binding's, bit for bit (same library, same inputs)."""
    import subprocess
    from test_capi import _build_c_smoke
    exe = _build_c_smoke(tmp_path)
    N, n = 64, 40
    p = subprocess.run([exe, str(N), str(n)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    got = dict(ln.split() for ln in p.stdout.splitlines())
    i = np.arange(2 * N)
    x = 0.3 * np.sin(2.0 * np.pi * 3.0 * i / (2 * N)) + 0.1 * np.cos(2.0 * np.pi * 5.0 * i / (2 * N))
    ctx = _capi.Context(_capi.SMO_SH23, N, (0., 12. * np.pi), 0.1, n, -0.3)
    J = ctx.forward([x]); g = ctx.adjoint(None)[0]
    assert float(got["J"]) == J or abs(float(got["J"]) - J) <= 1e-14 * abs(J)          # libm sin/cos of the C program vs NumPy's
    assert abs(float(got["inner"]) - ctx.inner(x, g)) <= 1e-12 * abs(ctx.inner(x, g)) and got["inner"] == got["inner_dev"]
    assert abs(float(got["combo_norm2"]) - float(np.sum((2 * x - g) ** 2))) <= 1e-12 * float(np.sum((2 * x - g) ** 2))

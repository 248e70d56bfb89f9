"""Plane-Poiseuille optimal mixing on the device (through the C-ABI) against the oracle: the four transforms (dense MFMA GEMMs),
the tau-operator time stepping, both cost functionals and the exact discrete adjoint (1e-6 relative on J and grad J)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

from spheremanopt_amd import _capi, poiseuille as pz
from spheremanopt_amd.test_grad import taylor_table

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


def hermitian_full(o, c):
    """(a, Nz) coefficients of the non-negative wavenumbers -> the reference's full complex spectrum (Nxc, Nz) of a REAL field."""
    full = np.zeros((o.Nxc, o.Nz), dtype=complex)
    for i, n in enumerate(o.n):
        full[i] = c[n] if n >= 0 else np.conj(c[-n])
    full[0] = full[0].real
    return full


@pytest.mark.parametrize("Nx,Nz", [(24, 24), (48, 36), (30, 66)])
def test_transforms_match_oracle(Nx, Nz):
    from oracle.poiseuille import PoiseuilleOracle
    o = PoiseuilleOracle(Nx, Nz)
    dom = pz.PoiseuilleDomain(Nx, Nz)
    rs = np.random.RandomState(1)
    g = rs.standard_normal((Nx, Nz))
    c = rs.standard_normal((o.ax, Nz)) + 1j * rs.standard_normal((o.ax, Nz))
    cf = hermitian_full(o, c)
    assert rel(pz.transform(g, dom), o.transform(g)[:o.ax]) < 1e-12
    assert rel(pz.transformInverseAdjoint(g, dom), o.transformInverseAdjoint(g)[:o.ax]) < 1e-12
    assert rel(pz.transformInverse(c, dom), o.transformInverse(cf).real) < 1e-12
    assert rel(pz.transformAdjoint(c, dom), o.transformAdjoint(cf).real) < 1e-12
    assert np.abs(o.transformInverse(cf).imag).max() < 1e-12
    dom.drop_contexts()


@pytest.mark.parametrize("Nx,Nz,n,s", [(24, 24, 6, 0), (24, 24, 6, 1), (48, 36, 10, 0), (36, 48, 8, 1), (96, 48, 4, 0)])
def test_forward_adjoint_vs_oracle(Nx, Nz, n, s):
    from oracle.poiseuille import PoiseuilleOracle, synthetic_ic
    o = PoiseuilleOracle(Nx, Nz, dt=5e-3, N_ITERS=n, s=s, delta=0.3)
    X = synthetic_ic(o, 42)
    dom = pz.PoiseuilleDomain(Nx, Nz)
    buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
    args = [dom, 500., 0.05, n, buf, 5e-3, s, 1., 0.3]
    J = pz.FWD_Solve_Discrete([X], *args)
    g = pz.ADJ_Solve_Discrete([X], *args)
    Jo = o.forward([X]); go = o.adjoint([X])
    assert abs(J - Jo) <= RTOL * abs(Jo), (J, Jo)
    assert len(g) == 1 and g[0].shape == X.shape and rel(g[0], go[0]) < RTOL, rel(g[0], go[0])
    for f, key in enumerate(('u_fwd', 'w_fwd', 'b_fwd')):
        for i in (0, 1, -2, -1):
            assert rel(buf[key][:, :, i], o.stack[f, :o.ax, :, i]) < 1e-8, (key, i)
    ip = o.inner(X, go[0])
    assert abs(pz.Inner_Prod_Discrete(X, g[0], dom) - ip) <= RTOL * abs(ip)
    assert np.array_equal(pz.weightMatrixDisc(dom), o.W)
    dom.drop_contexts()


@pytest.mark.parametrize("sw", [0, 1])
def test_against_committed_oracle_output(sw):
    """96 x 48, 40 steps: expected values from the committed oracle run (tools/gen_golden_oracle.py)."""
    gold = np.load(os.path.join(GOLDEN, "oracle_poiseuille_96x48_s%d.npz" % sw))
    dom = pz.PoiseuilleDomain(96, 48)
    buf = pz.GEN_BUFFER(96, 48, dom, 40)
    args = [dom, 500., 0.05, 40, buf, 5e-3, sw, 1., 0.3]
    J = pz.FWD_Solve([gold["X"]], *args)
    g = pz.ADJ_Solve([gold["X"]], *args)[0]
    assert abs(J - gold["J"]) <= RTOL * abs(gold["J"])
    assert rel(g, gold["grad"]) < RTOL
    assert rel(buf['b_fwd'][:, :, -1], gold["b_last"]) < 1e-8
    if sw == 0:
        assert rel(buf['u_fwd'][:, :, -1], gold["u_last"]) < 1e-8
    dom.drop_contexts()


def test_taylor_ic_and_errors():
    dom, U0 = pz.Generate_IC(48, 36, E_0=0.02, seed=42)
    _, dU0 = pz.Generate_IC(48, 36, E_0=0.02, seed=7)
    assert abs(pz.Inner_Prod(U0[0], U0[0], dom) - 0.02) < 1e-14
    u = U0[0][:48 * 36].reshape(48, 36)
    c = pz.transform(u, dom)                                     # de-aliased like the reference's prepared IC (u['c'] *= DA, POIS:608)
    assert np.abs(c[dom.ada:]).max() < 1e-12 * np.abs(c).max() and np.abs(c[:, 24:]).max() < 1e-12 * np.abs(c).max()
    buf = pz.GEN_BUFFER(48, 36, dom, 10)
    args_f = [dom, 500., 0.05, 10, buf, 5e-3, 0, 1., 0.3]
    AA = taylor_table(U0, dU0, pz.FWD_Solve, pz.ADJ_Solve, pz.Inner_Prod, args_f, [dom, None], epsilon=1e-3)
    assert np.all(np.abs(AA[4, :4] - 2.0) < 2e-2), AA
    with pytest.raises(_capi.SmoError):
        dom.context(500., 0.05, 10, 5e-3, 0, 1., 0.3).adjoint(None, "Continuous")
    with pytest.raises(_capi.SmoError):
        pz.PoiseuilleDomain(50, 36).any_context()
    dom.drop_contexts()


# ---- "Continuous" formulation (Dedalus IVPs of FWD_Solve_Cnts / ADJ_Solve_Cnts) --------------------------------------------------------------

@pytest.mark.parametrize("Nx,Nz,n,s", [(16, 16, 6, 0), (16, 24, 10, 1), (32, 24, 12, 0), (64, 32, 5, 1)])
def test_continuous_forward_adjoint_vs_oracle(Nx, Nz, n, s):
    from oracle.poiseuille import PoiseuilleCntsOracle, synthetic_ic_cnts
    o = PoiseuilleCntsOracle(Nx, Nz, dt=5e-3, N_ITERS=n, s=s, delta=0.3)
    X = (10. if s == 1 else 1.) * synthetic_ic_cnts(o, 42)
    dom = pz.PoiseuilleDomain(Nx, Nz, continuous=True)
    buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
    args = [dom, 500., 0.05, n, buf, 5e-3, s, 1., 0.3]
    J = pz.FWD_Solve_Cnts([X], *args)
    g = pz.ADJ_Solve_Cnts([X], *args)
    Jo = o.forward([X]); go = o.adjoint([X])
    assert abs(J - Jo) <= RTOL * abs(Jo), (J, Jo)
    assert len(g) == 1 and g[0].shape == X.shape and rel(g[0], go[0]) < RTOL, rel(g[0], go[0])
    for f, key in enumerate(('u_fwd', 'w_fwd', 'b_fwd')):
        for i in (0, 1, -1):
            assert rel(buf[key][:, :, i], o.stack[f, :, :, i]) < 1e-8, (key, i)
    ip = o.inner(X, go[0])
    assert abs(pz.Inner_Prod_Cnts(X, g[0], dom) - ip) <= RTOL * abs(ip)
    with pytest.raises(_capi.SmoError):
        dom.context(*args[1:4], 5e-3, s, 1., 0.3).adjoint(None, "Discrete")
    with pytest.raises(ValueError):
        pz.FWD_Solve_Cnts([X], pz.PoiseuilleDomain(24, 24), *args[1:])
    dom.drop_contexts()


def test_reference_resolution_properties():
    """384 x 192 (the reference script's 3/2 * (256, 128)), 100 steps: size-independent checks — Taylor test of the exact discrete
    gradient (kinetic-energy cost), boundary conditions and incompressibility of the last snapshot, de-aliased snapshots."""
    Nx, Nz, n = 384, 192, 100
    dom, U0 = pz.Generate_IC(Nx, Nz, E_0=0.02, seed=42)
    _, dU0 = pz.Generate_IC(Nx, Nz, E_0=0.02, seed=7)
    buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
    args_f = [dom, 500., 0.05, n, buf, 5e-3, 0, 1., 0.125]
    AA = taylor_table(U0, dU0, pz.FWD_Solve, pz.ADJ_Solve, pz.Inner_Prod, args_f, [dom, None], epsilon=1e-3)
    assert np.all(np.abs(AA[4, :4] - 2.0) < 2e-2), AA
    u, w = buf['u_fwd'][:, :, -1], buf['w_fwd'][:, :, -1]
    sg = (-1.) ** np.arange(Nz)
    scale = np.abs(u).max()
    assert np.abs(u.sum(axis=1)).max() < 1e-10 * scale and np.abs((u * sg).sum(axis=1)).max() < 1e-10 * scale       # u(+-1) = 0
    assert np.abs(w.sum(axis=1)).max() < 1e-10 * scale and np.abs((w * sg).sum(axis=1)).max() < 1e-10 * scale       # w(+-1) = 0
    assert np.abs(u[dom.ada:]).max() == 0.0 and np.abs(buf['b_fwd'][1:, :, 0]).max() == 0.0                         # de-aliased in x; rho_0 = rho_0(z)
    # continuity of the tau solution: i k u + wz = 0 with wz = Dz w up to the tau term in the highest mode
    k = 2. * np.pi * np.arange(dom.a) / (4. * np.pi)
    D = np.zeros((Nz, Nz))
    for i in range(Nz):
        for j in range(i + 1, Nz):
            D[i, j] = 2. * j * ((j - i) % 2)
    D[0] /= 2.
    div = 1j * k[:, None] * u + w @ D.T
    assert np.abs(div[:, :Nz - 2]).max() < 1e-6 * np.abs(w @ D.T).max()
    dom.drop_contexts()


@pytest.mark.parametrize("Nx,Nz,n,s,split", [(48, 36, 6, 1, 2), (96, 48, 5, 0, 1), (30, 66, 5, 1, 3), (384, 192, 12, 1, 2), (384, 192, 4, 0, 0), (96, 384, 3, 1, 3)])
def test_hodlr_apply_matches_dense_apply(Nx, Nz, n, s, split, monkeypatch):
    """The tau operators in HODLR form (csrc/hodlr.hpp, the default) against the dense operator stream they replace (SMO_POIS_APPLY=dense):
    cost, gradient and snapshots to 1e-11, for several task splits and non-power-of-two trees; ranks stay the structural ones."""
    _, U0 = pz.Generate_IC(Nx, Nz, E_0=0.02, seed=3)
    res = {}
    for mode in ("dense", "hodlr"):
        monkeypatch.setenv("SMO_POIS_APPLY", mode)
        monkeypatch.setenv("SMO_POIS_HODLR_SPLIT", str(split))
        dom = pz.PoiseuilleDomain(Nx, Nz)
        buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
        args = [dom, 500., 0.05, n, buf, 5e-3, s, 1., 0.125]
        J = pz.FWD_Solve_Discrete(U0, *args)
        g = pz.ADJ_Solve_Discrete(U0, *args)[0]
        res[mode] = (J, g, buf['u_fwd'][:, :, -1].copy(), buf['b_fwd'][:, :, -1].copy())
        dom.drop_contexts()
    (Jd, gd, ud, bd), (Jh, gh, uh, bh) = res["dense"], res["hodlr"]
    assert abs(Jh - Jd) <= 1e-11 * abs(Jd), (Jh, Jd)
    assert rel(gh, gd) < 1e-11 and rel(uh, ud) < 1e-11 and rel(bh, bd) < 1e-11, (rel(gh, gd), rel(uh, ud), rel(bh, bd))


def test_apply_mode_knob_is_validated(monkeypatch):
    monkeypatch.setenv("SMO_POIS_APPLY", "banded")
    dom = pz.PoiseuilleDomain(24, 24)
    with pytest.raises(Exception):
        dom.context(500., 0.05, 2, 5e-3, 0, 1., 0.125)
    dom.drop_contexts()


@pytest.mark.parametrize("Nx,Nz,n,s", [(32, 24, 6, 0), (64, 32, 5, 1), (128, 96, 6, 1)])
def test_continuous_hodlr_apply_matches_dense_apply(Nx, Nz, n, s, monkeypatch):
    """Continuous formulation: both IVP operators (forward and adjoint) in HODLR form against the dense stream."""
    from oracle.poiseuille import PoiseuilleCntsOracle, synthetic_ic_cnts
    X = synthetic_ic_cnts(PoiseuilleCntsOracle(Nx, Nz, dt=5e-3, N_ITERS=1, s=s, delta=0.3), 5)
    res = {}
    for mode in ("dense", "hodlr"):
        monkeypatch.setenv("SMO_POIS_APPLY", mode)
        dom = pz.PoiseuilleDomain(Nx, Nz, continuous=True)
        buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
        args = [dom, 500., 0.05, n, buf, 5e-3, s, 1., 0.3]
        J = pz.FWD_Solve_Cnts([X], *args)
        g = pz.ADJ_Solve_Cnts([X], *args)[0]
        res[mode] = (J, g, buf['b_fwd'][:, :, -1].copy())
        dom.drop_contexts()
    (Jd, gd, bd), (Jh, gh, bh) = res["dense"], res["hodlr"]
    assert abs(Jh - Jd) <= 1e-11 * abs(Jd), (Jh, Jd)
    assert rel(gh, gd) < 1e-11 and rel(bh, bd) < 1e-11, (rel(gh, gd), rel(bh, bd))


@pytest.mark.parametrize("sw", [1, 0])
def test_reference_resolution_full_length_fixture(sw):
    """The workload of bench.py's Poiseuille line — 384 x 192, 1000 steps (T = 5), mix-norm cost (sw = 1; sw = 0: the kinetic-energy cost with
    its forcing terms), the same seeded input — against the oracle runs committed as tests/golden/oracle_poiseuille_384x192_n1000_s<sw>.npz
    (tools/gen_golden_poiseuille_full.py): the HODLR operators, the fused epilogues and the in-place snapshot reads at the size and length
    that is timed."""
    path = os.path.join(GOLDEN, "oracle_poiseuille_384x192_n1000_s%d.npz" % sw)
    if not os.path.exists(path):
        pytest.skip("%s not generated (tools/gen_golden_poiseuille_full.py)" % os.path.basename(path))
    gold = np.load(path)
    Nx, Nz, n, s = int(gold["Nx"]), int(gold["Nz"]), int(gold["steps"]), int(gold["s"])
    X = float(gold["amplitude"]) * np.random.RandomState(int(gold["seed"])).standard_normal(2 * Nx * Nz)
    dom = pz.PoiseuilleDomain(Nx, Nz)
    buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
    args = [dom, 500., 0.05, n, buf, 5e-3, s, 1., 0.125]
    J = pz.FWD_Solve_Discrete([X], *args)
    g = pz.ADJ_Solve_Discrete([X], *args)[0]
    assert abs(J - float(gold["J"])) <= RTOL * abs(float(gold["J"])), (J, float(gold["J"]))
    nrm = float(gold["grad_norm"])
    assert abs(np.linalg.norm(g) - nrm) <= RTOL * nrm
    assert np.linalg.norm(g[gold["idx"]] - gold["grad"]) <= RTOL * np.linalg.norm(gold["grad"])
    w = np.random.RandomState(77).standard_normal(g.size)
    assert abs(float(np.dot(g, w)) - float(gold["grad_proj"])) <= RTOL * nrm * np.sqrt(g.size)
    for key, arr in (("u_last", buf['u_fwd'][:, :, -1]), ("b_last", buf['b_fwd'][:, :, -1]), ("b_prev", buf['b_fwd'][:, :, -2])):
        ref = gold[key + "_sample"]
        assert np.linalg.norm(arr.ravel()[::97] - ref) <= 1e-8 * np.linalg.norm(ref), key
        assert abs(np.linalg.norm(arr) - float(gold[key + "_norm"])) <= 1e-8 * float(gold[key + "_norm"]), key
    dom.drop_contexts()


@pytest.mark.parametrize("Nx,Nz,n,s", [(24, 24, 5, 0), (30, 66, 5, 1), (36, 48, 4, 0), (96, 48, 4, 1), (384, 192, 6, 1), (384, 192, 3, 0), (60, 36, 3, 1)])
def test_x_transforms_as_ffts_match_the_dense_products(Nx, Nz, n, s, monkeypatch):
    """The x phases as LDS FFTs (pois_x_to_grid / pois_x_to_coeff, the default where Nx has an instantiation) against the dense products with
    the x matrices (SMO_POIS_XFFT=0): cost, gradient and snapshots to 1e-11, including a ragged last tile of z columns (Nz = 66)."""
    _, U0 = pz.Generate_IC(Nx, Nz, E_0=0.02, seed=11)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SMO_POIS_XFFT", mode)
        dom = pz.PoiseuilleDomain(Nx, Nz)
        buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
        args = [dom, 500., 0.05, n, buf, 5e-3, s, 1., 0.125]
        J = pz.FWD_Solve_Discrete(U0, *args)
        g = pz.ADJ_Solve_Discrete(U0, *args)[0]
        res[mode] = (J, g, buf['u_fwd'][:, :, -1].copy(), buf['b_fwd'][:, :, -2].copy())
        dom.drop_contexts()
    (Jd, gd, ud, bd), (Jf, gf, uf, bf) = res["0"], res["1"]
    assert abs(Jf - Jd) <= 1e-11 * abs(Jd), (Jf, Jd)
    assert rel(gf, gd) < 1e-11 and rel(uf, ud) < 1e-11 and rel(bf, bd) < 1e-11, (rel(gf, gd), rel(uf, ud), rel(bf, bd))


@pytest.mark.parametrize("Nx,Nz,n,s", [(24, 24, 6, 0), (48, 66, 7, 1), (96, 48, 6, 0), (192, 96, 4, 1), (384, 192, 3, 0), (768, 48, 2, 1)])
def test_products_folded_into_the_x_transform_match_the_separate_kernels(Nx, Nz, n, s, monkeypatch):
    """Round 4: the grid stage of a step with the pointwise kernel folded into the loads of the forward x transform (pois_x_prod_to_coeff,
    SMO_POIS_XPROD=1) against the separate kernels (the default): both cost functionals (s = 0 carries the two forcing products: 10 instead
    of 8 product fields), a ragged last tile of z columns (Nz = 66).  Same arithmetic per element: 1e-12."""
    _, U0 = pz.Generate_IC(Nx, Nz, E_0=0.02, seed=11)
    res = {}
    for mode in ("0", "prod"):
        monkeypatch.setenv("SMO_POIS_XPROD", "1" if mode == "prod" else "0")
        dom = pz.PoiseuilleDomain(Nx, Nz)
        buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
        args = [dom, 500., 0.05, n, buf, 5e-3, s, 1., 0.125]
        J = pz.FWD_Solve_Discrete(U0, *args)
        g = pz.ADJ_Solve_Discrete(U0, *args)[0]
        J2 = pz.FWD_Solve_Discrete(U0, *args)                     # a second solve on the same context: the partial-sum rows are reset
        res[mode] = (J, g, buf['u_fwd'][:, :, -1].copy(), buf['b_fwd'][:, :, -2].copy())
        assert J2 == J
        dom.drop_contexts()
    J3, g3, u3, b3 = res["0"]
    for mode in ("prod",):
        Jf, gf, uf, bf = res[mode]
        assert abs(Jf - J3) <= 1e-12 * abs(J3), (mode, Jf, J3)
        assert rel(gf, g3) < 1e-12 and rel(uf, u3) < 1e-12 and rel(bf, b3) < 1e-12, (mode, rel(gf, g3), rel(uf, u3), rel(bf, b3))


@pytest.mark.parametrize("Nx,Nz,n,s", [(16, 16, 6, 0), (32, 24, 8, 1), (64, 32, 5, 1), (128, 64, 4, 0)])
def test_continuous_x_transforms_as_ffts_match_the_dense_products(Nx, Nz, n, s, monkeypatch):
    """Continuous formulation (Nx modes on the 3 Nx / 2 grid: a zero-padded transform): x phases as FFTs against the dense products."""
    from oracle.poiseuille import PoiseuilleCntsOracle, synthetic_ic_cnts
    X = synthetic_ic_cnts(PoiseuilleCntsOracle(Nx, Nz, dt=5e-3, N_ITERS=1, s=s, delta=0.3), 9)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SMO_POIS_XFFT", mode)
        dom = pz.PoiseuilleDomain(Nx, Nz, continuous=True)
        buf = pz.GEN_BUFFER(Nx, Nz, dom, n)
        args = [dom, 500., 0.05, n, buf, 5e-3, s, 1., 0.3]
        J = pz.FWD_Solve_Cnts([X], *args)
        g = pz.ADJ_Solve_Cnts([X], *args)[0]
        res[mode] = (J, g, buf['b_fwd'][:, :, -1].copy())
        dom.drop_contexts()
    (Jd, gd, bd), (Jf, gf, bf) = res["0"], res["1"]
    assert abs(Jf - Jd) <= 1e-11 * abs(Jd), (Jf, Jd)
    assert rel(gf, gd) < 1e-11 and rel(bf, bd) < 1e-11, (rel(gf, gd), rel(bf, bd))

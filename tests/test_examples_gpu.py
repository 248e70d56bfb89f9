"""The reference's example scripts (its `__main__` blocks) on the device path: same call sequence, reduced lengths."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_kdyn_script_at_the_reference_default_resolution(in_tmp_cwd):
    """Npts = 24 (G = 36 = 4*3*3, the reference script's default): Taylor test + a few optimiser iterations; J must grow."""
    from spheremanopt_amd.examples import kdyn_optimise
    R, F, X, AA = kdyn_optimise.main(["--npts", "24", "--dt", "5e-4", "--steps", "60", "--max-iters", "3", "--test-gradient", "--quiet"])
    assert np.all(np.abs(AA[4, :4] - 2.0) < 2e-2), AA
    assert len(F) == 3 and F[-1] > F[0] > 0
    assert len(X) == 2 and X[0].shape == (3 * 36 ** 3,)


def test_kdyn_script_in_one_process_over_several_devices(in_tmp_cwd):
    """The same script with --devices 0,0 (the multi-device context, the box's GPU listed twice) and the optimiser's vectors distributed over
    the devices: the same J_k sequence as the one-GPU run."""
    from spheremanopt_amd.examples import kdyn_optimise
    base = ["--npts", "24", "--dt", "5e-4", "--steps", "40", "--max-iters", "2", "--quiet"]
    _, F1, X1, _ = kdyn_optimise.main(base)
    _, F2, X2, _ = kdyn_optimise.main(base + ["--devices", "0,0", "--device-vectors"])
    assert len(F1) == len(F2) and np.allclose(F1, F2, rtol=1e-10, atol=0)
    x2 = X2[0].numpy()
    assert np.linalg.norm(x2 - X1[0]) <= 1e-9 * np.linalg.norm(X1[0])


def test_kdyn_npts24_matches_oracle():
    from oracle.kdyn import KDynOracle
    from spheremanopt_amd import kdyn
    for N in (24, 48):
        dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
        n = 3
        buf = kdyn.GEN_BUFFER(N, dom, n)
        args = [dom, 1., 1e-3, n, n, buf, "Final", "Discrete"]
        J = kdyn.FWD_Solve_IVP_Lin([B, U], *args)
        g = kdyn.ADJ_Solve_IVP_Lin([B, U], *args)
        o = KDynOracle(N, Rm=1., dt=1e-3, N_ITERS=n)
        Jo = o.forward([B, U]); go = o.adjoint([B, U])
        assert abs(J - Jo) <= 1e-6 * abs(Jo)
        for a, b in zip(g, go):
            assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b)
        dom.drop_contexts()


def test_sh23_and_shb23_scripts(in_tmp_cwd):
    from spheremanopt_amd.examples import sh23_optimise, shb23_optimise
    R, F, X, AA = sh23_optimise.main(["--T", "10", "--max-iters", "4", "--test-gradient", "--quiet"])
    assert np.all(np.abs(AA[4, :4] - 2.0) < 1e-2), AA
    assert len(F) == 4 and all(F[i + 1] >= F[i] for i in range(3))
    R, F, X, AA = shb23_optimise.main(["--npts", "256", "--T", "2", "--max-iters", "3", "--test-gradient", "--quiet"])
    assert np.all(np.abs(AA[4, :4] - 2.0) < 1e-2), AA
    assert len(F) >= 1 and all(F[i + 1] >= F[i] for i in range(len(F) - 1))


def test_poiseuille_script(in_tmp_cwd):
    """FWD_Solve_Poiseuille.py's __main__ at a reduced resolution: Taylor test with the kinetic-energy cost, then a few CG/Wolfe iterations with
    the mix-norm cost (the reference's default, s = 1), Discrete formulation; then the script's default "Continuous" formulation.  The cost must
    never increase.  Line-search warnings are attributed run by run: the Discrete formulation has the exact gradient of its cost (Taylor
    exponent 2) and must not produce any; the Continuous one differentiates the continuous problem — its gradient is only O(dt)-consistent
    with the discrete cost (as in the reference, FWD_Solve_Poiseuille.py:1161-1318), so a Wolfe search near the optimum may give up: that is
    the one run allowed to warn, and only with the optimiser's own LineSearchWarning."""
    import warnings
    from spheremanopt_amd.examples import poiseuille_optimise
    from spheremanopt_amd.sphere_opt import LineSearchWarning

    def run(args):
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            out = poiseuille_optimise.main(args)
        return out, rec

    (R, F, X, AA), w_taylor = run(["--nx", "32", "--nz", "24", "--T", "0.1", "--s", "0", "--max-iters", "3", "--test-gradient", "--quiet"])
    assert np.all(np.abs(AA[4, :4] - 2.0) < 2e-2), AA
    assert len(F) >= 1 and all(F[i + 1] >= F[i] for i in range(len(F) - 1))
    (R, F, X, _), w_mix = run(["--nx", "32", "--nz", "24", "--T", "0.25", "--s", "1", "--max-iters", "3", "--quiet"])
    assert len(F) >= 1 and all(F[i + 1] >= F[i] for i in range(len(F) - 1))
    assert len(X) == 1 and X[0].shape == (2 * 48 * 36,)
    # the script's default formulation ("Continuous"): same grid, 32 x 24 modes; the cost must not increase either
    (R, F, X, _), w_cnts = run(["--nx", "32", "--nz", "24", "--T", "0.25", "--s", "0", "--max-iters", "3", "--continuous", "--quiet"])
    assert len(F) >= 1 and all(F[i + 1] >= F[i] for i in range(len(F) - 1)) and X[0].shape == (2 * 48 * 36,)
    names = {"Discrete s=0 (+Taylor test)": w_taylor, "Discrete s=1": w_mix, "Continuous s=0": w_cnts}
    ls = {k: [str(w.message) for w in v if issubclass(w.category, LineSearchWarning)] for k, v in names.items()}
    # Which run warns, and why (measured, round 3): the kinetic-energy runs — Discrete AND Continuous — are silent.  The mix-norm run (s = 1)
    # warns twice, "could not find a solution less than or equal to amax: 100.0" + "did not converge", and the optimiser then stops with
    # "Couldn't find a descent direction" exactly like the reference's (Sphere_Grad_Descent.py:436, 477, 791-793): over T = 0.25 the mix-norm
    # hardly moves, phi keeps decreasing all the way to the step cap alpha_max = alpha_k = 100 the script passes (FWD_Solve_Poiseuille.py:1777)
    # and scalar_search_wolfe2 gives up at the cap.  It is the reference's line search meeting its own cap, not a gradient defect: the mix-norm
    # gradient of exactly this configuration is checked against a central difference below.
    assert not ls["Discrete s=0 (+Taylor test)"] and not ls["Continuous s=0"], ls
    assert all("amax" in m or "did not converge" in m for m in ls["Discrete s=1"]), ls
    from spheremanopt_amd import poiseuille as pz
    dom, U0 = pz.Generate_IC(48, 36, E_0=0.02, seed=42)
    _, dU0 = pz.Generate_IC(48, 36, E_0=0.02, seed=7)
    buf = pz.GEN_BUFFER(48, 36, dom, 50)
    args_f = [dom, 500., 0.05, 50, buf, 5e-3, 1, 1., 0.125]
    J0 = pz.FWD_Solve(U0, *args_f); g = pz.ADJ_Solve(U0, *args_f)
    d = pz.Inner_Prod(g[0], dU0[0], dom)
    h = 1e-2
    fd = (pz.FWD_Solve([U0[0] + h * dU0[0]], *args_f) - pz.FWD_Solve([U0[0] - h * dU0[0]], *args_f)) / (2 * h)
    assert abs(d - fd) <= 2e-3 * abs(fd), (d, fd, J0)
    dom.drop_contexts()


def test_on_disk_products(in_tmp_cwd):
    """write_products=True: scalar_data/ and CheckPoints/ with the groups the reference's plot scripts read; values against the snapshots."""
    import os
    from spheremanopt_amd import kdyn, products, sh23
    dom, X = sh23.Generate_IC(0.0725, Npts=64)
    dom.write_products = True
    buf = sh23.GEN_BUFFER(dom, 40)
    J = sh23.FWD_Solve_IVP_Lin([X], dom, 0.1, 40, 40, buf, None, "Discrete")
    sd = products.read_products([f for f in (os.path.join("scalar_data", "scalar_data_s1.h5"), os.path.join("scalar_data", "scalar_data_s1.npz")) if os.path.exists(f)][0])
    assert sd["tasks/Kinetic energy"].shape == (3, 1) and np.allclose(sd["scales/sim_time"], [0., 2., 4.])
    assert abs(sd["tasks/Kinetic energy"][0, 0] - 0.0725) < 1e-12                     # <X,X> of the initial condition
    cp = products.read_products([f for f in (os.path.join("CheckPoints", "CheckPoints_s1.h5"), os.path.join("CheckPoints", "CheckPoints_s1.npz")) if os.path.exists(f)][0])
    assert cp["tasks/u"].shape == (2, 96) and cp["tasks/u_hat"].shape == (2, 32)
    assert abs(np.mean(cp["tasks/u"][0] ** 2) - 0.0725) < 1e-12
    sh23.File_Manips(0)
    # KDyn, "Final" cost: the last sample of the magnetic energy is -J
    domk, B, U = kdyn.Generate_IC(16, U_Noise=True)
    domk.write_products = True
    bufk = kdyn.GEN_BUFFER(16, domk, 20)
    Jk = kdyn.FWD_Solve_IVP_Lin([B, U], domk, 1., 1e-2, 20, 20, bufk, "Final", "Discrete")
    sd = products.read_products([f for f in (os.path.join("scalar_data", "scalar_data_s1.h5"), os.path.join("scalar_data", "scalar_data_s1.npz")) if os.path.exists(f)][0])
    me = sd["tasks/Magnetic energy"]
    assert me.shape == (2, 1, 1, 1) and abs(me[0, 0, 0, 0] - 1.0) < 1e-12 and abs(me[1, 0, 0, 0] + Jk) < 1e-12 * abs(Jk)
    cp = products.read_products([f for f in (os.path.join("CheckPoints", "CheckPoints_s1.h5"), os.path.join("CheckPoints", "CheckPoints_s1.npz")) if os.path.exists(f)][0])
    assert cp["tasks/A"].shape == (2, 24, 24, 24) and np.allclose(cp["tasks/A"][0].ravel(), B[:24 ** 3], atol=1e-12)
    kdyn.File_Manips(1)
    assert any(f.startswith("CheckPoints_iter_1") for f in os.listdir("."))


def test_on_disk_products_of_the_hand_stepped_solvers(in_tmp_cwd):
    """SHB23 / Poiseuille (Discrete) write scalar_data_s1 and CheckPoints_s1 into the working directory, every step, like the reference."""
    import glob
    from spheremanopt_amd import poiseuille as pz, products, shb23
    dom, X = shb23.Generate_IC(64, M_0=0.0019)
    dom.write_products = True
    buf = shb23.GEN_BUFFER(64, dom, 20)
    J = shb23.FWD_Solve([X], dom, buf, 20)
    sd = products.read_products(glob.glob("scalar_data_s1.*")[0])
    ke = sd["tasks/Kinetic energy"]
    assert ke.shape == (20,) and abs(ke[0] - 0.0019) < 1e-14 and np.allclose(sd["scales/sim_time"], 1e-2 * np.arange(20))
    last = shb23.Inner_Prod(buf['A_fwd'][:, 20], buf['A_fwd'][:, 20], dom)
    assert abs(-J - 1e-2 * (ke.sum() + last)) < 1e-12 * abs(J)                        # J = dt * sum_{n=0}^{N} <u_n,u_n>
    cp = products.read_products(glob.glob("CheckPoints_s1.*")[0])
    assert cp["tasks/u"].shape == (2, 64) and np.allclose(cp["tasks/u"][0], X, atol=1e-13)
    shb23.File_Manips(2)
    assert glob.glob("scalar_data_iter_2.*") and glob.glob("CheckPoints_iter_2.*")
    for f in glob.glob("*_s1.*"):
        os.remove(f)
    domp, U0 = pz.Generate_IC(48, 36, E_0=0.02)
    domp.write_products = True
    bufp = pz.GEN_BUFFER(48, 36, domp, 12)
    Jp = pz.FWD_Solve(U0, domp, 500., 0.05, 12, bufp, 5e-3, 0, 1., 0.3)
    sd = products.read_products(glob.glob("scalar_data_s1.*")[0])
    ke = sd["tasks/Kinetic  energy"]
    assert ke.shape == (13,) and abs(ke[0] - 0.02) < 1e-10 and abs(Jp + 0.5 * 5e-3 * ke.sum()) < 1e-10 * abs(Jp)
    cp = products.read_products(glob.glob("CheckPoints_s1.*")[0])
    assert cp["tasks/vorticity"].shape == (2, 48, 36) and np.allclose(cp["tasks/u"][0].ravel(), U0[0][:48 * 36], atol=1e-10)

"""The reference's example scripts (its `__main__` blocks) on the device path: same call sequence, reduced lengths."""
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
    """FWD_Solve_Poiseuille.py's __main__ (Discrete formulation) at a reduced resolution: Taylor test with the kinetic-energy cost, then
    a few CG/Wolfe iterations with the mix-norm cost (the reference's default, s = 1); the cost must not increase."""
    from spheremanopt_amd.examples import poiseuille_optimise
    R, F, X, AA = poiseuille_optimise.main(["--nx", "32", "--nz", "24", "--T", "0.1", "--s", "0", "--max-iters", "3", "--test-gradient", "--quiet"])
    assert np.all(np.abs(AA[4, :4] - 2.0) < 2e-2), AA
    assert len(F) >= 1 and all(F[i + 1] >= F[i] for i in range(len(F) - 1))
    R, F, X, _ = poiseuille_optimise.main(["--nx", "32", "--nz", "24", "--T", "0.25", "--s", "1", "--max-iters", "3", "--quiet"])
    assert len(F) >= 1 and all(F[i + 1] >= F[i] for i in range(len(F) - 1))
    assert len(X) == 1 and X[0].shape == (2 * 48 * 36,)
    # the script's default formulation ("Continuous"): same grid, 32 x 24 modes; the cost must not increase either
    R, F, X, _ = poiseuille_optimise.main(["--nx", "32", "--nz", "24", "--T", "0.25", "--s", "0", "--max-iters", "3", "--continuous", "--quiet"])
    assert len(F) >= 1 and all(F[i + 1] >= F[i] for i in range(len(F) - 1)) and X[0].shape == (2 * 48 * 36,)

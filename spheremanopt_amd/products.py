"""On-disk products of the forward solves: what Dedalus' file handlers write in the reference and what its
plot scripts read — `scalar_data/scalar_data_s1.h5` (energy time series every 20 iterations) and `CheckPoints/CheckPoints_s1.h5`
(fields at t = 0 and t = T) with the groups `tasks/...` and `scales/...` (FWD_Solve_SH23.py:478-483, FWD_Solve_KDyn.py:603-611), plus
`File_Manips(k)` (FWD_Solve_SH23.py:731-746, FWD_Solve_KDyn.py:1006-1021), the optimiser callback that keeps one copy per iteration.

HDF5 when h5py is importable, otherwise `.npz` archives whose keys are the HDF5 paths ('tasks/Kinetic energy', 'scales/sim_time', ...):
`read_products(path)` returns the same dict either way.  Opt-in (`domain.write_products = True`): the solves write nothing by default.
Everything here is host-side post-processing of the device snapshot stack; none of it is on the hot path."""
import glob
import os
import shutil

import numpy as np

try:                                              # optional dependency, absent in this image
    import h5py
except ImportError:                               # pragma: no cover
    h5py = None

CADENCE = 20                                      # analysis1 = add_file_handler("scalar_data", iter=20)


def write_products(path_base, groups):
    """groups: {'tasks/name': array, 'scales/sim_time': array, ...} -> path_base + ('.h5' | '.npz'); returns the file name."""
    os.makedirs(os.path.dirname(path_base) or ".", exist_ok=True)
    if h5py is not None:
        with h5py.File(path_base + ".h5", "w") as f:
            for k, v in groups.items():
                f[k] = v
        return path_base + ".h5"
    np.savez(path_base + ".npz", **groups)
    return path_base + ".npz"


def read_products(path):
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    out = {}
    with h5py.File(path, "r") as f:
        f.visititems(lambda name, obj: out.__setitem__(name, obj[()]) if hasattr(obj, "shape") else None)
    return out


def File_Manips(k):
    """Keep the products of optimiser iteration k: scalar_data_iter_k / CheckPoints_iter_k next to DAL_PROGRESS (reference callback)."""
    for stem in ("scalar_data", "CheckPoints"):
        for src in glob.glob(os.path.join(stem, stem + "_s1.*")):
            shutil.copyfile(src, "%s_iter_%i%s" % (stem, k, os.path.splitext(src)[1]))


def sample_iterations(N_ITERS):
    return np.arange(0, N_ITERS + 1, CADENCE)


# ---- SH23 ------------------------------------------------------------------------------------------------------------------------
def write_sh23(domain, ctx, dt, N_ITERS):
    """'Kinetic energy' = (1/L) integ u^2 = |c_0|^2 + 2 sum_{k>0} |c_k|^2 of the amplitude-normalised coefficients; CheckPoints: u on the
    3/2 grid and u_hat at t = 0 and t = T."""
    its = sample_iterations(N_ITERS)
    ke = np.empty((len(its), 1))
    for i, n in enumerate(its):
        c = ctx.snapshot(int(n)).view(np.complex128)
        ke[i, 0] = abs(c[0]) ** 2 + 2. * np.sum(np.abs(c[1:]) ** 2)
    f1 = write_products(os.path.join("scalar_data", "scalar_data_s1"), {"tasks/Kinetic energy": ke, "scales/sim_time": its * dt,
                                                                        "scales/iteration": its})
    G15 = 3 * domain.Npts // 2
    u, uh = np.empty((2, G15)), np.empty((2, domain.Nc), dtype=complex)
    for i, n in enumerate((0, N_ITERS)):
        c = ctx.snapshot(n).view(np.complex128)
        pad = np.zeros(G15 // 2 + 1, dtype=complex)
        pad[:domain.Nc] = c
        pad[0] = pad[0].real
        u[i], uh[i] = np.fft.irfft(pad, n=G15) * G15, c
    L = domain.interval[1] - domain.interval[0]
    f2 = write_products(os.path.join("CheckPoints", "CheckPoints_s1"),
                        {"tasks/u": u, "tasks/u_hat": uh, "scales/x/1.5": domain.interval[0] + L * np.arange(G15) / G15,
                         "scales/kx": 2. * np.pi * np.arange(domain.Nc) / L, "scales/sim_time": np.array([0., N_ITERS * dt])})
    return f1, f2


# ---- KDyn ------------------------------------------------------------------------------------------------------------------------
def write_kdyn(domain, ctx, X0, dt, N_ITERS, coeff_to_grid):
    """'Magnetic energy' = <B,B> (sum over components of the grid mean) every 20 iterations, shape (nt,1,1,1) like Dedalus writes a
    volume integral on a 3-D domain; CheckPoints: A, B, C (the components of B) and the velocity on the 3/2 grid at t = 0 and t = T."""
    its = sample_iterations(N_ITERS)
    a, m = domain.a, domain.m
    w = np.full(a, 2.); w[0] = 1.                                # Hermitian half spectrum: kx = 0 counts once
    me = np.empty((len(its), 1, 1, 1))
    for i, n in enumerate(its):
        c = ctx.snapshot(int(n)).view(np.complex128).reshape(3, a, m, m)
        me[i, 0, 0, 0] = np.sum(w[None, :, None, None] * np.abs(c) ** 2)
    f1 = write_products(os.path.join("scalar_data", "scalar_data_s1"), {"tasks/Magnetic energy": me, "scales/sim_time": its * dt,
                                                                        "scales/iteration": its})
    G = domain.G
    fields = np.empty((3, 2, G, G, G))
    for i, n in enumerate((0, N_ITERS)):
        c = ctx.snapshot(n).view(np.complex128).reshape(3, a, m, m)
        for comp in range(3):
            fields[comp, i] = coeff_to_grid(domain, c[comp])
    s = domain.interval[0] + (domain.interval[1] - domain.interval[0]) * np.arange(G) / G
    U = np.asarray(X0[1]).reshape(3, G, G, G)
    groups = {"tasks/A": fields[0], "tasks/B": fields[1], "tasks/C": fields[2], "scales/sim_time": np.array([0., N_ITERS * dt]),
              "scales/x/1.5": s, "scales/y/1.5": s, "scales/z/1.5": s}
    for name, comp in (("u-velocity", 0), ("v-velocity", 1), ("w-velocity", 2)):
        groups["tasks/" + name] = np.stack([U[comp], U[comp]])
    f2 = write_products(os.path.join("CheckPoints", "CheckPoints_s1"), groups)
    return f1, f2

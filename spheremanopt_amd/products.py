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
    """Keep the products of optimiser iteration k: scalar_data_iter_k / CheckPoints_iter_k next to DAL_PROGRESS (reference callback).
    The Dedalus file handlers write into sub-directories, the hand-stepped Discrete solvers into the working directory
    (FWD_Solve_SHB23.py:931-946 handles both the same way)."""
    for stem in ("scalar_data", "CheckPoints"):
        for src in glob.glob(os.path.join(stem, stem + "_s1.*")) + glob.glob(stem + "_s1.*"):
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


# ---- SHB23 (Discrete): files in the working directory, every step (FWD_Solve_SHB23.py:604-672) ---------------------------------------
def write_shb23(domain, ctx, dt, N_ITERS, W):
    """'Kinetic energy' = <u_i,u_i>_W for i = 0..N-1; CheckPoints: u on the Gauss grid at i = 0 and i = N-1."""
    Lz = domain.interval[1] - domain.interval[0]
    ke = np.array([float(np.dot(u, W * u) / Lz) for u in (ctx.snapshot(i) for i in range(N_ITERS))])
    f1 = write_products("scalar_data_s1", {"tasks/Kinetic energy": ke, "scales/sim_time": dt * np.arange(N_ITERS)})
    f2 = write_products("CheckPoints_s1", {"tasks/u": np.stack([ctx.snapshot(0), ctx.snapshot(N_ITERS - 1)]), "scales/z/1.5": domain.grid()})
    return f1, f2


# ---- Poiseuille (Discrete): files in the working directory (FWD_Solve_Poiseuille.py:945-1151) ----------------------------------------
def write_poiseuille(domain, ctx, dt, N_ITERS, s):
    """'Kinetic  energy' (two spaces, as in the reference) and 'Buoyancy energy' with the discrete inner product, for i = 0..N-1 and,
    when s = 0, the final state; CheckPoints: vorticity, b, u, w on the (Nx, Nz) grid at i = 0 and i = N-1."""
    Nx, Nz, a = domain.Nx, domain.Nz, domain.a
    Lx = domain.interval[1] - domain.interval[0]
    z = domain.grid(1)
    dz = np.empty(Nz); dz[0] = z[1] - z[0]; dz[1:] = z[1:] - z[:-1]
    W, V = dz * (Lx / Nx), domain.hypervolume
    j, i = np.arange(Nz)[None, :], np.arange(Nz)[:, None]
    Ti = np.cos(np.pi * j * (2 * i + 1) / (2. * Nz)) * (-1.) ** j                   # T coefficients -> Gauss grid (POIS:67-76)
    D = np.zeros((Nz, Nz))
    for r in range(Nz):
        for c in range(r + 1, Nz):
            D[r, c] = 2. * c * ((c - r) % 2)
    D[0] /= 2.
    k = 2. * np.pi * np.arange(a) / Lx

    def grid(c):                                                                     # (a, Nz) coefficients -> (Nx, Nz) grid
        F = np.zeros((Nx // 2 + 1, Nz), dtype=complex)
        F[:a] = c @ Ti.T
        F[0] = F[0].real
        return np.fft.irfft(F, n=Nx, axis=0) * Nx

    last = N_ITERS + 1 if s == 0 else N_ITERS                                       # with s = 1 the last slot holds the mix-norm fields
    ke, de, saved = np.empty(last), np.empty(last), {}
    for n in range(last):
        c = ctx.snapshot(n).view(np.complex128).reshape(3, a, Nz)
        u, w, b = grid(c[0]), grid(c[1]), grid(c[2])
        ke[n] = np.sum(W * (u * u + w * w)) / V
        de[n] = np.sum(W * b * b) / V
        if n in (0, N_ITERS - 1):
            saved[n] = (grid(1j * k[:, None] * c[1]) - grid(c[0] @ D.T), b, u, w)
    f1 = write_products("scalar_data_s1", {"tasks/Kinetic  energy": ke, "tasks/Buoyancy energy": de, "scales/sim_time": dt * np.arange(last)})
    keys = ("vorticity", "b", "u", "w")
    groups = {"tasks/" + name: np.stack([saved[0][q], saved[N_ITERS - 1][q]]) for q, name in enumerate(keys)}
    groups.update({"scales/x/1.5": domain.grid(0), "scales/z/1.5": z})
    f2 = write_products("CheckPoints_s1", groups)
    return f1, f2

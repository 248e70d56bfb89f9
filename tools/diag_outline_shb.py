"""Diagnose the out-of-line-call failure of the SHB23 any-N kernels: run with SMO_LIB pointing at an experimental
libsmo build whose dct2<0>/dct3<0> are NOT inlined, print — never assert — how every any-N entry point compares with the oracle, so that ONE
GPU run says which kernel, which sizes and which part of the result go wrong.  (tools/run_outline_abi.sh builds the variants and runs this.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spheremanopt_amd import _capi, shb23      # noqa: E402
from oracle import shb23 as osh                # noqa: E402


def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    if not np.all(np.isfinite(a)):
        return float("nan")
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    print("lib:", _capi.LIB_PATH, flush=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", "shb_helpers.npz"))
    dom = shb23.SHBDomain(8)
    v = g["v8"]
    print("transforms N=8 (shb_transform_kernel<0>): T %.1e Tinv %.1e Tadj %.1e Tinvadj %.1e" % (
        rel(shb23.transform(v, dom), g["T8"]), rel(shb23.transformInverse(v, dom), g["Tinv8"]),
        rel(shb23.transformAdjoint(v, dom), g["Tadj8"]), rel(shb23.transformInverseAdjoint(v, dom), g["Tinvadj8"])), flush=True)
    for N, n in ((20, 30), (127, 30), (500, 20), (1023, 6)):
        o = osh.SHB23Oracle(N, dt=1e-2, N_ITERS=n)
        X = osh.synthetic_ic(o, 42, 0.0019)
        d = shb23.SHBDomain(N)
        buf = shb23.GEN_BUFFER(N, d, n)
        J = shb23.FWD_Solve_IVP_Discrete([X], d, buf, n, 1e-2)
        gr = shb23.ADJ_Solve_IVP_Discrete([X], d, buf, n, 1e-2)[0]
        Jo = o.forward([X]); go = o.adjoint([X])[0]
        print("Discrete   N=%4d n=%3d (LDS %6d B): J err %.1e  snapshot[-1] err %.1e  grad err %.1e" % (
            N, n, 104 * N + 16512, abs(J - Jo) / abs(Jo), rel(buf['A_fwd'][:, -1], o.stack[:, -1]), rel(gr, go)), flush=True)
    # prediction of the root cause: with N = 1024 = the workgroup size no lane is ever masked by `tid < N`, so the
    # saves under a narrowed exec mask lose nothing and even the failing builds must give the right answer here
    os.environ["SMO_SHB_ANY"] = "1"
    for N, n in ((1024, 6),):
        o = osh.SHB23Oracle(N, dt=1e-2, N_ITERS=n)
        X = osh.synthetic_ic(o, 42, 0.0019)
        d = shb23.SHBDomain(N)
        buf = shb23.GEN_BUFFER(N, d, n)
        J = shb23.FWD_Solve_IVP_Discrete([X], d, buf, n, 1e-2)
        gr = shb23.ADJ_Solve_IVP_Discrete([X], d, buf, n, 1e-2)[0]
        Jo = o.forward([X]); go = o.adjoint([X])[0]
        print("Discrete   N=%4d n=%3d forced through the any-N kernels (no lane masked by tid < N): J err %.1e  grad err %.1e" % (
            N, n, abs(J - Jo) / abs(Jo), rel(gr, go)), flush=True)
    os.environ.pop("SMO_SHB_ANY")
    for N, n in ((20, 30), (50, 40), (100, 60), (250, 20), (333, 10)):
        o = osh.SHB23CntsOracle(N, dt=1e-2, N_ITERS=n)
        X = osh.synthetic_ic_cnts(o, 42, 0.0019)
        d = shb23.SHBDomain(N, dealias=2)
        buf = shb23.GEN_BUFFER(N, d, n)
        J = shb23.FWD_Solve_IVP_Cnts([X], d, buf, n, 1e-2)
        gr = shb23.ADJ_Solve_IVP_Cnts([X], d, buf, n, 1e-2)[0]
        Jo = o.forward([X]); go = o.adjoint([X])[0]
        e = np.abs(gr - go) / np.abs(go).max()
        print("Continuous N=%4d n=%3d (grid %4d, LDS %6d B): J err %.1e  snapshot[-1] err %.1e  grad err %.1e  (worst entry %d: %.3e vs %.3e; entries off by > 1e-6: %d of %d)" % (
            N, n, 2 * N, 104 * 2 * N + 16512, abs(J - Jo) / abs(Jo), rel(buf['A_fwd'][:, -1], o.stack[:, -1]), rel(gr, go),
            int(np.nanargmax(e)) if np.isfinite(e).any() else -1, gr[int(np.nanargmax(e))] if np.isfinite(e).any() else float("nan"),
            go[int(np.nanargmax(e))] if np.isfinite(e).any() else float("nan"), int((~(e < 1e-6)).sum()), e.size), flush=True)


if __name__ == "__main__":
    main()

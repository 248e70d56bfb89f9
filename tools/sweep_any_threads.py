import os, sys, time, json
sys.path.insert(0, os.getcwd())
from spheremanopt_amd import kdyn
from spheremanopt_amd.devvec import DeviceVector, to_device
n = 50
for N in (64, 128, 66):
    for nt in (128, 256, 512, 1024):
        os.environ["SMO_KD_ANY"] = "1"; os.environ["SMO_KD_ANY_NT"] = str(nt)
        dom = kdyn.KDynDomain(N); ctx = dom.context(1., 1e-3, n, "Final")
        X = to_device([kdyn.synthetic_field(dom.G, 1), kdyn.synthetic_field(dom.G, 2)])
        g = [DeviceVector(ctx.vec_len), DeviceVector(ctx.vec_len)]
        J = ctx.forward_dev(X); ctx.adjoint_dev(X, g)
        t0 = time.perf_counter()
        for _ in range(2): J = ctx.forward_dev(X); ctx.adjoint_dev(X, g)
        print(N, nt, "%.1f ms" % (1e3 * (time.perf_counter() - t0) / 2), flush=True)
        dom.drop_contexts()

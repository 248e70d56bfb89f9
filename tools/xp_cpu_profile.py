import os, sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from oracle.kdyn import ThreadedKDynOracle, synthetic_field
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G = 3 * N // 2
B, U = synthetic_field(G, 1), synthetic_field(G, 2)
topo = bench.cpu_topology()
os.sched_setaffinity(0, topo["cpus_used"])
for th in (1, 4, 8, 16):
    o = ThreadedKDynOracle(N, Rm=1., dt=1e-3, N_ITERS=2, threads=th); o.prewarm()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); o.forward([B, U]); o.adjoint([B, U]); pr.disable()
    print("threads", th, "%.2f s" % (time.perf_counter() - t0))
    if th in (1, 16):
        pstats.Stats(pr).sort_stats('tottime').print_stats(12)

#!/usr/bin/env python3
"""Headline benchmark: forward+adjoint gradient evaluations per second (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sh23|kdyn|shb23] [--no-cpu-baseline]

One "step" = one gradient evaluation = one forward solve (J) + one adjoint solve (grad J) at the same X over the
full time window.  On one GPU the timed steps hand over HOST vectors (H2D of X and D2H of grad J inside the timed
region); the device-resident rate rides along as config.value_device_vectors.  Prints ONE JSON line (rank 0).  See DESIGN.md
"Measurement".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


TIMING_STRIDE = int(os.environ.get("SMO_BENCH_TIMING_STRIDE", "8"))      # of the dominant kernel class, every n-th launch carries HIP events in the timed region
KDYN_SOURCES = ("kdyn.hip", "kdyn_any.hpp", "fft_lds.hpp", "comm.hpp", "smo_common.hpp")     # the translation unit the KDyn kernels are compiled from


def source_sha():
    """sha256 over the sources of the KDyn kernels (kdyn.hip and the headers it includes; names + contents): the identity a PMC summary
    must carry for bench.py to quote its traffic figures."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spheremanopt_amd", "csrc")
    for f in KDYN_SOURCES:
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_name, npts):
    """(HBM bytes per launch, provenance) of `kernel_name` at this grid from the rocprofv3 --pmc summary that tools/profile_round.sh
    leaves under profiles/ ((2*FETCH_SIZE + WRITE_SIZE)*1024, separate counter passes) — but ONLY if that summary was collected from
    exactly the kernel sources of this tree (source_sha stored beside it); otherwise (None, reason): a counter value from another
    build is not a measurement of this one."""
    import glob
    sha = source_sha()
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kdyn%d_pmc.json" % npts)))
    for f in reversed(cands):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("source_sha") != sha:
            continue
        k = d.get("kernels", {}).get(kernel_name)
        if k:
            return k["hbm_bytes_per_launch"], {"file": os.path.relpath(f, ROOT), "source_sha": sha, "instantiation": k.get("instantiation")}
    return None, {"reason": "no profiles/r*_kdyn%d_pmc.json collected from these kernel sources (source_sha %s)" % (npts, sha)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="number of GPUs = ranks (default: WORLD_SIZE if launched by torchrun, else 1); "
                    "N > 1 without a launcher: bench.py starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, help="sh23 | shb23 | kdyn | pois (default: kdyn, the largest single-GPU config of BASELINE.json)")
    ap.add_argument("--npts", type=int, default=None)
    ap.add_argument("--iters", type=int, default=None, help="override N_ITERS (the result is then flagged as reduced)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-steps", type=int, default=None, help="time steps of the CPU baseline sample (default 4)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short SH23 / SHB23 lines appended to the default run")
    ap.add_argument("--no-host-vectors", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg of the kdyn workload")
    ap.add_argument("--replicas", action="store_true", help="N>1: run N independent gradients instead of the slab decomposition")
    ap.add_argument("--devices", default=None, help="kdyn in ONE process over these GPUs (e.g. 0,1,2,3,4,5,6,7; a one-GPU box: 0,0): the multi-device "
                    "context of smo_create_multi — full host vectors in and out, peer pulls instead of RCCL; not combinable with --gpus > 1")
    ap.add_argument("--allow-replica-fallback", action="store_true", help="N>1: if the slab-decomposed path fails, report independent replicas "
                    "(flagged in config.slab_path_error) instead of exiting non-zero")
    return ap.parse_args()


def cpu_baseline_sh23(Npts, dt, n_iters, X, budget_s=10.0):
    """Oracle (NumPy/pocketfft restatement, 1 thread — the reference forces OMP_NUM_THREADS=1) timed on the host."""
    from oracle.sh23 import SH23Oracle
    o = SH23Oracle(Npts, dt=dt, N_ITERS=n_iters)
    o.forward([X]); o.adjoint([X])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 200:
        o.forward([X]); o.adjoint([X]); n += 1
    el = time.perf_counter() - t0
    return {"value": n / el, "unit": "gradient evals/s", "cores": 1, "kind": "port",
            "sample": "%d full forward+adjoint evaluations of the same workload (NumPy restatement of the Dedalus path)" % n}


def bench_sh23(a, torch, rank, world):
    from spheremanopt_amd import sh23
    Npts = a.npts or 256
    dt, n_iters = 0.1, a.iters or 500
    steps = a.steps if a.steps is not None else 200
    warm = a.warmup if a.warmup is not None else 20
    dom, X = sh23.Generate_IC(0.0725, Npts=Npts, seed=42 + rank)
    dom.device = torch.cuda.current_device()
    ctx = dom.context(dt, n_iters, batch=a.batch)
    Xd = torch.from_numpy(np.tile(X, a.batch)).cuda()
    Gd = torch.empty_like(Xd)
    for _ in range(warm):
        ctx.forward_dev([Xd]); ctx.adjoint_dev([Xd], [Gd])
    ctx.timing_enable(True)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.forward_dev([Xd]); ctx.adjoint_dev([Xd], [Gd])
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    tim = ctx.timing()
    dom_k = max(tim, key=lambda t: t["total_ms"])
    avg_ms = dom_k["total_ms"] / max(dom_k["launches"], 1)
    roof = {"bound": "hbm", "kernel": dom_k["kernel"], "achieved": dom_k["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9,
            "peak": 8000.0, "unit": "GB/s", "traffic": None, "avg_launch_ms": avg_ms,
            "note": "latency-bound config: one workgroup per problem, %d dependent steps per launch; us/step = %.3f"
                    % (n_iters, avg_ms * 1e3 / n_iters)}
    roof["frac"] = roof["achieved"] / roof["peak"]
    cfg = {"workload": "Swift-Hohenberg 1D Fourier Npts=%d T=%g dt=%g discrete adjoint" % (Npts, dt * n_iters, dt),
           "grid": 2 * Npts, "n_iters": n_iters, "batch": a.batch, "parallelism": "replicas only (x%d)" % world}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline_sh23(Npts, dt, n_iters, X)
    return steps, warm, el, a.batch, roof, cfg, cpu


def bench_shb23(a, torch, rank, world):
    from spheremanopt_amd import shb23
    N = a.npts or 512
    dt, n_iters = 1e-2, a.iters or 2000
    steps = a.steps if a.steps is not None else 20
    warm = a.warmup if a.warmup is not None else 2
    dom, X = shb23.Generate_IC(N, M_0=0.0019, seed=42 + rank, device=torch.cuda.current_device())
    ctx = dom.context(dt, n_iters, batch=a.batch)
    Xd = torch.from_numpy(np.tile(X, a.batch)).cuda()
    Gd = torch.empty_like(Xd)
    for _ in range(warm):
        ctx.forward_dev([Xd]); ctx.adjoint_dev([Xd], [Gd])
    ctx.timing_enable(True)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.forward_dev([Xd]); ctx.adjoint_dev([Xd], [Gd])
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    tim = ctx.timing()
    dom_k = max(tim, key=lambda t: t["total_ms"])
    avg_ms = dom_k["total_ms"] / max(dom_k["launches"], 1)
    roof = {"bound": "hbm", "kernel": dom_k["kernel"], "achieved": dom_k["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9,
            "peak": 8000.0, "unit": "GB/s", "traffic": None, "avg_launch_ms": avg_ms,
            "note": "latency-bound config: %d dependent steps per launch; batch 1 = a cluster of N^2/8192 workgroups with the %.1f MB "
                    "tau operator resident in their LDS + one all-gather per step, batch > 1 = one workgroup per problem streaming "
                    "it from L2; us/step = %.3f" % (n_iters, N * N * 8 / 1e6, avg_ms * 1e3 / n_iters)}
    roof["frac"] = roof["achieved"] / roof["peak"]
    cfg = {"workload": "Swift-Hohenberg 1D Chebyshev N=%d T=%g dt=%g discrete adjoint" % (N, dt * n_iters, dt),
           "grid": N, "n_iters": n_iters, "batch": a.batch, "parallelism": "replicas only (x%d)" % world}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import shb23 as osh
        o = osh.SHB23Oracle(N, dt=dt, N_ITERS=n_iters)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0 and n < 50:
            o.forward([X]); o.adjoint([X]); n += 1
        cpu = {"value": n / (time.perf_counter() - t0), "unit": "gradient evals/s", "cores": 1, "kind": "port",
               "sample": "%d full forward+adjoint evaluations of the same workload (NumPy restatement)" % n}
    return steps, warm, el, a.batch, roof, cfg, cpu


def bench_pois(a, torch, rank, world):
    """Plane-Poiseuille optimal mixing at the reference script's resolution (FWD_Solve_Poiseuille.py:1746-1762): a 'next' row of SURVEY 8f."""
    from spheremanopt_amd import poiseuille as pz
    Nx, Nz = 384, 192                                   # 3/2 * (256, 128)
    dt, n_iters = 5e-3, a.iters or 1000
    steps = a.steps if a.steps is not None else 3
    warm = a.warmup if a.warmup is not None else 1
    t0 = time.perf_counter()
    dom = pz.PoiseuilleDomain(Nx, Nz, device=torch.cuda.current_device())
    ctx = dom.context(500., 0.05, n_iters, dt, 1, 1., 0.125)
    build_s = time.perf_counter() - t0
    X = 1e-3 * np.random.RandomState(42 + rank).standard_normal(2 * Nx * Nz)
    Xd = torch.from_numpy(X).cuda()
    Gd = torch.empty_like(Xd)
    # warm-up passes time EVERY kernel class (the breakdown in `all_kernels`); the timed region then records HIP events only around the launches
    # of the HBM-bound class that takes longest (an operator apply), so that the instrumentation does not slow the other 13 launches of a
    # step pair (the path is bound by the hand-over between short dependent kernels)
    ctx.timing_enable(True)
    for _ in range(max(warm, 1)):
        ctx.forward_dev([Xd]); ctx.adjoint_dev([Xd], [Gd])
    tim_all = ctx.timing()
    dom_i = max(range(len(tim_all)), key=lambda i: tim_all[i]["total_ms"] if tim_all[i]["bytes_per_launch"] > 0 else -1.0)
    # (every TIMING_STRIDE-th launch of that class: a uniform sample at a fraction of the event overhead, see bench_kdyn)
    ctx.timing_enable(only=dom_i, every=TIMING_STRIDE)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        J = ctx.forward_dev([Xd]); ctx.adjoint_dev([Xd], [Gd])
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    tim = ctx.timing()
    dom_k = tim[dom_i]                                  # the HBM-bound kernel: an operator apply (the MFMA GEMMs are launch-bound); timed live in the region above
    avg_ms = dom_k["total_ms"] / max(dom_k["launches"], 1)
    roof = {"bound": "hbm", "kernel": dom_k["kernel"], "achieved": dom_k["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9, "peak": 8000.0,
            "unit": "GB/s", "traffic": None, "avg_launch_ms": avg_ms,
            "all_kernels": [{"kernel": t["kernel"], "launches": t["launches"], "avg_ms": t["total_ms"] / max(t["launches"], 1),
                             "GBps": t["bytes_per_launch"] * t["launches"] / max(t["total_ms"], 1e-9) / 1e6} for t in tim_all],
            "note": "bytes_per_launch = the tau operators one launch streams (HODLR form; the mean of the forward apply over the de-aliased "
                    "wavenumbers or the transposed apply over all of them, whichever class took longer)"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    cfg = {"workload": "Plane-Poiseuille optimal mixing (Discrete), Nx x Nz = %d x %d, Re=500, Ri=0.05, T=%g, dt=%g, mix-norm cost"
                       % (Nx, Nz, dt * n_iters, dt),
           "grid": [Nx, Nz], "n_iters": n_iters, "operator_build_s": build_s, "J": J, "parallelism": "replicas only (x%d)" % world}
    fx = os.path.join(ROOT, "tests", "golden", "oracle_poiseuille_%dx%d_n%d_s1.npz" % (Nx, Nz, n_iters))     # expected value of exactly this workload (data, not oracle code)
    if rank == 0 and os.path.exists(fx):
        try:
            Jo = float(np.load(fx)["J"])
            cfg["J_oracle_fixture"] = Jo
            cfg["J_matches_oracle_1e-6"] = bool(abs(J - Jo) <= 1e-6 * abs(Jo))
        except Exception:
            pass
    return steps, warm, el, 1, roof, cfg, None


def cpu_topology():
    """What the all-core CPU leg may use: the physical cores of ONE socket (socket of the first CPU this process may run on), cut down to
    the process' affinity mask and to the cgroup's CPU quota.  Returns a dict with the model name, sockets, physical cores per socket, the
    CPUs chosen (one hardware thread per physical core) and the limits that applied."""
    info = {"model": None, "sockets": None, "physical_cores_per_socket": None, "affinity_cpus": None, "cgroup_cpu_quota": None}
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                info["model"] = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    info["affinity_cpus"] = len(allowed)
    topo = {}
    for c in range(os.cpu_count() or 1):
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            topo[c] = (int(open(base + "physical_package_id").read()), int(open(base + "core_id").read()))
        except (OSError, ValueError):
            pass
    chosen = allowed
    if topo:
        socks = sorted({v[0] for v in topo.values()})
        info["sockets"] = len(socks)
        info["physical_cores_per_socket"] = len({v[1] for v in topo.values() if v[0] == socks[0]})
        s0 = topo.get(allowed[0], (socks[0], 0))[0]
        seen, chosen = set(), []
        for c in allowed:
            if c in topo and topo[c][0] == s0 and topo[c][1] not in seen:
                seen.add(topo[c][1]); chosen.append(c)
        info["socket_used"] = s0
    quota = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(f).read().split()
            if f.endswith("cpu.max"):
                quota = None if t[0] == "max" else float(t[0]) / float(t[1])
            else:
                q = float(t[0]); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = None if q <= 0 else q / per
            break
        except (OSError, ValueError, IndexError):
            continue
    info["cgroup_cpu_quota"] = quota
    if quota is not None and quota >= 1 and len(chosen) > int(quota):
        # fewer CPUs than the socket has cores: take them SPREAD over the socket (every len/quota-th physical core), not the first few — on a
        # chiplet CPU the first 16 cores are two CCDs that share two links to the memory controllers, 16 cores over eight CCDs have eight
        n, q = len(chosen), int(quota)
        chosen = [chosen[(i * n) // q] for i in range(q)]
    info["cpus_used"] = chosen
    return info


def cpu_baseline_kdyn(N, Rm, dt, n_iters, B, U, workers, sample_steps=4, cpus=None):
    """Oracle timed on a bounded sample: `sample_steps` forward + adjoint steps at the full grid, scaled to n_iters.  The restatement with
    its pointwise stages threaded (oracle.kdyn.ThreadedKDynOracle: same arithmetic, bit-identical gradients); `cpus`: pin this leg to
    those CPUs (one socket's physical cores) for its duration."""
    from oracle.kdyn import ThreadedKDynOracle
    old = None
    if cpus and hasattr(os, "sched_setaffinity"):
        try:
            old = os.sched_getaffinity(0)
            os.sched_setaffinity(0, cpus)          # threads started from here on (the pool's, pocketfft's) inherit the mask
        except OSError:
            old = None
    try:
        o = ThreadedKDynOracle(N, Rm=Rm, dt=dt, N_ITERS=sample_steps, threads=workers)
        o.prewarm()                                   # work arrays allocated and touched before the clock starts
        t0 = time.perf_counter()
        o.forward([B, U]); o.adjoint([B, U])
        el = time.perf_counter() - t0
        if o.pool is not None:
            o.pool.shutdown()
    finally:
        if old is not None:
            os.sched_setaffinity(0, old)
    per_step = el / sample_steps          # includes the one-off transforms of X and the final gradient transforms
    return {"value": 1.0 / (per_step * n_iters), "unit": "gradient evals/s", "cores": workers, "kind": "port",
            "sample": "%d of %d forward+adjoint time steps at the full %d^3 grid (NumPy/pocketfft restatement of the Dedalus "
                      "path, %d thread(s)%s), %.1f s, extrapolated linearly" % (sample_steps, n_iters, N, workers,
                      ", pinned to one socket's physical cores" if cpus else "", el)}


class _PyLoop:
    """kdyn_slab.SlabKDyn (Python time loop, torch.distributed collectives) behind the attribute names bench.py reads from LibSlabKDyn."""

    def __init__(self, p):
        self.p, self.ctx, self.K = p, p.ops.ctx, p.K
        self.transport = "torch.distributed all_to_all_single (Python loop)"
        self.exchanges_per_step_pair = 3 + p.adj_groups
        self.forward, self.adjoint, self.local_slab = p.forward, p.adjoint, p.local_slab


def _rccl_library():
    """File of the librccl that libsmo bound for its own communicator (smo_comm_library: dladdr of its ncclGetUniqueId)."""
    from spheremanopt_amd import _capi
    try:
        return _capi.lib().smo_comm_library().decode() or None
    except Exception:
        return None


def _slab_elems(N, world):
    """complex128 elements of ONE field group of one rank's exchange buffer, all peers: 3 * (a/W) * m * G."""
    return 3 * (N // 2 // world) * (N - 1) * (3 * N // 2)


def _slab_run(torch, N, Rm, dt, n_iters, steps, warm, ckpt=1):
    """Build the slab solver for an N^3 problem on this rank's GPU and time `steps` gradients (barrier + sync on both sides)."""
    from spheremanopt_amd import kdyn
    from spheremanopt_amd.kdyn_slab import LibSlabKDyn
    G = 3 * N // 2
    # a rank that cannot build its solver (e.g. not enough HBM for its share of the stack) must not leave the others waiting in the
    # first all-to-all: every rank reports, and all raise together.  Two product paths are tried in turn, both HIP + RCCL:
    #   1. time loop + transposes inside libsmo on its own RCCL communicator (torch.distributed only carries the 128-byte unique id;
    #      with gloo — ranks sharing a GPU in tests — the host-staged callback transport)
    #   2. the Python loop over the phase-level entry with torch.distributed's all_to_all_single (round 1's path), should the
    #      library's communicator not come up on this node
    from spheremanopt_amd.kdyn_slab import SlabKDyn
    cdev = "cpu" if torch.distributed.get_backend() == "gloo" else "cuda"
    s, last_err = None, None
    for attempt, make in enumerate((lambda: LibSlabKDyn(N, Rm, dt, n_iters, "Final", device=torch.cuda.current_device(), ckpt=ckpt),
                                    lambda: _PyLoop(SlabKDyn(N, Rm, dt, n_iters, "Final", device=torch.cuda.current_device())))):
        if attempt == 1 and ckpt != 1:
            break                                            # the Python loop has no windowed checkpoints with slabs
        err = None
        try:
            s = make()
        except Exception as e:
            err = e
        if attempt == 0 and os.environ.get("SMO_BENCH_INJECT_FAILURE") == str(torch.distributed.get_rank()):     # test hook
            s, err = None, RuntimeError("injected construction failure")
        if attempt == 0 and os.environ.get("SMO_BENCH_INJECT_LIB_FAILURE") == "1":                               # test hook: every rank
            s, err = None, RuntimeError("injected failure of the in-library communicator")
        if attempt == 1 and os.environ.get("SMO_BENCH_INJECT_FAILURE") is not None:
            s, err = None, RuntimeError("injected construction failure")
        bad = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=cdev)
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
        if float(bad.item()) == 0:
            break
        s = None
        last_err = err if err is not None else RuntimeError("slab solver construction failed on another rank")
        sys.stderr.write("rank %d: slab solver path %d failed (%r)\n" % (torch.distributed.get_rank(), attempt, last_err))
        torch.cuda.empty_cache()
    if s is None:
        raise RuntimeError("slab solver construction failed: %r" % (last_err,))
    Bl = s.local_slab(kdyn.synthetic_field(G, 1)); Ul = s.local_slab(kdyn.synthetic_field(G, 2))
    out = [torch.empty_like(Bl), torch.empty_like(Ul)]
    # how many pipelined chunks hide the transposes best depends on the node's xGMI rate: try the candidates once (untimed region), unless
    # SMO_SLAB_CHUNKS pins the choice
    s.chunk_autotune = None
    if hasattr(s, "autotune_chunks") and "SMO_SLAB_CHUNKS" not in os.environ and os.environ.get("SMO_BENCH_AUTOTUNE", "1") != "0":
        s.chunk_autotune = {str(k): v for k, v in s.autotune_chunks([Bl, Ul]).items()}
    # warm-up passes time every kernel class (breakdown); the timed region records HIP events only around the dominant one, as on one GPU
    # (events around all ~8 launches of a step pair would cost about as much as a thin slab's kernel)
    s.ctx.timing_enable(True)
    for _ in range(warm):
        s.forward([Bl, Ul]); s.adjoint("Discrete", out)
    tim_all = s.ctx.timing() if warm else None
    if tim_all:
        dom_i = max(range(len(tim_all)), key=lambda i: tim_all[i]["total_ms"] if tim_all[i]["hbm_bytes_per_launch"] > 0 else -1.0)
        ex_i = [i for i, t in enumerate(tim_all) if t["kernel"].startswith("slab_exchange")]
        s.ctx.timing_enable(select=[dom_i] + ex_i)       # the dominant kernel class + every grouped send/recv (events on the stream it runs on)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        J = s.forward([Bl, Ul]); s.adjoint("Discrete", out)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    el = time.perf_counter() - t0
    s.warmup_timing = tim_all
    return s, J, el


def _slab_time_split(s, tim, el, steps, warm, n_iters):
    """Where a step pair's time goes on this rank: kernels (every class but the exchanges, from the warm-up breakdown, which times every
    launch) and the transposes (HIP events around every grouped send/recv in the TIMED region, on the stream they run on: with K > 1 chunks
    they overlap the kernels, so compute + exchange may exceed the wall time per step pair)."""
    def is_ex(t):
        return t["kernel"].startswith("slab_exchange")
    brk = s.warmup_timing or tim
    ex_t = [t for t in tim if is_ex(t)]
    return {"compute_ms_per_step_pair": sum(t["total_ms"] for t in brk if not is_ex(t)) / (max(warm, 1) if s.warmup_timing else steps) / n_iters,
            "exchange_ms_per_step_pair": (sum(t["total_ms"] for t in ex_t) / steps / n_iters) if ex_t and ex_t[0]["launches"] else None,
            "exchange_calls_per_step_pair": (sum(t["launches"] for t in ex_t) / steps / n_iters) if ex_t else None,
            "wall_ms_per_step_pair": 1e3 * el / steps / n_iters}


def bench_kdyn_slab(a, torch, rank, world):
    """N > 1: ONE 128^3 gradient slab-decomposed over the N GPUs (strong scaling), RCCL all-to-all pencil transposes."""
    from spheremanopt_amd import kdyn
    N = a.npts or 128
    Rm, dt = 1.0, 1e-3
    n_iters = a.iters or 1000
    steps = a.steps if a.steps is not None else 2
    warm = a.warmup if a.warmup is not None else 1
    G = 3 * N // 2
    # integrity check of the decomposition: rank 0 first evaluates J on its own GPU with the single-GPU path (same kernels, no
    # exchange); every rank's slab result must agree with it to 1e-9 relative
    cdev = "cpu" if torch.distributed.get_backend() == "gloo" else "cuda"
    J_single = torch.zeros(2, dtype=torch.float64, device=cdev)            # [J, failed]
    if rank == 0:
        try:
            dom1 = kdyn.KDynDomain(N, device=torch.cuda.current_device(), ckpt=0)      # windowed checkpoints if the whole stack does not fit one GPU
            J_single[0] = dom1.context(Rm, dt, n_iters, "Final").forward([kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)])
            dom1.drop_contexts()
        except Exception as e:                   # the other ranks are waiting in the broadcast: tell them instead of leaving
            sys.stderr.write("rank 0: single-GPU reference J failed (%r)\n" % (e,))
            J_single[1] = 1.0
    torch.distributed.broadcast(J_single, 0)
    if float(J_single[1].item()) > 0:
        raise RuntimeError("single-GPU reference J could not be computed on rank 0")       # raised on EVERY rank
    s, J, el = _slab_run(torch, N, Rm, dt, n_iters, steps, warm)
    tim = s.ctx.timing()                                       # timed region: the dominant class only (all of them if there was no warm-up)
    dom_k = max(tim, key=lambda t: t["total_ms"] if t["hbm_bytes_per_launch"] > 0 else -1.0)
    avg_ms = dom_k["total_ms"] / max(dom_k["launches"], 1)
    brk = s.warmup_timing or tim                               # per-class breakdown from the warm-up gradients
    brk_wall = (el / steps * max(warm, 1)) if s.warmup_timing else el
    split = _slab_time_split(s, tim, el, steps, warm, n_iters)
    roof = {"bound": "hbm", "kernel": dom_k["kernel"], "achieved": dom_k["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9, "peak": 8000.0,
            "unit": "GB/s", "traffic": None, "avg_launch_ms": avg_ms, "per_gpu": True,
            "bytes_per_launch": dom_k["hbm_bytes_per_launch"],
            "achieved_algorithmic": dom_k["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9,
            "kernel_busy_fraction_of_wall": sum(t["total_ms"] for t in brk) / (1e3 * brk_wall),
            "all_kernels": [{"kernel": t["kernel"], "launches": t["launches"], "avg_ms": t["total_ms"] / max(t["launches"], 1)} for t in brk]}
    roof["frac"] = roof["achieved"] / roof["peak"]
    cfg = {"workload": "Kinematic dynamo 3D Fourier %d^3, Rm=%g, T=%g, dt=%g, two-field (U,B) gradient, Final cost, discrete adjoint"
                       % (N, Rm, dt * n_iters, dt),
           "grid": [G, G, G], "n_iters": n_iters, "J": J, "stack_GB_per_gpu": s.ctx.stack_bytes / 1e9,
           "slab_J_matches_single_gpu": bool(abs(J - float(J_single[0].item())) <= 1e-9 * abs(float(J_single[0].item()))),
           "J_single_gpu": float(J_single[0].item()),
           "parallelism": "slab x%d (kx / z decomposition; all-to-all transposes between the z and y passes, transport %s; "
                          "%d field-group exchanges per step pair, %d pipelined z chunks)"
                          % (world, {"rccl": "RCCL grouped send/recv", "callback": "callback (host-staged, test only)"}.get(s.transport, s.transport),
                             s.exchanges_per_step_pair, s.K),
           "transport": s.transport, "chunk_autotune_s": getattr(s, "chunk_autotune", None),
           **split,
           "rccl_library": _rccl_library() if s.transport == "rccl" else None,
           "exchange_MB_sent_per_gpu_per_step_pair": s.exchanges_per_step_pair * _slab_elems(N, world) * 16 / 1e6 * (world - 1) / world,
           "grid_states_kept_GB_per_gpu": s.ctx.get(1) / 1e9}
    # BASELINE configs[4] rides along when the default workload is run: ONE 256^3 gradient over the same GPUs (not `value`)
    big = int(os.environ.get("SMO_BENCH_SLAB_EXTRA_NPTS", "256"))
    if a.npts is None and a.iters is None and not a.no_secondary and cfg["slab_J_matches_single_gpu"] and (big // 2) % world == 0:
        del s
        torch.cuda.empty_cache()
        try:
            # ckpt = 0: every rank takes the smallest checkpoint interval whose share of the stack fits its free HBM (2 GPUs: every
            # snapshot fits; the ranks agree on the interval inside smo_comm_init)
            st2 = int(os.environ.get("SMO_BENCH_256_STEPS", "2"))
            s2, J2, el2 = _slab_run(torch, big, Rm, dt, n_iters, st2, 1, ckpt=0)
            cfg["config_256"] = {"workload": "Kinematic dynamo 3D Fourier %d^3 slab-decomposed across %d GPUs, T=%g, dt=%g" % (big, world, dt * n_iters, dt),
                                 "ms_per_gradient": 1e3 * el2 / st2, "gradient_evals_per_s": st2 / el2, "steps": st2, "warmup": 1, "J": J2,
                                 **_slab_time_split(s2, s2.ctx.timing(), el2, st2, 1, n_iters),
                                 "stack_GB_per_gpu": s2.ctx.stack_bytes / 1e9, "checkpoint_interval": int(s2.ctx.get(0)),
                                 "chunks": s2.K, "chunk_autotune_s": getattr(s2, "chunk_autotune", None),
                                 "exchange_MB_sent_per_gpu_per_step_pair": s2.exchanges_per_step_pair * _slab_elems(big, world) * 16 / 1e6 * (world - 1) / world}
            del s2
        except Exception as e:                       # never lose the main line because of the extra
            cfg["config_256"] = {"error": repr(e)}
    return steps, warm, el, 1, roof, cfg, None, "strong"


def bench_kdyn(a, torch, rank, world):
    from spheremanopt_amd import _capi, kdyn
    N = a.npts or 128
    Rm, dt = 1.0, 1e-3
    n_iters = a.iters or 1000
    steps = a.steps if a.steps is not None else (5 if N <= 128 else 2)        # 0.46 s per step at 128^3, 5.1 s at 256^3
    warm = a.warmup if a.warmup is not None else 1
    dom, B, U = kdyn.Generate_IC(N, U_Noise=True, device=torch.cuda.current_device())
    dom.ckpt = 0                              # keep every snapshot if the stack fits the HBM, else the smallest window that does
    ctx = dom.context(Rm, dt, n_iters, "Final")
    ck = int(ctx.get(0))
    Bd, Ud = torch.from_numpy(B).cuda(), torch.from_numpy(U).cuda()
    gB, gU = torch.empty_like(Bd), torch.empty_like(Ud)
    # warm-up passes time EVERY kernel class (breakdown + which kernel dominates); the timed region then records HIP events only
    # around the dominant kernel's launches so that the instrumentation does not slow the other 14 launches of a step pair
    ctx.timing_enable(True)
    for _ in range(max(warm, 1)):
        ctx.forward_dev([Bd, Ud]); ctx.adjoint_dev([Bd, Ud], [gB, gU])
    tim = ctx.timing()
    tot_ms = sum(t["total_ms"] for t in tim)
    # dominant kernel = the byte-moving class with the largest share (the misc class — setup, reductions — has no byte model)
    dom_i = max(range(len(tim)), key=lambda i: tim[i]["total_ms"] if tim[i]["hbm_bytes_per_launch"] > 0 else -1.0)
    share = tim[dom_i]["total_ms"] / tot_ms
    every_ms = tim[dom_i]["total_ms"] / max(tim[dom_i]["launches"], 1)      # warm-up gradient(s): HIP events around EVERY launch of every class
    # THE TIMED REGION.  SURVEY 8d's metric: one f + one Grad_f at the same X "including H2D of X and D2H of grad J" — the callbacks
    # the reference's optimiser calls hand over host vectors.  On one GPU the timed steps therefore go through the host-buffer entry
    # points (smo_forward / smo_adjoint on pinned vectors); the device-resident rate (vectors already in HBM: what devvec.DeviceVector
    # callers get) is measured right after and rides along as config.value_device_vectors.  Replica / slab legs (world > 1, or
    # --no-host-vectors) time the device-resident form.
    host_primary = world == 1 and not getattr(a, "no_host_vectors", False)
    if host_primary:
        hX = [_capi.pinned_copy(B), _capi.pinned_copy(U)]
        hG = [_capi.pinned_empty(B.size), _capi.pinned_empty(U.size)]
        ctx.timing_enable(False)
        ctx.forward(hX); ctx.adjoint(None, out=hG)           # first touch of the staging path
    # every 8th launch of the dominant class is timed: a uniform sample of its launches over the whole timed region (125 per gradient at
    # 1000 steps) at an eighth of the event overhead — two event records per launch on every launch cost 6-7 ms per 128^3 gradient
    # (each is a marker in the queue that keeps the next kernel from starting under the previous one's tail)
    ctx.timing_enable(only=dom_i, every=TIMING_STRIDE)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        if host_primary:
            J = ctx.forward(hX); ctx.adjoint(None, out=hG)
        else:
            J = ctx.forward_dev([Bd, Ud]); ctx.adjoint_dev([Bd, Ud], [gB, gU])
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    dom_k = ctx.timing()[dom_i]
    avg_ms = dom_k["total_ms"] / max(dom_k["launches"], 1)

    def rate(nbytes, ms):
        return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

    # roofline of the dominant kernel.  `achieved` = compulsory HBM bytes of the kernel AS FUSED (each input read once, each output
    # written once: DESIGN.md section 4; what a perfect implementation of this kernel must move) / measured launch time, so that
    # frac <= 1 by construction.  SURVEY 8d's unfused count (every axis pass of every field reads + writes HBM) is carried beside
    # it as `achieved_algorithmic`: it prices passes the fusion removed and can exceed the peak.
    roof = {"bound": "hbm", "kernel": dom_k["kernel"], "achieved": rate(dom_k["hbm_bytes_per_launch"], avg_ms),
            "peak": 8000.0, "unit": "GB/s", "traffic": None, "avg_launch_ms": avg_ms, "launches_timed": dom_k["launches"],
            "timing_stride": TIMING_STRIDE, "bytes_per_launch": dom_k["hbm_bytes_per_launch"],
            # two views of the same kernel's launch time: the sample taken INSIDE the timed region (what `achieved` / `frac` use) and the
            # warm-up gradient's every-launch average (events around all classes: every kernel runs fenced, at ~9 % more wall time)
            "avg_launch_ms_sampled": avg_ms, "avg_launch_ms_every_launch": every_ms,
            "frac_basis": "avg_launch_ms_sampled: every %d-th launch of the dominant class INSIDE the timed region, begin / end timestamps of the "
                          "dispatch itself (hipExtLaunchKernelGGL start / stop events: what rocprofv3 --kernel-trace reports, no marker packets "
                          "in the queue).  The same kernel reads 3-5 %% shorter when every launch of every class is instrumented "
                          "(avg_launch_ms_every_launch, and any rocprofv3 run): an instrumented queue has idle gaps between its kernels and the "
                          "chip runs them at a higher clock (DESIGN.md section 5)" % TIMING_STRIDE,
            "frac_every_launch": rate(dom_k["hbm_bytes_per_launch"], every_ms) / 8000.0,
            "achieved_algorithmic": rate(dom_k["bytes_per_launch"], avg_ms), "algorithmic_bytes_per_launch": dom_k["bytes_per_launch"],
            "kernel_time_share": share,
            "all_kernels": [{"kernel": t["kernel"], "launches": t["launches"], "avg_ms": t["total_ms"] / max(t["launches"], 1),
                             "GBps": rate(t["hbm_bytes_per_launch"], t["total_ms"] / max(t["launches"], 1)) if t["launches"] else 0.0,
                             "GBps_algorithmic": rate(t["bytes_per_launch"], t["total_ms"] / max(t["launches"], 1)) if t["launches"] else 0.0}
                            for t in tim]}
    roof["frac"] = roof["achieved"] / roof["peak"]
    if world == 1:
        # HBM bytes per launch from the PMC counters: quoted only from a summary collected from exactly these kernel sources
        roof["traffic"], roof["traffic_source"] = pmc_traffic(dom_k["kernel"], N)
        if roof["traffic"]:
            roof["measured_traffic_GBps"] = rate(roof["traffic"], avg_ms)
    roof["note"] = ("achieved = compulsory HBM bytes of the fused kernel / launch time measured with HIP events in this run; "
                    "achieved_algorithmic = SURVEY 8d's unfused byte count / the same time (may exceed the peak); traffic = PMC "
                    "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, or null when no counter run of these sources is committed")
    # whole-job figures: the compulsory bytes of every launch of one gradient (from the warm-up breakdown) and SURVEY 8d's per-step
    # bytes (fwd 6T+9S3+12S0, adj 12T+15S3+24S0)
    a_, m_, G_ = N // 2, N - 1, 3 * N // 2
    S0, S1, S2, S3 = 16. * a_ * m_ * m_, 16. * a_ * m_ * G_, 16. * a_ * G_ * G_, 8. * G_ ** 3
    T = S0 + 2 * S1 + 2 * S2 + S3
    per_grad = n_iters * ((6 * T + 9 * S3 + 12 * S0) + (12 * T + 15 * S3 + 24 * S0))
    per_grad_hbm = sum(t["hbm_bytes_per_launch"] * t["launches"] for t in tim) / max(warm, 1)
    roof["whole_gradient_TB"] = per_grad_hbm / 1e12
    roof["whole_gradient_GBps"] = per_grad_hbm / (el / steps) / 1e9
    roof["whole_gradient_frac"] = roof["whole_gradient_GBps"] / roof["peak"]
    roof["whole_gradient_algorithmic_TB"] = per_grad / 1e12
    roof["whole_gradient_algorithmic_GBps"] = per_grad / (el / steps) / 1e9
    # the same gradient with the vectors already resident in HBM (smo_forward_dev / smo_adjoint_dev), a few evaluations: beside `value`
    dev = None
    if host_primary:
        ds = max(1, min(steps, 3))
        ctx.timing_enable(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ds):
            Jd = ctx.forward_dev([Bd, Ud]); ctx.adjoint_dev([Bd, Ud], [gB, gU])
        torch.cuda.synchronize()
        ed = time.perf_counter() - t1
        dev = {"value": ds / ed, "unit": "gradient evals/s", "ms_per_step": 1e3 * ed / ds, "steps": ds, "J_equal": bool(Jd == J),
               "grad_equal": bool(np.array_equal(gB.cpu().numpy(), hG[0]) and np.array_equal(gU.cpu().numpy(), hG[1])),
               "note": "smo_forward_dev + smo_adjoint_dev: X and grad J stay in HBM (no PCIe traffic in the timed steps)"}
    cfg = {"workload": "Kinematic dynamo 3D Fourier %d^3, Rm=%g, T=%g, dt=%g, two-field (U,B) gradient, Final cost, discrete adjoint"
                       % (N, Rm, dt * n_iters, dt),
           "grid": [G_, G_, G_], "n_iters": n_iters, "stack_GB": ctx.stack_bytes / 1e9, "checkpoint_interval": ck, "y_side_stack_GB": ctx.get(1) / 1e9, "J": J,
           "parallelism": "1 GPU" if world == 1 else "replicas only (x%d independent gradients)" % world}
    cfg["vectors"] = ("host (pinned): H2D of X (2 x %.0f MB) and D2H of grad J (2 x %.0f MB) inside the timed region, SURVEY 8d" % (B.nbytes / 1e6, B.nbytes / 1e6)
                      if host_primary else "device-resident (HBM)")
    if dev:
        cfg["value_device_vectors"] = dev
    # the timed gradient against the committed oracle value for exactly this workload (data under tests/golden/, not oracle code)
    fx = os.path.join(ROOT, "tests", "golden", "oracle_kdyn_c4_%d_n%d.npz" % (N, n_iters))
    if os.path.exists(fx):
        try:
            Jo = float(np.load(fx)["J_Final"])
            cfg["J_oracle_fixture"] = Jo
            cfg["J_matches_oracle_1e-6"] = bool(abs(J - Jo) <= 1e-6 * abs(Jo))
        except Exception:
            pass
    # Inner_Prod_3 (2 x vector bytes per call; the optimiser calls it ~30x per iteration): HIP events on every launch of a short
    # series of device-resident calls, and the wall time of a call (kernel + the D2H of the 1024 partial sums + the host reduction)
    try:
        dot_i = [i for i, t in enumerate(tim) if t["kernel"].startswith("kd_dot")][0]
        nd = 20
        ctx.inner_dev(Bd, gB)
        ctx.timing_enable(only=dot_i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nd):
            ip = ctx.inner_dev(Bd, gB)
        wall_ms = 1e3 * (time.perf_counter() - t1) / nd
        dk = ctx.timing()[dot_i]
        dms = dk["total_ms"] / max(dk["launches"], 1)
        roof["inner_product"] = {"kernel": dk["kernel"], "bytes_per_call": dk["hbm_bytes_per_launch"], "avg_launch_ms": dms, "launches_timed": dk["launches"],
                                 "GBps": rate(dk["hbm_bytes_per_launch"], dms), "frac": rate(dk["hbm_bytes_per_launch"], dms) / 8000.0,
                                 "wall_ms_per_call": wall_ms, "value": ip,
                                 "note": "smo_inner_dev (Inner_Prod_3, FWD_Solve_KDyn.py:173-181) on device-resident vectors: reads both vectors once"}
        ctx.timing_enable(False)
    except Exception as e:                               # never lose the main line because of the extra
        roof["inner_product"] = {"error": repr(e)}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        ss = getattr(a, "cpu_sample_steps", None) or 4      # (--cpu-sample-steps; the 256^3 leg of the default run takes 2)
        topo = cpu_topology()
        cpu = cpu_baseline_kdyn(N, Rm, dt, n_iters, B, U, 1, sample_steps=ss, cpus=topo["cpus_used"][:1])
        # the all-core leg: ONE socket's physical cores (what north_star's "single-socket CPU baseline" names), never the whole box's threads
        nthr = max(1, len(topo["cpus_used"]))
        allc = cpu_baseline_kdyn(N, Rm, dt, n_iters, B, U, nthr, sample_steps=ss, cpus=topo["cpus_used"])
        allc["cpu"] = {k: topo[k] for k in ("model", "sockets", "physical_cores_per_socket", "affinity_cpus", "cgroup_cpu_quota")}
        allc["cores"] = nthr
        allc["scaling_over_1_core"] = allc["value"] / cpu["value"]
        cfg["cpu_all_cores"] = allc
        cfg["cpu_single_socket"] = allc
    return steps, warm, el, 1, roof, cfg, cpu


def bench_kdyn_multi(a, torch, devices):
    """ONE process, ONE context (smo_create_multi): the gradient slab-decomposed over `devices` with the reference's full host vectors."""
    from spheremanopt_amd import _capi, kdyn
    N = a.npts or 128
    Rm, dt = 1.0, 1e-3
    n_iters = a.iters or 1000
    steps = a.steps if a.steps is not None else 2
    warm = a.warmup if a.warmup is not None else 1
    G = 3 * N // 2
    B, U = kdyn.synthetic_field(G, 1), kdyn.synthetic_field(G, 2)
    ctx = _capi.MultiContext(N, (0., 2. * np.pi), dt, n_iters, Rm, devices, cost="Final", ckpt=0)
    hX = [_capi.pinned_copy(B), _capi.pinned_copy(U)]
    hG = [_capi.pinned_empty(B.size), _capi.pinned_empty(U.size)]
    ctx.timing_enable(True)                                   # device 0's kernels: breakdown from the warm-up gradient
    for _ in range(max(warm, 1)):
        ctx.forward(hX); ctx.adjoint(None, out=hG)
    tim = ctx.timing()
    dom_i = max(range(len(tim)), key=lambda i: tim[i]["total_ms"] if tim[i]["hbm_bytes_per_launch"] > 0 else -1.0)
    ex_i = [i for i, t in enumerate(tim) if t["kernel"].startswith("slab_exchange")]
    ctx.timing_enable(select=[dom_i] + ex_i)
    t0 = time.perf_counter()
    for _ in range(steps):
        J = ctx.forward(hX); ctx.adjoint(None, out=hG)
    el = time.perf_counter() - t0
    t2 = ctx.timing()
    dom_k = t2[dom_i]
    avg_ms = dom_k["total_ms"] / max(dom_k["launches"], 1)
    ex = [t2[i] for i in ex_i]
    pull = {2.0: "one gather kernel per exchange reading every peer's send buffer (peer access)", 1.0: "hipMemcpyPeerAsync, one call per peer"}.get(ctx.comm_get(3), "?")
    rendezvous = ctx.comm_get(4)
    # host time the slowest worker spent ISSUING the last forward + adjoint solve (smo_get key 4: call entry until everything is enqueued, minus
    # its waits for the other workers in host rendezvous), per step pair — at the bench grid itself
    issue_ms = ctx.get(4) / n_iters
    # What the HOST costs per step pair with this many workers: the same loop at a grid whose kernels take microseconds (same launches, events and
    # spin-barrier rendezvous per step pair as at the bench grid; the GPU is never the bottleneck there), so its wall time per step pair is the
    # time the slowest worker needs to ISSUE a step pair — to be set against the kernels' time per step pair of the real grid.
    host_issue = None
    try:
        Nh = 32 if (16 % len(devices) == 0 and (48 // len(devices)) % 2 == 0) else None
        if Nh:
            hi = 20                                       # few steps: every packet of a solve fits the queues, no launch ever waits for the GPU
            hctx = _capi.MultiContext(Nh, (0., 2. * np.pi), dt, hi, Rm, devices, cost="Final", ckpt=1)
            Gh = 3 * Nh // 2
            hx = [_capi.pinned_copy(kdyn.synthetic_field(Gh, 1)), _capi.pinned_copy(kdyn.synthetic_field(Gh, 2))]
            hg = [_capi.pinned_empty(hx[0].size), _capi.pinned_empty(hx[0].size)]
            hctx.forward(hx); hctx.adjoint(None, out=hg)
            r0 = hctx.comm_get(4)
            th = time.perf_counter()
            iss = []
            for _ in range(3):
                hctx.forward(hx); hctx.adjoint(None, out=hg)
                iss.append(hctx.get(4) / hi)
            host_issue = {"wall_ms_per_step_pair": 1e3 * (time.perf_counter() - th) / 3 / hi, "issue_ms_per_step_pair_min_of_3": min(iss), "issue_ms_per_step_pair": hctx.get(4) / hi,
                          "grid": "%d^3" % Nh, "n_iters": hi, "workers": len(devices),
                          "chunks": int(hctx.comm_get(0)), "rendezvous_per_step_pair": (hctx.comm_get(4) - r0) / 3 / hi,
                          "note": "the same multi-device loop at a grid whose kernels take microseconds (same launches, event records / waits and "
                                  "spin-barrier rendezvous per step pair): wall time per step pair (an upper bound of the loop's fixed costs; with "
                                  "all workers on ONE GPU it includes that GPU executing every rank's launches) and the slowest worker's issue time"}
            hctx.close()
    except Exception as e:
        host_issue = {"error": repr(e)}
    roof = {"bound": "hbm", "kernel": dom_k["kernel"], "achieved": dom_k["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9, "peak": 8000.0,
            "unit": "GB/s", "traffic": None, "avg_launch_ms": avg_ms, "per_gpu": True, "bytes_per_launch": dom_k["hbm_bytes_per_launch"],
            "all_kernels": [{"kernel": t["kernel"], "launches": t["launches"], "avg_ms": t["total_ms"] / max(t["launches"], 1)} for t in tim]}
    roof["frac"] = roof["achieved"] / roof["peak"]
    cfg = {"workload": "Kinematic dynamo 3D Fourier %d^3, Rm=%g, T=%g, dt=%g, two-field (U,B) gradient, Final cost, discrete adjoint" % (N, Rm, dt * n_iters, dt),
           "grid": [G, G, G], "n_iters": n_iters, "J": J, "devices": list(devices),
           "parallelism": "ONE process, slab x%d over devices %s (smo_create_multi: one persistent worker thread per device, transposes = peer pulls "
                          "ordered by HIP events — %s; no RCCL)" % (len(devices), list(devices), pull),
           "transpose_pull": pull, "host_rendezvous_per_step_pair": rendezvous / max(steps + max(warm, 1), 1) / n_iters,
           "host_issue_ms_per_step_pair": issue_ms, "host_bound_loop": host_issue,
           "vectors": "host (pinned), the reference's full vectors: scatter of X and gather of grad J inside the timed region",
           "checkpoint_interval": int(ctx.get(0)), "chunks": int(ctx.comm_get(0)),
           "compute_ms_per_step_pair": sum(t["total_ms"] for t in tim if not t["kernel"].startswith("slab_exchange")) / max(warm, 1) / n_iters,
           "exchange_ms_per_step_pair": (sum(t["total_ms"] for t in ex) / steps / n_iters) if ex else None,
           "wall_ms_per_step_pair": 1e3 * el / steps / n_iters, "stack_GB": ctx.stack_bytes / 1e9}
    fx = os.path.join(ROOT, "tests", "golden", "oracle_kdyn_c4_%d_n%d.npz" % (N, n_iters))
    if os.path.exists(fx):
        Jo = float(np.load(fx)["J_Final"])
        cfg["J_oracle_fixture"], cfg["J_matches_oracle_1e-6"] = Jo, bool(abs(J - Jo) <= 1e-6 * abs(Jo))
    ctx.close()
    return steps, warm, el, 1, roof, cfg, None


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run, one process per GPU) and
    pass their output through.  Runs BEFORE this process imports torch or touches the GPU — a process that has initialised the GPU
    must never be replaced or forked into ranks."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus is not None and a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus is not None and a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to report a line whose "
                         "n_gpus would not be the number of GPUs asked for" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("SMO_BENCH_BACKEND", "nccl")       # "gloo": ranks sharing one GPU (tests on a one-GPU box)
    if world > ndev and backend == "nccl":
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s): RCCL needs one GPU per rank (SMO_BENCH_BACKEND=gloo lets ranks "
                         "share a GPU for tests)" % (world, ndev))
    torch.cuda.set_device(local % ndev)
    if world > 1:
        import datetime
        # a rank that dies inside a collective must end the job with an error, not leave the others waiting for ever
        torch.distributed.init_process_group(backend, timeout=datetime.timedelta(minutes=int(os.environ.get("SMO_BENCH_PG_TIMEOUT_MIN", "15"))))
    wl = a.workload or "kdyn"
    scaling = "weak"
    watchdog = None
    if world > 1:
        # A collective that never completes (a rank lost inside an RCCL call, a fabric problem) must end this job with an error, not sit
        # on the node until an outer limit kills it: torch's process-group timeout covers torch's collectives, this timer covers the rest.
        import threading
        limit = 60.0 * float(os.environ.get("SMO_BENCH_WATCHDOG_MIN", "25"))

        def _give_up():
            sys.stderr.write("rank %d: bench.py watchdog: no result after %.0f minutes, giving up\n" % (rank, limit / 60.0))
            sys.stderr.flush()
            os._exit(124)
        watchdog = threading.Timer(limit, _give_up)
        watchdog.daemon = True
        watchdog.start()
    n_gpus_line = world
    if a.devices:
        if world > 1 or wl != "kdyn":
            raise SystemExit("bench.py: --devices is the single-process form of the kdyn workload (no launcher, no --gpus > 1)")
        devs = [int(d) for d in a.devices.split(",")]
        steps, warm, el, per_step_units, roof, cfg, cpu = bench_kdyn_multi(a, torch, devs)
        scaling, n_gpus_line = "strong", len(set(devs))
        a.no_secondary = True
    elif wl == "sh23":
        steps, warm, el, per_step_units, roof, cfg, cpu = bench_sh23(a, torch, rank, world)
    elif wl == "shb23":
        steps, warm, el, per_step_units, roof, cfg, cpu = bench_shb23(a, torch, rank, world)
    elif wl == "pois":
        steps, warm, el, per_step_units, roof, cfg, cpu = bench_pois(a, torch, rank, world)
    elif wl == "kdyn" and world > 1 and not a.replicas:
        cdev = "cpu" if torch.distributed.get_backend() == "gloo" else "cuda"
        slab_err = None
        try:
            steps, warm, el, per_step_units, roof, cfg, cpu, scaling = bench_kdyn_slab(a, torch, rank, world)
            if not cfg["slab_J_matches_single_gpu"]:
                raise RuntimeError("slab-decomposed J %r differs from the single-GPU J %r" % (cfg["J"], cfg["J_single_gpu"]))
            per_step_units = 1.0 / world          # ONE gradient is shared by all ranks (value = steps / time)
            if not a.no_secondary:
                # the other way to use N GPUs (line-search trial points, Taylor-test perturbations, ensembles): one independent gradient
                # per rank, no exchange at all.  Reported next to the slab figure, never as `value`.  bench_kdyn runs with world=1 here (no
                # barrier inside), so a rank that fails still reaches the all_reduce below.
                el2, st2, info, fail = 0.0, 2, {}, 0.0
                try:
                    torch.cuda.empty_cache()
                    b = argparse.Namespace(**{**vars(a), "steps": 2, "warmup": 1, "no_cpu_baseline": True, "no_host_vectors": True})
                    torch.distributed.barrier()
                    st2, _, el2, _, _, info, _ = bench_kdyn(b, torch, rank, 1)
                except Exception as e2:
                    fail, info = 1.0, {"error": repr(e2)}
                t2 = torch.tensor([el2, fail], device=cdev, dtype=torch.float64)
                torch.distributed.all_reduce(t2, op=torch.distributed.ReduceOp.MAX)
                if float(t2[1].item()) > 0:
                    cfg["independent_gradients"] = {"error": info.get("error", "failed on another rank")}
                else:
                    cfg["independent_gradients"] = {"value": st2 * world / float(t2[0].item()), "unit": "gradient evals/s", "scaling": "weak",
                                                    "ms_per_gradient_per_gpu": 1e3 * float(t2[0].item()) / st2, "steps": st2, "warmup": 1,
                                                    "checkpoint_interval": info["checkpoint_interval"], "y_side_stack_GB": info["y_side_stack_GB"]}
        except Exception as e:
            slab_err = e
        # every rank takes the same branch: the slab result stands only if it succeeded everywhere
        flag = torch.tensor([0.0 if slab_err is None else 1.0], dtype=torch.float64, device=cdev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if float(flag.item()) > 0:
            # A scaling driver must never record N independent gradients as if they were one decomposed solve: a failed slab path ends the
            # job with a non-zero exit code on every rank (no JSON line) unless --allow-replica-fallback asks for the replica line
            sys.stderr.write("rank %d: slab path failed (%r)%s\n" % (rank, slab_err or "on another rank",
                             "; every rank falls back to independent replicas" if a.allow_replica_fallback else "; exiting (no fallback without --allow-replica-fallback)"))
            if not a.allow_replica_fallback:
                if watchdog is not None:
                    watchdog.cancel()
                torch.distributed.destroy_process_group()
                raise SystemExit(3)
            torch.cuda.empty_cache()
            steps, warm, el, per_step_units, roof, cfg, cpu = bench_kdyn(a, torch, rank, world)
            cfg["slab_path_error"] = repr(slab_err) if slab_err is not None else "failed on another rank"
            scaling = "weak"
    elif wl == "kdyn":
        steps, warm, el, per_step_units, roof, cfg, cpu = bench_kdyn(a, torch, rank, world)
    else:
        raise SystemExit("workload %s not built yet" % wl)
    secondary = None
    if world == 1 and a.workload is None and not a.no_secondary:
        # the two latency-bound 1-D configs of BASELINE.json (configs[1], configs[2]) ride along as secondary lines
        secondary = []
        for fn, kw in ((bench_sh23, dict(steps=50, warmup=5)), (bench_shb23, dict(steps=10, warmup=2)), (bench_pois, dict(steps=2, warmup=1))):
            b = argparse.Namespace(**{**vars(a), "npts": None, "iters": None, "batch": 1, "no_cpu_baseline": True, **kw})
            st, wm, e2, units, rf, cf, _ = fn(b, torch, rank, world)
            secondary.append({"workload": cf["workload"], "value": st * units / e2, "unit": "gradient evals/s", "ms_per_step": 1e3 * e2 / st,
                              "us_per_time_step": (1e6 * e2 / st / (2 * cf["n_iters"])) if fn is bench_pois else rf["avg_launch_ms"] * 1e3 / cf["n_iters"],
                              "steps": st, "warmup": wm, **{k: cf[k] for k in ("J", "J_matches_oracle_1e-6") if k in cf}})
        # BASELINE configs[4]'s grid on this ONE GPU (windowed checkpoints): the 1-GPU end of the 256^3 strong-scaling series whose
        # N-GPU points the N > 1 runs report under the same key
        if wl == "kdyn" and a.npts is None and a.iters is None:
            try:
                torch.cuda.empty_cache()
                # warm-up 1, two timed gradients through the host-buffer entry points like the main line, its own CPU leg (a 2-step sample of
                # the oracle at 256^3, 1 thread and all cores): north_star's "x the CPU baseline on the 256^3 gradient at 1 MI355X" as a number
                b = argparse.Namespace(**{**vars(a), "npts": 256, "steps": int(os.environ.get("SMO_BENCH_256_STEPS", "2")), "warmup": 1,
                                          "no_cpu_baseline": a.no_cpu_baseline, "cpu_sample_steps": 2})
                st, wm, e2, _, rf, cf, cpu2 = bench_kdyn(b, torch, rank, world)
                c256 = {"workload": cf["workload"] + " on 1 GPU", "ms_per_gradient": 1e3 * e2 / st, "gradient_evals_per_s": st / e2,
                        "steps": st, "warmup": wm, "J": cf["J"], "stack_GB_per_gpu": cf["stack_GB"], "vectors": cf["vectors"],
                        "checkpoint_interval": cf["checkpoint_interval"], "value_device_vectors": cf.get("value_device_vectors"),
                        # the same roofline accounting as the main line, for the dominant kernel of the G = 384 instantiations
                        "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                             "avg_launch_ms", "avg_launch_ms_sampled", "avg_launch_ms_every_launch", "frac_every_launch",
                                                             "launches_timed", "timing_stride", "bytes_per_launch", "achieved_algorithmic",
                                                             "whole_gradient_GBps", "whole_gradient_frac", "inner_product", "all_kernels")}}
                if cpu2:
                    c256["cpu_baseline"] = cpu2
                    c256["cpu_all_cores"] = cf.get("cpu_all_cores")
                    c256["speedup_vs_cpu_1_core"] = c256["gradient_evals_per_s"] / cpu2["value"]
                    if cf.get("cpu_all_cores"):
                        c256["speedup_vs_cpu_all_cores"] = c256["gradient_evals_per_s"] / cf["cpu_all_cores"]["value"]
                        # north_star's sentence as a number: this gradient against the same restatement on one socket's physical cores
                        # (a reported baseline, never the target)
                        c256["speedup_vs_cpu_single_socket"] = c256["speedup_vs_cpu_all_cores"]
                cfg["config_256"] = c256
            except Exception as e:                   # never lose the main line because of the extra
                cfg["config_256"] = {"error": repr(e)}
    if world > 1:
        t = torch.tensor([el], device="cpu" if torch.distributed.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        scaling = scaling if "slab" in cfg.get("parallelism", "") else "weak"
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        total = steps * per_step_units * world
        out = {"metric": "forward+adjoint gradient evals/sec", "value": total / el, "unit": "gradient evals/s",
               "n_gpus": n_gpus_line, "rccl_ranks": (torch.distributed.get_world_size() if (world > 1 and torch.distributed.get_backend() == "nccl") else 0),      # 0: no RCCL communicator exists in this run
               "backend": (torch.distributed.get_backend() if world > 1 else None),
               "steps": steps, "warmup": warm, "ms_per_step": 1e3 * el / steps, "higher_is_better": True,
               "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": cfg,
               "roofline": roof, "cpu_baseline": cpu}
        if secondary:
            out["secondary"] = secondary
        print(json.dumps(out))
    if watchdog is not None:
        watchdog.cancel()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""bench.py keeps the driver's contract: ONE JSON line with the required keys (run here at a reduced size so that it takes seconds)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _run(args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_kdyn_line_contract():
    d = _run(["--npts", "32", "--iters", "20", "--steps", "2", "--warmup", "1", "--no-secondary"])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "achieved_algorithmic", "bytes_per_launch", "traffic_source"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # frac is taken against the compulsory bytes of the fused kernel: a fraction of the peak, never above it; the unfused count rides along
    assert 0 < r["frac"] <= 1.0 and r["achieved_algorithmic"] >= r["achieved"]
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # the dominant class carries HIP events on every 8th of its launches inside the timed region (a uniform sample)
    assert r["timing_stride"] == 8 and 1 <= r["launches_timed"] <= (2 * 2 * 22 + 7) // 8      # a class is launched at most ~2x per time step
    # frac is taken from the sample inside the timed region; the warm-up gradient's every-launch average rides along
    assert r["avg_launch_ms"] == r["avg_launch_ms_sampled"] and r["avg_launch_ms_every_launch"] > 0 and "avg_launch_ms_sampled" in r["frac_basis"]
    assert abs(r["frac_every_launch"] - r["bytes_per_launch"] / (r["avg_launch_ms_every_launch"] * 1e-3) / 1e9 / r["peak"]) < 1e-9
    # Inner_Prod_3 is part of SURVEY 8d's measurement: 2 x vector bytes per call, timed with HIP events, checked against NumPy's value
    ip = r["inner_product"]
    assert "error" not in ip and ip["bytes_per_call"] == 2 * 8 * 3 * 48 ** 3 and ip["launches_timed"] == 20 and 0 < ip["frac"] <= 1.0 and ip["wall_ms_per_call"] > 0
    # no PMC summary of a 32^3 run is committed: traffic must be null with the reason, not a number from another build / size
    assert r["traffic"] is None and "reason" in r["traffic_source"]
    # `value` is SURVEY 8d's metric: the timed steps hand over HOST vectors (H2D of X and D2H of grad J inside the timed region); the
    # device-resident rate rides along, same J and the same gradient bit for bit
    assert d["config"]["vectors"].startswith("host (pinned)")
    dv = d["config"]["value_device_vectors"]
    assert dv["value"] > 0 and dv["J_equal"] is True and dv["grad_equal"] is True and dv["value"] >= 0.7 * d["value"]
    assert d["rccl_ranks"] == 0 and d["backend"] is None             # no RCCL communicator exists in a one-rank run
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    # the all-core leg is pinned to ONE socket's physical cores and says which CPU that is
    ac = d["config"]["cpu_single_socket"]
    assert ac["cores"] >= 1 and ac["value"] > 0 and ac["scaling_over_1_core"] > 0 and "model" in ac["cpu"] and ac["cpu"]["sockets"] is not None
    assert ac["cores"] <= (ac["cpu"]["physical_cores_per_socket"] or ac["cores"])


def test_single_process_multi_device_line():
    """`--devices 0,0`: the kdyn gradient through ONE multi-device context (smo_create_multi) in this one process, no launcher."""
    d = _run(["--devices", "0,0", "--npts", "32", "--iters", "20", "--steps", "2", "--warmup", "1"])
    c = d["config"]
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["rccl_ranks"] == 0 and c["devices"] == [0, 0]
    assert "ONE process" in c["parallelism"] and c["chunks"] >= 1
    assert c["compute_ms_per_step_pair"] > 0 and c["exchange_ms_per_step_pair"] > 0 and c["wall_ms_per_step_pair"] > 0
    one = _run(["--npts", "32", "--iters", "20", "--steps", "1", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"])
    assert abs(c["J"] - one["config"]["J"]) <= 1e-12 * abs(one["config"]["J"])
    # the transport in use is named (ranks sharing a device: the gather kernel) and the host side of the loop is measured
    assert "gather kernel" in c["transpose_pull"] and c["host_bound_loop"]["workers"] == 2 and c["host_issue_ms_per_step_pair"] > 0
    assert 3.9 <= c["host_bound_loop"]["rendezvous_per_step_pair"] <= 5.5      # 2 exchanges per forward step + 2 per adjoint step, ONE rendezvous each inside the time loop (chained), two around the transforms of X / the gradients


def test_single_process_line_at_the_north_star_decomposition(fields384):
    """`--devices 0 x 8 --npts 256 --iters 20`: config 5's decomposition (16 kx planes / 48 z planes per rank, G = 384 kernels, default chunks)
    through the benchmark's own single-process path, eight persistent workers."""
    d = _run(["--devices", "0,0,0,0,0,0,0,0", "--npts", "256", "--iters", "20", "--steps", "1", "--warmup", "1"])
    c = d["config"]
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and c["devices"] == [0] * 8 and c["chunks"] == 1 and c["grid"] == [384, 384, 384]
    assert np.isfinite(c["J"]) and c["compute_ms_per_step_pair"] > 0 and c["exchange_ms_per_step_pair"] > 0
    hi = c["host_bound_loop"]
    assert hi["workers"] == 8 and hi["chunks"] == 1 and 0 < c["host_issue_ms_per_step_pair"] < 5.0 and hi["issue_ms_per_step_pair"] > 0
    from spheremanopt_amd import _capi
    one = _capi.Context(_capi.SMO_KDYN, 256, (0., 2. * np.pi), 1e-3, 20, 1.0)           # the same 20 steps on the plain single-GPU context
    J1 = one.forward(list(fields384))
    one.close()
    assert abs(c["J"] - J1) <= 1e-12 * abs(J1)


@pytest.mark.parametrize("wl", ["sh23", "shb23", "pois"])
def test_other_workloads_emit_one_line(wl):
    d = _run(["--workload", wl, "--steps", "1", "--warmup", "1", "--iters", "40", "--no-cpu-baseline"])
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and wl[:2].lower() in d["config"]["workload"].lower().replace("swift-hohenberg", "sh").replace("plane-poiseuille", "po")


def _launch(args, extra_env=None, launcher_ranks=None, port=29741):
    env = dict(os.environ, PYTHONPATH=ROOT, SMO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable]
    if launcher_ranks:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher_ranks), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    p = subprocess.run(cmd + [os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    return p, lines


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher (the shape of the driver's command line): bench.py must start two ranks itself —
    before it touches the GPU — and report n_gpus = 2 (here the two ranks share the box's one GPU and exchange through gloo)."""
    p, lines = _launch(["--gpus", "2", "--npts", "32", "--iters", "20", "--steps", "2", "--warmup", "1", "--no-secondary"])
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["backend"] == "gloo" and d["rccl_ranks"] == 0
    assert d["config"]["slab_J_matches_single_gpu"] and "slab_path_error" not in d["config"]
    # the line says where a step pair's time goes: kernels, transposes (events around every exchange call), wall
    c = d["config"]
    for k in ("compute_ms_per_step_pair", "exchange_ms_per_step_pair", "exchange_calls_per_step_pair", "wall_ms_per_step_pair", "rccl_library"):
        assert k in c, k
    assert c["compute_ms_per_step_pair"] > 0 and c["exchange_ms_per_step_pair"] > 0 and 4.0 <= c["exchange_calls_per_step_pair"] <= 4.5      # 4 per step pair + the transforms of X and of the gradients (20 steps here)
    assert c["wall_ms_per_step_pair"] >= 0.5 * c["compute_ms_per_step_pair"]
    assert c["rccl_library"] is None                                 # callback transport: no RCCL communicator, none claimed


def test_world_size_other_than_gpus_is_refused():
    p, lines = _launch(["--gpus", "4", "--npts", "32", "--iters", "20", "--steps", "1", "--warmup", "0", "--no-secondary"], launcher_ranks=2, port=29743)
    assert p.returncode != 0 and not lines
    assert "--gpus 4" in p.stderr and "2 rank" in p.stderr


def test_failed_slab_path_is_loud():
    """A rank that cannot build its slab solver (here: injected on rank 1) must not leave the other in a collective — and the job must
    not print a line that looks like a perfect N x: every rank agrees on the failure and exits non-zero, no JSON line."""
    p, lines = _launch(["--gpus", "2", "--npts", "32", "--iters", "20", "--steps", "1", "--warmup", "1", "--no-secondary"],
                       extra_env={"SMO_BENCH_INJECT_FAILURE": "1", "SMO_BENCH_PG_TIMEOUT_MIN": "3"})
    assert p.returncode != 0 and not lines, (p.returncode, lines)
    assert "slab path failed" in p.stderr and "--allow-replica-fallback" in p.stderr


def test_failure_on_one_rank_is_agreed_on_by_all():
    """... unless --allow-replica-fallback asks for it: then every rank falls back to independent replicas together and the contract's one
    JSON line appears, flagged (scaling weak, config.slab_path_error)."""
    p, lines = _launch(["--gpus", "2", "--npts", "32", "--iters", "20", "--steps", "1", "--warmup", "1", "--no-secondary", "--allow-replica-fallback"],
                       extra_env={"SMO_BENCH_INJECT_FAILURE": "1", "SMO_BENCH_PG_TIMEOUT_MIN": "3"}, port=29745)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "slab_path_error" in d["config"] and "replicas" in d["config"]["parallelism"]


def test_second_slab_path_when_the_library_communicator_fails():
    """Should libsmo's own communicator not come up (here: injected on every rank), the slab decomposition still runs — Python loop over
    the phase-level entry with torch.distributed's all-to-all — before anything falls back to replicas."""
    p, lines = _launch(["--gpus", "2", "--npts", "32", "--iters", "20", "--steps", "1", "--warmup", "1", "--no-secondary"],
                       extra_env={"SMO_BENCH_INJECT_LIB_FAILURE": "1"}, port=29747)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "Python loop" in d["config"]["transport"]
    assert d["config"]["slab_J_matches_single_gpu"] and "slab_path_error" not in d["config"]


def test_two_rank_line_reports_slab_and_independent_gradients():
    """The N>1 launch of the contract (torch.distributed.run, one rank per process) on the one GPU of the test box: ranks share cuda:0
    and exchange through gloo.  One JSON line from rank 0, strong scaling, slab J equal to the single-GPU J, plus the exchange-free figure."""
    env = dict(os.environ, PYTHONPATH=ROOT, SMO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--npts", "32", "--iters", "20", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "slab" in d["config"]["parallelism"], d["config"]
    assert d["config"]["slab_J_matches_single_gpu"] and "slab_path_error" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]              # ONE gradient shared by the ranks
    ig = d["config"]["independent_gradients"]
    assert "error" not in ig and ig["scaling"] == "weak" and ig["value"] > 0

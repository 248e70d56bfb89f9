"""bench.py logic that needs no GPU: `--gpus N` starts the ranks itself before anything touches a GPU, a world size other than N is
refused, and PMC traffic is quoted only from a counter summary collected from exactly the kernel sources of this tree."""
import importlib.util
import json
import os
import subprocess
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_flag_spawns_ranks_without_touching_a_gpu():
    """Here there is no GPU: both child ranks must come up (torch.distributed.run) and each refuse with the no-GPU message; the parent
    only forwards their exit status.  A parent that ignored --gpus would print ONE such message, from itself."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--npts", "16", "--iters", "2"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert p.stderr.count("bench.py needs a GPU") == 2, p.stderr[-2000:]
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_refused_before_any_gpu_work():
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode != 0 and "--gpus 8" in p.stderr and "2 rank" in p.stderr


def test_pmc_traffic_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    b = _bench()
    sha = b.source_sha()
    assert len(sha) == 16 and sha == b.source_sha()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    monkeypatch.setattr(b, "source_sha", lambda: sha)
    val, src = b.pmc_traffic("kd_x_pass<fused_adj>", 128)
    assert val is None and "reason" in src
    rec = {"source_sha": "0" * 16, "kernels": {"kd_x_pass<fused_adj>": {"hbm_bytes_per_launch": 1.0, "instantiation": "x"}}}
    (prof / "r07_kdyn128_pmc.json").write_text(json.dumps(rec))
    assert b.pmc_traffic("kd_x_pass<fused_adj>", 128)[0] is None              # another build's counters are not quoted
    rec["source_sha"] = sha
    rec["kernels"]["kd_x_pass<fused_adj>"]["hbm_bytes_per_launch"] = 7.5e8
    (prof / "r08_kdyn128_pmc.json").write_text(json.dumps(rec))
    val, src = b.pmc_traffic("kd_x_pass<fused_adj>", 128)
    assert val == 7.5e8 and src["file"].endswith("r08_kdyn128_pmc.json") and src["source_sha"] == sha
    assert b.pmc_traffic("kd_x_pass<fused_adj>", 256)[0] is None               # other grid: other file
    assert b.pmc_traffic("kd_y_pass<inv>", 128)[0] is None                     # kernel not in the summary


def test_committed_pmc_summaries_match_this_tree():
    """The summaries under profiles/ that bench.py would quote must have been collected from the kernel sources as committed — rerun
    tools/profile_round.sh (and copy its pmc_<N>.json) after changing anything under spheremanopt_amd/csrc/."""
    import warnings
    b = _bench()
    for n in (128, 256):
        val, src = b.pmc_traffic("kd_x_pass<fused_adj>", n)
        if val is None:                       # not an error of the code: bench.py then reports traffic = null, as it must
            warnings.warn("profiles/ holds no PMC summary of the current kernel sources for %d^3: %s" % (n, src["reason"]))
        else:
            assert val > 0 and src["source_sha"] == b.source_sha()

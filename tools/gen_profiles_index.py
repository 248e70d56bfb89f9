#!/usr/bin/env python3
"""profiles/README.md, newest round: the rows are GENERATED from the files they index (round 3's hand-written rows quoted
155.3 us and a source_sha the files beside them did not hold).

    python tools/gen_profiles_index.py r04            rewrite the block between <!-- r04:begin --> and <!-- r04:end -->
    python tools/gen_profiles_index.py r04 --check    exit 1 if the committed block differs from what the files say (tests/test_profiles_index.py)

Every number in a generated row is read from the CSV / JSON it describes; the free-text files (r04_*.txt) are indexed by their first line."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
P = os.path.join(ROOT, "profiles")


def short(name):
    name = name.replace("void ", "").replace("smo::(anonymous namespace)::", "").replace("smo::", "")
    return re.sub(r"\(.*", "", name).strip()


def sq_row(path):
    """tools/profile_sq.sh summary: the dominant kernel's wave-cycle split and LDS conflict share."""
    txt = open(path).read()
    out = []
    for kern in re.findall(r"^(kd_x_pass<\d+, 4, [^>]*>|kd_z_forward<\d+, 1, 1, [^>]*>)\s*$", txt, flags=re.M):
        blk = txt[txt.index(kern):].split("\nkd_", 1)[0]
        c = {m.group(1): float(m.group(2)) for m in re.finditer(r"(SQ_\w+)\s+avg ([0-9.e+]+)", blk)}
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
            s = "`%s`: parked %.0f %%, issue-stalled %.0f %%, issuing %.0f %% of the wave cycles" % (
                kern, 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                100 * c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"])
            if c.get("SQ_LDS_IDX_ACTIVE"):
                s += ", LDS bank conflicts %.1f %% of the LDS-active cycles" % (100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"])
            out.append(s)
    return "SQ counters per kernel (`tools/profile_sq.sh`, six `--pmc` passes): " + "; ".join(out)


def stats_row(path, top=8):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    parts = ["`%s` %.1f µs × %d (%.1f %%)" % (short(r["Name"]), float(r["AverageNs"]) / 1e3, int(r["Calls"]), 100 * float(r["TotalDurationNs"]) / tot)
             for r in rows[:top] if "kd_" in r["Name"] or "pois" in r["Name"] or "sh" in r["Name"]]
    return "; ".join(parts)


def bench_row(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    if "roofline" not in d:                      # a small timing record, not a bench line
        return "; ".join("%s %s" % (k, ("%.4g" % v) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (dict, list)))[:300]
    r, c = d["roofline"], d["config"]
    s = "`value` %.4f %s (%.1f ms per step, %d steps)" % (d["value"], d["unit"], d["ms_per_step"], d["steps"])
    if "value_device_vectors" in c:
        s += "; device-resident %.1f ms" % c["value_device_vectors"]["ms_per_step"]
    s += "; `roofline` of `%s`: %.1f µs sampled in the timed region" % (r["kernel"], 1e3 * r["avg_launch_ms"])
    if "avg_launch_ms_every_launch" in r:
        s += " / %.1f µs with every launch instrumented" % (1e3 * r["avg_launch_ms_every_launch"])
    s += ", `frac` %.4f" % r["frac"]
    if r.get("frac_every_launch"):
        s += " (%.4f)" % r["frac_every_launch"]
    if r.get("traffic"):
        s += ", `traffic` %.1f MB from `%s`" % (r["traffic"] / 1e6, r["traffic_source"].get("file"))
    if r.get("whole_gradient_frac"):
        s += "; whole gradient %.3f of the peak" % r["whole_gradient_frac"]
    ip = r.get("inner_product")
    if ip and "error" not in ip:
        s += "; inner product %.1f µs, %.0f GB/s (%.3f)" % (1e3 * ip["avg_launch_ms"], ip["GBps"], ip["frac"])
    if d.get("cpu_baseline"):
        s += "; `cpu_baseline` %.3g/s on %d core" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
    ss = c.get("cpu_single_socket")
    if ss:
        s += ", %.3g/s on %d cores of one socket (%s; cgroup quota %s)" % (ss["value"], ss["cores"], ss["cpu"]["model"], ss["cpu"]["cgroup_cpu_quota"])
    c2 = c.get("config_256")
    if c2 and "error" not in c2:
        s += "; `config_256`: %.1f ms per gradient" % c2["ms_per_gradient"]
        if c2.get("roofline"):
            s += ", `%s` %.1f µs, frac %.4f" % (c2["roofline"]["kernel"], 1e3 * c2["roofline"]["avg_launch_ms"], c2["roofline"]["frac"])
        if c2.get("speedup_vs_cpu_single_socket"):
            s += ", × %.0f the single-socket CPU leg" % c2["speedup_vs_cpu_single_socket"]
    for k in ("host_issue_ms_per_step_pair", "compute_ms_per_step_pair", "exchange_ms_per_step_pair", "wall_ms_per_step_pair"):
        if c.get(k) is not None:
            s += "; %s %.3f" % (k, c[k])
    return s


def pmc_row(path):
    d = json.load(open(path))
    ks = d["kernels"]
    keys = [k for k in ("kd_y_pass<inv>", "kd_x_pass<fused_fwd>", "kd_x_pass<fused_adj>", "kd_z_forward<fwd_update>") if k in ks]
    return "`source_sha` %s; HBM bytes per launch (2·FETCH_SIZE + WRITE_SIZE)·1024: %s" % (
        d["source_sha"], ", ".join("%s %.1f MB" % (k.replace("kd_", ""), ks[k]["hbm_bytes_per_launch"] / 1e6) for k in keys))


def rows(rnd):
    out = []
    for f in sorted(glob.glob(os.path.join(P, rnd + "_*"))):
        b = os.path.basename(f)
        try:
            if b.endswith("kernel_stats.csv"):
                desc = "rocprofv3 --kernel-trace --stats: " + stats_row(f)
            elif b.endswith("_pmc.json"):
                desc = pmc_row(f)
            elif b.endswith(".json"):
                desc = bench_row(f)
            elif b.endswith("_sq_counters.txt"):
                desc = sq_row(f)
            elif b.endswith("_pmc_summary.txt"):
                desc = "per-kernel averages of the FETCH_SIZE / WRITE_SIZE passes (`tools/summarize_pmc.py`): the table `..._pmc.json` is reduced from"
            elif b.endswith("pytest_gpu.log"):
                desc = "`python -m pytest tests -m gpu -q`: " + [ln.strip() for ln in open(f) if ln.strip()][-1]
            elif b.endswith(".jsonl"):
                desc = "%d JSON lines" % sum(1 for ln in open(f) if ln.strip())
            else:
                first = [ln.strip() for ln in open(f, errors="replace") if ln.strip()]
                desc = first[0][:260] if first else ""
        except Exception as e:        # a file this script cannot read is still listed
            desc = "(not summarised: %s)" % type(e).__name__
        out.append("| `%s` | %s |" % (b, desc.replace("|", "\\|")))
    return out


def block(rnd):
    return "\n".join(["<!-- %s:begin (generated by tools/gen_profiles_index.py %s; do not edit by hand) -->" % (rnd, rnd),
                      "| file | what it holds (numbers read from the file itself) |", "|---|---|"] + rows(rnd) + ["<!-- %s:end -->" % rnd])


def main():
    rnd = sys.argv[1]
    readme = os.path.join(P, "README.md")
    txt = open(readme).read()
    new = block(rnd)
    m = re.search(r"<!-- %s:begin.*?<!-- %s:end -->" % (rnd, rnd), txt, flags=re.S)
    if "--check" in sys.argv:
        if not m or m.group(0) != new:
            sys.stderr.write("profiles/README.md: the %s block is stale; run tools/gen_profiles_index.py %s\n" % (rnd, rnd))
            sys.exit(1)
        return
    if m:
        txt = txt[:m.start()] + new + txt[m.end():]
    else:
        head, rest = txt.split("\n", 2)[0], txt.split("\n", 2)[2] if txt.count("\n") >= 2 else ""
        txt = head + "\n\n## Round %s (`%s_*`)\n\n" % (rnd[1:].lstrip("0"), rnd) + new + "\n\n" + rest
    open(readme, "w").write(txt)


if __name__ == "__main__":
    main()

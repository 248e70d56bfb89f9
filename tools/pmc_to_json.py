#!/usr/bin/env python3
"""rocprofv3 --pmc csv directories -> profiles/<name>.json keyed by the library's timing-class names.
usage: python tools/pmc_to_json.py <pmc_dir> <out.json> [grid]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

XMODE = {0: "kd_misc(x to grid)", 1: "kd_misc(x from grid)", 2: "kd_x_pass<fused_fwd>", 3: "kd_x_pass<fused_adj>",
         4: "kd_x_pass<fused_adj>"}      # 4 = X_FUSED_ADJ_SEQ: the field groups one after the other (the default adjoint pass since round 2)
ZI = {0: "kd_z_inverse", 1: "kd_z_inverse<curl>", 2: "kd_misc(z inverse scaled)"}
ZF = {0: "kd_misc(z forward plain)", 1: "kd_z_forward<fwd_update>", 2: "kd_z_forward<adj_update>", 3: "kd_misc(z forward nu)"}


def classify(name):
    m = re.search(r"(kd_\w+)(?:<([^>]*)>)?", name)
    if not m:
        return None
    base, args = m.group(1), [a.strip() for a in (m.group(2) or "").split(",")]
    if base == "kd_x_pass":
        return XMODE[int(args[1])]
    if base == "kd_z_inverse":
        return ZI[int(args[1])]
    if base == "kd_z_forward":          # template arguments: L, MODE, NEXT, ...; the last step of a solve runs without the fused next pass
        mode, nxt = int(args[1]), int(args[2])
        return ZF[mode] if (mode in (0, 3) or nxt != 0) else "kd_misc(z forward update, last step)"
    if base == "kd_y_pass":
        return "kd_y_pass<inv>" if args[1] == "true" else "kd_y_pass<fwd>"
    return base


def source_sha():
    """Identity of the kernel sources the counters were collected from (same function as bench.py's: kdyn.hip and the headers it includes):
    bench.py quotes `traffic` only from a summary whose source_sha equals the one of the tree it runs in."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spheremanopt_amd", "csrc")
    h = hashlib.sha256()
    for f in ("kdyn.hip", "kdyn_any.hpp", "fft_lds.hpp", "comm.hpp", "smo_common.hpp"):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    src, out = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    shapes = {}
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = classify(row.get("Kernel_Name", ""))
            if k is None:
                continue
            c = acc[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
            shapes[k] = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void smo::(anonymous namespace)::", ""))[:60]
    res = {}
    for k, cs in acc.items():
        v = {n: t / c for n, (t, c) in cs.items()}
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        res[k] = {"instantiation": shapes[k], "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
                  "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024}
        if "SQ_LDS_BANK_CONFLICT" in v and "SQ_LDS_IDX_ACTIVE" in v:          # only when the SQ passes were collected too
            res[k]["lds_conflict_fraction"] = v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)
        if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v:
            res[k]["wait_any_fraction"] = v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1)
    json.dump({"note": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), averages per dispatch; hbm_bytes = "
                       "(2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE tallies 128-B "
                       "requests as 64 B; narrower accesses are uncalibrated)", "grid": sys.argv[3] if len(sys.argv) > 3 else "", "source_sha": source_sha(),
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items()):
        print("%-34s %8.1f MB%s" % (k, v["hbm_bytes_per_launch"] / 1e6,
                                    ("  lds-conflict %.2f" % v["lds_conflict_fraction"]) if "lds_conflict_fraction" in v else ""))


if __name__ == "__main__":
    main()

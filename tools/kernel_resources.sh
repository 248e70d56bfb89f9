#!/bin/bash
# VGPRs / spills / occupancy / LDS of the KDyn kernels at the bench grids (compile-time report of hipcc; no GPU needed).
#   tools/kernel_resources.sh [extra compiler flags]        e.g.  tools/kernel_resources.sh -DSMO_FFT_MAX_RADIX=8
cd "$(dirname "$0")/../spheremanopt_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -Rpass-analysis=kernel-resource-usage -c kdyn.hip -o /tmp/kdyn_res.o 2>&1 | python3 -c '
import re, sys
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name.replace("void smo::(anonymous namespace)::", ""))
    if not re.search(r"<(%s)," % (__import__("os").environ.get("SMO_RES_G", "192|384")), name):
        continue
    print("%-44s VGPR %3d  spill %d  scratch %d  occupancy %d  LDS %6d" % (name, v.get("VGPRs", -1), v.get("VGPRs Spill", -1),
          v.get("ScratchSize", -1), v.get("Occupancy", -1), v.get("LDS Size", -1)))
'

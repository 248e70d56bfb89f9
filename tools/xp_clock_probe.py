#!/usr/bin/env python3
"""Does the GPU clock differ between an un-instrumented gradient and one whose every launch is timed?  

The dominant kernel reads 3-5 % shorter whenever every launch of every class carries timing events (or under rocprofv3) than when only every
8th launch of its own class does.  Hypothesis: an instrumented queue has idle gaps between kernels, the chip draws less power and clocks
higher.  This probe samples the shader clock and the socket power from sysfs while the same 128^3 gradient runs (a) un-instrumented,
(b) with kernel-stamped timing on every launch, (c) with marker events around every launch, and prints the averages per mode.
usage: python tools/xp_clock_probe.py [npts] [iters] [gradients per mode]"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402


def find_nodes():
    sclk = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    pwr = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
    freq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    return sclk, pwr, freq


def read_sclk(path):
    try:
        for ln in open(path):
            if "*" in ln:
                return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    except (OSError, ValueError, IndexError):
        pass
    return None


def read_num(path, scale):
    try:
        return float(open(path).read().strip()) * scale
    except (OSError, ValueError):
        return None


class Sampler(threading.Thread):
    def __init__(self, nodes):
        super().__init__(daemon=True)
        self.nodes, self.stop_flag, self.rows = nodes, False, []

    def run(self):
        sclk, pwr, freq = self.nodes
        while not self.stop_flag:
            self.rows.append((read_sclk(sclk[0]) if sclk else None, read_num(pwr[0], 1e-6) if pwr else None, read_num(freq[0], 1e-6) if freq else None))
            time.sleep(0.02)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    import torch
    from spheremanopt_amd import kdyn
    nodes = find_nodes()
    print("sysfs nodes:", [x[:1] for x in nodes])
    dom, B, U = kdyn.Generate_IC(N, U_Noise=True)
    ctx = dom.context(1.0, 1e-3, n, "Final")
    Bd, Ud = torch.from_numpy(B).cuda(), torch.from_numpy(U).cuda()
    g = [torch.empty_like(Bd), torch.empty_like(Ud)]
    ctx.forward_dev([Bd, Ud]); ctx.adjoint_dev([Bd, Ud], g)
    for mode, env in (("un-instrumented", None), ("stamped, every launch of every class", "1"), ("marker events, every launch of every class", "0"),
                      ("un-instrumented (again)", None)):
        if env is None:
            ctx.timing_enable(False)
        else:
            os.environ["SMO_TIMING_STAMP"] = env        # (read once per process by the library: the first instrumented mode decides; printed below)
            ctx.timing_enable(True)
        s = Sampler(nodes); s.start()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            ctx.forward_dev([Bd, Ud]); ctx.adjoint_dev([Bd, Ud], g)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        s.stop_flag = True; s.join()
        cols = list(zip(*s.rows)) if s.rows else [[], [], []]
        avg = [float(np.mean([v for v in c if v is not None])) if any(v is not None for v in c) else None for c in cols]
        line = "%-46s %.1f ms/gradient  sclk %s MHz  power %s W  freq1 %s MHz  (%d samples)" % (mode, 1e3 * el / reps, avg[0], avg[1], avg[2], len(s.rows))
        if env is not None:
            k = [t for t in ctx.timing() if t["kernel"] == "kd_x_pass<fused_adj>"][0]
            line += "  x<fused_adj> %.2f us" % (1e3 * k["total_ms"] / max(k["launches"], 1))
        print(line)


if __name__ == "__main__":
    main()

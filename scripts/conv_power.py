#!/usr/bin/env python
"""Power / clock / time attribution of the gate convolution (conv3x3_halo2_kernel) on one MI355X.

For every (operand fill, ablation mask) the kernel runs back to back for --seconds while a sampler thread reads the socket
power and the shader clock (hwmon sysfs; `rocm-smi --json` as fall-back); every launch is timed with its own event pair, so
the file also shows how the launch time moves from the first launch after an idle period (boost clocks) to the sustained
state -- the evidence behind DESIGN.md's "power-limited on random operands".  Ablation masks (only in the -DDH_ABLATION
build, `--ablation`; wrong results by construction): 1 no fragment reads, 2 no weight DMA, 4 no halo fetch, 8 no epilogue,
16 no MFMAs (and their sums).

    python scripts/conv_power.py [--ablation] [--edges 1024] [--seconds 1.5] [--out gpurun_out/conv_power.json]
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--ablation", action="store_true", help="load droid-slam_amd/ablation/ (DROID_HIP_ABLATION=1 python droid-slam_amd/build.py)")
ap.add_argument("--edges", type=int, default=1024)
ap.add_argument("--seconds", type=float, default=1.5)
ap.add_argument("--masks", default="0,1,2,4,6,7,8,15,16,22")
ap.add_argument("--shapes", default="zr,c128")
ap.add_argument("--fills", default="zero,randn")
ap.add_argument("--opt", action="append", default=[], help="library option name=value set before the runs (e.g. conv_halo3=1)")
ap.add_argument("--quick", type=int, default=0, help="N > 0: N launches per case, no idle gaps, no sampler (for rocprofv3 --pmc passes)")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "conv_power.json"))
a = ap.parse_args()
sys.path[:0] = [ROOT] + ([os.path.join(ROOT, "droid-slam_amd", "ablation")] if a.ablation else []) + [os.path.join(ROOT, "droid-slam_amd")]
import torch
import droid_backends as db
from droid_amd.update import pack_conv, pack_conv_halo, EPI_RELU

assert (db.get_option("ablation_build") == 1) == bool(a.ablation), db.__file__
for kv in a.opt:
    k_, v_ = kv.split("=")
    db.set_option(k_, int(v_))


# ---------------------------------------------------------------------------------------------- power / clock sampler
def _hwmon_files():
    power, clock = None, None
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for name in ("power1_average", "power1_input"):
            f = os.path.join(hw, name)
            if power is None and os.path.exists(f):
                try:
                    float(open(f).read()); power = f
                except (OSError, ValueError):
                    pass
        f = os.path.join(hw, "freq1_input")
        if clock is None and os.path.exists(f):
            clock = f
    return power, clock


class Sampler(threading.Thread):
    """socket power / shader clock while the kernels run.  Two sources side by side: `rocm-smi --showpower --showclocks --json`
    (one process per sample, ~5 Hz) and the hwmon files (first session of round 4: the hwmon values did not follow the load at
    100 Hz on the pool's boxes, so they are recorded but not trusted on their own)."""

    def __init__(self):
        super().__init__(daemon=True)
        self.power_f, self.clock_f = _hwmon_files()
        self.source = "rocm-smi --showpower --showclocks --json (+ hwmon %s, %s)" % (self.power_f, self.clock_f)
        self.samples = []                 # (t, smi_watts, smi_sclk_mhz, hwmon_watts, hwmon_mhz)
        self.run_flag = True
        self.raw_example = None

    def _hwmon(self):
        try:
            w = float(open(self.power_f).read()) * 1e-6 if self.power_f else None
            c = float(open(self.clock_f).read()) * 1e-6 if self.clock_f else None
            return w, c
        except (OSError, ValueError):
            return None, None

    def _smi(self):
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(k for k in d if k.startswith("card"))[0]]
            if self.raw_example is None:
                self.raw_example = card
            w = c = None
            for k, v in card.items():
                kl = k.lower()
                if w is None and "power" in kl and "(w)" in kl:
                    try:
                        w = float(v)
                    except ValueError:
                        pass
                if c is None and "sclk" in kl and "level" not in kl.replace("clock level", "level"):
                    digits = "".join(ch if (ch.isdigit() or ch == ".") else " " for ch in str(v)).split()
                    if digits:
                        c = float(digits[-1])
            if c is None:
                for k, v in card.items():
                    if "sclk" in k.lower():
                        digits = "".join(ch if (ch.isdigit() or ch == ".") else " " for ch in str(v)).split()
                        if digits:
                            c = float(digits[-1]); break
            return w, c
        except Exception:
            return None, None

    def run(self):
        while self.run_flag:
            t = time.perf_counter()
            w, c = self._smi()
            hw, hc = self._hwmon()
            self.samples.append((0.5 * (t + time.perf_counter()), w, c, hw, hc))

    def window(self, t0, t1):
        sel = [s_ for s_ in self.samples if t0 <= s_[0] <= t1]
        avg = lambda v: sum(v) / len(v) if v else None
        col = lambda i: [s_[i] for s_ in sel if s_[i] is not None]
        return {"power_W_avg": avg(col(1)), "power_W_max": max(col(1)) if col(1) else None, "sclk_MHz_avg": avg(col(2)), "n_samples": len(col(1)),
                "hwmon_power_W_avg": avg(col(3)), "hwmon_sclk_MHz_avg": avg(col(4))}


sampler = Sampler()
if not a.quick:
    sampler.start()
h, w = 48, 64
SHAPES = {"zr": ("gates z|r 3x3 320->256", (128, 128, 64), 256), "c128": ("3x3 128->128", (128,), 128), "q": ("gate q 3x3 320->128", (128, 128, 64), 128)}
results = {"device": torch.cuda.get_device_name(0), "library": os.path.dirname(db.__file__), "power_source": sampler.source,
           "fills": a.fills, "options": a.opt,
           "edges": a.edges, "seconds_per_case": a.seconds, "cases": []}
masks = [int(m) for m in a.masks.split(",")] if a.ablation else [0]
for key in a.shapes.split(","):
    name, cins, cout = SHAPES[key]
    for fill in a.fills.split(","):
        torch.manual_seed(0)
        mk = torch.zeros if fill == "zero" else torch.randn
        xs = [mk(a.edges, h, w, c, device="cuda").half() for c in cins]
        wgt = mk(cout, sum(cins), 3, 3, device="cuda") / (sum(cins) * 9) ** 0.5
        wp, bp = pack_conv(wgt, torch.zeros(cout, device="cuda"))
        wh = pack_conv_halo(wgt)
        out = torch.empty(a.edges, h, w, cout, device="cuda", dtype=torch.float16)
        run = lambda: db.conv2d_nhwc(xs, wp, wh, bp, 3, 3, cout, EPI_RELU, out, cout, None, None, None, None)
        flops = 2.0 * a.edges * h * w * sum(cins) * 9 * cout
        for m in masks:
            if a.ablation:
                db.set_option("conv_abl", m)
            run(); torch.cuda.synchronize()
            if a.quick:
                for _ in range(a.quick):
                    run()
                torch.cuda.synchronize()
                continue
            time.sleep(0.4)                                   # idle: the first launches below start from boost clocks
            idle = sampler.window(time.perf_counter() - 0.3, time.perf_counter())
            evs = []
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < a.seconds:
                for _ in range(8):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(); e1.record(); evs.append((e0, e1))
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            ms = [x.elapsed_time(y) for x, y in evs]
            tail = sorted(ms[len(ms) // 2:])
            steady = tail[len(tail) // 2]
            win = sampler.window(t0 + 0.5 * (t1 - t0), t1)
            case = {"shape": name, "fill": fill, "abl_mask": m, "launches": len(ms), "ms_first8": [round(v, 4) for v in ms[:8]],
                    "ms_steady_median": steady, "ms_min": min(ms), "TFLOPs_steady": flops / steady / 1e9, "TFLOPs_first": flops / ms[0] / 1e9,
                    "idle_before": idle, "steady": win}
            if win["power_W_avg"]:
                case["pJ_per_flop_steady"] = win["power_W_avg"] * steady * 1e-3 / flops * 1e12
            results["cases"].append(case)
            print("%-24s %-5s abl %2d: first %.3f ms (%.0f TF/s) steady %.3f ms (%.0f TF/s)  %s W  %s MHz  idle %s W" % (
                name, fill, m, ms[0], flops / ms[0] / 1e9, steady, flops / steady / 1e9,
                "%.0f" % win["power_W_avg"] if win["power_W_avg"] else "?", "%.0f" % win["sclk_MHz_avg"] if win["sclk_MHz_avg"] else "?",
                "%.0f" % idle["power_W_avg"] if idle["power_W_avg"] else "?"), flush=True)
        if a.ablation:
            db.set_option("conv_abl", 0)
        del xs, out
sampler.run_flag = False
results["rocm_smi_example"] = sampler.raw_example
if a.quick:
    sys.exit(0)
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(results, open(a.out, "w"), indent=1)
print("wrote", a.out)

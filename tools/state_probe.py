#!/usr/bin/env python3
"""How does a kernel's rate depend on how long the device has been busy?  (the "two kinds of boxes")

    python tools/state_probe.py [--cases diffX,dY,cumZ,cumY] [--launches 600]

For every case: an idle pause, then `--launches` back-to-back launches with a HIP event every 10 launches; printed is the
mean launch time of consecutive windows (fraction of 8 TB/s) -- the first 10 launches after the pause, launches 10-50,
50-100, ... -- next to what the driver's sysfs nodes report while the stream runs (shader / memory / fabric clock levels
marked active, package power), sampled from a thread.  One JSON line per case."""
import argparse
import glob
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from xgcm_amd import device as D  # noqa: E402


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return ""


class Sampler(threading.Thread):
    """clock levels / power from sysfs, every 5 ms"""

    def __init__(self):
        super().__init__(daemon=True)
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.dev = os.path.dirname(cards[0]) if cards else None
        self.hwmon = (sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*"))) or [None])[0] if self.dev else None
        self.samples, self.stop = [], False

    @staticmethod
    def _active(text):
        for ln in text.splitlines():
            if ln.rstrip().endswith("*"):
                return ln.split(":")[1].replace("*", "").strip()
        return None

    def run(self):
        while not self.stop and self.dev:
            s = {"t": time.perf_counter()}
            for key, node in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk"), ("socclk", "pp_dpm_socclk")):
                s[key] = self._active(_read(os.path.join(self.dev, node)))
            if self.hwmon:
                for key, node in (("power_uW", "power1_average"), ("power_in_uW", "power1_input"), ("freq1_Hz", "freq1_input")):
                    v = _read(os.path.join(self.hwmon, node)).strip()
                    if v:
                        s[key] = int(v)
            self.samples.append(s)
            time.sleep(0.005)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="diffX,dY,cumZ,cumY")
    ap.add_argument("--launches", type=int, default=600)
    ap.add_argument("--pause", type=float, default=2.0, help="idle seconds before every case")
    a = ap.parse_args()
    nz, ny, nx = 75, 2400, 3600
    T = D.synthetic((nz, ny, nx), 2)
    dx = D.synthetic((1, ny, nx), 31, 0, 1000.0, 1000.0)
    cases = {
        "diffX": (lambda: D.stencil1d("diff", T, 2, 1, 0, "periodic"), 16),
        "diffY": (lambda: D.stencil1d("diff", T, 1, 1, 0, "extend"), 16),
        "dY": (lambda: D.stencil1d("diff", T, 1, 1, 0, "extend", m_out=dx), 16 + 8 / nz),
        "cumZ": (lambda: D.cumsum1d(T, 0, 0, 1, 1, 0, "fill"), 16),
        "cumY": (lambda: D.cumsum1d(T, 1, 0, 1, 1, 0, "fill"), 16),
        "sumZ": (lambda: D.reduce1d(T, 0, None), 8),
        "copy": (lambda: T.clone(), 16),
    }
    for c in a.cases.split(","):
        fn, bpc = cases[c]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        time.sleep(a.pause)
        smp = Sampler()
        smp.start()
        n = a.launches
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n // 10 + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(n):
            fn()
            if (i + 1) % 10 == 0:
                ev[(i + 1) // 10].record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        smp.stop = True
        smp.join()
        per10 = [ev[i].elapsed_time(ev[i + 1]) / 10 for i in range(n // 10)]
        frac = lambda ms: round(T.numel() * bpc / (ms * 1e-3) / 8e12, 4)  # noqa: E731
        edges = [0, 1, 5, 10, 20, 30, 40, n // 10]
        windows = {f"launch_{10 * lo}_{10 * hi}": frac(sum(per10[lo:hi]) / (hi - lo)) for lo, hi in zip(edges, edges[1:]) if hi <= len(per10) and hi > lo}
        busy = [s for s in smp.samples if t0 <= s["t"] <= t1]
        def col(key):
            vals = [s.get(key) for s in busy if s.get(key) is not None]
            return vals
        out = {"case": c, "launches": n, "busy_s": round(t1 - t0, 3), "frac_8TBps_by_window": windows, "samples": len(busy)}
        for key in ("sclk", "mclk", "fclk", "socclk"):
            v = col(key)
            if v:
                out[key] = {"first": v[0], "last": v[-1], "distinct": sorted(set(v))[:6]}
        for key in ("power_uW", "power_in_uW", "freq1_Hz"):
            v = col(key)
            if v:
                k = max(1, len(v) // 5)
                out[key] = {"first_fifth_mean": round(sum(v[:k]) / k), "last_fifth_mean": round(sum(v[-k:]) / k), "max": max(v)}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

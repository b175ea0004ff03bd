#!/usr/bin/env python3
"""Board power + shader clock during a command, sampled at >= 10 Hz (GPU box):

    python tools/power_sampler.py --out gpurun_out/r03/power_clock.txt [--hz 20] -- python bench.py --steps 40 ...

The command runs with RSR_BENCH_MARKS=<file>: bench.py writes the wall-clock start / end of its timed region there, and the
summary (mean / max W, mean / min sclk) is taken over exactly that window -- the evidence behind DESIGN.md 4.1's "the board
sits at its power cap, ~1.75 GHz of 2.4" -- next to the whole-run numbers.  Sources, first one that answers: the amdsmi Python
binding, the amdgpu hwmon files in sysfs (power1_average / power1_input in uW, freq1_input in Hz), the rocm-smi CLI (~3 Hz)."""
import argparse
import glob
import os
import re
import subprocess
import sys
import threading
import time


def src_amdsmi():
    import amdsmi  # noqa: PLC0415
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]

    def read():
        p = amdsmi.amdsmi_get_power_info(h)
        w = p.get("current_socket_power") or p.get("average_socket_power")
        c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
        return float(w), float(c.get("clk") or c.get("cur_clk"))

    read()
    cap = None
    try:
        cap = float(amdsmi.amdsmi_get_power_cap_info(h).get("power_cap")) / (1e6 if amdsmi.amdsmi_get_power_cap_info(h).get("power_cap", 0) > 1e5 else 1)
    except Exception:  # noqa: BLE001
        pass
    return "amdsmi python binding", read, cap


def src_sysfs():
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        pw = [p for p in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(p)]
        fr = os.path.join(hw, "freq1_input")
        if pw and os.path.exists(fr):
            def read(pw=pw[0], fr=fr):
                return int(open(pw).read()) / 1e6, int(open(fr).read()) / 1e6
            read()
            cap = None
            try:
                cap = int(open(os.path.join(hw, "power1_cap")).read()) / 1e6
            except Exception:  # noqa: BLE001
                pass
            return "sysfs %s (power1_*, freq1_input)" % hw, read, cap
    raise RuntimeError("no amdgpu hwmon with power1_* and freq1_input")


def src_cli():
    def read():
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        m1 = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
        m2 = re.search(r"Power \(W\): ([0-9.]+)", o)
        return float(m2.group(1)), float(m1.group(1))
    read()
    return "rocm-smi CLI", read, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        raise SystemExit("no command")
    name = read = cap = None
    errs = []
    for mk in (src_amdsmi, src_sysfs, src_cli):
        try:
            name, read, cap = mk()
            break
        except Exception as e:  # noqa: BLE001
            errs.append("%s: %r" % (mk.__name__, e))
    if read is None:
        raise SystemExit("no power / clock source: " + "; ".join(errs))
    samples, stop = [], [False]

    def loop():
        dt = 1.0 / a.hz
        while not stop[0]:
            t = time.time()
            try:
                w, f = read()
                samples.append((t, w, f))
            except Exception:  # noqa: BLE001
                pass
            time.sleep(max(0.0, dt - (time.time() - t)))

    marks = a.out + ".marks"
    if os.path.exists(marks):
        os.remove(marks)
    th = threading.Thread(target=loop)
    th.start()
    t_begin = time.time()
    rc = subprocess.call(cmd, env=dict(os.environ, RSR_BENCH_MARKS=marks))
    t_end = time.time()
    stop[0] = True
    th.join()
    win = None
    try:
        t0, t1 = [float(x) for x in open(marks).read().split()[:2]]
        win = (t0, t1)
        os.remove(marks)
    except Exception:  # noqa: BLE001
        pass

    def stats(ss):
        if not ss:
            return "no samples"
        ws, fs = [s[1] for s in ss], [s[2] for s in ss]
        return "n=%d  power mean %.0f W  max %.0f W  min %.0f W | sclk mean %.0f MHz  min %.0f  max %.0f" % (
            len(ss), sum(ws) / len(ws), max(ws), min(ws), sum(fs) / len(fs), min(fs), max(fs))

    with open(a.out, "w") as f:
        f.write("# %s\n# source: %s; %.1f Hz requested, %d samples in %.1f s (%.1f Hz delivered)\n" % (
            " ".join(cmd), name, a.hz, len(samples), t_end - t_begin, len(samples) / max(t_end - t_begin, 1e-9)))
        if cap:
            f.write("power cap reported by the driver: %.0f W\n" % cap)
        f.write("whole command : %s\n" % stats(samples))
        if win:
            inner = [s for s in samples if win[0] <= s[0] <= win[1]]
            f.write("timed region  : %s   (%.3f s, bench.py's own marks)\n" % (stats(inner), win[1] - win[0]))
            idle = [s for s in samples if s[0] < win[0] - 0.5][:40]
            f.write("before it     : %s\n" % stats(idle))
        else:
            f.write("timed region  : no marks written by the command\n")
        f.write("command rc    : %d\n" % rc)
        f.write("# samples: t - t_begin [s], W, MHz (every 4th)\n")
        for s in samples[::4]:
            f.write("%.3f %.0f %.0f\n" % (s[0] - t_begin, s[1], s[2]))
    print(open(a.out).read()[:1500])
    return rc


if __name__ == "__main__":
    sys.exit(main())

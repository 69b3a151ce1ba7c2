"""What the box lets an ordinary user read about the GPU's clocks, power and temperature while a loop runs (sysfs hwmon /
pp_dpm_*; `rocm-smi` as a cross-check).  Question behind it: the headline moves 175-190 it/s between processes
and boxes while the streaming ceilings measured in the same process do not -- is it the core clock?

    python scripts/clock_probe.py
"""
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read(p):
    try:
        return open(p).read().strip()
    except Exception as e:
        return "<%s>" % type(e).__name__


def main():
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        dev = os.path.join(card, "device")
        if not os.path.exists(os.path.join(dev, "vendor")) or read(os.path.join(dev, "vendor")) != "0x1002":
            continue
        print("==", card, os.path.basename(os.path.realpath(dev)))
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "power_dpm_force_performance_level", "gpu_busy_percent"):
            print(f, "->", read(os.path.join(dev, f)).replace("\n", " | "))
        for hw in glob.glob(os.path.join(dev, "hwmon/hwmon*")):
            for f in sorted(os.listdir(hw)):
                if f.startswith(("freq", "power", "temp")) and f.endswith(("input", "average", "label", "cap")):
                    print(" ", f, "->", read(os.path.join(hw, f)))
    try:
        print(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=60).stdout[-3000:])
    except Exception as e:
        print("rocm-smi:", e)

    import numpy as np
    import cuda_mat_amd as cm
    ctx = cm.Context(0)
    n, per = 10_000_000, 50
    rp, ci, va = ctx.empty(n + 1, np.int32), ctx.empty(n * per, np.int32), ctx.empty(n * per)
    ctx.gen_rand_rows(n, per, 7, 0, n, 0, rp, ci, va)
    s = cm.Solver(ctx, n, n, n * per, rp, ci, va, 0)
    x, y = ctx.empty(n), ctx.empty(n)
    ctx.gen_xstar(0, n, 3, x)
    s.spmv(x, y)
    ctx.sync()
    hw = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(os.path.join(h, "freq1_input"))]
    stop, samples = threading.Event(), []

    def sampler():
        while not stop.is_set():
            row = []
            for h in hw:
                row.append((read(os.path.join(h, "freq1_input")), read(os.path.join(h, "freq2_input")), read(os.path.join(h, "power1_average")) if os.path.exists(os.path.join(h, "power1_average")) else read(os.path.join(h, "power1_input")), read(os.path.join(h, "temp1_input"))))
            samples.append((time.perf_counter(), row))
            time.sleep(0.1)

    th = threading.Thread(target=sampler)
    th.start()
    for rnd in range(12):
        t0 = time.perf_counter()
        for _ in range(400):
            s.spmv(x, y)
        ctx.sync()
        dt = time.perf_counter() - t0
        last = samples[-1][1] if samples else None
        print("round %2d: %.3f ms per SpMV   last sample %s" % (rnd, dt / 400 * 1e3, last), flush=True)
    stop.set()
    th.join()
    s.close()


if __name__ == "__main__":
    main()

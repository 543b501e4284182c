"""What the chip draws and clocks at while a workload runs (round 6; DESIGN 4.2 "the chip is power-limited under this kernel").

Samples the amdgpu hwmon files of GPU 0 (socket power, shader clock) every ~10 ms while a command runs and prints, for the BUSY part of the
run (power above the midpoint between the idle level and the maximum seen), the mean and maximum power and the mean / minimum / maximum shader
clock. The workloads are the bench's own: the headline batch (256 rows: one work-group per CU in every 3x3 launch), 136 rows (the 8-wave
shapes on 136 of the 256 CUs) and 32 rows (the small shapes), and the MFMA + LDS + barrier loop of `box` on 128 and on 256 work-groups.

    python tools/power_probe.py            (on the GPU box)
"""
import glob
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hwmon_files():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "power1_cap"):
            p = os.path.join(card, name)
            if os.path.exists(p) and name not in out:
                out[name] = p
        if out:
            break
    return out


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def sample_while(cmd, env=None):
    files = hwmon_files()
    pkey = "power1_average" if "power1_average" in files else "power1_input" if "power1_input" in files else None
    samples = []
    stop = threading.Event()

    def run():
        while not stop.is_set():
            row = (time.time(), read_int(files[pkey]) if pkey else None, read_int(files["freq1_input"]) if "freq1_input" in files else None)
            samples.append(row)
            time.sleep(0.01)

    th = threading.Thread(target=run)
    th.start()
    t0 = time.time()
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    t1 = time.time()
    stop.set()
    th.join()
    return proc, samples, t1 - t0, files


def summarise(name, proc, samples, wall, files, grep=()):
    pw = [s[1] / 1e6 for s in samples if s[1] is not None]
    line = "[power probe] %s: " % name
    if pw:
        lo, hi = min(pw), max(pw)
        thr = lo + 0.5 * (hi - lo)
        busy = [s for s in samples if s[1] is not None and s[1] / 1e6 >= thr]
        bp = [s[1] / 1e6 for s in busy]
        bf = [s[2] / 1e6 for s in busy if s[2] is not None]
        line += "idle %.0f W, busy (%d of %d samples) mean %.0f W, max %.0f W" % (lo, len(busy), len(samples), sum(bp) / max(1, len(bp)), hi)
        if bf:
            line += "; shader clock while busy mean %.0f MHz (min %.0f, max %.0f)" % (sum(bf) / len(bf), min(bf), max(bf))
        cap = read_int(files["power1_cap"]) if "power1_cap" in files else None
        if cap:
            line += "; power cap %.0f W" % (cap / 1e6)
    else:
        line += "no hwmon power file (%s)" % ", ".join(sorted(files))
    line += "; wall %.1f s, rc %d" % (wall, proc.returncode)
    print(line, flush=True)
    for ln in proc.stdout.splitlines():
        if any(g in ln for g in grep):
            print("    " + ln[:400], flush=True)


def main():
    py = sys.executable
    bench = [py, "bench.py", "--no-cpu-baseline", "--no-callers", "--no-pmc", "--no-profile", "--warmup", "5"]
    jobs = [
        ("bench batch 256, 1500 passes", bench + ["--batch", "256", "--steps", "1500"], ('"value"',)),
        ("bench batch 136, 2000 passes", bench + ["--batch", "136", "--steps", "2000"], ('"value"',)),
        ("bench batch 32, 4000 passes", bench + ["--batch", "32", "--steps", "4000"], ('"value"',)),
    ]
    loop = ("import ctypes,sys; sys.path.insert(0,'.'); from katago_amd import capi; lib=capi.load_library();"
            "ms,tf,mhz=ctypes.c_double(),ctypes.c_double(),ctypes.c_double();"
            "rc=lib.kmx_bench_mfma(8,%d,3,540,%d,ctypes.byref(ms),ctypes.byref(tf),ctypes.byref(mhz));"
            "print('LOOP wgs=%d rc',rc,'ms',round(ms.value,4),'tflops',round(tf.value,1),'mhz',round(mhz.value))")
    for wgs, iters in ((256, 8000), (128, 8000), (64, 8000)):
        jobs.append(("MFMA + LDS + barrier loop on %d work-groups" % wgs, [py, "-c", loop % (wgs, iters, wgs)], ("LOOP",)))
    for name, cmd, grep in jobs:
        proc, samples, wall, files = sample_while(cmd)
        summarise(name, proc, samples, wall, files, grep)


if __name__ == "__main__":
    main()

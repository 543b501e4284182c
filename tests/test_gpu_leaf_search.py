"""SURVEY 8 row f2 on the MI355X: the reference's own `benchmark` (its definition of nnEvals/s: NNEvaluator rows / search
seconds, cpp/program/playutils.cpp:843,991-1000) on b18c384nbt 19x19 with this repo's NNEvaluator and the reference's
UNMODIFIED search running its search threads as fibers - 16 descents per OS thread, 1024 leaves in flight on 64 OS
threads (integration/katamx_fibers.cpp) - against the device-resident rate of bench.py on the same box.

Round 2 measured 22-32 k nnEvals/s through the reference's callers (one blocked OS thread per leaf) beside 37-42 k
device-resident. The bar here: >= 90 % of the device-resident rate, from a real search - asserted on searches long enough for their
ramp-up and drain not to dominate (32000 visits; 1024 threads in one tree start on a narrow tree, and the first passes of a search
hold a handful of rows); the 8000- and 1600-visit rates are reported beside it."""
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import REPO, ref_binary
from katago_amd import modelgen
import test_gpu_reference_harness as h

pytestmark = pytest.mark.gpu


def benchmark(binary, model, cfg, threads, visits, leaves, n=5):
    env = dict(os.environ, KATAMX_FIBER_STATS="1", KATAMX_LEAVES_PER_THREAD=str(leaves))
    r = subprocess.run([binary, "benchmark", "-model", model, "-config", cfg, "-v", str(visits), "-t", str(threads), "-boardsize", "19", "-n", str(n)],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(binary))
    out = (r.stdout + r.stderr).replace("\r", "\n")
    assert r.returncode == 0, out[-3000:]
    m = re.findall(r"numSearchThreads = +(\d+):.*visits/s = ([\d.]+) nnEvals/s = ([\d.]+).*avgBatchSize = ([\d.]+)", out)
    assert m, out[-2000:]
    f = re.search(r"katamx fibers: (\d+) fibers run, (\d+) parks, (\d+) blocking waits, (\d+) carrier threads", out)
    return float(m[-1][2]), float(m[-1][1]), float(m[-1][3]), [int(x) for x in f.groups()] if f else None


def test_search_driven_rate_reaches_the_device_rate(tmp_path):
    binary = ref_binary("katago_hip")
    model = str(tmp_path / "b18.bin.gz")
    modelgen.write_model(model, "b18c384nbt", seed=7)
    cfg = tmp_path / "bench.cfg"
    cfg.write_text(h.BENCH_CFG + "nnMaxBatchSize = 256\nnumNNServerThreadsPerModel = 2\n")
    # the device-resident rate of this box, same process layout as the driver's bench run
    # (only `value` is wanted: without --no-callers bench.py would play its whole self-play leg here, minutes of the suite's 20)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-callers", "--no-profile", "--steps", "40", "--warmup", "5"],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    device = json.loads(r.stdout.strip().splitlines()[-1])["value"]
    lines = ["device-resident (bench.py, batch 256): %.0f evals/s" % device]
    threads_rate, _, _, c = benchmark(binary, model, str(cfg), 512, 8000, 1)
    assert c == [0, 0, 0, 0]
    lines.append("reference search, 512 OS threads (one per leaf): %.0f nnEvals/s" % threads_rate)
    rate, visits, avg_batch, c = benchmark(binary, model, str(cfg), 1024, 8000, 16)
    lines.append("reference search, 1024 search threads as fibers on 64 OS threads: %.0f nnEvals/s, %.0f visits/s, avg batch %.0f, %s"
                 % (rate, visits, avg_batch, c))
    # 1024 threads in ONE tree ramp up and drain once per search (the first passes hold a handful of rows: the tree is narrow), so the
    # rate of a search grows with its length; a 32000-visit search is 0.8 s of device time
    long_rate, _, long_batch, _ = benchmark(binary, model, str(cfg), 1024, 32000, 16, n=3)
    lines.append("same, 32000 visits per search: %.0f nnEvals/s (%.0f %% of the device-resident rate), avg batch %.0f" % (long_rate, 100.0 * long_rate / device, long_batch))
    short, _, _, _ = benchmark(binary, model, str(cfg), 1024, 1600, 16)
    lines.append("same, 1600 visits per search (BASELINE configs[1]; a search this short spends a third of its time ramping up and draining): %.0f nnEvals/s" % short)
    print("\n".join(lines))
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "search_driven_rate.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    assert c[3] == 63 and c[1] == c[2] and c[1] > 10000, c
    # Round 4's final kernels and batcher, boxes whose device does 38.5 / 43.6 k evals/s: 32000-visit searches reach 38.7 k (100 %) /
    # 42.4 k (97 %), 8000-visit searches 35.4 k (92 %) / 40.9 k (94 %), 1600-visit searches 36.3 k (83 %) on the faster box.
    assert long_rate >= 0.9 * device, lines
    assert rate >= 0.8 * device, lines
    # ... and an absolute floor for the short-search regime beside the ratio: 8000-visit searches measured 35.4 k on the
    # slowest box of round 4 (device 38.5 k), 1600-visit ones 33.8-36.3 k
    assert rate >= 31000.0 and short >= 28000.0, lines
    assert rate >= 0.9 * threads_rate, lines  # (since the batcher seals at the device's granule, 512 OS threads are not far behind)

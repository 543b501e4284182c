"""How many host cores does one MI355X need - and eight? Measured WITHOUT a GPU: the
whole host stack of the product binding - the reference's unmodified search on fibers, this repo's NNEvaluator (hash, cache,
featurisation into bit planes, submit by ticket, post-processing), the leaf batcher's three threads, the engine's launch calls
(~122 per pass of b18c384nbt) and its staging copies - runs against the fake HIP runtime (tests/fakehip/fakehip.cpp, quiet
mode): device memory is host memory, a kernel launch returns at once, so the device costs nothing and what is left is
exactly the host work per evaluated row. The reference's `benchmark` command runs twice with different visit counts; the
difference of the process CPU times over the difference of the evaluated rows is the marginal host cost of a row (start-up -
model load, weight re-tiling - cancels).

The network's outputs are whatever the never-written device buffers hold (zeros: a uniform policy), so the search tree is wider
and shallower than a real one; the per-row costs - featurise ~68 us, post-process ~41 us, descent and
backup - are of the same kind. The figure is a budget, not a benchmark."""
import os
import re
import resource
import subprocess
import sys

import pytest

from conftest import REPO, ref_binary

sys.path.insert(0, os.path.join(REPO, "tests"))
import test_schedule_dryrun as dry  # noqa: E402

CFG = """logDir = %s
logAllGTPCommunication = false
logSearchInfo = false
logToStderr = false
rules = tromp-taylor
allowResignation = false
maxVisits = 200
numSearchThreads = 8
nnCacheSizePowerOfTwo = 18
nnMutexPoolSizePowerOfTwo = 14
nnRandomize = true
ponderingEnabled = false
lagBuffer = 1.0
searchFactorAfterOnePass = 0.5
searchFactorAfterTwoPass = 0.25
searchFactorWhenWinning = 0.4
searchFactorWhenWinningThreshold = 0.95
nnMaxBatchSize = 256
numNNServerThreadsPerModel = 2
"""
DEVICE_ROWS_PER_S = 40000.0  # what one MI355X evaluates (bench.py, b18c384nbt batch 256)


def run_benchmark(binary, fake_so, model, cfg, visits, positions, tmp):
    env = dict(os.environ, LD_PRELOAD=fake_so, KMX_FAKEHIP_QUIET="1", KMX_FAKEHIP_LOG=os.path.join(tmp, "fake_%d.log" % visits),
               KATAMX_LEAVES_PER_THREAD="16")
    before = resource.getrusage(resource.RUSAGE_CHILDREN)
    p = subprocess.run([binary, "benchmark", "-model", model, "-config", cfg, "-v", str(visits), "-t", "256", "-boardsize", "19", "-n", str(positions)],
                       capture_output=True, text=True, timeout=900, env=env, cwd=tmp)
    after = resource.getrusage(resource.RUSAGE_CHILDREN)
    out = (p.stdout + p.stderr).replace("\r", "\n")
    assert p.returncode == 0, out[-2000:]
    m = re.findall(r"visits/s = ([\d.]+) nnEvals/s = ([\d.]+) nnBatches/s = ([\d.]+) avgBatchSize = ([\d.]+)", out)
    assert m, out[-2000:]
    vps, eps, bps, avg = (float(x) for x in m[-1])
    rows = visits * positions * eps / vps
    cpu = (after.ru_utime + after.ru_stime) - (before.ru_utime + before.ru_stime)
    return rows, cpu, eps, avg


def test_host_cores_per_device(tmp_path):
    from katago_amd import modelgen

    binary = ref_binary("katago_hip")
    tmp = str(tmp_path)
    fake_so = dry.build_fakehip(tmp)
    model = os.path.join(tmp, "b18.bin.gz")
    modelgen.write_model(model, "b18c384nbt", seed=7)
    cfg = os.path.join(tmp, "bench.cfg")
    with open(cfg, "w") as f:
        f.write(CFG % os.path.join(tmp, "gtp_logs"))
    # (the difference of two CPU-time totals: on a host that is busy with something else - the suite's other workers, a compiler - the
    # start-up part of a run, identical in both, can swing by more than the difference; such a pair is measured again, three times at most)
    for attempt in range(3):
        r1, c1, _, _ = run_benchmark(binary, fake_so, model, cfg, 800, 4, tmp)
        r2, c2, eps, avg = run_benchmark(binary, fake_so, model, cfg, 6400, 4, tmp)
        us_per_row = (c2 - c1) / (r2 - r1) * 1e6
        if 30.0 < us_per_row < 1500.0:
            break
        print("attempt %d: %.0f -> %.0f rows, %.2f -> %.2f s of CPU (%.0f us per row): measured again" % (attempt, r1, r2, c1, c2, us_per_row))
    cores_one = DEVICE_ROWS_PER_S * us_per_row * 1e-6
    line = ("host capacity (fake device, katago_hip benchmark on 256 descents / 16 carrier threads, b18c384nbt 19x19, own evaluator + featuriser + "
            "fibers + leaf batcher): %.0f -> %.0f rows, %.2f -> %.2f s of CPU: %.0f us of host CPU per evaluated row = %.1f cores per MI355X at "
            "%.0f k rows/s, %.0f cores for 8 (this host: %d cores; the run itself reached %.0f nnEvals/s at an average batch of %.0f rows)"
            % (r1, r2, c1, c2, us_per_row, cores_one, DEVICE_ROWS_PER_S / 1e3, 8 * cores_one, os.cpu_count() or 0, eps, avg))
    print(line)
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "host_capacity.txt"), "w") as f:
            f.write(line + "\n")
    # a budget, not a benchmark: guard against a collapse only (round 3 measured ~110 us of featurisation + post-processing per row)
    assert 30.0 < us_per_row < 1500.0, line

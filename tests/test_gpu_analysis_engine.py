"""-m gpu, BASELINE configs[3]: b28c512nbt (random weights) through the reference's ANALYSIS ENGINE (`katago analysis`,
cpp/command/analysis.cpp; settings of cpp/configs/analysis_example.cfg:95-132 scaled to one MI355X: 512 positions in flight,
nnMaxBatchSize = 512) on this repo's NNEvaluator and HIP backend. JSON queries in, JSON lines out; the rate is NN rows (the
engine's own "NN rows" log line, analysis.cpp:1292) over the wall time of the query stream, beside the device-resident rate of
the same net and batch size on the same box.

Two ways to keep 512 leaves in flight: the reference's way, 512 analysis threads of one search thread each (512 OS threads), and
64 analysis threads x 8 search threads with the search threads as fibers (64 OS threads, integration/katamx_fibers.cpp)."""
import json
import os
import random
import re
import subprocess
import sys
import time

import pytest

from conftest import REPO, ref_binary
from katago_amd import modelgen

pytestmark = pytest.mark.gpu

CFG = """logDir = analysis_logs
logToStderr = true
logAllRequests = false
logAllResponses = false
numAnalysisThreads = %d
numSearchThreadsPerAnalysisThread = %d
nnMaxBatchSize = 512
nnCacheSizePowerOfTwo = 20
nnMutexPoolSizePowerOfTwo = 16
nnRandomize = true
numNNServerThreadsPerModel = 2
reportAnalysisWinratesAs = BLACK
"""


def queries(n, visits):
    rng = random.Random(11)
    cols = "ABCDEFGHJKLMNOPQRST"
    out = []
    for i in range(n):
        moves, used, pla = [], set(), "B"
        for _ in range(10 + i % 60):
            while True:
                xy = (rng.randrange(19), rng.randrange(19))
                if xy not in used:
                    used.add(xy)
                    break
            moves.append([pla, "%s%d" % (cols[xy[0]], xy[1] + 1)])
            pla = "W" if pla == "B" else "B"
        out.append(json.dumps({"id": "q%d" % i, "moves": moves, "rules": "tromp-taylor", "komi": 7.5, "boardXSize": 19, "boardYSize": 19,
                               "maxVisits": visits}))
    return "\n".join(out) + "\n"


def device_rate(model, batch):
    code = ("import sys, time, ctypes, numpy as np; sys.path.insert(0, %r)\n"
            "from katago_amd import nninterface as nn\n"
            "sys.path.insert(0, %r); from conftest import make_rows\n"
            "nn.globalInitialize(); ctx = nn.createComputeContext([0], 19, 19, precision='auto')\n"
            "h = nn.createComputeHandle(ctx, nn.loadModelFile(%r), %d)\n"
            "sp, gl = make_rows(np.random.default_rng(0), %d)\n"
            "nn.getOutput(h, sp, gl)\n"
            "t = time.time(); k = 6\n"
            "for _ in range(k): nn.getOutput(h, sp, gl)\n"
            "print('RATE', k * %d / (time.time() - t))\n") % (REPO, os.path.join(REPO, "tests"), model, batch, batch, batch)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return float(r.stdout.split("RATE")[1])


def test_analysis_engine_b28_batch_512(tmp_path):
    binary = ref_binary("katago_hip")
    model = str(tmp_path / "b28.bin.gz")
    modelgen.write_model(model, "b28c512nbt", seed=28)
    dev = device_rate(model, 512)  # host rows in, host rows out (kmx_eval): what a caller of the boundary can get at best
    lines = ["b28c512nbt 19x19, batch 512, kmx_eval from host rows on this box: %.0f evals/s" % dev]
    text = queries(1024, 100)
    rates = {}
    for name, threads, search_threads, leaves in (("512 analysis threads x 1 search thread (512 OS threads)", 512, 1, 1),
                                                  ("64 analysis threads x 8 search threads on fibers (64 OS threads)", 64, 8, 8)):
        cfg = tmp_path / ("analysis_%d.cfg" % threads)
        cfg.write_text(CFG % (threads, search_threads))
        env = dict(os.environ, KATAMX_LEAVES_PER_THREAD=str(leaves))
        t0 = time.time()
        r = subprocess.run([binary, "analysis", "-model", model, "-config", str(cfg)], input=text, capture_output=True, text=True, timeout=900,
                           cwd=str(tmp_path), env=env)
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-3000:]
        answers = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(answers) == 1024 and all("moveInfos" in a and a["rootInfo"]["visits"] >= 90 for a in answers)
        rows = int(re.search(r"NN rows: (\d+)", r.stderr).group(1))
        avg = float(re.search(r"NN avg batch size: ([\d.]+)", r.stderr).group(1))
        # the engine's start-up (model load, batcher engines) is not part of the stream: time from "ready" to the last answer is not
        # logged by the reference, so the whole process is timed and the start-up measured with an empty query stream is subtracted
        e = subprocess.run([binary, "analysis", "-model", model, "-config", str(cfg)], input="", capture_output=True, text=True, timeout=900,
                           cwd=str(tmp_path), env=env)
        t1 = time.time()
        startup = t1 - (t0 + wall)
        rates[name] = rows / max(wall - startup, 1e-3)
        lines.append("analysis engine, %s: %d NN rows, avg device batch %.0f, %.1f s of query stream (%.1f s start-up subtracted): %.0f nnEvals/s = %.0f %% of the device rate"
                     % (name, rows, avg, wall - startup, startup, rates[name], 100.0 * rates[name] / dev))
        assert e.returncode == 0
    print("\n".join(lines))
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "analysis_engine_b28.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    assert max(rates.values()) >= 0.7 * dev, lines

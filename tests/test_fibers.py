"""K leaves in flight per OS thread (integration/katamx_fibers.cpp), on the CPU.

katago_oraclex = the reference's host code + this repo's NNEvaluator + the CPU oracle behind the leaf port, linked with
--wrap so that the reference's UNMODIFIED search runs its search threads as fibers when KATAMX_LEAVES_PER_THREAD > 1.
With the variable unset the reference's own thread pool runs (tests/test_nneval_own.py pins that path to the reference's
evaluator character for character). Here: the same analysis queries with 8 search threads on 8 OS threads (the reference's
way) and as 8 fibers on ONE OS thread. A multi-threaded search is not deterministic in the reference either (virtual losses
depend on timing), so the comparison is statistical, as the reference's own multi-thread tests are."""
import json
import os
import random
import re
import subprocess

from conftest import REPO, ref_binary

G170 = os.path.join(REPO, "oracle", "_ref", "models", "g170-b6c96-s175395328-d26788732.bin.gz")

CFG = """logDir = analysis_logs
logToStderr = false
numAnalysisThreads = 1
numSearchThreadsPerAnalysisThread = 8
nnMaxBatchSize = 8
nnCacheSizePowerOfTwo = 16
nnMutexPoolSizePowerOfTwo = 12
nnRandomize = false
reportAnalysisWinratesAs = BLACK
"""


def queries(n, visits):
    rng = random.Random(5)
    cols = "ABCDEFGHJ"
    out = []
    for i in range(n):
        moves, used, pla = [], set(), "B"
        for _ in range(4 + 3 * i):
            while True:
                xy = (rng.randrange(9), rng.randrange(9))
                if xy not in used:
                    used.add(xy)
                    break
            moves.append([pla, "%s%d" % (cols[xy[0]], xy[1] + 1)])
            pla = "W" if pla == "B" else "B"
        out.append(json.dumps({"id": "q%d" % i, "moves": moves, "rules": "tromp-taylor", "komi": 7.0, "boardXSize": 9, "boardYSize": 9,
                               "maxVisits": visits}))
    return "\n".join(out) + "\n"


def analyse(tmp_path, leaves, text):
    cfg = tmp_path / "analysis.cfg"
    cfg.write_text(CFG)
    env = dict(os.environ, KATAMX_FIBER_STATS="1", KATAMX_LEAVES_PER_THREAD=str(leaves))
    r = subprocess.run([ref_binary("katago_oraclex"), "analysis", "-model", G170, "-config", str(cfg)], input=text, capture_output=True,
                       text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    res = {}
    for line in r.stdout.splitlines():
        d = json.loads(line)
        res[d["id"]] = d
    m = re.search(r"katamx fibers: (\d+) fibers run, (\d+) parks, (\d+) blocking waits, (\d+) carrier threads", r.stderr)
    assert m, r.stderr[-500:]
    return res, [int(x) for x in m.groups()]


def test_search_threads_as_fibers_on_one_os_thread(tmp_path):
    text = queries(6, 200)
    ref, c0 = analyse(tmp_path, 1, text)
    fib, c1 = analyse(tmp_path, 8, text)
    assert c0 == [0, 0, 0, 0]  # fibers off: the reference's thread pool
    # 8 logical search threads per search -> 8 fibers per fan-out on the calling thread, no carrier thread needed;
    # every leaf that went to the port parked its fiber, and every park ended in exactly one collected ticket
    assert c1[0] >= 6 * 8 and c1[1] > 6 * 100 and c1[1] == c1[2] and c1[3] == 0, c1
    assert set(ref) == set(fib) and len(ref) == 6
    for q in sorted(ref):
        a, b = ref[q], fib[q]
        assert abs(a["rootInfo"]["visits"] - b["rootInfo"]["visits"]) <= 8  # the budget, give or take the descents in flight at the end
        assert abs(a["rootInfo"]["winrate"] - b["rootInfo"]["winrate"]) <= 0.08, (q, a["rootInfo"], b["rootInfo"])
        # the move the 8-OS-thread search preferred keeps at least 40 % of its share of the visits (flat positions split their
        # visits over many near-equal moves, so "same best move" would be a coin toss there)
        best = max(a["moveInfos"], key=lambda m: m["visits"])
        same = [m for m in b["moveInfos"] if m["move"] == best["move"]]
        assert same and same[0]["visits"] >= 0.4 * best["visits"], (q, best["move"], best["visits"], same and same[0]["visits"])

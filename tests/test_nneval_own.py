"""This repo's NNEvaluator (integration/katamx_nneval.cpp: no server threads; hash -> cache -> featurise -> leaf ticket ->
post-process on the caller's thread) against the reference's own NNEvaluator
(cpp/neuralnet/nneval.cpp), on the CPU.

oracle/_ref/katago_oracle and oracle/_ref/katago_oraclex are the SAME reference host code and the SAME backend (the CPU
oracle) and differ in exactly one translation unit: neuralnet/nneval.cpp against integration/katamx_nneval.cpp. Whatever a
reference command prints through one must therefore be printed, character for character, through the other: policies and
values after post-processing, ownership maps, the batching test's cache behaviour, whole searches (visit counts, principal
variations), and the random outputs of the no-neural-net mode. Only backend log lines (start with ':') may differ - the
reference's server thread logs "GPU 0 finishing" when it exits, and there is no such thread here.

Since round 3 this evaluator also featurises (row a2): inputs version 7 comes from integration/katamx_features.cpp as bit planes,
not from NNInputs::fillRowV7 as an fp32 row. The small commands run it as shipped; the large ones run it in its checking mode,
where every row is additionally compared with the reference's featuriser (tests/test_features_own.py is the direct test)."""
import os
import re
import subprocess

import pytest

from conftest import REPO, ref_binary

G170 = os.path.join(REPO, "oracle", "_ref", "models", "g170-b6c96-s175395328-d26788732.bin.gz")


def body(binary, *args, timeout=900, cwd=None, env=None):
    r = subprocess.run([binary] + list(args), capture_output=True, text=True, timeout=timeout, cwd=cwd or os.path.dirname(binary),
                       env=dict(os.environ, **(env or {})))
    text = re.sub(r": GPU \d+ finishing, processed \d+ rows \d+ batches", "", r.stdout + r.stderr)
    return r.returncode, [l for l in text.splitlines() if l.strip() and not l.startswith(":")]


def same_output(*args, min_lines=1, features="own", **kw):
    """features: KATAMX_FEATURES of this repo's evaluator - "own" = the product default (inputs featurised as bit planes by
    integration/katamx_features.cpp), "check" = the same rows submitted, and every one of them also compared, plane by plane and
    global by global, with NNInputs::fillRowV7 (a difference aborts the command)."""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(2) as ex:  # the two binaries side by side
        f0 = ex.submit(body, ref_binary("katago_oracle"), *args, **kw)
        f1 = ex.submit(body, ref_binary("katago_oraclex"), *args, env={"KATAMX_FEATURES": features}, **kw)
        (rc0, ref), (rc1, own) = f0.result(), f1.result()
    assert rc0 == 0 and rc1 == 0, (rc0, rc1, ref[-5:], own[-5:])
    assert len(ref) >= min_lines, ref[-5:]
    assert len(own) == len(ref)
    for i, (a, b) in enumerate(zip(ref, own)):
        assert a == b, "line %d differs:\n  reference NNEvaluator: %s\n  this repo's:           %s" % (i, a, b)
    return own


def test_tiny_board_and_symmetries_identical():
    """runnnontinyboardtest (policy / value / ownership of a 5x5 board in a 6x6 buffer, post-processed) and
    runnnsymmetriestest (all eight symmetries of four positions)."""
    out = same_output("runnnontinyboardtest", G170, "false", "false", "3", "false", min_lines=25)
    gold = [l for l in open(os.path.join(REPO, "tests", "golden", "ref_runNNOnTinyBoardTest.txt")).read().splitlines() if l.strip() and not l.startswith(":")]
    assert len(out) == len(gold)  # and the golden itself is what test_oracle_pinned.py checks numerically
    same_output("runnnsymmetriestest", G170, "false", "false", "false", min_lines=400)


def test_tiny_nets_end_to_end(tmp_path):
    """cpp/tests/tinymodel.cpp: two embedded nets with expected outputs; throws on a mismatch."""
    binary = ref_binary("katago_oraclex")
    r = subprocess.run([binary, "runtinynntests", str(tmp_path), "1.0"], capture_output=True, text=True, timeout=600, cwd=os.path.dirname(binary))
    assert r.returncode == 0 and "Tiny net sanity check complete" in r.stdout + r.stderr, (r.stdout + r.stderr)[-2000:]


def test_many_positions_identical():
    """runnnonmanyposestest: 92 000 lines of post-processed outputs over the positions of tests/testnnevalcanary.cpp, symmetry 5."""
    same_output("runnnonmanyposestest", G170, "false", "false", "5", "false", min_lines=90000, features="check")


@pytest.mark.slow
def test_batching_cache_and_threads_identical():
    """runnnbatchingtest: many threads, cache on, results independent of batch composition (cpp/tests/results/runNNBatchingTest*.txt)."""
    out = same_output("runnnbatchingtest", G170, "true", "true", "false", features="check")
    ref = os.path.join(os.environ.get("KATAGO_REFERENCE", "/root/reference"), "cpp", "tests", "results", "runNNBatchingTestNHWC.txt")
    if os.path.exists(ref):
        assert "\n".join(out).strip() == open(ref).read().strip()


@pytest.mark.slow
def test_whole_searches_identical():
    """runsearchtests on the real g170 net: visit counts, values and principal variations of every search of
    cpp/tests/testsearch.cpp (fixed seeds, one search thread) - the same through both evaluators (2 x 80 s on the CPU)."""
    same_output("runsearchtests", G170, "false", "false", "0", "false", min_lines=1400, timeout=1500, features="check")

"""-m gpu: the reference's OWN test commands running on the HIP backend (oracle/_ref/katago_hip = unmodified
reference host code + integration/katamxbackend.cpp + libkatamx.so)."""
import os
import re
import subprocess

import pytest

from conftest import REPO, ref_binary

pytestmark = pytest.mark.gpu
G170 = os.path.join(REPO, "oracle", "_ref", "models", "g170-b6c96-s175395328-d26788732.bin.gz")


def run(*args, timeout=600):
    b = ref_binary("katago_hip")
    r = subprocess.run([b] + list(args), capture_output=True, text=True, timeout=timeout, cwd=os.path.dirname(b))
    return r.returncode, r.stdout + r.stderr


def test_nn_layer_known_answers_on_hip():
    """cpp/tests/testnn.cpp with its fp16 tolerance 0.03*max(|x|,3) (:8-15); the fp32 variants report unsupported."""
    rc, out = run("runnnlayertests")
    assert rc == 0, out[-3000:]
    assert "Test failed" not in out, out[-3000:]
    m = re.search(r"Tested (\d+) configurations", out)
    assert m and int(m.group(1)) == 14, out[-500:]


def test_tiny_model_on_hip(tmp_path):
    """cpp/tests/tinymodel.cpp through the reference NNEvaluator (featurisation, batching threads, post-processing)."""
    rc, out = run("runtinynntests", str(tmp_path), "1.0")
    assert rc == 0 and "Tiny net sanity check complete" in out, out[-3000:]
    assert "katamx (HIP/gfx950) backend" in out


def test_real_net_tiny_board_golden_on_hip():
    """g170-b6c96 (a real trained net, 5x5 stem, v8) vs cpp/tests/results/runNNOnTinyBoardTest.txt at 16-bit tolerance."""
    if not os.path.exists(G170):
        pytest.skip("g170 net not packaged")
    rc, out = run("runnnontinyboardtest", G170, "true", "true", "3", "true")
    assert rc == 0, out[-3000:]
    gold = open(os.path.join(REPO, "tests", "golden", "ref_runNNOnTinyBoardTest.txt")).read()

    def parse(t):
        lines = [l for l in t.splitlines() if l.strip() and not l.startswith(":")]
        scal = {}
        for l in lines:
            m = re.match(r"(Win|Loss|NoResult|ScoreMean|ScoreMeanSq|Lead)\s+(-?[\d.]+)", l)
            if m:
                scal[m.group(1)] = float(m.group(2))
        k = next(i for i, l in enumerate(lines) if l.startswith("Pass"))
        grid = lambda rows: [float(x) if x != "-" else None for l in rows for x in l.split()]
        return scal, float(lines[k].split()[1]), grid(lines[k + 1:k + 6]), grid(lines[k + 6:k + 11])

    sa, passa, pola, owna = parse(out)
    sb, passb, polb, ownb = parse(gold)
    # 16-bit device arithmetic against an fp32 golden: value within 1.5 %, score/lead within 0.5 point,
    # policy within 1.5 % (printed per-mille), ownership within 0.025 (printed per-mille)
    assert abs(sa["Win"] - sb["Win"]) <= 1.5 and abs(sa["Loss"] - sb["Loss"]) <= 1.5 and abs(sa["NoResult"] - sb["NoResult"]) <= 0.5
    assert abs(sa["ScoreMean"] - sb["ScoreMean"]) <= 0.5 and abs(sa["Lead"] - sb["Lead"]) <= 0.5
    assert abs(sa["ScoreMeanSq"] - sb["ScoreMeanSq"]) <= 0.05 * sb["ScoreMeanSq"]
    assert abs(passa - passb) <= 15 and len(pola) == len(polb) == 25 and len(owna) == len(ownb) == 25
    for u, v in zip(pola, polb):
        assert (u is None) == (v is None) and (u is None or abs(u - v) <= 15), (u, v)  # same legality pattern
    for u, v in zip(owna, ownb):
        assert abs(u - v) <= 25, (u, v)

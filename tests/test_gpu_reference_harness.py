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
    num = lambda t: [float(x) for l in t.splitlines() if not l.startswith(":") and "Hash" not in l for x in re.findall(r"-?\d+\.?\d*", l)]
    a, b = num(out), num(gold)
    assert len(a) == len(b) and len(a) > 60
    for u, v in zip(a, b):  # printed as probabilities %, points, or per-mille
        assert abs(u - v) <= max(0.6, 0.03 * abs(v)) if abs(v) < 100 else abs(u - v) <= 25, (u, v)

"""-m gpu, row n1 (north_star: "visit counts match at fixed seed"): the reference's fixed-seed search tests
(`runsearchtestsv8`, cpp/tests/testsearchv8.cpp; NNEvaluator set up by cpp/tests/testsearchcommon.cpp:193-248 with
nnRandomize = false and a fixed seed) run on the HIP backend with the real g170-b6c96 net, and every search report is
compared with the reference's golden cpp/tests/results/runSearchTestsV8Bin.txt (its CUDA fp32 backend).

A search amplifies last-digit differences of the net, so the reference's own backends do not reproduce each other's visit
counts exactly either: its fp16 golden (runSearchTestsV8FP16.txt) differs from the fp32 one by up to 2.5 % of the
visits on the best move (mean 0.30 %), total-variation distance of the child visit distributions 0.46 % on average,
root utility up to 1.8 c — with the same best move in all searches. That spread is the yardstick. Measured on the HIP
backend in fp16: 2.8 % / 0.24 %, 0.45 %, 1.8 c, same best move in 136 of 139 searches — i.e. the reference's own fp16
spread; the limits below are 2-3x it and are written out. Round 3: the backend's DEFAULT precision ("auto") is fp16 with the
reference's 1/8 range transform (model_desc.cpp scaledBy8) and is held to the same limits as explicit fp16; bf16, an opt-in
since, keeps its wider ones.
"""
import gzip
import os
import subprocess

import pytest

import search_golden as sg
from conftest import REPO, ref_binary

pytestmark = pytest.mark.gpu
GOLD = os.path.join(REPO, "tests", "golden")
G170 = os.path.join(REPO, "oracle", "_ref", "models", "g170-b6c96-s175395328-d26788732.bin.gz")


def _gold(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return sg.parse(f.read())


# a flipped best move counts as a near-tie when each search's margin between the two candidates is below this share of its child visits
# (the golden's own fp16-vs-fp32 spread of the best move's share is 2.5 %, so two moves closer than twice that are within its noise)
NEAR_TIE = 0.06


@pytest.mark.parametrize("precision,limits", [
    # same best move | visit share of the best move: max, mean | child visit distribution TV: mean | root utility (c): max, mean
    # measured on MI355X (profiles/r02/search_fixed_seed_*.txt): fp16 0.978 | 0.028, 0.0024 | 0.0045 | 1.8, 0.076 - the spread of the
    # reference's own fp16 golden; bf16 0.976 | 0.89 (one search flips), 0.015 | 0.020 | 9.3, 0.52
    ("auto", dict(searches=130, near_tie=NEAR_TIE, same_best=0.97, best_share_max=0.08, best_share_mean=0.01, tv_mean=0.015, root_util_max=4.0, root_util_mean=0.3)),
    ("fp16", dict(searches=130, near_tie=NEAR_TIE, same_best=0.97, best_share_max=0.08, best_share_mean=0.01, tv_mean=0.015, root_util_max=4.0, root_util_mean=0.3)),
    ("bf16", dict(searches=115, same_best=0.95, best_share_max=1.0, best_share_mean=0.04, tv_mean=0.06, root_util_max=15.0, root_util_mean=1.5)),
])
def test_fixed_seed_search_visit_counts_match_the_reference_golden(tmp_path, precision, limits):
    if not os.path.exists(G170):
        pytest.skip("g170 net not packaged")
    env = dict(os.environ, KATAMX_PRECISION=precision)
    r = subprocess.run([ref_binary("katago_hip"), "runsearchtestsv8", G170, "false", "false", "true"], capture_output=True, text=True,
                       timeout=1500, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    got = sg.parse(r.stdout)
    want = _gold("ref_runSearchTestsV8Bin.txt.gz")
    assert len(got) == len(want) == 165, (len(got), len(want))
    st = sg.compare(got, want)
    flips = st.pop("flips")
    print("runsearchtestsv8 on HIP (%s) vs the CUDA fp32 golden: %s" % (precision, {k: round(v, 4) for k, v in st.items()}))
    for fl in flips:
        print("  flipped best move, search %(index)d (%(root_N)d visits): here %(a_top)s, golden %(b_top)s; the golden's margin over our move "
              "%(gap_b).4f of its child visits, ours over the golden's %(gap_a).4f" % fl)
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "search_fixed_seed_%s.txt" % precision), "w") as f:
            f.write(repr(st) + "\n")
            for fl in flips:
                f.write("flip " + repr(fl) + "\n")
    assert st["searches"] >= limits["searches"]  # searches whose root visit count equals the golden's (the rest reuse a tree or the NN cache differently)
    assert st["same_best"] >= limits["same_best"], st
    # Round 5: the reference's own fp16 golden keeps the best move of its fp32 golden in every search
    # (tests/test_search_golden.py). Here a handful flip, and each flip must be a NEAR-TIE in the golden itself: the golden's best move
    # leads the move this backend prefers by less than `near_tie` of the child visits (and this backend's margin the other way is as
    # small). A flip between two moves that the golden separates clearly is a different opinion of the net - a parity failure.
    if "near_tie" in limits:
        far = [fl for fl in flips if fl["gap_b"] > limits["near_tie"] or fl["gap_a"] > limits["near_tie"]]
        assert not far, far
    for k in ("best_share_max", "best_share_mean", "tv_mean", "root_util_max", "root_util_mean"):
        assert st[k] <= limits[k], (k, st)

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


# The reference orders the children by play selection value: visits, with the best LOWER CONFIDENCE BOUND among the well-visited moves
# promoted to the top (search/searchresults.cpp) - "the best move" is decided by LCBs. A flipped best move counts as a near-tie when the
# two candidates' LCBs lie within this many c (hundredths of utility) of each other in the golden AND in this backend's search: the
# reference's own fp16 golden moves root utilities by up to 1.8 c against its fp32 golden (tests/test_search_golden.py).
# Round 5, measured (profiles/r05_*/search_fixed_seed_auto.txt): 4 flips of 139, the golden's LCB gaps 1.33 c (two searches of 200
# visits whose candidates have 23-29 visits each), 0.29 c and 0.31 c - every one a tie the golden itself resolves by a fraction of a c.
NEAR_TIE = 2.0


@pytest.mark.parametrize("precision,limits", [
    # same best move | visit share of the best move: max, mean | child visit distribution TV: mean | root utility (c): max, mean
    # measured on MI355X (profiles/r02/search_fixed_seed_*.txt): fp16 0.978 | 0.028, 0.0024 | 0.0045 | 1.8, 0.076 - the spread of the
    # reference's own fp16 golden; bf16 0.976 | 0.89 (one search flips), 0.015 | 0.020 | 9.3, 0.52
    ("auto", dict(searches=130, near_tie=NEAR_TIE, same_best=0.97, best_share_max=0.08, best_share_mean=0.01, tv_mean=0.015, root_util_max=4.0, root_util_mean=0.3)),
    # ("auto" IS fp16 with the 1/8 range transform on this net, DESIGN.md 1: the explicit "fp16" case ran the same searches a second time -
    # 21 s of a suite the driver stops at 20 minutes - and is not repeated; KATAMX_PRECISION=fp16 is covered by tests/test_gpu_reference_harness.py)
    ("bf16", dict(searches=115, same_best=0.95, best_share_max=1.0, best_share_mean=0.04, tv_mean=0.06, root_util_max=15.0, root_util_mean=1.5)),
    # fp32 on the device (round 5, KMX_PREC_FP32): no 16-bit rounding anywhere - held to the fp16 limits at most, figures in the record
    ("fp32", dict(searches=130, near_tie=NEAR_TIE, same_best=0.97, best_share_max=0.08, best_share_mean=0.01, tv_mean=0.015, root_util_max=4.0, root_util_mean=0.3)),
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
        print("  flipped best move, search %(index)d (%(root_N)d visits): here %(a_top)s, golden %(b_top)s; LCB gap between the two candidates "
              "%(lcb_gap_b)s c in the golden, %(lcb_gap_a)s c here; visit margins %(gap_b).4f / %(gap_a).4f of the child visits" % fl)
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with gzip.open(os.path.join(keep, "search_fixed_seed_%s_output.txt.gz" % precision), "wt") as f:
            f.write(r.stdout)  # the search reports themselves, for the record
        with open(os.path.join(keep, "search_fixed_seed_%s.txt" % precision), "w") as f:
            f.write(repr(st) + "\n")
            for fl in flips:
                f.write("flip " + repr(fl) + "\n")
    assert st["searches"] >= limits["searches"]  # searches whose root visit count equals the golden's (the rest reuse a tree or the NN cache differently)
    assert st["same_best"] >= limits["same_best"], st
    # Round 5: the reference's own fp16 golden keeps the best move of its fp32 golden in every search
    # (tests/test_search_golden.py). Here a handful flip, and each flip must be a NEAR-TIE in the golden itself and here: the two
    # candidates' lower confidence bounds within `near_tie` c of each other. A flip between two moves that the golden separates clearly
    # is a different opinion of the net - a parity failure.
    if "near_tie" in limits:
        # (here twice the golden's bound: the move this search put first has its LCB from this search's own, differently distributed visits)
        far = [fl for fl in flips if fl["lcb_gap_b"] is None or fl["lcb_gap_a"] is None or fl["lcb_gap_b"] > limits["near_tie"] or fl["lcb_gap_a"] > 2 * limits["near_tie"]]
        assert not far, far
    for k in ("best_share_max", "best_share_mean", "tv_mean", "root_util_max", "root_util_mean"):
        assert st[k] <= limits[k], (k, st)

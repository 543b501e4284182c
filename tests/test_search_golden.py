"""The yardstick of tests/test_gpu_search_fixed_seed.py, recomputed from the committed goldens (no GPU): how far the
reference's own fp16 backend is from its fp32 backend on the fixed-seed search tests."""
import gzip
import os

import search_golden as sg
from conftest import REPO


def _gold(name):
    with gzip.open(os.path.join(REPO, "tests", "golden", name), "rt") as f:
        return sg.parse(f.read())


def test_reference_goldens_parse_and_their_own_spread():
    fp32, fp16 = _gold("ref_runSearchTestsV8Bin.txt.gz"), _gold("ref_runSearchTestsV8FP16.txt.gz")
    assert len(fp32) == len(fp16) == 165
    assert fp32[0]["root_N"] == 200 and fp32[0]["children"][0][0] == "G6" and fp32[0]["children"][0][3] == 126
    st = sg.compare(fp16, fp32)
    assert st["searches"] >= 140 and st["same_best"] == 1.0
    assert st["best_share_max"] < 0.03 and st["tv_mean"] < 0.006 and st["root_util_max"] < 2.0
    same = sg.compare(fp32, fp32)
    assert same["tv_max"] == 0.0 and same["root_util_max"] == 0.0

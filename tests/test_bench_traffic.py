"""bench.py's accounting of measured HBM traffic (--pmc), on the committed counter summaries of round 4's final run
(profiles/r04_final/self_benchpmc_{FETCH,WRITE}_SIZE_pmc.csv): since the chained 3x3 kernel, a launch of the bench line's conv3x3 class
is one CONVOLUTION, and most of them run inside convChainKernel dispatches of two or four - the class's bytes are both kernels' bytes over
the convolutions the counted passes held, not the unchained kernel's bytes per dispatch."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_traffic_of_the_3x3_class_counts_chained_dispatches():
    t = bench.traffic_totals(lambda c: os.path.join(REPO, "profiles", "r04_final", "self_benchpmc_%s_pmc.csv" % c))
    assert set(t) == {"conv3x3", "conv1x1_pair"}
    c = t["conv3x3"]
    assert c["passes"] == [5, 5]          # 2 warm-up + 3 timed passes, one input stage each
    assert c["dispatches"] == [145, 145]  # per pass 18 chained dispatches (13 of four, 5 of two) + 11 single convolutions
    conv = bench.finish_traffic(c, 73.0)  # b18c384nbt: 73 3x3 convolutions per pass (13 * 4 + 5 * 2 + 11)
    # 104 MB per convolution against 84.8 MB algorithmic (bench.json of the same run): 1.23x
    assert 100e6 < conv["hbm_bytes_per_launch"] < 108e6, conv
    assert abs((2 * conv["fetch_kib_raw"] + conv["write_kib_raw"]) * 1024 - conv["hbm_bytes_per_launch"]) < 2048
    seam = bench.finish_traffic(t["conv1x1_pair"], 17.0)
    assert 265e6 < seam["hbm_bytes_per_launch"] < 275e6, seam  # unchanged by the re-accounting: 269.9 MB (bench_pmc.json)
    assert bench.finish_traffic(None, 73.0) is None

"""this repo's featuriser (integration/katamx_features.cpp - inputs version 7 written straight into the boundary's
bit-plane row) against the reference's NNInputs::fillRowV7 (cpp/neuralnet/nninputs.cpp:2288-2731).

oracle/_ref/features_selftest links both (the reference's core/, game/ and nninputs.cpp compiled where they lie) and plays random
games under every combination of ko rule x scoring x tax x suicide x button x handicap bonus x friendly pass, on square and
rectangular boards inside 19x19 and exact-size buffers, through both encore phases and to the end of the game; at every position,
for both colours to move and several MiscNNInputParams (conservative pass at the root, passing hacks, history limits, playout
doubling advantage, draw equivalence, pass-alive override), the expanded bit planes must equal the reference's fp32 row byte for
byte and the 19 globals bit for bit. Runs with the ladder memo at its default size, off, and at 1024 entries (evictions)."""
import os
import re
import subprocess

import pytest

from conftest import ref_binary


def run(games, seed, memo_log2=None, threads=1):
    binary = ref_binary("features_selftest")
    env = dict(os.environ)
    if memo_log2 is not None:
        env["KATAMX_LADDER_MEMO_LOG2"] = str(memo_log2)
    r = subprocess.run([binary, str(games), str(seed)] + (["threads", str(threads)] if threads > 1 else []), capture_output=True, text=True,
                       timeout=900, env=env)
    m = re.search(r"(\d+) comparisons, (\d+) mismatches \(positions in encore 1: (\d+), encore 2: (\d+), where a pass would end the game: (\d+), finished games: (\d+)", r.stdout)
    assert m, r.stdout[-2000:] + r.stderr[-2000:]
    comparisons, mismatches, enc1, enc2, pass_ends, finished = map(int, m.groups())
    assert r.returncode == 0 and mismatches == 0, r.stdout[-3000:]
    return comparisons, enc1, enc2, pass_ends, finished


@pytest.mark.parametrize("memo_log2", [None, 0, 10])
def test_bit_planes_equal_the_reference_rows(memo_log2):
    comparisons, enc1, enc2, pass_ends, finished = run(120, 20260922 + (memo_log2 or 0), memo_log2)
    # the corpus must actually reach the rare states
    assert comparisons > 30000 and enc1 > 300 and enc2 > 300 and pass_ends > 200 and finished > 100


@pytest.mark.parametrize("memo_log2", [None, 10])
def test_concurrent_featurisation_shares_the_ladder_memo(memo_log2):
    """Eight threads featurise their own games at once - the way a search's threads do - through the ONE process-wide ladder memo
    (default size, and 1024 entries so that they evict each other's entries all the time): still no difference from the reference."""
    comparisons, enc1, enc2, pass_ends, finished = run(30, 77 + (memo_log2 or 0), memo_log2, threads=8)
    assert comparisons > 60000 and finished > 200

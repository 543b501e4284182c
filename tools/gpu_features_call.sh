#!/bin/bash
# One short GPU call: kmx_batcher_submit_packed on the device, then the reference's selfplay and benchmark on katago_hipx with this
# repo's featuriser (KATAMX_FEATURES=own, the default) and, last, with the reference's (A/B of the host side on the GPU box).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c15
rm -rf $OUT; mkdir -p $OUT
timeout 70 python -m pytest tests/test_gpu_batcher.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/batcher.log
KATAMX_FEATURES=own timeout 80 python -m pytest "tests/test_gpu_selfplay.py::test_selfplay_rate_with_leaves_in_flight_per_game" -m gpu -x -q -s -p no:cacheprovider > $OUT/selfplay_own.log 2>&1
cp gpurun_out/selfplay_rate_b18_own_evaluator.txt $OUT/selfplay_rate_own_features.txt 2>/dev/null; tail -2 $OUT/selfplay_own.log
KATAMX_FEATURES=own timeout 70 python -m pytest tests/test_gpu_leaf_search.py -m gpu -x -q -s -p no:cacheprovider > $OUT/leaf_search_own.log 2>&1
cp gpurun_out/search_driven_rate.txt $OUT/search_driven_rate_own_features.txt 2>/dev/null; tail -2 $OUT/leaf_search_own.log
KATAMX_FEATURES=reference timeout 80 python -m pytest "tests/test_gpu_selfplay.py::test_selfplay_rate_with_leaves_in_flight_per_game" -m gpu -x -q -s -p no:cacheprovider > $OUT/selfplay_reference.log 2>&1
cp gpurun_out/selfplay_rate_b18_own_evaluator.txt $OUT/selfplay_rate_reference_features.txt 2>/dev/null; tail -2 $OUT/selfplay_reference.log
cat $OUT/selfplay_rate_own_features.txt $OUT/selfplay_rate_reference_features.txt $OUT/search_driven_rate_own_features.txt 2>/dev/null

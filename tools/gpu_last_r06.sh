#!/bin/bash
# Round 6, last call: the full -m gpu suite EXACTLY as the driver runs it at round end (`python -m pytest tests/ -x -q -m gpu`, which the
# driver stops at 1200 s - GPUTEST_r05.json steps[0].timeout_s), with every test's duration kept. The suite had grown to 1103 s
# (profiles/r06_final/pytest_gpu.log) because test_gpu_leaf_search.py's bench.py child played bench.py's whole self-play leg; this run is the
# check that the trimmed suite fits with room. Output under gpurun_out/last_r06; the log is copied to profiles/r06_final.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/last_r06
rm -rf $OUT; mkdir -p $OUT
t0=$SECONDS
timeout ${KMX_SUITE_LIMIT_S:-1190} python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=60 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $? after $((SECONDS - t0)) s" | tee $OUT/suite_seconds.txt
tail -75 $OUT/pytest_gpu.log | cut -c1-200

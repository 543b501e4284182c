#!/bin/sh
# Copies the reference's own golden OUTPUT files (data, not code) that pin the hot path into tests/golden/.
# Source: cpp/tests/results/, produced by cpp/runsearchtests.sh:40 with the CUDA fp32 backend.
set -e
REF=${KATAGO_REFERENCE:-/root/reference}
cp "$REF/cpp/tests/results/runNNOnTinyBoardTest.txt" "$(dirname "$0")/../tests/golden/ref_runNNOnTinyBoardTest.txt"
# fixed-seed search reports of the g170-b6c96 net (row n1): the CUDA fp32 golden and the reference's own fp16 golden,
# whose mutual distance is the yardstick for "visit counts match" (tests/test_gpu_search_fixed_seed.py)
gzip -9 -n -c "$REF/cpp/tests/results/runSearchTestsV8Bin.txt" > "$(dirname "$0")/../tests/golden/ref_runSearchTestsV8Bin.txt.gz"
gzip -9 -n -c "$REF/cpp/tests/results/runSearchTestsV8FP16.txt" > "$(dirname "$0")/../tests/golden/ref_runSearchTestsV8FP16.txt.gz"

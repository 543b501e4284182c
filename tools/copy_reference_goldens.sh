#!/bin/sh
# Copies the reference's own golden OUTPUT files (data, not code) that pin the hot path into tests/golden/.
# Source: cpp/tests/results/, produced by cpp/runsearchtests.sh:40 with the CUDA fp32 backend.
set -e
REF=${KATAGO_REFERENCE:-/root/reference}
cp "$REF/cpp/tests/results/runNNOnTinyBoardTest.txt" "$(dirname "$0")/../tests/golden/ref_runNNOnTinyBoardTest.txt"

#!/bin/bash
# What the leaf batcher + device deliver at self-play's numbers of leaves in flight when the HOST does nothing per row (round 6): the leaf
# pump (integration/leaf_pump.cpp: T threads x K tickets, rows resubmitted the moment they return) at 64 leaves (8 games x 8) and 256 leaves
# (32 x 8), 1-3 batches in flight. Beside `katago_hip selfplay` at the same leaf counts (20-22 k / 29-30 k NN rows/s) this says how much of the
# gap to the device-resident rate is the search's turn-around and how much is pass latency at those batch sizes.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/r06/pump}
mkdir -p $OUT
M=/tmp/kmx_pump_b18.bin.gz
python - <<EOF
import sys; sys.path.insert(0, '.')
from katago_amd import modelgen
modelgen.write_model("$M", "b18c384nbt")
EOF
for spec in "8 8 1" "8 8 2" "8 8 3" "16 4 2" "64 1 2" "32 8 1" "32 8 2" "32 8 3" "100 1 2"; do
  set -- $spec
  echo -n "threads $1 x tickets $2, in flight $3: " | tee -a $OUT/pump.txt
  timeout 60 katago_amd/leaf_pump $M 19 256 $3 $1 $2 4 2>&1 | tail -1 | tee -a $OUT/pump.txt
done

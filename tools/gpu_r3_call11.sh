#!/bin/bash
# round 3, call 11: non-temporal hints in the convolution (epilogue stores / residual loads, image DMA): A/B on one box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c11
rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
for mode in 0 1; do
for var in 3000 35768 68536 101304; do
  timeout 60 python tools/conv_one.py 3 23 $var 192 192 $mode 40 2>/dev/null | tee -a $OUT/nt.txt
done; done; done
timeout 120 python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $OUT/smoke.txt

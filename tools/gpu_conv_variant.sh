#!/bin/bash
# A/B of convolution kernel variants on one box (conv_bench.hip): python tools/conv_one.py ks cfg variant cin cout mode iters
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/conv_variant
rm -rf $OUT; mkdir -p $OUT
for rep in 1 2 3; do
for mode in 0 1; do
for var in 3000 134072 396216; do
  timeout 60 python tools/conv_one.py 3 23 $var 192 192 $mode 40 2>/dev/null | tee -a $OUT/variants.txt
done; done; done

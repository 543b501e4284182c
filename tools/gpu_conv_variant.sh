#!/bin/bash
# A/B of convolution kernel variants on one box (conv_bench.hip): python tools/conv_one.py ks cfg variant cin cout mode iters
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/conv_variant
rm -rf $OUT; mkdir -p $OUT
VARS="${VARS:-0 265144}"
for rep in 1 2 3 4; do
for mode in ${MODES:-1}; do
for var in $VARS; do
  timeout 60 python tools/conv_one.py 3 23 $var 192 192 $mode 40 2>/dev/null | tee -a $OUT/variants.txt
done; done; done
timeout 300 python -m pytest tests/test_gpu_layers.py -m gpu -x -q -p no:cacheprovider -k "residual or conv_product" 2>&1 | tail -2

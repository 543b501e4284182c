#!/bin/bash
# the small-batch 3x3 shape with dedicated fetching waves (conv_small_kernel.h): parity, then the small-batch scan with and without it
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c6; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_layers.py "tests/test_gpu_model.py::test_model_vs_oracle" tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_small.log 2>&1
tail -3 $OUT/pytest_small.log
for v in "KMX_CONV_LOADERS=0" "KMX_CONV_LOADERS=1" "KMX_CONV_LOADERS=1 KMX_CONV_LOADERS_MAX_WGS=512"; do
  echo "== $v" >> $OUT/small_batch_scan.txt
  env $v timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN >> $OUT/small_batch_scan.txt
done
cut -c1-700 $OUT/small_batch_scan.txt

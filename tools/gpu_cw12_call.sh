#!/bin/bash
# One short GPU call for the 12-wave small-batch 3x3 shape (cfg 111): layer parity with it on, then the small-batch scan with and without it
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c16
rm -rf $OUT; mkdir -p $OUT
KMX_CONV_CW12=1 timeout 75 python -m pytest tests/test_gpu_layers.py -m gpu -x -q -p no:cacheprovider -k "test_conv or residual or gpool" 2>&1 | tail -3 | tee $OUT/layers_cw12.log
KMX_CONV_CW12=1 timeout 50 python tools/small_batch_scan.py 2>&1 | grep SCAN | tee $OUT/scan_cw12_on.txt
KMX_CONV_CW12=0 timeout 40 python tools/small_batch_scan.py 2>&1 | grep SCAN | tee $OUT/scan_cw12_off.txt

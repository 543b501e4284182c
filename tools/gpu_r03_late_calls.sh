#!/bin/bash
# The short gpurun calls at the end of round 3 (what profiles/r03_steps/features and profiles/r03_steps/small_batch_cw12 come from):
#   features      kmx_batcher_submit_packed on the device; the reference's selfplay / benchmark on katago_hipx with this repo's
#                 featuriser (KATAMX_FEATURES=own, the default) and with the reference's (A/B of the host side on one box)
#   cw12          the 12-wave small-batch 3x3 shape forced on: layer parity, then tools/small_batch_scan.py with and without it
#   cw12_default  the shape as the default: whole nets against the oracle / the PyTorch goldens, fuzzed nets, packed rows, layers
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
MODE=${1:-features}
OUT=gpurun_out/r03_late_$MODE
rm -rf $OUT; mkdir -p $OUT
case $MODE in
features)
  timeout 70 python -m pytest tests/test_gpu_batcher.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/batcher.log
  for F in own reference; do
    KATAMX_FEATURES=$F timeout 80 python -m pytest "tests/test_gpu_selfplay.py::test_selfplay_rate_with_leaves_in_flight_per_game" -m gpu -x -q -s -p no:cacheprovider > $OUT/selfplay_$F.log 2>&1
    cp gpurun_out/selfplay_rate_b18_own_evaluator.txt $OUT/selfplay_rate_${F}_features.txt 2>/dev/null; tail -2 $OUT/selfplay_$F.log
  done
  KATAMX_FEATURES=own timeout 70 python -m pytest tests/test_gpu_leaf_search.py -m gpu -x -q -s -p no:cacheprovider > $OUT/leaf_search_own.log 2>&1
  cp gpurun_out/search_driven_rate.txt $OUT/search_driven_rate_own_features.txt 2>/dev/null; tail -2 $OUT/leaf_search_own.log ;;
cw12)
  KMX_CONV_CW12=1 timeout 75 python -m pytest tests/test_gpu_layers.py -m gpu -x -q -p no:cacheprovider -k "test_conv or residual or gpool" 2>&1 | tail -3 | tee $OUT/layers_cw12.log
  KMX_CONV_CW12=1 timeout 50 python tools/small_batch_scan.py 2>&1 | grep SCAN | tee $OUT/scan_cw12_on.txt
  KMX_CONV_CW12=0 timeout 40 python tools/small_batch_scan.py 2>&1 | grep SCAN | tee $OUT/scan_cw12_off.txt ;;
cw12_default)
  timeout 100 python -m pytest tests/test_gpu_model.py tests/test_gpu_fuzz.py tests/test_gpu_layers.py -m gpu -x -q -p no:cacheprovider \
    -k "torch_golden or model_vs_oracle or metadata or packed or fuzz or test_conv or residual or gpool" --durations=5 2>&1 | tail -12 | tee $OUT/parity_default_on.log ;;
esac

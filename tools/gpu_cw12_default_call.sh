#!/bin/bash
# The 12-wave small-batch shape as the DEFAULT: whole nets against the oracle / the PyTorch goldens, fuzzed nets, packed rows, the batcher
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c17
rm -rf $OUT; mkdir -p $OUT
timeout 100 python -m pytest tests/test_gpu_model.py tests/test_gpu_fuzz.py tests/test_gpu_layers.py -m gpu -x -q -p no:cacheprovider \
  -k "torch_golden or model_vs_oracle or metadata or packed or fuzz or test_conv or residual or gpool" --durations=5 2>&1 | tail -12 | tee $OUT/parity_default_on.log

#!/bin/bash
# parity of the layers / whole nets and driver-style bench lines (a quick check after a kernel change)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py tests/test_gpu_fuzz.py tests/test_gpu_pointwise.py tests/test_gpu_transformer.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity.log
for rep in 1 2 3; do
for mode in 0 1; do
for var in 0 134072; do
  timeout 60 python tools/conv_one.py 3 23 $var 192 192 $mode 40 2>/dev/null | tee -a $OUT/conv_spread_step.txt
done; done; done
for c in "1 23 0 384 192 1" "1 23 134072 384 192 1" "3 13 0 192 192 1" "3 13 134072 192 192 1"; do timeout 60 python tools/conv_one.py $c 40 2>/dev/null | tee -a $OUT/conv_spread_step.txt; done
for rep in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-callers --steps 60 --warmup 5 2>> $OUT/bench.err | grep -o '"value": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/bench.txt; echo | tee -a $OUT/bench.txt
timeout 200 python bench.py --no-cpu-baseline --no-callers --steps 60 --warmup 5 --dtype bf16 2>> $OUT/bench.err | grep -o '"value": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/bench.txt; echo " (bf16)" | tee -a $OUT/bench.txt
done

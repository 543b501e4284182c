#!/bin/bash
# parity of the layers / whole nets and a small-batch scan with and without the loader-wave shape
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity.log
timeout 100 python tools/small_batch_timing.py 2>&1 | grep -A1 "product\|4 waves x 32" | grep -v "^\[timing\]\|^--" | paste - - | tee $OUT/small_kernel.txt
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --no-profile --steps 60 --warmup 5 "$@" 2>>"$OUT/err.txt" | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/small_batch_scan.txt"; }
for n in 1 8 16 32 42; do
b "batch $n loader waves" KMX_CONV_LW=1 -- --batch $n
b "batch $n plain 4-wave shape" KMX_CONV_LW=0 -- --batch $n
done

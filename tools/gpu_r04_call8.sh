#!/bin/bash
# mid batches: the fetching-waves shape two per CU (cfg 119) against the shapes the chooser takes today
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c8; rm -rf $OUT; mkdir -p $OUT
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --no-profile "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/mid_batch_packed.txt"; }
for n in 43 48 64 85 96 128; do
  b "batch $n default" A=1 -- --batch $n --steps 40 --warmup 5
  b "batch $n packed<=512" KMX_CONV_LOADERS_PACKED_MAX_WGS=512 -- --batch $n --steps 40 --warmup 5
  b "batch $n packed<=768" KMX_CONV_LOADERS_PACKED_MAX_WGS=768 -- --batch $n --steps 40 --warmup 5
done
KMX_CONV_LOADERS_PACKED_MAX_WGS=512 timeout 300 python -m pytest tests/test_gpu_layers.py -k "test_conv" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2

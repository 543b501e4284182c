#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/call5
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_pointwise.py -m gpu -x -q -s -p no:cacheprovider 2>&1 | tail -30 > "$OUT/pointwise_tests.log"; tail -30 "$OUT/pointwise_tests.log"
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 300 python3 bench.py --no-cpu-baseline "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"kernel_time_share": {[^}]*}\|"frac": [0-9.]*\|"profiled_ms_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/scan.txt"; }
b "seams off" KMX_FUSE_SEAMS=0 -- --steps 50 --warmup 5
b "seams on " KMX_FUSE_SEAMS=1 -- --steps 50 --warmup 5
b "seams off" KMX_FUSE_SEAMS=0 -- --steps 50 --warmup 5
b "seams on " KMX_FUSE_SEAMS=1 -- --steps 50 --warmup 5
b "seams on ways1" KMX_FUSE_SEAMS=1 KMX_SPLIT_MIN=0 -- --steps 50 --warmup 5
for n in 8 24 32 64 128; do
  b "batch$n seams off" KMX_FUSE_SEAMS=0 -- --batch $n --steps 50 --warmup 5 --no-profile
  b "batch$n seams on " KMX_FUSE_SEAMS=1 KMX_FUSE_MIN_ROWS=1 -- --batch $n --steps 50 --warmup 5 --no-profile
done
timeout 900 python -m pytest tests/test_gpu_search_fixed_seed.py tests/test_gpu_model.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee "$OUT/model_tests.log"

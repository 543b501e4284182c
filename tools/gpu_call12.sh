#!/bin/bash
# Seam kernel: 4 waves x 64 cells, two work-groups per CU (default) against 8 waves x 128 cells (KMX_PW_WAVES=8).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/call12
mkdir -p "$OUT"
timeout 200 python -m pytest tests/test_gpu_pointwise.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee "$OUT/parity4.log"
KMX_PW_WAVES=8 timeout 200 python -m pytest tests/test_gpu_pointwise.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee "$OUT/parity8.log"
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 200 python3 bench.py --no-cpu-baseline "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"kernel_time_share": {[^}]*}\|"frac": [0-9.]*\|"profiled_ms_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/scan.txt"; }
b "pw8" KMX_PW_WAVES=8 -- --steps 50 --warmup 5
b "pw4" -- --steps 50 --warmup 5
b "pw8" KMX_PW_WAVES=8 -- --steps 50 --warmup 5
b "pw4" -- --steps 50 --warmup 5
b "pw4 ways1" KMX_SPLIT_MIN=0 -- --steps 50 --warmup 5

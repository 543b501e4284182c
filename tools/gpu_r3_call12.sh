#!/bin/bash
# round 3, call 12: balanced grid of the persistent seam kernel (two half-batch streams), A/B on one box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c12
rm -rf $OUT; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_pointwise.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity.log
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --no-profile --steps 60 --warmup 5 "$@" 2>>"$OUT/err.txt" | grep -o '"value": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/balance.txt"; }
for rep in 1 2 3; do
b "v2 balanced grid" KMX_PW_BALANCE=1 --
b "v2 one per CU" KMX_PW_BALANCE=0 --
b "v1" KMX_PW_V2=0 --
done
b "v2 balanced, one stream" KMX_PW_BALANCE=1 KMX_SPLIT_MIN=0 --
b "v2 one per CU, one stream" KMX_PW_BALANCE=0 KMX_SPLIT_MIN=0 --
b "v2 balanced, batch 512" KMX_PW_BALANCE=1 -- --batch 512
b "v2 one per CU, batch 512" KMX_PW_BALANCE=0 -- --batch 512

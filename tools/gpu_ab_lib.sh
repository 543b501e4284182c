#!/bin/bash
# A/B of two builds of libkatamx.so on one box: katago_amd/libkatamx_prev.so (KMX_LIBRARY) against the current one; parity first
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/ab
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_model.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/parity.log
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --steps 60 --warmup 5 "$@" 2>>"$OUT/err.txt" | grep -o '"value": [0-9.]*\|"conv1x1_pair": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/ab.txt"; }
for rep in 1 2 3; do
b "new" A=1 --
b "prev" KMX_LIBRARY=$PWD/katago_amd/libkatamx_prev.so --
done
b "new, seam v1 alone (KMX_PW_V2=0, one stream)" KMX_PW_V2=0 KMX_SPLIT_MIN=0 --
b "prev, seam v1 alone" KMX_LIBRARY=$PWD/katago_amd/libkatamx_prev.so KMX_PW_V2=0 KMX_SPLIT_MIN=0 --

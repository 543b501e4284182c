#!/bin/bash
# Round 3, call 2: the persistent seam kernel (pointwise2_kernel.h) - parity, then A/B against the round-2 kernel on one box.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c2
rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_pointwise.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/parity_pw2.log
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 "$@" 2>>"$OUT/seam.err" | grep -o '"value": [0-9.]*\|"kernel_avg_launch_us": {[^}]*}\|"frac": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/seam.txt"; }
b "v2 (two streams)" --
b "v1 (two streams)" KMX_PW_V2=0 --
b "v2 (two streams)" --
b "v1 (two streams)" KMX_PW_V2=0 --
b "v2 one stream" KMX_SPLIT_MIN=0 --
b "v1 one stream" KMX_SPLIT_MIN=0 KMX_PW_V2=0 --
b "v2 grid 512" KMX_PW_GRID=512 --
b "v2 grid 192" KMX_PW_GRID=192 --
b "v2 fp16" -- --dtype fp16
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
tail -c 300 $OUT/bench_driver_cmd.json

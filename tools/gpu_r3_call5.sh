#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c5
rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_layers.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/parity.log
timeout 120 python tools/seam_timing.py 256 > $OUT/seam_v2.txt 2>&1
KMX_PW_V2=0 timeout 120 python tools/seam_timing.py 256 > $OUT/seam_v1.txt 2>&1
cat $OUT/seam_v2.txt $OUT/seam_v1.txt
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 "$@" 2>>"$OUT/seam.err" | grep -o '"value": [0-9.]*\|"kernel_avg_launch_us": {[^}]*}\|"frac": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/seam.txt"; }
b "v2 (two streams)" --
b "v1 (two streams)" KMX_PW_V2=0 --
b "v2 one stream" KMX_SPLIT_MIN=0 --
b "v1 one stream" KMX_SPLIT_MIN=0 KMX_PW_V2=0 --
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/parity_model.log

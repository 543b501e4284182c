#!/bin/bash
# round 3, call 13: the seam kernel chosen by what runs beside it (persistent when alone, short work-groups beside other streams)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c13
rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_model.py tests/test_gpu_batcher.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity.log
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --no-profile --steps 60 --warmup 5 "$@" 2>>"$OUT/err.txt" | grep -o '"value": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/seam_two_streams.txt"; }
for rep in 1 2 3; do
b "new default (two streams: short work-groups)" A=1 --
b "persistent forced (KMX_PW_V2=2)" KMX_PW_V2=2 --
b "never persistent (KMX_PW_V2=0)" KMX_PW_V2=0 --
done
b "one stream, default (persistent)" KMX_SPLIT_MIN=0 --
b "one stream, never persistent" KMX_SPLIT_MIN=0 KMX_PW_V2=0 --
b "batch 512, default" A=1 -- --batch 512
b "batch 512, persistent forced" KMX_PW_V2=2 -- --batch 512
b "batch 128 (one stream), default" A=1 -- --batch 128
b "batch 128, never persistent" KMX_PW_V2=0 -- --batch 128
timeout 60 katago_amd/leaf_pump /tmp/kmx_bench_b18c384nbt_r0.bin 19 256 2 8 128 3 | tail -1 | tee -a $OUT/seam_two_streams.txt
KMX_PW_V2=2 timeout 60 katago_amd/leaf_pump /tmp/kmx_bench_b18c384nbt_r0.bin 19 256 2 8 128 3 | tail -1 | tee -a $OUT/seam_two_streams.txt

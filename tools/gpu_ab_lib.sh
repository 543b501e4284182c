#!/bin/bash
# A/B of two builds of libkatamx.so on one box: katago_amd/libkatamx_prev.so (KMX_LIBRARY) against the current one; parity first
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/ab
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_layers.py "tests/test_gpu_model.py::test_large_nets_of_the_analysis_config" "tests/test_gpu_model.py::test_error_statistics_large_nets_default_precision" -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/parity.log
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers "$@" 2>>"$OUT/err.txt" | grep -o '"value": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/ab.txt"; }
for rep in 1 2 3; do
b "b28c512nbt batch 512 new" A=1 -- --model b28c512nbt --batch 512 --steps 12 --warmup 3
b "b28c512nbt batch 512 prev" KMX_LIBRARY=$PWD/katago_amd/libkatamx_prev.so -- --model b28c512nbt --batch 512 --steps 12 --warmup 3
done
b "b40c256 batch 512 new" A=1 -- --model b40c256 --batch 512 --steps 12 --warmup 3
b "b40c256 batch 512 prev" KMX_LIBRARY=$PWD/katago_amd/libkatamx_prev.so -- --model b40c256 --batch 512 --steps 12 --warmup 3
b "b18 new" A=1 -- --steps 40 --warmup 5
b "b18 prev" KMX_LIBRARY=$PWD/katago_amd/libkatamx_prev.so -- --steps 40 --warmup 5

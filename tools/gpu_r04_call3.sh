#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c3
rm -rf $OUT; mkdir -p $OUT
timeout 300 python tools/chain_timing.py 256 > $OUT/chain_timing_bf16.txt 2>&1
KMX_BENCH_DTYPE=fp16 timeout 300 python tools/chain_timing.py 256 > $OUT/chain_timing_fp16.txt 2>&1
cat $OUT/chain_timing_bf16.txt | cut -c1-200
grep "convolutions" $OUT/chain_timing_fp16.txt

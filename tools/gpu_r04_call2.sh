#!/bin/bash
# Round 4, GPU call 2: the chained 3x3 convolutions (conv_chain_kernel.h) - parity on the MI355X, then A/B of the driver's metric with
# chains of 0 / 2 / 4 convolutions, on one box. Everything under gpurun_out/r4c2.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c2
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_layers.py -k "chain" "tests/test_gpu_model.py::test_full_batch_properties" \
  "tests/test_gpu_model.py::test_headline_batch_vs_oracle_default_precision" "tests/test_gpu_model.py::test_model_vs_oracle" -m gpu -q -x -p no:cacheprovider > $OUT/pytest_chain.log 2>&1
tail -4 $OUT/pytest_chain.log
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 200 python3 bench.py --no-cpu-baseline --no-callers "$@" 2>>"$OUT/scan.err" | tee -a $OUT/bench_lines.jsonl | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline") or {}
print("value %.0f ms %.3f frac %s conv3x3_us %s pair_us %s" % (d["value"], d["ms_per_step"], r.get("frac"), (r.get("kernel_avg_launch_us") or {}).get("conv3x3"), (r.get("kernel_avg_launch_us") or {}).get("conv1x1_pair")))')
  echo "$name | $v" | tee -a "$OUT/chain_ab.txt"; }
for rep in 1 2; do
  b "chain 0 (separate launches)" KMX_CONV_CHAIN=0 -- --steps 40 --warmup 5
  b "chain 2" KMX_CONV_CHAIN=2 -- --steps 40 --warmup 5
  b "chain 4" KMX_CONV_CHAIN=4 -- --steps 40 --warmup 5
done
b "chain 4, one stream" KMX_CONV_CHAIN=4 KMX_SPLIT_MIN=0 -- --steps 40 --warmup 5 --no-profile
b "chain 0, one stream" KMX_CONV_CHAIN=0 KMX_SPLIT_MIN=0 -- --steps 40 --warmup 5 --no-profile
b "chain 4, bf16" KMX_CONV_CHAIN=4 -- --steps 40 --warmup 5 --dtype bf16
b "chain 4, batch 512" KMX_CONV_CHAIN=4 -- --steps 20 --warmup 3 --batch 512 --no-profile
b "chain 0, batch 512" KMX_CONV_CHAIN=0 -- --steps 20 --warmup 3 --batch 512 --no-profile
KMX_SPLIT_MIN=0 timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/trace_chain4 -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-callers --no-profile > $OUT/trace_chain4.log 2>&1
timeout 100 python tools/rocpd_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
head -8 $OUT/summary/*kernel_stats.csv | cut -c1-200

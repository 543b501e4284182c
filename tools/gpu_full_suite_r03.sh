#!/bin/bash
# round 3: the full -m gpu suite at HEAD + smoke + split-ways scan
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/full_suite_r03
rm -rf $OUT; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 2>&1 | tail -40 > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
for f in search_fixed_seed_auto.txt search_fixed_seed_fp16.txt search_fixed_seed_bf16.txt search_driven_rate.txt selfplay_rate_b18.txt selfplay_rate_b18_own_evaluator.txt analysis_engine_b28.txt leaf_pump_b18.txt reference_benchmark_batcher.txt reference_benchmark_b18_19x19.txt reference_benchmark_b6c96_9x9.txt testgpuerror_g170_auto.txt testgpuerror_g170_fp16.txt testgpuerror_g170_bf16.txt; do [ -f gpurun_out/$f ] && cp gpurun_out/$f $OUT/; done
cp -r gpurun_out/transformer $OUT/ 2>/dev/null
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --no-profile --steps 60 --warmup 5 "$@" 2>>"$OUT/err.txt" | grep -o '"value": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/split_ways.txt"; }
for rep in 1 2; do
b "two half-batch streams (default)" A=1 --
b "three streams" KMX_SPLIT_WAYS=3 --
b "four streams" KMX_SPLIT_WAYS=4 --
done

#!/bin/bash
# round 3, call 8: the full -m gpu suite after fibers + fp16 range transform + bench changes
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c8
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=15 2>&1 | tail -60 > $OUT/pytest.log
cat $OUT/pytest.log
for f in search_fixed_seed_auto.txt search_fixed_seed_fp16.txt search_fixed_seed_bf16.txt search_driven_rate.txt selfplay_rate_b18.txt leaf_pump_b18.txt reference_benchmark_batcher.txt; do [ -f gpurun_out/$f ] && cp gpurun_out/$f $OUT/; done
for d in fp16 bf16; do timeout 100 python3 bench.py --no-cpu-baseline --no-callers --dtype $d --steps 40 --warmup 5 2>> $OUT/bench.err | grep -o '"value": [0-9.]*\|"dtype": "[a-z0-9]*"\|"frac": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/bench_dtypes.txt; echo | tee -a $OUT/bench_dtypes.txt; done
KMX_FP16_SCALE8=0 timeout 100 python3 bench.py --no-cpu-baseline --no-callers --dtype fp16 --steps 40 --warmup 5 2>> $OUT/bench.err | grep -o '"value": [0-9.]*' | tee -a $OUT/bench_dtypes.txt

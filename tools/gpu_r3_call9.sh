#!/bin/bash
# round 3, call 9: the rest of the -m gpu suite (call 8 stopped at an assertion about OS threads vs fibers)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c9
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_leaf_search.py tests/test_gpu_model.py tests/test_gpu_pointwise.py tests/test_gpu_layers.py tests/test_gpu_fuzz.py tests/test_gpu_reference_harness.py tests/test_gpu_search_fixed_seed.py tests/test_gpu_selfplay.py tests/test_gpu_transformer.py -m gpu -q -p no:cacheprovider --durations=10 2>&1 | tail -70 > $OUT/pytest.log
cat $OUT/pytest.log
for f in search_fixed_seed_auto.txt search_fixed_seed_fp16.txt search_fixed_seed_bf16.txt search_driven_rate.txt selfplay_rate_b18.txt selfplay_rate_b18_own_evaluator.txt; do [ -f gpurun_out/$f ] && cp gpurun_out/$f $OUT/; done
ls gpurun_out/

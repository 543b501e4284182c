#!/bin/bash
# the tests that go through the leaf batcher (after a change to it), with their recorded rates
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_batcher.py tests/test_gpu_leaf_pump.py tests/test_gpu_leaf_search.py tests/test_gpu_selfplay.py tests/test_gpu_analysis_engine.py "tests/test_gpu_bench_command.py::test_driver_command_exits_zero_with_roofline_and_cpu_baseline" -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/parity.log
for f in search_driven_rate.txt selfplay_rate_b18.txt selfplay_rate_b18_own_evaluator.txt analysis_engine_b28.txt leaf_pump_b18.txt; do cp gpurun_out/$f $OUT/ 2>/dev/null; done
cat $OUT/search_driven_rate.txt $OUT/selfplay_rate_b18_own_evaluator.txt $OUT/analysis_engine_b28.txt | cut -c1-250

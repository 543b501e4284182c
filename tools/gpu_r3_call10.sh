#!/bin/bash
# round 3, call 10: new tests (full-batch conv shapes, large-net statistics, analysis engine, bench line with callers)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c10
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_analysis_engine.py "tests/test_gpu_model.py::test_error_statistics_large_nets_default_precision" "tests/test_gpu_model.py::test_large_nets_of_the_analysis_config" "tests/test_gpu_bench_command.py::test_driver_command_exits_zero_with_roofline_and_cpu_baseline" -m gpu -q -p no:cacheprovider --durations=8 -s 2>&1 | tail -80 > $OUT/pytest.log
cat $OUT/pytest.log
cp gpurun_out/analysis_engine_b28.txt $OUT/ 2>/dev/null
timeout 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json

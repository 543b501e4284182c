#!/bin/bash
# the round's last sources: smoke(), the tests that exercise small batches end to end, the driver's bench line
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c16; rm -rf $OUT; mkdir -p $OUT
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_batcher.py tests/test_gpu_transformer.py tests/test_gpu_selfplay.py -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_transformer.py::test_oracle_transformer_agrees_with_reference_opencl_backend > $OUT/pytest_last.log 2>&1
tail -3 $OUT/pytest_last.log
KMX_BENCH_SELFPLAY_TIMEOUT=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['small_batches']['ms_per_pass'], d.get('reference_benchmark_nn_evals_per_s'), d.get('host_rows_through_batcher_per_s'))"

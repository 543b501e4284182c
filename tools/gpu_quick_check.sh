#!/bin/bash
# parity of the layers / whole nets and one driver-style bench line (scratch: a quick check after a small kernel change)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/quick/bench.json') if l.startswith('{')][0])
print(d['value'], d['roofline']['frac'], d['roofline_seam']['frac'], d['roofline']['kernel_avg_launch_us'], d['box'])
print(d['cpu_baseline'])
print({k: d.get(k) for k in ('host_rows_through_batcher_per_s','reference_benchmark_nn_evals_per_s','selfplay_nn_rows_per_s','callers_error')})
P

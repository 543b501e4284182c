#!/bin/bash
# 1x1 convolutions at small batch with a ring of four / five steps instead of two (cfg 114 / 115 / 124): parity on hardware, then the
# small-batch scan and three mid batches per setting on the same box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c13; rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_layers.py "tests/test_gpu_model.py::test_model_vs_oracle" tests/test_gpu_fuzz.py tests/test_gpu_model.py::test_headline_batch_vs_oracle_default_precision -m gpu -q -x -p no:cacheprovider > $OUT/pytest_deep.log 2>&1
tail -2 $OUT/pytest_deep.log
KMX_CONV_DEEP1X1=5 timeout 200 python -m pytest tests/test_gpu_layers.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_deep5.log 2>&1
tail -1 $OUT/pytest_deep5.log
for d in 0 4 5 0 4; do
  echo "== KMX_CONV_DEEP1X1=$d" >> $OUT/small_batch_scan.txt
  KMX_CONV_DEEP1X1=$d timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN >> $OUT/small_batch_scan.txt
done
cut -c1-330 $OUT/small_batch_scan.txt
for b in 24 48 85; do for d in 0 4; do
  KMX_CONV_DEEP1X1=$d timeout 120 python bench.py --batch $b --no-cpu-baseline --no-callers --no-profile --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b deep $d', d['value'], d['ms_per_step'])" | tee -a $OUT/mid_batches.txt
done; done

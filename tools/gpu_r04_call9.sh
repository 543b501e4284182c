#!/bin/bash
# (1) the fetching-waves shape at three fetch depths (conv_small_kernel.h SG<PACK, DEPTH>): parity on hardware at each, then the small-batch
#     scan per depth on the same box; (2) the leaf batcher sealing at granule multiples below max_batch_size: the search-driven rate again
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c9; rm -rf $OUT; mkdir -p $OUT
for d in 1 0 2; do
  KMX_CONV_LOADERS_DEPTH=$d timeout 400 python -m pytest tests/test_gpu_layers.py "tests/test_gpu_model.py::test_model_vs_oracle" tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_depth$d.log 2>&1
  echo "depth $d: $(tail -1 $OUT/pytest_depth$d.log)"
done
for rep in 1 2; do
  for d in 0 1 2; do
    echo "== KMX_CONV_LOADERS_DEPTH=$d (run $rep)" >> $OUT/small_batch_scan.txt
    KMX_CONV_LOADERS_DEPTH=$d timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN >> $OUT/small_batch_scan.txt
  done
done
cut -c1-600 $OUT/small_batch_scan.txt
timeout 900 python -m pytest tests/test_gpu_leaf_search.py tests/test_gpu_batcher.py tests/test_gpu_leaf_pump.py tests/test_gpu_selfplay.py -m gpu -q -p no:cacheprovider -s > $OUT/pytest_batcher.log 2>&1
tail -3 $OUT/pytest_batcher.log
cp gpurun_out/search_driven_rate.txt gpurun_out/leaf_pump_b18.txt gpurun_out/selfplay_rate*.txt $OUT/ 2>/dev/null
cat $OUT/search_driven_rate.txt

#!/bin/bash
# last check of the round: smoke(), the search-driven-rate test with its final thresholds, the driver's launch line for N > 1 with one rank
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c12; rm -rf $OUT; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 600 python -m pytest tests/test_gpu_leaf_search.py tests/test_gpu_batcher.py -m gpu -q -x -p no:cacheprovider -s > $OUT/pytest_leaf.log 2>&1
tail -3 $OUT/pytest_leaf.log; cp gpurun_out/search_driven_rate.txt $OUT/ 2>/dev/null; cat $OUT/search_driven_rate.txt
KMX_BENCH_SELFPLAY_TIMEOUT=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_torchrun.json 2> $OUT/bench_torchrun.err
tail -c 600 $OUT/bench_torchrun.json; tail -3 $OUT/bench_torchrun.err

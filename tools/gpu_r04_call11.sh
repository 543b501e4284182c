#!/bin/bash
# the leaf batcher's granule rule, A/B on one box: a batch is sealed at a granule multiple while fewer than KMX_BATCH_GROW_AHEAD batches are
# ahead of it (2 = the previous call's rule, default = max_in_flight = a slot is free)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c11; rm -rf $OUT; mkdir -p $OUT
python - > $OUT/setup.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "tests")
import test_gpu_reference_harness as h
from katago_amd import modelgen
modelgen.write_model("/tmp/b18.bin.gz", "b18c384nbt", seed=7)
open("/tmp/bench.cfg", "w").write(h.BENCH_CFG + "nnMaxBatchSize = 256\nnumNNServerThreadsPerModel = 2\n")
open("/tmp/bench1.cfg", "w").write(h.BENCH_CFG + "nnMaxBatchSize = 256\nnumNNServerThreadsPerModel = 1\n")
PY
python bench.py --no-cpu-baseline --no-callers --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device-resident', d['value'], d['roofline']['frac'])" | tee $OUT/rates.txt
cd oracle/_ref
run() {  # label, cfg, visits, extra env...
  local label=$1 cfg=$2 v=$3; shift 3
  env KATAMX_FIBER_STATS=1 KATAMX_LEAVES_PER_THREAD=16 "$@" timeout 300 ./katago_hip benchmark -model /tmp/b18.bin.gz -config $cfg -v $v -t 1024 -boardsize 19 -n 3 2>&1 | tr '\r' '\n' | grep "nnEvals/s" | tail -1 | sed "s/^/$label | /" | tee -a ../../$OUT/rates.txt
}
for rep in 1 2; do
run "v 32000, grow when 2 ahead" /tmp/bench.cfg 32000 KMX_BATCH_GROW_AHEAD=2
run "v 32000, grow when every slot is taken (3)" /tmp/bench.cfg 32000 A=1
run "v 32000, never grow (99)" /tmp/bench.cfg 32000 KMX_BATCH_GROW_AHEAD=99
done
run "v 8000, default" /tmp/bench.cfg 8000 A=1
run "v 8000, grow when 2 ahead" /tmp/bench.cfg 8000 KMX_BATCH_GROW_AHEAD=2
run "v 32000, one server thread (2 in flight), default" /tmp/bench1.cfg 32000 A=1
run "v 32000, one server thread, never grow" /tmp/bench1.cfg 32000 KMX_BATCH_GROW_AHEAD=99

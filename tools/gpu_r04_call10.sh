#!/bin/bash
# the reference's search on fibers against the device rate of the same box: how the rate depends on the length of a search (ramp-up and
# drain of 1024 threads in one tree are per search), and on glibc's allocator settings
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c10; rm -rf $OUT; mkdir -p $OUT
python - > $OUT/setup.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "tests")
import test_gpu_reference_harness as h
from katago_amd import modelgen
modelgen.write_model("/tmp/b18.bin.gz", "b18c384nbt", seed=7)
open("/tmp/bench.cfg", "w").write(h.BENCH_CFG + "nnMaxBatchSize = 256\nnumNNServerThreadsPerModel = 2\n")
PY
python bench.py --no-cpu-baseline --no-callers --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device-resident', d['value'], d['roofline']['frac'])" | tee $OUT/rates.txt
cd oracle/_ref
run() {  # label, visits, threads, leaves, extra env...
  local label=$1 v=$2 t=$3 k=$4; shift 4
  env KATAMX_FIBER_STATS=1 KATAMX_LEAVES_PER_THREAD=$k "$@" timeout 300 ./katago_hip benchmark -model /tmp/b18.bin.gz -config /tmp/bench.cfg -v $v -t $t -boardsize 19 -n 3 2>&1 | tr '\r' '\n' | grep "nnEvals/s" | tail -1 | sed "s/^/$label | /" | tee -a ../../$OUT/rates.txt
}
run "v 8000 t 1024 k 16" 8000 1024 16 A=1
run "v 32000 t 1024 k 16" 32000 1024 16 A=1
run "v 100000 t 1024 k 16" 100000 1024 16 A=1
run "v 32000 t 1024 k 16 malloc tunables" 32000 1024 16 GLIBC_TUNABLES=glibc.malloc.mmap_threshold=536870912:glibc.malloc.trim_threshold=17179869184:glibc.malloc.top_pad=67108864
run "v 32000 t 2048 k 32" 32000 2048 32 A=1
run "v 32000 t 768 k 12" 32000 768 12 A=1
run "v 8000 t 1024 k 16 quantum 0" 8000 1024 16 KMX_BATCH_QUANTUM=0
run "v 32000 t 1024 k 16 quantum 0" 32000 1024 16 KMX_BATCH_QUANTUM=0
nproc | sed 's/^/host threads /' | tee -a ../../$OUT/rates.txt

#!/bin/bash
# round 3, call 6: the reference's `benchmark` through this repo's NNEvaluator with K leaves per OS thread (fibers)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c6
rm -rf $OUT; mkdir -p $OUT
python - <<'P' > $OUT/model.log 2>&1
import sys; sys.path.insert(0, '.')
from katago_amd import modelgen
modelgen.write_model('/tmp/b18.bin.gz', 'b18c384nbt', seed=7)
P
cat > /tmp/bench.cfg <<'C'
logDir = /tmp/gtp_logs
logAllGTPCommunication = false
logSearchInfo = false
logToStderr = false
rules = tromp-taylor
allowResignation = false
maxVisits = 200
numSearchThreads = 8
nnCacheSizePowerOfTwo = 18
nnMutexPoolSizePowerOfTwo = 14
nnRandomize = true
ponderingEnabled = false
lagBuffer = 1.0
searchFactorAfterOnePass = 0.5
searchFactorAfterTwoPass = 0.25
searchFactorWhenWinning = 0.4
searchFactorWhenWinningThreshold = 0.95
nnMaxBatchSize = 256
C
run() { local name=$1 k=$2 t=$3 v=$4; shift 4
  local line=$(KATAMX_FIBER_STATS=1 KATAMX_LEAVES_PER_THREAD=$k timeout 300 oracle/_ref/katago_hipx benchmark -model /tmp/b18.bin.gz -config /tmp/bench.cfg -v $v -t $t -boardsize 19 -n 4 2>&1 | tr '\r' '\n' | grep -E "nnEvals/s|katamx fibers" | sed 's/^ *//' | tr '\n' '|')
  echo "$name K=$k t=$t v=$v | $line" | tee -a $OUT/fibers.txt; }
nproc | tee -a $OUT/fibers.txt
run "own nneval, OS threads" 1 256 8000
run "own nneval, OS threads" 1 512 8000
run "fibers" 8 512 8000
run "fibers" 8 1024 8000
run "fibers" 16 1024 8000
run "fibers" 32 1024 8000
run "fibers" 16 512 8000
run "fibers 1600 visits" 8 512 1600
run "OS threads 1600 visits" 1 256 1600
timeout 200 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json

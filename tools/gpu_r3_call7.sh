#!/bin/bash
# round 3, call 7: fibers + the batcher's seal quantum
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c7
rm -rf $OUT; mkdir -p $OUT
python - <<'P' > $OUT/model.log 2>&1
import sys; sys.path.insert(0, '.')
from katago_amd import modelgen
modelgen.write_model('/tmp/b18.bin.gz', 'b18c384nbt', seed=7)
P
mkcfg() { cat > /tmp/bench$1.cfg <<C
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
numNNServerThreadsPerModel = $1
C
}
mkcfg 1; mkcfg 2; mkcfg 3
run() { local name=$1 k=$2 t=$3 v=$4 srv=$5; shift 5
  local line=$(env "$@" KATAMX_FIBER_STATS=1 KATAMX_LEAVES_PER_THREAD=$k timeout 300 oracle/_ref/katago_hipx benchmark -model /tmp/b18.bin.gz -config /tmp/bench$srv.cfg -v $v -t $t -boardsize 19 -n 5 2>&1 | tr '\r' '\n' | grep -E "nnEvals/s|katamx fibers" | sed 's/^ *//' | tr '\n' '|')
  echo "$name K=$k t=$t v=$v inflight=$((srv+1)) $* | $line" | tee -a $OUT/fibers.txt; }
run "quantum" 16 512 8000 1 A=1
run "quantum" 16 768 8000 1 A=1
run "quantum" 16 1024 8000 1 A=1
run "quantum" 16 768 8000 2 A=1
run "quantum" 16 1024 8000 2 A=1
run "quantum" 16 1024 8000 3 A=1
run "quantum" 32 1024 8000 2 A=1
run "no quantum" 16 1024 8000 1 KMX_BATCH_QUANTUM=0
run "no quantum" 16 1024 8000 2 KMX_BATCH_QUANTUM=0
run "quantum 1600 visits" 16 768 1600 2 A=1
run "quantum 1600 visits" 16 1024 1600 2 A=1
timeout 100 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2> $OUT/bench.err | grep -o '"value": [0-9.]*' | tee $OUT/bench.txt

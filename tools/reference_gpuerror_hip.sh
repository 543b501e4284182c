#!/bin/bash
# The reference's cross-backend acceptance test `testgpuerror` (command/gputest.cpp, tests/testnnevalcanary.cpp:573-829) on
# the katamx backend, against the reference file written by the oracle (= the role of the Eigen build). The test also
# builds an "fp32" evaluator; katamx has no fp32 arithmetic, so katamxPrecision pins BOTH evaluators to the 16-bit mode
# under test: the lines that matter are "... error vs reference" with the reference's REDUCED-precision limits
# (p99 2.0 % winrate / 1.0 lead-score / 2.5 % top policy / 0.002 KL, max 5 / 3 / 6 / 0.004); the strict fp32-limit lines
# are expected to exceed and the exit code to be 1.  Output -> gpurun_out/gpuerror_hip/
set -u
cd "$(dirname "$0")/.."
REPO=$(pwd)
OUT=$REPO/gpurun_out/gpuerror_hip
mkdir -p $OUT
G170=$REPO/oracle/_ref/models/g170-b6c96-s175395328-d26788732.bin.gz
cat > $OUT/bench.cfg <<CFG
logDir = $OUT/logs
logAllGTPCommunication = false
logSearchInfo = false
logToStderr = false
rules = tromp-taylor
maxVisits = 200
numSearchThreads = 16
nnCacheSizePowerOfTwo = 18
nnMutexPoolSizePowerOfTwo = 14
nnRandomize = true
CFG
cd $OUT
for SIZE in 9 19; do
  timeout 900 $REPO/oracle/_ref/katago_oracle testgpuerror -model $G170 -config bench.cfg -boardsize $SIZE -quick -reference-file $OUT/ref_$SIZE.txt > /dev/null 2>&1
  for PREC in bf16 fp16; do
    timeout 600 $REPO/integration/_build/katago_hip testgpuerror -model $G170 -config bench.cfg -boardsize $SIZE -quick -reference-file $OUT/ref_$SIZE.txt -override-config katamxPrecision=$PREC > $OUT/hip_${PREC}_$SIZE.log 2>&1
    echo "== $PREC ${SIZE}x$SIZE rc=$?"
    grep -E "batched current error vs reference (winrateError|leadError|scoreMeanError|topPolicyDelta|policyKLDiv|ownershipError|closest)" $OUT/hip_${PREC}_$SIZE.log
  done
  rm -f $OUT/ref_$SIZE.txt
done

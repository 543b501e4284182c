#!/bin/bash
# The reference's OWN GPU backend (OpenCL, built from its sources by `make -C oracle ref` -> oracle/_ref/katago_opencl)
# on the MI355X, next to katamx:
#  (1) parity of the oracle against it: katago_oracle (in the role of the Eigen build) writes `testgpuerror`'s reference
#      file on the CPU; katago_opencl checks its fp32 / fp16, batched / unbatched outputs against that file with the
#      reference's own cross-backend thresholds (tests/testnnevalcanary.cpp:573-829) — real trained net, real positions;
#  (2) the reference's performance on this hardware: its `benchmark` command on b18c384nbt 19x19 (random weights),
#      same command for katago_hip.
# Results -> gpurun_out/opencl_reference/. The OpenCL autotuner runs first (minutes); its files are kept there too.
set -u
cd "$(dirname "$0")/.."
REPO=$(pwd)
OUT=$REPO/gpurun_out/opencl_reference
mkdir -p $OUT/home
G170=$REPO/oracle/_ref/models/g170-b6c96-s175395328-d26788732.bin.gz
cat > $OUT/bench.cfg <<CFG
logDir = $OUT/logs
logAllGTPCommunication = false
logSearchInfo = false
logToStderr = false
rules = tromp-taylor
allowResignation = false
maxVisits = 200
numSearchThreads = 16
nnCacheSizePowerOfTwo = 18
nnMutexPoolSizePowerOfTwo = 14
nnRandomize = true
ponderingEnabled = false
lagBuffer = 1.0
searchFactorAfterOnePass = 0.5
searchFactorAfterTwoPass = 0.25
searchFactorWhenWinning = 0.4
searchFactorWhenWinningThreshold = 0.95
homeDataDir = $OUT/home
CFG
cd $OUT
for SIZE in 9 19; do
  timeout 900 $REPO/oracle/_ref/katago_oracle testgpuerror -model $G170 -config bench.cfg -boardsize $SIZE -quick -reference-file $OUT/ref_g170_$SIZE.txt > $OUT/oracle_write_$SIZE.log 2>&1
  echo "oracle write $SIZE rc=$?"
  timeout 1200 $REPO/oracle/_ref/katago_opencl testgpuerror -model $G170 -config bench.cfg -boardsize $SIZE -quick -reference-file $OUT/ref_g170_$SIZE.txt > $OUT/opencl_vs_oracle_$SIZE.log 2>&1
  echo "opencl vs oracle $SIZE rc=$?"
  grep -E "vs reference (winrateError|topPolicyDelta|policyKLDiv|closest margin)|ERROR|exceed" $OUT/opencl_vs_oracle_$SIZE.log | head -20
  rm -f $OUT/ref_g170_$SIZE.txt
done
python - <<PY
import sys
sys.path.insert(0, "$REPO")
from katago_amd import modelgen
modelgen.write_model("$OUT/b18rand.bin.gz", "b18c384nbt", seed=7)
PY
for BIN in oracle/_ref/katago_opencl integration/_build/katago_hip; do
  timeout 1500 $REPO/$BIN benchmark -model $OUT/b18rand.bin.gz -config bench.cfg -v 1600 -t 64,256 -boardsize 19 -n 3 > $OUT/benchmark_b18_$(basename $BIN).log 2>&1
  echo "$BIN benchmark rc=$?"
  tr '\r' '\n' < $OUT/benchmark_b18_$(basename $BIN).log | grep "nnEvals/s" | tail -2
done
rm -f $OUT/b18rand.bin.gz

#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/call4
mkdir -p "$OUT"
python -m pytest tests/test_gpu_transformer.py tests/test_gpu_search_fixed_seed.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40 > "$OUT/tf_search.log"; tail -12 "$OUT/tf_search.log"
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 300 python3 bench.py --no-cpu-baseline --no-profile "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*')
  echo "$name | $v" | tee -a "$OUT/scan.txt"; }
b "cfg23 ways1" KMX_SPLIT_MIN=0 -- --steps 50 --warmup 5
for w in 1 2 3 4; do
  if [ $w = 1 ]; then b "4-wave shapes ways1" KMX_MIN_WGS8=100000 KMX_SPLIT_MIN=0 -- --steps 50 --warmup 5
  else b "4-wave shapes ways$w" KMX_MIN_WGS8=100000 KMX_SPLIT_WAYS=$w -- --steps 50 --warmup 5; fi
done
python - <<'PY' 2>&1 | tee "$OUT/ref_benchmark_visits.txt"
import os, sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_reference_harness as h
from katago_amd import modelgen
tmp = tempfile.mkdtemp()
model = os.path.join(tmp, "b18.bin.gz"); modelgen.write_model(model, "b18c384nbt", seed=7)
for threads, visits, t in ((1, 1600, "256"), (1, 16000, "256"), (2, 16000, "256"), (2, 16000, "512"), (1, 16000, "128")):
    cfg = os.path.join(tmp, "bench%d.cfg" % threads)
    open(cfg, "w").write(h.BENCH_CFG + "numNNServerThreadsPerModel = %d\nnnMaxBatchSize = 256\n" % threads)
    rc, out = h.run("benchmark", "-model", model, "-config", cfg, "-v", str(visits), "-t", t, "-boardsize", "19", "-n", "3", timeout=900)
    for l in out.replace("\r", "\n").splitlines():
        if "nnEvals/s" in l: print("serverThreads", threads, "visits", visits, l.strip())
PY

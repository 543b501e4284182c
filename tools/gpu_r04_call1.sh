#!/bin/bash
# Round 4, first GPU call: the new parity tests, the two parked small-batch experiments, the driver's bench line (small-batch leg with
# warm-up + median), a kernel trace of a batch-8 pass (launch-bound or kernel-bound?), mid-batch shape scan, and self-play probes at
# the reference's production settings (tools/selfplay_cfg.py). Everything under gpurun_out/r4c1.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c1
rm -rf $OUT; mkdir -p $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^$\|====" > $OUT/smi.txt
nproc > $OUT/host.txt; grep -c processor /proc/cpuinfo >> $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null

timeout 900 python -m pytest "tests/test_gpu_model.py::test_full_batch_properties" "tests/test_gpu_model.py::test_headline_batch_vs_oracle_default_precision" \
  "tests/test_gpu_selfplay.py::test_mixed_board_sizes_b18_own_evaluator_writes_valid_shards" "tests/test_gpu_selfplay.py::test_selfplay_writes_shards_on_hip" \
  tests/test_gpu_reference_harness.py -k "not opencl" -m gpu -q -x -s -p no:cacheprovider > $OUT/pytest_new.log 2>&1
tail -5 $OUT/pytest_new.log
cp gpurun_out/selfplay_mixed_sizes_b18.txt $OUT/ 2>/dev/null

# small batches: default, read-ahead twelve-wave shape, small fused seams, both
for v in "A=1" "KMX_CONV_CW12_AHEAD=1" "KMX_FUSE_SMALL_ROWS=1" "KMX_CONV_CW12_AHEAD=1 KMX_FUSE_SMALL_ROWS=1"; do
  echo "== $v" >> $OUT/small_batch_scan.txt
  env $v timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN >> $OUT/small_batch_scan.txt
done
cat $OUT/small_batch_scan.txt | cut -c1-900

# is a small pass bound by its launches or by its kernels? sum of kernel durations against the wall time of the passes
cat > $OUT/pass8.py <<'PY'
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from katago_amd import modelgen, nninterface as nn
from conftest import make_rows
nn.globalInitialize()
p = "/tmp/kmx_scan_b18.bin"
if not os.path.exists(p): modelgen.write_model(p, "b18c384nbt", seed=5)
h = nn.createComputeHandle(nn.createComputeContext([0], 19, 19), nn.loadModelFile(p), 64)
sp, gl = make_rows(np.random.default_rng(3), 8)
sym = np.zeros(8, np.int32)
for _ in range(10): nn.getOutput(h, sp, gl, sym)
t0 = time.perf_counter()
for _ in range(50): nn.getOutput(h, sp, gl, sym)
print("WALL_MS_PER_PASS %.4f" % ((time.perf_counter() - t0) / 50 * 1e3))
PY
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_pass8 -o pass8 -- python $OUT/pass8.py > $OUT/trace_pass8.log 2>&1
grep WALL $OUT/trace_pass8.log
timeout 100 python tools/rocpd_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
ls $OUT/summary 2>/dev/null | head

# mid batches: which shape
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --no-profile "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/mid_batch_scan.txt"; }
for n in 64 96 128 192; do
  b "batch $n default" A=1 -- --batch $n --steps 40 --warmup 5
  b "batch $n KMX_MIN_WGS8=60" KMX_MIN_WGS8=60 -- --batch $n --steps 40 --warmup 5
done

# the driver's command
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c1/bench.json").read().strip().splitlines()[-1])
print("BENCH value", d["value"], "frac", d["roofline"]["frac"], "small", d.get("small_batches"), "selfplay", d.get("selfplay_nn_rows_per_s"), "refbench", d.get("reference_benchmark_nn_evals_per_s"))
PY

# self-play at the reference's production settings, b18c384nbt 19x19
python - <<'PY'
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import selfplay_cfg
from katago_amd import modelgen
d = "/tmp/sp_main"; os.makedirs(d + "/models", exist_ok=True)
modelgen.write_model(d + "/models/b18c384nbt-s1-d1.bin.gz", "b18c384nbt", seed=7)
selfplay_cfg.write(d + "/g8.cfg", numGameThreads=8, numSearchThreads=8, nnMaxBatchSize=64, logGamesEvery=1, switchNetsMidGame="false", nnCacheSizePowerOfTwo=21, nnMutexPoolSizePowerOfTwo=16, **selfplay_cfg.ONLY_19)
selfplay_cfg.write(d + "/g100.cfg", numGameThreads=100, numSearchThreads=1, nnMaxBatchSize=192, logGamesEvery=10, switchNetsMidGame="false", nnCacheSizePowerOfTwo=21, nnMutexPoolSizePowerOfTwo=16, **selfplay_cfg.ONLY_19)
selfplay_cfg.write(d + "/g100x4.cfg", numGameThreads=100, numSearchThreads=4, nnMaxBatchSize=256, logGamesEvery=10, switchNetsMidGame="false", nnCacheSizePowerOfTwo=21, nnMutexPoolSizePowerOfTwo=16, **selfplay_cfg.ONLY_19)
PY
sp() { local name=$1 cfg=$2 secs=$3; shift 3
  rm -rf /tmp/sp_main/out
  ( cd /tmp/sp_main && env "$@" timeout -s INT $secs $OLDPWD/oracle/_ref/katago_hip selfplay -config $cfg -models-dir models -output-dir out > $OLDPWD/$OUT/selfplay_$name.log 2>&1 )
  python - "$OUT/selfplay_$name.log" "$name" <<'PY' | tee -a $OUT/selfplay_probe.txt
import re, sys
t = open(sys.argv[1]).read()
g = lambda k: (re.findall(k + r": ([\d.]+)", t) or ["?"])[-1]
secs = g("Total selfplay runtime \(seconds\)")
rows, moves, games, batches = g("Final NN rows"), g("Final moves played"), g("Final games finished"), g("Final NN batches")
try:
    print("%s: %s s, games finished %s, moves %s, NN rows %s = %.0f rows/s, %.1f moves/s, avg batch %.1f, cache hits %s" % (sys.argv[2], secs, games, moves, rows, float(rows) / float(secs), float(moves) / float(secs), float(rows) / float(batches), g("Final NN cache hits")))
except Exception as e:
    print(sys.argv[2], "unparsed", e, t[-600:])
PY
}
sp g8_k8 g8.cfg 80 KATAMX_LEAVES_PER_THREAD=8
sp g8_k4 g8.cfg 40 KATAMX_LEAVES_PER_THREAD=4

sp g100 g100.cfg 45 A=1
sp g100x4 g100x4.cfg 45 KATAMX_LEAVES_PER_THREAD=4
grep -c "Started\|Games finished" $OUT/selfplay_g8_k8.log | head -2
grep "Moves played\|Games finished" $OUT/selfplay_g8_k8.log | tail -4

#!/bin/bash
# The first hour on an 8 x MI355X node (no such node was available in rounds 1-6: nothing below has run on more than
# one real device - the in-process host stack has, on eight FAKE devices: tests/test_schedule_dryrun.py::test_eight_devices_selfplay_in_one_process).
#
#   tools/scale_day_one.sh [out dir = gpurun_out/scale_day_one] [seconds per self-play window = 120]
#
#  1. NN evals/s at 1 / 2 / 4 / 8 GPUs: the driver's own command (bench.py --gpus N, one process per GPU over gloo, replicas: DESIGN.md 6).
#  2. Self-play on all GPUs in ONE process (the reference's multi-GPU mode: one evaluator, a leaf port per device; tools/selfplay_8gpu.sh),
#     BASELINE configs[2] - 8 games per GPU x 8 leaves - under BOTH port policies (KATAMX_PORT_POLICY = spread: a row goes to the device with the
#     fewest rows in flight; fill: to the first device below KATAMX_PORT_FILL_ROWS), then 32 games per GPU; each for a fixed window, interrupted with
#     SIGINT (the reference then writes its totals): NN rows/s, average device batch, rows per device.
# Prints one line per measurement and keeps the logs.
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
OUT=${1:-gpurun_out/scale_day_one}; WINDOW=${2:-120}
NGPU=${KMX_NUM_GPUS:-$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  [ "$n" -gt "$NGPU" ] && break
  if [ "$n" = 1 ]; then cmd="python bench.py --gpus 1 --steps 20 --warmup 5 --no-callers"
  else cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 20 --warmup 5 --no-callers"; fi
  KMX_BENCH_SELFPLAY_TIMEOUT=0 timeout 900 $cmd > "$OUT/bench_n$n.json" 2> "$OUT/bench_n$n.err"
  echo "evals/s at $n GPU(s): $(grep -o '"value": [0-9.]*' "$OUT/bench_n$n.json" | head -1) $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_n$n.json" | head -1)" | tee -a "$OUT/summary.txt"
done
D=$(mktemp -d /tmp/scale_day_one.XXXXXX); mkdir -p "$D/models"
python3 -c "import sys; sys.path.insert(0, '.'); from katago_amd import modelgen; modelgen.write_model('$D/models/b18c384nbt-s1-d1.bin.gz', 'b18c384nbt', seed=7)"
window() { # window <tag> <games per GPU> <leaves> [ENV=...]
  local tag=$1 games=$2 leaves=$3; shift 3
  rm -rf "$D/out_$tag"
  ( env "$@" KMX_NUM_GPUS=$NGPU KMX_BATCH_TRACE=0 KMX_LAUNCH_PREFIX="timeout -s INT $WINDOW" tools/selfplay_8gpu.sh "$D/models" "$D/out_$tag" $games $leaves \
      logGamesEvery=1000 switchNetsMidGame=false nnCacheSizePowerOfTwo=23 nnMutexPoolSizePowerOfTwo=17 bSizes=19 bSizeRelProbs=1 allowRectangleProb=0.0 ) > "$OUT/selfplay_$tag.log" 2>&1
  python3 - "$OUT/selfplay_$tag.log" "$tag" "$NGPU" "$games" "$leaves" <<'PY' | tee -a "$OUT/summary.txt"
import re, sys
t = open(sys.argv[1]).read()
g = lambda k: float((re.findall(k + r": ([\d.]+)", t) or ["nan"])[-1])
secs, rows, batches, fin = g(r"Total selfplay runtime \(seconds\)"), g("Final NN rows"), g("Final NN batches"), g("Final games finished")
fault = re.findall(r"HSA_STATUS[A-Z_]*|Memory access fault[^\n]*", t)[:1]
print("self-play %s: %s GPUs x %s games x %s leaves: %.0f NN rows/s, average device batch %.1f, %d games finished in %.0f s %s"
      % (sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], rows / secs, rows / max(batches, 1), fin, secs, fault or ""))
PY
}
window spread_8x8 8 8 KATAMX_PORT_POLICY=spread
window fill_8x8 8 8 KATAMX_PORT_POLICY=fill
window spread_32x8 32 8 KATAMX_PORT_POLICY=spread
window fill_32x8 32 8 KATAMX_PORT_POLICY=fill
rm -rf "$D"
echo "summary: $OUT/summary.txt"

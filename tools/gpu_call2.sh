#!/bin/bash
# stress of the driver's command + the new bench-path tests + first hardware run of the transformer path
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/call2
mkdir -p "$OUT"
fails=0
for i in $(seq 1 30); do
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/stress_$i.out" 2> "$OUT/stress_$i.err"
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "stress $i rc=$rc"; tail -3 "$OUT/stress_$i.err"; else rm -f "$OUT/stress_$i.err"; fi
done
echo "stress: $fails failures of 30"
grep -h -o '"value": [0-9.]*' $OUT/stress_*.out | sort | uniq -c | head -40
python -m pytest tests/test_gpu_bench_command.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -25 > "$OUT/bench_tests.log"; tail -25 "$OUT/bench_tests.log"
bash tools/verify_transformer_gpu.sh

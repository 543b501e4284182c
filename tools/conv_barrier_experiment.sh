#!/bin/bash
# Second GPU call of the next round: the even-tap-barrier kernels and the ring-depth-4 puzzle.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/conv_barrier_experiment.sh'
# 1. parity of the experiment kernels (KMX_CONV_BP2=1 makes the product pick them for the 8-wave 3x3 shapes): layer tests +
#    whole nets at the batch sizes where those shapes are chosen;
# 2. timing, 3x3 192->192 at batch 256, epilogue mode 0 and 1 (second argument 23 = 8-wave x 192 channels):
#    product (variant 0) | ring depth 4 | even-tap barriers with ring depth 3 and 4 | the same with cycle stamps.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/conv_bp2
mkdir -p "$OUT"
KMX_CONV_BP2=1 python -m pytest tests/test_gpu_layers.py tests/test_gpu_model.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > "$OUT/parity_bp2.log"
for mode in 0 1; do
  for var in 0 3000 4000 7096 8096; do
    python tools/conv_one.py 3 23 $var 192 192 $mode 50 2>&1 | tail -1
  done
done > "$OUT/timing.log" 2>&1
for var in 5048 6048 9144 10144; do   # 3000+2048, 4000+2048, 3000+6144, 4000+6144: per-wave cycle accounting on stderr
  python tools/conv_one.py 3 23 $var 192 192 1 20
done > "$OUT/cycles.log" 2>&1
tail -4 "$OUT/parity_bp2.log"; cat "$OUT/timing.log"; grep -c timing "$OUT/cycles.log"

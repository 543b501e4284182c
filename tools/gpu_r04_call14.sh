#!/bin/bash
# the small-batch 3x3 shape with a board's cell tiles over three work-groups (cfg 117): parity on hardware, then the small-batch scan with
# and without it (and with / without the deep 1x1 ring) on the same box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c14; rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_layers.py "tests/test_gpu_model.py::test_model_vs_oracle" tests/test_gpu_fuzz.py tests/test_gpu_model.py::test_headline_batch_vs_oracle_default_precision -m gpu -q -x -p no:cacheprovider > $OUT/pytest_split.log 2>&1
tail -2 $OUT/pytest_split.log
for v in "KMX_CONV_LOADERS_SPLIT=0 KMX_CONV_DEEP1X1=0" "KMX_CONV_LOADERS_SPLIT=1 KMX_CONV_DEEP1X1=0" "KMX_CONV_LOADERS_SPLIT=1 KMX_CONV_DEEP1X1=4" "KMX_CONV_LOADERS_SPLIT=0 KMX_CONV_DEEP1X1=0" "KMX_CONV_LOADERS_SPLIT=1 KMX_CONV_DEEP1X1=0"; do
  echo "== $v" >> $OUT/small_batch_scan.txt
  env $v timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN >> $OUT/small_batch_scan.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r4c14/small_batch_scan.txt"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l.split("SCAN ")[1])
    print("  ms", d["ms_per_pass"], "same", d["rows_bit_identical_across_batch_sizes"], d["digest"][:10])
    print("  b1", d["us_per_launch_batch_1"]); print("  b8", d["us_per_launch_batch_8"])
PY

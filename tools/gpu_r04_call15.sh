#!/bin/bash
# 1x1 layers with a board's cell tiles over three work-groups (cfg 113) and the 3x3 split shape two per CU (cfg 116): parity on hardware,
# then the small-batch scan per setting on one box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r4c15; rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_layers.py "tests/test_gpu_model.py::test_model_vs_oracle" tests/test_gpu_fuzz.py tests/test_gpu_model.py::test_headline_batch_vs_oracle_default_precision -m gpu -q -x -p no:cacheprovider > $OUT/pytest_split2.log 2>&1
tail -2 $OUT/pytest_split2.log
for v in "KMX_CONV_SPLIT1X1=0 KMX_CONV_LOADERS_SPLIT_PACKED_MAX_WGS=0" "KMX_CONV_SPLIT1X1=1 KMX_CONV_LOADERS_SPLIT_PACKED_MAX_WGS=0" "KMX_CONV_SPLIT1X1=1 KMX_CONV_LOADERS_SPLIT_PACKED_MAX_WGS=512" "KMX_CONV_SPLIT1X1=0 KMX_CONV_LOADERS_SPLIT_PACKED_MAX_WGS=0" "KMX_CONV_SPLIT1X1=1 KMX_CONV_LOADERS_SPLIT_PACKED_MAX_WGS=512"; do
  echo "== $v" >> $OUT/small_batch_scan.txt
  env $v timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN >> $OUT/small_batch_scan.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r4c15/small_batch_scan.txt"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l.split("SCAN ")[1])
    print("  ms", d["ms_per_pass"], "same", d["rows_bit_identical_across_batch_sizes"], d["digest"][:10])
    print("  b1", d["us_per_launch_batch_1"]); print("  b8", d["us_per_launch_batch_8"])
PY
for b in 20 28; do for m in 0 512; do
  KMX_CONV_LOADERS_SPLIT_PACKED_MAX_WGS=$m timeout 120 python bench.py --batch $b --no-cpu-baseline --no-callers --no-profile --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b split-packed<=$m', d['value'], d['ms_per_step'])" | tee -a $OUT/mid_batches.txt
done; done

#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c3
rm -rf $OUT; mkdir -p $OUT
timeout 120 python tools/seam_timing.py 256 > $OUT/seam_v2.txt 2>&1
KMX_PW_V2=0 timeout 120 python tools/seam_timing.py 256 > $OUT/seam_v1.txt 2>&1
timeout 120 python tools/seam_timing.py 64 > $OUT/seam_v2_b64.txt 2>&1
cat $OUT/seam_v2.txt $OUT/seam_v1.txt $OUT/seam_v2_b64.txt

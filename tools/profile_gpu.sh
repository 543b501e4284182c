#!/bin/bash
# rocprofv3 evidence for profiles/: (1) kernel-trace stats of the bench command, (2) PMC passes on the dominant
# kernel (3x3 192->192, batch 256). Counters are collected in their own runs (no trace domains combined with --pmc).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass -d $OUT/pmc_$tag -o conv -- python tools/conv_one.py 3 3 0 192 192 1 3 > $OUT/pmc_$tag.log 2>&1
done
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
find $OUT -name "*.csv" | head -50 > $OUT/files.txt

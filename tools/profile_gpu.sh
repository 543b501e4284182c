#!/bin/bash
# rocprofv3 evidence for profiles/: (1) kernel-trace stats of the bench command, (2) PMC passes over the same bench
# command (HBM traffic of the dominant kernel averaged over its launches) and over one launch shape of the dominant
# kernel (3x3 192->192 with residual, batch 256). Counters are collected in their own runs, never combined with a
# trace domain.   usage: tools/profile_gpu.sh [outdir under gpurun_out]
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# kernels are profiled with the chip to themselves (one stream), which is what bench.py's roofline pass measures
export KMX_SPLIT_MIN=0
OUT=gpurun_out/${1:-prof}
rm -rf $OUT; mkdir -p $OUT
BENCH="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass -d $OUT/benchpmc_$tag -o bench -- $BENCH > $OUT/benchpmc_$tag.log 2>&1
done
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass -d $OUT/pmc_$tag -o conv -- python tools/conv_one.py 3 23 0 192 192 1 3 > $OUT/pmc_$tag.log 2>&1
done
KMX_SPLIT_MIN=224 rocprofv3 --kernel-trace --stats -d $OUT/bench_trace_two_streams -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $OUT/bench_trace_two_streams.log 2>&1
python tools/rocpd_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
python tools/make_traffic.py $OUT/summary > $OUT/make_traffic.log 2>&1

#!/bin/bash
# Round 3, call 1: baseline of this box + diagnostics that decide the kernel work of the round.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r3c1
rm -rf $OUT; mkdir -p $OUT
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
tail -c 400 $OUT/bench_driver_cmd.json
timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_100.json 2>> $OUT/bench_driver_cmd.err
timeout 300 python tools/conv_streams.py 40 > $OUT/conv_streams.txt 2>&1
for n in 1 8 32; do
  timeout 120 python bench.py --batch $n --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_batch$n.json 2>> $OUT/scan.err
done
rocprofv3 -L > $OUT/counters.txt 2>&1
for var in 0 4000; do
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
              "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
              "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    timeout 100 rocprofv3 --pmc $pass -d $OUT/pmc_v${var}_$tag -o conv -- python tools/conv_one.py 3 23 $var 192 192 1 3 > $OUT/pmc_v${var}_$tag.log 2>&1
  done
done
python tools/rocpd_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
ls $OUT | head -50
# seam kernel: persistent work-groups and a start offset for odd ones (pointwise_kernel.h EXPERIMENT)
s() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env KMX_SPLIT_MIN=0 "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2>>"$OUT/seam.err" | grep -o '"value": [0-9.]*\|"kernel_avg_launch_us": {[^}]*}' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/seam.txt"; }
s "pw8 default" -- 
s "pw8 persist256" KMX_PW_PERSIST=256 --
s "pw8 persist256 delay15k" KMX_PW_PERSIST=256 KMX_PW_DELAY=15000 --
s "pw8 persist256 delay30k" KMX_PW_PERSIST=256 KMX_PW_DELAY=30000 --
s "pw4 default" KMX_PW_WAVES=4 --
s "pw4 persist512" KMX_PW_WAVES=4 KMX_PW_PERSIST=512 --
s "pw4 persist512 delay8k" KMX_PW_WAVES=4 KMX_PW_PERSIST=512 KMX_PW_DELAY=8000 --
s "pw4 persist512 delay15k" KMX_PW_WAVES=4 KMX_PW_PERSIST=512 KMX_PW_DELAY=15000 --
s "pw4 persist512 delay25k" KMX_PW_WAVES=4 KMX_PW_PERSIST=512 KMX_PW_DELAY=25000 --
timeout 200 python -m pytest tests/test_gpu_pointwise.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity_pw.log
KMX_PW_WAVES=4 KMX_PW_PERSIST=512 KMX_PW_DELAY=15000 timeout 200 python -m pytest tests/test_gpu_pointwise.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity_pw4p.log

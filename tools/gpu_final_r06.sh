#!/bin/bash
# Round-6 evidence run at HEAD, most important first: the driver's bench command, rocprofv3 kernel-trace stats + a PMC pass of the bench command,
# a small pass under the kernel trace, the small-batch scan, the 8-GPU day-one script on the one GPU there is, two more runs of 8 full-length
# games. (The full -m gpu suite is `tools/gpu_r06.sh suite`: 260 tests, 18 minutes.) Everything under gpurun_out/final_r06; summaries are copied
# to profiles/r06_final.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/final_r06
rm -rf $OUT; mkdir -p $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^$\|====" > $OUT/smi.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
export KMX_SPLIT_MIN=0   # kernels are profiled with the chip to themselves (one stream), as bench.py's roofline pass measures them
BENCH="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-callers --no-pmc"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-callers --no-pmc > $OUT/bench_trace.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $pass -d $OUT/benchpmc_$tag -o bench -- $BENCH > $OUT/benchpmc_$tag.log 2>&1
done
unset KMX_SPLIT_MIN
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_batch32 -o bench -- python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-callers --no-profile --no-pmc > $OUT/trace_batch32.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_batch136 -o bench -- python bench.py --batch 136 --steps 20 --warmup 5 --no-cpu-baseline --no-callers --no-profile --no-pmc > $OUT/trace_batch136.log 2>&1
timeout 100 python tools/rocpd_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
cp -r gpurun_out/bench_pmc/summary $OUT/bench_pmc_summary 2>/dev/null
timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN > $OUT/small_batch_scan.txt
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers --no-pmc "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"dtype": "[a-z0-9]*"' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/scan.txt"; }
for n in 8 32 64 96 128 144 192; do b "b18c384nbt default precision batch $n" A=1 -- --batch $n --steps 40 --warmup 5 --no-profile; done
b "b18c384nbt bf16 batch 256" A=1 -- --dtype bf16 --steps 40 --warmup 5
b "b28c512nbt default batch 512" A=1 -- --model b28c512nbt --batch 512 --steps 10 --warmup 2
KMX_NUM_GPUS=1 timeout 600 tools/scale_day_one.sh $OUT/scale_day_one_on_one_gpu 45 > $OUT/scale_day_one.log 2>&1
cat $OUT/scale_day_one_on_one_gpu/summary.txt
for run in 2 3; do
  if [ $((SECONDS + 420)) -lt ${KMX_FINAL_BUDGET_S:-2400} ]; then
    tools/selfplay_full_games.sh games_run$run 8 8 8 8 400 > /dev/null 2>&1
    cat gpurun_out/selfplay_full_games_run$run.txt | tee -a $OUT/games_per_hour_more_runs.txt
  else
    echo "run $run not started: $SECONDS s of the call's allowance used" | tee -a $OUT/games_per_hour_more_runs.txt
  fi
done
echo "script time: $SECONDS s" | tee $OUT/script_seconds.txt

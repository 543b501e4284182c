#!/bin/bash
# Round-5 evidence run at HEAD, most important first (a call that runs out of its allowance loses only the tail): the full -m gpu suite, the
# driver's bench command, rocprofv3 kernel-trace stats + a PMC pass of the bench command, the bench command with --pmc, the launch floor and the
# small shapes' cycle stamps, small-batch scans with and without the register-weights shapes, a batch scan, two more full-game self-play runs
# (with bench.py's own: three). Everything under gpurun_out/final_r05 (summaries are copied to profiles/r05_final).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/final_r05
rm -rf $OUT; mkdir -p $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^$\|====" > $OUT/smi.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
for f in search_driven_rate.txt selfplay_rate_b18.txt selfplay_rate_b18_own_evaluator.txt selfplay_mixed_sizes_b18.txt analysis_engine_b28.txt leaf_pump_b18.txt \
         reference_benchmark_b18_19x19.txt reference_benchmark_b6c96_9x9.txt reference_benchmark_batcher.txt testgpuerror_g170_auto.txt testgpuerror_g170_bf16.txt \
         testgpuerror_g170_fp16.txt testgpuerror_g170_fp32_evaluator.txt search_fixed_seed_auto.txt search_fixed_seed_bf16.txt search_fixed_seed_fp16.txt search_fixed_seed_fp32.txt; do
  cp gpurun_out/$f $OUT/ 2>/dev/null; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
export KMX_SPLIT_MIN=0   # kernels are profiled with the chip to themselves (one stream), as bench.py's roofline pass measures them
BENCH="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-callers"
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-callers > $OUT/bench_trace.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 150 rocprofv3 --pmc $pass -d $OUT/benchpmc_$tag -o bench -- $BENCH > $OUT/benchpmc_$tag.log 2>&1
done
unset KMX_SPLIT_MIN
# a small pass (batch 32, where self-play sits) under the kernel trace: the register-weights 3x3 shape, the 1x1 and small kernels per launch
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/trace_batch32 -o bench -- python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-callers --no-profile > $OUT/trace_batch32.log 2>&1
timeout 100 python tools/rocpd_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
KMX_BENCH_SELFPLAY_TIMEOUT=0 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc > $OUT/bench_pmc.json 2> $OUT/bench_pmc.err
cp -r gpurun_out/bench_pmc/summary $OUT/bench_pmc_summary 2>/dev/null
timeout 60 python tools/launch_floor.py > $OUT/launch_floor.txt 2>&1
timeout 100 python tools/small_conv_timing.py > $OUT/small_conv_timing.txt 2>&1
for k in 3 2; do
  echo "== KMX_PW_KERNEL=$k fp16" >> $OUT/seam_timing.log
  KMX_PW_KERNEL=$k KMX_BENCH_DTYPE=fp16 timeout 120 python tools/seam_timing.py 256 >> $OUT/seam_timing.log 2>&1
done
timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN > $OUT/small_batch_scan.txt
KMX_CONV_TUNE=regw=0 timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN >> $OUT/small_batch_scan.txt
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"dtype": "[a-z0-9]*"' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/scan.txt"; }
[ $SECONDS -lt $(( ${KMX_FINAL_BUDGET_S:-2160} - 300 )) ] && for n in 8 32 64 128; do b "b18c384nbt default precision batch $n" A=1 -- --batch $n --steps 40 --warmup 5 --no-profile; done
b "b18c384nbt default precision batch 32, round 4's small shapes (KMX_CONV_TUNE=regw=0)" KMX_CONV_TUNE=regw=0 -- --batch 32 --steps 40 --warmup 5 --no-profile
b "b18c384nbt default precision batch 64, round 4's small shapes (KMX_CONV_TUNE=regw=0)" KMX_CONV_TUNE=regw=0 -- --batch 64 --steps 40 --warmup 5 --no-profile
b "b18c384nbt bf16 batch 256" A=1 -- --dtype bf16 --steps 40 --warmup 5
b "b28c512nbt default batch 512" A=1 -- --model b28c512nbt --batch 512 --steps 10 --warmup 2
b "b40c256 default batch 512" A=1 -- --model b40c256 --batch 512 --steps 10 --warmup 2
# configs[2], games/hour as command/selfplay.cpp:388-389 defines it: two more runs of 8 full-length games (bench.json above holds the first)
# (only while the call's allowance lasts: KMX_FINAL_BUDGET_S seconds for the whole script, default 36 minutes)
for run in 2 3; do
  if [ $((SECONDS + 420)) -lt ${KMX_FINAL_BUDGET_S:-2160} ]; then
    tools/selfplay_full_games.sh games_run$run 8 8 8 8 400 > /dev/null 2>&1
    cat gpurun_out/selfplay_full_games_run$run.txt | tee -a $OUT/games_per_hour_more_runs.txt
  else
    echo "run $run not started: $SECONDS s of the call's allowance used" | tee -a $OUT/games_per_hour_more_runs.txt
  fi
done
echo "script time: $SECONDS s" | tee $OUT/script_seconds.txt

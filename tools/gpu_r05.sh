#!/bin/bash
# Round-5 GPU calls, by section (one gpurun call runs one or more sections; everything lands under gpurun_out/r05/<section>):
#   tools/gpu_r05.sh seam        the resident-weights seam kernel: parity on hardware, launch time and cycle stamps against round 3's kernel, whole-net A/B
#   tools/gpu_r05.sh parity      the round's changed tests (configs[3] in bf16 + fp16, fixed-seed search with the flip list, bench command)
#   tools/gpu_r05.sh small       small / mid batch scan
#   tools/gpu_r05.sh regw        the small-batch 3x3 shapes with their weights in registers against the slab-ring shapes
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
b() { local out=$1; shift; local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 200 python3 bench.py --no-cpu-baseline --no-callers "$@" 2>>"$out/err.txt" | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$out/scan.txt"; }
for section in "$@"; do
OUT=gpurun_out/r05/$section
rm -rf $OUT; mkdir -p $OUT
case $section in
seam)
  timeout 600 python -m pytest tests/test_gpu_pointwise.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity.log
  for k in 3 2; do for dt in fp16 bf16; do
    echo "== KMX_PW_KERNEL=$k $dt" | tee -a $OUT/seam_timing.txt
    KMX_PW_KERNEL=$k KMX_BENCH_DTYPE=$dt timeout 120 python tools/seam_timing.py 256 2>&1 | tee -a $OUT/seam_timing.txt
    KMX_PW_KERNEL=$k KMX_BENCH_DTYPE=$dt timeout 120 python tools/seam_timing.py 128 2>&1 | grep -v "timing\]" | tee -a $OUT/seam_timing.txt
  done; done
  for rep in 1 2; do
    b $OUT "b18 default, seam kernel 3 (resident weights)" KMX_PW_KERNEL=3 -- --steps 40 --warmup 5
    b $OUT "b18 default, seam kernel 2 (round 3)" KMX_PW_KERNEL=2 -- --steps 40 --warmup 5
  done
  b $OUT "b18 one stream, seam kernel 3" KMX_PW_KERNEL=3 KMX_SPLIT_MIN=0 -- --steps 40 --warmup 5 --no-profile
  b $OUT "b18 one stream, seam kernel 2" KMX_PW_KERNEL=2 KMX_SPLIT_MIN=0 -- --steps 40 --warmup 5 --no-profile
  b $OUT "b18 bf16, seam kernel 3" KMX_PW_KERNEL=3 -- --dtype bf16 --steps 40 --warmup 5
  b $OUT "b18 bf16, seam kernel 2" KMX_PW_KERNEL=2 -- --dtype bf16 --steps 40 --warmup 5
  # the kernels' own durations inside the net (rocprofv3 kernel trace, one stream), both seam kernels
  for k in 3 2; do
    KMX_PW_KERNEL=$k KMX_SPLIT_MIN=0 timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/trace_k$k -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-callers --no-profile > $OUT/trace_k$k.log 2>&1
    f=$(find $OUT/trace_k$k -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && { echo "== in-net kernel stats, seam kernel $k"; head -8 "$f" | cut -c1-200; } | tee -a $OUT/in_net_kernel_stats.txt
    [ -n "$f" ] && cp "$f" $OUT/trace_k${k}_kernel_stats.csv
    rm -rf $OUT/trace_k$k
  done
  ;;
parity)
  timeout 1500 python -m pytest "tests/test_gpu_model.py::test_large_nets_of_the_analysis_config" "tests/test_gpu_search_fixed_seed.py" tests/test_gpu_layers.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -60 | tee $OUT/pytest.log
  cp gpurun_out/search_fixed_seed_*.txt $OUT/ 2>/dev/null
  KMX_BENCH_SELFPLAY_TIMEOUT=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
  tail -c 3000 $OUT/bench.json
  ;;
search)
  timeout 900 python -m pytest "tests/test_gpu_search_fixed_seed.py" -m gpu -q -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -30 | tee $OUT/pytest.log
  cp gpurun_out/search_fixed_seed_*.txt gpurun_out/search_fixed_seed_*_output.txt.gz $OUT/ 2>/dev/null
  ;;
fp32)
  timeout 1200 python -m pytest tests/test_gpu_layers.py::test_fp32_mode_layers "tests/test_gpu_model.py::test_model_vs_oracle" tests/test_gpu_reference_harness.py::test_reference_gpuerror_as_the_reference_runs_it \
     tests/test_gpu_reference_harness.py::test_nn_layer_known_answers_on_hip "tests/test_gpu_search_fixed_seed.py" -k "fp32 or gpuerror or known_answers" -m gpu -q -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -40 | tee $OUT/pytest.log
  cp gpurun_out/search_fixed_seed_fp32.txt gpurun_out/testgpuerror_g170_fp32_evaluator.txt $OUT/ 2>/dev/null
  ;;
games)
  # configs[2], games/hour as command/selfplay.cpp:388-389 defines it: three runs of 8 full-length games (8 game threads x 8 search threads on fibers)
  for run in 1 2 3; do
    tools/selfplay_full_games.sh games_run$run 8 8 8 8 400 > /dev/null 2>&1
    cat gpurun_out/selfplay_full_games_run$run.txt | tee -a $OUT/games_per_hour_three_runs.txt
  done
  ;;
carriers)
  # configs[2] (8 games x 8 search threads): how many of a game's 8 descents share an OS thread (KATAMX_LEAVES_PER_THREAD); 50 s each, rows/s only
  for lpt in 8 4 2 1; do
    tools/selfplay_full_games.sh carriers_$lpt 8 8 $lpt 8 50 > /dev/null 2>&1
    cat gpurun_out/selfplay_full_carriers_$lpt.txt | tee -a $OUT/carriers.txt
  done
  ;;
regw)
  # the fetching-waves 3x3 shapes with their weights in registers (conv_small_kernel.h REGW, cfg 128 / 126 / 127 / 125; the default) against
  # round 4's slab-ring shapes (KMX_CONV_TUNE regw=0: cfg 118 / 119 / 117): parity on hardware, pass time at batch 1 .. 85 from host rows with
  # the digest of fixed rows, launch classes, self-play rows/s. (The round's calls 1-3 ran earlier forms of this section: regw=0 / 2 / 3 with the
  # shapes still off by default, then regw_early=0 / 1 - keys that are gone with the forms they selected; profiles/r05_steps/regw/README.md.)
  timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_fuzz.py tests/test_gpu_small_shapes.py "tests/test_gpu_model.py::test_full_batch_properties" -m gpu -q -p no:cacheprovider 2>&1 | tail -40 | cut -c1-600 | tee $OUT/parity.log
  for t in regw=0 regw=3 regw=0 regw=3; do
    KMX_CONV_TUNE=$t timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN | tee -a $OUT/small_batch_scan.txt
  done
  for t in regw=0 regw=3; do for n in 8 32 64; do
    echo "== $t batch $n: per-class launch times (hipEvent pair per launch, one stream)" | tee -a $OUT/kernel_classes.txt
    KMX_CONV_TUNE=$t timeout 200 python3 bench.py --no-cpu-baseline --no-callers --batch $n --steps 30 --warmup 5 2>>$OUT/err.txt | grep -o '"kernel_avg_launch_us": {[^}]*}\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/kernel_classes.txt; echo | tee -a $OUT/kernel_classes.txt
  done; done
  for t in regw=0 regw=3; do
    KMX_CONV_TUNE=$t tools/selfplay_full_games.sh regw_${t#*=} 8 8 8 8 60 > /dev/null 2>&1
    echo "$t: $(cat gpurun_out/selfplay_full_regw_${t#*=}.txt)" | tee -a $OUT/selfplay_rows.txt
  done
  timeout 60 python tools/launch_floor.py | tee $OUT/launch_floor.txt
  timeout 100 python tools/small_conv_timing.py 2>&1 | tee $OUT/small_conv_timing.txt
  ;;
half)
  # cfg 125: the register-weights shape with its cell tiles over TWO work-groups (KMX_CONV_TUNE regw_half=1: batch 15-21 at 192 channels;
  # =2 forces it wherever the three-way split would be taken, for the parity run)
  KMX_CONV_TUNE=regw_half=2 timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_fuzz.py "tests/test_gpu_model.py::test_full_batch_properties" -m gpu -q -p no:cacheprovider 2>&1 | tail -40 | cut -c1-600 | tee $OUT/parity.log
  for t in regw_half=0 regw_half=1 regw_half=0 regw_half=1; do
    KMX_CONV_TUNE=$t timeout 200 python tools/small_batch_scan.py 2>&1 | grep SCAN | tee -a $OUT/small_batch_scan.txt
  done
  ;;
small)
  timeout 900 python -m pytest tests/test_gpu_layers.py "tests/test_gpu_model.py::test_model_vs_oracle" "tests/test_gpu_model.py::test_full_batch_properties" tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity.log
  timeout 300 python tools/small_batch_scan.py 2>&1 | grep SCAN | tee $OUT/small_batch_scan.txt
  for n in 1 8 32 256; do
    echo "== batch $n: per-class launch times (hipEvent pair per launch, one stream)" | tee -a $OUT/kernel_classes.txt
    timeout 200 python3 bench.py --no-cpu-baseline --no-callers --batch $n --steps 30 --warmup 5 2>>$OUT/err.txt | grep -o '"kernel_avg_launch_us": {[^}]*}\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | tee -a $OUT/kernel_classes.txt; echo | tee -a $OUT/kernel_classes.txt
  done
  ;;
esac
done

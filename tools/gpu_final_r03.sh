#!/bin/bash
# Round-3 evidence run at HEAD: rocprofv3 kernel-trace stats + PMC passes of the bench command, in-kernel cycle stamps of both
# kernel families, batch scan, other configs, the driver's bench command (also with --pmc). Everything under gpurun_out/final_r03
# (summaries are copied to profiles/r03_final).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/final_r03
rm -rf $OUT; mkdir -p $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^$\|====" > $OUT/smi.txt
export KMX_SPLIT_MIN=0   # kernels are profiled with the chip to themselves (one stream), as bench.py's roofline pass measures them
BENCH="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-callers"
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-callers > $OUT/bench_trace.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 150 rocprofv3 --pmc $pass -d $OUT/benchpmc_$tag -o bench -- $BENCH > $OUT/benchpmc_$tag.log 2>&1
done
unset KMX_SPLIT_MIN
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/bench_trace_two_streams -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-callers > $OUT/bench_trace_two_streams.log 2>&1
timeout 100 python tools/rocpd_summary.py $OUT $OUT/summary > $OUT/summary.log 2>&1
# in-kernel cycle stamps: the 3x3 shape with and without the residual, the seam kernel
python - <<'PY' > $OUT/conv_timing.log 2>&1
import ctypes, sys
sys.path.insert(0, ".")
from katago_amd import capi
lib = capi.load_library(); capi.check(lib.kmx_global_init(), lib)
for name, c in (("3x3 192->192 8-wave D3, batch 256, act out only", (3, 23, 3000 + 2048, 192, 192, 0)), ("same, residual in, raw + act out", (3, 23, 3000 + 2048, 192, 192, 1))):
    print("==", name, flush=True)
    ms = ctypes.c_double()
    lib.kmx_bench_conv(c[0], c[1], c[2], c[3], c[4], 256, 19, 19, c[5], 5, ctypes.byref(ms))
    print("   %.2f us per launch (instrumented)" % (ms.value * 1e3), flush=True)
for c in ((3, 23, 0, 192, 192, 0), (3, 23, 0, 192, 192, 1), (1, 23, 0, 384, 192, 1), (1, 23, 0, 192, 384, 1), (3, 23, 3001, 192, 192, 1), (3, 23, 3002, 192, 192, 1), (3, 23, 3004, 192, 192, 1)):
    ms = ctypes.c_double()
    lib.kmx_bench_conv(c[0], c[1], c[2], c[3], c[4], 256, 19, 19, c[5], 30, ctypes.byref(ms))
    fl = 2.0 * c[0] * c[0] * c[3] * c[4] * 361 * 256
    print("ks%d cfg%d var%-5d %d->%d mode%d: %7.2f us  %7.1f TFLOP/s" % (c + (ms.value * 1e3, fl / ms.value / 1e9)), flush=True)
PY
timeout 120 python tools/seam_timing.py 256 > $OUT/seam_timing.log 2>&1
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 150 python3 bench.py --no-cpu-baseline --no-callers "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"dtype": "[a-z0-9]*"' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/scan.txt"; }
for n in 1 8 32 64 128 512; do b "b18c384nbt default precision batch $n" A=1 -- --batch $n --steps 40 --warmup 5 --no-profile; done
b "b18c384nbt bf16 batch 256" A=1 -- --dtype bf16 --steps 40 --warmup 5
b "b18c384nbt fp16 batch 256, file's own values (KMX_FP16_SCALE8=0)" KMX_FP16_SCALE8=0 -- --dtype fp16 --steps 40 --warmup 5 --no-profile
b "b18c384nbt default batch 256 one stream" KMX_SPLIT_MIN=0 -- --steps 40 --warmup 5 --no-profile
b "b18c384nbt default batch 256 seam v1 (KMX_PW_V2=0)" KMX_PW_V2=0 -- --steps 40 --warmup 5
b "b28c512nbt default batch 512" A=1 -- --model b28c512nbt --batch 512 --steps 10 --warmup 2
b "b40c256 default batch 512" A=1 -- --model b40c256 --batch 512 --steps 10 --warmup 2
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc > $OUT/bench_pmc.json 2> $OUT/bench_pmc.err
cp -r gpurun_out/bench_pmc/summary $OUT/bench_pmc_summary 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json

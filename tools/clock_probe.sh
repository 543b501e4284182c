#!/bin/bash
# The shader clock under the bench's workloads (tools/clock_meter.hip beside bench.py / the MFMA loop), round 6.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/r06/clock}
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-callers --no-pmc --no-profile --warmup 5"
run() { local name=$1 secs=$2; shift 2
  tools/_build/clock_meter $secs 200 $OUT/$name.csv > $OUT/$name.meter.txt 2>&1 &
  local mp=$!
  sleep 1
  "$@" > $OUT/$name.out 2>&1
  wait $mp
  echo "== $name" | tee -a $OUT/clock.txt
  cat $OUT/$name.meter.txt | tee -a $OUT/clock.txt
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|LOOP.*' $OUT/$name.out | tr '\n' ' ' | tee -a $OUT/clock.txt; echo | tee -a $OUT/clock.txt
}
run idle 3 sleep 2
run bench_248 24 $B --batch 248 --steps 2000
run bench_256 24 $B --batch 256 --steps 2000
run bench_136 22 $B --batch 136 --steps 2500
run bench_32 20 $B --batch 32 --steps 5000
LOOP="import ctypes,sys; sys.path.insert(0,'.'); from katago_amd import capi; lib=capi.load_library(); ms,tf,mhz=ctypes.c_double(),ctypes.c_double(),ctypes.c_double(); rc=lib.kmx_bench_mfma(8,248,3,540,12000,ctypes.byref(ms),ctypes.byref(tf),ctypes.byref(mhz)); print('LOOP 248 wgs rc',rc,'ms',round(ms.value,4),'tflops',round(tf.value,1),'mhz',round(mhz.value))"
run mfma_loop_248 12 python -c "$LOOP"

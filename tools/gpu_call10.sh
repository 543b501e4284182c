#!/bin/bash
# Single-loader DMA roles + residual add in the epilogue: parity, A/B against the previous library and the two-loader variant.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/call10
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_pointwise.py tests/test_gpu_model.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee "$OUT/parity.log"
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 300 python3 bench.py --no-cpu-baseline "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"kernel_time_share": {[^}]*}\|"frac": [0-9.]*\|"profiled_ms_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/scan.txt"; }
PREV=$PWD/katago_amd/libkatamx_prev.so
b "prev" KMX_LIBRARY=$PREV -- --steps 50 --warmup 5
b "new " -- --steps 50 --warmup 5
b "new ways1" KMX_SPLIT_MIN=0 -- --steps 50 --warmup 5
b "new " -- --steps 50 --warmup 5
python - <<'PY' 2>&1 | tee "$OUT/conv_ab.txt"
import ctypes, os, sys
sys.path.insert(0, ".")
from katago_amd import capi
libs = {"prev": capi.load_library(path=os.path.abspath("katago_amd/libkatamx_prev.so")), "new": capi.load_library(path=os.path.abspath("katago_amd/libkatamx.so"))}
for l in libs.values(): capi.check(l.kmx_global_init(), l)
def run(lib, ks, cfg, var, cin, cout, mode, batch=256, iters=30):
    ms = ctypes.c_double()
    rc = lib.kmx_bench_conv(ks, cfg, var, cin, cout, batch, 19, 19, mode, iters, ctypes.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")
cases = [(3, 23, 0, 192, 192, 0), (3, 23, 0, 192, 192, 1), (3, 13, 0, 192, 192, 1), (1, 23, 0, 384, 192, 1), (1, 23, 0, 192, 384, 1), (3, 23, 3001, 192, 192, 1)]
for rep in range(2):
    for c in cases:
        print("ks%d cfg%d var%-5d %d->%d mode%d | " % c + " ".join("%s %7.2f us" % (k, run(l, *c)) for k, l in libs.items()), flush=True)
new = libs["new"]
print("== all DMA on waves 0-3 (3000) vs weights on 0-3, image on 4-7 (16384)")
for rep in range(2):
    for c in [(3, 23, 3000, 192, 192, 1), (3, 23, 3000 + 16384, 192, 192, 1), (3, 23, 3000, 192, 192, 0), (3, 23, 3000 + 16384, 192, 192, 0)]:
        print("ks%d cfg%d var%-5d %d->%d mode%d | " % c + "%7.2f us" % run(new, *c), flush=True)
for c in [(3, 23, 3000 + 2048, 192, 192, 1), (3, 23, 3000 + 16384 + 2048, 192, 192, 1), (3, 23, 3000 + 2048, 192, 192, 0)]:
    print("ks%d cfg%d var%-5d %d->%d mode%d | " % c + "%7.2f us" % run(new, *c, iters=5), flush=True)
PY

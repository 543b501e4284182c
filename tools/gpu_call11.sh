#!/bin/bash
# New mish formulation (packed, shared reciprocal): parity + A/B against the previous library.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/call11
mkdir -p "$OUT"
timeout 400 python -m pytest tests/test_gpu_layers.py tests/test_gpu_pointwise.py tests/test_gpu_model.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 | tee "$OUT/parity.log"
b() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 200 python3 bench.py --no-cpu-baseline "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*\|"kernel_time_share": {[^}]*}\|"frac": [0-9.]*\|"profiled_ms_per_step": [0-9.]*\|"box": {[^}]*}' | tr '\n' ' ')
  echo "$name | $v" | tee -a "$OUT/scan.txt"; }
PREV=$PWD/katago_amd/libkatamx_prev2.so
b "prev2" KMX_LIBRARY=$PREV -- --steps 50 --warmup 5
b "new  " -- --steps 50 --warmup 5
b "prev2" KMX_LIBRARY=$PREV -- --steps 50 --warmup 5
b "new  " -- --steps 50 --warmup 5
python - <<'PY' 2>&1 | tee "$OUT/conv_ab.txt"
import ctypes, os, sys
sys.path.insert(0, ".")
from katago_amd import capi
libs = {"prev2": capi.load_library(path=os.path.abspath("katago_amd/libkatamx_prev2.so")), "new": capi.load_library(path=os.path.abspath("katago_amd/libkatamx.so"))}
for l in libs.values(): capi.check(l.kmx_global_init(), l)
def run(lib, ks, cfg, var, cin, cout, mode, batch=256, iters=30):
    ms = ctypes.c_double()
    rc = lib.kmx_bench_conv(ks, cfg, var, cin, cout, batch, 19, 19, mode, iters, ctypes.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")
cases = [(3, 23, 0, 192, 192, 0), (3, 23, 0, 192, 192, 1), (1, 23, 0, 384, 192, 1), (1, 23, 0, 192, 384, 1)]
for rep in range(2):
    for c in cases:
        print("ks%d cfg%d var%-5d %d->%d mode%d | " % c + " ".join("%s %7.2f us" % (k, run(l, *c)) for k, l in libs.items()), flush=True)
PY

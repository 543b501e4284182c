#!/bin/bash
# Memory-safety check of the DEVICE code on the CPU: the fully emulated build of the library (tests/fakehip; the real convolution, chain,
# small-batch and seam kernels through the test suite's rewrite rules) compiled with AddressSanitizer; device allocations are heap
# blocks with red zones, so an out-of-bounds global read or write of any kernel aborts the run. ~15 min on 8 cores.
# (LDS is one mapped buffer: overruns inside it are not seen.)
#   bash tools/emulated_asan.sh
set -eu
REPO="$(cd "$(dirname "$0")/.." && pwd)"
D=$(mktemp -d /tmp/kmx_emuasan.XXXXXX)
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
python3 - "$REPO" "$D" <<'PY'
import sys
repo, d = sys.argv[1], sys.argv[2]
sys.path.insert(0, repo + "/tests"); sys.path.insert(0, repo)
import test_engine_emulated as E
print(E.build_emu_full(d, extra_flags=("-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-shared-libasan"), so_name="libkatamx_emuasan.so"))
PY
cd "$D"
cat > run.py <<PY
import ctypes, os, sys
sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/tests")
import numpy as np
from katago_amd import capi
lib = ctypes.CDLL("$D/libkatamx_emuasan.so")   # no torch in this process: ASAN must own the allocator
for name, (res, args) in capi.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
capi._lib = lib
from katago_amd import modelgen, nninterface as nn
from conftest import make_rows
rng = np.random.default_rng(0)
# layers: 3x3 / 1x1 / 5x5, channel counts that are not multiples of the tile, several boards, a rectangular and a tiny board
for ks, cin, cout, X, Y, n in ((3, 32, 32, 9, 9, 1), (1, 64, 96, 19, 19, 1), (5, 22, 64, 13, 9, 1), (3, 40, 200, 19, 19, 1), (3, 64, 192, 19, 19, 2),
                              (1, 96, 384, 19, 19, 1), (3, 96, 64, 7, 11, 3), (1, 40, 192, 5, 4, 2)):
    w = (rng.normal(size=(cout, cin, ks, ks)) * 0.1).astype(np.float32)
    x = rng.normal(size=(n, Y * X, cin)).astype(np.float32)
    assert np.isfinite(np.asarray(nn.testEvaluateConv(w, n, X, Y, True, x))).all()
nn.globalInitialize()
ctx = nn.createComputeContext([0], 19, 19, precision="bf16")
p = "$D/b2c32nbt.bin"; modelgen.write_model(p, "b2c32nbt", seed=4)
sp, gl = make_rows(rng, 2, 19, [(19, 19), (9, 13)])
h = nn.createComputeHandle(ctx, nn.loadModelFile(p), 2)
assert all(np.isfinite(v).all() for v in nn.getOutput(h, sp, gl, np.array([3, 6], np.int32), np.array([0, 1], np.float32)).values())
h.close()
print("clean:", os.environ.get("KMX_CONV_TUNE"))
PY
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
# the default small-batch shapes (round 5: the register-weights 3x3 shapes, cfg 127 - a board's cell tiles over three work-groups -, 125 - over
# two -, 128, 126; cfg 113 / 114 for 1x1), round 4's slab-ring shapes (regw=0: 117 / 118 / 119), the 4-wave shapes of conv_kernel.h, the 8-wave shapes
for v in "A=1" "KMX_CONV_TUNE=regw_half=2" "KMX_CONV_TUNE=regw=0" "KMX_CONV_TUNE=loaders_split=0,split1x1=0" "KMX_CONV_TUNE=loaders_max_wgs=0" "KMX_CONV_TUNE=loaders_max_wgs=0,regw=0" "KMX_CONV_TUNE=loaders=0,deep1x1=0" "KMX_CONV_TUNE=min_wgs8=1"; do
  env LD_PRELOAD="$ASAN" $v python3 run.py
done
echo "emulated ASAN run: clean"

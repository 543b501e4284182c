#!/bin/bash
# Memory-safety check of the DEVICE code on the CPU: the fully emulated build of the library (tests/fakehip, real convolution
# kernel) compiled with AddressSanitizer; device allocations are heap blocks with red zones, so an out-of-bounds global
# read or write of any kernel aborts the run. Takes a few minutes. (LDS is one mapped buffer: overruns inside it are not seen.)
#   bash tools/emulated_asan.sh
set -eu
REPO="$(cd "$(dirname "$0")/.." && pwd)"
D=$(mktemp -d /tmp/kmx_emuasan.XXXXXX)
CLANG=/opt/rocm/lib/llvm/bin/clang++
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
python3 - "$REPO" "$D" <<'PY'
import re, sys
repo, d = sys.argv[1], sys.argv[2]
txt = open(repo + "/tests/test_engine_emulated.py").read()
ns = {}
exec(txt[txt.index("CONV_REWRITES = ["):txt.index("]\n", txt.index("CONV_REWRITES = [")) + 1], ns)
src = open(repo + "/katago_amd/csrc/conv_kernel.h").read()
for pat, rep, count in ns["CONV_REWRITES"]:
    src, k = re.subn(pat, rep, src)
    assert k == count, (pat, k)
open(d + "/conv_kernel.h", "w").write(src)
open(d + "/conv_mfma.hip", "w").write(open(repo + "/katago_amd/csrc/conv_mfma.hip").read())
PY
CXX="$CLANG -x c++ -std=c++20 -O1 -g -fPIC -pthread -fsanitize=address -fno-omit-frame-pointer -I$REPO/tests/fakehip/emul -I$REPO/tests/fakehip -I$REPO/katago_amd/csrc -DKMX_EMU_REAL_CONV"
cd "$D"
$CXX -c conv_mfma.hip -o conv_mfma.o &
$CXX -c "$REPO/tests/fakehip/emulate_engine.cpp" -o ee.o &
for f in misc_kernels.hip transformer_kernels.hip engine.cpp model_desc.cpp kmx_api.cpp; do $CXX -c "$REPO/katago_amd/csrc/$f" -o "${f%.*}.o" & done
wait
$CLANG -shared -fPIC -pthread -fsanitize=address -shared-libasan -o libkatamx_emuasan.so ./*.o -lz
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
for ks, cin, cout, X, Y, n in ((3, 32, 32, 9, 9, 1), (1, 64, 96, 19, 19, 1), (5, 22, 64, 13, 9, 1), (3, 40, 200, 19, 19, 1), (3, 64, 192, 19, 19, 1), (1, 96, 384, 19, 19, 1)):
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
if os.environ.get("KMX_EXPERIMENTAL_TRANSFORMER") == "1":
    v = np.load("$REPO/tests/golden/torch_tfb_vectors.npz")
    h = nn.createComputeHandle(ctx, nn.loadModelFile("$REPO/tests/golden/torch_tfb.bin.gz"), 2)
    assert all(np.isfinite(o).all() for o in nn.getOutput(h, v["spatial_nhwc"][2:4], v["glob"][2:4], None, np.zeros(2, np.float32)).values())
    h.close()
print("clean: KMX_MIN_WGS8=%s KMX_ATTENTION_VALU=%s" % tuple(os.environ.get(k) for k in ("KMX_MIN_WGS8", "KMX_ATTENTION_VALU")))
PY
export LD_PRELOAD="$ASAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 KMX_EXPERIMENTAL_TRANSFORMER=1
python3 run.py                                   # narrow shapes, matrix-core attention
KMX_MIN_WGS8=1 python3 run.py                    # 8-wave product shapes
KMX_MIN_WGS8=1 KMX_ATTENTION_VALU=1 python3 run.py   # 8-wave shapes at a small batch, plain attention
rm -rf "$D"

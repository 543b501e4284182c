#!/bin/bash
# ThreadSanitizer over the emulated convolution kernel (tests/fakehip): a COARSE check of work-group barrier placement.
# LDS-DMA copies are immediate under emulation, so what can be seen is a ring slot or image buffer written by one wave and
# read by another with no barrier in between. Sensitivity is limited (a kernel with all loop barriers removed is flagged, a
# ring that is one slot too shallow was not), so a clean run is weak evidence; the GPU remains the judge.
#   bash tools/emulated_tsan.sh            # product 8-wave shapes, a no-barrier control
set -eu
REPO="$(cd "$(dirname "$0")/.." && pwd)"
D=$(mktemp -d /tmp/kmx_emutsan.XXXXXX)
CLANG=/opt/rocm/lib/llvm/bin/clang++
python3 - "$REPO" "$D" <<'PY'
import os, re, sys
repo, d = sys.argv[1], sys.argv[2]
txt = open(repo + "/tests/test_engine_emulated.py").read()
ns = {}
exec(txt[txt.index("CONV_REWRITES = ["):txt.index("]\n", txt.index("CONV_REWRITES = [")) + 1], ns)
src = open(repo + "/katago_amd/csrc/conv_kernel.h").read()
for pat, rep, count in ns["CONV_REWRITES"]:
    src, k = re.subn(pat, rep, src)
    assert k == count, (pat, k)
# sanitizers skip accesses through non-default address spaces: make the LDS / global pointers ordinary ones
src = src.replace("__attribute__((address_space(3)))", "").replace("__attribute__((address_space(1)))", "")
os.makedirs(d + "/control")
open(d + "/conv_kernel.h", "w").write(src)
open(d + "/control/conv_kernel.h", "w").write(src.replace("      waitStep(t);\n      __builtin_amdgcn_s_barrier();\n", "      waitStep(t);\n"))
for sub in ("", "/control"):
    open(d + sub + "/conv_mfma.hip", "w").write(open(repo + "/katago_amd/csrc/conv_mfma.hip").read())
PY
cat > "$D/driver.cpp" <<'CPP'
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "katamx.h"
int main(int argc, char** argv) {
  (void)argc;
  int ks = atoi(argv[1]), cin = atoi(argv[2]), cout = atoi(argv[3]), X = atoi(argv[4]), Y = atoi(argv[5]), n = atoi(argv[6]);
  std::vector<float> w((size_t)ks * ks * cin * cout), x((size_t)n * X * Y * cin), out((size_t)n * X * Y * cout);
  unsigned s = 1;
  for(float& v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f * 0.2f - 0.1f; }
  for(float& v : x) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  kmx_conv_desc d; d.conv_y_size = ks; d.conv_x_size = ks; d.in_channels = cin; d.out_channels = cout; d.weights = w.data();
  int rc = kmx_test_conv(&d, n, X, Y, KMX_PREC_BF16, x.data(), out.data());
  printf("rc %d (%s)\n", rc, rc ? kmx_last_error() : "ok");
  return rc;
}
CPP
CXX="$CLANG -x c++ -std=c++20 -O1 -g -fPIC -pthread -fsanitize=thread -I$REPO/tests/fakehip/emul -I$REPO/tests/fakehip -I$REPO/katago_amd/csrc -I$REPO/include -DKMX_EMU_REAL_CONV"
cd "$D"
$CXX -c conv_mfma.hip -o conv_mfma.o &
(cd control && $CXX -c conv_mfma.hip -o conv_mfma.o) &
$CXX -c "$REPO/tests/fakehip/emulate_engine.cpp" -o ee.o &
for f in misc_kernels.hip transformer_kernels.hip engine.cpp model_desc.cpp kmx_api.cpp numa.cpp; do $CXX -c "$REPO/katago_amd/csrc/$f" -o "${f%.*}.o" & done
$CXX -c driver.cpp -o driver.o &
wait
OBJS="driver.o ee.o misc_kernels.o transformer_kernels.o engine.o model_desc.o kmx_api.o numa.o"
$CLANG -pthread -fsanitize=thread -o driver $OBJS conv_mfma.o -lz
$CLANG -pthread -fsanitize=thread -o control/driver $OBJS control/conv_mfma.o -lz
export TSAN_OPTIONS=halt_on_error=0
count() { grep -c "WARNING: ThreadSanitizer" "$1" || true; }
KMX_CONV_TUNE=min_wgs8=1 ./driver 3 96 192 19 19 2 > run.log 2>&1 || true
echo "8-wave 3x3 96->192: $(count run.log) reports"
KMX_CONV_TUNE=min_wgs8=1 control/driver 3 96 192 9 9 1 > control.log 2>&1 || true
echo "control (the kernel with its loop barriers removed): $(count control.log) reports (must be > 0)"
rm -rf "$D"

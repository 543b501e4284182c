#!/bin/bash
# ThreadSanitizer over the emulated kernels (tests/fakehip): a COARSE check of work-group barrier placement.
# LDS-DMA copies are immediate under emulation, so what can be seen is a ring slot or image buffer written by one wave and
# read by another with no barrier in between. Sensitivity is limited (a kernel with all loop barriers removed is flagged, a
# ring that is one slot too shallow was not), so a clean run is weak evidence; the GPU remains the judge.
# Round 6: built through the test suite's own builder and rewrite rules (tests/test_engine_emulated.py build_emu_full, like
# tools/emulated_asan.sh) - the round-1 version rewrote conv_kernel.h alone and stopped compiling when conv_mfma.hip gained
# conv_small_kernel.h. The emulator's barrier is seen through its acquire / release atomics (tests/fakehip/emul/hip/hip_runtime.h).
#   bash tools/emulated_tsan.sh            # product 8-wave 3x3 shape, a small-batch register-weights shape, a no-barrier control
set -eu
REPO="$(cd "$(dirname "$0")/.." && pwd)"
D=$(mktemp -d /tmp/kmx_emutsan.XXXXXX)
CLANG=/opt/rocm/lib/llvm/bin/clang++
mkdir -p "$D/product" "$D/control"
python3 - "$REPO" "$D" <<'PY'
import sys
repo, d = sys.argv[1], sys.argv[2]
sys.path.insert(0, repo + "/tests"); sys.path.insert(0, repo)
import test_engine_emulated as E
flags = ("-O1", "-g", "-fsanitize=thread")
print(E.build_emu_full(d + "/product", extra_flags=flags, so_name="libkatamx_emutsan.so", plain_pointers=True))
# the control: conv_kernel.h with the barrier of its step loop removed - the tool must see THAT
print(E.build_emu_full(d + "/control", extra_flags=flags, so_name="libkatamx_emutsan.so", plain_pointers=True,
                       conv_mutations=[(r"      waitStep\(t\);\n      __builtin_amdgcn_s_barrier\(\);\n", "      waitStep(t);\n", 1)]))
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
cd "$D"
for v in product control; do
  $CLANG -std=c++20 -O1 -g -pthread -fsanitize=thread -I"$REPO/include" driver.cpp -o $v/driver -L"$D/$v" -lkatamx_emutsan -Wl,-rpath,"$D/$v"
done
export TSAN_OPTIONS=halt_on_error=0
# A report counts as a SINK pair when both accesses are writes issued by the same statement through the same call path: the kernels send
# requests and stores they do not want to designated sink addresses (conv_kernel.h: `mySlack` for the padding LDS-DMA requests, `trash` for the
# rows of off-board cells), where lanes of different waves overwrite one another by design. Everything else - a read racing a write, writes
# of two different statements - is a finding.
cat > classify.py <<'PY'
import re, sys
text = open(sys.argv[1]).read()
blocks = text.split("WARNING: ThreadSanitizer: data race")[1:]
sink = 0
for b in blocks:
    m = re.search(r"^\s*(Atomic write|Atomic read|Write|Read) of size.*?\n(.*?)\n\s*Previous (atomic write|atomic read|write|read) of size.*?\n(.*?)(\n\n|\Z)", b, re.S | re.M)
    if not m:
        continue
    path = lambda t: re.findall(r"#\d+ .*? (\S+:\d+):\d+ \(", t)[:5]
    if "rite" in m.group(1) and "write" in m.group(3) and path(m.group(2)) and path(m.group(2)) == path(m.group(4)):
        sink += 1
print("%d reports, %d of them sink pairs, %d findings" % (len(blocks), sink, len(blocks) - sink))
sys.exit(0 if len(blocks) == sink else 1)
PY
ok=1
KMX_CONV_TUNE=min_wgs8=1 product/driver 3 96 192 19 19 2 > run.log 2>&1 || true
echo "8-wave 3x3 96->192 (conv_kernel.h): $(tail -1 run.log | cut -c1-40): $(python3 classify.py run.log)"
python3 classify.py run.log > /dev/null || ok=0
product/driver 3 96 192 13 13 2 > small.log 2>&1 || true
echo "small-batch 3x3 96->192 at its default shape (conv_small_kernel.h, weights in registers): $(tail -1 small.log | cut -c1-40): $(python3 classify.py small.log)"
python3 classify.py small.log > /dev/null || ok=0
KMX_CONV_TUNE=min_wgs8=1 control/driver 3 96 192 9 9 1 > control.log 2>&1 || true
echo "control (conv_kernel.h with its loop barrier removed): $(python3 classify.py control.log) (findings must be > 0)"
python3 classify.py control.log > /dev/null && ok=0
[ $ok = 1 ] && echo "emulated TSAN run: no finding in the product kernels, control flagged"
[ -n "${KMX_TSAN_KEEP:-}" ] && echo "kept: $D" || rm -rf "$D"

#!/usr/bin/env python3
"""In-kernel cycle accounting of the convolution's main loop (ABL_TIMING variants: s_memtime stamps between the segments of a
step, summed over the steps of one work-group; printed per wave by kmx_bench_conv on stderr)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
for label, ks, cfg, var, cin, cout, mode, batch in (
        ("3x3 192->192 8-wave D3, batch 256", 3, 23, 3000 + 2048, 192, 192, 0, 256),
        ("  same, no DMA", 3, 23, 3000 + 2052, 192, 192, 0, 256),
        ("  same, LDS reads replaced by constants", 3, 23, 3000 + 2056, 192, 192, 0, 256),
        ("  same, no epilogue", 3, 23, 3000 + 2049, 192, 192, 0, 256),
        ("1x1 192->384 8-wave D3, batch 256", 1, 23, 3000 + 2048, 192, 384, 1, 256),
        ("3x3 192->192 4-wave x96 D2, batch 16", 3, 13, 2000 + 2048, 192, 192, 0, 16),
        ("3x3 192->192 4-wave x32 D2, batch 1", 3, 11, 2000 + 2048, 192, 192, 0, 1)):
    ms = ctypes.c_double()
    print("== %s" % label, flush=True)
    sys.stderr.flush()
    rc = lib.kmx_bench_conv(ks, cfg, var, cin, cout, batch, 19, 19, mode, 5, ctypes.byref(ms))
    print("   %.2f us per launch (instrumented)" % (ms.value * 1e3) if rc == 0 else "   error " + lib.kmx_last_error().decode(), flush=True)

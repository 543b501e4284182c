#!/usr/bin/env python3
"""In-kernel cycle stamps of the register-weights small-batch 3x3 shapes (conv_small_kernel.h REGW / TIMING) on a 192 -> 192 layer: cfg 127 at
batch 8 (a board's cell tiles over three work-groups), 128 at batch 32, 126 at batch 64 (64 channels per work-group); with and without the
residual epilogue; beside each the product instantiation's time per launch (back-to-back launches of the same layer: weights and images
stay in the caches, unlike inside a pass).   python tools/small_conv_timing.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
for cfg, batch in ((127, 8),):  # (round 5 also stamped cfg 128 and 126: those instantiations spilled - conv_bench.hip - and are gone)
    for mode in (0, 1):
        for variant, what in ((0, "product"), (9999, "cycle stamps")):
            ms = ctypes.c_double()
            print("== cfg %d, batch %d, 3x3 192->192, epilogue mode %d (%s), %s" % (cfg, batch, mode, "residual + raw + act" if mode else "act only", what), flush=True)
            rc = lib.kmx_bench_conv(3, cfg, variant, 192, 192, batch, 19, 19, mode, 20, ctypes.byref(ms))
            if rc != 0:
                print("   error:", lib.kmx_last_error().decode(), flush=True)
            else:
                print("   %.2f us per launch" % (ms.value * 1e3), flush=True)

#!/usr/bin/env python3
"""In-kernel cycle stamps of the small-batch 3x3 shape (4 waves x 32 channels, D = 2) at batch 1 / 8 / 32, and the product's launch
times there: where a step's ~700 cycles go when a wave has 6 MFMAs per step.   python tools/small_batch_timing.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
for batch in (1, 8, 32):
    for name, cfg, var, mode in (("4 waves x 32 channels, cycle stamps (the round-2 step form)", 11, 2000 + 2048, 1), ("product", 11, 0, 0), ("product", 11, 0, 1),
                                 ("4 waves x 96 channels, product", 13, 0, 1)):
        ms = ctypes.c_double()
        print("== batch %d, 3x3 192->192, %s, mode %d" % (batch, name, mode), flush=True)
        rc = lib.kmx_bench_conv(3, cfg, var, 192, 192, batch, 19, 19, mode, 10, ctypes.byref(ms))
        if rc != 0:
            print("   error:", lib.kmx_last_error().decode(), flush=True)
        else:
            print("   %.2f us per launch" % (ms.value * 1e3), flush=True)

#!/usr/bin/env python3
"""The chained 3x3 convolutions (conv_chain_kernel.h) against separate launches on `batch` boards: microseconds per sequence of 2 / 4
convolutions, and the cycle stamps of a chained launch's phases (per wave of the middle board)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for n_conv, chained in ((2, 0), (2, 2), (4, 0), (4, 2), (4, 4)):
    ms = ctypes.c_double()
    capi.check(lib.kmx_bench_conv_chain(batch, n_conv, chained, 30, 0, ctypes.byref(ms)), lib)
    fl = 2.0 * 9 * 192 * 192 * 361 * batch * n_conv
    print("%d convolutions, chained %d: %8.2f us per sequence = %6.2f us per convolution, %6.1f TFLOP/s" % (n_conv, chained, ms.value * 1e3, ms.value * 1e3 / n_conv, fl / ms.value / 1e9), flush=True)
for n_conv, chained in ((4, 4), (2, 2)):
    print("== cycle stamps, %d convolutions chained" % n_conv, flush=True)
    ms = ctypes.c_double()
    capi.check(lib.kmx_bench_conv_chain(batch, n_conv, chained, 5, 1, ctypes.byref(ms)), lib)
    print("   %.2f us per sequence (instrumented)" % (ms.value * 1e3), flush=True)

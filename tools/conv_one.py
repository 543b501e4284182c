#!/usr/bin/env python3
"""Launch ONE convolution shape a few times (for rocprofv3 --pmc passes).  python tools/conv_one.py ks wn variant cin cout mode [iters]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

ks, wn, var, cin, cout, mode = [int(x) for x in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
ms = ctypes.c_double()
capi.check(lib.kmx_bench_conv(ks, wn, var, cin, cout, 256, 19, 19, mode, iters, ctypes.byref(ms)), lib)
print("ks%d wn%d var%d %d->%d mode%d: %.4f ms" % (ks, wn, var, cin, cout, mode, ms.value))

#!/usr/bin/env python3
"""The seam kernel alone: average launch time (kmx_bench_seam) and, for the persistent kernel, per-wave cycle sums of its phases.
    python tools/seam_timing.py [batch]        (KMX_PW_V2=0: the one-tile-per-work-group kernel)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
ms = ctypes.c_double()
capi.check(lib.kmx_bench_seam(batch, 30, 0, ctypes.byref(ms)), lib)
print("seam, batch %d: %.2f us per launch" % (batch, ms.value * 1e3), flush=True)
if os.environ.get("KMX_PW_V2", "1") != "0":
    capi.check(lib.kmx_bench_seam(batch, 5, 1, ctypes.byref(ms)), lib)
    print("  instrumented: %.2f us per launch" % (ms.value * 1e3), flush=True)

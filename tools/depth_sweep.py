#!/usr/bin/env python3
"""Pipeline depth D (weight-slab ring of D+1, requests D steps ahead) for the narrow work-group shapes at small batch."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
for ks, cfg, cin, cout, mode, depths in ((3, 11, 192, 192, 1, (2, 3, 4, 5)), (3, 12, 192, 192, 1, (2, 3, 4)), (3, 13, 192, 192, 1, (2, 3, 4)),
                                         (3, 23, 192, 192, 1, (2, 3, 4, 5)), (1, 11, 384, 192, 0, (2, 3, 4)), (1, 12, 384, 192, 0, (2, 3, 4))):
    for batch in (1, 16, 64, 256):
        row = []
        for d in depths:
            ms = ctypes.c_double()
            rc = lib.kmx_bench_conv(ks, cfg, d * 1000, cin, cout, batch, 19, 19, mode, 20, ctypes.byref(ms))
            row.append("D%d %6.1f us" % (d, ms.value * 1e3) if rc == 0 else "D%d   n/a   " % d)
        print("ks%d cfg%d %3d->%3d batch %3d : %s" % (ks, cfg, cin, cout, batch, "  ".join(row)), flush=True)

#!/usr/bin/env python3
"""Which 4-wave work-group width (32/64/96 channels) is fastest for small batches? 3x3 192->192 and 1x1 384->192."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
for ks, cin, cout, mode, cfgs in ((3, 192, 192, 1, (11, 12, 13, 22, 23)), (1, 384, 192, 0, (11, 12, 22, 23)), (1, 192, 384, 1, (11, 12, 22, 23))):
    for batch in (1, 4, 16, 32, 64, 128):
        row = []
        for cfg in cfgs:
            ms = ctypes.c_double()
            rc = lib.kmx_bench_conv(ks, cfg, 0, cin, cout, batch, 19, 19, mode, 20, ctypes.byref(ms))
            row.append("cfg%d %6.1f us" % (cfg, ms.value * 1e3) if rc == 0 else "cfg%d   n/a   " % cfg)
        print("ks%d %3d->%3d batch %3d : %s" % (ks, cin, cout, batch, "  ".join(row)), flush=True)

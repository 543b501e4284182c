#!/usr/bin/env python3
"""Ablation table of the 3x3 192->192 convolution (8-wave work-group, D=2): which part of the kernel costs what.
mask bits: 1 no epilogue, 2 no MFMA/LDS reads, 4 no DMA, 8 LDS reads replaced by constants."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
for ks, cfg, depth, cin, cout, mode in ((3, 23, 3000, 192, 192, 0), (3, 23, 3000, 192, 192, 1), (1, 23, 3000, 192, 384, 1)):
    for abl in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13):
        ms = ctypes.c_double()
        rc = lib.kmx_bench_conv(ks, cfg, depth + abl, cin, cout, 256, 19, 19, mode, 20, ctypes.byref(ms))
        if rc != 0:
            print("abl %d: %s" % (abl, lib.kmx_last_error().decode()))
            continue
        tags = [t for bit, t in ((1, "noEpi"), (2, "noMFMA"), (4, "noDMA"), (8, "noLdsRead")) if abl & bit]
        print("ks%d cfg%d D%d %d->%d mode%d abl %2d (%s): %7.2f us" % (ks, cfg, depth // 1000, cin, cout, mode, abl, " ".join(tags), ms.value * 1e3),
              flush=True)

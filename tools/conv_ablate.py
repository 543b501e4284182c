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
for depth, mode in ((3000, 0),):
    for abl in (0, 1, 4, 5, 8, 9, 13):
        ms = ctypes.c_double()
        rc = lib.kmx_bench_conv(3, 23, depth + abl, 192, 192, 256, 19, 19, mode, 20, ctypes.byref(ms))
        if rc != 0:
            print("abl %d: %s" % (abl, lib.kmx_last_error().decode()))
            continue
        print("D%d mode%d abl %2d (%s%s%s%s): %7.2f us" % (depth // 1000, mode, abl, "noEpi " if abl & 1 else "", "noMFMA " if abl & 2 else "",
                                                      "noDMA " if abl & 4 else "", ("noLdsRead " if abl & 8 else "") + ("noVmWait " if abl & 16 else "") + ("noWdma " if abl & 32 else "") + ("noAdma" if abl & 64 else ""), ms.value * 1e3), flush=True)

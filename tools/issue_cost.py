#!/usr/bin/env python3
"""Shader cycles per step of a wave that issues 6 MFMAs per step and (a) nothing else, (b) two LDS-DMA instructions, (c) two plain
global loads + two ds_write_b128 (register staging) - one wave per SIMD, L2-resident sources (kmx_bench_mfma, mode 256 + variant)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)
for wgs in (8, 256):
    for mode, name in ((256, "6 MFMAs"), (257, "6 MFMAs + 2 LDS-DMA (global_load_lds 16 B/lane)"), (258, "6 MFMAs + 2 global_load_dwordx4 + 2 ds_write_b128")):
        ms, cyc, mhz = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        capi.check(lib.kmx_bench_mfma(4, wgs, mode, 2000, 5, ctypes.byref(ms), ctypes.byref(cyc), ctypes.byref(mhz)), lib)
        print("%3d work-groups of 4 waves, %-52s: %7.1f cycles per step  (%.0f MHz)" % (wgs, name, cyc.value, mhz.value), flush=True)

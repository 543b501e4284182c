#!/usr/bin/env python3
"""Practical MFMA ceiling of the box: issue rate of v_mfma_f32_32x32x16_bf16 in the convolution's step shape
(18 per wave per step), alone / with the step barrier / with the step's LDS reads, plus the shader clock observed."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from katago_amd import capi  # noqa: E402


def main():
    lib = capi.load_library()
    capi.check(lib.kmx_global_init(), lib)
    def name(mode):
        parts = ["mfma"]
        if mode & 2:
            parts.append("lds")
        if mode & 4:
            parts.append("valu")
        if mode & 1:
            parts.append("barrier/%d" % max(mode >> 4, 1))
        return "+".join(parts)

    for waves, wgs in ((4, 256), (4, 512), (8, 256)):
        for mode in (0, 1, 2, 3, 4, 7, 1 + 32, 3 + 32, 7 + 32, 1 + 48, 7 + 48):
            ms, tf, mhz = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            rc = lib.kmx_bench_mfma(waves, wgs, mode, 540, 20, ctypes.byref(ms), ctypes.byref(tf), ctypes.byref(mhz))
            if rc != 0:
                print("error", lib.kmx_last_error().decode())
                continue
            print("waves/WG %2d WGs %4d %-24s: %8.4f ms %7.1f TFLOP/s  clock %6.0f MHz" % (
                waves, wgs, name(mode), ms.value, tf.value, mhz.value), flush=True)


if __name__ == "__main__":
    main()

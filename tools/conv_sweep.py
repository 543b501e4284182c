#!/usr/bin/env python3
"""Times single launches of the MFMA convolution (product kernel and ablated variants) with kmx_bench_conv.
    python tools/conv_sweep.py [--batch 256] [--iters 20]
variant = depth*1000 + ablation mask (1 no epilogue, 2 no MFMA/LDS reads, 4 no DMA, 8 no LDS reads, 16 setprio)."""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from katago_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    lib = capi.load_library()
    capi.check(lib.kmx_global_init(), lib)

    def run(ks, wn, variant, cin, cout, mode, batch=None):
        ms = ctypes.c_double()
        b = batch or a.batch
        rc = lib.kmx_bench_conv(ks, wn, variant, cin, cout, b, 19, 19, mode, a.iters, ctypes.byref(ms))
        if rc != 0:
            print("ks%d wn%d var%-5d %d->%d mode%d: error %s" % (ks, wn, variant, cin, cout, mode, lib.kmx_last_error().decode()))
            return
        flops = 2.0 * ks * ks * cin * cout * 361 * b
        print("ks%d wn%d var%-5d %3d->%3d mode%d batch%-4d: %8.4f ms  %7.1f TFLOP/s (%4.1f%% of 2.5PF)" % (
            ks, wn, variant, cin, cout, mode, b, ms.value, flops / ms.value / 1e9, flops / ms.value / 1e9 / 25.0), flush=True)

    print("== 3x3 192->192 (the dominant shape of b18c384nbt) ==")
    for mode in (0, 1):
        for var in (1000, 2000, 3000):
            run(3, 3, var, 192, 192, mode)
    for var in (2032, 2001, 2002, 2004, 2005, 2012, 2013, 2016, 2064, 2256, 2257, 3256):
        run(3, 3, var, 192, 192, 1)
    run(3, 3, 2032, 192, 192, 0)
    run(3, 3, 2256, 192, 192, 0)
    print("== tile width ==")
    run(3, 2, 2000, 192, 128, 1)
    run(3, 1, 2000, 192, 192, 1)
    run(3, 1, 3000, 192, 192, 1)
    run(3, 3, 2000, 128, 192, 1)
    print("== 1x1 384->192 (pre) and 192->384 (post) ==")
    for var in (1000, 2000, 2032, 2001, 2002, 2004, 2005, 2256, 2257):
        run(1, 3, var, 384, 192, 1)
    for var in (2000, 3000):
        run(1, 1, var, 384, 192, 1)
    run(1, 2, 2000, 384, 128, 1)
    for var in (1000, 2000, 2032, 2001):
        run(1, 3, var, 192, 384, 1)
    print("== batch scaling (product kernel) ==")
    for b in (32, 64, 128, 512):
        run(3, 3, 0, 192, 192, 1, b)
        run(3, 1, 0, 192, 192, 1, b)


if __name__ == "__main__":
    main()

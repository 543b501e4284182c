#!/usr/bin/env python3
"""Times single launches of the MFMA convolution (product kernel and ablated variants) with kmx_bench_conv.
    python tools/conv_sweep.py [--batch 256] [--iters 20]
variant = depth*1000 + ablation mask (1 no epilogue, 2 no MFMA/LDS reads, 4 no DMA, 8 no LDS reads, 512 no activation, 1024 no stores)."""
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

    # cfg = 10*WNW + WN: 13 = 4-wave work-group of 96 channels, 23 = 8-wave work-group of 192 channels
    print("== 3x3 192->192 (the dominant shape of b18c384nbt) ==")
    for mode in (0, 1):
        for cfg, var in ((13, 2000), (23, 2000), (23, 3000), (23, 4000), (12, 2000), (22, 2000), (22, 3000)):
            run(3, cfg, var, 192, 192 if cfg % 10 == 3 else 128, mode)
    for cfg, d in ((13, 2000), (23, 2000)):
        for abl in (1, 2, 4, 5, 8, 512, 1024):
            run(3, cfg, d + abl, 192, 192, 1)
    print("== 1x1 384->192 (pre) and 192->384 (post) ==")
    for cfg, var in ((13, 2000), (23, 2000), (23, 3000)):
        run(1, cfg, var, 384, 192, 1)
        run(1, cfg, var, 192, 384, 1)
    for cfg, d in ((13, 2000), (23, 2000)):
        for abl in (1, 2, 4, 5):
            run(1, cfg, d + abl, 192, 384, 1)
    print("== batch scaling (product dispatch) ==")
    for b in (16, 32, 64, 128, 512):
        run(3, 13, 0, 192, 192, 1, b)
        run(3, 23, 0, 192, 192, 1, b)


if __name__ == "__main__":
    main()

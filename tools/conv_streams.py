#!/usr/bin/env python3
"""Two streams running the same 3x3 layer on their own half batch, the second one started a FRACTION of a launch later
(kmx_bench_conv_streams). Round 2 only staggered by whole launches, which leaves co-resident work-groups phase-aligned.
    python tools/conv_streams.py [launches]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lib = capi.load_library()
capi.check(lib.kmx_global_init(), lib)


def run(cfg, batch, nstreams, delay, mode):
    ms = ctypes.c_double()
    rc = lib.kmx_bench_conv_streams(3, cfg, 192, 192, batch, nstreams, ctypes.c_double(delay), launches, mode, ctypes.byref(ms))
    if rc != 0:
        return None
    return ms.value * 1e3 / launches  # us per (layer over all streams' boards), including the delay


for mode in (0, 1):
    for cfg in (13, 23):
        base = run(cfg, 256, 1, 0.0, mode)
        print("cfg%d mode%d: 1 stream x 256 boards: %.2f us per layer" % (cfg, mode, base), flush=True)
        for delay in (0, 8, 16, 24, 32, 40, 48, 56):
            v = run(cfg, 128, 2, float(delay), mode)
            # the delay itself is paid once per run: also report the rate with it subtracted
            print("cfg%d mode%d: 2 streams x 128 boards, stream 1 delayed %2d us: %.2f us per layer (%.2f without the delay)"
                  % (cfg, mode, delay, v, v - delay / launches), flush=True)

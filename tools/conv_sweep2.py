#!/usr/bin/env python3
"""Epilogue ablation (see conv_sweep.py): 512 = identity instead of mish, 1024 = no global stores."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi
lib = capi.load_library(); capi.check(lib.kmx_global_init(), lib)
def run(ks, wn, var, cin, cout, mode, b=256):
    ms = ctypes.c_double()
    rc = lib.kmx_bench_conv(ks, wn, var, cin, cout, b, 19, 19, mode, 20, ctypes.byref(ms))
    print("ks%d wn%d var%-5d %3d->%3d mode%d: %s" % (ks, wn, var, cin, cout, mode, ("%.4f ms" % ms.value) if rc == 0 else lib.kmx_last_error().decode()), flush=True)
for mode in (0, 1):
    for var in (2000, 2001, 2512, 3024, 3536):
        run(3, 3, var, 192, 192, mode)

"""Does the ORDER in which a k half's nine MFMAs are issued change what the chip sustains on noise-like operands? (round 6)
kmx_bench_mfma_sustained, shape bits 1-2: 0 weight fragment outer (the convolution's order: at every fourth MFMA both operands change), 1 the
same as a snake (exactly one operand changes between consecutive MFMAs), 2 image fragment outer. fp16, uniform noise, 256 work-groups,
A/B/C/A/B/C.   python tools/mfma_order_probe.py [seconds]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    lib = capi.load_library()
    tf, mhz = ctypes.c_double(), ctypes.c_double()
    names = ("weight fragment outer", "snake", "image fragment outer")
    for lds in (0, 1):
        for rep in range(3):
            for order in (0, 1, 2):
                capi.check(lib.kmx_bench_mfma_sustained(256, order * 2 + lds, 2, capi.PREC_FP16, seconds, ctypes.byref(tf), ctypes.byref(mhz)), lib)
                print("[mfma order] %-32s, %-21s: %7.1f TFLOP/s, %4.0f MHz" % ("step shape (LDS reads + barrier)" if lds else "bare MFMA chain", names[order], tf.value, mhz.value), flush=True)


if __name__ == "__main__":
    main()

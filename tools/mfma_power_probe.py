"""What the matrix cores of an MI355X SUSTAIN, by operand data (kmx_bench_mfma_sustained, conv_bench.hip; round 6).

The 2.5 PFLOP/s the roofline is quoted against are 256 CUs x 2.4 GHz. The convolution multiplies activations and weights that look like
noise, and the chip clocks down under such operands whatever else the kernel does. This runs the convolution's step shape (per wave and
step 18 v_mfma_f32_32x32x16 on a 3 x 3 tile of accumulators, 12 ds_read_b128, one s_barrier; 8 waves per work-group, one work-group per CU)
and the bare MFMA chain for ~2.5 s each with four kinds of operand data - zeros, a smooth ramp of small positive numbers (what `box` used to
multiply), uniform noise in [-1, 1), the distributions of the bench's own operands - in bf16 and fp16, and prints TFLOP/s over the run and the shader clock inside the last launch.

    python tools/mfma_power_probe.py [seconds] [work-groups]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    lib = capi.load_library()
    tf, mhz = ctypes.c_double(), ctypes.c_double()
    kinds = ("zeros", "smooth ramp", "uniform noise", "net-like")
    shapes = ("bare MFMA chain", "step shape (LDS reads + barrier)")
    runs = [(capi.PREC_BF16, 1, 1, wgs), (capi.PREC_BF16, 1, 2, wgs), (capi.PREC_FP16, 1, 0, wgs), (capi.PREC_FP16, 1, 1, wgs),
            (capi.PREC_FP16, 1, 2, wgs), (capi.PREC_FP16, 0, 1, wgs), (capi.PREC_FP16, 0, 2, wgs), (capi.PREC_BF16, 0, 2, wgs),
            (capi.PREC_FP16, 1, 2, wgs // 2), (capi.PREC_FP16, 0, 2, wgs // 2),
            # the distributions of the bench's own operands: normal weights of a random-init 192-channel 3x3 layer x mish of a unit normal at 1/8
            (capi.PREC_FP16, 0, 3, wgs), (capi.PREC_FP16, 1, 3, wgs), (capi.PREC_BF16, 0, 3, wgs), (capi.PREC_FP16, 0, 2, wgs)]
    for prec, shape, kind, n in runs:
        capi.check(lib.kmx_bench_mfma_sustained(n, shape, kind, prec, seconds, ctypes.byref(tf), ctypes.byref(mhz)), lib)
        print("[mfma power] %s, %-13s, %-32s, %3d work-groups: %7.1f TFLOP/s over %.1f s, shader clock in the last launch %4.0f MHz (%.3f of the 2.5 PFLOP/s nominal)"
              % (capi.PREC_NAMES[prec], kinds[kind], shapes[shape], n, tf.value, seconds, mhz.value, tf.value / 2500.0), flush=True)


if __name__ == "__main__":
    main()

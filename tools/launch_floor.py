"""What a dependent launch costs on the MI355X before it does anything (kmx_bench_launch_floor, conv_bench.hip): chains of dependent launches
with the geometry of the small-batch 3x3 shapes (512 threads, 90 KB of LDS; 18 / 192 / 256 work-groups) and of a small kernel (8 KB), inside one
hipGraph as the engine replays its schedule. mode 0: the kernel ends at once; 1: every lane loads 16 bytes the launch before stored and stores
them again; 2: the same through LDS and a barrier. A small pass of b18c384nbt is ~122 such launches."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from katago_amd import capi  # noqa: E402


def main():
    lib = capi.load_library()
    us = ctypes.c_double()
    for lds in (8192, 90112):
        for wgs in (18, 192, 256):
            row = []
            for mode in (0, 1, 2):
                capi.check(lib.kmx_bench_launch_floor(wgs, lds, mode, 122, 20, ctypes.byref(us)), lib)
                row.append(us.value)
            print("[launch floor] %3d work-groups x 512 threads, %5d bytes of LDS: %.2f us per dependent launch (empty), %.2f (load + store), %.2f (through LDS + barrier)"
                  % (wgs, lds, row[0], row[1], row[2]), flush=True)


if __name__ == "__main__":
    main()

"""Dry-run of the launch schedule under the fake HIP runtime (tests/fakehip/fakehip.cpp). Usage (see
tests/test_schedule_dryrun.py):
    LD_PRELOAD=libfakehip.so KMX_FAKEHIP_LOG=out.log python run_schedule.py <libkatamx.so> <model file> <max batch> <n> [<n> ...]
Loads the library with plain ctypes (no torch: its own HIP runtime must not be the one that answers), builds a handle and
evaluates batches of the given sizes through kmx_eval on zero rows. No kernel runs; the log is the product."""
import ctypes
import sys

import numpy as np


def main():
    lib_path, model_path, max_batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    sizes = [int(x) for x in sys.argv[4:]]
    lib = ctypes.CDLL(lib_path)
    lib.kmx_last_error.restype = ctypes.c_char_p

    def check(rc, what):
        if rc != 0:
            raise SystemExit("%s failed: %d %s" % (what, rc, (lib.kmx_last_error() or b"").decode()))

    check(lib.kmx_global_init(), "kmx_global_init")
    model = ctypes.c_void_p()
    check(lib.kmx_model_load(model_path.encode(), None, ctypes.byref(model)), "kmx_model_load")
    ctx = ctypes.c_void_p()
    gpu = (ctypes.c_int * 1)(0)
    check(lib.kmx_context_create(gpu, 1, 19, 19, 0, ctypes.byref(ctx)), "kmx_context_create")
    handle = ctypes.c_void_p()
    check(lib.kmx_handle_create(ctx, model, max_batch, 1, 0, ctypes.byref(handle)), "kmx_handle_create")
    S = 361
    FP = ctypes.POINTER(ctypes.c_float)
    for n in sizes:
        sp = np.zeros((n, S, 22), np.float32)
        sp[:, :, 0] = 1.0
        gl = np.zeros((n, 19), np.float32)
        pol = np.zeros((n, S + 1), np.float32)
        own = np.zeros((n, S), np.float32)
        val = np.zeros((n, 3), np.float32)
        sc = np.zeros((n, 6), np.float32)
        sym = (np.arange(n) % 8).astype(np.int32)
        opt = np.zeros(n, np.float32)
        rows = lambda a: (FP * n)(*[a[i].ctypes.data_as(FP) for i in range(n)])
        print("eval n=%d" % n, flush=True)
        check(lib.kmx_eval(handle, n, rows(sp), rows(gl), sym.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), opt.ctypes.data_as(FP),
                           rows(pol), val.ctypes.data_as(FP), sc.ctypes.data_as(FP), rows(own)), "kmx_eval")
    lib.kmx_handle_free(handle)
    lib.kmx_context_free(ctx)
    lib.kmx_model_free(model)
    print("done")


if __name__ == "__main__":
    main()

"""Two fake devices, one server thread per device - the reference's in-process multi-GPU mode (cpp/neuralnet/nneval.cpp:399-407:
one NNEvaluator server thread per GPU, createComputeHandle called ON that thread with its gpuIdxForThisThread) - under the fake
HIP runtime (tests/fakehip/fakehip.cpp with KMX_FAKEHIP_DEVICES=2). Thread k owns a plain handle and a leaf batcher on device k;
the batcher's own dispatcher and completion threads never had a device made current by the caller. The log must hold launches
of both devices and not one VIOLATION line (a stream, event or launch touched while another device was current).
    LD_PRELOAD=libfakehip.so KMX_FAKEHIP_DEVICES=2 KMX_FAKEHIP_LOG=out.log python run_two_devices.py <libkatamx.so> <model>"""
import ctypes
import sys
import threading

import numpy as np


def main():
    lib = ctypes.CDLL(sys.argv[1])
    model_path = sys.argv[2]
    lib.kmx_last_error.restype = ctypes.c_char_p
    errors = []

    def check(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %d %s" % (what, rc, (lib.kmx_last_error() or b"").decode()))

    check(lib.kmx_global_init(), "kmx_global_init")
    n_dev = lib.kmx_device_count()
    assert n_dev == 2, n_dev
    model = ctypes.c_void_p()
    check(lib.kmx_model_load(model_path.encode(), None, ctypes.byref(model)), "kmx_model_load")
    ctx = ctypes.c_void_p()
    gpus = (ctypes.c_int * 2)(0, 1)
    check(lib.kmx_context_create(gpus, 2, 19, 19, 0, ctypes.byref(ctx)), "kmx_context_create")
    S = 361
    FP = ctypes.POINTER(ctypes.c_float)

    def server(k):
        try:
            handle = ctypes.c_void_p()
            check(lib.kmx_handle_create(ctx, model, 16, 1, k, ctypes.byref(handle)), "kmx_handle_create")
            batcher = ctypes.c_void_p()
            check(lib.kmx_batcher_create(ctx, model, 8, 2, k, ctypes.byref(batcher)), "kmx_batcher_create")
            for n in (1, 5, 16):
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
                check(lib.kmx_eval(handle, n, rows(sp), rows(gl), sym.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), opt.ctypes.data_as(FP),
                                   rows(pol), val.ctypes.data_as(FP), sc.ctypes.data_as(FP), rows(own)), "kmx_eval")
                tickets = []
                for i in range(n):
                    t = ctypes.c_uint64()
                    check(lib.kmx_batcher_submit(batcher, sp[i].ctypes.data_as(FP), gl[i].ctypes.data_as(FP), None, int(sym[i]), ctypes.c_float(0.0),
                                                 pol[i].ctypes.data_as(FP), val[i].ctypes.data_as(FP), sc[i].ctypes.data_as(FP), own[i].ctypes.data_as(FP),
                                                 ctypes.byref(t)), "kmx_batcher_submit")
                    tickets.append(t.value)
                for t in tickets:
                    check(lib.kmx_batcher_wait(batcher, ctypes.c_uint64(t)), "kmx_batcher_wait")
            # per-launch profiling reads events back on the caller's thread
            check(lib.kmx_handle_set_profiling(handle, 1), "kmx_handle_set_profiling")
            check(lib.kmx_handle_set_profiling(handle, 0), "kmx_handle_set_profiling")
            lib.kmx_batcher_free(batcher)
            lib.kmx_handle_free(handle)
        except Exception as e:  # noqa: BLE001
            errors.append("device %d: %s" % (k, e))

    lib.kmx_batcher_wait.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    lib.kmx_batcher_free.argtypes = [ctypes.c_void_p]
    lib.kmx_handle_free.argtypes = [ctypes.c_void_p]
    threads = [threading.Thread(target=server, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    lib.kmx_context_free(ctx)
    lib.kmx_model_free(model)
    if errors:
        raise SystemExit("\n".join(errors))
    print("done")


if __name__ == "__main__":
    main()

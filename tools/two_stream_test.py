#!/usr/bin/env python3
"""Does splitting a batch of 256 into two half-batches on two HIP streams (two compute handles) raise throughput?
The two kernel streams drift apart, so one half's memory-bound phases (residual fetch, epilogue stores) can overlap
the other's MFMA loops instead of all work-groups hitting HBM at the same moment."""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from katago_amd import capi, modelgen, nninterface as nn  # noqa: E402
from bench import synthetic_rows  # noqa: E402


def run(handles, B, steps=60, warmup=5):
    lib = capi.load_library()
    S = 361
    bufs = []
    for i, h in enumerate(handles):
        sp, gl = synthetic_rows(B, 100 + i)
        bufs.append(dict(sp=torch.from_numpy(sp).cuda(), gl=torch.from_numpy(gl).cuda(),
                         sym=(np.arange(B) % 8).astype(np.int32), opt=np.zeros(B, dtype=np.float32),
                         pol=torch.empty((B, S + 1), device="cuda"), val=torch.empty((B, 3), device="cuda"),
                         sc=torch.empty((B, 6), device="cuda"), own=torch.empty((B, S), device="cuda")))
    torch.cuda.synchronize()

    def step():
        for h, b in zip(handles, bufs):
            capi.check(lib.kmx_eval_device(h._p, B, b["sp"].data_ptr(), b["gl"].data_ptr(),
                                           b["sym"].ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                           b["opt"].ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                           b["pol"].data_ptr(), b["val"].data_ptr(), b["sc"].data_ptr(), b["own"].data_ptr(), 0), lib)

    for _ in range(warmup):
        step()
    for h in handles:
        h.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for h in handles:
        h.sync()
    el = time.perf_counter() - t0
    return steps * B * len(handles) / el


def main():
    nn.globalInitialize()
    path = "/tmp/kmx_two_stream.bin"
    modelgen.write_model(path, "b18c384nbt", seed=1)
    model = nn.loadModelFile(path)
    ctx = nn.createComputeContext([0], 19, 19, precision="bf16")
    for nh, B in ((1, 256), (2, 128), (4, 64), (8, 32), (4, 128), (2, 128), (4, 64)):
        hs = [nn.createComputeHandle(ctx, model, B, True, 0) for _ in range(nh)]
        v = run(hs, B)
        print("%d handle(s) x batch %3d : %9.1f evals/s" % (nh, B, v), flush=True)
        for h in hs:
            h.close()


if __name__ == "__main__":
    main()

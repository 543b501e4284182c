"""Small batches on the MI355X: ms per pass of b18c384nbt 19x19 through kmx_eval (host rows) at batch 1 ... 64, the 3x3
convolution's time per launch at batch 1 / 8 (hipEvents), and a digest of the outputs of fixed rows - the same rows must give the
same bits at every batch size and with either work-group shape (KMX_CONV_TUNE=loaders=0 / 1; run once each, compare the digests)."""
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from katago_amd import capi, modelgen, nninterface as nn  # noqa: E402
from conftest import make_rows  # noqa: E402


def main():
    nn.globalInitialize()
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "kmx_scan_b18.bin")
    if not os.path.exists(path):
        modelgen.write_model(path, "b18c384nbt", seed=5)
    ctx = nn.createComputeContext([0], 19, 19)
    model = nn.loadModelFile(path)
    rng = np.random.default_rng(3)
    sp, gl = make_rows(rng, 64, 19, [(19, 19), (19, 19), (13, 13), (9, 9)] * 16)
    sym = (np.arange(64) % 8).astype(np.int32)
    h = nn.createComputeHandle(ctx, model, 256)
    lib = h._lib
    out = {"tune": os.environ.get("KMX_CONV_TUNE", "default"), "precision": h.precision, "ms_per_pass": {}, "rows_per_s": {}}
    ref = nn.getOutput(h, sp, gl, sym)  # 64 rows: the 4-wave shapes
    same = True
    for n in (1, 2, 8, 16, 20, 42, 48):
        got = nn.getOutput(h, sp[:n], gl[:n], sym[:n])
        same = same and all(np.array_equal(got[k], ref[k][:n]) for k in ref)
    big = nn.getOutput(h, np.tile(sp, (4, 1, 1)), np.tile(gl, (4, 1)), np.tile(sym, 4))  # 256 rows: the 8-wave shapes, two streams
    same = same and all(np.array_equal(big[k][:64], ref[k]) for k in ref)
    out["rows_bit_identical_across_batch_sizes"] = bool(same)
    out["digest"] = hashlib.sha1(b"".join(np.ascontiguousarray(ref[k]).tobytes() for k in sorted(ref))).hexdigest()
    big85 = nn.getOutput(h, np.tile(sp, (2, 1, 1))[:85], np.tile(gl, (2, 1))[:85], np.tile(sym, 2)[:85])
    same = same and all(np.array_equal(big85[k][:64], ref[k]) for k in ref)
    out["rows_bit_identical_across_batch_sizes"] = bool(same)
    for n in (1, 2, 4, 8, 15, 16, 18, 21, 24, 32, 42, 48, 64, 85):
        spn, gln, symn = (np.tile(sp, (2, 1, 1))[:n], np.tile(gl, (2, 1))[:n], np.tile(sym, 2)[:n]) if n > 64 else (sp[:n], gl[:n], sym[:n])
        for _ in range(3):
            nn.getOutput(h, spn, gln, symn)
        t0 = time.perf_counter()
        reps = 25
        for _ in range(reps):
            nn.getOutput(h, spn, gln, symn)
        ms = (time.perf_counter() - t0) / reps * 1e3
        out["ms_per_pass"][n] = round(ms, 3)
        out["rows_per_s"][n] = round(n / ms * 1e3)
    capi.check(lib.kmx_handle_set_split_min(h._p, 0), lib)
    for n in (1, 8):
        capi.check(lib.kmx_handle_set_profiling(h._p, 1), lib)
        for _ in range(5):
            nn.getOutput(h, sp[:n], gl[:n], sym[:n])
        ent = (capi.ProfileEntry * 32)()
        cnt = ctypes.c_int()
        capi.check(lib.kmx_handle_get_profile(h._p, ent, 32, ctypes.byref(cnt)), lib)
        out["us_per_launch_batch_%d" % n] = {ent[i].name.decode(): round(ent[i].total_ms / max(ent[i].launches, 1) * 1e3, 2) for i in range(cnt.value)}
        capi.check(lib.kmx_handle_set_profiling(h._p, 0), lib)
    h.close()
    print("SCAN " + json.dumps(out))


if __name__ == "__main__":
    main()

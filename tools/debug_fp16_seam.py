"""Stress for nondeterminism: two handles (KMX_FUSE_SEAMS=0/1) alive together, repeated passes, every output compared with the
first unfused pass. Prints which handle deviates, in which rows."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from katago_amd import nninterface as nn, modelgen
from conftest import make_rows

nn.globalInitialize()
tmp = tempfile.mkdtemp()
p = os.path.join(tmp, "b18.bin")
modelgen.write_model(p, "b18c384nbt", seed=31)
rng = np.random.default_rng(31)
sp, gl = make_rows(rng, 256, 19, [(19, 19), (13, 13), (9, 9), (19, 10)] * 64)
sym = rng.integers(0, 8, 256).astype(np.int32)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for dtype in ("bf16", "fp16"):
    ctx = nn.createComputeContext([0], 19, 19, precision=dtype)
    model = nn.loadModelFile(p)
    os.environ["KMX_FUSE_SEAMS"] = "0"
    h0 = nn.createComputeHandle(ctx, model, 256)
    os.environ["KMX_FUSE_SEAMS"] = "1"
    h1 = nn.createComputeHandle(ctx, model, 256)
    ref = {}
    nbad = 0
    for rep in range(reps):
        for n in (256, 100, 37, 24, 5):
            for name, h in (("unfused", h0), ("fused", h1)):
                o = nn.getOutput(h, sp[:n], gl[:n], sym[:n])
                if n not in ref:
                    ref[n] = o
                    continue
                for k in o:
                    d = np.abs(o[k].astype(np.float64) - ref[n][k].astype(np.float64)).reshape(n, -1)
                    bad = np.where(~np.isfinite(d).all(axis=1) | (d.max(axis=1) > 0))[0]
                    if len(bad):
                        nbad += 1
                        print(dtype, "rep", rep, "n", n, name, k, "rows", bad[:16].tolist(), "of", len(bad), "max", float(np.nanmax(d)), flush=True)
    print(dtype, "deviating outputs:", nbad, flush=True)

#!/usr/bin/env python3
"""One-shot GPU bring-up report: every kernel family against the CPU oracle, with enough detail on a
mismatch to localise a layout bug from a single run, plus a first timing of the full net.
Usage (on the GPU box):  python tools/gpu_selftest.py [--quick] [--dtypes bf16,fp16]
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from katago_amd import capi, modelgen, nninterface as nn  # noqa: E402
from oracle import oracle  # noqa: E402

FAILS = []


def report(name, got, want, tol_rel, tol_abs, detail_axes=None):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    lim = tol_abs + tol_rel * np.maximum(np.abs(want), np.abs(got))
    bad = err > lim
    nbad = int(bad.sum())
    finite = bool(np.isfinite(got).all())
    status = "ok  " if (nbad == 0 and finite) else "FAIL"
    print("[%s] %-46s maxerr %.4g  rms %.4g  scale %.3g  bad %d/%d%s" % (
        status, name, err.max() if err.size else 0, np.sqrt((err ** 2).mean()) if err.size else 0,
        np.abs(want).max() if want.size else 0, nbad, bad.size, "" if finite else "  NON-FINITE"), flush=True)
    if nbad or not finite:
        FAILS.append(name)
        idx = np.argwhere(bad)[:8]
        for i in idx:
            print("        at %s got %.5g want %.5g" % (tuple(int(v) for v in i), got[tuple(i)], want[tuple(i)]))
        if detail_axes is not None and got.ndim >= 2:
            for ax, label, mod in detail_axes:
                other = tuple(a for a in range(got.ndim) if a != ax)
                frac = bad.mean(axis=other)
                if mod:
                    m = np.zeros(mod)
                    for k in range(len(frac)):
                        m[k % mod] += frac[k]
                    frac = m / max(1, len(bad.mean(axis=other)) // mod)
                print("        bad fraction by %s: %s" % (label, np.array2string(frac, precision=2, max_line_width=200)))
    return nbad == 0 and finite


def rand_inputs(rng, n, X, Y, xs=None, ys=None):
    S = X * Y
    sp = np.zeros((n, Y, X, 22), dtype=np.float32)
    for b in range(n):
        bx = xs[b] if xs else X
        by = ys[b] if ys else Y
        sp[b, :by, :bx, 0] = 1
        st = rng.random((by, bx))
        sp[b, :by, :bx, 1] = st < 0.25
        sp[b, :by, :bx, 2] = (st >= 0.25) & (st < 0.5)
        for c in range(3, 22):
            sp[b, :by, :bx, c] = rng.random((by, bx)) < 0.08
    gl = rng.normal(0, 0.5, (n, 19)).astype(np.float32)
    return sp.reshape(n, S, 22), gl


def tol_for(dtype):
    return (0.02, 0.03) if dtype == "bf16" else (0.004, 0.006)


def layer_checks(dtype, quick):
    rng = np.random.default_rng(3)
    rel, ab = tol_for(dtype)
    cases = [(1, 32, 64, 19, 19, 2), (3, 32, 64, 19, 19, 2), (3, 192, 192, 19, 19, 2), (1, 384, 192, 19, 19, 1),
             (3, 22, 96, 19, 19, 1), (5, 22, 64, 19, 19, 1), (3, 64, 48, 9, 13, 3), (1, 96, 4, 7, 7, 2), (3, 128, 192, 19, 19, 1)]
    if quick:
        cases = cases[:4]
    for ks, cin, cout, X, Y, n in cases:
        w = (rng.standard_normal((cout, cin, ks, ks)) * np.sqrt(1.0 / (ks * ks * cin))).astype(np.float32)
        # asymmetric, channel/cell-dependent input so that any transpose or permutation shows up
        x = rng.standard_normal((n, Y, X, cin)).astype(np.float32)
        x += (np.arange(cin) % 7 - 3)[None, None, None, :] * 0.1 + (np.arange(X) % 5)[None, None, :, None] * 0.05
        want = oracle.testEvaluateConv(w, n, X, Y, x)
        try:
            got = nn.testEvaluateConv(w, n, X, Y, dtype, x)
            report("conv%dx%d %d->%d %dx%d n%d [%s]" % (ks, ks, cin, cout, X, Y, n, dtype), got.reshape(n, X * Y, cout),
                   want.reshape(n, X * Y, cout), rel, ab * 3, detail_axes=[(2, "cout%32", 32), (1, "cell%32", 32)])
        except Exception as e:
            FAILS.append("conv exception")
            print("[FAIL] conv%dx%d %d->%d exception: %s" % (ks, ks, cin, cout, e))
    # batch norm with mask
    n, X, Y, C = 2, 9, 7, 20
    x = rng.standard_normal((n, Y, X, C)).astype(np.float32)
    mask = (rng.random((n, Y, X)) < 0.8).astype(np.float32)
    sc, bi = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 0.3, C).astype(np.float32)
    for act in (capi.ACT_IDENTITY, capi.ACT_RELU, capi.ACT_MISH, capi.ACT_SILU):
        want = oracle.testEvaluateBatchNorm(sc, bi, act, n, X, Y, x, mask)
        got = nn.testEvaluateBatchNorm(sc, bi, act, n, X, Y, dtype, x, mask)
        report("bnact act=%d [%s]" % (act, dtype), got, want, rel, ab)

    def bn(c, act):
        return (rng.uniform(0.6, 1.4, c).astype(np.float32), rng.normal(0, 0.25, c).astype(np.float32), act)

    def cw(co, ci, k, g=1.0):
        return (rng.standard_normal((co, ci, k, k)) * g * np.sqrt(2.0 / (k * k * ci))).astype(np.float32)

    for (C, M, X, Y, n) in ([(64, 64, 19, 19, 2), (192, 192, 19, 19, 1)] if not quick else [(64, 64, 19, 19, 2)]):
        blk = dict(pre=bn(C, capi.ACT_MISH), conv1=cw(M, C, 3), mid=bn(M, capi.ACT_MISH), conv2=cw(C, M, 3, 0.5))
        x = rng.standard_normal((n, Y, X, C)).astype(np.float32)
        mask = np.ones((n, Y, X), dtype=np.float32)
        mask[0, :, X - 3:] = 0
        x *= mask[..., None]
        want = oracle.testEvaluateResidualBlock(blk, n, X, Y, x, mask)
        got = nn.testEvaluateResidualBlock(blk, n, X, Y, dtype, x, mask)
        on = mask.reshape(n, -1) > 0
        report("resblock C%d [%s] (on-board cells)" % (C, dtype), got.reshape(n, X * Y, C)[on], want.reshape(n, X * Y, C)[on], rel, ab * 3)
    for (C, R, G, X, Y, n) in ([(64, 32, 16, 13, 13, 2), (192, 128, 64, 19, 19, 1)] if not quick else [(64, 32, 16, 13, 13, 2)]):
        blk = dict(pre=bn(C, capi.ACT_MISH), convr=cw(R, C, 3), convg=cw(G, C, 3), gbn=bn(G, capi.ACT_MISH),
                   gmul=(rng.standard_normal((3 * G, R)) * 0.5 / np.sqrt(3 * G)).astype(np.float32), mid=bn(R, capi.ACT_MISH),
                   conv2=cw(C, R, 3, 0.5))
        x = rng.standard_normal((n, Y, X, C)).astype(np.float32)
        mask = np.ones((n, Y, X), dtype=np.float32)
        mask[0, Y - 4:, :] = 0
        x *= mask[..., None]
        want = oracle.testEvaluateGlobalPoolingResidualBlock(blk, n, X, Y, x, mask)
        got = nn.testEvaluateGlobalPoolingResidualBlock(blk, n, X, Y, dtype, x, mask)
        on = mask.reshape(n, -1) > 0
        report("gpoolblock C%d [%s] (on-board cells)" % (C, dtype), got.reshape(n, X * Y, C)[on], want.reshape(n, X * Y, C)[on], rel, ab * 3)


def compare_outputs(tag, o, w, mask, rel, ab):
    ok = True
    full = np.concatenate([mask, np.ones((mask.shape[0], 1), bool)], axis=1)
    ok &= report(tag + " policy(on-board+pass)", o["policy"][full], w["policy"][full], rel, ab * 4)
    ok &= report(tag + " value", o["value"], w["value"], rel, ab * 2)
    ok &= report(tag + " score", o["score"], w["score"], rel, ab * 2)
    ok &= report(tag + " ownership(on-board)", o["ownership"][mask], w["ownership"][mask], rel, ab * 4)
    return ok


def model_checks(dtype, quick, tmpdir):
    rng = np.random.default_rng(11)
    rel, ab = tol_for(dtype)
    rel, ab = rel * 2.5, ab * 2.5
    # reference-torch golden net (masks: 13x9 and 9x9 boards inside the 19x19 buffer)
    gold = np.load(os.path.join(REPO, "tests", "golden", "torch_nbt_vectors.npz"))
    path = os.path.join(REPO, "tests", "golden", "torch_nbt.bin.gz")
    ctx = nn.createComputeContext([0], 19, 19, precision=dtype)
    model = nn.loadModelFile(path)
    h = nn.createComputeHandle(ctx, model, 8)
    mask = gold["spatial_nhwc"][:, :, 0] > 0
    for opt in (0.0, 1.0):
        o = nn.getOutput(h, gold["spatial_nhwc"], gold["glob"], None, np.full(4, opt, np.float32))
        w = {"policy": gold["policy"][:, int(opt), :], "value": gold["value"], "score": gold["score"], "ownership": gold["ownership"]}
        compare_outputs("torch-golden opt=%g [%s]" % (opt, dtype), o, w, mask, rel, ab)
    h.close()
    archs = [("b2c32nbt", 5, 3), ("b6c96", 5, 3), ("b18c384nbt", 6, 3)]
    if quick:
        archs = archs[:2]
    for arch, n, stem in archs:
        p = os.path.join(tmpdir, arch + ".bin")
        if not os.path.exists(p):
            modelgen.write_model(p, arch, stem_kernel=5 if arch == "b6c96" else 3)
        om = oracle.loadModelFile(p)
        model = nn.loadModelFile(p)
        h = nn.createComputeHandle(ctx, model, 16)
        xs = [19, 19, 13, 9, 19, 7][:n]
        ys = [19, 19, 13, 9, 10, 11][:n]
        sp, gl = rand_inputs(rng, n, 19, 19, xs, ys)
        sym = np.array([0, 5, 3, 6, 7, 1][:n], dtype=np.int32)
        opt = np.array([0, 0.3, 1.0, 0.0, 0.5, 0.2][:n], dtype=np.float32)
        t0 = time.time()
        w = oracle.getOutput(om, 19, 19, sp, gl, sym, opt)
        t1 = time.time()
        o = nn.getOutput(h, sp, gl, sym, opt)
        # on-board mask in OUTPUT (unsymmetrised) coordinates = the original input mask
        mask = sp[:, :, 0] > 0
        print("  (%s: oracle %.2fs for %d rows, hip %.3fs incl. first-launch)" % (arch, t1 - t0, n, time.time() - t1))
        compare_outputs("%s [%s]" % (arch, dtype), o, w, mask, rel, ab)
        # batching determinism: row results must not depend on batch composition
        o1 = nn.getOutput(h, sp[:1], gl[:1], sym[:1], opt[:1])
        report("%s batch-of-1 == row 0 of batch [%s]" % (arch, dtype), o1["policy"][0], o["policy"][0], 0, 1e-6)
        h.close()
    ctx.close()


def timing(dtype, tmpdir, batch=256, steps=5):
    import torch
    p = os.path.join(tmpdir, "b18c384nbt.bin")
    if not os.path.exists(p):
        modelgen.write_model(p, "b18c384nbt")
    ctx = nn.createComputeContext([0], 19, 19, precision=dtype)
    model = nn.loadModelFile(p)
    h = nn.createComputeHandle(ctx, model, batch)
    rng = np.random.default_rng(5)
    sp, gl = rand_inputs(rng, batch, 19, 19)
    dsp = torch.from_numpy(sp).cuda()
    dgl = torch.from_numpy(gl).cuda()
    S = 361
    dpol = torch.empty((batch, S + 1), device="cuda")
    dval = torch.empty((batch, 3), device="cuda")
    dsc = torch.empty((batch, 6), device="cuda")
    down = torch.empty((batch, S), device="cuda")
    torch.cuda.synchronize()
    lib = capi.load_library()

    def run(sync):
        capi.check(lib.kmx_eval_device(h._p, batch, dsp.data_ptr(), dgl.data_ptr(), None, None, dpol.data_ptr(), dval.data_ptr(),
                                       dsc.data_ptr(), down.data_ptr(), 1 if sync else 0), lib)

    run(True)
    run(True)
    t0 = time.time()
    for _ in range(steps):
        run(False)
    h.sync()
    dt = (time.time() - t0) / steps
    flops = model.info.flops_per_position * S * batch
    print("[time] b18c384nbt batch %d %s: %.3f ms/step  %.0f evals/s  %.1f TFLOP/s (%.1f%% of 2.5 PF)" % (
        batch, dtype, dt * 1e3, batch / dt, flops / dt / 1e12, flops / dt / 2.5e15 * 100), flush=True)
    capi.check(lib.kmx_handle_set_profiling(h._p, 1), lib)
    for _ in range(3):
        run(False)
    ent = (capi.ProfileEntry * 32)()
    cnt = __import__("ctypes").c_int()
    capi.check(lib.kmx_handle_get_profile(h._p, ent, 32, __import__("ctypes").byref(cnt)), lib)
    for i in range(cnt.value):
        e = ent[i]
        ms = e.total_ms / max(e.launches, 1)
        print("[prof] %-16s launches %4d  avg %.4f ms  %.1f TFLOP/s  %.2f TB/s(alg)" % (
            e.name.decode(), e.launches, ms, e.flops / max(e.total_ms, 1e-9) / 1e9, e.bytes / max(e.total_ms, 1e-9) / 1e9))
    capi.check(lib.kmx_handle_set_profiling(h._p, 0), lib)
    print("  sample outputs: value[0] %s policy[0,:4] %s" % (dval[0].cpu().numpy(), dpol[0, :4].cpu().numpy()))
    h.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--dtypes", default="bf16,fp16")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--tmpdir", default="/tmp/kmx_models")
    a = ap.parse_args()
    os.makedirs(a.tmpdir, exist_ok=True)
    nn.globalInitialize()
    nn.printDevices()
    for dtype in a.dtypes.split(","):
        for fn in (lambda: layer_checks(dtype, a.quick), lambda: model_checks(dtype, a.quick, a.tmpdir)):
            try:
                fn()
            except Exception:
                FAILS.append("exception")
                traceback.print_exc()
        if not a.no_timing:
            try:
                timing(dtype, a.tmpdir)
            except Exception:
                FAILS.append("timing exception")
                traceback.print_exc()
    print("SELFTEST %s (%d failing checks)%s" % ("PASSED" if not FAILS else "FAILED", len(FAILS), "" if not FAILS else ": " + "; ".join(FAILS[:12])))
    return 0 if not FAILS else 1


if __name__ == "__main__":
    sys.exit(main())

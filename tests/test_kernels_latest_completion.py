"""The kernels' asynchronous-completion logic on the CPU (tests/fakehip/README.md, "Asynchronous completion"): conv_kernel.h and
the seam kernels emulated with every LDS-DMA copy landing as LATE as the issuing wave's s_waitcnt vmcnt(N) statements allow
(KMX_EMU_LATE_DMA=1; requests retire in order). Compile-time request counts, ring depths and barrier placement must be right for
the answers to be - and the second test shows that they are not when a count is one too generous. (A file of its own so that the
CPU suite's workers share the emulated builds' time; helpers and rewrite rules live in test_engine_emulated.py.)"""
import json
import os
import sys

from test_engine_emulated import PW2_CODE, build_emu_full, conv_only, emu_full_lib, run_parallel  # noqa: F401  (emu_full_lib is a fixture)


def test_convolution_waitcnt_logic_with_latest_completion(emu_full_lib):
    """KMX_EMU_LATE_DMA=1: an LDS-DMA copy lands only when an s_waitcnt vmcnt(N) of its wave forces it (requests retire in order) -
    the latest the hardware may complete it; =2: at the barrier that follows that wait - the latest another wave may first see it
    (lanes are OS threads here: a wave that only fetches reaches its next wait long before the others have read anything). The convolution's compile-time counts, ring depths and barrier placement must be
    right for the answers to be: 4-wave and 8-wave shapes, 1x1 / 3x3 / 5x5."""
    for rc, so, se in run_parallel([conv_only(emu_full_lib, {"KMX_EMU_LATE_DMA": mode}) for mode in ("1", "2")]):
        assert rc == 0 and "RESULT " in so, (so + se)[-3000:]
        for k, v in json.loads(so.split("RESULT ")[1]).items():
            assert v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5, (k, v)


def test_latest_completion_catches_a_wrong_count(tmp_path):
    """The emulator's teeth: the same kernels with every top-of-step wait ONE request too generous (the padded shapes' VMCNT
    constant, the loader waves' computed count). With immediate copies nothing shows; with the latest legal completion the slab a
    step needs has not landed and the answers are wrong."""
    lib = build_emu_full(str(tmp_path), conv_mutations=[
        (r"static constexpr int VMCNT = SPREAD \? PPS \+ \(D - 2\) \* \(NPW \+ PPS\) : \(D - 2\) \* \(NPW \+ NPA\);",
         "static constexpr int VMCNT = 1 + (SPREAD ? PPS + (D - 2) * (NPW + PPS) : (D - 2) * (NPW + NPA));", 1),
        (r"        waitVmSel\(n\);\n      \}\n      return;", "        waitVmSel(n + 1);\n      }\n      return;", 1),
        # the weight waves of the role-split shapes (the twelve-wave small-batch shape among them)
        (r"static constexpr int VMCNT_W = \(D - 2\) \* NPW,", "static constexpr int VMCNT_W = 1 + (D - 2) * NPW,", 1),
    ], small_mutations=[
        # the fetching waves of the small-batch shape let one request more stay in flight than slab s + 1 allows
        (r"if\(slabWave\) convk::waitVmSel\(G::vmAt\(t\)\);", "if(slabWave) convk::waitVmSel(G::vmAt(t) + 1);", 1),
        # the shape with its weights in registers: a multiplying wave lets one request more stay in flight than the 16 younger fragments
        # (the fragment it is about to multiply may then not have landed), a fetching wave lets the image a barrier publishes stay in flight
        (r'waitFragSel\(wf\[slot\]\[wn\], younger \* WN\);', 'waitFragSel(wf[slot][wn], younger * WN + 1);', 1),
        (r'        waitVm<0>\(\);  // image chunk \+ 1', '        waitVm<NPA>();  // image chunk + 1', 1),
    ], pw2_mutations=[
        # the persistent seam kernel: phase 2 of a part lets one request more stay in flight than its loads and stores account for
        (r"waitVmSel\(G::nR\(q\) \+ 2\);", "waitVmSel(G::nR(q) + 2 + 1);", 1),
    ], pw3_mutations=[
        # the seam kernel with resident weights: the wait for the next tile's X lets one request more stay in flight than a work-group's
        # first tile issues after it - the last chunk of that X may then land after barrier BA1 has published it
        (r"static constexpr int VM_AFTER_X = N_RS \+ N_RS \+ N_S1S \+ N_S1S;", "static constexpr int VM_AFTER_X = N_RS + N_RS + N_S1S + N_S1S + 1;", 1),
    ])
    runs = run_parallel([conv_only(lib, {"KMX_EMU_LATE_DMA": late}) for late in ("0", "1")])
    (rc0, so0, se0), (rc1, so1, se1) = runs
    assert rc0 == 0 and rc1 == 0, (so0 + se0 + so1 + se1)[-3000:]
    early, late = json.loads(so0.split("RESULT ")[1]), json.loads(so1.split("RESULT ")[1])
    for k, v in early.items():
        assert v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5, ("immediate copies hide the defect", k, v)
    wrong = [k for k, v in late.items() if not v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5]
    print("late completion, one request too generous:", late)
    assert "conv3_64_32" in wrong and "conv3_96_192" in wrong, late  # the padded 4-wave shape and the 8-wave loader shape
    # the small-batch shape with its fetching waves' wait one request too generous
    runs = run_parallel([([sys.executable, "-c", CW12_CODE, lib], dict(os.environ, KMX_CONV_TUNE="loaders=1,regw=0", KMX_EMU_LATE_DMA=late_)) for late_ in ("0", "2")])
    (rc0, so0, se0), (rc1, so1, se1) = runs
    assert rc0 == 0 and rc1 == 0, (so0 + se0 + so1 + se1)[-3000:]
    c0, c1 = json.loads(so0.split("RESULT ")[1]), json.loads(so1.split("RESULT ")[1])
    ok = lambda v: v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5  # noqa: E731
    print("fetching-waves shape, one request too generous: latest completion", {k: v[:2] for k, v in c1.items()})
    # (the fetching waves run ahead of the multiplying ones between two barriers, so under emulation a late copy may still land before a
    # small case reads it: the defect must show in at least one case - the 19x19 one in practice - and never with immediate copies)
    assert all(ok(v) for v in c0.values()) and not all(ok(v) for v in c1.values()), (c0, c1)
    # ... and with its weights in registers: the fragment wait (mode 1: a stale fragment is multiplied) and, with cell tiles split, the image wait
    runs = run_parallel([([sys.executable, "-c", CW12_CODE, lib], dict(os.environ, KMX_CONV_TUNE="loaders=1,loaders_split=%s,regw=%s" % (sp_, rw_), KMX_EMU_LATE_DMA=late_))
                         for sp_, late_, rw_ in (("0", "0", "1"), ("0", "1", "1"), ("1", "2", "3"), ("0", "1", "2,loaders_max_wgs=0"))])
    for rc, so, se in runs:
        assert rc == 0, (so + se)[-3000:]
    c0, c1, c2, c3 = (json.loads(so.split("RESULT ")[1]) for rc, so, se in runs)
    print("weights in registers, one request too generous: latest completion", {k: v[:2] for k, v in c1.items()}, {k: v[:2] for k, v in c2.items()},
          {k: v[:2] for k, v in c3.items()})
    assert all(ok(v) for v in c0.values()) and not any(ok(v) for v in c1.values()) and not all(ok(v) for v in c2.values()), (c0, c1, c2)
    assert not all(ok(v) for v in c3.values()), c3  # the 64-channel shape where it is taken (an even number of channel tiles)
    # the seam kernel with its defect: right with immediate copies, wrong when a W2 slab may land as late as the count allows
    for kern in ("2", "3"):  # round 3's persistent kernel, round 5's with resident weights
        runs = run_parallel([([sys.executable, "-c", PW2_CODE, lib], dict(os.environ, KMX_PW_KERNEL=kern, KMX_PW_GRID="1", KMX_EMU_LATE_DMA=late_))
                             for late_ in ("0", "1")])
        (rc0, so0, se0), (rc1, so1, se1) = runs
        assert rc0 == 0 and rc1 == 0, (so0 + se0 + so1 + se1)[-3000:]
        r0, r1 = json.loads(so0.split("RESULT ")[1]), json.loads(so1.split("RESULT ")[1])
        print("seam kernel %s, one request too generous: immediate" % kern, r0["same"], "latest", r1["same"])
        assert all(r0["same"]) and not all(r1["same"]), (kern, r0, r1)


CW12_CODE = r"""
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn
rng = np.random.default_rng(7)
out = {}
for (cin, cout, X, Y, n) in ((64, 32, 9, 9, 1), (96, 192, 13, 13, 2), (40, 200, 19, 19, 1), (64, 64, 13, 9, 1), (32, 96, 7, 11, 3), (160, 32, 9, 9, 1)):
    w = (rng.normal(size=(cout, cin, 3, 3)) * 0.1).astype(np.float32)
    x = rng.normal(size=(n, Y * X, cin)).astype(np.float32)
    got = np.asarray(nn.testEvaluateConv(w, n, X, Y, True, x))
    xt = torch.from_numpy(x.reshape(n, Y, X, cin).transpose(0, 3, 1, 2)).to(torch.bfloat16).float()
    wt = torch.from_numpy(w).to(torch.bfloat16).float()
    want = torch.nn.functional.conv2d(xt, wt, padding=1).numpy().transpose(0, 2, 3, 1).reshape(n, Y * X, cout)
    out["%%d_%%d_%%dx%%d_n%%d" %% (cin, cout, X, Y, n)] = [float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(want).max()),
                                                   hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()]
print("RESULT " + json.dumps(out))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)


def test_small_batch_shape_with_fetching_waves(emu_full_lib):
    """cfg 118 (conv_small_kernel.h: four multiplying waves of three cell tiles, four waves that issue every LDS-DMA request): against
    conv2d with immediate and with the latest legal completion of its requests (at the wait; at the barrier after the wait - the mode
    that fails round 4's deleted third depth, which the MI355X also did), at both fetch depths (SG<PACK, DEPTH>: slabs 3 / 6 steps
    ahead, the image 1 / 2 chunks) and with a board's cell tiles split over three work-groups (cfg 117, MTW = 1), and BIT-IDENTICAL to the 4-wave shapes of
    conv_kernel.h the same layers take without it (same MFMAs per output in the same K order) - square, rectangular and several
    boards, channel counts that are not multiples of the tile, one to five chunks. Round 5: the same shapes with the WEIGHTS IN REGISTERS
    (REGW, cfg 128 / 127 / 126: hand-written loads and waits, one barrier per chunk, three image buffers) under all three completion modes - the
    weight fragments are entered in the lane's in-order queue and land when a wait forces them."""
    # (loaders, fetch depth, cell tiles split over three work-groups (cfg 117), completion mode, weights in registers (cfg 128 / 127))
    variants = (("0", "0", "0", "0", "0"), ("1", "0", "0", "0", "0"), ("1", "0", "0", "2", "0"), ("1", "1", "0", "0", "0"), ("1", "1", "0", "1", "0"),
                ("1", "1", "0", "2", "0"), ("1", "1", "1", "0", "0"), ("1", "1", "1", "2", "0"),
                ("1", "1", "0", "0", "1"), ("1", "1", "0", "1", "1"), ("1", "1", "0", "2", "1"), ("1", "1", "1", "0", "3"), ("1", "1", "1", "1", "3"),
                ("1", "1", "1", "2", "3"),
                # (regw 2 with loaders_max_wgs=0: layers with an even number of channel tiles take the 64-channel register-weights shape, cfg 126)
                ("1", "1", "0", "0", "2,loaders_max_wgs=0"), ("1", "1", "0", "1", "2,loaders_max_wgs=0"), ("1", "1", "0", "2", "2,loaders_max_wgs=0"),
                # (regw_half=2: the cell tiles over TWO work-groups, cfg 125, also where the three-way split would be taken)
                ("1", "1", "1", "0", "3,regw_half=2"), ("1", "1", "1", "1", "3,regw_half=2"), ("1", "1", "1", "2", "3,regw_half=2"))
    runs = run_parallel([([sys.executable, "-c", CW12_CODE, emu_full_lib],
                          dict(os.environ, KMX_CONV_TUNE="loaders=%s,loaders_depth=%s,loaders_split=%s,regw=%s" % (ld, depth, split, regw), KMX_EMU_LATE_DMA=late))
                         for ld, depth, split, late, regw in variants])
    res = []
    for rc, so, se in runs:
        assert rc == 0 and "RESULT " in so, (so + se)[-3000:]
        res.append(json.loads(so.split("RESULT ")[1]))
    for r in res:
        for k, v in r.items():
            assert v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5, (k, v)
    for k in res[0]:
        assert len({r[k][2] for r in res}) == 1, ("not bit-identical across shapes", k, [r[k] for r in res])


def test_deep_ring_1x1_shapes(emu_full_lib):
    """cfg 113 / 114 / 124 (conv_mfma.hip, round 4): the 1x1 shapes of conv_kernel.h with a ring of four steps instead of two
    (and, 113, a board's cell tiles over three work-groups: conv_kernel.h ABL_SPLIT) -
    a 1x1 step is a whole image chunk, and at small batch every step of the two-step ring waited out a memory round trip. Layers with
    fewer steps than the ring is deep (96 channels: 3), with many (384: 12), several boards, a rectangular board; with immediate copies
    and both latest-completion modes; BIT-IDENTICAL to the two-step ring (same MFMAs per output in the same K order)."""
    shapes = [(1, 96, 64, 13, 13, 1), (1, 384, 96, 19, 19, 2), (1, 40, 192, 9, 7, 3)]
    # (conv_only adds min_wgs8=1,loaders=0; KMX_CONV_TUNE_MORE is appended to it)
    one = "split1x1=0,"  # a board's cell tiles in ONE work-group
    envs = [{"KMX_CONV_TUNE_MORE": one + "deep1x1=0"},
            {"KMX_CONV_TUNE_MORE": one + "deep1x1=1"}, {"KMX_CONV_TUNE_MORE": one + "deep1x1=1", "KMX_EMU_LATE_DMA": "2"},
            {"KMX_CONV_TUNE_MORE": one + "deep1x1=1", "KMX_EMU_LATE_DMA": "1"},
            {"KMX_CONV_TUNE_MORE": one + "deep1x1=1,deep1x1_max_wgs=0", "KMX_EMU_LATE_DMA": "2"},  # the 64-channel deep shape where tiles are even
            {"KMX_CONV_TUNE_MORE": "deep1x1=1"}, {"KMX_CONV_TUNE_MORE": "deep1x1=1", "KMX_EMU_LATE_DMA": "2"}]  # cfg 113: the cell tiles over three work-groups
    res = []
    for env, (rc, so, se) in zip(envs, run_parallel([conv_only(emu_full_lib, env, shapes) for env in envs])):
        assert rc == 0 and "RESULT " in so, (env, (so + se)[-3000:])
        res.append(json.loads(so.split("RESULT ")[1]))
        for k, v in res[-1].items():
            assert v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5, (env, k, v)
    for k in res[0]:
        assert len({r[k][2] for r in res}) == 1, ("not bit-identical across ring depths", k, [r[k] for r in res])


NET_REGW_CODE = r"""
import sys, json, hashlib, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn, modelgen
from conftest import make_rows
rng = np.random.default_rng(0)
nn.globalInitialize()
out = {}
# a 13x13 buffer with a smaller board inside, and a 2x3 buffer (6 cells: most of a wave's cell tiles lie off the board)
for X, Y, sizes in ((13, 13, [(13, 13), (9, 7)]), (2, 3, [(2, 3), (2, 2), (2, 3)])):
    L, n = max(X, Y), len(sizes)
    ctx = nn.createComputeContext([0], X, Y, precision="bf16")
    sp_full, gl = make_rows(rng, n, L, sizes)
    sp = np.ascontiguousarray(sp_full.reshape(n, L, L, 22)[:, :Y, :X, :]).reshape(n, X * Y, 22)
    sym = np.array([3, 6, 1][:n], np.int32); opt = np.array([0.0, 1.0, 0.5][:n], np.float32)
    h = nn.createComputeHandle(ctx, nn.loadModelFile(sys.argv[2]), 4)
    got = nn.getOutput(h, sp, gl, sym, opt)
    out.update({"%%dx%%d %%s" %% (X, Y, k): hashlib.sha1(np.ascontiguousarray(got[k]).tobytes()).hexdigest() for k in sorted(got)})
    h.close()
print("RESULT " + json.dumps(out))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))


def test_register_weights_shapes_whole_net(emu_full_lib, tmp_path):
    """A nested-bottleneck net with 64-channel inner 3x3 layers (residuals, per-board biases, raw and activated channel ranges, a board
    smaller than the buffer) through the slab-ring shapes and through the shapes with their weights in registers - cfg 128, its cell tiles
    over three work-groups (127) or two (125), and the 64-channel shape (126: two channel tiles per wave, half a chunk of fragments in the ring) -
    with immediate and with the latest legal completion: the same bits in every output. The 2x3 buffer is the case the MI355X found in
    the first version of cfg 126 (fuzz case 34): cell tiles off the board were skipped, and the residual requests, which run one tile ahead
    in a fixed order, then delivered the wrong tile's residual to the second channel tile."""
    from katago_amd import modelgen
    modelgen.ARCHS["b2c128nbt"] = dict(C=128, mid=64, gpool=16, blocks=["n", "ng"], p1=16, g1=16, v1=24, v2=32)
    model = str(tmp_path / "b2c128nbt.bin")
    modelgen.write_model(model, "b2c128nbt", seed=4)
    variants = (("regw=0", "0"), ("regw=3", "2"), ("regw=1,loaders_split=0", "1"), ("regw=2,loaders_max_wgs=0", "1"), ("regw=2,loaders_max_wgs=0", "2"),
                ("regw=3,regw_half=2", "1"), ("regw=3,regw_half=2", "2"))
    runs = run_parallel([([sys.executable, "-c", NET_REGW_CODE, emu_full_lib, model], dict(os.environ, KMX_CONV_TUNE=tune, KMX_EMU_LATE_DMA=late))
                         for tune, late in variants])
    res = []
    for rc, so, se in runs:
        assert rc == 0 and "RESULT " in so, (so + se)[-3000:]
        res.append(json.loads(so.split("RESULT ")[1]))
    for r in res[1:]:
        assert r == res[0], (variants, res)

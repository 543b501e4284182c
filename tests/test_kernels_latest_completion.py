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

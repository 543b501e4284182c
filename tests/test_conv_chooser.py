"""Host logic of the convolution launcher, no GPU: for every (kernel size, padded channel count, batch) the engine can
ask for, chooseConvCfg must name a work-group shape for which a kernel is instantiated and which tiles the channels —
otherwise the launch fails at run time with hipErrorInvalidValue (found this way: 5x5 stems of 192-channel v8 nets at
batch >= 150 used to pick an 8-wave x 192 shape that only exists for 1x1 and 3x3)."""
import ctypes

import pytest

from katago_amd import capi


def test_every_choice_is_launchable():
    lib = capi.load_library()
    cfg, ok = ctypes.c_int(), ctypes.c_int()
    seen = {}
    for ks in (1, 3, 5):
        for cout_pad in range(64, 1536 + 1, 64):
            for batch in list(range(1, 66)) + [96, 112, 128, 129, 140, 141, 149, 150, 151, 192, 224, 255, 256, 300, 420, 421, 512, 1024, 2048]:
                capi.check(lib.kmx_debug_conv_cfg(ks, cout_pad, batch, ctypes.byref(cfg), ctypes.byref(ok)), lib)
                assert ok.value == 1, (ks, cout_pad, batch, cfg.value)
                seen.setdefault(ks, set()).add(cfg.value)
    # (3x3 at small batch: the fetching-waves shapes with their weights in registers, cfg 126 / 127 / 128, since round 5; their slab-ring
    # twins 119 / 117 / 118 stay instantiated behind KMX_CONV_TUNE regw=0)
    # (3x3: the 4-wave x 96 shape, cfg 13, is no longer chosen since the 8-wave shapes start at 129 work-groups - round 6 -: it was the shape of
    # batch 141-149, more than one work-group per CU; it stays instantiated for 5x5 and for the tools that force a shape)
    assert seen[3] >= {125, 126, 127, 128, 12, 22, 23} and 13 not in seen[3] and seen[1] >= {113, 114, 124, 12, 22, 23} and seen[5] >= {11, 13, 22}
    assert not ({117, 118, 119} & seen[3])
    assert not ({117, 118, 119, 125, 126, 127, 128} & (seen[1] | seen[5]))  # the small-batch shapes with fetching waves exist for 3x3 only
    assert not ({113, 114, 124} & (seen[3] | seen[5]))  # the deep-ring shapes for 1x1 only
    assert 23 not in seen[5] and 12 not in seen[5] and 13 not in seen[1]  # (5x5, 12) spilled registers: removed in round 3


def test_baseline_shapes_unchanged():
    """The b18c384nbt layers at batch 256 (128 per stream x 2 streams) keep the shapes the measurements were taken with."""
    lib = capi.load_library()
    cfg, ok = ctypes.c_int(), ctypes.c_int()
    for ks, cout_pad, want in ((3, 192, 23), (1, 384, 23), (1, 192, 23), (3, 384, 23), (1, 256, 22)):
        capi.check(lib.kmx_debug_conv_cfg(ks, cout_pad, 256, ctypes.byref(cfg), ctypes.byref(ok)), lib)
        assert (cfg.value, ok.value) == (want, 1), (ks, cout_pad, cfg.value)
    for bad in ((2, 64, 1), (3, 100, 1), (3, 64, 0)):
        assert lib.kmx_debug_conv_cfg(bad[0], bad[1], bad[2], ctypes.byref(cfg), ctypes.byref(ok)) == capi.KMX_ERR_INVALID_ARG


def test_small_batch_3x3_ranges_of_b18():
    """192-channel 3x3 layers (b18c384nbt's trunk) by batch: the cell tiles over three work-groups (cfg 127) while batch x 6 x 3 <= 256, over
    two (125) while batch x 6 x 2 <= 256, one work-group per board x 32 channels (128) while batch x 6 <= 256, a board x 64 channels (126)
    while batch x 3 <= 256, then the 4-wave shapes of conv_kernel.h and from 129 work-groups (round 6; 150 before) the 8-wave x 192 one. The net's 64-channel
    layers keep the split shapes longer. (cfg 125 is the shape under which round 5's driver run faulted; the cause was in the kernel family,
    not in the choice: DESIGN.md 0e.)"""
    lib = capi.load_library()
    cfg, ok = ctypes.c_int(), ctypes.c_int()

    def choice(cout_pad, batch):
        capi.check(lib.kmx_debug_conv_cfg(3, cout_pad, batch, ctypes.byref(cfg), ctypes.byref(ok)), lib)
        assert ok.value == 1
        return cfg.value

    want = {1: 127, 14: 127, 15: 125, 21: 125, 22: 128, 42: 128, 43: 126, 85: 126, 86: 12, 128: 12, 129: 23, 150: 23, 256: 23}
    assert {b: choice(192, b) for b in want} == want
    assert [choice(64, b) for b in (1, 42, 43, 64, 65, 128, 129, 256)] == [127, 127, 125, 125, 128, 128, 126, 126]


def test_a_typo_in_the_debug_override_is_an_error_not_an_abort():
    """KMX_CONV_TUNE with an unknown key: reported through the normal error path (KMX_ERR_INVALID_ARG from engine construction and from
    kmx_debug_conv_cfg, hipErrorInvalidValue from launchConv) - until round 5 the shared library called abort() in the embedder's process
    (ADVICE round 5). A leftover environment variable of the nine that KMX_CONV_TUNE replaced is named on stderr."""
    import os
    import subprocess
    import sys

    code = ("import ctypes, sys; sys.path.insert(0, %r); from katago_amd import capi; lib = capi.load_library(); c, k = ctypes.c_int(), ctypes.c_int();"
            "rc = lib.kmx_debug_conv_cfg(3, 192, 8, ctypes.byref(c), ctypes.byref(k)); print('RC', rc, lib.kmx_last_error().decode())"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, KMX_CONV_TUNE="regw_haf=1", KMX_MIN_WGS8="100"))
    assert p.returncode == 0 and ("RC %d" % capi.KMX_ERR_INVALID_ARG) in p.stdout and "unknown item 'regw_haf=1'" in p.stdout, p.stdout + p.stderr
    assert "KMX_MIN_WGS8 is no longer read" in p.stderr, p.stderr
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, KMX_CONV_TUNE="regw_half=0"))
    assert p.returncode == 0 and "RC 0" in p.stdout, p.stdout + p.stderr

"""The transformer device kernels (katago_amd/csrc/transformer_kernels.hip), compiled for x86 and EXECUTED on the CPU by
tests/fakehip/emulate_transformer.cpp (work-group = OS threads, __syncthreads = barrier, __shfl_xor = exchange array,
dynamic LDS = static buffer), against numpy restatements of the reference formulas — the same restatements the GPU unit
tests (tests/test_gpu_transformer.py) use. This checks the kernels' logic (indexing, masking, RoPE, running-max softmax,
grouped-query heads, zeroed channel padding, 16-bit layouts) without hardware; hardware-specific behaviour is what the
GPU tests remain for."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO
from katago_amd import capi
from test_gpu_transformer import _masks, _q16, _rope, _rope_tables

FP = ctypes.POINTER(ctypes.c_float)


def _p(a):
    return None if a is None else a.ctypes.data_as(FP)


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = os.path.join(str(tmp_path_factory.mktemp("emutf")), "libemutf.so")
    cmd = ["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++20", "-O2", "-fPIC", "-shared", "-pthread",
           "-I" + os.path.join(REPO, "tests", "fakehip", "emul"), "-o", so, os.path.join(REPO, "tests", "fakehip", "emulate_transformer.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return ctypes.CDLL(so)


def _tol(dtype):
    return 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("C,per_board,with_beta,act", [(32, False, False, capi.ACT_IDENTITY), (96, False, False, capi.ACT_IDENTITY),
                                                         (32, True, True, capi.ACT_MISH), (104, False, True, capi.ACT_SILU),
                                                         (96, True, True, capi.ACT_RELU)])
def test_rmsnorm_kernel_emulated(emu, dtype, C, per_board, with_beta, act):
    rng = np.random.default_rng(C + per_board)
    n, X, Y = 3, 19, 19
    S = X * Y
    mask = _masks(rng, n, X, Y)
    x = _q16(rng.normal(0, 2.0, (n, S, C)), dtype)
    w = (1.0 + 0.3 * rng.normal(size=C)).astype(np.float32)
    beta = (0.2 * rng.normal(size=C)).astype(np.float32) if with_beta else None
    eps = 1e-6
    if per_board:
        ss = (x.astype(np.float64) ** 2 * mask[:, :, None]).sum(axis=(1, 2)) / (mask.sum(axis=1) * C)
        r = 1.0 / np.sqrt(ss + eps)[:, None, None]
    else:
        r = 1.0 / np.sqrt((x.astype(np.float64) ** 2).mean(axis=2, keepdims=True) + eps)
    y = x * r * w + (beta if beta is not None else 0.0)
    if act == capi.ACT_RELU:
        y = np.maximum(y, 0)
    elif act == capi.ACT_SILU:
        y = y / (1 + np.exp(-y))
    elif act == capi.ACT_MISH:
        y = y * np.tanh(np.log1p(np.exp(np.minimum(y, 20.0))))
    want = (y * mask[:, :, None]).astype(np.float32)
    got = np.full((n, S, C), np.nan, np.float32)
    rc = emu.emu_rmsnorm(1 if dtype == "bf16" else 0, n, S, C, ctypes.c_float(eps), _p(w), _p(beta), act, int(per_board), _p(x), _p(mask), _p(got))
    assert rc == 0
    assert np.isfinite(got).all() and np.all(np.abs(got - want) <= 1.5 * _tol(dtype) * np.maximum(1.0, np.abs(want)))
    assert np.all(got[mask == 0] == 0)


@pytest.mark.parametrize("kernel", ["mfma", "valu"])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("H,KVH,QD,VD,rope", [(4, 4, 8, 8, "fixed"), (4, 2, 8, 4, "learnable"), (3, 3, 32, 32, "fixed"),
                                               (6, 3, 32, 16, "learnable"), (2, 1, 64, 64, "none"), (2, 2, 16, 32, "none"), (1, 1, 7, 3, "none")])
def test_attention_kernel_emulated(emu, monkeypatch, kernel, dtype, H, KVH, QD, VD, rope):
    """Both attention kernels: the matrix-core one (default; v_mfma_f32_32x32x16 is emulated as a wave-collective with the
    hardware's register layout) and the plain one (KMX_ATTENTION_VALU=1)."""
    if kernel == "valu":
        monkeypatch.setenv("KMX_ATTENTION_VALU", "1")
    else:
        monkeypatch.delenv("KMX_ATTENTION_VALU", raising=False)
    if kernel == "mfma" and dtype == "fp16" and (H, QD) in ((3, 32), (2, 64)):
        pytest.skip("emulating MFMAs is slow: the large shapes run once, in bf16")
    rng = np.random.default_rng(H * 100 + QD)
    n, X, Y = 2, 19, 19
    S = X * Y
    mask = _masks(rng, n, X, Y)[::-1].copy()  # board 0 partial, board 1 full
    q = _q16(rng.normal(0, 1.0, (n, S, H, QD)), dtype)
    k = _q16(rng.normal(0, 1.0, (n, S, KVH, QD)), dtype)
    v = _q16(rng.normal(0, 1.0, (n, S, KVH, VD)), dtype)
    cos = sin = None
    qr, kr = q, k
    if rope != "none":
        th = KVH if rope == "learnable" else 1
        cos, sin = _rope_tables(rng, th, QD, X, Y, rope == "learnable")
        qr = _rope(q, cos, sin, [(h * KVH // H) if th > 1 else 0 for h in range(H)])
        kr = _q16(_rope(k, cos, sin, [h if th > 1 else 0 for h in range(KVH)]), dtype)  # the kernel keeps rotated K in 16 bits
    want = np.zeros((n, S, H, VD), np.float64)
    for b in range(n):
        on = mask[b] > 0
        for h in range(H):
            g = h // (H // KVH)
            logits = (qr[b, :, h].astype(np.float64) @ kr[b, on, g].astype(np.float64).T) / np.sqrt(QD)
            p = np.exp(logits - logits.max(axis=1, keepdims=True))
            p /= p.sum(axis=1, keepdims=True)
            want[b, :, h] = (p @ v[b, on, g].astype(np.float64)) * mask[b][:, None]
    got = np.full((n, S, H * VD), np.nan, np.float32)
    rc = emu.emu_attention(1 if dtype == "bf16" else 0, n, S, H, KVH, QD, VD, _p(cos), _p(sin), 1 if rope != "learnable" else KVH,
                           _p(np.ascontiguousarray(q.reshape(n, S, -1))), _p(np.ascontiguousarray(k.reshape(n, S, -1))),
                           _p(np.ascontiguousarray(v.reshape(n, S, -1))), _p(mask), _p(got))
    assert rc == 0
    err = np.abs(got.reshape(n, S, H, VD) - want)
    print("%s %s max err %.4g" % (kernel, dtype, err.max()))
    assert np.isfinite(got).all() and err.max() <= 3 * _tol(dtype) * max(1.0, np.abs(want).max()), err.max()
    assert np.all(got[mask == 0] == 0)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("F", [48, 256])
def test_swiglu_kernel_emulated(emu, dtype, F):
    rng = np.random.default_rng(F)
    n, S = 2, 13 * 9
    a = _q16(rng.normal(0, 2.0, (n, S, F)), dtype)
    g = _q16(rng.normal(0, 2.0, (n, S, F)), dtype)
    want = a / (1 + np.exp(-a.astype(np.float64))) * g
    got = np.full((n, S, F), np.nan, np.float32)
    assert emu.emu_swiglu(1 if dtype == "bf16" else 0, n, S, F, _p(a), _p(g), _p(got)) == 0
    assert np.all(np.abs(got - want) <= 1.5 * _tol(dtype) * np.maximum(1.0, np.abs(want)))


def test_unsupported_shapes_are_refused(emu):
    z = np.zeros((1, 4, 80), np.float32)
    m = np.ones((1, 4), np.float32)
    out = np.zeros((1, 4, 80), np.float32)
    assert emu.emu_attention(1, 1, 4, 1, 1, 80, 80, None, None, 1, _p(z), _p(z), _p(z), _p(m), _p(out)) != 0  # head dim > 64
    w = np.ones(20, np.float32)
    x = np.zeros((1, 4, 20), np.float32)
    assert emu.emu_rmsnorm(1, 1, 4, 20, ctypes.c_float(1e-6), _p(w), None, 0, 0, _p(x), _p(m), _p(x.copy())) != 0  # C % 8 != 0

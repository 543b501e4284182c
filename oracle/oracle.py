"""ctypes binding of the CPU ORACLE (oracle/libkmxoracle.so).  TEST INFRASTRUCTURE ONLY.

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from
katago_amd/. Mirrors katago_amd.nninterface so that a test can run the same call on both sides.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libkmxoracle.so")
sys.path.insert(0, _REPO) if _REPO not in sys.path else None

from katago_amd import capi  # noqa: E402  (struct definitions only: shared with include/katamx.h)

_FP = ctypes.POINTER(ctypes.c_float)
_FPP = ctypes.POINTER(_FP)
_IP = ctypes.POINTER(ctypes.c_int)
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "oracle"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        L.okmx_last_error.restype = ctypes.c_char_p
        L.okmx_model_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        L.okmx_model_free.argtypes = [ctypes.c_void_p]
        L.okmx_model_info_get.argtypes = [ctypes.c_void_p, ctypes.POINTER(capi.ModelInfo)]
        L.okmx_eval.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FPP, _FPP, _IP, _FP, _FPP, _FP, _FP, _FPP,
                                ctypes.c_int]
        L.okmx_eval_meta.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FPP, _FPP, _FPP, _IP, _FP, _FPP, _FP, _FP,
                                     _FPP, ctypes.c_int]
        L.okmx_eval_trunk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, ctypes.c_int, _FP]
        L.okmx_test_conv.argtypes = [ctypes.POINTER(capi.ConvDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP]
        L.okmx_test_bnact.argtypes = [ctypes.POINTER(capi.BnActDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP]
        L.okmx_test_resblock.argtypes = [ctypes.POINTER(capi.ResBlockDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP]
        L.okmx_test_gpoolblock.argtypes = [ctypes.POINTER(capi.GPoolBlockDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP]
        L.okmx_copy_with_symmetry.argtypes = [_FP, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.okmx_copy_with_symmetry.restype = None
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _check(status):
    if status != 0:
        raise OracleError("oracle error %d: %s" % (status, (lib().okmx_last_error() or b"").decode("utf-8", "replace")))


def _fp(a):
    return a.ctypes.data_as(_FP)


class OracleModel:
    def __init__(self, file):
        p = ctypes.c_void_p()
        _check(lib().okmx_model_load(file.encode(), b"", ctypes.byref(p)))
        self._p = p
        self.info = capi.ModelInfo()
        _check(lib().okmx_model_info_get(self._p, ctypes.byref(self.info)))

    def close(self):
        if self._p:
            lib().okmx_model_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def loadModelFile(file):
    return OracleModel(file)


def usable_cores(cap=32):
    """Threads for the OpenMP loops: the cgroup's CPUs, at most `cap`. (The GPU box shows 256 logical CPUs; the
    oracle's batch x channel-tile loops do not scale there — 0.26 evals/s of b18c384nbt with 256 threads.)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, cap))


def getOutput(model, nnXLen, nnYLen, rowSpatial, rowGlobal, symmetries=None, policyOptimisms=None, includeOwnerMap=True,
              numThreads=0, rowMeta=None):
    if numThreads <= 0:
        numThreads = usable_cores()  # never oversubscribe a cgroup-limited box with one thread per host core
    rowSpatial = np.ascontiguousarray(rowSpatial, dtype=np.float32)
    rowGlobal = np.ascontiguousarray(rowGlobal, dtype=np.float32)
    n = rowSpatial.shape[0]
    S = nnXLen * nnYLen
    sp2 = rowSpatial.reshape(n, -1)
    gl2 = rowGlobal.reshape(n, -1)
    assert sp2.shape[1] == S * 22 and gl2.shape[1] == 19
    sym = np.ascontiguousarray(symmetries if symmetries is not None else np.zeros(n), dtype=np.int32)
    opt = np.ascontiguousarray(policyOptimisms if policyOptimisms is not None else np.zeros(n), dtype=np.float32)
    policy = np.empty((n, S + 1), dtype=np.float32)
    value = np.empty((n, 3), dtype=np.float32)
    score = np.empty((n, 6), dtype=np.float32)
    ownership = np.empty((n, S), dtype=np.float32) if includeOwnerMap else None
    PT = _FP * n
    sp_ptrs = PT(*[_fp(sp2[i]) for i in range(n)])
    gl_ptrs = PT(*[_fp(gl2[i]) for i in range(n)])
    pol_ptrs = PT(*[_fp(policy[i]) for i in range(n)])
    own_ptrs = PT(*[_fp(ownership[i]) for i in range(n)]) if includeOwnerMap else None
    mt_ptrs = None
    if rowMeta is not None:
        mt2 = np.ascontiguousarray(rowMeta, dtype=np.float32).reshape(n, -1)
        mt_ptrs = PT(*[_fp(mt2[i]) for i in range(n)])
    _check(lib().okmx_eval_meta(model._p, nnXLen, nnYLen, n, sp_ptrs, gl_ptrs, mt_ptrs, sym.ctypes.data_as(_IP), _fp(opt), pol_ptrs,
                                _fp(value), _fp(score), own_ptrs, numThreads))
    return {"policy": policy, "value": value, "score": score, "ownership": ownership}


def evalTrunk(model, nnXLen, nnYLen, spatial, glob, which=0):
    spatial = np.ascontiguousarray(spatial, dtype=np.float32)
    glob = np.ascontiguousarray(glob, dtype=np.float32)
    n = spatial.shape[0]
    out = np.empty((n, nnYLen * nnXLen, model.info.trunk_num_channels), dtype=np.float32)
    _check(lib().okmx_eval_trunk(model._p, nnXLen, nnYLen, n, _fp(spatial), _fp(glob), which, _fp(out)))
    return out


def copyWithSymmetry(src_hwc, symmetry, reverse):
    src = np.ascontiguousarray(src_hwc, dtype=np.float32)
    h, w, c = src.shape
    dst = np.empty_like(src)
    lib().okmx_copy_with_symmetry(_fp(src), _fp(dst), h, w, c, symmetry, 1 if reverse else 0)
    return dst


# ---- layer hooks, same signatures as katago_amd.nninterface.testEvaluate* --------------------------
def _conv_desc(w_oihw):
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    oc, ic, ky, kx = w.shape
    return capi.ConvDesc(ky, kx, ic, oc, _fp(w)), w


def _bn_desc(scale, bias, activation):
    s = np.ascontiguousarray(scale, dtype=np.float32)
    b = np.ascontiguousarray(bias, dtype=np.float32)
    return capi.BnActDesc(len(s), activation, _fp(s), _fp(b)), (s, b)


def testEvaluateConv(w_oihw, batchSize, nnXLen, nnYLen, inputNHWC):
    d, keep = _conv_desc(w_oihw)
    x = np.ascontiguousarray(inputNHWC, dtype=np.float32)
    out = np.empty((batchSize, nnYLen, nnXLen, d.out_channels), dtype=np.float32)
    _check(lib().okmx_test_conv(ctypes.byref(d), batchSize, nnXLen, nnYLen, _fp(x), _fp(out)))
    return out


def testEvaluateBatchNorm(scale, bias, activation, batchSize, nnXLen, nnYLen, inputNHWC, maskNHW):
    d, keep = _bn_desc(scale, bias, activation)
    x = np.ascontiguousarray(inputNHWC, dtype=np.float32)
    m = np.ascontiguousarray(maskNHW, dtype=np.float32)
    out = np.empty_like(x)
    _check(lib().okmx_test_bnact(ctypes.byref(d), batchSize, nnXLen, nnYLen, _fp(x), _fp(m), _fp(out)))
    return out


def testEvaluateResidualBlock(block, batchSize, nnXLen, nnYLen, inputNHWC, maskNHW):
    pre, k1 = _bn_desc(*block["pre"])
    c1, k2 = _conv_desc(block["conv1"])
    mid, k3 = _bn_desc(*block["mid"])
    c2, k4 = _conv_desc(block["conv2"])
    d = capi.ResBlockDesc(pre, c1, mid, c2)
    x = np.ascontiguousarray(inputNHWC, dtype=np.float32)
    m = np.ascontiguousarray(maskNHW, dtype=np.float32)
    out = np.empty_like(x)
    _check(lib().okmx_test_resblock(ctypes.byref(d), batchSize, nnXLen, nnYLen, _fp(x), _fp(m), _fp(out)))
    return out


def testEvaluateGlobalPoolingResidualBlock(block, batchSize, nnXLen, nnYLen, inputNHWC, maskNHW):
    pre, k1 = _bn_desc(*block["pre"])
    cr, k2 = _conv_desc(block["convr"])
    cg, k3 = _conv_desc(block["convg"])
    gbn, k4 = _bn_desc(*block["gbn"])
    gw = np.ascontiguousarray(block["gmul"], dtype=np.float32)
    gm = capi.MatMulDesc(gw.shape[0], gw.shape[1], _fp(gw))
    mid, k5 = _bn_desc(*block["mid"])
    c2, k6 = _conv_desc(block["conv2"])
    d = capi.GPoolBlockDesc(pre, cr, cg, gbn, gm, mid, c2)
    x = np.ascontiguousarray(inputNHWC, dtype=np.float32)
    m = np.ascontiguousarray(maskNHW, dtype=np.float32)
    out = np.empty_like(x)
    _check(lib().okmx_test_gpoolblock(ctypes.byref(d), batchSize, nnXLen, nnYLen, _fp(x), _fp(m), _fp(out)))
    return out

"""Python mirror of the reference's backend interface `namespace NeuralNet`
(cpp/neuralnet/nninterface.h:32-182) over the katamx C ABI.

Same names, argument meaning and error behaviour as the reference (errors raise KatamxError where the
reference throws StringError), so that the parity tests read like the reference's own tests:

    model   = loadModelFile(path, expectedSha256)
    ctx     = createComputeContext([0], nnXLen, nnYLen, useFP16Mode="auto")
    handle  = createComputeHandle(ctx, model, maxBatchSize, requireExactNNLen, gpuIdxForThisThread)
    outputs = getOutput(handle, rowSpatial, rowGlobal, symmetries, policyOptimisms)

This file is host plumbing for tests and bench.py; the product is the C ABI underneath it.
"""
import ctypes
import threading

import numpy as np

from . import capi
from .capi import KatamxError  # noqa: F401  (re-export)

_FP = ctypes.POINTER(ctypes.c_float)


def _fp(a):
    return a.ctypes.data_as(_FP)


def globalInitialize():
    lib = capi.load_library()
    capi.check(lib.kmx_global_init(), lib)


def globalCleanup():
    capi.load_library().kmx_global_cleanup()


def printDevices():
    lib = capi.load_library()
    n = lib.kmx_device_count()
    out = []
    for i in range(max(n, 0)):
        buf = ctypes.create_string_buffer(256)
        if lib.kmx_device_name(i, buf, 256) == capi.KMX_OK:
            out.append("Found HIP device %d: %s" % (i, buf.value.decode()))
    print("\n".join(out))
    return out


class LoadedModel:
    def __init__(self, file, expectedSha256=""):
        self._lib = capi.load_library()
        p = ctypes.c_void_p()
        capi.check(self._lib.kmx_model_load(file.encode(), (expectedSha256 or "").encode(), ctypes.byref(p)), self._lib)
        self._p = p
        info = capi.ModelInfo()
        capi.check(self._lib.kmx_model_info_get(self._p, ctypes.byref(info)), self._lib)
        self.info = info
        self.file = file

    def close(self):
        if self._p:
            self._lib.kmx_model_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def loadModelFile(file, expectedSha256=""):
    return LoadedModel(file, expectedSha256)


def freeLoadedModel(model):
    model.close()


def getModelDesc(model):
    """The fields NNEvaluator reads from ModelDesc (nneval.cpp:138-143,292,306,327)."""
    i = model.info
    return {
        "name": i.name.decode(),
        "modelVersion": i.model_version,
        "numInputChannels": i.num_input_channels,
        "numInputGlobalChannels": i.num_input_global_channels,
        "numPolicyChannels": i.num_policy_channels,
        "numValueChannels": i.num_value_channels,
        "numScoreValueChannels": i.num_score_value_channels,
        "numOwnershipChannels": i.num_ownership_channels,
        "trunkNumChannels": i.trunk_num_channels,
        "numBlocks": i.num_blocks,
        "numParameters": i.num_parameters,
        "flopsPerPosition": i.flops_per_position,
        "postProcessParams": {
            "tdScoreMultiplier": i.td_score_multiplier,
            "scoreMeanMultiplier": i.score_mean_multiplier,
            "scoreStdevMultiplier": i.score_stdev_multiplier,
            "leadMultiplier": i.lead_multiplier,
            "varianceTimeMultiplier": i.variance_time_multiplier,
            "shorttermValueErrorMultiplier": i.shortterm_value_error_multiplier,
            "shorttermScoreErrorMultiplier": i.shortterm_score_error_multiplier,
            "outputScaleMultiplier": i.output_scale_multiplier,
        },
    }


class ComputeContext:
    def __init__(self, gpuIdxs, nnXLen, nnYLen, useFP16Mode="auto", precision=None):
        self._lib = capi.load_library()
        if precision is None:
            # tri-state enabled_t (cpp/core/commontypes.h): False -> fp32, True/Auto -> backend 16-bit default
            precision = capi.PREC_FP32 if useFP16Mode in (False, "false") else capi.PREC_AUTO
        elif isinstance(precision, str):
            precision = {"auto": capi.PREC_AUTO, "fp32": capi.PREC_FP32, "fp16": capi.PREC_FP16, "bf16": capi.PREC_BF16}[precision]
        idxs = (ctypes.c_int * max(len(gpuIdxs), 1))(*gpuIdxs)
        p = ctypes.c_void_p()
        capi.check(self._lib.kmx_context_create(idxs, len(gpuIdxs), nnXLen, nnYLen, precision, ctypes.byref(p)), self._lib)
        self._p = p
        self.nnXLen, self.nnYLen = nnXLen, nnYLen

    def close(self):
        if self._p:
            self._lib.kmx_context_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def createComputeContext(gpuIdxs, nnXLen, nnYLen, useFP16Mode="auto", precision=None):
    return ComputeContext(gpuIdxs, nnXLen, nnYLen, useFP16Mode, precision)


def freeComputeContext(ctx):
    ctx.close()


class ComputeHandle:
    def __init__(self, context, loadedModel, maxBatchSize, requireExactNNLen=False, gpuIdxForThisThread=-1):
        self._lib = capi.load_library()
        p = ctypes.c_void_p()
        capi.check(self._lib.kmx_handle_create(context._p, loadedModel._p, maxBatchSize, 1 if requireExactNNLen else 0,
                                                gpuIdxForThisThread, ctypes.byref(p)), self._lib)
        self._p = p
        self.context = context
        self.model = loadedModel
        self.maxBatchSize = maxBatchSize

    @property
    def precision(self):
        return capi.PREC_NAMES[self._lib.kmx_handle_precision(self._p)]

    def stats(self):
        r, b = ctypes.c_uint64(), ctypes.c_uint64()
        capi.check(self._lib.kmx_handle_stats(self._p, ctypes.byref(r), ctypes.byref(b)), self._lib)
        return r.value, b.value

    def sync(self):
        capi.check(self._lib.kmx_handle_sync(self._p), self._lib)

    def close(self):
        if self._p:
            self._lib.kmx_handle_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def createComputeHandle(context, loadedModel, maxBatchSize, requireExactNNLen=False, gpuIdxForThisThread=-1):
    return ComputeHandle(context, loadedModel, maxBatchSize, requireExactNNLen, gpuIdxForThisThread)


def freeComputeHandle(handle):
    handle.close()


def isUsingFP16(handle):
    return handle.precision != "fp32"


def getOutput(handle, rowSpatial, rowGlobal, symmetries=None, policyOptimisms=None, includeOwnerMap=True, rowMeta=None):
    """NeuralNet::getOutput. rowSpatial: float32 [n, nnY*nnX, 22] NHWC (unsymmetrised), rowGlobal: [n, 19], rowMeta: [n, 192]
    for nets with an sgf-metadata encoder (NNResultBuf::rowMetaBuf), else None.
    Returns dict of logits: policy [n, S+1] (last = pass), value [n,3], score [n,6], ownership [n,S] or None."""
    lib = handle._lib
    rowSpatial = np.ascontiguousarray(rowSpatial, dtype=np.float32)
    rowGlobal = np.ascontiguousarray(rowGlobal, dtype=np.float32)
    n = rowSpatial.shape[0]
    S = handle.context.nnXLen * handle.context.nnYLen
    assert rowSpatial.reshape(n, -1).shape[1] == S * handle.model.info.num_input_channels
    assert rowGlobal.reshape(n, -1).shape[1] == handle.model.info.num_input_global_channels
    sp2 = rowSpatial.reshape(n, -1)
    gl2 = rowGlobal.reshape(n, -1)
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
        assert handle.model.info.num_input_meta_channels in (0, mt2.shape[1])  # 0: the library rejects the call
        mt_ptrs = PT(*[_fp(mt2[i]) for i in range(n)])
    capi.check(lib.kmx_eval_meta(handle._p, n, sp_ptrs, gl_ptrs, mt_ptrs, sym.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(opt),
                                 pol_ptrs, _fp(value), _fp(score), own_ptrs), lib)
    return {"policy": policy, "value": value, "score": score, "ownership": ownership}


def packRows(rowSpatial, nnXLen, nnYLen):
    """fp32 NHWC rows [n, nnY*nnX, C] of 0/1 features -> uint8 [n, C * ceil(S/8)] in the reference's binaryInputNCHWPacked
    layout (dataio/trainingwrite.h:180-183): plane by plane, 8 cells per byte, most significant bit first."""
    x = np.ascontiguousarray(rowSpatial, dtype=np.float32)
    n = x.shape[0]
    S = nnXLen * nnYLen
    planes = np.ascontiguousarray((x.reshape(n, S, -1) != 0).transpose(0, 2, 1))  # [n, C, S]
    return np.ascontiguousarray(np.packbits(planes, axis=2, bitorder="big")).reshape(n, -1)


def getOutputPacked(handle, rowPacked, rowGlobal, symmetries=None, policyOptimisms=None, includeOwnerMap=True, rowMeta=None):
    """kmx_eval_packed: getOutput on bit-packed spatial rows (packRows / kmx_pack_row)."""
    lib = handle._lib
    rowPacked = np.ascontiguousarray(rowPacked, dtype=np.uint8)
    rowGlobal = np.ascontiguousarray(rowGlobal, dtype=np.float32)
    n = rowPacked.shape[0]
    S = handle.context.nnXLen * handle.context.nnYLen
    assert rowPacked.reshape(n, -1).shape[1] == handle.model.info.num_input_channels * ((S + 7) // 8)
    pk2, gl2 = rowPacked.reshape(n, -1), rowGlobal.reshape(n, -1)
    sym = np.ascontiguousarray(symmetries if symmetries is not None else np.zeros(n), dtype=np.int32)
    opt = np.ascontiguousarray(policyOptimisms if policyOptimisms is not None else np.zeros(n), dtype=np.float32)
    policy = np.empty((n, S + 1), dtype=np.float32)
    value = np.empty((n, 3), dtype=np.float32)
    score = np.empty((n, 6), dtype=np.float32)
    ownership = np.empty((n, S), dtype=np.float32) if includeOwnerMap else None
    U8P = ctypes.POINTER(ctypes.c_uint8)
    pk_ptrs = (U8P * n)(*[pk2[i].ctypes.data_as(U8P) for i in range(n)])
    PT = _FP * n
    gl_ptrs = PT(*[_fp(gl2[i]) for i in range(n)])
    pol_ptrs = PT(*[_fp(policy[i]) for i in range(n)])
    own_ptrs = PT(*[_fp(ownership[i]) for i in range(n)]) if includeOwnerMap else None
    mt_ptrs = None
    if rowMeta is not None:
        mt2 = np.ascontiguousarray(rowMeta, dtype=np.float32).reshape(n, -1)
        mt_ptrs = PT(*[_fp(mt2[i]) for i in range(n)])
    capi.check(lib.kmx_eval_packed(handle._p, n, pk_ptrs, gl_ptrs, mt_ptrs, sym.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(opt),
                                   pol_ptrs, _fp(value), _fp(score), own_ptrs), lib)
    return {"policy": policy, "value": value, "score": score, "ownership": ownership}


def getOutputDevice(handle, dSpatial, dGlobal, symmetries, policyOptimisms, dPolicy, dValue, dScore, dOwnership=None, sync=True):
    """kmx_eval_device: the same pass on device-resident buffers (extension; what a device-side batcher would call).
    d* are device addresses (ints) of float32 arrays laid out like getOutput's host arrays; symmetries / optimisms are
    host arrays. With sync=False the call returns after enqueueing; handle.sync() completes it."""
    n = len(symmetries)
    sym = np.ascontiguousarray(symmetries, dtype=np.int32)
    opt = np.ascontiguousarray(policyOptimisms if policyOptimisms is not None else np.zeros(n), dtype=np.float32)
    capi.check(handle._lib.kmx_eval_device(handle._p, n, dSpatial, dGlobal, sym.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(opt),
                                           dPolicy, dValue, dScore, dOwnership, 1 if sync else 0), handle._lib)


class Batcher:
    """kmx_batcher_*: the persistent leaf batcher (include/katamx.h). The role of NNEvaluator's server half (nneval.cpp:562-752)
    for callers that submit rows themselves: submit() from any thread returns a ticket, wait(ticket) returns that row's outputs.
    Rows are bit-packed into pinned staging by their submitters, batches are sealed greedily and up to max_in_flight of them
    are on the device at once. A thread may hold any number of tickets."""

    def __init__(self, context, model, maxBatchSize, maxInFlight=2, gpuIdx=0):
        self._lib = capi.load_library()
        self.context, self.model = context, model
        self._p = ctypes.c_void_p()
        capi.check(self._lib.kmx_batcher_create(context._p, model._p, maxBatchSize, maxInFlight, gpuIdx, ctypes.byref(self._p)), self._lib)
        self.S = context.nnXLen * context.nnYLen
        self.numInputChannels = model.info.num_input_channels
        self._out = {}
        self._lock = threading.Lock()

    def close(self):
        if self._p:
            self._lib.kmx_batcher_free(self._p)
            self._p = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, rowSpatial, rowGlobal, symmetry=0, policyOptimism=0.0, wantOwnership=True, rowMeta=None, packed=False):
        """Returns a ticket. The row's output arrays are allocated here (the library writes into them) and handed out by wait().
        packed=True: rowSpatial is one row of packRows() (uint8 bit planes) and goes through kmx_batcher_submit_packed."""
        if packed:
            sp = np.ascontiguousarray(rowSpatial, dtype=np.uint8).reshape(-1)
            assert sp.size == self.numInputChannels * ((self.S + 7) // 8)
        else:
            sp = np.ascontiguousarray(rowSpatial, dtype=np.float32)
        gl = np.ascontiguousarray(rowGlobal, dtype=np.float32)
        mt = None if rowMeta is None else np.ascontiguousarray(rowMeta, dtype=np.float32)
        out = {"policy": np.empty(self.S + 1, np.float32), "value": np.empty(3, np.float32), "score": np.empty(6, np.float32),
               "ownership": np.empty(self.S, np.float32) if wantOwnership else None}
        t = ctypes.c_uint64()
        entry = self._lib.kmx_batcher_submit_packed if packed else self._lib.kmx_batcher_submit
        capi.check(entry(self._p, sp.ctypes.data_as(ctypes.c_void_p) if packed else _fp(sp), _fp(gl), None if mt is None else _fp(mt), int(symmetry),
                         float(policyOptimism), _fp(out["policy"]), _fp(out["value"]), _fp(out["score"]),
                         None if out["ownership"] is None else _fp(out["ownership"]), ctypes.byref(t)), self._lib)
        with self._lock:
            self._out[t.value] = out
        return t.value

    def wait(self, ticket):
        rc = self._lib.kmx_batcher_wait(self._p, ticket)
        with self._lock:
            out = self._out.pop(ticket, None)
        capi.check(rc, self._lib)
        return out

    def stats(self):
        r, b = ctypes.c_uint64(), ctypes.c_uint64()
        capi.check(self._lib.kmx_batcher_stats(self._p, ctypes.byref(r), ctypes.byref(b)), self._lib)
        return r.value, b.value

    def effectiveBatch(self):
        """kmx_batcher_effective_batch: the largest batch this batcher ever launches (min(maxBatchSize, the device's granule))."""
        return int(self._lib.kmx_batcher_effective_batch(self._p))


# ---- layer test hooks (nninterface.h:134-180) ------------------------------------------------------

def _conv_desc(w_oihw):
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    oc, ic, ky, kx = w.shape
    return capi.ConvDesc(ky, kx, ic, oc, _fp(w)), w


def _bn_desc(scale, bias, activation):
    s = np.ascontiguousarray(scale, dtype=np.float32)
    b = np.ascontiguousarray(bias, dtype=np.float32)
    return capi.BnActDesc(len(s), activation, _fp(s), _fp(b)), (s, b)


def _prec(useFP16):
    if isinstance(useFP16, str):
        return {"fp16": capi.PREC_FP16, "bf16": capi.PREC_BF16, "fp32": capi.PREC_FP32, "auto": capi.PREC_AUTO}[useFP16]
    return capi.PREC_AUTO if useFP16 else capi.PREC_FP32


def testEvaluatePointwisePair(batchSize, nnXLen, nnYLen, useFP16, x, resid, w1, s1, b1, act1, w2, s2, b2, act2, mask, fused):
    """kmx_test_pointwise_pair: the seam of two 1x1 convolutions between nested-bottleneck blocks, as the one-launch kernel
    (fused=True) or as the two convolution launches it replaces. Arrays: x [cells][c1], resid [cells][c2], w [out][in],
    mask [cells] or None. Returns (trunk_raw [cells][c2], mid_raw [cells][c3], mid_act [cells][c3])."""
    lib = capi.load_library()
    cells = batchSize * nnXLen * nnYLen
    c = [np.ascontiguousarray(v, dtype=np.float32) for v in (x, resid, w1, s1, b1, w2, s2, b2)]
    c1, c2, c3 = c[0].shape[1], c[1].shape[1], c[5].shape[0]
    assert c[0].shape[0] == cells and c[2].shape == (c2, c1) and c[5].shape == (c3, c2)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.float32)
    out = [np.empty((cells, c2), np.float32), np.empty((cells, c3), np.float32), np.empty((cells, c3), np.float32)]
    capi.check(lib.kmx_test_pointwise_pair(batchSize, nnXLen, nnYLen, _prec(useFP16), c1, c2, c3, _fp(c[0]), _fp(c[1]), _fp(c[2]), _fp(c[3]),
                                           _fp(c[4]), act1, _fp(c[5]), _fp(c[6]), _fp(c[7]), act2, None if m is None else _fp(m),
                                           1 if fused else 0, _fp(out[0]), _fp(out[1]), _fp(out[2])), lib)
    return out


def testEvaluateConvChain(batchSize, nnXLen, nnYLen, useFP16, x, r, w, scale, bias, activation, mask, chained):
    """kmx_test_conv_chain: 2 or 4 convolutions 3x3 192 -> 192 (one or two residual blocks on a 192-channel stream) as separate
    launches (chained=0) or as launches of `chained` convolutions with the activated image handed over inside the CU. Arrays: x, r
    [cells][192]; w [n_conv][192][192][3][3]; scale, bias [n_conv][192]; mask [cells] or None. Returns (r, x) after the last block."""
    lib = capi.load_library()
    cells = batchSize * nnXLen * nnYLen
    c = [np.ascontiguousarray(v, dtype=np.float32) for v in (x, r, w, scale, bias)]
    n_conv = c[2].shape[0]
    assert c[0].shape == (cells, 192) and c[1].shape == (cells, 192) and c[2].shape == (n_conv, 192, 192, 3, 3) and c[3].shape == (n_conv, 192)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.float32)
    out = [np.empty((cells, 192), np.float32), np.empty((cells, 192), np.float32)]
    capi.check(lib.kmx_test_conv_chain(batchSize, nnXLen, nnYLen, _prec(useFP16), n_conv, _fp(c[0]), _fp(c[1]), _fp(c[2]), _fp(c[3]), _fp(c[4]),
                                       activation, None if m is None else _fp(m), chained, _fp(out[0]), _fp(out[1])), lib)
    return out


def testEvaluateConv(w_oihw, batchSize, nnXLen, nnYLen, useFP16, inputNHWC):
    lib = capi.load_library()
    d, keep = _conv_desc(w_oihw)
    x = np.ascontiguousarray(inputNHWC, dtype=np.float32)
    out = np.empty((batchSize, nnYLen, nnXLen, d.out_channels), dtype=np.float32)
    capi.check(lib.kmx_test_conv(ctypes.byref(d), batchSize, nnXLen, nnYLen, _prec(useFP16), _fp(x), _fp(out)), lib)
    return out


def testEvaluateBatchNorm(scale, bias, activation, batchSize, nnXLen, nnYLen, useFP16, inputNHWC, maskNHW):
    lib = capi.load_library()
    d, keep = _bn_desc(scale, bias, activation)
    x = np.ascontiguousarray(inputNHWC, dtype=np.float32)
    m = np.ascontiguousarray(maskNHW, dtype=np.float32)
    out = np.empty_like(x)
    capi.check(lib.kmx_test_bnact(ctypes.byref(d), batchSize, nnXLen, nnYLen, _prec(useFP16), _fp(x), _fp(m), _fp(out)), lib)
    return out


def testEvaluateResidualBlock(block, batchSize, nnXLen, nnYLen, useFP16, inputNHWC, maskNHW):
    """block: dict(pre=(scale,bias,act), conv1=w_oihw, mid=(scale,bias,act), conv2=w_oihw)"""
    lib = capi.load_library()
    pre, k1 = _bn_desc(*block["pre"])
    c1, k2 = _conv_desc(block["conv1"])
    mid, k3 = _bn_desc(*block["mid"])
    c2, k4 = _conv_desc(block["conv2"])
    d = capi.ResBlockDesc(pre, c1, mid, c2)
    x = np.ascontiguousarray(inputNHWC, dtype=np.float32)
    m = np.ascontiguousarray(maskNHW, dtype=np.float32)
    out = np.empty_like(x)
    capi.check(lib.kmx_test_resblock(ctypes.byref(d), batchSize, nnXLen, nnYLen, _prec(useFP16), _fp(x), _fp(m), _fp(out)), lib)
    return out


def testEvaluateGlobalPoolingResidualBlock(block, batchSize, nnXLen, nnYLen, useFP16, inputNHWC, maskNHW):
    """block: dict(pre, convr, convg, gbn, gmul (w [3G][R]), mid, conv2)"""
    lib = capi.load_library()
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
    capi.check(lib.kmx_test_gpoolblock(ctypes.byref(d), batchSize, nnXLen, nnYLen, _prec(useFP16), _fp(x), _fp(m), _fp(out)), lib)
    return out

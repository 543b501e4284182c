"""ctypes binding of libkatamx.so (include/katamx.h). No torch types cross this boundary.

The library is HIP-only. Importing this module does not need a GPU (model parsing works anywhere);
creating a compute handle without a usable MI355X raises KatamxError — there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkatamx.so")

KMX_OK = 0
KMX_ERR_INVALID_ARG, KMX_ERR_IO, KMX_ERR_MODEL, KMX_ERR_DEVICE, KMX_ERR_UNSUPPORTED, KMX_ERR_INTERNAL = -1, -2, -3, -4, -5, -6
PREC_AUTO, PREC_FP32, PREC_FP16, PREC_BF16 = 0, 1, 2, 3
PREC_NAMES = {PREC_AUTO: "auto", PREC_FP32: "fp32", PREC_FP16: "fp16", PREC_BF16: "bf16"}
ACT_IDENTITY, ACT_RELU, ACT_MISH, ACT_SILU = 0, 1, 2, 3


class KatamxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("katamx error %d: %s" % (code, msg))
        self.code = code


class ModelInfo(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char * 128),
        ("model_version", ctypes.c_int32),
        ("num_input_channels", ctypes.c_int32),
        ("num_input_global_channels", ctypes.c_int32),
        ("num_input_meta_channels", ctypes.c_int32),
        ("num_policy_channels", ctypes.c_int32),
        ("num_value_channels", ctypes.c_int32),
        ("num_score_value_channels", ctypes.c_int32),
        ("num_ownership_channels", ctypes.c_int32),
        ("trunk_num_channels", ctypes.c_int32),
        ("mid_num_channels", ctypes.c_int32),
        ("num_blocks", ctypes.c_int32),
        ("meta_encoder_version", ctypes.c_int32),
        ("td_score_multiplier", ctypes.c_float),
        ("score_mean_multiplier", ctypes.c_float),
        ("score_stdev_multiplier", ctypes.c_float),
        ("lead_multiplier", ctypes.c_float),
        ("variance_time_multiplier", ctypes.c_float),
        ("shortterm_value_error_multiplier", ctypes.c_float),
        ("shortterm_score_error_multiplier", ctypes.c_float),
        ("output_scale_multiplier", ctypes.c_float),
        ("num_parameters", ctypes.c_int64),
        ("flops_per_position", ctypes.c_double),
    ]


class ConvDesc(ctypes.Structure):
    _fields_ = [("conv_y_size", ctypes.c_int32), ("conv_x_size", ctypes.c_int32), ("in_channels", ctypes.c_int32),
                ("out_channels", ctypes.c_int32), ("weights", ctypes.POINTER(ctypes.c_float))]


class BnActDesc(ctypes.Structure):
    _fields_ = [("num_channels", ctypes.c_int32), ("activation", ctypes.c_int32),
                ("merged_scale", ctypes.POINTER(ctypes.c_float)), ("merged_bias", ctypes.POINTER(ctypes.c_float))]


class MatMulDesc(ctypes.Structure):
    _fields_ = [("in_channels", ctypes.c_int32), ("out_channels", ctypes.c_int32), ("weights", ctypes.POINTER(ctypes.c_float))]


class ResBlockDesc(ctypes.Structure):
    _fields_ = [("pre_bn", BnActDesc), ("regular_conv", ConvDesc), ("mid_bn", BnActDesc), ("final_conv", ConvDesc)]


class GPoolBlockDesc(ctypes.Structure):
    _fields_ = [("pre_bn", BnActDesc), ("regular_conv", ConvDesc), ("gpool_conv", ConvDesc), ("gpool_bn", BnActDesc),
                ("gpool_to_bias_mul", MatMulDesc), ("mid_bn", BnActDesc), ("final_conv", ConvDesc)]


class ProfileEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", ctypes.c_uint64), ("total_ms", ctypes.c_double),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


_FP = ctypes.POINTER(ctypes.c_float)
_FPP = ctypes.POINTER(_FP)
_IP = ctypes.POINTER(ctypes.c_int)

# every symbol include/katamx.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "kmx_abi_version": (ctypes.c_int, []),
    "kmx_global_init": (ctypes.c_int, []),
    "kmx_global_cleanup": (None, []),
    "kmx_device_count": (ctypes.c_int, []),
    "kmx_device_name": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]),
    "kmx_last_error": (ctypes.c_char_p, []),
    "kmx_model_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]),
    "kmx_model_free": (None, [ctypes.c_void_p]),
    "kmx_model_info_get": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelInfo)]),
    "kmx_context_create": (ctypes.c_int, [_IP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "kmx_context_free": (None, [ctypes.c_void_p]),
    "kmx_handle_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "kmx_handle_free": (None, [ctypes.c_void_p]),
    "kmx_handle_precision": (ctypes.c_int, [ctypes.c_void_p]),
    "kmx_eval": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _FPP, _FPP, _IP, _FP, _FPP, _FP, _FP, _FPP]),
    "kmx_eval_packed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), _FPP, _FPP, _IP, _FP,
                                       _FPP, _FP, _FP, _FPP]),
    "kmx_pack_row": (ctypes.c_int, [_FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint8)]),
    "kmx_eval_meta": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _FPP, _FPP, _FPP, _IP, _FP, _FPP, _FP, _FP, _FPP]),
    "kmx_eval_device_meta": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _IP, _FP,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "kmx_eval_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _IP, _FP,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "kmx_batcher_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "kmx_batcher_free": (None, [ctypes.c_void_p]),
    "kmx_batcher_submit": (ctypes.c_int, [ctypes.c_void_p, _FP, _FP, _FP, ctypes.c_int, ctypes.c_float, _FP, _FP, _FP, _FP, ctypes.POINTER(ctypes.c_uint64)]),
    "kmx_batcher_submit_packed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _FP, _FP, ctypes.c_int, ctypes.c_float, _FP, _FP, _FP, _FP, ctypes.POINTER(ctypes.c_uint64)]),
    "kmx_batcher_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64]),
    "kmx_batcher_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "kmx_batcher_precision": (ctypes.c_int, [ctypes.c_void_p]),
    "kmx_batcher_effective_batch": (ctypes.c_int, [ctypes.c_void_p]),
    "kmx_handle_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "kmx_handle_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "kmx_handle_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "kmx_handle_set_profiling": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "kmx_handle_set_graphs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "kmx_handle_graph_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]),
    "kmx_handle_get_profile": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ProfileEntry), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "kmx_handle_set_split_min": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "kmx_test_pointwise_pair": (ctypes.c_int, [ctypes.c_int] * 7 + [_FP, _FP, _FP, _FP, _FP, ctypes.c_int, _FP, _FP, _FP, ctypes.c_int, _FP,
                                               ctypes.c_int, _FP, _FP, _FP]),
    "kmx_test_conv_chain": (ctypes.c_int, [ctypes.c_int] * 5 + [_FP, _FP, _FP, _FP, _FP, ctypes.c_int, _FP, ctypes.c_int, _FP, _FP]),
    "kmx_test_conv": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP]),
    "kmx_test_bnact": (ctypes.c_int, [ctypes.POINTER(BnActDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP]),
    "kmx_test_resblock": (ctypes.c_int, [ctypes.POINTER(ResBlockDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP]),
    "kmx_test_gpoolblock": (ctypes.c_int, [ctypes.POINTER(GPoolBlockDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP]),
    "kmx_test_rmsnorm": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.c_float, _FP, _FP, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP]),
    "kmx_test_attention": (ctypes.c_int, [ctypes.c_int] * 8 + [_FP, _FP, ctypes.c_int, _FP, _FP, _FP, _FP, _FP]),
    "kmx_test_swiglu": (ctypes.c_int, [ctypes.c_int] * 5 + [_FP, _FP, _FP]),
}

# kernel-tuning instrumentation (katago_amd/csrc/katamx_tuning.h): exported by the library, not part of include/katamx.h
TUNING_SIGNATURES = {
    "kmx_bench_conv": (ctypes.c_int, [ctypes.c_int] * 10 + [ctypes.POINTER(ctypes.c_double)]),
    "kmx_bench_conv_streams": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "kmx_bench_conv_chain": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)]),
    "kmx_bench_seam": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_double)]),
    "kmx_debug_conv_cfg": (ctypes.c_int, [ctypes.c_int] * 3 + [_IP, _IP]),
    "kmx_bench_mfma": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)] * 3),
    "kmx_bench_launch_floor": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)]),
    "kmx_bench_mfma_sustained": (ctypes.c_int, [ctypes.c_int] * 4 + [ctypes.c_double] + [ctypes.POINTER(ctypes.c_double)] * 2),
}

_lib = None


def load_library(path=None):
    """Load libkatamx.so (built by katago_amd.build). Fails loudly if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("KMX_LIBRARY") or LIB_PATH  # KMX_LIBRARY: A/B against another build of the same ABI
    # torch wheels bundle their own ROCm runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7). Two HIP
    # runtimes in one process cannot both own the GPU, so torch's must be loaded first: libkatamx.so then binds
    # to the already-loaded soname. (torch is only plumbing here: device buffers, streams, torch.distributed.)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less deployments use the system runtime
        pass
    if not os.path.exists(p):
        raise KatamxError(KMX_ERR_INTERNAL, "%s not found: run `python -m katago_amd.build` (the HIP extension is mandatory)" % p)
    lib = ctypes.CDLL(p)
    for name, (res, args) in list(SIGNATURES.items()) + list(TUNING_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status, lib=None):
    if status != KMX_OK:
        lib = lib or load_library()
        raise KatamxError(status, (lib.kmx_last_error() or b"").decode("utf-8", "replace"))

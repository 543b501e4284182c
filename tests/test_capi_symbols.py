"""The C ABI library loads (no GPU needed) and exports every symbol include/katamx.h declares."""
import ctypes
import os
import re

from katago_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header=os.path.join("include", "katamx.h")):
    text = open(os.path.join(REPO, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kmx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 25
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libkatamx.so does not export %s" % s
    assert sorted(capi.SIGNATURES) == syms, "capi.SIGNATURES is out of sync with include/katamx.h"
    # the public header holds the boundary only: kernel-tuning instrumentation lives in a private header (ABI 7)
    assert not [s for s in syms if s.startswith(("kmx_bench_", "kmx_debug_"))]
    tuning = declared_symbols(os.path.join("katago_amd", "csrc", "katamx_tuning.h"))
    assert sorted(capi.TUNING_SIGNATURES) == tuning and len(tuning) == 8
    for s in tuning:
        assert hasattr(lib, s), "libkatamx.so does not export %s" % s


def test_abi_version_and_errors_without_gpu():
    lib = capi.load_library()
    assert lib.kmx_abi_version() == 7
    p = ctypes.c_void_p()
    rc = lib.kmx_model_load(b"/nonexistent/model.bin.gz", b"", ctypes.byref(p))
    assert rc == capi.KMX_ERR_IO and b"model.bin.gz" in lib.kmx_last_error()
    rc = lib.kmx_model_load(b"/nonexistent/model.xyz", b"", ctypes.byref(p))
    assert rc == capi.KMX_ERR_MODEL


def test_no_oracle_in_product():
    """The product path must never route through the oracle (or any CPU fallback)."""
    for root, _, files in os.walk(os.path.join(REPO, "katago_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "okmx_" not in src and "kmx_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_tuning_hooks_refuse_bad_arguments_without_gpu():
    """kmx_bench_launch_floor (round 5) checks its arguments before it looks for a device."""
    lib = capi.load_library()
    us = ctypes.c_double()
    for args in ((0, 8192, 0, 10, 1), (18, 4096, 0, 10, 1), (18, 8192, 3, 10, 1), (18, 8192, 0, 0, 1), (18, 200 * 1024, 0, 10, 1)):
        assert lib.kmx_bench_launch_floor(*args, ctypes.byref(us)) == capi.KMX_ERR_INVALID_ARG, args
    assert lib.kmx_bench_launch_floor(18, 8192, 0, 10, 1, None) == capi.KMX_ERR_INVALID_ARG
    # kmx_bench_mfma_sustained (round 6): work-groups, loop shape 0 | 1, operand kind 0..3, a 16-bit precision, 0 < seconds <= 30
    tf, mhz = ctypes.c_double(), ctypes.c_double()
    for args in ((0, 1, 2, capi.PREC_FP16, 1.0), (256, 6, 2, capi.PREC_FP16, 1.0), (256, 1, 4, capi.PREC_FP16, 1.0),
                 (256, 1, 2, capi.PREC_FP32, 1.0), (256, 1, 2, capi.PREC_BF16, 0.0), (256, 1, 2, capi.PREC_BF16, 31.0)):
        assert lib.kmx_bench_mfma_sustained(*args, ctypes.byref(tf), ctypes.byref(mhz)) == capi.KMX_ERR_INVALID_ARG, args
    assert lib.kmx_bench_mfma_sustained(256, 1, 2, capi.PREC_FP16, 1.0, None, None) == capi.KMX_ERR_INVALID_ARG


def test_the_cpu_twin_of_the_abi_defines_what_the_binding_can_call(tmp_path):
    """oracle/kmx_abi_on_oracle.cpp implements include/katamx.h on the CPU oracle so that the SAME binding object links against either
    (test infrastructure: katago_oracle, katago_oraclex). An entry point added to the header and missed there is an unresolved symbol for
    every embedder built against the header and linked with the twin (round 5: kmx_batcher_effective_batch, ABI 7). The layer / device
    test hooks (kmx_test_*, kmx_eval_device ...) that only exist on a GPU are listed and must not grow silently."""
    import subprocess

    obj = str(tmp_path / "twin.o")
    subprocess.run(["g++", "-std=c++17", "-c", "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(REPO, "oracle"),
                    os.path.join(REPO, "oracle", "kmx_abi_on_oracle.cpp"), "-o", obj], check=True)
    out = subprocess.run(["nm", "-g", "--defined-only", obj], check=True, capture_output=True, text=True).stdout
    defined = set(re.findall(r"\b(kmx_[a-z0-9_]+)$", out, re.M))
    # what the binding and this repo's evaluator call (integration/*.cpp, *.h): all of it must exist in the twin
    used = set()
    for f in os.listdir(os.path.join(REPO, "integration")):
        if f.endswith((".cpp", ".h")) and f != "leaf_pump.cpp":  # (leaf_pump is the synthetic consumer of the device library only)
            text = re.sub(r"//[^\n]*|/\*.*?\*/", "", open(os.path.join(REPO, "integration", f)).read(), flags=re.S)
            used |= set(re.findall(r"\b(kmx_[a-z0-9_]+)\s*\(", text))
    used &= set(declared_symbols())
    assert len(used) >= 20 and "kmx_batcher_submit_packed" in used, sorted(used)
    assert sorted(used - defined) == [], "the CPU twin lacks entry points the binding calls"
    # and the one the advisor found missing in round 5: declared, exported by the device library, callable on the twin
    assert "kmx_batcher_effective_batch" in defined

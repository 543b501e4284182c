"""Whole nets through a CPU build of the product: engine.cpp, model_desc.cpp, kmx_api.cpp, misc_kernels.hip and
transformer_kernels.hip are compiled UNCHANGED for x86 against tests/fakehip/emul/hip/hip_runtime.h (work-group = OS
threads, __syncthreads = barrier, LDS = static buffer, device memory = host memory) and run on the CPU; only the MFMA
convolution kernel, which cannot be emulated (and is verified on the MI355X), is replaced by a plain-loop executor of the
ConvArgs contract of kernels.h (tests/fakehip/emulate_engine.cpp). Outputs are compared with the reference PyTorch goldens
and with the oracle at the 16-bit tolerances of the GPU tests.

This covers on the CPU what the schedule dry-run cannot: that the launches, taken together, COMPUTE the right thing —
weight re-tiling and swizzle, buffer reuse, strides and channel offsets, the small kernels — and it is the only numerical
check so far of the transformer device path, whose kernels have not run on hardware yet."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

FAKE = os.path.join(REPO, "tests", "fakehip")
CSRC = os.path.join(REPO, "katago_amd", "csrc")
SOURCES = [os.path.join(FAKE, "emulate_engine.cpp")] + [os.path.join(CSRC, f) for f in
                                                          ("misc_kernels.hip", "transformer_kernels.hip", "engine.cpp", "model_desc.cpp", "kmx_api.cpp")]


@pytest.fixture(scope="module")
def emu_lib(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("emueng"))
    cxx = ["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++20", "-O2", "-fPIC", "-pthread", "-I" + os.path.join(FAKE, "emul"), "-I" + FAKE]
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(d, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen(cxx + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, "%s:\n%s" % (src, out[-3000:])
    so = os.path.join(d, "libkatamx_emu.so")
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-shared", "-fPIC", "-pthread", "-o", so] + objs + ["-lz"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return so


def run_cases(emu_lib, cases, transformer=False, attention="valu"):
    """attention: which of the two attention kernels the transformer nets use. Emulating a matrix-core instruction costs two
    thread barriers per MFMA and wave, so the whole-net runs use the plain kernel except where stated; the matrix-core
    kernel is checked alone, shape by shape, in tests/test_transformer_kernels_emulated.py."""
    env = dict(os.environ)
    env.pop("KMX_EXPERIMENTAL_TRANSFORMER", None)
    env.pop("KMX_ATTENTION_VALU", None)
    if transformer:
        env["KMX_EXPERIMENTAL_TRANSFORMER"] = "1"
        if attention == "valu":
            env["KMX_ATTENTION_VALU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(FAKE, "run_emulated_nets.py"), emu_lib] + cases, capture_output=True, text=True, timeout=1800, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    return json.loads(p.stdout.split("RESULT ")[1])


def check(res, rel, ab):
    for case, dev in res.items():
        assert dev["finite"], case
        for k in ("policy", "value", "score", "ownership"):
            err, scale = dev[k]
            assert err <= ab + rel * scale, (case, k, err, scale)


def test_convolutional_nets_emulated(emu_lib):
    """Nested bottleneck + gpool (v15), the v8 layout with a 5x5 stem and gpool blocks, the reference PyTorch golden and
    the sgf-metadata net; masked 13x9 and 9x9 boards, symmetries, optimism."""
    res = run_cases(emu_lib, ["bf16:gen_b3c64nbt_v15", "fp16:gen_b6c96_v8", "bf16:torch_nbt", "fp16:torch_nbt", "bf16:torch_meta"])
    check({k: v for k, v in res.items() if k.startswith("bf16")}, 0.03, 0.08)
    check({k: v for k, v in res.items() if k.startswith("fp16")}, 0.03, 0.02)


def test_entry_point_variants_emulated(emu_lib):
    """kmx_eval_packed, the two-engine split of a batch, single rows: bit-identical to kmx_eval; counters add up."""
    res = run_cases(emu_lib, ["bf16:api_variants"])["bf16:api_variants"]
    assert res["finite"] and res["packed_equal"] and res["alone_equal"] and res["split_equal"] and res["small_equal"], res
    assert res["stats"] == [7 + 7 + 1, 3, 7 + 3, 2], res


def test_transformer_nets_emulated(emu_lib):
    """The transformer device path against the reference PyTorch goldens: attention + SwiGLU FFN trunk with fixed RoPE and a
    per-cell RMSNorm tip (tfa); grouped-query attention, learnable RoPE, a nested transformer bottleneck beside a
    convolutional one, per-board RMSNorm tip (tfb)."""
    res = run_cases(emu_lib, ["bf16:torch_tfa", "bf16:torch_tfb", "fp16:torch_tfa", "fp16:torch_tfb"], transformer=True)
    check({k: v for k, v in res.items() if k.startswith("bf16")}, 0.03, 0.08)
    check({k: v for k, v in res.items() if k.startswith("fp16")}, 0.03, 0.02)
    # fp16 keeps 11 bits: the path is not merely "within tolerance", it tracks the fp32 reference to ~1e-3
    assert res["fp16:torch_tfa"]["policy"][0] < 5e-3 and res["fp16:torch_tfb"]["policy"][0] < 5e-3
    # the default (matrix-core) attention kernel inside a whole net
    res = run_cases(emu_lib, ["fp16:torch_tfa"], transformer=True, attention="mfma")
    check(res, 0.03, 0.02)
    assert res["fp16:torch_tfa"]["policy"][0] < 5e-3


def test_reference_transformer_nets_emulated(emu_lib):
    """The two trained transformer nets the reference ships for its GPU tests, bf16, against the oracle."""
    nets = ["b7c96h3tfrs-test5-cnorm.bin.gz", "b7c96h6kv3qk32v16tflrs-fson-bnh.bin.gz"]
    if not all(os.path.exists(os.path.join(REPO, "oracle", "_ref", "models", f)) for f in nets):
        pytest.skip("reference test nets not packaged")
    res = run_cases(emu_lib, ["bf16:" + f for f in nets], transformer=True)
    check(res, 0.05, 0.15)


def test_emulated_library_refuses_transformer_nets_without_opt_in(emu_lib):
    env = dict(os.environ)
    env.pop("KMX_EXPERIMENTAL_TRANSFORMER", None)
    p = subprocess.run([sys.executable, os.path.join(FAKE, "run_emulated_nets.py"), emu_lib, "bf16:torch_tfa"], capture_output=True, text=True, env=env)
    assert p.returncode != 0 and "not supported" in (p.stdout + p.stderr)

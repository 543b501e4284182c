"""Build libkatamx.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m katago_amd.build            # incremental
    python -m katago_amd.build --force

The shared library lands next to this file (katago_amd/libkatamx.so): it is git-ignored but travels
with the tree to the GPU box. hipcc cross-compiles gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libkatamx.so")

SOURCES = ["conv_mfma.hip", "conv_chain.hip", "pointwise.hip", "conv_bench.hip", "conv_f32.hip", "misc_kernels.hip", "transformer_kernels.hip", "engine.cpp", "model_desc.cpp", "kmx_api.cpp", "batcher.cpp", "numa.cpp"]
HEADERS = ["kernels.h", "device_common.h", "conv_kernel.h", "conv_chain_kernel.h", "conv_small_kernel.h", "pointwise_kernel.h", "pointwise2_kernel.h", "pointwise3_kernel.h", "engine.h", "model_desc.h", "katamx_tuning.h", "numa.h", os.path.join("..", "..", "include", "katamx.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: left on, the SLP vectoriser packs adjacent fp32 epilogue arithmetic into v_pk_*_f32 instructions - which
# issue no faster than two plain ones on gfx950 and cost ~1.6 v_mov_b64 per value to marshal operands into aligned register
# pairs (233 of them per tile of the seam kernel; profiles/r03_steps)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-x", "hip"]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def _source_hash():
    """sha256 over every source and header (and the flags): what the shipped library was built from."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    for rel in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read())
    return h.hexdigest()


STAMP = LIB + ".srchash"


def up_to_date():
    """True when libkatamx.so exists and was built from exactly the sources in the tree. Decided by content, not by
    mtimes: the object directory does not travel to the GPU box (.gpurunignore) and a copied tree's mtimes mean
    nothing, so tests (conftest) and bench.py both load the SAME shipped binary instead of one of them rebuilding it."""
    try:
        with open(STAMP) as f:
            return os.path.exists(LIB) and f.read().strip() == _source_hash()
    except OSError:
        return False


PUMP = os.path.join(HERE, "leaf_pump")
PUMP_SRC = os.path.join(HERE, "..", "integration", "leaf_pump.cpp")


def build_leaf_pump(verbose=True):
    """integration/leaf_pump.cpp -> katago_amd/leaf_pump: a plain C++ consumer of include/katamx.h (g++, no HIP) that keeps
    many leaves in flight through the batcher; what tests/test_gpu_leaf_pump.py runs."""
    if os.path.exists(PUMP) and _mtime(PUMP) >= max(_mtime(PUMP_SRC), _mtime(LIB)):
        return PUMP
    cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-I" + os.path.join(HERE, "..", "include"), "-o", PUMP, PUMP_SRC,
           "-L" + HERE, "-lkatamx", "-Wl,-rpath,$ORIGIN", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building leaf_pump failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("[katago_amd.build] built", PUMP, flush=True)
    return PUMP


def build(force=False, verbose=True):
    if not force and up_to_date():
        build_leaf_pump(verbose)
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    newest_header = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
    todo = []
    objs = []
    for s in SOURCES:
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _mtime(obj) < max(_mtime(os.path.join(CSRC, s)), newest_header):
            todo.append(s)
    if todo:
        if verbose:
            print("[katago_amd.build] hipcc gfx950:", " ".join(todo), flush=True)
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(_compile, todo))
    if todo or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[katago_amd.build] linked", LIB, flush=True)
    with open(STAMP, "w") as f:
        f.write(_source_hash() + "\n")
    build_leaf_pump(verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

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
                                                          ("misc_kernels.hip", "transformer_kernels.hip", "engine.cpp", "model_desc.cpp", "kmx_api.cpp", "batcher.cpp", "numa.cpp")]


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


# ---- the "real convolution" build: conv_mfma.hip / conv_kernel.h themselves, emulated -----------------------------------
# The kernel source is used as it is except for the statements that only exist on the GPU, which are rewritten at test time
# (the product file is not touched): s_waitcnt / register-class asm statements are dropped, the dynamic LDS declaration
# becomes the emulator's buffer, global_load_lds becomes a per-lane copy that happens at once or (KMX_EMU_LATE_DMA=1 / 2) as late as
# the wave's s_waitcnt vmcnt(N) statements - which become emu::waitVm(N) - allow. v_mfma_f32_32x32x16 and
# v_permlane32_swap run as wave-collectives in the hardware's register layout.
CONV_REWRITES = [
    (r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\(N\) : "memory"\);', "emu::waitVm(N);", 1),
    (r'asm volatile\("" : "\+s"\(sTap\)\);', ";", 2),
    (r'asm volatile\("" ::"v"\((rawQ|actQ)\[j\]\)\);', ";", 2),
    (r'asm volatile\("" : "\+v"\(pOff\)\);', ";", 1),
    (r'extern __shared__ __attribute__\(\(aligned\(256\)\)\) char smem\[\];', "char* const smem = (char*)emu::dynLds();", 1),
    (r'__builtin_amdgcn_global_load_lds\(', "emu::globalLoadLds(", 2),
    (r'__attribute__\(\(amdgpu_waves_per_eu\(2, 2\)\)\)', "", 1),
]


# pointwise_kernel.h (the fused seam of two 1x1 convolutions) gets the same treatment
# Plain global loads and stores take their place in the wave's in-order request queue (emu::vmNote): the seam kernels' counts include
# them ("in flight at most this part's residual loads and stores").
PW_REWRITES = [
    (r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\(N\) : "memory"\);', "emu::waitVm(N);", 1),
    (r'for\(int j = 0; j < 2; j\+\+\) rq\[pt\]\[ct\]\[j\] = \*\(const u32x4\*\)\(live \? rrow \+ ct \* 32 \+ 16 \* j : rrow\);',
     "for(int j = 0; j < 2; j++) { rq[pt][ct][j] = *(const u32x4*)(live ? rrow + ct * 32 + 16 * j : rrow); emu::vmNote(); }", 1),
    (r'\*\(u32x4\*\)\(live \? rawRow \+ c : trash\) = rawQ\[j\];', "*(u32x4*)(live ? rawRow + c : trash) = rawQ[j]; emu::vmNote();", 1),
    (r'if\(hasActOut\) \*\(u32x4\*\)\(live \? actRow \+ c : trash\) = oq\[j\];',
     "if(hasActOut) { *(u32x4*)(live ? actRow + c : trash) = oq[j]; emu::vmNote(); }", 1),
    (r'\*\(u32x4\*\)\(rawRow \+ c\) = rq\[j\];', "*(u32x4*)(rawRow + c) = rq[j]; emu::vmNote();", 1),
    (r'\*\(u32x4\*\)\(actRow \+ c\) = oq\[j\];', "*(u32x4*)(actRow + c) = oq[j]; emu::vmNote();", 1),
    (r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', ";", 1),
    (r'extern __shared__ __attribute__\(\(aligned\(256\)\)\) char smemPw\[\];', "char* const smemPw = (char*)emu::dynLds();", 1),
    (r'__builtin_amdgcn_global_load_lds\(', "emu::globalLoadLds(", 1),
    (r'__attribute__\(\(amdgpu_waves_per_eu\(2, 2\)\)\)', "", 1),
]


# pointwise2_kernel.h (the persistent, software-pipelined seam kernel)
PW2_REWRITES = [
    (r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\(N\) : "memory"\);', "emu::waitVm(N);", 1),
    (r'for\(int j = 0; j < 2; j\+\+\) dst\[j\] = \*\(const GLOBAL u32x4\*\)\(rrow \+ 64 \* q \+ 16 \* j\);',
     "for(int j = 0; j < 2; j++) { dst[j] = *(const GLOBAL u32x4*)(rrow + 64 * q + 16 * j); emu::vmNote(); }", 1),
    (r'for\(int j = 0; j < 2; j\+\+\) \*\(GLOBAL u32x4\*\)\(rawRow \+ 64 \* q \+ 16 \* j\) = rawQ\[j\];',
     "for(int j = 0; j < 2; j++) { *(GLOBAL u32x4*)(rawRow + 64 * q + 16 * j) = rawQ[j]; emu::vmNote(); }", 1),
    (r'\*\(GLOBAL u32x4\*\)\(rawRow2 \+ ct \* 32 \+ 16 \* j\) = rawQ\[j\];', "*(GLOBAL u32x4*)(rawRow2 + ct * 32 + 16 * j) = rawQ[j]; emu::vmNote();", 1),
    (r'\*\(GLOBAL u32x4\*\)\(actRow2 \+ ct \* 32 \+ 16 \* j\) = oq\[j\];', "*(GLOBAL u32x4*)(actRow2 + ct * 32 + 16 * j) = oq[j]; emu::vmNote();", 1),
    (r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', ";", 1),
    (r'asm volatile\("" : "\+s"\((w1|w2)\)\);', ";", 2),
    (r'extern __shared__ __attribute__\(\(aligned\(256\)\)\) char smemPw2\[\];', "char* const smemPw2 = (char*)emu::dynLds();", 1),
    (r'__builtin_amdgcn_global_load_lds\(', "emu::globalLoadLds(", 2),
    (r'__attribute__\(\(amdgpu_waves_per_eu\(2, 2\)\)\)', "", 1),
]


# pointwise3_kernel.h (round 5: the seam with its weights resident in AGPRs + LDS, one wave per SIMD). Its inline-asm MFMAs are guarded
# by __AMDGCN__ in the header itself (the emulator's MFMA stand-in runs instead); what is rewritten are the waits, the opaque-register
# statements, the LDS declaration, the LDS-DMA builtin and the plain loads / stores that take their place in the in-order queue.
PW3_REWRITES = [
    (r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\(N\) : "memory"\);', "emu::waitVm(N);", 1),
    (r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', ";", 1),
    (r'for\(int i = 0; i < 2; i\+\+\) rq\[S\]\[j\]\[i\] = \*\(const GLOBAL u32x4\*\)\(rrow \+ 32 \* j \+ 16 \* i\);',
     "for(int i = 0; i < 2; i++) { rq[S][j][i] = *(const GLOBAL u32x4*)(rrow + 32 * j + 16 * i); emu::vmNote(); }", 1),
    (r'for\(int i = 0; i < 2; i\+\+\) \*\(GLOBAL u32x4\*\)\(rawRow \+ 32 \* J \+ 16 \* i\) = rawQ\[i\];',
     "for(int i = 0; i < 2; i++) { *(GLOBAL u32x4*)(rawRow + 32 * J + 16 * i) = rawQ[i]; emu::vmNote(); }", 1),
    (r'\*\(GLOBAL u32x4\*\)\(rawRow2 \+ 16 \* i\) = rawQ\[i\];', "*(GLOBAL u32x4*)(rawRow2 + 16 * i) = rawQ[i]; emu::vmNote();", 1),
    (r'\*\(GLOBAL u32x4\*\)\(actRow2 \+ 16 \* i\) = oq\[i\];', "*(GLOBAL u32x4*)(actRow2 + 16 * i) = oq[i]; emu::vmNote();", 1),
    (r'w1f\[j\]\[c\]\[kk\] = (\*\(const V8\*\)[^;]*;)', r"{ w1f[j][c][kk] = \1 emu::vmNote(); }", 1),
    (r'w2f\[c\]\[kk\] = (\*\(const V8\*\)[^;]*;)', r"{ w2f[c][kk] = \1 emu::vmNote(); }", 1),
    (r'asm volatile\("" : "\+v"\([^;]*\);', ";", 7),
    (r'extern __shared__ __attribute__\(\(aligned\(256\)\)\) char smemPw3\[\];', "char* const smemPw3 = (char*)emu::dynLds();", 1),
    (r'__builtin_amdgcn_global_load_lds\(', "emu::globalLoadLds(", 2),
    (r'__attribute__\(\(amdgpu_waves_per_eu\(1, 1\)\)\)', "", 1),
]


# conv_chain_kernel.h (2 or 4 chained 3x3 convolutions, the activated image handed over in LDS)
CHAIN_REWRITES = [
    (r'asm volatile\("" : "\+s"\(sTap\)\);', ";", 2),
    (r'asm volatile\("" : "\+v"\((pOff|ln|lnE)\)\);', ";", 3),
    (r'extern __shared__ __attribute__\(\(aligned\(256\)\)\) char smemChain\[\];', "char* const smemChain = (char*)emu::dynLds();", 1),
    (r'__attribute__\(\(amdgpu_waves_per_eu\(2, 2\)\)\)', "", 1),
    # plain global loads and stores take their place in the wave's in-order request queue (the waits between two convolutions are
    # vmcnt(0): they must cover the stores to the scratch tensor that the next loop's LDS-DMA reads)
    (r'for\(int j = 0; j < 2; j\+\+\) dst\[j\] = \*\(const u32x4\*\)\(rrow \+ wn \* \(32 \* WN\) \+ ct \* 32 \+ 16 \* j \+ 8 \* khalf\);',
     "for(int j = 0; j < 2; j++) { dst[j] = *(const u32x4*)(rrow + wn * (32 * WN) + ct * 32 + 16 * j + 8 * khalf); emu::vmNote(); }", 1),
    (r'\*\(u32x4\*\)dst = rawQ\[j\];', "*(u32x4*)dst = rawQ[j]; emu::vmNote();", 1),
    (r'\*\(u32x4\*\)dst = actQ\[j\];', "*(u32x4*)dst = actQ[j]; emu::vmNote();", 1),
]


# conv_small_kernel.h (the small-batch 3x3 shape with dedicated fetching waves)
SMALL_REWRITES = [
    (r'asm volatile\("" : "\+s"\(sTap\)\);', ";", 2),
    (r'asm volatile\("" : "\+v"\(pOff\)\);', ";", 1),
    (r'extern __shared__ __attribute__\(\(aligned\(256\)\)\) char smemSmall\[\];', "char* const smemSmall = (char*)emu::dynLds();", 1),
    (r'__attribute__\(\(amdgpu_waves_per_eu\(PACK \? 4 : 2, PACK \? 4 : REGW \? 2 : 3\)\)\)', "", 1),
    # REGW: the weight fragments are plain loads whose waits the kernel writes itself - they take their place in the lane's in-order queue
    # and land when a wait forces them
    (r'dst = \*\(const V8\*\)\(base \+ off\);', "emu::globalLoadReg(&dst, base + off, 16);", 1),
    (r'\(void\)frag;  // \(the CPU emulation: emu::waitVm\(N\)\)', "emu::waitVm(N);", 1),
]


def _tree_digest(paths, extra):
    import hashlib

    h = hashlib.sha1(repr(extra).encode())
    for path in sorted(paths):
        for root, _, files in (os.walk(path) if os.path.isdir(path) else [(os.path.dirname(path), [], [os.path.basename(path)])]):
            for f in sorted(files):
                if f.endswith((".h", ".hip", ".cpp", ".inc")):
                    h.update(f.encode())
                    h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:20]


def build_emu_full(d, conv_mutations=(), pw2_mutations=(), chain_mutations=(), small_mutations=(), extra_flags=(), so_name="libkatamx_emufull.so", pw3_mutations=(),
                   plain_pointers=False):
    """conv_mutations: further (pattern, replacement, count) rewrites of conv_kernel.h - deliberate defects for the tests of the
    emulator's own teeth. extra_flags: compile AND link flags (tools/emulated_asan.sh: -fsanitize=address ...). plain_pointers: the
    kernels' address_space(3) / (1) qualifiers dropped (tools/emulated_tsan.sh: sanitizers skip accesses through non-default address spaces).
    The build takes a minute and a half (conv_mfma.hip's instantiations) and several test files want the same one: a library is kept under
    $TMPDIR/kmx_emu_builds/<digest of every source, rewrite rule and flag> and built by whoever asks first; the others wait on its lock."""
    import fcntl
    import shutil

    key = _tree_digest([CSRC, FAKE], [CONV_REWRITES, PW_REWRITES, PW2_REWRITES, PW3_REWRITES, CHAIN_REWRITES, SMALL_REWRITES, list(conv_mutations),
                                      list(pw2_mutations), list(chain_mutations), list(small_mutations), list(pw3_mutations), list(extra_flags), so_name, plain_pointers])
    keep = os.path.join(os.environ.get("TMPDIR", "/tmp"), "kmx_emu_builds", key)
    os.makedirs(keep, exist_ok=True)
    kept = os.path.join(keep, so_name)
    with open(os.path.join(keep, "lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(kept):
            import time

            for other in os.listdir(os.path.dirname(keep)):  # builds of sources that no longer exist: gone after three days
                o = os.path.join(os.path.dirname(keep), other)
                if other != key and os.path.isdir(o) and time.time() - os.path.getmtime(o) > 3 * 86400:
                    shutil.rmtree(o, ignore_errors=True)
            built = _build_emu_full(d, conv_mutations, pw2_mutations, chain_mutations, small_mutations, extra_flags, so_name, pw3_mutations, plain_pointers)
            shutil.copyfile(built, kept + ".part")
            os.replace(kept + ".part", kept)
    mine = os.path.join(d, so_name)  # (callers such as tools/emulated_asan.sh expect the library in the directory they named)
    if not os.path.exists(mine):
        try:
            os.link(kept, mine)
        except OSError:
            shutil.copyfile(kept, mine)
    return mine


def _build_emu_full(d, conv_mutations, pw2_mutations, chain_mutations, small_mutations, extra_flags, so_name, pw3_mutations, plain_pointers):
    import re
    import shutil

    def put(name, text):
        if plain_pointers:
            text = text.replace("__attribute__((address_space(3)))", "").replace("__attribute__((address_space(1)))", "")
        open(os.path.join(d, name), "w").write(text)

    src = open(os.path.join(CSRC, "conv_kernel.h")).read()
    for pat, rep, count in list(CONV_REWRITES) + list(conv_mutations):
        src, k = re.subn(pat, rep, src)
        assert k == count, "conv_kernel.h changed: %r matched %d times, expected %d" % (pat, k, count)
    put("conv_kernel.h", src)
    shutil.copy(os.path.join(CSRC, "conv_mfma.hip"), os.path.join(d, "conv_mfma.hip"))
    src = open(os.path.join(CSRC, "conv_chain_kernel.h")).read()
    for pat, rep, count in list(CHAIN_REWRITES) + list(chain_mutations):
        src, k = re.subn(pat, rep, src)
        assert k == count, "conv_chain_kernel.h changed: %r matched %d times, expected %d" % (pat, k, count)
    put("conv_chain_kernel.h", src)
    shutil.copy(os.path.join(CSRC, "conv_chain.hip"), os.path.join(d, "conv_chain.hip"))
    src = open(os.path.join(CSRC, "conv_small_kernel.h")).read()
    for pat, rep, count in list(SMALL_REWRITES) + list(small_mutations):
        src, k = re.subn(pat, rep, src)
        assert k == count, "conv_small_kernel.h changed: %r matched %d times, expected %d" % (pat, k, count)
    put("conv_small_kernel.h", src)
    src = open(os.path.join(CSRC, "pointwise_kernel.h")).read()
    for pat, rep, count in PW_REWRITES:
        src, k = re.subn(pat, rep, src)
        assert k == count, "pointwise_kernel.h changed: %r matched %d times, expected %d" % (pat, k, count)
    put("pointwise_kernel.h", src)
    src = open(os.path.join(CSRC, "pointwise2_kernel.h")).read()
    for pat, rep, count in list(PW2_REWRITES) + list(pw2_mutations):
        src, k = re.subn(pat, rep, src)
        assert k == count, "pointwise2_kernel.h changed: %r matched %d times, expected %d" % (pat, k, count)
    put("pointwise2_kernel.h", src)
    src = open(os.path.join(CSRC, "pointwise3_kernel.h")).read()
    for pat, rep, count in list(PW3_REWRITES) + list(pw3_mutations):
        src, k = re.subn(pat, rep, src)
        assert k == count, "pointwise3_kernel.h changed: %r matched %d times, expected %d" % (pat, k, count)
    put("pointwise3_kernel.h", src)
    shutil.copy(os.path.join(CSRC, "pointwise.hip"), os.path.join(d, "pointwise.hip"))
    cxx = ["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++20", "-O2", "-fPIC", "-pthread", "-I" + os.path.join(FAKE, "emul"), "-I" + FAKE,
           "-I" + CSRC, "-DKMX_EMU_REAL_CONV"] + list(extra_flags)
    procs, objs = [], []
    for src_file in [os.path.join(d, "conv_mfma.hip"), os.path.join(d, "conv_chain.hip"), os.path.join(d, "pointwise.hip"), os.path.join(CSRC, "conv_f32.hip")] + SOURCES:
        obj = os.path.join(d, os.path.splitext(os.path.basename(src_file))[0] + ".o")
        objs.append(obj)
        procs.append((src_file, subprocess.Popen(cxx + ["-c", src_file, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src_file, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, "%s:\n%s" % (src_file, out[-3000:])
    so = os.path.join(d, so_name)
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-shared", "-fPIC", "-pthread"] + list(extra_flags) + ["-o", so] + objs + ["-lz"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return so


@pytest.fixture(scope="module")
def emu_full_lib(tmp_path_factory):
    return build_emu_full(str(tmp_path_factory.mktemp("emufull")))


CONV_ONLY = r"""
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn
rng = np.random.default_rng(2)
out = {}
for (ks, cin, cout, X, Y, n) in json.loads(sys.argv[2]):
    w = (rng.normal(size=(cout, cin, ks, ks)) * 0.1).astype(np.float32)
    x = rng.normal(size=(n, Y * X, cin)).astype(np.float32)
    got = np.asarray(nn.testEvaluateConv(w, n, X, Y, True, x))
    xt = torch.from_numpy(x.reshape(n, Y, X, cin).transpose(0, 3, 1, 2)).to(torch.bfloat16).float()
    wt = torch.from_numpy(w).to(torch.bfloat16).float()
    want = torch.nn.functional.conv2d(xt, wt, padding=ks // 2).numpy().transpose(0, 2, 3, 1).reshape(n, Y * X, cout)
    out["conv%%d_%%d_%%d" %% (ks, cin, cout)] = [float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(want).max()),
                                              hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()]
print("RESULT " + json.dumps(out))
""" % (REPO,)
# 3x3 on the 4-wave x 32-channel shape (every wave issues weights and image pieces, padded to a constant count), 1x1 (whole images
# ride the ring), 5x5, and 3x3 96 -> 192 which KMX_CONV_TUNE min_wgs8=1 sends to the 8-wave shape whose waves 0-3 issue all requests
LATE_DMA_SHAPES = [(3, 64, 32, 9, 9, 1), (1, 96, 64, 13, 13, 1), (5, 32, 64, 9, 9, 1), (3, 96, 192, 19, 19, 1)]


def conv_only(lib, env, shapes=None):
    # (loaders=0: the small 3x3 case is to take the padded 4-wave shape of conv_kernel.h here; the shape with fetching waves,
    # conv_small_kernel.h, has its own tests. The 1x1 case takes the deep-ring shape, cfg 114; the two-step ring and the other deep
    # shapes are in test_kernels_latest_completion.py::test_deep_ring_1x1_shapes)
    env = dict(env)
    tune = "min_wgs8=1,loaders=0" + ("," + env.pop("KMX_CONV_TUNE_MORE") if "KMX_CONV_TUNE_MORE" in env else "")
    return ([sys.executable, "-c", CONV_ONLY, lib, json.dumps(shapes or LATE_DMA_SHAPES)], dict(os.environ, KMX_CONV_TUNE=tune, **env))


def run_parallel(cmds_envs, timeout=1800):
    """The emulated kernels are slow (a rendezvous of 64 OS threads per emulated MFMA): independent variants run side by side."""
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for cmd, env in cmds_envs]
    out = []
    for p in procs:
        o, e = p.communicate(timeout=timeout)
        out.append((p.returncode, o, e))
    return out


def run_cases(emu_lib, cases, transformer=False, attention="valu"):
    """attention: which of the two attention kernels the transformer nets use. Emulating a matrix-core instruction costs a
    thread barrier per MFMA and wave, so the whole-net runs use the plain kernel except where stated; the matrix-core
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


def test_fp32_verification_mode_emulated(emu_lib, emu_full_lib):
    """KMX_PREC_FP32 (useFP16Mode = False, nninterface.h:50-63; round 5): the same schedule on four-byte tensors, the small kernels
    instantiated for float, one plain launch per convolution. Against the fp32 oracle and the PyTorch goldens at 1e-4 of the value (what
    a different summation order and the hardware's exp / reciprocal leave), with the engine's contract executor and with the product's
    own fp32 convolution kernel (conv_f32.hip, the "real convolution" build); the handle reports fp32."""
    cases = ["fp32:gen_b3c64nbt_v15", "fp32:gen_b6c96_v8", "fp32:torch_nbt", "fp32:torch_meta"]
    check(run_cases(emu_lib, cases), 1e-4, 1e-4)
    check(run_cases(emu_full_lib, ["fp32:gen_b3c64nbt_v15", "fp32:torch_nbt"]), 1e-4, 1e-4)
    code = r"""
import sys
sys.path.insert(0, %r)
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn, modelgen
nn.globalInitialize()
import os
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "kmx_emu_fp32.bin"); modelgen.write_model(p, "b2c32nbt", seed=4)
h = nn.createComputeHandle(nn.createComputeContext([0], 9, 9, useFP16Mode=False), nn.loadModelFile(p), 2)
print("PRECISION", h.precision)
""" % (REPO,)
    r = subprocess.run([sys.executable, "-c", code, emu_lib], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PRECISION fp32" in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_fp16_range_transform_emulated(emu_lib):
    """SURVEY 8 row a23(ii), model_desc.cpp scaledBy8: fp16 engines run the net at 1/8 of its values (the reference's
    applyScale8ToReduceActivations, desc.cpp:2718-2736) with unchanged outputs. Nested bottleneck + gpool + pass MLP with mish
    (v15), the metadata encoder, the v8 layout with relu - against the oracle / the PyTorch goldens, which know nothing of the
    transform; then a net whose activations pass 65504: finite and right with the transform, overflowed without it."""
    res = run_cases(emu_lib, ["fp16:gen_b3c64nbt_v15", "fp16:torch_meta", "fp16:gen_b10c128_v14"])
    check(res, 0.03, 0.02)
    big = run_cases(emu_lib, ["fp16:big_activations"])["fp16:big_activations"]
    assert big["finite"] and not big["finite_without_transform"], big
    for k in ("policy", "value", "score", "ownership"):
        err, scale = big[k]
        assert scale > 1.0e3 and err <= 0.02 * scale, (k, err, scale)  # logits of a net 2e4 times too big, relative to themselves


def test_entry_point_variants_emulated(emu_lib):
    """kmx_eval_packed, the two-engine split of a batch, single rows: bit-identical to kmx_eval; counters add up."""
    res = run_cases(emu_lib, ["bf16:api_variants"])["bf16:api_variants"]
    assert res["finite"] and res["packed_equal"] and res["alone_equal"] and res["split_equal"] and res["small_equal"], res
    assert res["stats"] == [7 + 7 + 1, 3, 7 + 3, 2], res


def test_leaf_batcher_emulated(emu_lib):
    """kmx_batcher_*: three submitter threads, batches of at most 4 rows, two in flight, rows handed over as fp32 planes and as
    bit planes (kmx_batcher_submit_packed) side by side - every row bit-identical to kmx_eval;
    rows/batches counters; a non-binary feature plane is refused by its own submit call."""
    res = run_cases(emu_lib, ["bf16:batcher"])["bf16:batcher"]
    assert res["equal"] and res["after_error_equal"], res
    assert res["many_tickets_equal"], res  # threads that hold more tickets than the staging sets have rows must not dead-lock
    assert res["rows"] == 10 and 3 <= res["batches"] <= 10, res
    assert "0 or 1" in res["error"], res
    # batches sealed at multiples of a granule below the batch size (the first row may go alone, the device being idle)
    assert res["granule_equal"] and res["granule_stats"][0] == 10 and 2 <= res["granule_stats"][1] <= 6, res


def test_transformer_nets_emulated(emu_lib):
    """The transformer device path against the reference PyTorch goldens: attention + SwiGLU FFN trunk with fixed RoPE and a
    per-cell RMSNorm tip (tfa); grouped-query attention, learnable RoPE, a nested transformer bottleneck beside a
    convolutional one, per-board RMSNorm tip (tfb)."""
    res = run_cases(emu_lib, ["bf16:torch_tfa", "fp16:torch_tfb"], transformer=True)
    check({k: v for k, v in res.items() if k.startswith("bf16")}, 0.03, 0.08)
    check({k: v for k, v in res.items() if k.startswith("fp16")}, 0.03, 0.02)
    # fp16 keeps 11 bits: the path is not merely "within tolerance", it tracks the fp32 reference to ~1e-3
    assert res["fp16:torch_tfb"]["policy"][0] < 5e-3
    # the default (matrix-core) attention kernel inside a whole net
    res = run_cases(emu_lib, ["fp16:torch_tfa@2"], transformer=True, attention="mfma")  # the 13x9 board only: MFMA emulation is slow
    check(res, 0.03, 0.02)
    assert res["fp16:torch_tfa@2"]["policy"][0] < 5e-3


def test_reference_transformer_nets_emulated(emu_lib):
    """The two trained transformer nets the reference ships for its GPU tests, bf16, against the oracle."""
    nets = ["b7c96h3tfrs-test5-cnorm.bin.gz", "b7c96h6kv3qk32v16tflrs-fson-bnh.bin.gz"]
    if not all(os.path.exists(os.path.join(REPO, "oracle", "_ref", "models", f)) for f in nets):
        pytest.skip("reference test nets not packaged")
    res = run_cases(emu_lib, ["bf16:" + f for f in nets], transformer=True)
    check(res, 0.05, 0.15)


PW2_CODE = r"""
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn
import pointwise_ref as ref
rng = np.random.default_rng(5)
batch, L = 2, 13
mask = np.ones((batch, L, L), np.float32); mask[1, :, 9:] = 0; mask[0, 11:, :] = 0
x, resid, w1, s1, b1, w2, s2, b2, m = ref.make_case(rng, batch * L * L, 192, 384, 192, mask.reshape(-1))
fused = nn.testEvaluatePointwisePair(batch, L, L, "bf16", x, resid, w1, s1, b1, 2, w2, s2, b2, 2, m, True)
plain = nn.testEvaluatePointwisePair(batch, L, L, "bf16", x, resid, w1, s1, b1, 2, w2, s2, b2, 2, m, False)
want = ref.seam(x, resid, w1, s1, b1, 2, w2, s2, b2, 2, m, "bf16")
import hashlib
print("RESULT " + json.dumps({"same": [bool(np.array_equal(f, p)) for f, p in zip(fused, plain)],
                              "err": [float(np.abs(f - w).max()) for f, w in zip(fused, want)], "scale": [float(np.abs(w).max()) for w in want],
                              "off_board_zero": bool((fused[2][m != 1.0] == 0).all()),
                              "digest": hashlib.sha1(b"".join(np.ascontiguousarray(f).tobytes() for f in fused)).hexdigest()}))
""" % (REPO, os.path.join(REPO, "tests"))


def test_bench_small_batch_block_emulated(emu_lib):
    """bench.py's small_batch_rates() - the block that puts the small-batch regime into the driver's bench line - on the CPU
    emulation of the library: same call sequence as on the GPU (kmx_eval from host rows at batch 1 / 8 on a handle made for more),
    and its promise not to raise: a handle that was closed gives an error record, not an exception."""
    code = r"""
import sys, json, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from katago_amd import capi
capi._lib = capi.load_library(path=sys.argv[1])
from katago_amd import nninterface as nn, modelgen
import bench
nn.globalInitialize()
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "kmx_emu_benchsmall.bin")
modelgen.write_model(p, "b2c32nbt", seed=9)
ctx = nn.createComputeContext([0], 19, 19, precision="bf16")
h = nn.createComputeHandle(ctx, nn.loadModelFile(p), 12)
sp, gl = bench.synthetic_rows(12, 1)
sym = (np.arange(12) %% 8).astype(np.int32); opt = np.zeros(12, np.float32)
ok = bench.small_batch_rates(nn, h, sp, gl, sym, opt, "bf16", sizes=(1, 8, 32), reps=2, warm=1)
h.close()
bad = bench.small_batch_rates(nn, h, sp, gl, sym, opt, "bf16", sizes=(1,), reps=1, warm=1)
print("RESULT " + json.dumps({"ok": ok, "bad": bad}))
""" % (REPO, os.path.join(REPO, "tests"))
    p = subprocess.run([sys.executable, "-c", code, emu_lib], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "RESULT " in p.stdout, (p.stdout + p.stderr)[-3000:]
    res = json.loads(p.stdout.split("RESULT ")[1])
    # 32 > 12 rows: skipped. (rows_per_s is an integer: an emulated pass on a loaded machine takes seconds and rounds to 0 - the times say it ran)
    assert sorted(res["ok"]["ms_per_pass"]) == ["1", "8"] and all(v > 0 for v in res["ok"]["ms_per_pass"].values()), res
    assert sorted(res["ok"]["rows_per_s"]) == ["1", "8"] and all(v >= 0 for v in res["ok"]["rows_per_s"].values()), res
    assert "small_batches_error" in res["bad"], res

"""Build-time guard for the hand-written asynchronous loads of the register-weights convolution shapes (conv_small_kernel.h REGW:
inline-asm global_load_dwordx4 / ds_read_b128 with hand-counted s_waitcnt). The compiler knows nothing about a load in flight: it may
copy, spill or re-use the destination registers between the load and its wait, and does so silently. tools/check_async_loads.py walks the
SHIPPED code objects (katago_amd/libkatamx.so, gfx950) with the hardware's in-order vmcnt / lgkmcnt queues and reports every instruction
that touches a register whose load is still in flight, and any scratch use of those kernels.

Round 6 found production self-play dying of a GPU exception (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION) exactly this way: LDS reads for
a chunk that does not exist were never waited for, the compiler gave their registers to a row pointer of the epilogue, and the data landed
there after the pointer was computed. No GPU needed: hipcc cross-compiles, llvm-objdump disassembles."""
import os
import subprocess
import sys

from conftest import REPO

TOOL = os.path.join(REPO, "tools", "check_async_loads.py")


def test_no_register_of_a_load_in_flight_is_touched_in_the_shipped_library():
    from katago_amd import build as kbuild

    lib = kbuild.build(verbose=False)
    p = subprocess.run([sys.executable, TOOL, lib], capture_output=True, text=True, timeout=600)
    kernels = [l for l in p.stdout.splitlines() if l.startswith("_Z")]
    # the four register-weights shapes (cfg 125 / 126 / 127 / 128) in fp16 and bf16, and the stamped cfg 127 of conv_bench.hip
    assert sum("convSmallKernel" in l for l in kernels) >= 9, p.stdout[-2000:]
    assert p.returncode == 0 and all(l.rstrip().endswith(": 0 instruction(s) touch a register whose load is in flight") for l in kernels), p.stdout[-4000:]


def test_the_checker_has_teeth(tmp_path):
    """The shape of round 5's bug, and of a spilled fragment: both must be reported."""
    bad = tmp_path / "bad.s"
    bad.write_text("""
_Z9lateLdsReadv:
	global_load_dwordx4 v[10:13], v0, s[4:5]
	s_waitcnt vmcnt(0)
	ds_read_b128 v[40:43], v36
	v_mfma_f32_32x32x16_f16 v[18:33], v[10:13], v[82:85], v[18:33]
	v_cndmask_b32_e32 v41, v37, v44, vcc
	v_cndmask_b32_e32 v40, v36, v45, vcc
	global_load_dwordx4 v[40:43], v[40:41], off
	s_waitcnt vmcnt(0)
	s_endpgm
.Lfunc_end0:
_Z12copiedInFlightv:
	global_load_dwordx4 v[10:13], v0, s[4:5]
	v_mov_b32_e32 v50, v10
	s_waitcnt vmcnt(0)
	v_mfma_f32_32x32x16_f16 v[18:33], v[10:13], v[82:85], v[18:33]
	s_endpgm
.Lfunc_end1:
_Z4finev:
	global_load_dwordx4 v[10:13], v0, s[4:5]
	ds_read_b128 v[40:43], v36
	s_waitcnt vmcnt(0)
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_32x32x16_f16 v[18:33], v[10:13], v[40:43], v[18:33]
	v_mov_b32_e32 v40, v1
	s_endpgm
.Lfunc_end2:
""")
    p = subprocess.run([sys.executable, TOOL, str(bad)], capture_output=True, text=True, timeout=60)
    out = p.stdout
    assert p.returncode == 1, out
    assert "_Z9lateLdsReadv: 3 instruction(s)" in out and "_Z12copiedInFlightv: 1 instruction(s)" in out and "_Z4finev: 0 instruction(s)" in out, out

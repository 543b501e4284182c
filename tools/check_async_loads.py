#!/usr/bin/env python3
"""Build-time guard for the hand-written asynchronous loads (the GPU exception of round 5's driver run).

The register-weights shapes (conv_small_kernel.h REGW) request their weight fragments with an inline-asm `global_load_dwordx4` and wait for
them many instructions later with an inline-asm `s_waitcnt vmcnt(N)`. The compiler believes the destination registers hold their value as
soon as the load statement has "executed": nothing stops it from copying them, or from using them for something else, while the load is
still in flight - the copy reads stale bits, and a register that was given another job (an address, say) is overwritten when the load
lands. Both are silent at compile time.

This script reads the device assembly of a translation unit (hipcc --cuda-device-only -S) and walks every kernel that contains such
loads, instruction by instruction, with the hardware's in-order vmcnt queue: a register is IN FLIGHT from the load that names it as its
destination until an s_waitcnt vmcnt(N) retires that load - and, since round 6, the same for LDS reads (ds_read_*) and lgkmcnt: the
hand-written ds_read_b128 of the same kernels read fragments several k halves ahead, and the reads issued for a "next chunk" that does not
exist were never waited for: their destination registers are dead to the compiler, which gave them to the epilogue's row addresses - and
the LDS data that landed there later made production self-play die of HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.
Any instruction that reads or writes a register in flight is reported.
Loops are walked twice (state carried over the back edge). Exit status 1 if anything is reported.

    python tools/check_async_loads.py /tmp/conv_mfma.s [kernel-name-substring]      # hipcc --cuda-device-only -S output
    python tools/check_async_loads.py katago_amd/libkatamx.so [kernel-name-substring] # the shipped library: its gfx950 code objects are
                                                                                      # unbundled and disassembled (llvm-objdump), seconds
Also reports, per kernel with such loads, scratch use (a spilled fragment register is a copy at the wrong time by construction).
"""
import os
import subprocess
import tempfile
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


VM_LOAD = ("global_load", "buffer_load", "flat_load", "scratch_load")
VM_STORE = ("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic")


def parse(body):
    prog = []
    for l in body:
        s = l.split(";")[0].split("//")[0].strip()
        if re.match(r"^<L\d+>:$", s):  # llvm-objdump --symbolize-operands
            prog.append(("label", s[:-1]))
            continue
        if not s or s.startswith((".", "/")):
            if re.match(r"^\.LBB\d+_\d+:", s):
                prog.append(("label", s[:-1]))
            continue
        if s.endswith(":"):
            prog.append(("label", s[:-1]))
            continue
        parts = s.split(None, 1)
        prog.append(("ins", parts[0], parts[1] if len(parts) > 1 else "", s))
    return prog


def check(name, body, verbose=False):
    prog = parse(body)
    labels = {p[1]: i for i, p in enumerate(prog) if p[0] == "label"}
    queue = []  # in-flight VMEM operations in issue order: (frozenset(dest regs), text)
    lgkm = []   # in-flight LDS / scalar-memory operations in issue order (LDS operations of a wave return in order)
    problems = []
    taken = set()
    i = 0
    steps = 0
    while i < len(prog) and steps < 4 * len(prog):
        steps += 1
        p = prog[i]
        if p[0] == "label":
            i += 1
            continue
        _, op, rest, text = p
        ops = [o.strip() for o in rest.split(",")] if rest else []
        inflight = set()
        for d, _t in queue + lgkm:
            inflight |= d
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                while len(lgkm) > n:
                    lgkm.pop(0)
            if not re.search(r"cnt\(", rest):  # raw immediate: treat as a full wait
                queue.clear()
                lgkm.clear()
            i += 1
            continue
        if op in ("s_endpgm",):
            # the hardware waits for everything outstanding before it ends the wave; what matters is what ran before
            queue.clear()
            lgkm.clear()
            i += 1
            continue
        if op.startswith("ds_"):
            touched_all = set(regs(rest))
            hit = touched_all & inflight
            if hit:
                problems.append((text, sorted(hit)))
            is_read = op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in op
            lgkm.append((frozenset(regs(ops[0])) if is_read and ops else frozenset(), text))
            i += 1
            continue
        if op.startswith(("s_load", "s_buffer_load", "s_sendmsg", "s_memtime", "s_memrealtime", "s_dcache")):
            lgkm.append((frozenset(), text))
            i += 1
            continue
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = ops[-1] if ops else ""
            if tgt in labels and labels[tgt] < i and i not in taken:
                taken.add(i)
                i = labels[tgt]
                continue
            i += 1
            continue
        touched = set(regs(rest))
        is_lds_dma = op.startswith(VM_LOAD) and (" lds" in (" " + rest) or "_lds_" in op)  # no register destination: the operands are its address
        if op.startswith(VM_LOAD) and not is_lds_dma:
            dest = frozenset(regs(ops[0]))
            src = set()
            for o in ops[1:]:
                src |= set(regs(o))
            hit = (src | dest) & inflight
            if hit:
                problems.append((text, sorted(hit)))
            queue.append((dest, text))
        elif op.startswith(VM_LOAD) or op.startswith(VM_STORE):
            hit = touched & inflight
            if hit:
                problems.append((text, sorted(hit)))
            queue.append((frozenset(), text))
        else:
            hit = touched & inflight
            if hit:
                problems.append((text, sorted(hit)))
        i += 1
    return problems


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def disassemble_library(so_path):
    """The gfx950 code objects of a shared library (its .hip_fatbin section: one offload bundle per translation unit), disassembled."""
    text = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path], check=True)
        blob = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        for k, o in enumerate(offs):
            piece = os.path.join(d, "b%d.bin" % k)
            with open(piece, "wb") as f:
                f.write(blob[o:offs[k + 1] if k + 1 < len(offs) else len(blob)])
            co = os.path.join(d, "co%d.o" % k)
            subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
            r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", "--symbolize-operands", "--no-show-raw-insn",
                                "--no-leading-addr", co], check=True, capture_output=True, text=True)
            text.append(r.stdout)
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            text.append("\n".join("; NOTE " + l for l in notes.split("\n")))
    return "\n".join(text)


def scratch_of(notes_lines, name):
    """private_segment_fixed_size / vgpr_spill_count of a kernel from the code object's metadata (the '; NOTE' lines)."""
    t = "\n".join(notes_lines)
    for block in re.split(r"\n; NOTE\s+- \.agpr_count", t)[1:]:
        if re.search(r"\.name:\s+" + re.escape(name) + r"\b", block):
            g = lambda k: int((re.findall(k + r":\s+(\d+)", block) or ["0"])[0])
            return g(r"\.private_segment_fixed_size"), g(r"\.vgpr_spill_count")
    return None


def main():
    path = sys.argv[1]
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    from_library = path.endswith(".so")
    lines = (disassemble_library(path) if from_library else open(path).read()).split("\n")
    notes = [l for l in lines if l.startswith("; NOTE")]
    if from_library:
        starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^<(_Z\w+)>:\s*$", l)] if m]
    else:
        starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s*(;.*)?$", l)] if m]
    bad = 0
    for k, (i, name) in enumerate(starts):
        if only and only not in name:
            continue
        j = i + 1
        if from_library:
            j = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        else:
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
        body = lines[i + 1:j]
        if not any(("global_load_dwordx4" in l and re.search(r", s\[\d+:\d+\]", l)) for l in body) and not os.environ.get("KMX_CHECK_ALL_KERNELS"):
            continue
        problems = check(name, body)
        sc = scratch_of(notes, name) if notes else None
        if sc is not None and (sc[0] or sc[1]):
            print("%s: %d bytes of scratch per lane, %d spilled registers" % (name, sc[0], sc[1]))
            bad += 1
        print("%s: %d instruction(s) touch a register whose load is in flight" % (name, len(problems)))
        for t, h in problems[:12]:
            print("    %-70s  in flight: v%s" % (t, ",v".join(map(str, h[:8]))))
        bad += len(problems)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

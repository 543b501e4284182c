#!/usr/bin/env python3
"""Build-time guard for the hand-written asynchronous loads (the GPU exception of round 5's driver run).

The register-weights shapes (conv_small_kernel.h REGW) request their weight fragments with an inline-asm `global_load_dwordx4` and wait for
them many instructions later with an inline-asm `s_waitcnt vmcnt(N)`. The compiler believes the destination registers hold their value as
soon as the load statement has "executed": nothing stops it from copying them, or from using them for something else, while the load is
still in flight - the copy reads stale bits, and a register that was given another job (an address, say) is overwritten when the load
lands. Both are silent at compile time.

This script reads the device assembly of a translation unit (hipcc --cuda-device-only -S) and walks every kernel that contains such
loads, instruction by instruction, with the hardware's in-order vmcnt queue: a register is IN FLIGHT from the load that names it as its
destination until an s_waitcnt vmcnt(N) retires that load. Any instruction that reads or writes a register in flight is reported.
Loops are walked twice (state carried over the back edge). Exit status 1 if anything is reported.

    python tools/check_async_loads.py /tmp/conv_mfma.s [kernel-name-substring]
"""
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
        s = l.split(";")[0].strip()
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
        for d, _t in queue:
            inflight |= d
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
            elif re.fullmatch(r"\s*\d+\s*|0x[0-9a-f]+", rest):  # raw immediate: treat as a full wait
                queue.clear()
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
        if op in ("s_endpgm",):
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


def main():
    path = sys.argv[1]
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s*(;.*)?$", l)] if m]
    bad = 0
    for i, name in starts:
        if only and only not in name:
            continue
        j = i
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i + 1:j]
        if not any(("global_load_dwordx4" in l and re.search(r", s\[\d+:\d+\]", l)) for l in body):
            continue
        problems = check(name, body)
        print("%s: %d instruction(s) touch a register whose load is in flight" % (name, len(problems)))
        for t, h in problems[:12]:
            print("    %-70s  in flight: v%s" % (t, ",v".join(map(str, h[:8]))))
        bad += len(problems)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

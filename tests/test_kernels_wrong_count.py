"""The emulator's teeth (tests/fakehip/README.md, "Asynchronous completion"): the kernels with a wait count ONE request too generous pass
with immediate copies and fail with the latest legal completion. A file of its own (its build of the mutated kernels and its runs are the
longest single item of the CPU suite); helpers live in test_engine_emulated.py and test_kernels_latest_completion.py."""
import json
import os
import sys

from test_engine_emulated import PW2_CODE, build_emu_full, conv_only, run_parallel
from test_kernels_latest_completion import CW12_CODE


def test_latest_completion_catches_a_wrong_count(tmp_path):
    """The emulator's teeth: the same kernels with every top-of-step wait ONE request too generous (the padded shapes' VMCNT
    constant, the loader waves' computed count). With immediate copies nothing shows; with the latest legal completion the slab a
    step needs has not landed and the answers are wrong."""
    lib = build_emu_full(str(tmp_path), conv_mutations=[
        (r"static constexpr int VMCNT = SPREAD \? PPS \+ \(D - 2\) \* \(NPW \+ PPS\) : \(D - 2\) \* \(NPW \+ NPA\);",
         "static constexpr int VMCNT = 1 + (SPREAD ? PPS + (D - 2) * (NPW + PPS) : (D - 2) * (NPW + NPA));", 1),
        (r"        waitVmSel\(n\);\n      \}\n      return;", "        waitVmSel(n + 1);\n      }\n      return;", 1),
        # the weight waves of the role-split shapes (the twelve-wave small-batch shape among them)
        (r"static constexpr int VMCNT_W = \(D - 2\) \* NPW,", "static constexpr int VMCNT_W = 1 + (D - 2) * NPW,", 1),
    ], small_mutations=[
        # the fetching waves of the small-batch shape let one request more stay in flight than slab s + 1 allows
        (r"if\(slabWave\) convk::waitVmSel\(G::vmAt\(t\)\);", "if(slabWave) convk::waitVmSel(G::vmAt(t) + 1);", 1),
        # the shape with its weights in registers: a multiplying wave lets one request more stay in flight than the 16 younger fragments
        # (the fragment it is about to multiply may then not have landed), a fetching wave lets the image a barrier publishes stay in flight
        (r'waitFragSel\(wf\[slot\]\[wn\], younger \* WN\);', 'waitFragSel(wf[slot][wn], younger * WN + 1);', 1),
        (r'        waitVm<0>\(\);  // image chunk \+ 1', '        waitVm<NPA>();  // image chunk + 1', 1),
    ], pw2_mutations=[
        # the persistent seam kernel: phase 2 of a part lets one request more stay in flight than its loads and stores account for
        (r"waitVmSel\(G::nR\(q\) \+ 2\);", "waitVmSel(G::nR(q) + 2 + 1);", 1),
    ], pw3_mutations=[
        # the seam kernel with resident weights: the wait for the next tile's X lets one request more stay in flight than a work-group's
        # first tile issues after it - the last chunk of that X may then land after barrier BA1 has published it
        (r"static constexpr int VM_AFTER_X = N_RS \+ N_RS \+ N_S1S \+ N_S1S;", "static constexpr int VM_AFTER_X = N_RS + N_RS + N_S1S + N_S1S + 1;", 1),
    ])
    runs = run_parallel([conv_only(lib, {"KMX_EMU_LATE_DMA": late}) for late in ("0", "1")])
    (rc0, so0, se0), (rc1, so1, se1) = runs
    assert rc0 == 0 and rc1 == 0, (so0 + se0 + so1 + se1)[-3000:]
    early, late = json.loads(so0.split("RESULT ")[1]), json.loads(so1.split("RESULT ")[1])
    for k, v in early.items():
        assert v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5, ("immediate copies hide the defect", k, v)
    wrong = [k for k, v in late.items() if not v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5]
    print("late completion, one request too generous:", late)
    assert "conv3_64_32" in wrong and "conv3_96_192" in wrong, late  # the padded 4-wave shape and the 8-wave loader shape
    # the small-batch shape with its fetching waves' wait one request too generous
    runs = run_parallel([([sys.executable, "-c", CW12_CODE, lib], dict(os.environ, KMX_CONV_TUNE="loaders=1,regw=0", KMX_EMU_LATE_DMA=late_)) for late_ in ("0", "2")])
    (rc0, so0, se0), (rc1, so1, se1) = runs
    assert rc0 == 0 and rc1 == 0, (so0 + se0 + so1 + se1)[-3000:]
    c0, c1 = json.loads(so0.split("RESULT ")[1]), json.loads(so1.split("RESULT ")[1])
    ok = lambda v: v[0] <= 2.0 ** -7 * max(1.0, v[1]) * 1.5  # noqa: E731
    print("fetching-waves shape, one request too generous: latest completion", {k: v[:2] for k, v in c1.items()})
    # (the fetching waves run ahead of the multiplying ones between two barriers, so under emulation a late copy may still land before a
    # small case reads it: the defect must show in at least one case - the 19x19 one in practice - and never with immediate copies)
    assert all(ok(v) for v in c0.values()) and not all(ok(v) for v in c1.values()), (c0, c1)
    # ... and with its weights in registers: the fragment wait (mode 1: a stale fragment is multiplied) and, with cell tiles split, the image wait
    runs = run_parallel([([sys.executable, "-c", CW12_CODE, lib], dict(os.environ, KMX_CONV_TUNE="loaders=1,loaders_split=%s,regw=%s" % (sp_, rw_), KMX_EMU_LATE_DMA=late_))
                         for sp_, late_, rw_ in (("0", "0", "1"), ("0", "1", "1"), ("1", "2", "3"), ("0", "1", "2,loaders_max_wgs=0"))])
    for rc, so, se in runs:
        assert rc == 0, (so + se)[-3000:]
    c0, c1, c2, c3 = (json.loads(so.split("RESULT ")[1]) for rc, so, se in runs)
    print("weights in registers, one request too generous: latest completion", {k: v[:2] for k, v in c1.items()}, {k: v[:2] for k, v in c2.items()},
          {k: v[:2] for k, v in c3.items()})
    assert all(ok(v) for v in c0.values()) and not any(ok(v) for v in c1.values()) and not all(ok(v) for v in c2.values()), (c0, c1, c2)
    assert not all(ok(v) for v in c3.values()), c3  # the 64-channel shape where it is taken (an even number of channel tiles)
    # the seam kernel with its defect: right with immediate copies, wrong when a W2 slab may land as late as the count allows
    for kern in ("2", "3"):  # round 3's persistent kernel, round 5's with resident weights
        runs = run_parallel([([sys.executable, "-c", PW2_CODE, lib], dict(os.environ, KMX_PW_KERNEL=kern, KMX_PW_GRID="1", KMX_EMU_LATE_DMA=late_))
                             for late_ in ("0", "1")])
        (rc0, so0, se0), (rc1, so1, se1) = runs
        assert rc0 == 0 and rc1 == 0, (so0 + se0 + so1 + se1)[-3000:]
        r0, r1 = json.loads(so0.split("RESULT ")[1]), json.loads(so1.split("RESULT ")[1])
        print("seam kernel %s, one request too generous: immediate" % kern, r0["same"], "latest", r1["same"])
        assert all(r0["same"]) and not all(r1["same"]), (kern, r0, r1)

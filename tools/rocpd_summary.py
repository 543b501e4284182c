#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the text files kept under profiles/.
    python tools/rocpd_summary.py gpurun_out/prof profiles/r01
Kernel-trace runs -> <name>_kernel_stats.csv (per-kernel calls / total / avg / min / max ns, like --stats);
PMC runs          -> <name>_pmc.csv (counter sums per kernel and per-dispatch averages)."""
import csv
import glob
import os
import re
import sqlite3
import sys


def short(name):
    # template arguments: traits, kernel size, WN, WNW, ring depth, ablation mask and (since round 3) the number of cell waves
    m = re.search(r"convMfmaKernel<kmx::(Traits\w+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?>", name)
    if not m:
        m = re.search(r"convMfmaKernelINS_\d+(Traits\w+?)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E(?:Li(\d+)E)?", name)  # mangled form
    if m:
        g = m.groups()
        return "convMfmaKernel<%s,KS=%s,WN=%s,WNW=%s,D=%s,ABL=%s%s>" % (g[:6] + ("" if g[6] in (None, "4") else ",CW=" + g[6],))
    return re.sub(r"\(.*", "", name)[:80]


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for db in sorted(glob.glob(os.path.join(src, "*", "*.db"))):
        tag = os.path.basename(os.path.dirname(db))
        con = sqlite3.connect(db)
        tabs = {r[0].split("_0000")[0]: r[0] for r in con.execute("select name from sqlite_master where type='table'")}
        kd, ks = tabs["rocpd_kernel_dispatch"], tabs["rocpd_info_kernel_symbol"]
        rows = con.execute(
            f"select k.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
            f"max(k.arch_vgpr_count), max(k.accum_vgpr_count), max(k.sgpr_count), max(d.group_segment_size) "
            f"from '{kd}' d join '{ks}' k on d.kernel_id = k.id group by k.kernel_name order by 3 desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR", "SGPR", "LDS_bytes"])
            for r in rows:
                w.writerow([short(r[0]), r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / total), r[4], r[5], r[6], r[7], r[8], r[9]])
        pe, pi = tabs.get("rocpd_pmc_event"), tabs.get("rocpd_info_pmc")
        n = con.execute(f"select count(*) from '{pe}'").fetchone()[0]
        if n:
            q = (f"select k.kernel_name, p.name, count(distinct d.id), sum(e.value) from '{pe}' e join '{pi}' p on e.pmc_id = p.id "
                 f"join '{kd}' d on e.event_id = d.event_id join '{ks}' k on d.kernel_id = k.id group by k.kernel_name, p.name")
            with open(os.path.join(dst, tag + "_pmc.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Kernel", "Counter", "Dispatches", "Sum", "PerDispatch"])
                for kn, cn, nd, sv in con.execute(q):
                    w.writerow([short(kn), cn, nd, "%.6g" % sv, "%.6g" % (sv / max(nd, 1))])
        print("summarised", tag, "kernels", len(rows), "pmc rows", n)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

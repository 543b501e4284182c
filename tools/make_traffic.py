#!/usr/bin/env python3
"""traffic.json for bench.py's roofline.traffic: HBM bytes per launch of the dominant kernel, from the rocprofv3 PMC
passes over the bench command (tools/profile_gpu.sh). FETCH_SIZE and WRITE_SIZE are reported in KiB and summed over a
kernel's dispatches; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes for wide streaming reads, so it is
doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is (uncalibrated there).
    python tools/make_traffic.py <summary dir>"""
import csv
import json
import os
import sys


def per_dispatch(path, counter, key):
    tot = disp = 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter"] == counter and key in r["Kernel"]:
            tot += float(r["Sum"])
            disp += float(r["Dispatches"])
    return (tot / disp if disp else None), int(disp)


def main(d):
    fetch, nf = per_dispatch(os.path.join(d, "benchpmc_FETCH_SIZE_pmc.csv"), "FETCH_SIZE", "KS=3")
    write, nw = per_dispatch(os.path.join(d, "benchpmc_WRITE_SIZE_pmc.csv"), "WRITE_SIZE", "KS=3")
    if fetch is None or write is None:
        raise SystemExit("no convMfmaKernel<KS=3> dispatches in the PMC summaries")
    out = {"kernel": "conv3x3", "model": "b18c384nbt", "batch": 256, "dtype": "bf16",
           "hbm_bytes_per_launch": round((2.0 * fetch + write) * 1024.0),
           "fetch_size_kib_per_launch_raw": round(fetch, 1), "write_size_kib_per_launch_raw": round(write, 1),
           "dispatches_counted": [nf, nw],
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 2`, all "
                     "convMfmaKernel<KS=3> dispatches; FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE x1"}
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])

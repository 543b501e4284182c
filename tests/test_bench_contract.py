"""The one JSON line bench.py prints (the driver's contract): the committed sample line profiles/r01_final/bench.json is checked
for every required key, type and relation — so that a change of bench.py's output that breaks the contract also has to
change the committed evidence. No GPU needed."""
import json
import os

from conftest import REPO


def test_committed_bench_line_follows_the_contract():
    d = json.load(open(os.path.join(REPO, "profiles", "r01_final", "bench.json")))
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert d["metric"] == "nn_evals_per_s" and d["unit"] == "evals/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] in ("bf16", "fp16")
    assert "workload" in d["config"] and "b18c384nbt" in d["config"]["workload"] and "model" not in d["config"]
    # value = rows of all ranks / time: consistent with ms_per_step at batch 256 per GPU
    assert abs(d["value"] - d["n_gpus"] * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic flops per launch / average launch duration
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) / r["achieved"] < 0.01
    assert r["traffic"] is None or r["traffic"]["hbm_bytes_per_launch"] > 0.5 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    # the rocprofv3 summary of the same command agrees with the live hipEvent average (same call, same box)
    import csv
    rows = list(csv.DictReader(open(os.path.join(REPO, "profiles", "r01_final", "bench_trace_kernel_stats.csv"))))
    conv3 = [x for x in rows if "KS=3" in x["Name"]]
    assert conv3 and abs(float(conv3[0]["AverageNs"]) * 1e-6 - r["avg_launch_ms"]) / r["avg_launch_ms"] < 0.05

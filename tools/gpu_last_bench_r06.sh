#!/bin/bash
# Round 6, last call: what the driver does at round end after the suite - smoke(), then its bench command - at the round's final commit, timed.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/last_r06; mkdir -p $OUT
t0=$SECONDS
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $? after $((SECONDS - t0)) s" | tee $OUT/bench_seconds.txt
t0=$SECONDS
timeout 560 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_last.json 2> $OUT/bench_last.err; echo "bench rc $? after $((SECONDS - t0)) s" | tee -a $OUT/bench_seconds.txt
tail -3 $OUT/smoke.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/last_r06/bench_last.json").read().strip().splitlines()[-1])
keep = {k: d[k] for k in d if k in ("value", "ms_per_step", "dtype") or k.startswith("selfplay") and not isinstance(d[k], dict) or k.startswith("reference_benchmark") and k.endswith("per_s") or k == "host_rows_through_batcher_per_s"}
keep["roofline.frac"] = d["roofline"]["frac"]; keep["frac_of_sustained"] = d["roofline"].get("frac_of_sustained"); keep["seam.frac"] = d["roofline_seam"]["frac"]
keep["traffic"] = (d["roofline"]["traffic"] or {}).get("hbm_bytes_per_launch"); keep["box"] = d["box"]; keep["small_batches"] = d.get("small_batches", {}).get("ms_per_pass")
keep["selfplay_error"] = d.get("selfplay_error")
print(json.dumps(keep, indent=1))
PY

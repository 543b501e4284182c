#!/bin/bash
# graphs + K-way split scan, full GPU suite, transformer nets at fp16/bf16, reference benchmark with 1/2 server threads
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/call3
mkdir -p "$OUT"
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" > "$OUT/smi_before.txt"
python -m pytest tests/test_gpu_bench_command.py -m gpu -x -q -p no:cacheprovider -k "graph or async or stream" 2>&1 | tail -15 > "$OUT/graph_tests.log"; tail -4 "$OUT/graph_tests.log"
b() { # label env... -- args
  local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local v=$(env "${envs[@]}" timeout 300 python3 bench.py --no-cpu-baseline --no-profile "$@" 2>>"$OUT/scan.err" | grep -o '"value": [0-9.]*, [^,]*, [^,]*, [^,]*, [^,]*, "ms_per_step": [0-9.]*' | sed 's/"n_gpus.*"ms_per_step"/ms/')
  echo "$name | $v" | tee -a "$OUT/scan.txt"
}
for g in 0 1; do
  b "batch256 ways1 graphs$g" KMX_GRAPHS=$g KMX_SPLIT_MIN=0 -- --steps 50 --warmup 5
  for w in 2 3 4; do b "batch256 ways$w graphs$g" KMX_GRAPHS=$g KMX_SPLIT_WAYS=$w -- --steps 50 --warmup 5; done
done
b "batch256 ways2 graphs0 stagger20" KMX_GRAPHS=0 KMX_SPLIT_STAGGER=20 -- --steps 50 --warmup 5
b "batch256 ways2 graphs0 stagger60" KMX_GRAPHS=0 KMX_SPLIT_STAGGER=60 -- --steps 50 --warmup 5
for g in 0 1; do for n in 1 8 32 64 128; do b "batch$n graphs$g" KMX_GRAPHS=$g -- --batch $n --steps 50 --warmup 5; done; done
b "batch512 ways2 graphs1" KMX_GRAPHS=1 -- --batch 512 --steps 20 --warmup 3
b "batch512 ways4 graphs1" KMX_GRAPHS=1 KMX_SPLIT_WAYS=4 -- --batch 512 --steps 20 --warmup 3
# full suite (transformer tests stay gated)
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 > "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
# transformer nets: margins per precision
export KMX_EXPERIMENTAL_TRANSFORMER=1
python - <<'PY' 2>&1 | tee "$OUT/tf_margins.txt"
import os, re, subprocess, sys, tempfile, pathlib
sys.path.insert(0, "tests")
import test_gpu_transformer as t
from conftest import ref_binary
for net in t.TF_NETS:
    tmp = pathlib.Path(tempfile.mkdtemp())
    args = t._reference_file(tmp, net)
    for prec in ("fp16", "bf16"):
        r = subprocess.run([ref_binary("katago_hip")] + args + ["-override-config", "katamxPrecision=" + prec], capture_output=True, text=True, timeout=900, cwd=str(tmp))
        out = r.stdout + r.stderr
        print(net, prec, "rc", r.returncode, {k: float(v) for k, v in re.findall(t._MARGIN, out)})
        for l in out.splitlines():
            if "current cfg" in l and ("winrateError" in l or "topPolicyDelta" in l or "policyKLDiv" in l or "scoreError" in l): print("   ", l.strip()[:200])
PY
unset KMX_EXPERIMENTAL_TRANSFORMER
# reference benchmark, 1 vs 2 server threads on the same GPU
python - <<'PY' 2>&1 | tee "$OUT/ref_benchmark_threads.txt"
import os, re, subprocess, sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_reference_harness as h
from katago_amd import modelgen
tmp = tempfile.mkdtemp()
model = os.path.join(tmp, "b18.bin.gz"); modelgen.write_model(model, "b18c384nbt", seed=7)
for threads in (1, 2, 3):
    cfg = os.path.join(tmp, "bench%d.cfg" % threads)
    open(cfg, "w").write(h.BENCH_CFG + "numNNServerThreadsPerModel = %d\n" % threads)
    rc, out = h.run("benchmark", "-model", model, "-config", cfg, "-v", "1600", "-t", "256", "-boardsize", "19", "-n", "3", timeout=900)
    for l in out.replace("\r", "\n").splitlines():
        if "nnEvals/s" in l: print("serverThreads", threads, l.strip())
PY
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" > "$OUT/smi_after.txt"

"""-m gpu: the reference's OWN test commands running on the HIP backend (integration/_build/katago_hip = unmodified reference host
code with this repo's NNEvaluator + featuriser in place of neuralnet/nneval.cpp, + integration/katamxbackend.cpp + libkatamx.so:
the default binding build since round 4; katago_hip_refeval keeps the reference's own evaluator)."""
import os
import re
import subprocess

import pytest

from conftest import REPO, ref_binary

pytestmark = pytest.mark.gpu
G170 = os.path.join(REPO, "oracle", "_ref", "models", "g170-b6c96-s175395328-d26788732.bin.gz")


def run(*args, timeout=600, binary="katago_hip"):
    b = ref_binary(binary)
    r = subprocess.run([b] + list(args), capture_output=True, text=True, timeout=timeout, cwd=os.path.dirname(b))
    return r.returncode, r.stdout + r.stderr


def test_nn_layer_known_answers_on_hip():
    """cpp/tests/testnn.cpp with its fp16 tolerance 0.03*max(|x|,3) (:8-15) and, since round 5, its fp32 variants at their own (the
    device serves useFP16 = false)."""
    rc, out = run("runnnlayertests")
    assert rc == 0, out[-3000:]
    assert "Test failed" not in out, out[-3000:]
    m = re.search(r"Tested (\d+) configurations", out)
    assert m and int(m.group(1)) == 28, out[-500:]  # 7 layer tests x {NHWC, NCHW} x {fp16, fp32}


def test_tiny_model_on_hip(tmp_path):
    """cpp/tests/tinymodel.cpp through the reference NNEvaluator (featurisation, batching threads, post-processing)."""
    rc, out = run("runtinynntests", str(tmp_path), "1.0")
    assert rc == 0 and "Tiny net sanity check complete" in out, out[-3000:]
    assert "katamx (HIP/gfx950)" in out and "CPU oracle" not in out


def test_real_net_tiny_board_golden_on_hip():
    """g170-b6c96 (a real trained net, 5x5 stem, v8) vs cpp/tests/results/runNNOnTinyBoardTest.txt at 16-bit tolerance."""
    if not os.path.exists(G170):
        pytest.skip("g170 net not packaged")
    rc, out = run("runnnontinyboardtest", G170, "true", "true", "3", "true")
    assert rc == 0, out[-3000:]
    gold = open(os.path.join(REPO, "tests", "golden", "ref_runNNOnTinyBoardTest.txt")).read()

    def parse(t):
        lines = [l for l in t.splitlines() if l.strip() and not l.startswith(":")]
        scal = {}
        for l in lines:
            m = re.match(r"(Win|Loss|NoResult|ScoreMean|ScoreMeanSq|Lead)\s+(-?[\d.]+)", l)
            if m:
                scal[m.group(1)] = float(m.group(2))
        k = next(i for i, l in enumerate(lines) if l.startswith("Pass"))
        grid = lambda rows: [float(x) if x != "-" else None for l in rows for x in l.split()]
        return scal, float(lines[k].split()[1]), grid(lines[k + 1:k + 6]), grid(lines[k + 6:k + 11])

    sa, passa, pola, owna = parse(out)
    sb, passb, polb, ownb = parse(gold)
    # 16-bit device arithmetic against an fp32 golden: value within 1.5 %, score/lead within 0.5 point,
    # policy within 1.5 % (printed per-mille), ownership within 0.025 (printed per-mille)
    assert abs(sa["Win"] - sb["Win"]) <= 1.5 and abs(sa["Loss"] - sb["Loss"]) <= 1.5 and abs(sa["NoResult"] - sb["NoResult"]) <= 0.5
    assert abs(sa["ScoreMean"] - sb["ScoreMean"]) <= 0.5 and abs(sa["Lead"] - sb["Lead"]) <= 0.5
    assert abs(sa["ScoreMeanSq"] - sb["ScoreMeanSq"]) <= 0.05 * sb["ScoreMeanSq"]
    assert abs(passa - passb) <= 15 and len(pola) == len(polb) == 25 and len(owna) == len(ownb) == 25
    for u, v in zip(pola, polb):
        assert (u is None) == (v is None) and (u is None or abs(u - v) <= 15), (u, v)  # same legality pattern
    for u, v in zip(owna, ownb):
        assert abs(u - v) <= 25, (u, v)


BENCH_CFG = """logDir = gtp_logs
logAllGTPCommunication = false
logSearchInfo = false
logToStderr = false
rules = tromp-taylor
allowResignation = false
maxVisits = 200
numSearchThreads = 8
nnCacheSizePowerOfTwo = 18
nnMutexPoolSizePowerOfTwo = 14
nnRandomize = true
ponderingEnabled = false
lagBuffer = 1.0
searchFactorAfterOnePass = 0.5
searchFactorAfterTwoPass = 0.25
searchFactorWhenWinning = 0.4
searchFactorWhenWinningThreshold = 0.95
"""


def test_reference_benchmark_command_on_hip(tmp_path):
    """The reference's own definition of the headline metric: `katago benchmark` (command/benchmark.cpp; nnEvals/s =
    NNEvaluator::numRowsProcessed / search seconds, playutils.cpp:843,991-1000) — BASELINE configs[0] (g170-b6c96, 9x9,
    200 visits) with the HIP backend in place of Eigen. Search, featurisation, batching and cache are the reference's."""
    if not os.path.exists(G170):
        pytest.skip("g170 net not packaged")
    cfg = tmp_path / "bench.cfg"
    cfg.write_text(BENCH_CFG)
    rc, out = run("benchmark", "-model", G170, "-config", str(cfg), "-v", "200", "-t", "8,32", "-boardsize", "9", "-n", "4")
    assert rc == 0, out[-3000:]
    assert "katamx (HIP/gfx950)" in out and "CPU oracle" not in out
    rates = [float(x) for x in re.findall(r"nnEvals/s = ([\d.]+)", out)]
    assert len(rates) >= 2 and all(r > 0 for r in rates), out[-2000:]
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "reference_benchmark_b6c96_9x9.txt"), "w") as f:
            f.write("\n".join(l for l in out.replace("\r", "\n").splitlines() if "nnEvals/s" in l) + "\n")


def test_reference_benchmark_b18_19x19_on_hip(tmp_path):
    """BASELINE configs[1] through the reference's own `benchmark`: b18c384nbt (random weights), 19x19, as many search
    threads as the reference needs to fill batches (its batch size = #threads, benchmark.cpp:206-217). The number is
    host-bound (one blocked OS thread per in-flight leaf) and recorded, not asserted."""
    from katago_amd import modelgen

    model = str(tmp_path / "b18.bin.gz")
    modelgen.write_model(model, "b18c384nbt", seed=7)
    cfg = tmp_path / "bench.cfg"
    cfg.write_text(BENCH_CFG)
    rc, out = run("benchmark", "-model", model, "-config", str(cfg), "-v", "1600", "-t", "64,256", "-boardsize", "19", "-n", "3",
                  timeout=900)
    assert rc == 0, out[-3000:]
    rates = [float(x) for x in re.findall(r"nnEvals/s = ([\d.]+)", out)]
    assert len(rates) >= 2 and all(r > 0 for r in rates), out[-2000:]
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "reference_benchmark_b18_19x19.txt"), "w") as f:
            f.write("\n".join(l for l in out.replace("\r", "\n").splitlines() if "nnEvals/s" in l) + "\n")


def test_oracle_agrees_with_reference_opencl_backend(tmp_path):
    """Pins the ORACLE against the reference's own GPU implementation of the path, on the MI355X: katago_opencl is the
    reference's OpenCL backend built from its sources (oracle/Makefile). katago_oracle, in the role of the Eigen build,
    writes `testgpuerror`'s reference file (CPU); katago_opencl then checks its fp32/fp16, batched/unbatched outputs against
    it with the reference's own cross-backend thresholds (tests/testnnevalcanary.cpp:573-829) — real trained net
    (g170-b6c96), the reference's own 9x9 position set. Exit code 0 = every statistic within its limit."""
    if not os.path.exists(G170):
        pytest.skip("g170 net not packaged")
    try:
        ocl = ref_binary("katago_opencl")
        orc = ref_binary("katago_oracle")
    except Exception:
        pytest.skip("reference OpenCL build not present")
    cfg = tmp_path / "bench.cfg"
    cfg.write_text(BENCH_CFG + "homeDataDir = %s\n" % (tmp_path / "home"))
    ref = str(tmp_path / "ref.txt")
    args = ["testgpuerror", "-model", G170, "-config", str(cfg), "-boardsize", "9", "-quick", "-reference-file", ref]
    r = subprocess.run([orc] + args, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0 and os.path.getsize(ref) > 100000, (r.stdout + r.stderr)[-2000:]
    r = subprocess.run([ocl] + args, capture_output=True, text=True, timeout=1200, cwd=str(tmp_path))
    out = r.stdout + r.stderr
    if "No OpenCL" in out or "clGetPlatformIDs" in out:
        pytest.skip("no OpenCL platform on this box")
    assert "Loaded reference values for" in out, out[-3000:]
    # What is asserted is the ORACLE: the OpenCL backend's fp32 outputs (unbatched and batched) against the oracle's values,
    # as a fraction of the reference's fp32 cross-backend limit (measured 0.0009x: fp32 rounding only). The exit code also
    # covers the OpenCL backend's own autotuned FP16 mode against its fp32 one, which is the reference's business and varies
    # with what its tuner picks on a given box, so it is reported but not required.
    margins = {k: float(v) for k, v in re.findall(r": ((?:batched )?(?:fp32|current cfg)) error vs reference closest margin:\s+([\d.eE+-]+)x of limit", out)}
    assert "fp32" in margins and "batched fp32" in margins, out[-3000:]
    assert margins["fp32"] < 0.05 and margins["batched fp32"] < 0.05, (margins, out[-1500:])
    if r.returncode != 0:
        print("katago_opencl testgpuerror exit code %d (its own fp16-vs-fp32 check); margins %s" % (r.returncode, margins))


def test_reference_gpuerror_as_the_reference_runs_it(tmp_path):
    """`testgpuerror` builds TWO evaluators on the device (command/gputest.cpp:122-133): the configured one and an fp32 one
    (useFP16 = false). Round 5: the backend serves fp32 (KMX_PREC_FP32 -> conv_f32.hip and the small kernels on float), so the command
    runs as the reference runs it on its own backends - no precision override. Reference values by the oracle in the role of the Eigen
    build. Exit code 0: every statistic within its limit - the fp32 evaluator against the strict fp32 limits, the default (fp16 with
    the 1/8 range transform) against the reduced-precision ones; the fp32 evaluator sits at fp32 rounding of the oracle (the reference's
    own OpenCL backend: 0.0009x of the limit)."""
    if not os.path.exists(G170):
        pytest.skip("g170 net not packaged")
    cfg = tmp_path / "bench.cfg"
    cfg.write_text(BENCH_CFG)
    ref = str(tmp_path / "ref.txt")
    args = ["testgpuerror", "-model", G170, "-config", str(cfg), "-boardsize", "9", "-quick", "-reference-file", ref]
    r = subprocess.run([ref_binary("katago_oracle")] + args, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0 and os.path.getsize(ref) > 100000, (r.stdout + r.stderr)[-2000:]
    r = subprocess.run([ref_binary("katago_hip")] + args, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    out = r.stdout + r.stderr
    assert "Loaded reference values for" in out, out[-3000:]
    margins = {k: float(v) for k, v in re.findall(r": ((?:batched )?(?:fp32|current cfg)) error vs reference closest margin:\s+([\d.eE+-]+)x of limit", out)}
    assert len(margins) == 4 and r.returncode == 0, (margins, out[-3000:])
    assert margins["fp32"] < 0.05 and margins["batched fp32"] < 0.05, margins  # fp32 rounding of the oracle's values
    assert margins["current cfg"] <= 0.25 and margins["batched current cfg"] <= 0.25, margins
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "testgpuerror_g170_fp32_evaluator.txt"), "w") as f:
            f.write("katago_hip testgpuerror -quick on g170-b6c96 9x9, no precision override (fp32 evaluator on the device + the default): margins "
                    "(x of the reference's limits) %s\n" % margins)


@pytest.mark.parametrize("prec", ["auto", "fp16", "bf16"])
def test_reference_gpuerror_acceptance_on_hip(tmp_path, prec):
    """The reference's cross-backend acceptance test `testgpuerror` (command/gputest.cpp; thresholds
    tests/testnnevalcanary.cpp:787-788 fp32, :806-807 reduced precision) on the katamx backend: real net (g170-b6c96), the
    reference's own 9x9 positions, reference values written by the oracle in the role of the Eigen build. The test also
    builds an "fp32" evaluator; katamx has none, so katamxPrecision pins both evaluators to the mode under test.
      auto: the backend default = fp16 with the 1/8 range transform (round 3): at most a quarter of the reduced-precision limits;
      fp16: passes outright — even the strict fp32-vs-fp32 limits (exit code 0);
      bf16: within the reduced-precision limits ("current cfg error vs reference"); only the strict fp32 rows exceed."""
    if not os.path.exists(G170):
        pytest.skip("g170 net not packaged")
    cfg = tmp_path / "bench.cfg"
    cfg.write_text(BENCH_CFG)
    ref = str(tmp_path / "ref.txt")
    args = ["testgpuerror", "-model", G170, "-config", str(cfg), "-boardsize", "9", "-quick", "-reference-file", ref]
    r = subprocess.run([ref_binary("katago_oracle")] + args, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0 and os.path.getsize(ref) > 100000, (r.stdout + r.stderr)[-2000:]
    r = subprocess.run([ref_binary("katago_hip")] + args + ["-override-config", "katamxPrecision=" + prec], capture_output=True,
                       text=True, timeout=600, cwd=str(tmp_path))
    out = r.stdout + r.stderr
    assert "Loaded reference values for" in out, out[-3000:]
    margins = {k: float(v) for k, v in re.findall(r": ((?:batched )?(?:fp32|current cfg)) error vs reference closest margin:\s+([\d.eE+-]+)x of limit", out)}
    assert len(margins) == 4, out[-3000:]
    assert margins["current cfg"] < 1.0 and margins["batched current cfg"] < 1.0, margins
    if prec in ("fp16", "auto"):
        assert r.returncode == 0 and margins["fp32"] < 1.0, (margins, out[-1500:])
    if prec == "auto":
        assert margins["current cfg"] <= 0.25 and margins["batched current cfg"] <= 0.25, margins
    keep = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "testgpuerror_g170_%s.txt" % prec), "w") as f:
            f.write("katago_hip testgpuerror -quick on g170-b6c96 9x9, katamxPrecision=%s: margins (x of the reference's limits) %s\n" % (prec, margins))

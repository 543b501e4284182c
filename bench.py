#!/usr/bin/env python3
"""bench.py — NN evals/s of the katamx HIP backend on the BASELINE.json workload.

    python bench.py --gpus 1 --steps 100 --warmup 10      (defaults; ~2 s of GPU time plus ~15 s of CPU baseline)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (NeuralNet::getOutput of the reference, cpp/neuralnet/nninterface.h:117)
over one batch of 256 synthetic 19x19 positions of a random-weight b18c384nbt (BASELINE.json configs[1]),
inputs already resident in HBM. Multi-GPU = independent replicas, one process per GPU, no collective on the data
path (the reference's only multi-GPU mode: one server thread per GPU on a shared queue, nneval.cpp:399-407);
value = rows evaluated by all ranks / max-over-ranks time ("weak" scaling: per-GPU work is fixed).

Rank 0 prints ONE JSON line, including
  roofline     : the dominant kernel (3x3 implicit-GEMM convolution) — algorithmic FLOPs per launch / average
                 launch duration measured live with hipEvents on the engine's own stream. The K timed steps that give
                 `value` run WITHOUT per-launch events (recording ~130 event pairs per step costs ~10 % of the step and
                 would understate `value`); min(K, 20) of the same steps are then repeated with the events on and that pass gives
                 `roofline` (its own ms_per_step is reported next to it); that pass runs the batch on ONE stream so that
                 a launch has the chip to itself, whereas the timed pass uses the product default for batch 256 — two
                 half-batches on two streams, whose kernels share the CUs (+6 % evals/s). `traffic` = HBM bytes per launch measured
                 IN THIS RUN when --pmc is given (bench.py re-runs itself for 3 steps under `rocprofv3 --pmc FETCH_SIZE` and
                 `--pmc WRITE_SIZE`, separate passes, and applies the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE x2),
                 null otherwise - a constant read from a committed file is not a measurement of the run that prints it;
  roofline_seam: the second kernel family (the fused 1x1 -> 1x1 seam between nested-bottleneck blocks, HBM-bound): algorithmic
                 bytes per launch / average launch duration of the same instrumented pass, against 8 TB/s;
  callers      : (N = 1, outside the timed region, ~15 s; --no-callers skips) what the callers of the path get on this box:
                 host rows through the persistent leaf batcher (katago_amd/leaf_pump, this repo's C++ consumer of the C ABI)
                 and, when integration/_build/katago_hip was built, the reference's own `benchmark` (nnEvals/s as
                 cpp/program/playutils.cpp:843,991-1000 defines it) from its unmodified search on 1024 fibers;
  small_batches: (N = 1, outside the timed region, < 1 s) ms per pass and rows/s of the same net at batch 1 / 8 / 32 from host rows
                 through kmx_eval - the latency-bound regime of a few games per GPU (BASELINE configs[2]);
  cpu_baseline : the CPU oracle (a port of the reference's Eigen path; Eigen itself is not buildable offline)
                 timed on this host's cores on a bounded sample of the same workload. Reported, not optimised against.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def synthetic_rows(n, seed):
    """Binary feature planes shaped like fillRowV7 output (SURVEY.md 8d fallback recipe) + globals."""
    rng = np.random.default_rng(seed)
    L = 19
    sp = np.zeros((n, L, L, 22), dtype=np.float32)
    sp[..., 0] = 1.0
    st = rng.random((n, L, L))
    sp[..., 1] = st < 0.2
    sp[..., 2] = (st >= 0.2) & (st < 0.4)
    occupied = (sp[..., 1] + sp[..., 2]) > 0
    for c in range(3, 22):
        plane = rng.random((n, L, L)) < 0.05
        sp[..., c] = plane & (occupied if c in (3, 4, 5, 14, 15, 16, 17) else ~occupied if c in (6,) else plane)
    gl = rng.normal(0.0, 0.5, (n, 19)).astype(np.float32)
    gl[:, 5] = rng.uniform(-0.5, 0.5, n)
    return sp.reshape(n, L * L, 22), gl


def cpu_baseline(model_path, batch_rows, min_seconds=10.0, max_seconds=30.0):
    from oracle import oracle

    om = oracle.loadModelFile(model_path)
    sp, gl = synthetic_rows(batch_rows, 4242)
    # the GPU box reports 256 logical CPUs but does not deliver them: one OpenMP team of 256 threads measures 0.26 evals/s, eight
    # pinned 32-thread processes together 4.9 (round 3, profiles/r03_steps/cpu_baseline_256_threads.txt) against ~20 for ONE team of
    # 32 - the container's CPU share, not the oracle's loops, is the limit. 32 threads are used and reported as `cores`.
    cores = oracle.usable_cores(32)
    rows = 0
    t0 = time.time()
    while True:
        oracle.getOutput(om, 19, 19, sp, gl, None, None, True, cores)
        rows += batch_rows
        el = time.time() - t0
        if el >= min_seconds or el + el / (rows / batch_rows) > max_seconds:
            break
    el = time.time() - t0
    return {"value": rows / el, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d b18c384nbt 19x19 evals in batches of %d, fp32 C oracle (OpenMP), %.1f s" % (rows, batch_rows, el)}


def measured_traffic(args):
    """HBM bytes per launch of the two dominant kernel families, measured NOW: this script re-run for 3 steps under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (counters in their own runs, one stream so that a launch is a whole
    batch), summed per kernel over its dispatches. FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts the 128-byte
    requests of wide streaming reads at 64 bytes and is doubled (MI355X_MICROARCH.md, HBM section)."""
    import csv
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import rocpd_summary

    keep = os.path.join(REPO, "gpurun_out")
    d = os.path.join(keep, "bench_pmc") if os.path.isdir(keep) else tempfile.mkdtemp(prefix="kmx_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-profile", "--no-callers", "--no-pmc",
           "--batch", str(args.batch), "--model", args.model, "--dtype", args.dtype]
    env = dict(os.environ, KMX_SPLIT_MIN="0")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        r = subprocess.run(["rocprofv3", "--pmc", c, "-d", os.path.join(d, "benchpmc_" + c), "-o", "bench", "--"] + cmd,
                           capture_output=True, text=True, timeout=240, env=env, cwd=os.environ.get("TMPDIR", "/tmp"))
        if r.returncode != 0:
            return {"error": "rocprofv3 --pmc %s failed: %s" % (c, (r.stdout + r.stderr)[-300:])}
    import contextlib

    with contextlib.redirect_stdout(sys.stderr):  # the summariser reports what it wrote; stdout carries ONE line, the JSON
        rocpd_summary.main(d, os.path.join(d, "summary"))

    return traffic_totals(lambda counter: os.path.join(d, "summary", "benchpmc_%s_pmc.csv" % counter))


def traffic_totals(csv_of):
    """Per kernel class, the FETCH_SIZE / WRITE_SIZE totals of a counted run (csv_of(counter) -> tools/rocpd_summary.py's *_pmc.csv)."""
    import csv

    def totals(counter, keys):
        tot = disp = 0.0
        passes = 0
        for r in csv.DictReader(open(csv_of(counter))):
            if r["Counter"] != counter:
                continue
            if any(k in r["Kernel"] for k in keys):
                tot += float(r["Sum"])
                disp += float(r["Dispatches"])
            if "inputExpandKernel" in r["Kernel"]:
                passes = int(float(r["Dispatches"]))  # one per pass (one stream, no split)
        return tot, int(disp), passes

    # A "launch" of the bench line's conv3x3 class is one CONVOLUTION: a chained dispatch (conv_chain_kernel.h) holds two or four, so
    # the class's bytes are summed over both kernels' dispatches and divided by the convolutions the passes held (finish_traffic).
    out = {}
    for name, keys in (("conv3x3", ("KS=3", "convChainKernel", "convSmallKernel")), ("conv1x1_pair", ("pointwisePair",))):
        f, nf, pf = totals("FETCH_SIZE", keys)
        w, nw, pw = totals("WRITE_SIZE", keys)
        if nf and nw and pf and pw:
            out[name] = {"fetch_kib_total": f, "write_kib_total": w, "dispatches": [nf, nw], "passes": [pf, pw]}
    return out


def finish_traffic(t, launches_per_pass):
    """HBM bytes per launch from measured_traffic's totals: a class's bytes over its launches (for the 3x3 class: convolutions) in the
    counted passes; FETCH_SIZE x2 on gfx950 (see measured_traffic)."""
    if not t or "error" in t or launches_per_pass <= 0:
        return t
    f = t["fetch_kib_total"] / (t["passes"][0] * launches_per_pass)
    w = t["write_kib_total"] / (t["passes"][1] * launches_per_pass)
    return {"hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0), "fetch_kib_raw": round(f, 1), "write_kib_raw": round(w, 1),
            "dispatches": t["dispatches"], "launches_per_pass": launches_per_pass,
            "source": "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 2`, one stream, summed "
                      "over the class's dispatches (3x3: convMfmaKernel KS=3 + convChainKernel + convSmallKernel) and divided by its launches in those passes (a "
                      "chained dispatch counts as the convolutions it holds); FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE x1"}


CALLER_CFG = """logDir = %s
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
nnMaxBatchSize = 256
numNNServerThreadsPerModel = 2
"""


def caller_rates(model_path, tmp):
    """What the CALLERS of the path get on this box (outside the timed region): (1) host rows through the persistent leaf batcher
    from this repo's C++ consumer of the C ABI; (2) the reference's `benchmark` - its own definition of nnEvals/s - from its
    unmodified search running 1024 search threads as fibers on 64 OS threads, through this repo's NNEvaluator."""
    import re
    import subprocess

    out = {}
    pump = os.path.join(REPO, "katago_amd", "leaf_pump")
    if os.path.exists(pump):
        r = subprocess.run([pump, model_path, "19", "256", "2", "8", "128", "3"], capture_output=True, text=True, timeout=120)
        if r.returncode == 0:
            out["host_rows_through_batcher_per_s"] = round(json.loads(r.stdout.strip().splitlines()[-1])["rows_per_s"], 1)
    hipx = os.path.join(REPO, "integration", "_build", "katago_hip")
    if os.path.exists(hipx):
        cfg = os.path.join(tmp, "kmx_bench_callers.cfg")
        with open(cfg, "w") as f:
            f.write(CALLER_CFG % os.path.join(tmp, "kmx_bench_gtp_logs"))
        env = dict(os.environ, KATAMX_LEAVES_PER_THREAD="16")
        r = subprocess.run([hipx, "benchmark", "-model", model_path, "-config", cfg, "-v", "8000", "-t", "1024", "-boardsize", "19", "-n", "4"],
                           capture_output=True, text=True, timeout=120, env=env, cwd=tmp)
        m = re.findall(r"visits/s = ([\d.]+) nnEvals/s = ([\d.]+).*avgBatchSize = ([\d.]+)", (r.stdout + r.stderr).replace("\r", "\n"))
        if r.returncode == 0 and m:
            out["reference_benchmark_nn_evals_per_s"] = float(m[-1][1])
            out["reference_benchmark"] = ("katago benchmark -v 8000 -t 1024 -boardsize 19 (4 positions), unmodified reference search on fibers "
                                          "(16 per OS thread), this repo's NNEvaluator + leaf batcher: %s visits/s, avg device batch %s rows" % (m[-1][0], m[-1][2]))
        # The same command at BASELINE configs[1]'s OWN setting (`benchmark -v 1600 -t 256 -fixed-batch-size 256`,
        # cpp/command/benchmark.cpp:206-217; its default 10 positions): 256 descents in one tree and searches of 1600 visits - a search
        # this short spends much of its time filling and draining its descents, so this is the lower of the two caller rates and is
        # reported beside the long-search one, not instead of it.
        r = subprocess.run([hipx, "benchmark", "-model", model_path, "-config", cfg, "-v", "1600", "-t", "256", "-fixed-batch-size", "256",
                            "-boardsize", "19"], capture_output=True, text=True, timeout=120, env=env, cwd=tmp)
        m = re.findall(r"visits/s = ([\d.]+) nnEvals/s = ([\d.]+).*avgBatchSize = ([\d.]+)", (r.stdout + r.stderr).replace("\r", "\n"))
        if r.returncode == 0 and m:
            out["reference_benchmark_configs1_nn_evals_per_s"] = float(m[-1][1])
            out["reference_benchmark_configs1"] = ("katago benchmark -v 1600 -t 256 -fixed-batch-size 256 -boardsize 19 (10 positions; BASELINE configs[1] as "
                                                   "written), same stack: %s visits/s, avg device batch %s rows" % (m[-1][0], m[-1][2]))
    if os.path.exists(hipx):
        # BASELINE configs[2], the second half of the metric: games/hour as the reference defines it (command/selfplay.cpp:388-389) -
        # `selfplay` with 8 parallel games on this GPU, the reference's production settings (tools/selfplay_cfg.py =
        # cpp/configs/training/selfplay8mainb18.cfg: 2000 / 350 visits, its rules, komi, forks ...), 19x19 only, every game's 8 search
        # threads as fibers on the game's own OS thread (8 leaves in flight per game), games played to the reference's OWN end conditions.
        # 8 full-length games of a random-weight net take a few minutes; KMX_BENCH_SELFPLAY_TIMEOUT (seconds, default 420; 0 skips the
        # leg) bounds it: a run that is cut short reports its NN rows/s and no games/hour - the figure is measured or absent, never derived.
        out.update(selfplay_rates(hipx, model_path, tmp))
    return out


def selfplay_rates(binary, model_path, tmp, game_threads=8, search_threads=8, timeout_s=None):
    import re
    import shutil
    import subprocess
    import tempfile

    sys.path.insert(0, os.path.join(REPO, "tools"))
    import selfplay_cfg

    if timeout_s is None:
        timeout_s = int(os.environ.get("KMX_BENCH_SELFPLAY_TIMEOUT", "420"))
    if timeout_s <= 0:
        return {}
    d = tempfile.mkdtemp(prefix="kmx_bench_selfplay_", dir=tmp)
    try:
        os.makedirs(os.path.join(d, "models"))
        shutil.copy(model_path, os.path.join(d, "models", "b18c384nbt-s1-d1.bin"))
        cfg = selfplay_cfg.write(os.path.join(d, "main.cfg"), numGameThreads=game_threads, numSearchThreads=search_threads, nnMaxBatchSize=64,
                                 logGamesEvery=1000, switchNetsMidGame="false", nnCacheSizePowerOfTwo=21, nnMutexPoolSizePowerOfTwo=16,
                                 **selfplay_cfg.ONLY_19)
        # KMX_BATCH_TRACE: one line per launched device batch on stderr (csrc/batcher.cpp) - the batch-size distribution of the run and, if the
        # child dies, the sizes that were on the device; HSA_DISABLE_COREDUMP_ON_EXCEPTION: a device exception is reported by the runtime's
        # own message (kind of fault) instead of by its core-dump attempt, which in round 5's driver run left nothing but "execvp failed"
        env = dict(os.environ, KATAMX_LEAVES_PER_THREAD=str(search_threads), KMX_BATCH_TRACE="1", HSA_DISABLE_COREDUMP_ON_EXCEPTION="1")
        p = subprocess.Popen([binary, "selfplay", "-config", cfg, "-models-dir", os.path.join(d, "models"), "-output-dir", os.path.join(d, "out"),
                              "-max-games-total", str(game_threads)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=d, env=env)
        try:
            log, _ = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal

            p.send_signal(signal.SIGINT)  # the reference stops its games, writes its totals and exits cleanly
            try:
                log, _ = p.communicate(timeout=60)
            except subprocess.TimeoutExpired:
                p.kill()
                log, _ = p.communicate()
        sizes = [int(v) for v in re.findall(r"^\[kmx batch\] slot \d+ rows (\d+)", log, re.M)]
        trace_tail = re.findall(r"^\[kmx batch\] (.*)$", log, re.M)[-8:]
        log = re.sub(r"^\[kmx batch\].*\n", "", log, flags=re.M)
        g = lambda k: float((re.findall(k + r": ([\d.]+)", log) or ["nan"])[-1])
        secs, total = g(r"Total selfplay runtime \(seconds\)"), g("Total games")
        rows, moves, fin, batches = g("Final NN rows"), g("Final moves played"), g("Final games finished"), g("Final NN batches")
        hist = {}
        for v in sizes:
            hist[v] = hist.get(v, 0) + 1
        if not secs > 0 or not rows > 0:
            # the child did not reach its totals: keep what a diagnosis needs - how it ended, the runtime's fault line, what was on the device
            import signal

            rc = p.returncode
            sig = signal.Signals(-rc).name if rc is not None and rc < 0 else None
            fault = [l.strip()[-300:] for l in log.splitlines() if re.search(r"HSA_STATUS|Memory access fault|GPU coredump|Aborted|terminate called|what\(\)", l)]
            return {"selfplay_error": {"returncode": rc, "signal": sig, "fault": fault[:4], "last_device_batches": trace_tail,
                                       "device_batch_rows_histogram": {str(k): hist[k] for k in sorted(hist)}, "log_tail": log[-600:]}}
        out = {"selfplay_nn_rows_per_s": round(rows / secs, 1)}
        if sizes:
            srt = sorted(sizes)
            out["selfplay_device_batch_rows"] = {"batches": len(srt), "mean": round(sum(srt) / len(srt), 1), "p10": srt[len(srt) // 10],
                                                 "p50": srt[len(srt) // 2], "p90": srt[(len(srt) * 9) // 10], "max": srt[-1]}
        what = ("katago selfplay (command/selfplay.cpp), b18c384nbt 19x19 random weights, %d game threads x %d search threads on fibers, the reference's "
                "production settings (selfplay8mainb18.cfg: 2000 / 350 visits), product path (own evaluator + featuriser + leaf batcher)" % (game_threads, search_threads))
        if fin >= game_threads and "Exited cleanly after signal" not in log:
            out["selfplay_games_per_hour"] = round(fin * 3600.0 / secs, 1)
            out["selfplay_games_per_hour_as_reference_counts"] = round(total * 3600.0 / secs, 1)
            out["selfplay"] = ("%s: %d full-length games (reference's own end conditions) finished in %.1f s - MEASURED, finished games x 3600 / 'Total selfplay "
                               "runtime'; the reference's own line counts 'Total games' = %d (it adds one per game thread at shutdown, selfplay.cpp:293); %.0f moves and "
                               "%.0f NN rows per game, average device batch %.1f rows" % (what, fin, secs, total, moves / fin, rows / fin, rows / max(batches, 1)))
        else:
            out["selfplay"] = ("%s: interrupted after %.0f s with %d of %d games finished - NN rows/s only, no games/hour (average device batch %.1f rows)"
                               % (what, secs, fin, game_threads, rows / max(batches, 1)))
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def small_batch_rates(nn, handle, sp, gl, sym, opt, dtype, sizes=(1, 8, 32, 64), reps=40, warm=8):
    """Small batches of the same net on this box (outside the timed region; host rows through kmx_eval - H2D, pass, D2H, synchronous):
    what BASELINE configs[2]'s regime (a few games per GPU) sees of the device. Per size: `warm` untimed passes (the first pass of a
    size loads the code objects of the work-group shapes it picks, and the device's clocks have dropped while the caller legs above ran
    their child processes), then `reps` passes timed ONE BY ONE; the MEDIAN is reported (round 3 reported a mean of 20 after 3 warm-ups
    and the driver's run showed 4.27 ms at batch 8 between 2.29 at batch 1 and 2.63 at batch 32: one slow outlier pass is enough for that),
    the fastest and slowest pass beside it. Informative: a failure here is recorded, not raised."""
    try:
        B = sp.shape[0]
        out = {"path": "kmx_eval from host rows (PCIe included), %s; median of %d passes after %d warm-up passes" % (dtype, reps, warm),
               "ms_per_pass": {}, "rows_per_s": {}, "ms_min_max": {}}
        sp3, gl2 = sp.reshape(B, -1, sp.shape[-1]), gl.reshape(B, -1)
        for n in sizes:
            if n > B:
                continue
            for _ in range(warm):
                nn.getOutput(handle, sp3[:n], gl2[:n], sym[:n], opt[:n])
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                nn.getOutput(handle, sp3[:n], gl2[:n], sym[:n], opt[:n])
                ts.append((time.perf_counter() - t0) * 1e3)
            ms = float(np.median(ts))
            out["ms_per_pass"][str(n)] = round(ms, 3)
            out["ms_min_max"][str(n)] = [round(min(ts), 3), round(max(ts), 3)]
            out["rows_per_s"][str(n)] = round(n / ms * 1e3)
        return out
    except Exception as e:  # noqa: BLE001
        return {"small_batches_error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="b18c384nbt")
    ap.add_argument("--dtype", default="auto", choices=["auto", "bf16", "fp16"],
                    help="auto = the backend's default: fp16 with the reference's 1/8 range transform for convolutional nets")
    # (round 5: the counter passes are part of the DEFAULT full run, so that the driver's own line carries `traffic`; a reduced run
    # (--no-callers / --no-profile / --no-cpu-baseline) takes them with --pmc only; --no-pmc skips them)
    ap.add_argument("--pmc", action="store_true", help="(the default) measure HBM traffic per launch in this run: two extra rocprofv3 passes of 3 steps, ~1 min")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 counter passes: `traffic` is null")
    ap.add_argument("--no-callers", action="store_true", help="skip the caller-side rates (leaf pump, reference benchmark on fibers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the second, hipEvent-instrumented pass (no roofline object)")
    ap.add_argument("--host-buffers", action="store_true",
                    help="also time the host-pointer entry kmx_eval (PCIe-inclusive; reported beside, never as, value)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: katamx has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        # replicas only, no collective on the data path (north_star: "no RCCL needed"): the bookkeeping of the measurement - a barrier,
        # the slowest rank's time, the total row count - goes over gloo on CPU tensors; nothing in this process initialises RCCL
        dist.init_process_group("gloo")

    traffic = None
    # by default in the FULL run only (the driver's command); the reduced runs of tools/ and tests ask for it with --pmc
    full_run = not (args.no_callers or args.no_profile or args.no_cpu_baseline)
    if (args.pmc or full_run) and not args.no_pmc and rank == 0 and world == 1:
        try:
            traffic = measured_traffic(args)  # before this process touches the GPU: the passes have the device to themselves
        except Exception as e:  # a measurement beside the timed region must not cost the line
            print("bench.py: HBM counter passes failed (%s: %s): traffic stays null" % (type(e).__name__, e), file=sys.stderr)
            traffic = None
        if isinstance(traffic, dict) and "error" in traffic:
            print("bench.py: %s" % traffic["error"], file=sys.stderr)

    from katago_amd import capi, modelgen, nninterface as nn

    lib = capi.load_library()
    nn.globalInitialize()
    tmp = os.environ.get("TMPDIR", "/tmp")
    model_path = os.path.join(tmp, "kmx_bench_%s_r%d.bin" % (args.model, rank))
    modelgen.write_model(model_path, args.model, seed=20260921)
    model = nn.loadModelFile(model_path)
    ctx = nn.createComputeContext([local_rank], 19, 19, precision=args.dtype)
    handle = nn.createComputeHandle(ctx, model, args.batch, True, local_rank)
    dtype = handle.precision  # what "auto" resolved to: "fp16" / "bf16"

    S, B = 361, args.batch
    sp, gl = synthetic_rows(B, 20260921 + rank)
    d_sp = torch.from_numpy(sp).cuda()
    d_gl = torch.from_numpy(gl).cuda()
    sym = (np.arange(B) % 8).astype(np.int32)
    opt = np.zeros(B, dtype=np.float32)
    d_pol = torch.empty((B, S + 1), device="cuda")
    d_val = torch.empty((B, 3), device="cuda")
    d_sc = torch.empty((B, 6), device="cuda")
    d_own = torch.empty((B, S), device="cuda")
    torch.cuda.synchronize()
    if os.environ.get("KMX_DEBUG_ALLOC"):  # fault triage: the caller's buffers beside the library's own ([kmx alloc] lines)
        for nm, t in (("d_sp", d_sp), ("d_gl", d_gl), ("d_pol", d_pol), ("d_val", d_val), ("d_sc", d_sc), ("d_own", d_own)):
            print("[bench alloc] %s 0x%x .. 0x%x" % (nm, t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()), file=sys.stderr, flush=True)
    sym_p = sym.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    opt_p = opt.ctypes.data_as(ctypes.POINTER(ctypes.c_float))

    def step(sync=False):
        capi.check(lib.kmx_eval_device(handle._p, B, d_sp.data_ptr(), d_gl.data_ptr(), sym_p, opt_p, d_pol.data_ptr(),
                                       d_val.data_ptr(), d_sc.data_ptr(), d_own.data_ptr(), 1 if sync else 0), lib)

    from katago_amd import replicas

    def barrier():
        handle.sync()
        replicas.barrier(torch.device("cuda", local_rank))

    for _ in range(args.warmup):
        step(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    handle.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    profiled_elapsed = None
    if not args.no_profile:
        # second pass over the same K steps with a hipEvent pair around every launch (engine stream)
        prof_steps = min(args.steps, 20)  # ~130 event pairs per step are kept until they are read back
        # the roofline is the kernel's with the chip to itself: one stream (the timed pass above runs the product
        # default, two half-batches on two streams, where two launches share the CUs and each takes longer)
        capi.check(lib.kmx_handle_set_split_min(handle._p, 0), lib)
        capi.check(lib.kmx_handle_set_profiling(handle._p, 1), lib)
        handle.sync()
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            step(False)
        handle.sync()
        torch.cuda.synchronize()
        profiled_elapsed = (time.perf_counter() - t1) / prof_steps * args.steps
    total_rows, elapsed = replicas.whole_job(args.steps * B, elapsed)

    assert torch.isfinite(d_pol).all() and torch.isfinite(d_val).all(), "non-finite outputs"

    prof_entries = None
    if not args.no_profile:
        ent = (capi.ProfileEntry * 32)()
        cnt = ctypes.c_int()
        capi.check(lib.kmx_handle_get_profile(handle._p, ent, 32, ctypes.byref(cnt)), lib)
        prof_entries = {ent[i].name.decode(): (ent[i].launches, ent[i].total_ms, ent[i].flops, ent[i].bytes) for i in range(cnt.value)}
        capi.check(lib.kmx_handle_set_profiling(handle._p, 0), lib)
        capi.check(lib.kmx_handle_set_split_min(handle._p, -1), lib)  # back to the creation value (KMX_SPLIT_MIN or the default)

    host_rate = host_packed_rate = None
    if args.host_buffers and rank == 0:
        # the reference's getOutput hands over host rows: same batch through kmx_eval (H2D + pass + D2H, synchronous)
        pol = np.empty((B, S + 1), dtype=np.float32)
        val = np.empty((B, 3), dtype=np.float32)
        sco = np.empty((B, 6), dtype=np.float32)
        own = np.empty((B, S), dtype=np.float32)
        FP = ctypes.POINTER(ctypes.c_float)
        PT = FP * B
        sp2, gl2 = sp.reshape(B, -1), gl.reshape(B, -1)
        f = lambda a: a.ctypes.data_as(FP)
        ptrs = (PT(*[f(sp2[i]) for i in range(B)]), PT(*[f(gl2[i]) for i in range(B)]), PT(*[f(pol[i]) for i in range(B)]),
                PT(*[f(own[i]) for i in range(B)]))
        call = lambda: capi.check(lib.kmx_eval(handle._p, B, ptrs[0], ptrs[1], sym_p, opt_p, ptrs[2], f(val), f(sco), ptrs[3]), lib)
        call()
        th = time.perf_counter()
        for _ in range(args.steps):
            call()
        host_rate = args.steps * B / (time.perf_counter() - th)
        # the same rows bit-packed (kmx_eval_packed): 1012 bytes per row over PCIe instead of 31768
        pk = nn.packRows(sp, 19, 19)
        U8P = ctypes.POINTER(ctypes.c_uint8)
        pk_ptrs = (U8P * B)(*[pk[i].ctypes.data_as(U8P) for i in range(B)])
        callp = lambda: capi.check(lib.kmx_eval_packed(handle._p, B, pk_ptrs, ptrs[1], None, sym_p, opt_p, ptrs[2], f(val), f(sco), ptrs[3]), lib)
        callp()
        th = time.perf_counter()
        for _ in range(args.steps):
            callp()
        host_packed_rate = args.steps * B / (time.perf_counter() - th)

    # What this box gives: the convolution's step shape (18 MFMAs per wave and step, its LDS reads and its barrier) without any
    # memory traffic, and the shader clock observed during it. Whole-net numbers vary by ~20 % between gpurun boxes; this line
    # says which kind of box a number came from.
    box = None
    if rank == 0:
        ms_, tf_, mhz_ = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        if lib.kmx_bench_mfma(8, 256, 3, 540, 10, ctypes.byref(ms_), ctypes.byref(tf_), ctypes.byref(mhz_)) == 0:
            box = {"mfma_lds_barrier_loop_tflops": round(tf_.value, 1), "shader_mhz_during_it": round(mhz_.value)}
        # Round 6: what the matrix cores SUSTAIN on operands that look like the convolution's (uniform noise) - the chip clocks down under
        # them whatever else a kernel does (tools/mfma_power_probe.py): the bare chain of MFMAs and the convolution's step shape,
        # in the pass's own precision, ~1.2 s each (the power controller needs tens of ms to settle; the 4 ms loop above runs at the clock
        # the passes left the chip at).
        if box is not None and world == 1:
            prec = capi.PREC_FP16 if dtype == "fp16" else capi.PREC_BF16
            for key, shape in (("mfma_chain", 0), ("mfma_lds_barrier_loop", 1)):
                if lib.kmx_bench_mfma_sustained(256, shape, 2, prec, 1.2, ctypes.byref(tf_), ctypes.byref(mhz_)) == 0:
                    box[key + "_sustained_on_noise_operands_tflops"] = round(tf_.value, 1)
                    box[key + "_sustained_on_noise_operands_mhz"] = round(mhz_.value)
            # ... and the bare chain on operands DISTRIBUTED like this bench's own (normal weights of a random-init 192-channel 3x3 layer x mish
            # of a unit normal at 1/8, data kind 3): a little BELOW uniform noise on the boxes measured (1 555 against 1 679 TFLOP/s in fp16)
            if lib.kmx_bench_mfma_sustained(256, 0, 3, prec, 1.2, ctypes.byref(tf_), ctypes.byref(mhz_)) == 0:
                box["mfma_chain_sustained_on_net_like_operands_tflops"] = round(tf_.value, 1)
                box["mfma_chain_sustained_on_net_like_operands_mhz"] = round(mhz_.value)

    roofline = None
    if prof_entries:
        total_ms = sum(v[1] for v in prof_entries.values())
        name, (launches, ms, flops, nbytes) = max(prof_entries.items(), key=lambda kv: kv[1][1])
        achieved = flops / (ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[dtype]
        roofline = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": finish_traffic((traffic or {}).get(name), launches / args.steps),
                    "pass": "single stream, hipEvent pair per launch (the timed pass runs two half-batch streams)",
                    "profiled_ms_per_step": round(profiled_elapsed / args.steps * 1e3, 4),
                    "avg_launch_ms": round(ms / launches, 5), "launches": int(launches),
                    "flops_per_launch": flops / launches, "algorithmic_bytes_per_launch": nbytes / launches,
                    "kernel_time_share": {k: round(v[1] / total_ms, 4) for k, v in prof_entries.items()},
                    "kernel_avg_launch_us": {k: round(v[1] / max(v[0], 1) * 1e3, 2) for k, v in prof_entries.items()}}
        sustained = (box or {}).get("mfma_chain_sustained_on_noise_operands_tflops")
        if sustained:
            # `peak` stays the nominal dense figure (256 CUs x 2.4 GHz); this is the same `achieved` against what a kernel of nothing but
            # MFMAs sustains on this box, on noise-like operands, in this precision
            roofline["sustained_mfma_only_on_noise_operands_tflops"] = sustained
            roofline["frac_of_sustained"] = round(achieved / sustained, 4)
        net_like = (box or {}).get("mfma_chain_sustained_on_net_like_operands_tflops")
        if net_like:
            roofline["frac_of_sustained_on_net_like_operands"] = round(achieved / net_like, 4)

    roofline_seam = None
    if prof_entries and prof_entries.get("conv1x1_pair", (0, 0, 0, 0))[0] > 0:
        launches, ms, flops, nbytes = prof_entries["conv1x1_pair"]
        gbs = nbytes / (ms * 1e-3) / 1e9
        roofline_seam = {"bound": "hbm", "kernel": "conv1x1_pair", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": finish_traffic((traffic or {}).get("conv1x1_pair"), launches / args.steps),
                         "avg_launch_ms": round(ms / launches, 5), "launches": int(launches),
                         "algorithmic_bytes_per_launch": nbytes / launches, "flops_per_launch": flops / launches}

    callers = None
    if rank == 0 and world == 1 and not args.no_callers:
        handle.sync()
        try:  # informative extras: whatever goes wrong here (a missing binary, a hung child) must not cost the bench line
            callers = caller_rates(model_path, tmp)
        except Exception as e:  # noqa: BLE001
            callers = {"callers_error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    small = None
    if rank == 0 and world == 1 and not args.no_callers:
        handle.sync()
        small = small_batch_rates(nn, handle, sp, gl, sym, opt, dtype)

    if rank == 0:
        value = total_rows / elapsed
        flops_eval = model.info.flops_per_position * S
        out = {
            "metric": "nn_evals_per_s", "value": round(value, 1), "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype,
            "precision": ("fp16 storage with the reference's 1/8 activation-range transform (desc.cpp:2718-2736), fp32 accumulate"
                          if dtype == "fp16" else "bf16 storage, fp32 accumulate") + (" (backend default)" if args.dtype == "auto" else ""),
            "data": "synthetic positions (SURVEY 8d recipe) on random-init weights of the named architecture (no trained b18c384nbt is available offline)",
            "config": {"workload": "%s 19x19 random weights, batch %d per GPU, NeuralNet::getOutput pass (device-resident inputs)" % (args.model, B),
                       "gflop_per_eval": round(flops_eval / 1e9, 3), "parallelism": "replicas x%d" % world,
                       "whole_net_tflops": round(value * flops_eval / 1e12, 1),
                       "whole_net_frac_of_mfma_peak": round(value * flops_eval / 1e12 / (MFMA_PEAK_TFLOPS[dtype] * world), 4)},
            "roofline": roofline,
            "roofline_seam": roofline_seam,
            "box": box,
        }
        if callers:
            out.update(callers)
        if small:
            out["small_batches"] = small
        if host_rate is not None:
            out["host_buffer_evals_per_s"] = round(host_rate, 1)
            out["host_buffer_packed_evals_per_s"] = round(host_packed_rate, 1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model_path, 8)
        print(json.dumps(out), flush=True)
    handle.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The launch schedule, checked without a GPU. tests/fakehip/fakehip.cpp stands in for the HIP runtime (LD_PRELOAD): device
memory is host memory, kernels are not executed, every launch is logged — kernel, grid, block, LDS bytes, every byte of
the argument struct with pointers rewritten as buffer-ordinal+offset, and a content hash of every buffer (weights!) when
it first appears. tests/fakehip/run_schedule.py drives libkatamx.so through kmx_eval for a list of batch sizes.

Two uses:
  * regression guard for the convolutional nets: the md5 of the log must equal tests/golden/schedule_md5.json, which was
    generated with the library built from the last commit whose kernels and schedule ran green on the MI355X (126 -m gpu
    tests). A refactor of the engine that changes any launch argument, any weight re-tiling or the shape chooser shows up
    here before it reaches a GPU. (Regenerate deliberately with `python tests/test_schedule_dryrun.py --regenerate`.)
  * the transformer schedule (off by default) is constructed and walked: strides, offsets and LDS
    sizes of every launch are checked against the model's dimensions."""
import hashlib
import json
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
FAKE_DIR = os.path.join(REPO, "tests", "fakehip")
GOLDEN = os.path.join(REPO, "tests", "golden", "schedule_md5.json")
LIB = os.path.join(REPO, "katago_amd", "libkatamx.so")
# (architecture, modelgen keywords, max batch, batch sizes): small batches pick the narrow shapes, 224+ is split in two
CASES = {
    "b3c64nbt_v15": ("b3c64nbt", {}, 256, [1, 7, 64, 150, 223, 224, 256]),
    "b6c96_v8": ("b6c96", {"version": 8}, 256, [1, 7, 64, 150, 223, 224, 256]),
    "b10c128_v14": ("b10c128", {"version": 14}, 256, [1, 7, 64, 150, 223, 224, 256]),
    "b2c32nbt_v16": ("b2c32nbt", {"version": 16}, 256, [1, 7, 64, 150, 223, 224, 256]),
    "b18c384nbt_v15": ("b18c384nbt", {}, 256, [1, 7, 64, 150, 223, 224, 256]),
}


def build_fakehip(out_dir):
    so = os.path.join(out_dir, "libfakehip.so")
    cmd = ["/opt/rocm/bin/hipcc", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-DKMX_FAKEHIP_TRANSFORMER", "-o", so, os.path.join(FAKE_DIR, "fakehip.cpp")]
    subprocess.run(cmd, check=True, capture_output=True)
    return so


def dry_run(fake_so, model_path, max_batch, sizes, log_path, env_extra=None, lib=LIB):
    env = dict(os.environ, LD_PRELOAD=fake_so, KMX_FAKEHIP_LOG=log_path)
    env.pop("KMX_EXPERIMENTAL_TRANSFORMER", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(FAKE_DIR, "run_schedule.py"), lib, model_path, str(max_batch)] + [str(s) for s in sizes],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "done" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    return open(log_path).read()


def generate(case, tmp_dir, fake_so, lib=LIB):
    from katago_amd import modelgen

    arch, kw, max_batch, sizes = CASES[case]
    model = os.path.join(tmp_dir, case + ".bin")
    modelgen.write_model(model, arch, seed=1, **kw)
    log = dry_run(fake_so, model, max_batch, sizes, os.path.join(tmp_dir, case + ".log"), lib=lib)
    return hashlib.md5(log.encode()).hexdigest(), log


@pytest.fixture(scope="module")
def fake_so(tmp_path_factory):
    return build_fakehip(str(tmp_path_factory.mktemp("fakehip")))


@pytest.mark.parametrize("case", sorted(CASES))
def test_convnet_schedule_matches_the_gpu_verified_one(case, fake_so, tmp_path):
    golden = json.load(open(GOLDEN))
    digest, log = generate(case, str(tmp_path), fake_so)
    launches = [l for l in log.splitlines() if l.startswith("launch ")]
    assert len(launches) > 50
    assert digest == golden["md5"][case], "launch log of %s differs from the GPU-verified schedule (%d launches)" % (case, len(launches))


def _fields(line):
    m = re.match(r"launch (\S+) grid (\d+),(\d+),(\d+) block (\d+) lds (\d+) args(.*)", line)
    return m.group(1), tuple(int(m.group(i)) for i in (2, 3, 4)), int(m.group(5)), int(m.group(6)), m.group(7).split()


@pytest.mark.parametrize("name,C,H,KVH,QD,VD,F,blocks", [("torch_tfa", 32, 4, 4, 8, 8, 48, ("attn", "ffn", "attn", "ffn")),
                                                           ("torch_tfb", 32, 4, 2, 8, 4, 48, None)])
def test_transformer_schedule_is_consistent(name, C, H, KVH, QD, VD, F, blocks, fake_so, tmp_path):
    model = os.path.join(REPO, "tests", "golden", name + ".bin.gz")
    n = 3
    log = dry_run(fake_so, model, 8, [n], str(tmp_path / "tf.log"), {"KMX_EXPERIMENTAL_TRANSFORMER": "1"})
    launches = [_fields(l) for l in log.splitlines() if l.startswith("launch ")]
    kinds = [("attention" if "attentionMfmaKernel" in k else "rmsnorm" if "rmsNormKernel" in k else "boardrms" if "boardRmsKernel" in k else
              "swiglu" if "swiGluKernel" in k else "conv" if ("convMfmaKernel" in k or "convSmallKernel" in k) else "other") for k, *_ in launches]
    S = 361
    att = [l for l, k in zip(launches, kinds) if k == "attention"]
    assert att
    for kname, grid, block, lds, args in att:
        assert grid[1] == n and block == 256 and "Li16ELi32E" in kname  # head dims 8 / 4..8 padded to 16 / 32 for the MFMA tiles
    if blocks is not None:
        # per block: rmsnorm, fused projection conv, attention|swiglu, residual conv; then the tip rmsnorm
        body = kinds[2:2 + 4 * len(blocks)]
        want = []
        for b in blocks:
            want += ["rmsnorm", "conv", "attention" if b == "attn" else "swiglu", "conv"]
        assert body == want and kinds[2 + 4 * len(blocks)] == "rmsnorm"
        kname, grid, block, lds, args = att[0]
        assert grid == (H, n, 1) and lds == 384 * 16 * 2 + 32 * 384 * 2 + 384 * 4  # K[384][16], V^T[32][384], mask
        qkv_stride = int(args[1], 16) & 0xFFFFFFFF
        k_off = int(args[1], 16) >> 32
        v_off = int(args[2], 16) & 0xFFFFFFFF
        assert (qkv_stride, k_off, v_off) == (96, H * QD, H * QD + KVH * QD)
        assert int(args[2], 16) >> 32 == H and int(args[3], 16) == (QD << 32 | KVH) and int(args[4], 16) == VD
        sw = [l for l, k in zip(launches, kinds) if k == "swiglu"][0]
        assert int(sw[4][1], 16) == (F << 32 | 2 * F) and int(sw[4][2], 16) == F and int(sw[4][5], 16) == n * S
    else:
        assert "boardrms" in kinds and kinds.count("attention") == 3  # two inside the nested block, one in the trunk
    # every convolution tiles its padded channels with the chosen shape
    for (kname, grid, block, lds, args), k in zip(launches, kinds):
        if k == "conv" and "convSmallKernel" in kname:  # the small-batch 3x3 shapes: a board x 32 (or 64) channels, 4 + 4 waves
            cout_pad = int(args[5], 16) & 0xFFFFFFFF  # (ConvArgs: in, w, wFrag, zeroPage, inC | nChunks, coutPad | N, ...)
            wn = 2 if kname.endswith("ELb1ELi2ELb0EEEvNS_8ConvArgsE") else 1  # cfg 126: two channel tiles per wave
            # (grid z = 3: the cell tiles of a board over three work-groups while batch x channel tiles x 3 <= 256, cfg 127 / 117)
            assert grid[:2] == (cout_pad // (32 * wn), n) and block == 512 and lds <= 160 * 1024
            # (z = 2: over two work-groups, cfg 125, while batch x channel tiles x 2 <= 256)
            assert grid[2] == (3 if n * (cout_pad // 32) * 3 <= 256 else 2 if n * (cout_pad // 32) * 2 <= 256 else 1), (grid, n, cout_pad)
            assert wn == (2 if n * (cout_pad // 32) > 256 else 1), (kname, n, cout_pad)
        elif k == "conv":
            # kernel size, WN, WNW, ring depth, flags (0, or 262144 = ABL_SPLIT: a board's cell tiles over three work-groups, cfg 113)
            m = re.search(r"ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(0|262144)EEE", kname)
            ks, wn, wnw = int(m.group(1)), int(m.group(2)), int(m.group(3))
            cout_pad = int(args[5], 16) & 0xFFFFFFFF
            assert grid == (cout_pad // (32 * wn * wnw), n, 3 if m.group(5) != "0" else 1) and block == 256 * wnw and lds <= 160 * 1024


def test_reference_transformer_nets_build_a_schedule(fake_so, tmp_path):
    d = os.path.join(REPO, "oracle", "_ref", "models")
    nets = [os.path.join(d, f) for f in ("b7c96h3tfrs-test5-cnorm.bin.gz", "b7c96h6kv3qk32v16tflrs-fson-bnh.bin.gz")]
    if not all(os.path.exists(p) for p in nets):
        pytest.skip("reference test nets not packaged")
    for p in nets:  # q/k dim 32, v dim 32 or 16 (padded to 32)
        log = dry_run(fake_so, p, 16, [16], str(tmp_path / "ref.log"), {"KMX_EXPERIMENTAL_TRANSFORMER": "1"})
        att = [l for l in log.splitlines() if l.startswith("launch ") and "attentionMfmaKernel" in l]
        assert len(att) == 7 and all("Li32ELi32E" in l for l in att)
    log = dry_run(fake_so, nets[1], 16, [16], str(tmp_path / "valu.log"), {"KMX_EXPERIMENTAL_TRANSFORMER": "1", "KMX_ATTENTION_VALU": "1"})
    att = [l for l in log.splitlines() if l.startswith("launch ") and "attentionKernel" in l]
    assert len(att) == 7 and all("Li32ELi16E" in l and "block 192" in l for l in att)  # the plain kernel, selected by the environment


if __name__ == "__main__":
    if "--regenerate" in sys.argv:
        import tempfile

        lib = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else LIB
        tmp = tempfile.mkdtemp(prefix="kmxsched")
        so = build_fakehip(tmp)
        out = {"generated_with": lib, "md5": {c: generate(c, tmp, so, lib=lib)[0] for c in sorted(CASES)}}
        json.dump(out, open(GOLDEN, "w"), indent=1, sort_keys=True)
        print(json.dumps(out, indent=1))


def test_two_devices_in_one_process(fake_so, tmp_path):
    """The reference's in-process multi-GPU mode (one NNEvaluator server thread per GPU, nneval.cpp:399-407; config keys
    INTEGRATION.md section 2) without an 8-GPU node: two fake devices, two server threads, a plain handle and a leaf batcher per
    device (tests/fakehip/run_two_devices.py). Every stream, event and launch of device k must be touched with device k current -
    including by the batcher's own dispatcher / completion threads, which no caller ever bound to a device."""
    from katago_amd import modelgen

    model = str(tmp_path / "b2.bin")
    modelgen.write_model(model, "b2c32nbt", seed=3, version=16)
    log_path = str(tmp_path / "two.log")
    env = dict(os.environ, LD_PRELOAD=fake_so, KMX_FAKEHIP_LOG=log_path, KMX_FAKEHIP_DEVICES="2")
    p = subprocess.run([sys.executable, os.path.join(FAKE_DIR, "run_two_devices.py"), LIB, model], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "done" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    log = open(log_path).read().splitlines()
    violations = [l for l in log if l.startswith("VIOLATION")]
    assert not violations, violations[:10]
    per_dev = {d: sum(1 for l in log if l.startswith("dev %d launch" % d)) for d in (0, 1)}
    assert per_dev[0] > 100 and per_dev[1] > 100, per_dev  # (how the batchers cut their rows into batches depends on timing)


@pytest.mark.parametrize("policy", ["spread", "fill"])
def test_eight_devices_selfplay_in_one_process(fake_so, tmp_path, policy):
    """The 8-GPU recipe without the node: tools/selfplay_8gpu.sh - `katago_hip selfplay` with the reference's production settings, one
    leaf port per device chosen by ...DeviceToUseThread0..7 (program/setup.cpp:174-220) - on EIGHT fake devices. The whole host
    stack runs (search on fibers, this repo's evaluator and featuriser, eight leaf batchers with their dispatcher / completion
    threads, engines, staging copies); kernels are not executed, so the games are random play. Checked: eight ports on eight
    devices, no stream / event / launch touched while another device was current (also at tear-down: the main thread frees all
    eight batchers), every device gets its share of the rows (the evaluator sends a row to the device with the fewest rows in
    flight; with KATAMX_PORT_POLICY=fill to the first port that holds fewer than a target - the A/B switch for the day a node exists),
    every port's helper threads bound to the CPUs of its device's NUMA node (a fake two-node sysfs tree), all games finish and the shards
    hold all three board sizes."""
    import shard_checks
    from katago_amd import modelgen

    binary = os.path.join(REPO, "integration", "_build", "katago_hip")
    if not os.path.exists(binary):
        pytest.skip("integration/_build/katago_hip not built (make -C integration needs the reference checkout)")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "models"))
    modelgen.write_model(os.path.join(d, "models", "b2c32nbt-s1-d1.bin.gz"), "b2c32nbt", seed=3)
    log_path = os.path.join(d, "fake.log")
    # a two-socket node in miniature (numa.h): fake devices 0-3 sit on NUMA node 0, 4-7 on node 1 (sysfs tree under KMX_SYSFS_ROOT; the
    # fake runtime's PCI ids are 0000:1A:00.0 + device, upper case as the real runtime prints them), the CPUs this process may use are
    # dealt to the two nodes half and half
    cpus = sorted(os.sched_getaffinity(0))
    halves = [cpus[:max(1, len(cpus) // 2)], cpus[len(cpus) // 2:] or cpus]
    sysfs = os.path.join(d, "sys")
    for dev in range(8):
        pci = os.path.join(sysfs, "bus", "pci", "devices", "0000:%02x:00.0" % (0x1A + dev))
        os.makedirs(pci)
        with open(os.path.join(pci, "numa_node"), "w") as f:
            f.write("%d\n" % (dev // 4))
    for node in range(2):
        nd = os.path.join(sysfs, "devices", "system", "node", "node%d" % node)
        os.makedirs(nd)
        with open(os.path.join(nd, "cpulist"), "w") as f:
            f.write(",".join(str(c) for c in halves[node]) + "\n")
    env = dict(os.environ, KMX_LAUNCH_PREFIX="env LD_PRELOAD=%s KMX_FAKEHIP_DEVICES=8 KMX_FAKEHIP_QUIET=1 KMX_FAKEHIP_LOG=%s" % (fake_so, log_path),
               KMX_SELFPLAY_ARGS="-max-games-total 48", KATAMX_LEAVES_PER_THREAD="2", KMX_SYSFS_ROOT=sysfs, KATAMX_NUMA_VERBOSE="1",
               KATAMX_PORT_POLICY=policy, KATAMX_PORT_FILL_ROWS="8")
    extra = ["nnMaxBatchSize=16", "maxVisits=12", "cheapSearchVisits=6", "reducedVisitsMin=6", "estimateLeadVisits=3", "maxMovesPerGame=24",
             "logGamesEvery=1000", "nnCacheSizePowerOfTwo=14", "nnMutexPoolSizePowerOfTwo=10", "handicapAsymmetricPlayoutProb=0.0",
             "normalAsymmetricPlayoutProb=0.0", "switchNetsMidGame=false", "maxRowsPerTrainFile=100", "firstFileRandMinProp=1.0",
             "bSizes=9,13,19", "bSizeRelProbs=1,1,1", "allowRectangleProb=0.0"]
    p = subprocess.run([os.path.join(REPO, "tools", "selfplay_8gpu.sh"), os.path.join(d, "models"), os.path.join(d, "out"), "3", "2"] + extra,
                       capture_output=True, text=True, timeout=900, env=env, cwd=d)
    log = p.stdout + p.stderr
    assert p.returncode == 0 and "All cleaned up, quitting" in log, log[-3000:]
    ports = re.findall(r"leaf port: device (\d+) ", log)
    assert sorted(int(x) for x in ports) == list(range(8)), ports
    fake = open(log_path).read().splitlines()
    assert not [l for l in fake if l.startswith("VIOLATION")], [l for l in fake if l.startswith("VIOLATION")][:10]
    per_dev = {int(m.group(1)): int(m.group(2)) for m in (re.match(r"dev (\d+) launches (\d+)", l) for l in fake) if m}
    assert sorted(per_dev) == list(range(8)) and (policy == "fill" or min(per_dev.values()) > 1000), per_dev
    if policy == "spread":
        assert max(per_dev.values()) < 1.5 * min(per_dev.values()), per_dev  # least rows in flight first: no device is left behind
        assert "rows go to the port with the fewest rows in flight" in log
    else:
        # fill-first: a port is passed over only while it holds KATAMX_PORT_FILL_ROWS rows, so the first devices carry most of the load
        assert "rows go to the first port below 8 rows in flight (KATAMX_PORT_POLICY=fill)" in log
        assert per_dev[0] > 2 * per_dev[7], per_dev
    # NUMA (round 5): every device's dispatcher and completion threads run on the CPUs of ITS node, pinned staging prefers that node
    def cpulist(text):
        out = []
        for part in text.split(","):
            a, _, b = part.partition("-")
            out.extend(range(int(a), int(b or a) + 1))
        return out
    bound = re.findall(r"\[katamx numa\] device (\d+) -> node (\d+): (dispatcher|completion) thread bound to cpus ([\d,\-]+)", log)
    seen = {(int(dv), what) for dv, _, what, _ in bound}
    assert seen == {(dv, what) for dv in range(8) for what in ("dispatcher", "completion")}, sorted(seen)
    for dv, node, what, cl in bound:
        assert int(node) == int(dv) // 4 and cpulist(cl) == halves[int(node)], (dv, node, what, cl, halves)
    staged = {(int(dv), int(node)) for dv, node in re.findall(r"\[katamx numa\] device (\d+) -> node (\d+): pinned staging", log)}
    assert staged == {(dv, dv // 4) for dv in range(8)}, sorted(staged)
    games = int(log.split("Final games finished: ")[1].split()[0])
    rows = int(log.split("Final data rows: ")[1].split()[0])
    assert games >= 48
    per_size = shard_checks.check_shards(shard_checks.shard_files(os.path.join(d, "out")), 19, (9, 13, 19), rows)
    assert all(v > 0 for v in per_size.values()), per_size
    print("8 fake devices: launches per device", per_dev, "training rows by board size", per_size)

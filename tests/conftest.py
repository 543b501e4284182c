import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

REF = os.environ.get("KATAGO_REFERENCE", "/root/reference")
REF_BIN_DIR = os.path.join(REPO, "oracle", "_ref")  # test binaries bound to the CPU oracle, the reference's OpenCL backend, test nets
PRODUCT_BIN_DIR = os.path.join(REPO, "integration", "_build")  # the product binding: katago_hip, katago_hip_refeval (integration/Makefile)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~30 s on CPU")
    # The CPU suite (-m "not gpu") is dominated by emulated kernels and by the reference's own test commands run as subprocesses:
    # independent files, spread over a few pytest-xdist workers when the plugin is there and the caller did not choose (-n / -p
    # no:xdist / KMX_TEST_WORKERS=0). Never for -m gpu: those tests time the device and must have it to themselves.
    opt = config.option
    if (getattr(opt, "numprocesses", None) in (None, 0) and getattr(opt, "dist", None) == "no" and not hasattr(config, "workerinput")
            and "not gpu" in (opt.markexpr or "") and os.environ.get("KMX_TEST_WORKERS", "") != "0" and not getattr(opt, "collectonly", False)):
        # (round 6, 8 cores: 4 workers 8 min 45 s, 6 workers 5 min - the emulated kernels' tests are rendezvous-bound, not core-bound)
        n = min(int(os.environ.get("KMX_TEST_WORKERS", "6")), os.cpu_count() or 1)
        if n > 1:
            opt.numprocesses = n
            opt.dist = "loadfile"
            opt.tx = ["popen"] * n


# The CPU suite's long files, longest first: with the workers above a file is handed to whichever worker is free, in collection order - and the
# longest one (the mutated build of the emulated kernels + its runs, four minutes) came last in the alphabet, alone on the tail of the run.
LONGEST_FIRST = ["test_kernels_wrong_count.py", "test_engine_emulated.py", "test_nneval_own.py", "test_kernels_latest_completion.py",
                 "test_kernels_emulated_chain.py", "test_kernels_emulated_seam.py", "test_kernels_emulated_conv.py"]


def pytest_collection_modifyitems(config, items):
    if "not gpu" not in (config.option.markexpr or ""):
        return
    rank = {name: i for i, name in enumerate(LONGEST_FIRST)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(rank)))  # (stable: everything else keeps its order)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The HIP library and the oracle must exist; build them if a fresh checkout lacks them."""
    from katago_amd import build as kbuild

    kbuild.build(verbose=False)
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"], check=True)


@pytest.fixture(scope="session", autouse=True)
def _model_files_written_once(tmp_path_factory):
    """A random-weight model file is a pure function of (architecture, seed, options, file format): the second test that asks for the
    same one gets a copy of the first's file instead of 7 s (b18c384nbt) or 20 s (b28c512nbt) of gzip - a minute and a half of the GPU
    suite, which the driver stops at 20 minutes."""
    import json
    import shutil

    from katago_amd import modelgen

    real, kept = modelgen.write_model, {}
    store = str(tmp_path_factory.mktemp("kmx_model_files"))

    def write_model(path, arch, *args, **kw):
        if args:
            return real(path, arch, *args, **kw)
        suffix = next(s for s in (".bin.gz", ".txt.gz", ".bin", ".txt", "") if path.endswith(s))
        key = json.dumps([arch, suffix, sorted(kw.items())], sort_keys=True, default=str)
        if key not in kept:
            made = os.path.join(store, "%d%s" % (len(kept), suffix))
            kept[key] = (made, real(made, arch, **kw))
        shutil.copyfile(kept[key][0], path)
        return dict(kept[key][1]) if isinstance(kept[key][1], dict) else kept[key][1]

    modelgen.write_model = write_model
    yield
    modelgen.write_model = real


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("kmx_models"))


@pytest.fixture(scope="session")
def small_model(model_dir):
    from katago_amd import modelgen

    p = os.path.join(model_dir, "b3c64nbt.bin.gz")
    modelgen.write_model(p, "b3c64nbt", seed=123)
    return p


def ref_binary(name):
    """integration/_build/<name> (the product binding, katago_hip*) or oracle/_ref/<name> (everything bound to the oracle): prebuilt
    (GPU box) or built here from the mounted reference."""
    product = name.startswith("katago_hip")
    path = os.path.join(PRODUCT_BIN_DIR if product else REF_BIN_DIR, name)
    if not os.path.exists(path):
        if not os.path.isdir(os.path.join(REF, "cpp")):
            pytest.skip("%s not built and the reference tree is not mounted" % name)
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "integration" if product else "oracle"), "-j%d" % (os.cpu_count() or 4),
                        "all" if product else "ref", "REF=" + REF], check=True, stdout=subprocess.DEVNULL)
    return path


def make_rows(rng, n, L=19, sizes=None):
    """Binary V7-like feature planes on an LxL buffer; sizes[b] = (x_size, y_size) of the real board."""
    sp = np.zeros((n, L, L, 22), dtype=np.float32)
    for b in range(n):
        xs, ys = sizes[b] if sizes else (L, L)
        sp[b, :ys, :xs, 0] = 1
        st = rng.random((ys, xs))
        sp[b, :ys, :xs, 1] = st < 0.25
        sp[b, :ys, :xs, 2] = (st >= 0.25) & (st < 0.5)
        for c in range(3, 22):
            sp[b, :ys, :xs, c] = rng.random((ys, xs)) < 0.08
    gl = rng.normal(0, 0.5, (n, 19)).astype(np.float32)
    return sp.reshape(n, L * L, 22), gl

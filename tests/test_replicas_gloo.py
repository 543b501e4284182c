"""N>1 path on CPU: two gloo ranks, replicas only — shard arithmetic, barrier, max-time / total-rows
aggregation exactly as bench.py uses them. The per-rank 'evaluation' is the CPU oracle on a tiny net, so the test also
shows that shards evaluated by different ranks reassemble to the single-process result (no cross-rank dependence)."""
import os
import socket
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, make_rows

sys.path.insert(0, REPO)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, model_path, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from katago_amd import replicas
    from oracle import oracle

    n_rows = 7  # ragged on purpose: 4 + 3
    sp, gl = make_rows(np.random.default_rng(11), n_rows, 9)
    b, e = replicas.shard_rows(n_rows, rank, world)
    om = oracle.loadModelFile(model_path)
    replicas.barrier()
    res = oracle.getOutput(om, 9, 9, sp[b:e], gl[b:e], None, None, True, 1)
    seconds = 1.0 + rank  # deterministic stand-in for the measured time: the slowest rank must win
    rows, tmax = replicas.whole_job(e - b, seconds)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), policy=res["policy"], value=res["value"], begin=b, end=e, rows=rows, tmax=tmax)
    replicas.barrier()
    dist.destroy_process_group()


def test_two_replicas_gloo(tmp_path):
    from katago_amd import modelgen, replicas
    from oracle import oracle

    path = str(tmp_path / "tiny.bin")
    modelgen.write_model(path, "b2c32nbt", seed=5)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), path, str(tmp_path)), nprocs=world, join=True)
    sp, gl = make_rows(np.random.default_rng(11), 7, 9)
    ref = oracle.getOutput(oracle.loadModelFile(path), 9, 9, sp, gl, None, None, True, 1)
    covered = []
    for r in range(world):
        d = np.load(str(tmp_path / ("rank%d.npz" % r)))
        assert int(d["rows"]) == 7 and float(d["tmax"]) == 2.0  # SUM of rows, MAX of times, identical on every rank
        b, e = int(d["begin"]), int(d["end"])
        covered += list(range(b, e))
        np.testing.assert_array_equal(d["policy"], ref["policy"][b:e])
        np.testing.assert_array_equal(d["value"], ref["value"][b:e])
    assert covered == list(range(7))


def test_shard_rows_cover_exactly():
    from katago_amd import replicas

    for n in (0, 1, 5, 8, 255, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [replicas.shard_rows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1

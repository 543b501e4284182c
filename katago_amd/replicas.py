"""Multi-GPU = independent replicas: one process per GPU, every rank evaluates its own rows, there is
NO collective on the data path. This mirrors the reference's only multi-GPU mode — one NN server thread per GPU pulling
from a shared queue (cpp/neuralnet/nneval.cpp:399-407, gpuIdxByServerThread) — with processes instead of threads.
The only communication is the bookkeeping of a measurement: a barrier, the slowest rank's time, the total row count - over
whatever backend the caller initialised torch.distributed with; bench.py uses gloo (CPU tensors): no RCCL anywhere."""
import torch
import torch.distributed as dist


def shard_rows(n_rows, rank, world):
    """Contiguous shard [begin, end) of a global batch for `rank`; shards differ by at most one row and cover it exactly
    (how a front end would spread one big query list over the replicas)."""
    base, extra = divmod(n_rows, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def barrier(device=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def whole_job(rows_this_rank, seconds_this_rank, device=None):
    """(total rows over all ranks, max seconds over ranks): whole-job throughput = rows / seconds."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(rows_this_rank), float(seconds_this_rank)
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([seconds_this_rank], dtype=torch.float64, device=dev)
    r = torch.tensor([rows_this_rank], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    return int(r.item()), float(t.item())

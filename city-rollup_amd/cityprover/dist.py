"""Control plane for N>1 GPUs: one process per GPU, units (independent traces / proof jobs) are
sharded across ranks with NO data-path collective; torch.distributed is used
only for the barrier and the max-over-ranks of the timed region. Works with gloo (CPU tests) or
nccl (= RCCL) as the launcher prefers."""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend="gloo"):
    rank, local_rank, world = env_rank()
    if world == 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def shard_units(n_units, rank, world):
    """Round-robin by unit index, like N workers popping one shared job queue in order
    (city_rollup_core_worker/src/lib.rs:131-145)."""
    return list(range(rank, n_units, world))


def unit_seed(base_seed, unit):
    """Synthetic input seed of unit `unit` (independent of which rank processes it)."""
    return (base_seed + unit * 0x1000003) & 0xFFFFFFFFFFFFFFFF


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sum_over_ranks(dist, value):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def broadcast_str(dist, value, src=0):
    """rank `src`'s string on every rank (e.g. the private directory rank 0 wrote a fixture into)"""
    if dist is None:
        return value
    obj = [value]
    dist.broadcast_object_list(obj, src=src)
    return obj[0]

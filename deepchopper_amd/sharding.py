"""Read sharding across GPUs: reads are independent, so rank r of W takes one
contiguous range of the input and nothing is exchanged on the data path.  (The reference uses
Lightning's DistributedSampler, which interleaves and pads the last shard by repeating samples;
`chop` tolerates that because it keys predictions by read id -- src/bin/predict.rs:222-235.
Contiguous, non-repeating shards give the same set of predictions.)"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """[lo, hi) of rank's shard: sizes differ by at most one, earlier ranks get the longer ones."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def merge_by_id(parts: list[dict]) -> dict:
    """Host-side gather: union of per-rank {read_id: result} maps; duplicate ids must agree."""
    out: dict = {}
    for p in parts:
        for k, v in p.items():
            if k in out and out[k] != v:
                raise ValueError(f"conflicting results for read '{k}'")
            out[k] = v
    return out


def dist_max(value: float, device=None) -> float:
    """max over ranks of a host scalar (bench timing); identity when no process group is up."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dist_barrier() -> None:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def gather_results(local: dict) -> dict | None:
    """Host-side gather of per-rank {read_id: result} maps to rank 0 (control plane only: the
    device never exchanges data).  Returns the merged map on rank 0, None elsewhere."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local)
    parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local, parts, dst=0)
    return merge_by_id(parts) if dist.get_rank() == 0 else None

"""Multi-GPU seam of the ICP hot path: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

The path shards over TARGET POINTS: correspondence search and residual/Jacobian rows are independent per point, only
the 6x6 / 6x1 normal equations couple them.  Every rank holds the whole local map (a few MB) and a
contiguous slice of the scan; per ICP iteration it accumulates its packed normal equations (32 doubles: 21 upper
triangular JtJ, 6 Jtr, loss, |r|^2, row count, pad), the vectors are summed with ONE all-reduce of 256 bytes, and
every rank applies the identical 6x6 solve and pose update — so the poses stay bit-identical across ranks without
broadcasting them.  The message is latency-bound (256 B never approaches the ~153 GB/s of one xGMI link).

The reference has no multi-GPU code at all; this is the MI355X-side addition.
"""
from typing import Optional, Tuple

__all__ = ["shard_bounds", "sharded_register", "NEQ_SIZE"]

NEQ_SIZE = 32


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous (ring-major) slice [begin, end) of `n` target rows owned by `rank`; slices tile [0, n) exactly."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise AssertionError(f"bad rank {rank} / world size {world_size}")
    per = (n + world_size - 1) // world_size
    begin = min(n, rank * per)
    return begin, min(n, begin + per)


def sharded_register(engine, local_points, init_pose=None, iterations: Optional[int] = None, group=None,
                     skip_null: bool = False):
    """Registers one scan whose rows are spread over the ranks of `group`.

    `engine` is an `IcpContext` (or anything with the same five calls): register_begin / iteration_accumulate /
    normal_equations_tensor / iteration_solve / register_end.  `local_points` is this rank's slice of the scan.
    With world size 1 (or torch.distributed not initialised) the loop degenerates to the single-GPU registration.
    """
    import torch.distributed as dist
    use_dist = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if hasattr(engine, "use_torch_stream"):
        engine.use_torch_stream()  # accumulate / all-reduce / solve must share one stream: torch's current one
    neq = engine.normal_equations_tensor()
    iters = int(iterations if iterations is not None else engine.config.max_num_alignments)
    engine.register_begin(local_points, init_pose, skip_null=skip_null)
    for _ in range(iters):
        engine.iteration_accumulate()
        if use_dist:
            dist.all_reduce(neq, op=dist.ReduceOp.SUM, group=group)
        engine.iteration_solve()
    return engine.register_end()

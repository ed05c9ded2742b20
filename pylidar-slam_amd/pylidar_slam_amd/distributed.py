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

__all__ = ["shard_bounds", "sharded_register", "sharded_map_normals", "connect_exchange", "NEQ_SIZE"]

NEQ_SIZE = 32


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous (ring-major) slice [begin, end) of `n` target rows owned by `rank`; slices tile [0, n) exactly."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise AssertionError(f"bad rank {rank} / world size {world_size}")
    per = (n + world_size - 1) // world_size
    begin = min(n, rank * per)
    return begin, min(n, begin + per)


def connect_exchange(engine, group=None) -> bool:
    """Switches `engine` (an `IcpContext`) to the in-library exchange: the IPC handles of the per-rank inboxes are
    all-gathered over `group` once, after which `engine.register(local_points, ...)` alone registers a scan that is
    sharded over the ranks — per ICP iteration ONE kernel sums this rank's rows, writes them into every peer's inbox
    over xGMI, waits for the others and solves (`k_sum_exchange_solve`), with no collective call and no host step in the
    loop.  Returns False (and changes nothing) when torch.distributed is not initialised or the world size is 1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    handle = engine.exchange_create(rank, world)
    handles = [None] * world
    dist.all_gather_object(handles, handle, group=group)
    engine.exchange_connect(handles)
    dist.barrier(group)  # every inbox is mapped everywhere before the first registration writes into one
    return True


def sharded_register(engine, local_points, init_pose=None, iterations: Optional[int] = None, group=None,
                     skip_null: bool = False):
    """Registers one scan whose rows are spread over the ranks of `group` with a torch.distributed all-reduce (RCCL
    with backend "nccl") per iteration — the portable driver; `connect_exchange` + `engine.register` is the one that
    keeps the whole loop on the devices.

    `engine` is an `IcpContext` (or anything with the same five calls): register_begin / iteration_accumulate /
    normal_equations_tensor / iteration_solve / register_end.  `local_points` is this rank's slice of the scan.
    Without a process group the loop degenerates to the single-GPU registration (a group of one rank still all-reduces).
    """
    import torch.distributed as dist
    # (a process group of ONE rank still issues its collective: that is how the 1-GPU test box exercises RCCL)
    use_dist = dist.is_available() and dist.is_initialized()
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


def sharded_map_normals(engine, group=None):
    """Map-sharded normal estimation after a map update (BASELINE configs[3]): every rank estimates the
    normals of the map points whose spatial bucket it owns (`icp_map_normals_owned`), ONE all-reduce sums the arrays
    over the ranks by original map index (16 B per map point: 16 MB for a 1M-point map, i.e. 2 MB per rank and ring
    step — bandwidth-bound on the ~153 GB/s xGMI links, ~0.1 ms), and every rank installs the full set into its own
    normal cache.  Every index has exactly one owner, so the sum is exact and all ranks end with identical normals.
    Without a process group (or with one rank) this is the eager single-GPU estimation."""
    import torch.distributed as dist
    use_dist = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if use_dist else 0
    world = dist.get_world_size(group) if use_dist else 1
    if hasattr(engine, "set_option"):
        # this driver estimates the map's normals, shard by shard: the library must not ALSO estimate all of them on
        # every rank behind each map update (its schedule for maps of up to 2^20 points)
        engine.set_option("eager_normals_limit", 0)
    shard = engine.map_normals_owned(rank, world)
    if use_dist:
        dist.all_reduce(shard, op=dist.ReduceOp.SUM, group=group)
    engine.map_normals_install(shard)
    return shard

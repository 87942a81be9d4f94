"""Multi-GPU helpers: one process per GPU, independent proofs per rank, proving key broadcast once.

The path shards by construction: proof i depends only on its own inputs plus the read-only key,
so there is no data-path collective; the only exchange is one broadcast of the key blob at load time
(RCCL over xGMI on the GPU box; gloo in the CPU tests).
"""


def shard_range(total, rank, world):
    """Contiguous block of batch indices owned by `rank`: sizes differ by at most one, union is range(total)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_blob(dist, blob, src=0, device="cpu"):
    """Broadcast a bytes object from `src` to every rank; returns bytes on all ranks."""
    import torch
    rank = dist.get_rank()
    if rank == src:
        data = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
        size = torch.tensor([data.numel()], dtype=torch.int64, device=device)
    else:
        size = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(size, src)
    if rank != src:
        data = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(data, src)
    return bytes(data.cpu().numpy().tobytes())


def gather_blobs(dist, blob, device="cpu"):
    """all_gather of equal-length bytes objects: the list of every rank's blob, in rank order, on every rank."""
    import torch
    mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [bytes(p.cpu().numpy().tobytes()) for p in parts]


def msm_g1_sharded(dist, n, partial_fn, sum_fn, device="cpu"):
    """One large G1 MSM over the GPUs of a node (SURVEY 8e for BASELINE.json configs[4]: "shard the 2^24 points 8-way, gather 8
    partial sums"): rank r computes the partial sum of its contiguous share of the points -- partial_fn(lo, hi) -> 64 B, on the GPU:
    Context.msm_g1_pippenger_bench_shard -- the world_size partial sums (64 B each: the ONLY exchange, one all_gather over RCCL / xGMI)
    are added on every rank by sum_fn(list of 64 B) -> 64 B (on the GPU: Context.msm_g1 with unit scalars).  Without a process group
    (dist None) it is the plain MSM."""
    if dist is None:
        return partial_fn(0, n)
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(n, rank, world)
    part = partial_fn(lo, hi) if hi > lo else bytes(64)
    return sum_fn(gather_blobs(dist, part, device))

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

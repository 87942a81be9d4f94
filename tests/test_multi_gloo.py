"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- key blob broadcast from rank 0, disjoint shards."""
import os
import sys
import tempfile

import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, blob_path, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
    import torch.distributed as dist
    from spp.multi import broadcast_blob, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = open(blob_path, "rb").read() if rank == 0 else None
    got = broadcast_blob(dist, blob, 0, "cpu")
    lo, hi = shard_range(1024, rank, world)
    with open(os.path.join(out_dir, "r%d" % rank), "wb") as f:
        f.write(got)
    with open(os.path.join(out_dir, "s%d" % rank), "w") as f:
        f.write("%d %d" % (lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_key_broadcast_and_sharding_world2(withdraw_artifacts):
    d = tempfile.mkdtemp()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, withdraw_artifacts["pk"], d), nprocs=2, join=True)
    ref = open(withdraw_artifacts["pk"], "rb").read()
    assert open(os.path.join(d, "r0"), "rb").read() == ref
    assert open(os.path.join(d, "r1"), "rb").read() == ref
    s0 = tuple(map(int, open(os.path.join(d, "s0")).read().split()))
    s1 = tuple(map(int, open(os.path.join(d, "s1")).read().split()))
    assert s0 == (0, 512) and s1 == (512, 1024)

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


def _run_bench(extra, env_extra):
    import json
    import subprocess
    env = dict(os.environ, SPP_BENCH_DRYRUN="1", SPP_BENCH_BACKEND="gloo", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_launcher_starts_n_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` with no launcher around it: bench.py itself starts 2 rank processes (before importing
    torch), they rendezvous (gloo here, nccl = RCCL on the GPU box), broadcast the key blob, take their shard, time the
    steps between barriers with MAX over ranks, and rank 0 prints one line claiming n_gpus = 2.  Dry run: the proving is
    replaced by a sleep (there is no GPU in this container); the rest is the code path of the real run."""
    j = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8"], {})
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3 and j["config"]["dry_run"] is True
    assert j["config"]["proofs_per_step_all_gpus"] == 16 and j["config"]["parallelism"] == "independent proofs x2"
    assert abs(j["value"] - 16 / (j["ms_per_step"] * 1e-3)) / j["value"] < 0.01


def test_bench_strong_mode_splits_a_fixed_batch():
    """BASELINE.json configs[2]: a FIXED batch of 1024 audit proofs over the GPUs of a node -- strong scaling, contiguous
    blocks from spp.multi.shard_range; with 3 ranks the blocks are 342 + 341 + 341."""
    j = _run_bench(["--gpus", "3", "--steps", "2", "--mode", "strong", "--total", "1024"], {})
    assert j["n_gpus"] == 3 and j["scaling"] == "strong" and j["config"]["proofs_per_step_all_gpus"] == 1024
    j1 = _run_bench(["--gpus", "1", "--steps", "2", "--mode", "strong", "--total", "64"], {})
    assert j1["n_gpus"] == 1 and j1["config"]["proofs_per_step_all_gpus"] == 64


def test_bench_under_an_external_launcher():
    """The driver's way: torch.distributed.run sets RANK / WORLD_SIZE and starts bench.py once per rank; bench.py must not
    spawn again."""
    import json
    import subprocess
    port = 29500 + ((os.getpid() + 777) % 2000)
    env = dict(os.environ, SPP_BENCH_DRYRUN="1", SPP_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 2


import pytest


@pytest.mark.gpu
def test_bench_two_ranks_strong_mode_on_one_gpu(tmp_path):
    """The N > 1 path of bench.py with real proving: two rank processes started by bench.py itself, gloo for the collectives
    (both ranks sit on the one GPU of the box, which RCCL refuses), each proving its contiguous block of a fixed 64-proof audit
    batch with small tables; one line, n_gpus = 2, scaling = strong."""
    import json
    import subprocess
    env = dict(os.environ, SPP_BENCH_BACKEND="gloo", SPP_FORCE_DEVICE="0", SPP_TABLE_BUDGET_GB="40")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", "strong", "--total", "64",
                          "--no-host-leg"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["proofs_per_step_all_gpus"] == 64 and j["config"]["batch_per_gpu"] == 32
    assert j["config"]["pk_bcast_ms"] > 0 and j["config"]["circuit"] == "audit" and j["value"] > 0


@pytest.mark.gpu
def test_bench_sharded_msm_two_ranks_on_one_gpu():
    """BASELINE.json configs[4] cut over ranks with real kernels: two rank processes on the one GPU (gloo for the gather),
    each computes the partial sum of its half of a 2^18-point MSM, the two 64-byte partials are gathered and added; the result is
    the one-rank MSM of the same points."""
    import json
    import subprocess
    outs = []
    for gpus in ("2", "1"):
        env = dict(os.environ, SPP_BENCH_BACKEND="gloo", SPP_FORCE_DEVICE="0")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--msm-shard-leg-only", "--msm-points", str(1 << 18)],
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        outs.append(json.loads(lines[0])["msm_g1_sharded"])
    assert outs[0]["n_gpus"] == 2 and outs[1]["n_gpus"] == 1 and outs[0]["result_hex"] == outs[1]["result_hex"] and len(outs[0]["result_hex"]) == 128


@pytest.mark.gpu
def test_bench_single_rank_over_rccl(tmp_path):
    """RCCL itself (backend nccl) with one rank: rendezvous, the key broadcast, barriers and the reductions of bench.py run on
    the GPU through RCCL -- the multi-rank code path minus the peers, which a one-GPU box cannot supply."""
    import json
    import subprocess
    env = dict(os.environ, SPP_BENCH_FORCE_DIST="1", SPP_TABLE_BUDGET_GB="40", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64", "--circuit", "withdraw",
                          "--no-extras", "--no-cpu-baseline", "--no-single", "--no-host-leg"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["pk_bcast_backend"] == "nccl" and j["config"]["pk_bcast_ms"] > 0


def _msm_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
    sys.path.insert(0, ROOT)
    import random
    import torch.distributed as dist
    from spp.multi import msm_g1_sharded
    from oracle import bn254 as B
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 37                                        # not a multiple of the world size
    rng = random.Random(11)
    pts = [B.g1_mul(B.G1_GEN, rng.randrange(1, B.R)) for _ in range(n)]
    sc = [rng.randrange(B.R) for _ in range(n)]

    def partial(lo, hi):                           # the CPU oracle stands in for Context.msm_g1_pippenger_bench_shard
        acc = None
        for i in range(lo, hi):
            acc = B.g1_add(acc, B.g1_mul(pts[i], sc[i]))
        return B.g1_to_bytes(acc)

    def total(parts):
        acc = None
        for b in parts:
            acc = B.g1_add(acc, B.g1_from_bytes(b))
        return B.g1_to_bytes(acc)
    got = msm_g1_sharded(dist, n, partial, total)
    with open(os.path.join(out_dir, "m%d" % rank), "wb") as f:
        f.write(got + partial(0, n))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_msm_gathers_the_partial_sums_world2():
    """SURVEY 8e for BASELINE.json configs[4]: the points of ONE large MSM cut into contiguous shares, one partial sum per rank, one
    all_gather of 64-byte points, the sum on every rank (spp/multi.py msm_g1_sharded; gloo here, RCCL on the GPU node): every rank ends
    with the MSM of all points."""
    d = tempfile.mkdtemp()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_msm_worker, args=(2, port, d), nprocs=2, join=True)
    for r in range(2):
        blob = open(os.path.join(d, "m%d" % r), "rb").read()
        assert len(blob) == 128 and blob[:64] == blob[64:]

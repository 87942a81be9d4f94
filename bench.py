#!/usr/bin/env python3
"""Headline benchmark: Groth16 proofs/sec on MI355X (BASELINE.json metric; headline circuit = audit_circuit, the one the
target names).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--circuit audit|withdraw] [--batch B] [--mode weak|strong] [--total T]

One "step" = one batch of independent proofs, every row a DIFFERENT witness (spp/workload.py: audit rows sk_i = 12345 + i,
Random(1000 + i) as SURVEY 8d Config 3; withdraw rows = distinct notes of one tree), inputs already resident in HBM, proofs
(388 B) and public witnesses written to HBM.

N > 1: one process per GPU.  Launched by the driver through torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or, when WORLD_SIZE is absent, by this script itself: `python bench.py --gpus N` starts N rank processes
BEFORE anything touches the GPU and waits for them.  Rank 0 runs the trusted setup, the proving key is broadcast once over
RCCL (timed as pk_bcast_ms, outside the metric), and every rank proves its own rows with no data-path collective.
  --mode weak   (default) B proofs per GPU per step                      -> "scaling": "weak"
  --mode strong a fixed total of T proofs per step (default 1024 = BASELINE.json configs[2]) cut into contiguous blocks with
                spp.multi.shard_range                                    -> "scaling": "strong"
Rank 0 prints ONE JSON line.  At N = 1 the line also carries the withdraw circuit (own shape, the reference's R1CS size, the
depth-20 variant), the RLWE witness kernel on 2^16 instances (configs[3]) and the 2^24-point Pippenger MSM (configs[4]).
`roofline` prices the dominant kernel (k_msm_flat<G1>, the table walk of kernels_msm.hip) against the HBM peak from a serialised probe (three steps on one
stream after the timed region, durations from the dispatches' own timestamps) -- in the pipelined timed region the same
dispatches share the chip with the other batch and last longer; both figures are in the line.  `cpu_baseline` times the
oracle's C/OpenMP prover on a bounded sample of the same rows on this host (reported baseline, not the target).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch / HIP initialise: see spp/lib.py (streams of unrelated batches must not share a queue)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SETS = ["CB", "A", "B1", "K", "Z", "CS", "B2(G2)"]          # launch order of spp_msm_kernel_ms
INFO_ORDER = ["A", "B1", "K", "Z", "CB", "CS", "B2(G2)"]    # order of spp_circuit_msm_sizes / _windows
WORKLOADS = {"withdraw": "noir_circuit withdraw (Poseidon Merkle depth 16 + Grumpkin)",
             "withdraw_refshape": "noir_circuit withdraw padded with ballast multiplications to the reference's gnark R1CS size (12 452 constraints, 2^14)",
             "withdraw_depth20": "withdraw statement over a depth-20 Poseidon tree (synthetic variant; the reference is depth 16)",
             "withdraw_acir": "noir_circuit withdraw, R1CS compiled from the reference's own ACIR (noir_circuit/target/shielded_pool_verifier.json via spp compile)",
             "audit": "audit_circuit (RLWE, const-PK; scripts/generate_audit.py:246-465)"}
DEFAULT_BATCH = {"withdraw": 4096, "audit": 2048, "withdraw_refshape": 2048, "withdraw_depth20": 2048, "withdraw_acir": 4096}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)    # SURVEY 8d timing rule: warm-up 3, >= 10 iterations or >= 2 s
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU per step in weak mode (0 = 2048 audit / 4096 withdraw)")
    ap.add_argument("--circuit", default=os.environ.get("SPP_BENCH_CIRCUIT", "audit"), choices=["withdraw", "audit", "withdraw_acir"])
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"])
    ap.add_argument("--total", type=int, default=1024, help="strong mode: proofs per step over all GPUs (BASELINE.json configs[2]: 1024)")
    ap.add_argument("--window", type=int, default=int(os.environ.get("SPP_WINDOW", "0")), help="MSM window bits; 0 = auto (largest tables within the HBM budget)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline circuit only (profiling runs)")
    ap.add_argument("--no-other-circuits", action="store_true", help="headline circuit with its own legs (rehearsal, end to end), none of the other circuits / configs")
    ap.add_argument("--msm-shard-leg", action="store_true", help="N > 1: also run the 2^24-point MSM cut over the GPUs (BASELINE configs[4], SURVEY 8e)")
    ap.add_argument("--msm-shard-leg-only", action="store_true", help="only that leg")
    ap.add_argument("--msm-points", type=int, default=1 << 24)
    ap.add_argument("--pair-leg-only", action="store_true", help="diagnostic: only the leg with both circuits resident")
    ap.add_argument("--ccs-leg-only", action="store_true", help="diagnostic: only the leg that proves the reference's own gnark R1CS")
    ap.add_argument("--no-single", action="store_true", help="skip the single-proof latency leg (keeps profiler per-kernel averages clean)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg: its chunks overlap on two workspaces and would "
                                                                "stretch a profiler's per-kernel averages")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters, nothing in this process has
    touched the GPU), one device each, and return the worst exit code.  Rank 0 inherits stdout and prints the JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank stuck in a rendezvous must not hold the run until the driver's own limit: SPP_BENCH_RANK_TIMEOUT_S (default 1500)
    # for the whole run; on expiry -- or as soon as one rank has failed and the others have had a minute to notice -- the
    # remaining children (fresh processes started above, nothing else) are terminated, then killed, and the exit code is 124
    limit = float(os.environ.get("SPP_BENCH_RANK_TIMEOUT_S", "1500"))
    t_end = time.time() + limit
    rc, failed_at = 0, None
    while any(p.poll() is None for p in procs):
        now = time.time()
        for p in procs:
            if p.returncode not in (None, 0) and failed_at is None:
                failed_at, rc = now, p.returncode
        if now > t_end or (failed_at is not None and now - failed_at > 60.0):
            print("bench.py: rank processes still running after %s; terminating them" % ("%.0f s" % limit if now > t_end else "a rank failed"), file=sys.stderr)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10.0
            while time.time() < t_kill and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            return rc or 124
        time.sleep(0.2)
    for p in procs:
        rc = rc or p.returncode
    return rc


def libspp_sha256():
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd", "libspp.so"), "rb").read()).hexdigest()


def dry_run(args, rank, world, dist, coll_dev):
    """SPP_BENCH_DRYRUN=1 (CPU tests of the launcher / rendezvous / sharding logic, tests/test_multi_gloo.py): everything
    bench.py does around the GPU work -- key broadcast, shard ranges, barrier-bracketed timing with MAX over ranks, one JSON
    line from rank 0 -- with the proving itself replaced by a sleep."""
    import torch
    from spp.multi import broadcast_blob, shard_range
    blob = bytes(range(256)) * 64
    if dist:
        blob = broadcast_blob(dist, blob if rank == 0 else None, 0, coll_dev)
    assert len(blob) == 256 * 64 and blob[:4] == bytes([0, 1, 2, 3])
    per = args.batch or 8
    lo, hi = shard_range(args.total, rank, world) if args.mode == "strong" else (rank * per, (rank + 1) * per)
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.0005 * (hi - lo))
    if dist:
        dist.barrier()
    el = time.perf_counter() - t0
    total = hi - lo
    if dist:
        t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        c = torch.tensor([hi - lo], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total = int(c.item())
    if rank == 0:
        print(json.dumps({"metric": "Groth16 proofs/sec", "value": round(total * args.steps / el, 3), "unit": "proofs/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": args.mode, "vs_baseline": None, "dtype": "u32x8 (254-bit Montgomery integers)", "data": "synthetic",
                          "config": {"workload": "DRY RUN (no GPU work): launcher / rendezvous / sharding rehearsal", "dry_run": True,
                                     "proofs_per_step_all_gpus": total, "parallelism": "independent proofs x%d" % world}}), flush=True)


def rlwe_leg(ctx, dev, pk, iters=200):
    """BASELINE.json configs[3]: RLWE negacyclic witness generation, 2^16 instances on one GPU (scripts/generate_audit.py:507-584)."""
    import ctypes
    import numpy as np
    import torch
    import spp
    from oracle import native
    cnt = 1 << 16
    g = torch.Generator(device="cpu").manual_seed(4)
    a = torch.tensor(pk["a"], dtype=torch.int32, device=dev)
    b = torch.tensor(pk["b"], dtype=torch.int32, device=dev)
    r = torch.randint(-3, 4, (cnt, 1024), generator=g, dtype=torch.int8).to(dev)
    e1 = torch.randint(-3, 4, (cnt, 64), generator=g, dtype=torch.int8).to(dev)
    e2 = torch.randint(-3, 4, (cnt, 1024), generator=g, dtype=torch.int8).to(dev)
    msg = torch.randint(0, 256, (cnt, 64), generator=g, dtype=torch.uint8).to(dev)
    c0 = torch.zeros((cnt, 64), dtype=torch.int32, device=dev)
    c1 = torch.zeros((cnt, 1024), dtype=torch.int32, device=dev)
    k0 = torch.zeros((cnt, 64), dtype=torch.int32, device=dev)
    k1 = torch.zeros((cnt, 1024), dtype=torch.int32, device=dev)
    packed = torch.zeros((cnt, 157 * 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    L = ctx.L

    def run():
        spp.lib.check(L.spp_rlwe_witness_batch_device(ctx.h, a.data_ptr(), b.data_ptr(), cnt, r.data_ptr(), e1.data_ptr(), e2.data_ptr(),
                                                      msg.data_ptr(), c0.data_ptr(), c1.data_ptr(), k0.data_ptr(), k1.data_ptr(), packed.data_ptr()))
    for _ in range(3):
        run()
    spp.lib.check(L.spp_ctx_sync(ctx.h))
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    spp.lib.check(L.spp_ctx_sync(ctx.h))
    ms = (time.perf_counter() - t0) / iters * 1e3
    # parity spot check of the timed outputs against the oracle (C, scalar) + its time as the CPU figure
    sel = [0, 1, 4097, cnt - 1]
    an, bn = np.array(pk["a"], dtype=np.uint32), np.array(pk["b"], dtype=np.uint32)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    t0 = time.perf_counter()
    for i in sel:
        oc0 = np.zeros(64, dtype=np.uint32); oc1 = np.zeros(1024, dtype=np.uint32)
        ok0 = np.zeros(64, dtype=np.int64); ok1 = np.zeros(1024, dtype=np.int64)
        native.lib().orc_rlwe_witness(p(an), p(bn), p(r[i].cpu().numpy().astype(np.int32)), p(e1[i].cpu().numpy().astype(np.int32)),
                                      p(e2[i].cpu().numpy().astype(np.int32)), p(msg[i].cpu().numpy().astype(np.uint32)), p(oc0), p(oc1), p(ok0), p(ok1))
        assert (c1[i].cpu().numpy().astype(np.uint32) == oc1).all() and (k1[i].cpu().numpy() == ok1).all(), "RLWE instance %d differs from the oracle" % i
        assert (c0[i].cpu().numpy().astype(np.uint32) == oc0).all() and (k0[i].cpu().numpy() == ok0).all()
    cpu_s = (time.perf_counter() - t0) / len(sel)
    alg = 21504.0 * cnt + 8192     # 21 504 B per instance + the public key once
    traffic = None
    try:   # PMC passes of this leg, committed under profiles/ (rocprofv3 --pmc cannot run inside this process)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "round2_rlwe_pmc_hbm.json")))
        if pmc.get("instances") == cnt:
            traffic = pmc["hbm_bytes_per_launch"]
    except Exception:
        pass
    return {"metric": "RLWE witness instances/sec", "value": round(cnt / (ms * 1e-3), 1), "unit": "instances/s", "iters": iters,
            "config": {"workload": "2^16 RLWE instances: two negacyclic n=1024 products, 1088 exact quotients, 7x32-bit packing (BASELINE configs[3])"},
            "ms_per_batch": round(ms, 3),
            "roofline": {"bound": "hbm", "kernel": "k_rlwe_witness", "achieved": round(alg / (ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic, "alg_bytes_per_launch": int(alg)},
            "cpu_baseline": {"value": round(1.0 / cpu_s, 2), "unit": "instances/s", "cores": 1, "kind": "port",
                             "sample": "%d instances, oracle C schoolbook (the reference's CPython path: 1.4 instances/s/core, BASELINE.md)" % len(sel)}}


def pippenger_leg(ctx, iters=10):
    """BASELINE.json configs[4]: 2^24-point BN254 G1 MSM over general bases (Pippenger), uniform and witness-like scalars."""
    import ctypes
    import random
    from oracle import native, bn254 as B
    n = 1 << 24
    res, ms, ms_bucket = ctx.msm_g1_pippenger_bench(n, seed=5, iters=iters)
    res_w, ms_w, ms_bucket_w = ctx.msm_g1_pippenger_bench(n, seed=5, iters=iters, small_permille=700)
    # linearity (size-independent property at the full size): MSM(3 * scalars) == 3 * MSM(scalars)
    res3, _, _ = ctx.msm_g1_pippenger_bench(n, seed=5, scale=3, iters=1)
    assert B.g1_to_bytes(B.g1_mul(B.g1_from_bytes(res), 3)) == res3, "Pippenger 2^24 is not linear in the scalars"
    alg = 96.0 * n
    ns = 1 << 16
    rng = random.Random(1)
    bases = B.g1_to_bytes(B.g1_mul(B.G1_GEN, 12345)) * ns
    sc = b"".join(rng.randrange(B.R).to_bytes(32, "big") for _ in range(ns))
    out = ctypes.create_string_buffer(64)
    t0 = time.perf_counter()
    native.lib().orc_msm_g1(bases, sc, ns, ctypes.cast(out, ctypes.c_void_p))
    cpu_s = time.perf_counter() - t0
    traffic = None
    try:   # PMC passes of this leg (profiles/run_pippenger_profile.sh), valid for the library they were taken on
        pmc = json.load(open(os.path.join(ROOT, "profiles", "round3_pippenger_pmc_hbm.json")))
        if pmc.get("points") == n and pmc.get("libspp_sha256") == libspp_sha256():
            traffic = pmc["hbm_bytes_per_msm"]
    except Exception:
        pass
    return {"metric": "G1 MSM points/sec (Pippenger, general bases)", "value": round(n / (ms * 1e-3), 1), "unit": "points/s", "iters": iters,
            "config": {"workload": "2^24-point BN254 G1 MSM, uniform 253-bit scalars, bases k_i*G generated on device (BASELINE configs[4])"},
            "ms_per_msm": round(ms, 3),
            "witness_like_70pct_small": {"ms_per_msm": round(ms_w, 3), "points_per_s": round(n / (ms_w * 1e-3), 1), "bucket_kernel_ms": round(ms_bucket_w, 3)},
            "roofline": {"bound": "hbm", "kernel": "whole MSM (sort + bucket accumulation + reduction)", "achieved": round(alg / (ms * 1e-3) / 1e9, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "alg_bytes_per_launch": int(alg), "bucket_kernel_ms": round(ms_bucket, 3)},
            "cpu_baseline": {"value": round(ns / cpu_s, 1), "unit": "points/s", "cores": native.max_threads(), "kind": "port",
                             "sample": "2^16-point MSM, oracle C Pippenger"}}


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))        # nothing above this line imports torch or touches HIP

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    dist = None
    # rehearsal switches (one-GPU boxes / CPU tests): SPP_BENCH_BACKEND=gloo runs the collectives on CPU tensors and
    # SPP_FORCE_DEVICE=k puts every rank on GPU k; the real multi-GPU run uses neither (nccl = RCCL, one GPU per rank)
    backend = os.environ.get("SPP_BENCH_BACKEND", "nccl")
    dry = os.environ.get("SPP_BENCH_DRYRUN") == "1"
    if "SPP_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SPP_FORCE_DEVICE"])
    # SPP_BENCH_FORCE_DIST=1: create the process group even for one rank (exercises the RCCL path -- rendezvous, key broadcast,
    # barriers, the MAX / SUM reductions -- on a single-GPU box)
    if world > 1 or os.environ.get("SPP_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if dry:
        dry_run(args, rank, world, dist, coll_dev)
        if dist is not None:
            dist.destroy_process_group()
        return
    dev = torch.device("cuda", local_rank)

    import spp
    from spp import workload
    from spp.multi import broadcast_blob, shard_range
    from oracle import native
    rlwe_pk = json.load(open(os.path.join(ROOT, "tests", "golden", "rlwe_pk.json")))

    def run_circuit(circuit, B, steps, warmup, want_cpu, strong_total=0):
        """One circuit: setup, load, distinct rows, timed steps, checks.  Returns the result dict on rank 0, None elsewhere.
        strong_total > 0: this rank proves rows shard_range(strong_total, rank, world) of the fixed batch."""
        tmp = tempfile.mkdtemp(prefix="spp_bench_%d_" % rank)
        sppc, pkp, vkp = (os.path.join(tmp, "c." + e) for e in ("sppc", "pk", "vk"))
        if circuit == "withdraw_acir":
            from spp import acir
            acir.compile_to_sppc(os.path.join(ROOT, "tests", "golden", "reference_withdraw_acir.json"), sppc)
        else:
            cid = {"withdraw": 1, "audit": 2, "withdraw_refshape": 3, "withdraw_depth20": 4}[circuit]
            spp.build_circuit(cid, sppc, aux=(list(rlwe_pk["a"]) + list(rlwe_pk["b"])) if cid == 2 else None)
        ctx = spp.Context(local_rank)
        # ---- proving key: GPU setup on rank 0, one RCCL broadcast over xGMI ----
        t0 = time.time()
        bcast_ms = 0.0
        if rank == 0:
            ctx.setup(sppc, b"\x2a" * 32, pkp, vkp)
        setup_s = time.time() - t0
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            tb = time.time()
            blob = broadcast_blob(dist, open(pkp, "rb").read() if rank == 0 else None, 0, coll_dev)   # RCCL over xGMI
            torch.cuda.synchronize()
            bcast_ms = (time.time() - tb) * 1e3
            if rank != 0:
                open(pkp, "wb").write(blob)
            pk_bytes = len(blob)
            del blob
        else:
            pk_bytes = os.path.getsize(pkp)
        t0 = time.time()
        h = ctx.load_circuit(sppc, pkp, args.window)
        load_s = time.time() - t0

        # ---- this rank's rows: all distinct, built with the HIP witness-input kernels, resident in HBM ----
        if strong_total:
            lo, hi = shard_range(strong_total, rank, world)
        else:
            lo, hi = rank * B, (rank + 1) * B
        B = hi - lo
        t0 = time.time()
        if circuit == "withdraw_depth20":
            rows_b = workload.withdraw_rows(ctx, B, seed=20 + rank, depth=20)
        elif circuit.startswith("withdraw"):
            rows_b = workload.withdraw_rows(ctx, B, seed=2 + rank)
        else:
            rows_b = workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], B, first=lo)     # proofs lo .. hi-1 of SURVEY 8d Config 3
        rows_s = time.time() - t0
        n_distinct = len({rows_b[32 * h.n_inputs * i + 32:32 * h.n_inputs * (i + 1)] for i in range(B)})
        inp = torch.frombuffer(bytearray(rows_b), dtype=torch.uint8).to(dev)
        rs_bytes = b"".join((1000003 * (lo + i) + 17).to_bytes(32, "big") + (998244353 * (lo + i) + 29).to_bytes(32, "big") for i in range(B))
        rs = torch.frombuffer(bytearray(rs_bytes), dtype=torch.uint8).to(dev)
        # two output sets: consecutive batches are pipelined on two streams inside libspp
        proofs = [torch.zeros(B * 388, dtype=torch.uint8, device=dev) for _ in range(2)]
        pws = [torch.zeros(B * h.pw_len, dtype=torch.uint8, device=dev) for _ in range(2)]
        status = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        step_no = [0]

        def step():
            k = step_no[0] & 1
            step_no[0] += 1
            h.prove_batch_device(B, inp.data_ptr(), rs.data_ptr(), proofs[k].data_ptr(), pws[k].data_ptr(), status[k].data_ptr())

        for _ in range(warmup):
            step()
        h.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        acc = {"stage": [0.0] * 7, "n": 0, "kern": [0.0] * 7}

        def take(which):
            tm = h.last_timings(which)
            km = h.msm_kernel_ms(which)
            acc["n"] += 1
            for i in range(7):
                acc["stage"][i] += tm[i]
                acc["kern"][i] += km[i]

        for it in range(steps):
            step()
            if it >= 1:
                take(1)   # events of the previous step; the step just enqueued keeps the GPU busy
        h.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        total_proofs = B
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            cnt = torch.tensor([B], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            total_proofs = int(cnt.item())
        assert int(status[0].abs().sum().item()) + int(status[1].abs().sum().item()) == 0, "some synthetic proofs were refused"
        take(0)           # events of the last timed step (already complete)

        if rank != 0:
            h.close()
            ctx.close()
            return None

        # ---- serialised roofline probe: one stream, one batch in flight, so a dispatch's duration is its own ----
        h.set_serial(True)
        probe = {"kern": [0.0] * 7, "stage": [0.0] * 7, "n": 0}
        for _ in range(3):
            step()
            h.sync()
            km, tm = h.msm_kernel_ms(0), h.last_timings(0)
            probe["n"] += 1
            for i in range(7):
                probe["kern"][i] += km[i]
                probe["stage"][i] += tm[i]
        h.set_serial(False)

        # ---- BASELINE.json configs[2] rehearsed on ONE GPU: 1 024 audit proofs over 8 GPUs are 128 per rank and step.  The same
        # handle proves the first 128 rows of this rank's batch, pipelined exactly like the headline (libspp keeps up to six
        # such batches in flight); its rate beside the 2 048-proof rate is what one rank of a strong-scaling run delivers.
        def pipelined_rate(Bs, n_steps, n_warm):
            outs = [(torch.zeros(Bs * 388, dtype=torch.uint8, device=dev), torch.zeros(Bs * h.pw_len, dtype=torch.uint8, device=dev),
                     torch.zeros(Bs, dtype=torch.int32, device=dev)) for _ in range(8)]   # more than libspp keeps in flight (6)
            def go(i):
                pr, pw, st_ = outs[i & 7]
                h.prove_batch_device(Bs, inp.data_ptr(), rs.data_ptr(), pr.data_ptr(), pw.data_ptr(), st_.data_ptr())
            for i in range(n_warm):
                go(i)
            h.sync(); torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(n_steps):
                go(n_warm + i)
            h.sync(); torch.cuda.synchronize()
            el = time.perf_counter() - ts
            assert sum(int(o[2].abs().sum().item()) for o in outs) == 0
            assert bytes(outs[0][0].cpu().numpy()) == bytes(outs[1][0].cpu().numpy()), "the batches in flight disagree"
            return Bs * n_steps / el, el / n_steps * 1e3

        rehearsal = None
        if circuit == "audit" and not strong_total and world == 1 and not args.no_extras and B >= 1024:
            rate128, ms128 = pipelined_rate(128, 48, 8)
            rehearsal = {"what": "one of 8 ranks of `--mode strong --total 1024` (BASELINE.json configs[2]): 128 audit proofs per step, pipelined",
                         "batch": 128, "value": round(rate128, 1), "unit": "proofs/s", "ms_per_step": round(ms128, 3), "steps": 48, "warmup": 8,
                         "fraction_of_the_%d_proof_rate" % B: round(rate128 / (total_proofs * steps / elapsed), 4),
                         "note": "predicted per-GPU efficiency of the literal strong-scaling config; weak scaling (2 048 per GPU) has no such loss"}

        # ---- end to end: (sk, r, e1, e2) -> proof.  Everything scripts/generate_audit.py:468-641 computes before `nargo execute`
        # (keygen, wa_commitment, RLWE encryption + quotients, packing, ct_commitment) runs on the device INSIDE the clock, each
        # step from the raw secrets of its batch; the rows it produces are checked against the ones the headline used.
        e2e = None
        if circuit == "audit" and not strong_total and world == 1 and not args.no_extras:
            import numpy as np
            from spp.lib import check as spp_check
            sks, r8, e18, e28 = workload.audit_noise(lo, B)
            up = lambda raw: torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
            d_a, d_b = up(np.asarray(rlwe_pk["a"], dtype=np.uint32).tobytes()), up(np.asarray(rlwe_pk["b"], dtype=np.uint32).tobytes())
            d_sk = up(b"".join(int(v).to_bytes(32, "big") for v in sks))
            d_r, d_e1, d_e2 = up(r8.tobytes()), up(e18.tobytes()), up(e28.tobytes())
            d_rows = torch.zeros(B * h.n_inputs * 32, dtype=torch.uint8, device=dev)
            # the stand-alone input pipeline reproduces the rows the headline used (the same kernels run inside the fused call)
            spp_check(ctx.L.spp_audit_inputs_batch_device(ctx.h, d_a.data_ptr(), d_b.data_ptr(), B, d_sk.data_ptr(), d_r.data_ptr(),
                                                          d_e1.data_ptr(), d_e2.data_ptr(), d_rows.data_ptr()))
            assert torch.equal(d_rows, inp), "device-built rows differ from the headline's rows"
            del d_rows

            def e2e_step(i):
                k = i & 1
                h.prove_audit_from_secrets_device(B, d_a.data_ptr(), d_b.data_ptr(), d_sk.data_ptr(), d_r.data_ptr(), d_e1.data_ptr(),
                                                  d_e2.data_ptr(), rs.data_ptr(), proofs[k].data_ptr(), pws[k].data_ptr(), status[k].data_ptr())
            ref_proofs = proofs[(step_no[0] - 1) & 1].clone()   # bytes of the last batch proved from precomputed rows
            for i in range(2):
                e2e_step(i)
            h.sync(); torch.cuda.synchronize()
            assert torch.equal(proofs[0], ref_proofs) and torch.equal(proofs[1], ref_proofs), "proofs from secrets differ from proofs from rows"
            ts = time.perf_counter()
            n_e2e = max(6, steps)
            for i in range(n_e2e):
                e2e_step(i)
            h.sync(); torch.cuda.synchronize()
            el = time.perf_counter() - ts
            assert int(status[0].abs().sum().item()) + int(status[1].abs().sum().item()) == 0
            e2e = {"what": "clock starts at (secret_key, r, e1, e2) resident in HBM: spp_prove_audit_from_secrets_device per step (keygen, wa_commitment, "
                           "RLWE encryption + quotients, packing, ct_commitment on the proving stream in front of the solver)",
                   "value": round(B * n_e2e / el, 1), "unit": "proofs/s", "ms_per_step": round(el / n_e2e * 1e3, 3), "steps": n_e2e,
                   "rows_equal_the_headline_rows": True, "proof_bytes_equal_the_proofs_from_rows": True}

        # the same rows through the host-buffer entry point (spp_prove_batch: H2D of the inputs, D2H of proofs / public
        # witnesses, synchronous): the PCIe-inclusive rate, reported beside `value`, never as `value`
        import ctypes
        host_rate = None
        if not args.no_host_leg:
            HB = 4 * B                       # one call with four batches' worth: libspp cuts it into chunks and pipelines them
            in_host, rs_host = rows_b * 4, rs_bytes * 4
            ph = ctypes.create_string_buffer(388 * HB)
            wh = ctypes.create_string_buffer(h.pw_len * HB)
            sh = (ctypes.c_int32 * HB)()
            args_h = (h.h, HB, in_host, rs_host, ctypes.cast(ph, ctypes.c_void_p), ctypes.cast(wh, ctypes.c_void_p), ctypes.cast(sh, ctypes.c_void_p))
            assert h.L.spp_prove_batch(*args_h) == 0
            assert ph.raw[388 * (HB - 1):388 * HB] == ph.raw[388 * (B - 1):388 * B]      # same inputs, same blinding, same bytes
            th = time.perf_counter()
            assert h.L.spp_prove_batch(*args_h) == 0
            host_rate = HB / (time.perf_counter() - th)
            del in_host, rs_host

        # SURVEY 8d Config 1/2: ONE proof from the reference's own inputs, end to end on the device-resident entry point
        single = None
        if circuit in ("withdraw", "audit", "withdraw_acir") and not args.no_single and world == 1:
            if circuit != "audit":
                from oracle import circuit as OC
                kat_row = OC.withdraw_inputs(json.load(open(os.path.join(ROOT, "tests", "golden", "withdraw_kat.json"))))
                kat_name = "client/prover-params.toml (tests/golden/withdraw_kat.json)"
            else:   # the reference's own run: sk = 12345, Random(999) (scripts/generate_audit.py:469-470)
                kat_row = workload.row_ints(workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], 1, first=0, seed_base=999), h.n_inputs, 0)
                kat_name = "scripts/generate_audit.py defaults: sk = 12345, Random(999), demo rlwe_pk.json"
            one_in = torch.frombuffer(bytearray(b"".join(int(v).to_bytes(32, "big") for v in kat_row)), dtype=torch.uint8).to(dev)
            # full-size blinding factors, as a real prover draws them (the s*Ar / r*Bs1 work depends on their length)
            one_r = 0x1f3a9c0de4b5a697887766554433221100ffeeddccbbaa998877665544332211 % (1 << 253)
            one_s = 0x0e2d4c6b8a79685746352413021f0e0dccbbaa99887766554433221100fedcba
            one_rs = torch.frombuffer(bytearray(one_r.to_bytes(32, "big") + one_s.to_bytes(32, "big")), dtype=torch.uint8).to(dev)
            one_pr = torch.zeros(388, dtype=torch.uint8, device=dev)
            one_pw = torch.zeros(h.pw_len, dtype=torch.uint8, device=dev)
            one_st = torch.zeros(1, dtype=torch.int32, device=dev)
            lat = []
            for _ in range(12):
                torch.cuda.synchronize()
                tl = time.perf_counter()
                h.prove_batch_device(1, one_in.data_ptr(), one_rs.data_ptr(), one_pr.data_ptr(), one_pw.data_ptr(), one_st.data_ptr())
                h.sync()
                lat.append((time.perf_counter() - tl) * 1e3)
            assert int(one_st.item()) == 0
            assert spp.verify(open(vkp, "rb").read(), one_pr.cpu().numpy().tobytes(), one_pw.cpu().numpy().tobytes())
            lat = sorted(lat[2:])
            single = {"inputs": kat_name, "latency_ms_median": round(lat[len(lat) // 2], 3), "latency_ms_min": round(lat[0], 3),
                      "proofs_per_s_at_batch_1": round(1e3 / lat[len(lat) // 2], 2), "blinding": "full-size r, s",
                      "tables": "the throughput handle above (window_bits = 0: one table row per base, a pass per window, Horner combine)"}
            # the same proof on the tables the drop-in helpers load (generateProof / generateAuditProof: window_bits = 8, one row per
            # window, no passes, ~6 GB): the latency configuration
            try:
                h8 = ctx.load_circuit(sppc, pkp, 8)
                lat8 = []
                for _ in range(12):
                    torch.cuda.synchronize()
                    tl = time.perf_counter()
                    h8.prove_batch_device(1, one_in.data_ptr(), one_rs.data_ptr(), one_pr.data_ptr(), one_pw.data_ptr(), one_st.data_ptr())
                    h8.sync()
                    lat8.append((time.perf_counter() - tl) * 1e3)
                assert int(one_st.item()) == 0
                lat8 = sorted(lat8[2:])
                single["latency_ms_median_8bit_tables"] = round(lat8[len(lat8) // 2], 3)
                single["table_bytes_8bit"] = h8.table_bytes
                h8.close()
            except spp.SppError as e:   # not enough HBM left beside the big tables
                single["latency_ms_median_8bit_tables"] = None
                single["latency_8bit_note"] = str(e)

        # the timed batches produced real proofs: every proof of the last pipelined batch through the batched GPU verifier,
        # two of them also through the host verifier
        last = (warmup + steps - 1) & 1
        pbytes, wbytes = proofs[last].cpu().numpy().tobytes(), pws[last].cpu().numpy().tobytes()
        vkb = open(vkp, "rb").read()
        for i in (0, B - 1):
            assert spp.verify(vkb, pbytes[388 * i:388 * (i + 1)], wbytes[h.pw_len * i:h.pw_len * (i + 1)]), "proof %d does not verify" % i
        all_ok = ctx.verify_batch(vkb, [pbytes[388 * i:388 * (i + 1)] for i in range(B)], [wbytes[h.pw_len * i:h.pw_len * (i + 1)] for i in range(B)])
        assert all(all_ok), "%d proofs of the last timed batch do not verify" % (B - sum(all_ok))

        value = total_proofs * steps / elapsed
        ms_per_step = elapsed / steps * 1e3
        sizes = dict(zip(INFO_ORDER, h.msm_sizes()))
        windows = dict(zip(INFO_ORDER, h.msm_windows()))
        # algorithmic bytes of one k_msm_flat launch over a batch (64 B per base once + one 32 B scalar per (base, proof))
        alg = {s: (128 if s.startswith("B2") else 64) * sizes[s] + 32 * sizes[s] * B for s in SETS}
        g1 = ["A", "B1", "K", "Z"]     # the four k_msm_flat<Fq> launches of a step (CB / CS: small row-per-window tables, k_msm_rows)
        ser = {s: probe["kern"][i] / probe["n"] for i, s in enumerate(SETS)}
        pip = {s: acc["kern"][i] / acc["n"] for i, s in enumerate(SETS)}
        ser_g1_ms = sum(ser[s] for s in g1)
        alg_g1 = sum(alg[s] for s in g1)
        achieved = alg_g1 / (ser_g1_ms * 1e-3) / 1e9 if ser_g1_ms > 0 else 0.0
        traffic, traffic_src, valu_util = None, None, None
        try:   # PMC passes of the same workload, committed under profiles/ (rocprofv3 --pmc cannot run inside this process)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")))
            # ... and only of THIS library: the file records the sha256 of the libspp.so the counters were taken on, so a kernel
            # change cannot inherit an older kernel's traffic
            if pmc.get("circuit") == circuit and pmc.get("batch") == B and pmc.get("n_distinct_witnesses") == B \
                    and pmc.get("libspp_sha256") == libspp_sha256():
                traffic = pmc["k_msm_flat_g1_hbm_bytes_per_launch"]
                traffic_src = pmc.get("source")
                valu_util = pmc.get("k_msm_flat_g1_valu_issue_util_serialised")
        except Exception:
            pass
        out = {
            "value": round(value, 3), "ms_per_step": round(ms_per_step, 3), "steps": steps, "warmup": warmup,
            "config": {"workload": "%s, batch of %d independent proofs per GPU per step, every row a distinct witness" % (WORKLOADS[circuit], B),
                       "circuit": circuit, "n_constraints": h.n_constraints, "n_wires": h.n_wires, "domain": 1 << h.domain_log,
                       "batch_per_gpu": B, "n_distinct_witnesses": n_distinct, "proofs_per_step_all_gpus": total_proofs,
                       "msm_windows": windows, "msm_sizes": sizes, "table_bytes": h.table_bytes, "parallelism": "independent proofs x%d" % world,
                       "pk_bcast_ms": round(bcast_ms, 3), "pk_bcast_backend": (backend if dist is not None else None), "pk_bytes": pk_bytes, "setup_s": round(setup_s, 2), "load_s": round(load_s, 2),
                       "rows_synth_s": round(rows_s, 2), "host_buffer_entry_proofs_per_s": None if host_rate is None else round(host_rate, 1),
                       "last_timed_batch_verified": "all %d proofs accepted by spp_verify_batch (GPU), two of them also by spp_verify (host)" % B,
                       "host_buffer_entry_note": "one spp_prove_batch call with host pointers for 4 batches' worth of proofs: PCIe copies included, chunks pipelined inside libspp"},
            "stage_ms_per_step_pipelined": {k: round(v / acc["n"], 3) for k, v in zip(
                ["solve+commit", "matrix_eval", "ntt_qap", "msm_g1", "msm_g2_side_stream_join", "assemble", "total"], acc["stage"])},
            "stage_ms_per_step_serialised": {k: round(v / probe["n"], 3) for k, v in zip(
                ["solve+commit", "matrix_eval", "ntt_qap", "msm_g1", "msm_g2", "assemble", "total"], probe["stage"])},
            "roofline": {"bound": "hbm", "kernel": "k_msm_flat<Fq>", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": int(alg_g1 / 4), "avg_launch_ms": round(ser_g1_ms / 4, 4), "launches_per_step": 4,
                         "launch_ms_serialised": {s: round(ser[s], 3) for s in SETS},
                         "launch_ms_in_pipelined_timed_region": {s: round(pip[s], 3) for s in SETS},
                         "launch_GBps_serialised": {s: round(alg[s] / (ser[s] * 1e-3) / 1e9, 2) if ser[s] > 0 else None for s in SETS},
                         "timing": "dispatch timestamps (hipExtLaunchKernelGGL events); `achieved` uses the serialised probe (3 steps on one stream after "
                                   "the timed region): 4 launches x avg_launch_ms <= ms_per_step; in the pipelined region the same dispatches share the chip",
                         "valu_issue_util_pmc": valu_util,
                         "msm_table_rows": dict(zip(INFO_ORDER, h.msm_table_rows())),
                         "note": "integer-VALU bound (about 2.3K instructions, 1.5K of them v_mad_u64_u32, per mixed addition; 16-17 additions per "
                                 "full-size scalar with single-row 15/16-bit tables); HBM fraction of the algorithmic bytes reported as mandated"},
        }
        if single is not None:
            out["single_proof"] = single
        if rehearsal is not None:
            out["strong_scaling_rank_rehearsal"] = rehearsal
        if e2e is not None:
            out["audit_end_to_end_from_secrets"] = e2e
        if want_cpu:
            orc = native.Prover(sppc, pkp)
            cores = native.max_threads()
            sample = [workload.row_ints(rows_b, h.n_inputs, i) for i in range(min(B, cores))]
            # throughput mode: one proof per host thread, rounds of `cores` proofs until ~10 s have elapsed
            t1 = time.perf_counter()
            done = 0
            while done == 0 or (time.perf_counter() - t1 < 10.0 and done < 4096):
                batch = [sample[(done + i) % len(sample)] for i in range(cores)]
                rc, _, _ = native.prove_many(orc, batch, [(7 + done + i, 11 + done + i) for i in range(cores)])
                assert rc == 0
                done += cores
            dt = time.perf_counter() - t1
            t2 = time.perf_counter()
            for i in range(3):
                assert orc.prove(sample[i % len(sample)], 3 + i, 4 + i)[0] == 0
            one_ms = (time.perf_counter() - t2) / 3 * 1e3
            out["cpu_baseline"] = {"value": round(done / dt, 3), "unit": "proofs/s", "cores": cores, "kind": "port",
                                   "single_proof_latency_ms": round(one_ms, 1),
                                   "sample": "%d %s proofs (rows of the timed batch), one per host thread, oracle C/OpenMP prover (stands in for the "
                                             "Sunspot Go/CPU path, which cannot run here: no Go toolchain, no pk)" % (done, circuit)}
        h.close()
        ctx.close()
        del inp, rs, proofs, pws, status
        torch.cuda.empty_cache()
        return out

    strong = args.mode == "strong"
    def reference_r1cs_leg(B=4096, steps=12, warmup=3):
        """the reference's OWN gnark constraint system (tests/golden/reference_withdraw.ccs = noir_circuit/target/
        shielded_pool_verifier.ccs) solved AND proved on the GPU: the container spp/ccs.py to_sppc_solved writes takes the 26 withdraw
        inputs of client/proof.helper.ts:34-50; the 6 163 other ACIR witnesses (`nargo execute` in the reference) and the 6 749
        internal wires (gnark's solver with its nine hints) are computed by the device solver inside the timed region.  B distinct
        notes per step, rows resident in HBM when the clock starts, as for every other leg."""
        from spp import ccs, acir
        golden = os.path.join(ROOT, "tests", "golden")
        ccs_path, acir_path = os.path.join(golden, "reference_withdraw.ccs"), os.path.join(golden, "reference_withdraw_acir.json")
        tmp = tempfile.mkdtemp(prefix="spp_bench_ccs_")
        sppc, pkp, vkp = (os.path.join(tmp, "c." + e) for e in ("sppc", "pk", "vk"))
        c = ccs.load_ccs(ccs_path)
        system = ccs.decode_system(c)
        n_rows, solved = ccs.to_sppc_solved(system, c, acir.load_program(acir_path), sppc)
        ctx = spp.Context(local_rank)
        ctx.setup(sppc, b"\x2a" * 32, pkp, vkp)
        t0 = time.time()
        h = ctx.load_circuit(sppc, pkp, args.window)
        load_s = time.time() - t0
        assert h.n_inputs == 26
        notes = workload.withdraw_rows(ctx, B, seed=91)               # distinct notes of one tree; keys below 2^128
        inp = torch.frombuffer(bytearray(notes), dtype=torch.uint8).to(dev)
        import random as _r
        rng = _r.Random(4)
        R_ = 21888242871839275222246405745257275088548364400416034343698204186575808495617
        rs = torch.frombuffer(bytearray(b"".join(rng.randrange(R_).to_bytes(32, "big") + rng.randrange(R_).to_bytes(32, "big") for _ in range(B))),
                              dtype=torch.uint8).to(dev)
        outs = [(torch.zeros(388 * B, dtype=torch.uint8, device=dev), torch.zeros(h.pw_len * B, dtype=torch.uint8, device=dev),
                 torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(3)]      # libspp keeps three batches of this circuit in flight

        def step(k):
            pr, pw, st = outs[k % 3]
            h.prove_batch_device(B, inp.data_ptr(), rs.data_ptr(), pr.data_ptr(), pw.data_ptr(), st.data_ptr())
        for k in range(warmup):
            step(k)
        h.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            step(k)
        h.sync(); torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        stages = h.last_timings(0)
        pr, pw, st = outs[(warmup + steps - 1) % 3]
        assert int(st.abs().sum().item()) == 0
        pb, wb = pr.cpu().numpy().tobytes(), pw.cpu().numpy().tobytes()
        vkb = open(vkp, "rb").read()
        assert all(ctx.verify_batch(vkb, [pb[388 * i:388 * (i + 1)] for i in range(B)], [wb[h.pw_len * i:h.pw_len * (i + 1)] for i in range(B)]))
        assert spp.verify(vkb, pb[:388], wb[:h.pw_len])
        lat = []
        for _ in range(8):
            torch.cuda.synchronize(); tl = time.perf_counter()
            h.prove_batch_device(1, inp.data_ptr(), rs.data_ptr(), pr.data_ptr(), pw.data_ptr(), st.data_ptr()); h.sync()
            lat.append((time.perf_counter() - tl) * 1e3)
        res = {"value": round(B * steps / elapsed, 1), "unit": "proofs/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup,
               "config": {"workload": "the reference's own gnark R1CS (noir_circuit/target/shielded_pool_verifier.ccs decoded by spp/ccs.py: 12 452 rows, "
                                      "41 hint calls), batch of %d distinct notes; the clock starts at the 26 withdraw inputs resident in HBM: all 12 939 "
                                      "wires are solved on the device (the reference runs `nargo execute` + gnark's solver for them)" % B,
                          "n_constraints": n_rows, "n_wires": system.n_wires, "domain": 1 << 14, "batch_per_gpu": B, "n_inputs": 26,
                          "solver_program_words": len(solved.program), "load_s": round(load_s, 2),
                          "msm_windows": dict(zip(INFO_ORDER, h.msm_windows())), "table_bytes": h.table_bytes,
                          "stage_ms_last_step": {k: round(v, 3) for k, v in zip(["solve+commit", "matrix_eval", "ntt_qap", "msm_g1", "join", "assemble", "total"], stages[:7])},
                          "verified": "all %d proofs of the last batch by spp_verify_batch, one by spp_verify" % B},
               "single_proof_latency_ms": round(sorted(lat[2:])[len(lat[2:]) // 2], 3)}
        h.close(); ctx.close()
        torch.cuda.empty_cache()
        return res

    def coresident_leg(Bp=2048, steps=10, warmup=3, budget=228e9):
        """One GPU serving BOTH circuits: the reference's relayer submits an audit proof and a withdraw proof per withdrawal
        (demo-frontend/app/api/relay/withdraw/route.ts:238-276).  The windows of both circuits' MSM sets are planned under ONE budget
        (spp_plan_windows: the greedy chooser over the union of their sets), both handles stay loaded, a step = Bp audit proofs + Bp
        withdraw proofs (distinct rows), their batches alternating on the handles' proving streams; value = pairs/s."""
        tmp = tempfile.mkdtemp(prefix="spp_bench_pair_")
        paths = {}
        ctx = spp.Context(local_rank)
        for name, cid in (("audit", 2), ("withdraw", 1)):
            sppc, pkp, vkp = (os.path.join(tmp, name + "." + e) for e in ("sppc", "pk", "vk"))
            spp.build_circuit(cid, sppc, aux=(list(rlwe_pk["a"]) + list(rlwe_pk["b"])) if cid == 2 else None)
            ctx.setup(sppc, b"\x2a" * 32, pkp, vkp)
            paths[name] = (sppc, pkp, vkp)
        plans = ctx.plan_windows([paths["audit"][1], paths["withdraw"][1]], budget)
        t0 = time.time()
        hs = {"audit": ctx.load_circuit(paths["audit"][0], paths["audit"][1], bits=plans[0]),
              "withdraw": ctx.load_circuit(paths["withdraw"][0], paths["withdraw"][1], bits=plans[1])}
        load_s = time.time() - t0
        rows = {"audit": workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], Bp), "withdraw": workload.withdraw_rows(ctx, Bp, seed=31)}
        up = lambda raw: torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        rs_b = b"".join((1000003 * i + 17).to_bytes(32, "big") + (998244353 * i + 29).to_bytes(32, "big") for i in range(Bp))
        d_rs = up(rs_b)
        d_in = {k: up(v) for k, v in rows.items()}
        outs = {k: [(torch.zeros(388 * Bp, dtype=torch.uint8, device=dev), torch.zeros(h.pw_len * Bp, dtype=torch.uint8, device=dev),
                     torch.zeros(Bp, dtype=torch.int32, device=dev)) for _ in range(2)] for k, h in hs.items()}

        def step(i):
            for k in ("audit", "withdraw"):
                pr, pw, st_ = outs[k][i & 1]
                hs[k].prove_batch_device(Bp, d_in[k].data_ptr(), d_rs.data_ptr(), pr.data_ptr(), pw.data_ptr(), st_.data_ptr())
        def sync():
            for h in hs.values():
                h.sync()
            torch.cuda.synchronize()
        for i in range(warmup):
            step(i)
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        sync()
        elapsed = time.perf_counter() - t0
        verified = {}
        for k, h in hs.items():
            pr, pw, st_ = outs[k][(warmup + steps - 1) & 1]
            assert int(st_.abs().sum().item()) == 0
            pb, wb = pr.cpu().numpy().tobytes(), pw.cpu().numpy().tobytes()
            ok = ctx.verify_batch(open(paths[k][2], "rb").read(), [pb[388 * i:388 * (i + 1)] for i in range(Bp)],
                                  [wb[h.pw_len * i:h.pw_len * (i + 1)] for i in range(Bp)])
            assert all(ok), "%s: %d proofs of the last batch do not verify" % (k, Bp - sum(ok))
            verified[k] = Bp
        # serialised probe per handle (one stream each, one after the other): the dispatches' own durations
        alg_total, ms_total, per = 0.0, 0.0, {}
        for k, h in hs.items():
            h.set_serial(True)
            km = None
            for i in range(2):
                pr, pw, st_ = outs[k][i & 1]
                h.prove_batch_device(Bp, d_in[k].data_ptr(), d_rs.data_ptr(), pr.data_ptr(), pw.data_ptr(), st_.data_ptr())
                h.sync()
                km = h.msm_kernel_ms(0)
            h.set_serial(False)
            sizes = dict(zip(INFO_ORDER, h.msm_sizes()))
            alg = sum(64 * sizes[s_] + 32 * sizes[s_] * Bp for s_ in SETS[:6])
            ms = sum(km[:6])
            alg_total += alg
            ms_total += ms
            per[k] = {"msm_windows": dict(zip(INFO_ORDER, h.msm_windows())), "table_bytes": h.table_bytes, "g1_msm_kernels_ms_serialised": round(ms, 3),
                      "alg_bytes_g1_msm": int(alg)}
        achieved = alg_total / (ms_total * 1e-3) / 1e9
        res = {"metric": "pairs (one audit proof + one withdraw proof) per second", "value": round(Bp * steps / elapsed, 1), "unit": "pairs/s",
               "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup,
               "config": {"workload": "audit_circuit and noir_circuit withdraw CO-RESIDENT on one GPU, %d distinct rows of each per step" % Bp,
                          "pairs_per_step": Bp, "table_budget_bytes": budget, "table_bytes_total": sum(p_["table_bytes"] for p_ in per.values()),
                          "circuits": per, "load_s": round(load_s, 2), "verified": "all proofs of the last batch of both circuits by spp_verify_batch"},
               "roofline": {"bound": "hbm", "kernel": "k_msm_flat<Fq>", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                            "note": "algorithmic bytes of the 12 G1 table walks of a pair of batches / their serialised durations"}}
        for h in hs.values():
            h.close()
        ctx.close()
        torch.cuda.empty_cache()
        return res

    def sharded_msm_leg(npts=1 << 24, iters=10):
        """BASELINE.json configs[4] on N GPUs: the 2^24 points cut into contiguous shares, one partial sum per rank, ONE
        all_gather of 64-byte points over RCCL, the sum on every rank (spp/multi.py msm_g1_sharded).  Opt-in (--msm-shard-leg): the
        default multi-GPU run is the proving benchmark alone."""
        from spp.multi import msm_g1_sharded
        ctxm = spp.Context(local_rank)
        it_ms = {}

        def partial(lo, hi):
            outb, ms, _ = ctxm.msm_g1_pippenger_bench_shard(npts, lo, hi - lo, seed=5, iters=iters)
            it_ms["ms"] = ms
            return outb
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        tg = time.perf_counter()
        total_pt = msm_g1_sharded(dist, npts, partial, lambda parts: ctxm.msm_g1(b"".join(parts), [1] * len(parts)), coll_dev)
        torch.cuda.synchronize()
        slowest = it_ms.get("ms", 0.0)
        if dist is not None:
            dist.barrier()
            tms = torch.tensor([slowest], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tms, op=dist.ReduceOp.MAX)
            slowest = float(tms.item())
        wall = time.perf_counter() - tg
        ctxm.close()
        return {"metric": "G1 MSM points/sec (Pippenger, general bases), points cut over the GPUs", "n_gpus": world,
                "ms_per_msm_slowest_rank": round(slowest, 3), "value": round(npts / (slowest * 1e-3), 1) if slowest else None,
                "unit": "points/s", "iters": iters, "points": npts, "result_hex": total_pt.hex(),
                "exchange": "one all_gather of %d x 64 B partial sums" % world, "wall_s_including_base_generation": round(wall, 2)}

    if args.msm_shard_leg_only:
        leg = sharded_msm_leg(npts=args.msm_points)
        if rank == 0:
            print(json.dumps({"msm_g1_sharded": leg}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.pair_leg_only:
        print(json.dumps({"audit_plus_withdraw_coresident": coresident_leg()}), flush=True)
        return
    if args.ccs_leg_only:
        print(json.dumps({"withdraw_reference_gnark_r1cs": reference_r1cs_leg()}), flush=True)
        return
    B0 = args.batch or DEFAULT_BATCH[args.circuit]
    main_res = run_circuit(args.circuit, B0, args.steps, args.warmup, not args.no_cpu_baseline and world == 1, args.total if strong else 0)
    extras = {}
    if world == 1 and not args.no_extras and not args.no_other_circuits and not strong:
        # the other circuits / configs of BASELINE.json, each at >= 10 timed steps, in the same line so that the driver times them
        want = not args.no_cpu_baseline
        other = "withdraw" if args.circuit == "audit" else "audit"
        extras[other + "_circuit"] = run_circuit(other, DEFAULT_BATCH[other], 10, 3, want)
        extras["withdraw_at_reference_r1cs_size"] = run_circuit("withdraw_refshape", DEFAULT_BATCH["withdraw_refshape"], 10, 3, False)
        extras["withdraw_depth20_variant"] = run_circuit("withdraw_depth20", DEFAULT_BATCH["withdraw_depth20"], 10, 3, False)
        extras["withdraw_compiled_from_reference_acir"] = run_circuit("withdraw_acir", DEFAULT_BATCH["withdraw_acir"], 10, 3, False)
        try:     # an auxiliary leg with a host-side process pool: a failure here must not cost the headline line
            extras["withdraw_reference_gnark_r1cs"] = reference_r1cs_leg()
        except Exception as e:   # noqa: BLE001
            extras["withdraw_reference_gnark_r1cs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            extras["audit_plus_withdraw_coresident"] = coresident_leg()
        except Exception as e:   # noqa: BLE001
            extras["audit_plus_withdraw_coresident"] = {"error": "%s: %s" % (type(e).__name__, e)}
        ctx = spp.Context(local_rank)
        extras["rlwe_witness_2p16"] = rlwe_leg(ctx, dev, rlwe_pk)
        extras["msm_g1_2p24"] = pippenger_leg(ctx)
        ctx.close()
    if world > 1 and args.msm_shard_leg and not strong:
        leg = sharded_msm_leg()
        if rank == 0:
            extras["msm_g1_2p24_sharded"] = leg
    if rank == 0:
        line = {"metric": "Groth16 proofs/sec", "value": main_res["value"], "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": args.mode,
                "vs_baseline": None, "dtype": "u32x8 (254-bit Montgomery integers)", "data": "synthetic"}
        for k in ("config", "stage_ms_per_step_pipelined", "stage_ms_per_step_serialised", "roofline", "cpu_baseline", "single_proof",
                  "strong_scaling_rank_rehearsal", "audit_end_to_end_from_secrets"):
            if k in main_res:
                line[k] = main_res[k]
        line.update(extras)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

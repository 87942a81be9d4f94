#!/usr/bin/env python3
"""Headline benchmark: Groth16 proofs/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--circuit withdraw|audit] [--window C]

One "step" = one batch of B independent proofs of the circuit, inputs already resident in HBM, proofs
(388 B) and public witnesses written to HBM.  N > 1: one process per GPU (torch.distributed / RCCL); the
proving key is produced on rank 0 and broadcast once over RCCL (timed separately, excluded from the metric);
each rank then proves its own B proofs with no data-path collective ("weak" scaling: B per GPU fixed).
Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (k_msm_fixed<G1>) against HBM peak with
live HIP-event timings; `cpu_baseline` times the oracle's C/OpenMP prover on a bounded sample of the same
workload on this host (reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)    # SURVEY 8d timing rule: warm-up 3, >= 10 iterations or >= 2 s
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU per step (0 = 4096 for withdraw, 2048 for audit)")
    ap.add_argument("--circuit", default=os.environ.get("SPP_BENCH_CIRCUIT", "withdraw"), choices=["withdraw", "audit"])
    ap.add_argument("--window", type=int, default=int(os.environ.get("SPP_WINDOW", "0")), help="MSM window bits; 0 = auto (largest tables within the HBM budget)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second circuit")
    ap.add_argument("--no-single", action="store_true", help="skip the single-proof latency leg (keeps profiler per-kernel averages clean)")
    ap.add_argument("--no-refshape", action="store_true", help="skip the withdraw circuit padded to the reference's R1CS size")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # rehearsal switches (one-GPU boxes): SPP_BENCH_BACKEND=gloo runs the collectives on CPU tensors and
    # SPP_FORCE_DEVICE=k puts every rank on GPU k; the real multi-GPU run uses neither (nccl = RCCL, one GPU per rank)
    backend = os.environ.get("SPP_BENCH_BACKEND", "nccl")
    if "SPP_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SPP_FORCE_DEVICE"])
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    import spp
    from oracle import native

    def run_circuit(circuit, B, steps, warmup, want_cpu):
        """Returns the result dict for one circuit (rank 0) or None (other ranks)."""
        tmp = tempfile.mkdtemp(prefix="spp_bench_%d_" % rank)
        sppc, pkp, vkp = (os.path.join(tmp, "c." + e) for e in ("sppc", "pk", "vk"))
        if circuit == "withdraw":
            spp.build_circuit(1, sppc)
        elif circuit == "withdraw_refshape":
            spp.build_circuit(3, sppc)       # same statement, padded to the reference's gnark R1CS size (12 452 constraints, 2^14)
        elif circuit == "withdraw_depth20":
            spp.build_circuit(4, sppc)       # SURVEY 8d Config 2's synthetic variant: a depth-20 tree (the reference is depth 16)
        else:
            pk = json.load(open(os.path.join(ROOT, "tests", "golden", "rlwe_pk.json")))
            spp.build_circuit(2, sppc, aux=list(pk["a"]) + list(pk["b"]))
        ctx = spp.Context(local_rank)
        # ---- proving key: GPU setup on rank 0, one RCCL broadcast over xGMI ----
        t0 = time.time()
        bcast_ms = 0.0
        if rank == 0:
            ctx.setup(sppc, b"\x2a" * 32, pkp, vkp)
        setup_s = time.time() - t0
        if world > 1:
            from spp.multi import broadcast_blob
            torch.cuda.synchronize()
            dist.barrier()
            tb = time.time()
            blob = broadcast_blob(dist, open(pkp, "rb").read() if rank == 0 else None, 0, coll_dev)   # RCCL over xGMI
            torch.cuda.synchronize()
            bcast_ms = (time.time() - tb) * 1e3
            if rank != 0:
                open(pkp, "wb").write(blob)
            del blob
        t0 = time.time()
        h = ctx.load_circuit(sppc, pkp, args.window)
        load_s = time.time() - t0

        # ---- synthetic batch of DISTINCT rows (spp/workload.py, built with the HIP witness-input kernels), resident in HBM ----
        from spp import workload
        if circuit == "withdraw_depth20":
            rows_b = workload.withdraw_rows(ctx, B, seed=20 + rank, depth=20)
        elif circuit.startswith("withdraw"):
            rows_b = workload.withdraw_rows(ctx, B, seed=2 + rank)
        else:
            rows_b = workload.audit_rows(ctx, pk["a"], pk["b"], B, first=rank * B)
        rows = [workload.row_ints(rows_b, h.n_inputs, i) for i in range(min(B, 256))]   # sample for the CPU baseline leg
        inp = torch.frombuffer(bytearray(rows_b), dtype=torch.uint8).to(dev)
        rs_bytes = b"".join((1000003 * (rank * B + i) + 17).to_bytes(32, "big") + (998244353 * (rank * B + i) + 29).to_bytes(32, "big")
                            for i in range(B))
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
        acc = {"kern_ms": 0.0, "kern_n": 0, "stage": [0.0] * 7}

        def take(tm):
            acc["kern_ms"] += tm[7] * tm[8]
            acc["kern_n"] += int(tm[8])
            for i in range(7):
                acc["stage"][i] += tm[i]

        for it in range(steps):
            step()
            if it >= 1:
                take(h.last_timings(1))   # HIP events of the previous step; the step just enqueued keeps the GPU busy
        h.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert int(status[0].abs().sum().item()) + int(status[1].abs().sum().item()) == 0, "some synthetic proofs were refused"
        take(h.last_timings(0))           # events of the last timed step (already complete)

        # the same batch through the host-buffer entry point (spp_prove_batch: H2D of the inputs, D2H of proofs/public
        # witnesses, synchronous): the PCIe-inclusive rate, reported beside `value`, never as `value`
        host_rate = None
        if rank == 0:
            import ctypes
            HB = 4 * B                       # one call with four batches' worth: libspp cuts it into chunks and pipelines them
            in_host = bytes(inp.cpu().numpy().tobytes()) * 4
            rs_host = bytes(rs.cpu().numpy().tobytes()) * 4
            ph = ctypes.create_string_buffer(388 * HB)
            wh = ctypes.create_string_buffer(h.pw_len * HB)
            sh = (ctypes.c_int32 * HB)()
            args_h = (h.h, HB, in_host, rs_host, ctypes.cast(ph, ctypes.c_void_p), ctypes.cast(wh, ctypes.c_void_p), ctypes.cast(sh, ctypes.c_void_p))
            assert h.L.spp_prove_batch(*args_h) == 0
            assert ph.raw[388 * (HB - 1):388 * HB] == ph.raw[388 * (B - 1):388 * B]      # same inputs, same blinding, same bytes
            th = time.perf_counter()
            for _ in range(2):
                assert h.L.spp_prove_batch(*args_h) == 0
            host_rate = 2 * HB / (time.perf_counter() - th)

        # SURVEY 8d Config 2: ONE withdraw proof from the reference's own inputs (client/prover-params.toml, committed as
        # tests/golden/withdraw_kat.json), end to end on the device-resident entry point: latency, and proofs/s at batch 1
        single = None
        if rank == 0 and circuit in ("withdraw", "audit") and not args.no_single:
            if circuit == "withdraw":
                from oracle import circuit as OC
                kat_row = OC.withdraw_inputs(json.load(open(os.path.join(ROOT, "tests", "golden", "withdraw_kat.json"))))
                kat_name = "client/prover-params.toml (tests/golden/withdraw_kat.json)"
            else:   # SURVEY 8d Config 1: the reference's own run, sk = 12345, Random(999) (scripts/generate_audit.py:469-470)
                import random
                from oracle import rlwe
                pkj = json.load(open(os.path.join(ROOT, "tests", "golden", "rlwe_pk.json")))
                kat_row = rlwe.audit_input_vector(rlwe.audit_inputs(pkj["a"], pkj["b"], 12345, random.Random(999)))
                kat_name = "scripts/generate_audit.py defaults: sk = 12345, Random(999), demo rlwe_pk.json"
            one_in = torch.frombuffer(bytearray(b"".join(int(v).to_bytes(32, "big") for v in kat_row)), dtype=torch.uint8).to(dev)
            one_rs = torch.frombuffer(bytearray((5).to_bytes(32, "big") + (6).to_bytes(32, "big")), dtype=torch.uint8).to(dev)
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
            single = {"inputs": kat_name, "latency_ms_median": round(lat[len(lat) // 2], 3),
                      "latency_ms_min": round(lat[0], 3), "proofs_per_s_at_batch_1": round(1e3 / lat[len(lat) // 2], 2)}

        out = None
        if rank == 0:
            # the timed batches produced real proofs: check two of the last step with the product's own pairing verifier
            last = (step_no[0] - 1) & 1
            pbytes, wbytes = proofs[last].cpu().numpy().tobytes(), pws[last].cpu().numpy().tobytes()
            vkb = open(vkp, "rb").read()
            for i in (0, B - 1):
                assert spp.verify(vkb, pbytes[388 * i:388 * (i + 1)], wbytes[h.pw_len * i:h.pw_len * (i + 1)]), "proof %d does not verify" % i
            # ... and every proof of that batch with the batched GPU verifier (spp_verify_batch)
            all_ok = ctx.verify_batch(vkb, [pbytes[388 * i:388 * (i + 1)] for i in range(B)],
                                      [wbytes[h.pw_len * i:h.pw_len * (i + 1)] for i in range(B)])
            assert all(all_ok), "%d proofs of the last timed batch do not verify" % (B - sum(all_ok))
            total_proofs = B * world * steps
            value = total_proofs / elapsed
            sizes = h.msm_sizes()
            g1_sizes = sizes[:6]
            # algorithmic bytes of one k_msm_fixed<G1> launch: every base once (64 B) + one 32 B scalar per (base, proof)
            alg_bytes = sum(64 * n + 32 * n * B for n in g1_sizes) / len(g1_sizes)
            avg_ms = acc["kern_ms"] / max(acc["kern_n"], 1)
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            traffic = None
            valu_util = None
            try:   # PMC pass of the same workload, committed under profiles/ (rocprofv3 --pmc cannot run inside this process)
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")))
                if pmc.get("circuit") == circuit and pmc.get("batch") == B and pmc.get("window_bits") == h.window_bits:
                    traffic = pmc["k_msm_fixed_g1_hbm_bytes_per_launch"]
                    valu_util = pmc.get("k_msm_fixed_g1_valu_issue_util_serialised")
            except Exception:
                pass
            out = {
                "value": round(value, 3), "ms_per_step": round(elapsed / steps * 1e3, 3),
                "config": {"workload": "%s, batch of %d independent proofs per GPU per step" % (
                    {"withdraw": "noir_circuit withdraw (Poseidon Merkle depth 16 + Grumpkin)",
                     "withdraw_refshape": "noir_circuit withdraw padded with ballast multiplications to the reference's gnark R1CS size",
                     "withdraw_depth20": "withdraw statement over a depth-20 Poseidon tree (synthetic variant; the reference is depth 16)",
                     "audit": "audit_circuit (RLWE, const-PK)"}[circuit], B),
                    "circuit": circuit, "n_constraints": h.n_constraints, "n_wires": h.n_wires, "domain": 1 << h.domain_log,
                    "batch_per_gpu": B, "window_bits": h.window_bits, "msm_windows": dict(zip(["A", "B1", "K", "Z", "CB", "CS", "B2(G2)"], h.msm_windows())), "msm_sizes": dict(zip(["A", "B1", "K", "Z", "CB", "CS", "B2(G2)"], sizes)),
                    "table_bytes": h.table_bytes, "parallelism": "independent proofs x%d" % world,
                    "pk_bcast_ms": round(bcast_ms, 3), "setup_s": round(setup_s, 2), "load_s": round(load_s, 2),
                    "host_buffer_entry_proofs_per_s": round(host_rate, 1),
                    "last_timed_batch_verified": "all %d proofs accepted by spp_verify_batch (GPU), two of them also by spp_verify (host)" % B,
                    "host_buffer_entry_note": "one spp_prove_batch call with host pointers for 4 batches' worth of proofs: PCIe copies included, chunks pipelined inside libspp"},
                "stage_ms_per_step": {k: round(v / steps, 3) for k, v in zip(
                    ["solve+commit", "matrix_eval", "ntt_qap", "msm_g1", "msm_g2_side_stream_join", "assemble", "total"], acc["stage"])},
                "roofline": {"bound": "hbm", "kernel": "k_msm_fixed<Fq>", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "alg_bytes_per_launch": int(alg_bytes),
                             "avg_launch_ms": round(avg_ms, 4), "launches_timed": acc["kern_n"],
                             "valu_issue_util_pmc": valu_util,   # SQ counters of the serialised run of the same workload (profiles/)
                             "note": "integer-VALU bound (about 2.3K instructions, 1.5K of them v_mad_u64_u32, per mixed addition); HBM fraction reported as mandated"},
            }
            if single is not None:
                out["single_proof"] = single
            if want_cpu:
                orc = native.Prover(sppc, pkp)
                cores = native.max_threads()
                # throughput mode: one proof per host thread, rounds of `cores` proofs until ~10 s have elapsed
                t1 = time.perf_counter()
                done = 0
                while done == 0 or (time.perf_counter() - t1 < 10.0 and done < 4096):
                    batch = [rows[(done + i) % len(rows)] for i in range(cores)]
                    rc, _, _ = native.prove_many(orc, batch, [(7 + done + i, 11 + done + i) for i in range(cores)])
                    assert rc == 0
                    done += cores
                dt = time.perf_counter() - t1
                t2 = time.perf_counter()
                for i in range(3):
                    assert orc.prove(rows[i % len(rows)], 3 + i, 4 + i)[0] == 0
                one_ms = (time.perf_counter() - t2) / 3 * 1e3
                out["cpu_baseline"] = {"value": round(done / dt, 3), "unit": "proofs/s", "cores": cores, "kind": "port",
                                       "single_proof_latency_ms": round(one_ms, 1),
                                       "sample": "%d %s proofs, one per host thread, oracle C/OpenMP prover (stands in for the Sunspot "
                                                 "Go/CPU path, which cannot run here: no Go toolchain, no pk)" % (done, circuit)}
        h.close()
        ctx.close()
        del inp, rs, proofs, pws, status
        torch.cuda.empty_cache()
        return out

    default_batch = {"withdraw": 4096, "audit": 2048, "withdraw_refshape": 2048, "withdraw_depth20": 2048}
    main_res = run_circuit(args.circuit, args.batch or default_batch[args.circuit], args.steps, args.warmup,
                           not args.no_cpu_baseline and world == 1)   # CPU baseline: rank 0 at N=1 only
    # the other circuit of BASELINE.json's metric, as a secondary figure (single GPU runs only)
    other = None
    if world == 1 and not args.no_secondary:
        oc = "audit" if args.circuit == "withdraw" else "withdraw"
        other = run_circuit(oc, default_batch[oc], 3, 1, not args.no_cpu_baseline)
    # like-for-like size check: the withdraw statement at the reference R1CS's dimensions (12 452 constraints, 2^14)
    refshape = None
    if world == 1 and not args.no_refshape and args.circuit == "withdraw":
        refshape = run_circuit("withdraw_refshape", default_batch["withdraw_refshape"], 3, 1, False)
    depth20 = None
    if world == 1 and not args.no_refshape and args.circuit == "withdraw":
        depth20 = run_circuit("withdraw_depth20", default_batch["withdraw_depth20"], 3, 1, False)
    if rank == 0:
        line = {"metric": "Groth16 proofs/sec", "value": main_res["value"], "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u32x8 (254-bit Montgomery integers)", "data": "synthetic"}
        for k in ("config", "stage_ms_per_step", "roofline", "cpu_baseline", "single_proof"):
            if k in main_res:
                line[k] = main_res[k]
        if other is not None:
            line["secondary_" + ("audit" if args.circuit == "withdraw" else "withdraw") + "_circuit"] = other
        if refshape is not None:
            line["withdraw_at_reference_r1cs_size"] = refshape
        if depth20 is not None:
            line["withdraw_depth20_variant"] = depth20
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

import json, os, sys, tempfile, time
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
import torch, spp
# full-size blinding factors (what a real prover draws): the s*Ar / r*Bs1 part of the assembly depends on their length
R1 = 0x1f3a9c0de4b5a697887766554433221100ffeeddccbbaa998877665544332211 % (1 << 253)
R2 = 0x0e2d4c6b8a79685746352413021f0e0dccbbaa99887766554433221100fedcba
from spp import workload
dev=torch.device("cuda",0)
tmp=tempfile.mkdtemp()
pk=json.load(open(os.path.join(ROOT,"tests/golden/rlwe_pk.json")))
for cid,name in ((1,"withdraw"),(2,"audit"),(5,"withdraw_acir")):
    sppc,pkp,vkp=(os.path.join(tmp,name+e) for e in (".sppc",".pk",".vk"))
    if cid==5:
        from spp import acir
        acir.compile_to_sppc(os.path.join(ROOT,"tests","golden","reference_withdraw_acir.json"),sppc)
    else:
        spp.build_circuit(cid,sppc,aux=(list(pk["a"])+list(pk["b"])) if cid==2 else None)
    ctx=spp.Context(0); ctx.setup(sppc,b"\x2a"*32,pkp,vkp)
    for win in ((8,0) if (cid!=5 and not os.environ.get('SPP_STAGES_QUICK')) else (8,)):
        t0=time.time(); h=ctx.load_circuit(sppc,pkp,win); load=time.time()-t0
        rows=workload.withdraw_rows(ctx,1) if cid!=2 else workload.audit_rows(ctx,pk["a"],pk["b"],1)
        inp=torch.frombuffer(bytearray(rows),dtype=torch.uint8).to(dev)
        rs=torch.frombuffer(bytearray((R1).to_bytes(32,"big")+(R2).to_bytes(32,"big")),dtype=torch.uint8).to(dev)
        pr=torch.zeros(388,dtype=torch.uint8,device=dev); pw=torch.zeros(h.pw_len,dtype=torch.uint8,device=dev); st=torch.zeros(1,dtype=torch.int32,device=dev)
        lat=[]
        for _ in range(8):
            torch.cuda.synchronize(); t=time.perf_counter()
            h.prove_batch_device(1,inp.data_ptr(),rs.data_ptr(),pr.data_ptr(),pw.data_ptr(),st.data_ptr()); h.sync()
            lat.append((time.perf_counter()-t)*1e3)
        print(name,"window",win,"load_s",round(load,2),"lat_ms",round(sorted(lat)[len(lat)//2],2),"stages",[round(x,2) for x in h.last_timings(0)[:7]],"msm",[round(x,3) for x in h.msm_kernel_ms(0)], flush=True)
        h.close()
    ctx.close()

import sys, ctypes as C, os
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib; lib=L.lib()
for rows in (21760, 9216):
    d=512
    x=torch.randn(rows,d,device="cuda"); dy=torch.randn(rows,d,device="cuda"); a=torch.randn(d,device="cuda"); b=torch.randn(d,device="cuda")
    y=torch.empty(rows,d,device="cuda"); st=torch.empty(rows,2,device="cuda"); dres=torch.randn(rows,d,device="cuda")
    dx=torch.empty_like(x); da=torch.zeros(d,device="cuda"); db=torch.zeros(d,device="cuda"); dz=torch.empty(rows,d,device="cuda",dtype=torch.bfloat16)
    L.check(lib.ortk_layernorm_fwd(L.ptr(x),L.ptr(a),L.ptr(b),L.ptr(y),0,L.ptr(st),rows,d,1e-6,L.stream_ptr()),"f")
    for name,dzp,p in (("plain",None,0.0),("dz bf16",dz,0.0),("dz bf16 + dropout",dz,0.1)):
        f=lambda: lib.ortk_layernorm_bwd_drop(L.ptr(dy),L.ptr(x),L.ptr(a),L.ptr(st),L.ptr(dres),L.ptr(dx),L.ptr(da),L.ptr(db),rows,d,C.c_float(1e-6),L.ptr(dzp),1,C.c_float(p),123,L.stream_ptr())
        for _ in range(3): L.check(f(),"b")
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)*1e3/20
        print(f"rows {rows} {name:20s}: {us:.1f} us", flush=True)

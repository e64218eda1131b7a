import sys, ctypes as C, os
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
import os as _os
if "ORTK_ATTN_IMPL" in _os.environ:      # (the library itself reads no environment: forward the old switch through ortk_set_tuning)
    P._lib.set_tuning(attn_impl=int(_os.environ["ORTK_ATTN_IMPL"]))

L=P._lib
def run(name,nkv,H,Lq,Lk,dk,causal,bias,reps=10):
    d=H*dk
    q=torch.randn(nkv*Lq,d,device="cuda"); k=torch.randn(nkv*Lk,d,device="cuda"); v=torch.randn(nkv*Lk,d,device="cuda"); do=torch.randn(nkv*Lq,d,device="cuda")
    o=torch.empty_like(q); p=torch.empty(nkv,H,Lq,Lk,device="cuda"); km=torch.ones(nkv,Lk,device="cuda")
    a=L.AttnArgs(); a.q,a.k,a.v,a.o,a.p=q.data_ptr(),k.data_ptr(),v.data_ptr(),o.data_ptr(),p.data_ptr()
    a.ldq=a.ldk=a.ldv=a.ldo=d; a.kmask=km.data_ptr(); a.nkv,a.H,a.Lq,a.Lk,a.dk,a.causal_period=nkv,H,Lq,Lk,dk,causal
    if bias: b=torch.randn(nkv,H,Lq,Lk,device="cuda"); a.bias=b.data_ptr()
    dq,dk_,dv=torch.empty_like(q),torch.empty_like(k),torch.empty_like(v)
    a.d_o,a.dq,a.d_k,a.dv=do.data_ptr(),dq.data_ptr(),dk_.data_ptr(),dv.data_ptr(); a.lddo=a.lddq=a.lddk=a.lddv=d
    res=[]
    for fn in (L.lib().ortk_attention_fwd, L.lib().ortk_attention_bwd):
        for _ in range(2): L.check(fn(C.byref(a),L.stream_ptr()),"a")
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn(C.byref(a),L.stream_ptr())
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)*1e3/reps)
    print(f"impl {os.environ.get('ORTK_ATTN_IMPL','0')} {name:6s}: fwd {res[0]:7.1f} us  bwd {res[1]:7.1f} us", flush=True)
run("enc",256,8,36,36,64,0,True)
run("self",1280,8,17,17,64,17,False)
run("cross",256,8,85,36,64,0,False)

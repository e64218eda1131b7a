import sys, ctypes as C
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
nkv,H,Lq,Lk,dk=256,8,85,36,64
d=H*dk
q=torch.randn(nkv*Lq,d,device="cuda"); k=torch.randn(nkv*Lk,d,device="cuda"); v=torch.randn(nkv*Lk,d,device="cuda"); do=torch.randn(nkv*Lq,d,device="cuda")
o=torch.empty_like(q); p=torch.empty(nkv,H,Lq,Lk,device="cuda")
a=L.AttnArgs(); a.q,a.k,a.v,a.o,a.p=q.data_ptr(),k.data_ptr(),v.data_ptr(),o.data_ptr(),p.data_ptr()
a.ldq=a.ldk=a.ldv=a.ldo=d; a.nkv,a.H,a.Lq,a.Lk,a.dk=nkv,H,Lq,Lk,dk
L.check(L.lib().ortk_attention_fwd(C.byref(a),L.stream_ptr()),"f")
dq,dk_,dv=torch.empty_like(q),torch.empty_like(k),torch.empty_like(v)
a.d_o,a.dq,a.d_k,a.dv=do.data_ptr(),dq.data_ptr(),dk_.data_ptr(),dv.data_ptr(); a.lddo=a.lddq=a.lddk=a.lddv=d
for part in (0,1,2):
    a.bwd_part=part
    for _ in range(3): L.lib().ortk_attention_bwd(C.byref(a),L.stream_ptr())
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): L.lib().ortk_attention_bwd(C.byref(a),L.stream_ptr())
    e1.record(); torch.cuda.synchronize(); print("part",part,round(e0.elapsed_time(e1)*1e3/20,1),"us")

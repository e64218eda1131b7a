import sys, ctypes as C
sys.path[:0]=["/root/repo","/root/repo/tests","/root/repo/tests/golden"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
torch.manual_seed(0)
nkv,H,Lq,Lk,dk=1,1,2,3,4
d=H*dk
q=torch.randn(nkv*Lq,d).cuda(); k=torch.randn(nkv*Lk,d).cuda(); v=torch.randn(nkv*Lk,d).cuda(); do=torch.randn(nkv*Lq,d).cuda()
o=torch.empty_like(q); p=torch.empty(nkv,H,Lq,Lk).cuda()
a=L.AttnArgs()
a.q,a.k,a.v,a.o,a.p=q.data_ptr(),k.data_ptr(),v.data_ptr(),o.data_ptr(),p.data_ptr()
a.ldq=a.ldk=a.ldv=a.ldo=d
a.nkv,a.H,a.Lq,a.Lk,a.dk=nkv,H,Lq,Lk,dk
L.check(L.lib().ortk_attention_fwd(C.byref(a),L.stream_ptr()),"f")
dq=torch.full_like(q,7.); dk_=torch.full_like(k,7.); dv=torch.full_like(v,7.); ds=torch.full((nkv,H,Lq,Lk),7.).cuda()
a.d_o,a.dq,a.d_k,a.dv,a.dscore=do.data_ptr(),dq.data_ptr(),dk_.data_ptr(),dv.data_ptr(),ds.data_ptr()
a.lddo=a.lddq=a.lddk=a.lddv=d
L.check(L.lib().ortk_attention_bwd(C.byref(a),L.stream_ptr()),"b")
torch.cuda.synchronize()
qr,kr,vr=[t.cpu().clone().requires_grad_() for t in (q,k,v)]
s=(qr@kr.t())/dk**0.5; pr=torch.softmax(s,-1); (pr@vr).backward(do.cpu())
print("P",p.cpu().flatten(), pr.detach().flatten())
print("dscore",ds.cpu().flatten())
print("dq",dq.cpu(), qr.grad)
print("dk",dk_.cpu(), kr.grad)
print("dv",dv.cpu(), vr.grad)

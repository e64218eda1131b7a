"""Decode-time cross-attention (5 beams x 36 regions, bf16 K/V): does the row stride of the projected memory matter?"""
import sys, ctypes as C
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
def run(nkv, Lq, Lk, ld, kvdt, reps=30):
    H, dk = 8, 64; d=H*dk
    q=torch.randn(nkv*Lq, d, device="cuda"); o=torch.empty(nkv*Lq, d, device="cuda", dtype=torch.bfloat16)
    kv=torch.randn(nkv*Lk, ld, device="cuda").to(torch.bfloat16 if kvdt else torch.float32)
    km=torch.ones(nkv, Lk, device="cuda")
    a=L.AttnArgs(); a.q=q.data_ptr(); a.k=kv.data_ptr(); a.v=kv.data_ptr()+d*kv.element_size(); a.o=o.data_ptr()
    a.ldq=d; a.ldk=a.ldv=ld; a.ldo=d; a.kmask=km.data_ptr(); a.nkv,a.H,a.Lq,a.Lk,a.dk=nkv,H,Lq,Lk,dk; a.o_dtype=1; a.kv_dtype=kvdt; a.precision=1
    for _ in range(3): L.check(L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()),"a")
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
for nkv,Lq in ((1024,5),(256,6)):
    for ld in (1024, 6144):
        print(f"nkv {nkv} Lq {Lq} Lk 36 bf16 K/V row stride {ld:5d}: {run(nkv,Lq,36,ld,1):6.1f} us")

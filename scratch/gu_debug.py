import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.sparse import SparsePlan
L = P._lib; lib = L.lib()
torch.manual_seed(0)
M, N, K, sp = 300, 130, 70, 0.8
W = torch.randn(N, K) * (torch.rand(N, K) >= sp).float()
X = torch.randn(M, K)
for xdt in (1, 0):
    plan = SparsePlan([dict(offset=0, N=N, K=K, ld=K)], L.SP_GU16, "cuda")
    plan.build(W.cuda())
    torch.cuda.synchronize()
    print("smax", plan.perm.tolist(), "nsteps", plan.chunk_len.tolist())
    Xd = (X.bfloat16() if xdt else X).cuda()
    Y = torch.full((M, N), float("nan"), device="cuda")
    a = L.SpmmArgs(); a.X, a.Y, a.ldx, a.ldy, a.M, a.x_dtype, a.y_dtype = Xd.data_ptr(), Y.data_ptr(), K, N, M, xdt, 0
    lib.ortk_spmm(plan.ref(), 0, C.byref(a), L.stream_ptr()); torch.cuda.synchronize()
    ref = Xd.float().cpu().bfloat16().float() @ W.bfloat16().float().t()
    err = (Y.cpu() - ref).abs()
    bad = (~torch.isfinite(Y.cpu())) | (err > 1e-2)
    print("xdt", xdt, "bad count", int(bad.sum()), "bad cols", sorted(set(bad.nonzero()[:, 1].tolist()))[:20], "bad rows", sorted(set(bad.nonzero()[:, 0].tolist()))[:20])

"""One sparse-product shape in a loop (for rocprofv3 --pmc / --kernel-trace): python scratch/spmm_one.py M N K sparsity reps"""
import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.sparse import SparsePlan, capacity_for
L = P._lib; lib = L.lib()
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); sp = float(sys.argv[4]); reps = int(sys.argv[5])
torch.manual_seed(0)
W = (torch.randn(N, K, device="cuda") * (torch.rand(N, K, device="cuda") >= sp).float()).bfloat16()
fmt = int(sys.argv[6]) if len(sys.argv) > 6 else 2
plan = SparsePlan([dict(offset=0, N=N, K=K, ld=K, capacity=capacity_for(N, K, 0.1))], fmt, "cuda")
plan.build(W)
X = torch.randn(M, K, device="cuda").bfloat16(); Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
sa = L.SpmmArgs(); sa.X, sa.Y, sa.ldx, sa.ldy, sa.M, sa.x_dtype, sa.y_dtype = X.data_ptr(), Y.data_ptr(), K, N, M, 1, 1
for _ in range(reps):
    lib.ortk_spmm(plan.ref(), 0, C.byref(sa), L.stream_ptr())
torch.cuda.synchronize()
print("steps/len sum", int(plan.chunk_len.sum()), "nnz", plan.nnz)

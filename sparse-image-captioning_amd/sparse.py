"""CSR images of pruned weight blocks for the sparse decode path (``ortk_spmm_csr``).

The reference evaluates pruned checkpoints as dense linears on zero-filled weights (``scripts/eval_model.py:64-88``:
COO state dict -> ``densify_state_dict`` -> dense model).  Here the zero pattern is turned into a chunked CSR once
per weight version (plumbing: torch index ops on the device), and every projection whose block is sparse enough is
multiplied by ``ortk_spmm_csr`` inside ``ortk_decode``.  Layout: see ``include/ortk.h`` (``ortk_csr``).
"""
import ctypes as C

import torch

from . import _lib as L

CHUNK = 512   # K columns per chunk; must match KC in csrc/ortk_sparse.hip
PAD = 4       # every (chunk,row) entry list is padded to a multiple of PAD entries


def csr_from_dense(w):
    """(N, K) device tensor -> (row_ptr int32 [nchunk*N+1], col int16 [n] (relative to chunk), val fp32 [n]);
    lists padded to multiples of PAD with (0, 0.0) entries, PAD spare entries at the end (include/ortk.h: ortk_csr)."""
    N, K = w.shape
    nch = (K + CHUNK - 1) // CHUNK
    wp = w
    if nch * CHUNK != K:
        wp = torch.zeros(N, nch * CHUNK, dtype=w.dtype, device=w.device)
        wp[:, :K] = w
    wc = wp.view(N, nch, CHUNK).permute(1, 0, 2).reshape(nch * N, CHUNK)   # chunk-major rows
    nz = wc != 0
    counts = nz.sum(1)
    padded = (counts + PAD - 1) // PAD * PAD
    row_ptr = torch.zeros(nch * N + 1, dtype=torch.int64, device=w.device)
    row_ptr[1:] = padded.cumsum(0)
    idx = nz.nonzero()                       # sorted by (row, col)
    first = counts.cumsum(0) - counts        # rank of each row's first non-zero in `idx`
    rank = torch.arange(idx.size(0), device=w.device) - first[idx[:, 0]]
    dest = row_ptr[idx[:, 0]] + rank
    total = int(row_ptr[-1]) + PAD           # spare batch at the end (read-ahead)
    col = torch.zeros(total, dtype=torch.int16, device=w.device)
    val = torch.zeros(total, dtype=torch.float32, device=w.device)
    col[dest] = idx[:, 1].to(torch.int16)
    val[dest] = wc[idx[:, 0], idx[:, 1]].float()
    return row_ptr.to(torch.int32), col, val


class SparseTable:
    """Owns the CSR tensors and the ctypes ``ortk_csr`` array handed to ``ortk_decode``."""

    def __init__(self, ccfg, flat, min_sparsity=0.9):
        lib = L.lib()
        n = lib.ortk_linear_block(C.byref(ccfg), -1, None, None, None)
        if n < 0:
            L.check(n, "ortk_linear_block")
        self.blocks, self._keep = [], []
        ents = []
        off, N, K = C.c_int64(), C.c_int32(), C.c_int32()
        for i in range(n):
            L.check(lib.ortk_linear_block(C.byref(ccfg), i, C.byref(off), C.byref(N), C.byref(K)), "ortk_linear_block")
            w = flat[off.value: off.value + N.value * K.value].view(N.value, K.value)
            sparsity = 1.0 - float((w != 0).float().mean())
            if sparsity < min_sparsity:
                continue
            rp, col, val = csr_from_dense(w)
            self._keep += [rp, col, val]
            ents.append(L.Csr(L.ptr(rp), L.ptr(col), L.ptr(val), N.value, K.value, off.value))
            self.blocks.append({"offset": off.value, "N": N.value, "K": K.value, "nnz": int((val != 0).sum()), "sparsity": sparsity})
        self.n = len(ents)
        self.array = (L.Csr * max(self.n, 1))(*ents)

    @property
    def nnz(self):
        return sum(b["nnz"] for b in self.blocks)

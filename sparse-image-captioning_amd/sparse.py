"""Sparse plans for pruned weight blocks (``include/ortk.h``: ``ortk_sparse_plan``).

The reference multiplies by the zero-filled weight ``s * W`` in every masked layer (``pruning/masked_layer.py:84-110,
134-135``) and evaluates pruned checkpoints as dense linears on zero-filled weights (``scripts/eval_model.py:64-88``).
Here a *plan* names the weight blocks that are sparse enough and owns the device buffers of their sparse images; the
images themselves are (re)built on the device inside ``ortk_forward`` / ``ortk_decode`` from the effective weights of that
very call, so nothing on the host can go stale when the weights, the masks or the mask sample change.

Formats (``_lib.SP_*``): ``SP_GU16`` — group-union MFMA images, the mixed-precision fast path; ``SP_ELL32`` — sorted ELL with
fp32 values, the fp32 parity mode; ``SP_ELL16`` — sorted ELL with bf16 pairs (VALU kernel, kept for comparison).

This module only does plumbing: block selection (one-off, at ``enable_sparse_kernels``), buffer sizes, ctypes tables.
"""
import ctypes as C
import os

import torch

from . import _lib as L

KMAX = 2048      # ELL: input columns one product launch takes (csrc/ortk_sparse.hip); wider blocks are cut along their inputs
GKC, GSTEPS = 512, 16     # GU16: input columns per chunk, k-steps reserved per (group, chunk) slot


def linear_blocks(ccfg):
    """[(arena offset, N, K)] of every weight matrix the executor multiplies by (packed projections count once)."""
    lib = L.lib()
    n = lib.ortk_linear_block(C.byref(ccfg), -1, None, None, None)
    if n < 0:
        L.check(n, "ortk_linear_block")
    out, seen = [], set()
    off, N, K = C.c_int64(), C.c_int32(), C.c_int32()
    for i in range(n):
        L.check(lib.ortk_linear_block(C.byref(ccfg), i, C.byref(off), C.byref(N), C.byref(K)), "ortk_linear_block")
        key = (off.value, N.value, K.value)
        if key not in seen:                       # ACORT layer sharing lists a shared block once per position
            seen.add(key)
            out.append(key)
    return out


class SparsePlan:
    """One ``ortk_sparse_plan``: a set of (N outputs, K inputs) blocks of a dense buffer + the device buffers of their images.

    ``blocks``: list of dicts ``offset, N, K, ld`` (element offset / leading dimension inside the dense buffer the builder
    will be given) and, for the ELL formats, ``capacity`` (entries)."""

    def __init__(self, blocks, fmt, device):
        assert blocks, "a sparse plan needs at least one block"
        self.blocks = blocks
        self.format = fmt
        n = len(blocks)
        self._host = (L.EllBlock * n)()
        rows = chunks = entries = slots = 0
        for i, b in enumerate(blocks):
            h = self._host[i]
            h.src_offset, h.ld, h.N, h.K = b["offset"], b["ld"], b["N"], b["K"]
            if fmt == L.SP_GU16:
                assert b["K"] <= GKC or b["N"] <= 512, "GU16: blocks with more than 512 inputs may have at most 512 outputs"
                h.chunk0 = slots
                slots += ((b["N"] + 15) // 16) * ((b["K"] + GKC - 1) // GKC)
            else:
                assert 1 <= b["K"] <= KMAX and 1 <= b["N"] <= 16384
                h.stream_offset, h.capacity = entries, b["capacity"]
                h.chunk0, h.row0 = chunks, rows
                rows += b["N"]
                chunks += (b["N"] + 63) // 64
                entries += b["capacity"]
        self._dev = torch.frombuffer(bytearray(bytes(self._host)), dtype=torch.uint8).to(device)
        z = lambda k, dt=torch.int32: torch.zeros(max(int(k), 1), dtype=dt, device=device)
        if fmt == L.SP_GU16:
            self.stream = z(slots * GSTEPS * 64 * 8, torch.bfloat16)       # wfrag
            self.chunk_ptr = z(slots * GSTEPS * 64)                         # kofs
            self.chunk_len = z(slots)                                       # nsteps
            self.perm = z(n)                                                # per-block maximum of nsteps
            self.count = z(slots)                                           # non-zeros per slot
            self.overflow = z(1)
            total = slots
        else:
            assert entries < 2 ** 31, "entry offsets are 32-bit"
            self.stream = z(entries * (4 if fmt == L.SP_ELL16 else 8) // 4)
            self.chunk_ptr, self.chunk_len = z(chunks), z(chunks)
            self.perm = torch.full((max(chunks, 1) * 64,), -1, dtype=torch.int32, device=device)
            self.count = z(rows)
            self.overflow = z(1)
            total = rows
        self.n, self.total_rows = n, total
        p = L.EllPlanStruct()
        p.blocks_host = C.cast(self._host, C.POINTER(L.EllBlock))
        p.blocks_dev = self._dev.data_ptr()
        p.nblocks, p.format = n, fmt
        p.stream, p.chunk_ptr, p.chunk_len = self.stream.data_ptr(), self.chunk_ptr.data_ptr(), self.chunk_len.data_ptr()
        p.perm = self.perm.data_ptr() if self.perm is not None else None
        p.count_scratch, p.overflow = self.count.data_ptr(), self.overflow.data_ptr()
        p.total_rows = total
        self.struct = p

    def ref(self):
        return C.pointer(self.struct)

    def build(self, dense):
        """(Re)build every image from ``dense`` (fp32 or bf16 device tensor) — what ortk_forward / ortk_decode do themselves;
        exposed for the operator-level tests and tools."""
        dt = {torch.float32: 0, torch.bfloat16: 1}[dense.dtype]
        L.check(L.lib().ortk_sparse_build(self.ref(), L.ptr(dense), dt, L.stream_ptr()), "ortk_sparse_build")

    def spmm(self, block, args):
        L.check(L.lib().ortk_spmm(self.ref(), block, C.byref(args), L.stream_ptr()), "ortk_spmm")

    def check_overflow(self):
        """Host sync: raises if a build SINCE THE LAST CHECK dropped entries (ELL: a block denser than the capacity it was
        planned with; the GU16 format is worst-case sized and cannot overflow).  The device flag is sticky across builds and
        cleared here."""
        hit = int(self.overflow.item())
        self.overflow.zero_()
        if hit:
            raise L.OrtkError("sparse plan overflow: a weight block has more non-zeros than the capacity reserved at "
                              "enable_sparse_kernels(); call it again with a lower min_sparsity")

    @property
    def nnz(self):
        """Non-zeros counted by the last build (host sync)."""
        return int(self.count.sum().item())


def capacity_for(N, K, max_density):
    """ELL: entries reserved for an (N, K) block: the sorted chunks pad each group of 64 columns to its longest member (about
    +15 % at K = 512 / 5 % density, +5 % at K = 2048) and to a multiple of 8 entries (4 in the fp32 format)."""
    n = int(max_density * N * K * 1.3) + 8 * 64 * ((N + 63) // 64)
    return (n + 63) // 64 * 64


# Measured crossover (mixed precision, ELL16 vs this library's dense bf16 GEMM on the same zero-filled weights, MI355X;
# scratch/spmm_crossover.py -> profiles/r04_spmm_crossover.txt): fraction of zeros from which the sparse product of an (N outputs,
# K inputs) block beats the dense MFMA GEMM by >= 5 % (the images are rebuilt per call: a tie is a loss) at the row count it runs
# over — "enc": 9 216 rows (256 images x 36 regions; also the packed cross-attention K|V projection), "dec": 16 640-21 760 rows
# (the captions' positions), "decode": 5 120 rows (1 024 images x 5 beams).  Keys are the shapes of the PRODUCT: forward blocks
# (N, K) of the weight, data-gradient blocks (K, N) of its transpose.  1.01 = never measured faster (up to 99.5 % zeros).
# At 95 % zeros no shape pays (0.65-0.93x of the dense GEMM's speed); at the reference's published 98.8 % models most do.
CROSSOVER = {
    "enc": {(512, 2048): 0.975, (512, 512): 0.975, (1536, 512): 0.988, (2048, 512): 0.975, (6144, 512): 0.98,
            (512, 1536): 0.975, (512, 6144): 0.98},
    "dec": {(512, 512): 0.975, (1536, 512): 0.985, (2048, 512): 0.98, (512, 2048): 0.985, (10112, 512): 0.975,
            (512, 1536): 0.975},
    "decode": {(512, 512): 0.99, (1536, 512): 0.985, (2048, 512): 0.985, (512, 2048): 0.995, (10112, 512): 0.975, (6144, 512): 0.98},
}
DEFAULT_CROSSOVER = 0.99


def crossover(N, K, cls):
    return CROSSOVER[cls].get((N, K), DEFAULT_CROSSOVER)


def chains_serve(ccfg, precision):
    """Whether the executor's forward pass runs the row-wise operators of the layers as rows-stationary chains — dense products on
    the zero-filled weights, one launch per chain (csrc/ortk_model.hip: chain_cfg_ok)."""
    t = L.Tuning()
    L.lib().ortk_get_tuning(C.byref(t))
    return (bool(precision) and t.row_chain >= 1 and ccfg.d_model == 512 and ccfg.d_ff % 512 == 0 and ccfg.d_ff // 512 <= 8 and
            ccfg.n_heads == 8 and not ccfg.share_att_enc and not ccfg.share_att_dec and ccfg.n_layers <= 6)


def select_blocks(ccfg, eff, min_sparsity, train=False, precision=0):
    """Blocks of the arena ``eff`` whose fraction of zeros is >= ``min_sparsity`` — or, with ``"auto"``, those whose forward
    (``fwd``) / data-gradient (``bwd``) product is past the measured crossover of its shape and row class (host syncs: one-off).
    Training with ``"auto"``: the products inside the layers stay with the forward chains where those run (one launch per chain
    beats per-operator sparse products: 12.6 vs 13.1 ms per step at 98.8 %); their data gradients still go sparse where that pays."""
    dec_off = int(L.lib().ortk_arena_decoder_offset(C.byref(ccfg)))
    blocks = linear_blocks(ccfg)
    chains = train and min_sparsity == "auto" and chains_serve(ccfg, precision)
    out = []
    for off, N, K in blocks:
        w = eff[off: off + N * K]
        sparsity = 1.0 - float(torch.count_nonzero(w)) / (N * K)
        if min_sparsity == "auto":
            # (the packed cross-attention K|V projection lives behind the decoder offset but runs over the encoder's rows)
            cls = ("enc" if (off < dec_off or N >= 2 * 1536) and N < 8192 else "dec") if train else "decode"
            # a (512, 10112) data-gradient product is cut into 2 048-input pieces: the (512, 2048) figure of its class
            # (with the chains: only the K|V projection of the memory and the generator, whose plan is then built beside the encoder)
            fwd = sparsity >= crossover(N, K, cls) and not (chains and N < 6144)
            bwd = sparsity >= crossover(K, min(N, KMAX), cls)
        else:
            fwd = bwd = sparsity >= min_sparsity
        if fwd or bwd:
            out.append({"offset": off, "N": N, "K": K, "sparsity": sparsity, "fwd": fwd, "bwd": bwd})
    return out


def default_format(precision):
    if not precision:
        return L.SP_ELL32
    # Measured on MI355X (scratch/spmm_bench.py): the VALU kernel on sorted ELL is the faster of the two
    # on the decode shapes and on the 2048-input blocks, the MFMA kernel on group unions on the widest outputs; neither beats
    # the dense MFMA GEMM at 95 %.  ELL16 is the default; `enable_sparse_kernels(..., fmt="gu16")` selects the other.
    return L.SP_ELL16


def make_plans(ccfg, eff, min_sparsity, precision, backward=False, fmt=None, density_of=None):
    """(forward plan, backward plan | None) for the blocks of ``eff`` that are sparse enough.

    Forward: the (N, K) weight blocks at their arena offsets.  Backward (mixed precision only): the same blocks transposed
    — (K outputs, N inputs), leading dimension N, at the same offsets of the transposed bf16 copy the executor keeps for its
    data-gradient GEMMs (ELL: a block with more than KMAX inputs — the generator: 10 112 — is cut into KMAX-wide pieces that
    the executor accumulates).  The region embedding (block 0) has no input gradient."""
    sel = select_blocks(ccfg, eff, min_sparsity, train=backward, precision=precision)
    if not sel:
        return None, None
    fmt = default_format(precision) if fmt is None else fmt
    dens0 = (1.0 - min_sparsity) if min_sparsity != "auto" else 0.0
    # capacity density per block: the planned bound, or — training with Bernoulli(sigmoid(m)) samples, whose density is the
    # MEAN of sigmoid(m), above the eval-mode round(sigmoid(m)) the selection saw — what `density_of(offset, N, K)` expects
    for b in sel:
        d0 = dens0 if min_sparsity != "auto" else min(1.0, 1.5 * (1.0 - b["sparsity"]) + 0.002)      # (auto: 1.5 x the density seen)
        b["dens"] = max(d0, density_of(b["offset"], b["N"], b["K"])) if density_of is not None else d0
    first = linear_blocks(ccfg)[0][0]
    gu = fmt == L.SP_GU16
    # (blocks the formats do not take — more than 16 384 outputs: the ELL planner's per-range tables — stay dense)
    ok = (lambda N, K: K <= GKC or N <= 512) if gu else (lambda N, K: K <= KMAX and N <= 16384)
    fwd = [dict(offset=b["offset"], N=b["N"], K=b["K"], ld=b["K"], capacity=capacity_for(b["N"], b["K"], b["dens"]), sparsity=b["sparsity"])
           for b in sel if b["fwd"] and ok(b["N"], b["K"])]
    plan_f = SparsePlan(fwd, fmt, eff.device) if fwd else None
    plan_b = None
    if backward and precision:
        bwd = []
        for b in sel:
            if b["offset"] == first or not b["bwd"]:
                continue
            if gu:
                if ok(b["K"], b["N"]):
                    bwd.append(dict(offset=b["offset"], N=b["K"], K=b["N"], ld=b["N"], sparsity=b["sparsity"]))
                continue
            if b["K"] > 16384:
                continue
            for k0 in range(0, b["N"], KMAX):            # inputs of the transposed block = outputs of the weight
                kw = min(KMAX, b["N"] - k0)
                bwd.append(dict(offset=b["offset"] + k0, N=b["K"], K=kw, ld=b["N"], capacity=capacity_for(b["K"], kw, b["dens"]),
                                sparsity=b["sparsity"]))
        plan_b = SparsePlan(bwd, fmt if gu else L.SP_ELL16, eff.device) if bwd else None
    return plan_f, plan_b

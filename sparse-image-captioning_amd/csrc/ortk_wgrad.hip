// ortk_wgrad.hip — the weight gradients of a whole layer in ONE grouped launch.
//
// Replaces the autograd of torch.nn.functional.linear with respect to the weight and the bias (the reference's step,
// scripts/train_transformer.py:65-81, differentiates every nn.Linear of models/transformer.py:214-358 one by one):
//     dW_i += dY_i^T X_i      (N_i x K_i, fp32, the gradient arena)          db_i += column sums of dY_i
// for up to ORTK_WGRAD_MAX projections that share the batch's rows (the six linears of a decoder layer, the four of an
// encoder layer).  Both operands are bf16, stored (rows, N_i) / (rows, K_i) row-major: the reduction runs over the rows, so
// both are k-major.
//
// Why grouped: alone, a 512 x 512 gradient has 4 (256 x 256) output tiles and needed split-K 24 on 128 x 128 tiles to fill
// the chip — 11 K-steps per workgroup between a pipeline prologue and a 64-KB atomic epilogue (62 launches, 4.8 ms of an
// 11.3-ms step at 280 TF/s; 1.5 GB of split-K atomics at the memory side's 1.3 TB/s).  A decoder layer's six gradients
// together are 56 tiles of 256 x 256: split-K 4 fills the chip with 224 workgroups that each run 130 K-steps, a quarter
// of the atomics, and one launch instead of six.
//
// Kernel: 256 x 256 tile, 8 waves (2 x 4, each 128 x 64 = 8 x 4 MFMA 16x16x32), a ring of four 32-row stages filled by
// LDS-DMA (global_load_lds_dwordx4: a k-row of a tile is 512 contiguous bytes), one raw s_barrier per stage, counted vmcnt;
// [k][m] images XOR-swizzled on the source address, fragments by ds_read_b64_tr_b16 (the images and reads of
// gemm_bf16_glds_kernel<true, true, true, 4>, ortk_gemm.hip).  Rows past a multiple of 32 go through one guarded,
// register-staged step (any row count: the valid-position decoder layout has one row per caption token).  The bias gradient
// is one more MFMA per A fragment against a vector of ones (first column tile only).  Epilogue: the fp32 tile is staged
// through the (free) ring in two halves and added to the arena as 256 contiguous bytes per atomic wave-instruction
// (MI355X_MICROARCH.md, Global float atomics: the accumulator layout itself would be the 17x slower access shape).
//
// Workgroup -> (K split, tile): ids are remapped so that one XCD walks consecutive tiles of ONE K range: the tiles of a
// projection share their dY rows (same m tile) and X rows (same n tile) in that XCD's L2.
#include <cstring>
#include <mutex>
#include "ortk_internal.h"

namespace {

constexpr int TM = 256;                // tile rows = tile columns
constexpr int WBK = 32, WNS = 4;       // rows of the batch per stage, stages
constexpr int W_IMG = TM * WBK;        // bf16 elements per operand image (16 KB)
constexpr int WCP = TM + 4;            // fp32 pitch of the staged half tile (1 040 B rows)
constexpr size_t W_RING_BYTES = (size_t)WNS * 2 * W_IMG * sizeof(__bf16);          // 128 KB
constexpr size_t W_EPI_BYTES = (size_t)128 * WCP * sizeof(float);                   // 133 120 B
constexpr size_t W_LDS_BYTES = W_RING_BYTES > W_EPI_BYTES ? W_RING_BYTES : W_EPI_BYTES;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct WgItem {
    const __bf16* A; const __bf16* B; float* C; float* cs;
    int lda, ldb, ldc, M, N, tilesN, tile0;
};
struct WgArgs {
    WgItem it[ORTK_WGRAD_MAX];
    int n, K, kchunk, tiles, lockstep, sk;
    f32x4* slabs;      // (tiles, sk, 256 x 256 fp32) partial tiles, or NULL: atomics
    int* tickets;      // (tiles) zeroed before the launch
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
__device__ __forceinline__ int swz_km(int k) { return 2 * ((k & 3) | ((k >> 1) & 4)); }

// Byte offset (from the first row of a stage) of the 16-byte chunk that lane `lane` of wave-instruction `inst` (0..15) of a 32 x 256 [k][m]
// stage fetches: chunk f = inst * 64 + lane of the image = (k-row f / 32, stored chunk f % 32), whose content is column chunk
// (f % 32) ^ swz_km(k) (columns past the operand's width are clamped: their products land in output rows / columns the epilogue drops).
// The offsets do not depend on the stage: a lane keeps them in registers and a stage's fetch is SGPR base + VGPR offset (no 64-bit
// vector address arithmetic in the loop: as pointers they cost four v_mad_i64 + a v_readfirstlane for M0 per stage and wave).
__device__ __forceinline__ uint32_t src_off(int ld, int tile0, int width, int inst, int lane) {
    const int f = inst * 64 + lane, kr = f >> 5, c = (f & 31) ^ swz_km(kr);
    return (uint32_t)((kr * ld + min(tile0 + c * 8, width - 8)) * 2);
}
__device__ __forceinline__ const __bf16* src_chunk(const __bf16* __restrict__ base, int ld, int tile0, int width, int k, int inst, int lane) {
    return reinterpret_cast<const __bf16*>(reinterpret_cast<const char*>(base + (int64_t)k * ld) + src_off(ld, tile0, width, inst, lane));
}
// one operand's stage: `rows` = the operand at the stage's first row (wave-uniform), off[u] = src_off of this wave's two instructions
__device__ __forceinline__ void dma_tile(const __bf16* __restrict__ rows, const uint32_t (&off)[2], __bf16* img, int wave) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(rows) + off[u]), (lds_void*)(img + (wave * 2 + u) * 512), 16, 0, 0);
}
// the same stage through registers, rows at or past k_end as zeros (the last, partial step of the last K range)
__device__ __forceinline__ void guarded_tile(const __bf16* __restrict__ base, int ld, int tile0, int width, int k, int k_end, __bf16* img, int wave,
                                             int lane) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int inst = wave * 2 + u;
        const int kr = (inst * 64 + lane) >> 5;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (k + kr < k_end) v = *reinterpret_cast<const u32x4*>(src_chunk(base, ld, tile0, width, k, inst, lane));
        *reinterpret_cast<u32x4*>(img + inst * 512 + lane * 8) = v;
    }
}
// Byte offset, inside a [32 k][256 m] stage image, of what this lane supplies to the pair of transposing reads that gather the MFMA
// 16x16x32 operand of columns m0 .. m0 + 15 (lane: column lane & 15, k = 8 (lane >> 4) .. + 7): lane 4q + p of a 16-lane group
// supplies &img[k = 8 lg + q][m0 + 4 p] and receives column (lane & 15) of 4 k-rows; the second read is 4 k-rows (2 048 B) on
// (same swizzle: bit 2 of k is not used).
__device__ __forceinline__ uint32_t frag_off(int m0, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    const int q = lr >> 2, pp = lane & 3;
    const int chunk = ((m0 >> 3) + (pp >> 1)) ^ (2 * q + 8 * (lg & 1));
    return (uint32_t)(((8 * lg + q) * TM + (chunk << 3) + 4 * (pp & 1)) * 2);
}
// The fragment reads are INLINE ASM: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every __builtin_amdgcn_ds_read_tr16_b64
// that follows an LDS-DMA in program order (it cannot tell the stages apart), which drains the three stages in flight at every
// step — the k-major instances of gemm_bf16_glds_kernel carry that wait, and lost to the register-staged kernel for it.  The
// compiler does not count asm reads in lgkmcnt either: the waits below are explicit, each one tied to the registers it
// releases ("+v") so that no MFMA can be scheduled above it.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ void tr_pair(u32x2& lo, u32x2& hi, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(addr));
}
__device__ __forceinline__ bf16x8 as_frag(u32x2 lo, u32x2 hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
#define ORTK_LGKM_WAIT(N, reg) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(reg))

// one stage: the wave's 128 x 64 sub-tile += A[32 k][128 m]^T B[32 k][64 n] (+ the column sums of its A rows)
// st: LDS byte address of the stage (A image, B image 16 KB behind); offA / offB: frag_off of the wave's 8 + 4 fragments
// cs_i: -1, or the row fragment (of each half: i and 4 + i) whose column sums THIS wave takes: the four column waves of a row half hold
// the same A fragments, so each takes a quarter of the bias gradient (one extra MFMA per 16 instead of four on one wave, which
// the other seven would wait for at every barrier)
__device__ __forceinline__ void multiply_stage(uint32_t st, const uint32_t (&offA)[8], const uint32_t (&offB)[4], int cs_i, const bf16x8& ones,
                                               f32x4 (&acc)[8][4], f32x4 (&acc_cs)[2]) {
    u32x2 blo[4], bhi[4], alo[8], ahi[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) tr_pair(blo[j], bhi[j], st + (uint32_t)(W_IMG * 2) + offB[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) tr_pair(alo[i], ahi[i], st + offA[i]);
    // 24 reads are out (the LDS returns them in order; the counter holds 15): 14 left = the B fragments and A fragment 0 are in
    bf16x8 b[4], a[8];
    ORTK_LGKM_WAIT(14, alo[0]); ORTK_LGKM_WAIT(14, ahi[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ORTK_LGKM_WAIT(14, blo[j]); ORTK_LGKM_WAIT(14, bhi[j]); b[j] = as_frag(blo[j], bhi[j]); }
#define ORTK_WG_ROW(I, N)                                                                                              \
    ORTK_LGKM_WAIT(N, alo[I]); ORTK_LGKM_WAIT(N, ahi[I]); a[I] = as_frag(alo[I], ahi[I]);                               \
    if (cs_i == (I & 3)) acc_cs[I >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[I], acc_cs[I >> 2], 0, 0, 0);   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[I], acc[I][j], 0, 0, 0);
    ORTK_WG_ROW(0, 14) ORTK_WG_ROW(1, 12) ORTK_WG_ROW(2, 10) ORTK_WG_ROW(3, 8)
    ORTK_WG_ROW(4, 6) ORTK_WG_ROW(5, 4) ORTK_WG_ROW(6, 2) ORTK_WG_ROW(7, 0)
#undef ORTK_WG_ROW
}

// ---- the main loop as a PING-PONG of the two waves of every SIMD (waves w and w + 4: the two row halves of the tile)
// A 512-thread workgroup puts two waves on each SIMD.  Run in lock step (one barrier per stage, ortk_wgrad_group_args.flags & 1) both
// read their fragments at the same time — the matrix pipe idles — and then both want the pipe.  Here waves 4-7 run ONE BARRIER
// BEHIND waves 0-3 and every wave alternates [read a stage's fragments] barrier [32 MFMAs] barrier: between two barriers one wave
// of a SIMD multiplies while its partner reads (cdna_hip_programming.md, "The 256^2 8-phase template": the same staggering).
// Measured alone on the chip (scratch/wgrad_lowwg.py, 56 workgroups of 520 stages): 920 ns per stage against 986 with half stages as
// phases (four barriers per stage) — 46 % of the nominal MFMA peak of the units the launch holds.
//
// Barrier j (counted over the kernel; j = 0 publishes stage 0) and what runs behind it, stage t:
//     waves 0-3:  j = 2t: read stage t | 2t+1: DMA t+3, MFMA t, wait(t+1) | 2t+2 ...
//     waves 4-7:  j = 2t+1: DMA t+3, read stage t, wait(t+1) | 2t+2: MFMA t | 2t+3 ...
// Stage t + 1 is read from barrier 2t + 2 on: every wave has waited for its own share of it before arriving there.  Slot
// (t + 3) & 3 held stage t - 1, whose last reads (waves 4-7, behind barrier 2t - 1) are complete before those waves' MFMAs behind
// barrier 2t: a DMA issued behind barrier 2t + 1 is safe.
struct Dma { const __bf16* A; const __bf16* B; int lda, ldb, k_begin; uint32_t offA[2], offB[2]; };
constexpr uint32_t STAGE_BYTES = 2 * W_IMG * 2;
__device__ __forceinline__ void issue_stage(const Dma& d, int t, __bf16* ring, int wave, int lane) {
    __bf16* st = ring + (size_t)(t & (WNS - 1)) * 2 * W_IMG;
    const int k = d.k_begin + t * WBK;
    dma_tile(d.A + (int64_t)k * d.lda, d.offA, st, wave);
    dma_tile(d.B + (int64_t)k * d.ldb, d.offB, st + W_IMG, wave);
}
// this wave's share of stage t has landed (stages t + 1 .. min(t + 2, T - 1) may stay in flight: 4 DMA instructions each)
__device__ __forceinline__ void wait_stage(int t, int T) {
    if (t >= T) return;
    const int ahead = min(T - 1 - t, WNS - 2);
    if (ahead == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int G>
__device__ __forceinline__ void pingpong2(const Dma& d, int T, __bf16* ring, uint32_t lds0, const uint32_t (&offA)[8], const uint32_t (&offB)[4],
                                          int cs_i, const bf16x8& ones, f32x4 (&acc)[8][4], f32x4 (&acc_cs)[2], int wave, int lane) {
    u32x2 blo[4], bhi[4], alo[8], ahi[8];
    if (G == 1) wg_barrier();
    for (int t = 0; t < T; ++t) {
        const uint32_t st = lds0 + (uint32_t)(t & (WNS - 1)) * STAGE_BYTES;
        if (G == 1 && t + WNS - 1 < T) issue_stage(d, t + WNS - 1, ring, wave, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_pair(blo[j], bhi[j], st + (uint32_t)(W_IMG * 2) + offB[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) tr_pair(alo[i], ahi[i], st + offA[i]);
        if (G == 1) wait_stage(t + 1, T);
        wg_barrier();
        if (G == 0 && t + WNS - 1 < T) issue_stage(d, t + WNS - 1, ring, wave, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ORTK_LGKM_WAIT(0, blo[j]); ORTK_LGKM_WAIT(0, bhi[j]); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { ORTK_LGKM_WAIT(0, alo[i]); ORTK_LGKM_WAIT(0, ahi[i]); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8 a = as_frag(alo[i], ahi[i]);
            if (cs_i == (i & 3)) acc_cs[i >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a, acc_cs[i >> 2], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(blo[j], bhi[j]), a, acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (G == 0) wait_stage(t + 1, T);
        wg_barrier();
    }
    if (G == 0) wg_barrier();
}

__global__ __launch_bounds__(512, 1) void wgrad_group_kernel(WgArgs p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ks_ = bid / p.tiles, t_ = bid - ks_ * p.tiles;
    int ii = 0;
#pragma unroll
    for (int i = 1; i < ORTK_WGRAD_MAX; ++i) if (i < p.n && t_ >= p.it[i].tile0) ii = i;
    const WgItem& it = p.it[ii];
    const int loc = t_ - it.tile0, nt = loc % it.tilesN, mt = loc / it.tilesN;
    const int mb = mt * TM, nb = nt * TM;
    const int k_begin = ks_ * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int T = (k_end - k_begin) / WBK, rem = (k_end - k_begin) - T * WBK;
    const __bf16* Ap = it.A; const __bf16* Bp = it.B;
    const int lda = it.lda, ldb = it.ldb, M = it.M, N = it.N;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cs_i = __builtin_amdgcn_readfirstlane((it.cs != nullptr && nt == 0) ? wn : -1);
    f32x4 acc_cs[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    uint32_t offA[8], offB[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) offA[i] = frag_off(wm * 128 + 16 * i, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) offB[j] = frag_off(wn * 64 + 16 * j, lane);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) __bf16*)smem16;

    const Dma dma{Ap, Bp, lda, ldb, k_begin, {src_off(lda, mb, M, wave * 2, lane), src_off(lda, mb, M, wave * 2 + 1, lane)},
                  {src_off(ldb, nb, N, wave * 2, lane), src_off(ldb, nb, N, wave * 2 + 1, lane)}};
    if (p.lockstep) {
        // measurement only (ortk_wgrad_group_args.flags & 1): both waves of a SIMD in the same phase, one barrier per stage
        for (int t = 0; t < WNS - 1 && t < T; ++t) issue_stage(dma, t, smem16, wave, lane);
        for (int t = 0; t < T; ++t) {
            wait_stage(t, T);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + WNS - 1 < T) issue_stage(dma, t + WNS - 1, smem16, wave, lane);      // every wave is past its reads of stage t - 1
            multiply_stage(lds0 + (uint32_t)(t & (WNS - 1)) * STAGE_BYTES, offA, offB, cs_i, ones, acc, acc_cs);
        }
    } else if (T > 0) {
        for (int t = 0; t < WNS - 1 && t < T; ++t) issue_stage(dma, t, smem16, wave, lane);
        wait_stage(0, T);
        wg_barrier();                                                       // barrier 0: stage 0 is published
        if (wm == 0) pingpong2<0>(dma, T, smem16, lds0, offA, offB, cs_i, ones, acc, acc_cs, wave, lane);
        else         pingpong2<1>(dma, T, smem16, lds0, offA, offB, cs_i, ones, acc, acc_cs, wave, lane);
    }
    if (rem > 0) {
        // slot T & 3 held stage T - 4 (read before the barrier of step T - 3): free for every wave
        __bf16* st = smem16 + (size_t)(T & (WNS - 1)) * 2 * W_IMG;
        guarded_tile(Ap, lda, mb, M, k_begin + T * WBK, k_end, st, wave, lane);
        guarded_tile(Bp, ldb, nb, N, k_begin + T * WBK, k_end, st + W_IMG, wave, lane);
        __syncthreads();
        multiply_stage(lds0 + (uint32_t)(T & (WNS - 1)) * STAGE_BYTES, offA, offB, cs_i, ones, acc, acc_cs);
    }
    if (cs_i >= 0 && lane < 16) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = mb + wm * 128 + 16 * (4 * h + cs_i) + lane;
            if (m < M) atomicAdd(it.cs + m, acc_cs[h][0]);
        }
    }
    float* Cp = it.C;
    const int ldc = it.ldc;
    if (p.slabs != nullptr || p.sk == 1) {
        // ---- epilogue without atomics: the K ranges of a tile meet in memory, the LAST one to arrive adds them up.
        // Every workgroup stores its 256 x 256 fp32 partial tile as a slab in register order (one contiguous KB per wave-instruction),
        // publishes it (every wave drains its stores, barrier, one agent-scope release, a ticket from the tile's counter), and leaves —
        // nobody waits for anybody, so the grid may be larger than what is resident.  The workgroup that draws the last ticket acquires,
        // adds the other slabs to its own accumulators and updates the arena with plain loads and stores: it is the only writer of
        // that tile (launches that share a weight block are ordered by their stream).  (cdna_hip_programming.md section 5, "In-launch
        // split-K reduction"; the 57 MB of split-K atomics of a decoder layer ran at the memory side's 1.3 TB/s: 45 of the launch's 165 us.)
        const int sk = p.sk;
        if (sk > 1) {
            // (write-through stores and sc1 loads instead of release / acquire fences: a release writes back the XCD's whole L2 — 7 MB
            //  of fresh slabs per XCD here — and made this epilogue slower than the atomics it replaces; MI355X_MICROARCH.md, "Valid
            //  forms": every byte stored sc1, every storing wave's vmcnt(0), the workgroup barrier, ONE lane's agent-scope add on the
            //  tile's one counter, the last adder's workgroup loads every byte sc1 behind a barrier; one workgroup per compute unit)
            typedef __amdgpu_buffer_rsrc_t rsrc_t;
            constexpr unsigned SLAB = TM * TM * 4;        // bytes
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slabs + (size_t)t_ * sk * (size_t)(TM * TM / 4), 0, (int)(sk * SLAB), 0x00020000);
            const unsigned mine = (unsigned)ks_ * SLAB + (unsigned)(wave * 32 * 64 + lane) * 16u;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, mine + (unsigned)(i * 4 + j) * 1024u, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(smem16);
            if (tid == 0) {
                const int ticket = __hip_atomic_fetch_add(p.tickets + t_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = ticket == sk - 1;
            }
            __syncthreads();
            if (!*flag) return;
            for (int s2 = 0; s2 < sk; ++s2) {
                if (s2 == ks_) continue;
                const unsigned theirs = (unsigned)s2 * SLAB + (unsigned)(wave * 32 * 64 + lane) * 16u;
                // 16 loads (64 registers) in flight per lane, twice: all 32 at once do not fit beside the accumulators
#pragma unroll
                for (int ih = 0; ih < 2; ++ih) {
                    u32x4 tmp[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) tmp[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, theirs + (unsigned)(ih * 16 + q) * 1024u, 0, 16);
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[ih * 4 + (q >> 2)][q & 3] += __builtin_bit_cast(f32x4, tmp[q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // dW += the tile: lane = row 16 i + (lane & 15), four consecutive columns
        const int m0 = mb + wm * 128 + (lane & 15), n0 = nb + wn * 64 + 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + 16 * i;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + 16 * j;
                if (n >= N) continue;              // (N is a multiple of 8: the four columns are in or out together)
                f32x4* c = reinterpret_cast<f32x4*>(Cp + (int64_t)m * ldc + n);
                if ((reinterpret_cast<uintptr_t>(c) & 15) == 0) *c += acc[i][j];
                else { float* cf = reinterpret_cast<float*>(c); for (int r = 0; r < 4; ++r) cf[r] += acc[i][j][r]; }
            }
        }
        return;
    }
    // ---- epilogue with atomics (no workspace given): two halves of 128 output rows through LDS -> 256-byte atomic rows
    float* sC = reinterpret_cast<float*>(smem16);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                   // the ring (h = 0) / the previous half's reads are done
        if (wm == h) {
            const int mr = lane & 15, nc = wn * 64 + 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(sC + (mr + 16 * i) * WCP + nc + 16 * j) = acc[i][j];
        }
        __syncthreads();
        // 512 row segments of 64 columns, 64 per wave
        for (int q = 0; q < 64; ++q) {
            const int seg = q * 8 + wave, r = seg >> 2, c = (seg & 3) * 64 + lane;
            const int m = mb + h * 128 + r, n = nb + c;
            if (m < M && n < N) atomicAdd(Cp + (int64_t)m * ldc + n, sC[r * WCP + c]);
        }
    }
}

}  // namespace

extern "C" int ortk_wgrad_group(const ortk_wgrad_group_args* a, ortk_stream stream) {
    if (!a || a->n < 1 || a->n > ORTK_WGRAD_MAX || a->rows < 0 || a->rows > 0x7FFFFFFF || a->splitk < 0 || (a->flags & ~9)) return ORTK_EINVAL;
    if (a->rows == 0) return 0;
    WgArgs p; std::memset(&p, 0, sizeof(p));
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    int tiles = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < a->n; ++i) {
        const ortk_wgrad_item& s = a->item[i];
        if (!s.dY || !s.X || !s.dW || s.Nout < 8 || s.Kin < 8 || (s.Nout & 7) || (s.Kin & 7) || (s.lddy & 7) || (s.ldx & 7) ||
            s.lddy < s.Nout || s.ldx < s.Kin || s.lddw < s.Kin || s.lddy > 0x7FFFFFFF || s.ldx > 0x7FFFFFFF || s.lddw > 0x7FFFFFFF ||
            !al16(s.dY) || !al16(s.X) || (reinterpret_cast<uintptr_t>(s.dW) & 3)) return ORTK_EINVAL;
        WgItem& d = p.it[i];
        d.A = reinterpret_cast<const __bf16*>(s.dY); d.B = reinterpret_cast<const __bf16*>(s.X); d.C = s.dW; d.cs = s.db;
        d.lda = (int)s.lddy; d.ldb = (int)s.ldx; d.ldc = (int)s.lddw; d.M = s.Nout; d.N = s.Kin;
        d.tilesN = (int)ortk_cdiv(s.Kin, TM); d.tile0 = tiles;
        tiles += (int)ortk_cdiv(s.Nout, TM) * d.tilesN;
        flops += 2.0 * (double)a->rows * s.Nout * s.Kin;
        bytes += (double)a->rows * (s.Nout + s.Kin) * 2 + (double)s.Nout * s.Kin * 4;
    }
    p.n = a->n; p.K = (int)a->rows; p.tiles = tiles; p.lockstep = a->flags & 1;
    // K ranges: one round of workgroups (256 compute units) when the tiles allow it, never fewer than 8 stages per workgroup
    const int ksteps = (int)ortk_cdiv(a->rows, WBK);
    int sk = a->splitk > 0 ? a->splitk : (tiles >= 256 ? 1 : 256 / tiles);
    sk = std::max(1, std::min(sk, std::max(1, ksteps / 8)));
    p.kchunk = (int)ortk_cdiv(ksteps, sk) * WBK;
    sk = (int)ortk_cdiv(a->rows, p.kchunk);
    hipStream_t s = ortk_s(stream);
    p.sk = sk;
    if (sk > 1 && a->ws && !(a->flags & 8)) {
        // workspace: the tiles' tickets (zeroed here, in stream order), then the slabs
        const size_t tk = ((size_t)tiles * sizeof(int) + 255) & ~(size_t)255;
        const size_t need = tk + (size_t)tiles * sk * TM * TM * sizeof(float);
        if ((reinterpret_cast<uintptr_t>(a->ws) & 255) || a->ws_bytes < need) return ORTK_ENOSPC;
        p.tickets = reinterpret_cast<int*>(a->ws);
        p.slabs = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a->ws) + tk);
        if (hipMemsetAsync(p.tickets, 0, (size_t)tiles * sizeof(int), s) != hipSuccess) return ORTK_EINVAL;
    }
    if (ortk::lds_attr(reinterpret_cast<const void*>(wgrad_group_kernel), W_LDS_BYTES)) return ORTK_EINVAL;
    ortk::ProfMark pm;
    ortk::prof_begin(ortk::PROF_KEY_WGRAD_GROUP, flops, bytes, s, pm);
    pm.units = (double)tiles * sk;
    hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)(tiles * sk)), dim3(512), W_LDS_BYTES, s, p);
    ortk::prof_end(pm, s);
    ORTK_CHECK_LAUNCH();
    return 0;
}

// bytes of ortk_wgrad_group_args.ws that let the launch reduce its K ranges without atomics (0: a single K range, nothing needed)
extern "C" size_t ortk_wgrad_group_workspace_bytes(const ortk_wgrad_group_args* a) {
    if (!a || a->n < 1 || a->n > ORTK_WGRAD_MAX || a->rows <= 0) return 0;
    int64_t tiles = 0;
    for (int i = 0; i < a->n; ++i) tiles += ortk_cdiv(a->item[i].Nout, TM) * ortk_cdiv(a->item[i].Kin, TM);
    if (tiles <= 0) return 0;
    const int ksteps = (int)ortk_cdiv(a->rows, WBK);
    int64_t sk = a->splitk > 0 ? a->splitk : (tiles >= 256 ? 1 : 256 / tiles);
    sk = std::max<int64_t>(1, std::min<int64_t>(sk, std::max(1, ksteps / 8)));
    const int64_t kchunk = ortk_cdiv(ksteps, sk) * WBK;
    sk = ortk_cdiv(a->rows, kchunk);
    if (sk <= 1) return 0;
    return (((size_t)tiles * sizeof(int) + 255) & ~(size_t)255) + (size_t)tiles * sk * TM * TM * sizeof(float);
}

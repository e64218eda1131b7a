// ortk_sparse.hip — sparse x dense products for pruned weights:  Y = epi(X . W~^T),  W~ (N outputs, K inputs) sparse.
//
// The reference multiplies by the zero-filled weight s (.) W in every masked layer (pruning/masked_layer.py:84-110,
// 134-135) and evaluates pruned checkpoints as DENSE linears on zero-filled weights (scripts/eval_model.py:64-88).  At
// 95 % sparsity only 5 % of those multiply-adds touch a non-zero.  Two formats (include/ortk.h: ortk_sparse_plan), both
// rebuilt on the device from the effective weights of every call (a new Bernoulli mask sample per training step):
//
// GU16 ("group union", mixed precision — the fast path).  For every group of 16 output columns and every chunk of 512 input
//   columns the builder lists the input columns in which ANY of the 16 outputs is non-zero (56 % of K at 95 % sparsity, 33 %
//   at 97.5 %) and packs the 16 x |union| weights in MFMA operand order.  The product kernel keeps a [k][rows] bf16 tile of X
//   in LDS and runs DENSE v_mfma_f32_16x16x32_bf16 over the compacted K: the weight fragment comes straight from global
//   memory (1 KB per wave and k-step, shared by all row tiles of the workgroup), the activation fragment is GATHERED from the
//   LDS tile by ds_read_b64_tr_b16 — every lane supplies the address of its own k-row, so the transposing read is a row
//   gather for free.  The builder orders each union so that the 8 k-rows a 32-lane half reads have distinct k mod 8 and the
//   tile's row pitch is 32 bytes mod 256: bank-conflict-free gathers.  Outputs leave the accumulators with 4 consecutive
//   columns per lane (the dense GEMM's layout): no LDS epilogue.  Bounds: MFMA work = |union| / K of the dense product;
//   L2 -> CU weight stream = compacted weights once per row tile (up to 128 rows).
// ELL ("sorted padded ELL": lane = output column, VALU products, LDS gathers with ds_read_b128).  ELL32 is the fp32 parity
//   mode; ELL16 (bf16 pairs, v_dot2c) is kept as the reference point the GU16 kernel is measured against: it is bound by
//   VALU issue (4 cycles per wave instruction) and by 3-way bank conflicts of random 16-byte gathers and stays BELOW the dense
//   MFMA GEMM at 95 % (profiles/r02_spmm_*).
#include <algorithm>
#include "ortk_common.h"
#include "ortk_internal.h"

namespace {

constexpr int RANGE = 512;             // output columns per workgroup = sort range of the format (8 chunks of 64)
constexpr int OP = RANGE + 4;          // fp32 row pitch of the staged output tile
constexpr int KMAX = 2048;             // offsets are slot*16 < 65536 and two planes must fit the LDS beside the output tile
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__host__ __device__ inline int ell_slot(int k) { return (k & ~7) | ((k + (k >> 3)) & 7); }

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned int)b << 16); }
__device__ __forceinline__ unsigned int f32_to_bf16_bits(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }

// 8 consecutive elements X[row][k0 .. k0+8) as floats, zeros beyond K
__device__ __forceinline__ void load8(const void* X, int dt, int64_t ld, int64_t row, int k0, int K, float (&v)[8]) {
    const int64_t i0 = row * ld + k0;
    if (dt == ORTK_BF16) {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(X) + i0;
        if (k0 + 8 <= K && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
            for (int d = 0; d < 4; ++d) { v[2 * d] = __uint_as_float(t[d] << 16); v[2 * d + 1] = __uint_as_float(t[d] & 0xFFFF0000u); }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = k0 + i < K ? bf16_bits_to_f32(p[i]) : 0.f;
        }
    } else {
        const float* p = reinterpret_cast<const float*>(X) + i0;
        if (k0 + 8 <= K && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = k0 + i < K ? p[i] : 0.f;
        }
    }
}

struct SpmmP {
    const void* stream; const int32_t* chunk_ptr; const int32_t* chunk_len; const int32_t* perm;   // already at the block's chunk0
    int32_t N, K, nchunks, nranges;
    ortk_spmm_args a;
};

// 4 consecutive fp32 of a row (zero beyond N)
__device__ __forceinline__ void ld4f(const float* __restrict__ p, int n0, int N, bool vec, float (&o)[4]) {
    if (vec && n0 + 3 < N) { const f32x4 t = *reinterpret_cast<const f32x4*>(p); o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; return; }
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = n0 + q < N ? p[q] : 0.f;
}

// one PAIR of entries of a bf16 chunk: e = {off1 | off2 << 16, bf16 w1 | bf16 w2 << 16}; 16 rows (two planes of 8):
// acc[r] += w1 * X[r, k1] + w2 * X[r, k2]   (v_dot2c_f32_bf16: fp32 accumulate of the two exact bf16 products)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ float dot2(unsigned int x, unsigned int w, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x), __builtin_bit_cast(bf16x2, w), c, false);
}
// The gathers of a pair and its arithmetic are separate steps so that the loop can keep the NEXT pairs' gathers in flight
// under the arithmetic of the current ones (PMC of the un-pipelined loop: LDS 50 % busy, VALU 47 % busy, and the two added
// up to the run time — every wave of a CU gathered, then every wave computed).
struct PairRows { u32x4 a0, a1, b0, b1; };
template <int PB>
__device__ __forceinline__ PairRows gather_pair(const unsigned char* planes, unsigned int offs) {
    const unsigned int o1 = offs & 0xFFFFu, o2 = offs >> 16;
    PairRows g;
    g.a0 = *reinterpret_cast<const u32x4*>(planes + o1); g.a1 = *reinterpret_cast<const u32x4*>(planes + PB + o1);
    g.b0 = *reinterpret_cast<const u32x4*>(planes + o2); g.b1 = *reinterpret_cast<const u32x4*>(planes + PB + o2);
    return g;
}
__device__ __forceinline__ void fma_pair16(const PairRows& g, unsigned int w, float (&acc)[16]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        // dword d of a plane read = rows (2d, 2d+1) of that input column: lo halves -> row 2d, hi halves -> row 2d+1
        acc[2 * d] = dot2(__builtin_amdgcn_perm(g.b0[d], g.a0[d], 0x05040100u), w, acc[2 * d]);
        acc[2 * d + 1] = dot2(__builtin_amdgcn_perm(g.b0[d], g.a0[d], 0x07060302u), w, acc[2 * d + 1]);
        acc[8 + 2 * d] = dot2(__builtin_amdgcn_perm(g.b1[d], g.a1[d], 0x05040100u), w, acc[8 + 2 * d]);
        acc[8 + 2 * d + 1] = dot2(__builtin_amdgcn_perm(g.b1[d], g.a1[d], 0x07060302u), w, acc[8 + 2 * d + 1]);
    }
}
// fp32 parity mode: 8 rows (two planes of 4)
template <int PB>
__device__ __forceinline__ void fma_entry8(const unsigned char* planes, u32x2 w, float (&acc)[8]) {
    const float val = __uint_as_float(w[1]);
    const f32x4 a = *reinterpret_cast<const f32x4*>(planes + w[0]);
    const f32x4 b = *reinterpret_cast<const f32x4*>(planes + PB + w[0]);
#pragma unroll
    for (int d = 0; d < 4; ++d) { acc[d] = fmaf(val, a[d], acc[d]); acc[4 + d] = fmaf(val, b[d], acc[4 + d]); }
}

// XT = __bf16: 4-byte entries, bf16 activation planes, 16 rows per workgroup;  XT = float: 8-byte entries, fp32 planes, 8 rows.
// KT: compile-time plane capacity (input columns): the second plane is an immediate offset of the first.
// NT: threads per workgroup (512 on short grids: one chunk per wave instead of two halves the latency of a workgroup).
// A workgroup stages its X tile once and then walks `rpw` consecutive 512-column ranges (long grids: the staging and the
// redundant X reads are shared by up to 4 ranges).
// ALIAS (one range per workgroup, rpw == 1): the staged output tile lies OVER the planes — every wave keeps the accumulators of its
// chunks (fixed assignment: chunk w, w + waves, ..) in registers until the whole workgroup is done gathering — so a 2 048-column tile
// takes 64 KB instead of 97 and two workgroups share a compute unit: one stages its X tile while the other gathers (the wide-input
// data-gradient products, one workgroup per CU before, were a chain of stage -> gather -> store per compute unit).
constexpr int MAXRPW = 32;
template <typename XT, int KT, int NT, bool ALIAS>
__global__ __launch_bounds__(NT) void spmm_ell_kernel(SpmmP p, int rpw, int ngroups) {
    constexpr bool F32 = sizeof(XT) == 4;
    constexpr int RB = F32 ? 8 : 16;
    constexpr int NPAIR = RB / 2;
    constexpr int PB = KT * 16;
    constexpr int CPW = ALIAS ? (RANGE / 64) / (NT / 64) : 1;      // chunks per wave held in registers
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* planes = smem;
    constexpr size_t TILE_B = (size_t)RB * OP * sizeof(float);
    float* sOut = reinterpret_cast<float*>(smem + (ALIAS ? 0 : 2 * PB));
    int* sNext = reinterpret_cast<int*>(smem + (ALIAS ? ((size_t)2 * PB > TILE_B ? (size_t)2 * PB : TILE_B) : (size_t)2 * PB + TILE_B));
    const int tid = threadIdx.x, lane = tid & 63;
    const int rg0 = (blockIdx.x % ngroups) * rpw;
    const int64_t m0 = (int64_t)(blockIdx.x / ngroups) * RB;
    const ortk_spmm_args& a = p.a;
    if (tid < MAXRPW) sNext[tid] = 0;
    // ---- stage the X tile transposed: plane h, slot(k): rows 8h..8h+7 (bf16) / 4h..4h+3 (fp32) of input column k
    const int Kp = (p.K + 7) & ~7;
    const bool xfast = !F32 && a.x_dtype == ORTK_BF16 && (p.K & 7) == 0 && (a.ldx & 7) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
    if (xfast) {
        // bf16 rows straight into bf16 planes: the two rows of a pair are interleaved with byte permutes, no float round trip
        const unsigned short* X16 = reinterpret_cast<const unsigned short*>(a.X);
        for (int task = tid; task < NPAIR * (Kp >> 3); task += NT) {
            const int pr = task % NPAIR, q = task / NPAIR;
            const int64_t r0 = min(m0 + 2 * pr, a.M - 1), r1 = min(m0 + 2 * pr + 1, a.M - 1);
            const u32x4 t0 = *reinterpret_cast<const u32x4*>(X16 + r0 * a.ldx + 8 * q);
            const u32x4 t1 = *reinterpret_cast<const u32x4*>(X16 + r1 * a.ldx + 8 * q);
            unsigned char* dst = planes + (pr >> 2) * PB + (pr & 3) * 4 + 8 * q * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned int w = __builtin_amdgcn_perm(t1[i >> 1], t0[i >> 1], (i & 1) ? 0x07060302u : 0x05040100u);
                *reinterpret_cast<unsigned int*>(dst + ((i + q) & 7) * 16) = w;          // slot 8 q + ((i + q) & 7) == ell_slot(8 q + i)
            }
        }
    } else {
        for (int task = tid; task < NPAIR * (Kp >> 3); task += NT) {
            const int pr = task % NPAIR, q = task / NPAIR;
            const int64_t r0 = min(m0 + 2 * pr, a.M - 1), r1 = min(m0 + 2 * pr + 1, a.M - 1);
            float x0[8], x1[8];
            load8(a.X, a.x_dtype, a.ldx, r0, 8 * q, p.K, x0);
            load8(a.X, a.x_dtype, a.ldx, r1, 8 * q, p.K, x1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int slot = 8 * q + ((i + q) & 7);            // == ell_slot(8 q + i)
                if (!F32) {
                    const unsigned int w = f32_to_bf16_bits(x0[i]) | (f32_to_bf16_bits(x1[i]) << 16);
                    *reinterpret_cast<unsigned int*>(planes + (pr >> 2) * PB + slot * 16 + (pr & 3) * 4) = w;
                } else {
                    *reinterpret_cast<f32x2*>(planes + (pr >> 1) * PB + slot * 16 + (pr & 1) * 8) = (f32x2){x0[i], x1[i]};
                }
            }
        }
    }
    __syncthreads();
    const bool al_b = !a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0;
    const bool al_r = !a.resid || ((reinterpret_cast<uintptr_t>(a.resid) & 15) == 0 && (a.ldr & 3) == 0);
    const bool al_g = a.gate && (reinterpret_cast<uintptr_t>(a.gate) & (a.gate_dtype == ORTK_BF16 ? 7 : 15)) == 0 && (a.ldg & 3) == 0;
    const bool al_y = (reinterpret_cast<uintptr_t>(a.Y) & (a.y_dtype == ORTK_BF16 ? 7 : 15)) == 0 && (a.ldy & 3) == 0;
    // lean epilogue: bias / ReLU / residual only, everything vector-aligned, whole groups of 4 columns
    const bool lean = !a.gate && a.drop_p == 0.f && !a.rowscale && al_b && al_r && al_y && (p.N & 3) == 0;
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    for (int ri = 0; ri < rpw; ++ri) {
        const int rg = rg0 + ri;
        if (rg >= p.nranges) break;
        // ---- lane = output column; chunks of this range are taken longest first
        const int c_begin = rg * (RANGE / 64), c_end = min(p.nchunks, c_begin + RANGE / 64);
        // acc[r] = the chunk's 64 output columns (lane = column) of row r of the tile
        auto run_chunk = [&](int c, float (&acc)[RB]) {
            const int len = p.chunk_len[c];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = 0.f;
            if (len > 0) {
                if constexpr (!F32) {
                    // entry pairs (8 bytes) of lane l: pair jp at (chunk_ptr / 2 + jp * 64 + l); len is a multiple of 8 entries
                    const u32x2* e = reinterpret_cast<const u32x2*>(p.stream) + (p.chunk_ptr[c] >> 1) + lane;
                    u32x2 w0 = e[0], w1 = e[64], w2 = e[128], w3 = e[192];
                    PairRows ga = gather_pair<PB>(planes, w0[0]), gb = gather_pair<PB>(planes, w1[0]);
                    for (int j = 0; j < len; j += 8) {
                        const bool more = j + 8 < len;
                        u32x2 n0 = {0u, 0u}, n1 = {0u, 0u}, n2 = {0u, 0u}, n3 = {0u, 0u};
                        if (more) {
                            const u32x2* en = e + (int64_t)((j + 8) >> 1) * 64;
                            n0 = en[0]; n1 = en[64]; n2 = en[128]; n3 = en[192];
                        }
                        const PairRows gc = gather_pair<PB>(planes, w2[0]), gd = gather_pair<PB>(planes, w3[0]);
                        fma_pair16(ga, w0[1], acc); fma_pair16(gb, w1[1], acc);
                        // (a zero pair gathers slot 0: harmless, and keeps the loop body branch-free apart from the stream load)
                        ga = gather_pair<PB>(planes, n0[0]); gb = gather_pair<PB>(planes, n1[0]);
                        fma_pair16(gc, w2[1], acc); fma_pair16(gd, w3[1], acc);
                        w0 = n0; w1 = n1; w2 = n2; w3 = n3;
                    }
                } else {
                    const u32x2* e = reinterpret_cast<const u32x2*>(p.stream) + p.chunk_ptr[c] + lane;
                    u32x2 w0 = e[0], w1 = e[64], w2 = e[128], w3 = e[192];
                    for (int j = 0; j < len; j += 4) {
                        u32x2 n0 = {0u, 0u}, n1 = {0u, 0u}, n2 = {0u, 0u}, n3 = {0u, 0u};
                        if (j + 4 < len) {
                            const u32x2* en = e + (int64_t)(j + 4) * 64;
                            n0 = en[0]; n1 = en[64]; n2 = en[128]; n3 = en[192];
                        }
                        fma_entry8<PB>(planes, w0, acc); fma_entry8<PB>(planes, w1, acc);
                        fma_entry8<PB>(planes, w2, acc); fma_entry8<PB>(planes, w3, acc);
                        w0 = n0; w1 = n1; w2 = n2; w3 = n3;
                    }
                }
            }
        };
        auto put_chunk = [&](int c, const float (&acc)[RB]) {
            const int col = p.perm[c * 64 + lane];
            if (col >= 0) {
                const int nloc = col - rg * RANGE;
#pragma unroll
                for (int r = 0; r < RB; ++r) sOut[r * OP + nloc] = acc[r];
            }
        };
        if constexpr (ALIAS) {
            float hold[CPW][RB];
#pragma unroll
            for (int ci = 0; ci < CPW; ++ci) {
                const int c = c_begin + ci * (NT / 64) + (tid >> 6);
                if (c < c_end) run_chunk(c, hold[ci]);
            }
            __syncthreads();                      // every wave is done with the planes the tile overwrites
#pragma unroll
            for (int ci = 0; ci < CPW; ++ci) {
                const int c = c_begin + ci * (NT / 64) + (tid >> 6);
                if (c < c_end) put_chunk(c, hold[ci]);
            }
        } else {
            for (;;) {
                int c = 0;
                if (lane == 0) c = atomicAdd(sNext + ri, 1);
                c = __builtin_amdgcn_readfirstlane(c) + c_begin;
                if (c >= c_end) break;
                float acc[RB];
                run_chunk(c, acc);
                put_chunk(c, acc);
            }
        }
        __syncthreads();
        // ---- epilogue on whole row segments (same order of operations as the dense GEMM epilogue, ortk_gemm.hip)
        if (lean) {
            for (int task = tid; task < RB * (RANGE / 4); task += NT) {
                const int r = task >> 7, c4 = (task & 127) * 4;
                const int64_t m = m0 + r;
                const int n0 = rg * RANGE + c4;
                if (m >= a.M || n0 >= p.N) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(sOut + r * OP + c4);
                if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n0);
                if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + m * a.ldr + n0);
                st_elem4(a.Y, m * a.ldy + n0, a.y_dtype, make_float4(v[0], v[1], v[2], v[3]));
            }
        } else {
            for (int task = tid; task < RB * (RANGE / 4); task += NT) {
                const int r = task >> 7, c4 = (task & 127) * 4;
                const int64_t m = m0 + r;
                const int n0 = rg * RANGE + c4;
                if (m >= a.M || n0 >= p.N) continue;
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sOut + r * OP + c4);
                float bb[4] = {0.f, 0.f, 0.f, 0.f}, rr[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {1.f, 1.f, 1.f, 1.f};
                if (a.bias) ld4f(a.bias + n0, n0, p.N, al_b, bb);
                if (a.resid) ld4f(a.resid + m * a.ldr + n0, n0, p.N, al_r, rr);
                if (a.gate) {
                    if (a.gate_dtype == ORTK_F32) ld4f(reinterpret_cast<const float*>(a.gate) + m * a.ldg + n0, n0, p.N, al_g, gg);
                    else if (al_g && n0 + 3 < p.N) { const float4 t = ld_elem4(a.gate, m * a.ldg + n0, ORTK_BF16); gg[0] = t.x; gg[1] = t.y; gg[2] = t.z; gg[3] = t.w; }
                    else for (int q = 0; q < 4; ++q) if (n0 + q < p.N) gg[q] = ld_elem(a.gate, m * a.ldg + n0 + q, ORTK_BF16);
                }
                const float rs = a.rowscale ? a.rowscale[m] : 1.f;
                bool kp[4] = {true, true, true, true};
                if (a.drop_p > 0.f) ortk_keep4(a.drop_seed, (uint64_t)(a.drop_rows ? (int64_t)a.drop_rows[m] : (int64_t)m) * (uint64_t)p.N + n0, a.drop_p, kp);
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = s4[q] + bb[q];
                    if (a.relu) x = fmaxf(x, 0.f);
                    x *= rs;
                    if (a.drop_p > 0.f) x = kp[q] ? x * inv_keep : 0.f;
                    if (a.gate) x = gg[q] > 0.f ? x * a.gate_scale : 0.f;
                    v[q] = x + rr[q];
                }
                const int64_t yi = m * a.ldy + n0;
                if (al_y && n0 + 3 < p.N) st_elem4(a.Y, yi, a.y_dtype, make_float4(v[0], v[1], v[2], v[3]));
                else for (int q = 0; q < 4; ++q) if (n0 + q < p.N) st_elem(a.Y, yi + q, a.y_dtype, v[q]);
            }
        }
        if (ri + 1 < rpw) __syncthreads();       // the next range's chunks overwrite the output tile
    }
}

// ------------------------------------------------------------------------------------------------ builder
__device__ __forceinline__ int find_block_by_row(const ortk_sparse_block* __restrict__ blocks, int nblocks, int64_t g) {
    int b = 0;
    for (int i = 1; i < nblocks; ++i) if ((int64_t)blocks[i].row0 <= g) b = i;     // blocks are ordered by row0
    return b;
}
__device__ __forceinline__ bool elem_nz(const void* base, int dt, int64_t i) {
    if (dt == ORTK_BF16) return (reinterpret_cast<const unsigned short*>(base)[i] & 0x7FFFu) != 0;
    return reinterpret_cast<const float*>(base)[i] != 0.f;
}

// one wave per output column (= row of the dense block): number of non-zeros
__global__ __launch_bounds__(256) void ell_count_kernel(const ortk_sparse_block* __restrict__ blocks, int nblocks, const void* dense, int dt,
                                                        int32_t* __restrict__ cnt, int64_t total_rows) {
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (g >= total_rows) return;
    const ortk_sparse_block& bk = blocks[find_block_by_row(blocks, nblocks, g)];
    const int64_t row = bk.src_offset + (g - bk.row0) * bk.ld;
    int n = 0;
    const size_t es = dt == ORTK_BF16 ? 2 : 4;
    const bool vec = (bk.K & 7) == 0 && ((reinterpret_cast<uintptr_t>(dense) + (size_t)row * es) & 15) == 0 && ((bk.ld * es) & 15) == 0;
    if (vec && dt == ORTK_BF16) {
        for (int k = lane * 8; k < bk.K; k += 512) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(dense) + row + k);
#pragma unroll
            for (int d = 0; d < 4; ++d) n += ((t[d] & 0x7FFFu) != 0) + ((t[d] & 0x7FFF0000u) != 0);
        }
    } else if (vec) {
        for (int k = lane * 4; k < bk.K; k += 256) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dense) + row + k);
            n += (t[0] != 0.f) + (t[1] != 0.f) + (t[2] != 0.f) + (t[3] != 0.f);
        }
    } else {
        for (int k = lane; k < bk.K; k += 64) n += elem_nz(dense, dt, row + k) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0) cnt[g] = n;
}

// one workgroup per (block, 512-column range): order the range's columns by count (descending); the chunk lengths (the longest
// member of every 64 columns, rounded up to the entry granule) go to chunk_len.  (Round 3 ran ONE workgroup per block over all its
// ranges — the generator's 20 ranges took 0.47 ms per build, 1 ms of every sparse training step.)
__global__ __launch_bounds__(256) void ell_rank_kernel(const ortk_sparse_block* __restrict__ blocks, int nblocks, const int32_t* __restrict__ cnt,
                                                       int32_t* __restrict__ chunk_len, int32_t* __restrict__ perm, int gran) {
    __shared__ int key[RANGE];
    int b = 0, first = 0;                       // block of this range, index of the block's first range
    for (; b < nblocks; ++b) {
        const int nr = (blocks[b].N + RANGE - 1) / RANGE;
        if ((int)blockIdx.x < first + nr) break;
        first += nr;
    }
    if (b >= nblocks) return;
    const ortk_sparse_block bk = blocks[b];
    const int tid = threadIdx.x;
    const int nch = (bk.N + 63) >> 6;
    const int n0 = ((int)blockIdx.x - first) * RANGE;
    const int nn = min(RANGE, bk.N - n0);
    for (int i = tid; i < RANGE; i += 256) key[i] = i < nn ? cnt[bk.row0 + n0 + i] : -1;
    __syncthreads();
    const int slots = min(RANGE, nch * 64 - n0);            // lane slots of this range (the last chunk is padded to 64)
    for (int i = tid; i < RANGE; i += 256) {
        if (i < nn) {
            const int ki = key[i];
            int rank = 0;
            for (int j = 0; j < nn; ++j) { const int kj = key[j]; rank += (kj > ki || (kj == ki && j < i)) ? 1 : 0; }
            perm[(int64_t)bk.chunk0 * 64 + n0 + rank] = n0 + i;
            if ((rank & 63) == 0) chunk_len[bk.chunk0 + ((n0 + rank) >> 6)] = (ki + gran - 1) & ~(gran - 1);
        } else if (i < slots) {
            perm[(int64_t)bk.chunk0 * 64 + n0 + i] = -1;
        }
    }
}
// one thread per block: chunk offsets inside the block's share of the entry stream; a block denser than its capacity is cut (and
// reported through the sticky overflow word)
__global__ __launch_bounds__(64) void ell_offsets_kernel(const ortk_sparse_block* __restrict__ blocks, int nblocks, int32_t* __restrict__ chunk_ptr,
                                                         int32_t* __restrict__ chunk_len, int32_t* overflow, int gran) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nblocks) return;
    const ortk_sparse_block bk = blocks[b];
    const int nch = (bk.N + 63) >> 6;
    int64_t off = bk.stream_offset;
    const int64_t end = bk.stream_offset + bk.capacity;
    for (int c = 0; c < nch; ++c) {
        int l = chunk_len[bk.chunk0 + c];
        if (off + (int64_t)l * 64 > end) {
            l = (int)(((end - off) / 64) & ~(int64_t)(gran - 1));
            if (l < 0) l = 0;
            *overflow = 1;
            chunk_len[bk.chunk0 + c] = l;
        }
        chunk_ptr[bk.chunk0 + c] = (int32_t)off;
        off += (int64_t)l * 64;
    }
}

// one wave per lane slot of a chunk: compact the non-zeros of its column into entries j*64 + slot, zero-pad to the chunk length
template <int EB>
__global__ __launch_bounds__(256) void ell_fill_kernel(const ortk_sparse_block* __restrict__ blocks, int nblocks, const void* dense, int dt,
                                                       const int32_t* __restrict__ chunk_ptr, const int32_t* __restrict__ chunk_len,
                                                       const int32_t* __restrict__ perm, void* stream, int64_t total_slots) {
    const int64_t gs = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (gs >= total_slots) return;
    int b = 0;
    for (int i = 1; i < nblocks; ++i) if ((int64_t)blocks[i].chunk0 * 64 <= gs) b = i;       // ordered by chunk0
    const ortk_sparse_block& bk = blocks[b];
    const int col = perm[gs];
    if (col < 0) return;
    const int c = (int)(gs >> 6), slot = (int)(gs & 63);
    const int64_t start = chunk_ptr[c];
    const int len = chunk_len[c];
    const int64_t row = bk.src_offset + (int64_t)col * bk.ld;
    int j = 0;
    for (int k0 = 0; k0 < bk.K; k0 += 64) {
        const int k = k0 + lane;
        unsigned int bits = 0; bool nz = false;
        if (k < bk.K) {
            if (dt == ORTK_BF16) {
                const unsigned short h = reinterpret_cast<const unsigned short*>(dense)[row + k];
                nz = (h & 0x7FFFu) != 0;
                bits = EB == 4 ? (unsigned int)h : ((unsigned int)h << 16);
            } else {
                const float f = reinterpret_cast<const float*>(dense)[row + k];
                nz = f != 0.f;
                bits = EB == 4 ? f32_to_bf16_bits(f) : __float_as_uint(f);
            }
        }
        const unsigned long long mask = __ballot(nz);
        const int pos = j + __popcll(mask & ((1ull << lane) - 1ull));
        if (nz && pos < len) {
            if (EB == 4) {     // pair pos/2 of this lane: halfwords {off(even), off(odd), w(even), w(odd)}
                unsigned short* q = reinterpret_cast<unsigned short*>(stream) + ((start >> 1) + (int64_t)(pos >> 1) * 64 + slot) * 4 + (pos & 1);
                q[0] = (unsigned short)(ell_slot(k) * 16); q[2] = (unsigned short)bits;
            } else {
                reinterpret_cast<u32x2*>(stream)[start + (int64_t)pos * 64 + slot] = (u32x2){(unsigned int)(ell_slot(k) * 16), bits};
            }
        }
        j += __popcll(mask);
    }
    for (int pos = j + lane; pos < len; pos += 64) {
        if (EB == 4) {
            unsigned short* q = reinterpret_cast<unsigned short*>(stream) + ((start >> 1) + (int64_t)(pos >> 1) * 64 + slot) * 4 + (pos & 1);
            q[0] = 0; q[2] = 0;
        } else {
            reinterpret_cast<u32x2*>(stream)[start + (int64_t)pos * 64 + slot] = (u32x2){0u, 0u};
        }
    }
}

// ================================================================================================ GU16
constexpr int GKC = 512;        // input columns per LDS chunk
constexpr int GSTEPS = 16;      // k-steps reserved per (group, chunk) slot: 8 residue classes x 64 members / 32 = worst case
typedef __attribute__((address_space(3))) void lds_ptr_t;

__device__ __forceinline__ bf16x4 tr_read4(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(p));
}

// One wave per (block, group of 16 outputs, chunk of 512 inputs).  Lane l owns the input columns 8 l .. 8 l + 7 of the chunk:
// the residue class k mod 8 of its i-th column is i, so "position of a column inside its residue class" is a prefix
// population count of one ballot.  Class i, position pos -> k-step pos >> 2, lane group lg = pos & 3, element
// j = (lg & 1) ? (i + 4) & 7 : i   (entry e = 8 lg + j of the step has residue (j + 4 (lg & 1)) & 7: see ortk.h).
__global__ __launch_bounds__(256) void gu_build_kernel(const ortk_sparse_block* __restrict__ blocks, int nblocks, const void* dense, int dt,
                                                       unsigned short* __restrict__ wfrag, unsigned short* __restrict__ kofs16,
                                                       int32_t* __restrict__ nsteps, int32_t* __restrict__ nnz, int32_t* __restrict__ blockmax,
                                                       int64_t total_slots) {
    const int64_t gs = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (gs >= total_slots) return;
    int b = 0;
    for (int i = 1; i < nblocks; ++i) if ((int64_t)blocks[i].chunk0 <= gs) b = i;       // ordered by slot0
    const ortk_sparse_block& bk = blocks[b];
    const int nkc = (bk.K + GKC - 1) / GKC;
    const int local = (int)(gs - bk.chunk0), g = local / nkc, kc = local - g * nkc;
    const int k0 = kc * GKC + 8 * lane;
    unsigned short w[16][8];
    const bool vec = (bk.ld & 7) == 0 && ((bk.src_offset + k0) & 7) == 0 && k0 + 8 <= bk.K;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int row = 16 * g + n;
#pragma unroll
        for (int i = 0; i < 8; ++i) w[n][i] = 0;
        if (row < bk.N && k0 < bk.K) {
            const int64_t at = bk.src_offset + (int64_t)row * bk.ld + k0;
            if (dt == ORTK_BF16) {
                const unsigned short* p = reinterpret_cast<const unsigned short*>(dense) + at;
                if (vec && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
                    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
                    for (int d = 0; d < 4; ++d) { w[n][2 * d] = (unsigned short)(t[d] & 0xFFFFu); w[n][2 * d + 1] = (unsigned short)(t[d] >> 16); }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) if (k0 + i < bk.K) w[n][i] = p[i];
                }
            } else {
                const float* p = reinterpret_cast<const float*>(dense) + at;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (k0 + i < bk.K) w[n][i] = (unsigned short)f32_to_bf16_bits(p[i]);
            }
        }
    }
    int pos[8], cnt[8], maxcnt = 0, mine = 0;
    bool flag[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        unsigned int any = 0;
#pragma unroll
        for (int n = 0; n < 16; ++n) { any |= (unsigned int)(w[n][i] & 0x7FFFu); mine += (w[n][i] & 0x7FFFu) != 0; }
        flag[i] = any != 0;
        const unsigned long long m = __ballot(flag[i]);
        pos[i] = __popcll(m & ((1ull << lane) - 1ull));
        cnt[i] = __popcll(m);
        maxcnt = max(maxcnt, cnt[i]);
    }
    const int S = (maxcnt + 3) >> 2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) { nsteps[gs] = S; nnz[gs] = mine; atomicMax(blockmax + b, S); }
    unsigned short* wf = wfrag + gs * (int64_t)(GSTEPS * 64 * 8);
    unsigned short* ko = kofs16 + gs * (int64_t)(GSTEPS * 64 * 2);
    // The product kernel runs every group of a block for the block-wide maximum of S (branch-free k-loop): the steps behind
    // this slot's own S must multiply by zero (and gather a valid row: row 0).
    for (int s_ = S; s_ < GSTEPS; ++s_) {
        reinterpret_cast<u32x4*>(wf)[s_ * 64 + lane] = (u32x4){0u, 0u, 0u, 0u};
        reinterpret_cast<uint32_t*>(ko)[s_ * 64 + lane] = 0u;
    }
    auto put = [&](int i, int p_, unsigned short krel, bool real) {
        const int s_ = p_ >> 2, lg = p_ & 3, j = (lg & 1) ? ((i + 4) & 7) : i;
#pragma unroll
        for (int n = 0; n < 16; ++n) wf[((s_ * 64 + lg * 16 + n) * 8) + j] = real ? w[n][i] : (unsigned short)0;
        // the 4 lanes (p = 0..3) of row q = j & 3 of lane group lg supply this entry's LDS row: lo half for j < 4, hi for j >= 4
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) ko[(s_ * 64 + lg * 16 + 4 * (j & 3) + pp) * 2 + (j >> 2)] = krel;
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (flag[i]) put(i, pos[i], (unsigned short)(8 * lane + i), true);
        for (int p_ = cnt[i] + lane; p_ < 4 * S; p_ += 64) put(i, p_, (unsigned short)i, false);      // zero-weight dummies of residue i
    }
}

struct GuP {
    const unsigned short* wfrag; const uint32_t* kofs; const int32_t* nsteps; const int32_t* smax;   // smax: this block's max steps
    int32_t slot0, N, K, G, nkc, gsplit;      // gsplit: groups per workgroup (grid = row tiles x ceil(G / gsplit))
    ortk_spmm_args a;
};

// epilogue of 4 consecutive output columns n0 .. n0+3 of row m (same order of operations as ortk_gemm.hip's epilogue)
__device__ __forceinline__ void gu_store4(const ortk_spmm_args& a, int N, int64_t m, int n0, f32x4 acc, float inv_keep) {
    if (m >= a.M || n0 >= N) return;
    const bool full = n0 + 3 < N;
    float bb[4] = {0.f, 0.f, 0.f, 0.f}, rr[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.bias) ld4f(a.bias + n0, n0, N, (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0, bb);
    if (a.resid) ld4f(a.resid + m * a.ldr + n0, n0, N, (reinterpret_cast<uintptr_t>(a.resid) & 15) == 0 && (a.ldr & 3) == 0, rr);
    if (a.gate) {
        const bool al_g = (reinterpret_cast<uintptr_t>(a.gate) & (a.gate_dtype == ORTK_BF16 ? 7 : 15)) == 0 && (a.ldg & 3) == 0;
        if (a.gate_dtype == ORTK_F32) ld4f(reinterpret_cast<const float*>(a.gate) + m * a.ldg + n0, n0, N, al_g, gg);
        else if (al_g && full) { const float4 t = ld_elem4(a.gate, m * a.ldg + n0, ORTK_BF16); gg[0] = t.x; gg[1] = t.y; gg[2] = t.z; gg[3] = t.w; }
        else for (int q = 0; q < 4; ++q) if (n0 + q < N) gg[q] = ld_elem(a.gate, m * a.ldg + n0 + q, ORTK_BF16);
    }
    const float rs = a.rowscale ? a.rowscale[m] : 1.f;
    bool kp[4] = {true, true, true, true};
    if (a.drop_p > 0.f) ortk_keep4(a.drop_seed, (uint64_t)(a.drop_rows ? (int64_t)a.drop_rows[m] : (int64_t)m) * (uint64_t)N + n0, a.drop_p, kp);
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = acc[q] + bb[q];
        if (a.relu) x = fmaxf(x, 0.f);
        x *= rs;
        if (a.drop_p > 0.f) x = kp[q] ? x * inv_keep : 0.f;
        if (a.gate) x = gg[q] > 0.f ? x * a.gate_scale : 0.f;
        v[q] = x + rr[q];
    }
    const int64_t yi = m * a.ldy + n0;
    const bool al_y = (reinterpret_cast<uintptr_t>(a.Y) & (a.y_dtype == ORTK_BF16 ? 7 : 15)) == 0 && (a.ldy & 3) == 0;
    if (al_y && full) st_elem4(a.Y, yi, a.y_dtype, make_float4(v[0], v[1], v[2], v[3]));
    else for (int q = 0; q < 4; ++q) if (n0 + q < N) st_elem(a.Y, yi + q, a.y_dtype, v[q]);
}

// lean epilogue (bias / ReLU / residual only, everything 16-byte aligned, N a multiple of 4): the generic one above costs ~60
// VALU instructions per 4 outputs — more issue time than the MFMAs that produced them
__device__ __forceinline__ void gu_store4_lean(const ortk_spmm_args& a, int N, int64_t m, int n0, f32x4 v, f32x4 bias4) {
    if (m >= a.M || n0 >= N) return;
    v += bias4;
    if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + m * a.ldr + n0);
    st_elem4(a.Y, m * a.ldy + n0, a.y_dtype, make_float4(v[0], v[1], v[2], v[3]));
}

// MT row tiles of 16 per workgroup, 8 waves.  MULTI = false: K <= 512, a wave walks its groups one after the other and
// stores each at once.  MULTI = true: K > 512 (a wave then owns at most GPW groups: N <= 512): the X tile is restaged per
// chunk and the accumulators of all the wave's groups persist across the chunks.
// The weights of a WHOLE group (up to 16 k-steps x 16 bytes per lane) are requested at once and the next group's while the
// current one is multiplied: with one or two waves per SIMD nothing else hides the L2 latency of the weight stream (the
// first version fetched one k-step ahead and ran at 150 TF/s of MFMA work).
constexpr int GPW = 4;
constexpr int GNT = 512;
struct GroupW { u32x4 w[GSTEPS]; uint32_t k[GSTEPS]; };

// Multiply the group held in `gw` over SMAX k-steps (the block-wide bound: steps past a group's own count carry zero
// weights) and, step by step, refill the registers of every consumed step with the same step of the NEXT group: a rolling
// prefetch one whole group deep in a single register buffer.  NO branch inside: a conditional load makes hipcc wait
// vmcnt(0) at every later use (measured: ~1 us per k-step, 15x the MFMA time of a group).
template <int MT, int PITCH, int SMAX>
__device__ __forceinline__ void gu_mul_group(GroupW& gw, const unsigned char* xcol, f32x4 (&cur)[MT], const u32x4* wfn, const uint32_t* kon) {
#pragma unroll
    for (int s_ = 0; s_ < SMAX; ++s_) {
        const unsigned char* ra = xcol + (gw.k[s_] & 0xFFFFu) * PITCH;
        const unsigned char* rb = xcol + (gw.k[s_] >> 16) * PITCH;
        const bf16x8 wa = __builtin_bit_cast(bf16x8, gw.w[s_]);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const bf16x4 lo = tr_read4(ra + 32 * t), hi = tr_read4(rb + 32 * t);
            const bf16x8 xb = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            cur[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, cur[t], 0, 0, 0);
        }
        gw.w[s_] = wfn[s_ * 64]; gw.k[s_] = kon[s_ * 64];
    }
}

template <int MT, bool MULTI>
__global__ __launch_bounds__(GNT) void spmm_gu_kernel(GuP p) {
    constexpr int ROWS = 16 * MT;
    constexpr int PITCH = MT == 1 ? 32 : ROWS * 2 + 32;      // bytes per k-row: == 32 (mod 256) -> the 8 k-rows of a half-wave
    constexpr int NP = ROWS / 2;                             // read (distinct k mod 8) fall in 8 disjoint 8-bank windows
    constexpr int NW = GNT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngs = (p.G + p.gsplit - 1) / p.gsplit;
    const int64_t m0 = (int64_t)(blockIdx.x / ngs) * ROWS;
    const int g_begin = (blockIdx.x % ngs) * p.gsplit, g_end = min(p.G, g_begin + p.gsplit);
    const ortk_spmm_args& a = p.a;
    const int lg = lane >> 4, pp = lane & 3;
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const bool lean = !a.gate && a.drop_p == 0.f && !a.rowscale && (p.N & 3) == 0 && (a.ldy & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(a.Y) & (a.y_dtype == ORTK_BF16 ? 7 : 15)) == 0 &&
                      (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                      (!a.resid || ((reinterpret_cast<uintptr_t>(a.resid) & 15) == 0 && (a.ldr & 3) == 0));
    auto store_group = [&](int gg, f32x4 (&v)[MT]) {
        const int n0 = 16 * gg + 4 * lg;
        if (lean) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && n0 < p.N) b4 = *reinterpret_cast<const f32x4*>(a.bias + n0);
#pragma unroll
            for (int t = 0; t < MT; ++t) gu_store4_lean(a, p.N, m0 + 16 * t + (lane & 15), n0, v[t], b4);
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t) gu_store4(a, p.N, m0 + 16 * t + (lane & 15), n0, v[t], inv_keep);
        }
    };
    f32x4 acc[MULTI ? GPW : 1][MT];
#pragma unroll
    for (int gi = 0; gi < (MULTI ? GPW : 1); ++gi)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[gi][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool xfast = a.x_dtype == ORTK_BF16 && (a.ldx & 7) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
    const unsigned char* xcol = smem + 8 * pp;        // this lane's 4 columns inside a row tile (+ 32 t)
    const int smax = *p.smax;                          // block-wide bound of the k-steps per (group, chunk)
    const int sbound = smax <= 8 ? 8 : smax <= 12 ? 12 : 16;    // the unrolled body that serves it
    for (int kc = 0; kc < p.nkc; ++kc) {
        const int kbase = kc * GKC, kw = min(GKC, p.K - kbase), kwp = (kw + 7) & ~7;
        // the first group's weights travel while the tile is staged (clamped to a valid slot when the wave has no group)
        GroupW gw;
        int g = g_begin + wave;
        auto slot_of = [&](int gg) { return (int64_t)p.slot0 + (int64_t)min(gg, p.G - 1) * p.nkc + kc; };
        auto wf_of = [&](int64_t sl) { return reinterpret_cast<const u32x4*>(p.wfrag) + sl * (GSTEPS * 64) + lane; };
        auto ko_of = [&](int64_t sl) { return p.kofs + sl * (GSTEPS * 64) + lane; };
        {
            const int64_t sl = slot_of(g);
            const u32x4* wf = wf_of(sl); const uint32_t* ko = ko_of(sl);
#pragma unroll
            for (int s_ = 0; s_ < GSTEPS; ++s_) if (s_ < sbound) { gw.w[s_] = wf[s_ * 64]; gw.k[s_] = ko[s_ * 64]; }
        }
        if (kc > 0) __syncthreads();                   // every wave is done with the previous chunk's tile
        // ---- stage X[m0 .. m0+ROWS)[kbase .. kbase+kw) transposed: row k of the tile = ROWS bf16, two X rows per dword store
        const int ntask = NP * (kwp >> 3);
        if (xfast && kw == kwp) {
            const unsigned short* X16 = reinterpret_cast<const unsigned short*>(a.X);
            for (int task0 = tid; task0 < ntask; task0 += 4 * GNT) {     // four tasks' loads in flight per thread
                u32x4 t0[4], t1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int task = task0 + u * GNT;
                    if (task < ntask) {
                        const int pi = task % NP, q = task / NP;
                        const int64_t r0 = min(m0 + 2 * pi, a.M - 1), r1 = min(m0 + 2 * pi + 1, a.M - 1);
                        t0[u] = *reinterpret_cast<const u32x4*>(X16 + r0 * a.ldx + kbase + 8 * q);
                        t1[u] = *reinterpret_cast<const u32x4*>(X16 + r1 * a.ldx + kbase + 8 * q);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int task = task0 + u * GNT;
                    if (task < ntask) {
                        const int pi = task % NP, q = task / NP;
                        unsigned char* dst = smem + (8 * q) * PITCH + pi * 4;
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            *reinterpret_cast<unsigned int*>(dst + i * PITCH) =
                                __builtin_amdgcn_perm(t1[u][i >> 1], t0[u][i >> 1], (i & 1) ? 0x07060302u : 0x05040100u);
                    }
                }
            }
        } else {
            for (int task = tid; task < ntask; task += GNT) {
                const int pi = task % NP, q = task / NP;
                const int64_t r0 = min(m0 + 2 * pi, a.M - 1), r1 = min(m0 + 2 * pi + 1, a.M - 1);
                unsigned char* dst = smem + (8 * q) * PITCH + pi * 4;
                float x0[8], x1[8];
                load8(a.X, a.x_dtype, a.ldx, r0, kbase + 8 * q, p.K, x0);
                load8(a.X, a.x_dtype, a.ldx, r1, kbase + 8 * q, p.K, x1);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<unsigned int*>(dst + i * PITCH) = f32_to_bf16_bits(x0[i]) | (f32_to_bf16_bits(x1[i]) << 16);
            }
        }
        __syncthreads();
        // ---- groups of this wave: g = g_begin + wave, + NW, ...
#pragma unroll 1
        for (int gi = 0; g < g_end; ++gi) {
            const int gn = g + NW;
            const int64_t sln = slot_of(gn);              // (past the last group: any valid slot, its weights are never used)
            const u32x4* wfn = wf_of(sln); const uint32_t* kon = ko_of(sln);
            f32x4 cur[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) cur[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (smax <= 8) gu_mul_group<MT, PITCH, 8>(gw, xcol, cur, wfn, kon);
            else if (smax <= 12) gu_mul_group<MT, PITCH, 12>(gw, xcol, cur, wfn, kon);
            else gu_mul_group<MT, PITCH, 16>(gw, xcol, cur, wfn, kon);
            if (MULTI) {
#pragma unroll
                for (int u = 0; u < GPW; ++u)
                    if (u == gi)
#pragma unroll
                        for (int t = 0; t < MT; ++t) acc[MULTI ? u : 0][t] += cur[t];
            } else {
                store_group(g, cur);
            }
            g = gn;
        }
    }
    if (MULTI) {
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            const int g = g_begin + wave + NW * gi;
            if (g < g_end) store_group(g, acc[MULTI ? gi : 0]);
        }
    }
}

template <int MT, bool MULTI>
int launch_gu(const GuP& p, hipStream_t s) {
    constexpr int ROWS = 16 * MT;
    constexpr int PITCH = MT == 1 ? 32 : ROWS * 2 + 32;
    const size_t lds = (size_t)GKC * PITCH;
    ortk::lds_attr(reinterpret_cast<const void*>(spmm_gu_kernel<MT, MULTI>), lds);
    const int64_t tiles = ortk_cdiv(p.a.M, ROWS);
    const int ngs = (int)ortk_cdiv(p.G, p.gsplit);
    hipLaunchKernelGGL((spmm_gu_kernel<MT, MULTI>), dim3((unsigned)(tiles * ngs)), dim3(GNT), lds, s, p);
    ORTK_CHECK_LAUNCH();
    return 0;
}

bool plan_ok(const ortk_sparse_plan* p) {
    if (!p || !p->blocks_host || !p->blocks_dev || p->nblocks <= 0 || !p->stream || !p->chunk_ptr || !p->chunk_len || !p->count_scratch ||
        p->total_rows <= 0) return false;
    if (p->format == ORTK_SP_GU16) return p->perm != nullptr;      // perm: per-block maximum of the step counts
    return (p->format == ORTK_SP_ELL32 || p->format == ORTK_SP_ELL16) && p->perm && p->overflow;
}

template <typename XT, int KT>
int launch_spmm(const SpmmP& p, hipStream_t s) {
    constexpr int RB = sizeof(XT) == 4 ? 8 : 16;
    const size_t tile_b = (size_t)RB * OP * sizeof(float);
    const size_t lds = (size_t)2 * KT * 16 + tile_b + MAXRPW * sizeof(int);
    const size_t lds_alias = std::max((size_t)2 * KT * 16, tile_b) + MAXRPW * sizeof(int);
    ortk::lds_attr(reinterpret_cast<const void*>(spmm_ell_kernel<XT, KT, 256, false>), lds);
    ortk::lds_attr(reinterpret_cast<const void*>(spmm_ell_kernel<XT, KT, 512, false>), lds);
    ortk::lds_attr(reinterpret_cast<const void*>(spmm_ell_kernel<XT, KT, 256, true>), lds_alias);
    ortk::lds_attr(reinterpret_cast<const void*>(spmm_ell_kernel<XT, KT, 512, true>), lds_alias);
    const int64_t tiles = ortk_cdiv(p.a.M, RB);
    // long grids: a workgroup walks up to 4 ranges (X staged once); short grids: one range per workgroup and 8 waves
    const int rpw = tiles * p.nranges > 2048 ? std::min(4, p.nranges) : 1;
    const int ngroups = (int)ortk_cdiv(p.nranges, rpw);
    const dim3 grid((unsigned)(tiles * ngroups));
    // one range per workgroup: the output tile over the planes (wide inputs: two workgroups per compute unit instead of one)
    // (wide inputs only: with 512 input columns the planes are 16 KB, three workgroups share a CU either way, and the held tile measured
    //  no faster)
    const bool alias = rpw == 1 && KT > 512 && ortk::tuning().spmm_alias != 0;
    if (tiles * ngroups <= 640) {
        if (alias) hipLaunchKernelGGL((spmm_ell_kernel<XT, KT, 512, true>), grid, dim3(512), lds_alias, s, p, rpw, ngroups);
        else hipLaunchKernelGGL((spmm_ell_kernel<XT, KT, 512, false>), grid, dim3(512), lds, s, p, rpw, ngroups);
    } else {
        if (alias) hipLaunchKernelGGL((spmm_ell_kernel<XT, KT, 256, true>), grid, dim3(256), lds_alias, s, p, rpw, ngroups);
        else hipLaunchKernelGGL((spmm_ell_kernel<XT, KT, 256, false>), grid, dim3(256), lds, s, p, rpw, ngroups);
    }
    ORTK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

static int gu_slots(const ortk_sparse_block& b) { return (int)(ortk_cdiv(b.N, 16) * ortk_cdiv(b.K, GKC)); }

extern "C" int ortk_sparse_build(const ortk_sparse_plan* plan, const void* dense, int32_t dtype, ortk_stream stream) {
    if (!plan_ok(plan) || !dense || (dtype != ORTK_F32 && dtype != ORTK_BF16)) return ORTK_EINVAL;
    hipStream_t s = ortk_s(stream);
    if (plan->format == ORTK_SP_GU16) {
        int64_t slots = 0;
        for (int i = 0; i < plan->nblocks; ++i) {
            const ortk_sparse_block& b = plan->blocks_host[i];
            if (b.N < 1 || b.K < 1 || b.ld < b.K || b.chunk0 != slots) return ORTK_EINVAL;      // packed, in table order
            if (b.K > GKC && b.N > 16 * 8 * GPW) return ORTK_EINVAL;                              // multi-chunk blocks: N <= 512
            slots += gu_slots(b);
        }
        if (slots != plan->total_rows) return ORTK_EINVAL;
        if (hipMemsetAsync(plan->perm, 0, sizeof(int32_t) * plan->nblocks, s) != hipSuccess) return ORTK_EINVAL;
        hipLaunchKernelGGL(gu_build_kernel, dim3((unsigned)ortk_cdiv(slots, 4)), dim3(256), 0, s, plan->blocks_dev, plan->nblocks, dense, dtype,
                           reinterpret_cast<unsigned short*>(plan->stream), reinterpret_cast<unsigned short*>(plan->chunk_ptr),
                           plan->chunk_len, plan->count_scratch, plan->perm, slots);
        ORTK_CHECK_LAUNCH();
        return 0;
    }
    const int eb = plan->format == ORTK_SP_ELL16 ? 4 : 8;
    int64_t rows = 0, slots = 0, ranges = 0;
    for (int i = 0; i < plan->nblocks; ++i) {
        const ortk_sparse_block& b = plan->blocks_host[i];
        if (b.N < 1 || b.N > 16384 || b.K < 1 || b.K > KMAX || b.ld < b.K || b.capacity < 0 || ((b.stream_offset | b.capacity) & 1)) return ORTK_EINVAL;
        if (b.row0 != rows || (int64_t)b.chunk0 * 64 != slots) return ORTK_EINVAL;      // packed, in table order
        rows += b.N; slots += ortk_cdiv(b.N, 64) * 64; ranges += ortk_cdiv(b.N, RANGE);
    }
    if (rows != plan->total_rows) return ORTK_EINVAL;
    hipLaunchKernelGGL(ell_count_kernel, dim3((unsigned)ortk_cdiv(rows, 4)), dim3(256), 0, s, plan->blocks_dev, plan->nblocks, dense, dtype,
                       plan->count_scratch, rows);
    hipLaunchKernelGGL(ell_rank_kernel, dim3((unsigned)ranges), dim3(256), 0, s, plan->blocks_dev, plan->nblocks, plan->count_scratch,
                       plan->chunk_len, plan->perm, eb == 4 ? 8 : 4);
    hipLaunchKernelGGL(ell_offsets_kernel, dim3((unsigned)ortk_cdiv(plan->nblocks, 64)), dim3(64), 0, s, plan->blocks_dev, plan->nblocks,
                       plan->chunk_ptr, plan->chunk_len, plan->overflow, eb == 4 ? 8 : 4);
    if (eb == 4)
        hipLaunchKernelGGL(ell_fill_kernel<4>, dim3((unsigned)ortk_cdiv(slots, 4)), dim3(256), 0, s, plan->blocks_dev, plan->nblocks, dense,
                           dtype, plan->chunk_ptr, plan->chunk_len, plan->perm, plan->stream, slots);
    else
        hipLaunchKernelGGL(ell_fill_kernel<8>, dim3((unsigned)ortk_cdiv(slots, 4)), dim3(256), 0, s, plan->blocks_dev, plan->nblocks, dense,
                           dtype, plan->chunk_ptr, plan->chunk_len, plan->perm, plan->stream, slots);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_spmm(const ortk_sparse_plan* plan, int32_t block, const ortk_spmm_args* a, ortk_stream stream) {
    if (!plan_ok(plan) || !a || block < 0 || block >= plan->nblocks || !a->X || !a->Y || a->M < 0) return ORTK_EINVAL;
    auto dt_ok = [](int d) { return d == ORTK_F32 || d == ORTK_BF16; };
    if (!dt_ok(a->x_dtype) || !dt_ok(a->y_dtype) || !dt_ok(a->gate_dtype)) return ORTK_EINVAL;
    if (a->M == 0) return 0;
    const ortk_sparse_block& b = plan->blocks_host[block];
    hipStream_t s = ortk_s(stream);
    if (plan->format == ORTK_SP_GU16) {
        GuP p;
        p.wfrag = reinterpret_cast<const unsigned short*>(plan->stream); p.kofs = reinterpret_cast<const uint32_t*>(plan->chunk_ptr);
        p.nsteps = plan->chunk_len; p.smax = plan->perm + block; p.slot0 = b.chunk0; p.N = b.N; p.K = b.K;
        p.G = (int)ortk_cdiv(b.N, 16); p.nkc = (int)ortk_cdiv(b.K, GKC);
        p.a = *a;
        const bool multi = p.nkc > 1;
        if (multi && p.G > 8 * GPW) return ORTK_EINVAL;
        // Row tile: as many rows as still give the chip enough workgroups (the compacted weights are streamed once per row
        // tile); short grids are widened by splitting the groups of a row tile over several workgroups.
        const int mt = a->M >= 8192 ? 4 : a->M >= 2048 ? 2 : 1;      // (a 128-row tile spills registers: not offered)
        const int64_t tiles = ortk_cdiv(a->M, 16 * mt);
        int split = 1;                                   // workgroups per row tile
        while (tiles * split < 384 && p.G / (split * 2) >= 8) split *= 2;
        p.gsplit = (int)ortk_cdiv(p.G, split);
        p.gsplit = (int)ortk_cdiv(p.gsplit, 8) * 8;      // whole rounds of the 8 waves
        if (multi) return mt == 1 ? launch_gu<1, true>(p, s) : mt == 2 ? launch_gu<2, true>(p, s) : launch_gu<4, true>(p, s);
        return mt == 1 ? launch_gu<1, false>(p, s) : mt == 2 ? launch_gu<2, false>(p, s) : launch_gu<4, false>(p, s);
    }
    if (b.K > KMAX) return ORTK_EINVAL;
    SpmmP p;
    p.stream = plan->stream; p.chunk_ptr = plan->chunk_ptr + b.chunk0; p.chunk_len = plan->chunk_len + b.chunk0;
    p.perm = plan->perm + (int64_t)b.chunk0 * 64;
    p.N = b.N; p.K = b.K; p.nchunks = (int)ortk_cdiv(b.N, 64); p.nranges = (int)ortk_cdiv(b.N, RANGE);
    p.a = *a;
    const bool small = b.K <= 512;
    if (plan->format == ORTK_SP_ELL16) return small ? launch_spmm<__bf16, 512>(p, s) : launch_spmm<__bf16, 2048>(p, s);
    return small ? launch_spmm<float, 512>(p, s) : launch_spmm<float, 2048>(p, s);
}

// ortk_gemm.hip — MFMA GEMM with fused epilogues for every dense projection of the ORT path.
//
// Replaces torch.nn.functional.linear + its autograd in the reference
// (models/transformer.py:238,280,324-325,412; models/relation_transformer.py:168-176,191,331-333).
//
// One kernel template covers the three operand layouts the path needs, all on row-major fp32 storage:
//   forward   Y  = X  W^T   : A (M,K)        , B = W  (N,K)  -> transA=0, transB=0
//   dgrad     dX = dY W     : A (M,K'=N_out) , B = W  stored (K',N') -> transA=0, transB=1
//   wgrad     dW = dY^T X   : A = dY stored (K'=rows, M'=N_out), B = X stored (K', N'=K_in) -> transA=1, transB=1
//
// Tiling (gfx950, wave64): 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 =
// 4x4 MFMA 16x16 tiles; K is consumed 16 (fp32) or 32 (bf16) at a time through a double-buffered LDS image
// that is ALWAYS k-major ([k][m] / [k][n]); only the global->LDS staging differs per layout, so the fragment
// reads are identical for all three.  The MFMA is issued "swapped" (B-tile as the A operand) so that each lane
// ends up with 4 consecutive n for one m: epilogue loads/stores are float4 along the contiguous C dimension.
//
// Workgroup ids are remapped so that the blocks resident on one XCD (ids b, b+8, ...) walk CONSECUTIVE tiles
// (n fastest): they share the A panel in that XCD's L2 instead of fetching it 8 times.
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>
#include "ortk_internal.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int PITCH = 128 + 16;  // floats; 144 % 32 == 16 -> the two k-rows a 32-lane half reads hit disjoint banks

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective for any nwg: XCD x (= bid % 8) gets a contiguous chunk of logical ids
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__device__ __forceinline__ float4 ld4(const float* __restrict__ base, int64_t ld, int row, int col, int nrows, int ncols,
                                      bool vec) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows && col < ncols) {
        const float* p = base + (int64_t)row * ld + col;
        if (vec && col + 3 < ncols) {
            r = *reinterpret_cast<const float4*>(p);
        } else {
            r.x = p[0];
            if (col + 1 < ncols) r.y = p[1];
            if (col + 2 < ncols) r.z = p[2];
            if (col + 3 < ncols) r.w = p[3];
        }
    }
    return r;
}

struct Epi {
    void* C; int64_t ldc; int c_dt;
    const float* bias; const float* rowscale; const float* resid; int64_t ldr;
    const void* gate; int64_t ldg; int g_dt; float gate_scale;
    int relu; float drop_p; uint32_t drop_seed; int accumulate; bool first_split;
    int M, N;
    int drop_rs, drop_r0;      // dropout draw of element (m, n): index (m * drop_rs + drop_r0) * N + n   (1, 0: the plain m * N + n)
    const int32_t* drop_rows;  // optional: row m draws as row drop_rows[m] (applied before drop_rs / drop_r0; ortk_gemm_args.drop_rows)
};

// 4 consecutive elements of a row vector / matrix row, zero beyond N
__device__ __forceinline__ float4 ldrow4(const float* __restrict__ p, int n0, int N, bool vec) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && n0 + 3 < N) return *reinterpret_cast<const float4*>(p);
    if (n0 < N) r.x = p[0];
    if (n0 + 1 < N) r.y = p[1];
    if (n0 + 2 < N) r.z = p[2];
    if (n0 + 3 < N) r.w = p[3];
    return r;
}

// Epilogue of one wave's 64x64 sub-tile: acc[i][j] holds C[m = mrow0 + 16 i][n = ncol0 + 16 j + 0..3].
// Bias is loaded once per j; residual / gate rows are fetched as float4 for all four j of a row BEFORE any of that
// row's stores (independent loads in flight together instead of 16 load->store chains per thread).
// (MS / NS: row / column distance between neighbouring accumulators: 16 / 16 for the 16x16 MFMA grid, 32 / 8 for the 32x32 one)
// PAIR: accumulator j holds the columns ncol0 + 32 (j >> 1) + 4 (j & 1) + 0..3 instead (gemm_bf16_dma256_kernel's permuted B image:
// ncol0 = 8 x lane group), so that j = 2 J, 2 J + 1 are 8 consecutive columns: one 16-byte store of a bf16 result instead of two
// 8-byte ones (the epilogue is store-ISSUE bound: scratch/micro/store_width.hip, 16 640 x 2 048 bf16 20.8 -> 12.7 us).
template <bool FAST, int MI = 4, int NJ = 4, int MS = 16, int NS = 16, bool PAIR = false>
__device__ __forceinline__ void epilogue_tile(const Epi& e, int mrow0, int ncol0, f32x4 (&acc)[MI][NJ]) {
    static_assert(!PAIR || (NJ % 2 == 0), "paired columns");
    auto colof = [&](int j) { return PAIR ? ncol0 + 32 * (j >> 1) + 4 * (j & 1) : ncol0 + NS * j; };
    const bool first = e.first_split;
    // FAST: the launcher has verified full tiles and vector alignment of every pointer -> no bounds / alignment tests
    const bool vec_b = FAST || (e.bias && ((reinterpret_cast<uintptr_t>(e.bias) & 15) == 0));
    const bool vec_r = FAST || (e.resid && ((e.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.resid) & 15) == 0));
    const bool vec_g = FAST || (e.gate && ((e.ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.gate) & 15) == 0));
    const bool vec_c = FAST || (((e.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.C) & 15) == 0));
    const int EN = FAST ? 0x7FFFFFFF : e.N, EM = FAST ? 0x7FFFFFFF : e.M;   // bounds the compiler can fold away
    const bool c16 = e.c_dt == ORTK_BF16, g16 = e.g_dt == ORTK_BF16;
    const float inv_keep = e.drop_p > 0.f ? 1.f / (1.f - e.drop_p) : 1.f;
    float4 bias4[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n0 = colof(j);
        bias4[j] = (e.bias && first && n0 < EN) ? ldrow4(e.bias + n0, n0, EN, vec_b) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mrow0 + MS * i;
        if (m >= EM) continue;
        const float rs = e.rowscale ? e.rowscale[m] : 1.f;
        const uint64_t dm = (e.drop_p > 0.f && e.drop_rows) ? (uint64_t)e.drop_rows[m] : (uint64_t)m;
        float4 res[NJ], gat[NJ];
        bf16x4 held = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n0 = colof(j);
            res[j] = (e.resid && first && n0 < EN) ? ldrow4(e.resid + (int64_t)m * e.ldr + n0, n0, EN, vec_r) : make_float4(0.f, 0.f, 0.f, 0.f);
            gat[j] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (e.gate && n0 < EN) {
                if (!g16) gat[j] = ldrow4(reinterpret_cast<const float*>(e.gate) + (int64_t)m * e.ldg + n0, n0, EN, vec_g);
                else if (vec_g && n0 + 3 < EN) gat[j] = ld_elem4(e.gate, (int64_t)m * e.ldg + n0, ORTK_BF16);
                else {
                    float* gp = &gat[j].x;
                    for (int r = 0; r < 4; ++r) if (n0 + r < EN) gp[r] = ld_elem(e.gate, (int64_t)m * e.ldg + n0 + r, ORTK_BF16);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n0 = colof(j);
            if (n0 >= EN) continue;
            const float bb[4] = {bias4[j].x, bias4[j].y, bias4[j].z, bias4[j].w};
            const float rr[4] = {res[j].x, res[j].y, res[j].z, res[j].w};
            const float gg[4] = {gat[j].x, gat[j].y, gat[j].z, gat[j].w};
            float v[4];
            bool kp[4] = {true, true, true, true};
            if (e.drop_p > 0.f) ortk_keep4(e.drop_seed, (dm * (uint64_t)e.drop_rs + (uint64_t)e.drop_r0) * (uint64_t)e.N + n0, e.drop_p, kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][j][r] + bb[r];
                if (e.relu) x = fmaxf(x, 0.f);
                x *= rs;
                if (e.drop_p > 0.f) x = kp[r] ? x * inv_keep : 0.f;
                if (e.gate) x = gg[r] > 0.f ? x * e.gate_scale : 0.f;
                v[r] = x + rr[r];
            }
            const int64_t ci = (int64_t)m * e.ldc + n0;
            if (e.accumulate) {      // always fp32 (gradient arena)
                float* c = reinterpret_cast<float*>(e.C) + ci;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < EN) atomicAdd(c + r, v[r]);
            } else if (PAIR && c16 && vec_c && ((e.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(e.C) & 15) == 0) && n0 + 7 < EN + (j & 1) * 4) {
                // both halves of the 8 columns are whole: the even accumulator's half waits in `held`, the odd one stores all 16 bytes
                const bf16x4 h4 = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                if ((j & 1) == 0) held = h4;
                else *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(e.C) + ci - 4) = (bf16x8){held[0], held[1], held[2], held[3], h4[0], h4[1], h4[2], h4[3]};
            } else if (vec_c && n0 + 3 < EN) {
                st_elem4(e.C, ci, e.c_dt, make_float4(v[0], v[1], v[2], v[3]));   // 16-B (fp32) or 8-B (bf16) store
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < EN) st_elem(e.C, ci + r, e.c_dt, v[r]);
            }
            (void)c16;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The forward-layout LDS-DMA kernels' epilogue, LEAN: what the launcher has verified is compiled out (full column tiles, vector
// alignment, no split-K accumulation, no row scale) and dropout / gate are template switches, so that a workgroup's epilogue is a
// few hundred instructions.  The general epilogue_tile above, inlined for 8 row blocks x 4 column blocks x {checked, unchecked} with
// every option a run-time test, made gemm_bf16_dma256_kernel 31 000 instructions (~190 KB of code against a 64-KB instruction
// cache): with the loads and MFMAs switched off, a 16 640 x 2 048 launch took 33 us where its stores alone take 16
// (scratch/micro/store_like_gemm.hip; profiles/r06_gemm_epilogue.txt).
// acc[i][j]: rows mrow0 + 16 i; columns ncol0 + 16 j + 0..3, or with PAIR ncol0 + 32 (j >> 1) + 4 (j & 1) + 0..3 (ncol0 = 8 x lane
// group: blocks 2 J, 2 J + 1 are 8 consecutive columns -> one 16-byte store of a bf16 result).  Same arithmetic, in the same order,
// as epilogue_tile: (acc + bias) -> relu -> dropout -> gate -> + residual.
template <int MI, bool PAIR, bool DROP, bool GATE, bool G16, bool GWIDE>
__device__ __forceinline__ void epilogue_lean_impl(const ortk_gemm_args& p, int mrow0, int ncol0, f32x4 (&acc)[MI][4]) {
    static_assert(MI % 4 == 0, "row blocks in groups of four or two");
    auto colof = [&](int j) { return PAIR ? ncol0 + 32 * (j >> 1) + 4 * (j & 1) : ncol0 + 16 * j; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + colof(j)) : zero4;
    const bool c16 = p.c_dtype == ORTK_BF16;
    const bool pair_ok = PAIR && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
    const float inv_keep = DROP ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint64_t drs = p.drop_row_stride > 0 ? (uint64_t)p.drop_row_stride : 1ull;
    // G row blocks at a time: ALL their residual / gate loads are issued first (rows past a ragged M clamped: no control flow between
    // the loads, one round trip per group instead of one per row block — the epilogue has no other workgroup to hide behind), then the
    // arithmetic and the stores, the stores alone under the row test.
    // (row blocks per group: what fits the registers beside the accumulators — a bf16 gate is kept as loaded, 2 registers per 4 columns)
    constexpr int G = (GATE && !G16) ? 2 : 4;
    const uint32_t thr = DROP ? ortk_keep_thr(p.drop_p) : 0u;
#pragma unroll
    for (int h = 0; h < MI / G; ++h) {
        // (with a gate the residual rows — never both in the executor — are fetched per row block instead: registers)
        constexpr int GR = GATE ? 1 : G;
        f32x4 res[GR][4];
        typename std::conditional<G16, bf16x4, f32x4>::type gat[G][4];
        if constexpr (!GATE) {
            if (p.resid) {
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int64_t mc = min(mrow0 + 16 * (G * h + i), p.M - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) res[i][j] = *reinterpret_cast<const f32x4*>(p.resid + mc * p.ldr + colof(j));
                }
            } else {
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) res[i][j] = zero4;
            }
        }
        if (GATE) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int64_t mc = min(mrow0 + 16 * (G * h + i), p.M - 1);
                if constexpr (G16 && PAIR && GWIDE) {          // 8 consecutive columns of a bf16 gate: one 16-byte load
                    const __bf16* gp = reinterpret_cast<const __bf16*>(p.gate) + mc * p.ldg;
#pragma unroll
                    for (int J = 0; J < 2; ++J) {
                        const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gp + colof(2 * J));
                        gat[i][2 * J] = (bf16x4){g8[0], g8[1], g8[2], g8[3]}; gat[i][2 * J + 1] = (bf16x4){g8[4], g8[5], g8[6], g8[7]};
                    }
                } else if constexpr (G16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) gat[i][j] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.gate) + mc * p.ldg + colof(j));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) gat[i][j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.gate) + mc * p.ldg + colof(j));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int m = mrow0 + 16 * (G * h + i);
            const int mc = min(m, p.M - 1);
            if constexpr (GATE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) res[0][j] = p.resid ? *reinterpret_cast<const f32x4*>(p.resid + (int64_t)mc * p.ldr + colof(j)) : zero4;
            }
            const uint64_t dbase = DROP ? ((p.drop_rows ? (uint64_t)p.drop_rows[mc] : (uint64_t)mc) * drs + (uint64_t)p.drop_row_off) * (uint64_t)p.N : 0ull;
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 x = acc[G * h + i][j] + bias4[j];
                if (p.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
                }
                if (DROP) {
                    // (ortk_keep4's aligned form: N and the column are multiples of 4 here)
                    const uint2 kb = ortk_keep_bits(p.drop_seed, (dbase + (uint64_t)colof(j)) >> 2);
                    const bool kp[4] = {(kb.x & 0xFFFFu) >= thr, (kb.x >> 16) >= thr, (kb.y & 0xFFFFu) >= thr, (kb.y >> 16) >= thr};
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = kp[r] ? x[r] * inv_keep : 0.f;
                }
                if (GATE) {
                    const float gg[4] = {(float)gat[i][j][0], (float)gat[i][j][1], (float)gat[i][j][2], (float)gat[i][j][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = gg[r] > 0.f ? x[r] * p.gate_scale : 0.f;
                }
                v[j] = x + res[GATE ? 0 : i][j];
            }
            if (m < p.M) {                              // (a partial last row tile: ragged M)
                if (c16) {
                    __bf16* c = reinterpret_cast<__bf16*>(p.C) + (int64_t)m * p.ldc;
                    if (pair_ok) {
#pragma unroll
                        for (int J = 0; J < 2; ++J)
                            *reinterpret_cast<bf16x8*>(c + colof(2 * J)) = (bf16x8){(__bf16)v[2 * J][0], (__bf16)v[2 * J][1], (__bf16)v[2 * J][2], (__bf16)v[2 * J][3],
                                                                                    (__bf16)v[2 * J + 1][0], (__bf16)v[2 * J + 1][1], (__bf16)v[2 * J + 1][2], (__bf16)v[2 * J + 1][3]};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x4*>(c + colof(j)) = (bf16x4){(__bf16)v[j][0], (__bf16)v[j][1], (__bf16)v[j][2], (__bf16)v[j][3]};
                    }
                } else {
                    float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(c + colof(j)) = v[j];
                }
            }
        }
    }
}

template <int MI, bool PAIR, bool DROP, bool GATE>
__device__ __forceinline__ void epilogue_lean(const ortk_gemm_args& p, int mrow0, int ncol0, f32x4 (&acc)[MI][4]) {
    // (the gate's element type picks the instance once: a bf16 gate — the executor's — takes 16-byte loads of 8 columns with PAIR and
    //  four row blocks per group; 16-byte alignment of its rows is the launcher's fast_nk test + ldg % 8, else the 8-byte form)
    if constexpr (GATE) {
        if (p.gate_dtype == ORTK_BF16) {
            if (PAIR && (p.ldg & 7) == 0 && (reinterpret_cast<uintptr_t>(p.gate) & 15) == 0) epilogue_lean_impl<MI, PAIR, DROP, true, true, true>(p, mrow0, ncol0, acc);
            else epilogue_lean_impl<MI, PAIR, DROP, true, true, false>(p, mrow0, ncol0, acc);
        } else epilogue_lean_impl<MI, PAIR, DROP, true, false, false>(p, mrow0, ncol0, acc);
    } else epilogue_lean_impl<MI, PAIR, DROP, false, false, false>(p, mrow0, ncol0, acc);
}

// ------------------------------------------------------------------------------------------------ fp32 MFMA
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(ortk_gemm_args p, int tilesM, int tilesN, int kchunk) {
    constexpr int BK = 16;
    __shared__ __attribute__((aligned(16))) float sA[2][BK][PITCH];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN, rest = bid / tilesN, mt = rest % tilesM, ks = rest / tilesM;
    const int mb = mt * BM, nb = nt * BN;
    const int k_begin = ks * kchunk;
    const int k_end = min(p.K, k_begin + kchunk);

    const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
    const bool vecB = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* Af = reinterpret_cast<const float*>(p.A);
    const float* Bf = reinterpret_cast<const float*>(p.B);
    const bool fullA = vecA && mb + BM <= p.M, fullB = vecB && nb + BN <= p.N;   // workgroup-uniform
    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
        const bool kfull = k0 + BK <= k_end;
        if (fullA && kfull) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = tid + 256 * u;
                ra[u] = *reinterpret_cast<const float4*>(!TA ? Af + (int64_t)(mb + (f >> 2)) * p.lda + k0 + 4 * (f & 3)
                                                             : Af + (int64_t)(k0 + (f >> 5)) * p.lda + mb + 4 * (f & 31));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = tid + 256 * u;
                if (!TA) ra[u] = ld4(Af, p.lda, mb + (f >> 2), k0 + 4 * (f & 3), p.M, k_end, vecA);
                else     ra[u] = ld4(Af, p.lda, k0 + (f >> 5), mb + 4 * (f & 31), k_end, p.M, vecA);
            }
        }
        if (fullB && kfull) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = tid + 256 * u;
                rb[u] = *reinterpret_cast<const float4*>(!TB ? Bf + (int64_t)(nb + (f >> 2)) * p.ldb + k0 + 4 * (f & 3)
                                                             : Bf + (int64_t)(k0 + (f >> 5)) * p.ldb + nb + 4 * (f & 31));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = tid + 256 * u;
                if (!TB) rb[u] = ld4(Bf, p.ldb, nb + (f >> 2), k0 + 4 * (f & 3), p.N, k_end, vecB);
                else     rb[u] = ld4(Bf, p.ldb, k0 + (f >> 5), nb + 4 * (f & 31), k_end, p.N, vecB);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = tid + 256 * u;
            if (!TA) {
                const int m = f >> 2, k = 4 * (f & 3);
                sA[buf][k + 0][m] = ra[u].x; sA[buf][k + 1][m] = ra[u].y;
                sA[buf][k + 2][m] = ra[u].z; sA[buf][k + 3][m] = ra[u].w;
            } else {
                *reinterpret_cast<float4*>(&sA[buf][f >> 5][4 * (f & 31)]) = ra[u];
            }
            if (!TB) {
                const int n = f >> 2, k = 4 * (f & 3);
                sB[buf][k + 0][n] = rb[u].x; sB[buf][k + 1][n] = rb[u].y;
                sB[buf][k + 2][n] = rb[u].z; sB[buf][k + 3][n] = rb[u].w;
            } else {
                *reinterpret_cast<float4*>(&sB[buf][f >> 5][4 * (f & 31)]) = rb[u];
            }
        }
    };

    int buf = 0;
    if (k_begin < k_end) {
        gload(k_begin);
        sstore(0);
    }
    __syncthreads();
    const int lr = lane & 15, lk = lane >> 4;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool has_next = k0 + BK < k_end;
        if (has_next) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[buf][4 * kk + lk][wm * 64 + 16 * i + lr];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[buf][4 * kk + lk][wn * 64 + 16 * j + lr];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        if (has_next) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    Epi e{p.C, p.ldc, p.c_dtype, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_dtype, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, p.accumulate, ks == 0, p.M, p.N, p.drop_row_stride > 0 ? p.drop_row_stride : 1, p.drop_row_off, p.drop_rows};
    epilogue_tile<false>(e, mb + wm * 64 + lr, nb + wn * 64 + 4 * lk, acc);
}

// ------------------------------------------------------------------------------------------------ bf16 MFMA
// Same 128x128 tiling, K consumed 64 at a time (32 MFMA 16x16x32 per wave between barriers).  Each operand is
// either fp32 or bf16 in memory (template ET); fp32 is converted while staging, bf16 is moved as 16-byte chunks.
// The LDS image of an operand follows its GLOBAL layout so that the staging writes are vector stores:
//   * k-contiguous operand (activations X, weights W as (N,K)):  image [m][k], pitch 72 bf16;  the 8-element MFMA
//     fragment (8 consecutive k of one row) is one ds_read_b128;
//   * k-major operand (dY^T / X for wgrad, W for dgrad):           image [k][m], pitch 136 bf16;  the fragment is
//     gathered by two hardware-transposing reads (ds_read_b64_tr_b16).
constexpr int BK16 = 64;
// Both images are UNPADDED and XOR-swizzled at 16-byte-chunk granularity (the padded pitches 72 / 136 of the first version
// showed SQ_LDS_BANK_CONFLICT = 33 % of the LDS cycles: ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, ...,
// not in 16 consecutive lanes):
//   [m][k] image: 128-byte rows (8 chunks), chunk' = chunk ^ ((row >> 1) & 7);
//   [k][m] image: 256-byte rows (16 chunks), chunk' = chunk ^ 2*((k & 3) | ((k >> 1) & 4)).
// (the same layouts the LDS-DMA kernels below use, where the counter reads 0)
constexpr int PITCH_MK = BK16;
constexpr int PITCH_KM = 128;
constexpr int IMG_ELEMS = 128 * BK16;                             // 16 KB per operand image
__device__ __forceinline__ int swz_mk64(int row) { return (row >> 1) & 7; }
// the [n][k] image whose fragments take rows 8 a + 4 p + b (a, b = 0..3, p fixed) instead of 16 consecutive ones
// (gemm_bf16_dma256_kernel's column permutation): bit 1 of the row and bits 3-4 give the 8 distinct slots of such a group
__device__ __forceinline__ int swz_mk64p(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }
__device__ __forceinline__ int swz_km(int k) { return 2 * ((k & 3) | ((k >> 1) & 4)); }
// element offset of (row, col) inside a swizzled image
template <bool T> __device__ __forceinline__ int img_off(int row, int col) {
    return row * (T ? PITCH_KM : PITCH_MK) + ((((col >> 3) ^ (T ? swz_km(row) : swz_mk64(row))) << 3) | (col & 7));
}
constexpr size_t BF16_LDS_BYTES_C = (size_t)128 * (128 + 4) * sizeof(float);   // the staged C tile (epilogue) is the larger user

__device__ __forceinline__ bf16x4 tr_read(const __bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(p));
}

// One 128 x 64 operand tile in flight: global -> registers (16-byte chunks) -> LDS image (bf16).
// The staged chunks live in a plain local array that is only ever indexed with unrolled constants and never has its
// address taken: a first version kept them in a struct member and read them through reinterpret_cast — hipcc then put
// the array in SCRATCH and waited for every load right after issuing it (scratch_store + vmcnt), which serialised the
// whole prefetch.
template <typename ET> struct ChunkT;
// native LLVM vector types (HIP's float4 / uint4 are structs around unions, which defeated SROA for the bf16 case)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <> struct ChunkT<float> { typedef f32x4 type; };
template <> struct ChunkT<__bf16> { typedef u32x4 type; };

template <bool T, typename ET> struct StageCfg {
    static constexpr int EPC = 16 / (int)sizeof(ET);                 // elements per chunk: 4 (fp32) | 8 (bf16)
    static constexpr int NCH = 128 * BK16 / EPC / 256;               // chunks per thread:   8        | 4
    static constexpr int CPR = (T ? 128 : BK16) / EPC;               // chunks per storage row
    static constexpr int PITCH = T ? PITCH_KM : PITCH_MK;
    typedef typename ChunkT<ET>::type chunk_t;
};

template <typename ET> __device__ __forceinline__ typename ChunkT<ET>::type guarded_chunk(const ET* base, int64_t ld, int grow, int gcol, int nrows, int ncols);
template <> __device__ __forceinline__ f32x4 guarded_chunk<float>(const float* base, int64_t ld, int grow, int gcol, int nrows, int ncols) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (grow < nrows) {
        const float* p = base + (int64_t)grow * ld + gcol;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (gcol + q < ncols) v[q] = p[q];
    }
    return v;
}
template <> __device__ __forceinline__ u32x4 guarded_chunk<__bf16>(const __bf16* base, int64_t ld, int grow, int gcol, int nrows, int ncols) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (grow < nrows) {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + (int64_t)grow * ld + gcol;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (gcol + q < ncols) v[q >> 1] |= (unsigned int)p[q] << (16 * (q & 1));
    }
    return v;
}

template <bool T, typename ET, bool FAST, typename CT, int N>
__device__ __forceinline__ void stage_load(CT (&r)[N], const ET* __restrict__ base,
                                           int64_t ld, int tile0, int k0, int nmn, int k_end, bool fast, int tid) {
    typedef StageCfg<T, ET> S;
    typedef CT chunk_t;
    static_assert(N == S::NCH, "chunk array size");
#pragma unroll
    for (int u = 0; u < S::NCH; ++u) {
        const int f = tid + 256 * u;
        const int srow = f / S::CPR, scol = (f % S::CPR) * S::EPC;   // storage row / first column of the chunk
        const int grow = (T ? k0 : tile0) + srow, gcol = (T ? tile0 : k0) + scol;
        if (FAST || fast) {
            r[u] = *reinterpret_cast<const chunk_t*>(base + (int64_t)grow * ld + gcol);
        } else {
            const int nrows = T ? k_end : nmn, ncols = T ? nmn : k_end;
            r[u] = guarded_chunk<ET>(base, ld, grow, gcol, nrows, ncols);
        }
    }
}
__device__ __forceinline__ void lds_put(__bf16* dst, f32x4 v) {
    *reinterpret_cast<bf16x4*>(dst) = (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}
__device__ __forceinline__ void lds_put(__bf16* dst, u32x4 v) { *reinterpret_cast<u32x4*>(dst) = v; }

__device__ __forceinline__ void chunk_add(float (&cs)[8], f32x4 v) { cs[0] += v[0]; cs[1] += v[1]; cs[2] += v[2]; cs[3] += v[3]; }
__device__ __forceinline__ void chunk_add(float (&cs)[8], u32x4 v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cs[2 * q] += __uint_as_float(v[q] << 16);              // low  bf16 of the dword
        cs[2 * q + 1] += __uint_as_float(v[q] & 0xFFFF0000u);   // high bf16
    }
}

// CS: also accumulate, per thread, the sums over k of the chunk's columns (bias gradient fused into wgrad: for the
// k-major A operand every chunk of a thread covers the SAME columns, tid % CPR).
template <bool T, typename ET, bool CS, typename CT, int N>
__device__ __forceinline__ void stage_store(const CT (&r)[N], __bf16* img, int tid, float (&cs)[8]) {
    typedef StageCfg<T, ET> S;
    static_assert(N == S::NCH, "chunk array size");
#pragma unroll
    for (int u = 0; u < S::NCH; ++u) {
        const int f = tid + 256 * u;
        lds_put(img + img_off<T>(f / S::CPR, (f % S::CPR) * S::EPC), r[u]);
        if (CS) chunk_add(cs, r[u]);
    }
}

// fragment of 16 rows starting at m0 for k-sub-step ks (32 wide)
template <bool T>
__device__ __forceinline__ bf16x8 frag(const __bf16* img, int m0, int ks, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    if (!T) {
        return *reinterpret_cast<const bf16x8*>(img + img_off<false>(m0 + lr, ks * 32 + 8 * lg));
    } else {
        // lane 4q+p of a 16-lane group supplies &img[k0 + q][m0 + 4p]; it receives column (lane & 15) of 4 k-rows
        const __bf16* p = img + img_off<true>(ks * 32 + 8 * lg + (lr >> 2), m0 + 4 * (lane & 3));
        const bf16x4 lo = tr_read(p), hi = tr_read(p + 4 * PITCH_KM);     // k + 4: same swizzle (bit 2 of k is not used)
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

// Epilogue through LDS (bf16-MFMA kernels): the 128x128 fp32 tile is first laid out row-major in the (now free)
// staging LDS, then written out one full 512-byte row segment per half-wave — coalesced float4 / bf16x4 stores and
// float4 residual / gate loads — or, when accumulating (wgrad split-K), as 256 contiguous bytes per atomic
// wave-instruction.  The register-layout epilogue issued atomics that touched 16 rows x 4 scattered dwords per
// instruction: the 17x-slow access shape (MI355X_MICROARCH.md, Global float atomics); a 2048x512x21760 wgrad with
// 48 K-splits took 554 us, of which ~400 us were atomics.
constexpr int CP = 128 + 4;   // fp32 row pitch of the staged C tile (528 B: 16-B aligned)
template <bool FAST>
__device__ __forceinline__ void epilogue_from_lds(const Epi& e, float* sC, int mb, int nb, int lane, int wave);
template <bool FAST>
__device__ __forceinline__ void epilogue_staged(const Epi& e, float* sC, int mb, int nb, int wm, int wn, int lane, int wave,
                                                f32x4 (&acc)[4][4]) {
    {   // registers -> LDS (lane holds 4 consecutive n of row m)
        const int mr = wm * 64 + (lane & 15), nc = wn * 64 + 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(sC + (mr + 16 * i) * CP + nc + 16 * j) = acc[i][j];
    }
    __syncthreads();
    epilogue_from_lds<FAST>(e, sC, mb, nb, lane, wave);
}
// the 128 x 128 fp32 tile staged row-major (pitch CP) in sC -> memory (4 waves)
template <bool FAST>
__device__ __forceinline__ void epilogue_from_lds(const Epi& e, float* sC, int mb, int nb, int lane, int wave) {
    const int EN = FAST ? 0x7FFFFFFF : e.N, EM = FAST ? 0x7FFFFFFF : e.M;
    const bool first = e.first_split;
    if (e.accumulate) {
        // 64 lanes = 64 consecutive columns of one row: 256 contiguous bytes per atomic instruction
        for (int it = 0; it < 64; ++it) {
            const int r = it * 2 + (wave >> 1), c = (wave & 1) * 64 + lane;
            const int m = mb + r, n = nb + c;
            if (m < EM && n < EN) atomicAdd(reinterpret_cast<float*>(e.C) + (int64_t)m * e.ldc + n, sC[r * CP + c]);
        }
        return;
    }
    const bool vec_r = FAST || (e.resid && ((e.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.resid) & 15) == 0));
    const bool vec_g = FAST || (e.gate && ((e.ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.gate) & 15) == 0));
    const bool vec_c = FAST || (((e.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.C) & 15) == 0));
    const bool vec_b = FAST || (e.bias && ((reinterpret_cast<uintptr_t>(e.bias) & 15) == 0));
    const float inv_keep = e.drop_p > 0.f ? 1.f / (1.f - e.drop_p) : 1.f;
    const int c4 = (lane & 31) * 4, n0 = nb + c4;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.bias && first && n0 < EN) b4 = ldrow4(e.bias + n0, n0, EN, vec_b);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int r = it * 8 + wave * 2 + (lane >> 5);
        const int m = mb + r;
        if (m >= EM || n0 >= EN) continue;
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(sC + r * CP + c4);
        const float rs = e.rowscale ? e.rowscale[m] : 1.f;
        float4 res = make_float4(0.f, 0.f, 0.f, 0.f), gat = make_float4(1.f, 1.f, 1.f, 1.f);
        if (e.resid && first) res = ldrow4(e.resid + (int64_t)m * e.ldr + n0, n0, EN, vec_r);
        if (e.gate) {
            if (e.g_dt == ORTK_F32) gat = ldrow4(reinterpret_cast<const float*>(e.gate) + (int64_t)m * e.ldg + n0, n0, EN, vec_g);
            else if (vec_g && n0 + 3 < EN) gat = ld_elem4(e.gate, (int64_t)m * e.ldg + n0, ORTK_BF16);
            else { float* gp = &gat.x; for (int q = 0; q < 4; ++q) if (n0 + q < EN) gp[q] = ld_elem(e.gate, (int64_t)m * e.ldg + n0 + q, ORTK_BF16); }
        }
        const float rr[4] = {res.x, res.y, res.z, res.w}, gg[4] = {gat.x, gat.y, gat.z, gat.w};
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x = a4[q] + bb[q];
            if (e.relu) x = fmaxf(x, 0.f);
            x *= rs;
            if (e.drop_p > 0.f) x = ortk_keep(e.drop_seed, (((e.drop_rows ? (uint64_t)e.drop_rows[m] : (uint64_t)m)) * (uint64_t)e.drop_rs + (uint64_t)e.drop_r0) * (uint64_t)e.N + (n0 + q), e.drop_p) ? x * inv_keep : 0.f;
            if (e.gate) x = gg[q] > 0.f ? x * e.gate_scale : 0.f;
            v[q] = x + rr[q];
        }
        const int64_t ci = (int64_t)m * e.ldc + n0;
        if (vec_c && n0 + 3 < EN) st_elem4(e.C, ci, e.c_dt, make_float4(v[0], v[1], v[2], v[3]));
        else for (int q = 0; q < 4; ++q) if (n0 + q < EN) st_elem(e.C, ci + q, e.c_dt, v[q]);
    }
}

// FAST = every tile is full (M % 128 == N % 128 == 0, K-chunks multiples of 64) and every pointer is vector-aligned:
// the launcher checks this, and the kernel then contains no bounds test at all (the guarded variant is 10x the code).
template <bool TA, bool TB, typename AT, typename BT, bool FAST>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(ortk_gemm_args p, int tilesM, int tilesN, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    auto sA = [&](int b_) { return smem16 + (size_t)b_ * 2 * IMG_ELEMS; };
    auto sB = [&](int b_) { return smem16 + (size_t)b_ * 2 * IMG_ELEMS + IMG_ELEMS; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN, rest = bid / tilesN, mt = rest % tilesM, ks_ = rest / tilesM;
    const int mb = mt * BM, nb = nt * BN;
    const int k_begin = ks_ * kchunk;
    const int k_end = min(p.K, k_begin + kchunk);
    const AT* Ap = reinterpret_cast<const AT*>(p.A);
    const BT* Bp = reinterpret_cast<const BT*>(p.B);
    const bool vecA = ((p.lda * sizeof(AT)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
    const bool vecB = ((p.ldb * sizeof(BT)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);
    const bool fullA = vecA && mb + BM <= p.M, fullB = vecB && nb + BN <= p.N;   // workgroup-uniform

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    typename StageCfg<TA, AT>::chunk_t ra[StageCfg<TA, AT>::NCH];
    typename StageCfg<TB, BT>::chunk_t rb[StageCfg<TB, BT>::NCH];
    // fused bias gradient (wgrad layout only): column sums of A, taken by the workgroups of the first N-tile
    const bool do_cs = TA && p.colsum != nullptr && nt == 0;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cs_unused[8];
    int buf = 0;
    if (k_begin < k_end) {
        const bool kfull = k_begin + BK16 <= k_end;
        stage_load<TA, AT, FAST>(ra, Ap, p.lda, mb, k_begin, p.M, k_end, fullA && kfull, tid);
        stage_load<TB, BT, FAST>(rb, Bp, p.ldb, nb, k_begin, p.N, k_end, fullB && kfull, tid);
        if (do_cs) stage_store<TA, AT, TA>(ra, sA(0), tid, cs); else stage_store<TA, AT, false>(ra, sA(0), tid, cs_unused);
        stage_store<TB, BT, false>(rb, sB(0), tid, cs_unused);
    }
    __syncthreads();
    for (int k0 = k_begin; k0 < k_end; k0 += BK16) {
        const bool has_next = k0 + BK16 < k_end;
        if (has_next) {
            const bool kfull = k0 + 2 * BK16 <= k_end;
            stage_load<TA, AT, FAST>(ra, Ap, p.lda, mb, k0 + BK16, p.M, k_end, fullA && kfull, tid);
            stage_load<TB, BT, FAST>(rb, Bp, p.ldb, nb, k0 + BK16, p.N, k_end, fullB && kfull, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag<TA>(sA(buf), wm * 64 + 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = frag<TB>(sB(buf), wn * 64 + 16 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        if (has_next) {
            if (do_cs) stage_store<TA, AT, TA>(ra, sA(buf ^ 1), tid, cs); else stage_store<TA, AT, false>(ra, sA(buf ^ 1), tid, cs_unused);
            stage_store<TB, BT, false>(rb, sB(buf ^ 1), tid, cs_unused);
        }
        __syncthreads();
        buf ^= 1;
    }
    if (TA && do_cs) {
        // combine the 256 / CPR threads that own the same columns (the LDS images are free: the K loop ended with a barrier)
        typedef StageCfg<TA, AT> S;
        float* red = reinterpret_cast<float*>(smem16);
#pragma unroll
        for (int q = 0; q < S::EPC; ++q) red[tid * S::EPC + q] = cs[q];
        __syncthreads();
        if (tid < BM && (FAST || mb + tid < p.M)) {
            const int c = tid / S::EPC, eidx = tid % S::EPC;
            float sum = 0.f;
            for (int rr = 0; rr < 256 / S::CPR; ++rr) sum += red[(rr * S::CPR + c) * S::EPC + eidx];
            atomicAdd(p.colsum + mb + tid, sum);
        }
    }
    Epi e{p.C, p.ldc, p.c_dtype, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_dtype, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, p.accumulate, ks_ == 0, p.M, p.N, p.drop_row_stride > 0 ? p.drop_row_stride : 1, p.drop_row_off, p.drop_rows};
    if (TA && do_cs) __syncthreads();     // the column-sum scratch shares the LDS with the staged C tile
    static_assert(128 * CP * sizeof(float) <= BF16_LDS_BYTES_C, "staged C tile must fit the staging LDS");
    // plain stores are faster straight from the accumulator layout (64-B segments, no LDS round trip: 491 vs 436 TF
    // on 21760x2048x512); atomics need the contiguous 256-B shape that only the staged form provides
    if (p.accumulate) epilogue_staged<FAST>(e, reinterpret_cast<float*>(smem16), mb, nb, wm, wn, lane, wave, acc);
    else epilogue_tile<FAST>(e, mb + wm * 64 + (lane & 15), nb + wn * 64 + 4 * (lane >> 4), acc);
}
constexpr size_t BF16_LDS_BYTES = (size_t)4 * IMG_ELEMS * sizeof(__bf16) > BF16_LDS_BYTES_C ? (size_t)4 * IMG_ELEMS * sizeof(__bf16) : BF16_LDS_BYTES_C;


// ------------------------------------------------------------------------------------------------
// LDS-DMA pipelined variant (both operands bf16 in memory, every tile full).
//
// The register-staged kernel above keeps ONE K-tile in flight per workgroup; with this path's K = 512 the MFMA work of
// a tile (512 cycles) is far shorter than a global load round trip, so each K-step waited ~2 000 cycles (measured:
// 1.0 us per 64-wide K-step with the workgroup alone on the chip).  Here tiles travel global -> LDS by
// `global_load_lds_dwordx4` (no VGPRs), through a ring of GNS = 4 stages of 32 k-columns: three tiles are in flight
// while the fourth is multiplied, a counted `s_waitcnt vmcnt` retires exactly the oldest tile, and ONE raw s_barrier per
// tile both publishes it and frees the stage consumed in the previous iteration.
//
// An LDS-DMA instruction writes 64 x 16 B CONTIGUOUSLY (lane order), so the images cannot be padded; they are made
// bank-conflict free by an XOR swizzle applied on the SOURCE address (which 16-byte chunk a lane fetches) and again
// when the fragments are read:
//   [m][k] image, 64-B rows (4 chunks): chunk' = chunk ^ F[(row >> 2) & 3], F = {0,2,3,1}  -> the 16 lanes of every
//          ds_read_b128 service group hit 16 distinct 16-B slots of the 256-B bank row;
//   [k][m] image, 256-B rows (16 chunks): chunk' = chunk ^ 2*((k & 3) | ((k >> 1) & 4))  -> the 8 k-rows x 2 chunks a
//          32-lane half of ds_read_b64_tr_b16 touches are 16 distinct chunks.
constexpr int GBK = 32, GNS = 4;
constexpr int G_IMG = 128 * GBK;                       // bf16 elements per operand image (8 KB)
constexpr size_t GLDS_RING_BYTES = (size_t)GNS * 2 * G_IMG * sizeof(__bf16);   // 64 KB

__device__ __forceinline__ int swz_mk(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1} packed as 0b01111000
// 64-byte rows read as 8 a + 4 p + b (a, b = 0..3: the permuted column order of the lean epilogue's 16-byte bf16 stores): b picks the
// quarter of the 256 bytes, a the chunk
__device__ __forceinline__ int swz_mkp(int row) { return (row >> 3) & 3; }
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// issue this wave's share (2 wave-instructions) of one operand tile of TM rows / columns (128: 8 instructions over
// 4 waves; 256: 16 instructions over 8 waves)
// rmax: last valid row of an [m][k] operand (a partial last row tile re-reads it; its results are dropped by the epilogue)
// (PERM: the [n][k] image of the permuted column order, see swz_mkp)
template <bool T, int TM, bool PERM = false>
__device__ __forceinline__ void glds_tile(const __bf16* __restrict__ base, int64_t ld, int tile0, int k0, __bf16* img, int wave, int lane,
                                          int rmax = 0x7FFFFFFF) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int inst = wave * 2 + u;
        const __bf16* g;
        if (!T) {
            const int r = inst * 16 + (lane >> 2), c = (lane & 3) ^ (PERM ? swz_mkp(r) : swz_mk(r));
            g = base + (int64_t)min(tile0 + r, rmax) * ld + k0 + c * 8;
        } else {
            constexpr int CPR = TM / 8;                       // 16-byte chunks per k-row
            const int f = inst * 64 + lane, kr = f / CPR, c = (f % CPR) ^ swz_km(kr);
            g = base + (int64_t)(k0 + kr) * ld + tile0 + c * 8;
        }
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(img + inst * 512), 16, 0, 0);
    }
}

// fragment of MFMA block j (0..3) of a wave's 64 permuted columns (gfrag64p's order on the 32-column image)
__device__ __forceinline__ bf16x8 gfragp(const __bf16* img, int n0, int j, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    const int row = n0 + 32 * (j >> 1) + 8 * (lr >> 2) + 4 * (j & 1) + (lr & 3);
    return *reinterpret_cast<const bf16x8*>(img + row * GBK + ((lg ^ swz_mkp(row)) << 3));
}

template <bool T, int TM>
__device__ __forceinline__ bf16x8 gfrag(const __bf16* img, int m0, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    if (!T) {
        return *reinterpret_cast<const bf16x8*>(img + (m0 + lr) * GBK + ((lg ^ swz_mk(lr)) << 3));
    } else {
        // lane 4q+p of a 16-lane group supplies &img[k = 8*lg + q][m0 + 4p]; it receives column (lane & 15) of 4 k-rows
        const int q = lr >> 2, pp = lane & 3;
        const int chunk = ((m0 >> 3) + (pp >> 1)) ^ (2 * q + 8 * (lg & 1));
        const __bf16* a = img + (8 * lg + q) * TM + (chunk << 3) + 4 * (pp & 1);
        const bf16x4 lo = tr_read(a), hi = tr_read(a + 4 * TM);      // k + 4: same swizzle (bit 2 of k is not used)
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

// BIG = false: 128 x 128 tile, 4 waves (2 x 2), 64 KB ring, two workgroups per CU.
// BIG = true : 256 x 256 tile, 8 waves (2 x 4, each 128 x 64), 128 KB ring, one workgroup per CU: half the operand
//              bytes fetched per FLOP (128 FLOP/B instead of 64) — the measured bound of the small tile is the
//              L2 -> CU fetch rate (~13 B/clk/CU sustained), not MFMA issue.
// NS = ring depth (4, or 8 for grids of at most one workgroup per CU: with 7 tiles in flight almost the whole K = 512
// panel of a decode-sized GEMM is requested up front and the K loop stops being a chain of fetch latencies).
// Soft-max partials {max, sum exp(. - max)} of a wave's (16 MI) rows x 64 columns at (m0, n0) (ortk_gemm_args.tile_stats; natural column order)
template <int MI>
__device__ __forceinline__ void stats_partials(const ortk_gemm_args& p, int m0, int n0, int lane, f32x4 (&acc)[MI][4]) {
    // soft-max partials of this wave's 64 rows x 64 columns (bias included, columns past stat_ncols left out): in-lane over
    // the lane's 16 values of a row, two shuffles over the four lane groups
    const int lr = lane & 15, lg = lane >> 4;
    const int c0 = n0 + 4 * lg;
    const int nblk = p.N >> 6, blk = n0 >> 6;
    f32x4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c0 + 16 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
    // (a block that lies below stat_ncols as a whole — every block but the vocabulary's last — takes the form without the
    //  per-element bounds tests: the epilogue is vector-ALU time the CU's other workgroup cannot use for its MFMAs)
    const bool whole = n0 + 64 <= p.stat_ncols;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + 16 * i + lr;
        float v[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] = acc[i][j][r] + bias4[j][r];
        float mx = -INFINITY, sm = 0.f;
        if (whole) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, v[j][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sm += __expf(v[j][r] - mx);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + 16 * j + r < p.stat_ncols) mx = fmaxf(mx, v[j][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + 16 * j + r < p.stat_ncols) sm += __expf(v[j][r] - mx);
        }
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        if (lg == 0 && m < p.M) {
            float* q = p.tile_stats + ((int64_t)m * nblk + blk) * 2;
            q[0] = mx; q[1] = sm;
        }
    }
}

// Gumbel-max candidates of a wave's (16 MI) rows x 64 columns at (m0, n0) (ortk_gemm_args.tile_samp), FAST: the draw function
template <bool FAST, int MI>
__device__ __forceinline__ void samp_candidates(const ortk_gemm_args& p, int m0, int n0, int lane, f32x4 (&acc)[MI][4]) {
    // Gumbel-max candidates of this wave's 64 rows x 64 columns (ortk_gemm_args.tile_samp): per row the best key over the block's
    // columns below stat_ncols other than the row's previous token, in-lane over the lane's 16 values, two shuffles over the four
    // lane groups (total order: larger key, lower column)
    const int lr = lane & 15, lg = lane >> 4;
    const int c0 = n0 + 4 * lg;
    const int nblk = p.N >> 6, blk = n0 >> 6;
    f32x4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c0 + 16 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + 16 * i + lr;
        const int mc = min(m, p.M - 1);
        const int gs = p.samp_greedy_stride;
        const int64_t grow = p.samp_row_offset + mc;
        const bool is_greedy = gs > 0 && mc % gs == 0;          // (as sample_step: by the row of this call)
        const bool samp = p.samp_sample && !is_greedy;
        const int hrow = (int)(gs > 0 ? grow - grow / gs - 1 : grow);
        const int prev = (p.samp_seq && p.samp_t > 0) ? (int)p.samp_seq[(int64_t)mc * p.samp_L + p.samp_t - 1] : -1;
        float bv = -INFINITY, bz = 0.f; int bi = 0x7FFFFFFF;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (branch-free: a column past the vocabulary or the row's previous token competes as (-inf, no column) and never wins)
                const int v = c0 + 16 * j + r;
                const bool ok = v < p.stat_ncols && v != prev;
                const float z = acc[i][j][r] + bias4[j][r];
                const float xs = z * p.samp_inv_temperature + ortk_gumbel<FAST>(p.samp_seed, p.samp_t, hrow, v);
                const float x = ok ? (samp ? xs : z) : -INFINITY;
                const int vv = ok ? v : 0x7FFFFFFF;
                if (ortk_better(x, vv, bv, bi)) { bv = x; bi = vv; bz = z; }
            }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ov = __shfl_xor(bv, o, 64), oz = __shfl_xor(bz, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ortk_better(ov, oi, bv, bi)) { bv = ov; bi = oi; bz = oz; }
        }
        if (lg == 0 && m < p.M) *reinterpret_cast<f32x4*>(p.tile_samp + ((int64_t)m * nblk + blk) * 4) = (f32x4){bv, __int_as_float(bi), bz, 0.f};
    }
}

// EPI (128 x 128 forward layout without split-K accumulation or row scale; the launcher chooses): -1 = the general epilogue; 0..3 = the
// lean one (epilogue_lean: bit 0 dropout, bit 1 gate) on the permuted column order; 4 = soft-max partials, 5 = partials + sampling
// candidates, then the lean store.
template <bool TA, bool TB, bool BIG, int NS, int EPI = -1>
__global__ __launch_bounds__(BIG ? 512 : 256, (BIG || NS > 4) ? 1 : 2) void gemm_bf16_glds_kernel(ortk_gemm_args p, int tilesM, int tilesN, int kchunk) {
    static_assert(EPI < 0 || (!TA && !TB && !BIG), "lean epilogues: the 128 x 128 forward layout");
    constexpr bool PERM = EPI >= 0 && EPI < 4;      // (the statistics' column blocks keep the natural order)
    constexpr int TM = BIG ? 256 : 128;             // tile rows = tile columns
    constexpr int MI = BIG ? 8 : 4;                 // 16-row fragments per wave
    constexpr int IMG = TM * GBK;                   // elements per operand image
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BIG ? wave >> 2 : wave >> 1, wn = BIG ? wave & 3 : wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // (tilesN >> 16: first column tile of this launch — the launcher's remainder launch behind a 256 x 256-tile launch over the
    //  columns below the last multiple of 256)
    const int nt0 = tilesN >> 16;
    tilesN &= 0xFFFF;
    const int nt = bid % tilesN + nt0, rest = bid / tilesN, mt = rest % tilesM, ks_ = rest / tilesM;
    const int mb = mt * TM, nb = nt * TM;
    const int k_begin = ks_ * kchunk;
    const int k_end = min(p.K, k_begin + kchunk);
    const __bf16* Ap = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* Bp = reinterpret_cast<const __bf16*>(p.B);
    const int T = (k_end - k_begin) / GBK;

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fused bias gradient (wgrad layout): column sums of the k-major A tile = one more output column against a vector of
    // ones, taken by the wn = 0 waves of the first N-tile's workgroups with MI extra MFMAs per tile
    const bool do_cs = TA && p.colsum != nullptr && nt == 0 && wn == 0;
    f32x4 acc_cs[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) acc_cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};

    auto issue = [&](int t) {
        __bf16* st = smem16 + (size_t)(t & (NS - 1)) * 2 * IMG;
        glds_tile<TA, TM>(Ap, p.lda, mb, k_begin + t * GBK, st, wave, lane, TA ? 0x7FFFFFFF : p.M - 1);
        glds_tile<TB, TM, PERM>(Bp, p.ldb, nb, k_begin + t * GBK, st + IMG, wave, lane);
    };
    for (int t = 0; t < NS - 1 && t < T; ++t) issue(t);
    for (int t = 0; t < T; ++t) {
        // retire tile t (4 DMA instructions per tile and wave; up to NS-2 later tiles may stay in flight), then publish it
        const int rem = min(T - 1 - t, NS - 2);
        switch (rem) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // every wave is past its reads of tile t-1: its stage takes tile t+NS-1
        if (t + NS - 1 < T) issue(t + NS - 1);
        const __bf16* sA = smem16 + (size_t)(t & (NS - 1)) * 2 * IMG;
        const __bf16* sB = sA + IMG;
        bf16x8 a[MI], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = PERM ? gfragp(sB, wn * 64, j, lane) : gfrag<TB, TM>(sB, wn * 64 + 16 * j, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = gfrag<TA, TM>(sA, wm * (16 * MI) + 16 * i, lane);
        if (TA && do_cs) {
#pragma unroll
            for (int i = 0; i < MI; ++i) acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], acc_cs[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (TA && do_cs && lane < 16) {
#pragma unroll
        for (int i = 0; i < MI; ++i) atomicAdd(p.colsum + mb + wm * (16 * MI) + 16 * i + lane, acc_cs[i][0]);
    }
    if ((EPI < 0 || EPI >= 4) && !BIG && !TA && !TB && p.tile_stats)
        stats_partials<MI>(p, mb + wm * (16 * MI), nb + wn * 64, lane, *reinterpret_cast<f32x4(*)[MI][4]>(&acc[0]));
    if ((EPI < 0 || EPI == 5) && !BIG && !TA && !TB && p.tile_samp) {
        // (one instance per draw function, chosen once: the per-element choice doubled the straight-line code of the block)
        if (p.samp_fast) samp_candidates<true, MI>(p, mb + wm * (16 * MI), nb + wn * 64, lane, *reinterpret_cast<f32x4(*)[MI][4]>(&acc[0]));
        else             samp_candidates<false, MI>(p, mb + wm * (16 * MI), nb + wn * 64, lane, *reinterpret_cast<f32x4(*)[MI][4]>(&acc[0]));
        if (p.samp_no_store) return;
    }
    if constexpr (EPI >= 0) {
        epilogue_lean<4, PERM, (EPI & 1) != 0 && EPI < 4, (EPI & 2) != 0 && EPI < 4>(p, mb + wm * 64 + (lane & 15), nb + wn * 64 + (PERM ? 8 : 4) * (lane >> 4),
                                                                                   *reinterpret_cast<f32x4(*)[4][4]>(&acc[0]));
        return;
    }
    Epi e{p.C, p.ldc, p.c_dtype, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_dtype, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, p.accumulate, ks_ == 0, p.M, p.N, p.drop_row_stride > 0 ? p.drop_row_stride : 1, p.drop_row_off, p.drop_rows};
    if (!BIG && p.accumulate) {
        __syncthreads();      // the staged C tile reuses the ring
        epilogue_staged<true>(e, reinterpret_cast<float*>(smem16), mb, nb, wm, wn, lane, wave,
                              *reinterpret_cast<f32x4(*)[4][4]>(&acc[0]));
    } else if (mb + TM <= p.M) {
#pragma unroll
        for (int hh = 0; hh < MI / 4; ++hh)
            epilogue_tile<true>(e, mb + wm * (16 * MI) + 64 * hh + (lane & 15), nb + wn * 64 + 4 * (lane >> 4),
                                *reinterpret_cast<f32x4(*)[4][4]>(&acc[4 * hh]));
    } else {              // partial last row tile (forward layout, ragged M): bounds-checked stores
#pragma unroll
        for (int hh = 0; hh < MI / 4; ++hh)
            epilogue_tile<false>(e, mb + wm * (16 * MI) + 64 * hh + (lane & 15), nb + wn * 64 + 4 * (lane >> 4),
                                 *reinterpret_cast<f32x4(*)[4][4]>(&acc[4 * hh]));
    }
}
constexpr size_t GLDS_LDS_BYTES = GLDS_RING_BYTES > 128 * CP * sizeof(float) ? GLDS_RING_BYTES : 128 * CP * sizeof(float);
constexpr size_t GLDS_LDS_BYTES_BIG = (size_t)GNS * 2 * 256 * GBK * sizeof(__bf16);   // 128 KB


// ------------------------------------------------------------------------------------------------
// 256 x 256 tile, 64-column K-steps, two LDS stages (2 x 64 KB), 8 waves (2 x 4, each 128 x 64).
// The 32-column stages above request HALF cache lines from [m][k] operands (64 bytes of each 128-byte line per
// instruction, the other half one stage later): measured, that costs about as much as the 2x fewer bytes of the big
// tile save.  Here an [m][k] row is a full 128-byte line (8 chunks), swizzled by chunk' = chunk ^ ((row >> 1) & 7):
// the 16 lanes of every ds_read_b128 service group then hit 16 distinct 16-byte slots.
constexpr int HBK = 64;

template <bool T, bool PERM = false>
__device__ __forceinline__ void glds_tile64(const __bf16* __restrict__ base, int64_t ld, int tile0, int k0, __bf16* img, int wave, int lane,
                                            int rmax = 0x7FFFFFFF) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int inst = wave * 4 + u;                         // 32 wave-instructions of 1 KB per 256 x 64 operand tile
        const __bf16* g;
        if (!T) {
            const int r = inst * 8 + (lane >> 3), c = (lane & 7) ^ (PERM ? swz_mk64p(r) : swz_mk64(r));
            g = base + (int64_t)min(tile0 + r, rmax) * ld + k0 + c * 8;
        } else {
            const int f = inst * 64 + lane, kr = f >> 5, c = (f & 31) ^ swz_km(kr);     // 32 chunks per 512-byte k-row
            g = base + (int64_t)(k0 + kr) * ld + tile0 + c * 8;
        }
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(img + inst * 512), 16, 0, 0);
    }
}

// one 1-KB piece (wave-instruction) of a 256 x 64 operand tile: piece index u = 0..3 of this wave
template <bool T, bool PERM = false>
__device__ __forceinline__ void glds_piece64(const __bf16* __restrict__ base, int64_t ld, int tile0, int k0, __bf16* img, int wave, int lane, int u,
                                             int rmax = 0x7FFFFFFF) {
    const int inst = wave * 4 + u;
    const __bf16* g;
    if (!T) {
        const int r = inst * 8 + (lane >> 3), c = (lane & 7) ^ (PERM ? swz_mk64p(r) : swz_mk64(r));
        g = base + (int64_t)min(tile0 + r, rmax) * ld + k0 + c * 8;
    } else {
        const int f = inst * 64 + lane, kr = f >> 5, c = (f & 31) ^ swz_km(kr);
        g = base + (int64_t)(k0 + kr) * ld + tile0 + c * 8;
    }
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(img + inst * 512), 16, 0, 0);
}

// the fragment of MFMA block j (0..3) of a wave's 64 permuted columns: MFMA row 4 g + r of block j is column
// 32 (j >> 1) + 8 g + 4 (j & 1) + r, so that a lane's accumulators of blocks 2 J, 2 J + 1 are 8 CONSECUTIVE columns
__device__ __forceinline__ bf16x8 gfrag64p(const __bf16* img, int n0, int j, int ks, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    const int row = n0 + 32 * (j >> 1) + 8 * (lr >> 2) + 4 * (j & 1) + (lr & 3);
    return *reinterpret_cast<const bf16x8*>(img + row * HBK + (((ks * 4 + lg) ^ swz_mk64p(row)) << 3));
}

template <bool T>
__device__ __forceinline__ bf16x8 gfrag64(const __bf16* img, int m0, int ks, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    if (!T) {
        return *reinterpret_cast<const bf16x8*>(img + (m0 + lr) * HBK + (((ks * 4 + lg) ^ swz_mk64(lr)) << 3));
    } else {
        const int q = lr >> 2, pp = lane & 3;
        const int chunk = ((m0 >> 3) + (pp >> 1)) ^ (2 * q + 8 * (lg & 1));
        const __bf16* a = img + (ks * 32 + 8 * lg + q) * 256 + (chunk << 3) + 4 * (pp & 1);
        const bf16x4 lo = tr_read(a), hi = tr_read(a + 4 * 256);
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

// EPI: -1 = the general epilogue (every option a run-time test); 0..3 = the lean one, bit 0 dropout, bit 1 gate (forward layout, no
// split-K accumulation, no row scale: the launcher chooses)
template <bool TA, bool TB, int EPI = -1>
__global__ __launch_bounds__(512, 1) void gemm_bf16_dma256_kernel(ortk_gemm_args p, int tilesM, int tilesN, int kchunk) {
    constexpr bool PERM = !TB && EPI < 4;            // (EPI 4: soft-max partials per natural 64-column block, then the lean store — measured slower
                                                     //  than the 128 x 128 kernel on the decode-time generator, not dispatched)
    constexpr int IMG = 256 * HBK;                  // elements per operand image (32 KB)
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN, rest = bid / tilesN, mt = rest % tilesM, ks_ = rest / tilesM;
    const int mb = mt * 256, nb = nt * 256;
    const int k_begin = ks_ * kchunk;
    const int k_end = min(p.K, k_begin + kchunk);
    const __bf16* Ap = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* Bp = reinterpret_cast<const __bf16*>(p.B);
    const int T = (k_end - k_begin) / HBK;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int t) {
        __bf16* st = smem16 + (size_t)(t & 1) * 2 * IMG;
        glds_tile64<TA>(Ap, p.lda, mb, k_begin + t * HBK, st, wave, lane, TA ? 0x7FFFFFFF : p.M - 1);
        glds_tile64<TB, PERM>(Bp, p.ldb, nb, k_begin + t * HBK, st + IMG, wave, lane);
    };
    if (T > 0) issue(0);
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile t landed (this wave's share)
        __builtin_amdgcn_s_barrier();                          // ... everyone's share; and tile t-1 is fully consumed
        asm volatile("" ::: "memory");
        // The 8 DMA pieces of the next tile are issued ONE per group of 8 MFMAs instead of as a burst at the top of the
        // iteration: a burst of 64 pieces per CU queues in the texture-address unit while the waves wait to issue
        // (PMC: TA busy 44 %, 128-byte requests return in ~450 cycles, yet a tile took ~7 000 cycles to land).
        const bool next = t + 1 < T;
        __bf16* nst = smem16 + (size_t)((t + 1) & 1) * 2 * IMG;
        const int nk0 = k_begin + (t + 1) * HBK;
        const __bf16* sA = smem16 + (size_t)(t & 1) * 2 * IMG;
        const __bf16* sB = sA + IMG;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[8], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = PERM ? gfrag64p(sB, wn * 64, j, ks, lane) : gfrag64<TB>(sB, wn * 64 + 16 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = gfrag64<TA>(sA, wm * 128 + 16 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (next && (i & 1) == 0) {
                    const int u = ks * 4 + (i >> 1);               // 0..7: pieces 0-3 of A, then 0-3 of B
                    if (u < 4) glds_piece64<TA>(Ap, p.lda, mb, nk0, nst, wave, lane, u, TA ? 0x7FFFFFFF : p.M - 1);
                    else       glds_piece64<TB, PERM>(Bp, p.ldb, nb, nk0, nst + IMG, wave, lane, u - 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    if constexpr (EPI >= 0) {
        if (EPI == 4 && p.tile_stats) stats_partials<8>(p, mb + wm * 128, nb + wn * 64, lane, acc);
        epilogue_lean<8, PERM, (EPI & 1) != 0 && EPI < 4, (EPI & 2) != 0 && EPI < 4>(p, mb + wm * 128 + (lane & 15), nb + wn * 64 + (PERM ? 8 : 4) * (lane >> 4), acc);
        return;
    }
    Epi e{p.C, p.ldc, p.c_dtype, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_dtype, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, p.accumulate, ks_ == 0, p.M, p.N, p.drop_row_stride > 0 ? p.drop_row_stride : 1, p.drop_row_off, p.drop_rows};
    // (forward-layout B: the permuted column order, a lane's accumulators pair up into 8 consecutive columns)
    const int ncol0 = nb + wn * 64 + (PERM ? 8 : 4) * (lane >> 4);
    if (mb + 256 <= p.M) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            epilogue_tile<true, 4, 4, 16, 16, PERM>(e, mb + wm * 128 + 64 * hh + (lane & 15), ncol0, *reinterpret_cast<f32x4(*)[4][4]>(&acc[4 * hh]));
    } else {              // partial last row tile (ragged M)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            epilogue_tile<false, 4, 4, 16, 16, PERM>(e, mb + wm * 128 + 64 * hh + (lane & 15), ncol0, *reinterpret_cast<f32x4(*)[4][4]>(&acc[4 * hh]));
    }
}
constexpr size_t DMA256_LDS_BYTES = (size_t)2 * 2 * 256 * HBK * sizeof(__bf16);   // 128 KB


// ------------------------------------------------------------------------------------------------
// 128 x 512 "row panel" tile (forward operand layout, N = 512 = d_model): the workgroup owns WHOLE rows of the output, so
// the LayerNorm that follows the projection in the residual stream (forward: x' = x + dropout(proj), y = LN(x')) or that
// the product is the output gradient of (backward: dy = dY W, dx = dLN/dx(dy) + residual gradient) runs in the epilogue,
// on the accumulators, instead of as a second kernel that reads the (rows, 512) fp32 matrix back: 30 + 32 launches of an
// XE step and the 44 + 34 MB round trip of each.  Same pipeline as the 256 x 256 kernel above: two stages of 64 columns
// (A 16 KB + B 64 KB each: the whole 160 KB of LDS), 8 waves as 2 x 4, each 64 rows x 128 columns = 4 x 8 MFMA tiles; the
// 10 DMA pieces a wave moves per stage are issued between the MFMA groups.  Row reductions: in-lane over the lane's 32
// values of a row, two shuffles over the four lane groups, one LDS exchange over the four column waves.
constexpr int RP_M = 128, RP_N = 512;
constexpr int RP_IMG_A = RP_M * HBK, RP_IMG_B = RP_N * HBK;
constexpr size_t RP_LDS_BYTES = (size_t)2 * (RP_IMG_A + RP_IMG_B) * sizeof(__bf16);   // 163 840

// sum over the 16 lanes of a DPP row (the lanes that hold the same columns of 16 different rows); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
#define ORTK_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    ORTK_DPP_ADD(0xB1);     // quad_perm [1,0,3,2]
    ORTK_DPP_ADD(0x4E);     // quad_perm [2,3,0,1]
    ORTK_DPP_ADD(0x141);    // row_half_mirror: the other quad of the half row
    ORTK_DPP_ADD(0x140);    // row_mirror: the other half row
#undef ORTK_DPP_ADD
    return v;
}

template <int MODE>     // 1: LayerNorm forward of the epilogue's result; 2: LayerNorm backward of the product
__global__ __launch_bounds__(512, 1) void gemm_bf16_row512_kernel(ortk_gemm_args p, int, int, int) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int mb = blockIdx.x * RP_M;
    const __bf16* Ap = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* Bp = reinterpret_cast<const __bf16*>(p.B);
    const int T = p.K / HBK;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one 1-KB piece = 8 rows of 128 bytes of an [m][k] image (chunk' = chunk ^ ((row >> 1) & 7))
    auto piece = [&](const __bf16* base, int64_t ld, int tile0, int k0, __bf16* img, int inst, int rmax) {
        const int r = inst * 8 + (lane >> 3), c = (lane & 7) ^ swz_mk64(r);
        const __bf16* g = base + (int64_t)min(tile0 + r, rmax) * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(img + inst * 512), 16, 0, 0);
    };
    if (T > 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) piece(Ap, p.lda, mb, 0, smem16, wave * 2 + u, p.M - 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) piece(Bp, p.ldb, 0, 0, smem16 + RP_IMG_A, wave * 8 + u, RP_N - 1);
    }
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool next = t + 1 < T;
        __bf16* nst = smem16 + (size_t)((t + 1) & 1) * (RP_IMG_A + RP_IMG_B);
        const int nk0 = (t + 1) * HBK;
        const __bf16* sA = smem16 + (size_t)(t & 1) * (RP_IMG_A + RP_IMG_B);
        const __bf16* sB = sA + RP_IMG_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = gfrag64<false>(sB, wn * 128 + 16 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = gfrag64<false>(sA, wm * 64 + 16 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (next) {
                    const int u = ks * 4 + i;                         // 0..7
                    piece(Bp, p.ldb, 0, nk0, nst + RP_IMG_A, wave * 8 + u, RP_N - 1);
                    if (u < 2) piece(Ap, p.lda, mb, nk0, nst, wave * 2 + u, p.M - 1);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // the stages become reduction scratch
    // the epilogue's lane-derived addresses come from an opaque copy of the lane id: otherwise the compiler computes all
    // ~100 of them before the K loop and spills them (131 registers)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lr = lane_e & 15, lg = lane_e >> 4;
    const int tid_e = wave * 64 + lane_e;
    float* red0 = reinterpret_cast<float*>(smem16);    // [128 rows][4 column waves]
    float* red1 = red0 + RP_M * 4;
    float* colr = red1 + RP_M * 4;                     // [2 (da, db)][2 row waves][512]
    const int row0 = mb + wm * 64 + lr;                // + 16 i
    const int col0 = wn * 128 + 4 * lg;                // + 16 j (+ r)
    // sum over the row's 512 columns of one value per (lane, i): lane groups by two shuffles, column waves through LDS
    auto row_total = [&](float (&s)[4], float* red) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] += __shfl_xor(s[i], 16, 64);
            s[i] += __shfl_xor(s[i], 32, 64);
            if (lg == 0) red[(wm * 64 + 16 * i + lr) * 4 + wn] = s[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(red + (wm * 64 + 16 * i + lr) * 4);
            s[i] = (q[0] + q[1]) + (q[2] + q[3]);
        }
    };
    if (MODE == 1) {
        const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
        const int drs = p.drop_row_stride > 0 ? p.drop_row_stride : 1;
        f32x4 bias4[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (optional operands: an unconditional load from memory that is readable in any case + a select — a uniform branch
            // around the load makes the compiler wait for every load right after issuing it)
            const f32x4 t_ = *reinterpret_cast<const f32x4*>((p.bias ? p.bias : p.ln_a) + col0 + 16 * j);
            bias4[j] = p.bias ? t_ : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        float s[4], q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 16 * i, rowc = min(row, p.M - 1);
            const uint64_t drow = (p.drop_p > 0.f && p.drop_rows) ? (uint64_t)p.drop_rows[rowc] : (uint64_t)rowc;
            f32x4 res[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 t_ = *reinterpret_cast<const f32x4*>((p.resid ? p.resid + (int64_t)rowc * p.ldr : reinterpret_cast<const float*>(p.C) + (int64_t)rowc * p.ldc) + col0 + 16 * j);
                res[j] = p.resid ? t_ : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            float si = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bool kp[4] = {true, true, true, true};
                if (p.drop_p > 0.f)
                    ortk_keep4(p.drop_seed, (drow * (uint64_t)drs + (uint64_t)p.drop_row_off) * (uint64_t)RP_N + (col0 + 16 * j), p.drop_p, kp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[i][j][r] + bias4[j][r];
                    if (p.drop_p > 0.f) x = kp[r] ? x * inv_keep : 0.f;
                    x += res[j][r];
                    acc[i][j][r] = x;
                    si += x;
                }
                if (row < p.M) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)row * p.ldc + col0 + 16 * j) = acc[i][j];
            }
            s[i] = si;
        }
        row_total(s, red0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] *= (1.f / RP_N);                      // mean
            float qi = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float t_ = acc[i][j][r] - s[i]; qi += t_ * t_; }
            q[i] = qi;
        }
        row_total(q, red1);
        float rinv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sd = sqrtf(q[i] / (float)(RP_N - 1));
            rinv[i] = 1.f / (sd + p.ln_eps);
            const int row = row0 + 16 * i;
            if (p.ln_stats && wn == 0 && lg == 0 && row < p.M) { p.ln_stats[(int64_t)row * 2] = s[i]; p.ln_stats[(int64_t)row * 2 + 1] = sd; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.ln_a + col0 + 16 * j);
            const f32x4 gb = *reinterpret_cast<const f32x4*>(p.ln_b + col0 + 16 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 16 * i;
                if (row >= p.M) continue;
                float4 o;
                o.x = ga[0] * (acc[i][j][0] - s[i]) * rinv[i] + gb[0];
                o.y = ga[1] * (acc[i][j][1] - s[i]) * rinv[i] + gb[1];
                o.z = ga[2] * (acc[i][j][2] - s[i]) * rinv[i] + gb[2];
                o.w = ga[3] * (acc[i][j][3] - s[i]) * rinv[i] + gb[3];
                st_elem4(p.ln_y, (int64_t)row * RP_N + col0 + 16 * j, p.ln_y_dtype, o);
            }
        }
    } else {
        // dy = acc.  With xc = x - mean, rr = 1 / (sd + eps), g = dy * a:  dx = rr (g - mean(g)) - rr^2 sum(g xc) xc / (511 sd) + dres;
        // da += sum_rows dy xc rr, db += sum_rows dy  (ortk_norm.hip: ln_bwd_kernel, the same formulas)
        float mean[4], sd[4], rr[4], vf[4], sg[4], sgx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rowc = min(row0 + 16 * i, p.M - 1);
            mean[i] = p.ln_stats[(int64_t)rowc * 2]; sd[i] = p.ln_stats[(int64_t)rowc * 2 + 1];
            rr[i] = 1.f / (sd[i] + p.ln_eps);
            vf[i] = row0 + 16 * i < p.M ? 1.f : 0.f;
        }
        // four partial row sums per row (one per r): the same shape as the packed column sums, so that the compiler's pairing
        // of the r's does not leave a scalar chain behind that keeps every product alive (148 spilled registers)
        f32x4 sg4[4], sgx4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { sg4[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; sgx4[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        // pass 1, two column tiles at a time: row sums of g and g xc (g replaces dy in the accumulators), column sums of the
        // parameter gradients over the lane's 4 rows, then over the 16 rows of the wave's lane row by DPP
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            f32x4 xv[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    xv[i][jj] = *reinterpret_cast<const f32x4*>(p.ln_x + (int64_t)min(row0 + 16 * i, p.M - 1) * RP_N + col0 + 16 * (2 * jp + jj));
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * jp + jj;
                const f32x4 ga = *reinterpret_cast<const f32x4*>(p.ln_a + col0 + 16 * j);
                float pa[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (rows past M only have to stay out of the column sums: vf = 0; their own results are never stored)
                        const float dyv = acc[i][j][r];
                        const float t_ = dyv * (xv[i][jj][r] - mean[i]);        // dy xc
                        pa[r] += t_ * (rr[i] * vf[i]); pb[r] += dyv * vf[i];
                        const float g = dyv * ga[r];
                        acc[i][j][r] = g; sg4[i][r] += g; sgx4[i][r] += t_ * ga[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { pa[r] = row16_sum(pa[r]); pb[r] = row16_sum(pb[r]); }
                if (lr == 0) {
                    *reinterpret_cast<f32x4*>(colr + wm * RP_N + col0 + 16 * j) = (f32x4){pa[0], pa[1], pa[2], pa[3]};
                    *reinterpret_cast<f32x4*>(colr + 2 * RP_N + wm * RP_N + col0 + 16 * j) = (f32x4){pb[0], pb[1], pb[2], pb[3]};
                }
            }
            // keep the batches apart: hoisting all 32 row loads of the pass to its top costs 131 spilled registers
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sg[i] = (sg4[i][0] + sg4[i][1]) + (sg4[i][2] + sg4[i][3]);
            sgx[i] = (sgx4[i][0] + sgx4[i][1]) + (sgx4[i][2] + sgx4[i][3]);
        }
        row_total(sg, red0);
        row_total(sgx, red1);          // (its barrier also publishes the column sums)
        {
            atomicAdd(p.ln_da + tid_e, colr[tid_e] + colr[RP_N + tid_e]);
            atomicAdd(p.ln_db + tid_e, colr[2 * RP_N + tid_e] + colr[3 * RP_N + tid_e]);
        }
        const float ik = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 16 * i, rowc = min(row, p.M - 1);
            const float mg = sg[i] * (1.f / RP_N);
            const float coef = rr[i] * rr[i] * sgx[i] / ((float)(RP_N - 1) * sd[i]);
            f32x4 xv[8], dr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xv[j] = *reinterpret_cast<const f32x4*>(p.ln_x + (int64_t)rowc * RP_N + col0 + 16 * j);
                const f32x4 t_ = *reinterpret_cast<const f32x4*>((p.ln_dres ? p.ln_dres : p.ln_x) + (int64_t)rowc * RP_N + col0 + 16 * j);
                dr[j] = p.ln_dres ? t_ : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (row < p.M)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = rr[i] * (acc[i][j][r] - mg) - coef * (xv[j][r] - mean[i]) + dr[j][r];
                const int64_t i0 = (int64_t)row * RP_N + col0 + 16 * j;
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + i0) = o;
                if (p.ln_y) {
                    bool kp[4] = {true, true, true, true};
                    if (p.drop_p > 0.f) ortk_keep4(p.drop_seed, p.drop_rows ? (uint64_t)p.drop_rows[row] * RP_N + (uint64_t)(col0 + 16 * j) : (uint64_t)i0, p.drop_p, kp);
                    st_elem4(p.ln_y, i0, p.ln_y_dtype, make_float4(kp[0] ? o[0] * ik : 0.f, kp[1] ? o[1] * ik : 0.f, kp[2] ? o[2] * ik : 0.f, kp[3] ? o[3] * ik : 0.f));
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Data-gradient product + LayerNorm backward on SHORT row panels (ln_mode = 2, round 6): (16 MT rows) x 512 columns per workgroup,
// MT = 2 .. 5 chosen so that the grid is about one workgroup per compute unit (16 640 rows -> 80-row panels, 208 workgroups;
// 9 216 -> 48-row panels, 192).  The 128-row form above owns whole rows as well, but its 130 workgroups moved the epilogue's
// 0.6-1.1 MB each through half of the chip's load/store paths and lost to the separate kernels (section 7c of DESIGN.md); here
// every unit streams its share of x / dres / dx / dz, the LayerNorm input stays in registers between the two passes (the small
// accumulator tile leaves room), and the weight panel — 512 KB, re-read by every workgroup — comes out of L2 in two 64-column stages
// (LDS-DMA; the pieces of the next stage issued between this stage's MFMA groups).
// Replaces, per LayerNorm of the backward: one data-gradient GEMM launch + one ln_bwd launch and the fp32 (rows, 512) product
// between them (34 MB written and read back at 16 640 rows).  8 waves, wave w = columns 64 w .. 64 w + 63 of all MT row tiles.
constexpr int LB_BK = 64, LB_N = 512;
constexpr int LB_IMG_B = LB_N * LB_BK;                          // bf16 elements of a weight stage (64 KB)
template <int MT> constexpr size_t lb_lds_bytes() { return (size_t)2 * (LB_IMG_B + 16 * MT * LB_BK) * sizeof(__bf16); }

template <int MT>
__global__ __launch_bounds__(512, 1) void gemm_bf16_rowln_bwd_kernel(ortk_gemm_args p) {
    constexpr int RM = 16 * MT, IMG_A = RM * LB_BK, STAGE = LB_IMG_B + IMG_A;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = blockIdx.x * RM;
    const __bf16* Ap = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* Bp = reinterpret_cast<const __bf16*>(p.B);
    const int T = p.K / LB_BK;
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Two stages of 64 columns: a [512][64] weight image + a [RM][64] operand image, FULL 128-byte rows (chunk' = chunk ^ ((row >> 1) & 7):
    // the images of gemm_bf16_dma256_kernel; 32-column stages asked for half cache lines and ran the K loop at 7 B/clk per unit).
    // Ten 1-KB pieces (8 rows each) per wave and stage: 8 of the weights, 2 of the operand rows (pieces past 2 MT repeat earlier ones —
    // the same bytes to the same place — so that every wave counts the same vmcnt), issued between the MFMA groups of the stage before.
    const int prow = lane >> 3;
    uint32_t offB[8], offA[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int r = (wave * 8 + u) * 8 + prow; offB[u] = (uint32_t)((r * (int)p.ldb + (((lane & 7) ^ swz_mk64(r)) << 3)) * 2); }
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int r = ((wave * 2 + u) % (2 * MT)) * 8 + prow; offA[u] = (uint32_t)((min(mb + r, p.M - 1) * (int)p.lda + (((lane & 7) ^ swz_mk64(r)) << 3)) * 2); }
    auto piece = [&](int t, int q) {      // piece q = 0..9 of stage t
        __bf16* st = smem16 + (size_t)(t & 1) * STAGE;
        if (q < 8) __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(Bp + t * LB_BK) + offB[q]), (lds_void*)(st + (wave * 8 + q) * 512), 16, 0, 0);
        else       __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(Ap + t * LB_BK) + offA[q - 8]),
                                                    (lds_void*)(st + LB_IMG_B + ((wave * 2 + q - 8) % (2 * MT)) * 512), 16, 0, 0);
    };
    if (T > 0) {
#pragma unroll
        for (int q = 0; q < 10; ++q) piece(0, q);
    }
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool next = t + 1 < T;
        const __bf16* sB = smem16 + (size_t)(t & 1) * STAGE;
        const __bf16* sA = sB + LB_IMG_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[MT], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = gfrag64<false>(sB, wave * 64 + 16 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = gfrag64<false>(sA, 16 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (next) {
                    constexpr int SLOTS = 2 * MT;
                    const int slot = ks * MT + i;
#pragma unroll
                    for (int q = 0; q < 10; ++q) if ((q * SLOTS) / 10 == slot) piece(t + 1, q);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // the ring becomes reduction scratch
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                   // (epilogue addresses from an opaque lane id: not hoisted above the K loop)
    const int lr = lane_e & 15, lg = lane_e >> 4;
    float* red0 = reinterpret_cast<float*>(smem16);    // [RM rows][8 column waves]
    float* red1 = red0 + RM * 8;
    const int row0 = mb + lr;                          // + 16 i
    const int col0 = wave * 64 + 4 * lg;               // + 16 j (+ r)
    // dy = acc.  With xc = x - mean, rr = 1 / (sd + eps), g = dy * a:  dx = rr (g - mean(g)) - rr^2 sum(g xc) xc / (511 sd) + dres;
    // da += sum_rows dy xc rr, db += sum_rows dy  (ortk_norm.hip: ln_bwd_kernel, the same formulas)
    float mean[MT], sd[MT], rr[MT], vf[MT], sg[MT], sgx[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int rowc = min(row0 + 16 * i, p.M - 1);
        mean[i] = p.ln_stats[(int64_t)rowc * 2]; sd[i] = p.ln_stats[(int64_t)rowc * 2 + 1];
        rr[i] = 1.f / (sd[i] + p.ln_eps);
        vf[i] = row0 + 16 * i < p.M ? 1.f : 0.f;
        sg[i] = 0.f; sgx[i] = 0.f;
    }
    // pass 1, one column tile at a time: row sums of g = dy a and of g xc, column sums of the parameter gradients (x is read again in
    // pass 2 — 16 MT more registers per lane would hold it, and spill: the second read comes out of the Infinity Cache)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 xv[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xv[i] = *reinterpret_cast<const f32x4*>(p.ln_x + (int64_t)min(row0 + 16 * i, p.M - 1) * LB_N + col0 + 16 * j);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(p.ln_a + col0 + 16 * j);
        float pa[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dyv = acc[i][j][r];
                const float t_ = dyv * (xv[i][r] - mean[i]);
                pa[r] += t_ * (rr[i] * vf[i]); pb[r] += dyv * vf[i];       // (rows past M only have to stay out of the column sums)
                const float g = dyv * ga[r];
                acc[i][j][r] = g; sg[i] += g; sgx[i] += t_ * ga[r];
            }
        }
        // the wave owns its 64 columns: sum over the 16 rows of a lane row by DPP, one atomic per column and workgroup
#pragma unroll
        for (int r = 0; r < 4; ++r) { pa[r] = row16_sum(pa[r]); pb[r] = row16_sum(pb[r]); }
        if (lr == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { atomicAdd(p.ln_da + col0 + 16 * j + r, pa[r]); atomicAdd(p.ln_db + col0 + 16 * j + r, pb[r]); }
        }
    }
    // row totals over the 512 columns: the four lane groups by two shuffles, the eight column waves through LDS
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        sg[i] += __shfl_xor(sg[i], 16, 64); sg[i] += __shfl_xor(sg[i], 32, 64);
        sgx[i] += __shfl_xor(sgx[i], 16, 64); sgx[i] += __shfl_xor(sgx[i], 32, 64);
        if (lg == 0) { red0[(16 * i + lr) * 8 + wave] = sg[i]; red1[(16 * i + lr) * 8 + wave] = sgx[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(red0 + (16 * i + lr) * 8), q1 = *reinterpret_cast<const f32x4*>(red0 + (16 * i + lr) * 8 + 4);
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(red1 + (16 * i + lr) * 8), u1 = *reinterpret_cast<const f32x4*>(red1 + (16 * i + lr) * 8 + 4);
        sg[i] = ((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q1[0] + q1[1]) + (q1[2] + q1[3]));
        sgx[i] = ((u0[0] + u0[1]) + (u0[2] + u0[3])) + ((u1[0] + u1[1]) + (u1[2] + u1[3]));
    }
    const float ik = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = row0 + 16 * i, rowc = min(row, p.M - 1);
        const float mg = sg[i] * (1.f / LB_N);
        const float coef = rr[i] * rr[i] * sgx[i] / ((float)(LB_N - 1) * sd[i]);
        f32x4 dr[4], xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xv[j] = *reinterpret_cast<const f32x4*>(p.ln_x + (int64_t)rowc * LB_N + col0 + 16 * j);
            const f32x4 t_ = *reinterpret_cast<const f32x4*>((p.ln_dres ? p.ln_dres : p.ln_x) + (int64_t)rowc * LB_N + col0 + 16 * j);
            dr[j] = p.ln_dres ? t_ : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (row < p.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = rr[i] * (acc[i][j][r] - mg) - coef * (xv[j][r] - mean[i]) + dr[j][r];
                const int64_t i0 = (int64_t)row * LB_N + col0 + 16 * j;
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + i0) = o;
                if (p.ln_y) {
                    bool kp[4] = {true, true, true, true};
                    if (p.drop_p > 0.f) ortk_keep4(p.drop_seed, p.drop_rows ? (uint64_t)p.drop_rows[row] * LB_N + (uint64_t)(col0 + 16 * j) : (uint64_t)i0, p.drop_p, kp);
                    st_elem4(p.ln_y, i0, p.ln_y_dtype, make_float4(kp[0] ? o[0] * ik : 0.f, kp[1] ? o[1] * ik : 0.f, kp[2] ? o[2] * ik : 0.f, kp[3] ? o[3] * ik : 0.f));
                }
            }
        }
    }
}
template <int MT> static int launch_rowln_bwd(const ortk_gemm_args& p, hipStream_t s) {
    if (ortk::lds_attr(reinterpret_cast<const void*>(gemm_bf16_rowln_bwd_kernel<MT>), lb_lds_bytes<MT>())) return ORTK_EINVAL;
    hipLaunchKernelGGL(gemm_bf16_rowln_bwd_kernel<MT>, dim3((unsigned)ortk_cdiv(p.M, 16 * MT)), dim3(512), lb_lds_bytes<MT>(), s, p);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 64 x 64 tile, 64-column K-steps, forward layout only, for SHORT grids: the decode-time projections (rows = images x
// beams, a few thousand at most) give the 128 x 128 kernels 48-640 workgroups, so most CUs idle or the last round is
// nearly empty, and every workgroup walks its K panel as a chain of fetch round trips.  Four times as many workgroups
// of a quarter of the size; ring of NS stages of 16 KB (A 64 x 64 + B 64 x 64 bf16):
//   NS = 8 (128 KB, one workgroup per CU): a whole K = 512 panel is requested up front — one round trip, not K/BK of them;
//   NS = 3 ( 48 KB, three per CU) for the longer grids.
// 4 waves as 2 x 2, each 32 x 32 (2 x 2 MFMA 16x16x32 per 32 columns).  Same [m][k] image and swizzle as the 256-tile
// kernel above (full 128-byte rows, chunk' = chunk ^ ((row >> 1) & 7)).
template <int NS>
__global__ __launch_bounds__(256, NS > 4 ? 1 : NS == 4 ? 2 : 3) void gemm_bf16_dma64_kernel(ortk_gemm_args p, int tilesM, int tilesN, int kchunk) {
    constexpr int IMG = 64 * HBK;                   // elements per operand image (8 KB)
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN, mt = bid / tilesN;
    const int mb = mt * 64, nb = nt * 64;
    const __bf16* Ap = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* Bp = reinterpret_cast<const __bf16*>(p.B);
    const int T = p.K / HBK;
    (void)tilesM; (void)kchunk;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one 1-KB piece = 8 rows of 128 bytes; this wave moves pieces 2*wave, 2*wave + 1 of each operand image
    auto piece = [&](const __bf16* base, int64_t ld, int tile0, int k0, __bf16* img, int inst, int rmax) {
        const int r = inst * 8 + (lane >> 3), c = (lane & 7) ^ swz_mk64(r);
        const __bf16* g = base + (int64_t)min(tile0 + r, rmax) * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(img + inst * 512), 16, 0, 0);
    };
    auto issue = [&](int t) {
        __bf16* st = smem16 + (size_t)(t % NS) * 2 * IMG;
#pragma unroll
        for (int u = 0; u < 2; ++u) piece(Ap, p.lda, mb, t * HBK, st, wave * 2 + u, p.M - 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) piece(Bp, p.ldb, nb, t * HBK, st + IMG, wave * 2 + u, 0x7FFFFFFF);
    };
    // The epilogue's operands (bias, residual rows) are requested FIRST, ahead of the operand stages, instead of after the
    // K loop: one fetch round trip less on a kernel that is a handful of round trips long (being the oldest loads they
    // never make a counted stage wait stricter).  The launcher sends only epilogues without dropout / gate / row scale here.
    const int mrow0 = mb + wm * 32 + (lane & 15), ncol0 = nb + wn * 32 + 4 * (lane >> 4);
    f32x4 bias4[2], res4[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ncol0 + 16 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
            res4[i][j] = (p.resid && mrow0 + 16 * i < p.M) ? *reinterpret_cast<const f32x4*>(p.resid + (int64_t)(mrow0 + 16 * i) * p.ldr + ncol0 + 16 * j)
                                                            : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int t = 0; t < NS - 1 && t < T; ++t) issue(t);
    for (int t = 0; t < T; ++t) {
        // retire stage t (4 DMA instructions per stage and wave; up to NS-2 later stages stay in flight), then publish it
        const int rem = min(T - 1 - t, NS - 2);
        switch (rem) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NS - 1 < T) issue(t + NS - 1);          // every wave is past its reads of stage t-1
        const __bf16* sA = smem16 + (size_t)(t % NS) * 2 * IMG;
        const __bf16* sB = sA + IMG;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = gfrag64<false>(sB, wn * 32 + 16 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = gfrag64<false>(sA, wm * 32 + 16 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][j][r] + bias4[j][r];
                if (p.relu) x = fmaxf(x, 0.f);
                v[r] = x + res4[i][j][r];
            }
            if (mrow0 + 16 * i < p.M)        // (a partial last row tile: ragged M)
                st_elem4(p.C, (int64_t)(mrow0 + 16 * i) * p.ldc + ncol0 + 16 * j, p.c_dtype, make_float4(v[0], v[1], v[2], v[3]));
        }
}
constexpr size_t DMA64_STAGE_BYTES = (size_t)2 * 64 * HBK * sizeof(__bf16);   // 16 KB

typedef void (*gemm16_fn)(ortk_gemm_args, int, int, int);
template <bool TA, bool TB, bool FAST> gemm16_fn pick16t(int adt, int bdt) {
    if (adt == ORTK_F32 && bdt == ORTK_F32) return gemm_bf16_kernel<TA, TB, float, float, FAST>;
    if (adt == ORTK_F32 && bdt == ORTK_BF16) return gemm_bf16_kernel<TA, TB, float, __bf16, FAST>;
    if (adt == ORTK_BF16 && bdt == ORTK_F32) return gemm_bf16_kernel<TA, TB, __bf16, float, FAST>;
    return gemm_bf16_kernel<TA, TB, __bf16, __bf16, FAST>;
}
template <bool TA, bool TB> gemm16_fn pick16(int adt, int bdt, bool fast) {
    return fast ? pick16t<TA, TB, true>(adt, bdt) : pick16t<TA, TB, false>(adt, bdt);
}

// ------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix cores (forward layout, fp32 operands and results: the fp32 parity mode).
//
// `v_mfma_f32_16x16x4_f32` runs at 1/16 of the bf16 MFMA rate (157 TF/s on the chip; gemm_f32_kernel reached 58).  Every fp32 value is
// instead split, while its tile is staged, into THREE bf16 values that add up to it exactly:
//     x1 = bf16(x),  x2 = bf16(x - x1),  x3 = bf16(x - x1 - x2)          (round to nearest; the differences are exact in fp32;
//     |x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|, x - x1 - x2 - x3 = 0 for every normal fp32 x whose third part is not subnormal)
// and a product a b is accumulated as the SIX partial products of at least 2^-16 relative size,
//     a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// each exact in fp32 (8 x 8 significant bits) and summed in the MFMA's fp32 accumulators.  What is dropped (a2 b3 + a3 b2 + a3 b3) is
// at most 2^-23 |a b| per term, the size of ONE fp32 rounding of the product; the accumulation error over K terms is that of any fp32
// dot product.  Six bf16 MFMAs replace eight fp32 MFMAs of the same 32-deep k-slice at 1/16 of the cycles each: 2.67 x the fp32 matrix
// peak (418 TF/s of fp32-equivalent work on the chip).
// (tests/test_gpu_ops.py: test_gemm_f32_split_vs_float64 holds it to the native fp32 kernel's error against a float64 product.)
//
// Workgroup tile (64 WM) x (64 WN), 4 waves as 2 x 2, each wave WM x WN accumulators of `v_mfma_f32_32x32x16_bf16` (32 cycles, of
// which 24 are free for the vector ALU: the split costs 5.5 vector instructions per element).  K is consumed 32 at a time: the fp32
// tile travels global -> registers -> (split) -> three [m][k] bf16 images per operand in LDS (64-byte rows, the XOR swizzle of the
// LDS-DMA kernels; conflict-free for the 32-row ds_read_b128 fragments as well), single-buffered: 24 KB (64 x 64) .. 48 KB (128 x 128)
// per workgroup, three workgroups per compute unit — one splits while another multiplies.  (gemm_f32x3p_kernel below is the
// software-pipelined form; this one serves the grids that want many mid-sized workgroups.)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ unsigned int pk_bf16(float x, float y) {
    const bf16x2 v = __builtin_convertvector((f32x2){x, y}, bf16x2);     // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned int, v);
}
// two fp32 values -> three packed bf16 pairs
__device__ __forceinline__ void split3(float x, float y, unsigned int& p1, unsigned int& p2, unsigned int& p3) {
    p1 = pk_bf16(x, y);
    x -= __uint_as_float(p1 << 16); y -= __uint_as_float(p1 & 0xFFFF0000u);
    p2 = pk_bf16(x, y);
    x -= __uint_as_float(p2 << 16); y -= __uint_as_float(p2 & 0xFFFF0000u);
    p3 = pk_bf16(x, y);
}

template <int WM, int WN, int GM, int GN, int MINB>
__global__ __launch_bounds__(64 * GM * GN, MINB) void gemm_f32x3_kernel(ortk_gemm_args p, int tilesM, int tilesN, int) {
    constexpr int RA = 32 * WM * GM, RB = 32 * WN * GN, BK = 32, NT = 64 * GM * GN;
    constexpr int CA = RA * 4 / NT, CB = RB * 4 / NT;            // 8-column chunks per thread and k-step
    static_assert(CA * NT == RA * 4 && CB * NT == RB * 4, "tile rows must divide over the threads");
    __shared__ __attribute__((aligned(16))) __bf16 sA[3][RA * BK];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3][RB * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / GN, wn = wave % GN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN, mt = bid / tilesN;
    const int mb = mt * RA, nb = nt * RB;
    const float* Af = reinterpret_cast<const float*>(p.A);
    const float* Bf = reinterpret_cast<const float*>(p.B);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: thread -> (row f >> 2, 8-column chunk f & 3) of the tile, f = tid + 256 u; rows past the matrix re-read its last row
    // (their results are dropped by the epilogue)
    const float* ga[CA]; const float* gb[CB];
    int oa[CA], ob[CB];
#pragma unroll
    for (int u = 0; u < CA; ++u) {
        const int f = tid + NT * u, r = f >> 2, c = f & 3;
        ga[u] = Af + (int64_t)min(mb + r, p.M - 1) * p.lda + 8 * c;
        oa[u] = r * BK + ((c ^ swz_mk(r)) << 3);
    }
#pragma unroll
    for (int u = 0; u < CB; ++u) {
        const int f = tid + NT * u, r = f >> 2, c = f & 3;
        gb[u] = Bf + (int64_t)min(nb + r, p.N - 1) * p.ldb + 8 * c;
        ob[u] = r * BK + ((c ^ swz_mk(r)) << 3);
    }
    f32x4 ra[CA][2], rb[CB][2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < CA; ++u) {
            ra[u][0] = *reinterpret_cast<const f32x4*>(ga[u] + k0);
            ra[u][1] = *reinterpret_cast<const f32x4*>(ga[u] + k0 + 4);
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            rb[u][0] = *reinterpret_cast<const f32x4*>(gb[u] + k0);
            rb[u][1] = *reinterpret_cast<const f32x4*>(gb[u] + k0 + 4);
        }
    };
    auto put = [&](__bf16* i0, __bf16* i1, __bf16* i2, int off, const f32x4 (&v)[2]) {
        unsigned int q[3][4];
        split3(v[0][0], v[0][1], q[0][0], q[1][0], q[2][0]);
        split3(v[0][2], v[0][3], q[0][1], q[1][1], q[2][1]);
        split3(v[1][0], v[1][1], q[0][2], q[1][2], q[2][2]);
        split3(v[1][2], v[1][3], q[0][3], q[1][3], q[2][3]);
        *reinterpret_cast<u32x4*>(i0 + off) = (u32x4){q[0][0], q[0][1], q[0][2], q[0][3]};
        *reinterpret_cast<u32x4*>(i1 + off) = (u32x4){q[1][0], q[1][1], q[1][2], q[1][3]};
        *reinterpret_cast<u32x4*>(i2 + off) = (u32x4){q[2][0], q[2][1], q[2][2], q[2][3]};
    };

    const int l32 = lane & 31, lh = lane >> 5, sw = swz_mk(l32);
    const int fa0 = (wm * 32 * WM + l32) * BK, fb0 = (wn * 32 * WN + l32) * BK;
    if (p.K > 0) gload(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
#pragma unroll
        for (int u = 0; u < CA; ++u) put(sA[0], sA[1], sA[2], oa[u], ra[u]);
#pragma unroll
        for (int u = 0; u < CB; ++u) put(sB[0], sB[1], sB[2], ob[u], rb[u]);
        __syncthreads();
        if (k0 + BK < p.K) gload(k0 + BK);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int co = ((2 * s + lh) ^ sw) << 3;
            bf16x8 a[3][WM], b[3][WN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(sA[pl] + fa0 + i * 32 * BK + co);
#pragma unroll
                for (int j = 0; j < WN; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(sB[pl] + fb0 + j * 32 * BK + co);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    // smallest partial products first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2][j], a[0][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[1][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[2][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[0][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[1][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // 32 x 32 accumulator (operands swapped: rows = n): lane holds m = lane & 31 and n = 8 g + 4 (lane >> 5) + 0..3 for g = 0..3
    f32x4 acc4[WM][4 * WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc4[i][4 * j + g] = (f32x4){acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
    Epi e{p.C, p.ldc, p.c_dtype, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_dtype, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, 0, true, p.M, p.N, p.drop_row_stride > 0 ? p.drop_row_stride : 1, p.drop_row_off, p.drop_rows};
    epilogue_tile<false, WM, 4 * WN, 32, 8>(e, mb + wm * 32 * WM + l32, nb + wn * 32 * WN + 4 * lh, acc4);
}

// The same product, software-pipelined inside the wave: the three-plane images are DOUBLE-buffered (one barrier per k-step) and the
// split of tile t + 1 (vector ALU, no memory operand until its stores) is written between the fragment reads and the MFMAs of tile t,
// so that it issues in the 24 cycles per `v_mfma_f32_32x32x16_bf16` the matrix pipe leaves free; the raw fp32 tiles are fetched TWO
// k-steps ahead into two register sets (a whole k-step of latency budget instead of one MFMA phase).  Loads past the last tile are
// clamped to it (branch-free steps); what they deliver is split into a buffer nobody reads again.
template <int WM, int WN, int GM, int GN, int MINB>
__global__ __launch_bounds__(64 * GM * GN, MINB) void gemm_f32x3p_kernel(ortk_gemm_args p, int tilesM, int tilesN, int) {
    constexpr int RA = 32 * WM * GM, RB = 32 * WN * GN, BK = 32, NT = 64 * GM * GN;
    constexpr int CA = RA * 4 / NT, CB = RB * 4 / NT, NC = CA + CB;
    static_assert(CA * NT == RA * 4 && CB * NT == RB * 4, "tile rows must divide over the threads");
    constexpr int IA = RA * BK, IB = RB * BK, BUF = 3 * (IA + IB);       // bf16 elements: one plane of A / of B, one buffer
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / GN, wn = wave % GN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN, mt = bid / tilesN;
    const int mb = mt * RA, nb = nt * RB;
    const float* Af = reinterpret_cast<const float*>(p.A);
    const float* Bf = reinterpret_cast<const float*>(p.B);
    const int T = p.K / BK;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* gc[NC];
    int oc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const bool isA = c < CA;
        const int f = tid + NT * (isA ? c : c - CA), r = f >> 2, ch = f & 3;
        gc[c] = isA ? Af + (int64_t)min(mb + r, p.M - 1) * p.lda + 8 * ch : Bf + (int64_t)min(nb + r, p.N - 1) * p.ldb + 8 * ch;
        oc[c] = (isA ? 0 : 3 * IA) + r * BK + ((ch ^ swz_mk(r)) << 3);
    }
    f32x4 r0[NC][2], r1[NC][2];
    auto gload = [&](f32x4 (&x)[NC][2], int kt) {
        const int k0 = min(kt, T - 1) * BK;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            x[c][0] = *reinterpret_cast<const f32x4*>(gc[c] + k0);
            x[c][1] = *reinterpret_cast<const f32x4*>(gc[c] + k0 + 4);
        }
    };
    const int l32 = lane & 31, lh = lane >> 5, sw = swz_mk(l32);
    const int fa0 = (wm * 32 * WM + l32) * BK, fb0 = 3 * IA + (wn * 32 * WN + l32) * BK;
    // one k-step: multiply the tile in `rbuf`; split the raw tile `x` into `wbuf` on the way
    auto step = [&](const __bf16* rbuf, __bf16* wbuf, const f32x4 (&x)[NC][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int co = ((2 * s + lh) ^ sw) << 3;
            bf16x8 a[3][WM], b[3][WN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(rbuf + pl * IA + fa0 + i * 32 * BK + co);
#pragma unroll
                for (int j = 0; j < WN; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(rbuf + pl * IB + fb0 + j * 32 * BK + co);
            }
            unsigned int q[(NC + 1) / 2][3][4];
#pragma unroll
            for (int c = s; c < NC; c += 2) {
                split3(x[c][0][0], x[c][0][1], q[c >> 1][0][0], q[c >> 1][1][0], q[c >> 1][2][0]);
                split3(x[c][0][2], x[c][0][3], q[c >> 1][0][1], q[c >> 1][1][1], q[c >> 1][2][1]);
                split3(x[c][1][0], x[c][1][1], q[c >> 1][0][2], q[c >> 1][1][2], q[c >> 1][2][2]);
                split3(x[c][1][2], x[c][1][3], q[c >> 1][0][3], q[c >> 1][1][3], q[c >> 1][2][3]);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2][j], a[0][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[1][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[2][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[0][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[1][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
                }
#pragma unroll
            for (int c = s; c < NC; c += 2) {
                const int ps = c < CA ? IA : IB;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<u32x4*>(wbuf + oc[c] + pl * ps) = (u32x4){q[c >> 1][pl][0], q[c >> 1][pl][1], q[c >> 1][pl][2], q[c >> 1][pl][3]};
            }
            {
                // pin the interleave: behind every MFMA the share of this half's split instructions that fits its shadow (the
                // compiler's own order left runs of back-to-back MFMAs next to runs of vector instructions: 2-4 % slower)
                constexpr int NMF = WM * WN * 6, NCH = (NC + 1) / 2, VPM = (NCH * 46 + NMF - 1) / NMF;
#pragma unroll
                for (int m = 0; m < NMF; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
            }
        }
    };

    __bf16* buf0 = smem16;
    __bf16* buf1 = smem16 + BUF;
    if (T > 0) {
        gload(r0, 0);
        gload(r1, 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            unsigned int q[3][4];
            split3(r0[c][0][0], r0[c][0][1], q[0][0], q[1][0], q[2][0]);
            split3(r0[c][0][2], r0[c][0][3], q[0][1], q[1][1], q[2][1]);
            split3(r0[c][1][0], r0[c][1][1], q[0][2], q[1][2], q[2][2]);
            split3(r0[c][1][2], r0[c][1][3], q[0][3], q[1][3], q[2][3]);
            const int ps = c < CA ? IA : IB;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(buf0 + oc[c] + pl * ps) = (u32x4){q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
        }
        gload(r0, 2);
        __syncthreads();
        int t = 0;
        for (; t + 1 < T; t += 2) {
            step(buf0, buf1, r1);       // tile t; tile t + 1 -> buffer 1
            gload(r1, t + 3);
            __syncthreads();
            step(buf1, buf0, r0);       // tile t + 1; tile t + 2 (or a copy of the last tile) -> buffer 0
            gload(r0, t + 4);
            __syncthreads();
        }
        if (t < T) step(buf0, buf1, r1);   // odd tile count: the last tile (what is split on the way is not read)
    }

    f32x4 acc4[WM][4 * WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc4[i][4 * j + g] = (f32x4){acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
    Epi e{p.C, p.ldc, p.c_dtype, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_dtype, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, 0, true, p.M, p.N, p.drop_row_stride > 0 ? p.drop_row_stride : 1, p.drop_row_off, p.drop_rows};
    epilogue_tile<false, WM, 4 * WN, 32, 8>(e, mb + wm * 32 * WM + l32, nb + wn * 32 * WN + 4 * lh, acc4);
}
template <int WM, int WN, int GM, int GN, int MINB>
int launch_f32x3p(const ortk_gemm_args& p, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 3 * (32 * WM * GM + 32 * WN * GN) * 32 * sizeof(__bf16);
    ortk::lds_attr(reinterpret_cast<const void*>(gemm_f32x3p_kernel<WM, WN, GM, GN, MINB>), lds);
    const int tm = (int)ortk_cdiv(p.M, 32 * WM * GM), tn = (int)ortk_cdiv(p.N, 32 * WN * GN);
    hipLaunchKernelGGL((gemm_f32x3p_kernel<WM, WN, GM, GN, MINB>), dim3((unsigned)(tm * tn)), dim3(64 * GM * GN), lds, s, p, tm, tn, 0);
    return 0;
}

// The transposed-operand layouts of the same split product (fp32 parity mode: data gradients dX = dY W with W stored (K, N), weight
// gradients dW = dY^T X with both operands stored k-major and K = the batch's rows split over workgroups that accumulate with
// atomics).  A k-major operand is staged as float4s ALONG its contiguous dimension (4 m of one k-row), split, and written as 8-byte
// pieces into three [k][m] images (256-byte rows, chunk' = chunk ^ 4 (k & 3)); the 32-row fragments come out of
// `ds_read_b64_tr_b16` pairs (conflict-free with that swizzle: the 4 k-rows x 4 chunks of a 32-lane half are 16 distinct chunks).
// 128 x 128 tile, 4 waves of 64 x 64, single-buffered images; k-rows past the end of the reduction are staged as zeros, columns past
// the matrix re-read its last four (dropped by the epilogue).
__device__ __forceinline__ bf16x8 x3_frag_km(const __bf16* img, int m0, int s, int lane) {
    const int g = lane >> 4, j = lane & 15;
    const int k = 16 * s + 8 * (g >> 1) + (j >> 2), m = m0 + 16 * (g & 1) + 4 * (j & 3);
    const __bf16* a = img + k * 128 + ((((m >> 3) ^ (4 * (k & 3)))) << 3) + (m & 4);
    const bf16x4 lo = tr_read(a), hi = tr_read(a + 4 * 128);
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <bool TA, bool ACC>
__global__ __launch_bounds__(256, ACC ? 2 : 3) void gemm_f32x3t_kernel(ortk_gemm_args p, int tilesM, int tilesN, int kchunk) {
    constexpr int BK = 32, IMG = 128 * BK;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];      // 6 images (48 KB); ACC: the staged C tile (66 KB) over them
    __bf16* sA = smem16;
    __bf16* sB = smem16 + 3 * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % tilesN, rest = bid / tilesN, mt = rest % tilesM, ks = rest / tilesM;
    const int mb = mt * 128, nb = nt * 128;
    const int k_begin = ks * kchunk, k_end = min(p.K, k_begin + kchunk);
    const float* Af = reinterpret_cast<const float*>(p.A);
    const float* Bf = reinterpret_cast<const float*>(p.B);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // k-major operand: float4 number f = tid + 256 u (u < 4) of the 32 x 128 tile: k-row f >> 5, columns 4 (f & 31) ..+3
    // row-major A (dgrad): 8-column chunk f = tid + 256 u (u < 2): row f >> 2, chunk f & 3  (as gemm_f32x3_kernel)
    const int kr = tid >> 5, c4 = (tid & 31) * 4;
    const float* gbk = Bf + (int64_t)kr * p.ldb + min(nb + c4, p.N - 4);
    const float* gak = TA ? Af + (int64_t)kr * p.lda + min(mb + c4, p.M - 4) : nullptr;
    const int okm = kr * 128 + ((((c4 >> 3) ^ (4 * (kr & 3)))) << 3) + (c4 & 4);      // (+ 8 u rows: same swizzle, 8 * 128 elements further)
    const float* gar[2]; int oar[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = tid + 256 * u, r = f >> 2, c = f & 3;
        gar[u] = TA ? nullptr : Af + (int64_t)min(mb + r, p.M - 1) * p.lda + 8 * c;
        oar[u] = r * BK + ((c ^ swz_mk(r)) << 3);
    }
    f32x4 ra[4], rb[4];
    // fused bias gradient (weight-gradient layout): column sums of the A tile over this workgroup's k-range, by the first column tile's
    // workgroups; a thread's four float4s of a k-step cover the SAME four columns (c4 ..+3) of four k-rows
    const bool do_cs = TA && p.colsum != nullptr && nt == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = k0 + kr + 8 * u < k_end;
            rb[u] = in ? *reinterpret_cast<const f32x4*>(gbk + (int64_t)(k0 + 8 * u) * p.ldb) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (TA) ra[u] = in ? *reinterpret_cast<const f32x4*>(gak + (int64_t)(k0 + 8 * u) * p.lda) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (!TA) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ra[2 * u] = *reinterpret_cast<const f32x4*>(gar[u] + k0);
                ra[2 * u + 1] = *reinterpret_cast<const f32x4*>(gar[u] + k0 + 4);
            }
        }
    };
    auto put_km = [&](__bf16* img, int off, f32x4 v) {
        unsigned int q[3][2];
        split3(v[0], v[1], q[0][0], q[1][0], q[2][0]);
        split3(v[2], v[3], q[0][1], q[1][1], q[2][1]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(img + pl * IMG + off) = make_uint2(q[pl][0], q[pl][1]);
    };
    auto put_mk = [&](__bf16* img, int off, f32x4 v0, f32x4 v1) {
        unsigned int q[3][4];
        split3(v0[0], v0[1], q[0][0], q[1][0], q[2][0]);
        split3(v0[2], v0[3], q[0][1], q[1][1], q[2][1]);
        split3(v1[0], v1[1], q[0][2], q[1][2], q[2][2]);
        split3(v1[2], v1[3], q[0][3], q[1][3], q[2][3]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(img + pl * IMG + off) = (u32x4){q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
    };

    const int l32 = lane & 31, lh = lane >> 5, sw = swz_mk(l32);
    if (k_begin < k_end) gload(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            put_km(sB, okm + 8 * u * 128, rb[u]);
            if (TA) put_km(sA, okm + 8 * u * 128, ra[u]);
            if (TA && do_cs) cs += ra[u];
        }
        if (!TA) {
#pragma unroll
            for (int u = 0; u < 2; ++u) put_mk(sA, oar[u], ra[2 * u], ra[2 * u + 1]);
        }
        __syncthreads();
        if (k0 + BK < k_end) gload(k0 + BK);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[pl][i] = TA ? x3_frag_km(sA + pl * IMG, wm * 64 + 32 * i, s, lane)
                                  : *reinterpret_cast<const bf16x8*>(sA + pl * IMG + (wm * 64 + 32 * i + l32) * BK + (((2 * s + lh) ^ sw) << 3));
#pragma unroll
                for (int j = 0; j < 2; ++j) b[pl][j] = x3_frag_km(sB + pl * IMG, wn * 64 + 32 * j, s, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2][j], a[0][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[1][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[2][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[0][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[1][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    if (TA && do_cs) {
        // the 8 threads that own the same four columns (tid & 31 equal) -> one atomic per column (the images are free: the K loop
        // ended with a barrier); columns past M were clamped copies: not added
        f32x4* red = reinterpret_cast<f32x4*>(smem16);
        red[tid] = cs;
        __syncthreads();
        if (tid < 32) {
            f32x4 t = red[tid];
#pragma unroll
            for (int r = 1; r < 8; ++r) t += red[tid + 32 * r];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (mb + c4 + q < p.M) atomicAdd(p.colsum + mb + c4 + q, t[q]);
        }
        __syncthreads();
    }
    Epi e{p.C, p.ldc, p.c_dtype, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_dtype, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, p.accumulate, ks == 0, p.M, p.N, p.drop_row_stride > 0 ? p.drop_row_stride : 1, p.drop_row_off, p.drop_rows};
    if (ACC) {
        // split-K accumulation: the tile goes through LDS so that every atomic instruction adds 256 contiguous bytes
        float* sC = reinterpret_cast<float*>(smem16);
        const int mr = wm * 64 + l32, nc = wn * 64 + 4 * lh;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(sC + (mr + 32 * i) * CP + nc + 32 * j + 8 * g) =
                        (f32x4){acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        __syncthreads();
        epilogue_from_lds<false>(e, sC, mb, nb, lane, wave);
        return;
    }
    f32x4 acc4[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc4[i][4 * j + g] = (f32x4){acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
    epilogue_tile<false, 2, 8, 32, 8>(e, mb + wm * 64 + l32, nb + wn * 64 + 4 * lh, acc4);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ profiling hook
// Opt-in, measurement only (bench.py's roofline leg): HIP events around every GEMM launch on the launch stream,
// accumulated per kernel instance (precision, transA, transB).  Disabled by default; the timed region of bench.py
// never runs with it on.  This is the only process-global state in the library.
namespace {
struct ProfRec { hipEvent_t a, b; int key; double flops, bytes; int slot; double per_count; double units; };
constexpr int PROF_SLOTS = 1 << 16;
unsigned long long* g_prof_slots = nullptr;      // device counters (ortk::prof_slot)
int g_prof_slot_next = 0;
bool g_prof_on = false;
bool g_prof_serial = false;    // level 1: the executor keeps every launch on the caller's stream (kernels timed in isolation)
std::mutex g_prof_mu;          // decode chunks may be driven by several host threads
std::vector<ProfRec>* g_prof = nullptr;
}  // namespace

namespace ortk {
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (device, function): set once per pair, from any host thread
// (a process-wide `static bool` per call site left the second device of a process without it, and raced)
int lds_attr(const void* fn, size_t bytes) {
    struct Key { int dev; const void* fn; size_t bytes; };
    static std::mutex mu;
    static std::vector<Key>* seen = new std::vector<Key>();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ORTK_EINVAL;
    std::lock_guard<std::mutex> lk(mu);
    for (const Key& k : *seen) if (k.dev == dev && k.fn == fn && k.bytes >= bytes) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return ORTK_EINVAL;
    seen->push_back(Key{dev, fn, bytes});
    return 0;
}
bool ortk_prof_active() { return g_prof_on; }
bool ortk_prof_serial() { return g_prof_serial; }
// the same hook for launches that are not ortk_gemm (key >= 16): begin records the first event, end the second
bool prof_begin(int key, double flops, double bytes, hipStream_t s, ProfMark& m) {
    m.live = false;
    if (!g_prof_on) return false;
    if (hipEventCreate(&m.a) != hipSuccess || hipEventCreate(&m.b) != hipSuccess) return false;
    m.key = key; m.flops = flops; m.bytes = bytes; m.live = true;
    (void)hipEventRecord(m.a, s);
    return true;
}
void prof_end(const ProfMark& m, hipStream_t s) {
    if (!m.live) return;
    (void)hipEventRecord(m.b, s);
    ProfRec rec{m.a, m.b, m.key, m.flops, m.bytes, m.slot, m.per_count, m.units};
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof->push_back(rec);
}
unsigned long long* prof_slot(int* index) {
    if (index) *index = -1;
    if (!g_prof_on || !g_prof_slots || !index) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_slot_next >= PROF_SLOTS) return nullptr;
    *index = g_prof_slot_next++;
    return g_prof_slots + *index;
}
}  // namespace ortk

extern "C" int ortk_prof_enable(int32_t on) {
    if (!g_prof) g_prof = new std::vector<ProfRec>();
    for (auto& r : *g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof->clear();
    if (on && !g_prof_slots && hipMalloc(reinterpret_cast<void**>(&g_prof_slots), PROF_SLOTS * sizeof(unsigned long long)) != hipSuccess) g_prof_slots = nullptr;
    if (on && g_prof_slots && hipMemset(g_prof_slots, 0, PROF_SLOTS * sizeof(unsigned long long)) != hipSuccess) return ORTK_EINVAL;
    g_prof_slot_next = 0;
    g_prof_on = on != 0;
    g_prof_serial = on == 1;
    return 0;
}
// key = precision*4 + transA*2 + transB.  Waits for the recorded events (host sync: measurement only).
extern "C" int ortk_prof_collect(int32_t key, int64_t* launches, double* total_ms, double* total_flops) {
    if (!g_prof || !launches || !total_ms || !total_flops) return ORTK_EINVAL;
    *launches = 0; *total_ms = 0; *total_flops = 0;
    for (auto& r : *g_prof) {
        if (r.key != key) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return ORTK_EINVAL;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return ORTK_EINVAL;
        *launches += 1; *total_ms += ms; *total_flops += r.flops;
    }
    return 0;
}

// sum over the launches of `key` of the workgroups each one started (recorded by the launchers that size their grids themselves:
// the grouped weight gradients) — launches / this = the average share of the chip such a launch holds
extern "C" int ortk_prof_collect_units(int32_t key, double* total_workgroups) {
    if (!g_prof || !total_workgroups) return ORTK_EINVAL;
    *total_workgroups = 0;
    for (auto& r : *g_prof) if (r.key == key) *total_workgroups += r.units;
    return 0;
}

extern "C" int ortk_prof_collect_bytes(int32_t key, double* total_bytes) {
    if (!g_prof || !total_bytes) return ORTK_EINVAL;
    *total_bytes = 0;
    // (the counters of the slots: the caller has synchronised — ortk_prof_collect waits for every event — and this copy waits as well)
    std::vector<unsigned long long> slots;
    if (g_prof_slots && g_prof_slot_next > 0) {
        slots.resize((size_t)g_prof_slot_next);
        if (hipMemcpy(slots.data(), g_prof_slots, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return ORTK_EINVAL;
    }
    for (auto& r : *g_prof) {
        if (r.key != key) continue;
        *total_bytes += r.bytes;
        if (r.slot >= 0 && (size_t)r.slot < slots.size()) *total_bytes += r.per_count * (double)slots[(size_t)r.slot];
    }
    return 0;
}

extern "C" int ortk_gemm(const ortk_gemm_args* a, ortk_stream stream) {
    if (!a || !a->A || !a->B || !a->C || a->M < 0 || a->N < 0 || a->K < 0) return ORTK_EINVAL;
    if (a->M == 0 || a->N == 0) return 0;
    ortk_gemm_args p = *a;
    auto dt_ok = [](int d) { return d == ORTK_F32 || d == ORTK_BF16; };
    if (!dt_ok(p.a_dtype) || !dt_ok(p.b_dtype) || !dt_ok(p.c_dtype) || !dt_ok(p.gate_dtype)) return ORTK_EINVAL;
    if (!p.precision && (p.a_dtype || p.b_dtype)) return ORTK_EINVAL;   // fp32 MFMA path takes fp32 operands only
    if (p.accumulate && p.c_dtype) return ORTK_EINVAL;                   // accumulation targets the fp32 gradient arena
    if (p.transA && !p.transB) return ORTK_EINVAL;                       // layout not needed by the path
    if (p.colsum && !p.transA) return ORTK_EINVAL;                       // fused column sums: the weight-gradient layout only
    if (p.ln_mode) {
        // LayerNorm fused into the epilogue (forward: of the result; backward: the product is the LayerNorm's output gradient)
        if (p.ln_mode < 1 || p.ln_mode > 2 || p.c_dtype != ORTK_F32 || p.ldc != p.N || p.accumulate || p.transA || p.gate || p.rowscale ||
            p.N < 2 || p.N > 2048 || !p.ln_a || !p.ln_stats) return ORTK_EINVAL;
        if (p.ln_mode == 1 && (!p.ln_b || !p.ln_y)) return ORTK_EINVAL;
        if (p.ln_mode == 2 && (!p.ln_x || !p.ln_da || !p.ln_db || p.bias || p.resid || p.relu)) return ORTK_EINVAL;
        auto al = [](const void* q, size_t a_) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) % a_) == 0; };
        const bool fused = p.precision && !p.transB && p.a_dtype == ORTK_BF16 && p.b_dtype == ORTK_BF16 && p.N == RP_N && p.K > 0 &&
                           p.K % HBK == 0 && !p.relu && al(p.A, 16) && al(p.B, 16) && (p.lda % 8) == 0 && (p.ldb % 8) == 0 && al(p.C, 16) &&
                           al(p.bias, 16) && al(p.resid, 16) && (p.ldr % 4) == 0 && al(p.ln_a, 16) && al(p.ln_b, 16) && al(p.ln_y, 16) &&
                           al(p.ln_x, 16) && al(p.ln_dres, 16);
        if (!fused) {
            // any other shape / precision: the same result from the separate kernels
            ortk_gemm_args q = p; q.ln_mode = 0;
            if (p.ln_mode == 2) q.drop_p = 0.f;
            if (int e = ortk_gemm(&q, stream)) return e;
            float* Cf = reinterpret_cast<float*>(p.C);
            if (p.ln_mode == 1) return ortk_layernorm_fwd(Cf, p.ln_a, p.ln_b, p.ln_y, p.ln_y_dtype, p.ln_stats, p.M, p.N, p.ln_eps, stream);
            return ortk_layernorm_bwd_drop_rows(Cf, p.ln_x, p.ln_a, p.ln_stats, p.ln_dres, Cf, p.ln_da, p.ln_db, p.M, p.N, p.ln_eps,
                                                p.ln_y, p.ln_y_dtype, p.drop_p, p.drop_seed, p.drop_rows, stream);
        }
        if (p.ln_mode == 2 && p.K % LB_BK == 0 && p.lda < (1 << 30) / 2 && p.ldb < (1 << 30) / 2 && (int64_t)p.M * p.lda < (1ll << 30) && !(ortk::tuning().ln_fuse & 2)) {
            // short row panels, about one workgroup per compute unit
            const int mt = (int)std::min<int64_t>(5, std::max<int64_t>(2, ortk_cdiv(p.M, 16 * 256)));      // (96-row panels spill)
            hipStream_t s = ortk_s(stream);
            ProfRec rec{};
            if (g_prof_on) {
                if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) return ORTK_EINVAL;
                rec.key = 4; rec.flops = 2.0 * p.M * p.N * p.K;
                rec.bytes = (double)p.M * p.K * 2 + (double)p.N * p.K * 2 + (double)p.M * p.N * (4 + 4 + 4 + 2);
                (void)hipEventRecord(rec.a, s);
            }
            int e = 0;
            switch (mt) {
                case 2: e = launch_rowln_bwd<2>(p, s); break;
                case 3: e = launch_rowln_bwd<3>(p, s); break;
                case 4: e = launch_rowln_bwd<4>(p, s); break;
                default: e = launch_rowln_bwd<5>(p, s); break;
            }
            if (g_prof_on) { (void)hipEventRecord(rec.b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof->push_back(rec); }
            if (e) return e;
            ORTK_CHECK_LAUNCH();
            return 0;
        }
        gemm16_fn g = p.ln_mode == 1 ? gemm_bf16_row512_kernel<1> : gemm_bf16_row512_kernel<2>;
        ortk::lds_attr(reinterpret_cast<const void*>(g), RP_LDS_BYTES);
        hipStream_t s = ortk_s(stream);
        ProfRec rec{};
        if (g_prof_on) {
            if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) return ORTK_EINVAL;
            rec.key = 4; rec.flops = 2.0 * p.M * p.N * p.K;
            rec.bytes = (double)p.M * p.K * 2 + (double)p.N * p.K * 2 + (double)p.M * p.N * (4 + 4 + 2 + (p.ln_mode == 2 ? 8 : 0));
            (void)hipEventRecord(rec.a, s);
        }
        hipLaunchKernelGGL(g, dim3((unsigned)ortk_cdiv(p.M, RP_M)), dim3(512), RP_LDS_BYTES, s, p, 0, 0, 0);
        if (g_prof_on) { (void)hipEventRecord(rec.b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof->push_back(rec); }
        ORTK_CHECK_LAUNCH();
        return 0;
    }
    const int tilesM = (int)ortk_cdiv(p.M, BM), tilesN = (int)ortk_cdiv(p.N, BN);
    // (fp32 transposed layouts on the split kernels consume K 32 at a time: K-split chunks are made multiples of 32 for them)
    const int bk = p.precision ? BK16 : (ortk::tuning().f32_split && (p.transA || p.transB)) ? 32 : 16;
    int splitk = p.accumulate ? (p.splitk > 0 ? p.splitk : 1) : 1;
    int kchunk = bk;
    if (p.K <= 0) {
        splitk = 1;  // empty reduction: C = epilogue(0), bias / residual still applied
    } else {
        const int ksteps = (int)ortk_cdiv(p.K, bk);
        if (splitk > ksteps) splitk = ksteps;
        kchunk = (int)ortk_cdiv(ksteps, splitk) * bk;
        splitk = (int)ortk_cdiv(p.K, kchunk);
    }
    dim3 grid((unsigned)(tilesM * tilesN * splitk)), block(256);
    hipStream_t s = ortk_s(stream);
    const int key = (p.precision ? 4 : 0) | (p.transA ? 2 : 0) | (p.transB ? 1 : 0);
    ProfRec rec{};
    if (g_prof_on) {
        if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) return ORTK_EINVAL;
        rec.key = key; rec.flops = 2.0 * p.M * p.N * p.K;
        rec.bytes = (double)p.M * p.K * ortk_esize(p.a_dtype) + (double)p.N * p.K * ortk_esize(p.b_dtype) +
                    (double)p.M * p.N * (ortk_esize(p.c_dtype) + (p.resid ? 4 : 0) + (p.gate ? ortk_esize(p.gate_dtype) : 0));
        (void)hipEventRecord(rec.a, s);
    }
    if (!p.precision) {
        if (p.tile_stats || p.tile_samp) return ORTK_EINVAL;
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        if (key == 0 && ortk::tuning().f32_split && !p.accumulate && p.K > 0 && p.K % 32 == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0 &&
            al16(p.A) && al16(p.B)) {
            // three-way bf16 split on the bf16 matrix cores (gemm_f32x3_kernel); tile size by how many workgroups it gives the chip
            // Instance by how the grid fills the 256 compute units (measured on the shapes of the fp32 parity decode:
            // scratch/f32x3_bench.py, profiles/r05_f32_split_gemm.txt): the 256 x 128 pipelined tile wherever its grid fills at least
            // 3/4 of its last round of workgroups; short grids (at most 1.25 workgroups of 128 x 128 per unit) the 64 x 64 pipelined
            // tile; in between 128 x 64, three workgroups per unit.
            const int64_t t128 = (int64_t)tilesM * tilesN, big = ortk_cdiv(p.M, 256) * (int64_t)tilesN;
            const int force = ortk::tuning().f32_split;     // 1 automatic | 2..7 a fixed kernel instance
            const int shape = force > 1 ? force : big * 4 >= ortk_cdiv(big, 256) * 256 * 3 ? 7 : t128 < 320 ? 5 : 3;
#define ORTK_X3(WM_, WN_, GM_, GN_, MB_) do { const int tm = (int)ortk_cdiv(p.M, 32 * WM_ * GM_), tn = (int)ortk_cdiv(p.N, 32 * WN_ * GN_); \
            hipLaunchKernelGGL((gemm_f32x3_kernel<WM_, WN_, GM_, GN_, MB_>), dim3((unsigned)(tm * tn)), dim3(64 * GM_ * GN_), 0, s, p, tm, tn, 0); } while (0)
            switch (shape) {
                case 2:  ORTK_X3(1, 1, 2, 2, 2); break;               // single-buffered: 64 x 64 (24 KB)
                case 3:  ORTK_X3(2, 1, 2, 2, 3); break;               //                  128 x 64 (36 KB), three workgroups per unit
                case 4:  ORTK_X3(2, 2, 2, 2, 3); break;               //                  128 x 128 (48 KB), three
                case 5:  launch_f32x3p<1, 1, 2, 2, 3>(p, s); break;   // pipelined:       64 x 64 (48 KB)
                case 6:  launch_f32x3p<2, 1, 2, 2, 2>(p, s); break;   //                  128 x 64 (72 KB)
                default: launch_f32x3p<2, 2, 4, 2, 1>(p, s); break;   //                  256 x 128, 8 waves (144 KB)
            }
#undef ORTK_X3
            if (g_prof_on) { (void)hipEventRecord(rec.b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof->push_back(rec); }
            ORTK_CHECK_LAUNCH();
            return 0;
        }
        // transposed-operand layouts of the split product (gemm_f32x3t_kernel): dgrad without accumulation, wgrad with or without
        if (key != 0 && ortk::tuning().f32_split && p.K > 0 && (p.ldb & 3) == 0 && al16(p.B) && (p.N & 3) == 0 && p.N >= 4 && kchunk % 32 == 0 &&
            (key == 1 ? !p.accumulate && p.K % 32 == 0 && (p.lda & 3) == 0 && al16(p.A)
                      : (p.lda & 3) == 0 && al16(p.A) && (p.M & 3) == 0 && p.M >= 4)) {
            typedef void (*x3t_fn)(ortk_gemm_args, int, int, int);
            const x3t_fn fn = key == 1 ? gemm_f32x3t_kernel<false, false> : p.accumulate ? gemm_f32x3t_kernel<true, true> : gemm_f32x3t_kernel<true, false>;
            const size_t lds = p.accumulate ? BF16_LDS_BYTES_C : (size_t)6 * 128 * 32 * sizeof(__bf16);
            ortk::lds_attr(reinterpret_cast<const void*>(gemm_f32x3t_kernel<true, true>), BF16_LDS_BYTES_C);
            hipLaunchKernelGGL(fn, grid, block, lds, s, p, tilesM, tilesN, kchunk);
            if (g_prof_on) { (void)hipEventRecord(rec.b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof->push_back(rec); }
            ORTK_CHECK_LAUNCH();
            return 0;
        }
        // (the fp32 MFMA kernel has no fused column sums: the same sums from their own launch)
        if (p.colsum) { if (int e = ortk_colsum(p.A, ORTK_F32, p.lda, p.colsum, p.K, p.M, stream)) return e; }
        switch (key) {
            case 0: hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, p, tilesM, tilesN, kchunk); break;
            case 1: hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, p, tilesM, tilesN, kchunk); break;
            case 3: hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, p, tilesM, tilesN, kchunk); break;
            default: return ORTK_EINVAL;
        }
    } else {
        auto al = [](const void* q, size_t a_) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) % a_) == 0; };
        const size_t ea = ortk_esize(p.a_dtype), eb = ortk_esize(p.b_dtype), ec = ortk_esize(p.c_dtype), eg = ortk_esize(p.gate_dtype);
        // everything the unguarded kernels assume, except the row count: full column / K tiles and vector alignment
        const bool fast_nk = p.N % BN == 0 && p.K > 0 && p.K % BK16 == 0 && kchunk % BK16 == 0 &&
                             al(p.A, 16) && al(p.B, 16) && (p.lda * ea) % 16 == 0 && (p.ldb * eb) % 16 == 0 &&
                             al(p.C, 4 * ec) && (p.ldc % 4) == 0 && al(p.bias, 16) && al(p.resid, 16) && (p.ldr % 4) == 0 &&
                             al(p.gate, 4 * eg) && (p.ldg % 4) == 0;
        const bool fast = fast_nk && p.M % BM == 0;
        // The forward-layout LDS-DMA kernels also take a RAGGED row count (M = images x regions, captions x positions, images x
        // beams: whatever the batch is): the operand rows of the partial last row tile are clamped to M - 1 and only that tile
        // runs the bounds-checked epilogue.  (Without this every batch size that is not a multiple of 128 images fell back to
        // the guarded register-staged kernel: 13.4 ms per XE step at 127 images against 9.0 ms at 128.)
        const bool fast4 = fast_nk && key == 4 && !p.accumulate;
        if (p.tile_samp && !p.tile_stats) return ORTK_EINVAL;      // (the combine step needs the partials beside the candidates)
        const bool want_stats = p.tile_stats != nullptr;      // soft-max partials (+ sampling candidates): the 128 x 128 LDS-DMA kernel's epilogue
        if (want_stats && !(fast4 && p.a_dtype == ORTK_BF16 && p.b_dtype == ORTK_BF16 && kchunk % GBK == 0 && p.K % HBK == 0 && !p.relu &&
                            p.drop_p == 0.f && !p.gate && !p.rowscale && !p.resid && p.stat_ncols > 0 && p.stat_ncols <= p.N))
            return ORTK_EINVAL;
        const int impl = ortk::tuning().gemm_impl;   // experiments: 1 = register-staged kernel only, 2 = 128^2 DMA tiles only, 3 = 256^2 whenever legal
        // Measured in the XE step (bench.py, ms/step): register-staged kernel everywhere 17.9; DMA kernels everywhere
        // 21.3 (the k-major layouts lose: dgrad 4.1 vs 3.3 ms, wgrad 4.1 vs 3.4 ms per step in isolation); the DMA kernels
        // therefore serve the forward layout only unless ORTK_GEMM_IMPL >= 2 asks for them everywhere.
        const bool dma_layout = key == 4 || impl >= 2;
        // short forward grids (decode-time projections): 64 x 64 tiles
        // Measured with bias + residual epilogues (scratch/gemm_t64.py, us, 128^2/256^2 kernels -> 64^2): 1536x512x512 16.4 -> 7.9,
        // 1536x512x2048 33.3 -> 15.6, 5120x512x512 16.8 -> 10.5, 5120x512x2048 35.9 -> 23.4, 5120x2048x512 31.4 -> 29.1,
        // 9216x512x512 23.3 -> 17.6; past ~640 big tiles or with the generator's N = 10240 the small tiles lose (5120x10240x512
        // 163 -> 178, 21760x1536x512 87 -> 106).  Decode-sized row counts only (M <= 6 144): inside the training step, beside
        // the side stream's weight gradients, the 9 216- and 16 640-row projections are faster on the big tiles (XE step
        // 12.67 ms with them on the 64 x 64 tiles, 12.51 without).
        const int t64 = ortk::tuning().gemm_t64;     // use them while the 128 x 128 grid has at most this many workgroups (-1 = never)
        if (fast4 && impl != 1 && !want_stats && p.a_dtype == ORTK_BF16 && p.b_dtype == ORTK_BF16 && p.K % HBK == 0 &&
            p.drop_p == 0.f && !p.gate && !p.rowscale && (int64_t)tilesM * tilesN <= t64 && p.N <= 2048 && p.M <= 6144) {
            const int tm = (int)ortk_cdiv(p.M, 64), tn = p.N / 64;
            const bool one = (int64_t)tm * tn <= 256 + 64;          // one workgroup per CU: the whole K = 512 panel in flight
            gemm16_fn g = one ? gemm_bf16_dma64_kernel<8> : gemm_bf16_dma64_kernel<3>;    // else three per CU
            const size_t lds = (one ? 8 : 3) * DMA64_STAGE_BYTES;
            ortk::lds_attr(reinterpret_cast<const void*>(g), lds);
            hipLaunchKernelGGL(g, dim3((unsigned)(tm * tn)), dim3(256), lds, s, p, tm, tn, kchunk);
            if (g_prof_on) { (void)hipEventRecord(rec.b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof->push_back(rec); }
            ORTK_CHECK_LAUNCH();
            return 0;
        }
        if (want_stats && tilesN > 0xFFFF) return ORTK_EINVAL;       // (the 128 x 128 kernel's column-tile argument is 16 bits wide: 8.3 M columns)
        if ((fast || fast4) && (impl != 1 || want_stats) && dma_layout && p.a_dtype == ORTK_BF16 && p.b_dtype == ORTK_BF16 && kchunk % GBK == 0 && tilesN <= 0xFFFF) {
            // 256 x 256 tiles when they still give enough workgroups (and no split-K accumulation, which needs the
            // staged 128 x 128 epilogue); impl 2 = small tiles only, impl 3 = big tiles whenever legal
            const int64_t big_blocks = (int64_t)ortk_cdiv(p.M, 256) * (p.N / 256);
            // measured (scratch/gemm_shapes.py): the big tile wins whenever its grid fills >= 60 % of the CU slots of its
            // last round (170 blocks: 29.6 vs 33.7 us; 510: 58 vs 72 us) and loses on short grids (72 blocks: 26 vs 17 us;
            // 288 blocks = 1.1 rounds: 50 vs 44 us)
            const int64_t rounds = (big_blocks + 255) / 256;
            // In the training step the threshold is 50 %: the 130-tile launches of the valid-position decoder (16 640 x 512) leave
            // the other half of the chip to the weight-gradient GEMM of the side stream (XE step 12.47 -> 12.10 ms; 40 % and
            // 30 % measure the same)
            const bool fills = big_blocks * 10 >= rounds * 256 * 5;
            const bool lean = key == 4 && !p.accumulate && !p.rowscale && !(ortk::tuning().gemm_epilogue & 1);
            // Round 6: a column count of 256 n + 128 (the padded vocabulary: 10 112 = 39 x 256 + 128) runs its first 256 n columns on the
            // 256 x 256 tiles and the last 128 as a second launch of the 128 x 128 kernel on the same stream: the training-time generator
            // (16 640 / 21 760 rows; XE step 10.03 -> 10.00 ms, SCST step 23.06 -> 22.94).  NOT the launches that carry soft-max
            // partials: the decode-time generator (5 120 rows) is slower that way — 18.00 vs 17.65 ms per 1 024-image decode,
            // scratch/decode_tuning_ab.py — as it was in round 5 (the statistics epilogue has nothing to overlap with at one workgroup per
            // unit), so launches with statistics never take the big tile.
            // (ortk_tuning.gemm_epilogue & 2: never split.)
            const bool nsplit = lean && !want_stats && !(ortk::tuning().gemm_epilogue & 2) && p.N % 256 == 128 && p.N > 256 && p.K % HBK == 0;
            const bool big = !p.accumulate && !want_stats && (p.M % 256 == 0 || fast4) &&
                             (p.N % 256 == 0 || nsplit) && impl != 2 && (impl == 3 || fills);
            // 8-deep ring for grids of at most one workgroup per CU (decode-time projections): measured SLOWER in the
            // 1024-image decode (36.9 vs 35.8 ms) -> experiment only (ORTK_GEMM_IMPL=5)
            const bool deep = !big && (int64_t)tilesM * tilesN * splitk <= 256 && impl == 5;
            gemm16_fn gf;
            if (big)       gf = key == 4 ? gemm_bf16_glds_kernel<false, false, true, 4> : key == 5 ? gemm_bf16_glds_kernel<false, true, true, 4> : gemm_bf16_glds_kernel<true, true, true, 4>;
            else if (deep) gf = key == 4 ? gemm_bf16_glds_kernel<false, false, false, 8> : key == 5 ? gemm_bf16_glds_kernel<false, true, false, 8> : gemm_bf16_glds_kernel<true, true, false, 8>;
            else           gf = key == 4 ? gemm_bf16_glds_kernel<false, false, false, 4> : key == 5 ? gemm_bf16_glds_kernel<false, true, false, 4> : gemm_bf16_glds_kernel<true, true, false, 4>;
            // (the remainder launch of a split takes the small kernel's lean instance too)
            if (big && nsplit) gf = gemm_bf16_glds_kernel<false, false, false, 4>;
            if (lean && (!big || nsplit) && !deep) {
                if (want_stats) gf = p.tile_samp ? gemm_bf16_glds_kernel<false, false, false, 4, 5> : gemm_bf16_glds_kernel<false, false, false, 4, 4>;
                else switch ((p.drop_p > 0.f ? 1 : 0) | (p.gate ? 2 : 0)) {
                    case 0:  gf = gemm_bf16_glds_kernel<false, false, false, 4, 0>; break;
                    case 1:  gf = gemm_bf16_glds_kernel<false, false, false, 4, 1>; break;
                    case 2:  gf = gemm_bf16_glds_kernel<false, false, false, 4, 2>; break;
                    default: gf = gemm_bf16_glds_kernel<false, false, false, 4, 3>; break;
                }
            }
            const size_t lds = (big && !nsplit) ? GLDS_LDS_BYTES_BIG : deep ? 2 * GLDS_RING_BYTES : GLDS_LDS_BYTES;
            ortk::lds_attr(reinterpret_cast<const void*>(gf), lds);
            if (big && impl != 6 && p.K % HBK == 0) {
                // 64-column stages (full cache lines); impl 6 = the 32-column 4-stage ring for comparison
                gemm16_fn g2 = key == 4 ? gemm_bf16_dma256_kernel<false, false> : key == 5 ? gemm_bf16_dma256_kernel<false, true>
                                                                                              : gemm_bf16_dma256_kernel<true, true>;
                if (lean) {
                    switch ((p.drop_p > 0.f ? 1 : 0) | (p.gate ? 2 : 0)) {
                        case 0:  g2 = gemm_bf16_dma256_kernel<false, false, 0>; break;
                        case 1:  g2 = gemm_bf16_dma256_kernel<false, false, 1>; break;
                        case 2:  g2 = gemm_bf16_dma256_kernel<false, false, 2>; break;
                        default: g2 = gemm_bf16_dma256_kernel<false, false, 3>; break;
                    }
                }
                ortk::lds_attr(reinterpret_cast<const void*>(g2), DMA256_LDS_BYTES);
                hipLaunchKernelGGL(g2, dim3((unsigned)big_blocks), dim3(512), DMA256_LDS_BYTES, s, p, (int)ortk_cdiv(p.M, 256), p.N / 256, kchunk);
                if (nsplit)      // the last 128 columns: one column tile of the 128 x 128 kernel, starting at tile (N / 128 - 1)
                    hipLaunchKernelGGL(gf, dim3((unsigned)tilesM), block, lds, s, p, tilesM, 1 | ((tilesN - 1) << 16), kchunk);
            }
            else if (big) hipLaunchKernelGGL(gf, dim3((unsigned)big_blocks), dim3(512), lds, s, p, (int)ortk_cdiv(p.M, 256), p.N / 256, kchunk);
            else          hipLaunchKernelGGL(gf, grid, block, lds, s, p, tilesM, tilesN, kchunk);
            if (g_prof_on) { (void)hipEventRecord(rec.b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof->push_back(rec); }
            ORTK_CHECK_LAUNCH();
            return 0;
        }
        gemm16_fn fn = key == 4 ? pick16<false, false>(p.a_dtype, p.b_dtype, fast)
                     : key == 5 ? pick16<false, true>(p.a_dtype, p.b_dtype, fast)
                                : pick16<true, true>(p.a_dtype, p.b_dtype, fast);
        // > 64 KB of dynamic LDS needs the attribute once per kernel instance and device
        ortk::lds_attr(reinterpret_cast<const void*>(fn), BF16_LDS_BYTES);
        hipLaunchKernelGGL(fn, grid, block, BF16_LDS_BYTES, s, p, tilesM, tilesN, kchunk);
    }
    if (g_prof_on) { (void)hipEventRecord(rec.b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof->push_back(rec); }
    ORTK_CHECK_LAUNCH();
    return 0;
}

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
#include <vector>
#include "ortk_common.h"

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
    float* C; int64_t ldc;
    const float* bias; const float* rowscale; const float* resid; int64_t ldr;
    const float* gate; int64_t ldg; float gate_scale;
    int relu; float drop_p; uint32_t drop_seed; int accumulate; bool first_split;
    int M, N;
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
__device__ __forceinline__ void epilogue_tile(const Epi& e, int mrow0, int ncol0, f32x4 (&acc)[4][4]) {
    const bool first = e.first_split;
    const bool vec_b = e.bias && ((reinterpret_cast<uintptr_t>(e.bias) & 15) == 0);
    const bool vec_r = e.resid && ((e.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.resid) & 15) == 0);
    const bool vec_g = e.gate && ((e.ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.gate) & 15) == 0);
    const bool vec_c = ((e.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(e.C) & 15) == 0);
    const float inv_keep = e.drop_p > 0.f ? 1.f / (1.f - e.drop_p) : 1.f;
    float4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n0 = ncol0 + 16 * j;
        bias4[j] = (e.bias && first && n0 < e.N) ? ldrow4(e.bias + n0, n0, e.N, vec_b) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mrow0 + 16 * i;
        if (m >= e.M) continue;
        const float rs = e.rowscale ? e.rowscale[m] : 1.f;
        float4 res[4], gat[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = ncol0 + 16 * j;
            res[j] = (e.resid && first && n0 < e.N) ? ldrow4(e.resid + (int64_t)m * e.ldr + n0, n0, e.N, vec_r) : make_float4(0.f, 0.f, 0.f, 0.f);
            gat[j] = (e.gate && n0 < e.N) ? ldrow4(e.gate + (int64_t)m * e.ldg + n0, n0, e.N, vec_g) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = ncol0 + 16 * j;
            if (n0 >= e.N) continue;
            const float bb[4] = {bias4[j].x, bias4[j].y, bias4[j].z, bias4[j].w};
            const float rr[4] = {res[j].x, res[j].y, res[j].z, res[j].w};
            const float gg[4] = {gat[j].x, gat[j].y, gat[j].z, gat[j].w};
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][j][r] + bb[r];
                if (e.relu) x = fmaxf(x, 0.f);
                x *= rs;
                if (e.drop_p > 0.f) x = ortk_keep(e.drop_seed, (uint64_t)m * (uint64_t)e.N + (n0 + r), e.drop_p) ? x * inv_keep : 0.f;
                if (e.gate) x = gg[r] > 0.f ? x * e.gate_scale : 0.f;
                v[r] = x + rr[r];
            }
            float* c = e.C + (int64_t)m * e.ldc + n0;
            if (e.accumulate) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < e.N) atomicAdd(c + r, v[r]);
            } else if (vec_c && n0 + 3 < e.N) {
                *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < e.N) c[r] = v[r];
            }
        }
    }
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

    const bool fullA = vecA && mb + BM <= p.M, fullB = vecB && nb + BN <= p.N;   // workgroup-uniform
    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
        const bool kfull = k0 + BK <= k_end;
        if (fullA && kfull) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = tid + 256 * u;
                ra[u] = *reinterpret_cast<const float4*>(!TA ? p.A + (int64_t)(mb + (f >> 2)) * p.lda + k0 + 4 * (f & 3)
                                                             : p.A + (int64_t)(k0 + (f >> 5)) * p.lda + mb + 4 * (f & 31));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = tid + 256 * u;
                if (!TA) ra[u] = ld4(p.A, p.lda, mb + (f >> 2), k0 + 4 * (f & 3), p.M, k_end, vecA);
                else     ra[u] = ld4(p.A, p.lda, k0 + (f >> 5), mb + 4 * (f & 31), k_end, p.M, vecA);
            }
        }
        if (fullB && kfull) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = tid + 256 * u;
                rb[u] = *reinterpret_cast<const float4*>(!TB ? p.B + (int64_t)(nb + (f >> 2)) * p.ldb + k0 + 4 * (f & 3)
                                                             : p.B + (int64_t)(k0 + (f >> 5)) * p.ldb + nb + 4 * (f & 31));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = tid + 256 * u;
                if (!TB) rb[u] = ld4(p.B, p.ldb, nb + (f >> 2), k0 + 4 * (f & 3), p.N, k_end, vecB);
                else     rb[u] = ld4(p.B, p.ldb, k0 + (f >> 5), nb + 4 * (f & 31), k_end, p.N, vecB);
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

    Epi e{p.C, p.ldc, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, p.accumulate, ks == 0, p.M, p.N};
    epilogue_tile(e, mb + wm * 64 + lr, nb + wn * 64 + 4 * lk, acc);
}

// ------------------------------------------------------------------------------------------------ bf16 MFMA
// Same 128x128 tiling, K consumed 64 at a time (32 MFMA 16x16x32 per wave between barriers).  Operands are fp32 in
// memory and are converted to bf16 while staging.  The LDS image of an operand follows its GLOBAL layout so that the
// staging writes are always 8-byte vector stores:
//   * k-contiguous operand (activations X, weights W as (N,K)):  image [m][k], pitch 72 bf16;  the 8-element MFMA
//     fragment (8 consecutive k of one row) is one ds_read_b128;
//   * k-major operand (dY^T / X for wgrad, W for dgrad):           image [k][m], pitch 136 bf16;  the fragment is
//     gathered by two hardware-transposing reads (ds_read_b64_tr_b16).
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int BK16 = 64;
constexpr int PITCH_MK = BK16 + 8;     // [m][k] image: 144 B rows (16-B aligned, rows spread over banks)
constexpr int PITCH_KM = 128 + 8;      // [k][m] image: 272 B rows (8-B aligned for the transposing read)
constexpr int IMG_ELEMS = 128 * PITCH_MK > BK16 * PITCH_KM ? 128 * PITCH_MK : BK16 * PITCH_KM;

__device__ __forceinline__ bf16x4 tr_read(const __bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ bf16x4 cvt4(const float4& v) {
    return (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}

// stage one 128 x 64 operand tile: global fp32 -> registers (8 float4 per thread).
// `fast` (workgroup-uniform): the whole tile is in bounds and 16-B aligned -> 8 unconditional, independent
// global_load_dwordx4 that stay in flight together.  (Per-load bounds branches made hipcc drain vmcnt at every
// join: 16 serialized memory round trips per K-step, ~4 % MFMA utilisation in the first profile.)
template <bool T>
__device__ __forceinline__ void g2r(float4 (&r)[8], const float* __restrict__ base, int64_t ld, int tile0, int k0, int nmn, int k_end,
                                    bool vec, bool fast, int tid) {
    if (fast) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = tid + 256 * u;
            const float* p = !T ? base + (int64_t)(tile0 + (f >> 4)) * ld + k0 + 4 * (f & 15)
                                : base + (int64_t)(k0 + (f >> 5)) * ld + tile0 + 4 * (f & 31);
            r[u] = *reinterpret_cast<const float4*>(p);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = tid + 256 * u;
            if (!T) r[u] = ld4(base, ld, tile0 + (f >> 4), k0 + 4 * (f & 15), nmn, k_end, vec);
            else    r[u] = ld4(base, ld, k0 + (f >> 5), tile0 + 4 * (f & 31), k_end, nmn, vec);
        }
    }
}
template <bool T>
__device__ __forceinline__ void r2s(const float4 (&r)[8], __bf16* img, int tid) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int f = tid + 256 * u;
        if (!T) *reinterpret_cast<bf16x4*>(img + (f >> 4) * PITCH_MK + 4 * (f & 15)) = cvt4(r[u]);
        else    *reinterpret_cast<bf16x4*>(img + (f >> 5) * PITCH_KM + 4 * (f & 31)) = cvt4(r[u]);
    }
}
// fragment of 16 rows starting at m0 for k-sub-step ks (32 wide)
template <bool T>
__device__ __forceinline__ bf16x8 frag(const __bf16* img, int m0, int ks, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    if (!T) {
        return *reinterpret_cast<const bf16x8*>(img + (m0 + lr) * PITCH_MK + ks * 32 + 8 * lg);
    } else {
        // lane 4q+p of a 16-lane group supplies &img[k0 + q][m0 + 4p]; it receives column (lane & 15) of 4 k-rows
        const __bf16* p = img + (ks * 32 + 8 * lg + (lr >> 2)) * PITCH_KM + m0 + 4 * (lane & 3);
        const bf16x4 lo = tr_read(p), hi = tr_read(p + 4 * PITCH_KM);
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

template <bool TA, bool TB>
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
    const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
    const bool vecB = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool fullA = vecA && mb + BM <= p.M, fullB = vecB && nb + BN <= p.N;   // workgroup-uniform
    float4 ra[8], rb[8];
    int buf = 0;
    if (k_begin < k_end) {
        const bool kfull = k_begin + BK16 <= k_end;
        g2r<TA>(ra, p.A, p.lda, mb, k_begin, p.M, k_end, vecA, fullA && kfull, tid);
        g2r<TB>(rb, p.B, p.ldb, nb, k_begin, p.N, k_end, vecB, fullB && kfull, tid);
        r2s<TA>(ra, sA(0), tid);
        r2s<TB>(rb, sB(0), tid);
    }
    __syncthreads();
    for (int k0 = k_begin; k0 < k_end; k0 += BK16) {
        const bool has_next = k0 + BK16 < k_end;
        if (has_next) {
            const bool kfull = k0 + 2 * BK16 <= k_end;
            g2r<TA>(ra, p.A, p.lda, mb, k0 + BK16, p.M, k_end, vecA, fullA && kfull, tid);
            g2r<TB>(rb, p.B, p.ldb, nb, k0 + BK16, p.N, k_end, vecB, fullB && kfull, tid);
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
            r2s<TA>(ra, sA(buf ^ 1), tid);
            r2s<TB>(rb, sB(buf ^ 1), tid);
        }
        __syncthreads();
        buf ^= 1;
    }
    Epi e{p.C, p.ldc, p.bias, p.rowscale, p.resid, p.ldr, p.gate, p.ldg, p.gate_scale,
          p.relu, p.drop_p, p.drop_seed, p.accumulate, ks_ == 0, p.M, p.N};
    epilogue_tile(e, mb + wm * 64 + (lane & 15), nb + wn * 64 + 4 * (lane >> 4), acc);
}
constexpr size_t BF16_LDS_BYTES = (size_t)4 * IMG_ELEMS * sizeof(__bf16);

}  // namespace

// ------------------------------------------------------------------------------------------------ profiling hook
// Opt-in, measurement only (bench.py's roofline leg): HIP events around every GEMM launch on the launch stream,
// accumulated per kernel instance (precision, transA, transB).  Disabled by default; the timed region of bench.py
// never runs with it on.  This is the only process-global state in the library.
namespace {
struct ProfRec { hipEvent_t a, b; int key; double flops; };
bool g_prof_on = false;
std::vector<ProfRec>* g_prof = nullptr;
}  // namespace

extern "C" int ortk_prof_enable(int32_t on) {
    if (!g_prof) g_prof = new std::vector<ProfRec>();
    for (auto& r : *g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof->clear();
    g_prof_on = on != 0;
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

extern "C" int ortk_gemm(const ortk_gemm_args* a, ortk_stream stream) {
    if (!a || !a->A || !a->B || !a->C || a->M < 0 || a->N < 0 || a->K < 0) return ORTK_EINVAL;
    if (a->M == 0 || a->N == 0) return 0;
    ortk_gemm_args p = *a;
    const int tilesM = (int)ortk_cdiv(p.M, BM), tilesN = (int)ortk_cdiv(p.N, BN);
    const int bk = p.precision ? BK16 : 16;
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
    if (p.transA && !p.transB) return ORTK_EINVAL;  // not needed by the path
    dim3 grid((unsigned)(tilesM * tilesN * splitk)), block(256);
    hipStream_t s = ortk_s(stream);
    const int key = (p.precision ? 4 : 0) | (p.transA ? 2 : 0) | (p.transB ? 1 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF16_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF16_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF16_LDS_BYTES);
        attr_set = true;
    }
    ProfRec rec{};
    if (g_prof_on) {
        if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) return ORTK_EINVAL;
        rec.key = key; rec.flops = 2.0 * p.M * p.N * p.K;
        (void)hipEventRecord(rec.a, s);
    }
    switch (key) {
        case 0: hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, p, tilesM, tilesN, kchunk); break;
        case 1: hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, p, tilesM, tilesN, kchunk); break;
        case 3: hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, p, tilesM, tilesN, kchunk); break;
        case 4: hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, BF16_LDS_BYTES, s, p, tilesM, tilesN, kchunk); break;
        case 5: hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, BF16_LDS_BYTES, s, p, tilesM, tilesN, kchunk); break;
        case 7: hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, BF16_LDS_BYTES, s, p, tilesM, tilesN, kchunk); break;
        default: return ORTK_EINVAL;
    }
    if (g_prof_on) { (void)hipEventRecord(rec.b, s); g_prof->push_back(rec); }
    ORTK_CHECK_LAUNCH();
    return 0;
}

// ortk_box.hip — relative-geometry attention bias of the Object Relation Transformer.
//
// Replaces BoxMultiHeadedAttention.BoxRelationalEmbedding + the 8 per-head Linear(64->1)+ReLU and the
// log(clamp(.,1e-6)) of box_attention (models/relation_transformer.py:196-256,177-183,286).
// The reference materialises the (B,S,S,64) embedding (85 MB at B=256) in EVERY one of the 6 encoder layers
// through ~25 elementwise kernels; here one thread owns one (image, i, j) pair, keeps the 64-vector in
// registers and emits the bias of all layers and heads in one pass: HBM traffic = boxes in, (L,B,H,S,S) out.
//
// Numerics follow the reference's fp32 op order exactly: 100*p first, then * dim_mat[k];
// dim_mat comes from the host (as torch evaluates 1/1000^(k/8) in fp32); sinf/cosf/logf are the accurate
// OCML forms (arguments reach ~690 rad: fast-math intrinsics are NOT acceptable), division is IEEE.
#include "ortk_common.h"

namespace {

constexpr int MAXL = 16;
struct Layers {
    const float* wg[MAXL];
    const float* bg[MAXL];
    float* dwg[MAXL];
    float* dbg[MAXL];
    float dim_mat[8];
};

// DG = 64: sin / cos embedding; DG = 4: the raw log-ratios (`no_box_trigonometric_embedding`, relation_transformer.py:243-256)
template <int DG>
__device__ __forceinline__ void pair_embedding(const float* __restrict__ boxes, int b, int i, int j, int S,
                                               const float (&dm)[8], float (&e)[DG]) {
    const float4 bi = *reinterpret_cast<const float4*>(boxes + ((int64_t)b * S + i) * 4);
    const float4 bj = *reinterpret_cast<const float4*>(boxes + ((int64_t)b * S + j) * 4);
    const float cxi = (bi.x + bi.z) * 0.5f, cyi = (bi.y + bi.w) * 0.5f;
    const float cxj = (bj.x + bj.z) * 0.5f, cyj = (bj.y + bj.w) * 0.5f;
    const float wi = (bi.z - bi.x) + 1.0f, hi = (bi.w - bi.y) + 1.0f;
    const float wj = (bj.z - bj.x) + 1.0f, hj = (bj.w - bj.y) + 1.0f;
    float pos[4];
    pos[0] = logf(fmaxf(fabsf((cxi - cxj) / wi), 1e-3f));
    pos[1] = logf(fmaxf(fabsf((cyi - cyj) / hi), 1e-3f));
    pos[2] = logf(wi / wj);
    pos[3] = logf(hi / hj);
    if (DG == 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) e[c] = pos[c];
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float p100 = 100.0f * pos[c];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float arg = p100 * dm[k];
                float sn, cs;
                sincosf(arg, &sn, &cs);            // one range reduction for both (the same OCML kernels as sinf / cosf)
                e[(c * 8 + k) % DG] = sn;
                e[(32 + c * 8 + k) % DG] = cs;
            }
        }
    }
}

template <int DG>
__global__ __launch_bounds__(256) void box_embedding_kernel(const float* __restrict__ boxes, Layers ly, float* __restrict__ out,
                                                            int B, int S) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * S * S) return;
    const int j = idx % S, i = (idx / S) % S, b = idx / ((int64_t)S * S);
    float e[DG];
    pair_embedding<DG>(boxes, b, i, j, S, ly.dim_mat, e);
#pragma unroll
    for (int k = 0; k < DG; ++k) out[idx * DG + k] = e[k];
}

template <int DG>
__global__ __launch_bounds__(256) void box_logbias_fwd_kernel(const float* __restrict__ boxes, Layers ly,
                                                              float* __restrict__ out, int L, int B, int S, int H) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t SS = (int64_t)S * S;
    if (idx >= (int64_t)B * SS) return;
    const int j = idx % S, i = (idx / S) % S, b = idx / SS;
    float e[DG];
    pair_embedding<DG>(boxes, b, i, j, S, ly.dim_mat, e);
    for (int l = 0; l < L; ++l) {
        const float* __restrict__ W = ly.wg[l];
        const float* __restrict__ bb = ly.bg[l];
        for (int h = 0; h < H; ++h) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < DG; ++k) acc += W[h * DG + k] * e[k];   // W is wave-uniform -> scalar loads
            acc += bb[h];
            const float g = fmaxf(acc, 0.f);
            out[(((int64_t)l * B + b) * H + h) * SS + (int64_t)i * S + j] = logf(fmaxf(g, 1e-6f));
        }
    }
}

// Backward: dpre = dscore / pre where pre > 1e-6 (relu active and clamp inactive), else 0;
// dWG[l,h,:] += sum_pairs dpre * e ; dbG[l,h] += sum_pairs dpre.
// 128 pairs per workgroup: embeddings in LDS, then a (H x 128) x (128 x 64) product per layer on the VALU
// (2 outputs per thread) and one atomicAdd per output per workgroup.
constexpr int PAIRS = 128;
template <int DG>
__global__ __launch_bounds__(256) void box_logbias_bwd_kernel(const float* __restrict__ boxes, Layers ly,
                                                              const float* __restrict__ dscore, int L, int B, int S, int H) {
    __shared__ float sE[PAIRS][DG + 1];
    __shared__ float sD[PAIRS][9];
    const int tid = threadIdx.x;
    const int64_t SS = (int64_t)S * S, total = (int64_t)B * SS;
    const int64_t p0 = (int64_t)blockIdx.x * PAIRS;
    if (tid < PAIRS) {
        const int64_t idx = p0 + tid;
        float e[DG];
        if (idx < total) {
            const int j = idx % S, i = (idx / S) % S, b = idx / SS;
            pair_embedding<DG>(boxes, b, i, j, S, ly.dim_mat, e);
        } else {
#pragma unroll
            for (int k = 0; k < DG; ++k) e[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < DG; ++k) sE[tid][k] = e[k];
    }
    __syncthreads();
    const int pp = tid & (PAIRS - 1), hh = tid >> 7;  // 2 threads per pair, each half of the heads
    const int64_t idx = p0 + pp;
    const int64_t b = idx / SS, ij = idx - b * SS;
    for (int l = 0; l < L; ++l) {
        const float* __restrict__ W = ly.wg[l];
        const float* __restrict__ bb = ly.bg[l];
        for (int h = hh; h < H; h += 2) {
            float d = 0.f;
            if (idx < total) {
                float acc = 0.f;
                for (int k = 0; k < DG; ++k) acc += W[h * DG + k] * sE[pp][k];
                acc += bb[h];
                if (acc > 1e-6f) d = dscore[(((int64_t)l * B + b) * H + h) * SS + ij] / acc;
            }
            if (h < 8) sD[pp][h] = d;
        }
        __syncthreads();
        for (int o = tid; o < H * DG; o += 256) {
            const int h = o / DG, k = o % DG;
            float acc = 0.f;
            for (int q = 0; q < PAIRS; ++q) acc += sD[q][h] * sE[q][k];
            atomicAdd(&ly.dwg[l][o], acc);
        }
        if (tid < H) {
            float acc = 0.f;
            for (int q = 0; q < PAIRS; ++q) acc += sD[q][tid];
            atomicAdd(&ly.dbg[l][tid], acc);
        }
        __syncthreads();
    }
}

int fill_layers(Layers& ly, const float* const* wg, const float* const* bg, float* const* dwg, float* const* dbg,
                const float* dim_mat, int L) {
    if (L < 0 || L > MAXL) return ORTK_EINVAL;
    for (int l = 0; l < L; ++l) {
        ly.wg[l] = wg ? wg[l] : nullptr;
        ly.bg[l] = bg ? bg[l] : nullptr;
        ly.dwg[l] = dwg ? dwg[l] : nullptr;
        ly.dbg[l] = dbg ? dbg[l] : nullptr;
    }
    for (int k = 0; k < 8; ++k) ly.dim_mat[k] = dim_mat ? dim_mat[k] : 0.f;   // NULL = non-trigonometric 4-d embedding
    return 0;
}

}  // namespace

extern "C" int ortk_box_embedding(const float* boxes, const float* dim_mat, float* out, int32_t B, int32_t S, ortk_stream stream) {
    if (!boxes || !out || B < 0 || S < 1) return ORTK_EINVAL;
    Layers ly;
    if (int e = fill_layers(ly, nullptr, nullptr, nullptr, nullptr, dim_mat, 0)) return e;
    const int64_t n = (int64_t)B * S * S;
    if (n == 0) return 0;
    if (dim_mat) hipLaunchKernelGGL(box_embedding_kernel<64>, dim3((unsigned)ortk_cdiv(n, 256)), dim3(256), 0, ortk_s(stream), boxes, ly, out, B, S);
    else         hipLaunchKernelGGL(box_embedding_kernel<4>, dim3((unsigned)ortk_cdiv(n, 256)), dim3(256), 0, ortk_s(stream), boxes, ly, out, B, S);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_box_logbias_fwd(const float* boxes, const float* const* wg, const float* const* bg, const float* dim_mat,
                                    float* out, int32_t L, int32_t B, int32_t S, int32_t H, ortk_stream stream) {
    if (!boxes || !wg || !bg || !out || B < 0 || S < 1 || H < 1) return ORTK_EINVAL;
    Layers ly;
    if (int e = fill_layers(ly, wg, bg, nullptr, nullptr, dim_mat, L)) return e;
    const int64_t n = (int64_t)B * S * S;
    if (n == 0 || L == 0) return 0;
    if (dim_mat) hipLaunchKernelGGL(box_logbias_fwd_kernel<64>, dim3((unsigned)ortk_cdiv(n, 256)), dim3(256), 0, ortk_s(stream), boxes, ly, out, L, B, S, H);
    else         hipLaunchKernelGGL(box_logbias_fwd_kernel<4>, dim3((unsigned)ortk_cdiv(n, 256)), dim3(256), 0, ortk_s(stream), boxes, ly, out, L, B, S, H);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_box_logbias_bwd(const float* boxes, const float* const* wg, const float* const* bg, const float* dim_mat,
                                    const float* dscore, float* const* dwg, float* const* dbg, int32_t L, int32_t B, int32_t S,
                                    int32_t H, ortk_stream stream) {
    if (!boxes || !wg || !bg || !dscore || !dwg || !dbg || B < 0 || S < 1 || H < 1 || H > 8) return ORTK_EINVAL;
    Layers ly;
    if (int e = fill_layers(ly, wg, bg, dwg, dbg, dim_mat, L)) return e;
    const int64_t n = (int64_t)B * S * S;
    if (n == 0 || L == 0) return 0;
    if (dim_mat) hipLaunchKernelGGL(box_logbias_bwd_kernel<64>, dim3((unsigned)ortk_cdiv(n, PAIRS)), dim3(256), 0, ortk_s(stream), boxes, ly, dscore, L, B, S, H);
    else         hipLaunchKernelGGL(box_logbias_bwd_kernel<4>, dim3((unsigned)ortk_cdiv(n, PAIRS)), dim3(256), 0, ortk_s(stream), boxes, ly, dscore, L, B, S, H);
    ORTK_CHECK_LAUNCH();
    return 0;
}

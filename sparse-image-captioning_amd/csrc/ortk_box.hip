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
#include "ortk_internal.h"

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
// 128 pairs per block of work: embeddings in LDS, then a (H x 128) x (128 x 64) product per layer on the VALU (2 outputs per
// thread).  A workgroup walks SEVERAL blocks and keeps its sums in LDS ([layer][H * DG + H] floats), so the 520 atomics per layer
// go out once per workgroup, not once per block: with one workgroup per block (2 592 at 256 images x 36 regions) the 6.7 M atomics
// of a five-layer launch all hit the same 2 600 addresses — the slow form of MI355X_MICROARCH.md, "Global float atomics:
// contention" — and the launch took 309 us beside the last encoder layer's backward, whose kernels it slowed 2-3x.
constexpr int PAIRS = 128;
constexpr int BOX_MAX_WGS = 512;          // two workgroups per compute unit
template <int DG>
__global__ __launch_bounds__(256) void box_logbias_bwd_kernel(const float* __restrict__ boxes, Layers ly,
                                                              const float* __restrict__ dscore, int L, int B, int S, int H, int nblocks) {
    extern __shared__ float smem_box[];
    float (*sE)[DG + 1] = reinterpret_cast<float (*)[DG + 1]>(smem_box);                         // [PAIRS][DG + 1]
    float (*sD)[9] = reinterpret_cast<float (*)[9]>(smem_box + PAIRS * (DG + 1));                // [PAIRS][9]
    float* sAcc = smem_box + PAIRS * (DG + 1) + PAIRS * 9;                                       // [L][H * DG + H]
    const int tid = threadIdx.x;
    const int NO = H * DG + H;                // outputs per layer: the weight block, then the bias
    for (int o = tid; o < L * NO; o += 256) sAcc[o] = 0.f;
    const int64_t SS = (int64_t)S * S, total = (int64_t)B * SS;
    const int pp = tid & (PAIRS - 1), hh = tid >> 7;  // 2 threads per pair, each half of the heads
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t p0 = (int64_t)blk * PAIRS;
        __syncthreads();                      // the previous block's reads of sE / sD are done (and sAcc is cleared)
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
                sAcc[l * NO + o] += acc;       // (thread tid owns outputs tid, tid + 256, ...: no conflict)
            }
            if (tid < H) {
                float acc = 0.f;
                for (int q = 0; q < PAIRS; ++q) acc += sD[q][tid];
                sAcc[l * NO + H * DG + tid] += acc;
            }
            __syncthreads();
        }
    }
    for (int l = 0; l < L; ++l) {
        for (int o = tid; o < H * DG; o += 256) atomicAdd(&ly.dwg[l][o], sAcc[l * NO + o]);
        if (tid < H) atomicAdd(&ly.dbg[l][tid], sAcc[l * NO + H * DG + tid]);
    }
}
template <int DG> static size_t box_bwd_lds(int L, int H) { return sizeof(float) * ((size_t)PAIRS * (DG + 1) + PAIRS * 9 + (size_t)L * (H * DG + H)); }

// The same backward with both products on the matrix cores (fp32 MFMA 16x16x4: exact fp32 products, fp32 accumulation — both
// precisions), trigonometric embedding (DG = 64), H <= 8.  Per block of 128 pairs and layer:
//     pre[pair][head] = E[pair][:] . WG[head][:] + bG[head]        8 row tiles x 16 k-steps   (wave w: row tiles 2w, 2w + 1)
//     d[pair][head]   = dscore / pre  where pre > 1e-6
//     dWG[head][k]   += sum_pairs d[pair][head] E[pair][k]          4 column tiles x 32 k-steps (wave w: columns 16w .. 16w + 15)
// 64 MFMAs per wave instead of ~1 500 dependent LDS-read + FMA pairs per thread: the five-layer launch of the XE step went
// 367 -> (see profiles/r06_box_bwd.txt) us, and it runs beside the last encoder layer's backward, whose kernels wait for units.
constexpr int BEP = 65;       // fp32 pitch of the embedding rows in LDS
__device__ __forceinline__ bool SS_is_mult4(int S) { return ((S * S) & 3) == 0; }
__global__ __launch_bounds__(256) void box_logbias_bwd_mfma_kernel(const float* __restrict__ boxes, Layers ly, const float* __restrict__ dscore,
                                                                   int L, int B, int S, int H, int nblocks) {
    extern __shared__ float smem_box[];
    float* sE = smem_box;                              // [PAIRS][BEP]
    float* sD = sE + PAIRS * BEP;                      // [PAIRS][9]      d[pair][head]
    float* sW = sD + PAIRS * 9;                        // [L][8][64]      WG (heads past H: zeros)
    float* sB = sW + L * 512;                          // [L][8]
    float* sAcc = sB + L * 8;                          // [L][520]        dWG (8 x 64), then dbG (8)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    long long* sOff = reinterpret_cast<long long*>(sAcc + L * 520);      // [PAIRS] offset of (image, head 0, i, j) inside a layer's dscore, or -1
    float* sDS = reinterpret_cast<float*>(sOff + PAIRS);                 // [2][8][PAIRS] the block's score gradients of a layer (double-buffered)
    // The score gradients reach LDS as whole rows: thread t stages head t / 32, pairs 4 (t % 32) .. + 3 (consecutive floats of one image's
    // (S x S) block whenever S * S is a multiple of 4: one 16-byte load).  Read straight from memory in the accumulator layout they were
    // 8 loads x 32 scattered 4-byte accesses per wave and layer — 2.6 M requests per layer for 10 MB: the launch's bound (30 us per layer).
    const int st_h = tid >> 5, st_p = (tid & 31) * 4;
    const bool vec4 = (SS_is_mult4(S)) && (reinterpret_cast<uintptr_t>(dscore) & 15) == 0;
    auto load_ds = [&](int l) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (st_h < H) {
            const float* __restrict__ dl = dscore + (int64_t)l * B * H * ((int64_t)S * S) + (int64_t)st_h * ((int64_t)S * S);
            if (vec4) {
                const long long off = sOff[st_p];
                if (off >= 0) v = *reinterpret_cast<const f32x4*>(dl + off);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { const long long off = sOff[st_p + q]; if (off >= 0) v[q] = dl[off]; }
            }
        }
        return v;
    };
    for (int o = tid; o < L * 512; o += 256) { const int l = o >> 9, h = (o >> 6) & 7; sW[o] = h < H ? ly.wg[l][h * 64 + (o & 63)] : 0.f; }
    for (int o = tid; o < L * 8; o += 256) sB[o] = (o & 7) < H ? ly.bg[o >> 3][o & 7] : 0.f;
    for (int o = tid; o < L * 520; o += 256) sAcc[o] = 0.f;
    const int64_t SS = (int64_t)S * S, total = (int64_t)B * SS;
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t p0 = (int64_t)blk * PAIRS;
        __syncthreads();                      // the previous block's reads of sE / sD are done (first pass: the tables are written)
        if (tid < PAIRS) {
            const int64_t idx = p0 + tid;
            float e[64];
            if (idx < total) {
                const int j = idx % S, i = (idx / S) % S, b = idx / SS;
                pair_embedding<64>(boxes, b, i, j, S, ly.dim_mat, e);
                sOff[tid] = (long long)b * H * SS + (idx - (int64_t)b * SS);       // (the 64-bit divisions once per pair, not per layer and head)
            } else {
#pragma unroll
                for (int k = 0; k < 64; ++k) e[k] = 0.f;
                sOff[tid] = -1;
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) sE[tid * BEP + k] = e[k];
        }
        __syncthreads();
        f32x4 nx = load_ds(0);
        *reinterpret_cast<f32x4*>(sDS + st_h * PAIRS + st_p) = nx;
        if (L > 1) nx = load_ds(1);
        __syncthreads();
        for (int l = 0; l < L; ++l) {
            const float* sds = sDS + (l & 1) * 8 * PAIRS;
            float bsum = 0.f;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int m0 = (2 * wave + tt) * 16;
                // (every operand read is issued before the first product: read-wait-multiply per k-step left the LDS latency exposed 16 times)
                float ea[16], wb[16];
#pragma unroll
                for (int s4 = 0; s4 < 16; ++s4) { ea[s4] = sE[(m0 + lr) * BEP + 4 * s4 + lk]; wb[s4] = lr < 8 ? sW[l * 512 + (lr & 7) * 64 + 4 * s4 + lk] : 0.f; }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 16; s4 += 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s4], wb[s4], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s4 + 1], wb[s4 + 1], acc2, 0, 0, 0);
                }
                acc += acc2;
                if (lr < 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float pre = acc[j] + sB[l * 8 + lr];
                        const float d = pre > 1e-6f ? sds[lr * PAIRS + m0 + 4 * lk + j] / pre : 0.f;      // (0 for heads past H and pairs past the end)
                        sD[(m0 + 4 * lk + j) * 9 + lr] = d;
                        bsum += d;
                    }
                }
            }
            if (lr < H) atomicAdd(&sAcc[l * 520 + 512 + lr], bsum);
            __syncthreads();
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float da[16], eb[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int s4 = 16 * half + u;
                    da[u] = lr < 8 ? sD[(4 * s4 + lk) * 9 + lr] : 0.f; eb[u] = sE[(4 * s4 + lk) * BEP + 16 * wave + lr];
                }
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da[u], eb[u], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(da[u + 1], eb[u + 1], acc2, 0, 0, 0);
                }
            }
            acc += acc2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int h = 4 * lk + j;
                if (h < H) sAcc[l * 520 + h * 64 + 16 * wave + lr] += acc[j];      // (one lane per entry: no conflict)
            }
            if (l + 1 < L) {
                *reinterpret_cast<f32x4*>(sDS + ((l + 1) & 1) * 8 * PAIRS + st_h * PAIRS + st_p) = nx;      // (buffer (l + 1) & 1 was last read in layer l - 1)
                if (l + 2 < L) nx = load_ds(l + 2);
            }
            __syncthreads();
        }
    }
    for (int l = 0; l < L; ++l) {
        for (int o = tid; o < H * 64; o += 256) atomicAdd(&ly.dwg[l][o], sAcc[l * 520 + o]);
        if (tid < H) atomicAdd(&ly.dbg[l][tid], sAcc[l * 520 + 512 + tid]);
    }
}
static size_t box_bwd_mfma_lds(int L) { return sizeof(float) * ((size_t)PAIRS * BEP + PAIRS * 9 + (size_t)L * (512 + 8 + 520) + 2 * 8 * PAIRS) + PAIRS * sizeof(long long) + 16; }

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
    const int nblocks = (int)ortk_cdiv(n, PAIRS), grid = nblocks < BOX_MAX_WGS ? nblocks : BOX_MAX_WGS;
    if (dim_mat) {
        const size_t lds = box_bwd_mfma_lds(L);
        if (ortk::lds_attr(reinterpret_cast<const void*>(box_logbias_bwd_mfma_kernel), lds)) return ORTK_EINVAL;
        hipLaunchKernelGGL(box_logbias_bwd_mfma_kernel, dim3((unsigned)grid), dim3(256), lds, ortk_s(stream), boxes, ly, dscore, L, B, S, H, nblocks);
    } else {
        const size_t lds = box_bwd_lds<4>(L, H);
        hipLaunchKernelGGL(box_logbias_bwd_kernel<4>, dim3((unsigned)grid), dim3(256), lds, ortk_s(stream), boxes, ly, dscore, L, B, S, H, nblocks);
    }
    ORTK_CHECK_LAUNCH();
    return 0;
}

// ortk_optim.hip — arena-wide elementwise passes: fused clip+Adam and the pruning-mask algebra.
//
// All parameters live in one flat arena (include/ortk.h), so each of these is ONE launch over 55 M (or 111 M)
// contiguous floats, float4-vectorised and HBM-bound, instead of one small kernel per tensor:
//   * clip_grad_value_ + Adam  (utils/optim.py:116-126,187-191; torch.optim.Adam's rule, no amsgrad / weight decay)
//   * MaskMixin.get_masked_weight for all 147 masked tensors (pruning/masked_layer.py:84-110) and the
//     straight-through backward of pruning/sampler.py:10-66.
#include "ortk_common.h"

namespace {

// ZERO: the gradient is cleared on the way (the next step accumulates into it: no separate fill pass over the arena)
template <bool ZERO>
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float step_size, float b1, float b2,
                                                        float eps, float clip, float sqrt_bc2) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float gc = fminf(fmaxf(G[u], -clip), clip);
            M[u] = M[u] + (gc - M[u]) * (1.f - b1);           // exp_avg.lerp_(grad, 1 - beta1)
            V[u] = V[u] * b2 + gc * gc * (1.f - b2);
            const float denom = sqrtf(V[u]) / sqrt_bc2 + eps;
            P[u] = P[u] - step_size * (M[u] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (ZERO) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gc = fminf(fmaxf(g[i], -clip), clip);
        const float mi = m[i] + (gc - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + gc * gc * (1.f - b2);
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
        if (ZERO) g[i] = 0.f;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// sample of element i under `mode`
// `draws` (optional, mode 1 only): explicit uniforms u[i] instead of the counter hash — the Bernoulli sample is then
// u[i] < sigmoid(m[i]), exactly torch.bernoulli's definition for a given uniform (parity tests inject the reference's draws)
__device__ __forceinline__ float mask_sample(float m, int mode, uint32_t seed, uint64_t i, const float* __restrict__ draws) {
    if (mode == 2) return m;
    const float pr = sigmoidf_(m);
    if (mode == 0) return rintf(pr);  // round-half-to-even like torch.round (logit 0 -> 0.5 -> 0)
    if (draws) return draws[i] < pr ? 1.f : 0.f;
    const uint32_t h = ortk_mix32((uint32_t)i * 0x9E3779B1u + (uint32_t)(i >> 32) * 0x85EBCA77u + seed);
    return ortk_u01(h) < pr ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void mask_apply_kernel(const float* __restrict__ w, const float* __restrict__ m,
                                                         float* __restrict__ we, int64_t n, int mode, uint32_t seed,
                                                         const float* __restrict__ draws) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        we[i] = mask_sample(m[i], mode, seed, (uint64_t)i, draws) * w[i];
}

__global__ __launch_bounds__(256) void mask_bwd_kernel(const float* __restrict__ dwe, const float* __restrict__ w,
                                                       const float* __restrict__ m, float* __restrict__ dw, float* __restrict__ dm,
                                                       int64_t n, int mode, uint32_t seed,
                                                       const float* __restrict__ extra_coef, const float* __restrict__ draws, int64_t index0) {
    // extra_coef[0] = d(sparsity loss)/d(sample), the same for every mask element (pruning/prune.py:228-269)
    // index0: arena position of element 0 of this call (a RANGE of the arena: the hash draws are keyed by the arena position)
    const float ec = extra_coef ? extra_coef[0] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float g = dwe[i], mi = m[i];
        const float s = mask_sample(mi, mode, seed, (uint64_t)(index0 + i), draws ? draws - index0 : nullptr);
        if (dm) {
            float ds = g * w[i] + ec;                  // d/ds of (s*w) + sparsity-loss term
            if (mode != 2) { const float pr = sigmoidf_(mi); ds *= pr * (1.f - pr); }
            dm[i] += ds;
        }
        dw[i] = g * s;                                  // may alias dwe
    }
}

__global__ __launch_bounds__(256) void mask_count_kernel(const float* __restrict__ m, int64_t n, int mode, float* __restrict__ out) {
    __shared__ float sh[4];
    float c = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        c += mode == 2 ? (m[i] != 0.f ? 1.f : 0.f) : rintf(sigmoidf_(m[i]));
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

// The whole element-wise tail of a masked training step in ONE pass over (a range of) the arena: straight-through mask backward
// (mask_bwd_kernel), frozen scopes, clip + Adam on the weights (gradient cleared on the way), clip + Adam on the mask logits — 7-8
// arrays read and 7 written instead of the 22 array passes of the four separate launches (mask backward, dm clear, two Adam groups).
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float step_size, float b1, float b2, float eps, float clip, float sqrt_bc2) {
    const float gc = fminf(fmaxf(g, -clip), clip);
    m = m + (gc - m) * (1.f - b1);
    v = v * b2 + gc * gc * (1.f - b2);
    p = p - step_size * (m / (sqrtf(v) / sqrt_bc2 + eps));
}
__global__ __launch_bounds__(256) void masked_adam_kernel(ortk_masked_adam_args a, float step_w, float step_m, float sqrt_bc2) {
    const float ec = a.extra_coef ? a.extra_coef[0] : 0.f;
    const bool tm = a.mm != nullptr;
    const int64_t stride = (int64_t)gridDim.x * 256;
    auto one = [&](int64_t i, float g, float& w, float& ml, float& mw, float& vw, float& mm, float& mv, float act) {
        const float s = mask_sample(ml, a.mode, a.seed, (uint64_t)(a.index0 + i), a.draws);
        float ds = g * w + ec;
        if (a.mode != 2) { const float pr = sigmoidf_(ml); ds *= pr * (1.f - pr); }
        ds *= act;
        adam1(w, g * s, mw, vw, step_w, a.beta1, a.beta2, a.eps_w, a.clip, sqrt_bc2);
        if (tm) adam1(ml, ds, mm, mv, step_m, a.beta1, a.beta2, a.eps_m, a.clip, sqrt_bc2);
    };
    const int64_t n4 = a.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 g4 = reinterpret_cast<const float4*>(a.g)[i], w4 = reinterpret_cast<float4*>(a.w)[i], l4 = reinterpret_cast<float4*>(a.ml)[i];
        float4 mw4 = reinterpret_cast<float4*>(a.mw)[i], vw4 = reinterpret_cast<float4*>(a.vw)[i];
        float4 mm4 = make_float4(0.f, 0.f, 0.f, 0.f), mv4 = mm4, ac4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (tm) { mm4 = reinterpret_cast<float4*>(a.mm)[i]; mv4 = reinterpret_cast<float4*>(a.mv)[i]; }
        if (a.active) ac4 = reinterpret_cast<const float4*>(a.active)[i];
        float* G = &g4.x; float* W = &w4.x; float* Lg = &l4.x; float* MW = &mw4.x; float* VW = &vw4.x; float* MM = &mm4.x; float* MV = &mv4.x; float* AC = &ac4.x;
#pragma unroll
        for (int u = 0; u < 4; ++u) one(4 * i + u, G[u], W[u], Lg[u], MW[u], VW[u], MM[u], MV[u], AC[u]);
        reinterpret_cast<float4*>(a.w)[i] = w4; reinterpret_cast<float4*>(a.mw)[i] = mw4; reinterpret_cast<float4*>(a.vw)[i] = vw4;
        if (tm) { reinterpret_cast<float4*>(a.ml)[i] = l4; reinterpret_cast<float4*>(a.mm)[i] = mm4; reinterpret_cast<float4*>(a.mv)[i] = mv4; }
        reinterpret_cast<float4*>(a.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        float w = a.w[i], ml = a.ml[i], mw = a.mw[i], vw = a.vw[i], mm = tm ? a.mm[i] : 0.f, mv = tm ? a.mv[i] : 0.f;
        one(i, a.g[i], w, ml, mw, vw, mm, mv, a.active ? a.active[i] : 1.f);
        a.w[i] = w; a.mw[i] = mw; a.vw[i] = vw;
        if (tm) { a.ml[i] = ml; a.mm[i] = mm; a.mv[i] = mv; }
        a.g[i] = 0.f;
    }
}

inline unsigned ew_grid(int64_t n) { return (unsigned)std::min<int64_t>(ortk_cdiv(n, 256), 4096); }

}  // namespace

extern "C" int ortk_masked_adam_step(const ortk_masked_adam_args* a, ortk_stream stream) {
    if (!a || !a->w || !a->g || !a->mw || !a->vw || !a->ml || a->n < 0 || a->index0 < 0 || a->mode < 0 || a->mode > 2) return ORTK_EINVAL;
    if ((a->mm == nullptr) != (a->mv == nullptr) || a->bc1 <= 0.f || a->bc2 <= 0.f) return ORTK_EINVAL;
    if (a->n == 0) return 0;
    auto al = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!(al(a->w) && al(a->g) && al(a->mw) && al(a->vw) && al(a->ml) && al(a->mm) && al(a->mv) && al(a->active))) return ORTK_EINVAL;
    ortk_masked_adam_args k = *a;
    if (k.draws) { k.draws -= k.index0; }       // (the kernel indexes draws by arena position; the caller passes the range's pointer)
    hipLaunchKernelGGL(masked_adam_kernel, dim3(ew_grid(a->n / 4 + 1)), dim3(256), 0, ortk_s(stream), k, a->lr_w / a->bc1, a->lr_m / a->bc1,
                       (float)sqrt((double)a->bc2));
    ORTK_CHECK_LAUNCH();
    return 0;
}

static int adam_clip_launch(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float clip,
                            float bc1, float bc2, bool zero, ortk_stream stream) {
    if (!p || !g || !m || !v || n < 0 || bc1 <= 0.f || bc2 <= 0.f) return ORTK_EINVAL;
    if (n == 0) return 0;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!(al(p) && al(g) && al(m) && al(v))) return ORTK_EINVAL;
    const float step_size = lr / bc1;
    const float sqrt_bc2 = (float)sqrt((double)bc2);
    if (zero) hipLaunchKernelGGL(adam_clip_kernel<true>, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ortk_s(stream), p, g, m, v, n, step_size,
                                 beta1, beta2, eps, clip, sqrt_bc2);
    else      hipLaunchKernelGGL(adam_clip_kernel<false>, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ortk_s(stream), p, g, m, v, n, step_size,
                                 beta1, beta2, eps, clip, sqrt_bc2);
    ORTK_CHECK_LAUNCH();
    return 0;
}
extern "C" int ortk_adam_clip(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float clip, float bc1, float bc2, ortk_stream stream) {
    return adam_clip_launch(p, const_cast<float*>(g), m, v, n, lr, beta1, beta2, eps, clip, bc1, bc2, false, stream);
}
extern "C" int ortk_adam_clip_zero(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                   float eps, float clip, float bc1, float bc2, ortk_stream stream) {
    return adam_clip_launch(p, g, m, v, n, lr, beta1, beta2, eps, clip, bc1, bc2, true, stream);
}

extern "C" int ortk_mask_apply(const float* w, const float* m, float* w_eff, int64_t n, int32_t mode, uint32_t seed,
                               ortk_stream stream) {
    if (!w || !m || !w_eff || n < 0 || mode < 0 || mode > 2) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, ortk_s(stream), w, m, w_eff, n, mode, seed, (const float*)nullptr);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_mask_apply_draws(const float* w, const float* m, const float* draws, float* w_eff, int64_t n, ortk_stream stream) {
    if (!w || !m || !draws || !w_eff || n < 0) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, ortk_s(stream), w, m, w_eff, n, 1, 0u, draws);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_mask_bwd(const float* dw_eff, const float* w, const float* m, float* dw, float* dm, int64_t n, int32_t mode,
                             uint32_t seed, const float* extra_coef_dev, ortk_stream stream) {
    if (!dw_eff || !w || !m || !dw || n < 0 || mode < 0 || mode > 2) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ortk_s(stream), dw_eff, w, m, dw, dm, n, mode, seed, extra_coef_dev,
                       (const float*)nullptr, (int64_t)0);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_mask_bwd_draws(const float* dw_eff, const float* w, const float* m, const float* draws, float* dw, float* dm,
                                   int64_t n, const float* extra_coef_dev, ortk_stream stream) {
    if (!dw_eff || !w || !m || !draws || !dw || n < 0) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ortk_s(stream), dw_eff, w, m, dw, dm, n, 1, 0u, extra_coef_dev, draws, (int64_t)0);
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_mask_count(const float* m, int64_t n, int32_t mode, float* count_dev, ortk_stream stream) {
    if (!m || !count_dev || n < 0 || mode < 0 || mode > 2) return ORTK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)std::min<int64_t>(ortk_cdiv(n, 256), 1024)), dim3(256), 0,
                       ortk_s(stream), m, n, mode, count_dev);
    ORTK_CHECK_LAUNCH();
    return 0;
}

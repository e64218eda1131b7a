// ortk_norm.hip — the reference's non-standard LayerNorm (models/transformer.py:329-341):
//     y = a_2 * (x - mean) / (std_unbiased + eps) + b_2
// One wave (64 lanes) per row, reductions by wavefront shuffles; a row of d <= 2048 floats stays in registers
// (statically indexed, so nothing spills to scratch): lane l owns columns {4l..4l+3} + 256*it (vector form)
// or l + 64*it (scalar form for d % 4 != 0 or unaligned rows).
#include <cstdlib>
#include "ortk_internal.h"

namespace {

// NREG = floats per lane: 8 covers d <= 512 (the model width), 32 covers d <= 2048.

template <bool VEC, int NREG> struct Cols {
    static constexpr int W = VEC ? 4 : 1;           // consecutive columns per iteration
    static constexpr int NIT = NREG / W;            // iterations
    static __device__ __forceinline__ int col(int lane, int it) { return VEC ? lane * 4 + 256 * it : lane + 64 * it; }
};

template <bool VEC, int NREG>
__device__ __forceinline__ void load_row(const float* __restrict__ x, int d, int lane, float (&v)[NREG]) {
    using C = Cols<VEC, NREG>;
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
        const int c = C::col(lane, it);
        if (VEC) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < d) t = *reinterpret_cast<const float4*>(x + c);
            v[it * 4] = t.x; v[it * 4 + 1] = t.y; v[it * 4 + 2] = t.z; v[it * 4 + 3] = t.w;
        } else {
            v[it] = c < d ? x[c] : 0.f;
        }
    }
}

template <bool VEC, int NREG>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                     const float* __restrict__ b, void* __restrict__ y, int y_dt,
                                                     float* __restrict__ stats, int64_t rows, int d, float eps) {
    using C = Cols<VEC, NREG>;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NREG];
    load_row<VEC, NREG>(x + row * d, d, lane, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += v[i];  // out-of-range slots hold 0
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < C::NIT; ++it)
#pragma unroll
        for (int u = 0; u < C::W; ++u)
            if (C::col(lane, it) + u < d) { const float t = v[it * C::W + u] - mean; q += t * t; }
    const float sd = sqrtf(wave_sum(q) / (float)(d - 1));
    const float rinv = 1.f / (sd + eps);
    if (stats && lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = sd; }
    const int64_t yoff = row * d;
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
        const int c = C::col(lane, it);
        if (c < d) {
            if (VEC) {
                const float4 aa = *reinterpret_cast<const float4*>(a + c), bb = *reinterpret_cast<const float4*>(b + c);
                float4 o;
                o.x = aa.x * (v[it * 4] - mean) * rinv + bb.x;
                o.y = aa.y * (v[it * 4 + 1] - mean) * rinv + bb.y;
                o.z = aa.z * (v[it * 4 + 2] - mean) * rinv + bb.z;
                o.w = aa.w * (v[it * 4 + 3] - mean) * rinv + bb.w;
                st_elem4(y, yoff + c, y_dt, o);
            } else {
                st_elem(y, yoff + c, y_dt, a[c] * (v[it] - mean) * rinv + b[c]);
            }
        }
    }
}

// Backward.  With xc = x - mean, r = 1/(sd+eps), g = dy*a, n = d:
//   dx = r*(g - mean(g)) - r^2 * sum(g*xc) * xc / ((n-1)*sd)      [+ dres]
//   da += sum_rows dy*xc*r ; db += sum_rows dy
// Each workgroup walks LN_ROWS_PER_BLOCK rows (4 waves, round-robin); every lane owns fixed columns, so the
// per-column partial sums stay in registers and are combined through LDS + one atomicAdd per column per block.
constexpr int LN_ROWS_PER_BLOCK = 64;

template <bool VEC, int NREG>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ a, const float* __restrict__ stats,
                                                     const float* __restrict__ dres, float* __restrict__ dx,
                                                     float* __restrict__ da, float* __restrict__ db, int64_t rows, int d,
                                                     float eps, int rows_per_block, void* __restrict__ dz, int dz_dt,
                                                     float drop_p, uint32_t drop_seed, const int32_t* __restrict__ drop_rows) {
    using C = Cols<VEC, NREG>;
    extern __shared__ float red[];  // [4 waves][2][d] partial da / db
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pa[NREG], pb[NREG], av[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) { pa[i] = 0.f; pb[i] = 0.f; }
    load_row<VEC, NREG>(a, d, lane, av);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    // Software pipeline over the wave's rows: the loads of row k+1 (x, dy, the residual gradient, the statistics) are in
    // flight while row k is reduced and stored.  With one row per wave in flight the kernel ran at 2.9 TB/s (5 waves per CU,
    // ~4 KB each in flight: latency-bound); see DESIGN.md section 7.
    float xn[NREG], gn[NREG], rn[NREG];
    float mean_n = 0.f, sd_n = 1.f;
    auto fetch = [&](int64_t row) {
        load_row<VEC, NREG>(x + row * d, d, lane, xn);
        load_row<VEC, NREG>(dy + row * d, d, lane, gn);
        if (dres) load_row<VEC, NREG>(dres + row * d, d, lane, rn);
        mean_n = stats[row * 2]; sd_n = stats[row * 2 + 1];
    };
    if (r0 + wave < rows && wave < rows_per_block) fetch(r0 + wave);
    for (int rr = wave; rr < rows_per_block; rr += 4) {
        const int64_t row = r0 + rr;
        if (row >= rows) break;
        float xv[NREG], gv[NREG], rv[NREG];
#pragma unroll
        for (int i = 0; i < NREG; ++i) { xv[i] = xn[i]; gv[i] = gn[i]; rv[i] = dres ? rn[i] : 0.f; }
        const float mean = mean_n, sd = sd_n;
        if (rr + 4 < rows_per_block && row + 4 < rows) fetch(row + 4);
        const float r = 1.f / (sd + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int it = 0; it < C::NIT; ++it)
#pragma unroll
            for (int u = 0; u < C::W; ++u) {
                const int i = it * C::W + u;
                const bool in = C::col(lane, it) + u < d;
                const float xc = in ? xv[i] - mean : 0.f, dyv = gv[i];
                pa[i] += dyv * xc * r; pb[i] += dyv;
                const float g = dyv * av[i];
                xv[i] = xc; gv[i] = g; sg += g; sgx += g * xc;
            }
        sg = wave_sum(sg); sgx = wave_sum(sgx);
        const float mg = sg / (float)d;
        const float coef = r * r * sgx / ((float)(d - 1) * sd);
        float* dxr = dx + row * d;
        // (dz draws the dropout of row drop_rows[row]: the valid-position layout keyed like the padded one)
        const uint64_t krow = (dz && drop_rows) ? (uint64_t)drop_rows[row] : (uint64_t)row;
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) {
            const int c = C::col(lane, it);
            if (c < d) {
                if (VEC) {
                    float4 o;
                    o.x = r * (gv[it * 4] - mg) - coef * xv[it * 4];
                    o.y = r * (gv[it * 4 + 1] - mg) - coef * xv[it * 4 + 1];
                    o.z = r * (gv[it * 4 + 2] - mg) - coef * xv[it * 4 + 2];
                    o.w = r * (gv[it * 4 + 3] - mg) - coef * xv[it * 4 + 3];
                    o.x += rv[it * 4]; o.y += rv[it * 4 + 1]; o.z += rv[it * 4 + 2]; o.w += rv[it * 4 + 3];
                    *reinterpret_cast<float4*>(dxr + c) = o;
                    if (dz) {   // second output: the dropout-masked (and possibly bf16) copy the next out-projection's GEMMs read
                        const uint64_t i0 = (uint64_t)row * (uint64_t)d + (uint64_t)c;
                        const float ik = 1.f / (1.f - drop_p);
                        float4 z;
                        bool kp[4] = {true, true, true, true};
                        if (drop_p > 0.f) ortk_keep4(drop_seed, krow * (uint64_t)d + (uint64_t)c, drop_p, kp);
                        z.x = kp[0] ? o.x * ik : 0.f;
                        z.y = kp[1] ? o.y * ik : 0.f;
                        z.z = kp[2] ? o.z * ik : 0.f;
                        z.w = kp[3] ? o.w * ik : 0.f;
                        st_elem4(dz, (int64_t)i0, dz_dt, z);
                    }
                } else {
                    float o = r * (gv[it] - mg) - coef * xv[it];
                    o += rv[it];
                    dxr[c] = o;
                    if (dz) {
                        const uint64_t i0 = (uint64_t)row * (uint64_t)d + (uint64_t)c;
                        st_elem(dz, (int64_t)i0, dz_dt, (drop_p > 0.f && !ortk_keep(drop_seed, krow * (uint64_t)d + (uint64_t)c, drop_p)) ? 0.f : o * (1.f / (1.f - drop_p)));
                    }
                }
            }
        }
    }
    // combine the 4 waves' column partials: plain stores to per-wave LDS rows, then each thread sums 4 values per column
    // (LDS float atomics were the slow part of this tail)
    __syncthreads();
#pragma unroll
    for (int it = 0; it < C::NIT; ++it)
#pragma unroll
        for (int u = 0; u < C::W; ++u) {
            const int c = C::col(lane, it) + u;
            if (c < d) { red[wave * 2 * d + c] = pa[it * C::W + u]; red[wave * 2 * d + d + c] = pb[it * C::W + u]; }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += 256) {
        const float v = (red[c] + red[2 * d + c]) + (red[4 * d + c] + red[6 * d + c]);
        atomicAdd(c < d ? &da[c] : &db[c - d], v);
    }
}

// d = 512 (the model width), vector-aligned rows: lane l owns the EIGHT consecutive columns 8 l .. 8 l + 7, so that the masked bf16 copy
// dz leaves as one 16-byte store per lane (the form above stores bf16x4 = 8 bytes: half the rate per byte, MI355X_MICROARCH.md) and a
// bf16 dy (DYT = __bf16: the data-gradient GEMM's output stored as bf16) arrives as one 16-byte load.  Same arithmetic, same pipeline.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_n;
template <typename DYT>
__global__ __launch_bounds__(256) void ln_bwd512_kernel(const DYT* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ a,
                                                        const float* __restrict__ stats, const float* __restrict__ dres, float* __restrict__ dx,
                                                        float* __restrict__ da, float* __restrict__ db, int64_t rows, float eps, int rows_per_block,
                                                        void* __restrict__ dz, int dz_dt, float drop_p, uint32_t drop_seed,
                                                        const int32_t* __restrict__ drop_rows) {
    constexpr int d = 512;
    extern __shared__ float red[];  // [4 waves][2][d] partial da / db
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c0 = 8 * lane;
    float pa[8], pb[8], av[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { pa[i] = 0.f; pb[i] = 0.f; }
    { const f32x4 t0 = *reinterpret_cast<const f32x4*>(a + c0), t1 = *reinterpret_cast<const f32x4*>(a + c0 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = t0[i]; av[4 + i] = t1[i]; } }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    f32x4 xn[2], rn[2]; f32x4 gn32[2]; u32x4_n gn16;
    float mean_n = 0.f, sd_n = 1.f;
    auto fetch = [&](int64_t row) {
        xn[0] = *reinterpret_cast<const f32x4*>(x + row * d + c0); xn[1] = *reinterpret_cast<const f32x4*>(x + row * d + c0 + 4);
        if (sizeof(DYT) == 4) {
            gn32[0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + row * d + c0);
            gn32[1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + row * d + c0 + 4);
        } else {
            gn16 = *reinterpret_cast<const u32x4_n*>(reinterpret_cast<const __bf16*>(dy) + row * d + c0);
        }
        if (dres) { rn[0] = *reinterpret_cast<const f32x4*>(dres + row * d + c0); rn[1] = *reinterpret_cast<const f32x4*>(dres + row * d + c0 + 4); }
        mean_n = stats[row * 2]; sd_n = stats[row * 2 + 1];
    };
    if (r0 + wave < rows && wave < rows_per_block) fetch(r0 + wave);
    for (int rr = wave; rr < rows_per_block; rr += 4) {
        const int64_t row = r0 + rr;
        if (row >= rows) break;
        float xv[8], gv[8], rv[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xv[i] = xn[0][i]; xv[4 + i] = xn[1][i];
            rv[i] = dres ? rn[0][i] : 0.f; rv[4 + i] = dres ? rn[1][i] : 0.f;
            if (sizeof(DYT) == 4) { gv[i] = gn32[0][i]; gv[4 + i] = gn32[1][i]; }
            else { gv[2 * i] = __uint_as_float(gn16[i] << 16); gv[2 * i + 1] = __uint_as_float(gn16[i] & 0xFFFF0000u); }
        }
        const float mean = mean_n, sd = sd_n;
        if (rr + 4 < rows_per_block && row + 4 < rows) fetch(row + 4);
        const float r = 1.f / (sd + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xc = xv[i] - mean, dyv = gv[i];
            pa[i] += dyv * xc * r; pb[i] += dyv;
            const float g = dyv * av[i];
            xv[i] = xc; gv[i] = g; sg += g; sgx += g * xc;
        }
        sg = wave_sum(sg); sgx = wave_sum(sgx);
        const float mg = sg / (float)d;
        const float coef = r * r * sgx / ((float)(d - 1) * sd);
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = r * (gv[i] - mg) - coef * xv[i] + rv[i];
        float* dxr = dx + row * d + c0;
        *reinterpret_cast<f32x4*>(dxr) = (f32x4){o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(dxr + 4) = (f32x4){o[4], o[5], o[6], o[7]};
        if (dz) {
            const uint64_t krow = drop_rows ? (uint64_t)drop_rows[row] : (uint64_t)row;
            const float ik = 1.f / (1.f - drop_p);
            float z[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bool kp[4] = {true, true, true, true};
                if (drop_p > 0.f) ortk_keep4(drop_seed, krow * (uint64_t)d + (uint64_t)(c0 + 4 * h), drop_p, kp);
#pragma unroll
                for (int q = 0; q < 4; ++q) z[4 * h + q] = kp[q] ? o[4 * h + q] * ik : 0.f;
            }
            const int64_t i0 = row * d + c0;
            if (dz_dt == ORTK_BF16) {
                const bf16x8 pk = {(__bf16)z[0], (__bf16)z[1], (__bf16)z[2], (__bf16)z[3], (__bf16)z[4], (__bf16)z[5], (__bf16)z[6], (__bf16)z[7]};
                *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(dz) + i0) = pk;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dz) + i0) = (f32x4){z[0], z[1], z[2], z[3]};
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dz) + i0 + 4) = (f32x4){z[4], z[5], z[6], z[7]};
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[wave * 2 * d + c0 + i] = pa[i]; red[wave * 2 * d + d + c0 + i] = pb[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += 256) {
        const float v = (red[c] + red[2 * d + c]) + (red[4 * d + c] + red[6 * d + c]);
        atomicAdd(c < d ? &da[c] : &db[c - d], v);
    }
}

inline bool al16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int ortk_layernorm_fwd(const float* x, const float* a, const float* b, void* y, int32_t y_dtype, float* stats,
                                  int64_t rows, int32_t d, float eps, ortk_stream stream) {
    if (!x || !a || !b || !y || d < 2 || d > 2048 || rows < 0 || (y_dtype != ORTK_F32 && y_dtype != ORTK_BF16)) return ORTK_EINVAL;
    if (rows == 0) return 0;
    dim3 grid((unsigned)ortk_cdiv(rows, 4)), block(256);
    const bool vec = d % 4 == 0 && al16(x) && al16(y) && al16(a) && al16(b);
#define LN_F(V, N) hipLaunchKernelGGL((ln_fwd_kernel<V, N>), grid, block, 0, ortk_s(stream), x, a, b, y, (int)y_dtype, stats, rows, d, eps)
    if (d <= 512) { if (vec) LN_F(true, 8); else LN_F(false, 8); }
    else          { if (vec) LN_F(true, 32); else LN_F(false, 32); }
#undef LN_F
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_layernorm_bwd_drop(const float* dy, const float* x, const float* a, const float* stats, const float* dres,
                                       float* dx, float* da, float* db, int64_t rows, int32_t d, float eps, void* dz, int32_t dz_dtype,
                                       float drop_p, uint32_t drop_seed, ortk_stream stream) {
    return ortk_layernorm_bwd_drop_rows(dy, x, a, stats, dres, dx, da, db, rows, d, eps, dz, dz_dtype, drop_p, drop_seed, nullptr, stream);
}

extern "C" int ortk_layernorm_bwd_drop_rows(const float* dy, const float* x, const float* a, const float* stats, const float* dres,
                                            float* dx, float* da, float* db, int64_t rows, int32_t d, float eps, void* dz, int32_t dz_dtype,
                                            float drop_p, uint32_t drop_seed, const int32_t* drop_rows, ortk_stream stream) {
    return ortk_layernorm_bwd_dt(dy, ORTK_F32, x, a, stats, dres, dx, da, db, rows, d, eps, dz, dz_dtype, drop_p, drop_seed, drop_rows, stream);
}

extern "C" int ortk_layernorm_bwd_dt(const void* dy_, int32_t dy_dtype, const float* x, const float* a, const float* stats, const float* dres,
                                     float* dx, float* da, float* db, int64_t rows, int32_t d, float eps, void* dz, int32_t dz_dtype,
                                     float drop_p, uint32_t drop_seed, const int32_t* drop_rows, ortk_stream stream) {
    const float* dy = reinterpret_cast<const float*>(dy_);
    if (dy_dtype != ORTK_F32 && dy_dtype != ORTK_BF16) return ORTK_EINVAL;
    if (!dy || !x || !a || !stats || !dx || !da || !db || d < 2 || d > 2048 || rows < 0) return ORTK_EINVAL;
    if (dz && ((dz_dtype != ORTK_F32 && dz_dtype != ORTK_BF16) || drop_p < 0.f || drop_p >= 1.f)) return ORTK_EINVAL;
    if (rows == 0) return 0;
    // measured (scratch/ln_bench.py): 64 and 32 rows tie at 21 760 rows (47 us); 32 wins at 9 216 (31.6 vs 35.9 us);
    // 16 rows and fewer lose to the per-block atomics on da / db
    const int rpb = rows >= 16384 ? LN_ROWS_PER_BLOCK : LN_ROWS_PER_BLOCK / 2;
    dim3 grid((unsigned)ortk_cdiv(rows, rpb)), block(256);
    const size_t shm = 8 * (size_t)d * sizeof(float);
    const bool vec = d % 4 == 0 && al16(dy) && al16(x) && al16(a) && al16(dres) && al16(dx) && (dz == nullptr || (reinterpret_cast<uintptr_t>(dz) & 15) == 0);
    if (d == 512 && vec && (dy_dtype == ORTK_BF16 || !(ortk::tuning().ln_fuse & 4))) {       // the model width: eight consecutive columns per lane (16-byte bf16 accesses; ln_fuse & 4: the four-column form, measurement)
        if (dy_dtype == ORTK_BF16) hipLaunchKernelGGL(ln_bwd512_kernel<__bf16>, grid, block, shm, ortk_s(stream), reinterpret_cast<const __bf16*>(dy_), x, a, stats, dres, dx, da, db, rows, eps, rpb, dz, (int)dz_dtype, drop_p, drop_seed, drop_rows);
        else hipLaunchKernelGGL(ln_bwd512_kernel<float>, grid, block, shm, ortk_s(stream), dy, x, a, stats, dres, dx, da, db, rows, eps, rpb, dz, (int)dz_dtype, drop_p, drop_seed, drop_rows);
        ORTK_CHECK_LAUNCH();
        return 0;
    }
    if (dy_dtype != ORTK_F32) return ORTK_EINVAL;        // (bf16 output gradients: d = 512 only)
#define LN_B(V, N) hipLaunchKernelGGL((ln_bwd_kernel<V, N>), grid, block, shm, ortk_s(stream), dy, x, a, stats, dres, dx, da, db, rows, d, eps, rpb, dz, (int)dz_dtype, drop_p, drop_seed, drop_rows)
    if (d <= 512) { if (vec) LN_B(true, 8); else LN_B(false, 8); }
    else          { if (vec) LN_B(true, 32); else LN_B(false, 32); }
#undef LN_B
    ORTK_CHECK_LAUNCH();
    return 0;
}

extern "C" int ortk_layernorm_bwd(const float* dy, const float* x, const float* a, const float* stats, const float* dres,
                                  float* dx, float* da, float* db, int64_t rows, int32_t d, float eps, ortk_stream stream) {
    return ortk_layernorm_bwd_drop(dy, x, a, stats, dres, dx, da, db, rows, d, eps, nullptr, 0, 0.f, 0, stream);
}

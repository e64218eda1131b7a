// sddmm_probe.hip — how fast can the weight gradient of a BINARY-masked linear be computed at its non-zeros only?
//   dW[n][k] = sum_m dY[m][n] X[m][k]   for the (n, k) of the mask            (pruning/prune.py:296-433: magnitude / SNIP /
//   lottery fine-tuning needs no gradient at pruned positions; supermask training does: sampler.py:10-34)
// The kernel below is the fastest form we could write for the VALU: a workgroup stages MB rows of dY and X TRANSPOSED in LDS
// ([column][m], bf16: 8 consecutive m = one 16-byte read), every lane owns non-zeros and accumulates 8 multiply-adds per pair of
// ds_read_b128 with four v_dot2_f32_bf16; the M dimension is split over workgroups (fp32 atomics into the compact gradient).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 scratch/sddmm/sddmm_probe.hip -o /tmp/sddmm && /tmp/sddmm
// It prints the time of one (N x K) gradient over M rows at 95 % zeros and its dense-equivalent TFLOP/s; the dense MFMA
// weight-gradient GEMM of the library on the same shape is timed by scratch/sddmm/dense_wgrad.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <algorithm>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
constexpr int MB = 64;       // rows per workgroup step (2 x 512 columns x 64 rows x 2 B = 128 KB of LDS)
constexpr int NC = 512;      // columns of dY and of X (the path's 512 x 512 projections)
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
// dYt / Xt: [NC][M] bf16 (already transposed in memory: the best case for the staging); nz: (n << 16 | k) per non-zero
__global__ __launch_bounds__(512) void sddmm_kernel(const __bf16* __restrict__ dYt, const __bf16* __restrict__ Xt, const unsigned* __restrict__ nz,
                                                    int nnz, float* __restrict__ out, int M, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* sY = reinterpret_cast<__bf16*>(smem);                 // [NC][MB + 8]  (+8: rows 16 bytes apart in the bank row)
    __bf16* sX = sY + NC * (MB + 8);
    const int m_begin = blockIdx.x * rows_per_wg, tid = threadIdx.x;
    const int per = (nnz + 511) / 512;                            // non-zeros per lane (strided)
    for (int m0 = m_begin; m0 < m_begin + rows_per_wg && m0 < M; m0 += MB) {
        __syncthreads();
        for (int idx = tid; idx < NC * (MB / 8); idx += 512) {
            const int c = idx / (MB / 8), q = idx % (MB / 8);
            *reinterpret_cast<uint4*>(sY + c * (MB + 8) + 8 * q) = *reinterpret_cast<const uint4*>(dYt + (size_t)c * M + m0 + 8 * q);
            *reinterpret_cast<uint4*>(sX + c * (MB + 8) + 8 * q) = *reinterpret_cast<const uint4*>(Xt + (size_t)c * M + m0 + 8 * q);
        }
        __syncthreads();
        for (int i = 0; i < per; ++i) {
            const int e = tid + 512 * i;
            if (e >= nnz) break;
            const unsigned code = nz[e];
            const __bf16* py = sY + (code >> 16) * (MB + 8);
            const __bf16* px = sX + (code & 0xFFFF) * (MB + 8);
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < MB / 8; ++q) {
                const uint4 a = *reinterpret_cast<const uint4*>(py + 8 * q), b = *reinterpret_cast<const uint4*>(px + 8 * q);
                acc = dot2(a.x, b.x, acc); acc = dot2(a.y, b.y, acc); acc = dot2(a.z, b.z, acc); acc = dot2(a.w, b.w, acc);
            }
            atomicAdd(out + e, acc);
        }
    }
}
int main() {
    const int M = 21760, nnz = (int)(NC * NC * 0.05);
    std::vector<uint16_t> hy((size_t)NC * M), hx((size_t)NC * M);
    srand(1);
    for (auto& v : hy) v = (uint16_t)(0x3C00 + (rand() & 0x1FF));   // bf16 values around 0.01 - 0.03
    for (auto& v : hx) v = (uint16_t)(0x3C00 + (rand() & 0x1FF));
    std::vector<unsigned> hnz(nnz);
    for (auto& v : hnz) v = ((unsigned)(rand() % NC) << 16) | (unsigned)(rand() % NC);
    __bf16 *dY, *dX; unsigned* dnz; float* dout;
    hipMalloc(&dY, hy.size() * 2); hipMalloc(&dX, hx.size() * 2); hipMalloc(&dnz, nnz * 4); hipMalloc(&dout, nnz * 4);
    hipMemcpy(dY, hy.data(), hy.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dX, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dnz, hnz.data(), nnz * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)2 * NC * (MB + 8) * 2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(sddmm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int wgs : {85, 170, 256, 340}) {
        const int rows_per_wg = ((M + wgs - 1) / wgs + MB - 1) / MB * MB;
        const int grid = (M + rows_per_wg - 1) / rows_per_wg;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int it = 0; it < 3; ++it) { hipMemsetAsync(dout, 0, nnz * 4, 0); hipLaunchKernelGGL(sddmm_kernel, dim3(grid), dim3(512), lds, 0, dY, dX, dnz, nnz, dout, M, rows_per_wg); }
        hipEventRecord(a, 0);
        const int reps = 20;
        for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(sddmm_kernel, dim3(grid), dim3(512), lds, 0, dY, dX, dnz, nnz, dout, M, rows_per_wg);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
        printf("SDDMM  %d x %d over M = %d, 95 %% zeros (%d non-zeros), %3d workgroups x %d rows: %7.1f us = %6.1f GMAC/s at the non-zeros = %6.1f TFLOP/s dense-equivalent\n",
               NC, NC, M, nnz, grid, rows_per_wg, ms * 1e3, (double)nnz * M / (ms * 1e-3) / 1e9, 2.0 * NC * NC * M / (ms * 1e-3) / 1e12);
    }
    // spot check of a few outputs against the host
    std::vector<float> ho(nnz); hipMemsetAsync(dout, 0, nnz * 4, 0);
    hipLaunchKernelGGL(sddmm_kernel, dim3(256), dim3(512), lds, 0, dY, dX, dnz, nnz, dout, M, ((M + 255) / 256 + MB - 1) / MB * MB);
    hipMemcpy(ho.data(), dout, nnz * 4, hipMemcpyDeviceToHost);
    auto bf = [](uint16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; };
    double worst = 0;
    for (int e = 0; e < 5; ++e) {
        const int n = hnz[e] >> 16, k = hnz[e] & 0xFFFF; double s = 0;
        for (int m = 0; m < M; ++m) s += (double)bf(hy[(size_t)n * M + m]) * bf(hx[(size_t)k * M + m]);
        worst = std::max(worst, std::abs(s - ho[e]) / std::abs(s));
    }
    printf("spot check: worst relative error %.2e\n", worst);
    return 0;
}

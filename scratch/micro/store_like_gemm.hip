// The dma256 GEMM's epilogue store pattern alone: 256 x 256 tiles, 512 threads (8 waves as 2 x 4, each 128 rows x 64 columns),
// with and without the 128 KB of LDS that holds the kernel to one workgroup per compute unit.
// hipcc --offload-arch=gfx950 -O3 scratch/micro/store_like_gemm.hip -o scratch/micro/store_like_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
// MODE 0: fp32 16 B | 1: bf16 8 B | 2: bf16 16 B (paired columns) ; REMAP: XCD-aware tile order
template <int MODE, bool REMAP> __global__ __launch_bounds__(512) void k(void* out, int M, int N, float v) {
    extern __shared__ char lds[];
    const int tn = N / 256;
    const int bid = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int bm = bid / tn, bn = bid % tn;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, wm = w >> 2, wn = w & 3;
    const int r = l & 15, g = l >> 4;
    if (v == 777.f) lds[threadIdx.x] = 1;
    for (int i = 0; i < 8; ++i) {
        const int row = bm * 256 + wm * 128 + i * 16 + r;
        if (row >= M) continue;
        const size_t base = (size_t)row * N + bn * 256 + wn * 64;
        if (MODE == 0) { for (int j = 0; j < 4; ++j) { float4 x = {v, v + 1, v + 2, v + 3}; *reinterpret_cast<float4*>((float*)out + base + j * 16 + g * 4) = x; } }
        else if (MODE == 1) { for (int j = 0; j < 4; ++j) { uint2 x = {__float_as_uint(v), __float_as_uint(v + j)}; *reinterpret_cast<uint2*>((uint16_t*)out + base + j * 16 + g * 4) = x; } }
        else { for (int j = 0; j < 2; ++j) { uint4 x = {__float_as_uint(v), __float_as_uint(v + j), __float_as_uint(v), __float_as_uint(v)}; *reinterpret_cast<uint4*>((uint16_t*)out + base + j * 32 + g * 8) = x; } }
    }
}
template <int MODE, bool REMAP> float run(void* buf, int M, int N, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, REMAP>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int grid = ((M + 255) / 256) * (N / 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) k<MODE, REMAP><<<grid, 512, lds>>>(buf, M, N, 1.f);
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) k<MODE, REMAP><<<grid, 512, lds>>>(buf, M, N, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1); return t * 1e3f / 50;
}
int main() {
    const int M = 16640;
    void* buf; hipMalloc(&buf, (size_t)M * 2048 * 4);
    for (int N : {512, 2048})
        for (size_t lds : {(size_t)0, (size_t)131072}) {
            printf("M %d N %d LDS %6zu:  remap: fp32 %.1f  bf16 8-B %.1f  bf16 16-B %.1f | plain order: fp32 %.1f  bf16 8-B %.1f  bf16 16-B %.1f us\n", M, N, lds,
                   run<0, true>(buf, M, N, lds), run<1, true>(buf, M, N, lds), run<2, true>(buf, M, N, lds),
                   run<0, false>(buf, M, N, lds), run<1, false>(buf, M, N, lds), run<2, false>(buf, M, N, lds));
        }
    return 0;
}

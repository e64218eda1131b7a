// Epilogue store experiment: 16640 x N tile writes by 256-thread workgroups of 128x128 tiles, as the GEMM epilogue
// issues them (a lane owns 4 consecutive columns of one row per 16x16 block) against a lane owning 8 consecutive columns.
// hipcc --offload-arch=gfx950 -O3 scratch/micro/store_width.hip -o scratch/micro/store_width
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdint>
template <int MODE> __global__ __launch_bounds__(256) void k(void* out, int M, int N, float v) {
    const int tn = N / 128, bm = blockIdx.x / tn, bn = blockIdx.x % tn;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, wm = w >> 1, wn = w & 1;   // wave: 64 rows x 64 columns
    const int r = l & 15, g = l >> 4;
    for (int i = 0; i < 4; ++i) {          // 4 row blocks of 16
        const int row = bm * 128 + wm * 64 + i * 16 + r;
        if (row >= M) continue;
        if (MODE == 0) {                   // fp32, 16 B per lane, 4 column blocks
            for (int j = 0; j < 4; ++j) { float4 x = {v, v + 1, v + 2, v + 3}; *reinterpret_cast<float4*>((float*)out + (size_t)row * N + bn * 128 + wn * 64 + j * 16 + g * 4) = x; }
        } else if (MODE == 1) {            // bf16, 8 B per lane, 4 column blocks
            for (int j = 0; j < 4; ++j) { uint2 x = {__float_as_uint(v), __float_as_uint(v + j)}; *reinterpret_cast<uint2*>((uint16_t*)out + (size_t)row * N + bn * 128 + wn * 64 + j * 16 + g * 4) = x; }
        } else {                           // bf16, 16 B per lane, 2 column blocks of 32
            for (int j = 0; j < 2; ++j) { uint4 x = {__float_as_uint(v), __float_as_uint(v + j), __float_as_uint(v), __float_as_uint(v)}; *reinterpret_cast<uint4*>((uint16_t*)out + (size_t)row * N + bn * 128 + wn * 64 + j * 32 + g * 8) = x; }
        }
    }
}
int main() {
    const int M = 16640;
    void* buf; hipMalloc(&buf, (size_t)M * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int N : {512, 2048}) {
        const int grid = ((M + 127) / 128) * (N / 128);
        float t[3];
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&] { if (mode == 0) k<0><<<grid, 256>>>(buf, M, N, 1.f); else if (mode == 1) k<1><<<grid, 256>>>(buf, M, N, 1.f); else k<2><<<grid, 256>>>(buf, M, N, 1.f); };
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0); for (int i = 0; i < 50; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&t[mode], e0, e1); t[mode] *= 1e3f / 50;
        }
        printf("M %d N %d: fp32 16-B stores %.1f us | bf16 8-B stores %.1f us | bf16 16-B stores %.1f us\n", M, N, t[0], t[1], t[2]);
    }
    return 0;
}

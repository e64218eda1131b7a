// fetch_rate.hip — how many bytes per clock does ONE compute unit pull through its vector-memory path, as a function of the
// load form and of the bytes it keeps in flight?  One 512-thread workgroup per CU (grid = 256), every wave streams its own
// contiguous slice.   Forms:  V = global_load_dwordx4 into VGPRs (U loads in flight per lane);  D = global_load_lds_dwordx4
// (LDS-DMA, U 1-KB pieces in flight per wave).   Sources:  "own" = every workgroup reads a distinct 8 MB region (HBM / MALL),
// "same" = all workgroups read the same 512 KB (an L2-resident weight matrix).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 scratch/fetch_rate/fetch_rate.hip -o /tmp/fr && /tmp/fr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

template <int U>
__global__ __launch_bounds__(512, 1) void stream_vgpr(const u32x4* __restrict__ src, size_t wg_stride16, int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* p = src + (size_t)blockIdx.x * wg_stride16 + (size_t)wave * iters * U * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = p[u * 64];
    for (int it = 1; it < iters; ++it) {
        p += U * 64;
#pragma unroll
        for (int u = 0; u < U; ++u) { acc ^= r[u]; r[u] = p[u * 64]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= r[u];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int U>
__global__ __launch_bounds__(512, 1) void stream_dma(const u32x4* __restrict__ src, size_t wg_stride16, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* p = src + (size_t)blockIdx.x * wg_stride16 + (size_t)wave * iters * U * 64 + lane;
    char* mine = smem + wave * (2 * U * 1024);          // two groups of U pieces per wave
    for (int u = 0; u < U; ++u) __builtin_amdgcn_global_load_lds((glb_void*)(p + u * 64), (lds_void*)(mine + u * 1024), 16, 0, 0);
    for (int it = 1; it < iters; ++it) {
        p += U * 64;
        char* dst = mine + (it & 1) * U * 1024;
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_global_load_lds((glb_void*)(p + u * 64), (lds_void*)(dst + u * 1024), 16, 0, 0);
        // retire the previous group: U newer pieces may stay in flight
        if (U == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (U == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (U == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (U == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (U == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (reinterpret_cast<unsigned*>(smem)[threadIdx.x] == 0x12345678u) sink[0] = 1;
}

template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int NWG = 256;
    const size_t per_wg = (size_t)8 << 20;                 // 8 MB per workgroup
    u32x4* buf; unsigned* sink;
    hipMalloc(&buf, per_wg * NWG); hipMemset(buf, 1, per_wg * NWG); hipMalloc(&sink, 4);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock %d MHz\n", clk_khz / 1000);
    for (int same = 0; same < 2; ++same) {
        const size_t bytes_wg = same ? (size_t)512 << 10 : per_wg;     // bytes one workgroup streams per launch
        const size_t stride16 = same ? 0 : per_wg / 16;
#define RUN(KERN, U, NAME, LDS) { \
            const int iters = (int)(bytes_wg / 8 / (U * 1024)); \
            hipFuncSetAttribute(reinterpret_cast<const void*>(KERN<U>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
            float ms = time_ms([&] { hipLaunchKernelGGL(KERN<U>, dim3(NWG), dim3(512), LDS, 0, buf, stride16, iters, sink); }, 20); \
            double bpc = (double)bytes_wg / (ms * 1e-3) / (clk_khz * 1e3); \
            printf("%-4s %s U=%2d (%3d KB in flight per CU): %7.1f us per launch, %6.1f GB/s per CU = %5.1f B/clk, chip %5.2f TB/s\n", \
                   same ? "same" : "own", NAME, U, U * 8, ms * 1e3, bytes_wg / (ms * 1e-3) / 1e9, bpc, bytes_wg * NWG / (ms * 1e-3) / 1e12); }
        RUN(stream_vgpr, 1, "VGPR", 0) RUN(stream_vgpr, 2, "VGPR", 0) RUN(stream_vgpr, 4, "VGPR", 0) RUN(stream_vgpr, 8, "VGPR", 0) RUN(stream_vgpr, 16, "VGPR", 0)
        RUN(stream_dma, 1, "DMA ", 8 * 2 * 1 * 1024) RUN(stream_dma, 2, "DMA ", 8 * 2 * 2 * 1024) RUN(stream_dma, 4, "DMA ", 8 * 2 * 4 * 1024)
        RUN(stream_dma, 8, "DMA ", 8 * 2 * 8 * 1024)
    }
    return 0;
}

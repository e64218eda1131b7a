// xchg_probe.hip — what does an all-gather among the G workgroups of a group cost on one MI355X?  (The column-split decoder stack
// kernel of DESIGN.md section 7c needs 9 of them per layer.)  Every workgroup (512 threads) writes its [64 rows x 512/G columns]
// slice (bf16 or fp32) of a [64 x 512] tile into a global buffer, publishes a counter (release, agent scope), waits until all G
// members of its group have (acquire), and reads the whole tile back.  Group members sit on ONE XCD (block ids x + 8 k) or are
// spread over XCDs (consecutive ids), `iters` exchanges back to back.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 scratch/xchg_probe/xchg_probe.hip -o /tmp/xp && /tmp/xp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// LIGHT: no agent-scope fences (no L2 write-back / invalidate): stores drain to the XCD's L2 (the vector L1 is write-through), the flag
// is a relaxed L2 atomic, the tile is read back with sc1 loads (L1 bypassed).  Only valid when the members share one XCD's L2.
__device__ __forceinline__ u32x4 load_sc1(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int G, int ESZ, bool LIGHT>      // ESZ: bytes per element (2 = bf16 tile 64 KB, 4 = fp32 tile 128 KB)
__global__ __launch_bounds__(512, 1) void xchg_kernel(u32x4* __restrict__ xbuf, int* __restrict__ flags, int iters, int same_xcd, int ngroups,
                                                      unsigned* sink) {
    const int tid = threadIdx.x;
    int grp, c;
    if (same_xcd) { const int x = blockIdx.x & 7, k = blockIdx.x >> 3; grp = x * (ngroups / 8) + k / G; c = k % G; }
    else { grp = blockIdx.x / G; c = blockIdx.x % G; }
    constexpr int TILE16 = 64 * 512 * ESZ / 16;          // uint4 per tile
    constexpr int SL16 = TILE16 / G;                     // uint4 per slice
    u32x4* buf = xbuf + (size_t)grp * 2 * TILE16;
    int* flag = flags + grp * 32;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4* dst = buf + (it & 1) * TILE16 + c * SL16;
        for (int i = tid; i < SL16; i += 512) dst[i] = (u32x4){(unsigned)it, (unsigned)c, (unsigned)i, 1u};
        __syncthreads();                                   // all of this workgroup's stores issued
        const u32x4* src = buf + (it & 1) * TILE16;
        if (LIGHT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G * (it + 1)) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            constexpr int NL = TILE16 / 512;
            u32x4 v[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) v[u] = load_sc1(src + tid + 512 * u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int i = tid + 512 * u;
                if (v[u][0] != (unsigned)it || v[u][2] != (unsigned)(i % SL16) || v[u][1] != (unsigned)(i / SL16)) atomicAdd(sink + 1, 1u);   // stale data
                acc ^= v[u];
            }
        } else {
            if (tid == 0) {
                __atomic_thread_fence(__ATOMIC_RELEASE);      // (HIP: agent scope by default for __atomic_thread_fence)
                __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < G * (it + 1)) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int i = tid; i < TILE16; i += 512) { const u32x4 v = __builtin_nontemporal_load(src + i); acc ^= v; }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int G, int ESZ, bool LIGHT> static void run(int ngroups, int same_xcd, u32x4* xbuf, int* flags, unsigned* sink) {
    const int iters = 2000;
    hipMemset(flags, 0, 4096 * 32 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((xchg_kernel<G, ESZ, LIGHT>), dim3(ngroups * G), dim3(512), 0, 0, xbuf, flags, 10, same_xcd, ngroups, sink);
    hipDeviceSynchronize();
    hipMemset(flags, 0, 4096 * 32 * 4);
    hipEventRecord(a);
    hipLaunchKernelGGL((xchg_kernel<G, ESZ, LIGHT>), dim3(ngroups * G), dim3(512), 0, 0, xbuf, flags, iters, same_xcd, ngroups, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned h[2] = {0, 0}; hipMemcpy(h, sink, 8, hipMemcpyDeviceToHost);
    printf("%s G=%d %s tile, %3d groups (%3d workgroups), members %s: %6.2f us per exchange, stale reads %u\n", LIGHT ? "light " : "fenced", G,
           ESZ == 2 ? "bf16 64-KB" : "fp32 128-KB", ngroups, ngroups * G, same_xcd ? "on one XCD " : "across XCDs", ms * 1e3 / iters, h[1]);
    hipMemset(sink, 0, 8);
}

int main() {
    u32x4* xbuf; int* flags; unsigned* sink;
    hipMalloc(&xbuf, (size_t)256 * 2 * 64 * 512 * 4); hipMalloc(&flags, 4096 * 32 * 4); hipMalloc(&sink, 8); hipMemset(sink, 0, 8);
    for (int same = 1; same >= 0; --same) {
        run<2, 2, false>(80, same, xbuf, flags, sink); run<2, 4, false>(80, same, xbuf, flags, sink);
        run<4, 2, false>(48, same, xbuf, flags, sink); run<8, 2, false>(24, same, xbuf, flags, sink);
    }
    run<2, 2, true>(80, 1, xbuf, flags, sink); run<2, 4, true>(80, 1, xbuf, flags, sink);
    run<4, 2, true>(48, 1, xbuf, flags, sink); run<4, 4, true>(48, 1, xbuf, flags, sink);
    run<8, 2, true>(24, 1, xbuf, flags, sink); run<8, 4, true>(24, 1, xbuf, flags, sink);
    run<2, 2, true>(80, 0, xbuf, flags, sink);       // (across XCDs the light protocol is NOT valid: shows the stale reads)
    return 0;
}

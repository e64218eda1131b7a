// VALU cost of the bf16/fp16 multiply-accumulate idioms considered for the sparse product (one launch per idiom,
// 256 CUs x 8 waves, long dependent-free loops; prints cycles per wave-instruction group and G MAC-lanes/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
#define N_ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out) {
    unsigned x0 = in[threadIdx.x], x1 = in[threadIdx.x + 256], w = in[threadIdx.x + 512];
    float val = __uint_as_float(w & 0xFFFF0000u);
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = (float)i;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            unsigned xa = x0 + d * 0x00010001u + it, xb = x1 ^ (d * 0x01000100u);
            if (MODE == 0) {            // perm + dot2c: 2 MACs per (perm, dot2) pair
                unsigned lo = __builtin_amdgcn_perm(xb, xa, 0x05040100u), hi = __builtin_amdgcn_perm(xb, xa, 0x07060302u);
                a[2 * d] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, lo), __builtin_bit_cast(bf2, w), a[2 * d], false);
                a[2 * d + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, hi), __builtin_bit_cast(bf2, w), a[2 * d + 1], false);
            } else if (MODE == 1) {     // shift / and + fma: 1 MAC per (cvt, fma)
                a[2 * d] = fmaf(__uint_as_float(xa << 16), val, a[2 * d]);
                a[2 * d + 1] = fmaf(__uint_as_float(xa & 0xFFFF0000u), val, a[2 * d + 1]);
            } else if (MODE == 2) {     // v_fma_mix_f32 on fp16 halves: 1 MAC per instruction
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[2 * d]) : "v"(xa), "v"(val));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[2 * d + 1]) : "v"(xa), "v"(val));
            } else if (MODE == 3) {     // plain fp32 fma (fp32 planes)
                a[2 * d] = fmaf(__uint_as_float(xa), val, a[2 * d]);
                a[2 * d + 1] = fmaf(__uint_as_float(xb), val, a[2 * d + 1]);
            } else {                    // dot2c alone (operands ready)
                a[2 * d] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, xa), __builtin_bit_cast(bf2, w), a[2 * d], false);
                a[2 * d + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, xb), __builtin_bit_cast(bf2, w), a[2 * d + 1], false);
            }
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double macs_per_iter, unsigned* in, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 2;          // 2 blocks of 4 waves per CU: 2 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lanes = (double)blocks * 256 * N_ITER * macs_per_iter;
    printf("%-28s %8.3f ms   %7.2f T MAC-lanes/s   (%.1f SIMD-cycles per 16-MAC group per wave at 2.4 GHz, 2 waves/SIMD)\n", name, ms, lanes / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (N_ITER * 2.0) / (macs_per_iter / 16.0));
}
int main() {
    unsigned* in; float* out; hipMalloc(&in, 4096); hipMalloc(&out, 256 * 2 * 256 * 4);
    std::vector<unsigned> h(1024, 0x3F803F80u); hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<0>("perm + dot2c (bf16 pairs)", 32, in, out);
    run<4>("dot2c alone", 32, in, out);
    run<1>("shift/and + fma (bf16)", 16, in, out);
    run<2>("v_fma_mix_f32 (fp16)", 16, in, out);
    run<3>("v_fma_f32 (fp32 planes)", 16, in, out);
    return 0;
}

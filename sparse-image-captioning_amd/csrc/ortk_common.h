// Shared device/host helpers for libortk (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ortk.h"

#define ORTK_WAVE 64

#define ORTK_CHECK_LAUNCH()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

static inline hipStream_t ortk_s(ortk_stream s) { return (hipStream_t)s; }

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// ---------------------------------------------------------------------------------------------
// Element types of activation buffers.  All arithmetic is fp32; in the mixed-precision mode (precision = 1) the
// tensors that only feed MFMA operands are STORED as bf16 (halves their HBM / L2 traffic); ORTK_F32 everywhere
// in the fp32 parity mode.
// ---------------------------------------------------------------------------------------------
enum : int { ORTK_F32 = 0, ORTK_BF16 = 1 };
static inline size_t ortk_esize(int dt) { return dt == ORTK_BF16 ? 2 : 4; }

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int DT = ORTK_F32; };
template <> struct Elem<__bf16> { static constexpr int DT = ORTK_BF16; };

__device__ __forceinline__ float ld_elem(const void* p, int64_t i, int dt) {
    return dt == ORTK_BF16 ? (float)reinterpret_cast<const __bf16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_elem(void* p, int64_t i, int dt, float v) {
    if (dt == ORTK_BF16) reinterpret_cast<__bf16*>(p)[i] = (__bf16)v; else reinterpret_cast<float*>(p)[i] = v;
}
// 4 consecutive elements (16-B / 8-B aligned) as float4
__device__ __forceinline__ float4 ld_elem4(const void* p, int64_t i, int dt) {
    if (dt == ORTK_BF16) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p) + i);
        return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ void st_elem4(void* p, int64_t i, int dt, float4 v) {
    if (dt == ORTK_BF16) {
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p) + i) = (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Counter-based RNG: one 32-bit mix per element.  The oracle carries the same function
// (oracle/ort_oracle.py: gumbel_from_hash) so that sampled decodes are reproducible token for token.
// ---------------------------------------------------------------------------------------------
__host__ __device__ static inline uint32_t ortk_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// uniform in (0,1) with 24 random bits
__host__ __device__ static inline float ortk_u01(uint32_t h) { return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// Dropout keep decisions.  Four consecutive elements (idx & ~3 .. +3) share ONE hash evaluation: 64 bits = four 16-bit
// fields, element idx keeps iff its field >= floor(p * 65536) (drop probability within 1.6e-5 of p).  A kernel that owns an
// aligned group of four (GEMM epilogues, LayerNorm backward, the bf16 attention kernels) calls ortk_keep4 — one 32-bit mix
// plus one multiply-xorshift round per four elements instead of one mix and an int->float conversion per element (the
// per-element form cost 25 us of the 100-us FFN up-projection); every other site evaluates the same function per element.
__device__ static inline uint2 ortk_keep_bits(uint32_t seed, uint64_t group) {
    const uint32_t h = ortk_mix32((uint32_t)group * 0x9E3779B1u + (uint32_t)(group >> 32) * 0x85EBCA77u + seed);
    uint32_t g = (h ^ 0x68E31DA4u) * 0xB5297A4Du;
    g ^= g >> 15;
    return make_uint2(h, g);
}
__device__ static inline uint32_t ortk_keep_thr(float p) { return (uint32_t)(p * 65536.f); }
__device__ static inline bool ortk_keep(uint32_t seed, uint64_t idx, float p) {
    const uint2 b = ortk_keep_bits(seed, idx >> 2);
    const uint32_t w = (idx & 2) ? b.y : b.x;
    return ((w >> ((idx & 1) * 16)) & 0xFFFFu) >= ortk_keep_thr(p);
}
// k[r] = ortk_keep(seed, idx0 + r, p), r = 0..3
__device__ static inline void ortk_keep4(uint32_t seed, uint64_t idx0, float p, bool (&k)[4]) {
    if ((idx0 & 3) == 0) {
        const uint2 b = ortk_keep_bits(seed, idx0 >> 2);
        const uint32_t thr = ortk_keep_thr(p);
        k[0] = (b.x & 0xFFFFu) >= thr; k[1] = (b.x >> 16) >= thr; k[2] = (b.y & 0xFFFFu) >= thr; k[3] = (b.y >> 16) >= thr;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) k[r] = ortk_keep(seed, idx0 + r, p);
    }
}
// The same decisions for element indices below 2^32 (the group's high word is zero), in 32-bit arithmetic, against a precomputed
// threshold thr = ortk_keep_thr(p): the column-split decoder stack kernel (<= 8 192 rows: every teacher-forced index fits).
__device__ static inline bool ortk_keep_u32(uint32_t seed, uint32_t idx, uint32_t thr) {
    const uint32_t h = ortk_mix32((idx >> 2) * 0x9E3779B1u + seed);
    uint32_t w = h;
    if (idx & 2) { w = (h ^ 0x68E31DA4u) * 0xB5297A4Du; w ^= w >> 15; }
    return ((w >> ((idx & 1) * 16)) & 0xFFFFu) >= thr;
}
// k[r] = ortk_keep(seed, idx0 + r, p), r = 0..3, idx0 a multiple of 4
__device__ static inline void ortk_keep4_u32(uint32_t seed, uint32_t idx0, uint32_t thr, bool (&k)[4]) {
    const uint32_t h = ortk_mix32((idx0 >> 2) * 0x9E3779B1u + seed);
    uint32_t g = (h ^ 0x68E31DA4u) * 0xB5297A4Du;
    g ^= g >> 15;
    k[0] = (h & 0xFFFFu) >= thr; k[1] = (h >> 16) >= thr; k[2] = (g & 0xFFFFu) >= thr; k[3] = (g >> 16) >= thr;
}
// Gumbel noise of (step t, hash row, token v): identical to oracle/ort_oracle.py: gumbel_from_hash.  FAST (mixed precision): v_log_f32 for
// the two logarithms — the same uniforms, the noise to ~1 ulp.
template <bool FAST>
__device__ __forceinline__ float ortk_gumbel(uint64_t seed, int t, int row, int v) {
    uint32_t x = (uint32_t)row * 0x9E3779B1u + (uint32_t)v * 0x85EBCA77u + (uint32_t)(t + 1) * 0xC2B2AE3Du + (uint32_t)seed * 0x27D4EB2Fu;
    const float u = ortk_u01(ortk_mix32(x));
    return FAST ? -__logf(-__logf(u)) : -logf(-logf(u));
}
// total order of (key, index) candidates: larger key first, lower index on a tie
__device__ __forceinline__ bool ortk_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }
__host__ static inline uint32_t ortk_subseed(uint64_t seed, uint32_t op) {
    return ortk_mix32((uint32_t)seed ^ ortk_mix32((uint32_t)(seed >> 32) + 0x632BE5ABu) ^ (op * 0x9E3779B1u + 0x7F4A7C15u));
}

// ---------------------------------------------------------------------------------------------
// wave-level reductions (64 lanes, DPP/shuffle based)
// ---------------------------------------------------------------------------------------------
__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ static inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int64_t ortk_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t ortk_align(int64_t a, int64_t b) { return ortk_cdiv(a, b) * b; }

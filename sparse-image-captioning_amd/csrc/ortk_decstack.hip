// ortk_decstack.hip — one decode position of the WHOLE decoder stack in one kernel (mixed precision, d_model 512, 8 heads).
//
// The cached decoder step of the reference (models/transformer.py:172-210, 240-273: per layer LayerNorm -> packed QKV ->
// self-attention over the cache -> output projection + residual -> LayerNorm -> query projection -> attention over the
// projected memory -> output projection + residual -> LayerNorm -> FFN + residual) is eleven dependent launches per layer
// in the unfused executor.  At decode time a launch has a few thousand rows at most, and a dependent launch costs ~8 us
// on the device whatever it does (profiles/r02_decode_launch_floor.txt): 10.5 of the 28.6 ms of a 1 024-image beam-5
// decode are that floor.  Here the rows stay put and the weights stream:
//
//   * a workgroup (8 waves) owns 32 consecutive rows for ALL layers.  The residual stream of its rows lives in registers
//     (MFMA accumulator layout: wave w holds columns 64w .. 64w+63 of the 32 rows), LayerNorm is a register pass with two
//     small LDS exchanges, and every projection is a [32 x 512] x [512 x 512] "unit" product whose A operand is a 32-KB bf16
//     image in LDS and whose B operand goes from global memory straight into the lane that feeds the MFMA — wave w needs
//     exactly the 64 weight rows of its output columns, nobody else does, so nothing is staged or shared;
//   * the weights are re-packed once per decode call (stack_pack_kernel) into that order: per wave ONE contiguous stream of
//     1-KB MFMA fragments over all units of all layers (QKV 3 units, O, CQ, CO, then W1 / W2 interleaved per 512 hidden
//     units), read through a 4-k-step register ring that keeps running across unit boundaries and LayerNorms;
//   * self-attention: the wave that owns a row reads its cache rows (beam ancestry table or fixed stride) as 1-KB rows,
//     lane = 8 features, two rows side by side, online soft-max over small key batches; this position's K / V come from LDS and are appended to
//     the cache on the way.  Cross-attention: the rows of ONE image share every K / V load (up to 5 rows per pass).  Register
//     budget rules both: with 8 waves per CU a wave has 256 VGPRs, 32 of them hold the residual rows throughout, and any batch
//     size that makes the compiler spill costs more than the extra loads in flight buy (measured; parking the residual rows
//     in global memory during the two phases to free their registers cost 0.5 ms per decode and bought nothing).
//
//   * 8 extra workgroups (one per XCD) run ahead of the compute workgroups and touch the weight stream into that XCD's L2,
//     paced by a counter the first compute workgroup of the XCD publishes.
//
// What bounds it: 14 units x 512 KB of weights per layer pass through every CU's vector-memory path — 3.4 us per unit at the
// nominal 64 B/clk, ~6 us measured (85 GB/s per CU); 160 workgroups for 1 024 images x 5 beams.  DESIGN.md section 4.
#include "ortk_internal.h"
#include <mutex>
#include <cstdlib>

namespace ortk {
namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int SD = 512;            // d_model
constexpr int SPD = 4;             // k-steps of weight fragments in flight per wave
// sparse stream: per wave a 4-KB dense fragment buffer (4 column tiles x 64 lanes x 16 B: one k-step) + 256 B of pad slots
constexpr int SDBUF = 4096 + 256;
constexpr int SCAP = 128;          // scatter entries per step (64 lanes x 2)
constexpr int XNR = 5;             // rows of one image served per cross-attention pass
// Attention batch sizes (-D overrides are for the sweeps of scratch/variant_sweep.sh): measured best, all within 0.2 ms of
// each other as long as the kernel does not spill; 8-key batches spill and cost 2 ms per decode.
#ifndef XKB_
#define XKB_ 4
#endif
#ifndef SKB_
#define SKB_ 2
#endif
#ifndef SROWS_
#define SROWS_ 2
#endif
constexpr int XKB = XKB_;          // keys per cross-attention batch
constexpr int SKB = SKB_;          // keys per self-attention batch
constexpr int SROWS = SROWS_;      // rows a wave serves side by side in the self-attention (1, 2 or 4)
constexpr int FRAG = 64;           // uint4 per fragment (1 KB)
constexpr int KSTEP = 4 * FRAG;    // uint4 per k-step of one wave (4 column tiles)
constexpr int STACK_AHEAD = 3;     // units (512 KB each) the L2 prefetcher may run in front of the pace-maker

// A images: [row][64 chunks of 16 B], physical chunk = chunk ^ (row & 15): the MFMA operand reads (16 rows x 4 k-groups per
// instruction) and the row-wise attention reads / writes (64 chunks of one row) are both conflict-free.
__device__ __forceinline__ int img_off(int row, int chunk) { return row * 1024 + ((chunk ^ (row & 15)) << 4); }

__device__ __forceinline__ float dot2(unsigned int a, unsigned int b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
__device__ __forceinline__ float dpp_quad1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_quad2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }
// sum over the 8 lanes of a head (lanes 8h .. 8h+7); every lane of the group gets the sum
__device__ __forceinline__ float sum8(float v) {
    v += dpp_quad1(v);
    v += dpp_quad2(v);
    v += dpp_half_mirror(v);
    return v;
}
__device__ __forceinline__ float lo_f(unsigned int x) { return __builtin_bit_cast(float, x << 16); }
__device__ __forceinline__ float hi_f(unsigned int x) { return __builtin_bit_cast(float, x & 0xFFFF0000u); }
__device__ __forceinline__ unsigned int pack2(float a, float b) {
    const bf16x2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, t);
}
__device__ __forceinline__ float rdlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// ------------------------------------------------------------------------------------------------ weight stream
struct Ring { uint4 f[SPD][4]; };

__device__ __forceinline__ void ring_start(Ring& r, const uint4* wp, int lane) {
#pragma unroll
    for (int s = 0; s < SPD; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) r.f[s][nt] = wp[s * KSTEP + nt * FRAG + lane];
}

// acc[mt][nt] += A[16 mt .. +15][:] . W[64 w + 16 nt .. +15][:]^T over the 512 inputs of one unit.  NEXT: keep the ring
// running into the unit that follows in the stream (false: the ring is dead afterwards — an attention phase needs the
// registers — and the next unit calls ring_start again).
template <bool NEXT>
__device__ __forceinline__ void unit_gemm(f32x4 (&acc)[2][4], const char* A, const uint4*& wp, Ring& r, int lane) {
    const int m = lane & 15, kg = lane >> 4;
    // four k-steps per iteration (= one turn of the ring); not unrolled further: the scheduler would hoist all 32 operand
    // reads of a unit to its top and spill the accumulators
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int s = 0; s < SPD; ++s) {
            const int ks = it * SPD + s;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(A + img_off(m, 4 * ks + kg));
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(A + img_off(16 + m, 4 * ks + kg));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, r.f[s][nt]);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a0, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a1, acc[1][nt], 0, 0, 0);
            }
            if (NEXT || it < 3) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) r.f[s][nt] = wp[(ks + SPD) * KSTEP + nt * FRAG + lane];
            }
        }
    }
    wp += 16 * KSTEP;
}


// ------------------------------------------------------------------------------------------------ sparse weight stream
// The same unit product when the decoder weights are mostly zeros (the reference evaluates pruned checkpoints as dense
// linears on zero-filled weights, scripts/eval_model.py:64-88).  The dense stream is bound by the bytes a CU can pull
// through its vector-memory path (512 KB per unit per workgroup, ~85 GB/s); here a wave pulls only the NON-ZEROS of its 64
// weight rows — (position, value) scatter entries, 4 bytes each — expands one k-step (4 column tiles x 64 lanes x 8 weights =
// 4 KB) at a time into a private LDS buffer with two ds_write_b16 per lane, reads the four dense MFMA fragments back with
// ds_read_b128 and multiplies on the matrix cores as before (same accumulation order: bit-identical to the dense stream
// unless a k-step overflows its step, below).  The buffer is kept all-zero between steps by writing zeros to the same
// positions once the fragments are in registers (two more ds_write_b16), so nothing is ever cleared wholesale.  LDS
// operations of one wave execute in order: [clear step i] [scatter step i+1] [read step i+1] are queued back to back behind
// [read step i] with no waits, and the MFMAs of step i run while they complete.
//
// Entry: bits 0-15 the bf16 weight, 16-27 its halfword index in the 4-KB fragment buffer ((tile * 64 + lane) * 8 + j; pad
// entries point at private slots 2048 + 2 lane + j behind it), 28-31 the k-step (32 input columns) of the whole step.  A
// step = 64 lanes x 2 entries; a k-step with more than 128 non-zeros in the wave's 64 rows (0.5 % of them at 95 % zeros:
// 102 +- 10) takes further steps with the same k — so the stream is correct at ANY density — and the steps of a unit are
// padded to a multiple of four (the depth of the register ring that prefetches them).  Inside a step the builder deals the
// entries to the four 32-lane groups of the two store instructions so that no group has more than two entries on one of the
// 32 LDS banks where that is possible (a 2-way store conflict is free: the store's data transfer takes as long).
//
// What bounds it (counters: profiles/r03_sparse_decode_stack_pmc_*): a wave issues at most one
// instruction per 4 cycles, the 8 MFMAs of a k-step take 128, and the LDS serves 8 waves: four 2-byte scatter / clear stores,
// four fragment reads and two operand reads per wave and k-step are ~45 LDS cycles x 8 waves against 256 of MFMA time.  Four
// other expansions were built and measured this round, all token-identical, all SLOWER than this one (1 024-image beam-5
// decode; dense stream 20.7 ms): bitmap + compacted 8-byte pieces fetched (a) by one buffer load per piece with out-of-range
// offsets for the zero lanes: 27.7 ms (~18 cycles of texture-addresser time per sparse wave-instruction), (b) through an LDS
// ring filled by LDS-DMA: 24.3 ms, (c) through a 16-deep register ring + one LDS staging slot, masks by s_load: 29.2 ms,
// (d) the same with the masks travelling in the step's own registers (v_readlane): 26.6 ms — the v_mbcnt / select / address
// arithmetic is ~6 vector + 4 scalar instructions per piece, 150 per k-step, 600 cycles of issue time per wave.
#ifndef SSPD_
#define SSPD_ 4
#endif
constexpr int SSPD = SSPD_;      // steps of entries in flight per wave (2 VGPRs each)
struct SRing { uint2 e[SSPD]; };
typedef const int __attribute__((address_space(4)))* cint_ptr;       // constant address space: uniform loads become s_load

__device__ __forceinline__ void sring_start(SRing& r, const uint2* sp, int lane) {
#pragma unroll
    for (int s = 0; s < SSPD; ++s) r.e[s] = sp[s * 64 + lane];
}
__device__ __forceinline__ void s_scatter(char* D, const uint2 e) {
    *reinterpret_cast<unsigned short*>(D + ((e.x >> 15) & 0x1FFE)) = (unsigned short)e.x;
    *reinterpret_cast<unsigned short*>(D + ((e.y >> 15) & 0x1FFE)) = (unsigned short)e.y;
}
__device__ __forceinline__ void s_clear(char* D, const uint2 e) {
    *reinterpret_cast<unsigned short*>(D + ((e.x >> 15) & 0x1FFE)) = 0;
    *reinterpret_cast<unsigned short*>(D + ((e.y >> 15) & 0x1FFE)) = 0;
}
__device__ __forceinline__ void s_frags(const char* D, uint4 (&f)[4], int lane) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) f[nt] = *reinterpret_cast<const uint4*>(D + nt * 1024 + lane * 16);
}
__device__ __forceinline__ int s_kstep(const uint2 e) { return __builtin_amdgcn_readfirstlane((int)(e.x >> 28)); }

// NEXT: the first step of the unit that follows in the stream is expanded and read into F[0] on the way out (false: the
// buffer is left clean and F dead — an attention phase needs the registers — and the next unit calls s_unit_cold first).
__device__ __forceinline__ void s_unit_cold(char* D, const SRing& E, uint4 (&F)[2][4], int lane) {
    s_scatter(D, E.e[0]);
    s_frags(D, F[0], lane);
}
// nstp = steps of the unit (a multiple of SSPD) | its REAL steps << 16: the all-pad steps that round a unit up to the ring's four
// (a k-step that overflows its 128 entries costs one real step more — 7 % of the wave-units at 95 % zeros, so 45 % of the units have
// a wave with 20 steps instead of 16 and every other wave waits for it at the next barrier) skip their MFMAs.  (Skipping their
// scatter / clear / fragment reads as well — four more wave-uniform branches per step — breaks the software pipeline: 700 us per
// position against 593.)
template <bool NEXT>
__device__ __forceinline__ void unit_sparse(f32x4 (&acc)[2][4], const char* A, char* D, const uint2*& sp, int nstp, SRing& E, uint4 (&F)[2][4], int lane) {
    const int nst = nstp & 0xFFFF, nreal = nstp >> 16;
    const int m = lane & 15, kg = lane >> 4;
    bf16x8 a[2][2];
    {
        const int ks = s_kstep(E.e[0]);
        a[0][0] = *reinterpret_cast<const bf16x8*>(A + img_off(m, 4 * ks + kg));
        a[0][1] = *reinterpret_cast<const bf16x8*>(A + img_off(16 + m, 4 * ks + kg));
    }
#pragma unroll 1
    for (int g = 0; g < nst; g += SSPD) {
#pragma unroll
        for (int s = 0; s < SSPD; ++s) {
            const int p = s & 1, sn = (s + 1) & (SSPD - 1);
            const bool more = s < SSPD - 1 || g + SSPD < nst;        // another step of THIS unit follows
            s_clear(D, E.e[s]);
            if (NEXT || more) { s_scatter(D, E.e[sn]); s_frags(D, F[p ^ 1], lane); }
            if (more) {
                const int ks = s_kstep(E.e[sn]);
                a[p ^ 1][0] = *reinterpret_cast<const bf16x8*>(A + img_off(m, 4 * ks + kg));
                a[p ^ 1][1] = *reinterpret_cast<const bf16x8*>(A + img_off(16 + m, 4 * ks + kg));
            }
            E.e[s] = sp[(g + s + SSPD) * 64 + lane];                // (past the unit: the next unit's steps; past the stream: zeroed slack)
            if (g + s < nreal) {                                    // (wave-uniform; an all-pad step's fragments are zeros)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, F[p][nt]);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a[p][0], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a[p][1], acc[1][nt], 0, 0, 0);
                }
            }
        }
    }
    sp += (int64_t)nst * 64;
}

// ------------------------------------------------------------------------------------------------ gather form of the sparse unit
// From ~97 % zeros on (the reference's published 97.5 / 98.8 / 99.1 % models) even the scatter stream above does work that no
// non-zero asks for: 16 k-steps of 8 MFMAs per unit whatever the density.  Here a unit is a per-COLUMN list product instead (the
// shape of ortk_sparse.hip's ELL kernel): lane = one of the wave's 64 output columns, its non-zeros arrive in pairs {k1, k2, w1, w2},
// and the A operand sits in LDS TRANSPOSED — three planes of 8 rows, slot(k) = the 8 rows' bf16 values of input column k — so that
// one ds_read_b128 per plane fetches a non-zero's operand for 8 rows and one v_dot2_f32_bf16 per row multiplies a pair.  A wave
// runs as many pairs as its longest column has (padded with zero weights).  The 24 row sums of a lane's column then go through a
// 2.5-KB per-wave LDS tile into the MFMA accumulator layout the rest of the kernel works in (8 rows per pass).
constexpr int GPB = SD * 16;                 // bytes of one plane (512 slots of 16 B)
constexpr int GPL = 3 * GPB;                 // one transposed image: rows 0..23 (24 KB)
constexpr int GTP = 80;                      // fp32 pitch of the 8-row transposition tile (2-way bank conflicts on the b128 reads at most)
constexpr int GTB = 8 * GTP * 4;             // bytes of a wave's tile
__device__ __forceinline__ int g_slot(int k) { return (k & ~7) | (((k & 7) + (k >> 3)) & 7); }      // (spreads a slot's neighbours over the banks)
struct GRing { uint2 e[4]; };
__device__ __forceinline__ void gring_start(GRing& r, const uint2* sp, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) r.e[s] = sp[s * 64 + lane];
}
struct GRows { uint4 a[3], b[3]; };
__device__ __forceinline__ void g_gather(const char* P, unsigned int offs, GRows& r) {
    const unsigned int o1 = offs & 0xFFFFu, o2 = offs >> 16;
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        r.a[h] = *reinterpret_cast<const uint4*>(P + h * GPB + o1);
        r.b[h] = *reinterpret_cast<const uint4*>(P + h * GPB + o2);
    }
}
__device__ __forceinline__ void g_fma(const GRows& r, unsigned int w, float (&g)[24]) {
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const unsigned int ad[4] = {r.a[h].x, r.a[h].y, r.a[h].z, r.a[h].w}, bd[4] = {r.b[h].x, r.b[h].y, r.b[h].z, r.b[h].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // dword d of a slot = rows (2d, 2d + 1) of that input column: low halves -> row 2d, high halves -> row 2d + 1
            g[8 * h + 2 * d] = dot2(__builtin_amdgcn_perm(bd[d], ad[d], 0x05040100u), w, g[8 * h + 2 * d]);
            g[8 * h + 2 * d + 1] = dot2(__builtin_amdgcn_perm(bd[d], ad[d], 0x07060302u), w, g[8 * h + 2 * d + 1]);
        }
    }
}
// acc += A . W_u^T for the wave's 64 columns; np pairs (a multiple of 4) of this (wave, unit); P = the transposed A image; T = the wave's tile
__device__ __forceinline__ void unit_gather(f32x4 (&acc)[2][4], const char* P, char* T, const uint2*& sp, int np, GRing& E, int lane) {
    float g[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) g[i] = 0.f;
#pragma unroll 1
    for (int j = 0; j < np; j += 4) {
        GRows r0, r1;
        g_gather(P, E.e[0].x, r0);
        g_gather(P, E.e[1].x, r1);
        g_fma(r0, E.e[0].y, g); E.e[0] = sp[(j + 4) * 64 + lane];
        g_gather(P, E.e[2].x, r0);
        g_fma(r1, E.e[1].y, g); E.e[1] = sp[(j + 5) * 64 + lane];
        g_gather(P, E.e[3].x, r1);
        g_fma(r0, E.e[2].y, g); E.e[2] = sp[(j + 6) * 64 + lane];
        g_fma(r1, E.e[3].y, g); E.e[3] = sp[(j + 7) * 64 + lane];     // (past the unit: the next unit's pairs; past the stream: zeroed slack)
    }
    sp += (int64_t)np * 64;
    const int m = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float*>(T + (i * GTP + lane) * 4) = g[8 * p + i];
        __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the wave's LDS writes have landed (one wave owns T)
        __builtin_amdgcn_wave_barrier();
        const int rl = m - 8 * (p & 1);
        if (rl >= 0 && rl < 8) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[p >> 1][nt] += *reinterpret_cast<const f32x4*>(T + (rl * GTP + 16 * nt + 4 * q4) * 4);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}
// the transposed form of an A image (all 512 threads, thread = input column k; rows RB.. are zeros)
// LDS transpose read (gfx950): lane 4q + p of a 16-lane group supplies the address of 4 consecutive bf16 of "row" q (segment p); it
// receives element (lane & 15) of each of the group's 4 rows
__device__ __forceinline__ uint2 g_tr_read(const char* p) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
    const bf16x4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4_ __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(uint2, v);
}
template <int RB>
__device__ __forceinline__ void build_planes(const char* img, char* P, int tid) {
    static_assert(RB % 4 == 0 && RB <= 24, "four image rows per transpose read");
    // 32 groups of 16 lanes, one per 16 input columns: a transpose read turns 4 image rows x 16 columns into 4 rows of ONE column per
    // lane — half a slot — instead of one 2-byte read per element (20 per thread: the builds were 1.1 ms of a 16.3-ms decode)
    const int lr = tid & 15, q = lr >> 2, pp = lr & 3, k0 = 16 * (tid >> 4);
    const int ks = k0 + 4 * pp;                                // the columns this lane SUPPLIES (of row 4 b + q)
    const char* src = img + (ks & 7) * 2;
    char* dst = P + g_slot(k0 + lr) * 16;                      // the column this lane RECEIVES
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
        if (8 * h < RB) lo = g_tr_read(src + img_off(8 * h + q, ks >> 3));
        if (8 * h + 4 < RB) hi = g_tr_read(src + img_off(8 * h + 4 + q, ks >> 3));
        *reinterpret_cast<uint4*>(dst + h * GPB) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

__device__ __forceinline__ void zero(f32x4 (&a)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// the 16 columns this lane holds of a 512-wide vector (4 per column tile), e.g. a bias
__device__ __forceinline__ void load_cols(const float* p, int wave, int lane, f32x4 (&v)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) v[nt] = *reinterpret_cast<const f32x4*>(p + 64 * wave + 16 * nt + 4 * (lane >> 4));
}
// (acc + bias [relu]) as bf16 into an A image
template <int RB>
__device__ __forceinline__ void store_img(char* img, const f32x4 (&acc)[2][4], const f32x4 (&bias)[4], bool relu, int wave, int lane) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int row = 16 * mt + (lane & 15), col = 64 * wave + 16 * nt + 4 * (lane >> 4);
            if (RB < 32 && row >= RB) continue;          // rows past the block: no image row (their accumulators are don't-cares)
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mt][nt][r] + bias[nt][r]; if (relu) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<uint2*>(img + img_off(row, col >> 3) + ((col >> 2) & 1) * 8) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
        }
}

// LayerNorm of the register-resident rows (transformer.py:338-341: a (x - mean) / (std_unbiased + eps) + b), two exchanges
// of per-wave partial sums through LDS.  The caller puts a barrier between the result and its consumers.
template <bool RAW>
__device__ __forceinline__ void layer_norm(const f32x4 (&x)[2][4], const float* ga, const float* be, float eps, float* red1, float* red2,
                                           int wave, int lane, f32x4 (&y)[2][4]) {
#pragma clang fp contract(off)
    f32x4 a4[4], b4[4];
    load_cols(ga, wave, lane, a4);
    load_cols(be, wave, lane, b4);
    const int m = lane & 15;
    float mean[2], rinv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += x[mt][nt][r];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (lane < 16) red1[(16 * mt + m) * 8 + wave] = s;
    }
    if constexpr (RAW) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } else __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * 8), p1 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * 8 + 4);
        mean[mt] = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / SD);
        float q = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = x[mt][nt][r] - mean[mt]; q = __builtin_fmaf(d, d, q); }
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        if (lane < 16) red2[(16 * mt + m) * 8 + wave] = q;
    }
    if constexpr (RAW) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } else __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * 8), p1 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * 8 + 4);
        const float var = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / (SD - 1));
        rinv[mt] = 1.f / (sqrtf(var) + eps);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[mt][nt][r] = __builtin_fmaf(a4[nt][r] * (x[mt][nt][r] - mean[mt]), rinv[mt], b4[nt][r]);
    }
}
template <int RB>
__device__ __forceinline__ void store_img_plain(char* img, const f32x4 (&y)[2][4], int wave, int lane) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int row = 16 * mt + (lane & 15), col = 64 * wave + 16 * nt + 4 * (lane >> 4);
            if (RB < 32 && row >= RB) continue;
            *reinterpret_cast<uint2*>(img + img_off(row, col >> 3) + ((col >> 2) & 1) * 8) =
                make_uint2(pack2(y[mt][nt][0], y[mt][nt][1]), pack2(y[mt][nt][2], y[mt][nt][3]));
        }
}

// ------------------------------------------------------------------------------------------------ attention
// Online soft-max state of NR query rows that see the same keys; lane = features 8 lane .. 8 lane + 7 (head lane / 8).
template <int NR>
struct AttState {
    uint4 q[NR];       // packed bf16 query features of this lane
    float o[NR][8], m[NR], l[NR];
    __device__ __forceinline__ void init(const char* Q, int row0, int nr, int lane) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            q[i] = *reinterpret_cast<const uint4*>(Q + img_off(row0 + (i < nr ? i : 0), lane));
            m[i] = -INFINITY; l[i] = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) o[i][d] = 0.f;
        }
    }
    // Scores of KB keys against the NR rows + the online soft-max update; kk = this lane's 8 features of each key's K row;
    // kind[u] = 0 (a key), -1e9 (masked: REPLACES the score, like the reference's masked_fill) or -inf (past the last key).
    // Leaves the un-normalised probabilities in p.  pv() then adds p . V; the two halves are separate so that the K registers
    // can be refilled with the next batch while V of this one is still in flight.
    // (fp contract off + explicit fmaf in scores / pv / layer_norm: every row must get the SAME arithmetic whichever unrolled
    // instance — row of a chunk, row of a pair, tile position — serves it.  Left to -ffp-contract=fast, hipcc fuses some
    // instances and not others, and a decode stops being bit-for-bit equivariant under a permutation of the images:
    // tests/test_gpu_model.py::test_decode_at_bench_size_properties)
    template <int KB>
    __device__ __forceinline__ void scores(const uint4 (&kk)[KB], const float (&kind)[KB], float (&p)[NR][KB]) {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float bm = -INFINITY;
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                float d = dot2(q[i].x, kk[u].x, 0.f);
                d = dot2(q[i].y, kk[u].y, d);
                d = dot2(q[i].z, kk[u].z, d);
                d = dot2(q[i].w, kk[u].w, d);
                d = sum8(d) * 0.125f;                           // 1 / sqrt(64)
                d = kind[u] == 0.f ? d : kind[u];
                p[i][u] = d;
                bm = fmaxf(bm, d);
            }
            const float mn = fmaxf(m[i], bm);
            const float corr = __expf(m[i] - mn);
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < KB; ++u) { p[i][u] = __expf(p[i][u] - mn); s += p[i][u]; }
            l[i] = __builtin_fmaf(l[i], corr, s);
            m[i] = mn;
#pragma unroll
            for (int d = 0; d < 8; ++d) o[i][d] *= corr;
        }
    }
    template <int KB>
    __device__ __forceinline__ void pv(const uint4 (&vv)[KB], const float (&p)[NR][KB]) {
#pragma clang fp contract(off)
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const float v[8] = {lo_f(vv[u].x), hi_f(vv[u].x), lo_f(vv[u].y), hi_f(vv[u].y), lo_f(vv[u].z), hi_f(vv[u].z), lo_f(vv[u].w), hi_f(vv[u].w)};
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int d = 0; d < 8; ++d) o[i][d] = __builtin_fmaf(p[i][u], v[d], o[i][d]);
        }
    }
    __device__ __forceinline__ void finish(char* O, int row0, int nr, int lane) {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (i < nr) {
                const float inv = 1.f / l[i];
                *reinterpret_cast<uint4*>(O + img_off(row0 + i, lane)) =
                    make_uint4(pack2(o[i][0] * inv, o[i][1] * inv), pack2(o[i][2] * inv, o[i][3] * inv),
                               pack2(o[i][4] * inv, o[i][5] * inv), pack2(o[i][6] * inv, o[i][7] * inv));
            }
        }
    }
};

}  // namespace

// ------------------------------------------------------------------------------------------------ the kernel
// SPARSE: the weights arrive as the scatter-entry stream above instead of packed dense fragments.  RB: rows per workgroup
// (32: 160 workgroups for 1 024 images x 5 beams; 20: 256 workgroups, one per CU, four images of five beams each — what the
// sparse stream uses: its cost per workgroup no longer depends on L2 bandwidth shared with the others, so more, smaller
// workgroups only shorten the attention phases).  Rows RB .. 31 of the two MFMA row tiles do not exist: their operand reads
// land in whatever follows the image in LDS, their accumulators are never stored.
// GATHER (with SPARSE, RB <= 24): the units as per-column gather lists over transposed A images (above) instead of the scatter stream.
template <bool SPARSE, int RB, bool GATHER = false>
__global__ __launch_bounds__(512) void decoder_stack_kernel(StackArgs a) {
    static_assert(!GATHER || (SPARSE && RB <= 24), "gather form: sparse stream, at most 24 rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IMG = RB * SD * 2;  // bytes of one bf16 A image
    char* A0 = smem;                  // LayerNorm output / attention output: the A operand of the next projection
    char* A1 = smem + IMG;            // query image; FFN hidden chunk (even)
    char* KN = smem + 2 * IMG;        // this position's K; FFN hidden chunk (odd)
    char* VN = smem + 3 * IMG;        // this position's V
    float* red1 = reinterpret_cast<float*>(smem + 4 * IMG);
    float* red2 = red1 + 32 * 8;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = lane0;
    const int U = 6 + 2 * a.NC;
    if (!SPARSE && (int)blockIdx.x >= a.nblocks) {
        // L2 prefetcher of one XCD (workgroups are dealt to the 8 XCDs round-robin, so nblocks + j runs on XCD (nblocks + j) % 8
        // and workgroup k < 8 — the pace-maker that publishes its progress — on XCD k).  The compute workgroups of an XCD walk
        // the same weight stream in step; without this every one of their fragment loads is an L2 MISS that all of them wait
        // for (measured: 6.6 us per 512-KB unit = 33 B/clk per CU).  One dword per 128-byte line, kept AHEAD units in front.
        const int xcd = (int)(blockIdx.x & 7);
        const int base = a.t * a.L * U;
        unsigned int sink = 0;
        int spin = 0;
        for (int u = 0; u < a.L * U; ++u) {
            // (bounded: a prefetcher that loses its pace-maker — another dispatch order — stops waiting after ~1 ms and runs free)
            for (; spin < 4096 && __hip_atomic_load(a.progress + xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < base + u + 1 - STACK_AHEAD; ++spin)
                __builtin_amdgcn_s_sleep(8);
            spin = spin >= 4096 ? 4096 : 0;
            const unsigned int* src = reinterpret_cast<const unsigned int*>(a.wpk + ((int64_t)wave * a.L * U + u) * 16 * KSTEP) + lane * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) sink += src[i * 64 * 32];          // 8 x (64 lanes x 128 B) = this wave's 64 KB of the unit
        }
        if (sink == 0x9E3779B1u) a.progress[8] = 1;                          // (keeps the loads)
        return;
    }
    const bool pace = !SPARSE && a.progress != nullptr && blockIdx.x < 8 && tid == 0;
    int unit_no = a.t * a.L * U;
// a fresh, opaque copy of the lane id per phase: without it the compiler hoists every lane-derived address of every phase
    // out of the layer loop and keeps ~60 of them alive (spilled) through the whole kernel
#define STACK_FRESH_LANE() do { lane = lane0; asm volatile("" : "+v"(lane)); } while (0)
#define STACK_UNIT_BEGIN() do { ++unit_no; if (pace) __hip_atomic_store(a.progress + blockIdx.x, unit_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
    const int r0 = blockIdx.x * RB;
    const int m = lane & 15, q4 = lane >> 4;
    const int Lk = a.t + 1;

    // residual stream of the block's rows, accumulator layout
    f32x4 x[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int g = min(r0 + 16 * mt + m, a.rows - 1);
        if (a.tok) {        // Embeddings (transformer.py:366-373: lut(x) * sqrt(d_model)) + positional encoding, same arithmetic as embed_fwd_kernel
            const float* e = a.lut + a.tok[g] * SD;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = 64 * wave + 16 * nt + 4 * q4;
                const f32x4 ev = *reinterpret_cast<const f32x4*>(e + col), pv = *reinterpret_cast<const f32x4*>(a.pe_t + col);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[mt][nt][r] = ev[r] * a.emb_scale + pv[r];
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) x[mt][nt] = *reinterpret_cast<const f32x4*>(a.x_io + (int64_t)g * SD + 64 * wave + 16 * nt + 4 * q4);
        }
    }
    // dense stream state
    const uint4* wp = a.wpk + (int64_t)wave * a.L * U * 16 * KSTEP;
    Ring ring;
    // sparse stream state
    const uint2* sp = nullptr;
    cint_ptr nstp = nullptr;
    char* D = smem + 4 * IMG + 2 * 32 * 8 * 4 + wave * SDBUF;      // this wave's fragment buffer
    SRing E;
    uint4 F[2][4];
    // gather form: transposed images of A0 (PA) and of the FFN hidden chunk (PH), the wave's transposition tile, the pair ring
    char* PA = smem + 4 * IMG + 2 * 32 * 8 * 4;
    char* PH = PA + GPL;
    char* GT = PH + GPL + wave * GTB;
    GRing GE;
    if constexpr (GATHER) {
        auto uni64 = [](uint64_t v) { return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
        nstp = (cint_ptr)(uintptr_t)uni64((uint64_t)(uintptr_t)(a.snst + wave * a.L * U));
        sp = a.sstream + (int64_t)a.sstart[wave] * 64;
        gring_start(GE, sp, lane);
    } else if constexpr (SPARSE) {
        auto uni64 = [](uint64_t v) { return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
        nstp = (cint_ptr)(uintptr_t)uni64((uint64_t)(uintptr_t)(a.snst + wave * a.L * U));
        sp = a.sstream + (int64_t)a.sstart[wave] * 64;
        sring_start(E, sp, lane);
        const uint4 z4 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(D + i * 1024 + lane * 16) = z4;
        if (lane < 16) *reinterpret_cast<uint4*>(D + 4096 + lane * 16) = z4;
        s_unit_cold(D, E, F, lane);
    } else {
        ring_start(ring, wp, lane);
    }
    // one [32 x 512] x [512 x 512] unit: NEXT = the stream keeps running into the unit that follows
#define STACK_UNIT(ACC, AIMG, NEXT)                                                                    \
    do {                                                                                               \
        STACK_UNIT_BEGIN();                                                                            \
        if constexpr (GATHER) { const int np_ = *nstp++; unit_gather(ACC, (AIMG) == A0 ? PA : PH, GT, sp, np_, GE, lane); } \
        else if constexpr (SPARSE) { const int nst_ = *nstp++; unit_sparse<NEXT>(ACC, AIMG, D, sp, nst_, E, F, lane); } \
        else unit_gemm<NEXT>(ACC, AIMG, wp, ring, lane);                                               \
    } while (0)
#define STACK_RESTART()                                                                                \
    do { if constexpr (GATHER) gring_start(GE, sp, lane); else if constexpr (SPARSE) s_unit_cold(D, E, F, lane); else ring_start(ring, wp, lane); } while (0)
// gather form: a freshly published A image (behind the barrier that publishes it) gets its transposed planes
#define STACK_PUBLISH(AIMG)                                                                            \
    do { if constexpr (GATHER) { build_planes<RB>(AIMG, (AIMG) == A0 ? PA : PH, tid); __syncthreads(); } } while (0)

#define STACK_SYNC() __syncthreads()
    for (int l = 0; l < a.L; ++l) {
        const StackLayer& P = a.layer[l];
        f32x4 acc[2][4], y[2][4], bias[4];
        // ---- LayerNorm 0 -> A0
        STACK_FRESH_LANE();
        layer_norm<false>(x, P.n0a, P.n0b, a.eps, red1, red2, wave, lane, y);
        store_img_plain<RB>(A0, y, wave, lane);
        STACK_SYNC();
        STACK_PUBLISH(A0);
        // ---- packed QKV: three units -> q (A1), k (KN), v (VN)
        STACK_FRESH_LANE();
        load_cols(P.bqkv, wave, lane, bias); zero(acc); STACK_UNIT(acc, A0, true); store_img<RB>(A1, acc, bias, false, wave, lane);
        load_cols(P.bqkv + SD, wave, lane, bias); zero(acc); STACK_UNIT(acc, A0, true); store_img<RB>(KN, acc, bias, false, wave, lane);
        load_cols(P.bqkv + 2 * SD, wave, lane, bias); zero(acc); STACK_UNIT(acc, A0, false); store_img<RB>(VN, acc, bias, false, wave, lane);
        STACK_SYNC();
        // ---- self-attention: wave w serves rows w, w + 8, w + 16, .. of the block over their cache rows + this position; o -> A0
        STACK_FRESH_LANE();
        if (!(a.debug & 1)) {
            // The wave's SROWS rows run in lock-step, each with its own K / V registers: SROWS x SKB row loads of K and as
            // many of V are in flight, K of the next batch requested as soon as the scores have consumed this one's, V as
            // soon as p . V has.  The phase is bound by the bytes a wave can keep in flight (registers), not by its arithmetic.
            const uint4* ck = reinterpret_cast<const uint4*>(P.ck) + lane;
            const uint4* cv = reinterpret_cast<const uint4*>(P.cv) + lane;
            const int nb = (a.t + SKB - 1) / SKB;
            constexpr int NI = (RB + 7) / 8;          // rows per wave
#pragma unroll 1
            for (int i0 = 0; i0 < NI; i0 += SROWS) {
                auto rowof = [&](int i) { return RB == 32 ? 4 * wave + i : wave + 8 * i; };     // i-th row of this wave
                if (rowof(i0) >= RB || r0 + rowof(i0) >= a.rows) break;        // (wave-uniform) no row of this pass exists
                int idx[SROWS];          // lane j: physical cache row of key j of row r
                int rowi[SROWS];         // row inside the block (clamped: a missing second row repeats the first)
                AttState<1> st[SROWS];
                uint4 kq[SROWS][SKB], vq[SROWS][SKB];
#pragma unroll
                for (int r = 0; r < SROWS; ++r) {
                    const int row = rowof(i0 + r);
                    rowi[r] = (row < RB && r0 + row < a.rows) ? row : rowof(i0);
                    const int g = r0 + rowi[r];
                    idx[r] = a.kvidx ? a.kvidx[(int64_t)g * Lk + min(lane, a.t)] : g * a.T + min(lane, a.t);
                }
                auto issue_k = [&](int b) {
#pragma unroll
                    for (int r = 0; r < SROWS; ++r)
#pragma unroll
                        for (int u = 0; u < SKB; ++u) kq[r][u] = ck[(int64_t)__builtin_amdgcn_readlane(idx[r], min(b * SKB + u, a.t - 1)) * (SD / 8)];
                };
                auto issue_v = [&](int b) {
#pragma unroll
                    for (int r = 0; r < SROWS; ++r)
#pragma unroll
                        for (int u = 0; u < SKB; ++u) vq[r][u] = cv[(int64_t)__builtin_amdgcn_readlane(idx[r], min(b * SKB + u, a.t - 1)) * (SD / 8)];
                };
                if (nb > 0) { issue_k(0); issue_v(0); }
#pragma unroll
                for (int r = 0; r < SROWS; ++r) st[r].init(A1, rowi[r], 1, lane);
                for (int b = 0; b < nb; ++b) {
                    float kind[SKB], p[SROWS][1][SKB];
#pragma unroll
                    for (int u = 0; u < SKB; ++u) kind[u] = b * SKB + u < a.t ? 0.f : -INFINITY;
#pragma unroll
                    for (int r = 0; r < SROWS; ++r) st[r].scores<SKB>(kq[r], kind, p[r]);
                    if (b + 1 < nb) issue_k(b + 1);
#pragma unroll
                    for (int r = 0; r < SROWS; ++r) st[r].pv<SKB>(vq[r], p[r]);
                    if (b + 1 < nb) issue_v(b + 1);
                }
#pragma unroll
                for (int r = 0; r < SROWS; ++r) {
                    // this position: K / V from the LDS images, appended to the cache
                    const int row = rowi[r];
                    const uint4 kself[1] = {*reinterpret_cast<const uint4*>(KN + img_off(row, lane))};
                    const uint4 vself[1] = {*reinterpret_cast<const uint4*>(VN + img_off(row, lane))};
                    const float kindself[1] = {0.f};
                    float pself[1][1];
                    st[r].scores<1>(kself, kindself, pself);
                    st[r].pv<1>(vself, pself);
                    if (row == rowof(i0 + r)) {      // (a repeated row writes nothing)
                        const int64_t slot = (int64_t)__builtin_amdgcn_readlane(idx[r], a.t) * (SD / 8);
                        reinterpret_cast<uint4*>(P.ck)[slot + lane] = kself[0];
                        reinterpret_cast<uint4*>(P.cv)[slot + lane] = vself[0];
                        st[r].finish(A0, row, 1, lane);
                    }
                }
            }
        }
        STACK_SYNC();
        STACK_PUBLISH(A0);
        // ---- output projection + residual, LayerNorm 1 -> A0
        STACK_FRESH_LANE();
        STACK_RESTART();
        load_cols(P.bo, wave, lane, bias); zero(acc); STACK_UNIT(acc, A0, true);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) x[mt][nt] += acc[mt][nt] + bias[nt];
        layer_norm<false>(x, P.n1a, P.n1b, a.eps, red1, red2, wave, lane, y);       // (its barriers: every wave is done reading A0)
        store_img_plain<RB>(A0, y, wave, lane);
        STACK_SYNC();
        STACK_PUBLISH(A0);
        // ---- cross-attention query -> A1
        STACK_FRESH_LANE();
        load_cols(P.cqb, wave, lane, bias); zero(acc); STACK_UNIT(acc, A0, false); store_img<RB>(A1, acc, bias, false, wave, lane);
        STACK_SYNC();
        // ---- cross-attention: chunks of up to XNR rows of one image, dealt round-robin to the waves; o -> A0
        STACK_FRESH_LANE();
        if (!(a.debug & 2)) {
            const int last = min(r0 + RB, a.rows) - 1;
            const int img0 = r0 / a.per_img, img1 = last / a.per_img;
            int chunk = 0;
            for (int im = img0; im <= img1; ++im) {
                const int lo = max(im * a.per_img, r0), hi = min((im + 1) * a.per_img - 1, last);
                for (int c0 = lo; c0 <= hi; c0 += XNR, ++chunk) {
                    if ((chunk & 7) != wave) continue;
                    const int nr = min(XNR, hi - c0 + 1);
                    AttState<XNR> st;
                    st.init(A1, c0 - r0, nr, lane);
                    const uint4* xk = reinterpret_cast<const uint4*>(P.xk + (int64_t)im * a.S * a.ldx) + lane;
                    const uint4* xv = reinterpret_cast<const uint4*>(P.xv + (int64_t)im * a.S * a.ldx) + lane;
                    const int64_t pitch = a.ldx / 8;
                    const float* mk = a.att_masks + (int64_t)im * a.S;
                    const float mk0 = lane < a.S ? mk[lane] : 1.f, mk1 = lane + 64 < a.S ? mk[lane + 64] : 1.f;
                    const int nb = (a.S + XKB - 1) / XKB;
                    uint4 kq[XKB], vq[XKB];
#pragma unroll
                    for (int u = 0; u < XKB; ++u) { const int j = min(u, a.S - 1); kq[u] = xk[j * pitch]; vq[u] = xv[j * pitch]; }
                    for (int b = 0; b < nb; ++b) {
                        float kind[XKB], p[XNR][XKB];
#pragma unroll
                        for (int u = 0; u < XKB; ++u) {
                            const int j = b * XKB + u;
                            const float mv = j < 64 ? rdlane(mk0, j & 63) : rdlane(mk1, j & 63);
                            kind[u] = j >= a.S ? -INFINITY : (mv == 0.f ? -1e9f : 0.f);
                        }
                        st.scores<XKB>(kq, kind, p);
                        if (b + 1 < nb) {
#pragma unroll
                            for (int u = 0; u < XKB; ++u) kq[u] = xk[min((b + 1) * XKB + u, a.S - 1) * pitch];
                        }
                        st.pv<XKB>(vq, p);
                        if (b + 1 < nb) {
#pragma unroll
                            for (int u = 0; u < XKB; ++u) vq[u] = xv[min((b + 1) * XKB + u, a.S - 1) * pitch];
                        }
                    }
                    st.finish(A0, c0 - r0, nr, lane);
                }
            }
        }
        STACK_SYNC();
        STACK_PUBLISH(A0);
        // ---- output projection + residual, LayerNorm 2 -> A0
        STACK_FRESH_LANE();
        STACK_RESTART();
        load_cols(P.cob, wave, lane, bias); zero(acc); STACK_UNIT(acc, A0, true);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) x[mt][nt] += acc[mt][nt] + bias[nt];
        layer_norm<false>(x, P.n2a, P.n2b, a.eps, red1, red2, wave, lane, y);
        store_img_plain<RB>(A0, y, wave, lane);
        STACK_SYNC();
        STACK_PUBLISH(A0);
        STACK_FRESH_LANE();
        // ---- FFN, 512 hidden units at a time: h_c = relu(y W1_c^T + b1_c) -> LDS, acc2 += h_c W2[:, c]^T
        f32x4 acc2[2][4];
        zero(acc2);
        for (int c = 0; c < ((a.debug & 4) ? 0 : a.NC); ++c) {
            char* Hc = (c & 1) ? KN : A1;
            load_cols(P.b1 + c * SD, wave, lane, bias); zero(acc); STACK_UNIT(acc, A0, true); store_img<RB>(Hc, acc, bias, true, wave, lane);
            STACK_SYNC();
            STACK_PUBLISH(Hc);
            STACK_UNIT(acc2, Hc, true);
        }
        load_cols(P.b2, wave, lane, bias);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) x[mt][nt] += acc2[mt][nt] + bias[nt];
    }
    // ---- final LayerNorm -> bf16 rows for the generator
    STACK_FRESH_LANE();
    f32x4 y[2][4];
    layer_norm<false>(x, a.fa, a.fb, a.eps, red1, red2, wave, lane, y);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = 16 * mt + (lane & 15), g = r0 + row;
        if (g < a.rows && row < RB) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<uint2*>(a.y_out + (int64_t)g * SD + 64 * wave + 16 * nt + 4 * (lane >> 4)) =
                    make_uint2(pack2(y[mt][nt][0], y[mt][nt][1]), pack2(y[mt][nt][2], y[mt][nt][3]));
        }
    }
#undef STACK_SYNC
#undef STACK_UNIT
#undef STACK_RESTART
#undef STACK_PUBLISH
}

// wpk[((((w L + l) U + u) 16 + ks) 4 + nt) 64 + lane] = the 8 bf16 W_u[64 w + 16 nt + (lane & 15)][32 ks + 8 (lane >> 4) ..]
__global__ __launch_bounds__(256) void stack_pack_kernel(const __bf16* __restrict__ w16, uint4* __restrict__ wpk, StackPack t) {
    const int U = 6 + 2 * t.NC;
    const int64_t total = (int64_t)8 * t.L * U * 16 * KSTEP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), nt = (int)((i >> 6) & 3), ks = (int)((i >> 8) & 15);
        int64_t rest = i >> 12;
        const int u = (int)(rest % U); rest /= U;
        const int l = (int)(rest % t.L);
        const int w = (int)(rest / t.L);
        // unit -> (weight matrix, first output row, first input column, leading dimension)
        int64_t base; int ld;
        if (u < 3)       { base = t.off[l][0] + (int64_t)u * SD * SD; ld = SD; }
        else if (u < 6)  { base = t.off[l][u - 2]; ld = SD; }
        else {
            const int c = (u - 6) >> 1;
            if (((u - 6) & 1) == 0) { base = t.off[l][4] + (int64_t)c * SD * SD; ld = SD; }            // W1 rows 512 c ..
            else                    { base = t.off[l][5] + (int64_t)c * SD; ld = t.NC * SD; }           // W2 columns 512 c ..
        }
        const __bf16* src = w16 + base + (int64_t)(64 * w + 16 * nt + (lane & 15)) * ld + 32 * ks + 8 * (lane >> 4);
        wpk[i] = *reinterpret_cast<const uint4*>(src);
    }
}

size_t stack_packed_bytes(int L, int NC) {
    return ((size_t)8 * L * (6 + 2 * NC) * 16 * KSTEP + (size_t)SPD * KSTEP) * sizeof(uint4);   // + the ring's read-ahead past the end
}

int stack_pack(const void* w16, void* wpk, const StackPack& t, hipStream_t s) {
    hipLaunchKernelGGL(stack_pack_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const __bf16*>(w16), reinterpret_cast<uint4*>(wpk), t);
    ORTK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ sparse stream builder
namespace {
constexpr int SSLACK = SSPD + 4;        // all-zero steps behind every wave's stream (the ring's look-ahead: entry 0 = weight 0 at position 0)
// the 32 weights lane `lane` of wave w feeds to the MFMAs of k-step ks of unit u (4 column tiles x 8), as in stack_pack_kernel
__device__ __forceinline__ void sstack_load(const __bf16* w16, const StackPack& t, int w, int l, int u, int ks, int lane, uint4 (&f)[4]) {
    int64_t base; int ld;
    if (u < 3)       { base = t.off[l][0] + (int64_t)u * SD * SD; ld = SD; }
    else if (u < 6)  { base = t.off[l][u - 2]; ld = SD; }
    else {
        const int c = (u - 6) >> 1;
        if (((u - 6) & 1) == 0) { base = t.off[l][4] + (int64_t)c * SD * SD; ld = SD; }
        else                    { base = t.off[l][5] + (int64_t)c * SD; ld = t.NC * SD; }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        f[nt] = *reinterpret_cast<const uint4*>(w16 + base + (int64_t)(64 * w + 16 * nt + (lane & 15)) * ld + 32 * ks + 8 * (lane >> 4));
}
// bit (8 nt + j) = weight j of tile nt is non-zero (+0 and -0 are both zeros: a masked weight is w * 0)
__device__ __forceinline__ unsigned int sstack_mask(const uint4 (&f)[4]) {
    unsigned int mk = 0;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const unsigned int d[4] = {f[nt].x, f[nt].y, f[nt].z, f[nt].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (d[i] & 0x00007FFFu) mk |= 1u << (8 * nt + 2 * i);
            if (d[i] & 0x7FFF0000u) mk |= 1u << (8 * nt + 2 * i + 1);
        }
    }
    return mk;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int steps_of(int cnt) { return max(1, (cnt + SCAP - 1) / SCAP); }
}  // namespace

// cnt[(w LU + lu) 16 + ks] = non-zeros among the 64 x 32 weights of wave w, unit lu, k-step ks
__global__ __launch_bounds__(256) void sstack_count_kernel(const __bf16* __restrict__ w16, int32_t* __restrict__ cnt, StackPack t) {
    const int U = 6 + 2 * t.NC, LU = t.L * U;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= 8 * LU * 16) return;
    const int ks = item & 15, lu = (item >> 4) % LU, w = (item >> 4) / LU;
    uint4 f[4];
    sstack_load(w16, t, w, lu / U, lu % U, ks, lane, f);
    const int c = wave_sum(__popc(sstack_mask(f)));
    if (lane == 0) cnt[item] = c;
}
// thread w: the step layout of wave w's stream.  cnt[item] -> first step of the k-step (relative to the wave's stream);
// nst[w LU + lu] = steps of the unit (a multiple of SSPD); start[w] = first step of the wave in the buffer;
// stats = {steps in all, non-zeros}
__global__ void sstack_scan_kernel(int32_t* __restrict__ cnt, int32_t* __restrict__ nst, int64_t* __restrict__ start, int64_t* __restrict__ stats, int LU) {
    __shared__ int64_t tot[8], nzs[8];
    const int w = threadIdx.x;
    if (w < 8) {
        int64_t run = 0, z = 0;
        for (int lu = 0; lu < LU; ++lu) {
            int st = 0;
            for (int ks = 0; ks < 16; ++ks) {
                const int i = (w * LU + lu) * 16 + ks, c = cnt[i];
                z += c;
                cnt[i] = (int)(run + st);
                st += steps_of(c);
            }
            const int real = st;
            st = (st + SSPD - 1) / SSPD * SSPD;
            nst[w * LU + lu] = st | (real << 16);
            run += st;
        }
        tot[w] = run; nzs[w] = z;
    }
    __syncthreads();
    if (w == 0) {
        int64_t run = 0, z = 0;
        for (int i = 0; i < 8; ++i) { start[i] = run; run += tot[i] + SSLACK; z += nzs[i]; }
        stats[0] = run; stats[1] = z;
    }
}
// One wave per (w, lu, ks): its entries, dealt to the four 32-lane store groups of a step (group g = entries [32 g, 32 g + 32)
// of the step: lanes 0-31 / 32-63 of the .x store, then of the .y store) so that a group gets at most two entries per LDS bank
// where possible.  The wave of a unit's last k-step also writes the unit's all-pad steps.
__global__ __launch_bounds__(256) void sstack_fill_kernel(const __bf16* __restrict__ w16, const int32_t* __restrict__ first, const int32_t* __restrict__ nst,
                                                         const int64_t* __restrict__ start, uint2* __restrict__ stream, StackPack t) {
    __shared__ unsigned int ent[4][2048];            // the k-step's entries in (lane, bit) order
    __shared__ unsigned int slot[4][SCAP];           // one step being dealt
    const int U = 6 + 2 * t.NC, LU = t.L * U;
    const int wv = threadIdx.x >> 6, item = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    if (item >= 8 * LU * 16) return;
    const int ks = item & 15, lu = (item >> 4) % LU, w = (item >> 4) / LU;
    uint4 f[4];
    sstack_load(w16, t, w, lu / U, lu % U, ks, lane, f);
    unsigned int mk = sstack_mask(f);
    const int mine = __popc(mk);
    int pre = mine;                                    // inclusive scan over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pre, o, 64); if (lane >= o) pre += v; }
    const int total = __shfl(pre, 63, 64);
    int e = pre - mine;
    const unsigned short* h = reinterpret_cast<const unsigned short*>(f);
    while (mk) {
        const int b = __ffs(mk) - 1;
        mk &= mk - 1;
        const int nt = b >> 3, j = b & 7;
        ent[wv][e++] = (unsigned int)h[8 * nt + j] | ((unsigned int)((nt * 64 + lane) * 8 + j) << 16) | ((unsigned int)ks << 28);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): this wave's LDS writes are done (one wave owns ent[wv])
    __builtin_amdgcn_wave_barrier();
    unsigned int* out = reinterpret_cast<unsigned int*>(stream + (start[w] + first[item]) * 64);     // entry r of step s: out[(s 64 + (r & 63)) 2 + (r >> 6)]
    const int ns = steps_of(total);
    for (int st = 0; st < ns; ++st) {
        const int n = min(SCAP, total - st * SCAP);    // entries of this step
        // pad entries first (private slots behind the buffer: bank = lane, at most one per bank and group)
        for (int r = lane; r < SCAP; r += 64) slot[wv][r] = ((unsigned int)(2048 + 2 * (r & 63) + (r >> 6)) << 16) | ((unsigned int)ks << 28);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            // greedy deal: every entry goes to the group with the fewest entries on its bank so far (then the emptiest);
            // a group holds 32.  bank of an entry = dword index of its halfword mod 32.
            unsigned char load[4][32];
            int fill[4] = {0, 0, 0, 0};
            for (int g = 0; g < 4; ++g) for (int b = 0; b < 32; ++b) load[g][b] = 0;
            for (int i = 0; i < n; ++i) {
                const unsigned int v = ent[wv][st * SCAP + i];
                const int bank = (v >> 17) & 31;
                int best = -1, bl = 1 << 30;
                for (int g = 0; g < 4; ++g) {
                    if (fill[g] >= 32) continue;
                    const int score = load[g][bank] * 64 + fill[g];
                    if (score < bl) { bl = score; best = g; }
                }
                slot[wv][best * 32 + fill[best]] = v;
                ++fill[best]; ++load[best][bank];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // slot r of the deal = lane (r & 31) + 32 ((r >> 5) & 1) of store (r >> 6)
        for (int r = lane; r < SCAP; r += 64) out[((int64_t)st * 64 + (r & 63)) * 2 + (r >> 6)] = slot[wv][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    if (ks == 15) {
        const int64_t ustart = start[w] + first[item - 15], uend = ustart + (nst[w * LU + lu] & 0xFFFF);
        uint2* us = stream + (start[w] + first[item] + ns) * 64;
        for (int64_t st = start[w] + first[item] + ns; st < uend; ++st, us += 64)
            us[lane] = make_uint2((unsigned int)(2048 + 2 * lane) << 16, (unsigned int)(2048 + 2 * lane + 1) << 16);
        // behind the wave's last unit: the ring's look-ahead (entry 0 = weight 0 at position 0)
        if (lu == LU - 1)
            for (int i = 0; i < SSLACK; ++i, us += 64) us[lane] = make_uint2(0u, 0u);
    }
}

// Worst case (no zero at all): 16 steps per k-step, 256 per unit — 8 bytes per lane and step.
static size_t sstack_cap_steps(int L, int NC) { return 8 * ((size_t)L * (6 + 2 * NC) * 256 + SSLACK); }
size_t sstack_bytes(int L, int NC, SStackBufs* carve, void* base) {
    const size_t LU = (size_t)L * (6 + 2 * NC);
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? (char*)base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    SStackBufs b;
    b.stream_bytes = sstack_cap_steps(L, NC) * 64 * sizeof(uint2);
    b.stream = (uint2*)take(b.stream_bytes);
    b.cnt = (int32_t*)take(8 * LU * 16 * 4);
    b.nst = (int32_t*)take(8 * LU * 4);
    b.start = (int64_t*)take(8 * 8);
    b.stats = (int64_t*)take(2 * 8);
    if (carve) *carve = b;
    return off;
}

int sstack_pack(const void* w16, const SStackBufs& b, const StackPack& t, hipStream_t s) {
    const int LU = t.L * (6 + 2 * t.NC);
    const __bf16* w = reinterpret_cast<const __bf16*>(w16);
    const unsigned items = (unsigned)ortk_cdiv(8 * LU * 16, 4);
    hipLaunchKernelGGL(sstack_count_kernel, dim3(items), dim3(256), 0, s, w, b.cnt, t);
    ORTK_CHECK_LAUNCH();
    hipLaunchKernelGGL(sstack_scan_kernel, dim3(1), dim3(64), 0, s, b.cnt, b.nst, b.start, b.stats, LU);
    ORTK_CHECK_LAUNCH();
    hipLaunchKernelGGL(sstack_fill_kernel, dim3(items), dim3(256), 0, s, w, b.cnt, b.nst, b.start, b.stream, t);
    ORTK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ gather-list builder
// Stream of the gather form: per wave, per unit, np pair rows of 64 lanes x {off1 | off2 << 16, w1 | w2 << 16}: lane = output column
// 64 w + lane of the unit, its non-zeros in input order, two per row, off = byte offset of the input column's slot in a plane; np = the
// longest column of the wave's 64, in pairs, rounded up to the ring's four (the rest: zero weights at slot 0).  Same buffers as the
// scatter stream (SStackBufs: cnt = first pair row of every (wave, unit), nst = its pair rows).
namespace {
__device__ __forceinline__ void gstack_row(const StackPack& t, int w, int l, int u, int lane, int64_t& base) {
    int ld;
    if (u < 3)       { base = t.off[l][0] + (int64_t)u * SD * SD; ld = SD; }
    else if (u < 6)  { base = t.off[l][u - 2]; ld = SD; }
    else {
        const int c = (u - 6) >> 1;
        if (((u - 6) & 1) == 0) { base = t.off[l][4] + (int64_t)c * SD * SD; ld = SD; }
        else                    { base = t.off[l][5] + (int64_t)c * SD; ld = t.NC * SD; }
    }
    base += (int64_t)(64 * w + lane) * ld;          // the 512 inputs of this lane's output column
}
}  // namespace
// one wave per (w, lu): np[w LU + lu] = pair rows of the unit, np[8 LU + ..] = its non-zeros
__global__ __launch_bounds__(256) void gstack_count_kernel(const __bf16* __restrict__ w16, int32_t* __restrict__ np, StackPack t) {
    const int U = 6 + 2 * t.NC, LU = t.L * U;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= 8 * LU) return;
    const int lu = item % LU, w = item / LU;
    int64_t base; gstack_row(t, w, lu / U, lu % U, lane, base);
    int n = 0;
    for (int q = 0; q < SD / 8; ++q) {
        const uint4 f = *reinterpret_cast<const uint4*>(w16 + base + 8 * q);
        const unsigned int d[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) n += ((d[i] & 0x00007FFFu) ? 1 : 0) + ((d[i] & 0x7FFF0000u) ? 1 : 0);
    }
    int mx = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    const int tot = wave_sum(n);
    if (lane == 0) { np[item] = ((mx + 1) / 2 + 3) & ~3; np[8 * LU + item] = tot; }
}
// thread w: first[w LU + lu] (relative to the wave's stream), start[w]; stats = {pair rows in all, non-zeros}
__global__ void gstack_scan_kernel(const int32_t* __restrict__ np, int32_t* __restrict__ first, int64_t* __restrict__ start, int64_t* __restrict__ stats, int LU) {
    __shared__ int64_t tot[8], nzs[8];
    const int w = threadIdx.x;
    if (w < 8) {
        int64_t run = 0, z = 0;
        for (int lu = 0; lu < LU; ++lu) { first[w * LU + lu] = (int)run; run += np[w * LU + lu]; z += np[8 * LU + w * LU + lu]; }
        tot[w] = run; nzs[w] = z;
    }
    __syncthreads();
    if (w == 0) {
        int64_t run = 0, z = 0;
        for (int i = 0; i < 8; ++i) { start[i] = run; run += tot[i] + SSLACK; z += nzs[i]; }
        stats[0] = run; stats[1] = z;
    }
}
__global__ __launch_bounds__(256) void gstack_fill_kernel(const __bf16* __restrict__ w16, const int32_t* __restrict__ first, const int32_t* __restrict__ nst,
                                                         const int64_t* __restrict__ start, uint2* __restrict__ stream, StackPack t) {
    const int U = 6 + 2 * t.NC, LU = t.L * U;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= 8 * LU) return;
    const int lu = item % LU, w = item / LU;
    int64_t base; gstack_row(t, w, lu / U, lu % U, lane, base);
    uint2* out = stream + (start[w] + first[item]) * 64 + lane;
    const int np = nst[item];
    int j = 0, have = 0;
    unsigned int o1 = 0, w1 = 0;
    const unsigned short* row = reinterpret_cast<const unsigned short*>(w16 + base);
    for (int q = 0; q < SD / 8; ++q) {
        const uint4 f = *reinterpret_cast<const uint4*>(row + 8 * q);
        const unsigned int d[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned int h = (i & 1) ? d[i >> 1] >> 16 : d[i >> 1] & 0xFFFFu;
            if (h & 0x7FFFu) {
                const unsigned int off = (unsigned int)g_slot(8 * q + i) * 16u;
                if (!have) { o1 = off; w1 = h; have = 1; }
                else { out[(int64_t)j * 64] = make_uint2(o1 | (off << 16), w1 | (h << 16)); ++j; have = 0; }
            }
        }
    }
    if (have) { out[(int64_t)j * 64] = make_uint2(o1, w1); ++j; }           // (second half: weight 0 at slot 0)
    for (; j < np; ++j) out[(int64_t)j * 64] = make_uint2(0u, 0u);
    // behind the wave's last unit: the ring's look-ahead
    if (lu == LU - 1) for (int i = 0; i < SSLACK; ++i) out[(int64_t)(np + i) * 64] = make_uint2(0u, 0u);
}
int gstack_pack(const void* w16, const SStackBufs& b, const StackPack& t, hipStream_t s) {
    const int LU = t.L * (6 + 2 * t.NC);
    const __bf16* w = reinterpret_cast<const __bf16*>(w16);
    const unsigned items = (unsigned)ortk_cdiv(8 * LU, 4);
    // (b.cnt holds 8 LU 16 ints: first[8 LU] at its head, the count kernel's {pair rows, non-zeros}[2 x 8 LU] behind it; b.nst = the pair rows)
    int32_t* np = b.cnt + 8 * LU;
    hipLaunchKernelGGL(gstack_count_kernel, dim3(items), dim3(256), 0, s, w, np, t);
    ORTK_CHECK_LAUNCH();
    hipLaunchKernelGGL(gstack_scan_kernel, dim3(1), dim3(64), 0, s, np, b.cnt, b.start, b.stats, LU);
    ORTK_CHECK_LAUNCH();
    if (hipMemcpyAsync(b.nst, np, (size_t)8 * LU * sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return ORTK_EINVAL;
    hipLaunchKernelGGL(gstack_fill_kernel, dim3(items), dim3(256), 0, s, w, b.cnt, b.nst, b.start, b.stream, t);
    ORTK_CHECK_LAUNCH();
    return 0;
}

// ================================================================================================ column-split form
// The kernel above is bound by the bytes ONE compute unit pulls out of L2 (every workgroup streams all 42 MB of decoder weights
// per position at 79-85 GB/s; a CU gets 95 GB/s at most, profiles/r03_fetch_rate_probe.txt), whatever the number of rows it
// holds.  Here G workgroups SHARE 64 rows and split every unit's 512 output columns (and the 8 heads) G ways: each streams 1/G
// of the weights.  What a member needs of the others — the attention output and the FFN hidden chunk as the next unit's A
// operand (bf16), the out-projections' residual updates (fp32; every member keeps the FULL residual rows in registers, so
// LayerNorm stays local) — travels through a [64 x 512] tile in global memory: plain stores (the vector L1 is write-through),
// `s_waitcnt vmcnt(0)`, a relaxed counter in L2, `sc1` loads that bypass the reader's L1.  That is only coherent inside ONE
// XCD's L2, so the members of a group are the workgroups x + 8 k of one XCD x; with agent-scope fences (L2 write-back +
// invalidate) an exchange costs 19-23 us, this way 1.8 us (bf16 tile) / 2.8 us (fp32): profiles/r03_xchg_probe.txt.
// Every member spins on its group's counter, so all workgroups of a launch must be resident: groups x G <= 256.
namespace {
constexpr int TR = 64;                     // rows per group
typedef __attribute__((ext_vector_type(4))) unsigned int tpu4;
template <int G> struct TP {
    static constexpr int C = SD / G;                 // columns per slice
    static constexpr int LR = 64 / G;                // 16-byte chunks (lanes) per slice row
    static constexpr int NT = G == 2 ? 2 : 1;        // column tiles per wave
    static constexpr int MT = G == 8 ? 2 : 4;        // row tiles per wave
    static constexpr int WS = G == 8 ? 4 : 8;        // weight streams per slice (G = 8: waves w and w + 4 share one)
    static constexpr int KST = NT * FRAG;            // uint4 per k-step of one stream
    // k-steps of weight fragments in flight per wave.  The stream is bound by latency x steps in flight, not by bytes, so the
    // narrower streams would want 16 / NT steps (the 64 VGPRs decoder_stack_kernel spends) — measured SLOWER (25.5 vs 23.9 ms per
    // 1 024-image decode): with the full residual rows in registers (64 VGPRs) the deeper ring spills
    static constexpr int DEP = G == 2 ? SPD : 8;     // (G = 4, 8 measured at 1 536 rows, ms per decode: 4 steps 10.65, 8: 10.38, 16: 14.6 — spills)
    static constexpr int SLICE = TR * C * 2;         // bytes of a bf16 slice image
    static constexpr int R2 = 3 * SLICE > 65536 ? 3 * SLICE : 65536;
    static constexpr size_t LDS = 65536 + R2;
};


// slice image [64 rows][LR chunks of 16 B]: chunk' = chunk ^ (row & min(15, LR - 1)) (the accumulator-layout stores of 16 rows
// of one column tile land on distinct chunks)
template <int G> __device__ __forceinline__ int sl_off(int row, int chunk) {
    constexpr int LR = 64 / G;
    return row * (LR * 16) + ((chunk ^ (row & (LR - 1) & 15)) << 4);
}
__device__ __forceinline__ tpu4 load_sc1(const void* p) {
    tpu4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// the compiler does not know that the registers above are in flight: the wait names them as in-out operands, so that no copy or
// use of them can be scheduled in front of it
__device__ __forceinline__ void wait_sc1(tpu4 (&v)[8]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
}
// Stores into an exchange tile through its buffer descriptor: plain (the line stays in this XCD's L2: valid only when every member of
// the group has been SEEN on this XCD, below) or write-through (sc1: valid at any placement — cdna_hip_programming.md Guideline 16 R1:
// sc1 payload stores, every storing wave's vmcnt(0), the workgroup barrier, one lane's agent-scope counter add; sc1 loads to registers).
typedef __amdgpu_buffer_rsrc_t tp_rsrc;
__device__ __forceinline__ void tp_store16(tp_rsrc rs, unsigned off, tpu4 v, bool wt) {
    if (wt) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
}
__device__ __forceinline__ void tp_store8(tp_rsrc rs, unsigned off, uint2 v, bool wt) {
    typedef __attribute__((ext_vector_type(2))) unsigned int tpu2;
    const tpu2 w = {v.x, v.y};
    if (wt) __builtin_amdgcn_raw_buffer_store_b64(w, rs, off, 0, 16);
    else __builtin_amdgcn_raw_buffer_store_b64(w, rs, off, 0, 0);
}
// sum over the 64 lanes by DPP (no LDS crossbar, no lgkmcnt wait): 16-lane rows, then the four row totals through readlane
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_quad1(v);
    v += dpp_quad2(v);
    v += dpp_half_mirror(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));      // row_mirror
    return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}
template <int G> struct RingTP { uint4 f[TP<G>::DEP][TP<G>::NT]; };
template <int G> __device__ __forceinline__ void ring_tp_start(RingTP<G>& r, const uint4* wp, int lane) {
#pragma unroll
    for (int s = 0; s < TP<G>::DEP; ++s)
#pragma unroll
        for (int nt = 0; nt < TP<G>::NT; ++nt) r.f[s][nt] = wp[s * TP<G>::KST + nt * FRAG + lane];
}
// acc[mt][nt] += A[16 (mt0 + mt) .. +15][:] . W[this wave's column tile nt][:]^T over the 512 inputs of one unit
template <int G, bool NEXT>
__device__ __forceinline__ void unit_tp(f32x4 (&acc)[TP<G>::MT][TP<G>::NT], const char* A, const uint4*& wp, RingTP<G>& r, int lane, int mt0) {
    constexpr int NT = TP<G>::NT, MT = TP<G>::MT, KST = TP<G>::KST, DEP = TP<G>::DEP, NIT = 16 / DEP;
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int s = 0; s < DEP; ++s) {
            const int ks = it * DEP + s;
            bf16x8 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const bf16x8*>(A + img_off(16 * (mt0 + mt) + m, 4 * ks + kg));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, r.f[s][nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, av[mt], acc[mt][nt], 0, 0, 0);
            }
            if (NEXT || it < NIT - 1) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) r.f[s][nt] = wp[(ks + DEP) * KST + nt * FRAG + lane];
            }
        }
    }
    wp += 16 * KST;
}
}  // namespace

// TRAIN: rows draw the dropout of the teacher-forced pass (StackArgs.drop_*); G >= 4 only.
template <int G, bool TRAIN>
__global__ __launch_bounds__(512) void decoder_stack_tp_kernel(StackArgs a) {
    using T_ = TP<G>;
    constexpr int C = T_::C, LR = T_::LR, NT = T_::NT, MT = T_::MT, KST = T_::KST;
    static_assert(!TRAIN || G >= 4, "train-mode rows: one row per lane group in both attention phases");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A0 = smem;                        // [64 x 512] bf16: LayerNorm output / gathered attention output
    char* QI = smem + 65536;                // this member's query / key / value slices of the current position
    char* KI = QI + T_::SLICE;
    char* VI = KI + T_::SLICE;
    char* Hh = QI;                          // [64 x 512] bf16 gathered FFN hidden chunk (the slices are dead by then)
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = lane0;
    const int U = 6 + 2 * a.NC;
    // exchanges of one launch: per layer o(self), wo, o(cross), co, the hidden units (all chunks at once), w2 (a.debug: phases
    // skipped for measurements)
    const int NX = a.L * (3 + ((a.debug & 1) ? 0 : 1) + ((a.debug & 2) ? 0 : 1) + ((a.debug & 4) ? 0 : 1));
    // members of a group: the workgroups x + 8 k of the launch — dealt to ONE XCD by the dispatcher as observed; that is a speed
    // assumption only: the XCC ids the members report decide the exchange form below.  (debug 16: consecutive workgroups instead,
    // i.e. members on DIFFERENT XCDs — the test of that decision)
    const bool scatter = (a.debug & 16) != 0;
    const int xcd = (int)(blockIdx.x & 7), kk = (int)(blockIdx.x >> 3);
    const int c = scatter ? (int)(blockIdx.x % G) : kk % G, grp = scatter ? (int)(blockIdx.x / G) : (kk / G) * 8 + xcd;
    if ((int)blockIdx.x >= a.tp_groups * G) {
        // L2 prefetcher of one XCD (as in decoder_stack_kernel): the members of an XCD's groups walk the G x WS weight streams in
        // step, so without it every fragment load is an L2 miss all of them wait for.  One dword per 128-byte line of the
        // unit's 512 KB, kept STACK_AHEAD units in front of the XCD's pace-maker (its first workgroup).
        constexpr int NS = G * T_::WS;                     // streams; unit chunk of a stream: 16 k-steps x KST uint4
        constexpr int LINES = 16 * KST * 16 / 128;         // 128-byte lines of one stream's unit chunk
        const int base = a.tp_launch * a.L * U;
        unsigned int sink = 0;
        int spin = 0;
        for (int u = 0; u < a.L * U; ++u) {
            for (; spin < 4096 && __hip_atomic_load(a.progress + xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < base + u + 1 - STACK_AHEAD; ++spin)
                __builtin_amdgcn_s_sleep(8);
            spin = spin >= 4096 ? 4096 : 0;
            for (int sidx = wave; sidx < NS; sidx += 8) {
                const unsigned int* src = reinterpret_cast<const unsigned int*>(a.tp_wpk + ((int64_t)sidx * a.L * U + u) * 16 * KST);
                for (int ln = lane; ln < LINES; ln += 64) sink += src[ln * 32];
            }
        }
        if (sink == 0x9E3779B1u) a.progress[8] = 1;        // (keeps the loads)
        return;
    }
    const bool pace = a.progress != nullptr && blockIdx.x < 8 && tid == 0;
    int unit_no = a.tp_launch * a.L * U;
    int32_t* flag = a.tp_flag + grp * TP_FLAG_STRIDE;
    const int r0 = grp * TR;
    if (r0 >= a.rows) {                      // no rows in this launch (the first beam pass): keep the group's counter in step
        if (tid == 0) __hip_atomic_fetch_add(flag, NX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    int xn = a.tp_launch * NX;              // exchanges this group has completed
    const int xn_first = xn;
    const size_t XTILE = (size_t)a.tp_xtile;      // bytes of one exchange tile: the fp32 [64 x 512] form or the NC hidden chunks
    char* xb = a.tp_xbuf + (size_t)grp * 2 * XTILE;
    const tp_rsrc xrs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)(2 * XTILE), 0x00020000);
    // ---- placement and residency.  (1) Every member reports the XCD it runs on (HW_REG_XCC_ID) into the group's mask word of this
    // launch before its first arrival; the FIRST exchange of a launch uses write-through stores, which are valid at any placement;
    // once it has completed, every member has reported, every thread reads the same final mask, and only a group whose members
    // all sit on ONE XCD goes on with plain stores (they stay in that XCD's L2, which its CUs share: 1.8 us per exchange instead
    // of ~3).  (2) The wait is bounded: a member that never arrives (the launch is not fully resident: another kernel holds CUs)
    // makes the others give up after `spin_max` polls, raise StackArgs.tp_status and run on WITHOUT waiting — the launch (and the
    // rest of the decode, which sees the word at entry) ends in bounded time with results the host side discards
    // (ortk_decode poisons its outputs, ortk_decode_status returns ORTK_EEXCHANGE).
    bool wt = true;                          // write-through exchange stores
    bool dead = false;                       // (thread 0) stop waiting: the error word is up
    const int spin_max = (a.debug & 32) ? (1 << 12) : (1 << 20);
    const bool absent = (a.debug & 32) && grp == 0 && c == G - 1;      // test: this member never arrives
    if (tid == 0) {
        const int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;       // hwreg(HW_REG_XCC_ID, 0, 4)
        __hip_atomic_fetch_or(flag + 16 + (a.tp_launch & 63), 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = absent || __hip_atomic_load(a.tp_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    }
    // publish this member's part of tile (xn & 1) and wait for the others'
#define TP_XWAIT()                                                                                              \
    do {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
        __syncthreads();                                                                                         \
        if (tid == 0) {                                                                                          \
            if (!absent) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            \
            for (int sp_ = 0; !dead && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (xn + 1) * G; ++sp_) { \
                __builtin_amdgcn_s_sleep(1);                                                                     \
                if (sp_ >= spin_max) {                                                                           \
                    dead = true;                                                                                 \
                    __hip_atomic_fetch_or(a.tp_status, TP_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                }                                                                                                \
            }                                                                                                    \
        }                                                                                                        \
        __syncthreads();                                                                                         \
        if (xn == xn_first) {                                                                                    \
            const int mk_ = __hip_atomic_load(flag + 16 + (a.tp_launch & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            wt = __builtin_amdgcn_readfirstlane(__popc(mk_)) != 1;                                               \
        }                                                                                                        \
        ++xn;                                                                                                    \
    } while (0)
#define TP_FRESH_LANE() do { lane = lane0; asm volatile("" : "+v"(lane)); } while (0)
#define TP_UNIT_BEGIN() do { ++unit_no; if (pace) __hip_atomic_store(a.progress + blockIdx.x, unit_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
    const int Lk = a.t + 1;
    const int mt0 = G == 8 ? 2 * (wave >> 2) : 0;
    const int ws = G == 8 ? (wave & 3) : wave;
    const int tile0 = G == 2 ? 2 * wave : ws;       // column tile (of the slice) of accumulator 0; accumulator nt: tile0 + nt
    // train-mode rows: the row of the teacher-forced pass whose draws decode row g takes, or -1 (an eval-mode row: the greedy baseline);
    // one table entry per row of the group, behind the images in LDS
    int* tfrow = reinterpret_cast<int*>(smem + T_::LDS);
    if constexpr (TRAIN) {
        if (tid < TR) {
            const int g = min(r0 + tid, a.rows - 1);
            int v = g;
            if (a.greedy_stride > 0) { const int q = g / a.greedy_stride, k = g - q * a.greedy_stride; v = k == 0 ? -1 : q * (a.greedy_stride - 1) + k - 1; }
            tfrow[tid] = v;
        }
        __syncthreads();
    }
    // (every teacher-forced element index of a launch this kernel serves — at most 8 192 rows, 64 positions, d_ff 4 096 — is below 2^32)
    const float ik = TRAIN ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thr = TRAIN ? ortk_keep_thr(a.drop_p) : 0u;
    const int ns_tf = a.per_img - (a.greedy_stride > 0 ? 1 : 0);       // captions per image in the teacher-forced pass

    // residual rows, row layout: wave w holds rows 8 w .. 8 w + 7, lane l columns 8 l .. 8 l + 7
    float x[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int g = min(r0 + 8 * wave + i, a.rows - 1);
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(a.x_io + (int64_t)g * SD + 8 * lane);
        const f32x4 u1 = *reinterpret_cast<const f32x4*>(a.x_io + (int64_t)g * SD + 8 * lane + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[i][j] = u0[j]; x[i][4 + j] = u1[j]; }
    }
    // LayerNorm of the 8 rows of this wave -> bf16 rows of an A image (wave reductions only)
    // (NOT wave_sum(): inside namespace ortk that name finds the stream builder's int overload first and silently truncates)
    auto ln_rows = [&](const float* ga, const float* be, char* img) {
#pragma clang fp contract(off)
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(ga + 8 * lane), g1 = *reinterpret_cast<const f32x4*>(ga + 8 * lane + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(be + 8 * lane), b1 = *reinterpret_cast<const f32x4*>(be + 8 * lane + 4);
        const float gv[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sm += x[i][j];
            const float mean = wave_sum_dpp(sm) * (1.f / SD);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = x[i][j] - mean; q = __builtin_fmaf(d, d, q); }
            const float rinv = 1.f / (sqrtf(wave_sum_dpp(q) * (1.f / (SD - 1))) + a.eps);
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = __builtin_fmaf(gv[j] * (x[i][j] - mean), rinv, bv[j]);
            const uint4 pk = make_uint4(pack2(y[0], y[1]), pack2(y[2], y[3]), pack2(y[4], y[5]), pack2(y[6], y[7]));
            *reinterpret_cast<uint4*>(img + img_off(8 * wave + i, lane)) = pk;
        }
    };
    // (acc + bias [relu]) of this wave's tiles as bf16 into a slice image
    auto store_slice = [&](char* img, const f32x4 (&acc)[MT][NT], const float* bias) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cl = 16 * (tile0 + nt) + 4 * (lane >> 4);              // column inside the slice
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c * C + cl);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = 16 * (mt0 + mt) + (lane & 15);
                *reinterpret_cast<uint2*>(img + sl_off<G>(row, cl >> 3) + ((cl >> 2) & 1) * 8) =
                    make_uint2(pack2(acc[mt][nt][0] + b4[0], acc[mt][nt][1] + b4[1]), pack2(acc[mt][nt][2] + b4[2], acc[mt][nt][3] + b4[3]));
            }
        }
    };
    // (acc + bias) of this wave's tiles into the exchange tile at byte offset `toff` of the group's buffer: fp32 (a residual update) or
    // relu + bf16 (an FFN hidden chunk).  TRAIN: the dropout of that output in the teacher-forced pass — element (row, coff + col) of
    // its (rows x T, ncols) tensor (the GEMM epilogues' index: ortk_gemm_args.drop_row_stride / _off)
    auto store_tile = [&](unsigned toff, const f32x4 (&acc)[MT][NT], const float* bias, bool hidden, uint32_t seed, int ncols, int coff) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = c * C + 16 * (tile0 + nt) + 4 * (lane >> 4);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = 16 * (mt0 + mt) + (lane & 15);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[mt][nt][r] + b4[r]; if (hidden) v[r] = fmaxf(v[r], 0.f); }
                if constexpr (TRAIN) {
                    const int mtf = tfrow[row];
                    if (mtf >= 0) {
                        bool kp[4];
                        ortk_keep4_u32(seed, ((uint32_t)mtf * (uint32_t)a.T + (uint32_t)a.t) * (uint32_t)ncols + (uint32_t)(coff + col), thr, kp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * ik : 0.f;
                    }
                }
                if (hidden) tp_store8(xrs, toff + (unsigned)(row * SD + col) * 2, make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3])), wt);
                else tp_store16(xrs, toff + (unsigned)(row * SD + col) * 4, (tpu4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, wt);
            }
        }
    };
    // the gathered bf16 tile -> an A image (8 chunks of 16 B per thread)
    auto gather_img = [&](const char* tile, char* img) {
        tpu4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = load_sc1(tile + (size_t)(tid + 512 * u) * 16);
        wait_sc1(v);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int q = tid + 512 * u; *reinterpret_cast<tpu4*>(img + img_off(q >> 6, q & 63)) = v[u]; }
    };
    // the gathered fp32 tile: x += update
    auto add_tile = [&](const char* tile) {
        tpu4 v0[8], v1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v0[i] = load_sc1(tile + ((size_t)(8 * wave + i) * SD + 8 * lane) * 4);
            v1[i] = load_sc1(tile + ((size_t)(8 * wave + i) * SD + 8 * lane + 4) * 4);
        }
        wait_sc1(v0); wait_sc1(v1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[i][j] += __uint_as_float(v0[i][j]); x[i][4 + j] += __uint_as_float(v1[i][j]); }
    };
    auto zero_acc = [&](f32x4 (&acc)[MT][NT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    const uint4* wp = a.tp_wpk + (int64_t)(c * T_::WS + ws) * a.L * U * 16 * KST;
    RingTP<G> ring;
    ring_tp_start<G>(ring, wp, lane);

    for (int l = 0; l < a.L; ++l) {
        const StackLayer& P = a.layer[l];
        f32x4 acc[MT][NT];
        // ---- LayerNorm 0 -> A0
        TP_FRESH_LANE();
        ln_rows(P.n0a, P.n0b, A0);
        __syncthreads();
        // ---- packed QKV: this member's columns of q, k, v -> slice images
        TP_FRESH_LANE();
        zero_acc(acc); TP_UNIT_BEGIN(); unit_tp<G, true>(acc, A0, wp, ring, lane, mt0); store_slice(QI, acc, P.bqkv);
        zero_acc(acc); TP_UNIT_BEGIN(); unit_tp<G, true>(acc, A0, wp, ring, lane, mt0); store_slice(KI, acc, P.bqkv + SD);
        zero_acc(acc); TP_UNIT_BEGIN(); unit_tp<G, false>(acc, A0, wp, ring, lane, mt0); store_slice(VI, acc, P.bqkv + 2 * SD);
        __syncthreads();
        // ---- self-attention over this member's heads: G rows side by side in a wave (LR lanes each); o -> the exchange tile
        TP_FRESH_LANE();
        if (!(a.debug & 1)) {
            const unsigned toff = (unsigned)((xn & 1) * XTILE);
            char* tile = xb + toff;
            const int rs = lane / LR, fc = lane % LR;
            const int hd = (c * LR + fc) >> 3;               // head of this lane's 8 features
            uint4* ck = reinterpret_cast<uint4*>(P.ck);
            uint4* cv = reinterpret_cast<uint4*>(P.cv);
#pragma unroll 1
            for (int p0 = 0; p0 < 8; p0 += G) {
                const int row = 8 * wave + p0 + rs;
                const int g = r0 + row, gc = min(g, a.rows - 1);
                // TRAIN: probability (row, head, key j) of the teacher-forced (rows, 8, T, T) tensor at query position t
                const int mrow = TRAIN ? tfrow[row] : -1;
                const uint32_t sbase = (((uint32_t)mrow * 8 + hd) * a.T + a.t) * a.T;
                int idx[G];                  // this lane: cache rows of keys fc + LR u of its row
#pragma unroll
                for (int u = 0; u < G; ++u) { const int j = min(fc + LR * u, a.t); idx[u] = a.kvidx ? a.kvidx[(int64_t)gc * Lk + j] : gc * a.T + j; }
                // cache row of (wave-uniform) key J for this lane's row: register J / LR of lane rs LR + J % LR
#define TP_KEY_ROW(J, OUT)                                                                   \
                do {                                                                          \
                    int v_ = idx[0];                                                          \
                    _Pragma("unroll") for (int u_ = 1; u_ < G; ++u_) v_ = ((J) / LR == u_) ? idx[u_] : v_; \
                    OUT = __shfl(v_, rs * LR + ((J) % LR), 64);                               \
                } while (0)
                AttState<1> st;
                st.q[0] = *reinterpret_cast<const uint4*>(QI + sl_off<G>(row, fc));
                st.m[0] = -INFINITY; st.l[0] = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) st.o[0][d] = 0.f;
                constexpr int SKT = G == 2 ? SKB : 4;      // keys per batch (measured at 1 536 rows, ms per decode: 2: 11.1, 3 / 4: 10.65, 6: 11.5, 9: 12.1 — registers)
                const int nb = (a.t + SKT - 1) / SKT;
                uint4 kq[SKT], vq[SKT];
#define TP_ISSUE(B)                                                                           \
                do {                                                                          \
                    _Pragma("unroll") for (int u_ = 0; u_ < SKT; ++u_) {                      \
                        const int j_ = min((B) * SKT + u_, max(a.t - 1, 0));                  \
                        int kr_; TP_KEY_ROW(j_, kr_);                                         \
                        const int64_t ko_ = (int64_t)kr_ * (SD / 8) + c * LR + fc;            \
                        kq[u_] = ck[ko_]; vq[u_] = cv[ko_];                                   \
                    }                                                                         \
                } while (0)
                if (nb > 0) TP_ISSUE(0);
                for (int b = 0; b < nb; ++b) {
                    float kind[SKT], pr[1][SKT];
#pragma unroll
                    for (int u = 0; u < SKT; ++u) kind[u] = b * SKT + u < a.t ? 0.f : -INFINITY;
                    uint4 kc[SKT], vc[SKT];
#pragma unroll
                    for (int u = 0; u < SKT; ++u) { kc[u] = kq[u]; vc[u] = vq[u]; }
                    if (b + 1 < nb) TP_ISSUE(b + 1);
                    st.scores<SKT>(kc, kind, pr);
                    if constexpr (TRAIN) {
                        if (mrow >= 0) {
#pragma unroll
                            for (int u = 0; u < SKT; ++u) pr[0][u] = ortk_keep_u32(a.drop_seed[l][0], sbase + b * SKT + u, thr) ? pr[0][u] * ik : 0.f;
                        }
                    }
                    st.pv<SKT>(vc, pr);
                }
                const uint4 kself[1] = {*reinterpret_cast<const uint4*>(KI + sl_off<G>(row, fc))};
                const uint4 vself[1] = {*reinterpret_cast<const uint4*>(VI + sl_off<G>(row, fc))};
                const float kindself[1] = {0.f};
                float pself[1][1];
                st.scores<1>(kself, kindself, pself);
                if constexpr (TRAIN) {
                    if (mrow >= 0) pself[0][0] = ortk_keep_u32(a.drop_seed[l][0], sbase + a.t, thr) ? pself[0][0] * ik : 0.f;
                }
                st.pv<1>(vself, pself);
                int srow_; TP_KEY_ROW(a.t, srow_);
                const int64_t slot = (int64_t)srow_ * (SD / 8) + c * LR + fc;
                if (g < a.rows) { ck[slot] = kself[0]; cv[slot] = vself[0]; }
                {
#pragma clang fp contract(off)
                    const float inv = 1.f / st.l[0];
                    tp_store16(xrs, toff + (unsigned)(row * 64 + c * LR + fc) * 16,
                               (tpu4){pack2(st.o[0][0] * inv, st.o[0][1] * inv), pack2(st.o[0][2] * inv, st.o[0][3] * inv),
                                      pack2(st.o[0][4] * inv, st.o[0][5] * inv), pack2(st.o[0][6] * inv, st.o[0][7] * inv)}, wt);
                }
            }
            TP_XWAIT();
            gather_img(tile, A0);
        }
        __syncthreads();
        // ---- output projection: residual update through the exchange, LayerNorm 1 -> A0
        TP_FRESH_LANE();
        {
            const unsigned toff = (unsigned)((xn & 1) * XTILE);
            ring_tp_start<G>(ring, wp, lane);
            zero_acc(acc); TP_UNIT_BEGIN(); unit_tp<G, true>(acc, A0, wp, ring, lane, mt0);
            store_tile(toff, acc, P.bo, false, a.drop_seed[l][1], SD, 0);
            TP_XWAIT();
            add_tile(xb + toff);
        }
        ln_rows(P.n1a, P.n1b, A0);         // (every wave is past its reads of A0: the exchange's barriers)
        __syncthreads();
        // ---- cross-attention query -> QI
        TP_FRESH_LANE();
        zero_acc(acc); TP_UNIT_BEGIN(); unit_tp<G, false>(acc, A0, wp, ring, lane, mt0); store_slice(QI, acc, P.cqb);
        __syncthreads();
        // ---- cross-attention: chunks of up to XNR rows of one image share their key / value loads; one chunk per LR-lane group
        TP_FRESH_LANE();
        if (!(a.debug & 2) && G >= 4) {
            // narrow slices (16 / 8 lanes per row): one row per lane group and all 64 rows of the group in one or two passes — the
            // shared-load form below leaves two thirds of the lane groups idle there and runs five rows' state per lane
            const unsigned toff = (unsigned)((xn & 1) * XTILE);
            char* tile = xb + toff;
            const int rs = lane / LR, fc = lane % LR;
            const int hd = (c * LR + fc) >> 3;
            constexpr int XKT = 4;
            const int64_t pitch = a.ldx / 8;
#pragma unroll 1
            for (int p0 = 0; p0 < 8; p0 += G) {
                const int row = 8 * wave + p0 + rs;
                const int gc = min(r0 + row, a.rows - 1), imm = gc / a.per_img;
                // TRAIN: probability (image, head, caption i, position t, region j) of the teacher-forced (images, 8, ns T, S) tensor; an
                // eval-mode (greedy) row attends to the EVAL-mode memory's projection
                const int mrow = TRAIN ? tfrow[row] : -1;
                const uint32_t cbase = ((((uint32_t)(mrow / ns_tf) * 8 + hd) * ns_tf + (uint32_t)(mrow % ns_tf)) * a.T + a.t) * a.S;
                const __bf16* xkb = (TRAIN && mrow < 0 && P.xkg) ? P.xkg : P.xk;
                const __bf16* xvb = (TRAIN && mrow < 0 && P.xvg) ? P.xvg : P.xv;
                AttState<1> st;
                st.q[0] = *reinterpret_cast<const uint4*>(QI + sl_off<G>(row, fc));
                st.m[0] = -INFINITY; st.l[0] = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) st.o[0][d] = 0.f;
                const uint4* xk = reinterpret_cast<const uint4*>(xkb) + (int64_t)imm * a.S * pitch + c * LR + fc;
                const uint4* xv = reinterpret_cast<const uint4*>(xvb) + (int64_t)imm * a.S * pitch + c * LR + fc;
                const float* mk = a.att_masks + (int64_t)imm * a.S;
                const int nb = (a.S + XKT - 1) / XKT;
                uint4 kq[XKT], vq[XKT];
                float mq[XKT];
#pragma unroll
                for (int u = 0; u < XKT; ++u) { const int j = min(u, a.S - 1); kq[u] = xk[j * pitch]; vq[u] = xv[j * pitch]; mq[u] = mk[j]; }
                for (int b = 0; b < nb; ++b) {
                    float kind[XKT], pr[1][XKT];
#pragma unroll
                    for (int u = 0; u < XKT; ++u) kind[u] = b * XKT + u >= a.S ? -INFINITY : (mq[u] == 0.f ? -1e9f : 0.f);
                    st.scores<XKT>(kq, kind, pr);
                    if (b + 1 < nb) {
#pragma unroll
                        for (int u = 0; u < XKT; ++u) { const int j = min((b + 1) * XKT + u, a.S - 1); kq[u] = xk[j * pitch]; mq[u] = mk[j]; }
                    }
                    if constexpr (TRAIN) {
                        if (mrow >= 0) {
                            if (XKT == 4 && (a.S & 3) == 0) {        // (uniform) the batch is one aligned group of four: one hash evaluation
                                bool kp[4];
                                ortk_keep4_u32(a.drop_seed[l][2], cbase + b * XKT, thr, kp);
#pragma unroll
                                for (int u = 0; u < XKT; ++u) pr[0][u] = kp[u & 3] ? pr[0][u] * ik : 0.f;
                            } else {
#pragma unroll
                                for (int u = 0; u < XKT; ++u) pr[0][u] = ortk_keep_u32(a.drop_seed[l][2], cbase + b * XKT + u, thr) ? pr[0][u] * ik : 0.f;
                            }
                        }
                    }
                    st.pv<XKT>(vq, pr);
                    if (b + 1 < nb) {
#pragma unroll
                        for (int u = 0; u < XKT; ++u) vq[u] = xv[min((b + 1) * XKT + u, a.S - 1) * pitch];
                    }
                }
                {
#pragma clang fp contract(off)
                    const float inv = 1.f / st.l[0];
                    tp_store16(xrs, toff + (unsigned)(row * 64 + c * LR + fc) * 16,
                               (tpu4){pack2(st.o[0][0] * inv, st.o[0][1] * inv), pack2(st.o[0][2] * inv, st.o[0][3] * inv),
                                      pack2(st.o[0][4] * inv, st.o[0][5] * inv), pack2(st.o[0][6] * inv, st.o[0][7] * inv)}, wt);
                }
            }
            TP_XWAIT();
            gather_img(tile, A0);
        } else if (!(a.debug & 2)) {
            const unsigned toff = (unsigned)((xn & 1) * XTILE);
            char* tile = xb + toff;
            const int rs = lane / LR, fc = lane % LR;
            const int last = min(r0 + TR, a.rows) - 1;
            const int img0 = r0 / a.per_img, img1 = last / a.per_img;
            int nch = 0;
            for (int im = img0; im <= img1; ++im) {
                const int lo = max(im * a.per_img, r0), hi = min((im + 1) * a.per_img - 1, last);
                nch += (hi - lo) / XNR + 1;
            }
            for (int pass = 0; pass * 8 * G < nch; ++pass) {
                const int mine = pass * 8 * G + wave * G + rs;
                int c0 = r0, nr = 0, imm = img0, k = 0;
                for (int im = img0; im <= img1; ++im) {
                    const int lo = max(im * a.per_img, r0), hi = min((im + 1) * a.per_img - 1, last);
                    for (int cc = lo; cc <= hi; cc += XNR, ++k)
                        if (k == mine) { c0 = cc; nr = min(XNR, hi - cc + 1); imm = im; }
                }
                AttState<XNR> st;
#pragma unroll
                for (int i = 0; i < XNR; ++i) {
                    st.q[i] = *reinterpret_cast<const uint4*>(QI + sl_off<G>(c0 - r0 + (i < nr ? i : 0), fc));
                    st.m[i] = -INFINITY; st.l[i] = 0.f;
#pragma unroll
                    for (int d = 0; d < 8; ++d) st.o[i][d] = 0.f;
                }
                const int64_t pitch = a.ldx / 8;
                const uint4* xk = reinterpret_cast<const uint4*>(P.xk) + (int64_t)imm * a.S * pitch + c * LR + fc;
                const uint4* xv = reinterpret_cast<const uint4*>(P.xv) + (int64_t)imm * a.S * pitch + c * LR + fc;
                const float* mk = a.att_masks + (int64_t)imm * a.S;
                const int nb = (a.S + XKB - 1) / XKB;
                uint4 kq[XKB], vq[XKB];
                float mq[XKB];
#pragma unroll
                for (int u = 0; u < XKB; ++u) { const int j = min(u, a.S - 1); kq[u] = xk[j * pitch]; vq[u] = xv[j * pitch]; mq[u] = mk[j]; }
                for (int b = 0; b < nb; ++b) {
                    float kind[XKB], pr[XNR][XKB];
#pragma unroll
                    for (int u = 0; u < XKB; ++u) kind[u] = b * XKB + u >= a.S ? -INFINITY : (mq[u] == 0.f ? -1e9f : 0.f);
                    st.scores<XKB>(kq, kind, pr);
                    if (b + 1 < nb) {
#pragma unroll
                        for (int u = 0; u < XKB; ++u) { const int j = min((b + 1) * XKB + u, a.S - 1); kq[u] = xk[j * pitch]; mq[u] = mk[j]; }
                    }
                    st.pv<XKB>(vq, pr);
                    if (b + 1 < nb) {
#pragma unroll
                        for (int u = 0; u < XKB; ++u) vq[u] = xv[min((b + 1) * XKB + u, a.S - 1) * pitch];
                    }
                }
#pragma unroll
                for (int i = 0; i < XNR; ++i) {
#pragma clang fp contract(off)
                    if (i < nr) {
                        const float inv = 1.f / st.l[i];
                        tp_store16(xrs, toff + (unsigned)((c0 - r0 + i) * 64 + c * LR + fc) * 16,
                                   (tpu4){pack2(st.o[i][0] * inv, st.o[i][1] * inv), pack2(st.o[i][2] * inv, st.o[i][3] * inv),
                                          pack2(st.o[i][4] * inv, st.o[i][5] * inv), pack2(st.o[i][6] * inv, st.o[i][7] * inv)}, wt);
                    }
                }
            }
            TP_XWAIT();
            gather_img(tile, A0);
        }
        __syncthreads();
        // ---- output projection, LayerNorm 2 -> A0
        TP_FRESH_LANE();
        {
            const unsigned toff = (unsigned)((xn & 1) * XTILE);
            ring_tp_start<G>(ring, wp, lane);
            zero_acc(acc); TP_UNIT_BEGIN(); unit_tp<G, true>(acc, A0, wp, ring, lane, mt0);
            store_tile(toff, acc, P.cob, false, a.drop_seed[l][3], SD, 0);
            TP_XWAIT();
            add_tile(xb + toff);
        }
        ln_rows(P.n2a, P.n2b, A0);
        __syncthreads();
        // ---- FFN, 512 hidden units at a time: this member's columns of h_c = relu(y W1_c^T + b1_c) -> exchange -> Hh;
        //      acc2 += Hh W2[this member's rows, chunk c]^T
        TP_FRESH_LANE();
        f32x4 acc2[MT][NT];
        zero_acc(acc2);
        if (!(a.debug & 4)) {
            // all NC up-projection units first (the stream holds them back to back), ONE exchange of the [64 x NC 512] hidden
            // units, then the NC down-projection units, each on its gathered chunk
            const unsigned toff = (unsigned)((xn & 1) * XTILE);
            char* tile = xb + toff;
            for (int cc = 0; cc < a.NC; ++cc) {
                zero_acc(acc); TP_UNIT_BEGIN(); unit_tp<G, true>(acc, A0, wp, ring, lane, mt0);
                store_tile(toff + (unsigned)cc * (TR * SD * 2), acc, P.b1 + cc * SD, true, a.drop_seed[l][4], a.NC * SD, cc * SD);
            }
            TP_XWAIT();
            for (int cc = 0; cc < a.NC; ++cc) {
                gather_img(tile + (size_t)cc * (TR * SD * 2), Hh);
                __syncthreads();
                TP_UNIT_BEGIN(); unit_tp<G, true>(acc2, Hh, wp, ring, lane, mt0);
                __syncthreads();                     // every wave is past its reads of Hh
            }
        }
        {
            const unsigned toff = (unsigned)((xn & 1) * XTILE);
            store_tile(toff, acc2, P.b2, false, a.drop_seed[l][5], SD, 0);
            TP_XWAIT();
            add_tile(xb + toff);
        }
    }
    // ---- final LayerNorm -> bf16 rows for the generator (through A0; member 0 writes them out)
    TP_FRESH_LANE();
    ln_rows(a.fa, a.fb, A0);
    __syncthreads();
    if (c == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = tid + 512 * u, row = q >> 6, ch = q & 63;
            if (r0 + row < a.rows) *reinterpret_cast<uint4*>(a.y_out + (int64_t)(r0 + row) * SD + 8 * ch) = *reinterpret_cast<const uint4*>(A0 + img_off(row, ch));
        }
    }
#undef TP_XWAIT
#undef TP_FRESH_LANE
#undef TP_UNIT_BEGIN
#undef TP_KEY_ROW
#undef TP_ISSUE
}

// tp_wpk[((((c WS + ws) L + l) U + u) 16 + ks) NT + nt) 64 + lane] = the 8 bf16 W_u[row][32 ks + 8 (lane >> 4) ..] with
// row = c C + 16 tile + (lane & 15) of the unit's 512 output rows (tile = NT ws + nt)
template <int G>
__global__ __launch_bounds__(256) void stack_tp_pack_kernel(const __bf16* __restrict__ w16, uint4* __restrict__ wpk, StackPack t) {
    constexpr int NT = TP<G>::NT, WS = TP<G>::WS, C = TP<G>::C;
    const int U = 6 + 2 * t.NC;
    const int64_t total = (int64_t)G * WS * t.L * U * 16 * NT * 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        int64_t rest = i >> 6;
        const int nt = (int)(rest % NT); rest /= NT;
        const int ks = (int)(rest & 15); rest >>= 4;
        const int u = (int)(rest % U); rest /= U;
        const int l = (int)(rest % t.L); rest /= t.L;
        const int ws = (int)(rest % WS);
        const int c = (int)(rest / WS);
        int64_t base; int ld;
        if (u < 3)       { base = t.off[l][0] + (int64_t)u * SD * SD; ld = SD; }
        else if (u < 6)  { base = t.off[l][u - 2]; ld = SD; }
        else if (u - 6 < t.NC) { const int cc = u - 6;        base = t.off[l][4] + (int64_t)cc * SD * SD; ld = SD; }           // W1 rows 512 cc .. (all chunks first)
        else                   { const int cc = u - 6 - t.NC; base = t.off[l][5] + (int64_t)cc * SD; ld = t.NC * SD; }          // W2 columns 512 cc ..
        const int row = c * C + 16 * (NT * ws + nt) + (lane & 15);
        wpk[i] = *reinterpret_cast<const uint4*>(w16 + base + (int64_t)row * ld + 32 * ks + 8 * (lane >> 4));
    }
}

// compute units of the current device: every workgroup of a column-split launch must be resident (one per CU: > 80 KB of LDS)
static int tp_cu_budget() {
    static std::mutex mu;
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> g(mu);
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        // Only the unpartitioned MI355X (256 CUs in 8 XCDs, workgroup i on XCD i % 8) is served: the exchange is coherent
        // inside ONE XCD's L2, and the members of a group are placed by that mapping.  A partitioned device (DPX / QPX / CPX:
        // 128 / 64 / 32 CUs, another workgroup -> XCD mapping) runs the other executors.
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount == 256) ? 256 : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}
int stack_tp_degree(int64_t rows) {
    const int groups = stack_tp_groups(rows), budget = tp_cu_budget();
    for (int G = 8; G >= 2; G >>= 1)
        if (groups * G <= budget) return G;
    return 0;
}
size_t stack_tp_packed_bytes(int L, int NC, int G) {
    const int NT = G == 2 ? 2 : 1, WS = G == 8 ? 4 : 8;
    return ((size_t)G * WS * L * (6 + 2 * NC) * 16 * NT * 64 + (size_t)16 * NT * 64) * sizeof(uint4);   // + the ring's read-ahead
}
int stack_tp_pack(const void* w16, void* wpk, const StackPack& t, int G, hipStream_t s) {
    const __bf16* w = reinterpret_cast<const __bf16*>(w16);
    uint4* o = reinterpret_cast<uint4*>(wpk);
    if (G == 2) hipLaunchKernelGGL(stack_tp_pack_kernel<2>, dim3(2048), dim3(256), 0, s, w, o, t);
    else if (G == 4) hipLaunchKernelGGL(stack_tp_pack_kernel<4>, dim3(2048), dim3(256), 0, s, w, o, t);
    else if (G == 8) hipLaunchKernelGGL(stack_tp_pack_kernel<8>, dim3(2048), dim3(256), 0, s, w, o, t);
    else return ORTK_EINVAL;
    ORTK_CHECK_LAUNCH();
    return 0;
}

template <int G, bool TRAIN>
static int stack_tp_launch(const StackArgs& b, hipStream_t s) {
    constexpr size_t lds = TP<G>::LDS + (TRAIN ? TR * sizeof(int) : 0);        // (+ the teacher-forced row table)
    static_assert(lds <= 160 * 1024, "LDS budget");
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ORTK_EINVAL;
    {
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_stack_tp_kernel<G, TRAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            done[dev] = true;
        }
    }
    // (the pace-makers are the first member of groups 0..7: they must have rows in this launch)
    const bool pf = b.progress != nullptr && b.tp_groups * G + 8 <= tp_cu_budget() && !(b.debug & 8) && !(b.debug & 16) && b.rows > 7 * TR;
    StackArgs c = b;
    if (!pf) c.progress = nullptr;
    hipLaunchKernelGGL((decoder_stack_tp_kernel<G, TRAIN>), dim3((unsigned)(b.tp_groups * G + (pf ? 8 : 0))), dim3(512), lds, s, c);
    return 0;
}

template <bool SPARSE, int RB, bool GATHER = false>
static int stack_launch(const StackArgs& b, bool pf, hipStream_t s) {
    constexpr size_t lds = (size_t)4 * RB * SD * 2 + 2 * 32 * 8 * sizeof(float) + (GATHER ? 2 * GPL + 8 * GTB : SPARSE ? 8 * SDBUF : 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    // (per device: the attribute belongs to the function on ONE device; a process that drives several GPUs sets it on each)
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ORTK_EINVAL;
    {
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_stack_kernel<SPARSE, RB, GATHER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            done[dev] = true;
        }
    }
    hipLaunchKernelGGL((decoder_stack_kernel<SPARSE, RB, GATHER>), dim3((unsigned)(b.nblocks + (pf ? 8 : 0))), dim3(512), lds, s, b);
    return 0;
}

int stack_step(const StackArgs& a, hipStream_t s) {
    if (a.rows < 1 || a.S < 1 || a.S > 128 || a.t < 0 || a.t >= 64 || a.per_img < 1 || a.L < 1 || a.L > STACK_MAXL || a.NC < 1) return ORTK_EINVAL;
    const bool sparse = a.sstream != nullptr;
    // (the sparse stream's fragment buffers leave LDS for 20-row images.  Whole-image 30-row blocks — with 5 beams a 32-row block cuts
    // every third image in two and both workgroups read its projected memory — were measured SLOWER at 1 024 x 5 rows, 18.69 vs 18.50
    // ms per decode, same tokens: 171 instead of 160 workgroups stream the weights out of the L2s, and that stream is the bound)
    const int rb = (sparse || a.rb == 20) ? 20 : 32;
    StackArgs b = a;
    b.nblocks = (int)ortk_cdiv(a.rows, rb);
    const bool pf = !sparse && b.nblocks >= 8 && b.nblocks + 8 <= 256 && a.progress != nullptr && !(a.debug & 8);
    if (!pf && !a.tp) b.progress = nullptr;
    // algorithmic bytes of the launch: the weights once (sparse stream: 4 bytes per non-zero), every image's projected
    // memory (K and V) once per layer, every row's cached keys and values once per layer plus the appended position, the
    // residual rows in and the normalised rows out
    ProfMark pm;
    if (ortk_prof_active()) {
        const double U = 6 + 2 * a.NC, imgs = (double)ortk_cdiv(a.rows, a.per_img);
        const double wbytes = U * SD * SD * (sparse ? 0.05 * 4.0 : 2.0);       // (sparse: priced at the 95 % of BASELINE configs[4])
        // cached keys / values: every row's t positions — or, with a counter of the UNIQUE rows the beams of this pass reference
        // (StackArgs.uniq_slot: beams share ancestors through the ancestry table), that count — plus the appended position per row
        const bool uq = a.uniq_slot > 0 && a.kvidx;
        const double cached = uq ? (double)a.rows : (double)a.rows * (a.t + 1);
        const double bytes = a.L * (wbytes + imgs * a.S * 2.0 * SD * 2 + cached * 2.0 * SD * 2) + (double)a.rows * SD * (4 + 2);
        (void)prof_begin(PROF_KEY_DECSTACK, 2.0 * a.rows * a.L * U * SD * SD, bytes, s, pm);
        if (uq) { pm.slot = a.uniq_slot - 1; pm.per_count = a.L * 2.0 * SD * 2; }
    } else pm.live = false;
    int rc;
    if (a.tp) {
        if (sparse || !a.tp_wpk || !a.tp_xbuf || !a.tp_flag || !a.tp_status || a.tp_groups < 8 || a.tp_groups % 8 || a.tp_groups * a.tp > tp_cu_budget() ||
            (int64_t)a.tp_groups * TR < a.rows || a.tp_launch < 0 || a.tp_launch >= 64) return ORTK_EINVAL;
        const bool train = a.drop_p > 0.f;
        if (train && (a.tp < 4 || a.drop_p >= 1.f || (a.greedy_stride > 0 && a.greedy_stride != a.per_img))) return ORTK_EINVAL;
        if (train) rc = a.tp == 4 ? stack_tp_launch<4, true>(b, s) : a.tp == 8 ? stack_tp_launch<8, true>(b, s) : ORTK_EINVAL;
        else rc = a.tp == 2 ? stack_tp_launch<2, false>(b, s) : a.tp == 4 ? stack_tp_launch<4, false>(b, s) : a.tp == 8 ? stack_tp_launch<8, false>(b, s) : ORTK_EINVAL;
    }
    else if (a.drop_p > 0.f) return ORTK_EINVAL;             // (train-mode rows: the column-split form only)
    else if (sparse && a.gather) rc = stack_launch<true, 20, true>(b, pf, s);
    else if (sparse) rc = stack_launch<true, 20>(b, pf, s);
    else if (rb == 20) rc = stack_launch<false, 20>(b, pf, s);
    else rc = stack_launch<false, 32>(b, pf, s);
    prof_end(pm, s);
    if (rc) return rc;
    ORTK_CHECK_LAUNCH();
    return 0;
}

}  // namespace ortk

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

namespace ortk {
namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int SD = 512;            // d_model
constexpr int SRB = 32;            // rows per workgroup
constexpr int SPD = 4;             // k-steps of weight fragments in flight per wave
constexpr int SIMG = SRB * SD * 2; // bytes of one bf16 A image
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
__device__ __forceinline__ void store_img(char* img, const f32x4 (&acc)[2][4], const f32x4 (&bias)[4], bool relu, int wave, int lane) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int row = 16 * mt + (lane & 15), col = 64 * wave + 16 * nt + 4 * (lane >> 4);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mt][nt][r] + bias[nt][r]; if (relu) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<uint2*>(img + img_off(row, col >> 3) + ((col >> 2) & 1) * 8) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
        }
}

// LayerNorm of the register-resident rows (transformer.py:338-341: a (x - mean) / (std_unbiased + eps) + b), two exchanges
// of per-wave partial sums through LDS.  The caller puts a barrier between the result and its consumers.
__device__ __forceinline__ void layer_norm(const f32x4 (&x)[2][4], const float* ga, const float* be, float eps, float* red1, float* red2,
                                           int wave, int lane, f32x4 (&y)[2][4]) {
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
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * 8), p1 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * 8 + 4);
        mean[mt] = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / SD);
        float q = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = x[mt][nt][r] - mean[mt]; q += d * d; }
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        if (lane < 16) red2[(16 * mt + m) * 8 + wave] = q;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * 8), p1 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * 8 + 4);
        const float var = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / (SD - 1));
        rinv[mt] = 1.f / (sqrtf(var) + eps);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[mt][nt][r] = a4[nt][r] * (x[mt][nt][r] - mean[mt]) * rinv[mt] + b4[nt][r];
    }
}
__device__ __forceinline__ void store_img_plain(char* img, const f32x4 (&y)[2][4], int wave, int lane) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int row = 16 * mt + (lane & 15), col = 64 * wave + 16 * nt + 4 * (lane >> 4);
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
    template <int KB>
    __device__ __forceinline__ void scores(const uint4 (&kk)[KB], const float (&kind)[KB], float (&p)[NR][KB]) {
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
            l[i] = l[i] * corr + s;
            m[i] = mn;
#pragma unroll
            for (int d = 0; d < 8; ++d) o[i][d] *= corr;
        }
    }
    template <int KB>
    __device__ __forceinline__ void pv(const uint4 (&vv)[KB], const float (&p)[NR][KB]) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const float v[8] = {lo_f(vv[u].x), hi_f(vv[u].x), lo_f(vv[u].y), hi_f(vv[u].y), lo_f(vv[u].z), hi_f(vv[u].z), lo_f(vv[u].w), hi_f(vv[u].w)};
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int d = 0; d < 8; ++d) o[i][d] += p[i][u] * v[d];
        }
    }
    __device__ __forceinline__ void finish(char* O, int row0, int nr, int lane) {
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
__global__ __launch_bounds__(512) void decoder_stack_kernel(StackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A0 = smem;                  // LayerNorm output / attention output: the A operand of the next projection
    char* A1 = smem + SIMG;           // query image; FFN hidden chunk (even)
    char* KN = smem + 2 * SIMG;       // this position's K; FFN hidden chunk (odd)
    char* VN = smem + 3 * SIMG;       // this position's V
    float* red1 = reinterpret_cast<float*>(smem + 4 * SIMG);
    float* red2 = red1 + SRB * 8;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = lane0;
    const int U = 6 + 2 * a.NC;
    if ((int)blockIdx.x >= a.nblocks) {
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
    const bool pace = blockIdx.x < 8 && tid == 0;
    int unit_no = a.t * a.L * U;
// a fresh, opaque copy of the lane id per phase: without it the compiler hoists every lane-derived address of every phase
    // out of the layer loop and keeps ~60 of them alive (spilled) through the whole kernel
#define STACK_FRESH_LANE() do { lane = lane0; asm volatile("" : "+v"(lane)); } while (0)
#define STACK_UNIT_BEGIN() do { ++unit_no; if (pace) __hip_atomic_store(a.progress + blockIdx.x, unit_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
    const int r0 = blockIdx.x * SRB;
    const int m = lane & 15, q4 = lane >> 4;
    const int Lk = a.t + 1;

    // residual stream of the block's rows, accumulator layout
    f32x4 x[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int g = min(r0 + 16 * mt + m, a.rows - 1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) x[mt][nt] = *reinterpret_cast<const f32x4*>(a.x_io + (int64_t)g * SD + 64 * wave + 16 * nt + 4 * q4);
    }
    const uint4* wp = a.wpk + (int64_t)wave * a.L * U * 16 * KSTEP;
    Ring ring;
    ring_start(ring, wp, lane);

    for (int l = 0; l < a.L; ++l) {
        const StackLayer& P = a.layer[l];
        f32x4 acc[2][4], y[2][4], bias[4];
        // ---- LayerNorm 0 -> A0
        STACK_FRESH_LANE();
        layer_norm(x, P.n0a, P.n0b, a.eps, red1, red2, wave, lane, y);
        store_img_plain(A0, y, wave, lane);
        __syncthreads();
        // ---- packed QKV: three units -> q (A1), k (KN), v (VN)
        STACK_FRESH_LANE();
        load_cols(P.bqkv, wave, lane, bias); zero(acc); STACK_UNIT_BEGIN(); unit_gemm<true>(acc, A0, wp, ring, lane); store_img(A1, acc, bias, false, wave, lane);
        load_cols(P.bqkv + SD, wave, lane, bias); zero(acc); STACK_UNIT_BEGIN(); unit_gemm<true>(acc, A0, wp, ring, lane); store_img(KN, acc, bias, false, wave, lane);
        load_cols(P.bqkv + 2 * SD, wave, lane, bias); zero(acc); STACK_UNIT_BEGIN(); unit_gemm<false>(acc, A0, wp, ring, lane); store_img(VN, acc, bias, false, wave, lane);
        __syncthreads();
        // ---- self-attention of rows 4 wave .. 4 wave + 3 over their cache rows + this position; o -> A0
        STACK_FRESH_LANE();
        if (!(a.debug & 1)) {
            // The wave's SROWS rows run in lock-step, each with its own K / V registers: SROWS x SKB row loads of K and as
            // many of V are in flight, K of the next batch requested as soon as the scores have consumed this one's, V as
            // soon as p . V has.  The phase is bound by the bytes a wave can keep in flight (registers), not by its arithmetic.
            const uint4* ck = reinterpret_cast<const uint4*>(P.ck) + lane;
            const uint4* cv = reinterpret_cast<const uint4*>(P.cv) + lane;
            const int nb = (a.t + SKB - 1) / SKB;
#pragma unroll 1
            for (int i0 = 0; i0 < 4; i0 += SROWS) {
                int idx[SROWS];          // lane j: physical cache row of key j of row r
                AttState<1> st[SROWS];
                uint4 kq[SROWS][SKB], vq[SROWS][SKB];
#pragma unroll
                for (int r = 0; r < SROWS; ++r) {
                    const int g = min(r0 + 4 * wave + i0 + r, a.rows - 1);
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
                for (int r = 0; r < SROWS; ++r) st[r].init(A1, 4 * wave + i0 + r, 1, lane);
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
                    const int row = 4 * wave + i0 + r;
                    const uint4 kself[1] = {*reinterpret_cast<const uint4*>(KN + img_off(row, lane))};
                    const uint4 vself[1] = {*reinterpret_cast<const uint4*>(VN + img_off(row, lane))};
                    const float kindself[1] = {0.f};
                    float pself[1][1];
                    st[r].scores<1>(kself, kindself, pself);
                    st[r].pv<1>(vself, pself);
                    if (r0 + row < a.rows) {
                        const int64_t slot = (int64_t)__builtin_amdgcn_readlane(idx[r], a.t) * (SD / 8);
                        reinterpret_cast<uint4*>(P.ck)[slot + lane] = kself[0];
                        reinterpret_cast<uint4*>(P.cv)[slot + lane] = vself[0];
                    }
                    st[r].finish(A0, row, 1, lane);
                }
            }
        }
        __syncthreads();
        // ---- output projection + residual, LayerNorm 1 -> A0
        STACK_FRESH_LANE();
        ring_start(ring, wp, lane);
        load_cols(P.bo, wave, lane, bias); zero(acc); STACK_UNIT_BEGIN(); unit_gemm<true>(acc, A0, wp, ring, lane);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) x[mt][nt] += acc[mt][nt] + bias[nt];
        layer_norm(x, P.n1a, P.n1b, a.eps, red1, red2, wave, lane, y);       // (its barriers: every wave is done reading A0)
        store_img_plain(A0, y, wave, lane);
        __syncthreads();
        // ---- cross-attention query -> A1
        STACK_FRESH_LANE();
        load_cols(P.cqb, wave, lane, bias); zero(acc); STACK_UNIT_BEGIN(); unit_gemm<false>(acc, A0, wp, ring, lane); store_img(A1, acc, bias, false, wave, lane);
        __syncthreads();
        // ---- cross-attention: chunks of up to XNR rows of one image, dealt round-robin to the waves; o -> A0
        STACK_FRESH_LANE();
        if (!(a.debug & 2)) {
            const int last = min(r0 + SRB, a.rows) - 1;
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
        __syncthreads();
        // ---- output projection + residual, LayerNorm 2 -> A0
        STACK_FRESH_LANE();
        ring_start(ring, wp, lane);
        load_cols(P.cob, wave, lane, bias); zero(acc); STACK_UNIT_BEGIN(); unit_gemm<true>(acc, A0, wp, ring, lane);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) x[mt][nt] += acc[mt][nt] + bias[nt];
        layer_norm(x, P.n2a, P.n2b, a.eps, red1, red2, wave, lane, y);
        store_img_plain(A0, y, wave, lane);
        __syncthreads();
        STACK_FRESH_LANE();
        // ---- FFN, 512 hidden units at a time: h_c = relu(y W1_c^T + b1_c) -> LDS, acc2 += h_c W2[:, c]^T
        f32x4 acc2[2][4];
        zero(acc2);
        for (int c = 0; c < ((a.debug & 4) ? 0 : a.NC); ++c) {
            char* Hc = (c & 1) ? KN : A1;
            load_cols(P.b1 + c * SD, wave, lane, bias); zero(acc); STACK_UNIT_BEGIN(); unit_gemm<true>(acc, A0, wp, ring, lane); store_img(Hc, acc, bias, true, wave, lane);
            __syncthreads();
            STACK_UNIT_BEGIN(); unit_gemm<true>(acc2, Hc, wp, ring, lane);
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
    layer_norm(x, a.fa, a.fb, a.eps, red1, red2, wave, lane, y);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int g = r0 + 16 * mt + (lane & 15);
        if (g < a.rows) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<uint2*>(a.y_out + (int64_t)g * SD + 64 * wave + 16 * nt + 4 * (lane >> 4)) =
                    make_uint2(pack2(y[mt][nt][0], y[mt][nt][1]), pack2(y[mt][nt][2], y[mt][nt][3]));
        }
    }
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

int stack_step(const StackArgs& a, hipStream_t s) {
    static const size_t lds = (size_t)4 * SIMG + 2 * SRB * 8 * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_stack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (a.rows < 1 || a.S < 1 || a.S > 128 || a.t < 0 || a.t >= 64 || a.per_img < 1 || a.L < 1 || a.L > STACK_MAXL || a.NC < 1) return ORTK_EINVAL;
    StackArgs b = a;
    b.nblocks = (int)ortk_cdiv(a.rows, SRB);
    const bool pf = b.nblocks >= 8 && a.progress != nullptr && !(a.debug & 8);
    // algorithmic bytes of the launch: the weights once, every image's projected memory (K and V) once per layer, every row's
    // cached keys and values once per layer plus the appended position, the residual rows in and the normalised rows out
    ProfMark pm;
    if (ortk_prof_active()) {
        const double U = 6 + 2 * a.NC, imgs = (double)ortk_cdiv(a.rows, a.per_img);
        const double bytes = a.L * (U * SD * SD * 2.0 + imgs * a.S * 2.0 * SD * 2 + (double)a.rows * (a.t + 1) * 2.0 * SD * 2) +
                             (double)a.rows * SD * (4 + 2);
        (void)prof_begin(PROF_KEY_DECSTACK, 2.0 * a.rows * a.L * U * SD * SD, bytes, s, pm);
    } else pm.live = false;
    hipLaunchKernelGGL(decoder_stack_kernel, dim3((unsigned)(b.nblocks + (pf ? 8 : 0))), dim3(512), lds, s, b);
    prof_end(pm, s);
    ORTK_CHECK_LAUNCH();
    return 0;
}

}  // namespace ortk

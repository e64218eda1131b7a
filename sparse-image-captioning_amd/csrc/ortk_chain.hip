// ortk_chain.hip — rows-stationary chains of the row-wise operators of a transformer layer (mixed precision, d_model 512).
//
// Between two attention calls a layer of the reference is a chain of ROW-WISE operators (models/transformer.py:293-294,324-325,
// 338-341: SublayerConnection = x + dropout(sublayer(LayerNorm(x))), PositionwiseFeedForward = w_2(dropout(relu(w_1(x))))):
//
//     decoder   [o1 Wo^T + bo -> x + dropout -> LayerNorm -> Wcq]           (attention)
//               [o2 Wco^T + b -> x + dropout -> LayerNorm -> W1, relu, dropout -> W2 -> x + dropout -> LayerNorm -> Wqkv]   (attention)
//     encoder   [o Wo^T + bo -> x + dropout -> LayerNorm -> W1 ... W2 -> x + dropout -> LayerNorm -> Wqkv]
//
// As separate launches (rounds 1-3) every K = 512 projection is a memory-side kernel: it reads its bf16 operand rows AND the fp32
// residual rows and writes fp32 rows — 100 MB for 8.7 GFLOP at 16 640 rows — and the LayerNorm that follows reads them again
// (26 us alone, ~60 us beside the weight gradients, against 3.5 us of matrix-core time).  Here the rows
// stay put, as in the decoder stack kernel of the decode path (ortk_decstack.hip): a workgroup (8 waves) owns up to 48
// consecutive rows for the whole chain, the residual rows live in registers (MFMA accumulator layout: wave w = columns 64 w ..
// 64 w + 63), LayerNorm is a register pass with two small LDS exchanges, every projection is a [48 x 512] x [512 x 512] unit
// whose A operand is a swizzled bf16 image in LDS and whose weights stream global -> VGPR -> MFMA through a 4-k-step ring that
// keeps running across the units, the FFN runs in 512-hidden-unit chunks through LDS, and everything the backward pass needs
// (LayerNorm outputs and statistics, projections, hidden units, residual streams) is stored on the way in the layouts the
// separate kernels write.  Dropout draws are those of the GEMM epilogues (element (row, column) of the (rows, N) output, same
// site seeds), so the two executors are interchangeable under one seed.
//
// What bounds it: the weight stream per compute unit (every workgroup pulls all units of its chain — 0.5 MB each — out of L2 at
// ~90-100 GB/s), i.e. a chain of U units costs ~5.5 us x U per round of 256 workgroups whatever the row count up to 48 rows per
// workgroup.
#include "ortk_internal.h"
#include <mutex>

namespace ortk {
namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int CD = 512;            // d_model
constexpr int CMT = 3;             // 16-row tiles per workgroup
constexpr int CRB = 16 * CMT;      // rows per workgroup (at most)
constexpr int CSPD = 4;            // k-steps of weight fragments in flight per wave
constexpr int CFRAG = 64;          // uint4 per fragment (1 KB)
constexpr int CKSTEP = 4 * CFRAG;  // uint4 per k-step of one wave (4 column tiles)
constexpr int CUNIT = 16 * CKSTEP; // uint4 per unit of one wave (64 KB)
constexpr int CIMG = CRB * CD * 2; // bytes of one bf16 A image
// Dropout keys of the workgroup's rows, behind everything else in LDS: row r draws as row skey[r] = drop_rows[r0 + r] (the valid-position
// layout keyed like the padded one, ortk_chain_args.drop_rows) or r0 + r.  A table instead of a load at every site: the kernels sit at
// 256 VGPRs and a hoisted per-lane copy of the keys spills (28-36 bytes of scratch per lane in the first version).
constexpr int CKEY_BYTES = 512;
__device__ __forceinline__ void c_fill_keys(uint32_t* skey, const int32_t* __restrict__ drop_rows, int r0, int nrow, int tid) {
    if (tid < CKEY_BYTES / 4) {
        const int g = r0 + (tid < nrow ? tid : 0);
        skey[tid] = drop_rows ? (uint32_t)drop_rows[g] : (uint32_t)g;
    }
    __syncthreads();
}

// A images: [row][64 chunks of 16 B], physical chunk = chunk ^ (row & 15) (conflict-free MFMA operand reads, as ortk_decstack.hip)
__device__ __forceinline__ int c_off(int row, int chunk) { return row * 1024 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ unsigned int c_pack2(float a, float b) {
    const bf16x2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, t);
}

struct CRing { uint4 f[CSPD][4]; };
__device__ __forceinline__ void c_ring_start(CRing& r, const uint4* wp, int lane) {
#pragma unroll
    for (int s = 0; s < CSPD; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) r.f[s][nt] = wp[s * CKSTEP + nt * CFRAG + lane];
}
// acc[mt][nt] += A[16 mt .. +15][:] . W[64 w + 16 nt .. +15][:]^T over the 512 inputs of one unit; the ring keeps running into
// whatever follows in the stream (a slack unit of zeros behind the last one)
__device__ __forceinline__ void c_unit(f32x4 (&acc)[CMT][4], const char* A, const uint4*& wp, CRing& r, int lane) {
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll 1
    for (int it = 0; it < 16 / CSPD; ++it) {
#pragma unroll
        for (int s = 0; s < CSPD; ++s) {
            const int ks = it * CSPD + s;
            bf16x8 av[CMT];
#pragma unroll
            for (int mt = 0; mt < CMT; ++mt) av[mt] = *reinterpret_cast<const bf16x8*>(A + c_off(16 * mt + m, 4 * ks + kg));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, r.f[s][nt]);
#pragma unroll
                for (int mt = 0; mt < CMT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, av[mt], acc[mt][nt], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) r.f[s][nt] = wp[(ks + CSPD) * CKSTEP + nt * CFRAG + lane];
        }
    }
    wp += CUNIT;
}
__device__ __forceinline__ void c_zero(f32x4 (&a)[CMT][4]) {
#pragma unroll
    for (int i = 0; i < CMT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void c_cols(const float* p, int wave, int lane, f32x4 (&v)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) v[nt] = *reinterpret_cast<const f32x4*>(p + 64 * wave + 16 * nt + 4 * (lane >> 4));
}

struct ChainArgs {
    const uint4* wpk;              // this launch's packed stream: [wave][units + 1 slack][16 k-steps][4 tiles][64 lanes]
    int32_t n_units;
    int32_t M, rb;                 // rows, rows per workgroup (<= 48)
    const float* x_in;             // (M, 512) residual rows in
    const __bf16* a_in; const float* bias_r; float* x_mid; uint32_t seed_r;     // R: x += dropout(a_in W^T + b)  (a_in NULL: none)
    const float *g1, *b1; __bf16* y1; float* st1;                               // L1: LayerNorm -> A image (+ y1, statistics)
    int32_t n1; const float* bias_s1; __bf16* out1; int32_t ld1;                // S1: n1 projections of the LayerNorm output
    int32_t NC; const float *bias_h, *bias_o; __bf16* h; float* x_out; uint32_t seed_h, seed_o;   // F: feed-forward sublayer
    const float *g2, *b2; __bf16* y2; float* st2;                               // L2
    int32_t n2; const float* bias_s2; __bf16* out2; int32_t ld2;                // S2
    float drop_p, eps;
    int32_t* progress;             // [8] zeroed before the launch: units begun by the pace-maker workgroup of each XCD (NULL: no prefetchers)
    int32_t npf, slots;            // L2 prefetcher workgroups at the head of the grid (0 | 8); compute workgroups per round
    const int32_t* drop_rows;      // optional: row g draws its dropout as row drop_rows[g] (ortk_chain_args.drop_rows)
};
constexpr int CAHEAD = 3;          // units the L2 prefetcher may run in front of its pace-maker

__global__ __launch_bounds__(512) void row_chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A0 = smem;                  // LayerNorm output / projection operand
    char* H0 = smem + CIMG;           // FFN hidden chunk (even)
    char* H1 = smem + 2 * CIMG;       // FFN hidden chunk (odd)
    float* red1 = reinterpret_cast<float*>(smem + 3 * CIMG);
    float* red2 = red1 + CRB * 8;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = lane0;
#define CH_FRESH_LANE() do { lane = lane0; asm volatile("" : "+v"(lane)); } while (0)
    if ((int)blockIdx.x < a.npf) {
        // L2 prefetcher of one XCD (workgroups are dealt to the 8 XCDs round-robin — a speed assumption only): the compute workgroups
        // of an XCD walk the same weight stream roughly in step, and without this every fragment load of theirs is an L2 miss they
        // all wait for (ortk_decstack.hip: 6.6 -> 4.6 us per unit).  One dword per 128-byte line of a unit's 512 KB, CAHEAD units in
        // front of the XCD's pace-maker — the first compute workgroup of every round on that XCD — with a bounded wait.
        const int xcd = (int)(blockIdx.x & 7);
        const int rounds = (int)((((int64_t)a.M + a.rb - 1) / a.rb + a.slots - 1) / a.slots);
        unsigned int sink = 0;
        int spin = 0;
        for (int v = 0; v < rounds * a.n_units; ++v) {
            for (; spin < 4096 && __hip_atomic_load(a.progress + xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v + 1 - CAHEAD; ++spin)
                __builtin_amdgcn_s_sleep(8);
            spin = spin >= 4096 ? 4096 : 0;
            const int u = v % a.n_units;
            const unsigned int* src = reinterpret_cast<const unsigned int*>(a.wpk + ((int64_t)wave * (a.n_units + 1) + u) * CUNIT) + lane * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) sink += src[i * 64 * 32];          // 8 x (64 lanes x 128 B) = this wave's 64 KB of the unit
        }
        if (sink == 0x9E3779B1u) a.progress[8] = 1;                          // (keeps the loads)
        return;
    }
    const int bid = (int)blockIdx.x - a.npf;
    const bool pace = a.npf > 0 && (bid % a.slots) < 8 && tid == 0;
    int unit_no = (bid / a.slots) * a.n_units;
#define CH_UNIT_BEGIN() do { ++unit_no; if (pace) __hip_atomic_store(a.progress + (blockIdx.x & 7), unit_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
    const int r0 = bid * a.rb;
    const int nrow = min(a.rb, a.M - r0);          // rows of this workgroup that exist
    uint32_t* skey = reinterpret_cast<uint32_t*>(smem + 3 * CIMG + 2 * CRB * 8 * sizeof(float));
    c_fill_keys(skey, a.drop_rows, r0, nrow, tid);
    const float ik = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thr = ortk_keep_thr(a.drop_p);
    const bool drop = a.drop_p > 0.f;

    const uint4* wp = a.wpk + (int64_t)wave * (a.n_units + 1) * CUNIT;
    CRing ring;
    c_ring_start(ring, wp, lane);

    f32x4 xr[CMT][4];
    // residual rows (fp32) <- global, accumulator layout; rows that do not exist read the block's first row (never stored)
    auto load_x = [&](const float* src) {
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            const int row = 16 * mt + (lane & 15);
            const int64_t g = r0 + (row < nrow ? row : 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) xr[mt][nt] = *reinterpret_cast<const f32x4*>(src + g * CD + 64 * wave + 16 * nt + 4 * (lane >> 4));
        }
    };
    auto store_x = [&](float* dst) {
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            const int row = 16 * mt + (lane & 15);
            if (row < nrow) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(dst + (int64_t)(r0 + row) * CD + 64 * wave + 16 * nt + 4 * (lane >> 4)) = xr[mt][nt];
            }
        }
    };
    // xr += dropout(acc + bias): element (row, col) of the (M, 512) output (the GEMM epilogue's draw)
    auto resid = [&](const f32x4 (&acc)[CMT][4], const float* biasp, uint32_t seed) {
        f32x4 bias[4];
        c_cols(biasp, wave, lane, bias);
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            const int row = 16 * mt + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = 64 * wave + 16 * nt + 4 * (lane >> 4);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][r] + bias[nt][r];
                if (drop) {
                    bool kp[4];
                    ortk_keep4_u32(seed, skey[row < nrow ? row : 0] * (uint32_t)CD + (uint32_t)col, thr, kp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * ik : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[mt][nt][r] += v[r];
            }
        }
    };
    // LayerNorm of the register-resident rows (transformer.py:338-341: a (x - mean) / (std_unbiased + eps) + b) -> A0 (bf16) and,
    // optionally, the rows and their {mean, std} to global memory (what ln_fwd_kernel stores for the backward pass)
    auto layer_norm = [&](const float* ga, const float* be, __bf16* yout, float* stats) {
        f32x4 a4[4], b4[4];
        c_cols(ga, wave, lane, a4);
        c_cols(be, wave, lane, b4);
        const int m = lane & 15;
        float mean[CMT], sd[CMT];
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            float s = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += xr[mt][nt][r];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if (lane < 16) red1[(16 * mt + m) * 8 + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * 8), p1 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * 8 + 4);
            mean[mt] = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / CD);
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = xr[mt][nt][r] - mean[mt]; q = __builtin_fmaf(d, d, q); }
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) red2[(16 * mt + m) * 8 + wave] = q;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * 8), p1 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * 8 + 4);
            const float var = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / (CD - 1));
            sd[mt] = sqrtf(var);
            const float rinv = 1.f / (sd[mt] + a.eps);
            const int row = 16 * mt + m;
            if (stats && wave == 0 && lane < 16 && row < nrow) *reinterpret_cast<float2*>(stats + (int64_t)(r0 + row) * 2) = make_float2(mean[mt], sd[mt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = a4[nt][r] * (xr[mt][nt][r] - mean[mt]) * rinv + b4[nt][r];
                const int col = 64 * wave + 16 * nt + 4 * (lane >> 4);
                const uint2 pk = make_uint2(c_pack2(y[0], y[1]), c_pack2(y[2], y[3]));
                *reinterpret_cast<uint2*>(A0 + c_off(row, col >> 3) + ((col >> 2) & 1) * 8) = pk;
                if (yout && row < nrow) *reinterpret_cast<uint2*>(yout + (int64_t)(r0 + row) * CD + col) = pk;
            }
        }
    };
    // n projections of the A0 rows, each (acc + bias) as bf16 to out[:, 512 i ..]
    auto project = [&](int n, const float* biasp, __bf16* out, int ld) {
        for (int i = 0; i < n; ++i) {
            f32x4 acc[CMT][4], bias[4];
            c_cols(biasp + i * CD, wave, lane, bias);
            c_zero(acc);
            CH_UNIT_BEGIN();
            c_unit(acc, A0, wp, ring, lane);
#pragma unroll
            for (int mt = 0; mt < CMT; ++mt) {
                const int row = 16 * mt + (lane & 15);
                if (row < nrow) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int col = 64 * wave + 16 * nt + 4 * (lane >> 4);
                        *reinterpret_cast<uint2*>(out + (int64_t)(r0 + row) * ld + i * CD + col) =
                            make_uint2(c_pack2(acc[mt][nt][0] + bias[nt][0], acc[mt][nt][1] + bias[nt][1]),
                                       c_pack2(acc[mt][nt][2] + bias[nt][2], acc[mt][nt][3] + bias[nt][3]));
                    }
                }
            }
        }
    };

    const float* xsrc = a.x_in;
    // ---- R: the attention output's projection, residual
    if (a.a_in) {
        // operand rows (bf16) -> A0: 512 threads x 16 B = 8 rows per pass
#pragma unroll
        for (int p = 0; p < CRB / 8; ++p) {
            const int row = 8 * p + (tid >> 6), ch = tid & 63;
            const int64_t g = r0 + (row < nrow ? row : 0);
            *reinterpret_cast<uint4*>(A0 + c_off(row, ch)) = *reinterpret_cast<const uint4*>(a.a_in + g * CD + 8 * ch);
        }
        load_x(a.x_in);
        __syncthreads();
        CH_FRESH_LANE();
        f32x4 acc[CMT][4];
        c_zero(acc);
        CH_UNIT_BEGIN();
        c_unit(acc, A0, wp, ring, lane);
        resid(acc, a.bias_r, a.seed_r);
        store_x(a.x_mid);
        xsrc = a.x_mid;
        __syncthreads();                 // every wave is past its reads of A0
    } else {
        load_x(a.x_in);
    }
    // ---- L1 (+ S1)
    if (a.g1) {
        CH_FRESH_LANE();
        layer_norm(a.g1, a.b1, a.y1, a.st1);
        __syncthreads();
        CH_FRESH_LANE();
        project(a.n1, a.bias_s1, a.out1, a.ld1);
    }
    // ---- F: feed-forward sublayer, 512 hidden units at a time: h_c = dropout(relu(y W1_c^T + b1_c)) -> LDS + global,
    //      acc2 += h_c W2[:, c]^T; then x += dropout(acc2 + b2)
    if (a.NC > 0) {
        CH_FRESH_LANE();
        f32x4 acc2[CMT][4];
        c_zero(acc2);
        const uint32_t ffn = (uint32_t)a.NC * CD;
        for (int c = 0; c < a.NC; ++c) {
            char* Hc = (c & 1) ? H1 : H0;
            f32x4 acc[CMT][4], bias[4];
            c_cols(a.bias_h + c * CD, wave, lane, bias);
            c_zero(acc);
            CH_UNIT_BEGIN();
            c_unit(acc, A0, wp, ring, lane);
#pragma unroll
            for (int mt = 0; mt < CMT; ++mt) {
                const int row = 16 * mt + (lane & 15);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int col = 64 * wave + 16 * nt + 4 * (lane >> 4);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[mt][nt][r] + bias[nt][r], 0.f);
                    if (drop) {
                        bool kp[4];
                        ortk_keep4_u32(a.seed_h, skey[row < nrow ? row : 0] * ffn + (uint32_t)(c * CD + col), thr, kp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * ik : 0.f;
                    }
                    const uint2 pk = make_uint2(c_pack2(v[0], v[1]), c_pack2(v[2], v[3]));
                    *reinterpret_cast<uint2*>(Hc + c_off(row, col >> 3) + ((col >> 2) & 1) * 8) = pk;
                    if (a.h && row < nrow) *reinterpret_cast<uint2*>(a.h + (int64_t)(r0 + row) * ffn + c * CD + col) = pk;
                }
            }
            __syncthreads();
            CH_UNIT_BEGIN();
            c_unit(acc2, Hc, wp, ring, lane);
        }
        load_x(xsrc);                    // (the residual rows were not kept through the chunks: registers)
        resid(acc2, a.bias_o, a.seed_o);
        store_x(a.x_out);
        __syncthreads();                 // every wave is past its reads of the hidden chunks and of A0
    }
    // ---- L2 (+ S2)
    if (a.g2) {
        CH_FRESH_LANE();
        layer_norm(a.g2, a.b2, a.y2, a.st2);
        __syncthreads();
        CH_FRESH_LANE();
        project(a.n2, a.bias_s2, a.out2, a.ld2);
    }
#undef CH_FRESH_LANE
#undef CH_UNIT_BEGIN
}

// ================================================================================================ wide form of the forward chain
// OPT-IN / DEFAULT: see ortk_tuning.chain_wide (profiles/r04_row_chains.txt).
// The kernel above is bound by the weight stream per compute unit (~104 GB/s out of L2 whatever the rows a workgroup holds), and with
// three 48-row LDS images it cannot hold more than 48 rows: 16 640 decoder rows need TWO rounds of workgroups, every compute unit
// streams the chain's weights twice.  This form holds up to 76 rows (5 row tiles) in the SAME eight waves — wave w = 64 output
// columns, two waves per SIMD as above —: 16 640 rows = 256 workgroups x 65 rows = ONE round, the decode's 36 864 encoder rows two
// rounds instead of three.  What makes it fit: TWO LDS images (the A operand + ONE FFN hidden chunk: an extra barrier per chunk);
// one register image of the block's rows (a product's accumulators become, in place, the fp32 residual rows the LayerNorm reads:
// 80 VGPRs); an FFN up-projection unit streamed as two half units so that the hidden chunk's accumulators are 40 registers beside the
// 80 of the down-projection's running sum.  Rows past the 76th of a tile read the neighbouring LDS bytes (don't-cares) and are never
// written.  (A first version with FOUR waves of 128 columns — one per SIMD, accumulators in AGPRs — measured slower than two rounds of
// the 48-row kernel: one wave per SIMD does not keep the matrix pipe fed.)
constexpr int WMT = 5, WNW = 8, WNT = 4, WCW = 16 * WNT, WRB = 76;
#ifndef WSPD_
#define WSPD_ 4
#endif
constexpr int WSPD = WSPD_;                 // k-steps of weight fragments in flight per wave (4 KB each)
constexpr int WKSTEP = WNT * CFRAG;         // uint4 per k-step of one wave
constexpr int WUNIT = 16 * WKSTEP;          // uint4 per unit of one wave (64 KB)
constexpr int WIMG = WRB * 1024;            // bytes of an image (76 rows); MFMA tiles cover 80: rows 76.. of A0 read H0, of H0 the red arrays
constexpr size_t WIDE_LDS = (size_t)2 * WIMG + 2 * (16 * WMT) * WNW * sizeof(float) + CKEY_BYTES;
static_assert(2 * (16 * WMT) * WNW * sizeof(float) >= 4 * 1024, "the tile rows past the second image stay inside the allocation");
static_assert(WIDE_LDS <= 160 * 1024, "LDS budget");

struct WRing { uint4 f[WSPD][WNT]; };
__device__ __forceinline__ void w_ring_start(WRing& r, const uint4* wp, int lane) {
#pragma unroll
    for (int s = 0; s < WSPD; ++s)
#pragma unroll
        for (int nt = 0; nt < WNT; ++nt) r.f[s][nt] = wp[s * WKSTEP + nt * CFRAG + lane];
}
// (the A operand of the NEXT k-step is requested before the MFMAs of this one: with two waves per SIMD and 20 MFMAs per k-step the
//  LDS round trip of five ds_read_b128 was otherwise exposed once per k-step)
__device__ __forceinline__ void w_unit(f32x4 (&acc)[WMT][WNT], const char* A, const uint4*& wp, WRing& r, int lane) {
    const int m = lane & 15, kg = lane >> 4;
    bf16x8 av[2][WMT];
#pragma unroll
    for (int mt = 0; mt < WMT; ++mt) av[0][mt] = *reinterpret_cast<const bf16x8*>(A + c_off(16 * mt + m, kg));
#pragma unroll 1
    for (int it = 0; it < 16 / WSPD; ++it) {
#pragma unroll
        for (int s = 0; s < WSPD; ++s) {
            const int ks = it * WSPD + s;
            const int kn = ks + 1 < 16 ? ks + 1 : ks;                  // (the last step re-reads its own slice: no branch in the loop)
#pragma unroll
            for (int mt = 0; mt < WMT; ++mt) av[(s + 1) & 1][mt] = *reinterpret_cast<const bf16x8*>(A + c_off(16 * mt + m, 4 * kn + kg));
#pragma unroll
            for (int nt = 0; nt < WNT; ++nt) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, r.f[s][nt]);
#pragma unroll
                for (int mt = 0; mt < WMT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, av[s & 1][mt], acc[mt][nt], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < WNT; ++nt) r.f[s][nt] = wp[(ks + WSPD) * WKSTEP + nt * CFRAG + lane];
        }
    }
    wp += WUNIT;
}
// an FFN up-projection unit is streamed as TWO half units (8 groups of four fragments each): group g of half h holds k-steps 2g and
// 2g + 1 of the wave's column tiles 2h, 2h + 1 — the hidden chunk's accumulators are 40 registers instead of 80
__device__ __forceinline__ void w_unit_half(f32x4 (&acc)[WMT][2], const char* A, const uint4*& wp, WRing& r, int lane) {
    const int m = lane & 15, kg = lane >> 4;
    bf16x8 av[2][WMT];
#pragma unroll
    for (int mt = 0; mt < WMT; ++mt) av[0][mt] = *reinterpret_cast<const bf16x8*>(A + c_off(16 * mt + m, kg));
#pragma unroll 1
    for (int it = 0; it < 8 / WSPD; ++it) {
#pragma unroll
        for (int s = 0; s < WSPD; ++s) {
            const int g = it * WSPD + s;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int ks = 2 * g + k2, kn = ks + 1 < 16 ? ks + 1 : ks;
#pragma unroll
                for (int mt = 0; mt < WMT; ++mt) av[(k2 + 1) & 1][mt] = *reinterpret_cast<const bf16x8*>(A + c_off(16 * mt + m, 4 * kn + kg));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, r.f[s][2 * k2 + j]);
#pragma unroll
                    for (int mt = 0; mt < WMT; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, av[k2 & 1][mt], acc[mt][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < WNT; ++nt) r.f[s][nt] = wp[(g + WSPD) * WKSTEP + nt * CFRAG + lane];
        }
    }
    wp += 8 * WKSTEP;
}
__device__ __forceinline__ void w_zero(f32x4 (&a)[WMT][WNT]) {
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) a[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void w_cols(const float* p, int wave, int lane, f32x4 (&v)[WNT]) {
#pragma unroll
    for (int nt = 0; nt < WNT; ++nt) v[nt] = *reinterpret_cast<const f32x4*>(p + WCW * wave + 16 * nt + 4 * (lane >> 4));
}

__global__ __launch_bounds__(512) void row_chain_wide_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A0 = smem;
    char* H0 = smem + WIMG;
    float* red1 = reinterpret_cast<float*>(smem + 2 * WIMG);
    float* red2 = red1 + 16 * WMT * WNW;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = lane0;
#define CH_FRESH_LANE() do { lane = lane0; asm volatile("" : "+v"(lane)); } while (0)
    const int r0 = blockIdx.x * a.rb;
    const int nrow = min(a.rb, a.M - r0);
    uint32_t* skey = reinterpret_cast<uint32_t*>(smem + 2 * WIMG + 2 * (16 * WMT) * WNW * sizeof(float));
    c_fill_keys(skey, a.drop_rows, r0, nrow, tid);
    const float ik = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thr = ortk_keep_thr(a.drop_p);
    const bool drop = a.drop_p > 0.f;
    const uint4* wp = a.wpk + (int64_t)wave * (a.n_units + 1) * WUNIT;
    WRing ring;
    w_ring_start(ring, wp, lane);

    // ONE register image of the block's rows: a product's accumulators, then (in place) the fp32 residual rows the LayerNorm reads
    f32x4 xa[WMT][WNT];
    auto load_x = [&](const float* src) {
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) {
            const int row = 16 * mt + (lane & 15);
            const int64_t g = r0 + (row < nrow ? row : 0);
#pragma unroll
            for (int nt = 0; nt < WNT; ++nt) xa[mt][nt] = *reinterpret_cast<const f32x4*>(src + g * CD + WCW * wave + 16 * nt + 4 * (lane >> 4));
        }
    };
    // xa = src rows + dropout(xa + bias), stored to dst
    auto resid = [&](const float* biasp, uint32_t seed, const float* src, float* dst) {
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) {
            const int row = 16 * mt + (lane & 15);
            const uint32_t g = (uint32_t)(r0 + (row < nrow ? row : 0));
#pragma unroll
            for (int nt = 0; nt < WNT; ++nt) {
                const int col = WCW * wave + 16 * nt + 4 * (lane >> 4);
                const f32x4 x = *reinterpret_cast<const f32x4*>(src + (int64_t)g * CD + col);
                const f32x4 bias = *reinterpret_cast<const f32x4*>(biasp + col);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = xa[mt][nt][r] + bias[r];
                if (drop) {
                    bool kp[4];
                    ortk_keep4_u32(seed, skey[row < nrow ? row : 0] * (uint32_t)CD + (uint32_t)col, thr, kp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * ik : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) xa[mt][nt][r] = x[r] + v[r];
                if (row < nrow) *reinterpret_cast<f32x4*>(dst + (int64_t)g * CD + col) = xa[mt][nt];
            }
        }
    };
    auto layer_norm = [&](const float* ga, const float* be, __bf16* yout, float* stats) {
        const int m = lane & 15;
        float mean[WMT];
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) {
            float s = 0.f;
#pragma unroll
            for (int nt = 0; nt < WNT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += xa[mt][nt][r];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if (lane < 16) red1[(16 * mt + m) * WNW + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * WNW), p1 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * WNW + 4);
            mean[mt] = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / CD);
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < WNT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = xa[mt][nt][r] - mean[mt]; q = __builtin_fmaf(d, d, q); }
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (lane < 16) red2[(16 * mt + m) * WNW + wave] = q;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * WNW), p1 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * WNW + 4);
            const float var = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / (CD - 1));
            const float sd = sqrtf(var);
            const float rinv = 1.f / (sd + a.eps);
            const int row = 16 * mt + m;
            if (stats && wave == 0 && lane < 16 && row < nrow) *reinterpret_cast<float2*>(stats + (int64_t)(r0 + row) * 2) = make_float2(mean[mt], sd);
#pragma unroll
            for (int nt = 0; nt < WNT; ++nt) {
                const int col = WCW * wave + 16 * nt + 4 * (lane >> 4);
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ga + col), b4 = *reinterpret_cast<const f32x4*>(be + col);
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = a4[r] * (xa[mt][nt][r] - mean[mt]) * rinv + b4[r];
                const uint2 pk = make_uint2(c_pack2(y[0], y[1]), c_pack2(y[2], y[3]));
                if (row < WRB) *reinterpret_cast<uint2*>(A0 + c_off(row, col >> 3) + ((col >> 2) & 1) * 8) = pk;
                if (yout && row < nrow) *reinterpret_cast<uint2*>(yout + (int64_t)(r0 + row) * CD + col) = pk;
            }
        }
    };
    auto project = [&](int n, const float* biasp, __bf16* out, int ld) {
        for (int i = 0; i < n; ++i) {
            w_zero(xa);
            w_unit(xa, A0, wp, ring, lane);
#pragma unroll
            for (int mt = 0; mt < WMT; ++mt) {
                const int row = 16 * mt + (lane & 15);
#pragma unroll
                for (int nt = 0; nt < WNT; ++nt) {
                    const int col = WCW * wave + 16 * nt + 4 * (lane >> 4);
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(biasp + i * CD + col);
                    if (row < nrow)
                        *reinterpret_cast<uint2*>(out + (int64_t)(r0 + row) * ld + i * CD + col) =
                            make_uint2(c_pack2(xa[mt][nt][0] + bias[0], xa[mt][nt][1] + bias[1]), c_pack2(xa[mt][nt][2] + bias[2], xa[mt][nt][3] + bias[3]));
                }
            }
        }
    };

    const float* xsrc = a.x_in;
    if (a.a_in) {
        for (int row = tid >> 6; row < WRB; row += WNW) {
            const int ch = tid & 63;
            const int64_t g = r0 + (row < nrow ? row : 0);
            *reinterpret_cast<uint4*>(A0 + c_off(row, ch)) = *reinterpret_cast<const uint4*>(a.a_in + g * CD + 8 * ch);
        }
        __syncthreads();
        CH_FRESH_LANE();
        w_zero(xa);
        w_unit(xa, A0, wp, ring, lane);
        resid(a.bias_r, a.seed_r, a.x_in, a.x_mid);
        xsrc = a.x_mid;
        __syncthreads();
    } else {
        load_x(a.x_in);
    }
    if (a.g1) {
        CH_FRESH_LANE();
        layer_norm(a.g1, a.b1, a.y1, a.st1);
        __syncthreads();
        CH_FRESH_LANE();
        project(a.n1, a.bias_s1, a.out1, a.ld1);
    }
    if (a.NC > 0) {
        CH_FRESH_LANE();
        w_zero(xa);
        const uint32_t ffn = (uint32_t)a.NC * CD;
        for (int c = 0; c < a.NC; ++c) {
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 hacc[WMT][2];
#pragma unroll
                for (int i = 0; i < WMT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) hacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                w_unit_half(hacc, A0, wp, ring, lane);
                if (c > 0 && hf == 0) __syncthreads();          // every wave is past its reads of the previous hidden chunk
#pragma unroll
                for (int mt = 0; mt < WMT; ++mt) {
                    const int row = 16 * mt + (lane & 15);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = WCW * wave + 32 * hf + 16 * j + 4 * (lane >> 4);
                        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias_h + c * CD + col);
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(hacc[mt][j][r] + bias[r], 0.f);
                        if (drop) {
                            bool kp[4];
                            ortk_keep4_u32(a.seed_h, skey[row < nrow ? row : 0] * ffn + (uint32_t)(c * CD + col), thr, kp);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * ik : 0.f;
                        }
                        const uint2 pk = make_uint2(c_pack2(v[0], v[1]), c_pack2(v[2], v[3]));
                        if (row < WRB) *reinterpret_cast<uint2*>(H0 + c_off(row, col >> 3) + ((col >> 2) & 1) * 8) = pk;
                        if (a.h && row < nrow) *reinterpret_cast<uint2*>(a.h + (int64_t)(r0 + row) * ffn + c * CD + col) = pk;
                    }
                }
            }
            __syncthreads();
            w_unit(xa, H0, wp, ring, lane);
        }
        resid(a.bias_o, a.seed_o, xsrc, a.x_out);
        __syncthreads();
    } else if (a.g2) {
        load_x(xsrc);                        // (not kept across the projections above)
    }
    if (a.g2) {
        CH_FRESH_LANE();
        layer_norm(a.g2, a.b2, a.y2, a.st2);
        __syncthreads();
        CH_FRESH_LANE();
        project(a.n2, a.bias_s2, a.out2, a.ld2);
    }
#undef CH_FRESH_LANE
}

// ================================================================================================ backward chains
// The same idea for the backward pass of the row-wise operators (everything between two attention-backward calls except the
// weight gradients, which reduce over ALL rows and stay GEMMs on the side stream): data gradients through the TRANSPOSED bf16
// weight copies, the ReLU / dropout gate of the FFN, the LayerNorm backward on the register-resident gradient rows, the
// dropout-masked bf16 copy the next sublayer's products (and weight gradients) read.
//
//   P0 (nin = 1 | 3):  gy = sum_i ain[:, 512 i ..] . U_i            (dq . Wcq, or the packed dQ|dK|dV . Wqkv: K = 1 536)
//        or (nin = 0): the A image <- dz0 rows (a masked gradient that already exists)
//   LNa (nin > 0):     dx = LayerNorm'(gy; x, {mean, std}, gain) + dres  -> dxa;  d gain / d bias += column sums;
//                      dz = dropout-mask(dx) (site of the NEXT sublayer down the stack) -> A image (+ dza rows)
//   P1 (NC > 0):       gh_c = gate(h_c) (A . W2^T_c) -> LDS + gh rows;  gy2 += gh_c . W1^T_c        (PositionwiseFeedForward backward)
//   LNb (NC > 0):      as LNa on gy2 (x = the FFN sublayer's input rows), dres = dresb (may be the dxa rows just written)
//   P2 (n2 = 1):       out2 = A . U (the masked gradient through an attention out-projection) -> bf16 rows
//
// LayerNorm backward (transformer.py:338-341 differentiated; ortk_norm.hip: ln_bwd_kernel): with xc = x - mean, r = 1 / (std + eps),
// g = gy * gain, n = 512:  dx = r (g - mean(g)) - r^2 sum(g xc) xc / ((n - 1) std) [+ dres];  d gain += sum_rows gy xc r;  d bias += sum_rows gy.
struct BChainArgs {
    const uint4* wpk; int32_t n_units; int32_t M, rb;
    int32_t nin; const __bf16* ain; int32_t ld_ain; const __bf16* dz0;
    const float *xa, *sta, *ga, *dresa; float *dxa, *daa, *dba; __bf16* dza; uint32_t seed_a; int32_t mask_a;     // mask_a: dz wanted
    int32_t NC; const __bf16* hgate; __bf16* gh; float gate_scale;
    const float *xb, *stb, *gb, *dresb; float *dxb, *dab, *dbb; __bf16* dzb; uint32_t seed_b; int32_t mask_b;
    int32_t n2; __bf16* out2;
    float drop_p, eps;
    const int32_t* drop_rows;
};

__global__ __launch_bounds__(512) void row_bchain_kernel(BChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A0 = smem;
    char* H0 = smem + CIMG;
    char* H1 = smem + 2 * CIMG;
    float* red1 = reinterpret_cast<float*>(smem + 3 * CIMG);
    float* red2 = red1 + CRB * 8;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = lane0;
#define CH_FRESH_LANE() do { lane = lane0; asm volatile("" : "+v"(lane)); } while (0)
    const int r0 = blockIdx.x * a.rb;
    const int nrow = min(a.rb, a.M - r0);
    uint32_t* skey = reinterpret_cast<uint32_t*>(smem + 3 * CIMG + 2 * CRB * 8 * sizeof(float));
    c_fill_keys(skey, a.drop_rows, r0, nrow, tid);
    const float ik = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thr = ortk_keep_thr(a.drop_p);
    const bool drop = a.drop_p > 0.f;
    const uint4* wp = a.wpk + (int64_t)wave * (a.n_units + 1) * CUNIT;
    CRing ring;
    c_ring_start(ring, wp, lane);

    // bf16 rows (M, ld) columns col0 .. col0 + 511 -> an A image (512 threads x 16 B = 8 rows per pass)
    auto load_img = [&](char* img, const __bf16* src, int ld, int col0) {
#pragma unroll
        for (int p = 0; p < CRB / 8; ++p) {
            const int row = 8 * p + (tid >> 6), ch = tid & 63;
            const int64_t g = r0 + (row < nrow ? row : 0);
            *reinterpret_cast<uint4*>(img + c_off(row, ch)) = *reinterpret_cast<const uint4*>(src + g * ld + col0 + 8 * ch);
        }
    };
    // LayerNorm backward on the gradient rows in `gy` (accumulator layout); leaves dx in gy, stores it, adds the parameter gradients,
    // and (mask) writes the dropout-masked bf16 copy into A0 (+ dz rows).  Two barriers inside; the caller syncs before A0 is read.
    auto ln_bwd = [&](f32x4 (&gy)[CMT][4], const float* x, const float* stats, const float* gain, const float* dres, float* dxo, float* dap, float* dbp,
                      __bf16* dzo, uint32_t seed, bool mask) {
        const int m = lane & 15, q4 = lane >> 4;
        f32x4 xc[CMT][4];
        float r[CMT], sd[CMT];
        {
            f32x4 pa[4], pb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { pa[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; pb[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int mt = 0; mt < CMT; ++mt) {
                const int row = 16 * mt + m;
                const bool live = row < nrow;
                const int64_t g = r0 + (live ? row : 0);
                const float2 st = *reinterpret_cast<const float2*>(stats + g * 2);
                sd[mt] = st.y; r[mt] = 1.f / (st.y + a.eps);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + g * CD + 64 * wave + 16 * nt + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xc[mt][nt][e] = xv[e] - st.x;
                        const float gv = live ? gy[mt][nt][e] : 0.f;          // rows that do not exist add nothing to the parameter gradients
                        pa[nt][e] += gv * xc[mt][nt][e] * r[mt];
                        pb[nt][e] += gv;
                    }
                }
            }
            // column sums over the block's rows: the 16 lanes of a DPP row hold the 16 rows of a tile
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float va = pa[nt][e], vb = pb[nt][e];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) { va += __shfl_xor(va, o, 64); vb += __shfl_xor(vb, o, 64); }
                    if (m == 0) { atomicAdd(dap + 64 * wave + 16 * nt + 4 * q4 + e, va); atomicAdd(dbp + 64 * wave + 16 * nt + 4 * q4 + e, vb); }
                }
        }
        f32x4 gn[4];
        c_cols(gain, wave, lane, gn);
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            float sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) { gy[mt][nt][e] *= gn[nt][e]; sg += gy[mt][nt][e]; sgx = __builtin_fmaf(gy[mt][nt][e], xc[mt][nt][e], sgx); }
            sg += __shfl_xor(sg, 16, 64); sg += __shfl_xor(sg, 32, 64);
            sgx += __shfl_xor(sgx, 16, 64); sgx += __shfl_xor(sgx, 32, 64);
            if (lane < 16) { red1[(16 * mt + m) * 8 + wave] = sg; red2[(16 * mt + m) * 8 + wave] = sgx; }
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * 8), p1 = *reinterpret_cast<const f32x4*>(red1 + (16 * mt + m) * 8 + 4);
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * 8), s1 = *reinterpret_cast<const f32x4*>(red2 + (16 * mt + m) * 8 + 4);
            const float mg = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / CD);
            const float sx = ((s0[0] + s0[1]) + (s0[2] + s0[3])) + ((s1[0] + s1[1]) + (s1[2] + s1[3]));
            const float coef = r[mt] * r[mt] * sx / ((float)(CD - 1) * sd[mt]);
            const int row = 16 * mt + m;
            const bool live = row < nrow;
            const int64_t g = r0 + (live ? row : 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = 64 * wave + 16 * nt + 4 * q4;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = r[mt] * (gy[mt][nt][e] - mg) - coef * xc[mt][nt][e];
                if (dres) {
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(dres + g * CD + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += dv[e];
                }
                gy[mt][nt] = o;
                if (live) *reinterpret_cast<f32x4*>(dxo + g * CD + col) = o;
                if (mask) {
                    float z[4] = {o[0], o[1], o[2], o[3]};
                    if (drop) {
                        bool kp[4];
                        ortk_keep4_u32(seed, skey[row < nrow ? row : 0] * (uint32_t)CD + (uint32_t)col, thr, kp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[e] = kp[e] ? z[e] * ik : 0.f;
                    }
                    const uint2 pk = make_uint2(c_pack2(z[0], z[1]), c_pack2(z[2], z[3]));
                    *reinterpret_cast<uint2*>(A0 + c_off(row, col >> 3) + ((col >> 2) & 1) * 8) = pk;
                    if (dzo && live) *reinterpret_cast<uint2*>(dzo + g * CD + col) = pk;
                }
            }
        }
        __syncthreads();             // (red1 / red2 are free again; A0 is complete)
    };

    // ---- P0 / LNa
    if (a.nin > 0) {
        load_img(A0, a.ain, a.ld_ain, 0);
        if (a.nin > 1) { load_img(H0, a.ain, a.ld_ain, CD); load_img(H1, a.ain, a.ld_ain, 2 * CD); }
        __syncthreads();
        CH_FRESH_LANE();
        f32x4 acc[CMT][4];
        c_zero(acc);
        c_unit(acc, A0, wp, ring, lane);
        if (a.nin > 1) { c_unit(acc, H0, wp, ring, lane); c_unit(acc, H1, wp, ring, lane); }
        __syncthreads();             // every wave is past its reads of the three images
        ln_bwd(acc, a.xa, a.sta, a.ga, a.dresa, a.dxa, a.daa, a.dba, a.dza, a.seed_a, a.mask_a != 0);
    } else {
        load_img(A0, a.dz0, CD, 0);
        __syncthreads();
    }
    // ---- P1 / LNb: the feed-forward sublayer backwards, 512 hidden units at a time
    if (a.NC > 0) {
        CH_FRESH_LANE();
        f32x4 acc2[CMT][4];
        c_zero(acc2);
        const int64_t ffn = (int64_t)a.NC * CD;
        for (int c = 0; c < a.NC; ++c) {
            char* Hc = (c & 1) ? H1 : H0;
            f32x4 acc[CMT][4];
            c_zero(acc);
            c_unit(acc, A0, wp, ring, lane);
#pragma unroll
            for (int mt = 0; mt < CMT; ++mt) {
                const int row = 16 * mt + (lane & 15);
                const bool live = row < nrow;
                const int64_t g = r0 + (live ? row : 0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int col = 64 * wave + 16 * nt + 4 * (lane >> 4);
                    const uint2 hv = *reinterpret_cast<const uint2*>(a.hgate + g * ffn + c * CD + col);
                    // gate: h > 0 (ReLU active and the unit kept by its dropout) passes the gradient, scaled by 1 / (1 - p)
                    const bool k0 = (hv.x & 0x7FFFu) != 0 && !(hv.x & 0x8000u), k1 = (hv.x & 0x7FFF0000u) != 0 && !(hv.x & 0x80000000u);
                    const bool k2 = (hv.y & 0x7FFFu) != 0 && !(hv.y & 0x8000u), k3 = (hv.y & 0x7FFF0000u) != 0 && !(hv.y & 0x80000000u);
                    const float v0 = k0 ? acc[mt][nt][0] * a.gate_scale : 0.f, v1 = k1 ? acc[mt][nt][1] * a.gate_scale : 0.f;
                    const float v2 = k2 ? acc[mt][nt][2] * a.gate_scale : 0.f, v3 = k3 ? acc[mt][nt][3] * a.gate_scale : 0.f;
                    const uint2 pk = make_uint2(c_pack2(v0, v1), c_pack2(v2, v3));
                    *reinterpret_cast<uint2*>(Hc + c_off(row, col >> 3) + ((col >> 2) & 1) * 8) = pk;
                    if (live) *reinterpret_cast<uint2*>(a.gh + g * ffn + c * CD + col) = pk;
                }
            }
            __syncthreads();
            c_unit(acc2, Hc, wp, ring, lane);
        }
        __syncthreads();             // every wave is past its reads of A0 and of the hidden chunks
        ln_bwd(acc2, a.xb, a.stb, a.gb, a.dresb, a.dxb, a.dab, a.dbb, a.dzb, a.seed_b, a.mask_b != 0);
    }
    // ---- P2
    if (a.n2 > 0) {
        CH_FRESH_LANE();
        f32x4 acc[CMT][4];
        c_zero(acc);
        c_unit(acc, A0, wp, ring, lane);
#pragma unroll
        for (int mt = 0; mt < CMT; ++mt) {
            const int row = 16 * mt + (lane & 15);
            if (row < nrow) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int col = 64 * wave + 16 * nt + 4 * (lane >> 4);
                    *reinterpret_cast<uint2*>(a.out2 + (int64_t)(r0 + row) * CD + col) =
                        make_uint2(c_pack2(acc[mt][nt][0], acc[mt][nt][1]), c_pack2(acc[mt][nt][2], acc[mt][nt][3]));
                }
            }
        }
    }
#undef CH_FRESH_LANE
}

// out[(((w NU1 + u) 16 + ks) 4 + nt) 64 + lane] = the 8 bf16 W_u[64 w + 16 nt + (lane & 15)][32 ks + 8 (lane >> 4) ..], NU1 = units + 1
// (the slack unit behind every wave's stream — the ring's read-ahead — is zero-filled)
// The wide kernel's stream: full units (form 1) as above; an FFN up-projection unit (form 2) as two half units: group g = 8 h + g8 of four
// fragments holds k-steps 2 g8 + (j >> 1) of the wave's column tiles 2 h + (j & 1).  Same bytes per chain and wave.
struct PackUnit { int64_t offset, ld; };
__device__ __forceinline__ int64_t pack_src(int form, int w, int q, int64_t ld) {
    const int lane = q & 63;
    int nt, ks, col0;
    col0 = 64 * w;
    if (form != 2) { nt = (q >> 6) & 3; ks = q >> 8; }                         // (the wide kernel's full units stream like the 48-row kernel's)
    else { const int j = (q >> 6) & 3, g = q >> 8; ks = 2 * (g & 7) + (j >> 1); nt = 2 * (g >> 3) + (j & 1); }
    return (int64_t)(col0 + 16 * nt + (lane & 15)) * ld + 32 * ks + 8 * (lane >> 4);
}
__global__ __launch_bounds__(256) void chain_pack_kernel(const __bf16* __restrict__ w16, uint4* __restrict__ out, const PackUnit* __restrict__ units,
                                                         int n_units, int wide, int n_r, int n1, int NC) {
    const int64_t usz = CUNIT;
    const int64_t per_wave = (int64_t)(n_units + 1) * usz, total = 8 * per_wave;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i / per_wave);
        const int64_t rest = i - (int64_t)w * per_wave;
        const int u = (int)(rest / usz), q = (int)(rest % usz);
        if (u >= n_units) { out[i] = make_uint4(0, 0, 0, 0); continue; }
        const PackUnit pu = units[u];
        out[i] = *reinterpret_cast<const uint4*>(w16 + pu.offset + pack_src(chain_unit_form(wide != 0, u, n_r, n1, NC), w, q, pu.ld));
    }
}

__global__ __launch_bounds__(256) void chain_pack_all_kernel(const __bf16* __restrict__ w16, uint4* __restrict__ out, ChainPackTable t, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int c = 0;
        int64_t rest = i;
        for (; c < t.n_chains; ++c) {
            const int64_t sz = (int64_t)8 * (t.first[c + 1] - t.first[c] + 1) * CUNIT;
            if (rest < sz) break;
            rest -= sz;
        }
        const int n = t.first[c + 1] - t.first[c];
        const int64_t usz = CUNIT;
        const int64_t per_wave = (int64_t)(n + 1) * usz;
        const int w = (int)(rest / per_wave);
        const int64_t r2 = rest - (int64_t)w * per_wave;
        const int u = (int)(r2 / usz), q = (int)(r2 % usz);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (u < n) {
            const auto pu = t.u[t.first[c] + u];
            v = *reinterpret_cast<const uint4*>(w16 + (int64_t)pu.offset + pack_src(t.form[t.first[c] + u], w, q, pu.ld));
        }
        out[t.base[c] + rest] = v;
    }
}

constexpr size_t CHAIN_LDS = (size_t)3 * CIMG + 2 * CRB * 8 * sizeof(float) + CKEY_BYTES;

}  // namespace

size_t chain_packed_bytes(int n_units) { return (size_t)8 * (n_units + 1) * CUNIT * sizeof(uint4); }

// rows per workgroup for M rows: the fewest rounds of 256 workgroups that 48-row blocks allow, the rows spread evenly over them
int chain_rows_per_block(int64_t M, int slots) {
    const int64_t blocks48 = ortk_cdiv(M, CRB), rounds = ortk_cdiv(blocks48, slots);
    const int64_t rb = ortk_cdiv(M, rounds * slots);
    return (int)std::min<int64_t>(CRB, std::max<int64_t>(rb, 1));
}

bool chain_wide(int64_t M) {
    if (!tuning().chain_wide) return false;
    // ONE round of 76-row blocks instead of two of 48-row blocks (12 289 .. 19 456 rows: the training decoder's 16 640): 187 vs 221 us
    // on the 13-unit chain.  Two rounds instead of three (36 864 rows, the decode's encoder) measured slower, 380 vs 353 us: a 76-row
    // round costs 12 us per unit (MFMA + operand reads + weight stream barely overlap with two waves per SIMD) against 7.5.
    if (tuning().chain_wide == 2) return ortk_cdiv(ortk_cdiv(M, (int64_t)WRB), 256) < ortk_cdiv(ortk_cdiv(M, (int64_t)CRB), 256);      // (A/B: wherever it saves a round)
    return ortk_cdiv(ortk_cdiv(M, (int64_t)WRB), 256) == 1 && ortk_cdiv(ortk_cdiv(M, (int64_t)CRB), 256) == 2;
}

int chain_pack(const void* w16, const ortk_chain_unit* units_dev, int n_units, void* packed, hipStream_t s, bool wide, int n_r, int n1, int NC) {
    if (!w16 || !units_dev || n_units < 1 || !packed) return ORTK_EINVAL;
    static_assert(sizeof(PackUnit) == sizeof(ortk_chain_unit), "unit descriptor");
    static_assert(WUNIT == CUNIT, "both streams take the same bytes per unit and wave");
    hipLaunchKernelGGL(chain_pack_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<const __bf16*>(w16), reinterpret_cast<uint4*>(packed),
                       reinterpret_cast<const PackUnit*>(units_dev), n_units, wide ? 1 : 0, n_r, n1, NC);
    ORTK_CHECK_LAUNCH();
    return 0;
}

int chain_pack_all(const void* w16, void* packed, const ChainPackTable& t, hipStream_t s) {
    if (!w16 || !packed || t.n_chains < 1 || t.n_chains > CH_MAX_CHAINS) return ORTK_EINVAL;
    int64_t total = 0;
    for (int c = 0; c < t.n_chains; ++c) {
        if (t.base[c] != total || t.first[c + 1] <= t.first[c] || t.first[c + 1] > CH_MAX_UNITS) return ORTK_EINVAL;       // packed back to back
        total += (int64_t)8 * (t.first[c + 1] - t.first[c] + 1) * CUNIT;
    }
    hipLaunchKernelGGL(chain_pack_all_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const __bf16*>(w16), reinterpret_cast<uint4*>(packed), t, total);
    ORTK_CHECK_LAUNCH();
    return 0;
}

int chain_run(const ortk_chain_args* p, const void* packed, hipStream_t s) {
    if (!p || !packed || !p->x_in || p->M < 1 || p->M > (int64_t)1 << 22) return ORTK_EINVAL;
    if (p->n1 < 0 || p->n1 > 3 || p->n2 < 0 || p->n2 > 3 || p->NC < 0 || p->NC > 8) return ORTK_EINVAL;
    if ((p->n1 > 0 && !p->g1) || (p->n2 > 0 && !p->g2)) return ORTK_EINVAL;
    if (p->a_in && (!p->bias_r || !p->x_mid)) return ORTK_EINVAL;
    if (p->n1 > 0 && (!p->bias_s1 || !p->out1 || p->ld1 < p->n1 * CD || p->ld1 % 4)) return ORTK_EINVAL;
    if (p->n2 > 0 && (!p->bias_s2 || !p->out2 || p->ld2 < p->n2 * CD || p->ld2 % 4)) return ORTK_EINVAL;
    if (p->NC > 0 && (!p->g1 || !p->bias_h || !p->bias_o || !p->x_out)) return ORTK_EINVAL;      // (h NULL: inference, the hidden units are not kept)
    if (p->drop_p < 0.f || p->drop_p >= 1.f) return ORTK_EINVAL;
    const int n_units = (p->a_in ? 1 : 0) + p->n1 + 2 * p->NC + p->n2;
    if (n_units < 1 || n_units != p->n_units) return ORTK_EINVAL;
    if ((int64_t)p->M * std::max(p->NC, 1) * CD >= ((int64_t)1 << 32)) return ORTK_EINVAL;       // 32-bit dropout element indices
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ORTK_EINVAL;
    {
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(row_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(row_chain_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_LDS);
            if (e != hipSuccess) return (int)e;
            done[dev] = true;
        }
    }
    const bool wide = !p->progress && chain_wide(p->M);      // (the caller packed the stream for the same rule)
    ChainArgs a;
    a.wpk = reinterpret_cast<const uint4*>(packed); a.n_units = n_units;
    // with a progress array the grid starts with 8 L2 prefetcher workgroups (one per XCD): 248 compute workgroups per round
    a.progress = p->progress; a.npf = p->progress ? 8 : 0; a.slots = 256 - a.npf;
    if (p->progress && hipMemsetAsync(p->progress, 0, 16 * sizeof(int32_t), s) != hipSuccess) return ORTK_EINVAL;
    a.M = (int)p->M; a.rb = chain_rows_per_block(p->M, a.slots);
    if (wide) a.rb = (int)std::min<int64_t>(WRB, ortk_cdiv(p->M, ortk_cdiv(ortk_cdiv(p->M, (int64_t)WRB), 256) * 256));
    a.x_in = p->x_in;
    a.a_in = reinterpret_cast<const __bf16*>(p->a_in); a.bias_r = p->bias_r; a.x_mid = p->x_mid; a.seed_r = p->seed_r;
    a.g1 = p->g1; a.b1 = p->b1; a.y1 = reinterpret_cast<__bf16*>(p->y1); a.st1 = p->st1;
    a.n1 = p->n1; a.bias_s1 = p->bias_s1; a.out1 = reinterpret_cast<__bf16*>(p->out1); a.ld1 = (int)p->ld1;
    a.NC = p->NC; a.bias_h = p->bias_h; a.bias_o = p->bias_o; a.h = reinterpret_cast<__bf16*>(p->h); a.x_out = p->x_out;
    a.seed_h = p->seed_h; a.seed_o = p->seed_o;
    a.g2 = p->g2; a.b2 = p->b2; a.y2 = reinterpret_cast<__bf16*>(p->y2); a.st2 = p->st2;
    a.n2 = p->n2; a.bias_s2 = p->bias_s2; a.out2 = reinterpret_cast<__bf16*>(p->out2); a.ld2 = (int)p->ld2;
    a.drop_p = p->drop_p; a.eps = p->eps; a.drop_rows = p->drop_rows;
    const unsigned grid = (unsigned)(ortk_cdiv(p->M, a.rb) + a.npf);
    ProfMark pm;
    if (ortk_prof_active()) {
        // algorithmic HBM bytes: the weights once + every row tensor the chain reads or writes
        const double rowb = 2048.0 + (p->a_in ? 1024 + 2048 : 0) + (p->g1 ? (p->y1 ? 1024 : 0) + 8 : 0) + p->n1 * 1024.0 +
                            (p->NC ? (p->h ? p->NC * 1024.0 : 0.0) + 2048 + 2048 : 0) + (p->g2 ? (p->y2 ? 1024 : 0) + 8 : 0) + p->n2 * 1024.0;
        (void)prof_begin(PROF_KEY_CHAIN, 2.0 * p->M * n_units * CD * CD, (double)n_units * CD * CD * 2 + rowb * p->M, s, pm);
    } else pm.live = false;
    if (wide) hipLaunchKernelGGL(row_chain_wide_kernel, dim3(grid), dim3(512), WIDE_LDS, s, a);
    else hipLaunchKernelGGL(row_chain_kernel, dim3(grid), dim3(512), CHAIN_LDS, s, a);
    prof_end(pm, s);
    ORTK_CHECK_LAUNCH();
    return 0;
}

int bchain_run(const ortk_bchain_args* p, const void* packed, hipStream_t s) {
    if (!p || !packed || p->M < 1 || p->M > (int64_t)1 << 22) return ORTK_EINVAL;
    if ((p->nin != 0 && p->nin != 1 && p->nin != 3) || p->NC < 0 || p->NC > 8 || p->n2 < 0 || p->n2 > 1) return ORTK_EINVAL;
    if (p->nin > 0 && (!p->ain || p->ld_ain < p->nin * CD || p->ld_ain % 8 || !p->xa || !p->sta || !p->ga || !p->dxa || !p->daa || !p->dba)) return ORTK_EINVAL;
    if (p->nin == 0 && !p->dz0) return ORTK_EINVAL;
    if (p->NC > 0 && (!p->hgate || !p->gh || !p->xb || !p->stb || !p->gb || !p->dxb || !p->dab || !p->dbb)) return ORTK_EINVAL;
    if (p->n2 > 0 && !p->out2) return ORTK_EINVAL;
    if ((p->NC > 0 || p->n2 > 0) && p->nin > 0 && !p->mask_a) return ORTK_EINVAL;       // P1 / P2 read the masked image LNa leaves
    if (p->n2 > 0 && p->NC > 0 && !p->mask_b) return ORTK_EINVAL;
    if (p->drop_p < 0.f || p->drop_p >= 1.f) return ORTK_EINVAL;
    const int n_units = p->nin + 2 * p->NC + p->n2;
    if (n_units < 1 || n_units != p->n_units) return ORTK_EINVAL;
    if ((int64_t)p->M * CD >= ((int64_t)1 << 32)) return ORTK_EINVAL;
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ORTK_EINVAL;
    {
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(row_bchain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS);
            if (e != hipSuccess) return (int)e;
            done[dev] = true;
        }
    }
    BChainArgs a;
    a.wpk = reinterpret_cast<const uint4*>(packed); a.n_units = n_units; a.M = (int)p->M; a.rb = chain_rows_per_block(p->M, 256);
    a.nin = p->nin; a.ain = reinterpret_cast<const __bf16*>(p->ain); a.ld_ain = (int)p->ld_ain; a.dz0 = reinterpret_cast<const __bf16*>(p->dz0);
    a.xa = p->xa; a.sta = p->sta; a.ga = p->ga; a.dresa = p->dresa; a.dxa = p->dxa; a.daa = p->daa; a.dba = p->dba;
    a.dza = reinterpret_cast<__bf16*>(p->dza); a.seed_a = p->seed_a; a.mask_a = p->mask_a;
    a.NC = p->NC; a.hgate = reinterpret_cast<const __bf16*>(p->hgate); a.gh = reinterpret_cast<__bf16*>(p->gh); a.gate_scale = p->gate_scale;
    a.xb = p->xb; a.stb = p->stb; a.gb = p->gb; a.dresb = p->dresb; a.dxb = p->dxb; a.dab = p->dab; a.dbb = p->dbb;
    a.dzb = reinterpret_cast<__bf16*>(p->dzb); a.seed_b = p->seed_b; a.mask_b = p->mask_b;
    a.n2 = p->n2; a.out2 = reinterpret_cast<__bf16*>(p->out2);
    a.drop_p = p->drop_p; a.eps = p->eps; a.drop_rows = p->drop_rows;
    ProfMark pm;
    if (ortk_prof_active()) (void)prof_begin(PROF_KEY_CHAIN, 2.0 * p->M * n_units * CD * CD, (double)n_units * CD * CD * 2, s, pm); else pm.live = false;
    hipLaunchKernelGGL(row_bchain_kernel, dim3((unsigned)ortk_cdiv(p->M, a.rb)), dim3(512), CHAIN_LDS, s, a);
    prof_end(pm, s);
    ORTK_CHECK_LAUNCH();
    return 0;
}

}  // namespace ortk

// Operator form of the backward chain (tests, tools): packs the units of this call from `w16t` into `packed`, then runs it.
extern "C" int ortk_row_bchain(const ortk_bchain_args* p, ortk_stream stream) {
    if (!p || !p->w16t || !p->units_dev || !p->packed || p->packed_bytes < ortk::chain_packed_bytes(p->n_units)) return ORTK_EINVAL;
    hipStream_t s = ortk_s(stream);
    if (int e = ortk::chain_pack(p->w16t, p->units_dev, p->n_units, p->packed, s)) return e;
    return ortk::bchain_run(p, p->packed, s);
}

extern "C" size_t ortk_chain_packed_bytes(int32_t n_units) { return n_units < 1 ? 0 : ortk::chain_packed_bytes(n_units); }

// Operator form (tests, tools): packs the units of this call from `w16` into `packed`, then runs the chain.
extern "C" int ortk_row_chain(const ortk_chain_args* p, ortk_stream stream) {
    if (!p || !p->w16 || !p->units_dev || !p->packed || p->packed_bytes < ortk::chain_packed_bytes(p->n_units)) return ORTK_EINVAL;
    hipStream_t s = ortk_s(stream);
    const bool wide = !p->progress && ortk::chain_wide(p->M);
    if (int e = ortk::chain_pack(p->w16, p->units_dev, p->n_units, p->packed, s, wide, p->a_in ? 1 : 0, p->n1, p->NC)) return e;
    return ortk::chain_run(p, p->packed, s);
}

// ortk_sparse.hip — sparse x dense product for >= 90 %-pruned weights:  Y = epi(X . W~^T),  W~ (N,K) in CSR.
//
// The reference runs pruned models as DENSE linears on zero-filled weights (scripts/eval_model.py:64-88,
// pruning/masked_layer.py:134-135).  At 95 % sparsity only 5 % of those multiply-adds touch a non-zero, so here each
// output column n gathers just its nnz(n) ~ 0.05 K input columns.
//
// Layout: the K axis is cut in chunks of KC = 512 columns; the CSR is "chunked" (row_ptr has nchunk*N+1 entries,
// entry c*N+n starts the non-zeros of row n whose column lies in chunk c; col holds the column RELATIVE to its
// chunk as uint16).  Every (chunk,row) list is padded with (col 0, val 0) entries to a multiple of 4 and the arrays
// carry 4 spare entries at the end: the kernel streams them as aligned 4-entry batches, one batch ahead of use.  A workgroup owns ROWS consecutive rows of X, stages their current K-chunk in LDS as
// [row][k] (pitch odd in dwords: lane = row reads are bank-conflict free whatever the gathered column), and its waves
// walk groups of 16 output columns: the (col, val) stream of a group is wave-uniform (scalar loads), each lane
// accumulates its row's 16 outputs in registers and writes them as one 64-byte segment.  For K > KC the chunks are
// processed in sequence by the same workgroup (same threads own the same outputs: plain read-modify-write, no atomics).
// HBM traffic = X once + Y once (+ 6 bytes per non-zero from L2): activation-bandwidth bound.
#include "ortk_common.h"

namespace {

constexpr int KC = 512;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

struct SpmmP {
    const int32_t* row_ptr; const uint16_t* col; const float* val;
    const void* X; int64_t ldx; int x_dt;
    const float* bias; const float* resid; int64_t ldr;
    void* Y; int64_t ldy; int y_dt;
    int64_t M; int N, K, relu;
};

// XT: element type staged in LDS (float or bf16). ROWS rows per workgroup; a wave serves 64/ROWS column groups at once.
template <typename XT, int ROWS>
__global__ __launch_bounds__(256) void spmm_csr_kernel(SpmmP p) {
    constexpr int EPD = 4 / (int)sizeof(XT);                 // elements per dword
    constexpr int PITCH = KC + EPD;                           // odd number of dwords
    constexpr int NQ = 64 / ROWS;                             // column groups per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XT* sX = reinterpret_cast<XT*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * ROWS;
    const int ml = lane % ROWS, q = lane / ROWS;
    const int64_t m = m0 + ml;
    const int nchunk = (p.K + KC - 1) / KC;
    const int ngroups = (p.N + 15) / 16;
    for (int c = 0; c < nchunk; ++c) {
        const int k0 = c * KC, kw = min(KC, p.K - k0);
        __syncthreads();
        // stage X[m0 .. m0+ROWS)[k0 .. k0+kw) -> sX[row][k]   (coalesced along k)
        for (int idx = tid; idx < ROWS * (KC / 4); idx += 256) {
            const int r = idx / (KC / 4), k = (idx - r * (KC / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < p.M) {
                const int64_t gi = (m0 + r) * p.ldx + k0 + k;
                if (k + 3 < kw && (gi & 3) == 0) v = ld_elem4(p.X, gi, p.x_dt);
                else {
                    if (k < kw) v.x = ld_elem(p.X, gi, p.x_dt);
                    if (k + 1 < kw) v.y = ld_elem(p.X, gi + 1, p.x_dt);
                    if (k + 2 < kw) v.z = ld_elem(p.X, gi + 2, p.x_dt);
                    if (k + 3 < kw) v.w = ld_elem(p.X, gi + 3, p.x_dt);
                }
            }
            XT* dst = sX + r * PITCH + k;
            dst[0] = (XT)v.x; dst[1] = (XT)v.y; dst[2] = (XT)v.z; dst[3] = (XT)v.w;
        }
        __syncthreads();
        const int32_t* rp = p.row_ptr + (int64_t)c * p.N;
        const bool first = c == 0, last = c == nchunk - 1;
        for (int ng = wave * NQ + q; ng < ngroups; ng += 4 * NQ) {
            float acc[16];
            int r[17];
#pragma unroll
            for (int t = 0; t <= 16; ++t) r[t] = rp[min(ng * 16 + t, p.N)];
            // Rows are padded to multiples of 4 entries (col 0, val 0) and the arrays end with 4 spare entries, so the
            // stream of the 16 columns is read as aligned 4-entry batches, always one batch ahead of its use.
            int j = r[0];
            f32x4 v_next = *reinterpret_cast<const f32x4*>(p.val + j);
            u16x4 c_next = *reinterpret_cast<const u16x4*>(p.col + j);
            const XT* xrow = sX + ml * PITCH;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float a0 = 0.f, a1 = 0.f;
                while (j < r[t + 1]) {
                    const f32x4 v = v_next; const u16x4 cc = c_next;
                    j += 4;
                    v_next = *reinterpret_cast<const f32x4*>(p.val + j);
                    c_next = *reinterpret_cast<const u16x4*>(p.col + j);
                    a0 += v[0] * (float)xrow[cc[0]]; a1 += v[1] * (float)xrow[cc[1]];
                    a0 += v[2] * (float)xrow[cc[2]]; a1 += v[3] * (float)xrow[cc[3]];
                }
                acc[t] = a0 + a1;
            }
            if (m < p.M) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int n = ng * 16 + t;
                    if (n < p.N) {
                        float y = acc[t];
                        const int64_t yi = m * p.ldy + n;
                        if (first) { if (p.bias) y += p.bias[n]; } else y += ld_elem(p.Y, yi, p.y_dt);
                        if (last) {
                            if (p.relu) y = fmaxf(y, 0.f);
                            if (p.resid) y += p.resid[m * p.ldr + n];
                        }
                        st_elem(p.Y, yi, p.y_dt, y);
                    }
                }
            }
        }
    }
}

template <typename XT>
int launch(const SpmmP& p, hipStream_t s) {
    constexpr int EPD = 4 / (int)sizeof(XT);
    const size_t per_row = (size_t)(KC + EPD) * sizeof(XT);
    // fewer rows per workgroup for small batches so that the grid still covers the 256 CUs
    const int rows = p.M >= 32768 ? 64 : (p.M >= 8192 ? 32 : 16);
    const dim3 block(256);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_csr_kernel<XT, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * per_row));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_csr_kernel<XT, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(32 * per_row));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_csr_kernel<XT, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * per_row));
        attr = true;
    }
    if (rows == 64) hipLaunchKernelGGL((spmm_csr_kernel<XT, 64>), dim3((unsigned)ortk_cdiv(p.M, 64)), block, 64 * per_row, s, p);
    else if (rows == 32) hipLaunchKernelGGL((spmm_csr_kernel<XT, 32>), dim3((unsigned)ortk_cdiv(p.M, 32)), block, 32 * per_row, s, p);
    else hipLaunchKernelGGL((spmm_csr_kernel<XT, 16>), dim3((unsigned)ortk_cdiv(p.M, 16)), block, 16 * per_row, s, p);
    ORTK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int ortk_spmm_csr(const ortk_csr* w, const void* X, int32_t x_dtype, int64_t ldx, const float* bias, void* Y,
                             int32_t y_dtype, int64_t ldy, int64_t M, int32_t relu, const float* resid, int64_t ldr,
                             ortk_stream stream) {
    if (!w || !w->row_ptr || !w->col || !w->val || !X || !Y || M < 0 || w->N < 1 || w->K < 1) return ORTK_EINVAL;
    if ((x_dtype != ORTK_F32 && x_dtype != ORTK_BF16) || (y_dtype != ORTK_F32 && y_dtype != ORTK_BF16)) return ORTK_EINVAL;
    if (M == 0) return 0;
    SpmmP p{w->row_ptr, w->col, w->val, X, ldx, x_dtype, bias, resid, ldr, Y, ldy, y_dtype, M, w->N, w->K, relu};
    // bf16 activations stay bf16 in LDS; fp32 activations stay fp32 (parity mode)
    return x_dtype == ORTK_BF16 ? launch<__bf16>(p, ortk_s(stream)) : launch<float>(p, ortk_s(stream));
}

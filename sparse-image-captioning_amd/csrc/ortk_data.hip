// ortk_data.hip — host-side batch assembly (no device code; built into libortk.so with the rest of the C-ABI).
// Follows torch.nn.utils.rnn.pad_sequence(batch_first=True, padding_value=0) as used by the reference's collate functions
// (data/collate.py:130-131,153-161,213): row-for-row copies into a zero-initialised (B, max_len, ...) array.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/ortk_data.h"

extern "C" int ortk_pad_rows(const float* const* rows, const int64_t* n_rows, int64_t B, int64_t F, int64_t smax, float* out, float* mask,
                             int32_t nthreads) {
    if (B < 0 || F < 1 || smax < 0 || (B > 0 && (!rows || !n_rows || !out))) return -1;
    for (int64_t i = 0; i < B; ++i) if (n_rows[i] < 0 || n_rows[i] > smax || (n_rows[i] > 0 && !rows[i])) return -1;
    if (B == 0 || smax == 0) return 0;
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, B));
    auto work = [=](int t) {
        for (int64_t i = t; i < B; i += nt) {
            float* dst = out + i * smax * F;
            const int64_t n = n_rows[i];
            if (n > 0) std::memcpy(dst, rows[i], (size_t)(n * F) * sizeof(float));
            if (n < smax) std::memset(dst + n * F, 0, (size_t)((smax - n) * F) * sizeof(float));
            if (mask) for (int64_t j = 0; j < smax; ++j) mask[i * smax + j] = j < n ? 1.f : 0.f;
        }
    };
    if (nt == 1) { work(0); return 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
    return 0;
}

extern "C" int ortk_pad_seqs(const int64_t* const* seqs, const int64_t* len, int64_t B, int64_t smax, int64_t pad, int64_t* out, float* mask) {
    if (B < 0 || smax < 0 || (B > 0 && (!seqs || !len || !out))) return -1;
    for (int64_t i = 0; i < B; ++i) if (len[i] < 0 || len[i] > smax || (len[i] > 0 && !seqs[i])) return -1;
    for (int64_t i = 0; i < B; ++i)
        for (int64_t j = 0; j < smax; ++j) {
            out[i * smax + j] = j < len[i] ? seqs[i][j] : pad;
            if (mask) mask[i * smax + j] = j < len[i] ? 1.f : 0.f;
        }
    return 0;
}

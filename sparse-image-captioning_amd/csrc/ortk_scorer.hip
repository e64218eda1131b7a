// ortk_scorer.hip — SCST reward scorer on the HOST (no device code in this file; it is built into libortk.so with the
// rest of the C-ABI): CIDEr-D and per-sentence BLEU-1..4 over integer n-grams, multi-threaded over items.
//
// Follows, operation for operation in double precision:
//   precook / cook_refs / cook_test / counts2vec / sim / compute_cider   ciderD_scorer.py:18-214
//   precook / cook_refs / cook_test / compute_score (per-sentence list)  bleu_scorer.py:24-90,202-243
// The reference keeps n-grams as tuples of words in Python dicts; iteration order there is insertion order (n-gram
// length major, then first occurrence), which the cooked captions here reproduce so that the floating-point sums run
// in the same order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/ortk_scorer.h"

namespace {

constexpr int MAXN = 4;

struct Cooked {
    int len = 0;                          // number of words
    std::vector<uint64_t> key;            // packed n-gram, insertion order
    std::vector<int> cnt;
    std::vector<int8_t> ord;              // n-gram length - 1
    // CIDEr side (filled once the document frequencies are known)
    std::vector<double> vec;
    double norm[MAXN] = {0, 0, 0, 0};
    int length = 0;                       // sum of bigram counts (ciderD_scorer.py:152-153: `if n == 1`)
};

inline uint64_t pack(const int32_t* w, int k) {
    uint64_t v = 0;
    for (int i = 0; i < k; ++i) v = (v << 16) | (uint64_t)(w[i] + 1);
    return v;
}

bool cook(const int32_t* w, int len, int n, Cooked& c) {
    c.len = len;
    for (int i = 0; i < len; ++i) if (w[i] < 0 || w[i] >= 65535) return false;
    for (int k = 1; k <= n; ++k) {
        const size_t first = c.key.size();
        for (int i = 0; i + k <= len; ++i) {
            const uint64_t kk = pack(w + i, k);
            size_t j = first;
            for (; j < c.key.size(); ++j) if (c.key[j] == kk) break;
            if (j == c.key.size()) { c.key.push_back(kk); c.cnt.push_back(1); c.ord.push_back((int8_t)(k - 1)); }
            else ++c.cnt[j];
        }
    }
    return true;
}

template <typename F>
void parallel_for(int64_t n, int nthreads, F f) {
    if (n <= 0) return;
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n));
    if (nt == 1) { for (int64_t i = 0; i < n; ++i) f(i); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=]() { for (int64_t i = t; i < n; i += nt) f(i); });
    for (auto& x : th) x.join();
}

}  // namespace

struct ortk_scorer {
    int n;
    double sigma;
    bool cached = false;
    double ref_len_log = 0.0;
    std::unordered_map<uint64_t, double> df;
};

extern "C" ortk_scorer* ortk_scorer_create(int32_t n, double sigma) {
    if (n < 1 || n > MAXN || !(sigma > 0.0)) return nullptr;
    ortk_scorer* s = new ortk_scorer();
    s->n = n; s->sigma = sigma;
    return s;
}
extern "C" void ortk_scorer_destroy(ortk_scorer* s) { delete s; }

extern "C" int ortk_scorer_set_df(ortk_scorer* s, const int32_t* tokens, const int64_t* key_off, const double* df, int64_t nkeys,
                                  double ref_len) {
    if (!s || nkeys < 0 || (nkeys > 0 && (!tokens || !key_off || !df)) || !(ref_len > 0.0)) return -1;
    s->df.clear();
    s->df.reserve((size_t)nkeys * 2);
    for (int64_t i = 0; i < nkeys; ++i) {
        const int64_t k = key_off[i + 1] - key_off[i];
        if (k < 1 || k > MAXN) return -1;
        for (int64_t j = key_off[i]; j < key_off[i + 1]; ++j) if (tokens[j] < 0 || tokens[j] >= 65535) return -1;
        s->df[pack(tokens + key_off[i], (int)k)] = df[i];
    }
    s->ref_len_log = std::log(ref_len);
    s->cached = true;
    return 0;
}

extern "C" int ortk_scorer_score(const ortk_scorer* s, const int32_t* cap_tok, const int64_t* cap_off, int64_t ncaps,
                                 const int64_t* hyp_cap, const int64_t* ref_cap, const int64_t* item_ref_off, int64_t nitems,
                                 double* cider_out, double* bleu_out, int32_t nthreads) {
    if (!s || !cap_off || !hyp_cap || !ref_cap || !item_ref_off || ncaps < 0 || nitems < 0) return -1;
    if (nitems == 0) return 0;
    for (int64_t i = 0; i < nitems; ++i) {
        if (hyp_cap[i] < 0 || hyp_cap[i] >= ncaps || item_ref_off[i + 1] <= item_ref_off[i]) return -1;
        for (int64_t r = item_ref_off[i]; r < item_ref_off[i + 1]; ++r) if (ref_cap[r] < 0 || ref_cap[r] >= ncaps) return -1;
    }
    const int n = s->n;
    std::vector<Cooked> caps((size_t)ncaps);
    std::vector<char> ok((size_t)ncaps, 1);
    // BLEU always uses 4-grams (BleuSilent(4), scorers.py:52); CIDEr uses the first n orders of the same cooked captions
    parallel_for(ncaps, nthreads, [&](int64_t c) {
        ok[c] = cook(cap_tok + cap_off[c], (int)(cap_off[c + 1] - cap_off[c]), MAXN, caps[c]) ? 1 : 0;
    });
    for (int64_t c = 0; c < ncaps; ++c) if (!ok[c]) return -1;

    if (cider_out) {
        // document frequencies: cached table, or "corpus" mode over this call's items (one document per item)
        std::unordered_map<uint64_t, double> local;
        const std::unordered_map<uint64_t, double>* df = &s->df;
        double ref_len_log = s->ref_len_log;
        if (!s->cached) {
            std::vector<uint64_t> seen;
            for (int64_t i = 0; i < nitems; ++i) {
                seen.clear();
                for (int64_t r = item_ref_off[i]; r < item_ref_off[i + 1]; ++r) {
                    const Cooked& c = caps[ref_cap[r]];
                    for (size_t j = 0; j < c.key.size(); ++j) if (c.ord[j] < n) seen.push_back(c.key[j]);
                }
                std::sort(seen.begin(), seen.end());
                seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
                for (uint64_t k : seen) local[k] += 1.0;
            }
            df = &local;
            ref_len_log = std::log((double)nitems);
        }
        // counts2vec (ciderD_scorer.py:131-155)
        parallel_for(ncaps, nthreads, [&](int64_t ci) {
            Cooked& c = caps[ci];
            c.vec.assign(c.key.size(), 0.0);
            double nsq[MAXN] = {0, 0, 0, 0};
            c.length = 0;
            for (size_t j = 0; j < c.key.size(); ++j) {
                const int o = c.ord[j];
                if (o >= n) continue;
                const auto it = df->find(c.key[j]);
                const double d = std::log(std::max(1.0, it == df->end() ? 0.0 : it->second));
                const double v = (double)c.cnt[j] * (ref_len_log - d);
                c.vec[j] = v;
                nsq[o] += std::pow(v, 2);
                if (o == 1) c.length += c.cnt[j];
            }
            for (int o = 0; o < MAXN; ++o) c.norm[o] = std::sqrt(nsq[o]);
        });
        const double sigma = s->sigma;
        parallel_for(nitems, nthreads, [&](int64_t i) {
            const Cooked& h = caps[hyp_cap[i]];
            double score[MAXN] = {0, 0, 0, 0};
            const int64_t nref = item_ref_off[i + 1] - item_ref_off[i];
            for (int64_t r = item_ref_off[i]; r < item_ref_off[i + 1]; ++r) {
                const Cooked& rf = caps[ref_cap[r]];
                const double delta = (double)(h.length - rf.length);
                double val[MAXN] = {0, 0, 0, 0};
                for (size_t j = 0; j < h.key.size(); ++j) {
                    const int o = h.ord[j];
                    if (o >= n) continue;
                    double vr = 0.0;
                    for (size_t q = 0; q < rf.key.size(); ++q) if (rf.key[q] == h.key[j]) { vr = rf.vec[q]; break; }
                    val[o] += std::min(h.vec[j], vr) * vr;          // clipping (ciderD_scorer.py:176)
                }
                for (int o = 0; o < n; ++o) {
                    if (h.norm[o] != 0.0 && rf.norm[o] != 0.0) val[o] /= h.norm[o] * rf.norm[o];
                    val[o] *= std::pow(M_E, -(delta * delta) / (2.0 * sigma * sigma));
                    score[o] += val[o];
                }
            }
            double sum = 0.0;
            for (int o = 0; o < n; ++o) sum += score[o];
            double avg = sum / (double)n;
            avg /= (double)nref;
            avg *= 10.0;
            cider_out[i] = avg;
        });
    }

    if (bleu_out) {
        const double small = 1e-9, tiny = 1e-15;
        parallel_for(nitems, nthreads, [&](int64_t i) {
            const Cooked& h = caps[hyp_cap[i]];
            const int testlen = h.len;
            // closest reference length: min over (|l - testlen|, l) (bleu_scorer.py:74-75)
            int best_d = 1 << 30, reflen = 0;
            for (int64_t r = item_ref_off[i]; r < item_ref_off[i + 1]; ++r) {
                const int l = caps[ref_cap[r]].len, dd = std::abs(l - testlen);
                if (dd < best_d || (dd == best_d && l < reflen)) { best_d = dd; reflen = l; }
            }
            int correct[4] = {0, 0, 0, 0};
            for (size_t j = 0; j < h.key.size(); ++j) {
                int mx = 0;
                for (int64_t r = item_ref_off[i]; r < item_ref_off[i + 1]; ++r) {
                    const Cooked& rf = caps[ref_cap[r]];
                    for (size_t q = 0; q < rf.key.size(); ++q) if (rf.key[q] == h.key[j]) { mx = std::max(mx, rf.cnt[q]); break; }
                }
                correct[h.ord[j]] += std::min(mx, h.cnt[j]);
            }
            double bleu = 1.0, out[4];
            for (int k = 0; k < 4; ++k) {
                const int guess = std::max(0, testlen - k);
                bleu *= ((double)correct[k] + tiny) / ((double)guess + small);
                out[k] = std::pow(bleu, 1.0 / (double)(k + 1));
            }
            const double ratio = ((double)testlen + tiny) / ((double)reflen + small);
            if (ratio < 1.0) for (int k = 0; k < 4; ++k) out[k] *= std::exp(1.0 - 1.0 / ratio);
            for (int k = 0; k < 4; ++k) bleu_out[(int64_t)k * nitems + i] = out[k];
        });
    }
    return 0;
}

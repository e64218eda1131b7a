// ortk_model.hip — the native executor: parameter-arena layout, workspace carving, and the teacher-forced
// forward / fused loss / backward of the Object Relation Transformer as ONE sequence of HIP launches per call.
//
// Follows RelationTransformerModel._forward (models/relation_transformer.py:341-372), the encoder / decoder blocks
// (relation_transformer.py:77-113,148-191; models/transformer.py:172-210,315-358) and LanguageModelCriterion /
// RewardCriterion (utils/losses.py:15-43); the step order of the backward is the reverse of that graph.
// Design differences from the reference that do not change results:
//   * Q,K,V projections are one N=3d GEMM; cross-attention K/V of all 6 decoder layers are one N=12d GEMM over
//     the B*S encoder rows (the reference re-projects the repeated memory per caption row: 5x the work);
//   * the geometry bias is computed once for all layers by one kernel;
//   * bias / ReLU / dropout / residual are GEMM epilogues; the criterion never materialises log-probs.
#include <string>
#include <vector>
#include <mutex>
#include <utility>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "ortk_internal.h"

namespace ortk {

constexpr int MAXLAYERS = 16;

struct EncOff { int64_t wqkv, bqkv, wo, bo, wg, bg, w1, b1, w2, b2, n0a, n0b, n1a, n1b; };
struct DecOff { int64_t wqkv, bqkv, wo, bo, cqw, cqb, cow, cob, w1, b1, w2, b2, n0a, n0b, n1a, n1b, n2a, n2b; };
struct Offsets {
    int64_t att_w, att_b;
    EncOff enc[MAXLAYERS];
    int64_t enc_na, enc_nb;
    DecOff dec[MAXLAYERS];
    int64_t dec_na, dec_nb;
    int64_t ckv_w, ckv_b;
    int ckv_slot[MAXLAYERS];   // decoder layer -> its slice of the packed cross-attention K|V block (shared layers: same slice)
    int ckv_slots;             // number of distinct decoder layers
    int64_t cw;                // width of one slice: 2d ([K|V]; share_att "qk": [Q=K|V]), d with share_att "kv" (K = V)
    int64_t cv;                // column of V inside a slice (d, or 0 with "kv")
    int64_t lut, gen_w, gen_b;
    int64_t total;      // trainable floats (gradient / Adam mirrors have this size)
    int64_t pe;         // positional-encoding BUFFER (1, PE_ROWS, d), stored after the trainable part
    int64_t total_all;  // trainable + buffers
};
constexpr int64_t PE_ROWS = 5000;  // transformer.py:365 (max_len), kept for state_dict compatibility
struct Entry { std::string name; int64_t offset, numel; int ndim; int64_t shape[4]; int kind; };  // kind: 0 param, 1 maskable param, 2 buffer

// Projection sharing inside an attention module (ortk_config.share_att_*): which d-wide column block of the packed
// projection output holds Q (always 0), K and V; and where the attention backward puts dQ (0), dK, dV so that, after
// adding block `gsrc` into block `gdst`, the first n blocks are the gradient of the packed projection output.
struct AttMode { int n, k, v, gk, gv, gsrc, gdst; };
static inline AttMode att_mode(int m) {
    if (m == 1) return AttMode{2, 1, 1, 1, 2, 2, 1};      // "kv": [Q | K=V]
    if (m == 2) return AttMode{2, 0, 1, 2, 1, 2, 0};      // "qk": [Q=K | V]
    return AttMode{3, 1, 2, 1, 2, -1, -1};
}

static int check_cfg(const ortk_config* c) {
    if (!c) return ORTK_EINVAL;
    if (c->d_model < 8 || c->d_model > 2048 || c->d_ff < 1 || c->n_layers < 1 || c->n_layers > MAXLAYERS) return ORTK_EINVAL;
    if (c->n_heads < 1 || c->n_heads > 8 || c->d_model % c->n_heads) return ORTK_EINVAL;
    if (c->d_model / c->n_heads > 64) return ORTK_EINVAL;
    if (c->vocab < 2 || c->feat < 1 || c->seq_len < 1 || c->seq_len > 64) return ORTK_EINVAL;
    if (c->precision != 0 && c->precision != 1) return ORTK_EINVAL;
    if (c->share_att_enc < 0 || c->share_att_enc > 2 || c->share_att_dec < 0 || c->share_att_dec > 2) return ORTK_EINVAL;
    for (int l = 0; l < c->n_layers; ++l)
        for (const int32_t* sh : {c->share_enc, c->share_dec}) {
            const int k = sh[l];
            if (k < 0 || k > l || (k > 0 && sh[k - 1] != 0)) return ORTK_EINVAL;   // shares an earlier, itself unshared layer
        }
    return 0;
}

static void build_layout(const ortk_config& c, Offsets& o, std::vector<Entry>* entries) {
    const int64_t d = c.d_model, ff = c.d_ff, V = c.vocab, F = c.feat;
    const int L = c.n_layers, H = c.n_heads;
    int64_t off = 0;
    auto align = [&]() { off = ortk_align(off, 64); };
    auto add = [&](const std::string& name, std::initializer_list<int64_t> shp) -> int64_t {
        Entry e; e.name = name; e.offset = off; e.ndim = (int)shp.size(); e.numel = 1;
        int i = 0; for (int64_t s : shp) { e.shape[i++] = s; e.numel *= s; }
        for (; i < 4; ++i) e.shape[i] = 1;
        e.kind = e.ndim >= 2 ? 1 : 0;
        if (entries) entries->push_back(e);
        const int64_t at = off; off += e.numel; return at;
    };
    auto L_ = [](const std::string& p, int i, const char* w) { return p + ".linears." + std::to_string(i) + "." + w; };
    // packed projections per attention module: 3 (Q, K, V) or, with share_att, 2; the output projection is the next linear
    const int ne = att_mode(c.share_att_enc).n, nd = att_mode(c.share_att_dec).n;
    const bool cqk = c.share_att_dec == 2;     // cross-attention Q and K share linears.0: it lives in the packed K|V block
    o.cw = c.share_att_dec == 1 ? d : 2 * d; o.cv = c.share_att_dec == 1 ? 0 : d;
    const bool plain = c.no_box != 0;
    const std::string root = plain ? "core." : "model.", emb = plain ? "core.src_embed.0." : "att_embed.0.";
    align(); o.att_w = add(emb + "weight", {d, F});
    align(); o.att_b = add(emb + "bias", {d});
    for (int l = 0; l < L; ++l) {
        const std::string p = root + "encoder.layers." + std::to_string(l);
        EncOff& e = o.enc[l];
        if (c.share_enc[l] > 0) { e = o.enc[c.share_enc[l] - 1]; continue; }      // the same module: no storage of its own
        align(); e.wqkv = off; for (int i = 0; i < ne; ++i) add(L_(p + ".self_attn", i, "weight"), {d, d});
        align(); e.bqkv = off; for (int i = 0; i < ne; ++i) add(L_(p + ".self_attn", i, "bias"), {d});
        align(); e.wo = add(L_(p + ".self_attn", ne, "weight"), {d, d});
        align(); e.bo = add(L_(p + ".self_attn", ne, "bias"), {d});
        const int64_t dg = c.box_trig ? 64 : 4;   // relation_transformer.py:131-136
        e.wg = e.bg = -1;
        if (!plain) {
            align(); e.wg = off; for (int h = 0; h < H; ++h) add(p + ".self_attn.WGs." + std::to_string(h) + ".weight", {1, dg});
            align(); e.bg = off; for (int h = 0; h < H; ++h) add(p + ".self_attn.WGs." + std::to_string(h) + ".bias", {1});
        }
        align(); e.w1 = add(p + ".feed_forward.w_1.weight", {ff, d});
        align(); e.b1 = add(p + ".feed_forward.w_1.bias", {ff});
        align(); e.w2 = add(p + ".feed_forward.w_2.weight", {d, ff});
        align(); e.b2 = add(p + ".feed_forward.w_2.bias", {d});
        align(); e.n0a = add(p + ".sublayer.0.norm.a_2", {d}); align(); e.n0b = add(p + ".sublayer.0.norm.b_2", {d});
        align(); e.n1a = add(p + ".sublayer.1.norm.a_2", {d}); align(); e.n1b = add(p + ".sublayer.1.norm.b_2", {d});
    }
    align(); o.enc_na = add(root + "encoder.norm.a_2", {d}); align(); o.enc_nb = add(root + "encoder.norm.b_2", {d});
    o.ckv_slots = 0;
    for (int l = 0; l < L; ++l) o.ckv_slot[l] = c.share_dec[l] > 0 ? o.ckv_slot[c.share_dec[l] - 1] : o.ckv_slots++;
    for (int l = 0; l < L; ++l) {
        const std::string p = root + "decoder.layers." + std::to_string(l);
        DecOff& e = o.dec[l];
        if (c.share_dec[l] > 0) { e = o.dec[c.share_dec[l] - 1]; continue; }
        align(); e.wqkv = off; for (int i = 0; i < nd; ++i) add(L_(p + ".self_attn", i, "weight"), {d, d});
        align(); e.bqkv = off; for (int i = 0; i < nd; ++i) add(L_(p + ".self_attn", i, "bias"), {d});
        align(); e.wo = add(L_(p + ".self_attn", nd, "weight"), {d, d});
        align(); e.bo = add(L_(p + ".self_attn", nd, "bias"), {d});
        e.cqw = e.cqb = -1;                   // "qk": set below, inside the packed K|V block
        if (!cqk) { align(); e.cqw = add(L_(p + ".src_attn", 0, "weight"), {d, d}); align(); e.cqb = add(L_(p + ".src_attn", 0, "bias"), {d}); }
        align(); e.cow = add(L_(p + ".src_attn", nd, "weight"), {d, d});
        align(); e.cob = add(L_(p + ".src_attn", nd, "bias"), {d});
        align(); e.w1 = add(p + ".feed_forward.w_1.weight", {ff, d});
        align(); e.b1 = add(p + ".feed_forward.w_1.bias", {ff});
        align(); e.w2 = add(p + ".feed_forward.w_2.weight", {d, ff});
        align(); e.b2 = add(p + ".feed_forward.w_2.bias", {d});
        align(); e.n0a = add(p + ".sublayer.0.norm.a_2", {d}); align(); e.n0b = add(p + ".sublayer.0.norm.b_2", {d});
        align(); e.n1a = add(p + ".sublayer.1.norm.a_2", {d}); align(); e.n1b = add(p + ".sublayer.1.norm.b_2", {d});
        align(); e.n2a = add(p + ".sublayer.2.norm.a_2", {d}); align(); e.n2b = add(p + ".sublayer.2.norm.b_2", {d});
    }
    align(); o.dec_na = add(root + "decoder.norm.a_2", {d}); align(); o.dec_nb = add(root + "decoder.norm.b_2", {d});
    // cross-attention K / V projections of ALL distinct decoder layers: one (U*cw, d) matrix.  Slice of a layer:
    // [W_k; W_v] (linears.1, .2) | "kv": [W_kv] (linears.1) | "qk": [W_qk; W_v] (linears.0, .1; also the query projection)
    const int c0 = cqk ? 0 : 1, c1 = c.share_att_dec == 1 ? 1 : c0 + 1;
    align(); o.ckv_w = off;
    for (int l = 0; l < L; ++l) {
        if (c.share_dec[l] > 0) continue;
        const std::string p = root + "decoder.layers." + std::to_string(l) + ".src_attn";
        if (cqk) o.dec[l].cqw = off;
        for (int i = c0; i <= c1; ++i) add(L_(p, i, "weight"), {d, d});
    }
    align(); o.ckv_b = off;
    for (int l = 0; l < L; ++l) {
        if (c.share_dec[l] > 0) continue;
        const std::string p = root + "decoder.layers." + std::to_string(l) + ".src_attn";
        if (cqk) o.dec[l].cqb = off;
        for (int i = c0; i <= c1; ++i) add(L_(p, i, "bias"), {d});
    }
    if (cqk) for (int l = 0; l < L; ++l) if (c.share_dec[l] > 0) { o.dec[l].cqw = o.dec[c.share_dec[l] - 1].cqw; o.dec[l].cqb = o.dec[c.share_dec[l] - 1].cqb; }
    align(); o.lut = add(root + "tgt_embed.0.lut.weight", {V, d});
    // The generator rows are padded (with zeros that no state_dict entry covers) to a multiple of the GEMM tile:
    // logits are computed for Vp = align(V,128) columns with full tiles; the pad logits are exactly 0, have zero
    // gradient, and are ignored by the soft-max / criterion / decoders, which all take V.
    const int64_t Vp = ortk_align(V, 128);
    align(); o.gen_w = add(root + "generator.proj.weight", {V, d}); off += (Vp - V) * d;
    align(); o.gen_b = add(root + "generator.proj.bias", {V}); off += (Vp - V);
    align(); o.total = off;
    o.pe = add(root + "tgt_embed.1.pe", {1, PE_ROWS, d});
    if (entries) entries->back().kind = 2;
    align(); o.total_all = off;
}

// ------------------------------------------------------------------------------------------------ workspace
// ------------------------------------------------------------------------------------------------ rows-stationary chains
// The row-wise operators between two attention calls as ONE launch each (ortk_chain.hip): mixed precision, d_model 512, d_ff a
// multiple of 512, unshared projections, dense products.  Chains of the encoder: E0 = [LN -> Wqkv] of layer 0, EB(l) = [Wo + x ->
// LN -> FFN + x -> LN of layer l + 1 (or the stack's) -> Wqkv of layer l + 1]; of the decoder: D0 = [LN -> Wqkv] of layer 0,
// DB(l) = [Wo + x -> LN -> Wcq], DC(l) = [Wco + x -> LN -> FFN + x -> next LN -> next Wqkv].  Their weight units are packed into
// streaming order once per forward (chain_pack_all, from the bf16 weight copy of the call).
struct ChainSet {
    bool on = false;
    ChainPackTable t;
    int e0 = -1, eb[MAXLAYERS], d0 = -1, db[MAXLAYERS], dc[MAXLAYERS];
    // backward chains (bchain_layout): X = [FFN' -> LN' -> . Wout] on a masked gradient, Y = [dq . Wcq -> LN' -> . Wo],
    // ZX(l) = [dQKV . Wqkv -> LN' of layer l] + X of layer l - 1, Z0 = the bottom of a stack
    int bx_top = -1, by[MAXLAYERS], bzx[MAXLAYERS], bz0 = -1, ex_top = -1, ezx[MAXLAYERS], ez0 = -1;
    size_t bytes = 0;
    int units(int c) const { return t.first[c + 1] - t.first[c]; }
    const void* stream_of(const void* pk, int c) const { return reinterpret_cast<const char*>(pk) + (size_t)t.base[c] * 16; }
};
static bool chain_cfg_ok(const ortk_config& c, bool ignore_switch = false) {
    return (ignore_switch || tuning().row_chain) && c.precision == 1 && c.d_model == 512 && c.d_ff % 512 == 0 && c.d_ff / 512 <= 8 && c.n_heads == 8 &&
           c.share_att_enc == 0 && c.share_att_dec == 0 && c.n_layers <= 6;
}
// Me / Md: the rows the encoder's / the decoder's chains will run on (they pick the 48- or the 76-row form of the kernel, and with it the
// streaming order of the units: ortk::chain_wide)
// Which forward last built a data-gradient plan's images (the plan's buffers belong to the model, not to a workspace): the forward
// queues the build beside its own work, the backward rebuilds only when another forward — a second autograd graph, another mask
// sample — has done so since.  Host-side bookkeeping in call order = stream order on the caller's stream.
struct BwdPlanOwner { const ortk_sparse_plan* plan; const void* ws; uint64_t seed; int train; };
static std::mutex g_bwd_owner_mu;
static BwdPlanOwner g_bwd_owner[8] = {};
static void bwd_plan_built_by(const ortk_sparse_plan* plan, const void* ws, uint64_t seed, int train) {
    std::lock_guard<std::mutex> g(g_bwd_owner_mu);
    int slot = 0;
    for (int i = 0; i < 8; ++i) { if (g_bwd_owner[i].plan == plan) { slot = i; break; } if (!g_bwd_owner[i].plan) slot = i; }
    g_bwd_owner[slot] = BwdPlanOwner{plan, ws, seed, train};
}
static void bwd_plan_forget(const ortk_sparse_plan* plan) {
    std::lock_guard<std::mutex> g(g_bwd_owner_mu);
    for (auto& o : g_bwd_owner) if (o.plan == plan) o = BwdPlanOwner{};
}
static bool bwd_plan_is_of(const ortk_sparse_plan* plan, const void* ws, uint64_t seed, int train) {
    std::lock_guard<std::mutex> g(g_bwd_owner_mu);
    for (const auto& o : g_bwd_owner) if (o.plan == plan) return o.ws == ws && o.seed == seed && o.train == train;
    return false;
}
static bool plan_hits_chain(const ortk_sparse_plan* plan, const ChainSet& cs) {
    if (!plan || !cs.on) return false;
    const int nu = cs.t.first[cs.t.n_chains];
    for (int i = 0; i < plan->nblocks; ++i)
        for (int u = 0; u < nu; ++u)
            if (plan->blocks_host[i].src_offset == cs.t.u[u].offset) return true;       // (a block starts where its first unit starts)
    return false;
}
static void chain_layout(const ortk_config& c, const Offsets& o, bool enc, bool dec, ChainSet& cs, int64_t Me, int64_t Md, bool sizing = false) {
    cs.on = chain_cfg_ok(c);
    cs.bytes = 0;
    if (!cs.on && !(sizing && chain_cfg_ok(c, true))) return;
    const int L = c.n_layers, NC = c.d_ff / 512, ff = c.d_ff;
    ChainPackTable& t = cs.t;
    t.n_chains = 0; t.first[0] = 0;
    int nu = 0;
    int64_t base = 0;
    auto unit = [&](int64_t off, int ld) { t.u[nu].offset = (int32_t)off; t.u[nu].ld = ld; t.form[nu] = 0; ++nu; };
    // n_r / n1 / nc: the chain's shape (ortk_chain_args), which fixes the form of each of its units
    auto close = [&](bool wide, int n_r, int n1, int nc) {
        const int c_ = t.n_chains++;
        t.base[c_] = base; t.first[c_ + 1] = nu;
        for (int u = t.first[c_]; u < nu; ++u) t.form[u] = chain_unit_form(wide, u - t.first[c_], n_r, n1, nc);
        base += (int64_t)8 * (nu - t.first[c_] + 1) * 16 * 256;           // uint4 per unit (+ one of slack): 16 k-steps x 32 column tiles x 64 lanes
        return c_;
    };
    auto qkv = [&](int64_t w) { for (int i = 0; i < 3; ++i) unit(w + (int64_t)i * 512 * 512, 512); };
    auto ffn = [&](int64_t w1, int64_t w2) { for (int k = 0; k < NC; ++k) { unit(w1 + (int64_t)k * 512 * 512, 512); unit(w2 + (int64_t)k * 512, ff); } };
    if (enc) {
        const bool wide = Me > 0 && chain_wide(Me);
        qkv(o.enc[0].wqkv); cs.e0 = close(wide, 0, 3, 0);
        for (int l = 0; l < L; ++l) {
            unit(o.enc[l].wo, 512); ffn(o.enc[l].w1, o.enc[l].w2);
            if (l + 1 < L) qkv(o.enc[l + 1].wqkv);
            cs.eb[l] = close(wide, 1, 0, NC);
        }
    }
    if (dec) {
        const bool wide = Md > 0 && chain_wide(Md);
        qkv(o.dec[0].wqkv); cs.d0 = close(wide, 0, 3, 0);
        for (int l = 0; l < L; ++l) {
            unit(o.dec[l].wo, 512); unit(o.dec[l].cqw, 512); cs.db[l] = close(wide, 1, 1, 0);
            unit(o.dec[l].cow, 512); ffn(o.dec[l].w1, o.dec[l].w2);
            if (l + 1 < L) qkv(o.dec[l + 1].wqkv);
            cs.dc[l] = close(wide, 1, 0, NC);
        }
    }
    cs.bytes = (size_t)base * 16;
}
// units over the TRANSPOSED bf16 weight copy (block (N, K) of the arena stored (K, N) at the same offset)
static void bchain_layout(const ortk_config& c, const Offsets& o, ChainSet& cs, bool sizing = false) {
    cs.on = chain_cfg_ok(c);
    cs.bytes = 0;
    if (!cs.on && !(sizing && chain_cfg_ok(c, true))) return;
    const int L = c.n_layers, NC = c.d_ff / 512, ff = c.d_ff;
    ChainPackTable& t = cs.t;
    t.n_chains = 0; t.first[0] = 0;
    int nu = 0;
    int64_t base = 0;
    auto unit = [&](int64_t off, int ld) { t.u[nu].offset = (int32_t)off; t.u[nu].ld = ld; t.form[nu] = 0; ++nu; };
    auto close = [&]() {
        const int c_ = t.n_chains++;
        t.base[c_] = base; t.first[c_ + 1] = nu;
        base += (int64_t)8 * (nu - t.first[c_] + 1) * 16 * 256;
        return c_;
    };
    auto zq = [&](int64_t wqkv) { for (int i = 0; i < 3; ++i) unit(wqkv + (int64_t)i * 512, 1536); };          // dQKV . Wqkv: columns 512 i .. of Wqkv^T
    auto xf = [&](int64_t w1, int64_t w2, int64_t wout) {
        for (int k = 0; k < NC; ++k) { unit(w2 + (int64_t)k * 512 * 512, 512); unit(w1 + (int64_t)k * 512, ff); }
        unit(wout, 512);
    };
    // decoder
    xf(o.dec[L - 1].w1, o.dec[L - 1].w2, o.dec[L - 1].cow); cs.bx_top = close();
    for (int l = L - 1; l >= 0; --l) {
        unit(o.dec[l].cqw, 512); unit(o.dec[l].wo, 512); cs.by[l] = close();
        zq(o.dec[l].wqkv);
        if (l > 0) { xf(o.dec[l - 1].w1, o.dec[l - 1].w2, o.dec[l - 1].cow); cs.bzx[l] = close(); }
        else cs.bz0 = close();
    }
    // encoder
    xf(o.enc[L - 1].w1, o.enc[L - 1].w2, o.enc[L - 1].wo); cs.ex_top = close();
    for (int l = L - 1; l >= 0; --l) {
        zq(o.enc[l].wqkv);
        if (l > 0) { xf(o.enc[l - 1].w1, o.enc[l - 1].w2, o.enc[l - 1].wo); cs.ezx[l] = close(); }
        else cs.ez0 = close();
    }
    cs.bytes = (size_t)base * 16;
}

struct Bump {
    char* base; size_t off;
    template <typename T> T* take(int64_t n) { return reinterpret_cast<T*>(take_bytes((size_t)n * sizeof(T))); }
    void* take_bytes(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

// Buffers marked (A) hold MFMA operands only: bf16 in mixed precision (cfg.precision = 1), fp32 otherwise.
// Buffers marked (Q) — the projected Q / K / V, which only feed the attention products — are bf16 in mixed precision when
// the attention of their stack is served by the bf16-operand kernels (TrainWS.qdt_*), fp32 otherwise.
struct EncBuf {   // per encoder layer
    void *y1 /*A*/; void* qkv /*Q*/; float* P; void* o /*A*/; float* xm; void* y2 /*A*/; void* h /*A*/; float* xout; float *st1, *st2;
};
struct DecBuf {   // per decoder layer
    void* y1 /*A*/; void* qkv /*Q*/; float* Ps; void* o1 /*A*/; float* xm1; void* y2 /*A*/; void* qc /*Q*/; float* Pc; void* o2 /*A*/;
    float* xm2; void* y3 /*A*/; void* h /*A*/; float* xout; float *st1, *st2, *st3;
};
struct TrainWS {
    int64_t Me, Md, ldv; int adt;
    int qdt_enc, qdt_self, qdt_cross;       // element type of the (Q) buffers of the three attention stacks
    void* w16;                              // bf16 working copy of the weight arena (mixed precision)
    void* w16t;                             // ... and of every weight block TRANSPOSED (operand of the data-gradient GEMMs)
    float *x0, *logbias, *dscore; void* mem /*A*/; float* st_mem;
    EncBuf enc[MAXLAYERS];
    float *dx0, *keymask; void* ckv /*Q*/; void* dec_out /*A*/; float *st_out, *logits, *row_loss /*criterion terms, one per decoder row*/; void* dlogits /*A; aliases logits in fp32*/;
    DecBuf dec[MAXLAYERS];
    // backward temporaries
    float *ga, *gb, *gy; void* gdo /*Q: dO of an attention backward*/; void *gt /*A*/, *gt2 /*A*/, *gt3 /*A*/, *gqkv /*A*/, *gh /*A*/, *gkv /*A*/; float* scalar;
    void* chain_pk = nullptr;               // weight units of the rows-stationary chains in streaming order (chain_layout; mixed precision)
    void* chain_pk_b = nullptr;             // ... of the backward chains (bchain_layout, from the transposed copy)
    void* gh2 = nullptr;                    // second FFN hidden-gradient buffer: the backward chains alternate (a weight gradient still reads the other)
    // grouped weight gradients (ortk_wgrad_group, mixed precision): a layer's dY operands stay alive until the layer's ONE launch on the
    // side stream has read them, while the next layer's are already being written
    static constexpr int NGT = 10;          // (rows, d) gradient temporaries in rotation: 4 per decoder layer, two layers alive + slack
    void* gtp[NGT] = {};
    void* gqkv2 = nullptr;                  // layers alternate between gqkv / gqkv2 and gh / gh2
    void* wg_ws = nullptr; size_t wg_ws_bytes = 0;      // partial tiles + tickets of one grouped launch (launches are ordered by their stream)
    void* feats16 = nullptr;                // bf16 copy of the region features (mixed precision): operand of att_embed and of its weight gradient
    size_t bytes;
};

static void carve_train(const ortk_config& c, int B, int S, int R, int T, void* base, TrainWS& w) {
    const int64_t d = c.d_model, ff = c.d_ff, H = c.n_heads, L = c.n_layers;
    const int64_t Me = (int64_t)B * S, Md = (int64_t)R * T, Mx = Me > Md ? Me : Md;
    const int64_t spi = B > 0 ? R / B : 1;
    w.Me = Me; w.Md = Md; w.ldv = ortk_align(c.vocab, 128);
    w.adt = c.precision ? ORTK_BF16 : ORTK_F32;
    const size_t es = ortk_esize(w.adt);
    Bump b{reinterpret_cast<char*>(base), 0};
    auto act = [&](int64_t n) { return b.take_bytes((size_t)n * es); };
    const int dk_ = (int)(d / H);
    w.qdt_enc = (c.precision && attn16_shape_ok(S, S, dk_)) ? ORTK_BF16 : ORTK_F32;
    w.qdt_self = (c.precision && attn16_shape_ok(T, T, dk_)) ? ORTK_BF16 : ORTK_F32;
    w.qdt_cross = (c.precision && attn16_shape_ok((int)(spi * T), S, dk_)) ? ORTK_BF16 : ORTK_F32;
    auto qbuf = [&](int64_t n, int dt) { return b.take_bytes((size_t)n * ortk_esize(dt)); };
    Offsets o; build_layout(c, o, nullptr);
    w.w16 = c.precision ? b.take_bytes((size_t)o.total * 2) : nullptr;
    w.w16t = c.precision ? b.take_bytes((size_t)o.total * 2) : nullptr;
    w.x0 = b.take<float>(Me * d);
    w.logbias = b.take<float>(L * B * H * S * S);
    w.dscore = b.take<float>(L * B * H * S * S);
    for (int l = 0; l < L; ++l) {
        EncBuf& e = w.enc[l];
        e.y1 = act(Me * d); e.qkv = qbuf(Me * 3 * d, w.qdt_enc); e.P = b.take<float>((int64_t)B * H * S * S);
        e.o = act(Me * d); e.xm = b.take<float>(Me * d); e.y2 = act(Me * d);
        e.h = act(Me * ff); e.xout = b.take<float>(Me * d);
        e.st1 = b.take<float>(Me * 2); e.st2 = b.take<float>(Me * 2);
    }
    w.mem = act(Me * d); w.st_mem = b.take<float>(Me * 2);
    w.dx0 = b.take<float>(Md * d); w.keymask = b.take<float>(Md);
    w.ckv = qbuf(Me * L * 2 * d, w.qdt_cross);
    for (int l = 0; l < L; ++l) {
        DecBuf& e = w.dec[l];
        e.y1 = act(Md * d); e.qkv = qbuf(Md * 3 * d, w.qdt_self); e.Ps = b.take<float>((int64_t)R * H * T * T);
        e.o1 = act(Md * d); e.xm1 = b.take<float>(Md * d); e.y2 = act(Md * d);
        e.qc = qbuf(Md * d, w.qdt_cross); e.Pc = b.take<float>((int64_t)B * H * spi * T * S);
        e.o2 = act(Md * d); e.xm2 = b.take<float>(Md * d); e.y3 = act(Md * d);
        e.h = act(Md * ff); e.xout = b.take<float>(Md * d);
        e.st1 = b.take<float>(Md * 2); e.st2 = b.take<float>(Md * 2); e.st3 = b.take<float>(Md * 2);
    }
    w.dec_out = act(Md * d); w.st_out = b.take<float>(Md * 2);
    w.logits = b.take<float>(Md * w.ldv);
    w.row_loss = b.take<float>(xent_scratch_floats(Md));
    w.dlogits = c.precision ? act(Md * w.ldv) : (void*)w.logits;
    w.ga = b.take<float>(Mx * d); w.gb = b.take<float>(Mx * d); w.gy = b.take<float>(Mx * d);
    w.gdo = c.precision ? act(Mx * d) : nullptr;
    w.gt = act(Mx * d); w.gt2 = act(Mx * d); w.gt3 = act(Mx * d); w.gqkv = act(Mx * 3 * d); w.gh = act(Mx * ff); w.gkv = act(Me * L * 2 * d);
    w.scalar = b.take<float>(64);
    {   // (sized whatever the tuning switch says: the workspace size is a function of the configuration and the shapes only)
        ChainSet cs; chain_layout(c, o, true, true, cs, 0, 0, true);
        if (cs.bytes) w.chain_pk = b.take_bytes(cs.bytes);
        ChainSet bs; bchain_layout(c, o, bs, true);
        if (bs.bytes) { w.chain_pk_b = b.take_bytes(bs.bytes); w.gh2 = act(Mx * ff); }
    }
    if (c.precision) {
        w.gtp[0] = w.gt; w.gtp[1] = w.gt2; w.gtp[2] = w.gt3;
        for (int i = 3; i < TrainWS::NGT; ++i) w.gtp[i] = act(Mx * d);
        w.gqkv2 = act(Mx * 3 * d);
        if (!w.gh2) w.gh2 = act(Mx * ff);
        // the largest group: a decoder layer (6 projections), an encoder layer, the generator, the memory's K|V projection
        const int64_t lay_tiles = 2 * ortk_cdiv(d, 256) * ortk_cdiv(ff, 256) + 4 * ortk_cdiv(d, 256) * ortk_cdiv(d, 256) + ortk_cdiv(3 * d, 256) * ortk_cdiv(d, 256);
        const int64_t gen_tiles = ortk_cdiv(w.ldv, 256) * ortk_cdiv(d, 256), kv_tiles = ortk_cdiv(L * 2 * d, 256) * ortk_cdiv(d, 256);
        const int64_t tiles = std::max(lay_tiles, std::max(gen_tiles, kv_tiles));
        // (at most max(256, tiles) partial tiles per launch: the K ranges are chosen for one round of workgroups; a forced split of 8 at most)
        w.wg_ws_bytes = (size_t)(((tiles * 4 + 255) & ~(int64_t)255) + std::max<int64_t>(256 + tiles, tiles * 8) * 256 * 256 * 4);
        w.wg_ws = b.take_bytes(w.wg_ws_bytes);
        w.feats16 = b.take_bytes((size_t)Me * c.feat * 2);
    }
    w.bytes = (b.off + 255) & ~(size_t)255;
}

// ------------------------------------------------------------------------------------------------ op helpers
#define TRY(x) do { int e__ = (x); if (e__) return e__; } while (0)

// Side stream for the weight-gradient GEMMs of the backward.  dW = dY^T X and dX = dY W of a projection both only READ dY,
// nothing reads dW before the optimizer, and each of them alone leaves a partly filled last round of workgroups (680
// tiles on 512 slots, 384 split-K workgroups, ...): the wgrad runs on this stream beside the dgrad on the caller's stream
// (fork: the side stream waits for the event "dY produced"; lazy join: the caller's stream waits for a wgrad only right
// before a kernel that OVERWRITES the buffer this wgrad reads, so the side stream may lag by a few kernels).  Mixed
// precision only (there every dY is one of a few bf16 temporaries); ortk_tuning.side_stream = 0 keeps everything on the
// caller's stream.  Free-running microbenchmark (scratch/gemm_concurrent.py): dgrad + wgrad of w1 172 -> 128 us, of
// qkv 124 -> 87 us, of the 512 x 512 projections no change.
struct SideStream {
    hipStream_t s = nullptr;
    hipStream_t s2 = nullptr;      // a third queue for short independent launches at the head of a forward (the chain weights' packing)
    hipEvent_t ev[64];
    int next = 0;
    bool ok = false;
    bool init() {
        // (a side stream at the lowest priority the device offers measured the same step: 10.51 ms both ways, round 6)
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return false;
        for (auto& x : ev) if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) return false;
        ok = true;
        return true;
    }
    hipEvent_t take() { hipEvent_t e = ev[next]; next = (next + 1) & 63; return e; }
};
// One side stream (and event ring) per (device, caller stream): two models stepping from two host threads on two streams
// never share one — the C-ABI's "thread-safe per stream" holds for the executor too.  Created on first use, kept for the
// life of the process (a handful of entries: one per stream a host ever trains on).
static SideStream* side_for(hipStream_t caller) {
    if (!tuning().side_stream) return nullptr;
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, hipStream_t>, SideStream*>> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : table) if (e.first.first == dev && e.first.second == caller) return e.second->ok ? e.second : nullptr;
    SideStream* ss = new SideStream();
    ss->init();
    table.push_back({{dev, caller}, ss});
    return ss->ok ? ss : nullptr;
}

struct Ctx {
    const ortk_config* cfg; hipStream_t s; int prec; uint64_t seed; bool train;
    const float* P;          // fp32 parameter arena (biases, LayerNorm, embeddings, and weights in fp32 mode)
    const void* W16;         // bf16 weight arena (mixed precision) or nullptr
    int adt;                 // dtype of (A) buffers
    const ortk_sparse_plan* ell_f = nullptr;        // optional sparse plan over (N,K) weight blocks: forward-layout products
    const ortk_sparse_plan* ell_b = nullptr;        // ... over their transposed copies: data-gradient products (mixed precision)
    const void* W16T = nullptr;                  // transposed bf16 weight blocks (training workspaces, mixed precision)
    const int32_t* drop_rows = nullptr;          // valid-position decoder rows: row m draws its dropout as row drop_rows[m] of the padded
                                                 // (caption, position) layout (ortk_batch.row_pos; set around the DECODER's operators only)
    int drop_rs = 0, drop_r0 = 0;                // decode step in train mode: output row m draws like row m * drop_rs + drop_r0 of the
                                                 // teacher-forced pass (ortk_gemm_args.drop_row_stride / _off); 0: the natural index
    bool use_side = false;                       // weight-gradient GEMMs (and other independent work) on `side`
    SideStream* side = nullptr;                  // the side stream of this call's (device, caller stream)
    struct Pend { const void* buf; hipEvent_t done; uint64_t seq; };
    mutable Pend pend[32] = {};                  // buffers a forked, not yet joined wgrad reads
    // The side stream runs its launches in order: once the caller's stream has waited for launch number k, every buffer read by a
    // launch <= k is free.  before_write() therefore inserts ONE wait for the newest launch it needs and none for older ones — each
    // wait is a barrier packet on the caller's queue, ~6 us of idle time even when the event has long fired (12 layers x ~8 of them).
    mutable uint64_t side_seq = 0, waited_seq = 0;
    // weight gradients collected for ONE grouped launch (ortk_wgrad_group): wgrad_gemm appends while `grp_on`, flush_wgrads launches
    mutable ortk_wgrad_group_args grp = {};
    bool grp_on = false;
    void* wg_ws = nullptr; size_t wg_ws_bytes = 0;
    mutable hipEvent_t last_done = nullptr;
    uint64_t next_seq() const { return ++side_seq; }     // one per launch (or group of launches behind one event) queued on the side stream
    void reads(const void* buf, hipEvent_t done, uint64_t seq = 0) const {
        if (!seq) seq = next_seq();
        for (auto& p : pend) if (p.buf == buf || p.buf == nullptr) { p.buf = buf; p.done = done; p.seq = seq; last_done = done; return; }
        pend[0] = Pend{buf, done, seq}; last_done = done;     // table full (never with this path's temporaries): join() covers what fell out
    }
    int before_write(const void* buf) const {            // the caller's stream is about to overwrite `buf`
        for (auto& p : pend)
            if (p.buf == buf) {
                if (p.seq > waited_seq) {
                    if (hipStreamWaitEvent(s, p.done, 0) != hipSuccess) return ORTK_EINVAL;
                    waited_seq = p.seq;
                }
                p.buf = nullptr;
            }
        return 0;
    }
    // generic fork / mark / wait for other independent work (forward: geometry bias, cross-attention K/V projection;
    // backward: token-embedding and geometry-bias gradients)
    int fork() const {                                   // the side stream waits for everything queued on `s` so far
        hipEvent_t e = side->take();
        if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(side->s, e, 0) != hipSuccess) return ORTK_EINVAL;
        return 0;
    }
    int side_mark(hipEvent_t* out) const {               // completion point of the work queued on the side stream so far
        hipEvent_t e = side->take();
        if (hipEventRecord(e, side->s) != hipSuccess) return ORTK_EINVAL;
        last_done = e;
        if (out) *out = e;
        return 0;
    }
    int wait_ev(hipEvent_t e) const { return (e && hipStreamWaitEvent(s, e, 0) != hipSuccess) ? ORTK_EINVAL : 0; }
    Ctx on_side() const { Ctx t = *this; t.s = side->s; t.use_side = false; return t; }
    int join() const {                                   // everything forked so far (the side stream runs in order)
        if (last_done) { if (hipStreamWaitEvent(s, last_done, 0) != hipSuccess) return ORTK_EINVAL; last_done = nullptr; }
        waited_seq = side_seq;
        for (auto& p : pend) p.buf = nullptr;
        return 0;
    }
    // block of `plan` that is the (N outputs, K inputs) matrix at arena offset `off`, or -1
    // (with share_att "qk" a (d,d) projection starts where the K|V block does: the shape is part of the key)
    static int ell_block(const ortk_sparse_plan* plan, int64_t off, int N, int K) {
        if (!plan) return -1;
        for (int i = 0; i < plan->nblocks; ++i) {
            const ortk_sparse_block& b = plan->blocks_host[i];
            if (b.src_offset == off && b.N == N && b.K == K) return i;
        }
        return -1;
    }
    float p_drop() const { return train ? cfg->drop : 0.f; }
    float p_src() const { return train ? cfg->drop_src : 0.f; }
    uint32_t sub(uint32_t op) const { return ortk_subseed(seed, op); }
    const void* W(int64_t off) const {
        return prec ? (const void*)(reinterpret_cast<const __bf16*>(W16) + off) : (const void*)(P + off);
    }
    int wdt() const { return prec ? ORTK_BF16 : ORTK_F32; }
};
static inline void* off_elems(void* p, int64_t n, int dt) { return reinterpret_cast<char*>(p) + (size_t)n * ortk_esize(dt); }
static inline const void* off_elems(const void* p, int64_t n, int dt) { return reinterpret_cast<const char*>(p) + (size_t)n * ortk_esize(dt); }

// Y = epi(X W^T): forward projection (W = weight at arena offset woff, (N,K) row-major)
static int fwd_gemm(const Ctx& c, const void* X, int xdt, int64_t ldx, int64_t woff, const float* bias, void* Y, int ydt, int64_t ldy,
                    int64_t M, int N, int K, bool relu = false, float drop = 0.f, uint32_t seed = 0, const float* resid = nullptr,
                    int64_t ldr = 0, const float* rowscale = nullptr) {
    {
        const int blk = Ctx::ell_block(c.ell_f, woff, N, K);
        if (blk >= 0) {
            ortk_spmm_args sa; std::memset(&sa, 0, sizeof(sa));
            sa.X = X; sa.x_dtype = xdt; sa.ldx = ldx; sa.Y = Y; sa.y_dtype = ydt; sa.ldy = ldy; sa.M = M;
            sa.bias = bias; sa.rowscale = rowscale; sa.resid = resid; sa.ldr = ldr; sa.relu = relu; sa.drop_p = drop; sa.drop_seed = seed;
            sa.drop_rows = c.drop_rows;
            return ortk_spmm(c.ell_f, blk, &sa, (ortk_stream)c.s);
        }
    }
    ortk_gemm_args a; std::memset(&a, 0, sizeof(a));
    a.A = X; a.a_dtype = xdt; a.lda = ldx; a.B = c.W(woff); a.b_dtype = c.wdt(); a.ldb = K; a.C = Y; a.c_dtype = ydt; a.ldc = ldy;
    a.M = (int)M; a.N = N; a.K = K;
    a.bias = bias; a.relu = relu; a.drop_p = drop; a.drop_seed = seed; a.resid = resid; a.ldr = ldr; a.rowscale = rowscale;
    a.drop_row_stride = c.drop_rs; a.drop_row_off = c.drop_r0; a.drop_rows = c.drop_rows;
    a.precision = c.prec;
    return ortk_gemm(&a, (ortk_stream)c.s);
}
// dX = dY W   (W stored (N_out, K_in)); optional ReLU/dropout gate
// With the transposed bf16 copy of the block (W^T stored (K_in, N_out)) this is the FORWARD operand layout — both operands
// k-contiguous, LDS-DMA kernels: 3.34 -> 2.78 ms per step over the path's shapes in isolation (scratch/dgrad_layouts.py).
// `whole_block` = false: W is a sub-block of a packed projection (no transposed image of its own).
static int dgrad_gemm(const Ctx& c, const void* dY, int dydt, int64_t lddy, int64_t woff, void* dX, int dxdt, int64_t lddx, int64_t M,
                      int Nout, int Kin, const void* gate = nullptr, int gdt = 0, int64_t ldg = 0, float gate_scale = 1.f,
                      bool whole_block = true) {
    if (c.prec && c.W16T && whole_block) {
        // W~^T as a sparse plan block: (Kin outputs, Nout inputs) at the same offset
        constexpr int KMAX = 2048;          // inputs one sparse product launch takes (ortk_sparse.hip)
        const int blk = Ctx::ell_block(c.ell_b, woff, Kin, Nout);      // (ELL plans hold wide blocks as pieces: below)
        if (blk >= 0) {
            ortk_spmm_args sa; std::memset(&sa, 0, sizeof(sa));
            sa.X = dY; sa.x_dtype = dydt; sa.ldx = lddy; sa.Y = dX; sa.y_dtype = dxdt; sa.ldy = lddx; sa.M = M;
            sa.gate = gate; sa.gate_dtype = gdt; sa.ldg = ldg; sa.gate_scale = gate_scale;
            TRY(c.before_write(dX));
            return ortk_spmm(c.ell_b, blk, &sa, (ortk_stream)c.s);
        }
        if (Nout > KMAX && !gate && dxdt == ORTK_F32 && c.ell_b) {
            // more inputs than one launch takes (the generator: 10 112 logit columns): the plan holds the block cut into
            // KMAX-wide pieces along its inputs; the pieces accumulate into the fp32 output through the residual operand
            bool all = true;
            for (int k0 = 0; k0 < Nout; k0 += KMAX) all = all && Ctx::ell_block(c.ell_b, woff + k0, Kin, std::min(KMAX, Nout - k0)) >= 0;
            if (all) {
                TRY(c.before_write(dX));
                for (int k0 = 0; k0 < Nout; k0 += KMAX) {
                    ortk_spmm_args sa; std::memset(&sa, 0, sizeof(sa));
                    sa.X = off_elems(dY, k0, dydt); sa.x_dtype = dydt; sa.ldx = lddy; sa.Y = dX; sa.y_dtype = ORTK_F32; sa.ldy = lddx; sa.M = M;
                    if (k0) { sa.resid = reinterpret_cast<const float*>(dX); sa.ldr = lddx; }
                    TRY(ortk_spmm(c.ell_b, Ctx::ell_block(c.ell_b, woff + k0, Kin, std::min(KMAX, Nout - k0)), &sa, (ortk_stream)c.s));
                }
                return 0;
            }
        }
    }
    ortk_gemm_args a; std::memset(&a, 0, sizeof(a));
    a.A = dY; a.a_dtype = dydt; a.lda = lddy;
    if (c.prec && c.W16T && whole_block) {
        a.B = reinterpret_cast<const __bf16*>(c.W16T) + woff; a.b_dtype = ORTK_BF16; a.ldb = Nout; a.transB = 0;
    } else {
        a.B = c.W(woff); a.b_dtype = c.wdt(); a.ldb = Kin; a.transB = 1;
    }
    a.C = dX; a.c_dtype = dxdt; a.ldc = lddx; a.M = (int)M; a.N = Kin; a.K = Nout;
    a.gate = gate; a.gate_dtype = gdt; a.ldg = ldg; a.gate_scale = gate_scale; a.precision = c.prec;
    TRY(c.before_write(dX));
    return ortk_gemm(&a, (ortk_stream)c.s);
}
// the collected weight gradients of a layer in ONE launch (ortk_wgrad.hip), on the side stream when there is one: the launch waits for
// everything queued on the caller's stream so far (the last dY), and every dY it reads is marked pending until it is done
// Row ranges: a launch that fills the chip keeps the caller's stream waiting (its workgroups hold a compute unit's LDS each: 11.4 ms per XE
// step against 11.3 ungrouped), so a group gets about ortk_tuning.wgrad_group_wgs workgroups — with the ~130-workgroup data-gradient
// launches of the caller's stream the two queues then share the 256 units side by side (10.7 ms).  `urgent`: the caller's stream is about
// to wait for this launch (the last group before a join): one full round of workgroups.
static int flush_wgrads(const Ctx& c, bool urgent = false) {
    if (c.grp.n == 0) return 0;
    int64_t tiles = 0;
    for (int i = 0; i < c.grp.n; ++i) tiles += ortk_cdiv(c.grp.item[i].Nout, 256) * ortk_cdiv(c.grp.item[i].Kin, 256);
    const int forced = tuning().wgrad_group_splitk, target = tuning().wgrad_group_wgs;
    // (urgent: one round of workgroups, but never fewer than 64 stages of 32 rows per workgroup — each row range adds its whole tile to the
    //  arena: nine ranges of 32 stages made the step's last launch 137 us, 50 of them atomics)
    const int64_t full = std::max<int64_t>(1, std::min<int64_t>(256 / std::max<int64_t>(1, tiles), c.grp.rows / (64 * 32)));
    c.grp.splitk = forced > 0 ? forced : (urgent && tuning().wgrad_group_tail) ? (int)full : (int)std::max<int64_t>(1, (target + tiles / 2) / tiles);
    c.grp.flags = 0;
    c.grp.ws = nullptr; c.grp.ws_bytes = 0;
    if (tuning().wgrad_group & 8) {
        const size_t need = ortk_wgrad_group_workspace_bytes(&c.grp);
        if (need && need <= c.wg_ws_bytes) { c.grp.ws = c.wg_ws; c.grp.ws_bytes = c.wg_ws_bytes; }
    }
    int e = 0;
    if (c.use_side) {
        hipEvent_t ready = c.side->take(), done = c.side->take();
        if (hipEventRecord(ready, c.s) != hipSuccess || hipStreamWaitEvent(c.side->s, ready, 0) != hipSuccess) return ORTK_EINVAL;
        e = ortk_wgrad_group(&c.grp, (ortk_stream)c.side->s);
        if (!e && hipEventRecord(done, c.side->s) != hipSuccess) e = ORTK_EINVAL;
        if (!e) { const uint64_t seq = c.next_seq(); for (int i = 0; i < c.grp.n; ++i) c.reads(c.grp.item[i].dY, done, seq); }
    } else {
        e = ortk_wgrad_group(&c.grp, (ortk_stream)c.s);
    }
    c.grp.n = 0;
    return e;
}
// dW += dY^T X ; db += colsum(dY)
static int wgrad_gemm(const Ctx& c, const void* dY, int dydt, int64_t lddy, const void* X, int xdt, int64_t ldx, float* dW, float* db,
                      int64_t M, int Nout, int Kin) {
    if (c.grp_on && dydt == ORTK_BF16 && xdt == ORTK_BF16 && !(Nout & 7) && !(Kin & 7) && Nout >= 8 && Kin >= 8 && !(lddy & 7) && !(ldx & 7) &&
        !(reinterpret_cast<uintptr_t>(dY) & 15) && !(reinterpret_cast<uintptr_t>(X) & 15) && (c.grp.n == 0 || c.grp.rows == M)) {
        if (c.grp.n == ORTK_WGRAD_MAX) TRY(flush_wgrads(c));
        ortk_wgrad_item& it = c.grp.item[c.grp.n++];
        it.dY = dY; it.lddy = lddy; it.X = X; it.ldx = ldx; it.dW = dW; it.lddw = Kin; it.db = db; it.Nout = Nout; it.Kin = Kin;
        c.grp.rows = M;
        return 0;
    }
    TRY(flush_wgrads(c));      // (keeps the order of the additions into a shared weight block)
    ortk_gemm_args a; std::memset(&a, 0, sizeof(a));
    a.A = dY; a.a_dtype = dydt; a.lda = lddy; a.transA = 1; a.B = X; a.b_dtype = xdt; a.ldb = ldx; a.transB = 1; a.C = dW; a.ldc = Kin;
    a.M = Nout; a.N = Kin; a.K = (int)M; a.accumulate = 1; a.precision = c.prec;
    // K (= rows of the batch) is split over workgroups that accumulate with 256-B-contiguous atomics.  Tuned on the
    // path's shapes (scratch/wgrad_bench.py): ~384 workgroups (1.5 per CU) for the small outputs, 3 splits for the
    // 10112x512 generator (632 tiles alone leave a 23 % tail), never fewer than 8 K-steps per workgroup.
    const int64_t tiles = ortk_cdiv(Nout, 128) * ortk_cdiv(Kin, 128);
    const int64_t wgs = ortk::tuning().wgrad_wgs;
    int64_t sk = tiles >= 256 ? (M >= 16384 ? 3 : 1) : (wgs + tiles / 2) / tiles;
    const int64_t max_sk = std::max<int64_t>(1, M / 512);
    a.splitk = (int)std::max<int64_t>(1, std::min(sk, max_sk));
    a.colsum = db;                         // bias gradient fused into the wgrad kernel (both precisions; ortk_gemm falls back to ortk_colsum)
    if (c.use_side) {
        hipEvent_t ready = c.side->take(), done = c.side->take();
        if (hipEventRecord(ready, c.s) != hipSuccess || hipStreamWaitEvent(c.side->s, ready, 0) != hipSuccess) return ORTK_EINVAL;
        TRY(ortk_gemm(&a, (ortk_stream)c.side->s));
        if (hipEventRecord(done, c.side->s) != hipSuccess) return ORTK_EINVAL;
        c.reads(dY, done);
        return 0;
    }
    TRY(ortk_gemm(&a, (ortk_stream)c.s));
    return 0;
}
static int ln_fwd(const Ctx& c, const float* x, int64_t a, int64_t b, void* y, int ydt, float* st, int64_t rows) {
    return ortk_layernorm_fwd(x, c.P + a, c.P + b, y, ydt, st, rows, c.cfg->d_model, 1e-6f, (ortk_stream)c.s);
}
// next_op >= 0: also emit, into `dz`, the dropout-masked / bf16 copy of dx that the NEXT drop_bwd(dx, dz, ., next_op) would
// produce (that call then finds its work done, see drop_bwd)
static int ln_bwd(const Ctx& c, const void* dy, const float* x, float* G, int64_t a, int64_t b, const float* st,
                  const float* dres, float* dx, int64_t rows, void* dz = nullptr, int next_op = -1, int dy_dt = ORTK_F32) {
    const bool fuse = dz && next_op >= 0 && (c.p_drop() > 0.f || c.adt == ORTK_BF16);
    if (fuse) TRY(c.before_write(dz));
    return ortk_layernorm_bwd_dt(dy, dy_dt, x, c.P + a, st, dres, dx, G + a, G + b, rows, c.cfg->d_model, 1e-6f, fuse ? dz : nullptr,
                                 c.adt, c.p_drop(), fuse ? c.sub((uint32_t)next_op) : 0, c.drop_rows, (ortk_stream)c.s);
}
// gradient through a residual-branch dropout: the buffer (and its dtype) holding dx * keep/(1-p).
// Mixed precision always goes through `tmp` (it also performs the fp32 -> bf16 conversion of the GEMM operand).
static int drop_bwd(const Ctx& c, const float* dx, void* tmp, int64_t n, uint32_t op, const void** out, int* out_dt,
                    bool done_by_ln_bwd = false) {
    if (c.p_drop() > 0.f || c.adt == ORTK_BF16) {
        const int d_ = c.cfg->d_model;
        if (!done_by_ln_bwd) TRY(ortk_dropout_apply_rows(dx, tmp, c.adt, n / d_, d_, c.p_drop(), c.sub(op), c.drop_rows, (ortk_stream)c.s));
        *out = tmp; *out_dt = c.adt;
    } else { *out = dx; *out_dt = ORTK_F32; }
    return 0;
}

// dX = dLN/dx(dY W) + dres in ONE launch (ortk_gemm ln_mode 2: short row panels, LayerNorm backward in the epilogue), or — sparse plans,
// fp32 mode, widths other than 512, a sub-block of a packed projection — the data-gradient GEMM into `gy` followed by ln_bwd.
// (W stored (N_out, K_in) with K_in = d_model; the LayerNorm is the one whose OUTPUT the projection read.)
static int dgrad_ln_bwd(const Ctx& c, const void* dY, int dydt, int64_t lddy, int64_t woff, int64_t M, int Nout, int Kin, float* gy,
                        const float* x, float* G, int64_t na, int64_t nb, const float* st, const float* dres, float* dx, void* dz, int next_op,
                        bool whole_block = true) {
    const bool fusable = tuning().ln_fuse & 1 && c.prec && c.W16T && whole_block && dydt == ORTK_BF16 && Kin == 512 && Kin == c.cfg->d_model &&
                         (Nout % 64) == 0 && Ctx::ell_block(c.ell_b, woff, Kin, Nout) < 0 && !c.ell_b;
    if (!fusable) {
        // Mixed precision on the dense kernels: the LayerNorm's output gradient lives as bf16 between the two launches, like every other
        // gradient that is a GEMM operand there (half the bytes out of the GEMM's epilogue and into ln_bwd; tuning().ln_fuse & 8 keeps fp32)
        const int gdt = (c.prec && c.adt == ORTK_BF16 && c.W16T && whole_block && !c.ell_b && Kin == 512 && Kin == c.cfg->d_model && !(tuning().ln_fuse & 8)) ? ORTK_BF16 : ORTK_F32;
        TRY(dgrad_gemm(c, dY, dydt, lddy, woff, gy, gdt, Kin, M, Nout, Kin, nullptr, 0, 0, 1.f, whole_block));
        return ln_bwd(c, gy, x, G, na, nb, st, dres, dx, M, dz, next_op, gdt);
    }
    const bool mask = dz && next_op >= 0;        // (mixed precision: the masked copy is also the bf16 conversion)
    ortk_gemm_args a; std::memset(&a, 0, sizeof(a));
    a.A = dY; a.a_dtype = ORTK_BF16; a.lda = lddy;
    a.B = reinterpret_cast<const __bf16*>(c.W16T) + woff; a.b_dtype = ORTK_BF16; a.ldb = Nout;
    a.C = dx; a.c_dtype = ORTK_F32; a.ldc = Kin; a.M = (int)M; a.N = Kin; a.K = Nout; a.precision = 1;
    a.ln_mode = 2; a.ln_a = c.P + na; a.ln_stats = const_cast<float*>(st); a.ln_eps = 1e-6f; a.ln_x = x; a.ln_dres = dres;
    a.ln_da = G + na; a.ln_db = G + nb;
    a.ln_y = mask ? dz : nullptr; a.ln_y_dtype = c.adt;
    a.drop_p = c.p_drop(); a.drop_seed = mask ? c.sub((uint32_t)next_op) : 0; a.drop_rows = c.drop_rows;
    TRY(c.before_write(dx));
    if (mask) TRY(c.before_write(dz));
    return ortk_gemm(&a, (ortk_stream)c.s);
}

static const float DIM_MAT_SENTINEL = -1.f;
static float g_dim_mat[8] = {DIM_MAT_SENTINEL, 0, 0, 0, 0, 0, 0, 0};
static const float* dim_mat() {
    if (g_dim_mat[0] == DIM_MAT_SENTINEL) {
        // fp32 evaluation of 1 / 1000^(k/8) in the reference's op order (relation_transformer.py:241-243):
        // feat_range / 8 (exact), powf in fp32, reciprocal in fp32.  tests/test_lib_host.py pins these 8 values
        // against torch's.
        float t[8];
        for (int k = 0; k < 8; ++k) t[k] = 1.0f / powf(1000.0f, (float)k / 8.0f);
        for (int k = 7; k >= 0; --k) g_dim_mat[k] = t[k];
    }
    return g_dim_mat;
}

enum Op : uint32_t { OP_SRC = 1, OP_EMB = 2, OP_ENC = 16, OP_DEC = 16 + 16 * MAXLAYERS };
static inline uint32_t eop(int l, int k) { return OP_ENC + 16 * l + k; }
static inline uint32_t dop(int l, int k) { return OP_DEC + 16 * l + k; }
}  // namespace ortk
extern "C" uint32_t ortk_dropout_site_seed(uint64_t seed, int32_t stack, int32_t layer, int32_t k) {
    using namespace ortk;
    const uint32_t op = stack == 0 ? (uint32_t)OP_SRC : stack == 1 ? (uint32_t)OP_EMB : stack == 2 ? eop(layer, k) : dop(layer, k);
    return ortk_subseed(seed, op);
}
namespace ortk {

struct EncPtrs { void* y1; void* qkv; float* P; void* o; float* xm; void* y2; void* h; float* xout; float *st1, *st2; };

// encoder stack; `bufs[l]` may alias between layers when nothing has to be kept for a backward pass.
// `mem` receives the final LayerNorm in dtype `mem_dt`.
// the geometry bias of all encoder layers (depends on the boxes and the fp32 WG weights only), on the side stream when there is one
static int queue_box_bias(const Ctx& c, const Offsets& o, const float* boxes, float* logbias, int B, int S, hipEvent_t* done) {
    const ortk_config& cfg = *c.cfg;
    const int L = cfg.n_layers, H = cfg.n_heads;
    const float* wg[MAXLAYERS]; const float* bg[MAXLAYERS];
    for (int l = 0; l < L; ++l) { wg[l] = c.P + o.enc[l].wg; bg[l] = c.P + o.enc[l].bg; }
    *done = nullptr;
    if (c.use_side) {
        TRY(c.fork());
        TRY(ortk_box_logbias_fwd(boxes, wg, bg, cfg.box_trig ? dim_mat() : nullptr, logbias, L, B, S, H, (ortk_stream)c.side->s));
        return c.side_mark(done);
    }
    return ortk_box_logbias_fwd(boxes, wg, bg, cfg.box_trig ? dim_mat() : nullptr, logbias, L, B, S, H, (ortk_stream)c.s);
}

// `box_queued`: the caller has already queued the geometry bias (queue_box_bias; *box_queued = its completion event or NULL)
static int encoder_forward(const Ctx& c, const Offsets& o, const float* feats, const float* boxes, const float* masks, int B, int S,
                           float* x0, float* logbias, const EncPtrs* bufs, void* mem, int mem_dt, float* st_mem, int qdt = ORTK_F32,
                           const hipEvent_t* box_queued = nullptr, const ChainSet* cs = nullptr, const void* chain_pk = nullptr,
                           bool keep = true,         // keep = false (decode): the chains do not store what only a backward would read
                           const void* feats16 = nullptr, hipEvent_t feats16_done = nullptr,     // bf16 copy of the features, made on another queue
                           hipEvent_t pack_done = nullptr) {                                        // the chains' packed weights, likewise
    const ortk_config& cfg = *c.cfg;
    const float* P = c.P;
    const int d = cfg.d_model, ff = cfg.d_ff, H = cfg.n_heads, L = cfg.n_layers, dk = d / H, A = c.adt;
    const int64_t Me = (int64_t)B * S;
    // att_embed: relu(Linear) on valid regions, zeros elsewhere, dropout (relation_transformer.py:331-333,349-350)
    // (the plain `transformer` embeds every row, padded regions included: transformer.py:627-629)
    const bool plain = cfg.no_box != 0;
    // the geometry bias only depends on the boxes and WG: queued first, beside att_embed / the first LayerNorm and QKV projection
    hipEvent_t box_done = nullptr;
    if (plain) {
        // no geometry bias
    } else if (box_queued) {
        box_done = *box_queued;
    } else {
        TRY(queue_box_bias(c, o, boxes, logbias, B, S, &box_done));
    }
    if (feats16) {
        TRY(c.wait_ev(feats16_done));
        TRY(fwd_gemm(c, feats16, ORTK_BF16, cfg.feat, o.att_w, P + o.att_b, x0, ORTK_F32, d, Me, d, cfg.feat, true, c.p_src(), c.sub(OP_SRC),
                     nullptr, 0, plain ? nullptr : masks));
    } else {
        TRY(fwd_gemm(c, feats, ORTK_F32, cfg.feat, o.att_w, P + o.att_b, x0, ORTK_F32, d, Me, d, cfg.feat, true, c.p_src(), c.sub(OP_SRC),
                     nullptr, 0, plain ? nullptr : masks));
    }
    TRY(c.wait_ev(pack_done));
    const float* x = x0;
    const AttMode am = att_mode(cfg.share_att_enc);
    // rows-stationary chains (ortk_chain.hip) in place of the LayerNorm / projection launches: bf16 Q|K|V and memory, dense products
    const bool chains = cs && cs->on && cs->e0 >= 0 && chain_pk && qdt == ORTK_BF16 && mem_dt == ORTK_BF16 && A == ORTK_BF16;
    if (chains) {
        ortk_chain_args ca; std::memset(&ca, 0, sizeof(ca));
        ca.n_units = cs->units(cs->e0); ca.M = Me; ca.x_in = x0;
        ca.g1 = P + o.enc[0].n0a; ca.b1 = P + o.enc[0].n0b; ca.y1 = keep ? bufs[0].y1 : nullptr; ca.st1 = keep ? bufs[0].st1 : nullptr;
        ca.n1 = 3; ca.bias_s1 = P + o.enc[0].bqkv; ca.out1 = bufs[0].qkv; ca.ld1 = 3 * d;
        ca.eps = 1e-6f;
        TRY(chain_run(&ca, cs->stream_of(chain_pk, cs->e0), c.s));
    }
    for (int l = 0; l < L; ++l) {
        const EncOff& e = o.enc[l]; const EncPtrs& b = bufs[l];
        if (chains) {
            ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = c.prec; a.qkv_dtype = qdt;
            a.q = (const float*)b.qkv; a.k = (const float*)off_elems(b.qkv, am.k * d, qdt); a.v = (const float*)off_elems(b.qkv, am.v * d, qdt);
            a.ldq = a.ldk = a.ldv = 3 * d; a.o = b.o; a.o_dtype = A; a.ldo = d;
            a.kmask = masks; a.bias = plain ? nullptr : logbias + (int64_t)l * B * H * S * S; a.p = b.P;
            a.nkv = B; a.H = H; a.Lq = S; a.Lk = S; a.dk = dk; a.drop_p = c.p_drop(); a.drop_seed = c.sub(eop(l, 0));
            if (l == 0) TRY(c.wait_ev(box_done));
            TRY(ortk_attention_fwd(&a, (ortk_stream)c.s));
            ortk_chain_args ca; std::memset(&ca, 0, sizeof(ca));
            ca.n_units = cs->units(cs->eb[l]); ca.M = Me; ca.x_in = x;
            ca.a_in = b.o; ca.bias_r = P + e.bo; ca.x_mid = b.xm; ca.seed_r = c.sub(eop(l, 1));
            ca.g1 = P + e.n1a; ca.b1 = P + e.n1b; ca.y1 = keep ? b.y2 : nullptr; ca.st1 = keep ? b.st2 : nullptr;
            ca.NC = ff / 512; ca.bias_h = P + e.b1; ca.bias_o = P + e.b2; ca.h = keep ? b.h : nullptr; ca.x_out = b.xout;
            ca.seed_h = c.sub(eop(l, 2)); ca.seed_o = c.sub(eop(l, 3));
            if (l + 1 < L) {
                ca.g2 = P + o.enc[l + 1].n0a; ca.b2 = P + o.enc[l + 1].n0b; ca.y2 = keep ? bufs[l + 1].y1 : nullptr; ca.st2 = keep ? bufs[l + 1].st1 : nullptr;
                ca.n2 = 3; ca.bias_s2 = P + o.enc[l + 1].bqkv; ca.out2 = bufs[l + 1].qkv; ca.ld2 = 3 * d;
            } else {
                ca.g2 = P + o.enc_na; ca.b2 = P + o.enc_nb; ca.y2 = mem; ca.st2 = keep ? st_mem : nullptr;
            }
            ca.drop_p = c.p_drop(); ca.eps = 1e-6f;
            TRY(chain_run(&ca, cs->stream_of(chain_pk, cs->eb[l]), c.s));
            x = b.xout;
            continue;
        }
        TRY(ln_fwd(c, x, e.n0a, e.n0b, b.y1, A, b.st1, Me));
        TRY(fwd_gemm(c, b.y1, A, d, e.wqkv, P + e.bqkv, b.qkv, qdt, 3 * d, Me, am.n * d, d));
        ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = c.prec; a.qkv_dtype = qdt;
        a.q = (const float*)b.qkv; a.k = (const float*)off_elems(b.qkv, am.k * d, qdt); a.v = (const float*)off_elems(b.qkv, am.v * d, qdt);
        a.ldq = a.ldk = a.ldv = 3 * d; a.o = b.o; a.o_dtype = A; a.ldo = d;
        a.kmask = masks; a.bias = plain ? nullptr : logbias + (int64_t)l * B * H * S * S; a.p = b.P;
        a.nkv = B; a.H = H; a.Lq = S; a.Lk = S; a.dk = dk; a.drop_p = c.p_drop(); a.drop_seed = c.sub(eop(l, 0));
        if (l == 0) TRY(c.wait_ev(box_done));
        TRY(ortk_attention_fwd(&a, (ortk_stream)c.s));
        TRY(fwd_gemm(c, b.o, A, d, e.wo, P + e.bo, b.xm, ORTK_F32, d, Me, d, d, false, c.p_drop(), c.sub(eop(l, 1)), x, d));
        TRY(ln_fwd(c, b.xm, e.n1a, e.n1b, b.y2, A, b.st2, Me));
        TRY(fwd_gemm(c, b.y2, A, d, e.w1, P + e.b1, b.h, A, ff, Me, ff, d, true, c.p_drop(), c.sub(eop(l, 2))));
        TRY(fwd_gemm(c, b.h, A, ff, e.w2, P + e.b2, b.xout, ORTK_F32, d, Me, d, ff, false, c.p_drop(), c.sub(eop(l, 3)), b.xm, d));
        x = b.xout;
    }
    if (!chains) TRY(ln_fwd(c, x, o.enc_na, o.enc_nb, mem, mem_dt, st_mem, Me));
    return 0;
}

static void enc_ptrs_from_ws(const TrainWS& w, int L, EncPtrs* out) {
    for (int l = 0; l < L; ++l) {
        const EncBuf& e = w.enc[l];
        out[l] = EncPtrs{e.y1, e.qkv, e.P, e.o, e.xm, e.y2, e.h, e.xout, e.st1, e.st2};
    }
}

// bf16 working copy of the trainable arena (weights are the B operand of every forward / dgrad GEMM)
static int make_w16t(const ortk_config* cfg, const Offsets& o, const float* params, void* w16t, ortk_stream stream);
static int make_w16(const ortk_config* cfg, const Offsets& o, const float* params, void* w16, ortk_stream stream) {
    if (!cfg->precision) return 0;
    return ortk_cast_bf16(params, w16, o.total, stream);
}

}  // namespace ortk

using namespace ortk;

// ================================================================================================ C ABI
extern "C" int ortk_version(void) { return ORTK_VERSION; }

// ------------------------------------------------------------------------------------------------ tuning switches
namespace ortk {
static ortk_tuning g_tuning = {0, 640, 0, 33, 1, 1, 1, 1, 1, 384, 3, 0, 80, 1, 1, 0, 1, 0};
const ortk_tuning& tuning() { return g_tuning; }
}
extern "C" void ortk_get_tuning(ortk_tuning* out) { if (out) *out = ortk::g_tuning; }
extern "C" int ortk_set_tuning(const ortk_tuning* t) {
    if (!t || t->gemm_impl < 0 || t->gemm_impl > 3 || t->attn_impl < 0 || t->attn_impl > 4 || t->attn16_min_lq < 1 || t->f32_split < 0 || t->f32_split > 7 || t->wgrad_wgs < 1 ||
        t->wgrad_group < 0 || t->wgrad_group > 15 || t->wgrad_group_splitk < 0 || t->wgrad_group_splitk > 8 || t->wgrad_group_wgs < 1 || t->wgrad_group_tail < 0 || t->wgrad_group_tail > 1 ||
        t->feats_bf16 < 0 || t->feats_bf16 > 1 || t->ln_fuse < 0 || t->ln_fuse > 15 || t->samp_epilogue < 0 || t->samp_epilogue > 1 || t->gemm_epilogue < 0 || t->gemm_epilogue > 3) return ORTK_EINVAL;
    ortk::g_tuning = *t;
    return 0;
}

extern "C" int ortk_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return std::strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
}

extern "C" int64_t ortk_arena_numel(const ortk_config* cfg) {
    if (check_cfg(cfg)) return -1;
    Offsets o; build_layout(*cfg, o, nullptr);
    return o.total;
}
extern "C" int64_t ortk_arena_numel_with_buffers(const ortk_config* cfg) {
    if (check_cfg(cfg)) return -1;
    Offsets o; build_layout(*cfg, o, nullptr);
    return o.total_all;
}
extern "C" int32_t ortk_arena_entries(const ortk_config* cfg) {
    if (check_cfg(cfg)) return -1;
    Offsets o; std::vector<Entry> v; build_layout(*cfg, o, &v);
    return (int32_t)v.size();
}
extern "C" int ortk_arena_entry(const ortk_config* cfg, int32_t index, char* name_buf, int64_t* offset, int64_t* numel,
                                int32_t* ndim, int64_t* shape, int32_t* kind) {
    if (int e = check_cfg(cfg)) return e;
    Offsets o; std::vector<Entry> v; build_layout(*cfg, o, &v);
    if (index < 0 || index >= (int32_t)v.size() || !name_buf) return ORTK_EINVAL;
    const Entry& e = v[index];
    std::snprintf(name_buf, 128, "%s", e.name.c_str());
    if (offset) *offset = e.offset;
    if (numel) *numel = e.numel;
    if (ndim) *ndim = e.ndim;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = e.shape[i];
    if (kind) *kind = e.kind;
    return 0;
}

static int check_batch(const ortk_config* cfg, const ortk_batch* b, bool need_seq) {
    if (!b || !b->att_feats || (!b->boxes && !cfg->no_box) || !b->att_masks) return ORTK_EINVAL;
    if (b->B < 1 || b->S < 1 || b->S > 128) return ORTK_EINVAL;
    if (need_seq) {
        if (!b->seqs || b->R < 1 || b->T < 1 || b->R % b->B || b->seq_stride < b->T + 1) return ORTK_EINVAL;
        if (b->T > 64 || (int64_t)(b->R / b->B) * b->T > 4096) return ORTK_EINVAL;
        if (b->T > cfg->seq_len) return ORTK_EINVAL;
        if (b->cap_off || b->row_pos || b->Mc) {       // valid-position layout: all three, mixed precision only
            if (!b->cap_off || !b->row_pos || b->Mc < b->R || (int64_t)b->Mc > (int64_t)b->R * b->T || cfg->precision != 1) return ORTK_EINVAL;
        }
    }
    return 0;
}
// decoder rows of the step: the valid positions only (ortk_batch.cap_off / row_pos / Mc) or all R x T
static inline bool batch_compact(const ortk_batch* b) { return b->row_pos != nullptr; }

// 1 if a training batch of this geometry may use the valid-position decoder layout (ortk_batch.cap_off / row_pos): mixed
// precision with every decoder attention on the bf16-operand kernels (dk = 64 or 32, T and captions-per-image * T <= 128)
extern "C" int ortk_valid_positions_ok(const ortk_config* cfg, int32_t B, int32_t S, int32_t R, int32_t T) {
    if (check_cfg(cfg) || B < 1 || S < 1 || R < 1 || T < 1 || R % B) return 0;
    TrainWS w; carve_train(*cfg, B, S, R, T, nullptr, w);
    return cfg->precision == 1 && w.qdt_self && w.qdt_cross ? 1 : 0;
}

extern "C" size_t ortk_train_workspace_bytes(const ortk_config* cfg, int32_t B, int32_t S, int32_t R, int32_t T) {
    if (check_cfg(cfg) || B < 1 || S < 1 || R < 1 || T < 1) return 0;
    TrainWS w; carve_train(*cfg, B, S, R, T, nullptr, w);
    return w.bytes;
}

extern "C" int ortk_forward(const ortk_config* cfg, const float* params, const ortk_batch* bt, void* ws, size_t ws_bytes,
                            float* logp_out, int64_t ldv_out, int32_t train, uint64_t seed, ortk_stream stream) {
    return ortk_forward_phase(cfg, params, bt, ws, ws_bytes, logp_out, ldv_out, train, seed, 0, stream);
}

extern "C" void* ortk_train_workspace_memory(const ortk_config* cfg, int32_t B, int32_t S, int32_t R, int32_t T, void* ws, int32_t* dtype) {
    if (check_cfg(cfg) || B < 1 || S < 1 || R < 1 || T < 1 || !ws) return nullptr;
    TrainWS w; carve_train(*cfg, B, S, R, T, ws, w);
    if (dtype) *dtype = w.adt;
    return w.mem;
}

// phase 0: the whole forward.  phase 1: weight copies + encoder only (the memory stays in `ws`: ortk_train_workspace_memory);
// phase 2: the decoder and generator on the encoder state phase 1 left in the same workspace (same params / train / seed).
// An SCST step runs its encoder ONCE this way: phase 1, the rollout decode on that memory (ortk_decode_opts.memory), phase 2 on
// the sampled captions, backward.
extern "C" int ortk_forward_phase(const ortk_config* cfg, const float* params, const ortk_batch* bt, void* ws, size_t ws_bytes,
                                  float* logp_out, int64_t ldv_out, int32_t train, uint64_t seed, int32_t phase, ortk_stream stream) {
    if (phase < 0 || phase > 2) return ORTK_EINVAL;
    if (int e = check_cfg(cfg)) return e;
    if (int e = check_batch(cfg, bt, phase != 1)) return e;
    if (phase == 1 && (bt->R < 1 || bt->T < 1 || bt->R % bt->B || bt->T > cfg->seq_len)) return ORTK_EINVAL;     // (they shape the workspace)
    if (!params || !ws) return ORTK_EINVAL;
    Offsets o; build_layout(*cfg, o, nullptr);
    TrainWS w; carve_train(*cfg, bt->B, bt->S, bt->R, bt->T, ws, w);
    if (w.bytes > ws_bytes) return ORTK_ENOSPC;
    if (logp_out && (ldv_out < cfg->vocab)) return ORTK_EINVAL;
    Ctx c{cfg, ortk_s(stream), cfg->precision, seed, train != 0, params, w.w16, w.adt};
    c.side = (c.adt == ORTK_BF16 && !ortk_prof_serial()) ? side_for(c.s) : nullptr;
    c.use_side = c.side != nullptr;
    // The geometry bias reads the fp32 parameters: with the side stream it starts BEFORE the bf16 weight copies are made
    // (0.14 ms of casts it used to wait behind; the encoder's first attention then waited for it), and the transposed copy,
    // which only the backward reads, is made on the side stream behind it.
    hipEvent_t box_done = nullptr;
    const bool box_early = c.use_side && !cfg->no_box && phase != 2;
    // mixed precision with the side stream: the region features get a bf16 copy there, ahead of the geometry bias (25 us for 75 MB): the
    // att_embed product then runs on the LDS-DMA tiles instead of the fp32-operand kernel (89 -> 35 us at the head of the step), and its
    // weight gradient joins the last grouped launch of the backward (it was a 120-us launch of its own at the very end)
    hipEvent_t feats16_done = nullptr;
    const bool f16 = c.use_side && w.feats16 && phase != 2 && (cfg->feat % 64) == 0 && tuning().feats_bf16;
    if (phase != 2) {
        if (f16) {
            TRY(ortk_cast_bf16(bt->att_feats, w.feats16, w.Me * cfg->feat, (ortk_stream)c.side->s));
            TRY(c.side_mark(&feats16_done));
        }
        if (box_early) TRY(queue_box_bias(c, o, bt->boxes, w.logbias, bt->B, bt->S, &box_done));
        TRY(make_w16(cfg, o, params, w.w16, stream));
        if (c.use_side) {
            if (!box_early) TRY(c.fork());      // (the previous backward on the caller's stream still reads the old copy)
            TRY(make_w16t(cfg, o, params, w.w16t, (ortk_stream)c.side->s));     // done before the decoder prefix the forward waits for
            TRY(c.side_mark(nullptr));
        } else {
            TRY(make_w16t(cfg, o, params, w.w16t, stream));      // read by the backward that follows this forward
        }
    }
    // sparse plans are rebuilt from THIS call's effective weights (a new mask sample per step): no stale images
    c.ell_f = cfg->sparse_fwd;        // (built below, once it is known who uses it first)
    const float* P = params;
    const int d = cfg->d_model, ff = cfg->d_ff, H = cfg->n_heads, L = cfg->n_layers, dk = d / H, V = cfg->vocab, A = w.adt;
    const int B = bt->B, S = bt->S, R = bt->R, T = bt->T, spi = R / B;
    const bool compact = batch_compact(bt);
    if (compact && (logp_out || !w.qdt_self || !w.qdt_cross)) return ORTK_EINVAL;     // fused criterion + bf16-operand attention only
    // rows-stationary chains: their weight units go into streaming order once per forward (phase 2 finds phase 1's image)
    ChainSet cs; chain_layout(*cfg, o, true, true, cs, w.Me, batch_compact(bt) ? bt->Mc : w.Md);
    // (a sparse forward plan that takes a product of a chain switches the chains off; one over the other products — region embedding,
    //  K|V projection of the memory, generator — or a data-gradient plan alone leaves them on)
    if (cs.on && (!w.chain_pk || plan_hits_chain(c.ell_f, cs))) cs.on = false;
    const bool dchains = cs.on && w.qdt_self == ORTK_BF16 && w.qdt_cross == ORTK_BF16 && A == ORTK_BF16;
    hipEvent_t fplan_done = nullptr;
    if (cfg->sparse_fwd && phase != 2) {
        // With the chains on, no product of a layer is in the plan; unless the region embedding is, its first user is the K|V
        // projection of the memory, after the encoder: the build (0.2 ms) then runs beside the encoder on the side stream.
        const void* src = cfg->precision ? (const void*)w.w16 : (const void*)params;
        const int sdt = cfg->precision ? ORTK_BF16 : ORTK_F32;
        if (cs.on && c.use_side && Ctx::ell_block(c.ell_f, o.att_w, cfg->d_model, cfg->feat) < 0) {
            TRY(c.fork());                        // (the bf16 weight copy is made on the caller's stream)
            TRY(ortk_sparse_build(cfg->sparse_fwd, src, sdt, (ortk_stream)c.side->s));
            TRY(c.side_mark(&fplan_done));
        } else {
            TRY(ortk_sparse_build(cfg->sparse_fwd, src, sdt, stream));
        }
    }
    const int64_t Me = w.Me, Md = compact ? bt->Mc : w.Md;
    // Phase 1 has no captions, so it packs for the kernel forms of (Me, R * T).  A phase 2 on the valid positions whose row count
    // picks the OTHER form of the chain kernel (76-row blocks stream the FFN units in another order) packs again, from the same bf16
    // weight copy, for the layout chain_run() will derive from its own row count (the encoder's units, already consumed, ride along).
    const bool repack2 = phase == 2 && cs.on && chain_wide(w.Md) != chain_wide(Md);
    // (with the side streams: on the third queue, beside att_embed — 117 us the first chain used to wait for on the caller's stream;
    //  the caller's stream and the side stream's decoder prefix wait for it right before their first chain)
    hipEvent_t pack_done = nullptr;
    if (cs.on && (phase != 2 || repack2)) {
        if (c.use_side && phase != 2 && tuning().feats_bf16) {
            hipEvent_t w16_ready = c.side->take();
            pack_done = c.side->take();
            if (hipEventRecord(w16_ready, c.s) != hipSuccess || hipStreamWaitEvent(c.side->s2, w16_ready, 0) != hipSuccess) return ORTK_EINVAL;
            TRY(chain_pack_all(w.w16, w.chain_pk, cs.t, c.side->s2));
            if (hipEventRecord(pack_done, c.side->s2) != hipSuccess) return ORTK_EINVAL;
        } else {
            TRY(chain_pack_all(w.w16, w.chain_pk, cs.t, c.s));
        }
    }
    EncPtrs ep[MAXLAYERS]; enc_ptrs_from_ws(w, L, ep);
    const AttMode am = att_mode(cfg->share_att_dec);
    const int64_t cw = o.cw, cv = o.cv;
    // Self-attention sublayer (and the cross-attention query projection) of decoder layer l, on context `cx`'s stream.
    auto self_part = [&](const Ctx& cx, int l, const float* x) -> int {
        const DecOff& e = o.dec[l]; const DecBuf& b = w.dec[l];
        if (dchains) {
            if (l == 0) {             // (layers 1.. get their LayerNorm 0 and Q|K|V from the chain that ends layer l - 1)
                ortk_chain_args ca; std::memset(&ca, 0, sizeof(ca));
                ca.n_units = cs.units(cs.d0); ca.M = Md; ca.x_in = x;
                ca.g1 = P + e.n0a; ca.b1 = P + e.n0b; ca.y1 = b.y1; ca.st1 = b.st1;
                ca.n1 = 3; ca.bias_s1 = P + e.bqkv; ca.out1 = b.qkv; ca.ld1 = 3 * d; ca.eps = 1e-6f;
                TRY(chain_run(&ca, cs.stream_of(w.chain_pk, cs.d0), cx.s));
            }
        } else {
            TRY(ln_fwd(cx, x, e.n0a, e.n0b, b.y1, A, b.st1, Md));
            TRY(fwd_gemm(cx, b.y1, A, d, e.wqkv, P + e.bqkv, b.qkv, w.qdt_self, 3 * d, Md, am.n * d, d));
        }
        ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = cx.prec; a.qkv_dtype = w.qdt_self;
        a.q = (const float*)b.qkv; a.k = (const float*)off_elems(b.qkv, am.k * d, w.qdt_self); a.v = (const float*)off_elems(b.qkv, am.v * d, w.qdt_self);
        a.ldq = a.ldk = a.ldv = 3 * d; a.o = b.o1; a.o_dtype = A; a.ldo = d;
        a.kmask = w.keymask; a.p = b.Ps; a.nkv = R; a.H = H; a.Lq = T; a.Lk = T; a.dk = dk; a.causal_period = T;
        a.drop_p = cx.p_drop(); a.drop_seed = cx.sub(dop(l, 0));
        if (compact) { a.q_off = bt->cap_off; a.q_off_stride = 1; a.kv_ragged = 1; }      // a caption's keys are its own (valid) rows
        TRY(ortk_attention_fwd(&a, (ortk_stream)cx.s));
        if (dchains) {                // [Wo + x -> LayerNorm 1 -> Wcq]
            ortk_chain_args ca; std::memset(&ca, 0, sizeof(ca));
            ca.n_units = cs.units(cs.db[l]); ca.M = Md; ca.x_in = x;
            ca.a_in = b.o1; ca.bias_r = P + e.bo; ca.x_mid = b.xm1; ca.seed_r = cx.sub(dop(l, 1));
            ca.g1 = P + e.n1a; ca.b1 = P + e.n1b; ca.y1 = b.y2; ca.st1 = b.st2;
            ca.n1 = 1; ca.bias_s1 = P + e.cqb; ca.out1 = b.qc; ca.ld1 = d;
            ca.drop_p = cx.p_drop(); ca.eps = 1e-6f; ca.drop_rows = cx.drop_rows;
            return chain_run(&ca, cs.stream_of(w.chain_pk, cs.db[l]), cx.s);
        }
        TRY(fwd_gemm(cx, b.o1, A, d, e.wo, P + e.bo, b.xm1, ORTK_F32, d, Md, d, d, false, cx.p_drop(), cx.sub(dop(l, 1)), x, d));
        TRY(ln_fwd(cx, b.xm1, e.n1a, e.n1b, b.y2, A, b.st2, Md));
        TRY(fwd_gemm(cx, b.y2, A, d, e.cqw, P + e.cqb, b.qc, w.qdt_cross, d, Md, d, d));
        return 0;
    };
    // The token side of decoder layer 0 (embedding, self-attention sublayer, cross-attention query) does not depend on the
    // encoder: with the side stream it runs beside the encoder stack (after the geometry bias, which the encoder needs first).
    hipEvent_t prefix_done = nullptr;
    const bool prefix_side = c.use_side && phase == 0;       // (split phases: the token side has no encoder to run beside)
    if (prefix_side) {
        TRY(c.fork());                     // the side stream sees the bf16 weight copy
        if (pack_done && hipStreamWaitEvent(c.side->s, pack_done, 0) != hipSuccess) return ORTK_EINVAL;      // ... and the packed chain weights
    }
    if (phase != 2)
        TRY(encoder_forward(c, o, bt->att_feats, bt->boxes, bt->att_masks, B, S, w.x0, w.logbias, ep, w.mem, A, w.st_mem, w.qdt_enc,
                            box_early ? &box_done : nullptr, &cs, w.chain_pk, true, f16 ? w.feats16 : nullptr, feats16_done, pack_done));
    else TRY(c.wait_ev(pack_done));
    // From here on every operator with a dropout site runs on DECODER rows: on the valid positions it draws what the padded
    // (caption, position) layout draws (a step is the same function of its seed in both layouts; an SCST update on the valid
    // positions reproduces its train-mode rollout's masks).
    c.drop_rows = compact ? bt->row_pos : nullptr;
    if (phase == 1) {
        // (the data-gradient plan is built by phase 2, from this phase's transposed copy: until then no backward may take the
        //  images a previous step left in the plan for current)
        if (cfg->sparse_bwd) bwd_plan_forget(cfg->sparse_bwd);
        return c.join();                   // (the transposed weight copy of the side stream included)
    }
    TRY(c.wait_ev(fplan_done));
    {
        const Ctx cx = prefix_side ? c.on_side() : c;
        // (no_pad_keys: rollout captions — the key mask hides nothing, as in the cached passes that drew them; ortk_batch)
        TRY(embed_fwd_rows(bt->seqs, bt->seq_stride, P + o.lut, P + o.pe, w.dx0, w.keymask, Md, compact ? bt->row_pos : nullptr, T, 0, d,
                           bt->no_pad_keys ? -1 : cfg->pad_id, c.p_drop(), c.sub(OP_EMB), cx.s));
        if (prefix_side) { TRY(self_part(cx, 0, w.dx0)); TRY(c.side_mark(&prefix_done)); }
    }
    // the data-gradient plan's images (from this call's transposed weights: this step's mask sample) are built beside the forward
    // — 0.2 ms the backward used to start with — behind everything the forward waits for on the side stream
    hipEvent_t bplan_done = nullptr;
    if (cfg->precision && cfg->sparse_bwd && w.w16t) {          // (phases 0 and 2: EVERY forward that a backward can follow builds and marks)
        if (c.use_side) {
            TRY(ortk_sparse_build(cfg->sparse_bwd, w.w16t, ORTK_BF16, (ortk_stream)c.side->s));
            TRY(c.side_mark(&bplan_done));
        } else {
            TRY(ortk_sparse_build(cfg->sparse_bwd, w.w16t, ORTK_BF16, stream));
        }
        bwd_plan_built_by(cfg->sparse_bwd, ws, seed, train != 0);
    }
    // decoder
    const int U = o.ckv_slots;            // distinct decoder layers: one K|V slice each in the packed projection
    TRY(fwd_gemm(c, w.mem, A, d, o.ckv_w, P + o.ckv_b, w.ckv, w.qdt_cross, U * cw, Me, (int)(U * cw), d));
    const float* x = w.dx0;
    for (int l = 0; l < L; ++l) {
        const DecOff& e = o.dec[l]; const DecBuf& b = w.dec[l];
        if (l == 0 && prefix_side) TRY(c.wait_ev(prefix_done)); else TRY(self_part(c, l, x));
        ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
        a.qkv_dtype = w.qdt_cross; a.q = (const float*)b.qc; a.ldq = d;
        a.k = (const float*)off_elems(w.ckv, o.ckv_slot[l] * cw, w.qdt_cross); a.v = (const float*)off_elems(w.ckv, o.ckv_slot[l] * cw + cv, w.qdt_cross);
        a.ldk = a.ldv = U * cw;
        a.o = b.o2; a.o_dtype = A; a.ldo = d; a.kmask = bt->att_masks; a.p = b.Pc; a.nkv = B; a.H = H; a.Lq = spi * T; a.Lk = S; a.dk = dk;
        a.drop_p = c.p_drop(); a.drop_seed = c.sub(dop(l, 2));
        if (compact) { a.q_off = bt->cap_off; a.q_off_stride = spi; a.drop_rows = bt->row_pos; }      // an image's rows: those of its spi captions
        TRY(ortk_attention_fwd(&a, stream));
        if (dchains) {                // [Wco + x -> LayerNorm 2 -> FFN + x -> LayerNorm 0 of layer l + 1 (or the stack's) -> next Wqkv]
            ortk_chain_args ca; std::memset(&ca, 0, sizeof(ca));
            ca.n_units = cs.units(cs.dc[l]); ca.M = Md; ca.x_in = b.xm1;
            ca.a_in = b.o2; ca.bias_r = P + e.cob; ca.x_mid = b.xm2; ca.seed_r = c.sub(dop(l, 3));
            ca.g1 = P + e.n2a; ca.b1 = P + e.n2b; ca.y1 = b.y3; ca.st1 = b.st3;
            ca.NC = ff / 512; ca.bias_h = P + e.b1; ca.bias_o = P + e.b2; ca.h = b.h; ca.x_out = b.xout;
            ca.seed_h = c.sub(dop(l, 4)); ca.seed_o = c.sub(dop(l, 5));
            if (l + 1 < L) {
                const DecOff& en = o.dec[l + 1]; const DecBuf& bn = w.dec[l + 1];
                ca.g2 = P + en.n0a; ca.b2 = P + en.n0b; ca.y2 = bn.y1; ca.st2 = bn.st1;
                ca.n2 = 3; ca.bias_s2 = P + en.bqkv; ca.out2 = bn.qkv; ca.ld2 = 3 * d;
            } else {
                ca.g2 = P + o.dec_na; ca.b2 = P + o.dec_nb; ca.y2 = w.dec_out; ca.st2 = w.st_out;
            }
            ca.drop_p = c.p_drop(); ca.eps = 1e-6f; ca.drop_rows = c.drop_rows;
            TRY(chain_run(&ca, cs.stream_of(w.chain_pk, cs.dc[l]), c.s));
            x = b.xout;
            continue;
        }
        TRY(fwd_gemm(c, b.o2, A, d, e.cow, P + e.cob, b.xm2, ORTK_F32, d, Md, d, d, false, c.p_drop(), c.sub(dop(l, 3)), b.xm1, d));
        TRY(ln_fwd(c, b.xm2, e.n2a, e.n2b, b.y3, A, b.st3, Md));
        TRY(fwd_gemm(c, b.y3, A, d, e.w1, P + e.b1, b.h, A, ff, Md, ff, d, true, c.p_drop(), c.sub(dop(l, 4))));
        TRY(fwd_gemm(c, b.h, A, ff, e.w2, P + e.b2, b.xout, ORTK_F32, d, Md, d, ff, false, c.p_drop(), c.sub(dop(l, 5)), b.xm2, d));
        x = b.xout;
    }
    if (!dchains) TRY(ln_fwd(c, x, o.dec_na, o.dec_nb, w.dec_out, A, w.st_out, Md));
    const int Vp = (int)w.ldv;     // padded vocabulary (zero weight rows / bias): full GEMM tiles
    if (logp_out) {
        const int Nout = ldv_out >= Vp ? Vp : V;
        TRY(fwd_gemm(c, w.dec_out, A, d, o.gen_w, P + o.gen_b, logp_out, ORTK_F32, ldv_out, Md, Nout, d));
        TRY(ortk_log_softmax(logp_out, Md, V, ldv_out, 1.f, stream));
    } else {
        TRY(fwd_gemm(c, w.dec_out, A, d, o.gen_w, P + o.gen_b, w.logits, ORTK_F32, w.ldv, Md, Vp, d));
    }
    TRY(c.wait_ev(bplan_done));            // (long finished: orders the backward, a later call on this stream, behind the build)
    return 0;
}

extern "C" int ortk_loss(const ortk_config* cfg, const ortk_batch* bt, void* ws, size_t ws_bytes, const float* norm_dev,
                         float* loss_dev, ortk_stream stream) {
    if (int e = check_cfg(cfg)) return e;
    if (int e = check_batch(cfg, bt, true)) return e;
    if (!ws || !norm_dev || !loss_dev || !bt->tok_weight) return ORTK_EINVAL;
    TrainWS w; carve_train(*cfg, bt->B, bt->S, bt->R, bt->T, ws, w);
    if (w.bytes > ws_bytes) return ORTK_ENOSPC;
    const bool compact = batch_compact(bt);
    return xent_rows(w.logits, bt->seqs + 1, bt->seq_stride, bt->T, bt->tok_weight, norm_dev, loss_dev, w.row_loss, compact ? bt->Mc : w.Md,
                     compact ? bt->row_pos : nullptr, cfg->vocab, w.ldv, w.dlogits, w.adt, w.ldv, ortk_s(stream));
}

extern "C" int ortk_loss_external(const ortk_config* cfg, const ortk_batch* bt, void* ws, size_t ws_bytes, const float* logp,
                                  const float* dlogp, int64_t ldv, ortk_stream stream) {
    if (int e = check_cfg(cfg)) return e;
    if (int e = check_batch(cfg, bt, true)) return e;
    if (batch_compact(bt)) return ORTK_EINVAL;      // log-probs (R, T, V) exist in the padded layout only
    if (!ws || !logp || !dlogp || ldv < cfg->vocab) return ORTK_EINVAL;
    TrainWS w; carve_train(*cfg, bt->B, bt->S, bt->R, bt->T, ws, w);
    if (w.bytes > ws_bytes) return ORTK_ENOSPC;
    return ortk_log_softmax_bwd(logp, dlogp, ldv, w.dlogits, w.adt, w.ldv, w.Md, cfg->vocab, stream);
}

extern "C" int64_t ortk_arena_decoder_offset(const ortk_config* cfg) {
    if (check_cfg(cfg)) return -1;
    Offsets o; build_layout(*cfg, o, nullptr);
    return o.dec[0].wqkv;
}

extern "C" int ortk_backward(const ortk_config* cfg, const float* params, float* grads, const ortk_batch* bt, void* ws,
                             size_t ws_bytes, int32_t train, uint64_t seed, ortk_stream stream) {
    return ortk_backward_phase(cfg, params, grads, bt, ws, ws_bytes, train, seed, 0, stream);
}

extern "C" int ortk_backward_phase(const ortk_config* cfg, const float* params, float* grads, const ortk_batch* bt, void* ws,
                                   size_t ws_bytes, int32_t train, uint64_t seed, int32_t phase, ortk_stream stream) {
    if (phase < 0 || phase > 2) return ORTK_EINVAL;
    if (int e = check_cfg(cfg)) return e;
    if (int e = check_batch(cfg, bt, true)) return e;
    if (!params || !grads || !ws) return ORTK_EINVAL;
    Offsets o; build_layout(*cfg, o, nullptr);
    TrainWS w; carve_train(*cfg, bt->B, bt->S, bt->R, bt->T, ws, w);
    if (w.bytes > ws_bytes) return ORTK_ENOSPC;
    // (train, seed) must match the forward's; the bf16 weight copy made by the forward is still in the workspace
    Ctx c{cfg, ortk_s(stream), cfg->precision, seed, train != 0, params, w.w16, w.adt};
    c.W16T = w.w16t;
    if (cfg->precision && cfg->sparse_bwd) {
        // Rebuilt HERE from this workspace's transposed bf16 weights (the mask sample of THIS graph's forward): the plan's
        // buffers are shared by every workspace of the model, so another forward (a second autograd graph, another seed) may
        // have rebuilt them since.  Phase 2 of a split backward reuses phase 1's build.
        if (phase != 2 && !bwd_plan_is_of(cfg->sparse_bwd, ws, seed, train != 0)) {
            TRY(ortk_sparse_build(cfg->sparse_bwd, w.w16t, ORTK_BF16, stream));
            bwd_plan_built_by(cfg->sparse_bwd, ws, seed, train != 0);
        }
        c.ell_b = cfg->sparse_bwd;
    }
    c.side = (c.adt == ORTK_BF16 && !ortk_prof_serial()) ? side_for(c.s) : nullptr;
    c.use_side = c.side != nullptr;
    float* G = grads;
    const int d = cfg->d_model, ff = cfg->d_ff, H = cfg->n_heads, L = cfg->n_layers, dk = d / H, A = w.adt;
    const int B = bt->B, S = bt->S, R = bt->R, T = bt->T, spi = R / B;
    const bool compact = batch_compact(bt);
    if (compact && (!w.qdt_self || !w.qdt_cross)) return ORTK_EINVAL;
    const int64_t Me = w.Me, Md = compact ? bt->Mc : w.Md;
    const float inv_keep = c.p_drop() > 0.f ? 1.f / (1.f - c.p_drop()) : 1.f;
    // dK|dV slice of every decoder layer in w.gkv: distinct layers first (the order of the packed K|V weight block), the
    // layers that share one of them behind
    const int U = o.ckv_slots;
    int gslot[MAXLAYERS];
    { int extra = U; for (int l = 0; l < L; ++l) gslot[l] = cfg->share_dec[l] > 0 ? extra++ : o.ckv_slot[l]; }
    // share_att: column blocks of w.gqkv that receive dQ (0) / dK / dV, and the add that folds the shared projection's two
    // gradients together before the (rows, n*d) projection GEMMs.  Cross-attention with "kv": all dK slices (width d)
    // first, all dV slices behind them (column L*d on), folded by one add over L*d columns.
    const AttMode ame = att_mode(cfg->share_att_enc), amd = att_mode(cfg->share_att_dec);
    const int64_t cw = o.cw, cv = o.cv, ldg = (int64_t)L * 2 * d, gdv = cfg->share_att_dec == 1 ? (int64_t)L * d : d;
    auto fold = [&](const AttMode& m, int64_t rows, void* gq) -> int {
        if (m.gsrc < 0) return 0;
        return ortk_axpy_cols(off_elems(gq, (int64_t)m.gsrc * d, A), off_elems(gq, (int64_t)m.gdst * d, A), A, 3 * d, rows, d, stream);
    };

    float* dx = w.ga; float* dx2 = w.gb;
    // the (rows, d) bf16 gradient temporary rotates over three buffers: its producers (ln_bwd's masked copy, the cross-
    // attention dQ, the att_embed gate) never have to wait for the side-stream wgrad that still reads the previous one
    // (grouped weight gradients: ten, so that the four of a layer stay untouched until the layer's launch has read them)
    const int wgm = tuning().wgrad_group;
    const bool grouped = A == ORTK_BF16 && (wgm & 7) != 0 && w.gtp[TrainWS::NGT - 1] != nullptr;
    const int n_gt = grouped ? TrainWS::NGT : 3;
    void* gt_pool[TrainWS::NGT] = {w.gt, w.gt2, w.gt3};
    for (int i = 3; i < n_gt; ++i) gt_pool[i] = w.gtp[i];
    int gt_i = 0;
    void* gt_cur = w.gt;
    auto gt_new = [&]() { gt_i = (gt_i + 1) % n_gt; gt_cur = gt_pool[gt_i]; return gt_cur; };
    c.wg_ws = w.wg_ws; c.wg_ws_bytes = w.wg_ws_bytes;
    // Backward chains (ortk_chain.hip): everything between two attention-backward calls except the weight gradients in one
    // rows-stationary launch each.  Mixed precision, bf16 Q / K / V / dO in all three attention stacks, dense products.
    ChainSet bs; bchain_layout(*cfg, o, bs);
    const bool bch_any = bs.on && w.chain_pk_b && A == ORTK_BF16 && w.qdt_self == ORTK_BF16 && w.qdt_cross == ORTK_BF16 && w.qdt_enc == ORTK_BF16 &&
                         !c.ell_b && !cfg->sparse_fwd;
    // ortk_tuning.row_chain: 1 = forward chains only, 2 = + the encoder's backward (one round of workgroups at 9 216 rows), 3 = + the decoder's
    const bool bch_enc = bch_any && tuning().row_chain >= 2, bch = bch_any && tuning().row_chain >= 3;
    if (bch_enc && phase != 2) TRY(chain_pack_all(w.w16t, w.chain_pk_b, bs.t, c.s));
    const bool grp_layers = grouped && (wgm & 1), grp_gen = grouped && (wgm & 2), grp_kv = grouped && (wgm & 4);
    // the bf16 feature copy of this step's forward (same conditions there) lets att_embed's weight gradient join layer 0's group
    const bool tail_group = grp_layers && c.use_side && w.feats16 && (cfg->feat % 64) == 0 && tuning().feats_bf16 && !bch_enc;
    auto gq_of = [&](int l) { return (grp_layers && (l & 1)) ? w.gqkv2 : w.gqkv; };
    auto gh_of = [&](int l) { return (grp_layers && (l & 1)) ? w.gh2 : w.gh; };
    const int NCc = ff / 512;
    int gh_i = 0;
    auto gh_new = [&]() { gh_i ^= 1; return gh_i ? w.gh2 : w.gh; };
    // X = [FFN' -> LayerNorm' -> . Wout] of one layer, standalone (the top of a stack) or as the tail of a ZX chain: the fields of `ca`
    auto x_part = [&](ortk_bchain_args& ca, const void* hrow, void* ghbuf, const float* xrows, const float* strows, int64_t na, int64_t nb,
                      const float* dres, float* dxo, void* dzo, uint32_t seed, void* dO) {
        ca.NC = NCc; ca.hgate = hrow; ca.gh = ghbuf; ca.gate_scale = inv_keep;
        ca.xb = xrows; ca.stb = strows; ca.gb = params + na; ca.dresb = dres; ca.dxb = dxo; ca.dab = G + na; ca.dbb = G + nb;
        ca.dzb = dzo; ca.seed_b = seed; ca.mask_b = 1;
        ca.n2 = 1; ca.out2 = dO;
    };
    if (phase != 2) {
    c.drop_rows = compact ? bt->row_pos : nullptr;      // (decoder rows: the masked copies draw what the forward drew; reset below)
    // ---- decoder half: generator, decoder stack, token embedding, cross-attention K/V projections; leaves the gradient
    // of the encoder memory in w.gy.  Every gradient at arena offsets >= ortk_arena_decoder_offset is final afterwards.
    // generator (over the padded vocabulary: pad columns of dlogits are exact zeros)
    const int Vp = (int)w.ldv;
    c.grp_on = grp_gen;
    TRY(wgrad_gemm(c, w.dlogits, A, w.ldv, w.dec_out, A, d, G + o.gen_w, G + o.gen_b, Md, Vp, d));
    TRY(flush_wgrads(c));
    c.grp_on = false;
    TRY(dgrad_gemm(c, w.dlogits, A, w.ldv, o.gen_w, w.gy, ORTK_F32, d, Md, Vp, d));
    TRY(ln_bwd(c, w.gy, w.dec[L - 1].xout, G, o.dec_na, o.dec_nb, w.st_out, nullptr, dx, Md, gt_new(), dop(L - 1, 5)));
    if (bch) {
        // cur = gradient of the FFN sublayer's INPUT rows (xm2) of the layer in hand, oth = the other fp32 buffer; dtA / dtB = the masked
        // bf16 gradients the weight gradients of W2 / Wco read; ghb = the FFN hidden gradient (W1's weight gradient)
        float* cur = dx2; float* oth = dx;
        void* dtA = gt_cur; void* dtB = gt_new(); void* ghb = gh_new();
        {   // X of the top layer, on the masked gradient the final LayerNorm's backward left
            const DecOff& e = o.dec[L - 1]; const DecBuf& b = w.dec[L - 1];
            ortk_bchain_args ca; std::memset(&ca, 0, sizeof(ca)); ca.drop_rows = c.drop_rows;
            ca.n_units = bs.units(bs.bx_top); ca.M = Md; ca.nin = 0; ca.dz0 = dtA;
            x_part(ca, b.h, ghb, b.xm2, b.st3, e.n2a, e.n2b, dx, cur, dtB, c.sub(dop(L - 1, 3)), w.gdo);
            ca.drop_p = c.p_drop(); ca.eps = 1e-6f;
            TRY(c.before_write(dtB)); TRY(c.before_write(ghb));
            TRY(bchain_run(&ca, bs.stream_of(w.chain_pk_b, bs.bx_top), c.s));
        }
        for (int l = L - 1; l >= 0; --l) {
            const DecOff& e = o.dec[l]; const DecBuf& b = w.dec[l];
            const float* xin = l == 0 ? w.dx0 : w.dec[l - 1].xout;
            void* const gql = gq_of(l);
            c.grp_on = grp_layers;
            TRY(wgrad_gemm(c, dtA, A, d, b.h, A, ff, G + e.w2, G + e.b2, Md, d, ff));
            TRY(wgrad_gemm(c, ghb, A, ff, b.y3, A, d, G + e.w1, G + e.b1, Md, ff, d));
            TRY(wgrad_gemm(c, dtB, A, d, b.o2, A, d, G + e.cow, G + e.cob, Md, d, d));
            // cross-attention backward: dO = w.gdo (the chain's last product) -> dq, dK | dV
            void* dq = gt_new();
            ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
            a.qkv_dtype = w.qdt_cross; a.q = (const float*)b.qc; a.ldq = d;
            a.k = (const float*)off_elems(w.ckv, o.ckv_slot[l] * cw, w.qdt_cross); a.v = (const float*)off_elems(w.ckv, o.ckv_slot[l] * cw + cv, w.qdt_cross);
            a.ldk = a.ldv = U * cw;
            a.p = b.Pc; a.nkv = B; a.H = H; a.Lq = spi * T; a.Lk = S; a.dk = dk; a.drop_p = c.p_drop(); a.drop_seed = c.sub(dop(l, 2));
            a.d_o = (const float*)w.gdo; a.lddo = d; a.dq = dq; a.lddq = d; a.dqkv_dtype = A;
            a.d_k = off_elems(w.gkv, gslot[l] * cw, A); a.dv = off_elems(w.gkv, gslot[l] * cw + gdv, A);
            a.lddk = a.lddv = ldg;
            if (compact) { a.q_off = bt->cap_off; a.q_off_stride = spi; a.drop_rows = bt->row_pos; }
            TRY(c.before_write(dq));
            TRY(ortk_attention_bwd(&a, stream));
            TRY(wgrad_gemm(c, dq, A, d, b.y2, A, d, G + e.cqw, G + e.cqb, Md, d, d));
            // Y: [dq . Wcq -> LayerNorm 1' (+ cur) -> masked copy -> . Wo]
            void* dtD = gt_new();
            {
                ortk_bchain_args ca; std::memset(&ca, 0, sizeof(ca)); ca.drop_rows = c.drop_rows;
                ca.n_units = bs.units(bs.by[l]); ca.M = Md; ca.nin = 1; ca.ain = dq; ca.ld_ain = d;
                ca.xa = b.xm1; ca.sta = b.st2; ca.ga = params + e.n1a; ca.dresa = cur; ca.dxa = oth; ca.daa = G + e.n1a; ca.dba = G + e.n1b;
                ca.dza = dtD; ca.seed_a = c.sub(dop(l, 1)); ca.mask_a = 1;
                ca.n2 = 1; ca.out2 = w.gdo;
                ca.drop_p = c.p_drop(); ca.eps = 1e-6f;
                TRY(c.before_write(dtD));
                TRY(bchain_run(&ca, bs.stream_of(w.chain_pk_b, bs.by[l]), c.s));
            }
            TRY(wgrad_gemm(c, dtD, A, d, b.o1, A, d, G + e.wo, G + e.bo, Md, d, d));
            // self-attention backward: dO = w.gdo -> packed dQ | dK | dV
            std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
            a.qkv_dtype = w.qdt_self; a.q = (const float*)b.qkv; a.k = (const float*)off_elems(b.qkv, amd.k * d, w.qdt_self);
            a.v = (const float*)off_elems(b.qkv, amd.v * d, w.qdt_self); a.ldq = a.ldk = a.ldv = 3 * d;
            a.p = b.Ps; a.nkv = R; a.H = H; a.Lq = T; a.Lk = T; a.dk = dk; a.drop_p = c.p_drop(); a.drop_seed = c.sub(dop(l, 0));
            a.d_o = (const float*)w.gdo; a.lddo = d; a.dqkv_dtype = A;
            a.dq = gql; a.d_k = off_elems(gql, amd.gk * d, A); a.dv = off_elems(gql, amd.gv * d, A); a.lddq = a.lddk = a.lddv = 3 * d;
            if (compact) { a.q_off = bt->cap_off; a.q_off_stride = 1; a.kv_ragged = 1; }
            TRY(c.before_write(gql));
            TRY(ortk_attention_bwd(&a, stream));
            TRY(wgrad_gemm(c, gql, A, 3 * d, b.y1, A, d, G + e.wqkv, G + e.bqkv, Md, 3 * d, d));
            TRY(flush_wgrads(c, l == 0));
            c.grp_on = false;
            // Z (+ X of the layer below): [dQKV . Wqkv -> LayerNorm 0' (+ oth) -> masked copy] [-> FFN' -> LayerNorm 2' -> masked copy -> . Wco]
            ortk_bchain_args ca; std::memset(&ca, 0, sizeof(ca)); ca.drop_rows = c.drop_rows;
            ca.M = Md; ca.nin = 3; ca.ain = gql; ca.ld_ain = 3 * d;
            ca.xa = xin; ca.sta = b.st1; ca.ga = params + e.n0a; ca.dresa = oth; ca.dxa = cur; ca.daa = G + e.n0a; ca.dba = G + e.n0b;
            ca.drop_p = c.p_drop(); ca.eps = 1e-6f;
            if (l > 0) {
                const DecOff& en = o.dec[l - 1]; const DecBuf& bn = w.dec[l - 1];
                dtA = gt_new(); void* dtBn = gt_new(); ghb = gh_new();
                ca.n_units = bs.units(bs.bzx[l]);
                ca.dza = dtA; ca.seed_a = c.sub(dop(l - 1, 5)); ca.mask_a = 1;
                x_part(ca, bn.h, ghb, bn.xm2, bn.st3, en.n2a, en.n2b, cur, oth, dtBn, c.sub(dop(l - 1, 3)), w.gdo);
                TRY(c.before_write(dtA)); TRY(c.before_write(dtBn)); TRY(c.before_write(ghb));
                TRY(bchain_run(&ca, bs.stream_of(w.chain_pk_b, bs.bzx[l]), c.s));
                dtB = dtBn;
                std::swap(cur, oth);          // cur = gradient of xm2 of layer l - 1 (what the chain wrote last)
            } else {
                ca.n_units = bs.units(bs.bz0);
                TRY(bchain_run(&ca, bs.stream_of(w.chain_pk_b, bs.bz0), c.s));
                dx = cur; dx2 = oth;           // the gradient of the embedded tokens
            }
        }
    } else
    for (int l = L - 1; l >= 0; --l) {
        const DecOff& e = o.dec[l]; const DecBuf& b = w.dec[l];
        const float* xin = l == 0 ? w.dx0 : w.dec[l - 1].xout;
        const void* dt; int dtt;
        void* const ghl = gh_of(l); void* const gql = gq_of(l);
        c.grp_on = grp_layers;
        // feed-forward sublayer
        TRY(drop_bwd(c, dx, gt_cur, Md * d, dop(l, 5), &dt, &dtt, true));
        TRY(wgrad_gemm(c, dt, dtt, d, b.h, A, ff, G + e.w2, G + e.b2, Md, d, ff));
        TRY(dgrad_gemm(c, dt, dtt, d, e.w2, ghl, A, ff, Md, d, ff, b.h, A, ff, inv_keep));
        TRY(wgrad_gemm(c, ghl, A, ff, b.y3, A, d, G + e.w1, G + e.b1, Md, ff, d));
        TRY(dgrad_ln_bwd(c, ghl, A, ff, e.w1, Md, ff, d, w.gy, b.xm2, G, e.n2a, e.n2b, b.st3, dx, dx2, gt_new(), dop(l, 3)));
        // cross-attention sublayer
        TRY(drop_bwd(c, dx2, gt_cur, Md * d, dop(l, 3), &dt, &dtt, true));
        TRY(wgrad_gemm(c, dt, dtt, d, b.o2, A, d, G + e.cow, G + e.cob, Md, d, d));
        // the out-projection's input gradient only feeds the attention backward's products: bf16 beside bf16 Q / K / V
        void* dO = w.qdt_cross ? w.gdo : (void*)w.gy;
        TRY(dgrad_gemm(c, dt, dtt, d, e.cow, dO, w.qdt_cross, d, Md, d, d));
        ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
        a.qkv_dtype = w.qdt_cross; a.q = (const float*)b.qc; a.ldq = d;
        a.k = (const float*)off_elems(w.ckv, o.ckv_slot[l] * cw, w.qdt_cross); a.v = (const float*)off_elems(w.ckv, o.ckv_slot[l] * cw + cv, w.qdt_cross);
        a.ldk = a.ldv = U * cw;
        a.p = b.Pc; a.nkv = B; a.H = H; a.Lq = spi * T; a.Lk = S; a.dk = dk; a.drop_p = c.p_drop(); a.drop_seed = c.sub(dop(l, 2));
        a.d_o = (const float*)dO; a.lddo = d; a.dq = gt_new(); a.lddq = d; a.dqkv_dtype = A;
        a.d_k = off_elems(w.gkv, gslot[l] * cw, A); a.dv = off_elems(w.gkv, gslot[l] * cw + gdv, A);
        a.lddk = a.lddv = ldg;
        if (compact) { a.q_off = bt->cap_off; a.q_off_stride = spi; a.drop_rows = bt->row_pos; }
        TRY(c.before_write(gt_cur));
        TRY(ortk_attention_bwd(&a, stream));
        TRY(wgrad_gemm(c, gt_cur, A, d, b.y2, A, d, G + e.cqw, G + e.cqb, Md, d, d));
        {
            const void* dq_ = gt_cur;
            TRY(dgrad_ln_bwd(c, dq_, A, d, e.cqw, Md, d, d, w.gy, b.xm1, G, e.n1a, e.n1b, b.st2, dx2, dx, gt_new(), dop(l, 1), cfg->share_att_dec != 2));
        }
        // self-attention sublayer
        TRY(drop_bwd(c, dx, gt_cur, Md * d, dop(l, 1), &dt, &dtt, true));
        TRY(wgrad_gemm(c, dt, dtt, d, b.o1, A, d, G + e.wo, G + e.bo, Md, d, d));
        dO = w.qdt_self ? w.gdo : (void*)w.gy;
        TRY(dgrad_gemm(c, dt, dtt, d, e.wo, dO, w.qdt_self, d, Md, d, d));
        std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
        a.qkv_dtype = w.qdt_self; a.q = (const float*)b.qkv; a.k = (const float*)off_elems(b.qkv, amd.k * d, w.qdt_self);
        a.v = (const float*)off_elems(b.qkv, amd.v * d, w.qdt_self); a.ldq = a.ldk = a.ldv = 3 * d;
        a.p = b.Ps; a.nkv = R; a.H = H; a.Lq = T; a.Lk = T; a.dk = dk; a.drop_p = c.p_drop(); a.drop_seed = c.sub(dop(l, 0));
        a.d_o = (const float*)dO; a.lddo = d; a.dqkv_dtype = A;
        a.dq = gql; a.d_k = off_elems(gql, amd.gk * d, A); a.dv = off_elems(gql, amd.gv * d, A); a.lddq = a.lddk = a.lddv = 3 * d;
        if (compact) { a.q_off = bt->cap_off; a.q_off_stride = 1; a.kv_ragged = 1; }
        TRY(c.before_write(gql));
        TRY(ortk_attention_bwd(&a, stream));
        TRY(fold(amd, Md, gql));
        TRY(wgrad_gemm(c, gql, A, 3 * d, b.y1, A, d, G + e.wqkv, G + e.bqkv, Md, amd.n * d, d));
        TRY(flush_wgrads(c, l == 0));        // the layer's six weight gradients: one launch
        c.grp_on = false;
        TRY(dgrad_ln_bwd(c, gql, A, 3 * d, e.wqkv, Md, amd.n * d, d, w.gy, xin, G, e.n0a, e.n0b, b.st1, dx, dx2, gt_new(), l > 0 ? (int)dop(l - 1, 5) : -1));
        std::swap(dx, dx2);
    }
    // the embedding gradient (atomics into the 5 M-element table) only feeds the optimizer: beside the K/V projection GEMMs
    if (c.use_side) {
        TRY(c.fork());
        TRY(embed_bwd_rows(bt->seqs, bt->seq_stride, dx, G + o.lut, Md, compact ? bt->row_pos : nullptr, T, d, c.p_drop(), c.sub(OP_EMB), c.side->s));
        hipEvent_t e_done = nullptr;
        TRY(c.side_mark(&e_done));
        c.reads(dx, e_done);               // dx (w.ga / w.gb) is rewritten by the encoder half
    } else {
        TRY(embed_bwd_rows(bt->seqs, bt->seq_stride, dx, G + o.lut, Md, compact ? bt->row_pos : nullptr, T, d, c.p_drop(), c.sub(OP_EMB), ortk_s(stream)));
    }
    // cross-attention K/V projections of all layers, and the gradient of the encoder memory
    // layers that share weights also share the projected K|V: their dK|dV slices add up into the slice of the layer they share
    if (cfg->share_att_dec == 1)           // K = V: dK += dV, all layers at once
        TRY(ortk_axpy_cols(off_elems(w.gkv, gdv, A), w.gkv, A, ldg, Me, (int64_t)L * d, stream));
    for (int l = 0; l < L; ++l)
        if (cfg->share_dec[l] > 0)
            TRY(ortk_axpy_cols(off_elems(w.gkv, gslot[l] * cw, A), off_elems(w.gkv, o.ckv_slot[l] * cw, A), A, ldg, Me, cw, stream));
    c.grp_on = grp_kv;
    TRY(wgrad_gemm(c, w.gkv, A, ldg, w.mem, A, d, G + o.ckv_w, G + o.ckv_b, Me, (int)(U * cw), d));
    TRY(flush_wgrads(c, true));
    c.grp_on = false;
    TRY(dgrad_gemm(c, w.gkv, A, ldg, o.ckv_w, w.gy, ORTK_F32, d, Me, (int)(U * cw), d));
    TRY(c.join());
    c.drop_rows = nullptr;
    }
    if (phase == 1) return 0;
    // ---- encoder half (reads the memory gradient left in w.gy)
    // Gradient of the geometry-bias weights of layers [l0, l0 + n): only feeds the optimizer.  With the side stream the
    // layers 1 .. L-1 go as soon as layer 1's attention backward has written their score gradients — one 0.36 ms VALU
    // kernel for all layers at the very end of the step had the att_embed weight gradient as its only company (the
    // embedding's sin / cos are then evaluated twice: 2 x 0.08 ms of side-stream time against 0.25 ms of exposed tail).
    const bool box_split = c.use_side && !cfg->no_box && L > 2;
    hipEvent_t box_done_ev = nullptr;
    auto box_grad = [&](int l0, int n) -> int {
        const float* wg[MAXLAYERS]; const float* bg[MAXLAYERS]; float* dwg[MAXLAYERS]; float* dbg[MAXLAYERS];
        for (int l = 0; l < n; ++l) {
            wg[l] = params + o.enc[l0 + l].wg; bg[l] = params + o.enc[l0 + l].bg; dwg[l] = G + o.enc[l0 + l].wg; dbg[l] = G + o.enc[l0 + l].bg;
        }
        const float* ds = w.dscore + (int64_t)l0 * B * H * S * S;
        if (c.use_side) {
            // on the THIRD queue (in order behind the side stream's weight-gradient groups it delayed every later group by its 0.3 ms:
            // the side stream reached the end of the step 0.25 ms late); joined at the end of the backward
            hipEvent_t ready = c.side->take();
            box_done_ev = c.side->take();
            if (hipEventRecord(ready, c.s) != hipSuccess || hipStreamWaitEvent(c.side->s2, ready, 0) != hipSuccess) return ORTK_EINVAL;
            TRY(ortk_box_logbias_bwd(bt->boxes, wg, bg, cfg->box_trig ? dim_mat() : nullptr, ds, dwg, dbg, n, B, S, H, (ortk_stream)c.side->s2));
            if (hipEventRecord(box_done_ev, c.side->s2) != hipSuccess) return ORTK_EINVAL;
            return 0;
        }
        return ortk_box_logbias_bwd(bt->boxes, wg, bg, cfg->box_trig ? dim_mat() : nullptr, ds, dwg, dbg, n, B, S, H, stream);
    };
    dx = w.ga; dx2 = w.gb;
    TRY(ln_bwd(c, w.gy, w.enc[L - 1].xout, G, o.enc_na, o.enc_nb, w.st_mem, nullptr, dx, Me, gt_new(), eop(L - 1, 3)));
    if (bch_enc) {
        // cur = gradient of the FFN sublayer's input rows (xm) of the layer in hand; oth = the other fp32 buffer
        float* cur = dx2; float* oth = dx;
        void* dtA = gt_cur; void* dtB = gt_new(); void* ghb = gh_new();
        {
            const EncOff& e = o.enc[L - 1]; const EncBuf& b = w.enc[L - 1];
            ortk_bchain_args ca; std::memset(&ca, 0, sizeof(ca)); ca.drop_rows = c.drop_rows;
            ca.n_units = bs.units(bs.ex_top); ca.M = Me; ca.nin = 0; ca.dz0 = dtA;
            x_part(ca, b.h, ghb, b.xm, b.st2, e.n1a, e.n1b, dx, cur, dtB, c.sub(eop(L - 1, 1)), w.gdo);
            ca.drop_p = c.p_drop(); ca.eps = 1e-6f;
            TRY(c.before_write(dtB)); TRY(c.before_write(ghb));
            TRY(bchain_run(&ca, bs.stream_of(w.chain_pk_b, bs.ex_top), c.s));
        }
        for (int l = L - 1; l >= 0; --l) {
            const EncOff& e = o.enc[l]; const EncBuf& b = w.enc[l];
            const float* xin = l == 0 ? w.x0 : w.enc[l - 1].xout;
            void* const gql = gq_of(l);
            c.grp_on = grp_layers;
            TRY(wgrad_gemm(c, dtA, A, d, b.h, A, ff, G + e.w2, G + e.b2, Me, d, ff));
            TRY(wgrad_gemm(c, ghb, A, ff, b.y2, A, d, G + e.w1, G + e.b1, Me, ff, d));
            TRY(wgrad_gemm(c, dtB, A, d, b.o, A, d, G + e.wo, G + e.bo, Me, d, d));
            ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
            a.qkv_dtype = w.qdt_enc; a.q = (const float*)b.qkv; a.k = (const float*)off_elems(b.qkv, ame.k * d, w.qdt_enc);
            a.v = (const float*)off_elems(b.qkv, ame.v * d, w.qdt_enc); a.ldq = a.ldk = a.ldv = 3 * d;
            a.p = b.P; a.nkv = B; a.H = H; a.Lq = S; a.Lk = S; a.dk = dk; a.drop_p = c.p_drop(); a.drop_seed = c.sub(eop(l, 0));
            a.d_o = (const float*)w.gdo; a.lddo = d; a.dqkv_dtype = A;
            a.dq = gql; a.d_k = off_elems(gql, ame.gk * d, A); a.dv = off_elems(gql, ame.gv * d, A); a.lddq = a.lddk = a.lddv = 3 * d;
            a.dscore = cfg->no_box ? nullptr : w.dscore + (int64_t)l * B * H * S * S;
            TRY(c.before_write(gql));
            TRY(ortk_attention_bwd(&a, stream));
            if (l == 1 && box_split) TRY(box_grad(1, L - 1));
            TRY(wgrad_gemm(c, gql, A, 3 * d, b.y1, A, d, G + e.wqkv, G + e.bqkv, Me, 3 * d, d));
            TRY(flush_wgrads(c, l == 0));
            c.grp_on = false;
            ortk_bchain_args ca; std::memset(&ca, 0, sizeof(ca)); ca.drop_rows = c.drop_rows;
            ca.M = Me; ca.nin = 3; ca.ain = gql; ca.ld_ain = 3 * d;
            ca.xa = xin; ca.sta = b.st1; ca.ga = params + e.n0a; ca.dresa = cur; ca.dxa = oth; ca.daa = G + e.n0a; ca.dba = G + e.n0b;
            ca.drop_p = c.p_drop(); ca.eps = 1e-6f;
            if (l > 0) {
                const EncOff& en = o.enc[l - 1]; const EncBuf& bn = w.enc[l - 1];
                dtA = gt_new(); void* dtBn = gt_new(); ghb = gh_new();
                ca.n_units = bs.units(bs.ezx[l]);
                ca.dza = dtA; ca.seed_a = c.sub(eop(l - 1, 3)); ca.mask_a = 1;
                x_part(ca, bn.h, ghb, bn.xm, bn.st2, en.n1a, en.n1b, oth, cur, dtBn, c.sub(eop(l - 1, 1)), w.gdo);
                TRY(c.before_write(dtA)); TRY(c.before_write(dtBn)); TRY(c.before_write(ghb));
                TRY(bchain_run(&ca, bs.stream_of(w.chain_pk_b, bs.ezx[l]), c.s));
                dtB = dtBn;
            } else {
                ca.n_units = bs.units(bs.ez0);
                TRY(bchain_run(&ca, bs.stream_of(w.chain_pk_b, bs.ez0), c.s));
                dx = oth; dx2 = cur;           // the gradient of the embedded regions (att_embed's output)
            }
        }
    } else
    for (int l = L - 1; l >= 0; --l) {
        const EncOff& e = o.enc[l]; const EncBuf& b = w.enc[l];
        const float* xin = l == 0 ? w.x0 : w.enc[l - 1].xout;
        const void* dt; int dtt;
        void* const ghl = gh_of(l); void* const gql = gq_of(l);
        c.grp_on = grp_layers;
        TRY(drop_bwd(c, dx, gt_cur, Me * d, eop(l, 3), &dt, &dtt, true));
        TRY(wgrad_gemm(c, dt, dtt, d, b.h, A, ff, G + e.w2, G + e.b2, Me, d, ff));
        TRY(dgrad_gemm(c, dt, dtt, d, e.w2, ghl, A, ff, Me, d, ff, b.h, A, ff, inv_keep));
        TRY(wgrad_gemm(c, ghl, A, ff, b.y2, A, d, G + e.w1, G + e.b1, Me, ff, d));
        TRY(dgrad_ln_bwd(c, ghl, A, ff, e.w1, Me, ff, d, w.gy, b.xm, G, e.n1a, e.n1b, b.st2, dx, dx2, gt_new(), eop(l, 1)));
        TRY(drop_bwd(c, dx2, gt_cur, Me * d, eop(l, 1), &dt, &dtt, true));
        TRY(wgrad_gemm(c, dt, dtt, d, b.o, A, d, G + e.wo, G + e.bo, Me, d, d));
        // (layer 0 with the tail group: the three gradients whose operands exist go now; wqkv's and att_embed's, which the step ends with, form a small last group)
        if (l == 0 && tail_group) TRY(flush_wgrads(c));
        void* dOe = w.qdt_enc ? w.gdo : (void*)w.gy;
        TRY(dgrad_gemm(c, dt, dtt, d, e.wo, dOe, w.qdt_enc, d, Me, d, d));
        ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
        a.qkv_dtype = w.qdt_enc; a.q = (const float*)b.qkv; a.k = (const float*)off_elems(b.qkv, ame.k * d, w.qdt_enc);
        a.v = (const float*)off_elems(b.qkv, ame.v * d, w.qdt_enc); a.ldq = a.ldk = a.ldv = 3 * d;
        a.p = b.P; a.nkv = B; a.H = H; a.Lq = S; a.Lk = S; a.dk = dk; a.drop_p = c.p_drop(); a.drop_seed = c.sub(eop(l, 0));
        a.d_o = (const float*)dOe; a.lddo = d; a.dqkv_dtype = A;
        a.dq = gql; a.d_k = off_elems(gql, ame.gk * d, A); a.dv = off_elems(gql, ame.gv * d, A); a.lddq = a.lddk = a.lddv = 3 * d;
        a.dscore = cfg->no_box ? nullptr : w.dscore + (int64_t)l * B * H * S * S;
        TRY(c.before_write(gql));
        TRY(ortk_attention_bwd(&a, stream));
        if (l == 1 && box_split) TRY(box_grad(1, L - 1));     // layers 1 .. L-1: beside the last two encoder layers
        if (l == 0 && tail_group && !cfg->no_box) TRY(box_grad(0, box_split ? 1 : L));      // (third queue: done before the step's last launch needs the units)
        TRY(fold(ame, Me, gql));
        TRY(wgrad_gemm(c, gql, A, 3 * d, b.y1, A, d, G + e.wqkv, G + e.bqkv, Me, ame.n * d, d));
        if (l > 0 || !tail_group) TRY(flush_wgrads(c, l == 0));        // the layer's four weight gradients: one launch (layer 0: below)
        c.grp_on = false;
        TRY(dgrad_ln_bwd(c, gql, A, 3 * d, e.wqkv, Me, ame.n * d, d, w.gy, xin, G, e.n0a, e.n0b, b.st1, dx2, dx, gt_new(), l > 0 ? (int)eop(l - 1, 3) : -1));
    }
    // The tail.  With the grouped weight gradients and the bf16 feature copy: layer 0's last weight gradient was held back, att_embed's
    // joins the one that is still held back (wqkv's), and the two go as ONE launch on the side stream; the geometry-bias gradient of layer 0 went
    // to the third queue right behind layer 0's attention backward (its workgroups hold the LDS of every unit: beside the last launch they delayed it by their 63 us).  (Before: an urgent 4-gradient launch beside the last LayerNorm backward, which
    // took 116 instead of 24 us for it, then a 120-us att_embed launch of its own on the caller's stream.)
    if (!cfg->no_box && !tail_group) TRY(box_grad(0, box_split ? 1 : L));
    // att_embed: x0 = dropout(relu(.) * mask)  ->  d(pre-activation) = dx * [x0 > 0] / (1 - p_src)
    TRY(c.before_write(gt_new()));
    TRY(ortk_gate_apply(dx, w.x0, gt_cur, A, Me * d, c.p_src() > 0.f ? 1.f / (1.f - c.p_src()) : 1.f, stream));
    if (tail_group) {
        c.grp_on = true;
        TRY(wgrad_gemm(c, gt_cur, A, d, w.feats16, ORTK_BF16, cfg->feat, G + o.att_w, G + o.att_b, Me, d, cfg->feat));
        TRY(flush_wgrads(c, true));
        c.grp_on = false;
    } else {   // the last weight gradient stays on the caller's stream: the side stream is busy with the geometry-bias gradient
        const bool us = c.use_side;
        c.use_side = false;
        TRY(wgrad_gemm(c, gt_cur, A, d, bt->att_feats, ORTK_F32, cfg->feat, G + o.att_w, G + o.att_b, Me, d, cfg->feat));
        c.use_side = us;
    }
    TRY(c.wait_ev(box_done_ev));
    return c.join();       // every gradient is final in the caller's stream order
}

static int encode_impl(const ortk_config*, const float*, const float*, const float*, const float*, int32_t, int32_t, void*,
                       size_t, float*, ortk_stream);

extern "C" int ortk_encode(const ortk_config* cfg, const float* params, const float* att_feats, const float* boxes,
                           const float* att_masks, int32_t B, int32_t S, void* ws, size_t ws_bytes, float* memory_out,
                           ortk_stream stream) {
    if (int e = check_cfg(cfg)) return e;
    if (!params || !att_feats || (!boxes && !cfg->no_box) || !att_masks || !ws || !memory_out || B < 1 || S < 1 || S > 128) return ORTK_EINVAL;
    return encode_impl(cfg, params, att_feats, boxes, att_masks, B, S, ws, ws_bytes, memory_out, stream);
}

namespace ortk {
struct Blk { int64_t off; int N, K; };
// every weight matrix a forward GEMM reads as one (N, K) block (packed projections count once)
static std::vector<Blk> linear_blocks(const ortk_config* cfg, const Offsets& o) {
    const int d = cfg->d_model, ff = cfg->d_ff, L = cfg->n_layers;
    const int ne = att_mode(cfg->share_att_enc).n, nd = att_mode(cfg->share_att_dec).n;
    std::vector<Blk> v;
    v.push_back({o.att_w, d, cfg->feat});
    for (int l = 0; l < L; ++l) {
        const EncOff& e = o.enc[l];
        v.push_back({e.wqkv, ne * d, d}); v.push_back({e.wo, d, d}); v.push_back({e.w1, ff, d}); v.push_back({e.w2, d, ff});
    }
    for (int l = 0; l < L; ++l) {
        const DecOff& e = o.dec[l];
        v.push_back({e.wqkv, nd * d, d}); v.push_back({e.wo, d, d}); v.push_back({e.cow, d, d});
        if (cfg->share_att_dec != 2) v.push_back({e.cqw, d, d});      // "qk": part of the packed K|V block below
        v.push_back({e.w1, ff, d}); v.push_back({e.w2, d, ff});
    }
    v.push_back({o.ckv_w, (int)(o.ckv_slots * o.cw), d});
    v.push_back({o.gen_w, (int)ortk_align(cfg->vocab, 128), d});
    return v;
}
// bf16 transposed copies of the blocks the data-gradient GEMMs read (everything but the region embedding, whose input
// gradient is never needed); layers that share weights appear once
static int make_w16t(const ortk_config* cfg, const Offsets& o, const float* params, void* w16t, ortk_stream stream) {
    if (!cfg->precision || !w16t) return 0;
    const std::vector<Blk> v = linear_blocks(cfg, o);
    WBlockTable t; t.n = 0; t.tiles = 0;
    for (size_t i = 1; i < v.size(); ++i) {
        bool dup = false;
        for (int j = 0; j < t.n; ++j) dup |= t.b[j].off == (int32_t)v[i].off;
        if (dup) continue;
        if (t.n >= MAX_WBLOCKS || v[i].off > 0x7FFFFFFF) return ORTK_EINVAL;
        t.b[t.n] = WBlock{(int32_t)v[i].off, v[i].N, v[i].K, t.tiles};
        t.tiles += (int32_t)(ortk_cdiv(v[i].N, 64) * ortk_cdiv(v[i].K, 64));
        ++t.n;
    }
    return cast_bf16_transposed(params, w16t, t, ortk_s(stream));
}
}  // namespace ortk

extern "C" int ortk_linear_block(const ortk_config* cfg, int32_t i, int64_t* offset, int32_t* N, int32_t* K) {
    if (int e = check_cfg(cfg)) return e;
    Offsets o; build_layout(*cfg, o, nullptr);
    const std::vector<Blk> v = linear_blocks(cfg, o);
    if (i < 0) return (int)v.size();
    if (i >= (int)v.size() || !offset || !N || !K) return ORTK_EINVAL;
    *offset = v[i].off; *N = v[i].N; *K = v[i].K;
    return 0;
}

// ================================================================================================ decoding
namespace ortk {
struct DecodeWS {
    int64_t ldv; int adt;
    void* w16;
    float *x0, *logbias; void* mem /*A*/; float* st; void* ckv /*ckvdt*/;
    EncPtrs enc;                       // one set of encoder buffers, reused by every layer
    float *xa, *xb; void* y /*A*/; float* qkv; void* o /*A*/; void* q /*ckvdt when xq16*/; void* h /*A*/; float* logits;
    float* gstats = nullptr;           // soft-max partials of the logit rows (rows, ldv / 64, 2): mixed precision
    float* gsamp = nullptr;            // Gumbel-max candidates of the logit rows (rows, ldv / 64, 4): sampling decodes, mixed precision
    int ckvdt, xq16;                   // projected memory dtype; 1 = cross-attention through the bf16-operand kernels (bf16 query too)
    void *cache_k[MAXLAYERS], *cache_v[MAXLAYERS]; int kvdt;   // K/V caches + projected memory: bf16 in mixed precision when the
                                                               // decode attention kernels take them (kv16), fp32 otherwise
    int64_t* it; int32_t *unfinished, *last_step;
    int32_t *bseq[2], *kvidx[2], *done_seq, *done_len, *done_cnt; float *blp[2], *cum, *done_lp; double* done_p;
    void* wpk = nullptr;               // decoder weights in the stack kernel's streaming order (stack path, dense stream)
    int32_t* progress = nullptr;       // pace-maker counters of the stack kernel's L2 prefetchers
    SStackBufs ss{};                   // the sparse stream and its tables (stack path, sparse stream)
    int tp = 0;                        // column-split stack kernel: workgroups per group (0 = off), its weight image,
    void* tp_wpk = nullptr; char* tp_xbuf = nullptr; int32_t* tp_flag = nullptr;      // exchange tiles and counters
    void* chain_pk = nullptr;          // encoder chains' weight units in streaming order (chain_layout; mixed precision)
    int32_t* status = nullptr;         // [64] decode status word (ortk_decode_status): ALWAYS the first bytes of the workspace
    void* ckv_g = nullptr;             // train-mode decode with greedy rows: projection of the EVAL-mode encoder memory
    size_t bytes;
};

// The one-launch-per-position decoder stack (ortk_decstack.hip) serves the reference's configuration and the ACORT widths
// with d_model 512: mixed precision, 8 heads of 64, d_ff a multiple of 512, no projection sharing inside the decoder's
// attention modules.  Everything else — and every call that brings a sparse plan — runs the unfused executor below.
// ORTK_DEC_STACK=0 switches it off (A/B measurements).
static bool stack_ok(const ortk_config& c, int64_t rows, int32_t flags) {
    // Every workgroup of the stack kernel streams ALL decoder weights (42 MB per position at ~85 GB/s per CU: >= 0.5 ms per
    // position however few rows there are), the unfused GEMMs read each weight once per launch: measured crossover at 320
    // images x 5 beams (16.1 vs 16.0 ms; 512 images 17.0 vs 21.3, 50 images 14.1 vs 11.1, the SCST rollout of 256 x 6 rows
    // 29.9 vs 28.3 ms per step).  The SPARSE stream (ORTK_DEC_SPARSE_STREAM) is ~9x shorter and has no such floor.
    // ortk_decode_opts.exec_flags overrides the size rule (parity tests and A/B measurements run both executors).
    if (flags & ORTK_DEC_UNFUSED) return false;
    // ORTK_DEC_SPLIT_SMALL: the column-split form serves the small decodes (groups of 8 workgroups up to 2 048 rows, of 4 up to 4 096)
    const bool small_split = (flags & ORTK_DEC_SPLIT_SMALL) && !(flags & ORTK_DEC_SPARSE_STREAM) && stack_tp_degree(rows) >= 4;
    if (!(flags & (ORTK_DEC_STACK | ORTK_DEC_SPARSE_STREAM | ORTK_DEC_STACK_SPLIT)) && !small_split && rows < 1600) return false;
    return c.precision == 1 && c.d_model == 512 && c.n_heads == 8 && c.d_ff % 512 == 0 && c.d_ff / 512 <= 8 &&
           c.share_att_dec == 0 && c.n_layers <= STACK_MAXL && c.seq_len <= 64;     // (seq_len: one lane per cached key)
}

// workgroups per group of the column-split stack kernel for this decode (0: the plain kernel)
static int split_degree(bool dense_stack, int32_t flags, int64_t rows) {
    if (!dense_stack) return 0;
    const int G = stack_tp_degree(rows);
    if (flags & ORTK_DEC_STACK_SPLIT) return G;                                   // whenever its groups fit the chip
    return ((flags & ORTK_DEC_SPLIT_SMALL) && !(flags & ORTK_DEC_STACK) && G >= 4) ? G : 0;     // the small decodes only (8 or 4 per group)
}
static void carve_decode(const ortk_config& c, int B, int S, int K, bool beam, void* base, DecodeWS& w, bool stack = false, bool sstream = false,
                         bool train = false, int tp = 0, bool greedy_rows = false) {
    const int64_t d = c.d_model, ff = c.d_ff, H = c.n_heads, L = c.n_layers, T = c.seq_len;
    // bf16 K / V storage: only when both decode attention kernels that understand it will be the ones dispatched
    w.kvdt = (c.precision && H == 8 && d == 512 && S > 8 && S <= 48 && T <= 32 && K <= 16) ? ORTK_BF16 : ORTK_F32;
    // more than 48 regions (ragged 10-100 bottom-up features): the cross-attention of a step runs in the bf16-operand block
    // kernel on bf16 projected memory and a bf16 query; the self-attention caches stay as the row kernel wants them
    w.xq16 = (c.precision && w.kvdt == ORTK_F32 && S > 48 && attn16_shape_ok(K, S, (int)(d / H))) ? 1 : 0;
    w.ckvdt = w.xq16 ? ORTK_BF16 : w.kvdt;
    if (stack) { w.kvdt = w.ckvdt = ORTK_BF16; w.xq16 = 0; }      // the stack kernel reads bf16 caches at every S
    if (train && !stack) { w.kvdt = w.ckvdt = ORTK_F32; w.xq16 = 0; }       // train-mode sampling, unfused: the generic attention kernel (fp32 rows)
    const size_t kves = ortk_esize(w.kvdt);
    const int64_t Me = (int64_t)B * S, rows = (int64_t)B * K;
    w.ldv = ortk_align(c.vocab, 128);
    w.adt = c.precision ? ORTK_BF16 : ORTK_F32;
    const size_t es = ortk_esize(w.adt);
    Bump b{reinterpret_cast<char*>(base), 0};
    auto act = [&](int64_t n) { return b.take_bytes((size_t)n * es); };
    Offsets o; build_layout(c, o, nullptr);
    w.status = b.take<int32_t>(64);
    w.w16 = c.precision ? b.take_bytes((size_t)o.total * 2) : nullptr;
    w.x0 = b.take<float>(Me * d); w.logbias = b.take<float>(L * B * H * S * S);
    w.enc.y1 = act(Me * d); w.enc.qkv = b.take<float>(Me * 3 * d); w.enc.P = nullptr; w.enc.o = act(Me * d);
    w.enc.xm = b.take<float>(Me * d); w.enc.y2 = act(Me * d); w.enc.h = act(Me * ff);
    w.enc.xout = w.x0; w.enc.st1 = b.take<float>(Me * 2); w.enc.st2 = w.enc.st1;
    w.mem = b.take_bytes((size_t)Me * d * 4); w.st = b.take<float>(std::max(Me, rows) * 2);
    w.ckv = b.take_bytes((size_t)(Me * L * 2 * d) * ortk_esize(w.ckvdt));
    if (train && stack && greedy_rows) w.ckv_g = b.take_bytes((size_t)(Me * L * 2 * d) * ortk_esize(w.ckvdt));
    w.xa = b.take<float>(rows * d); w.xb = b.take<float>(rows * d); w.y = act(rows * d);
    w.qkv = b.take<float>(rows * 3 * d); w.o = act(rows * d); w.q = b.take<float>(rows * d);      // (fp32-sized; holds bf16 when xq16)
    w.h = act(rows * ff); w.logits = b.take<float>(rows * w.ldv);
    if (c.precision) w.gstats = b.take<float>(rows * (w.ldv / 64) * 2);
    if (!beam && c.precision) w.gsamp = b.take<float>(rows * (w.ldv / 64) * 4);
    for (int l = 0; l < L; ++l) { w.cache_k[l] = b.take_bytes((size_t)(rows * T * d) * kves); w.cache_v[l] = b.take_bytes((size_t)(rows * T * d) * kves); }
    w.it = b.take<int64_t>(rows);
    {
        ChainSet cs; chain_layout(c, o, true, false, cs, 0, 0, true);
        if (cs.bytes) w.chain_pk = b.take_bytes(cs.bytes);
    }
    if (stack && !sstream && tp > 0) {
        w.tp = tp;
        w.tp_wpk = b.take_bytes(stack_tp_packed_bytes((int)L, (int)(ff / 512), tp));
        w.tp_xbuf = reinterpret_cast<char*>(b.take_bytes(stack_tp_xbuf_bytes(rows, (int)(ff / 512))));
        w.tp_flag = b.take<int32_t>((int64_t)stack_tp_groups(rows) * TP_FLAG_STRIDE);
        w.progress = b.take<int32_t>(16);
    } else if (stack && !sstream) { w.wpk = b.take_bytes(stack_packed_bytes((int)L, (int)(ff / 512))); w.progress = b.take<int32_t>(16); }
    if (stack && sstream) {
        const size_t nb = sstack_bytes((int)L, (int)(ff / 512), nullptr, nullptr);
        void* p = b.take_bytes(nb);
        sstack_bytes((int)L, (int)(ff / 512), &w.ss, p);
    }
    if (beam) {
        for (int i = 0; i < 2; ++i) { w.bseq[i] = b.take<int32_t>(rows * T); w.blp[i] = b.take<float>(rows * T); w.kvidx[i] = b.take<int32_t>(rows * (T + 1)); }
        w.cum = b.take<float>(rows);
        w.done_seq = b.take<int32_t>(rows * T * T); w.done_lp = b.take<float>(rows * T * T);
        w.done_p = b.take<double>(rows * T); w.done_len = b.take<int32_t>(rows * T); w.done_cnt = b.take<int32_t>(B);
    } else {
        w.unfinished = b.take<int32_t>(rows); w.last_step = b.take<int32_t>(4);
    }
    w.bytes = (b.off + 255) & ~(size_t)255;
}
}  // namespace ortk

static int encode_impl(const ortk_config* cfg, const float* params, const float* att_feats, const float* boxes,
                       const float* att_masks, int32_t B, int32_t S, void* ws, size_t ws_bytes, float* memory_out,
                       ortk_stream stream) {
    Offsets o; build_layout(*cfg, o, nullptr);
    DecodeWS w; carve_decode(*cfg, B, S, 1, false, ws, w);
    if (w.bytes > ws_bytes) return ORTK_ENOSPC;
    TRY(make_w16(cfg, o, params, w.w16, stream));
    Ctx c{cfg, ortk_s(stream), cfg->precision, 0, false, params, w.w16, w.adt};
    EncPtrs ep[MAXLAYERS];
    for (int l = 0; l < cfg->n_layers; ++l) ep[l] = w.enc;
    return encoder_forward(c, o, att_feats, boxes, att_masks, B, S, w.x0, w.logbias, ep, memory_out, ORTK_F32, w.st);      // (fp32 memory out: no chains)
}

static int decode_K(const ortk_decode_opts* o) {
    if (o->num_random_sample > 0) return o->beam_size < 1 ? o->num_random_sample + (o->with_greedy ? 1 : 0) : -1;
    return o->beam_size >= 1 ? o->beam_size : -1;
}

// Which executor serves a decode call.  Train-mode rows (ortk_decode_opts.train) run on the column-split stack kernel with at least 4
// workgroups per group (its dropout sites, ortk_decstack.hip) or on the unfused executor; eval-mode (greedy) rows beside them
// (`with_greedy`) only on the former.  ok = false: the option combination is not served (ORTK_EINVAL).
struct DecodePlan { bool ok, stack, sstream; int split; bool gather = false; };
static DecodePlan plan_decode(const ortk_config& cfg, int B, int K, const ortk_decode_opts* o) {
    DecodePlan p{true, false, false, 0};
    const int64_t rows = (int64_t)B * K;
    p.stack = stack_ok(cfg, rows, o->exec_flags) && !o->sparse;
    p.sstream = p.stack && (o->exec_flags & ORTK_DEC_SPARSE_STREAM);
    p.gather = p.sstream && (o->exec_flags & ORTK_DEC_SPARSE_GATHER);
    p.split = split_degree(p.stack && !p.sstream, o->exec_flags, rows);
    if (o->train) {
        if (o->num_random_sample <= 0 || o->sparse) { p.ok = false; return p; }      // multinomial rollouts, dense products
        if (!(p.stack && !p.sstream && p.split >= 4)) { p.stack = p.sstream = false; p.split = 0; }
        if (o->with_greedy && !p.stack) p.ok = false;
    }
    return p;
}

extern "C" size_t ortk_decode_workspace_bytes(const ortk_config* cfg, int32_t B, int32_t S, const ortk_decode_opts* o) {
    if (check_cfg(cfg) || !o || B < 1 || S < 1) return 0;
    const int K = decode_K(o);
    if (K < 1) return 0;
    const DecodePlan pl = plan_decode(*cfg, B, K, o);
    if (!pl.ok) return 0;
    DecodeWS w; carve_decode(*cfg, B, S, K, o->num_random_sample <= 0 && o->beam_size > 1, nullptr, w, pl.stack, pl.sstream, o->train != 0,
                             pl.split, o->with_greedy != 0);
    return w.bytes;
}

// Status of the decode that last ran on workspace `ws` (host synchronisation: waits for `stream`): 0, or ORTK_EEXCHANGE when a
// workgroup of the column-split stack kernel never met its exchange group (the launch was not fully resident: another kernel
// held compute units) — the outputs of that decode are all-pad captions with NaN log-probs.
extern "C" int ortk_decode_status(const void* ws, ortk_stream stream) {
    if (!ws) return ORTK_EINVAL;
    int32_t h = 0;
    if (hipMemcpyAsync(&h, ws, sizeof(h), hipMemcpyDeviceToHost, ortk_s(stream)) != hipSuccess) return ORTK_EINVAL;
    if (hipStreamSynchronize(ortk_s(stream)) != hipSuccess) return ORTK_EINVAL;
    return h == 0 ? 0 : ORTK_EEXCHANGE;
}

// One position of the cached-attention decoder for `rows` rows (transformer.py:172-210 with the caches of :240-273):
// embeds the tokens at position t, appends this position's K/V to the self-attention caches ((row, tmax, d) per layer;
// the keys of a row are its own slots 0..t unless the beam ancestry table kvidx is given), attends to the projected
// memory of the row's group (`per_group` consecutive rows share one), and leaves the generator logits in b.logits.
struct StepBufs {
    const int64_t* it; float *xa, *xb; void* y; float* qkv; void* o; void* q; void* h; float* logits; float* st; int64_t ldv;
    const void* ckv; const float* att_masks;
    void* cache_k[MAXLAYERS]; void* cache_v[MAXLAYERS];
    int kvdt;      // element type of cache_k / cache_v
    int ckvdt = ORTK_F32, xq16 = 0;   // element type of ckv; 1 = bf16 cross-attention query + bf16-operand kernel
    float* gstats = nullptr; int stat_ncols = 0;    // soft-max partials of the logit rows wanted from the generator GEMM
    float* gsamp = nullptr; const SampleState* samp = nullptr; int samp_t = 0; bool samp_fast = false;     // ... and its sampling candidates (sample_combine)
};
// generator logits of a decode step (w.y: the final LayerNorm's output); with w.gstats also their soft-max partials
static int gen_gemm(const Ctx& c, const Offsets& o, const StepBufs& w, int ydt, int64_t rows) {
    const int d = c.cfg->d_model;
    if (!w.gstats) return fwd_gemm(c, w.y, ydt, d, o.gen_w, c.P + o.gen_b, w.logits, ORTK_F32, w.ldv, rows, (int)w.ldv, d);
    ortk_gemm_args a; std::memset(&a, 0, sizeof(a));
    a.A = w.y; a.a_dtype = ydt; a.lda = d; a.B = c.W(o.gen_w); a.b_dtype = c.wdt(); a.ldb = d; a.C = w.logits; a.c_dtype = ORTK_F32; a.ldc = w.ldv;
    a.M = (int)rows; a.N = (int)w.ldv; a.K = d; a.bias = c.P + o.gen_b; a.precision = c.prec;
    a.tile_stats = w.gstats; a.stat_ncols = w.stat_ncols;
    if (w.gsamp && w.samp) {
        const SampleState& ss = *w.samp;
        a.tile_samp = w.gsamp; a.samp_seq = ss.decoding_constraint ? ss.seq : nullptr; a.samp_seed = ss.seed; a.samp_row_offset = ss.row_offset;
        a.samp_L = ss.L; a.samp_t = w.samp_t; a.samp_greedy_stride = ss.greedy_stride; a.samp_sample = ss.sample; a.samp_fast = w.samp_fast;
        a.samp_no_store = 1; a.samp_inv_temperature = 1.f / ss.temperature;
    }
    return ortk_gemm(&a, (ortk_stream)c.s);
}
static int decoder_step(const Ctx& c, const Offsets& o, const StepBufs& w, int64_t rows, int groups, int per_group, int /*row_mult*/,
                        int S, int T, int t, const int32_t* kvidx) {
    const ortk_config* cfg = c.cfg;
    const float* P = c.P;
    ortk_stream stream = (ortk_stream)c.s;
    const int d = cfg->d_model, ff = cfg->d_ff, H = cfg->n_heads, L = cfg->n_layers, dk = d / H, A = c.adt;
    const float* att_masks = w.att_masks;
    const int B = groups, per_img = per_group;
    const AttMode am = att_mode(cfg->share_att_dec);
        // Train-mode sampling (c.train: utils/training.py:224-237 samples after model.train()): every dropout of the step draws
        // what the teacher-forced pass over [BOS, sample] draws at (row, position t) — same site keys, the element index of the
        // (rows x T, N) teacher-forced tensors (cd.drop_rs / drop_r0; ortk_attn_args.drop_tf_*).  pd = 0 in eval mode.
        Ctx cd = c; cd.drop_rs = T; cd.drop_r0 = t;
        const float pd = c.p_drop();
        TRY(embed_fwd_rows(w.it, 1, P + o.lut, P + o.pe, w.xa, nullptr, rows, nullptr, 1, t, d, cfg->pad_id, pd, c.sub(OP_EMB), c.s, T, t));
        float* x = w.xa; float* xn = w.xb;
        for (int l = 0; l < L; ++l) {
            const DecOff& e = o.dec[l];
            TRY(ln_fwd(c, x, e.n0a, e.n0b, w.y, A, w.st, rows));
            TRY(fwd_gemm(c, w.y, A, d, e.wqkv, P + e.bqkv, w.qkv, ORTK_F32, 3 * d, rows, am.n * d, d));
            ortk_attn_args a; std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
            a.k_new = w.qkv + am.k * d; a.v_new = w.qkv + am.v * d; a.ld_new = 3 * d;    // this position's K / V join the cache inside the kernel
            a.q = w.qkv; a.ldq = 3 * d; a.k = reinterpret_cast<const float*>(w.cache_k[l]); a.v = reinterpret_cast<const float*>(w.cache_v[l]);
            a.kv_dtype = w.kvdt; a.ldk = a.ldv = d; a.o = w.o; a.o_dtype = A; a.ldo = d;
            a.nkv = (int)rows; a.H = H; a.Lq = 1; a.Lk = t + 1; a.dk = dk;
            if (kvidx) a.kv_index = kvidx; else a.kv_group_stride = T;
            if (pd > 0.f) { a.drop_p = pd; a.drop_seed = c.sub(dop(l, 0)); a.drop_tf_T = T; a.drop_tf_t = t; a.drop_tf_lk = T; }
            TRY(ortk_attention_fwd(&a, stream));
            TRY(fwd_gemm(cd, w.o, A, d, e.wo, P + e.bo, xn, ORTK_F32, d, rows, d, d, false, pd, c.sub(dop(l, 1)), x, d));
            std::swap(x, xn);
            TRY(ln_fwd(c, x, e.n1a, e.n1b, w.y, A, w.st, rows));
            TRY(fwd_gemm(c, w.y, A, d, e.cqw, P + e.cqb, w.q, w.xq16 ? ORTK_BF16 : ORTK_F32, d, rows, d, d));
            std::memset(&a, 0, sizeof(a)); a.precision = c.prec;
            a.q = (const float*)w.q; a.ldq = d; a.k = reinterpret_cast<const float*>(off_elems(w.ckv, o.ckv_slot[l] * o.cw, w.ckvdt));
            a.v = reinterpret_cast<const float*>(off_elems(w.ckv, o.ckv_slot[l] * o.cw + o.cv, w.ckvdt));
            if (w.xq16) a.qkv_dtype = 1; else a.kv_dtype = w.ckvdt;
            a.ldk = a.ldv = o.ckv_slots * o.cw;
            a.o = w.o; a.o_dtype = A; a.ldo = d; a.kmask = att_masks; a.nkv = B; a.H = H; a.Lq = per_img; a.Lk = S; a.dk = dk;
            if (pd > 0.f) { a.drop_p = pd; a.drop_seed = c.sub(dop(l, 2)); a.drop_tf_T = T; a.drop_tf_t = t; a.drop_tf_lk = S; }
            TRY(ortk_attention_fwd(&a, stream));
            TRY(fwd_gemm(cd, w.o, A, d, e.cow, P + e.cob, xn, ORTK_F32, d, rows, d, d, false, pd, c.sub(dop(l, 3)), x, d));
            std::swap(x, xn);
            TRY(ln_fwd(c, x, e.n2a, e.n2b, w.y, A, w.st, rows));
            TRY(fwd_gemm(cd, w.y, A, d, e.w1, P + e.b1, w.h, A, ff, rows, ff, d, true, pd, c.sub(dop(l, 4))));
            TRY(fwd_gemm(cd, w.h, A, ff, e.w2, P + e.b2, xn, ORTK_F32, d, rows, d, ff, false, pd, c.sub(dop(l, 5)), x, d));
            std::swap(x, xn);
        }
        TRY(ln_fwd(c, x, o.dec_na, o.dec_nb, w.y, A, w.st, rows));
        TRY(gen_gemm(c, o, w, A, rows));
    return 0;
}

// The same position through the one-launch decoder stack: embed, stack kernel, generator.
struct SplitBufs { int G; const void* wpk; char* xbuf; int32_t* flag; int groups; int32_t* status; };      // column-split form (G = 0: off)
static int decoder_stack_step(const Ctx& c, const Offsets& o, const StepBufs& w, const void* wpk, const SStackBufs* ss, int32_t* progress, int32_t flags,
                              int64_t rows, int per_img, int S, int T, int t, const int32_t* kvidx, const SplitBufs* sp = nullptr,
                              int greedy_stride = 0, const void* ckv_g = nullptr, int uniq_slot = 0) {
    const ortk_config* cfg = c.cfg;
    const float* P = c.P;
    const int d = cfg->d_model;
    // c.train: every dropout of the position draws what the teacher-forced pass over [BOS, sample] draws at (row, t); rows with
    // row % greedy_stride == 0 stay in eval mode (the greedy baseline of the same launch)
    const float pd = c.p_drop();
    // (eval mode on the plain kernel: the embedding is the kernel's first load — one launch less per position)
    const bool fold_embed = pd == 0.f && !(sp && sp->G);
    if (!fold_embed) TRY(embed_fwd_rows(w.it, 1, P + o.lut, P + o.pe, w.xa, nullptr, rows, nullptr, 1, t, d, cfg->pad_id, pd, c.sub(OP_EMB), c.s, T, t, greedy_stride));
    StackArgs a; std::memset(&a, 0, sizeof(a));
    if (fold_embed) { a.tok = w.it; a.lut = P + o.lut; a.pe_t = P + o.pe + (int64_t)t * d; a.emb_scale = (float)std::sqrt((double)d); }
    for (int l = 0; l < cfg->n_layers; ++l) {
        const DecOff& e = o.dec[l];
        StackLayer& y = a.layer[l];
        y.n0a = P + e.n0a; y.n0b = P + e.n0b; y.bqkv = P + e.bqkv; y.bo = P + e.bo; y.n1a = P + e.n1a; y.n1b = P + e.n1b;
        y.cqb = P + e.cqb; y.cob = P + e.cob; y.n2a = P + e.n2a; y.n2b = P + e.n2b; y.b1 = P + e.b1; y.b2 = P + e.b2;
        y.ck = reinterpret_cast<__bf16*>(w.cache_k[l]); y.cv = reinterpret_cast<__bf16*>(w.cache_v[l]);
        y.xk = reinterpret_cast<const __bf16*>(w.ckv) + o.ckv_slot[l] * o.cw;
        y.xv = y.xk + o.cv;
        if (ckv_g) { y.xkg = reinterpret_cast<const __bf16*>(ckv_g) + o.ckv_slot[l] * o.cw; y.xvg = y.xkg + o.cv; }
        for (int k = 0; k < 6; ++k) a.drop_seed[l][k] = c.sub(dop(l, k));
    }
    a.drop_p = pd; a.greedy_stride = pd > 0.f ? greedy_stride : 0;
    a.wpk = reinterpret_cast<const uint4*>(wpk); a.progress = progress; a.x_io = w.xa; a.y_out = reinterpret_cast<__bf16*>(w.y);
    if (ss) { a.sstream = ss->stream; a.snst = ss->nst; a.sstart = ss->start; a.gather = (flags & ORTK_DEC_SPARSE_GATHER) ? 1 : 0; }
    a.rb = (flags & ORTK_DEC_STACK_RB20) ? 20 : 32;
    a.fa = P + o.dec_na; a.fb = P + o.dec_nb; a.att_masks = w.att_masks; a.kvidx = kvidx; a.ldx = o.ckv_slots * o.cw;
    a.rows = (int)rows; a.per_img = per_img; a.S = S; a.T = T; a.t = t; a.L = cfg->n_layers; a.NC = cfg->d_ff / 512; a.eps = 1e-6f;
    a.debug = (flags >> 8) & 0xFF;       // phase-skipping measurement switches and the exchange tests (StackArgs.debug)
    a.uniq_slot = uniq_slot;
    if (sp && sp->G) {
        a.tp = sp->G; a.tp_wpk = reinterpret_cast<const uint4*>(sp->wpk); a.tp_xbuf = sp->xbuf; a.tp_flag = sp->flag; a.tp_groups = sp->groups;
        a.tp_status = sp->status;
        a.tp_launch = t;                 // one launch per position: the exchange counters keep running through the decode
        a.tp_xtile = (int64_t)stack_tp_xtile_bytes(cfg->d_ff / 512);
    }
    TRY(stack_step(a, c.s));
    return gen_gemm(c, o, w, ORTK_BF16, rows);
}

extern "C" int ortk_decode(const ortk_config* cfg, const float* params, const float* att_feats, const float* boxes,
                           const float* att_masks, int32_t B, int32_t S, const ortk_decode_opts* op, void* ws, size_t ws_bytes,
                           int64_t* seq_out, float* logprob_out, float* score_out, ortk_stream stream) {
    if (int e = check_cfg(cfg)) return e;
    if (!params || !op || !att_masks || !ws || !seq_out || !logprob_out) return ORTK_EINVAL;
    if (!op->memory && (!att_feats || (!boxes && !cfg->no_box))) return ORTK_EINVAL;
    // (op->memory with op->train: the caller's memory must be the TRAIN-mode encoder output under drop_seed — ortk_forward_phase(train = 1,
    // seed = drop_seed, phase 1) computes exactly what this call's own train-mode encoder pass would)
    if (B < 1 || S < 1 || S > 128) return ORTK_EINVAL;
    const int K = decode_K(op);
    if (K < 1) return ORTK_EINVAL;   // the reference asserts the same option combinations (transformer.py:509,514)
    const bool beam = op->num_random_sample <= 0 && op->beam_size > 1;
    if (beam && (K > 8 || K > cfg->vocab)) return ORTK_EINVAL;
    if (op->temperature <= 0.f) return ORTK_EINVAL;
    Offsets o; build_layout(*cfg, o, nullptr);
    // train-mode sampling (dropout on while the captions are drawn): multinomial rollouts only; with the greedy baseline as eval-mode
    // rows of the same launches when the column-split stack kernel serves the call (plan_decode)
    const DecodePlan pl = plan_decode(*cfg, B, K, op);
    if (!pl.ok) return ORTK_EINVAL;
    const bool stack = pl.stack, sstream = pl.sstream;
    const int split = pl.split;
    const bool greedy_rows = op->train && op->with_greedy;         // (=> stack)
    if (greedy_rows && (!att_feats || (!boxes && !cfg->no_box))) return ORTK_EINVAL;      // their eval-mode encoder pass runs here
    DecodeWS w; carve_decode(*cfg, B, S, K, beam, ws, w, stack, sstream, op->train != 0, split, greedy_rows);
    if (w.bytes > ws_bytes) return ORTK_ENOSPC;
    hipStream_t s = ortk_s(stream);
    TRY(fill_i32(w.status, 64, 0, s));
    TRY(make_w16(cfg, o, params, w.w16, stream));
    // (the bf16 weight copy exists from here on in stream order: what the side stream's builders wait for)
    SideStream* const side0 = (w.adt == ORTK_BF16 && !ortk_prof_serial()) ? side_for(s) : nullptr;
    hipEvent_t w16_done = nullptr;
    if (side0 && stack && sstream) {
        w16_done = side0->take();
        if (hipEventRecord(w16_done, s) != hipSuccess) return ORTK_EINVAL;
    }
    StackPack tp; tp.L = cfg->n_layers; tp.NC = cfg->d_ff / 512;
    for (int l = 0; l < cfg->n_layers; ++l) {
        const DecOff& e = o.dec[l];
        const int64_t offs[6] = {e.wqkv, e.wo, e.cqw, e.cow, e.w1, e.w2};
        for (int i = 0; i < 6; ++i) tp.off[l][i] = offs[i];
    }
    if (stack) {
        if (sstream) { /* the sparse weight stream is built beside the encoder pass, below */ }
        else if (w.tp) {
            TRY(stack_tp_pack(w.w16, w.tp_wpk, tp, w.tp, s));
            TRY(fill_i32(w.tp_flag, (int64_t)stack_tp_groups((int64_t)B * K) * TP_FLAG_STRIDE, 0, s));
            TRY(fill_i32(w.progress, 16, 0, s));
        } else {
            TRY(stack_pack(w.w16, w.wpk, tp, s));
            TRY(fill_i32(w.progress, 16, 0, s));
        }
    }
    Ctx c{cfg, s, cfg->precision, op->train ? op->drop_seed : 0, op->train != 0, params, w.w16, w.adt};
    if (op->sparse) {
        TRY(ortk_sparse_build(op->sparse, cfg->precision ? (const void*)w.w16 : (const void*)params, cfg->precision ? ORTK_BF16 : ORTK_F32, stream));
        c.ell_f = op->sparse;
    }
    const float* P = params;
    const int d = cfg->d_model, L = cfg->n_layers, V = cfg->vocab, T = cfg->seq_len, A = w.adt;
    const int64_t Me = (int64_t)B * S;
    // No activations are kept: every encoder layer reuses one buffer set, and the residual stream is updated in
    // place (x is dead once xm = x + attn(...) exists, so the FFN sublayer writes its output back over x).
    EncPtrs ep[MAXLAYERS];
    for (int l = 0; l < L; ++l) ep[l] = w.enc;
    // the side stream carries the geometry bias of the encoder (0.4 ms at 1 024 images, VALU-bound) beside att_embed and the
    // first projection; nothing else of a decode runs there
    c.side = (c.adt == ORTK_BF16 && !ortk_prof_serial()) ? side_for(c.s) : nullptr;
    c.use_side = c.side != nullptr;
    // the encoder's row-wise operators as rows-stationary chains (ortk_chain.hip): bf16 Q|K|V (the buffer is fp32-sized) and memory
    ChainSet ecs; chain_layout(*cfg, o, true, false, ecs, Me, 0);
    const bool enc_bf16_attn = cfg->precision && attn16_shape_ok(S, S, d / cfg->n_heads);
    if (ecs.on && (!w.chain_pk || op->sparse || !enc_bf16_attn || (op->memory && !greedy_rows))) ecs.on = false;
    if (ecs.on) TRY(chain_pack_all(w.w16, w.chain_pk, ecs.t, s));
    const int enc_qdt = ecs.on ? ORTK_BF16 : ORTK_F32;
    // ortk_decode_opts.memory: the encoder output of these images already exists (the training forward's, ortk_forward_phase 1)
    if (greedy_rows) {
        // the greedy rows attend to the EVAL-mode encoder memory (utils/training.py:216-222 decodes the baseline under model.eval()):
        // its pass and projection first, then the buffers are free for the train-mode pass
        Ctx ce = c; ce.train = false; ce.seed = 0;
        TRY(encoder_forward(ce, o, att_feats, boxes, att_masks, B, S, w.x0, w.logbias, ep, w.mem, A, w.st, enc_qdt, nullptr, &ecs, w.chain_pk, false));
        ce.use_side = false;
        TRY(fwd_gemm(ce, w.mem, A, d, o.ckv_w, P + o.ckv_b, w.ckv_g, w.ckvdt, o.ckv_slots * o.cw, Me, (int)(o.ckv_slots * o.cw), d));
    }
    if (!op->memory) TRY(encoder_forward(c, o, att_feats, boxes, att_masks, B, S, w.x0, w.logbias, ep, w.mem, A, w.st, enc_qdt, nullptr, &ecs, w.chain_pk, false));
    // the decoder's sparse weight stream (count / scan / fill: 0.37 ms): on the side stream behind the geometry bias, beside the encoder
    // pass, which does not read it; the first decoder position waits for it
    hipEvent_t sstream_done = nullptr;
    if (stack && sstream) {
        if (c.use_side && w16_done) {
            if (hipStreamWaitEvent(c.side->s, w16_done, 0) != hipSuccess) return ORTK_EINVAL;     // (NOT a fork: the encoder pass is queued already)
            TRY(pl.gather ? gstack_pack(w.w16, w.ss, tp, c.side->s) : sstack_pack(w.w16, w.ss, tp, c.side->s));
            TRY(c.side_mark(&sstream_done));
        } else {
            TRY(pl.gather ? gstack_pack(w.w16, w.ss, tp, s) : sstack_pack(w.w16, w.ss, tp, s));
        }
    }
    c.use_side = false;
    TRY(fwd_gemm(c, op->memory ? op->memory : w.mem, A, d, o.ckv_w, P + o.ckv_b, w.ckv, w.ckvdt, o.ckv_slots * o.cw, Me, (int)(o.ckv_slots * o.cw), d));

    const int64_t rows_full = (int64_t)B * K;
    BeamState bs; std::memset(&bs, 0, sizeof(bs));
    SampleState ss; std::memset(&ss, 0, sizeof(ss));
    if (beam) {
        bs.B = B; bs.b = K; bs.L = T; bs.V = V; bs.eos = cfg->eos_id; bs.ldv = w.ldv;
        for (int i = 0; i < 2; ++i) { bs.seq[i] = w.bseq[i]; bs.tok_lp[i] = w.blp[i]; bs.kvidx[i] = w.kvidx[i]; }
        bs.cum = w.cum; bs.it = w.it; bs.done_seq = w.done_seq; bs.done_lp = w.done_lp; bs.done_p = w.done_p;
        bs.done_len = w.done_len; bs.done_cnt = w.done_cnt; bs.decoding_constraint = op->decoding_constraint;
        bs.length_penalty = op->length_penalty; bs.length_alpha = op->length_alpha; bs.tmax = T;
        // Soft-max partials from the generator GEMM instead of a second pass over the logits (mixed precision, no temperature,
        // dense generator, d_model a multiple of 64 and full 128-column tiles: what the GEMM's statistics epilogue serves)
        if (w.gstats && op->temperature == 1.f && !op->sparse && A == ORTK_BF16 && d % 64 == 0 && w.ldv % 128 == 0 && w.ldv / 64 <= 256) {
            bs.gstats = w.gstats; bs.nblk = (int32_t)(w.ldv / 64);
        }
        TRY(fill_i32(w.done_cnt, B, 0, s));
        TRY(fill_i64(w.it, B, cfg->bos_id, s));
        TRY(kvidx_init(w.kvidx[0], B, K, T, s));
    } else {
        ss.rows = (int)rows_full; ss.L = T; ss.V = V; ss.eos = cfg->eos_id; ss.ldv = w.ldv; ss.it = w.it; ss.seq = seq_out;
        ss.lp = logprob_out; ss.unfinished = w.unfinished; ss.last_step = w.last_step;
        ss.decoding_constraint = op->decoding_constraint; ss.sample = op->num_random_sample > 0; ss.temperature = op->temperature;
        ss.seed = op->seed;
        ss.greedy_stride = (op->num_random_sample > 0 && op->with_greedy) ? K : 0;
        ss.row_offset = op->sample_row_offset;
        TRY(sample_init(ss, cfg->bos_id, s));
    }
    // Sampling decodes in mixed precision on the dense generator: the generator GEMM's epilogue emits the Gumbel-max candidates and the
    // soft-max partials of every 64-logit block and a combine step picks the token — the (rows, V) fp32 logits are never stored
    // (the conditions are the statistics epilogue's: ortk_gemm; tuning().samp_epilogue = 0 keeps the logit rows and sample_step)
    const bool samp_epi = !beam && w.gsamp && w.gstats && tuning().samp_epilogue && !op->sparse && !c.ell_f && A == ORTK_BF16 && d % 64 == 0 && w.ldv % 128 == 0 &&
                          w.ldv / 64 <= 256 && op->temperature > 0.f;
    int uniq_slot = 0;          // (profiling only) counter of the unique cache rows the NEXT pass references, filled by this pass's beam step
    TRY(c.wait_ev(sstream_done));
    for (int t = 0; t < T; ++t) {
        // rows of this pass: the first beam pass runs one row per image (transformer.py:488), then b per image
        const bool first_beam = beam && t == 0;
        const int64_t rows = first_beam ? B : rows_full;
        const int per_img = first_beam ? 1 : K;
        const int row_mult = first_beam ? K : 1;
        StepBufs sb{w.it, w.xa, w.xb, w.y, w.qkv, w.o, w.q, w.h, w.logits, w.st, w.ldv, w.ckv, att_masks};
        sb.kvdt = w.kvdt; sb.ckvdt = w.ckvdt; sb.xq16 = w.xq16;
        if (beam && bs.gstats) { sb.gstats = w.gstats; sb.stat_ncols = V; }
        if (samp_epi) { sb.gstats = w.gstats; sb.stat_ncols = V; sb.gsamp = w.gsamp; sb.samp = &ss; sb.samp_t = t; sb.samp_fast = cfg->precision == 1; }
        for (int l = 0; l < L; ++l) { sb.cache_k[l] = w.cache_k[l]; sb.cache_v[l] = w.cache_v[l]; }
        const SplitBufs spb{w.tp, w.tp_wpk, w.tp_xbuf, w.tp_flag, stack_tp_groups(rows_full), w.status};
        if (stack) TRY(decoder_stack_step(c, o, sb, w.wpk, sstream ? &w.ss : nullptr, w.progress, op->exec_flags, rows, per_img, S, T, t, beam ? w.kvidx[t & 1] : nullptr, &spb,
                                          greedy_rows ? K : 0, w.ckv_g, uniq_slot));
        else TRY(decoder_step(c, o, sb, rows, B, per_img, row_mult, S, T, t, beam ? w.kvidx[t & 1] : nullptr));
        // first-step log-probs are plain log_softmax; later beam steps re-normalise logp / temperature
        // (transformer.py:488 vs caption_model.py:218); greedy / multinomial never rescale the log-probs themselves.
        const float scale = (beam && t > 0) ? 1.f / op->temperature : 1.f;
        const bool fast_exp = cfg->precision == 1;                     // (the fp32 parity mode keeps libm's expf)
        if (beam && stack && ortk_prof_active()) { int ix = -1; bs.uniq = prof_slot(&ix); uniq_slot = ix + 1; }
        if (beam) TRY(beam_step(bs, w.logits, t, s, true, scale, fast_exp));     // log-soft-max fused into the candidate scan
        else if (samp_epi) TRY(sample_combine(ss, w.gstats, w.gsamp, (int32_t)(w.ldv / 64), t, s, fast_exp));      // the generator's epilogue has the candidates
        else if (V <= 256 * 40) TRY(sample_step(ss, w.logits, t, s, true, fast_exp));     // log-soft-max fused (scale is 1 on this branch)
        else {
            TRY(ortk_log_softmax(w.logits, rows, V, w.ldv, scale, stream));
            TRY(sample_step(ss, w.logits, t, s));
        }
    }
    if (beam) TRY(beam_finalize(bs, seq_out, logprob_out, score_out, s));
    else {
        TRY(sample_finalize(ss, s));
        if (score_out) TRY(ortk_fill(score_out, rows_full, 0.f, stream));
    }
    // the column-split kernel's exchanges are bounded waits: a decode whose groups never met must not pass for a result
    if (w.tp) TRY(decode_poison(w.status, seq_out, logprob_out, score_out, rows_full * T, rows_full, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------ per-step host API
namespace ortk {
struct StepWS { void* w16; float *xa, *xb; void* y; float* qkv; void* o; float* q; void* h; float* logits; float* st; int64_t ldv; size_t bytes; };
static void carve_step(const ortk_config& c, int64_t rows, void* base, StepWS& w) {
    const int64_t d = c.d_model, ff = c.d_ff;
    const size_t es = ortk_esize(c.precision ? ORTK_BF16 : ORTK_F32);
    Bump b{reinterpret_cast<char*>(base), 0};
    Offsets o; build_layout(c, o, nullptr);
    w.ldv = ortk_align(c.vocab, 128);
    w.w16 = c.precision ? b.take_bytes((size_t)o.total * 2) : nullptr;
    w.xa = b.take<float>(rows * d); w.xb = b.take<float>(rows * d); w.y = b.take_bytes((size_t)rows * d * es);
    w.qkv = b.take<float>(rows * 3 * d); w.o = b.take_bytes((size_t)rows * d * es); w.q = b.take<float>(rows * d);
    w.h = b.take_bytes((size_t)rows * ff * es); w.logits = b.take<float>(rows * w.ldv); w.st = b.take<float>(rows * 2);
    w.bytes = (b.off + 255) & ~(size_t)255;
}
}  // namespace ortk

extern "C" size_t ortk_decode_step_workspace_bytes(const ortk_config* cfg, int32_t rows) {
    if (check_cfg(cfg) || rows < 1) return 0;
    StepWS w; carve_step(*cfg, rows, nullptr, w);
    return w.bytes;
}

extern "C" int ortk_project_memory(const ortk_config* cfg, const float* params, const float* memory, int64_t mem_rows, void* ws,
                                   size_t ws_bytes, float* cross_kv, ortk_stream stream) {
    if (int e = check_cfg(cfg)) return e;
    if (!params || !memory || !cross_kv || !ws || mem_rows < 1) return ORTK_EINVAL;
    Offsets o; build_layout(*cfg, o, nullptr);
    StepWS w; carve_step(*cfg, 1, ws, w);
    if (w.bytes > ws_bytes) return ORTK_ENOSPC;
    TRY(make_w16(cfg, o, params, w.w16, stream));
    Ctx c{cfg, ortk_s(stream), cfg->precision, 0, false, params, w.w16, cfg->precision ? ORTK_BF16 : ORTK_F32};
    const int d = cfg->d_model, L = cfg->n_layers;
    const int U = o.ckv_slots;      // distinct decoder layers (ortk_config.share_dec)
    (void)L;
    return fwd_gemm(c, memory, ORTK_F32, d, o.ckv_w, params + o.ckv_b, cross_kv, ORTK_F32, U * o.cw, mem_rows, (int)(U * o.cw), d);
}

extern "C" int ortk_decode_step(const ortk_config* cfg, const float* params, const int64_t* it, int32_t t, int32_t rows,
                                int32_t kv_groups, int32_t S, const float* cross_kv, const float* att_masks, float* self_k,
                                float* self_v, int32_t tmax, void* ws, size_t ws_bytes, float* logp_out, int64_t ld_out,
                                ortk_stream stream) {
    if (int e = check_cfg(cfg)) return e;
    if (!params || !it || !cross_kv || !att_masks || !self_k || !self_v || !ws || !logp_out) return ORTK_EINVAL;
    if (rows < 1 || kv_groups < 1 || rows % kv_groups || S < 1 || S > 128 || t < 0 || t >= tmax || ld_out < cfg->vocab) return ORTK_EINVAL;
    Offsets o; build_layout(*cfg, o, nullptr);
    StepWS w; carve_step(*cfg, rows, ws, w);
    if (w.bytes > ws_bytes) return ORTK_ENOSPC;
    TRY(make_w16(cfg, o, params, w.w16, stream));
    Ctx c{cfg, ortk_s(stream), cfg->precision, 0, false, params, w.w16, cfg->precision ? ORTK_BF16 : ORTK_F32};
    const int64_t d = cfg->d_model;
    StepBufs sb{it, w.xa, w.xb, w.y, w.qkv, w.o, w.q, w.h, w.logits, w.st, w.ldv, cross_kv, att_masks};
    sb.kvdt = ORTK_F32;        // the caller's caches are fp32 (reference state layout)
    for (int l = 0; l < cfg->n_layers; ++l) {
        sb.cache_k[l] = self_k + (int64_t)l * rows * tmax * d;
        sb.cache_v[l] = self_v + (int64_t)l * rows * tmax * d;
    }
    TRY(decoder_step(c, o, sb, rows, kv_groups, rows / kv_groups, 1, S, tmax, t, nullptr));
    TRY(ortk_log_softmax(w.logits, rows, cfg->vocab, w.ldv, 1.f, stream));
    if (hipMemcpy2DAsync(logp_out, (size_t)ld_out * 4, w.logits, (size_t)w.ldv * 4, (size_t)cfg->vocab * 4, (size_t)rows,
                         hipMemcpyDeviceToDevice, ortk_s(stream)) != hipSuccess) return ORTK_EINVAL;
    return 0;
}

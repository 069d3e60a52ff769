// Host-side executors of librdm_hip: context, weight blob/manifest, UNet / VQ-decoder / CLIP graph
// walkers, DDIM / DDPM loops, and the C ABI declared in include/rdm_hip.h.
//
// The executors mirror the reference's module graphs
//   UNetModel.__init__/forward      rdm/modules/diffusionmodules/openaimodel.py:66-371
//   SpatialTransformer / BasicTransformerBlock / CrossAttention   rdm/modules/attention.py:20-196
//   DDIMSampler                     rdm/models/diffusion/ddim.py:27-268
//   CLIP                            rdm/modules/custom_clip/model.py:152-336
//   [ldm, un-vendored] ResBlock, Down/Upsample, VQModelInterface.decode, p_sample_loop
// but are laid out for the hardware: NHWC bf16 activations (a 1x1 conv, a Linear and a token
// sequence are the same [M,C] matrix), fused epilogues (bias, time-embedding add, residual, GEGLU,
// SiLU), no materialised skip-concat, cross-attention K/V and all 22 emb_layers projections batched
// into one GEMM each, K/V of the retrieved neighbours computed once per sample() call instead of
// once per step per layer.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rdm_hip.h"
#include "kernels.h"
#include "knn.h"

#define RDM_CHECK_HIP(ctx, expr)                                                            \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return (ctx)->fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define RDM_TRY(expr)            \
    do {                         \
        int _r = (expr);         \
        if (_r != 0) return _r;  \
    } while (0)

// Every C-ABI entry binds the calling thread to the context's device for the duration of the call and restores the
// caller's current device on exit (a context for device 1 used from a thread whose current device is 0 must neither
// launch on device 0 nor leave the caller's device changed).
struct DevGuard {
    int prev = -1; bool changed = false;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard&) = delete; DevGuard& operator=(const DevGuard&) = delete;
};
#define RDM_ENTER(c) if (!(c)) return -1; DevGuard _dev_guard((c)->device)

// ------------------------------------------------------------------------------------ manifest
struct Manifest {
    std::string text;
    size_t total = 0;
    size_t add(const std::string& kind, const std::string& srcs, size_t nbytes) {      // kind may carry a padding recipe ("|R=..|C=..", build_unet)
        total = (total + 255) & ~(size_t)255;
        const size_t off = total;
        text += std::to_string(off) + " " + std::to_string(nbytes) + " " + kind + " "; text += srcs; text += "\n";
        total += nbytes;
        return off;
    }
};

// ------------------------------------------------------------------------------------ arena
struct Arena {
    char* base = nullptr; size_t cap = 0, off = 0, peak = 0; bool planning = false;
    void reset() { off = 0; }
    void* alloc(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : (void*)(uintptr_t)(off + 256);
        off += bytes;
        if (off > peak) peak = off;
        return p;
    }
};

// ------------------------------------------------------------------------------------ UNet description
// Channel padding.  Every GEMM / conv K loop advances in 64-channel slices and the skip-concat split points sit on slice boundaries,
// so an activation with C channels is held with P(C) = C rounded up to 64 of them, the tail zero: models/rdm/ffhq has
// model_channels 224 (224 / 448 / 672 / 896: multiples of 32 only).  Weights are padded to match by the packer -- the manifest kind
// carries the recipe, "|R=" row segments and "|C=" channel segments as logical>padded pairs, absent when nothing is padded (the
// ImageNet models: byte-identical manifests) -- with zero rows / columns / biases / norm affines in the tail, so a padded channel
// stays exactly zero through every layer.  Only the normalisations must know the LOGICAL counts (group membership and divisor of
// GroupNorm, mean / variance of LayerNorm); attention simply gains all-zero heads.  cin / cout / c below are PHYSICAL counts.
static inline int pad64(int c) { return (c + 63) & ~63; }
struct Seg { int n, p; };
static std::string seg_spec(const char* tag, std::initializer_list<Seg> segs) {
    bool any = false; for (const Seg& g : segs) any |= g.n != g.p;
    if (!any) return "";
    std::string o = std::string("|") + tag + "=";
    bool first = true;
    for (const Seg& g : segs) { char b[48]; snprintf(b, sizeof b, "%s%d>%d", first ? "" : ",", g.n, g.p); o += b; first = false; }
    return o;
}
struct ResW { int cin, cout; size_t gn1g, gn1b, w1, b1, gn2g, gn2b, w2, b2, wsk, bsk; int emb_off; bool skip; int l0, l1, lout, p0, p1; };
constexpr int XA_NP = 128;        // padded (heads x neighbours) width of the skinny cross-attention operands

struct StW {
    int c, heads; size_t gng, gnb, win, bin, ln1g, ln1b, wqk, wv, wo1, bo1, ln2g, ln2b, wq2, wo2, bo2, ln3g, ln3b, wff1,
        bff1, wff2, bff2, wout, bout, wfo, bfo; int kv_off; long long xa_unit;   // xa_unit: per-sample element offset of this layer's (G, U) pair
    int lc;                           // logical channels (c is padded)
    bool v_follows;
};
struct ConvW { int c; size_t w, b; int lc; };
struct ULayer { int kind; int idx; };            // 0 conv_in, 1 res, 2 st, 3 down, 4 up
struct UBlock { int where; std::vector<ULayer> layers; };   // where: 0 input, 1 middle, 2 output

struct UNet {
    rdm_unet_cfg cfg{};
    bool loaded = false;
    std::vector<UBlock> blocks; std::vector<ResW> res; std::vector<StW> st; std::vector<ConvW> down, up;
    size_t te0w, te0b, te2w, te2b, embw, embb, kvw, cinw, cinb, outg, outb, outw, outbias;
    int emb_total = 0, kv_total = 0; long long xa_total = 0;         // xa_total: per-sample elements of all (G, U) pairs
    bf16_t* xa_cache = nullptr; size_t xa_cache_bytes = 0;
    char* blob = nullptr; size_t blob_bytes = 0;
    Arena arena;
    // cached cross-attention K/V for the current conditioning
    bf16_t* kv_cache = nullptr; size_t kv_cache_bytes = 0;
    float* emb_table = nullptr; size_t emb_table_bytes = 0;      // rdm_ddim_sample: one row of emb_total floats per sampler timestep
    int ctx_rows = 0;            // samples [ctx_rows, B') of the cached conditioning have ALL-ZERO neighbours (the unconditional half of
                                 // a guided batch): their cross-attention is the output bias, no GEMM runs for them
};

static std::string key(const std::string& pre, const char* s) { return pre + s; }

static void build_unet(UNet& u, const rdm_unet_cfg& c, Manifest& mf) {
    u.cfg = c; u.blocks.clear(); u.res.clear(); u.st.clear(); u.down.clear(); u.up.clear();
    const int mc = c.model_channels, ted = mc * 4, mcp = pad64(mc);       // ted = 4 mc: a multiple of 64 whenever mc % 32 == 0... (128-aligned)
    // vec: fp32 vector(s) over channel segments; mat: bf16 [rows][cols]; both padded per segment
    auto vec = [&](const std::string& n, std::initializer_list<Seg> segs) {
        size_t tot = 0; for (const Seg& g : segs) tot += g.p;
        return mf.add("f32" + seg_spec("R", segs), n, tot * 4);
    };
    auto mat = [&](const char* kind, const std::string& n, std::initializer_list<Seg> rows, std::initializer_list<Seg> cols, size_t elt = 2, int taps = 1) {
        size_t r = 0, k = 0; for (const Seg& g : rows) r += g.p; for (const Seg& g : cols) k += g.p;
        return mf.add((std::string(kind) + seg_spec("R", rows) + seg_spec("C", cols)), n, r * k * taps * elt);
    };
    const Seg TED{ted, ted};
    u.te0w = mat("bf16", "time_embed.0.weight", {TED}, {{mc, mcp}}); u.te0b = vec("time_embed.0.bias", {TED});
    u.te2w = mat("bf16", "time_embed.2.weight", {TED}, {TED}); u.te2b = vec("time_embed.2.bias", {TED});
    std::string emb_w_srcs, emb_b_srcs, kv_srcs, emb_rspec, kv_rspec;
    bool emb_padded = false, kv_padded = false;
    u.emb_total = 0; u.kv_total = 0; u.xa_total = 0;
    auto in_attn = [&](int ds) { for (int i = 0; i < c.n_attention_resolutions; i++) if (c.attention_resolutions[i] == ds) return true; return false; };
    auto add_seg = [](std::string& spec, bool& padded, int n, int p) { char b[48]; snprintf(b, sizeof b, "%s%d>%d", spec.empty() ? "" : ",", n, p); spec += b; padded |= n != p; };
    auto add_res = [&](const std::string& pre, int l0, int l1, int lout) {       // input = [l0 | l1] logical channels (l1: the skip tensor, or 0)
        ResW r{}; r.l0 = l0; r.l1 = l1; r.lout = lout; r.p0 = pad64(l0); r.p1 = l1 ? pad64(l1) : 0;
        r.cin = r.p0 + r.p1; r.cout = pad64(lout); r.skip = (l0 + l1) != lout;
        const Seg S0{l0, r.p0}, S1{l1, r.p1}, SO{lout, r.cout};
        if (l1) { r.gn1g = vec(pre + ".in_layers.0.weight", {S0, S1}); r.gn1b = vec(pre + ".in_layers.0.bias", {S0, S1}); }
        else { r.gn1g = vec(pre + ".in_layers.0.weight", {S0}); r.gn1b = vec(pre + ".in_layers.0.bias", {S0}); }
        r.w1 = l1 ? mat("conv3", pre + ".in_layers.2.weight", {SO}, {S0, S1}, 2, 9) : mat("conv3", pre + ".in_layers.2.weight", {SO}, {S0}, 2, 9);
        r.b1 = vec(pre + ".in_layers.2.bias", {SO});
        r.gn2g = vec(pre + ".out_layers.0.weight", {SO}); r.gn2b = vec(pre + ".out_layers.0.bias", {SO});
        r.w2 = mat("conv3", pre + ".out_layers.3.weight", {SO}, {SO}, 2, 9); r.b2 = vec(pre + ".out_layers.3.bias", {SO});
        if (r.skip) {
            r.wsk = l1 ? mat("bf16", pre + ".skip_connection.weight", {SO}, {S0, S1}) : mat("bf16", pre + ".skip_connection.weight", {SO}, {S0});
            r.bsk = vec(pre + ".skip_connection.bias", {SO});
        }
        r.emb_off = u.emb_total; u.emb_total += r.cout;
        if (!emb_w_srcs.empty()) { emb_w_srcs += ","; emb_b_srcs += ","; }
        emb_w_srcs += pre + ".emb_layers.1.weight"; emb_b_srcs += pre + ".emb_layers.1.bias";
        add_seg(emb_rspec, emb_padded, lout, r.cout);
        u.res.push_back(r); return (int)u.res.size() - 1;
    };
    auto add_st = [&](const std::string& pre, int lch) {
        StW s{}; s.lc = lch; s.c = pad64(lch); s.heads = s.c / c.num_head_channels;
        const int ch = s.c;
        const Seg S{lch, ch}, F{4 * lch, 4 * lch};
        const std::string tb = pre + ".transformer_blocks.0";
        s.gng = vec(pre + ".norm.weight", {S}); s.gnb = vec(pre + ".norm.bias", {S});
        s.win = mat("bf16", pre + ".proj_in.weight", {S}, {S}); s.bin = vec(pre + ".proj_in.bias", {S});
        s.ln1g = vec(tb + ".norm1.weight", {S}); s.ln1b = vec(tb + ".norm1.bias", {S});
        s.wqk = mat("bf16", tb + ".attn1.to_q.weight," + tb + ".attn1.to_k.weight", {S, S}, {S});
        s.wv = mat("bf16", tb + ".attn1.to_v.weight", {S}, {S});
        s.v_follows = s.wv == s.wqk + (size_t)2 * ch * ch * 2;      // to_v directly behind to_q | to_k in the blob: q | k | v is ONE projection (else the separate V path runs)
        s.wo1 = mat("bf16", tb + ".attn1.to_out.0.weight", {S}, {S}); s.bo1 = vec(tb + ".attn1.to_out.0.bias", {S});
        s.ln2g = vec(tb + ".norm2.weight", {S}); s.ln2b = vec(tb + ".norm2.bias", {S});
        s.wq2 = mat("bf16", tb + ".attn2.to_q.weight", {S}, {S});
        s.wo2 = mat("bf16", tb + ".attn2.to_out.0.weight", {S}, {S}); s.bo2 = vec(tb + ".attn2.to_out.0.bias", {S});
        s.ln3g = vec(tb + ".norm3.weight", {S}); s.ln3b = vec(tb + ".norm3.bias", {S});
        // GEGLU hidden width 4 * lch (a multiple of 128): not padded, only its K side
        s.wff1 = mat("geglu_w", tb + ".ff.net.0.proj.weight", {{8 * lch, 8 * lch}}, {S});
        s.bff1 = mf.add("geglu_b", tb + ".ff.net.0.proj.bias", (size_t)8 * lch * 4);
        s.wff2 = mat("bf16", tb + ".ff.net.2.weight", {S}, {F}); s.bff2 = vec(tb + ".ff.net.2.bias", {S});
        s.wout = mat("bf16", pre + ".proj_out.weight", {S}, {S}); s.bout = vec(pre + ".proj_out.bias", {S});
        // ff.net.2 followed by proj_out is one linear map of [ff | t2]:  [W_out W_2 | W_out], bias W_out b_2 + b_out (packed in fp32)
        s.wfo = mat("fuse_w", tb + ".ff.net.2.weight," + pre + ".proj_out.weight", {S}, {F, S});
        s.bfo = mf.add("fuse_b" + seg_spec("R", {S}), tb + ".ff.net.2.bias," + pre + ".proj_out.weight," + pre + ".proj_out.bias", (size_t)ch * 4);
        s.kv_off = u.kv_total; u.kv_total += 2 * ch;
        s.xa_unit = u.xa_total; u.xa_total += 4LL * XA_NP * ch;      // G, U row-major + their fragment-ordered images (attention.hip: xattn_fused_kernel)
        if (!kv_srcs.empty()) kv_srcs += ",";
        kv_srcs += tb + ".attn2.to_k.weight," + tb + ".attn2.to_v.weight";
        add_seg(kv_rspec, kv_padded, lch, ch); add_seg(kv_rspec, kv_padded, lch, ch);
        u.st.push_back(s); return (int)u.st.size() - 1;
    };
    auto name_of = [](const char* grp, int i, int j) { char b[64]; snprintf(b, sizeof b, "%s.%d.%d", grp, i, j); return std::string(b); };

    // input blocks (openaimodel.py:144-215); all channel bookkeeping below is LOGICAL
    std::vector<int> chans;
    {
        UBlock b; b.where = 0; b.layers.push_back({0, 0});
        u.cinw = mf.add("f32" + seg_spec("R", {{mc, mcp}}), "input_blocks.0.0.weight", (size_t)mcp * c.in_channels * 9 * 4);     // [mc][in][3][3]: rows padded
        u.cinb = vec("input_blocks.0.0.bias", {{mc, mcp}});
        u.blocks.push_back(b); chans.push_back(mc);
    }
    int ch = mc, ds = 1, idx = 1;
    for (int level = 0; level < c.n_channel_mult; level++) {
        const int mult = c.channel_mult[level];
        for (int r = 0; r < c.num_res_blocks; r++) {
            UBlock b; b.where = 0;
            b.layers.push_back({1, add_res(name_of("input_blocks", idx, 0), ch, 0, mult * mc)});
            ch = mult * mc;
            if (in_attn(ds)) b.layers.push_back({2, add_st(name_of("input_blocks", idx, 1), ch)});
            u.blocks.push_back(b); idx++; chans.push_back(ch);
        }
        if (level != c.n_channel_mult - 1) {
            UBlock b; b.where = 0;
            ConvW d{}; d.lc = ch; d.c = pad64(ch); const std::string pre = name_of("input_blocks", idx, 0);
            d.w = mat("conv3", pre + ".op.weight", {{ch, d.c}}, {{ch, d.c}}, 2, 9); d.b = vec(pre + ".op.bias", {{ch, d.c}});
            u.down.push_back(d); b.layers.push_back({3, (int)u.down.size() - 1});
            u.blocks.push_back(b); idx++; chans.push_back(ch); ds *= 2;
        }
    }
    {   // middle (openaimodel.py:223-249)
        UBlock b; b.where = 1;
        b.layers.push_back({1, add_res("middle_block.0", ch, 0, ch)});
        b.layers.push_back({2, add_st("middle_block.1", ch)});
        b.layers.push_back({1, add_res("middle_block.2", ch, 0, ch)});
        u.blocks.push_back(b);
    }
    int oidx = 0;   // output blocks (openaimodel.py:252-305)
    for (int level = c.n_channel_mult - 1; level >= 0; level--) {
        const int mult = c.channel_mult[level];
        for (int i = 0; i <= c.num_res_blocks; i++) {
            const int ich = chans.back(); chans.pop_back();
            UBlock b; b.where = 2; int j = 0;
            b.layers.push_back({1, add_res(name_of("output_blocks", oidx, j++), ch, ich, mc * mult)});
            ch = mc * mult;
            if (in_attn(ds)) b.layers.push_back({2, add_st(name_of("output_blocks", oidx, j++), ch)});
            if (level && i == c.num_res_blocks) {
                ConvW up{}; up.lc = ch; up.c = pad64(ch); const std::string pre = name_of("output_blocks", oidx, j++);
                up.w = mat("conv3", pre + ".conv.weight", {{ch, up.c}}, {{ch, up.c}}, 2, 9); up.b = vec(pre + ".conv.bias", {{ch, up.c}});
                u.up.push_back(up); b.layers.push_back({4, (int)u.up.size() - 1}); ds /= 2;
            }
            u.blocks.push_back(b); oidx++;
        }
    }
    u.outg = vec("out.0.weight", {{mc, mcp}}); u.outb = vec("out.0.bias", {{mc, mcp}});
    // out conv weights stay [Cout][Cin][3][3] fp32 (misc.hip conv_out_kernel): the Cin axis is padded
    u.outw = mf.add(mc == mcp ? std::string("f32") : "f32_cin" + seg_spec("C", {{mc, mcp}}), "out.2.weight", (size_t)c.out_channels * mcp * 9 * 4);
    u.outbias = mf.add("f32", "out.2.bias", (size_t)c.out_channels * 4);
    u.embw = mf.add(std::string("bf16") + (emb_padded ? "|R=" + emb_rspec : ""), emb_w_srcs, (size_t)u.emb_total * ted * 2);
    u.embb = mf.add(std::string("f32") + (emb_padded ? "|R=" + emb_rspec : ""), emb_b_srcs, (size_t)u.emb_total * 4);
    u.kvw = mf.add(std::string("bf16") + (kv_padded ? "|R=" + kv_rspec : ""), kv_srcs, (size_t)u.kv_total * c.context_dim * 2);
}

// ------------------------------------------------------------------------------------ VQ decoder description
struct VqRes { int cin, cout; size_t n1g, n1b, w1, b1, n2g, n2b, w2, b2, wsk, bsk; bool skip; };
struct VqAttn { int c; size_t ng, nb, wq, bq, wk, bk, wv, bv, wo, bo; };
struct VqModel {
    rdm_vq_cfg cfg{}; bool loaded = false;
    bool wide = false;                             // z_channels > 4 (taming VQGAN-f16: 256): latent handled as bf16 NHWC tokens, GEMM-class
                                                   // post_quant_conv / conv_in; else the 3-channel VQ-f4 path (tiny fp32 stem kernels)
    size_t codebook, pqw, pqb, cinw, cinb, noutg, noutb, coutw, coutb;
    VqRes mid1, mid2; VqAttn attn;
    std::vector<std::vector<VqRes>> up_blocks;     // indexed by level
    std::vector<std::vector<VqAttn>> up_attn;      // indexed by level: one AttnBlock per res block when the level's resolution is in attn_resolutions
    std::vector<ConvW> upsample;                   // indexed by level (level 0 unused)
    char* blob = nullptr; size_t blob_bytes = 0; Arena arena;
};

static void build_vq(VqModel& v, const rdm_vq_cfg& c, Manifest& mf) {
    v.cfg = c;
    auto f32 = [&](const std::string& n, size_t numel) { return mf.add("f32", n, numel * 4); };
    auto bf = [&](const std::string& n, size_t numel) { return mf.add("bf16", n, numel * 2); };
    auto add_res = [&](const std::string& pre, int cin, int cout) {
        VqRes r{}; r.cin = cin; r.cout = cout; r.skip = cin != cout;
        r.n1g = f32(pre + ".norm1.weight", cin); r.n1b = f32(pre + ".norm1.bias", cin);
        r.w1 = mf.add("conv3", pre + ".conv1.weight", (size_t)cout * cin * 9 * 2); r.b1 = f32(pre + ".conv1.bias", cout);
        r.n2g = f32(pre + ".norm2.weight", cout); r.n2b = f32(pre + ".norm2.bias", cout);
        r.w2 = mf.add("conv3", pre + ".conv2.weight", (size_t)cout * cout * 9 * 2); r.b2 = f32(pre + ".conv2.bias", cout);
        if (r.skip) { r.wsk = bf(pre + ".nin_shortcut.weight", (size_t)cout * cin); r.bsk = f32(pre + ".nin_shortcut.bias", cout); }
        return r;
    };
    auto add_attn = [&](const std::string& p, int ch) {
        VqAttn a{}; a.c = ch;
        a.ng = f32(p + ".norm.weight", ch); a.nb = f32(p + ".norm.bias", ch);
        a.wq = bf(p + ".q.weight", (size_t)ch * ch); a.bq = f32(p + ".q.bias", ch);
        a.wk = bf(p + ".k.weight", (size_t)ch * ch); a.bk = f32(p + ".k.bias", ch);
        a.wv = bf(p + ".v.weight", (size_t)ch * ch); a.bv = f32(p + ".v.bias", ch);
        a.wo = bf(p + ".proj_out.weight", (size_t)ch * ch); a.bo = f32(p + ".proj_out.bias", ch);
        return a;
    };
    v.wide = c.z_channels > 4;
    if (!c.kl) v.codebook = f32("quantize.embedding.weight", (size_t)c.n_embed * c.embed_dim);
    int bin = c.ch * c.ch_mult[c.n_ch_mult - 1];
    if (v.wide) {
        v.pqw = bf("post_quant_conv.weight", (size_t)c.z_channels * c.embed_dim); v.pqb = f32("post_quant_conv.bias", c.z_channels);
        v.cinw = mf.add("conv3", "decoder.conv_in.weight", (size_t)bin * c.z_channels * 9 * 2); v.cinb = f32("decoder.conv_in.bias", bin);
    } else {
        v.pqw = f32("post_quant_conv.weight", (size_t)c.z_channels * c.embed_dim); v.pqb = f32("post_quant_conv.bias", c.z_channels);
        v.cinw = f32("decoder.conv_in.weight", (size_t)bin * c.z_channels * 9); v.cinb = f32("decoder.conv_in.bias", bin);
    }
    v.mid1 = add_res("decoder.mid.block_1", bin, bin);
    if (c.mid_attn) v.attn = add_attn("decoder.mid.attn_1", bin);
    v.mid2 = add_res("decoder.mid.block_2", bin, bin);
    v.up_blocks.assign(c.n_ch_mult, {}); v.upsample.assign(c.n_ch_mult, ConvW{}); v.up_attn.assign(c.n_ch_mult, {});
    int curr_res = c.resolution >> (c.n_ch_mult - 1);
    for (int lvl = c.n_ch_mult - 1; lvl >= 0; lvl--) {
        const int bout = c.ch * c.ch_mult[lvl];
        bool at = false;
        for (int i = 0; i < c.n_attn_resolutions; i++) at = at || c.attn_resolutions[i] == curr_res;
        for (int i = 0; i <= c.num_res_blocks; i++) {
            char pre[64]; snprintf(pre, sizeof pre, "decoder.up.%d.block.%d", lvl, i);
            v.up_blocks[lvl].push_back(add_res(pre, bin, bout)); bin = bout;
            if (at) { snprintf(pre, sizeof pre, "decoder.up.%d.attn.%d", lvl, i); v.up_attn[lvl].push_back(add_attn(pre, bin)); }
        }
        if (lvl != 0) {
            char pre[64]; snprintf(pre, sizeof pre, "decoder.up.%d.upsample.conv", lvl);
            ConvW u{}; u.c = bin; u.w = mf.add("conv3", std::string(pre) + ".weight", (size_t)bin * bin * 9 * 2);
            u.b = f32(std::string(pre) + ".bias", bin); v.upsample[lvl] = u;
            curr_res *= 2;
        }
    }
    v.noutg = f32("decoder.norm_out.weight", bin); v.noutb = f32("decoder.norm_out.bias", bin);
    v.coutw = f32("decoder.conv_out.weight", (size_t)c.out_ch * bin * 9); v.coutb = f32("decoder.conv_out.bias", c.out_ch);
}

// ------------------------------------------------------------------------------------ VQ-f4 first-stage ENCODER description (training input)
// ldm Encoder (ldm/modules/diffusionmodules/model.py; un-vendored: restated from the published code, parity unpinned) as reached from
// MinimalRETRODiffusion.get_input -> encode_first_stage -> VQModelInterface.encode = quant_conv(encoder(x)) under torch.no_grad()
// (rdm/models/diffusion/ddpm.py:390-391): conv_in, per level num_res_blocks ResnetBlocks (+ AttnBlocks at attn_resolutions) and a
// stride-2 Downsample conv with (0, 1, 0, 1) zero padding, mid res-attn-res, GroupNorm + swish + conv_out, quant_conv (1x1).
struct VqEncModel {
    rdm_vq_cfg cfg{}; bool loaded = false;
    size_t cinw, cinb, noutg, noutb, coutw, coutb, qw, qb;
    std::vector<std::vector<VqRes>> down; std::vector<std::vector<VqAttn>> down_attn; std::vector<ConvW> downsample;   // indexed by level
    VqRes mid1, mid2; VqAttn attn;
    char* blob = nullptr; size_t blob_bytes = 0; Arena arena;
};
static void build_vqenc(VqEncModel& v, const rdm_vq_cfg& c, Manifest& mf) {
    v.cfg = c;
    auto f32 = [&](const std::string& n, size_t numel) { return mf.add("f32", n, numel * 4); };
    auto bf = [&](const std::string& n, size_t numel) { return mf.add("bf16", n, numel * 2); };
    auto add_res = [&](const std::string& pre, int cin, int cout) {
        VqRes r{}; r.cin = cin; r.cout = cout; r.skip = cin != cout;
        r.n1g = f32(pre + ".norm1.weight", cin); r.n1b = f32(pre + ".norm1.bias", cin);
        r.w1 = mf.add("conv3", pre + ".conv1.weight", (size_t)cout * cin * 9 * 2); r.b1 = f32(pre + ".conv1.bias", cout);
        r.n2g = f32(pre + ".norm2.weight", cout); r.n2b = f32(pre + ".norm2.bias", cout);
        r.w2 = mf.add("conv3", pre + ".conv2.weight", (size_t)cout * cout * 9 * 2); r.b2 = f32(pre + ".conv2.bias", cout);
        if (r.skip) { r.wsk = bf(pre + ".nin_shortcut.weight", (size_t)cout * cin); r.bsk = f32(pre + ".nin_shortcut.bias", cout); }
        return r;
    };
    auto add_attn = [&](const std::string& p, int ch) {
        VqAttn a{}; a.c = ch;
        a.ng = f32(p + ".norm.weight", ch); a.nb = f32(p + ".norm.bias", ch);
        a.wq = bf(p + ".q.weight", (size_t)ch * ch); a.bq = f32(p + ".q.bias", ch);
        a.wk = bf(p + ".k.weight", (size_t)ch * ch); a.bk = f32(p + ".k.bias", ch);
        a.wv = bf(p + ".v.weight", (size_t)ch * ch); a.bv = f32(p + ".v.bias", ch);
        a.wo = bf(p + ".proj_out.weight", (size_t)ch * ch); a.bo = f32(p + ".proj_out.bias", ch);
        return a;
    };
    v.cinw = f32("encoder.conv_in.weight", (size_t)c.ch * c.out_ch * 9); v.cinb = f32("encoder.conv_in.bias", c.ch);
    v.down.assign(c.n_ch_mult, {}); v.down_attn.assign(c.n_ch_mult, {}); v.downsample.assign(c.n_ch_mult, ConvW{});
    int bin = c.ch, curr_res = c.resolution;
    for (int lvl = 0; lvl < c.n_ch_mult; lvl++) {
        const int bout = c.ch * c.ch_mult[lvl];
        bool at = false;
        for (int i = 0; i < c.n_attn_resolutions; i++) at = at || c.attn_resolutions[i] == curr_res;
        for (int i = 0; i < c.num_res_blocks; i++) {
            char pre[64]; snprintf(pre, sizeof pre, "encoder.down.%d.block.%d", lvl, i);
            v.down[lvl].push_back(add_res(pre, bin, bout)); bin = bout;
            if (at) { snprintf(pre, sizeof pre, "encoder.down.%d.attn.%d", lvl, i); v.down_attn[lvl].push_back(add_attn(pre, bin)); }
        }
        if (lvl != c.n_ch_mult - 1) {
            char pre[64]; snprintf(pre, sizeof pre, "encoder.down.%d.downsample.conv", lvl);
            ConvW d{}; d.c = bin; d.w = mf.add("conv3", std::string(pre) + ".weight", (size_t)bin * bin * 9 * 2);
            d.b = f32(std::string(pre) + ".bias", bin); v.downsample[lvl] = d;
            curr_res /= 2;
        }
    }
    v.mid1 = add_res("encoder.mid.block_1", bin, bin);
    if (c.mid_attn) v.attn = add_attn("encoder.mid.attn_1", bin);
    v.mid2 = add_res("encoder.mid.block_2", bin, bin);
    v.noutg = f32("encoder.norm_out.weight", bin); v.noutb = f32("encoder.norm_out.bias", bin);
    v.coutw = f32("encoder.conv_out.weight", (size_t)c.z_channels * bin * 9); v.coutb = f32("encoder.conv_out.bias", c.z_channels);
    v.qw = f32("quant_conv.weight", (size_t)c.embed_dim * c.z_channels); v.qb = f32("quant_conv.bias", c.embed_dim);
}

// ------------------------------------------------------------------------------------ CLIP description
struct ClipBlk { size_t ln1g, ln1b, wqkv, bqkv, wo, bo, ln2g, ln2b, wfc, bfc, wpj, bpj; };
struct ClipModel {
    rdm_clip_cfg cfg{}; bool loaded = false;
    std::vector<ClipBlk> text, vis;
    size_t tok, pos, lnfg, lnfb, tproj;                               // text
    size_t conv1, cls, vpos, lnpreg, lnpreb, lnpostg, lnpostb, vproj;  // vision
    char* blob = nullptr; size_t blob_bytes = 0; Arena arena;
};
static void build_clip(ClipModel& m, const rdm_clip_cfg& c, Manifest& mf) {
    m.cfg = c; m.text.clear(); m.vis.clear();
    auto f32 = [&](const std::string& n, size_t numel) { return mf.add("f32", n, numel * 4); };
    auto bf = [&](const std::string& n, size_t numel) { return mf.add("bf16", n, numel * 2); };
    auto tower = [&](std::vector<ClipBlk>& out, const std::string& pre, int w, int layers) {
        for (int i = 0; i < layers; i++) {
            char b[96]; snprintf(b, sizeof b, "%s.resblocks.%d", pre.c_str(), i); const std::string p = b;
            ClipBlk k{};
            k.ln1g = f32(p + ".ln_1.weight", w); k.ln1b = f32(p + ".ln_1.bias", w);
            k.wqkv = bf(p + ".attn.in_proj_weight", (size_t)3 * w * w); k.bqkv = f32(p + ".attn.in_proj_bias", 3 * w);
            k.wo = bf(p + ".attn.out_proj.weight", (size_t)w * w); k.bo = f32(p + ".attn.out_proj.bias", w);
            k.ln2g = f32(p + ".ln_2.weight", w); k.ln2b = f32(p + ".ln_2.bias", w);
            k.wfc = bf(p + ".mlp.c_fc.weight", (size_t)4 * w * w); k.bfc = f32(p + ".mlp.c_fc.bias", 4 * w);
            k.wpj = bf(p + ".mlp.c_proj.weight", (size_t)4 * w * w); k.bpj = f32(p + ".mlp.c_proj.bias", w);
            out.push_back(k);
        }
    };
    const int vw = c.vision_width, g = c.image_resolution / c.vision_patch_size, tw = c.transformer_width;
    m.conv1 = bf("visual.conv1.weight", (size_t)vw * 3 * c.vision_patch_size * c.vision_patch_size);
    m.cls = f32("visual.class_embedding", vw); m.vpos = f32("visual.positional_embedding", (size_t)(g * g + 1) * vw);
    m.lnpreg = f32("visual.ln_pre.weight", vw); m.lnpreb = f32("visual.ln_pre.bias", vw);
    tower(m.vis, "visual.transformer", vw, c.vision_layers);
    m.lnpostg = f32("visual.ln_post.weight", vw); m.lnpostb = f32("visual.ln_post.bias", vw);
    m.vproj = mf.add("bf16_t", "visual.proj", (size_t)vw * c.embed_dim * 2);
    tower(m.text, "transformer", tw, c.transformer_layers);
    m.tok = f32("token_embedding.weight", (size_t)c.vocab_size * tw);
    m.pos = f32("positional_embedding", (size_t)c.context_length * tw);
    m.lnfg = f32("ln_final.weight", tw); m.lnfb = f32("ln_final.bias", tw);
    m.tproj = mf.add("bf16_t", "text_projection", (size_t)tw * c.embed_dim * 2);
}

// ------------------------------------------------------------------------------------ RARM transformer description
struct RarmBlk { size_t ln1g, ln1b, wqkv, wo1, bo1, ln2g, ln2b, wq2, wo2, bo2, ln3g, ln3b, wff1, bff1, wff2, bff2; };
struct RarmModel {
    rdm_rarm_cfg cfg{}; bool loaded = false; int C = 0, kv_total = 0;
    std::vector<RarmBlk> blk;
    size_t emb, pos, kvw, wpo, bpo;
    char* blob = nullptr; size_t blob_bytes = 0; Arena arena;
    // sampling state: per-layer self-attention K/V cache [depth][2][B'][L][C], projected neighbours [B'*k][depth*2*C],
    // device step counter / completion counter / current tokens, logits of the current step
    char* cache = nullptr; size_t cache_bytes = 0;
    char* ctxkv = nullptr; size_t ctxkv_bytes = 0;
    char* state = nullptr; size_t state_bytes = 0;
    // decode-step cross-attention operands per layer (rarm_prepare): [depth][2][Bc][128][C] bf16 (G, UT), valid for xa_B conditional sequences and xa_k neighbours
    char* xa = nullptr; size_t xa_bytes = 0; int xa_B = 0, xa_k = 0;
    char* xws = nullptr; size_t xws_bytes = 0;          // split decode cross-attention: [B2][4][C] {fp32, epoch} granules, then B2 arrival counters
    unsigned xepoch = 0;                                // its launch tag (rarm.hip): incremented per launch
};
static void build_rarm(RarmModel& m, const rdm_rarm_cfg& c, Manifest& mf) {
    m.cfg = c; m.blk.clear(); m.C = c.n_heads * c.d_head;
    const int C = m.C;
    auto f32 = [&](const std::string& n, size_t numel) { return mf.add("f32", n, numel * 4); };
    auto bf = [&](const std::string& n, size_t numel) { return mf.add("bf16", n, numel * 2); };
    m.emb = f32("proj_in.weight", (size_t)c.vocab_in * C);
    m.pos = mf.add("f32_t", "positional_encoding", (size_t)C * c.sequence_length * 4);
    std::string kv_srcs;
    for (int i = 0; i < c.depth; i++) {
        char b[64]; snprintf(b, sizeof b, "transformer_blocks.%d", i); const std::string p = b;
        RarmBlk k{};
        k.ln1g = f32(p + ".norm1.weight", C); k.ln1b = f32(p + ".norm1.bias", C);
        k.wqkv = bf(p + ".attn1.to_q.weight," + p + ".attn1.to_k.weight," + p + ".attn1.to_v.weight", (size_t)3 * C * C);
        k.wo1 = bf(p + ".attn1.to_out.0.weight", (size_t)C * C); k.bo1 = f32(p + ".attn1.to_out.0.bias", C);
        k.ln2g = f32(p + ".norm2.weight", C); k.ln2b = f32(p + ".norm2.bias", C);
        k.wq2 = bf(p + ".attn2.to_q.weight", (size_t)C * C);
        k.wo2 = bf(p + ".attn2.to_out.0.weight", (size_t)C * C); k.bo2 = f32(p + ".attn2.to_out.0.bias", C);
        k.ln3g = f32(p + ".norm3.weight", C); k.ln3b = f32(p + ".norm3.bias", C);
        k.wff1 = mf.add("geglu_w", p + ".ff.net.0.proj.weight", (size_t)8 * C * C * 2);
        k.bff1 = mf.add("geglu_b", p + ".ff.net.0.proj.bias", (size_t)8 * C * 4);
        k.wff2 = bf(p + ".ff.net.2.weight", (size_t)C * 4 * C); k.bff2 = f32(p + ".ff.net.2.bias", C);
        if (!kv_srcs.empty()) kv_srcs += ",";
        kv_srcs += p + ".attn2.to_k.weight," + p + ".attn2.to_v.weight";
        m.blk.push_back(k);
    }
    m.kv_total = c.depth * 2 * C;
    m.kvw = mf.add("bf16", kv_srcs, (size_t)m.kv_total * c.context_dim * 2);      // the neighbours' K/V of ALL layers: one GEMM per sampling call
    m.wpo = bf("proj_out.weight", (size_t)c.vocab_out * C); m.bpo = f32("proj_out.bias", c.vocab_out);
}

// ------------------------------------------------------------------------------------ context
struct rdm_ctx {
    int device = 0; hipStream_t stream = nullptr; char err[512] = {0};
    // side stream for work that is independent of the main chain (the ResBlocks' skip_connection GEMM, round 6): forked / joined by events
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; hipStream_t side_saved = nullptr;
    void* zero_page = nullptr; float* eye3 = nullptr;
    UNet unet; VqModel vq; VqEncModel vqenc; ClipModel clip; RarmModel rarm; KnnDb db;
    float* gn_partial = nullptr; size_t gn_partial_bytes = 0;
    char* splitk_ws = nullptr; size_t splitk_ws_bytes = 0;   // fp32 partial planes of the K-split halo convs
    char* samp = nullptr; size_t samp_bytes = 0;     // sampler scratch
    // derived weight layouts, built on first use per weight and dropped when a model is reloaded: fragment-ordered copies of the 3x3 conv
    // weights (conv_halo4.hip) and of the Linear / 1x1 weights (lin4.hip; optionally scaled by a LayerNorm's gamma, with the (s, b') table
    // of the folded LayerNorm beside it).  Keyed on everything the copy depends on -- the entry is the copy of exactly that
    // (weight, shape, kind, gamma): two weights can never alias one entry.
    struct FragKey {
        const void* W; int N, K, kind; const void* aux;       // kind: 0 conv3x3, 1 linear, 2 linear GEGLU-ordered, 3 / 4 = 1 / 2 with LayerNorm folded in (aux = gamma)
        bool operator==(const FragKey& o) const { return W == o.W && N == o.N && K == o.K && kind == o.kind && aux == o.aux; }
    };
    struct FragKeyHash {
        size_t operator()(const FragKey& k) const {
            size_t h = std::hash<const void*>()(k.W);
            for (size_t v : {(size_t)k.N, (size_t)k.K, (size_t)k.kind, (size_t)(uintptr_t)k.aux}) h = (h ^ v) * 0x9E3779B97F4A7C15ull + (h >> 29);
            return h;
        }
    };
    struct FragVal { bf16_t* frag; float* sb; };
    std::unordered_map<FragKey, FragVal, FragKeyHash> wfrag;
    char* bwd_tmp = nullptr; size_t bwd_tmp_bytes = 0;          // scratch of the backward ops (backward.hip)
    char* wfrag_tmp = nullptr; size_t wfrag_tmp_bytes = 0;      // rdm_op_conv3x3 / rdm_op_linear: caller-owned weights are re-packed per call
    void drop_frags() { for (auto& kv : wfrag) { (void)hipFree(kv.second.frag); if (kv.second.sb) (void)hipFree(kv.second.sb); } wfrag.clear(); }
    // (the copies are packed on `stream`; rdm_set_stream synchronises the old stream before it installs another one, so a copy is
    //  complete before any other stream can launch a kernel that reads it)
    const bf16_t* frag_for(const bf16_t* W, int N, int Cin) {
        const FragKey key{W, N, Cin, 0, nullptr};
        auto it = wfrag.find(key);
        if (it != wfrag.end()) return it->second.frag;
        bf16_t* d = nullptr;
        if (hipMalloc((void**)&d, (size_t)N * 9 * Cin * 2) != hipSuccess) return nullptr;
        if (launch_conv_w_fragpack(W, d, N, Cin, stream) != hipSuccess) { (void)hipFree(d); return nullptr; }
        wfrag[key] = FragVal{d, nullptr};
        return d;
    }
    const bf16_t* phase_weights_for(const bf16_t* W, int N, int Cin) {           // [4][N][2][2][Cin] of a fused-upsample conv (igemm.hip CONV == 3)
        const FragKey key{W, N, Cin, 5, nullptr};
        auto it = wfrag.find(key);
        if (it != wfrag.end()) return it->second.frag;
        bf16_t* d = nullptr;
        if (hipMalloc((void**)&d, (size_t)16 * N * Cin * 2) != hipSuccess) return nullptr;
        if (launch_conv_phase_weights(W, d, N, Cin, stream) != hipSuccess) { (void)hipFree(d); return nullptr; }
        wfrag[key] = FragVal{d, nullptr};
        return d;
    }
    const bf16_t* frag_for_lin(const bf16_t* W, int N, int K, int geglu) {       // fragment-ordered copy of a Linear / 1x1 weight (lin4.hip)
        const FragKey key{W, N, K, geglu ? 2 : 1, nullptr};
        auto it = wfrag.find(key);
        if (it != wfrag.end()) return it->second.frag;
        bf16_t* d = nullptr;
        if (hipMalloc((void**)&d, (size_t)N * K * 2) != hipSuccess) return nullptr;
        if (launch_lin_w_fragpack(W, d, N, K, K, geglu, stream) != hipSuccess) { (void)hipFree(d); return nullptr; }
        wfrag[key] = FragVal{d, nullptr};
        return d;
    }
    // [2][C] fp32 = {0, b}: the rowvec of attn1.to_out over a guided batch's [conditional | unconditional] halves (unet_body: the
    // unconditional rows' cross-attention is exactly attn2.to_out's bias b, which then rides in attn1.to_out's start values)
    const float* zero_bias_pair(const float* b, int C) {
        const FragKey key{b, C, 2, 6, nullptr};
        auto it = wfrag.find(key);
        if (it != wfrag.end()) return it->second.sb;
        float* t = nullptr;
        if (hipMalloc((void**)&t, (size_t)2 * C * sizeof(float)) != hipSuccess) return nullptr;
        if (hipMemsetAsync(t, 0, (size_t)C * sizeof(float), stream) != hipSuccess ||
            hipMemcpyAsync(t + C, b, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) { (void)hipFree(t); return nullptr; }
        wfrag[key] = FragVal{nullptr, t};
        return t;
    }
    // ... with LayerNorm(gamma, beta) folded in: the copy holds bf16(gamma[k] W[n][k]), *sb the (s, b') table (lin4.hip: lin_ln_sb_kernel)
    const bf16_t* frag_for_lin_ln(const bf16_t* W, int N, int K, int geglu, const float* gamma, const float* beta, const float* bias, const float** sb) {
        const FragKey key{W, N, K, geglu ? 4 : 3, gamma};
        auto it = wfrag.find(key);
        if (it != wfrag.end()) { *sb = it->second.sb; return it->second.frag; }
        bf16_t* d = nullptr; float* t = nullptr;
        if (hipMalloc((void**)&d, (size_t)N * K * 2) != hipSuccess) return nullptr;
        if (hipMalloc((void**)&t, (size_t)N * 2 * sizeof(float)) != hipSuccess) { (void)hipFree(d); return nullptr; }
        if (launch_lin_w_fragpack(W, d, N, K, K, geglu, stream, gamma) != hipSuccess || launch_lin_ln_sb(W, gamma, beta, bias, t, N, K, stream) != hipSuccess) {
            (void)hipFree(d); (void)hipFree(t); return nullptr;
        }
        wfrag[key] = FragVal{d, t};
        *sb = t;
        return d;
    }
    // batch-invariant execution (rdm_set_deterministic / env RDM_DETERMINISTIC): every kernel-selection decision (skinny vs tiled GEMM,
    // halo vs generic conv, conv split-K, the zero-context shortcut) is a function of the PER-SAMPLE layer shape only, so a row's
    // result is bitwise independent of the batch it sits in and of the number of ranks the batch is sharded over
    bool deterministic = getenv("RDM_DETERMINISTIC") ? atoi(getenv("RDM_DETERMINISTIC")) != 0 : false;
    // debug tap (rdm_debug_tap): the output activation of top-level UNet block `tap_block` (NHWC bf16) is copied into tap_buf by the next forward
    void* tap_buf = nullptr; size_t tap_bytes = 0; int tap_block = -1, tap_sub = 0;      // tap_sub: 0 = the block's output, else 16 * layer-in-block + stage (unet_body)
    // RCCL communicator (rdm_comm_*): library handle from dlopen, function table, communicator
    void* rccl_lib = nullptr; void* comm = nullptr; int comm_world = 0;
    // optional per-launch HIP-event profiler for the GEMM-class kernels (bench.py roofline)
    unsigned prof = 0;           // bit k set: record HIP events around launches of kind k (RDM_PROF_* in rdm_hip.h)
    struct ProfRec { hipEvent_t a, b; int kind; double flops; const char* tag; int d0, d1, d2; };      // tag: the op's role in the graph (string literal), d*: its shape
    std::vector<ProfRec> prof_recs; std::vector<hipEvent_t> prof_pool;
    hipEvent_t prof_event() {
        if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
        hipEvent_t e; hipEventCreate(&e); return e;
    }
    int fail(int code, const char* fmt, ...) {
        va_list ap; va_start(ap, fmt); vsnprintf(err, sizeof err, fmt, ap); va_end(ap); return code;
    }
};

static int ensure_bytes(rdm_ctx* c, char** p, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*p) { RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream)); RDM_CHECK_HIP(c, hipFree(*p)); *p = nullptr; *have = 0; }
    RDM_CHECK_HIP(c, hipMalloc((void**)p, need));
    *have = need; return 0;
}

// ------------------------------------------------------------------------------------ op helpers
struct Ops {
    rdm_ctx* c; Arena* ar; const char* blob; bool plan; int rc = 0;
    template <typename T> const T* w(size_t off) const { return (const T*)(blob + off); }
    bf16_t* abf(size_t n) { return (bf16_t*)ar->alloc(n * 2); }
    float* af32(size_t n) { return (float*)ar->alloc(n * 4); }
    void check(hipError_t e, const char* what) {
        if (e != hipSuccess && rc == 0) rc = c->fail(-3, "%s: %s", what, hipGetErrorString(e));
    }
    // ---- side stream (round 6).  A ResBlock's skip_connection 1x1 conv depends on the block's INPUT only, while the main chain runs
    // GroupNorm -> conv1 -> GroupNorm before conv2 consumes it as the residual.  Issued on a second (non-blocking) stream at the top of the
    // block it fills the CUs the persistent conv kernels leave idle in their last, partial round of tiles (16 x 16 level: 384 tiles on
    // 256 CUs) instead of taking its own slot in the serial chain.  Same kernels, same arithmetic: only the issue order changes.
    // Opt-in, RDM_SKIP_OVERLAP=1 (resblock): two same-box A/Bs disagree in sign.  side_begin(): launches that follow go to the side stream (ordered after everything issued so far).
    bool side_begin() {
        // (not while kernel classes other than the conv are bracketed with events: concurrent side work would be billed to whatever runs beside it)
        if (plan || (c->prof & ~(1u << RDM_PROF_CONV3X3))) return false;
        if (!c->side) {
            if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) { c->side = nullptr; return false; }     // (a low-priority stream measured the same: profiles/r06_conv_tail_split_normal_priority_ab.log)
            if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
                (void)hipStreamDestroy(c->side); c->side = nullptr; return false;
            }
        }
        check(hipEventRecord(c->ev_fork, c->stream), "side stream fork");
        check(hipStreamWaitEvent(c->side, c->ev_fork, 0), "side stream fork");
        c->side_saved = c->stream; c->stream = c->side;
        return true;
    }
    void side_end() {                                           // back to the main stream; the side work is still in flight
        check(hipEventRecord(c->ev_join, c->stream), "side stream join");
        c->stream = c->side_saved;
    }
    void side_join() { check(hipStreamWaitEvent(c->stream, c->ev_join, 0), "side stream join"); }     // main stream: wait for the side work
    IgemmParams base(int M, int N, int K) {
        IgemmParams p{}; p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.ldo = N; p.zero_page = c->zero_page;
        p.Hin = p.Win = p.Hout = p.Wout = 1; p.stride = 1; p.rows_per_sample = 1; return p;
    }
    // out[M,N] = act(A[M,K] W^T + bias) (+res)
    // a1_wrap_rows > 0: A1 holds that many rows only, row m reads m % a1_wrap_rows (lin4 only: callers check lin4_takes first)
    bool lin4_takes(int M, int N, int C0, int C1, int a1_wrap_rows, int res_wrap_rows = 0) {
        if (c->deterministic) return false;
        IgemmParams t = base(M, N, C0 + C1);
        t.C0 = C0; t.C1 = C1; t.W = (const bf16_t*)blob; t.Wfrag = t.W; t.out_bf16 = (bf16_t*)blob; t.a1_wrap_rows = a1_wrap_rows;
        if (res_wrap_rows > 0) { t.res_bf16 = (const bf16_t*)blob; t.res_wrap_rows = res_wrap_rows; }
        return lin4_supported(t, 1);
    }
    // rowvec / rowvec_ld / rv_rows: a per-row-group per-column add (IgemmParams::rowvec with rows_per_sample = rv_rows)
    void linear(const bf16_t* A0, const bf16_t* A1, int C0, int C1, size_t woff, size_t boff, bool has_bias, int M, int N,
                int act, const bf16_t* res, bf16_t* out, float* out_f32 = nullptr, const float* res_f32 = nullptr, int a1_wrap_rows = 0,
                const float* rowvec = nullptr, int rowvec_ld = 0, int rv_rows = 1, int res_wrap_rows = 0) {
        if (plan) return;
        // skinny (weight-streaming) kernel for decode-sized operands.  Fast mode: whenever M <= 128.  Deterministic mode: exactly for
        // the ops with ONE row per sample (`single_row`: time embedding, RARM decode step, CLIP projection), at any batch
        // (one-row-per-sample operands of bigger batches -- RARM decode at 128+ sequences per GPU -- keep the skinny kernel: its row
        //  blocks scale with M, while the tiled kernels would run a dozen 256-row tiles)
        // RDM_SGEMM_MAX_ROWS / RDM_SGEMM_GEGLU_MAX_ROWS (dev): one-row-per-sample operands beyond these row counts take the tiled kernels
        static const int sg_max = getenv("RDM_SGEMM_MAX_ROWS") ? atoi(getenv("RDM_SGEMM_MAX_ROWS")) : 4096;      // (2048 sequences: 716 -> 780 img/s against the tiled kernels, round 5)
        // (round 5, same box: the GEGLU projection of the RARM decode step through the tiled kernel from ~200 rows on: 397.8 -> 409.5 img/s at 256
        //  sequences, 487.0 -> 514.8 at 512; the plain projections through it: 221 / 305 -- their N = 768 gives the tiled kernel 16-24 tiles)
        static const int sg_geglu_max = getenv("RDM_SGEMM_GEGLU_MAX_ROWS") ? atoi(getenv("RDM_SGEMM_GEGLU_MAX_ROWS")) : 192;
        const bool skinny = c->deterministic ? single_row : (M <= 128 || (single_row && M <= (act == ACT_GEGLU ? sg_geglu_max : sg_max)));
        if (skinny && !A1 && C1 == 0 && !rowvec) {         // N/32 x ceil(M/32) blocks (sgemm.hip)
            SgemmParams q{}; q.A = A0; q.lda = C0; q.W = w<bf16_t>(woff); q.M = M; q.N = N; q.K = C0; q.bias = has_bias ? w<float>(boff) : nullptr;
            q.act = act; q.res_f32 = res_f32; q.res_bf16 = res; q.out_f32 = out_f32; q.out_bf16 = out; q.ldo = act == ACT_GEGLU ? N / 2 : N;
            q.fixed_split = c->deterministic ? 1 : 0;
            // 1536+ rows: LDS-staged 64 x 64 tiles (mgemm.hip) -- the skinny kernel's per-wave operand fetch is 75 MB through the L2 -> CU
            // fabric for a [2048 x 768] x [768 x 768] product (33.6 us; 15.5 there).  Not in deterministic mode (the kernel choice would follow the batch).
            static const int mg_from = getenv("RDM_MGEMM_FROM") ? atoi(getenv("RDM_MGEMM_FROM")) : 1536;
            if (!c->deterministic && single_row && mg_from > 0 && M >= mg_from && act != ACT_GEGLU && mgemm_supported(q)) {
                prof_begin(RDM_PROF_LINEAR, 2.0 * M * N * (double)C0, M, N, C0);
                check(launch_mgemm(q, c->stream), "mid-size linear");
                prof_end();
                return;
            }
            if (sgemm_supported(q)) {
                prof_begin(RDM_PROF_LINEAR, 2.0 * M * N * (double)C0, M, N, C0);
                check(launch_sgemm(q, c->stream), "skinny linear");
                prof_end();
                return;
            }
        }
        IgemmParams p = base(M, N, C0 + C1);
        p.A0 = A0; p.A1 = A1; p.C0 = C0; p.C1 = C1; p.W = w<bf16_t>(woff); p.bias = has_bias ? w<float>(boff) : nullptr;
        p.act = act; p.res_bf16 = res; p.res_f32 = res_f32; p.out_bf16 = out; p.out_f32 = out_f32; p.a1_wrap_rows = a1_wrap_rows;
        p.res_wrap_rows = res_wrap_rows;
        if (rowvec) { p.rowvec = rowvec; p.rowvec_ld = rowvec_ld; p.rows_per_sample = rv_rows; }
        if (act == ACT_GEGLU) p.ldo = N / 2;
        // one-wave-per-SIMD kernel for the big-M projections (its tile choice follows M, so not in deterministic mode)
        // (deterministic mode: its use must not follow the batch -- exactly when the rows of ONE sample fill whole 128 / 256-row tiles,
        //  at any tile count; a row's sum order is the same in every tile position)
        {
            IgemmParams t = p; t.Wfrag = p.W;
            const int tile_rows = (N % 384 == 0) ? 128 : 256;
            const bool det_ok = rows_hint > 0 && rows_hint % tile_rows == 0;
            if (c->deterministic) t.l4_any_tiles = p.l4_any_tiles = 1;
            if ((!c->deterministic || det_ok) && lin4_supported(t, 1)) p.Wfrag = c->frag_for_lin(p.W, N, C0 + C1, act == ACT_GEGLU);
        }
        prof_begin(RDM_PROF_LINEAR, 2.0 * M * N * (double)(C0 + C1), M, N, C0 + C1);
        check(launch_igemm(p, false, 1, c->stream), "linear");
        prof_end();
    }
    // out = act(LayerNorm(x) W^T + bias) with the LayerNorm formed inside the skinny GEMM (sgemm.hip): decode-sized operands only.
    // false = not available for this shape (the caller runs layernorm + linear)
    bool linear_ln(const float* x, size_t g, size_t b, int C, size_t woff, size_t boff, bool has_bias, int M, int N, int act, bf16_t* out) {
        static const int off = getenv("RDM_NO_LNFUSE") ? atoi(getenv("RDM_NO_LNFUSE")) : 0;
        static const int ln_max_rows = getenv("RDM_SGEMM_LN_MAX_ROWS") ? atoi(getenv("RDM_SGEMM_LN_MAX_ROWS")) : 192;
        // (from ~200 rows on a separate LayerNorm pass + the 64 x 64-tile GEMM beats the LayerNorm-fused 32-row tiles: sgemm.hip)
        // (RDM_SGEMM_LN8_FROM=m: the eight-wave 64-row tiles take the LayerNorm themselves from m rows on -- measured slower, off: sgemm.hip)
        static const int ln8_from = getenv("RDM_SGEMM_LN8_FROM") ? atoi(getenv("RDM_SGEMM_LN8_FROM")) : 0;
        const bool ln8 = ln8_from > 0 && M >= ln8_from && act != ACT_GEGLU && C == 768;
        if (!c->deterministic && M > ln_max_rows && !ln8) return false;
        const bool skinny = c->deterministic ? single_row : (M <= 128 || (single_row && M <= 1024));
        SgemmParams q{}; q.ln_x = x; q.ln_g = w<float>(g); q.ln_b = w<float>(b); q.ln_eps = 1e-5f; q.W = w<bf16_t>(woff); q.M = M; q.N = N; q.K = C;
        q.bias = has_bias ? w<float>(boff) : nullptr; q.act = act; q.out_bf16 = out; q.ldo = act == ACT_GEGLU ? N / 2 : N;
        q.fixed_split = c->deterministic ? 1 : 0;
        if (off || !skinny || !sgemm_supported(q)) return false;
        if (plan) return true;
        prof_begin(RDM_PROF_LINEAR, 2.0 * M * N * (double)C, M, N, C);
        check(launch_sgemm(q, c->stream), "skinny linear on a LayerNorm");
        prof_end();
        return true;
    }
    // out = act(LayerNorm(x) W^T + bias) for the big-M projections, with the LayerNorm folded into lin4's GEMM (lin4.hip, <.., LN>): x is
    // the RAW bf16 tensor, the row statistics are taken inside the kernel.  false = not available for this shape / mode (the caller
    // runs layernorm + linear).  Clog: logical row width (zero padding beyond it).
    bool linear_ln_big(const bf16_t* x, size_t g, size_t b, int C, int Clog, size_t woff, size_t boff, bool has_bias, int M, int N, int act, bf16_t* out) {
        // OFF by default (round 4, measured on the headline bench, same box): 38.50 img/s with the fold against 38.75 without.  The
        // separate LayerNorm passes it removes are 42 ms of a 1650 ms step; the folded GEMMs cost 52 ms more -- every column block of a
        // row block repeats the row statistics (64 v_dot2_f32_bf16 per K-slice at ~11 cycles each beside the MFMAs: K loop + 30 %), and
        // the read-out gains 4 LDS reads + 16 FMAs per 8 outputs (GEGLU read-out + 50 %).  DESIGN.md section 8.  RDM_LNFOLD=1 enables it.
        static const int on = getenv("RDM_LNFOLD") ? atoi(getenv("RDM_LNFOLD")) : 0;
        if (!on || c->deterministic) return false;
        IgemmParams p = base(M, N, C);
        p.A0 = x; p.C0 = C; p.W = w<bf16_t>(woff); p.act = act; p.out_bf16 = out; if (act == ACT_GEGLU) p.ldo = N / 2;
        p.ln_inv_c = 1.0f / (float)Clog; p.ln_eps = 1e-5f;
        IgemmParams t = p; t.Wfrag = p.W; t.ln_sb = (const float*)blob;          // shape check only
        if (!lin4_supported(t, 1)) return false;
        if (plan) return true;
        p.Wfrag = c->frag_for_lin_ln(p.W, N, C, act == ACT_GEGLU, w<float>(g), w<float>(b), has_bias ? w<float>(boff) : nullptr, &p.ln_sb);
        if (!p.Wfrag) { if (rc == 0) rc = c->fail(-2, "out of memory for a LayerNorm-folded weight copy"); return true; }
        prof_begin(RDM_PROF_LINEAR, 2.0 * M * N * (double)C, M, N, C);
        check(launch_lin4(p, c->stream), "linear on a folded LayerNorm");
        prof_end();
        return true;
    }
    void conv3(const bf16_t* A0, const bf16_t* A1, int C0, int C1, size_t woff, size_t boff, int B, int Hin, int Win, int N,
               int stride, int ups, const float* rowvec, int rowvec_ld, const bf16_t* res, bf16_t* out, int asym = 0) {
        if (plan) return;
        const int Hout = ups ? Hin * 2 : (stride == 2 ? Hin / 2 : Hin), Wout = ups ? Win * 2 : (stride == 2 ? Win / 2 : Win);
        IgemmParams p = base(B * Hout * Wout, N, 9 * (C0 + C1));
        p.asym = asym;
        p.A0 = A0; p.A1 = A1; p.C0 = C0; p.C1 = C1; p.W = w<bf16_t>(woff); p.bias = w<float>(boff);
        p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout; p.stride = stride; p.ups = ups;
        p.rowvec = rowvec; p.rowvec_ld = rowvec_ld; p.rows_per_sample = Hout * Wout; p.res_bf16 = res; p.out_bf16 = out;
        // deterministic mode: the halo kernels need whole 256-pixel tiles, which at < 256 pixels per sample exist only for batches
        // that are multiples of 256 / HW -- there the generic implicit GEMM (another summation order) runs for EVERY batch; and no
        // split-K (its factor follows the tile count, i.e. the batch)
        // Upsample's conv by output phase: four 2 x 2-tap convs at source resolution on pre-summed weights, 2.25 x fewer FLOPs than the
        // nine taps at output resolution (igemm.hip CONV == 3).  RDM_NO_UPS_PHASE=1: the fused-upsample halo kernel as before.
        static const bool no_phase = getenv("RDM_NO_UPS_PHASE") != nullptr;
        if (ups && !no_phase && !A1 && C1 == 0 && C0 % 64 == 0 && N % 8 == 0 && !rowvec && !res && stride == 1) {
            const bf16_t* wp = c->phase_weights_for(p.W, N, C0);
            if (wp) {
                IgemmParams q = base(B * Hin * Win, N, 4 * C0);
                q.A0 = A0; q.C0 = C0; q.W = wp; q.bias = w<float>(boff); q.out_bf16 = out; q.phase2 = 1;
                q.Hin = Hin; q.Win = Win; q.Hout = Hout; q.Wout = Wout; q.stride = 1; q.rows_per_sample = Hin * Win;
                q.sA = 0; q.sW = (long long)N * 4 * C0; q.sO = 0;
                prof_begin(RDM_PROF_UPSCONV, 2.0 * 4.0 * q.M * N * (double)q.K, 4 * q.M, N, q.K);
                check(launch_igemm(q, true, 4, c->stream), "conv3x3 on a 2x upsample, by phase");
                prof_end();
                return;
            }
        }
        const bool det_generic = c->deterministic && ((Hout * Wout) % 256 != 0);
        const int ks = c->deterministic ? 1 : conv_halo_ksplit(p);
        if (ks > 1 && ensure_bytes(c, &c->splitk_ws, &c->splitk_ws_bytes, (size_t)ks * p.M * N * 4) == 0) { p.ksplit = ks; p.ws = (float*)c->splitk_ws; }
        if (!det_generic && (conv_halo_supported(p) || conv_halo4_strip_supported(p))) p.Wfrag = c->frag_for(p.W, N, C0 + C1);
        prof_begin(RDM_PROF_CONV3X3, 2.0 * p.M * N * (double)p.K, p.M, N, p.K);
        check(det_generic ? launch_igemm(p, true, 1, c->stream) : launch_conv3x3(p, c->stream), "conv3x3");
        prof_end();
    }
    // (Round 6, wave quantisation: a conv of q full rounds of the CUs plus a partial round -- the 16 x 16 level of a guided batch of 64 is 384
    //  tiles on 256 CUs -- was run as full rounds + a K-split remainder in three forms: tail rows on a second stream, halves handed out inside
    //  one launch, two launches on tile sub-ranges.  All parity-exact, all SLOWER (-1.0 .. -1.4 % on the headline), and the launch trace says
    //  why: under the package power cap a half-empty round is nearly half price -- 256 tiles take 140 us, 384 tiles 213 us: the 128
    //  remainder tiles cost 73 us on a chip that clocks up when half its CUs idle -- while 256 K halves + their fp32 planes and finisher cost
    //  118 us.  The code is in the history (commits "conv tail split", "conv_halo4 TAIL variant", "conv remainder split"), the numbers in
    //  profiles/r06_conv_tail_split_*.log, r06_conv_remainder_split_*.)
    int cur_block = -1, cur_layer = 0;           // position in the UNet's block table (debug tap)
    void tap(int stage, const void* ptr, size_t nbytes) {      // rdm_debug_tap: stage `stage` of layer cur_layer of block cur_block
        if (plan || !c->tap_buf || c->tap_block != cur_block || c->tap_sub != cur_layer * 16 + stage) return;
        if (nbytes > c->tap_bytes) nbytes = c->tap_bytes;
        check(hipMemcpyAsync(c->tap_buf, ptr, nbytes, hipMemcpyDeviceToDevice, c->stream), "debug tap");
    }
    bool single_row = false;     // set by callers around ops whose operand has one row per sample (see linear)
    int rows_hint = 0;           // rows per sample of the operand of the linear ops that follow (0 = unknown); set by the UNet block executors
    bool prof_open = false;
    const char* tag = "";        // role of the ops that follow in the graph ("st.proj_in", "res.conv1", ...): rdm_prof_dump groups by it
    void prof_begin(int kind, double work, int d0 = 0, int d1 = 0, int d2 = 0) {       // work: FLOPs (GEMM-class kinds) or bytes (bandwidth-class kinds)
        prof_open = (c->prof >> kind) & 1u;
        if (!prof_open) return;
        rdm_ctx::ProfRec r; r.a = c->prof_event(); r.b = c->prof_event(); r.kind = kind; r.flops = work; r.tag = tag; r.d0 = d0; r.d1 = d1; r.d2 = d2;
        hipEventRecord(r.a, c->stream); c->prof_recs.push_back(r);
    }
    void prof_end() { if (prof_open) hipEventRecord(c->prof_recs.back().b, c->stream); prof_open = false; }
    void groupnorm(const bf16_t* x0, const bf16_t* x1, int C0, int C1, int B, int HW, size_t g, size_t b, float eps, int silu,
                   bf16_t* out, int L0 = -1, int L1 = -1, int x1_bmod = 0, int x0_bmod = 0) {      // L0 / L1: logical channels of the (zero-padded) sources, default = all
        if (plan) return;
        GnParams p{}; p.x0 = x0; p.x1 = x1; p.C0 = C0; p.C1 = C1; p.HW = HW; p.B = B; p.groups = 32; p.x1_bmod = x1_bmod; p.x0_bmod = x0_bmod;
        p.L0 = L0 < 0 ? C0 : L0; p.L1 = L1 < 0 ? C1 : L1;
        int nchunk = HW / 64; if (nchunk < 1) nchunk = 1; if (nchunk > 32) nchunk = 32;
        p.nchunk = nchunk; p.partial = c->gn_partial; p.gamma = w<float>(g); p.beta = w<float>(b); p.eps = eps; p.silu = silu;
        p.out = out;
        prof_begin(RDM_PROF_GROUPNORM, (double)B * HW * (C0 + C1) * 4.0, B * HW, C0 + C1, silu);       // ALGORITHMIC bytes: one read + one write of the bf16 tensor (round 5: the one-pass kernel moves exactly these; the two-pass form of the 64 x 64 level reads twice)
        check(launch_groupnorm(p, c->stream), "groupnorm");
        prof_end();
    }
    // GroupNorm + SiLU + 3x3 conv to a few channels (UNet `out`, VQ decoder conv_out): one statistics pass + the fused MFMA head kernel
    // (misc.hip), or GroupNorm-apply into `tmp` + the VALU head conv where the fused kernel does not apply
    void head(const bf16_t* x, int B, int H, int W, int C, int Clog, size_t g, size_t b, float eps, size_t woff, size_t boff, int Cout,
              float* out, bf16_t* tmp, bf16_t* wp) {
        if (plan) return;
        static const int off = getenv("RDM_NO_HEADFUSE") ? atoi(getenv("RDM_NO_HEADFUSE")) : 0;
        HeadParams hp{}; hp.x = x; hp.B = B; hp.H = H; hp.W = W; hp.C = C; hp.groups = 32; hp.gamma = w<float>(g); hp.beta = w<float>(b); hp.eps = eps;
        hp.w = w<float>(woff); hp.wp = wp; hp.bias = w<float>(boff); hp.out = out; hp.Cout = Cout;
        int nchunk = H * W / 64; if (nchunk < 1) nchunk = 1; if (nchunk > 32) nchunk = 32;
        hp.partial = c->gn_partial; hp.nchunk = nchunk;
        if (!off && Clog == C && head_conv_supported(hp)) {
            GnParams p{}; p.x0 = x; p.C0 = C; p.HW = H * W; p.B = B; p.groups = 32; p.L0 = C; p.nchunk = nchunk; p.partial = c->gn_partial;
            prof_begin(RDM_PROF_GROUPNORM, (double)B * H * W * C * 2.0, B * H * W, C, 2);
            check(launch_gn_stats(p, c->stream), "head groupnorm statistics");
            prof_end();
            check(launch_head_conv(hp, c->stream), "head conv");
            return;
        }
        groupnorm(x, nullptr, C, 0, B, H * W, g, b, eps, 1, tmp, Clog, 0);
        check(launch_conv_out(tmp, w<float>(woff), w<float>(boff), out, B, H, W, C, Cout, c->stream), "conv_out");
    }
    void layernorm(const void* x, int in_f32, size_t g, size_t b, void* out, int out_f32, int M, int C, int Clog = -1) {
        if (plan) return;
        prof_begin(RDM_PROF_LAYERNORM, (double)M * C * ((in_f32 ? 4.0 : 2.0) + (out_f32 ? 4.0 : 2.0)), M, C, 0);
        check(launch_layernorm(x, in_f32, w<float>(g), w<float>(b), out, out_f32, M, C, 1e-5f, c->stream, Clog < 0 ? C : Clog), "layernorm");
        prof_end();
    }
};

static int ensure_gn_partial(rdm_ctx* c, int B) {
    const size_t need = (size_t)B * 32 * 64 * 2 * sizeof(float);
    return ensure_bytes(c, (char**)&c->gn_partial, &c->gn_partial_bytes, need);
}

// ------------------------------------------------------------------------------------ UNet forward
// kv: bf16 [B*k, kv_total] cross-attention keys/values for every SpatialTransformer (computed by unet_prepare_kv)
static void unet_compute_kv(Ops& o, UNet& u, const float* context, int B, int k, bf16_t* kv_out) {
    const int cd = u.cfg.context_dim;
    bf16_t* cb = o.abf((size_t)B * k * cd);
    if (!o.plan) o.check(launch_cast_f32_bf16(context, cb, (long long)B * k * cd, o.c->stream), "cast ctx");
    o.linear(cb, nullptr, cd, 0, u.kvw, 0, false, B * k, u.kv_total, ACT_NONE, nullptr, kv_out);
}

// ---- cross-attention over k neighbours as two skinny GEMMs.  softmax(q K^T / sqrt d) V W_o^T with q = x W_q^T is re-associated
// per sample:  scores = x G_b^T with G_b[(h,j), :] = (K_bj restricted to head h) W_q / sqrt d   -> [heads*k <= 128 columns]
//              out    = P U_b^T  with U_b[:, (h,j)] = W_o (V_bj restricted to head h)            -> K = 128
// G_b and U_b depend only on the conditioning (computed once per sampling call next to the K/V cache).  The per-forward work
// drops from two C x C projections + an attention kernel (9 tensor passes) to N = 128 / K = 128 GEMMs (3.3 passes); identical
// in exact arithmetic to rdm/modules/attention.py:52-72 (CrossAttention.forward).
static bool xattn_skinny_ok(const UNet& u, int k) {
    static const int off = getenv("RDM_NO_XSKINNY") ? atoi(getenv("RDM_NO_XSKINNY")) : 0;
    if (off || !(k == 1 || k == 2 || k == 4)) return false;
    for (const StW& s : u.st) if (s.heads * k > XA_NP || s.c != s.heads * 32) return false;
    return !u.st.empty();
}
static void unet_compute_xattn(Ops& o, UNet& u, const bf16_t* kv, int B, int k, bf16_t* xa) {
    int cmax = 0;
    for (const StW& s : u.st) cmax = s.c > cmax ? s.c : cmax;
    bf16_t* kexp = o.abf((size_t)B * XA_NP * cmax);
    bf16_t* vexp = o.abf((size_t)B * XA_NP * cmax);
    bf16_t* wqt = o.abf((size_t)cmax * cmax);
    if (o.plan) return;
    for (const StW& s : u.st) {
        const int C = s.c;
        bf16_t* G = xa + (size_t)B * s.xa_unit;                       // [B][NP][C]
        bf16_t* U = G + (size_t)B * XA_NP * C;                        // [B][C][NP]
        o.check(launch_expand_heads(kv + s.kv_off, u.kv_total, B, k, s.heads, 32, XA_NP, 1.0f / sqrtf(32.f), kexp, o.c->stream), "expand K");
        o.check(launch_expand_heads(kv + s.kv_off + C, u.kv_total, B, k, s.heads, 32, XA_NP, 1.0f, vexp, o.c->stream), "expand V");
        o.check(launch_transpose_bf16(o.w<bf16_t>(s.wq2), wqt, C, C, o.c->stream), "transpose Wq");
        {   // G = Kexp . Wq   (contract over Wq's ROW index: weights operand = Wq^T)
            IgemmParams p = o.base(B * XA_NP, C, C);
            p.A0 = kexp; p.C0 = C; p.W = wqt; p.out_bf16 = G;
            o.check(launch_igemm(p, false, 1, o.c->stream), "xattn G");
        }
        {   // U_b = Wo . Vexp_b^T  per sample (A shared)
            IgemmParams p = o.base(C, XA_NP, C);
            p.A0 = o.w<bf16_t>(s.wo2); p.C0 = C; p.W = vexp; p.sA = 0; p.sW = (long long)XA_NP * C; p.sO = (long long)C * XA_NP;
            p.out_bf16 = U; p.ldo = XA_NP;
            o.check(launch_igemm(p, false, B, o.c->stream), "xattn U");
        }
        o.check(launch_xattn_pack(G, U, U + (size_t)B * C * XA_NP, U + (size_t)B * C * XA_NP + (size_t)B * XA_NP * C, B, XA_NP, C, o.c->stream), "xattn pack");
    }
}

// time embedding (openaimodel.py:352-353) and all 22 ResBlock emb_layers of it as ONE row of u.emb_total floats per timestep row:
// emb is only ever consumed through SiLU (ResBlock.emb_layers[0]), so SiLU is folded into the two MLP outputs
static void unet_time_rows(Ops& o, UNet& u, const long long* t, int B, float* emb_all) {
    const int mcl = u.cfg.model_channels, mc = pad64(mcl), ted = mcl * 4;
    bf16_t* temb = o.abf((size_t)B * mc);
    if (!o.plan) o.check(launch_timestep_embedding(t, temb, B, mcl, mc, o.c->stream), "timestep_embedding");
    bf16_t* e1 = o.abf((size_t)B * ted);
    o.single_row = true;                                  // one row per sample (see Ops::linear)
    o.tag = "time_embed";
    o.linear(temb, nullptr, mc, 0, u.te0w, u.te0b, true, B, ted, ACT_SILU, nullptr, e1);
    bf16_t* semb = o.abf((size_t)B * ted);
    o.linear(e1, nullptr, ted, 0, u.te2w, u.te2b, true, B, ted, ACT_SILU, nullptr, semb);
    o.linear(semb, nullptr, ted, 0, u.embw, u.embb, true, B, u.emb_total, ACT_NONE, nullptr, nullptr, emb_all);      // all 22 emb_layers in one GEMM
    o.single_row = false;
}

// emb_row: the batch shares ONE timestep whose emb row was computed ahead of the sampling loop (rdm_ddim_sample: a table of the S rows, one
// launch set per call instead of three GEMMs per forward); null: the rows are formed here from t
static void unet_body(Ops& o, UNet& u, const float* x, const long long* t, const bf16_t* kv, const bf16_t* xa, const int Bfull, int k, int H, int W,
                      float* eps_out, int Bx /* samples [Bx, Bfull) have all-zero context */, int Bshared /* Bfull, or Bfull/2: see below */,
                      const float* emb_row = nullptr) {
    int B = Bfull;               // the batch the CURRENT layer runs on (Bshared inside the guidance prefix)
    const rdm_unet_cfg& c = u.cfg;
    const int mcl = c.model_channels, mc = pad64(mcl);        // mc: padded width of the base level (see build_unet)
    const float* emb_all = emb_row;
    const int emb_ld = emb_row ? 0 : u.emb_total;           // row stride per sample: 0 = every sample reads the same row
    if (!emb_row) {
        float* e = o.af32((size_t)B * u.emb_total);
        unet_time_rows(o, u, t, B, e);
        emb_all = e;
    }

    struct Act { bf16_t* p; int C, H, W, L; bool half = false; };          // C: padded channels (row stride), L: logical channels; half: only samples [0, Bfull/2) exist (a shared-prefix skip tensor read with a batch wrap)
    std::vector<Act> hs;
    Act h{nullptr, 0, H, W, 0};
    // Shared guidance prefix: with classifier-free guidance the batch is [x | x] with the SAME x and t in both halves and different
    // contexts (ddim.py:229-234), so every layer before the first SpatialTransformer (conv_in, the 64x64 ResBlocks, the first
    // Downsample, the first 32x32 ResBlock: 12 % of the conv FLOPs, 17 % of the GroupNorm bytes) computes identical values for the two
    // halves.  They run once on Bfull/2 samples; the activations (and the skip tensors already pushed) are duplicated right before the
    // first context-dependent layer.  B below is the batch the CURRENT layer runs on.
    B = (Bshared > 0 && Bshared * 2 == Bfull) ? Bshared : Bfull;
    // (block outputs produced inside the prefix are allocated for Bfull samples and written into the first half, so leaving the
    //  prefix costs one copy of B samples per tensor: first half -> second half)
    auto expand = [&](Act& a) {        // [B, H, W, C] in a [2B, H, W, C] allocation -> second half = copy of the first
        const size_t n = (size_t)(Bfull / 2) * a.H * a.W * a.C;
        if (!o.plan) o.check(hipMemcpyAsync(a.p + n, a.p, n * 2, hipMemcpyDeviceToDevice, o.c->stream), "expand prefix");
        a.half = false;
    };
    // Skip tensors pushed inside the prefix are NOT duplicated when their readers can wrap the batch index instead (GroupNorm's and the
    // skip_connection GEMM's second source: 4 of the 5 copies, 0.24 ms of a 34.5 ms forward); resblock() materialises one on demand.
    static const int no_wrap = getenv("RDM_NO_SKIPWRAP") ? atoi(getenv("RDM_NO_SKIPWRAP")) : 0;

    auto resblock = [&](const ResW& r, const Act& a, Act* skip) -> Act {
        const int C0 = a.C, C1 = skip ? skip->C : 0, HW = a.H * a.W, M = B * HW;
        o.rows_hint = HW;
        int wrap_b = 0;                                    // > 0: the skip tensor holds Bfull/2 samples, read with a batch wrap
        if (skip && skip->half) {
            if (B == Bfull && r.skip && o.lin4_takes(M, r.cout, C0, C1, (Bfull / 2) * HW)) wrap_b = Bfull / 2;
            else expand(*skip);
        }
        const bf16_t* x1 = skip ? skip->p : nullptr;
        // skip_connection first, on the side stream (Ops::side_begin): it needs the block's input only
        const bf16_t* res = a.p;
        bf16_t* sk = nullptr; bool forked = false;
        if (r.skip) {
            sk = o.abf((size_t)M * r.cout);
            o.tag = "res.skip";
            static const int skip_on = getenv("RDM_SKIP_OVERLAP") ? atoi(getenv("RDM_SKIP_OVERLAP")) : 0;     // opt-in: + 0.45 % on one box, - 0.25 % on another (profiles/r06_skip_overlap_ab*.log)
            forked = skip_on && o.side_begin();
            o.linear(a.p, x1, C0, C1, r.wsk, r.bsk, true, M, r.cout, ACT_NONE, nullptr, sk, nullptr, nullptr, wrap_b * HW);
            if (forked) o.side_end();
            res = sk;
        }
        bf16_t* n1 = o.abf((size_t)M * r.cin);
        o.tag = "res.gn1";
        o.groupnorm(a.p, x1, C0, C1, B, HW, r.gn1g, r.gn1b, 1e-5f, 1, n1, a.L, skip ? skip->L : 0, wrap_b);
        o.tap(1, n1, (size_t)M * r.cin * 2);
        bf16_t* h1 = o.abf((size_t)M * r.cout);
        o.tag = "res.conv1";
        o.conv3(n1, nullptr, r.cin, 0, r.w1, r.b1, B, a.H, a.W, r.cout, 1, 0, emb_all + r.emb_off, emb_ld, nullptr, h1);
        o.tap(2, h1, (size_t)M * r.cout * 2);
        bf16_t* n2 = o.abf((size_t)M * r.cout);
        o.tag = "res.gn2";
        o.groupnorm(h1, nullptr, r.cout, 0, B, HW, r.gn2g, r.gn2b, 1e-5f, 1, n2, r.lout, 0);
        o.tap(3, n2, (size_t)M * r.cout * 2);
        if (forked) o.side_join();
        if (r.skip) o.tap(4, sk, (size_t)M * r.cout * 2);
        bf16_t* out = o.abf((size_t)Bfull * HW * r.cout);           // Bfull: see expand()
        o.tag = "res.conv2";
        o.conv3(n2, nullptr, r.cout, 0, r.w2, r.b2, B, a.H, a.W, r.cout, 1, 0, nullptr, 0, res, out);
        o.tap(5, out, (size_t)M * r.cout * 2);
        return Act{out, r.cout, a.H, a.W, r.lout};
    };
    auto transformer = [&](const StW& s, const Act& a) -> Act {
        const int C = s.c, n = a.H * a.W, M = B * n;
        o.rows_hint = n;
        // a.half: the input left the shared guidance prefix and only its first Bfull / 2 samples exist: its two readers -- the entry GroupNorm and
        // the residual of ff.net.2 x proj_out -- wrap the batch index (round 5: no 50 MB duplication pass per forward)
        const int in_wrap = a.half ? Bfull / 2 : 0;
        bf16_t* xn = o.abf((size_t)M * C);
        o.tag = "st.gn";
        o.groupnorm(a.p, nullptr, C, 0, B, n, s.gng, s.gnb, 1e-6f, 0, xn, s.lc, 0, 0, in_wrap);
        o.tap(1, xn, (size_t)M * C * 2);
        bf16_t* t0 = o.abf((size_t)M * C);
        o.tag = "st.proj_in";
        o.linear(xn, nullptr, C, 0, s.win, s.bin, true, M, C, ACT_NONE, nullptr, t0);
        o.tap(2, t0, (size_t)M * C * 2);
        // --- attn1 (self)
        bf16_t* l1 = o.abf((size_t)M * C);
        // n % 64 == 0: q | k | v in ONE projection (to_v's rows follow to_q | to_k in the blob, asserted in build_unet); the flash kernel
        // reads the token-major V block through transpose reads, so no per-layer V^T GEMM (6.8 % of the forward as a batched
        // weights-as-A GEMM at 380 TFLOP/s)
        static const int no_vrow = getenv("RDM_NO_VROW") ? atoi(getenv("RDM_NO_VROW")) : 0;
        const bool vrow = (n % 64 == 0) && !no_vrow && s.v_follows;
        const int QW = vrow ? 3 * C : 2 * C;
        bf16_t* qk = o.abf((size_t)M * QW);
        o.tag = "st.norm1+qkv";
        // norm1 folded into the q | k | v projection where lin4 takes it (only the fused form: the other paths read l1 again)
        if (!(vrow && o.linear_ln_big(t0, s.ln1g, s.ln1b, C, s.lc, s.wqk, 0, false, M, QW, ACT_NONE, qk))) {
            o.layernorm(t0, 0, s.ln1g, s.ln1b, l1, 0, M, C, s.lc);
            o.tap(3, l1, (size_t)M * C * 2);
            o.linear(l1, nullptr, C, 0, s.wqk, 0, false, M, QW, ACT_NONE, nullptr, qk);
        }
        o.tap(4, qk, (size_t)M * QW * 2);
        bf16_t* ao = o.abf((size_t)M * C);
        o.tag = "st.self_attention";
        if (vrow) {
            if (!o.plan) {
                FlashParams f{}; f.q = qk; f.ldq = QW; f.k = qk + C; f.ldk = QW; f.v = qk + 2 * C; f.ldv = QW; f.out = ao; f.ldo = C;
                f.n = n; f.C = C; f.scale_log2e = (1.0f / sqrtf(32.f)) * 1.4426950408889634f;
                o.prof_begin(RDM_PROF_ATTENTION, 4.0 * B * s.heads * (double)n * n * 32, B, n, C);
                o.check(launch_flash_d32(f, s.heads, B, o.c->stream), "flash attention");
                o.prof_end();
            }
        } else if (n % 32 == 0) {
            bf16_t* vt = o.abf((size_t)M * C);      // V^T per sample: [B][C][n] via swapped-operand GEMM
            if (!o.plan) {
                IgemmParams p = o.base(C, n, C);
                p.A0 = o.w<bf16_t>(s.wv); p.C0 = C; p.W = l1; p.sA = 0; p.sW = (long long)n * C; p.sO = (long long)C * n;
                p.out_bf16 = vt; p.ldo = n;
                o.check(launch_igemm(p, false, B, o.c->stream), "v^T gemm");
                FlashParams f{}; f.q = qk; f.ldq = 2 * C; f.k = qk + C; f.ldk = 2 * C; f.vt = vt; f.out = ao; f.ldo = C;
                f.n = n; f.C = C; f.scale_log2e = (1.0f / sqrtf(32.f)) * 1.4426950408889634f;
                o.prof_begin(RDM_PROF_ATTENTION, 4.0 * B * s.heads * (double)n * n * 32, B, n, C);
                o.check(launch_flash_d32(f, s.heads, B, o.c->stream), "flash attention");
                o.prof_end();
            }
        } else {
            bf16_t* v = o.abf((size_t)M * C);
            o.linear(l1, nullptr, C, 0, s.wv, 0, false, M, C, ACT_NONE, nullptr, v);
            if (!o.plan) {
                SmallAttnParams p{}; p.q = qk; p.ldq = 2 * C; p.k = qk + C; p.ldk = 2 * C; p.v = v; p.ldv = C; p.out = ao;
                p.ldo = C; p.nq = n; p.nkv = n; p.causal = 0; p.scale = 1.0f / sqrtf(32.f);
                o.check(launch_small_attention(p, 32, s.heads, B, o.c->stream), "small self attention");
            }
        }
        o.tap(5, ao, (size_t)M * C * 2);
        bf16_t* t1 = o.abf((size_t)M * C);
        // --- attn2 (cross over the k neighbours); samples >= Bx have all-zero neighbours: t2 = t1 + b_o exactly (see add_bias_rows_kernel)
        const int Mx = Bx * n;
        // Round 5: in a guided batch [conditional | unconditional] (Bx = B / 2) with the LayerNorm-fused cross-attention kernel,
        //   * the unconditional rows' t2 = attn1.to_out(...) + t0 + b_o2 leaves attn1.to_out's GEMM directly (b_o2 rides in the start
        //     values of those rows: Ops::linear's rowvec; one rounding less than t1 -> + b_o2), no add_bias_rows pass;
        //   * the cross-attention kernel runs IN PLACE on the conditional rows (t2 aliases t1) and emits norm3 of its finished rows, so
        //     the separate LayerNorm-3 pass only covers the unconditional rows.
        // RDM_NO_XFOLD=1: the separate passes as before.
        static const int no_xfold = getenv("RDM_NO_XFOLD") ? atoi(getenv("RDM_NO_XFOLD")) : 0;
        static const int no_xfused_env = getenv("RDM_NO_XFUSED") ? atoi(getenv("RDM_NO_XFUSED")) : 0;
        XattnParams xq{}; xq.rows = Mx; xq.n = n; xq.C = C; xq.NP = XA_NP; xq.ncols = s.heads * k; xq.group = k;
        const bool xfold_shape = xa && Mx > 0 && !no_xfused_env && s.lc == C && C <= 2048 && xattn_fused_supported(xq) && !no_xfold && !o.c->deterministic;       // => xfused && xln below
        const bool bias_fold = xfold_shape && Bx * 2 == B;             // the unconditional half exists and is exactly the second half
        o.tag = "st.attn1.to_out";
        if (bias_fold) {
            const float* zb = o.plan ? nullptr : o.c->zero_bias_pair(o.w<float>(s.bo2), C);
            if (!o.plan && !zb && o.rc == 0) o.rc = o.c->fail(-2, "out of memory for a bias table");
            o.linear(ao, nullptr, C, 0, s.wo1, s.bo1, true, M, C, ACT_NONE, t0, t1, nullptr, nullptr, 0, zb, C, Mx);
        } else {
            o.linear(ao, nullptr, C, 0, s.wo1, s.bo1, true, M, C, ACT_NONE, t0, t1);
        }
        o.tap(6, t1, (size_t)M * C * 2);          // (with the bias fold: the unconditional rows already hold t2)
        o.tag = "st.norm2+attn2";
        bf16_t* l2 = o.abf((size_t)M * C);
        // norm2 + attn2 + residual in one kernel when the neighbours' operands are cached (xa) and no channel is padding
        static const int no_xfused = getenv("RDM_NO_XFUSED") ? atoi(getenv("RDM_NO_XFUSED")) : 0;       // 1: two GEMMs; 2: fused without the LayerNorm
        XattnParams xp{};
        if (xa && Mx > 0) {
            const bf16_t* G = xa + (size_t)B * s.xa_unit; const bf16_t* U = G + (size_t)B * XA_NP * C;
            xp.x = l2; xp.G = U + (size_t)B * C * XA_NP; xp.U = xp.G + (size_t)B * XA_NP * C; xp.bias = o.w<float>(s.bo2); xp.res = t1; xp.out = nullptr;
            xp.rows = Mx; xp.n = n; xp.C = C; xp.NP = XA_NP; xp.ncols = s.heads * k; xp.group = k;
        }
        const bool xfused = xa && Mx > 0 && no_xfused != 1 && xattn_fused_supported(xp);
        const bool xln = xfused && no_xfused != 2 && s.lc == C && C <= 2048;
        if (Mx > 0 && !xln) o.layernorm(t1, 0, s.ln2g, s.ln2b, l2, 0, Mx, C, s.lc);
        bf16_t* t2 = xfold_shape ? t1 : o.abf((size_t)M * C);
        bf16_t* l3 = o.abf((size_t)M * C);
        if (Bx < B && !bias_fold && !o.plan)
            o.check(launch_add_bias_rows(t1 + (size_t)Mx * C, o.w<float>(s.bo2), t2 + (size_t)Mx * C, (long long)(M - Mx), C, o.c->stream), "zero-context cross attention");
        if (Mx == 0) {
        } else if (xa) {       // two skinny per-sample GEMMs (see unet_compute_xattn)
            bf16_t* P = o.abf((size_t)M * XA_NP);
            if (!o.plan) {
                const bf16_t* G = xa + (size_t)B * s.xa_unit; const bf16_t* U = G + (size_t)B * XA_NP * C;
                xp.out = t2;
                if (xln) { xp.x = t1; xp.res = nullptr; xp.ln_g = o.w<float>(s.ln2g); xp.ln_b = o.w<float>(s.ln2b); xp.ln_eps = 1e-5f; }
                if (xfold_shape && xln && xfused) { xp.ln3_g = o.w<float>(s.ln3g); xp.ln3_b = o.w<float>(s.ln3b); xp.ln3_out = l3; }
                if (xfused) {       // both GEMMs, the softmax and the residual (and norm2) in one launch (attention.hip)
                    o.prof_begin(RDM_PROF_LINEAR, 4.0 * Mx * XA_NP * (double)C, Mx, XA_NP, C);
                    o.check(launch_xattn_fused(xp, o.c->stream), "fused cross attention");
                    o.prof_end();
                } else {
                IgemmParams p = o.base(n, XA_NP, C);
                p.A0 = l2; p.C0 = C; p.sA = (long long)n * C; p.W = G; p.sW = (long long)XA_NP * C; p.out_bf16 = P; p.sO = (long long)n * XA_NP;
                p.act = ACT_SOFTMAXG; p.sm_group = k;
                o.prof_begin(RDM_PROF_LINEAR, 2.0 * Mx * XA_NP * (double)C);
                o.check(launch_igemm(p, false, Bx, o.c->stream), "xattn scores");
                o.prof_end();
                IgemmParams q = o.base(n, C, XA_NP);
                q.A0 = P; q.C0 = XA_NP; q.sA = (long long)n * XA_NP; q.W = U; q.sW = (long long)C * XA_NP; q.bias = o.w<float>(s.bo2);
                q.res_bf16 = t1; q.out_bf16 = t2; q.sO = (long long)n * C;
                o.prof_begin(RDM_PROF_LINEAR, 2.0 * Mx * C * (double)XA_NP);
                o.check(launch_igemm(q, false, Bx, o.c->stream), "xattn out");
                o.prof_end();
                }
            }
        } else {
            bf16_t* q2 = o.abf((size_t)M * C);
            o.linear(l2, nullptr, C, 0, s.wq2, 0, false, Mx, C, ACT_NONE, nullptr, q2);
            bf16_t* ao2 = o.abf((size_t)M * C);
            if (!o.plan) {
                SmallAttnParams p{}; p.q = q2; p.ldq = C; p.k = kv + s.kv_off; p.ldk = u.kv_total; p.v = kv + s.kv_off + C;
                p.ldv = u.kv_total; p.out = ao2; p.ldo = C; p.nq = n; p.nkv = k; p.causal = 0; p.scale = 1.0f / sqrtf(32.f);
                o.check(launch_small_attention(p, 32, s.heads, Bx, o.c->stream), "cross attention");
            }
            o.linear(ao2, nullptr, C, 0, s.wo2, s.bo2, true, Mx, C, ACT_NONE, t1, t2);
        }
        o.tap(7, t2, (size_t)M * C * 2);
        // --- GEGLU feed-forward
        o.tag = "st.norm3+geglu";
        const int FI = 4 * s.lc;                     // GEGLU hidden width: 4 x the LOGICAL channels (a multiple of 128, never padded)
        bf16_t* ff = o.abf((size_t)M * FI);
        if (xfold_shape) {       // norm3 of the conditional rows left the cross-attention kernel; the unconditional rows' here
            if (M > Mx) o.layernorm(t2 + (size_t)Mx * C, 0, s.ln3g, s.ln3b, l3 + (size_t)Mx * C, 0, M - Mx, C, s.lc);
            o.linear(l3, nullptr, C, 0, s.wff1, s.bff1, true, M, 2 * FI, ACT_GEGLU, nullptr, ff);
        } else if (!o.linear_ln_big(t2, s.ln3g, s.ln3b, C, s.lc, s.wff1, s.bff1, true, M, 2 * FI, ACT_GEGLU, ff)) {       // norm3 folded into the GEGLU projection
            o.layernorm(t2, 0, s.ln3g, s.ln3b, l3, 0, M, C, s.lc);
            o.linear(l3, nullptr, C, 0, s.wff1, s.bff1, true, M, 2 * FI, ACT_GEGLU, nullptr, ff);
        }
        o.tap(8, l3, (size_t)M * C * 2);          // (only where a separate norm3 tensor exists: not with RDM_LNFOLD)
        o.tap(9, ff, (size_t)M * FI * 2);
        bf16_t* out = o.abf((size_t)M * C);
        o.tag = "st.ff2*proj_out";
        static const int no_ffout = getenv("RDM_NO_FFOUT") ? atoi(getenv("RDM_NO_FFOUT")) : 0;
        if (!no_ffout) {
            // t3 = ff W_2^T + b_2 + t2 and out = t3 W_out^T + b_out + x are one GEMM over the K-concatenated operand [ff | t2]
            // (dual-source A) with the product weights built by the packer: t3 never exists (2 of 9 tensor passes, one launch)
            o.linear(ff, t2, FI, C, s.wfo, s.bfo, true, M, C, ACT_NONE, a.p, out, nullptr, nullptr, 0, nullptr, 0, 1, in_wrap * n);
        } else {
            bf16_t* t3 = o.abf((size_t)M * C);
            o.linear(ff, nullptr, FI, 0, s.wff2, s.bff2, true, M, C, ACT_NONE, t2, t3);
            o.linear(t3, nullptr, C, 0, s.wout, s.bout, true, M, C, ACT_NONE, a.p, out);
        }
        o.tap(10, out, (size_t)M * C * 2);
        return Act{out, C, a.H, a.W, s.lc};
    };

    for (const UBlock& blk : u.blocks) {
        Act* skip = nullptr; Act sk{};
        if (blk.where == 2) { sk = hs.back(); hs.pop_back(); skip = &sk; }
        bool first = true;
        o.cur_block = (int)(&blk - &u.blocks[0]); o.cur_layer = 0;
        for (const ULayer& L : blk.layers) {
            switch (L.kind) {
                case 0: {
                    bf16_t* out = o.abf((size_t)Bfull * H * W * mc);
                    if (!o.plan) o.check(launch_conv_in(x, o.w<float>(u.cinw), o.w<float>(u.cinb), out, B, c.in_channels, H, W, mc, o.c->stream), "conv_in");
                    h = Act{out, mc, H, W, mcl};
                } break;
                case 1: h = resblock(u.res[L.idx], h, (first && skip) ? skip : nullptr); break;
                case 2:
                    if (B < Bfull) {       // first context-dependent layer: leave the shared prefix
                        {   // the activation itself is duplicated only where its readers cannot wrap the batch index instead (transformer())
                            const StW& s0 = u.st[L.idx];
                            static const int no_inwrap = getenv("RDM_NO_INWRAP") ? atoi(getenv("RDM_NO_INWRAP")) : 0;
                            static const int no_ffout_env = getenv("RDM_NO_FFOUT") ? atoi(getenv("RDM_NO_FFOUT")) : 0;
                            const int n0 = h.H * h.W, Mfull = Bfull * n0;
                            if (!no_inwrap && !no_wrap && !no_ffout_env && !o.c->deterministic &&
                                o.lin4_takes(Mfull, s0.c, 4 * s0.lc, s0.c, 0, (Bfull / 2) * n0)) h.half = true;
                            else expand(h);
                        }
                        for (Act& a : hs) { if (no_wrap || o.c->deterministic) expand(a); else a.half = true; }
                        B = Bfull;
                    }
                    h = transformer(u.st[L.idx], h); break;
                case 3: {
                    const ConvW& d = u.down[L.idx];
                    bf16_t* out = o.abf((size_t)Bfull * (h.H / 2) * (h.W / 2) * d.c);
                    o.tag = "downsample";
                    o.conv3(h.p, nullptr, d.c, 0, d.w, d.b, B, h.H, h.W, d.c, 2, 0, nullptr, 0, nullptr, out);
                    h = Act{out, d.c, h.H / 2, h.W / 2, d.lc};
                } break;
                case 4: {
                    const ConvW& d = u.up[L.idx];
                    bf16_t* out = o.abf((size_t)Bfull * (h.H * 2) * (h.W * 2) * d.c);
                    o.tag = "upsample";
                    o.conv3(h.p, nullptr, d.c, 0, d.w, d.b, B, h.H, h.W, d.c, 1, 1, nullptr, 0, nullptr, out);
                    h = Act{out, d.c, h.H * 2, h.W * 2, d.lc};
                } break;
            }
            first = false;
            o.cur_layer++;
        }
        if (blk.where == 0) hs.push_back(h);
        if (!o.plan && o.c->tap_buf && o.c->tap_sub == 0 && o.c->tap_block == (int)(&blk - &u.blocks[0])) {       // (inside the shared guidance prefix: B < Bfull samples exist)
            size_t nb = (size_t)B * h.H * h.W * h.C * 2; if (nb > o.c->tap_bytes) nb = o.c->tap_bytes;
            o.check(hipMemcpyAsync(o.c->tap_buf, h.p, nb, hipMemcpyDeviceToDevice, o.c->stream), "debug tap");
        }
    }
    if (B < Bfull) { expand(h); B = Bfull; }          // (a UNet without attention: the whole network was shared)
    bf16_t* no = o.abf((size_t)B * H * W * mc);
    bf16_t* hwp = o.abf(head_conv_wp_bytes(mc) / 2);
    o.tag = "out_head";
    o.head(h.p, B, H, W, mc, mcl, u.outg, u.outb, 1e-5f, u.outw, u.outbias, c.out_channels, eps_out, no, hwp);
}

// plan (count bytes) -> ensure arena -> run
template <typename F>
static int run_with_arena(rdm_ctx* c, Arena& ar, const char* blob, F&& body) {
    Ops plan{c, &ar, blob, true};
    char* keep = ar.base; ar.base = nullptr; ar.reset(); ar.peak = 0;
    body(plan);
    const size_t need = ar.peak + 4096;
    ar.base = keep;
    RDM_TRY(ensure_bytes(c, &ar.base, &ar.cap, need));
    ar.reset();
    Ops run{c, &ar, blob, false};
    body(run);
    return run.rc;
}

static int cfg_check_unet(rdm_ctx* c, const rdm_unet_cfg* g) {
    if (!g || g->n_channel_mult < 1 || g->n_channel_mult > RDM_MAX_LEVELS || g->n_attention_resolutions > RDM_MAX_LEVELS)
        return c ? c->fail(-1, "bad unet cfg") : -1;
    if (g->model_channels % 32 || g->num_head_channels != 32 || g->context_dim % 64 || g->in_channels > 4 || g->out_channels > 4)
        return c ? c->fail(-1, "unsupported unet cfg: model_channels %% 32 == 0 (GroupNorm32; widths that are not multiples of 64 run zero-padded), num_head_channels == 32, context_dim %% 64 == 0 required") : -1;
    return 0;
}

static long long write_manifest(const Manifest& mf, char* buf, size_t buflen, size_t* blob_bytes) {
    if (blob_bytes) *blob_bytes = (mf.total + 255) & ~(size_t)255;
    if (buf && buflen > mf.text.size()) { memcpy(buf, mf.text.c_str(), mf.text.size() + 1); }
    return (long long)mf.text.size();
}

template <typename M>
static int load_blob(rdm_ctx* c, M& m, const Manifest& mf, const void* packed, size_t nbytes) {
    const size_t need = (mf.total + 255) & ~(size_t)255;
    if (nbytes != need) return c->fail(-1, "packed blob is %zu bytes, manifest needs %zu", nbytes, need);
    RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    c->drop_frags();                                  // derived weight layouts refer to blob addresses that may be reused
    if (m.blob) { RDM_CHECK_HIP(c, hipFree(m.blob)); m.blob = nullptr; }
    RDM_CHECK_HIP(c, hipMalloc((void**)&m.blob, need));
    RDM_CHECK_HIP(c, hipMemcpy(m.blob, packed, need, hipMemcpyHostToDevice));
    m.blob_bytes = need; m.loaded = true;
    return 0;
}

// ------------------------------------------------------------------------------------ VQ decode
// decoder trunk shared by the VQ-f4 (ldm) and VQGAN-f16 (taming) first stages: ResnetBlocks, AttnBlocks, nearest-2x upsample convs
static void vq_trunk(Ops& o, VqModel& v, bf16_t* h, int B, int H, int W, float* img) {
    const rdm_vq_cfg& c = v.cfg;
    int bin = c.ch * c.ch_mult[c.n_ch_mult - 1];
    auto res = [&](const VqRes& r, bf16_t* x) -> bf16_t* {
        const int HW = H * W, M = B * HW;
        bf16_t* n1 = o.abf((size_t)M * r.cin);
        o.groupnorm(x, nullptr, r.cin, 0, B, HW, r.n1g, r.n1b, 1e-6f, 1, n1);
        bf16_t* h1 = o.abf((size_t)M * r.cout);
        o.conv3(n1, nullptr, r.cin, 0, r.w1, r.b1, B, H, W, r.cout, 1, 0, nullptr, 0, nullptr, h1);
        bf16_t* n2 = o.abf((size_t)M * r.cout);
        o.groupnorm(h1, nullptr, r.cout, 0, B, HW, r.n2g, r.n2b, 1e-6f, 1, n2);
        const bf16_t* rs = x;
        if (r.skip) { bf16_t* s = o.abf((size_t)M * r.cout); o.linear(x, nullptr, r.cin, 0, r.wsk, r.bsk, true, M, r.cout, ACT_NONE, nullptr, s); rs = s; }
        bf16_t* out = o.abf((size_t)M * r.cout);
        o.conv3(n2, nullptr, r.cout, 0, r.w2, r.b2, B, H, W, r.cout, 1, 0, nullptr, 0, rs, out);
        return out;
    };
    auto attn = [&](const VqAttn& a, bf16_t* x) -> bf16_t* {   // ldm / taming AttnBlock: single head over H*W tokens, scale C^-1/2
        const int C = a.c, n = H * W, M = B * n;
        bf16_t* hn = o.abf((size_t)M * C);
        o.groupnorm(x, nullptr, C, 0, B, n, a.ng, a.nb, 1e-6f, 0, hn);
        bf16_t* q = o.abf((size_t)M * C); bf16_t* kk = o.abf((size_t)M * C); bf16_t* vt = o.abf((size_t)M * C);
        o.linear(hn, nullptr, C, 0, a.wq, a.bq, true, M, C, ACT_NONE, nullptr, q);
        o.linear(hn, nullptr, C, 0, a.wk, a.bk, true, M, C, ACT_NONE, nullptr, kk);
        float* S = o.af32((size_t)B * n * n); bf16_t* P = o.abf((size_t)B * n * n); bf16_t* ao = o.abf((size_t)M * C);
        if (!o.plan) {
            IgemmParams p = o.base(C, n, C);     // V^T[b] = Wv . hn[b]^T   (bias b_v folded in after P.V: rows of P sum to 1)
            p.A0 = o.w<bf16_t>(a.wv); p.C0 = C; p.W = hn; p.sW = (long long)n * C; p.sO = (long long)C * n; p.out_bf16 = vt; p.ldo = n;
            o.check(launch_igemm(p, false, B, o.c->stream), "vq v^T");
            IgemmParams s = o.base(n, n, C);     // S[b] = q[b] k[b]^T * C^-1/2  (fp32 scores)
            s.A0 = q; s.C0 = C; s.W = kk; s.sA = (long long)n * C; s.sW = (long long)n * C; s.sO = (long long)n * n; s.out_f32 = S; s.ldo = n;
            s.alpha = 1.0f / sqrtf((float)C);
            o.check(launch_igemm(s, false, B, o.c->stream), "vq qk^T");
            o.check(launch_softmax_rows(S, P, (long long)B * n, n, o.c->stream), "vq softmax");
            IgemmParams pv = o.base(n, C, n);    // O[b] = P[b] V[b] + b_v
            pv.A0 = P; pv.C0 = n; pv.W = vt; pv.sA = (long long)n * n; pv.sW = (long long)C * n; pv.sO = (long long)n * C; pv.out_bf16 = ao; pv.ldo = C;
            pv.bias = o.w<float>(a.bv);
            o.check(launch_igemm(pv, false, B, o.c->stream), "vq pv");
        }
        bf16_t* out = o.abf((size_t)M * C);
        o.linear(ao, nullptr, C, 0, a.wo, a.bo, true, M, C, ACT_NONE, x, out);
        return out;
    };
    // debug tap (rdm_debug_tap): first-stage decoder layers are blocks 1000, 1001, ... in execution order (conv_in's output = 1000)
    int tapi = 1000;
    auto vtap = [&](const bf16_t* t, int ch) {
        o.cur_block = tapi++; o.cur_layer = 0;
        if (!o.plan && o.c->tap_buf && o.c->tap_sub == 0 && o.c->tap_block == o.cur_block) {
            size_t nb = (size_t)B * H * W * ch * 2; if (nb > o.c->tap_bytes) nb = o.c->tap_bytes;
            o.check(hipMemcpyAsync(o.c->tap_buf, t, nb, hipMemcpyDeviceToDevice, o.c->stream), "debug tap");
        }
    };
    vtap(h, bin);
    h = res(v.mid1, h); vtap(h, bin);
    if (c.mid_attn) { h = attn(v.attn, h); vtap(h, bin); }
    h = res(v.mid2, h); vtap(h, bin);
    for (int lvl = c.n_ch_mult - 1; lvl >= 0; lvl--) {
        for (size_t i = 0; i < v.up_blocks[lvl].size(); i++) {
            h = res(v.up_blocks[lvl][i], h); bin = v.up_blocks[lvl][i].cout; vtap(h, bin);
            if (i < v.up_attn[lvl].size()) { h = attn(v.up_attn[lvl][i], h); vtap(h, bin); }
        }
        if (lvl != 0) {
            const ConvW& u = v.upsample[lvl];
            bf16_t* out = o.abf((size_t)B * (H * 2) * (W * 2) * u.c);
            o.conv3(h, nullptr, u.c, 0, u.w, u.b, B, H, W, u.c, 1, 1, nullptr, 0, nullptr, out);
            h = out; H *= 2; W *= 2;
            vtap(h, u.c);
        }
    }
    bf16_t* no = o.abf((size_t)B * H * W * bin);
    bf16_t* hwp = o.abf(head_conv_wp_bytes(bin) / 2);
    o.head(h, B, H, W, bin, bin, v.noutg, v.noutb, 1e-6f, v.coutw, v.coutb, c.out_ch, img, no, hwp);
}

// VQ-f4 (3-channel latent): quantise (or not) + post_quant_conv + conv_in as tiny fp32 stem kernels, then the trunk
static void vq_body(Ops& o, VqModel& v, const float* z, int B, int force_not_quantize, float* img, int* idx_out) {
    const rdm_vq_cfg& c = v.cfg;
    const int zr = c.resolution >> (c.n_ch_mult - 1);
    const int HW0 = zr * zr;
    float* zq = o.af32((size_t)B * c.z_channels * HW0);
    const int quant = (!c.kl && !force_not_quantize) ? 1 : 0;
    if (!o.plan)
        o.check(launch_vq_quantize(z, c.kl ? nullptr : o.w<float>(v.codebook), c.kl ? 0 : c.n_embed, o.w<float>(v.pqw), o.w<float>(v.pqb),
                                   zq, idx_out, B, HW0, quant, o.c->stream), "vq_quantize");
    const int bin = c.ch * c.ch_mult[c.n_ch_mult - 1];
    bf16_t* h = o.abf((size_t)B * HW0 * bin);
    if (!o.plan) o.check(launch_conv_in(zq, o.w<float>(v.cinw), o.w<float>(v.cinb), h, B, c.z_channels, zr, zr, bin, o.c->stream), "vq conv_in");
    vq_trunk(o, v, h, B, zr, zr, img);
}

// VQGAN-f16 (wide latent): decode_to_img from code indices (taming Net2NetTransformer.decode_to_img -> quantize.get_codebook_entry ->
// VQModel.decode = post_quant_conv + Decoder; reached from transformer.py:296-312): codebook rows gathered straight into bf16
// NHWC tokens, post_quant_conv (1x1) as a GEMM, conv_in as a 3x3 halo conv.
static void vq_wide_body(Ops& o, VqModel& v, const long long* indices, int B, float* img) {
    const rdm_vq_cfg& c = v.cfg;
    const int zr = c.resolution >> (c.n_ch_mult - 1), HW0 = zr * zr, M = B * HW0;
    bf16_t* zq = o.abf((size_t)M * c.embed_dim);
    if (!o.plan) o.check(launch_codebook_gather(indices, o.w<float>(v.codebook), c.n_embed, c.embed_dim, M, zq, o.c->stream), "codebook gather");
    bf16_t* h0 = o.abf((size_t)M * c.z_channels);
    o.linear(zq, nullptr, c.embed_dim, 0, v.pqw, v.pqb, true, M, c.z_channels, ACT_NONE, nullptr, h0);
    const int bin = c.ch * c.ch_mult[c.n_ch_mult - 1];
    bf16_t* h = o.abf((size_t)M * bin);
    o.conv3(h0, nullptr, c.z_channels, 0, v.cinw, v.cinb, B, zr, zr, bin, 1, 0, nullptr, 0, nullptr, h);
    vq_trunk(o, v, h, B, zr, zr, img);
}

// VQ-f4 encode: image f32 [B, out_ch, R, R] -> z f32 [B, embed_dim, R / 2^(levels-1), ...]  (VQModelInterface.encode: no quantisation here)
static void vqenc_body(Ops& o, VqEncModel& v, const float* img, int B, float* z) {
    const rdm_vq_cfg& c = v.cfg;
    int H = c.resolution, W = c.resolution;
    bf16_t* h = o.abf((size_t)B * H * W * c.ch);
    if (!o.plan) o.check(launch_conv_in(img, o.w<float>(v.cinw), o.w<float>(v.cinb), h, B, c.out_ch, H, W, c.ch, o.c->stream), "encoder conv_in");
    auto res = [&](const VqRes& r, bf16_t* x) -> bf16_t* {
        const int HW = H * W, M = B * HW;
        bf16_t* n1 = o.abf((size_t)M * r.cin);
        o.groupnorm(x, nullptr, r.cin, 0, B, HW, r.n1g, r.n1b, 1e-6f, 1, n1);
        bf16_t* h1 = o.abf((size_t)M * r.cout);
        o.conv3(n1, nullptr, r.cin, 0, r.w1, r.b1, B, H, W, r.cout, 1, 0, nullptr, 0, nullptr, h1);
        bf16_t* n2 = o.abf((size_t)M * r.cout);
        o.groupnorm(h1, nullptr, r.cout, 0, B, HW, r.n2g, r.n2b, 1e-6f, 1, n2);
        const bf16_t* rs = x;
        if (r.skip) { bf16_t* sk = o.abf((size_t)M * r.cout); o.linear(x, nullptr, r.cin, 0, r.wsk, r.bsk, true, M, r.cout, ACT_NONE, nullptr, sk); rs = sk; }
        bf16_t* out = o.abf((size_t)M * r.cout);
        o.conv3(n2, nullptr, r.cout, 0, r.w2, r.b2, B, H, W, r.cout, 1, 0, nullptr, 0, rs, out);
        return out;
    };
    auto attn = [&](const VqAttn& a, bf16_t* x) -> bf16_t* {       // as vq_trunk's AttnBlock
        const int C = a.c, n = H * W, M = B * n;
        bf16_t* hn = o.abf((size_t)M * C);
        o.groupnorm(x, nullptr, C, 0, B, n, a.ng, a.nb, 1e-6f, 0, hn);
        bf16_t* q = o.abf((size_t)M * C); bf16_t* kk = o.abf((size_t)M * C); bf16_t* vt = o.abf((size_t)M * C);
        o.linear(hn, nullptr, C, 0, a.wq, a.bq, true, M, C, ACT_NONE, nullptr, q);
        o.linear(hn, nullptr, C, 0, a.wk, a.bk, true, M, C, ACT_NONE, nullptr, kk);
        float* S = o.af32((size_t)B * n * n); bf16_t* P = o.abf((size_t)B * n * n); bf16_t* ao = o.abf((size_t)M * C);
        if (!o.plan) {
            IgemmParams p = o.base(C, n, C);
            p.A0 = o.w<bf16_t>(a.wv); p.C0 = C; p.W = hn; p.sW = (long long)n * C; p.sO = (long long)C * n; p.out_bf16 = vt; p.ldo = n;
            o.check(launch_igemm(p, false, B, o.c->stream), "enc v^T");
            IgemmParams sc = o.base(n, n, C);
            sc.A0 = q; sc.C0 = C; sc.W = kk; sc.sA = (long long)n * C; sc.sW = (long long)n * C; sc.sO = (long long)n * n; sc.out_f32 = S; sc.ldo = n;
            sc.alpha = 1.0f / sqrtf((float)C);
            o.check(launch_igemm(sc, false, B, o.c->stream), "enc qk^T");
            o.check(launch_softmax_rows(S, P, (long long)B * n, n, o.c->stream), "enc softmax");
            IgemmParams pv = o.base(n, C, n);
            pv.A0 = P; pv.C0 = n; pv.W = vt; pv.sA = (long long)n * n; pv.sW = (long long)C * n; pv.sO = (long long)n * C; pv.out_bf16 = ao; pv.ldo = C;
            pv.bias = o.w<float>(a.bv);
            o.check(launch_igemm(pv, false, B, o.c->stream), "enc pv");
        }
        bf16_t* out = o.abf((size_t)M * C);
        o.linear(ao, nullptr, C, 0, a.wo, a.bo, true, M, C, ACT_NONE, x, out);
        return out;
    };
    int bin = c.ch;
    for (int lvl = 0; lvl < c.n_ch_mult; lvl++) {
        for (size_t i = 0; i < v.down[lvl].size(); i++) {
            h = res(v.down[lvl][i], h); bin = v.down[lvl][i].cout;
            if (i < v.down_attn[lvl].size()) h = attn(v.down_attn[lvl][i], h);
        }
        if (lvl != c.n_ch_mult - 1) {      // F.pad(x, (0, 1, 0, 1)) + Conv2d(stride 2, padding 0): window rows 2 oy .. 2 oy + 2, zero beyond the last row / column
            const ConvW& d = v.downsample[lvl];
            bf16_t* out = o.abf((size_t)B * (H / 2) * (W / 2) * d.c);
            o.conv3(h, nullptr, d.c, 0, d.w, d.b, B, H, W, d.c, 2, 0, nullptr, 0, nullptr, out, /*asym=*/1);
            h = out; H /= 2; W /= 2;
        }
    }
    h = res(v.mid1, h);
    if (c.mid_attn) h = attn(v.attn, h);
    h = res(v.mid2, h);
    bf16_t* no = o.abf((size_t)B * H * W * bin);
    bf16_t* hwp = o.abf(head_conv_wp_bytes(bin) / 2);
    float* ze = o.af32((size_t)B * c.z_channels * H * W);
    o.head(h, B, H, W, bin, bin, v.noutg, v.noutb, 1e-6f, v.coutw, v.coutb, c.z_channels, ze, no, hwp);
    if (!o.plan) o.check(launch_vq_quantize(ze, nullptr, 0, o.w<float>(v.qw), o.w<float>(v.qb), z, nullptr, B, H * W, 0, o.c->stream), "quant_conv");
}

// ------------------------------------------------------------------------------------ CLIP
static void clip_tower(Ops& o, const std::vector<ClipBlk>& blks, float* x, int B, int L, int Wd, int heads, int causal) {
    const int M = B * L;
    bf16_t* ln = o.abf((size_t)M * Wd); bf16_t* qkv = o.abf((size_t)M * 3 * Wd); bf16_t* ao = o.abf((size_t)M * Wd);
    bf16_t* hid = o.abf((size_t)M * 4 * Wd);
    for (const ClipBlk& k : blks) {
        o.layernorm(x, 1, k.ln1g, k.ln1b, ln, 0, M, Wd);
        o.linear(ln, nullptr, Wd, 0, k.wqkv, k.bqkv, true, M, 3 * Wd, ACT_NONE, nullptr, qkv);
        if (!o.plan) {
            SmallAttnParams p{}; p.q = qkv; p.ldq = 3 * Wd; p.k = qkv + Wd; p.ldk = 3 * Wd; p.v = qkv + 2 * Wd; p.ldv = 3 * Wd;
            p.out = ao; p.ldo = Wd; p.nq = L; p.nkv = L; p.causal = causal; p.scale = 1.0f / sqrtf((float)(Wd / heads));
            o.check(launch_small_attention(p, Wd / heads, heads, B, o.c->stream), "clip attention");
        }
        o.linear(ao, nullptr, Wd, 0, k.wo, k.bo, true, M, Wd, ACT_NONE, nullptr, nullptr, x, x);        // x += out_proj(attn)
        o.layernorm(x, 1, k.ln2g, k.ln2b, ln, 0, M, Wd);
        o.linear(ln, nullptr, Wd, 0, k.wfc, k.bfc, true, M, 4 * Wd, ACT_QUICKGELU, nullptr, hid);
        o.linear(hid, nullptr, 4 * Wd, 0, k.wpj, k.bpj, true, M, Wd, ACT_NONE, nullptr, nullptr, x, x);  // x += mlp
    }
}
static void clip_text_body(Ops& o, ClipModel& m, const long long* tokens, int B, float* out) {
    const rdm_clip_cfg& c = m.cfg; const int L = c.context_length, Wd = c.transformer_width;
    float* x = o.af32((size_t)B * L * Wd);
    if (!o.plan) o.check(launch_clip_embed(tokens, o.w<float>(m.tok), o.w<float>(m.pos), x, B, L, Wd, o.c->stream), "clip embed");
    clip_tower(o, m.text, x, B, L, Wd, c.transformer_heads, 1);
    float* eot = o.af32((size_t)B * Wd); bf16_t* ln = o.abf((size_t)B * Wd);
    if (!o.plan) o.check(launch_clip_gather_eot(tokens, x, eot, B, L, Wd, o.c->stream), "gather eot");
    o.layernorm(eot, 1, m.lnfg, m.lnfb, ln, 0, B, Wd);
    o.single_row = true;                                  // the pooled token: one row per sample
    o.linear(ln, nullptr, Wd, 0, m.tproj, 0, false, B, c.embed_dim, ACT_NONE, nullptr, nullptr, out);
    o.single_row = false;
}
// raw_h > 0: `img` is the un-preprocessed [B,3,raw_h,raw_w] image in [-1,1]; the bicubic resize + normalisation of
// ClipImageRetriever.preprocess (rdm/modules/retrievers.py:83-91) is fused into the patch gather
static void clip_image_body(Ops& o, ClipModel& m, const float* img, int B, float* out, int raw_h = 0, int raw_w = 0) {
    const rdm_clip_cfg& c = m.cfg; const int P = c.vision_patch_size, G = c.image_resolution / P, Wd = c.vision_width;
    const int K = 3 * P * P, L = G * G + 1;
    bf16_t* patches = o.abf((size_t)B * G * G * K); float* pe = o.af32((size_t)B * G * G * Wd);
    if (!o.plan) {
        if (raw_h > 0) o.check(launch_clip_preprocess(img, B, raw_h, raw_w, c.image_resolution, P, nullptr, patches, o.c->stream), "preprocess+patchify");
        else o.check(launch_clip_patchify(img, patches, B, c.image_resolution, P, o.c->stream), "patchify");
    }
    o.linear(patches, nullptr, K, 0, m.conv1, 0, false, B * G * G, Wd, ACT_NONE, nullptr, nullptr, pe);
    float* x0 = o.af32((size_t)B * L * Wd); float* x = o.af32((size_t)B * L * Wd);
    if (!o.plan) o.check(launch_clip_vit_assemble(pe, o.w<float>(m.cls), o.w<float>(m.vpos), x0, B, G * G, Wd, o.c->stream), "vit assemble");
    o.layernorm(x0, 1, m.lnpreg, m.lnpreb, x, 1, B * L, Wd);
    clip_tower(o, m.vis, x, B, L, Wd, Wd / 64, 0);
    float* cls = o.af32((size_t)B * Wd); bf16_t* ln = o.abf((size_t)B * Wd);
    if (!o.plan) o.check(launch_gather_rows_f32(x, cls, B, L, Wd, o.c->stream), "gather cls");
    o.layernorm(cls, 1, m.lnpostg, m.lnpostb, ln, 0, B, Wd);
    o.single_row = true;                                  // the pooled token: one row per sample
    o.linear(ln, nullptr, Wd, 0, m.vproj, 0, false, B, c.embed_dim, ACT_NONE, nullptr, nullptr, out);
    o.single_row = false;
}

// ==================================================================================== C ABI
extern "C" {

const char* rdm_version(void) { return "rdm_hip 0.1 (gfx950)"; }

int rdm_ctx_create(int device_id, rdm_ctx** out) {
    if (!out) return -1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -4;      // no HIP device: fail loudly, there is no CPU path
    if (device_id < 0 || device_id >= n) return -1;
    DevGuard guard(device_id);
    rdm_ctx* c = new rdm_ctx();
    c->device = device_id;
    {   // zero page + identity matrix (the residual-as-K-columns operand of the linear GEMMs, igemm.hip)
        const size_t bytes = RDM_EYE_OFFSET + (size_t)RDM_EYE_N * RDM_EYE_N * 2;
        std::vector<uint16_t> host(bytes / 2, 0);
        for (int i = 0; i < RDM_EYE_N; i++) host[RDM_EYE_OFFSET / 2 + (size_t)i * RDM_EYE_N + i] = 0x3F80;   // bf16 1.0
        if (hipMalloc(&c->zero_page, bytes) != hipSuccess || hipMemcpy(c->zero_page, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { delete c; return -2; }
    }
    *out = c;
    return 0;
}

void rdm_ctx_destroy(rdm_ctx* c) {
    if (!c) return;
    if (c->comm) rdm_comm_destroy(c);
    DevGuard guard(c->device);
    hipDeviceSynchronize();
    // (the side stream and its two events are left to the runtime: a context may be destroyed from a finaliser at interpreter shutdown, and tearing
    //  streams down there is not worth one lost handle per context)
    void* ptrs[] = {c->zero_page, c->unet.blob, c->unet.arena.base, c->unet.kv_cache, c->vq.blob, c->vq.arena.base,
                    c->clip.blob, c->clip.arena.base, c->gn_partial, c->samp, c->splitk_ws, c->unet.xa_cache,
                    c->rarm.blob, c->rarm.arena.base, c->rarm.cache, c->rarm.ctxkv, c->rarm.state, c->rarm.xa, c->rarm.xws, c->wfrag_tmp, c->bwd_tmp,
                    c->vqenc.blob, c->vqenc.arena.base, c->eye3, c->unet.emb_table};
    for (void* p : ptrs) if (p) hipFree(p);
    c->drop_frags();
    knn_free(c->db);
    delete c;
}

const char* rdm_last_error(rdm_ctx* c) { return c ? c->err : "null context"; }

// The grow-only work buffers (backward scratch: K-major operand copies + up to 256 MB of fp32 weight-gradient planes per conv -- over
// 1 GB for the 576 -> 192 block at 64 x 64 and batch 64 --, split-K planes, per-call weight re-packs, sampler scratch) are kept between
// calls so that steady-state steps allocate nothing; a caller switching from training back to sampling hands them back here.  They
// are re-created on demand.
int rdm_release_scratch(rdm_ctx* c) {
    RDM_ENTER(c);
    RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    char** bufs[] = {&c->bwd_tmp, &c->wfrag_tmp, &c->splitk_ws, &c->samp};
    size_t* sizes[] = {&c->bwd_tmp_bytes, &c->wfrag_tmp_bytes, &c->splitk_ws_bytes, &c->samp_bytes};
    for (int i = 0; i < 4; i++) { if (*bufs[i]) { (void)hipFree(*bufs[i]); *bufs[i] = nullptr; } *sizes[i] = 0; }
    return 0;
}
int rdm_set_deterministic(rdm_ctx* c, int on) { if (!c) return -1; c->deterministic = on != 0; return 0; }
int rdm_get_deterministic(rdm_ctx* c) { return c ? (c->deterministic ? 1 : 0) : -1; }
int rdm_set_stream(rdm_ctx* c, void* s) {
    RDM_ENTER(c);
    if ((hipStream_t)s != c->stream) {      // work queued (and derived weight copies packed) on the old stream completes before the new one is used
        RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        c->stream = (hipStream_t)s;
    }
    return 0;
}

long long rdm_unet_manifest(const rdm_unet_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes) {
    if (cfg_check_unet(nullptr, cfg)) return -1;
    UNet u; Manifest mf; build_unet(u, *cfg, mf);
    return write_manifest(mf, buf, buflen, blob_bytes);
}
long long rdm_vq_manifest(const rdm_vq_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes) {
    if (!cfg || cfg->n_ch_mult < 1 || cfg->n_ch_mult > RDM_MAX_LEVELS || cfg->n_attn_resolutions < 0 || cfg->n_attn_resolutions > RDM_MAX_LEVELS) return -1;
    VqModel v; Manifest mf; build_vq(v, *cfg, mf);
    return write_manifest(mf, buf, buflen, blob_bytes);
}
long long rdm_clip_manifest(const rdm_clip_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes) {
    if (!cfg) return -1;
    ClipModel m; Manifest mf; build_clip(m, *cfg, mf);
    return write_manifest(mf, buf, buflen, blob_bytes);
}

int rdm_load_unet(rdm_ctx* c, const rdm_unet_cfg* cfg, const void* packed, size_t nbytes) {
    RDM_ENTER(c);
    if (!c) return -1;
    RDM_TRY(cfg_check_unet(c, cfg));
    Manifest mf; build_unet(c->unet, *cfg, mf);
    return load_blob(c, c->unet, mf, packed, nbytes);
}
int rdm_load_vq(rdm_ctx* c, const rdm_vq_cfg* cfg, const void* packed, size_t nbytes) {
    RDM_ENTER(c);
    if (!c || !cfg) return -1;
    if (cfg->ch % 64 || cfg->out_ch > 4 || cfg->n_attn_resolutions < 0 || cfg->n_attn_resolutions > RDM_MAX_LEVELS ||
        !((cfg->embed_dim == 3 && cfg->z_channels == 3) || (cfg->embed_dim % 64 == 0 && cfg->z_channels % 64 == 0 && !cfg->kl)))
        return c->fail(-1, "unsupported vq cfg: ch %% 64 == 0 and either embed_dim == z_channels == 3 (VQ-f4 / KL-f4) or both multiples of 64 (VQGAN-f16)");
    Manifest mf; build_vq(c->vq, *cfg, mf);
    return load_blob(c, c->vq, mf, packed, nbytes);
}
static int cfg_check_vqenc(rdm_ctx* c, const rdm_vq_cfg* g) {
    const bool ok = g && g->n_ch_mult >= 1 && g->n_ch_mult <= RDM_MAX_LEVELS && g->n_attn_resolutions >= 0 && g->n_attn_resolutions <= RDM_MAX_LEVELS &&
                    g->ch % 64 == 0 && g->out_ch <= 4 && g->embed_dim == 3 && g->z_channels == 3 && !g->kl && g->resolution % (1 << (g->n_ch_mult - 1)) == 0;
    if (!ok) return c ? c->fail(-1, "unsupported first-stage encoder cfg: VQ interface (kl = 0) with embed_dim == z_channels == 3, ch %% 64 == 0") : -1;
    return 0;
}
long long rdm_vqenc_manifest(const rdm_vq_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes) {
    if (cfg_check_vqenc(nullptr, cfg)) return -1;
    VqEncModel v; Manifest mf; build_vqenc(v, *cfg, mf);
    return write_manifest(mf, buf, buflen, blob_bytes);
}
int rdm_load_vqenc(rdm_ctx* c, const rdm_vq_cfg* cfg, const void* packed, size_t nbytes) {
    RDM_ENTER(c);
    if (!packed) return c->fail(-1, "null blob");
    RDM_TRY(cfg_check_vqenc(c, cfg));
    Manifest mf; build_vqenc(c->vqenc, *cfg, mf);
    return load_blob(c, c->vqenc, mf, packed, nbytes);
}
int rdm_vq_encode(rdm_ctx* c, const float* img, int b, float* z_out) {
    RDM_ENTER(c);
    if (!img || !z_out || b < 1) return c->fail(-1, "rdm_vq_encode: bad argument");
    if (!c->vqenc.loaded) return c->fail(-1, "first-stage encoder weights not loaded (rdm_load_vqenc)");
    RDM_TRY(ensure_gn_partial(c, b));
    return run_with_arena(c, c->vqenc.arena, c->vqenc.blob, [&](Ops& o) { vqenc_body(o, c->vqenc, img, b, z_out); });
}
int rdm_load_clip(rdm_ctx* c, const rdm_clip_cfg* cfg, const void* packed, size_t nbytes) {
    RDM_ENTER(c);
    if (!c || !cfg) return -1;
    if (cfg->transformer_width % 64 || cfg->vision_width % 64 || cfg->embed_dim % 8 ||
        cfg->transformer_width / cfg->transformer_heads != 64)
        return c->fail(-1, "unsupported clip cfg: widths %% 64 == 0 and 64-d heads required");
    Manifest mf; build_clip(c->clip, *cfg, mf);
    return load_blob(c, c->clip, mf, packed, nbytes);
}

static int unet_forward_impl(rdm_ctx* c, const float* x, const int64_t* t, const float* context, const bf16_t* kv_cached,
                             int b, int k, int H, int W, float* eps_out, int ctx_rows = -1, int shared_half = 0, const float* emb_row = nullptr) {
    UNet& u = c->unet;
    if (!u.loaded) return c->fail(-1, "unet weights not loaded");
    const int down = 1 << (u.cfg.n_channel_mult - 1);
    if (b < 1 || k < 1 || H % down || W % down) return c->fail(-1, "bad unet_forward shape b=%d k=%d H=%d W=%d", b, k, H, W);
    if (k > 256) return c->fail(-1, "k=%d neighbours exceeds the cross-attention kernel limit (256)", k);
    RDM_TRY(ensure_gn_partial(c, b));
    return run_with_arena(c, u.arena, u.blob, [&](Ops& o) {
        const bf16_t* kv = kv_cached;
        const bf16_t* xa = (kv_cached && xattn_skinny_ok(u, k)) ? u.xa_cache : nullptr;
        if (!kv) {
            bf16_t* kvb = o.abf((size_t)b * k * u.kv_total);
            unet_compute_kv(o, u, context, b, k, kvb);
            kv = kvb;
            if (xattn_skinny_ok(u, k)) {
                bf16_t* xab = o.abf((size_t)b * u.xa_total);
                unet_compute_xattn(o, u, kv, b, k, xab);
                xa = xab;
            }
        }
        unet_body(o, u, x, (const long long*)t, kv, xa, b, k, H, W, eps_out, (ctx_rows >= 0 && ctx_rows <= b) ? ctx_rows : b, shared_half, emb_row);
    });
}

int rdm_unet_forward(rdm_ctx* c, const float* x, const int64_t* t, const float* context, int b, int k, int H, int W,
                     float* eps_out) {
    RDM_ENTER(c);
    if (!c || !x || !t || !context || !eps_out) return c ? c->fail(-1, "null argument") : -1;
    return unet_forward_impl(c, x, t, context, nullptr, b, k, H, W, eps_out);
}

// ---- samplers
int rdm_ddim_num_intermediates(int S, int log_every_t) {
    int n = 0;
    for (int i = 0; i < S; i++) { const int index = S - i - 1; if (index % log_every_t == 0 || index == S - 1) n++; }
    return n;
}

// precompute K/V of the conditioning once per sample() call (reference recomputes S*16 times)
static int prepare_kv(rdm_ctx* c, const float* cond, const float* uncond, int B, int k, int* Beff) {
    UNet& u = c->unet;
    const int nb = uncond ? 2 * B : B;
    *Beff = nb;
    const size_t cd = u.cfg.context_dim;
    RDM_TRY(ensure_bytes(c, (char**)&u.kv_cache, &u.kv_cache_bytes, (size_t)nb * k * u.kv_total * 2));
    // stage [cond | uncond] contiguous in the sampler scratch (fp32)
    float* cat = (float*)c->samp;
    RDM_CHECK_HIP(c, hipMemcpyAsync(cat, cond, (size_t)B * k * cd * 4, hipMemcpyDeviceToDevice, c->stream));
    if (uncond) RDM_CHECK_HIP(c, hipMemcpyAsync(cat + (size_t)B * k * cd, uncond, (size_t)B * k * cd * 4, hipMemcpyDeviceToDevice, c->stream));
    const bool skinny = xattn_skinny_ok(u, k);
    if (skinny) RDM_TRY(ensure_bytes(c, (char**)&u.xa_cache, &u.xa_cache_bytes, (size_t)nb * u.xa_total * 2));
    u.ctx_rows = nb;
    if (uncond) {      // how many trailing samples have all-zero neighbours?  (one small kernel + one 4*nb-byte read per sampling call)
        static const int off = getenv("RDM_NO_ZEROCTX") ? atoi(getenv("RDM_NO_ZEROCTX")) : 0;
        std::vector<int> flags(nb, 1);
        int* dflags = (int*)c->gn_partial;             // scratch: >= 16 KB once any forward ran; make sure it exists
        RDM_TRY(ensure_gn_partial(c, nb));
        dflags = (int*)c->gn_partial;
        RDM_CHECK_HIP(c, launch_row_nonzero(cat, nb, (long long)k * cd, dflags, c->stream));
        RDM_CHECK_HIP(c, hipMemcpyAsync(flags.data(), dflags, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
        RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        int rows = nb;
        while (rows > 0 && flags[rows - 1] == 0) rows--;
        if (!off && !c->deterministic) u.ctx_rows = rows;     // (which rows are "trailing" depends on the batch)
    }
    return run_with_arena(c, u.arena, u.blob, [&](Ops& o) {
        unet_compute_kv(o, u, cat, nb, k, u.kv_cache);
        if (skinny) unet_compute_xattn(o, u, u.kv_cache, nb, k, u.xa_cache);
    });
}

int rdm_ddim_sample(rdm_ctx* c, const rdm_ddim_args* a, const float* x_T, const float* cond, const float* uncond,
                    const float* noise, float* z_out, float* x_inter, float* pred_x0_inter) {
    RDM_ENTER(c);
    if (!c || !a || !x_T || !cond || !z_out) return c ? c->fail(-1, "null argument") : -1;
    UNet& u = c->unet;
    if (!u.loaded) return c->fail(-1, "unet weights not loaded");
    if (a->unconditional_guidance_scale < 1.0f) return c->fail(-1, "unconditional_guidance_scale must be >= 1 (ddim.py:223)");
    const bool cfg = a->unconditional_guidance_scale > 1.0f;
    if (cfg && !uncond) return c->fail(-1, "unconditional_conditioning required when scale > 1 (ddim.py:231)");
    if (a->eta != 0.f && !noise) return c->fail(-1, "eta > 0 needs an explicit noise stack [S,B,C,H,W] (device RNG parity is not defined)");
    if (a->S < 1 || a->S > a->T || !a->alphas_cumprod) return c->fail(-1, "bad schedule");
    const int B = a->batch, k = a->k, S = a->S;
    const long long n1 = (long long)B * a->channels * a->height * a->width;
    // schedule (ldm make_ddim_timesteps 'uniform' + make_ddim_sampling_parameters)
    const int step = a->T / S;
    std::vector<int> ts; for (int i = 0; i < a->T && (int)ts.size() < (a->T + step - 1) / step; i += step) ts.push_back(i + 1);
    const int total = (int)ts.size();
    for (int v : ts) if (v >= a->T) return c->fail(-1, "ddim timestep %d out of range for T=%d (S must divide the schedule like the reference)", v, a->T);
    std::vector<float> at(total), ap(total), sg(total), s1m(total);
    for (int i = 0; i < total; i++) {
        at[i] = a->alphas_cumprod[ts[i]];
        const double apd = (i == 0) ? (double)a->alphas_cumprod[0] : (double)a->alphas_cumprod[ts[i - 1]];
        ap[i] = (float)apd;
        // sigma computed in float64 from the fp32 alphas like numpy does with a float64 alphas_prev array
        const double atd = (double)at[i];
        sg[i] = (float)((double)a->eta * std::sqrt((1.0 - apd) / (1.0 - atd) * (1.0 - atd / apd)));
        s1m[i] = std::sqrt(1.0f - at[i]);         // np.sqrt on the fp32 tensor (ddim.py:52)
    }
    // scratch: [cond|uncond] f32, x2 [2B], t [total][2B] int64, eps [2B]
    const int nb = cfg ? 2 * B : B;
    const size_t cd_bytes = (size_t)nb * k * u.cfg.context_dim * 4;
    const size_t x2_bytes = (size_t)nb * (n1 / B) * 4, t_bytes = (size_t)total * nb * 8, eps_bytes = x2_bytes;
    const size_t off_x2 = (cd_bytes + 255) & ~(size_t)255, off_t = (off_x2 + x2_bytes + 255) & ~(size_t)255,
                 off_eps = (off_t + t_bytes + 255) & ~(size_t)255;
    RDM_TRY(ensure_bytes(c, &c->samp, &c->samp_bytes, off_eps + eps_bytes));
    int nbe = 0;
    RDM_TRY(prepare_kv(c, cond, cfg ? uncond : nullptr, B, k, &nbe));
    float* x2 = (float*)(c->samp + off_x2); long long* tdev = (long long*)(c->samp + off_t); float* eps = (float*)(c->samp + off_eps);
    {
        std::vector<long long> th((size_t)total * nb);
        for (int i = 0; i < total; i++) for (int j = 0; j < nb; j++) th[(size_t)i * nb + j] = ts[i];
        RDM_CHECK_HIP(c, hipMemcpyAsync(tdev, th.data(), t_bytes, hipMemcpyHostToDevice, c->stream));
        RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));   // th goes out of scope
    }
    RDM_CHECK_HIP(c, hipMemcpyAsync(x2, x_T, n1 * 4, hipMemcpyDeviceToDevice, c->stream));
    int n_logged = 0;
    static const int share_prefix = getenv("RDM_NO_SHARED_PREFIX") ? !atoi(getenv("RDM_NO_SHARED_PREFIX")) : 1;
    // Every sample of a step shares the step's timestep, and the S timesteps are known now: their time-embedding rows (MLP + the 22 emb_layers)
    // are computed ONCE per call as S-row GEMMs into a table, instead of three B-row GEMMs per forward (72 us of a 29 ms forward).  Not in
    // deterministic mode (the rows must come out of the same kernel configuration as rdm_unet_forward's).
    static const int no_emb_table = getenv("RDM_NO_EMB_TABLE") ? atoi(getenv("RDM_NO_EMB_TABLE")) : 0;
    const float* emb_table = nullptr;
    if (!no_emb_table && !c->deterministic) {
        RDM_TRY(ensure_bytes(c, (char**)&u.emb_table, &u.emb_table_bytes, (size_t)total * u.emb_total * 4 + (size_t)total * 8 + 256));
        long long* tuniq = (long long*)((char*)u.emb_table + (((size_t)total * u.emb_total * 4 + 255) & ~(size_t)255));
        std::vector<long long> th(ts.begin(), ts.end());
        RDM_CHECK_HIP(c, hipMemcpyAsync(tuniq, th.data(), (size_t)total * 8, hipMemcpyHostToDevice, c->stream));
        RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        RDM_TRY(run_with_arena(c, u.arena, u.blob, [&](Ops& o) { unet_time_rows(o, u, tuniq, total, u.emb_table); }));
        emb_table = u.emb_table;
    }
    for (int i = 0; i < total; i++) {
        const int index = total - i - 1;
        if (cfg && i == 0) RDM_CHECK_HIP(c, hipMemcpyAsync(x2 + n1, x2, n1 * 4, hipMemcpyDeviceToDevice, c->stream));   // later steps: ddim_step writes both halves
        RDM_TRY(unet_forward_impl(c, x2, (const int64_t*)(tdev + (size_t)index * nb), nullptr, u.kv_cache, nb, k, a->height, a->width, eps, u.ctx_rows,
                                  share_prefix && cfg ? B : 0, emb_table ? emb_table + (size_t)index * u.emb_total : nullptr));     // [x | x], same t: the context-independent prefix runs once
        const bool log = (index % a->log_every_t == 0) || (index == total - 1);
        DdimStepParams p{};
        p.x = x2; p.eps = eps; p.noise = (noise && a->eta != 0.f) ? noise + (size_t)i * n1 : nullptr;
        p.x_prev = x2; p.x_dup = cfg ? x2 + n1 : nullptr; p.pred_x0 = (log && pred_x0_inter) ? pred_x0_inter + (size_t)n_logged * n1 : nullptr;
        p.n_per_batch = n1; p.a_t = at[index]; p.a_prev = ap[index]; p.sigma_t = sg[index]; p.sqrt_one_minus_at = s1m[index];
        p.scale = a->unconditional_guidance_scale; p.temperature = a->temperature; p.cfg = cfg ? 1 : 0;
        RDM_CHECK_HIP(c, launch_ddim_step(p, c->stream));
        if (log) {
            if (x_inter) RDM_CHECK_HIP(c, hipMemcpyAsync(x_inter + (size_t)n_logged * n1, x2, n1 * 4, hipMemcpyDeviceToDevice, c->stream));
            n_logged++;
        }
    }
    RDM_CHECK_HIP(c, hipMemcpyAsync(z_out, x2, n1 * 4, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int rdm_ddpm_sample(rdm_ctx* c, const rdm_ddpm_args* a, const float* x_T, const float* cond, const float* noise,
                    float* z_out) {
    RDM_ENTER(c);
    if (!c || !a || !x_T || !cond || !noise || !z_out) return c ? c->fail(-1, "null argument") : -1;
    UNet& u = c->unet;
    if (!u.loaded) return c->fail(-1, "unet weights not loaded");
    if (a->timesteps < 1 || a->timesteps > a->T) return c->fail(-1, "bad timesteps");
    const int B = a->batch, k = a->k, T = a->timesteps;
    const long long n1 = (long long)B * a->channels * a->height * a->width;
    const size_t cd_bytes = (size_t)B * k * u.cfg.context_dim * 4;
    const size_t off_x = (cd_bytes + 255) & ~(size_t)255, off_t = (off_x + n1 * 4 + 255) & ~(size_t)255,
                 off_eps = (off_t + (size_t)T * B * 8 + 255) & ~(size_t)255;
    RDM_TRY(ensure_bytes(c, &c->samp, &c->samp_bytes, off_eps + n1 * 4));
    int nbe = 0;
    RDM_TRY(prepare_kv(c, cond, nullptr, B, k, &nbe));
    float* x = (float*)(c->samp + off_x); long long* tdev = (long long*)(c->samp + off_t); float* eps = (float*)(c->samp + off_eps);
    {
        std::vector<long long> th((size_t)T * B);
        for (int i = 0; i < T; i++) for (int j = 0; j < B; j++) th[(size_t)i * B + j] = i;
        RDM_CHECK_HIP(c, hipMemcpyAsync(tdev, th.data(), th.size() * 8, hipMemcpyHostToDevice, c->stream));
        RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    }
    RDM_CHECK_HIP(c, hipMemcpyAsync(x, x_T, n1 * 4, hipMemcpyDeviceToDevice, c->stream));
    // the T timesteps' time-embedding rows once per call (as rdm_ddim_sample: every sample of a step shares the step's timestep)
    static const int no_emb_table = getenv("RDM_NO_EMB_TABLE") ? atoi(getenv("RDM_NO_EMB_TABLE")) : 0;
    const float* emb_table = nullptr;
    if (!no_emb_table && !c->deterministic) {
        RDM_TRY(ensure_bytes(c, (char**)&u.emb_table, &u.emb_table_bytes, (size_t)T * u.emb_total * 4 + (size_t)T * 8 + 256));
        long long* tuniq = (long long*)((char*)u.emb_table + (((size_t)T * u.emb_total * 4 + 255) & ~(size_t)255));
        std::vector<long long> th((size_t)T);
        for (int i = 0; i < T; i++) th[i] = i;
        RDM_CHECK_HIP(c, hipMemcpyAsync(tuniq, th.data(), (size_t)T * 8, hipMemcpyHostToDevice, c->stream));
        RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
        RDM_TRY(run_with_arena(c, u.arena, u.blob, [&](Ops& o) { unet_time_rows(o, u, tuniq, T, u.emb_table); }));
        emb_table = u.emb_table;
    }
    for (int n = 0, i = T - 1; i >= 0; i--, n++) {
        RDM_TRY(unet_forward_impl(c, x, (const int64_t*)(tdev + (size_t)i * B), nullptr, u.kv_cache, B, k, a->height, a->width, eps, B, 0,
                                  emb_table ? emb_table + (size_t)i * u.emb_total : nullptr));
        DdpmStepParams p{};
        p.x = x; p.eps = eps; p.noise = noise + (size_t)n * n1; p.x_prev = x; p.n = n1;
        p.sqrt_recip = a->sqrt_recip_alphas_cumprod[i]; p.sqrt_recipm1 = a->sqrt_recipm1_alphas_cumprod[i];
        p.coef1 = a->posterior_mean_coef1[i]; p.coef2 = a->posterior_mean_coef2[i]; p.log_var = a->posterior_log_variance_clipped[i];
        p.clip = a->clip_denoised; p.nonzero = (i != 0); p.temperature = a->temperature;
        RDM_CHECK_HIP(c, launch_ddpm_step(p, c->stream));
    }
    RDM_CHECK_HIP(c, hipMemcpyAsync(z_out, x, n1 * 4, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

// Samples per decoder pass.  Decoding is per sample (GroupNorm statistics included), so a batch may be walked in ranges; a range is
// sized so that the decoder's largest activation stays below 2^30 elements (2 GiB of bf16): the halo convs address an operand through
// 32-bit offsets and leave bigger tensors to the generic implicit GEMM (RARM at 512 sequences per GPU: the seven 128-channel convs of
// the 256 x 256 level on an 8.6 GB activation ran there at 0.30 of peak, 94 of the step's 933 ms).  RDM_VQ_RANGE overrides.
static int vq_range(const VqModel& v, int b) {
    static const int env = getenv("RDM_VQ_RANGE") ? atoi(getenv("RDM_VQ_RANGE")) : 0;
    if (env > 0) return env < b ? env : b;
    const rdm_vq_cfg& c = v.cfg;
    long long per = 1;
    for (int l = 0; l < c.n_ch_mult; l++) {
        const long long r = c.resolution >> l, e = r * r * c.ch * c.ch_mult[l];
        if (e > per) per = e;
        // the level's Upsample output (and the first convs' input at the next finer level) keeps THIS level's channel count at twice the
        // resolution -- the decoder's largest activation (VQ-f4: 256 x 256 x 256 per image, twice the level maximum): a range sized without
        // it reached exactly 2^31 elements and pushed those convs off the 32-bit-offset halo kernels (advisor, round 5)
        if (l >= 1) { const long long u = 4 * r * r * c.ch * c.ch_mult[l]; if (u > per) per = u; }
    }
    long long n = (1LL << 30) / per;
    if (n < 1) n = 1;
    return n < b ? (int)n : b;
}

int rdm_vq_decode(rdm_ctx* c, const float* z, int b, int force_not_quantize, float* img_out, int32_t* indices_out) {
    RDM_ENTER(c);
    if (!c || !z || !img_out) return c ? c->fail(-1, "null argument") : -1;
    if (!c->vq.loaded) return c->fail(-1, "vq weights not loaded");
    if (c->vq.wide) return c->fail(-1, "this first stage has a wide latent (VQGAN-f16): decode from code indices with rdm_vq_decode_indices");
    const rdm_vq_cfg& q = c->vq.cfg;
    const int nb = vq_range(c->vq, b), zr = q.resolution >> (q.n_ch_mult - 1);
    RDM_TRY(ensure_gn_partial(c, nb));
    for (int b0 = 0; b0 < b; b0 += nb) {
        const int n = b - b0 < nb ? b - b0 : nb;
        RDM_TRY(run_with_arena(c, c->vq.arena, c->vq.blob, [&](Ops& o) {
            vq_body(o, c->vq, z + (size_t)b0 * q.z_channels * zr * zr, n, force_not_quantize, img_out + (size_t)b0 * q.out_ch * q.resolution * q.resolution,
                    indices_out ? indices_out + (size_t)b0 * zr * zr : nullptr);
        }));
    }
    return 0;
}
int rdm_vq_quantize(rdm_ctx* c, const float* z, int b, float* zq_out, int32_t* indices_out) {
    RDM_ENTER(c);
    if (!z || !zq_out || b < 1) return c->fail(-1, "rdm_vq_quantize: bad argument");
    VqModel& v = c->vq;
    if (!v.loaded) return c->fail(-1, "vq weights not loaded");
    if (v.wide || v.cfg.kl || v.cfg.embed_dim != 3) return c->fail(-1, "rdm_vq_quantize: needs a VQ first stage with a 3-channel latent (VQ-f4)");
    if (!c->eye3) {       // the quantiser kernel ends in a 3 x 3 map (post_quant_conv in decode): identity + zero bias here
        const float h[12] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f};
        RDM_CHECK_HIP(c, hipMalloc((void**)&c->eye3, sizeof h));
        RDM_CHECK_HIP(c, hipMemcpy(c->eye3, h, sizeof h, hipMemcpyHostToDevice));
    }
    const int zr = v.cfg.resolution >> (v.cfg.n_ch_mult - 1);
    RDM_CHECK_HIP(c, launch_vq_quantize(z, (const float*)(v.blob + v.codebook), v.cfg.n_embed, c->eye3, c->eye3 + 9, zq_out, indices_out, b, zr * zr, 1, c->stream));
    return 0;
}
int rdm_vq_decode_indices(rdm_ctx* c, const int64_t* indices, int b, float* img_out) {
    RDM_ENTER(c);
    if (!indices || !img_out || b < 1) return c->fail(-1, "bad argument");
    if (!c->vq.loaded) return c->fail(-1, "vq weights not loaded");
    if (!c->vq.wide || c->vq.cfg.kl) return c->fail(-1, "rdm_vq_decode_indices needs a VQGAN first stage with a wide latent (z_channels %% 64 == 0)");
    const rdm_vq_cfg& q = c->vq.cfg;
    const int nb = vq_range(c->vq, b), zr = q.resolution >> (q.n_ch_mult - 1);
    RDM_TRY(ensure_gn_partial(c, nb));
    for (int b0 = 0; b0 < b; b0 += nb) {
        const int n = b - b0 < nb ? b - b0 : nb;
        RDM_TRY(run_with_arena(c, c->vq.arena, c->vq.blob, [&](Ops& o) {
            vq_wide_body(o, c->vq, (const long long*)indices + (size_t)b0 * zr * zr, n, img_out + (size_t)b0 * q.out_ch * q.resolution * q.resolution);
        }));
    }
    return 0;
}

int rdm_to_uint8(rdm_ctx* c, const float* img, int b, int ch, int h, int w, uint8_t* out) {
    RDM_ENTER(c);
    if (!c || !img || !out) return -1;
    RDM_CHECK_HIP(c, launch_to_uint8_hwc(img, out, b, ch, h, w, c->stream));
    return 0;
}

int rdm_clip_encode_text(rdm_ctx* c, const int64_t* tokens, int b, float* out) {
    RDM_ENTER(c);
    if (!c || !tokens || !out) return c ? c->fail(-1, "null argument") : -1;
    if (!c->clip.loaded) return c->fail(-1, "clip weights not loaded");
    return run_with_arena(c, c->clip.arena, c->clip.blob, [&](Ops& o) { clip_text_body(o, c->clip, (const long long*)tokens, b, out); });
}
int rdm_clip_encode_image(rdm_ctx* c, const float* image, int b, float* out) {
    RDM_ENTER(c);
    if (!c || !image || !out) return c ? c->fail(-1, "null argument") : -1;
    if (!c->clip.loaded) return c->fail(-1, "clip weights not loaded");
    return run_with_arena(c, c->clip.arena, c->clip.blob, [&](Ops& o) { clip_image_body(o, c->clip, image, b, out); });
}

int rdm_clip_preprocess(rdm_ctx* c, const float* image, int b, int h, int w, float* out) {
    RDM_ENTER(c);
    if (!image || !out || b < 1 || h < 1 || w < 1) return c->fail(-1, "bad argument");
    if (!c->clip.loaded) return c->fail(-1, "clip weights not loaded (the target resolution comes from the clip cfg)");
    RDM_CHECK_HIP(c, launch_clip_preprocess(image, b, h, w, c->clip.cfg.image_resolution, c->clip.cfg.vision_patch_size, out, nullptr, c->stream));
    return 0;
}
int rdm_clip_encode_image_raw(rdm_ctx* c, const float* image, int b, int h, int w, float* out) {
    RDM_ENTER(c);
    if (!image || !out || b < 1 || h < 1 || w < 1) return c->fail(-1, "bad argument");
    if (!c->clip.loaded) return c->fail(-1, "clip weights not loaded");
    return run_with_arena(c, c->clip.arena, c->clip.blob, [&](Ops& o) { clip_image_body(o, c->clip, image, b, out, h, w); });
}

// ---- RARM (kernels in rarm.hip)
struct RarmState { int* pos; int* done; long long* tokens; float* logits; };
static RarmState rarm_state(RarmModel& m, int B2) {
    RarmState st{};
    st.pos = (int*)m.state; st.done = (int*)(m.state + 64); st.tokens = (long long*)(m.state + 256);
    st.logits = (float*)(m.state + 256 + (((size_t)B2 * 8 + 255) & ~(size_t)255));
    return st;
}
static int rarm_prepare(rdm_ctx* c, int B2, int k, const float* context /*[B,k,cd] dev*/, int B, bool cfg) {
    RarmModel& m = c->rarm; const rdm_rarm_cfg& g = m.cfg; const int C = m.C, L = g.sequence_length;
    RDM_TRY(ensure_bytes(c, &m.cache, &m.cache_bytes, (size_t)g.depth * 2 * B2 * L * C * 2));
    RDM_TRY(ensure_bytes(c, &m.ctxkv, &m.ctxkv_bytes, (size_t)B2 * k * m.kv_total * 2));
    RDM_TRY(ensure_bytes(c, &m.state, &m.state_bytes, 256 + (((size_t)B2 * 8 + 255) & ~(size_t)255) + (size_t)B2 * g.vocab_out * 4));
    RDM_CHECK_HIP(c, hipMemsetAsync(m.state, 0, 256, c->stream));
    // neighbours' keys / values of every layer in one GEMM; the unconditional half of a guided batch attends to ZERO neighbours
    // (transformer.py:237-239), whose projections are zero (to_k / to_v have no bias)
    RDM_CHECK_HIP(c, hipMemsetAsync(m.ctxkv, 0, (size_t)B2 * k * m.kv_total * 2, c->stream));
    static const int no_xf = getenv("RDM_NO_RARM_XFUSED") ? atoi(getenv("RDM_NO_RARM_XFUSED")) : 0;
    // Round 5: the per-sequence re-association pays per-sequence operands (G and U^T: 2 x 128 x C bf16 = 393 KB per sequence and layer
    // where the projections' weights are 2.4 MB per layer for ALL sequences): the one-launch form wins while launches are the cost
    // (<= 128 sequences); from RDM_RARM_XGEMM_FROM sequences on (default 384) the decode step takes norm2 + to_q as a GEMM, the k-key attention
    // and to_out + residual as a GEMM (same box, profiles/r05_rarm_sweep.log: 256 sequences 399 img/s fused vs 378 as GEMMs, 512 sequences 484 vs 489).
    static const int xgemm_from = getenv("RDM_RARM_XGEMM_FROM") ? atoi(getenv("RDM_RARM_XGEMM_FROM")) : 384;
    const bool fuse = !no_xf && g.n_heads * k <= 128 && C <= 1024 && C % 64 == 0 && (c->deterministic || B2 < xgemm_from);
    m.xa_B = 0; m.xa_k = 0;
    if (fuse) {
        RDM_TRY(ensure_bytes(c, &m.xa, &m.xa_bytes, (size_t)g.depth * 2 * B * 128 * C * 2));
        m.xa_B = B; m.xa_k = k;
        // partial rows as 8-byte {value, epoch} granules + one monotonic arrival counter per sequence; zeroed per sampling call (tag 0 = never
        // written; counters restart at a multiple of four)
        const size_t pbytes = ((size_t)B2 * 4 * C * 8 + 255) & ~(size_t)255;
        RDM_TRY(ensure_bytes(c, &m.xws, &m.xws_bytes, pbytes + (size_t)B2 * 4));
        RDM_CHECK_HIP(c, hipMemsetAsync(m.xws, 0, pbytes + (size_t)B2 * 4, c->stream));
    }
    return run_with_arena(c, m.arena, m.blob, [&](Ops& o) {
        bf16_t* cb = o.abf((size_t)B * k * g.context_dim);
        if (!o.plan) o.check(launch_cast_f32_bf16(context, cb, (long long)B * k * g.context_dim, c->stream), "cast ctx");
        o.linear(cb, nullptr, g.context_dim, 0, m.kvw, 0, false, B * k, m.kv_total, ACT_NONE, nullptr, (bf16_t*)m.ctxkv);
        // the decode step's cross-attention re-associated per sequence (rarm.hip: rarm_xattn_decode_kernel):
        //   G_l[b][(h,j)][:] = (K_bj restricted to head h) W_q / sqrt(d),   UT_l[b][(h,j)][:] = W_o (V_bj restricted to head h)
        if (fuse) {
            constexpr int NP = 128;
            bf16_t* kexp = o.abf((size_t)B * NP * C); bf16_t* vexp = o.abf((size_t)B * NP * C); bf16_t* wqt = o.abf((size_t)C * C);
            if (!o.plan) {
                const bf16_t* kv = (const bf16_t*)m.ctxkv;
                for (int l = 0; l < g.depth; l++) {
                    const RarmBlk& bk = m.blk[l];
                    bf16_t* G = (bf16_t*)m.xa + ((size_t)l * 2) * B * NP * C; bf16_t* UT = G + (size_t)B * NP * C;
                    o.check(launch_expand_heads(kv + (size_t)l * 2 * C, m.kv_total, B, k, g.n_heads, g.d_head, NP, 1.0f / sqrtf((float)g.d_head), kexp, c->stream), "rarm expand K");
                    o.check(launch_expand_heads(kv + (size_t)l * 2 * C + C, m.kv_total, B, k, g.n_heads, g.d_head, NP, 1.0f, vexp, c->stream), "rarm expand V");
                    o.check(launch_transpose_bf16(o.w<bf16_t>(bk.wq2), wqt, C, C, c->stream), "rarm transpose Wq");
                    IgemmParams p = o.base(B * NP, C, C);
                    p.A0 = kexp; p.C0 = C; p.W = wqt; p.out_bf16 = G;
                    o.check(launch_igemm(p, false, 1, c->stream), "rarm xattn G");
                    IgemmParams q = o.base(B * NP, C, C);
                    q.A0 = vexp; q.C0 = C; q.W = o.w<bf16_t>(bk.wo2); q.out_bf16 = UT;
                    o.check(launch_igemm(q, false, 1, c->stream), "rarm xattn UT");
                }
            }
        }
    });
}
// one decode step for B2 sequences: token at position *pos -> logits of the next token
static int rarm_step(rdm_ctx* c, int B2, int k, int pos_hint = -1 /* host's copy of the position processed (profiling only: the kernels read the device counter) */) {
    RarmModel& m = c->rarm; const rdm_rarm_cfg& g = m.cfg; const int C = m.C, L = g.sequence_length;
    RarmState st = rarm_state(m, B2);
    return run_with_arena(c, m.arena, m.blob, [&](Ops& o) {
        o.single_row = true;                              // a decode step is one row per sequence throughout
        float* x = o.af32((size_t)B2 * C);
        bf16_t* ln = o.abf((size_t)B2 * C); bf16_t* qkv = o.abf((size_t)B2 * 3 * C); bf16_t* ao = o.abf((size_t)B2 * C);
        bf16_t* q2 = o.abf((size_t)B2 * C); bf16_t* ff = o.abf((size_t)B2 * 4 * C);
        if (!o.plan) o.check(launch_rarm_embed(st.tokens, o.w<float>(m.emb), o.w<float>(m.pos), st.pos, x, B2, C, g.vocab_in, c->stream), "rarm embed");
        const float scale = 1.0f / sqrtf((float)g.d_head);
        static const int no_ln3 = getenv("RDM_NO_RARM_LN3") ? atoi(getenv("RDM_NO_RARM_LN3")) : 0;
        const bool ln3_fused = !no_ln3 && m.xa_B > 0 && m.xa_k == k;      // the fused cross-attention kernel also emits norm3 of its output rows
        for (int l = 0; l < g.depth; l++) {
            const RarmBlk& b = m.blk[l];
            if (!o.linear_ln(x, b.ln1g, b.ln1b, C, b.wqkv, 0, false, B2, 3 * C, ACT_NONE, qkv)) {
                o.layernorm(x, 1, b.ln1g, b.ln1b, ln, 0, B2, C);
                o.linear(ln, nullptr, C, 0, b.wqkv, 0, false, B2, 3 * C, ACT_NONE, nullptr, qkv);
            }
            if (!o.plan) {
                RarmAttnParams p{}; p.q = qkv; p.ldq = 3 * C; p.k_new = qkv + C; p.v_new = qkv + 2 * C;
                p.Kc = (bf16_t*)m.cache + ((size_t)l * 2) * B2 * L * C; p.Vc = (bf16_t*)m.cache + ((size_t)l * 2 + 1) * B2 * L * C;
                p.batch_stride = (long long)L * C; p.row_stride = C; p.nkv = L; p.pos = st.pos; p.scale = scale; p.out = ao; p.ldo = C;
                // Head-major cache [B][head][L][64] (round 5): a (head, sequence) block reads ONE contiguous run of (pos + 1) x 128 bytes
                // instead of 128-byte pieces 2 C bytes apart.  The cache is private to this kernel (it appends the new row itself).
                static const int rowmajor = getenv("RDM_RARM_CACHE_ROWMAJOR") ? atoi(getenv("RDM_RARM_CACHE_ROWMAJOR")) : 0;
                if (!rowmajor) { p.row_stride = g.d_head; p.head_stride = (long long)L * g.d_head; }
                // bytes of the K / V cache rows this step reads (positions 0 .. pos): what bounds the launch at big batches
                o.tag = "rarm.cache_attention";
                o.prof_begin(RDM_PROF_ATTENTION, pos_hint >= 0 ? (double)B2 * (pos_hint + 1) * C * 4.0 : 0.0, B2, pos_hint + 1, C);
                o.check(launch_rarm_decode_attention(p, g.n_heads, B2, c->stream), "rarm self attention");
                o.prof_end();
            }
            o.linear(ao, nullptr, C, 0, b.wo1, b.bo1, true, B2, C, ACT_NONE, nullptr, nullptr, x, x);
            if (m.xa_B > 0 && m.xa_k == k) {      // norm2 + to_q + attention over the neighbours + to_out + residual in one launch
                if (!o.plan) {
                    RarmXattnParams xp{}; xp.x = x; xp.ln_g = o.w<float>(b.ln2g); xp.ln_b = o.w<float>(b.ln2b); xp.ln_eps = 1e-5f;
                    xp.G = (const bf16_t*)m.xa + ((size_t)l * 2) * m.xa_B * 128 * C; xp.UT = xp.G + (size_t)m.xa_B * 128 * C;
                    xp.bias = o.w<float>(b.bo2); xp.B2 = B2; xp.Bc = m.xa_B; xp.C = C; xp.NP = 128; xp.heads = g.n_heads; xp.k = k;
                    xp.ws = (float*)m.xws; xp.ws_count = (int*)(m.xws + ((((size_t)B2 * 4 * C * 8) + 255) & ~(size_t)255));
                    if (++m.xepoch == 0u) m.xepoch = 1u;                 // unique per launch, never 0
                    xp.epoch = m.xepoch; xp.no_split = c->deterministic ? 1 : 0;
                    if (ln3_fused) { xp.ln3_g = o.w<float>(b.ln3g); xp.ln3_b = o.w<float>(b.ln3b); xp.ln3_out = ln; }
                    o.check(launch_rarm_xattn_decode(xp, c->stream), "rarm fused cross attention");
                }
            } else {
            if (!o.linear_ln(x, b.ln2g, b.ln2b, C, b.wq2, 0, false, B2, C, ACT_NONE, q2)) {
                o.layernorm(x, 1, b.ln2g, b.ln2b, ln, 0, B2, C);
                o.linear(ln, nullptr, C, 0, b.wq2, 0, false, B2, C, ACT_NONE, nullptr, q2);
            }
            if (!o.plan) {
                RarmAttnParams p{}; p.q = q2; p.ldq = C; p.Kc = (bf16_t*)m.ctxkv + (size_t)l * 2 * C; p.Vc = (bf16_t*)m.ctxkv + (size_t)l * 2 * C + C;
                p.batch_stride = (long long)k * m.kv_total; p.row_stride = m.kv_total; p.nkv = k; p.scale = scale; p.out = ao; p.ldo = C;
                o.check(launch_rarm_decode_attention(p, g.n_heads, B2, c->stream), "rarm cross attention");
            }
            o.linear(ao, nullptr, C, 0, b.wo2, b.bo2, true, B2, C, ACT_NONE, nullptr, nullptr, x, x);
            }
            if (ln3_fused) {      // norm3 left the cross-attention kernel with the finished rows: a plain GEGLU GEMM on the bf16 operand
                o.linear(ln, nullptr, C, 0, b.wff1, b.bff1, true, B2, 8 * C, ACT_GEGLU, nullptr, ff);
            } else if (!o.linear_ln(x, b.ln3g, b.ln3b, C, b.wff1, b.bff1, true, B2, 8 * C, ACT_GEGLU, ff)) {
                o.layernorm(x, 1, b.ln3g, b.ln3b, ln, 0, B2, C);
                o.linear(ln, nullptr, C, 0, b.wff1, b.bff1, true, B2, 8 * C, ACT_GEGLU, nullptr, ff);
            }
            o.linear(ff, nullptr, 4 * C, 0, b.wff2, b.bff2, true, B2, C, ACT_NONE, nullptr, nullptr, x, x);
        }
        if (!o.plan) o.check(launch_cast_f32_bf16(x, ln, (long long)B2 * C, c->stream), "cast x");
        o.linear(ln, nullptr, C, 0, m.wpo, m.bpo, true, B2, g.vocab_out, ACT_NONE, nullptr, nullptr, st.logits);
    });
}
static int rarm_check(rdm_ctx* c, int b, int k, int positions) {
    RarmModel& m = c->rarm;
    if (!m.loaded) return c->fail(-1, "rarm weights not loaded");
    if (b < 1 || k < 1 || k > 1024) return c->fail(-1, "bad rarm shape b=%d k=%d", b, k);
    if (positions < 1 || positions > m.cfg.sequence_length)
        return c->fail(-1, "%d positions exceed the positional encoding (sequence_length %d)", positions, m.cfg.sequence_length);
    return 0;
}
// column i of a [b, t] int64 token matrix -> the current-token buffer (optionally duplicated for the unconditional half)
static int rarm_set_tokens(rdm_ctx* c, RarmState& st, const int64_t* tokens, int b, int t, int i, bool dup) {
    RDM_CHECK_HIP(c, hipMemcpy2DAsync(st.tokens, 8, tokens + i, (size_t)t * 8, 8, b, hipMemcpyDeviceToDevice, c->stream));
    if (dup) RDM_CHECK_HIP(c, hipMemcpy2DAsync(st.tokens + b, 8, tokens + i, (size_t)t * 8, 8, b, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
long long rdm_rarm_manifest(const rdm_rarm_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes) {
    if (!cfg || cfg->depth < 1 || cfg->n_heads < 1) return -1;
    RarmModel m; Manifest mf; build_rarm(m, *cfg, mf);
    return write_manifest(mf, buf, buflen, blob_bytes);
}
int rdm_load_rarm(rdm_ctx* c, const rdm_rarm_cfg* cfg, const void* packed, size_t nbytes) {
    RDM_ENTER(c);
    if (!cfg) return c->fail(-1, "null cfg");
    if (cfg->d_head != 64 || cfg->context_dim % 64 || cfg->sequence_length > 1024 || cfg->vocab_out % 2 || cfg->depth < 1)
        return c->fail(-1, "unsupported rarm cfg: d_head == 64, context_dim %% 64 == 0, sequence_length <= 1024, even vocab_out required");
    Manifest mf; build_rarm(c->rarm, *cfg, mf);
    return load_blob(c, c->rarm, mf, packed, nbytes);
}
int rdm_rarm_forward(rdm_ctx* c, const int64_t* tokens, int b, int t, const float* context, int k, float* logits_out) {
    RDM_ENTER(c);
    if (!tokens || !context || !logits_out) return c->fail(-1, "null argument");
    RDM_TRY(rarm_check(c, b, k, t));
    RarmModel& m = c->rarm;
    RDM_TRY(rarm_prepare(c, b, k, context, b, false));
    RarmState st = rarm_state(m, b);
    const size_t V = m.cfg.vocab_out;
    for (int i = 0; i < t; i++) {
        RDM_TRY(rarm_set_tokens(c, st, tokens, b, t, i, false));
        RDM_CHECK_HIP(c, launch_set_int(st.pos, i, c->stream));
        RDM_TRY(rarm_step(c, b, k));
        RDM_CHECK_HIP(c, hipMemcpy2DAsync(logits_out + (size_t)i * V, (size_t)t * V * 4, st.logits, V * 4, V * 4, b, hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}
int rdm_rarm_sample(rdm_ctx* c, const rdm_rarm_sample_args* a, const int64_t* cond_tokens, const float* context, const float* uniforms,
                    int64_t* tokens_out) {
    RDM_ENTER(c);
    if (!a || !cond_tokens || !context || !uniforms || !tokens_out) return c->fail(-1, "null argument");
    if (a->cond_len < 1 || a->steps < 1 || a->temperature <= 0.f) return c->fail(-1, "bad sampling arguments");
    RDM_TRY(rarm_check(c, a->batch, a->k, a->cond_len + a->steps - 1));
    RarmModel& m = c->rarm;
    const bool cfg = a->guidance_scale > 1.0f;
    const int B = a->batch, B2 = cfg ? 2 * B : B, k = a->k;
    RDM_TRY(rarm_prepare(c, B2, k, context, B, cfg));
    RarmState st = rarm_state(m, B2);
    for (int i = 0; i < a->cond_len; i++) {                    // prefill the conditioning tokens (the sos token)
        RDM_TRY(rarm_set_tokens(c, st, cond_tokens, B, a->cond_len, i, cfg));
        RDM_CHECK_HIP(c, launch_set_int(st.pos, i, c->stream));
        if (i + 1 < a->cond_len) RDM_TRY(rarm_step(c, B2, k, i));
    }
    RarmSampleParams sp{}; sp.logits = st.logits; sp.vocab = m.cfg.vocab_out; sp.B = B; sp.cfg = cfg ? 1 : 0; sp.scale = a->guidance_scale;
    sp.temperature = a->temperature; sp.top_k = a->top_k > 0 ? a->top_k : m.cfg.vocab_out; sp.uniforms = uniforms; sp.pos = st.pos;
    sp.pos0 = a->cond_len - 1; sp.steps = a->steps; sp.tokens_out = (long long*)tokens_out; sp.next_tokens = st.tokens; sp.done = st.done;
    for (int s_ = 0; s_ < a->steps; s_++) {                    // every step: the same launches (the step counter lives on the device)
        RDM_TRY(rarm_step(c, B2, k, a->cond_len - 1 + s_));
        RDM_CHECK_HIP(c, launch_rarm_sample(sp, c->stream));
    }
    return 0;
}

// ---- retrieval (kernels in knn.hip)
int rdm_db_load(rdm_ctx* c, const void* emb, long long n, int dim, int dtype, int is_device) {
    RDM_ENTER(c);
    if (!c || !emb) return -1;
    const char* msg = knn_load(c->db, emb, n, dim, dtype, is_device, c->stream);
    return msg ? c->fail(-5, "rdm_db_load: %s", msg) : 0;
}
long long rdm_db_size(rdm_ctx* c) { return c ? c->db.n : -1; }
static int knn_entry(rdm_ctx* c, const float* q, int b, int k, uint32_t* idx_out, float* score_out, double* score64_out);
int rdm_knn(rdm_ctx* c, const float* q, int b, int k, uint32_t* idx_out, float* score_out) { return knn_entry(c, q, b, k, idx_out, score_out, nullptr); }
int rdm_knn_f64(rdm_ctx* c, const float* q, int b, int k, uint32_t* idx_out, double* score_out) { return knn_entry(c, q, b, k, idx_out, nullptr, score_out); }
static int knn_entry(rdm_ctx* c, const float* q, int b, int k, uint32_t* idx_out, float* score_out, double* score64_out) {
    RDM_ENTER(c);
    if (!c || !q || !idx_out) return c ? c->fail(-1, "null argument") : -1;
    const bool prof = (c->prof >> RDM_PROF_KNN) & 1u;
    if (prof) {     // whole search (query prep + database scan + candidate merge); work = bytes of the database passes
        rdm_ctx::ProfRec r; r.a = c->prof_event(); r.b = c->prof_event(); r.kind = RDM_PROF_KNN;
        r.flops = (double)((b + 63) / 64) * (double)c->db.n * c->db.dim * 2.0;
        r.tag = "knn"; r.d0 = b; r.d1 = k; r.d2 = c->db.dim;
        hipEventRecord(r.a, c->stream); c->prof_recs.push_back(r);
    }
    const char* msg = knn_search(c->db, q, b, k, idx_out, score_out, score64_out, c->stream);
    if (prof) hipEventRecord(c->prof_recs.back().b, c->stream);
    return msg ? c->fail(-5, "rdm_knn: %s", msg) : 0;
}
int rdm_knn_last_fallback(rdm_ctx* c) { return c ? c->db.last_fallbacks : -1; }
int rdm_db_gather(rdm_ctx* c, const uint32_t* idx, long long n_idx, float* out) {
    RDM_ENTER(c);
    if (!c || !idx || !out) return -1;
    const char* msg = knn_gather(c->db, idx, n_idx, out, c->stream);
    return msg ? c->fail(-5, "rdm_db_gather: %s", msg) : 0;
}

// ---- profiler
// ------------------------------------------------------------------------------------ RCCL wrappers (include/rdm_hip.h, multi-GPU)
// RCCL is resolved at run time: the ABI types are spelled out here (ncclUniqueId = 128 bytes, ncclComm_t = pointer, ncclInt8 = 0)
namespace {
struct RcclId { char b[128]; };
typedef int (*fn_get_id)(RcclId*);
typedef int (*fn_init_rank)(void**, int, RcclId, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
struct RcclFns { fn_get_id get_id; fn_init_rank init_rank; fn_all_gather all_gather; fn_all_reduce all_reduce; fn_destroy destroy; fn_errstr errstr; };
int rccl_load(rdm_ctx* c, RcclFns& f) {
    if (!c->rccl_lib) {
        c->rccl_lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!c->rccl_lib) c->rccl_lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!c->rccl_lib) return c->fail(-5, "rdm_comm: cannot load librccl.so (%s)", dlerror());
    }
    f.get_id = (fn_get_id)dlsym(c->rccl_lib, "ncclGetUniqueId"); f.init_rank = (fn_init_rank)dlsym(c->rccl_lib, "ncclCommInitRank");
    f.all_gather = (fn_all_gather)dlsym(c->rccl_lib, "ncclAllGather"); f.destroy = (fn_destroy)dlsym(c->rccl_lib, "ncclCommDestroy");
    f.all_reduce = (fn_all_reduce)dlsym(c->rccl_lib, "ncclAllReduce");
    f.errstr = (fn_errstr)dlsym(c->rccl_lib, "ncclGetErrorString");
    if (!f.get_id || !f.init_rank || !f.all_gather || !f.destroy) return c->fail(-5, "rdm_comm: librccl.so lacks the expected entry points");
    return 0;
}
}  // namespace
int rdm_comm_unique_id(rdm_ctx* c, void* id128) {
    RDM_ENTER(c);
    if (!id128) return c->fail(-1, "rdm_comm_unique_id: null id buffer");
    RcclFns f{}; RDM_TRY(rccl_load(c, f));
    RcclId id; const int r = f.get_id(&id);
    if (r != 0) return c->fail(-5, "ncclGetUniqueId failed: %s", f.errstr ? f.errstr(r) : "?");
    memcpy(id128, id.b, 128);
    return 0;
}
int rdm_comm_init(rdm_ctx* c, const void* id128, int rank, int world) {
    RDM_ENTER(c);
    if (!id128 || world < 1 || rank < 0 || rank >= world) return c->fail(-1, "rdm_comm_init: bad arguments (rank %d of %d)", rank, world);
    if (c->comm) return c->fail(-1, "rdm_comm_init: communicator already initialised (rdm_comm_destroy first)");
    RcclFns f{}; RDM_TRY(rccl_load(c, f));
    RcclId id; memcpy(id.b, id128, 128);
    void* comm = nullptr; const int r = f.init_rank(&comm, world, id, rank);
    if (r != 0) return c->fail(-5, "ncclCommInitRank failed: %s", f.errstr ? f.errstr(r) : "?");
    c->comm = comm; c->comm_world = world;
    return 0;
}
int rdm_comm_all_gather(rdm_ctx* c, const void* send, void* recv, size_t nbytes) {
    RDM_ENTER(c);
    if (!c->comm) return c->fail(-1, "rdm_comm_all_gather: no communicator (rdm_comm_init)");
    if (!send || !recv) return c->fail(-1, "rdm_comm_all_gather: null buffer");
    RcclFns f{}; RDM_TRY(rccl_load(c, f));
    const int r = f.all_gather(send, recv, nbytes, /*ncclInt8*/ 0, c->comm, c->stream);
    if (r != 0) return c->fail(-5, "ncclAllGather failed: %s", f.errstr ? f.errstr(r) : "?");
    return 0;
}
int rdm_comm_all_reduce_f32(rdm_ctx* c, float* buf, size_t count, int average) {
    RDM_ENTER(c);
    if (!c->comm) return c->fail(-1, "rdm_comm_all_reduce_f32: no communicator (rdm_comm_init)");
    if (!buf) return c->fail(-1, "rdm_comm_all_reduce_f32: null buffer");
    RcclFns f{}; RDM_TRY(rccl_load(c, f));
    if (!f.all_reduce) return c->fail(-5, "rdm_comm: librccl.so lacks ncclAllReduce");
    const int r = f.all_reduce(buf, buf, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, c->stream);
    if (r != 0) return c->fail(-5, "ncclAllReduce failed: %s", f.errstr ? f.errstr(r) : "?");
    if (average && c->comm_world > 1) RDM_CHECK_HIP(c, launch_scale_f32(buf, (long long)count, 1.0f / (float)c->comm_world, c->stream));
    return 0;
}
int rdm_comm_destroy(rdm_ctx* c) {
    RDM_ENTER(c);
    if (!c->comm) return 0;
    RcclFns f{}; RDM_TRY(rccl_load(c, f));
    RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    f.destroy(c->comm); c->comm = nullptr; c->comm_world = 0;
    return 0;
}

int rdm_prof_enable(rdm_ctx* c, int kind_mask) {
    if (!c) return -1;
    c->prof = (unsigned)kind_mask;
    return 0;
}
int rdm_prof_collect(rdm_ctx* c, int kind, long long* launches, double* ms, double* flops) {
    RDM_ENTER(c);
    if (!c) return -1;
    RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    long long n = 0; double t = 0, f = 0;
    for (auto& r : c->prof_recs) {
        if (r.kind != kind) continue;
        float e = 0; hipEventElapsedTime(&e, r.a, r.b);
        n++; t += e; f += r.flops;
    }
    if (launches) *launches = n; if (ms) *ms = t; if (flops) *flops = f;
    return 0;
}
int rdm_debug_tap(rdm_ctx* c, void* buf, size_t nbytes, int block, int sub) {
    if (!c) return -1;
    c->tap_buf = buf; c->tap_bytes = buf ? nbytes : 0; c->tap_block = buf ? block : -1; c->tap_sub = buf ? sub : 0;
    return 0;
}
int rdm_prof_dump(rdm_ctx* c, const char* path) {
    RDM_ENTER(c);
    if (!c || !path) return -1;
    RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    FILE* f = fopen(path, "w");
    if (!f) return c->fail(-1, "rdm_prof_dump: cannot open %s", path);
    fprintf(f, "kind,tag,d0,d1,d2,ms,work\n");
    for (auto& r : c->prof_recs) {
        float e = 0; hipEventElapsedTime(&e, r.a, r.b);
        fprintf(f, "%d,%s,%d,%d,%d,%.6f,%.6e\n", r.kind, r.tag ? r.tag : "", r.d0, r.d1, r.d2, e, r.flops);
    }
    fclose(f);
    return 0;
}
int rdm_op_ffn_fused(rdm_ctx* c, const void* l3, const void* t2, const void* xin, const void* w1, const float* b1, const void* wf, const float* bf,
                     void* out, int M, int C) {
    RDM_ENTER(c);
    if (!l3 || !t2 || !xin || !w1 || !b1 || !wf || !bf || !out) return c->fail(-1, "rdm_op_ffn_fused: null argument");
    if (!ffn_fused_supported(M, C)) return c->fail(-5, "rdm_op_ffn_fused: C = 384 and M %% 128 == 0 only (M = %d, C = %d)", M, C);
    RDM_TRY(ensure_bytes(c, &c->wfrag_tmp, &c->wfrag_tmp_bytes, ffn_fused_scratch_bytes(C)));
    static const int op_cache = getenv("RDM_OP_FRAG_CACHE") ? atoi(getenv("RDM_OP_FRAG_CACHE")) : 0;     // dev-only (tools/ffn_bench.py): the caller promises constant weights
    static const void* last_w1 = nullptr; static const void* last_wf = nullptr; static const void* last_buf = nullptr;
    const bool repack = !(op_cache && last_w1 == w1 && last_wf == wf && last_buf == c->wfrag_tmp);
    RDM_CHECK_HIP(c, launch_ffn_fused((const bf16_t*)l3, (const bf16_t*)t2, (const bf16_t*)xin, (const bf16_t*)w1, b1, (const bf16_t*)wf, bf, (bf16_t*)out, M, C,
                                      c->wfrag_tmp, repack, c->stream));
    last_w1 = w1; last_wf = wf; last_buf = c->wfrag_tmp;
    return 0;
}
int rdm_debug_counter(rdm_ctx* c, int which, unsigned long long* value) {
    RDM_ENTER(c);
    if (!value) return c->fail(-1, "rdm_debug_counter: null argument");
    RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    if (which == 0) { *value = rarm_xsplit_stale_count(); return 0; }
    return c->fail(-2, "rdm_debug_counter: unknown counter %d", which);
}
// Box calibration (calib.hip): fixed probes, independent of every product kernel.  buf: caller's device scratch.
int rdm_calib_probe(rdm_ctx* c, void* buf, size_t buf_bytes, double mfma_ms, size_t stream_bytes, int stream_reps, double* mfma_tflops, double* stream_gbps) {
    RDM_ENTER(c);
    if (!buf || buf_bytes < (1u << 20) || buf_bytes < 2 * stream_bytes || stream_bytes % 16 || stream_reps < 1 || !(mfma_ms > 0))
        return c->fail(-2, "rdm_calib_probe: buf must hold max(1 MiB, 2 * stream_bytes), stream_bytes a multiple of 16, stream_reps >= 1, mfma_ms > 0");
    RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    RDM_CHECK_HIP(c, run_calib_probes(buf, mfma_ms, stream_bytes, stream_reps, mfma_tflops, stream_gbps, c->stream));
    return 0;
}
int rdm_prof_reset(rdm_ctx* c) {
    RDM_ENTER(c);
    if (!c) return -1;
    RDM_CHECK_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& r : c->prof_recs) { c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b); }
    c->prof_recs.clear();
    return 0;
}

// ---- operator-level wrappers for the parity tests
static int op_linear_impl(rdm_ctx* c, const void* a, const void* w, const float* bias, const void* res, void* out, float* out_f32,
                          int M, int N, int K, int act, float alpha, const float* rowvec, int rows_per_group);
int rdm_op_linear(rdm_ctx* c, const void* a, const void* w, const float* bias, const void* res, void* out, float* out_f32,
                  int M, int N, int K, int act, float alpha) {
    return op_linear_impl(c, a, w, bias, res, out, out_f32, M, N, K, act, alpha, nullptr, 1);
}
int rdm_op_linear_rowvec(rdm_ctx* c, const void* a, const void* w, const float* bias, const float* rowvec, int rows_per_group, const void* res,
                         void* out, int M, int N, int K) {
    if (c && (!rowvec || rows_per_group < 1)) return c->fail(-1, "rdm_op_linear_rowvec: rowvec and rows_per_group >= 1 required");
    return op_linear_impl(c, a, w, bias, res, out, nullptr, M, N, K, ACT_NONE, 1.0f, rowvec, rows_per_group);
}
static int op_linear_impl(rdm_ctx* c, const void* a, const void* w, const float* bias, const void* res, void* out, float* out_f32,
                          int M, int N, int K, int act, float alpha, const float* rowvec, int rows_per_group) {
    RDM_ENTER(c);
    if (!c) return -1;
    if (M <= 128 && alpha == 1.0f && !c->deterministic && !rowvec) {       // same dispatch as the executors (Ops::linear): decode-sized batches take the skinny kernel.  Deterministic mode: the
                                                                // op has no notion of "one row per sample", so it never takes the batch-dependent shortcut (the tiled kernel for every M)
        SgemmParams q{}; q.A = (const bf16_t*)a; q.lda = K; q.W = (const bf16_t*)w; q.M = M; q.N = N; q.K = K; q.bias = bias; q.act = act;
        q.res_bf16 = (const bf16_t*)res; q.out_f32 = out_f32; q.out_bf16 = (bf16_t*)out; q.ldo = act == ACT_GEGLU ? N / 2 : N;
        if (sgemm_supported(q)) { RDM_CHECK_HIP(c, launch_sgemm(q, c->stream)); return 0; }
    }
    static const int mg_any = getenv("RDM_MGEMM_ANY") ? atoi(getenv("RDM_MGEMM_ANY")) : 0;      // dev / tests: plain ops of >= mg_any rows on the mid-size GEMM (mgemm.hip)
    if (mg_any > 0 && M >= mg_any && alpha == 1.0f && !rowvec && act != ACT_GEGLU) {
        SgemmParams q{}; q.A = (const bf16_t*)a; q.lda = K; q.W = (const bf16_t*)w; q.M = M; q.N = N; q.K = K; q.bias = bias; q.act = act;
        q.res_bf16 = (const bf16_t*)res; q.out_f32 = out_f32; q.out_bf16 = (bf16_t*)out; q.ldo = N;
        if (mgemm_supported(q)) { RDM_CHECK_HIP(c, launch_mgemm(q, c->stream)); return 0; }
    }
    IgemmParams p{}; p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.ldo = (act == ACT_GEGLU) ? N / 2 : N; p.zero_page = c->zero_page;
    p.Hin = p.Win = p.Hout = p.Wout = 1; p.stride = 1; p.rows_per_sample = 1;
    p.A0 = (const bf16_t*)a; p.C0 = K; p.W = (const bf16_t*)w; p.bias = bias; p.res_bf16 = (const bf16_t*)res;
    p.out_bf16 = (bf16_t*)out; p.out_f32 = out_f32; p.act = act;
    if (rowvec) { p.rowvec = rowvec; p.rowvec_ld = N; p.rows_per_sample = rows_per_group; }
    {
        IgemmParams t = p; t.Wfrag = p.W;
        if (!c->deterministic && lin4_supported(t, 1)) {
            static const int op_cache = getenv("RDM_OP_FRAG_CACHE") ? atoi(getenv("RDM_OP_FRAG_CACHE")) : 0;   // dev-only: the caller promises constant weights
            if (op_cache) p.Wfrag = c->frag_for_lin(p.W, N, K, act == ACT_GEGLU);
            else {
                RDM_TRY(ensure_bytes(c, &c->wfrag_tmp, &c->wfrag_tmp_bytes, (size_t)N * K * 2));
                RDM_CHECK_HIP(c, launch_lin_w_fragpack(p.W, (bf16_t*)c->wfrag_tmp, N, K, K, act == ACT_GEGLU, c->stream));
                p.Wfrag = (const bf16_t*)c->wfrag_tmp;
            }
        }
    }
    RDM_CHECK_HIP(c, launch_igemm(p, false, 1, c->stream));
    return 0;
}
int rdm_op_linear_ln(rdm_ctx* c, const void* x, const void* w, const float* bias, const float* gamma, const float* beta, void* out,
                     int M, int N, int K, int act, float eps) {
    RDM_ENTER(c);
    if (!x || !w || !gamma || !beta || !out) return c->fail(-1, "rdm_op_linear_ln: null argument");
    IgemmParams p{}; p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.ldo = (act == ACT_GEGLU) ? N / 2 : N; p.zero_page = c->zero_page;
    p.Hin = p.Win = p.Hout = p.Wout = 1; p.stride = 1; p.rows_per_sample = 1;
    p.A0 = (const bf16_t*)x; p.C0 = K; p.W = (const bf16_t*)w; p.out_bf16 = (bf16_t*)out; p.act = act; p.l4_any_tiles = 1;
    p.ln_inv_c = 1.0f / (float)K; p.ln_eps = eps;
    IgemmParams t = p; t.Wfrag = p.W; t.ln_sb = gamma;
    if (!lin4_supported(t, 1)) return c->fail(-5, "rdm_op_linear_ln: shape not taken by the folded-LayerNorm kernel (M %% 128/256, N %% 384/192, K %% 64, K >= 128)");
    // caller-owned weights: packed per call into the scratch copy [fragments | (s, b') table]
    const size_t wbytes = ((size_t)N * K * 2 + 255) & ~(size_t)255;
    RDM_TRY(ensure_bytes(c, &c->wfrag_tmp, &c->wfrag_tmp_bytes, wbytes + (size_t)N * 8));
    float* sb = (float*)(c->wfrag_tmp + wbytes);
    RDM_CHECK_HIP(c, launch_lin_w_fragpack(p.W, (bf16_t*)c->wfrag_tmp, N, K, K, act == ACT_GEGLU, c->stream, gamma));
    RDM_CHECK_HIP(c, launch_lin_ln_sb(p.W, gamma, beta, bias, sb, N, K, c->stream));
    p.Wfrag = (const bf16_t*)c->wfrag_tmp; p.ln_sb = sb;
    RDM_CHECK_HIP(c, launch_lin4(p, c->stream));
    return 0;
}
int rdm_op_conv3x3(rdm_ctx* c, const void* x0, const void* x1, int C0, int C1, const void* w, const float* bias,
                   const float* rowvec, int rowvec_ld, const void* res, void* out, int B, int Hin, int Win, int N, int stride,
                   int ups) {
    RDM_ENTER(c);
    if (!c) return -1;
    const int Hout = ups ? Hin * 2 : (stride == 2 ? Hin / 2 : Hin), Wout = ups ? Win * 2 : (stride == 2 ? Win / 2 : Win);
    IgemmParams p{}; p.M = B * Hout * Wout; p.N = N; p.K = 9 * (C0 + C1); p.alpha = 1.f; p.ldo = N; p.zero_page = c->zero_page;
    p.A0 = (const bf16_t*)x0; p.A1 = (const bf16_t*)x1; p.C0 = C0; p.C1 = C1; p.W = (const bf16_t*)w; p.bias = bias;
    p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout; p.stride = stride; p.ups = ups;
    p.rowvec = rowvec; p.rowvec_ld = rowvec_ld; p.rows_per_sample = Hout * Wout; p.res_bf16 = (const bf16_t*)res; p.out_bf16 = (bf16_t*)out;
    {   // the fused-upsample conv by output phase (as Ops::conv3 runs it inside the models); the phase weights are rebuilt per call
        static const bool no_phase = getenv("RDM_NO_UPS_PHASE") != nullptr;
        if (ups && !no_phase && !x1 && C1 == 0 && C0 % 64 == 0 && N % 8 == 0 && !rowvec && !res && stride == 1) {
            RDM_TRY(ensure_bytes(c, &c->wfrag_tmp, &c->wfrag_tmp_bytes, (size_t)16 * N * C0 * 2));
            RDM_CHECK_HIP(c, launch_conv_phase_weights(p.W, (bf16_t*)c->wfrag_tmp, N, C0, c->stream));
            IgemmParams q{}; q.M = B * Hin * Win; q.N = N; q.K = 4 * C0; q.alpha = 1.f; q.ldo = N; q.zero_page = c->zero_page;
            q.A0 = p.A0; q.C0 = C0; q.W = (const bf16_t*)c->wfrag_tmp; q.bias = bias; q.out_bf16 = (bf16_t*)out; q.phase2 = 1;
            q.Hin = Hin; q.Win = Win; q.Hout = Hout; q.Wout = Wout; q.stride = 1; q.rows_per_sample = Hin * Win; q.sW = (long long)N * 4 * C0;
            RDM_CHECK_HIP(c, launch_igemm(q, true, 4, c->stream));
            return 0;
        }
    }
    const bool det_generic = c->deterministic && ((Hout * Wout) % 256 != 0);
    if (det_generic) { RDM_CHECK_HIP(c, launch_igemm(p, true, 1, c->stream)); return 0; }
    const int ks = c->deterministic ? 1 : conv_halo_ksplit(p);
    if (ks > 1) { RDM_TRY(ensure_bytes(c, &c->splitk_ws, &c->splitk_ws_bytes, (size_t)ks * p.M * N * 4)); p.ksplit = ks; p.ws = (float*)c->splitk_ws; }
    static const int op_cache = getenv("RDM_OP_FRAG_CACHE") ? atoi(getenv("RDM_OP_FRAG_CACHE")) : 0;     // dev-only (tools/conv_bench.py): the caller promises constant weights
    const bool halo = conv_halo_supported(p) || conv_halo4_strip_supported(p);
    if (halo && op_cache) p.Wfrag = c->frag_for(p.W, N, C0 + C1);
    else if (halo) {
        RDM_TRY(ensure_bytes(c, &c->wfrag_tmp, &c->wfrag_tmp_bytes, (size_t)N * p.K * 2));
        RDM_CHECK_HIP(c, launch_conv_w_fragpack(p.W, (bf16_t*)c->wfrag_tmp, N, C0 + C1, c->stream));
        p.Wfrag = (const bf16_t*)c->wfrag_tmp;
    }
    RDM_CHECK_HIP(c, launch_conv3x3(p, c->stream));
    return 0;
}
int rdm_op_rarm_sampler(rdm_ctx* c, const float* logits, int b, int vocab, int cfg, float guidance_scale, float temperature, int top_k,
                        const float* uniforms, int64_t* tokens_out) {
    RDM_ENTER(c);
    if (!logits || !uniforms || !tokens_out || b < 1 || vocab < 1) return c->fail(-1, "rdm_op_rarm_sampler: bad arguments");
    if (!(temperature > 0.f)) return c->fail(-1, "rdm_op_rarm_sampler: temperature must be positive");
    RDM_TRY(ensure_bytes(c, &c->samp, &c->samp_bytes, (size_t)b * 8 + 64));
    int* pos = (int*)c->samp; int* done = pos + 1; long long* next = (long long*)(c->samp + 64);
    RDM_CHECK_HIP(c, launch_set_int(pos, 0, c->stream));
    RDM_CHECK_HIP(c, launch_set_int(done, 0, c->stream));
    RarmSampleParams sp{}; sp.logits = logits; sp.vocab = vocab; sp.B = b; sp.cfg = cfg ? 1 : 0; sp.scale = guidance_scale; sp.temperature = temperature;
    sp.top_k = top_k > 0 ? top_k : vocab; sp.uniforms = uniforms; sp.pos = pos; sp.pos0 = 0; sp.steps = 1; sp.tokens_out = (long long*)tokens_out;
    sp.next_tokens = next; sp.done = done;
    RDM_CHECK_HIP(c, launch_rarm_sample(sp, c->stream));
    return 0;
}
// ---- backward ops (kernels in backward.hip)
int rdm_op_conv3x3_dgrad(rdm_ctx* c, const void* dy, const void* w, void* dx, int B, int H, int W, int C, int N) {
    RDM_ENTER(c);
    if (!dy || !w || !dx || C % 64 || N % 64) return c->fail(-1, "rdm_op_conv3x3_dgrad: null argument or channel counts not multiples of 64");
    // dX = conv3x3(dY, W~): the flipped, transposed filter through the FORWARD kernel (input channels N, output channels C)
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, (size_t)N * 9 * C * 2));
    RDM_CHECK_HIP(c, launch_conv_w_dgrad((const bf16_t*)w, (bf16_t*)c->bwd_tmp, N, C, c->stream));
    return rdm_op_conv3x3(c, dy, nullptr, N, 0, c->bwd_tmp, nullptr, nullptr, 0, nullptr, dx, B, H, W, C, 1, 0);
}
int rdm_op_conv3x3_wgrad(rdm_ctx* c, const void* x, const void* dy, float* dw, int B, int H, int W, int C, int N) {
    RDM_ENTER(c);
    if (!x || !dy || !dw || C % 2 || N < 1) return c->fail(-1, "rdm_op_conv3x3_wgrad: bad arguments");
    if (conv_wgrad_tn_supported(B, H, W, C, N)) {
        RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, conv_wgrad_tn_scratch_bytes(B, H, W, C, N) + 256));
        RDM_CHECK_HIP(c, launch_wgrad_tn((const bf16_t*)dy, N, (const bf16_t*)x, C, dw, (long long)B * H * W, N, C, 9, H, W, c->bwd_tmp, c->zero_page, c->stream));
        return 0;
    }
    const size_t need = conv_wgrad_scratch_bytes(B, H, W, C, N, nullptr, nullptr, nullptr, nullptr, nullptr);
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, need));
    RDM_CHECK_HIP(c, launch_conv_wgrad((const bf16_t*)x, (const bf16_t*)dy, dw, B, H, W, C, N, c->bwd_tmp, c->zero_page, c->stream));
    return 0;
}
int rdm_op_groupnorm_bwd_add(rdm_ctx* c, const void* x, const void* dy, const float* gamma, const float* beta, int B, int HW, int C, float eps, int silu,
                             const void* residual, void* dx, float* dgamma, float* dbeta) {
    RDM_ENTER(c);
    if (!x || !dy || !gamma || !beta || !dx || !dgamma || !dbeta || C % 32) return c->fail(-1, "rdm_op_groupnorm_bwd: bad arguments");
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, groupnorm_bwd_scratch_bytes(B, HW, C, 32)));
    RDM_CHECK_HIP(c, launch_groupnorm_bwd((const bf16_t*)x, (const bf16_t*)dy, gamma, beta, B, HW, C, 32, eps, silu, (float*)c->bwd_tmp, (bf16_t*)dx,
                                          dgamma, dbeta, c->stream, (const bf16_t*)residual));
    return 0;
}
int rdm_op_groupnorm_bwd(rdm_ctx* c, const void* x, const void* dy, const float* gamma, const float* beta, int B, int HW, int C, float eps, int silu,
                         void* dx, float* dgamma, float* dbeta) {
    return rdm_op_groupnorm_bwd_add(c, x, dy, gamma, beta, B, HW, C, eps, silu, nullptr, dx, dgamma, dbeta);
}
int rdm_op_layernorm_bwd_add(rdm_ctx* c, const void* x, const void* dy, const float* gamma, int M, int C, float eps, const void* residual, void* dx,
                             float* dgamma, float* dbeta) {
    RDM_ENTER(c);
    if (!x || !dy || !gamma || !dx || !dgamma || !dbeta) return c->fail(-1, "rdm_op_layernorm_bwd: null argument");
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, (size_t)2 * ((M + 15) / 16) * C * 4));
    RDM_CHECK_HIP(c, launch_layernorm_bwd((const bf16_t*)x, (const bf16_t*)dy, gamma, M, C, eps, (float*)c->bwd_tmp, nullptr, (bf16_t*)dx, dgamma, dbeta,
                                          c->stream, (const bf16_t*)residual));
    return 0;
}
int rdm_op_layernorm_bwd(rdm_ctx* c, const void* x, const void* dy, const float* gamma, int M, int C, float eps, void* dx, float* dgamma, float* dbeta) {
    return rdm_op_layernorm_bwd_add(c, x, dy, gamma, M, C, eps, nullptr, dx, dgamma, dbeta);
}
int rdm_op_linear_wgrad(rdm_ctx* c, const void* dy, const void* a, float* dw, long long M, int N, int K) {
    RDM_ENTER(c);
    if (!dy || !a || !dw || M < 1 || M > 0x7fffffffLL || N < 2 || K < 2 || N % 2 || K % 2) return c->fail(-1, "rdm_op_linear_wgrad: bad argument (N, K even)");
    if (wgrad_tn_supported(M, N, K, N, K)) {
        RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, wgrad_tn_scratch_bytes(M, N, K, 1) + 256));
        RDM_CHECK_HIP(c, launch_wgrad_tn((const bf16_t*)dy, N, (const bf16_t*)a, K, dw, M, N, K, 1, 1, 1, c->bwd_tmp, c->zero_page, c->stream));
        return 0;
    }
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, linear_wgrad_scratch_bytes(M, N, K)));
    RDM_CHECK_HIP(c, launch_linear_wgrad((const bf16_t*)dy, (const bf16_t*)a, dw, M, N, K, c->bwd_tmp, c->zero_page, c->stream));
    return 0;
}
int rdm_op_colsum(rdm_ctx* c, const void* x, float* out, long long M, int N) {
    RDM_ENTER(c);
    if (!x || !out) return c->fail(-1, "rdm_op_colsum: null argument");
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, colsum_scratch_bytes(M, N) + 256));
    RDM_CHECK_HIP(c, launch_colsum((const bf16_t*)x, out, M, N, c->stream, (float*)c->bwd_tmp));
    return 0;
}
int rdm_op_transpose(rdm_ctx* c, const void* x, void* y, int rows, int cols) {
    RDM_ENTER(c);
    if (!x || !y) return c->fail(-1, "rdm_op_transpose: null argument");
    RDM_CHECK_HIP(c, launch_transpose_bf16((const bf16_t*)x, (bf16_t*)y, rows, cols, c->stream));
    return 0;
}
int rdm_op_add(rdm_ctx* c, const void* a, const void* b, void* out, long long n) {
    RDM_ENTER(c);
    if (!a || !b || !out) return c->fail(-1, "rdm_op_add: null argument");
    RDM_CHECK_HIP(c, launch_add_bf16((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n, c->stream));
    return 0;
}
int rdm_op_ema(rdm_ctx* c, float* shadow, const float* p, long long n, float one_minus_decay) {
    RDM_ENTER(c);
    if (!shadow || !p || n < 1) return c->fail(-1, "rdm_op_ema: bad argument");
    RDM_CHECK_HIP(c, launch_ema(shadow, p, n, one_minus_decay, c->stream));
    return 0;
}
int rdm_op_silu(rdm_ctx* c, const float* x, const float* dy, void* out, long long n) {
    RDM_ENTER(c);
    if (!x || !out || n < 1) return c->fail(-1, "rdm_op_silu: bad argument");
    RDM_CHECK_HIP(c, launch_silu(x, dy, dy ? nullptr : (bf16_t*)out, dy ? (float*)out : nullptr, n, c->stream));
    return 0;
}
int rdm_op_q_sample(rdm_ctx* c, const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, float* out, void* out_nhwc, int B, int C,
                    int H, int W, int cpad) {
    RDM_ENTER(c);
    if (!x0 || !noise || !sqrt_ac || !sqrt_1mac || (!out && !out_nhwc) || B < 1 || C < 1 || H < 1 || W < 1) return c->fail(-1, "rdm_op_q_sample: bad argument");
    RDM_CHECK_HIP(c, launch_q_sample(x0, noise, sqrt_ac, sqrt_1mac, out, (bf16_t*)out_nhwc, B, C, H * W, cpad, c->stream));
    return 0;
}
int rdm_op_mse_loss(rdm_ctx* c, const void* eps_nhwc, const float* target, const float* coef, float* se, void* deps_nhwc, int B, int C, int H, int W, int ldc) {
    RDM_ENTER(c);
    if (!eps_nhwc || !target || !se || (deps_nhwc && !coef) || B < 1 || C < 1 || ldc < C) return c->fail(-1, "rdm_op_mse_loss: bad argument");
    RDM_CHECK_HIP(c, launch_mse_loss((const bf16_t*)eps_nhwc, target, coef, se, (bf16_t*)deps_nhwc, B, C, H * W, ldc, c->stream));
    return 0;
}
int rdm_op_where_rows(rdm_ctx* c, const unsigned char* mask, const float* a, const float* x, float* out, long long rows, long long n) {
    RDM_ENTER(c);
    if (!mask || !a || !x || !out || rows < 1 || n < 1) return c->fail(-1, "rdm_op_where_rows: bad argument");
    RDM_CHECK_HIP(c, launch_where_rows(mask, a, x, out, rows, n, c->stream));
    return 0;
}
int rdm_op_timestep_embedding(rdm_ctx* c, const int64_t* t, void* out_bf16, int B, int dim, int ld) {
    RDM_ENTER(c);
    if (!t || !out_bf16 || B < 1 || dim < 2 || dim % 2 || ld < dim) return c->fail(-1, "rdm_op_timestep_embedding: bad argument");
    RDM_CHECK_HIP(c, launch_timestep_embedding((const long long*)t, (bf16_t*)out_bf16, B, dim, ld, c->stream));
    return 0;
}
int rdm_op_colsum_samples(rdm_ctx* c, const void* x, void* out, int B, int HW, int N) {
    RDM_ENTER(c);
    if (!x || !out || B < 1 || HW < 1 || N < 1) return c->fail(-1, "rdm_op_colsum_samples: bad argument");
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, colsum_samples_scratch_bytes(B, HW, N)));
    RDM_CHECK_HIP(c, launch_colsum_samples((const bf16_t*)x, (bf16_t*)out, B, HW, N, c->stream, (float*)c->bwd_tmp));
    return 0;
}
int rdm_op_expand2(rdm_ctx* c, const void* x, void* out, int B, int H, int W, int C, int mode) {
    RDM_ENTER(c);
    if (!x || !out || B < 1 || H < 1 || W < 1 || C < 8 || C % 8 || (mode != 0 && mode != 1)) return c->fail(-1, "rdm_op_expand2: bad argument (C %% 8 == 0, mode 0 / 1)");
    RDM_CHECK_HIP(c, launch_expand2((const bf16_t*)x, (bf16_t*)out, B, H, W, C, mode, c->stream));
    return 0;
}
int rdm_op_sumpool2(rdm_ctx* c, const void* x, void* out, int B, int H, int W, int C) {
    RDM_ENTER(c);
    if (!x || !out || B < 1 || H < 1 || W < 1 || C < 8 || C % 8) return c->fail(-1, "rdm_op_sumpool2: bad argument (C must be a multiple of 8)");
    RDM_CHECK_HIP(c, launch_sumpool2((const bf16_t*)x, (bf16_t*)out, B, H, W, C, c->stream));
    return 0;
}
int rdm_op_adamw(rdm_ctx* c, float* p, const float* g, float* m, float* v, void* p_bf16, long long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step) {
    RDM_ENTER(c);
    if (!p || !g || !m || !v || n < 1 || step < 1) return c->fail(-1, "rdm_op_adamw: bad argument");
    RDM_CHECK_HIP(c, launch_adamw(p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, weight_decay, step, c->stream));
    return 0;
}
int rdm_op_adamw_multi(rdm_ctx* c, int n, float* const* p, const float* const* g, float* const* m, float* const* v, void* const* p_bf16, const long long* numel,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
    RDM_ENTER(c);
    if (n < 1 || !p || !g || !m || !v || !numel || step < 1) return c->fail(-1, "rdm_op_adamw_multi: bad argument");
    for (int i = 0; i < n; i++) if (!p[i] || !g[i] || !m[i] || !v[i] || numel[i] < 1) return c->fail(-1, "rdm_op_adamw_multi: null tensor in the list");
    RDM_CHECK_HIP(c, launch_multi_tensor(n, p, g, m, v, p_bf16, numel, 0, lr, beta1, beta2, eps, weight_decay, step, 0.f, c->stream));
    return 0;
}
int rdm_op_ema_multi(rdm_ctx* c, int n, float* const* shadow, const float* const* param, const long long* numel, float one_minus_decay) {
    RDM_ENTER(c);
    if (n < 1 || !shadow || !param || !numel) return c->fail(-1, "rdm_op_ema_multi: bad argument");
    for (int i = 0; i < n; i++) if (!shadow[i] || !param[i] || numel[i] < 1) return c->fail(-1, "rdm_op_ema_multi: null tensor in the list");
    RDM_CHECK_HIP(c, launch_multi_tensor(n, shadow, param, nullptr, nullptr, nullptr, numel, 1, 0.f, 0.f, 0.f, 0.f, 0.f, 1, one_minus_decay, c->stream));
    return 0;
}
int rdm_op_attention_bwd(rdm_ctx* c, const void* q, const void* k, const void* v, const void* o, const void* dout, int B, int n, int m, int heads,
                         void* dq, void* dk, void* dv) {
    RDM_ENTER(c);
    if (!q || !k || !v || !o || !dout || !dq || !dk || !dv || B < 1 || heads < 1 || n < 32 || m < 32 || n % 32 || m % 32)
        return c->fail(-1, "rdm_op_attention_bwd: bad argument (d_head = 32; n and m multiples of 32)");
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, attn_bwd_scratch_bytes(B, heads, n, m)));
    RDM_CHECK_HIP(c, launch_attention_bwd((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)dout, B, n, m, heads,
                                          (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, c->bwd_tmp, c->stream));
    return 0;
}
int rdm_op_small_attention_bwd(rdm_ctx* c, const void* q, int ldq, const void* k, const void* v, int ldkv, const void* dout, int ldo, int B, int nq, int nkv,
                               int heads, float scale, void* dq, void* dk, void* dv) {
    RDM_ENTER(c);
    if (!q || !k || !v || !dout || !dq || !dk || !dv || B < 1 || heads < 1 || nq < 1 || nkv < 1 || nkv > 32)
        return c->fail(-1, "rdm_op_small_attention_bwd: bad argument (d_head = 32, 1..32 keys)");
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, small_attention_bwd_scratch_bytes(B, heads, nq, nkv)));
    RDM_CHECK_HIP(c, launch_small_attention_bwd((const bf16_t*)q, ldq, (const bf16_t*)k, (const bf16_t*)v, ldkv, (const bf16_t*)dout, ldo, B, nq, nkv, heads, scale,
                                                (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, c->bwd_tmp, c->stream));
    return 0;
}
int rdm_op_bmm(rdm_ctx* c, const void* a, const void* w, void* out_bf16, float* out_f32, int batch, int M, int N, int K, float alpha) {
    RDM_ENTER(c);
    if (!a || !w || (!out_bf16 && !out_f32) || batch < 1 || M < 1 || N < 2 || K < 64 || K % 64 || N % 2)
        return c->fail(-1, "rdm_op_bmm: bad argument (K must be a multiple of 64, N even)");
    IgemmParams p{}; p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.ldo = N; p.zero_page = c->zero_page;
    p.Hin = p.Win = p.Hout = p.Wout = 1; p.stride = 1; p.rows_per_sample = 1;
    p.A0 = (const bf16_t*)a; p.C0 = K; p.W = (const bf16_t*)w; p.out_bf16 = (bf16_t*)out_bf16; p.out_f32 = out_f32;
    p.sA = (long long)M * K; p.sW = (long long)N * K; p.sO = (long long)M * N;
    RDM_CHECK_HIP(c, launch_igemm(p, false, batch, c->stream));
    return 0;
}
int rdm_op_heads(rdm_ctx* c, const void* x, void* out, int B, int n, int H, int D, int ldx, int mode) {
    RDM_ENTER(c);
    if (!x || !out || B < 1 || n < 1 || H < 1) return c->fail(-1, "rdm_op_heads: bad argument");
    RDM_CHECK_HIP(c, launch_heads((const bf16_t*)x, (bf16_t*)out, B, n, H, D, ldx, mode, c->stream));
    return 0;
}
int rdm_op_transpose_batched(rdm_ctx* c, const void* x, void* y, int batch, int rows, int cols) {
    RDM_ENTER(c);
    if (!x || !y || batch < 1 || batch > 65535) return c->fail(-1, "rdm_op_transpose_batched: bad argument");
    RDM_CHECK_HIP(c, launch_transpose_bf16((const bf16_t*)x, (bf16_t*)y, rows, cols, c->stream, batch));
    return 0;
}
int rdm_op_softmax(rdm_ctx* c, const float* s, void* p_bf16, long long rows, int n, int n_valid) {
    RDM_ENTER(c);
    if (!s || !p_bf16 || n % 4) return c->fail(-1, "rdm_op_softmax: bad argument (n must be a multiple of 4)");
    RDM_CHECK_HIP(c, launch_softmax_rows(s, (bf16_t*)p_bf16, rows, n, c->stream, n_valid));
    return 0;
}
int rdm_op_softmax_bwd(rdm_ctx* c, const void* p_bf16, const float* dp, void* ds_bf16, long long rows, int n) {
    RDM_ENTER(c);
    if (!p_bf16 || !dp || !ds_bf16 || n % 4) return c->fail(-1, "rdm_op_softmax_bwd: bad argument (n must be a multiple of 4)");
    RDM_CHECK_HIP(c, launch_softmax_bwd((const bf16_t*)p_bf16, dp, (bf16_t*)ds_bf16, rows, n, c->stream));
    return 0;
}
int rdm_op_geglu(rdm_ctx* c, const void* pre, const void* dh, void* out, long long M, int F) {
    RDM_ENTER(c);
    if (!pre || !out || M < 1 || F < 8 || F % 8) return c->fail(-1, "rdm_op_geglu: bad argument (F must be a positive multiple of 8)");
    RDM_CHECK_HIP(c, launch_geglu((const bf16_t*)pre, (const bf16_t*)dh, (bf16_t*)out, M, F, c->stream));
    return 0;
}
int rdm_op_groupnorm(rdm_ctx* c, const void* x0, const void* x1, int C0, int C1, int B, int HW, const float* gamma,
                     const float* beta, float eps, int silu, void* out) {
    RDM_ENTER(c);
    if (!c) return -1;
    RDM_TRY(ensure_gn_partial(c, B));
    GnParams p{}; p.x0 = (const bf16_t*)x0; p.x1 = (const bf16_t*)x1; p.C0 = C0; p.C1 = C1; p.HW = HW; p.B = B; p.groups = 32;
    int nchunk = HW / 64; if (nchunk < 1) nchunk = 1; if (nchunk > 32) nchunk = 32;
    p.nchunk = nchunk; p.partial = c->gn_partial; p.gamma = gamma; p.beta = beta; p.eps = eps; p.silu = silu; p.out = (bf16_t*)out;
    RDM_CHECK_HIP(c, launch_groupnorm(p, c->stream));
    return 0;
}
int rdm_op_layernorm(rdm_ctx* c, const void* x, int in_is_f32, const float* gamma, const float* beta, int M, int C, float eps,
                     void* out) {
    RDM_ENTER(c);
    if (!c) return -1;
    RDM_CHECK_HIP(c, launch_layernorm(x, in_is_f32, gamma, beta, out, 0, M, C, eps, c->stream));
    return 0;
}
int rdm_op_self_attention(rdm_ctx* c, const void* qk, const void* vt, int B, int n, int heads, void* out) {
    RDM_ENTER(c);
    if (!c) return -1;
    const int C = heads * 32;
    FlashParams f{}; f.q = (const bf16_t*)qk; f.ldq = 2 * C; f.k = (const bf16_t*)qk + C; f.ldk = 2 * C; f.vt = (const bf16_t*)vt;
    f.out = (bf16_t*)out; f.ldo = C; f.n = n; f.C = C; f.scale_log2e = (1.0f / sqrtf(32.f)) * 1.4426950408889634f;
    RDM_CHECK_HIP(c, launch_flash_d32(f, heads, B, c->stream));
    return 0;
}
int rdm_op_self_attention_qkv(rdm_ctx* c, const void* qkv, int B, int n, int heads, void* out) {
    RDM_ENTER(c);
    if (!c) return -1;
    const int C = heads * 32;
    if (n % 64 != 0) return c->fail(-3, "rdm_op_self_attention_qkv: n = %d must be a multiple of 64 (token-major V is read by the LDS-shared kernel only)", n);
    FlashParams f{}; f.q = (const bf16_t*)qkv; f.ldq = 3 * C; f.k = f.q + C; f.ldk = 3 * C; f.v = f.q + 2 * C; f.ldv = 3 * C;
    f.out = (bf16_t*)out; f.ldo = C; f.n = n; f.C = C; f.scale_log2e = (1.0f / sqrtf(32.f)) * 1.4426950408889634f;
    RDM_CHECK_HIP(c, launch_flash_d32(f, heads, B, c->stream));
    return 0;
}
int rdm_op_xattn_fused(rdm_ctx* c, const void* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* G, const void* U,
                       const float* bias, const void* res, int B, int n, int C, int NP, int ncols, int group, void* out) {
    RDM_ENTER(c);
    if (!c) return -1;
    XattnParams q{}; q.x = (const bf16_t*)x; q.G = (const bf16_t*)G; q.U = (const bf16_t*)U; q.bias = bias; q.res = (const bf16_t*)res;
    q.out = (bf16_t*)out; q.rows = B * n; q.n = n; q.C = C; q.NP = NP; q.ncols = ncols; q.group = group;
    q.ln_g = ln_gamma; q.ln_b = ln_beta; q.ln_eps = ln_eps;
    if ((ln_gamma != nullptr) != (ln_beta != nullptr) || (ln_gamma && res)) return c->fail(-3, "rdm_op_xattn_fused: LayerNorm needs gamma and beta, and then the residual is x itself (res must be null)");
    if (!xattn_fused_supported(q)) return c->fail(-3, "rdm_op_xattn_fused: unsupported shape (n %% 32, C %% 64, NP %% 32, ncols <= min(NP, 128), group 1 / 2 / 4): n %d C %d NP %d ncols %d group %d", n, C, NP, ncols, group);
    const size_t img = (size_t)B * NP * C * 2;          // the kernel reads fragment-ordered images of G and U
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, 2 * img));
    bf16_t* Gp = (bf16_t*)c->bwd_tmp; bf16_t* Up = Gp + (size_t)B * NP * C;
    RDM_CHECK_HIP(c, launch_xattn_pack(q.G, q.U, Gp, Up, B, NP, C, c->stream));
    q.G = Gp; q.U = Up;
    RDM_CHECK_HIP(c, launch_xattn_fused(q, c->stream));
    return 0;
}
int rdm_op_xattn_fused_ln3(rdm_ctx* c, void* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* G, const void* U,
                           const float* bias, int B, int n, int C, int NP, int ncols, int group, const float* ln3_gamma, const float* ln3_beta, void* ln3_out) {
    RDM_ENTER(c);
    if (!c) return -1;
    if (!x || !ln_gamma || !ln_beta || !ln3_gamma || !ln3_beta || !ln3_out) return c->fail(-1, "rdm_op_xattn_fused_ln3: null argument");
    XattnParams q{}; q.x = (const bf16_t*)x; q.G = (const bf16_t*)G; q.U = (const bf16_t*)U; q.bias = bias; q.res = nullptr;
    q.out = (bf16_t*)x; q.rows = B * n; q.n = n; q.C = C; q.NP = NP; q.ncols = ncols; q.group = group;
    q.ln_g = ln_gamma; q.ln_b = ln_beta; q.ln_eps = ln_eps; q.ln3_g = ln3_gamma; q.ln3_b = ln3_beta; q.ln3_out = (bf16_t*)ln3_out;
    if (!xattn_fused_supported(q)) return c->fail(-3, "rdm_op_xattn_fused_ln3: unsupported shape: n %d C %d NP %d ncols %d group %d", n, C, NP, ncols, group);
    const size_t img = (size_t)B * NP * C * 2;
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, 2 * img));
    bf16_t* Gp = (bf16_t*)c->bwd_tmp; bf16_t* Up = Gp + (size_t)B * NP * C;
    RDM_CHECK_HIP(c, launch_xattn_pack(q.G, q.U, Gp, Up, B, NP, C, c->stream));
    q.G = Gp; q.U = Up;
    RDM_CHECK_HIP(c, launch_xattn_fused(q, c->stream));
    return 0;
}
int rdm_op_head_conv(rdm_ctx* c, const void* x, const float* gn_gamma, const float* gn_beta, float gn_eps, const float* w, const float* bias,
                     int B, int H, int W, int C, int Cout, float* out) {
    RDM_ENTER(c);
    if (!c) return -1;
    HeadParams hp{}; hp.x = (const bf16_t*)x; hp.B = B; hp.H = H; hp.W = W; hp.C = C; hp.groups = 32; hp.gamma = gn_gamma; hp.beta = gn_beta; hp.eps = gn_eps;
    hp.w = w; hp.bias = bias; hp.out = out; hp.Cout = Cout;
    if ((gn_gamma != nullptr) != (gn_beta != nullptr)) return c->fail(-3, "rdm_op_head_conv: GroupNorm needs gamma and beta");
    int nchunk = H * W / 64; if (nchunk < 1) nchunk = 1; if (nchunk > 32) nchunk = 32;
    if (gn_gamma) { RDM_TRY(ensure_gn_partial(c, B)); hp.partial = c->gn_partial; hp.nchunk = nchunk; }
    if (!head_conv_supported(hp)) return c->fail(-3, "rdm_op_head_conv: unsupported shape (C %% 32, C <= 240, W %% 32, H %% 2, Cout <= 8, C %% 32 groups): C %d H %d W %d Cout %d", C, H, W, Cout);
    RDM_TRY(ensure_bytes(c, &c->bwd_tmp, &c->bwd_tmp_bytes, head_conv_wp_bytes(C)));
    hp.wp = (bf16_t*)c->bwd_tmp;
    if (gn_gamma) {
        GnParams p{}; p.x0 = hp.x; p.C0 = C; p.HW = H * W; p.B = B; p.groups = 32; p.L0 = C; p.nchunk = nchunk; p.partial = c->gn_partial;
        RDM_CHECK_HIP(c, launch_gn_stats(p, c->stream));
    }
    RDM_CHECK_HIP(c, launch_head_conv(hp, c->stream));
    return 0;
}
int rdm_op_small_attention(rdm_ctx* c, const void* q, int ldq, const void* k, const void* v, int ldkv, int B, int nq, int nkv,
                           int heads, int D, int causal, float scale, void* out, int ldo) {
    RDM_ENTER(c);
    if (!c) return -1;
    SmallAttnParams p{}; p.q = (const bf16_t*)q; p.ldq = ldq; p.k = (const bf16_t*)k; p.ldk = ldkv; p.v = (const bf16_t*)v; p.ldv = ldkv;
    p.out = (bf16_t*)out; p.ldo = ldo; p.nq = nq; p.nkv = nkv; p.causal = causal; p.scale = scale;
    RDM_CHECK_HIP(c, launch_small_attention(p, D, heads, B, c->stream));
    return 0;
}

}  // extern "C"

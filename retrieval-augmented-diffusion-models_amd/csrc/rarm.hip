// Kernels of the RARM (retrieval-augmented autoregressive) sampling path on gfx950:
//   LatentImageRETRO.sample              rdm/models/autoregression/transformer.py:224-294
//   RetrievalPatchTransformer.forward    rdm/modules/attention.py:199-272 (continuous=False: token embedding + positional
//                                        encoding, 18 causal BasicTransformerBlocks with cross-attention to the k neighbours)
//   top_k_logits / softmax / multinomial transformer.py:256-270 (taming Net2NetTransformer.top_k_logits)
// The reference re-runs the transformer over the WHOLE prefix for each of the 256 new tokens (:241-248, 32 896 token-forwards
// per image); here every step processes one token per sequence against a K/V cache.  A decode step is M = B' (<= 128) rows, so
// the linear layers are weight-bandwidth-bound launches of the shared GEMM (igemm.hip); what is new is below.
//
// The step index lives in DEVICE memory (`pos`): every kernel of a step reads it, the sampler increments it — so the launches
// of all steps are identical and nothing on the host depends on the step.
#include "kernels.h"

// ---------------------------------------------------------------- token embedding + positional encoding
// x[b, :] = proj_in.weight[token[b]] + positional_encoding[:, pos]   (attention.py:252-258), fp32 residual stream
__global__ __launch_bounds__(256) void rarm_embed_kernel(const long long* tokens, const float* emb, const float* pos_t /*[L][C]*/,
                                                         const int* pos, float* x, int B, int C, int vocab) {
    const int b = blockIdx.x, t = *pos;
    long long tok = tokens[b];
    if (tok < 0 || tok >= vocab) tok = 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[(long long)b * C + c] = emb[tok * C + c] + pos_t[(long long)t * C + c];
}
hipError_t launch_rarm_embed(const long long* tokens, const float* emb, const float* pos_t, const int* pos, float* x, int B, int C,
                             int vocab, hipStream_t st) {
    rarm_embed_kernel<<<B, 256, 0, st>>>(tokens, emb, pos_t, pos, x, B, C, vocab);
    return hipGetLastError();
}

// ---------------------------------------------------------------- decode-step attention against a K/V cache (d_head = 64)
// One BLOCK of four waves per (head, sequence) (round 4; one wave before: its 2 x 8 chunk loops were 16 serialised round trips to
// the cache, 12.8 us at 256 cached rows -- four waves take every fourth 32-row chunk, and a wave's first K and V chunks are requested
// together before anything is computed).  Self-attention (attn1, causal): the new token's k / v rows are appended to the cache at
// position *pos and the query attends to rows 0..*pos.  Cross-attention (attn2): k_new == null, the cache holds the projected
// neighbours and all nkv rows are attended.  Phase 1: scores of the wave's chunks into LDS; block-wide maximum and sum; phase 2: the
// wave's chunks of sum_j p_j V[j][:], the four partial rows meet in LDS and are added in wave order.  CrossAttention.forward,
// attention.py:42-74.
// NW: waves per block = 32-row chunks in flight at once.  Eight waves (all 256 cached rows of a (head, sequence) requested in ONE round
// trip; RDM_RARM_ATTN_NW8_FROM=<sequences>) measured SLOWER, round 5: 419 / 578 / 683 -> 406 / 552 / 644 img/s at 256 / 512 / 1024 sequences
// (profiles/r05e_rarm_attn_8wave_sweep.log) -- averaged over the 256 positions the launch is a fixed ~6.5 us + the cache bytes at ~5.5 TB/s
// already; the bigger blocks raise the fixed part (idle waves at short prefixes, wider barriers).  Four waves stay the default.
template <int NW>
__global__ __launch_bounds__(64 * NW) void rarm_decode_attention_kernel(RarmAttnParams p) {
    // Eight lanes per cache row (16 bytes each), eight rows per load instruction: an instruction touches 8 cache lines.  (One lane
    // per key row -- 64 rows, 64 lines per instruction -- in the score phase and one 128-byte row per instruction in the value
    // phase made this launch 11.5 us of mostly address traffic.)
    constexpr int D = 64;
    __shared__ float sc[1024];
    __shared__ float red[2 * NW];
    __shared__ float part[NW][D];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jr = lane >> 3, c8 = (lane & 7) * 8;
    const int t = p.pos ? *p.pos : 0;
    const int n = p.k_new ? t + 1 : p.nkv;
    const int nc = p.k_new ? t : n;                     // rows read from the cache by the value phase
    const long long hs = p.head_stride > 0 ? p.head_stride : D;
    bf16_t* Kc = p.Kc + (long long)b * p.batch_stride + h * hs;
    bf16_t* Vc = p.Vc + (long long)b * p.batch_stride + h * hs;
    const bf16_t* knew = p.k_new ? p.k_new + (long long)b * p.ldq + h * D : nullptr;
    const bf16_t* vnew = p.k_new ? p.v_new + (long long)b * p.ldq + h * D : nullptr;
    float qv[8];
    {
        const bf16x8 qq = *(const bf16x8*)(p.q + (long long)b * p.ldq + h * D + c8);
#pragma unroll
        for (int e = 0; e < 8; e++) qv[e] = bf2f((bf16_t)qq[e]) * p.scale;
    }
    if (p.k_new && w == 0 && jr == 0) {       // append the new row (read back from the projection output below, never through the cache)
        *(bf16x8*)(Kc + (long long)t * p.row_stride + c8) = *(const bf16x8*)(knew + c8);
        *(bf16x8*)(Vc + (long long)t * p.row_stride + c8) = *(const bf16x8*)(vnew + c8);
    }
    // ---- the wave's first chunk of keys AND values in one batch of requests (the values do not depend on the scores)
    bf16x8 kk[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int j = w * 32 + u * 8 + jr;
        const bf16_t* kr = (p.k_new && j == t) ? knew : Kc + (long long)(j < n ? j : 0) * p.row_stride;
        kk[u] = *(const bf16x8*)(kr + c8);
        vv[u] = *(const bf16x8*)(Vc + (long long)(j < nc ? j : 0) * p.row_stride + c8);
    }
    // ---- scores: rows j0 + 8 u + jr of chunks w, w + 4, ...
    float m = -INFINITY;
    for (int j0 = w * 32; j0 < n; j0 += 32 * NW) {
        if (j0 != w * 32) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u * 8 + jr;
                const bf16_t* kr = (p.k_new && j == t) ? knew : Kc + (long long)(j < n ? j : 0) * p.row_stride;
                kk[u] = *(const bf16x8*)(kr + c8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * 8 + jr;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) s += qv[e] * bf2f((bf16_t)kk[u][e]);
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            if (j < n) { if ((lane & 7) == 0) sc[j] = s; m = fmaxf(m, s); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int u = 1; u < NW; u++) m = fmaxf(m, red[u]);
    float l = 0.f;
    for (int j = tid; j < n; j += 64 * NW) { const float e = __expf(sc[j] - m); sc[j] = e; l += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    if (lane == 0) red[NW + w] = l;
    __syncthreads();
    l = (red[NW] + red[NW + 1]) + (red[NW + 2] + red[NW + 3]);
    if constexpr (NW == 8) l += (red[NW + 4] + red[NW + 5]) + (red[NW + 6] + red[NW + 7]);
    // ---- values: the wave's chunks of the cached rows
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j0 = w * 32; j0 < nc; j0 += 32 * NW) {
        if (j0 != w * 32) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u * 8 + jr;
                vv[u] = *(const bf16x8*)(Vc + (long long)(j < nc ? j : 0) * p.row_stride + c8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * 8 + jr;
            const float pj = j < nc ? sc[j] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += pj * bf2f((bf16_t)vv[u][e]);
        }
    }
    if (p.k_new && w == 0 && jr == 0) {                   // ... and the new token's row straight from the projection output
        const bf16x8 vn = *(const bf16x8*)(vnew + c8);
        const float pt = sc[t];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += pt * bf2f((bf16_t)vn[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) { acc[e] += __shfl_xor(acc[e], 8); acc[e] += __shfl_xor(acc[e], 16); acc[e] += __shfl_xor(acc[e], 32); }
    if (jr == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) part[w][c8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < D / 2) {                                    // fixed order over the waves: deterministic
        const int c = tid * 2;
        const float inv = 1.f / l;
        float o0 = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        float o1 = (part[0][c + 1] + part[1][c + 1]) + (part[2][c + 1] + part[3][c + 1]);
        if constexpr (NW == 8) {
            o0 += (part[4][c] + part[5][c]) + (part[6][c] + part[7][c]);
            o1 += (part[4][c + 1] + part[5][c + 1]) + (part[6][c + 1] + part[7][c + 1]);
        }
        o0 *= inv; o1 *= inv;
        *(uint32_t*)(p.out + (long long)b * p.ldo + h * D + c) = pack2bf(o0, o1);
    }
}
// Cross-attention over <= 8 projected neighbours (the GEMM-form decode step of big batches, round 5): ONE block per sequence, a wave per
// head in turn (heads w, w + 4, ..): a wave instruction reads the head's eight 128-byte key rows (eight lanes per row), the eight partial dot
// products of a row meet by three lane exchanges, the softmax over the eight keys by three more, the values likewise -- the launch above
// spends a 256-thread block with two LDS exchanges on each of heads x sequences eight-key problems (17 us at 512 sequences).
__global__ __launch_bounds__(256) void rarm_fewkey_attention_kernel(RarmAttnParams p, int heads) {
    constexpr int D = 64;
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane >> 3, c8 = (lane & 7) * 8;
    const bool live = j < p.nkv;
    for (int h = w; h < heads; h += 4) {
        const bf16_t* Kr = p.Kc + (long long)b * p.batch_stride + (long long)(live ? j : 0) * p.row_stride + h * D + c8;
        const bf16_t* Vr = p.Vc + (long long)b * p.batch_stride + (long long)(live ? j : 0) * p.row_stride + h * D + c8;
        const bf16x8 qq = *(const bf16x8*)(p.q + (long long)b * p.ldq + h * D + c8);
        const bf16x8 kk = *(const bf16x8*)Kr, vv = *(const bf16x8*)Vr;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) s += bf2f((bf16_t)qq[e]) * p.scale * bf2f((bf16_t)kk[e]);
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        s = live ? s : -INFINITY;
        float m = s;
        m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
        const float e_ = live ? __expf(s - m) : 0.f;
        float l = e_;
        l += __shfl_xor(l, 8); l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
        const float pj = e_ / l;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[e] = pj * bf2f((bf16_t)vv[e]);
            o[e] += __shfl_xor(o[e], 8); o[e] += __shfl_xor(o[e], 16); o[e] += __shfl_xor(o[e], 32);
        }
        if (j == 0)
            *(uint4*)(p.out + (long long)b * p.ldo + h * D + c8) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
    }
}
hipError_t launch_rarm_decode_attention(const RarmAttnParams& p, int heads, int batch, hipStream_t st) {
    if (p.nkv > 1024) return hipErrorInvalidValue;
    static const int no_fewkey = getenv("RDM_NO_RARM_FEWKEY") ? atoi(getenv("RDM_NO_RARM_FEWKEY")) : 0;
    if (!no_fewkey && !p.k_new && !p.pos && p.nkv >= 1 && p.nkv <= 8 && batch >= 128) {       // cross-attention over few keys at big batches
        rarm_fewkey_attention_kernel<<<batch, 256, 0, st>>>(p, heads);
        return hipGetLastError();
    }
    static const int nw8_from = getenv("RDM_RARM_ATTN_NW8_FROM") ? atoi(getenv("RDM_RARM_ATTN_NW8_FROM")) : 0;
    if (p.k_new && nw8_from > 0 && batch >= nw8_from) rarm_decode_attention_kernel<8><<<dim3(heads, batch), 512, 0, st>>>(p);
    else rarm_decode_attention_kernel<4><<<dim3(heads, batch), 256, 0, st>>>(p);
    return hipGetLastError();
}

// ---------------------------------------------------------------- decode-step cross-attention, one launch
// norm2 + to_q + attention over the k neighbours + to_out + residual (RetrievalPatchTransformer block, rdm/modules/attention.py:238)
// for ONE new token per sequence.  With the neighbours fixed for the whole sampling call, softmax(q K^T / sqrt d) V W_o^T is
// re-associated per sequence (model.hip: rarm_prepare): scores = LN(x) G_b^T, out = P UT_b -- two matrix-vector products against 2 x
// heads*k x C bf16 per sequence instead of two C x C projections and an attention launch (3 launches of ~9 us each, all latency).
// One block per sequence: LayerNorm by block reduction (two-pass, rounded to bf16 as the GEMM operand was), a wave per group of
// score rows (coalesced 16-byte pieces, wave reduction), per-head softmax, then 4 output channels per thread over the heads*k rows.
// LayerNorm of ONE row spread one channel per thread over NW waves (two-pass, fp32), rounded to bf16: y[c] = (x - mean) rstd g[c] + b[c].
// red: >= 2 NW floats of LDS.  Every thread of the block must call it.
template <int NW>
__device__ __forceinline__ void rarm_emit_ln(float xv, bool live, int c, int C, const float* g, const float* bta, float eps, bf16_t* y, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s = live ? xv : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __syncthreads();                                   // red may still be read by the caller's previous phase
    if (lane == 0) red[w] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) tot += red[i];
    const float mean = tot / C;
    const float d = live ? xv - mean : 0.f;
    float q = d * d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) red[NW + w] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) tot += red[NW + i];
    const float rstd = rsqrtf(tot / C + eps);
    if (live) y[c] = f2bf(d * rstd * g[c] + bta[c]);
}
__global__ __launch_bounds__(1024) void rarm_xattn_decode_kernel(RarmXattnParams p) {
    // 16 waves per sequence: the two matrix-vector products are one batch of loads each (every row of G / UT a thread needs is
    // requested before the first one is used) -- with 4 waves and batches of 4 - 8 rows the kernel was 15 dependent HBM round trips
    __shared__ float xn[1024];            // LayerNorm(x), C <= 1024
    __shared__ float sc[128];             // scores -> probabilities
    __shared__ float red[32];
    __shared__ __attribute__((aligned(16))) float part[4 * 1024];        // output partials [row group][C]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, C = p.C;
    float* xr = p.x + (long long)b * C;
    if (b >= p.Bc) {                      // zero neighbours: K = V = 0 -> the attention output is 0, to_out leaves its bias
        float xv = 0.f;
        if (tid < C) { xv = xr[tid] + p.bias[tid]; xr[tid] = xv; }
        if (p.ln3_out) rarm_emit_ln<16>(xv, tid < C, tid, C, p.ln3_g, p.ln3_b, p.ln_eps, p.ln3_out + (long long)b * C, red);
        return;
    }
    // ---- LayerNorm (one channel per thread)
    const float v = tid < C ? xr[tid] : 0.f;
    float s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[w] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) tot += red[i];
    const float mean = tot / C;
    const float d = tid < C ? v - mean : 0.f;
    float q = d * d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) red[16 + w] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) tot += red[16 + i];
    const float rstd = rsqrtf(tot / C + p.ln_eps);
    if (tid < C) xn[tid] = bf2f(f2bf(d * rstd * p.ln_g[tid] + p.ln_b[tid]));
    __syncthreads();
    // ---- scores: wave w takes rows w, w + 16, ... (<= 8 rows); a row = C/8 16-byte pieces over the lanes (two rounds)
    const int nrow = p.heads * p.k, npc = C >> 3;
    const bf16_t* Gb = p.G + (long long)b * p.NP * C;
    {
        bf16x8 g[8][2];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = w + 16 * u;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int pc = lane + 64 * h;
                g[u][h] = (j < nrow && pc < npc) ? *(const bf16x8*)(Gb + (long long)j * C + pc * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = w + 16 * u;
            float a = 0.f;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int pc = lane + 64 * h;
                if (pc < npc) {
#pragma unroll
                    for (int e = 0; e < 8; e++) a += bf2f((bf16_t)g[u][h][e]) * xn[pc * 8 + e];
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0 && j < nrow) sc[j] = a;
        }
    }
    __syncthreads();
    // ---- softmax over each head's k columns
    float pr = 0.f;
    if (tid < nrow) {
        const int h0 = (tid / p.k) * p.k;
        float m = -INFINITY;
        for (int i = 0; i < p.k; i++) m = fmaxf(m, sc[h0 + i]);
        float sum = 0.f;
        for (int i = 0; i < p.k; i++) sum += __expf(sc[h0 + i] - m);
        pr = __expf(sc[tid] - m) / sum;
    }
    __syncthreads();
    if (tid < nrow) sc[tid] = pr;
    __syncthreads();
    // ---- output: thread = (4 channels, row group); NG = 4 row groups, each thread <= 32 rows, all requested at once
    const bf16_t* Ub = p.UT + (long long)b * p.NP * C;
    const int nc4 = C >> 2, c4 = tid % nc4, jg = tid / nc4;      // nc4 <= 256 -> at least 4 groups among the 1024 threads
    if (jg < 4) {
        uint2 u[32];
#pragma unroll
        for (int i = 0; i < 32; i++) { const int j = jg + 4 * i; u[i] = j < nrow ? *(const uint2*)(Ub + (long long)j * C + c4 * 4) : make_uint2(0u, 0u); }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int j = jg + 4 * i;
            const float pj = j < nrow ? sc[j] : 0.f;
            a0 += pj * __uint_as_float(u[i].x << 16); a1 += pj * __uint_as_float(u[i].x & 0xffff0000u);
            a2 += pj * __uint_as_float(u[i].y << 16); a3 += pj * __uint_as_float(u[i].y & 0xffff0000u);
        }
        *(float4*)(part + jg * 1024 + c4 * 4) = make_float4(a0, a1, a2, a3);
    }
    __syncthreads();
    if (tid < nc4) {
        float4 xo = *(float4*)(xr + tid * 4);
        const float4 bb = *(const float4*)(p.bias + tid * 4);
        float4 acc = bb;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { const float4 t = *(const float4*)(part + g4 * 1024 + tid * 4); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
        xo.x += acc.x; xo.y += acc.y; xo.z += acc.z; xo.w += acc.w;
        *(float4*)(xr + tid * 4) = xo;
        if (p.ln3_out) *(float4*)(xn + tid * 4) = xo;          // the finished row, for norm3 below (xn is free: the scores are done)
    }
    if (p.ln3_out) {       // the block holds the finished row: norm3 of the feed-forward that follows leaves with it (bf16 GEMM operand)
        __syncthreads();
        rarm_emit_ln<16>(tid < C ? xn[tid] : 0.f, tid < C, tid, C, p.ln3_g, p.ln3_b, p.ln_eps, p.ln3_out + (long long)b * C, red);
    }
}
// Four blocks per sequence (round 4).  The one-block form asks ONE CU for a sequence's whole operand pair (heads k rows of G and of
// UT: 295 KB at 12 heads x 8 neighbours x 768 channels) on 64 of 256 CUs: 16.4 us per layer.  The softmax is per head, so a block
// that owns a QUARTER of the heads needs only their rows of G and UT (74 KB): scores, softmax and the partial output row of its
// heads; the four partial rows meet in global memory -- written through to the coherence point (agent-scope stores), one
// agent-scope arrival counter per sequence -- and the block that arrives last adds them IN BLOCK ORDER to bias and residual
// (deterministic).  heads % 4 == 0.
// LayerNorm of one row held as channels tid + 256 i (i < 4) by a 256-thread block, rounded to bf16 (every thread calls it)
__device__ __forceinline__ void rarm_emit_ln4(const float (&xv)[4], int tid, int C, const float* g, const float* bta, float eps, bf16_t* y, float* red) {
    const int lane = tid & 63, w = tid >> 6;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) s += (tid + 256 * i < C) ? xv[i] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __syncthreads();
    if (lane == 0) red[w] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { const float d = (tid + 256 * i < C) ? xv[i] - mean : 0.f; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) red[4 + w] = q;
    __syncthreads();
    const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / C + eps);
#pragma unroll
    for (int i = 0; i < 4; i++) { const int c = tid + 256 * i; if (c < C) y[c] = f2bf((xv[i] - mean) * rstd * g[c] + bta[c]); }
}
// HAND-OVER OF THE PARTIAL ROWS (round 6: self-validating granules instead of an ordering assumption).  Round 4's form wrote the partial
// rows with agent-scope write-through stores, drained them (vmcnt(0) + workgroup barrier), arrived on a relaxed agent-scope counter, and the
// last arriver read the rows back with agent-scope loads: correct only if "store completed" means "visible to every other XCD" -- gfx950
// behaviour outside the C++ memory model, and round 5's stress test saw ONE bitwise mismatch in ~6 800 repeated decodes (~2.9 M launches)
// whose only possible source was this hand-over (everything else in the step is block-local): a last arriver reading one partial value of
// the PREVIOUS layer, which sits in the same buffer.  Now every value travels as ONE naturally aligned 8-byte granule {fp32 bits, tag},
// tag = this launch's epoch (host-incremented per launch, never 0; the buffer is zeroed when it is allocated), written by ONE agent-scope
// 8-byte store.  The last arriver checks the tag of every granule it reads and re-reads (agent-scope load, bounded spin) until the tag
// is this launch's: a value of an earlier launch can no longer be consumed, whatever the visibility delay -- no release / acquire edge and
// no assumption about write-through completion is needed, on any target.  Re-reads are counted (g_rarm_xsplit_stale, rdm_debug_counter(0)):
// tools/rarm_stress.py reports them, a non-zero count is the round-5 mismatch caught in the act.  The arrival counter is monotonic
// (last = every fourth arrival), nothing is re-armed.
__device__ unsigned long long g_rarm_xsplit_stale = 0ull;
unsigned long long rarm_xsplit_stale_count() {
    unsigned long long v = 0ull;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_rarm_xsplit_stale), sizeof(v));
    return v;
}
__global__ __launch_bounds__(256) void rarm_xattn_decode_split_kernel(RarmXattnParams p) {
    __shared__ float xn[1024];
    __shared__ float sc[32];
    __shared__ float red[8];
    __shared__ int s_last;
    const int b = blockIdx.x >> 2, q = blockIdx.x & 3, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, C = p.C;
    float* xr = p.x + (long long)b * C;
    if (b >= p.Bc) {                      // zero neighbours: the attention output is 0, to_out leaves its bias (block 0 of the sequence)
        if (q != 0) return;
        float xv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const int c = tid + 256 * i; xv[i] = 0.f; if (c < C) { xv[i] = xr[c] + p.bias[c]; xr[c] = xv[i]; } }
        if (p.ln3_out) rarm_emit_ln4(xv, tid, C, p.ln3_g, p.ln3_b, p.ln_eps, p.ln3_out + (long long)b * C, red);
        return;
    }
    // ---- every operand row this block will read, requested NOW: neither G nor UT depends on the LayerNorm or the scores, and the launch
    // is a chain of dependent round trips -- wave w takes score rows w, w + 4, ... (<= 8 each; a row = C/8 16-byte pieces: two rounds of
    // lanes), thread t the 4 output channels 4 t of all nr rows of UT
    const int nr = (p.heads >> 2) * p.k, r0 = q * nr, npc = C >> 3;
    const bf16_t* Gb = p.G + ((long long)b * p.NP + r0) * C;
    const bf16_t* Ub = p.UT + ((long long)b * p.NP + r0) * C;
    const int nc4 = C >> 2;
    bf16x8 g[8][2];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int j = w + 4 * u;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int pc = lane + 64 * h;
            g[u][h] = (j < nr && pc < npc) ? *(const bf16x8*)(Gb + (long long)j * C + pc * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    uint2 u[32];
    if (tid < nc4) {
#pragma unroll
        for (int i = 0; i < 32; i++) u[i] = i < nr ? *(const uint2*)(Ub + (long long)i * C + tid * 4) : make_uint2(0u, 0u);
    }
    // ---- LayerNorm (two-pass; up to 4 channels per thread), rounded to bf16 as the GEMM operand was
    float v[4]; float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { const int c = tid + 256 * i; v[i] = c < C ? xr[c] : 0.f; s += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[w] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / C;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { const int c = tid + 256 * i; const float d = c < C ? v[i] - mean : 0.f; v[i] = d; qq += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
    if (lane == 0) red[4 + w] = qq;
    __syncthreads();
    const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / C + p.ln_eps);
#pragma unroll
    for (int i = 0; i < 4; i++) { const int c = tid + 256 * i; if (c < C) xn[c] = bf2f(f2bf(v[i] * rstd * p.ln_g[c] + p.ln_b[c])); }
    __syncthreads();
    // ---- scores of this block's rows [r0, r0 + nr)
    {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = w + 4 * u;
            float a = 0.f;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int pc = lane + 64 * h;
                if (pc < npc) {
#pragma unroll
                    for (int e = 0; e < 8; e++) a += bf2f((bf16_t)g[u][h][e]) * xn[pc * 8 + e];
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0 && j < nr) sc[j] = a;
        }
    }
    __syncthreads();
    float pr = 0.f;
    if (tid < nr) {
        const int h0 = (tid / p.k) * p.k;
        float m = -INFINITY;
        for (int i = 0; i < p.k; i++) m = fmaxf(m, sc[h0 + i]);
        float sum = 0.f;
        for (int i = 0; i < p.k; i++) sum += __expf(sc[h0 + i] - m);
        pr = __expf(sc[tid] - m) / sum;
    }
    __syncthreads();
    if (tid < nr) sc[tid] = pr;
    __syncthreads();
    // ---- partial output row of this block's heads: thread = 4 channels over the nr rows requested above
    unsigned long long* const pw = (unsigned long long*)p.ws + ((long long)b * 4 + q) * C;
    const unsigned long long etag = (unsigned long long)p.epoch << 32;
    if (tid < nc4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const float pj = i < nr ? sc[i] : 0.f;
            a0 += pj * __uint_as_float(u[i].x << 16); a1 += pj * __uint_as_float(u[i].x & 0xffff0000u);
            a2 += pj * __uint_as_float(u[i].y << 16); a3 += pj * __uint_as_float(u[i].y & 0xffff0000u);
        }
        __hip_atomic_store(pw + tid * 4 + 0, etag | __float_as_uint(a0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pw + tid * 4 + 1, etag | __float_as_uint(a1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pw + tid * 4 + 2, etag | __float_as_uint(a2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pw + tid * 4 + 3, etag | __float_as_uint(a3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Drain this block's granule stores, then ONE relaxed agent-scope arrival per block (monotonic counter: every fourth arrival is a
    // sequence's last).  The drain makes the common case -- granules visible before the arrival -- the only case seen so far; the tag check
    // below is what makes the uncommon one harmless.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = ((__hip_atomic_fetch_add(p.ws_count + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 3) == 3) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    const unsigned long long* const pb = (const unsigned long long*)p.ws + (long long)b * 4 * C;
    auto granule = [&](const unsigned long long* g) -> float {
        unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) != p.epoch) {              // a value of an earlier launch: not yet this launch's -- count it, wait for ours
            atomicAdd(&g_rarm_xsplit_stale, 1ull);
            for (int spin = 0; spin < (1 << 24) && (unsigned)(v >> 32) != p.epoch; spin++) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if ((unsigned)(v >> 32) != p.epoch) return __uint_as_float(0x7fc00000u);      // never arrived (cannot happen: all four blocks have): poison, do not guess
        }
        return __uint_as_float((unsigned)v);
    };
    float xv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = tid + 256 * i;
        xv[i] = 0.f;
        if (c < C) {
            const float t0 = granule(pb + c), t1 = granule(pb + C + c), t2 = granule(pb + 2 * C + c), t3 = granule(pb + 3 * C + c);
            xv[i] = xr[c] + (p.bias[c] + ((t0 + t1) + (t2 + t3)));
            xr[c] = xv[i];
        }
    }
    // the last arriver holds the finished row: norm3 of the feed-forward that follows leaves with it (bf16 GEMM operand)
    if (p.ln3_out) rarm_emit_ln4(xv, tid, C, p.ln3_g, p.ln3_b, p.ln_eps, p.ln3_out + (long long)b * C, red);
}
#ifndef RARM_XSPLIT_DEFAULT
#define RARM_XSPLIT_DEFAULT 0
#endif
hipError_t launch_rarm_xattn_decode(const RarmXattnParams& p, hipStream_t st) {
    if (p.C % 8 || p.C > 1024 || p.heads * p.k > 128 || p.heads * p.k > p.NP || p.k < 1) return hipErrorInvalidValue;
    // The four-blocks-per-sequence form: OPT-IN (RDM_RARM_XSPLIT=1).  Round 5 made it opt-in after one unexplained bitwise mismatch in ~6 800
    // repeated decodes; round 6 replaced the hand-over's ordering assumption by self-validating granules (note above), stress-ran both forms clean
    // (profiles/r06_rarm_stress.log) -- and then measured the granule form SLOWER than one block per sequence (64 sequences 199.4 vs 198.0 img/s,
    // 128: 290.6 vs 284.5; profiles/r06_rarm_split_ab.log): the 8-byte granules and the tag checks cost more than the 1 us the split bought.
    // Never in deterministic mode (p.no_split: the choice follows the batch; the two forms add the heads in different orders).
    static const int split_on = getenv("RDM_RARM_XSPLIT") ? atoi(getenv("RDM_RARM_XSPLIT")) : RARM_XSPLIT_DEFAULT;
    // (from 128 sequences on the one-block form already fills the chip: measured equal at 256)
    if (split_on && !p.no_split && p.epoch && p.ws && p.ws_count && p.B2 <= 128 && p.heads % 4 == 0 && (p.heads / 4) * p.k <= 32 && p.C % 4 == 0) {
        rarm_xattn_decode_split_kernel<<<p.B2 * 4, 256, 0, st>>>(p);
        return hipGetLastError();
    }
    rarm_xattn_decode_kernel<<<p.B2, 1024, 0, st>>>(p);
    return hipGetLastError();
}

// ---------------------------------------------------------------- CFG + temperature + top-k filter + softmax + multinomial
// transformer.py:250-270.  One block per sequence: logits = l_u + s (l_c - l_u) (rows b and b + B of the doubled batch),
// / temperature, values below the k-th largest -> -inf (taming top_k_logits keeps ties with the k-th), softmax, then ONE draw by
// inverse CDF with the caller's uniform u in [0,1): the smallest index whose cumulative probability exceeds u * total, in
// vocabulary order (torch.multinomial's device-specific random stream has no cross-device definition; the uniform is an input
// so that the draw is reproducible and checkable).  The k-th largest value is found by a 32-round bit-wise threshold search on the
// order-preserving integer image of the floats (keys in registers).  Writes the token to out[b, *pos], to the next-step token buffer (both CFG
// halves), and — last block — advances *pos.
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

__global__ __launch_bounds__(256) void rarm_sample_kernel(RarmSampleParams p) {
    __shared__ uint32_t hist[256];
    __shared__ float red[256];
    __shared__ int chosen;
    const int b = blockIdx.x, tid = threadIdx.x, V = p.vocab;
    const int t = *p.pos;
    const float* lc = p.logits + (long long)b * V;
    const float* lu = p.cfg ? p.logits + (long long)(b + p.B) * V : nullptr;
    const float inv_t = 1.0f / p.temperature;
    auto logit_g = [&](int i) { const float c = lc[i]; return (lu ? lu[i] + p.scale * (c - lu[i]) : c) * inv_t; };
    // The guided, temperature-scaled logits are formed ONCE (coalesced reads) and kept in LDS: the select's four passes, the max,
    // the partial sums and the final scan had each gone back to global memory, the last two one strided / dependent element at a
    // time (117 us per step).  Element i lives at i + (i >> 6): a thread's contiguous 64-element chunk then spreads over the banks.
    extern __shared__ float lg[];
    for (int i0 = tid; i0 < V; i0 += 256 * 16) {              // sixteen independent (pairs of) loads in flight per thread
        float tmp[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { const int i = i0 + u * 256; tmp[u] = i < V ? logit_g(i) : 0.f; }
#pragma unroll
        for (int u = 0; u < 16; u++) { const int i = i0 + u * 256; if (i < V) lg[i + (i >> 6)] = tmp[u]; }
    }
    __syncthreads();
    auto logit = [&](int i) { return lg[i + (i >> 6)]; };
    // ---- the top_k-th largest value: bit-by-bit search for the largest threshold that at least top_k keys reach.  Keys (the
    // order-preserving integer image of the logits) sit in registers, a round is 64 compares per thread + one block-wide count:
    // 32 rounds.  (The 4-pass radix select it replaces spent its time in wave-aggregated LDS-atomic histogram updates: ~180 us.)
    const int remaining0 = p.top_k < V ? p.top_k : V;
    constexpr int KPT = 64;                                   // keys per thread: vocabularies up to 16 384; longer ones loop below
    uint32_t prefix = 0;
    if (V <= 256 * KPT) {
        uint32_t key[KPT];
#pragma unroll
        for (int u = 0; u < KPT; u++) { const int i = tid + u * 256; key[u] = i < V ? f2ord(logit(i)) : 0u; }      // 0 is below every real key
        // TWO bits per round (round 4): a round is a block-wide count + one barrier whatever it counts, so three thresholds per round
        // (prefix | 01, 10, 11 at the bit pair) halve the 32 rounds; counts of 01 / 10 ride in one register (<= 4096 per wave each)
        for (int bit = 30; bit >= 0; bit -= 2) {
            const uint32_t c1 = prefix | (1u << bit), c2 = prefix | (2u << bit), c3 = prefix | (3u << bit);
            uint32_t n12 = 0, n3 = 0;
#pragma unroll
            for (int u = 0; u < KPT; u++) { n12 += (key[u] >= c1 ? 1u : 0u) + (key[u] >= c2 ? 0x10000u : 0u); n3 += key[u] >= c3 ? 1u : 0u; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { n12 += __shfl_xor(n12, o); n3 += __shfl_xor(n3, o); }
            const int slot = ((bit >> 1) & 1) * 8;                                       // two alternating slots: one barrier per round
            if ((tid & 63) == 0) { hist[slot + (tid >> 6) * 2] = n12; hist[slot + (tid >> 6) * 2 + 1] = n3; }
            __syncthreads();
            const uint32_t t12 = hist[slot] + hist[slot + 2] + hist[slot + 4] + hist[slot + 6];          // <= 16384 per field
            const int tot3 = (int)(hist[slot + 1] + hist[slot + 3] + hist[slot + 5] + hist[slot + 7]);
            const int tot1 = (int)(t12 & 0xffffu), tot2 = (int)(t12 >> 16);
            if (tot3 >= remaining0) prefix = c3; else if (tot2 >= remaining0) prefix = c2; else if (tot1 >= remaining0) prefix = c1;
        }
    } else {
        for (int bit = 31; bit >= 0; bit--) {
            const uint32_t cand = prefix | (1u << bit);
            int cnt = 0;
            for (int i = tid; i < V; i += 256) cnt += f2ord(logit(i)) >= cand ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if ((tid & 63) == 0) hist[(bit & 1) * 4 + (tid >> 6)] = (uint32_t)cnt;
            __syncthreads();
            const int tot = (int)(hist[(bit & 1) * 4] + hist[(bit & 1) * 4 + 1] + hist[(bit & 1) * 4 + 2] + hist[(bit & 1) * 4 + 3]);
            if (tot >= remaining0) prefix = cand;
        }
    }
    __syncthreads();
    const uint32_t kth = prefix;                          // order image of the k-th largest logit: keep o >= kth
    // ---- max, then per-thread partial sums over a CONTIGUOUS chunk (vocabulary order), block scan, locate the draw
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) { const float v = logit(i); if (f2ord(v) >= kth) mx = fmaxf(mx, v); }
    red[tid] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    const int chunk = (V + 255) / 256, i0 = tid * chunk, i1 = min(V, i0 + chunk);
    // (round 4) the three scans below add in EXACTLY the order they always did -- the draw is defined by it -- but no longer wait for one
    // LDS read / one exponential per addition: loads and exponentials are issued in batches of 16, the additions then run out of registers
    // (the serial forms were ~30 of the kernel's 77 us: 256 + 64 + 64 dependent LDS round trips)
    float part = 0.f;
    for (int i = i0; i < i1; i += 16) {
        float e[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { const float v = (i + u < i1) ? logit(i + u) : 0.f; e[u] = (i + u < i1 && f2ord(v) >= kth) ? __expf(v - mx) : -1.f; }
#pragma unroll
        for (int u = 0; u < 16; u++) if (e[u] >= 0.f) part += e[u];
    }
    red[tid] = part; __syncthreads();
    if (tid == 0) {
        float total = 0.f;
        for (int j = 0; j < 256; j += 16) {
            float r[16];
#pragma unroll
            for (int u = 0; u < 16; u++) r[u] = red[j + u];
#pragma unroll
            for (int u = 0; u < 16; u++) total += r[u];
        }
        const float target = p.uniforms[(long long)(t - p.pos0) * p.B + b] * total;
        float run = 0.f; int c = -1;
        for (int j = 0; j < 256; j += 16) {
            float r[16];
#pragma unroll
            for (int u = 0; u < 16; u++) r[u] = red[j + u];
#pragma unroll
            for (int u = 0; u < 16; u++) { const float nr = run + r[u]; if (c < 0) { if (nr > target) c = j + u; else run = nr; } }
        }
        if (c < 0) c = 255;
        // (if rounding pushed the target past the total, fall into the last non-empty chunk)
        while (c > 0 && red[c] == 0.f) c--;
        chosen = c; red[0] = run; red[1] = target;
    }
    __syncthreads();
    if (tid == chosen) {
        float run = red[0]; const float target = red[1];
        int pick = -1, last = -1;
        for (int i = i0; i < i1 && pick < 0; i += 16) {
            float e[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const float v = (i + u < i1) ? logit(i + u) : 0.f; e[u] = (i + u < i1 && f2ord(v) >= kth) ? __expf(v - mx) : -1.f; }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                if (e[u] >= 0.f && pick < 0) { last = i + u; run += e[u]; if (run > target) pick = i + u; }
            }
        }
        if (pick < 0) pick = last;
        p.tokens_out[(long long)b * p.steps + (t - p.pos0)] = pick;
        p.next_tokens[b] = pick;
        if (p.cfg) p.next_tokens[b + p.B] = pick;
        __threadfence();
        if (atomicAdd(p.done, 1) == p.B - 1) { *p.done = 0; *p.pos = t + 1; }     // last sequence of the step advances the clock
    }
}
hipError_t launch_rarm_sample(const RarmSampleParams& p, hipStream_t st) {
    const size_t sm = ((size_t)p.vocab + (p.vocab >> 6) + 64) * sizeof(float);
    if (sm > 150 * 1024) return hipErrorInvalidValue;          // vocabulary beyond ~37 k entries: not an RARM configuration
    static bool attr_dev[RDM_MAX_DEVICES] = {false};
    bool& attr = attr_dev[rdm_cur_device()];
    if (!attr && sm > 32 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)rarm_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    rarm_sample_kernel<<<p.B, 256, sm, st>>>(p);
    return hipGetLastError();
}

// ---------------------------------------------------------------- VQGAN codebook lookup (taming VectorQuantizer2.get_codebook_entry)
// indices [B*HW] -> bf16 token-major [B*HW, E] (== NHWC), the layout post_quant_conv (1x1) reads as a GEMM operand
__global__ __launch_bounds__(256) void codebook_gather_kernel(const long long* idx, const float* codebook, int n_embed, int E, long long n,
                                                              bf16_t* out) {
    const long long total = n * E;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / E; const int c = (int)(i - r * E);
        long long id = idx[r]; if (id < 0 || id >= n_embed) id = 0;
        out[i] = f2bf(codebook[id * E + c]);
    }
}
hipError_t launch_codebook_gather(const long long* idx, const float* codebook, int n_embed, int E, long long n, bf16_t* out, hipStream_t st) {
    const long long total = n * E;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    codebook_gather_kernel<<<grid, 256, 0, st>>>(idx, codebook, n_embed, E, n, out);
    return hipGetLastError();
}
// f32 [rows, C] += nothing; bf16 copy of the fp32 residual stream for the logits GEMM is launch_cast_f32_bf16 (misc.hip)
__global__ void set_int_kernel(int* p, int v) { *p = v; }
hipError_t launch_set_int(int* p, int v, hipStream_t st) { set_int_kernel<<<1, 1, 0, st>>>(p, v); return hipGetLastError(); }

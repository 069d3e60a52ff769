// Internal kernel-launch interface of librdm_hip (host side). All launches are asynchronous on `st`.
#pragma once
#include "common.h"

struct GnParams {
    const bf16_t* x0; const bf16_t* x1;   // [B, HW, C0], [B, HW, C1] (x1 may be null)
    int C0, C1, HW, B, groups;
    int L0, L1;                           // logical channels of x0 / x1 (<= C0 / C1: the tails are zero padding, build_unet); 0 = C0 / C1
    int nchunk;                           // pixel chunks per sample
    float* partial;                       // [B, nchunk, groups, 2] (sum, sumsq)
    const float* gamma; const float* beta;
    float eps; int silu;
    bf16_t* out;                          // [B, HW, C0+C1]
    int x1_bmod;                          // > 0: x1 holds only x1_bmod samples and sample b reads b % x1_bmod (a shared-prefix skip tensor whose second half was never materialised)
    int b0;                               // first sample of this launch (launch_groupnorm walks big tensors in sample ranges: blockIdx.y + b0)
    int x0_bmod;                          // the same for x0 (round 5: the activation leaving the shared guidance prefix is not duplicated either)
};

hipError_t launch_gn_stats(GnParams p, hipStream_t st);          // the statistics pass alone (p.partial), for consumers that apply the norm themselves

// GroupNorm-apply + SiLU + 3x3 conv to a few output channels in one kernel (the UNet's `out` head, the VQ decoder's conv_out)
struct HeadParams {
    const bf16_t* x; int B, H, W, C;                  // NHWC bf16, raw (pre-norm) when partial is given
    const float* partial; int nchunk, groups;         // GroupNorm statistics from launch_gn_stats ([B][nchunk][groups][2]) or null: x is used as is
    const float* gamma; const float* beta; float eps;
    const float* w;                                   // [Cout][C][3][3] fp32
    bf16_t* wp;                                       // scratch for the packed weights: head_conv_wp_bytes(C)
    const float* bias; float* out; int Cout;          // out NCHW fp32
};
size_t head_conv_wp_bytes(int C);
bool head_conv_supported(const HeadParams& p);
hipError_t launch_head_conv(const HeadParams& p, hipStream_t st);

struct FlashParams {
    const bf16_t* q; int ldq;        // q[(b*n + i)*ldq + h*32 + d]
    const bf16_t* k; int ldk;        // k[(b*n + j)*ldk + h*32 + d]
    const bf16_t* vt;                // vt[((b*C) + h*32 + d)*n + j]   (V transposed per sample), or null with:
    const bf16_t* v; int ldv;        // v[(b*n + j)*ldv + h*32 + d]     (token-major V, n % 64 == 0 only: read through ds_read_b64_tr_b16)
    bf16_t* out; int ldo;            // out[(b*n + i)*ldo + h*32 + d]
    int n, C;                        // tokens, channels (= heads*32)
    float scale_log2e;               // d^-0.5 * log2(e)
    int xcd_remap;                   // set by the launcher: XCD-aware (sample, head) grouping of the query blocks
};

// cross-attention over k retrieved neighbours in the re-associated form (model.hip: unet_compute_xattn): per sample b
//   out = softmax_groups(x G_b^T) U_b^T + bias + res      x [rows][C], G_b [NP][C], U_b [C][NP]  (all bf16, row-major)
struct XattnParams {
    const bf16_t* x; const bf16_t* G; const bf16_t* U;
    const float* bias; const bf16_t* res; bf16_t* out;
    int rows, n, C, NP;              // rows = samples * n (n rows per sample, n % 32 == 0); NP = row count of G (128)
    int ncols, group;                // heads * k used score columns; softmax over groups of `group` (1, 2, 4) adjacent columns
    const float* ln_g; const float* ln_b; float ln_eps;   // given: x holds the RAW rows, the scores are taken on LayerNorm(x) and the residual is x itself (res unused)
    // LN-fused form only: out may alias x (a block reads its 32 rows before it writes them); ln3_out given: LayerNorm(out rows; ln3_g,
    // ln3_b, ln_eps) -- norm3 of BasicTransformerBlock (attention.py:239) -- leaves with the finished tile as the next GEMM's operand
    const float* ln3_g; const float* ln3_b; bf16_t* ln3_out;
};
bool xattn_fused_supported(const XattnParams& p);
hipError_t launch_xattn_fused(const XattnParams& p, hipStream_t st);        // G / U: FRAGMENT-ORDERED images, written by:
hipError_t launch_xattn_pack(const bf16_t* G, const bf16_t* U, bf16_t* Gp, bf16_t* Up, int B, int NP, int C, hipStream_t st);

struct SmallAttnParams {
    const bf16_t* q; int ldq;        // q[(b*nq + i)*ldq + h*D + d]
    const bf16_t* k; int ldk; const bf16_t* v; int ldv;   // k[(b*nkv + j)*ldk + h*D + d]
    bf16_t* out; int ldo;
    int nq, nkv, causal;
    float scale;
};

struct DdimStepParams {
    const float* x; const float* eps; const float* noise;   // noise may be null (eta == 0)
    float* x_prev; float* pred_x0;                           // pred_x0 may be null
    float* x_dup;                                            // second copy of x_prev (the CFG-doubled batch) or null
    long long n_per_batch;                                   // B*C*H*W
    float a_t, a_prev, sigma_t, sqrt_one_minus_at, scale, temperature;
    int cfg;
};

struct DdpmStepParams {
    const float* x; const float* eps; const float* noise; float* x_prev; long long n;
    float sqrt_recip, sqrt_recipm1, coef1, coef2, log_var; int clip, nonzero; float temperature;
};

// RARM decode step (rarm.hip)
struct RarmAttnParams {
    const bf16_t* q; int ldq;                 // q[b*ldq + h*64 + d]; k_new / v_new share ldq (one fused qkv row per sequence)
    const bf16_t* k_new; const bf16_t* v_new; // self-attention: the new token's rows (appended to the cache at *pos); null for cross-attention
    bf16_t* Kc; bf16_t* Vc;                   // cache [B][rows][row_stride]
    long long batch_stride; int row_stride;
    long long head_stride;                    // elements between the heads of a cache row set: 0 / 64 = heads side by side inside a row (row_stride = C); rows * 64 = head-major [B][head][rows][64] (row_stride = 64)
    int nkv;                                  // cross-attention: rows to attend; self-attention: capacity (<= 1024)
    const int* pos;                           // device step counter (self-attention attends rows 0..*pos)
    float scale; bf16_t* out; int ldo;
};
// decode-step cross-attention over the k neighbours with the per-sequence operands re-associated once per sampling call (as the UNet's
// xattn kernels): x += softmax_heads(LN(x) G_b^T) UT_b + bias for sequences b < Bc; x += bias for the others (zero neighbours)
struct RarmXattnParams {
    float* x;                                  // [B2][C] fp32 residual stream, updated in place
    const float* ln_g; const float* ln_b; float ln_eps;
    const bf16_t* G; const bf16_t* UT;         // [Bc][NP][C] each: G row h*k + j = scale * (key j restricted to head h) W_q;  UT row h*k + j = W_o (value j restricted to head h)
    const float* bias;                         // [C]
    int B2, Bc, C, NP, heads, k;
    const float* ln3_g; const float* ln3_b; bf16_t* ln3_out;   // given: LayerNorm (gamma, beta) of the FINISHED rows leaves with them as bf16 [B2][C] (norm3: the operand of the feed-forward's first GEMM)
    float* ws; int* ws_count;                  // four-blocks-per-sequence form: partial output rows [B2][4][C] as 8-byte {fp32, epoch tag} granules (zeroed once) and one monotonic arrival counter per sequence; null = one block per sequence
    unsigned epoch;                            // four-blocks form: this launch's tag, unique per launch over the buffer's lifetime, never 0 (rarm.hip: self-validating hand-over)
    int no_split;                              // deterministic mode: the form must not follow the batch -- always one block per sequence
};
unsigned long long rarm_xsplit_stale_count();     // granules the four-blocks form's last arrivers had to re-read since library load (debug counter 0)
hipError_t launch_rarm_xattn_decode(const RarmXattnParams& p, hipStream_t st);

struct RarmSampleParams {
    const float* logits; int vocab; int B; int cfg; float scale, temperature; int top_k;
    const float* uniforms;                    // [steps][B], row = *pos - pos0
    int* pos; int pos0; int steps;            // tokens_out[b*steps + (*pos - pos0)]
    long long* tokens_out; long long* next_tokens; int* done;
};
hipError_t launch_rarm_embed(const long long* tokens, const float* emb, const float* pos_t, const int* pos, float* x, int B, int C,
                             int vocab, hipStream_t st);
hipError_t launch_rarm_decode_attention(const RarmAttnParams& p, int heads, int batch, hipStream_t st);
hipError_t launch_rarm_sample(const RarmSampleParams& p, hipStream_t st);
hipError_t launch_codebook_gather(const long long* idx, const float* codebook, int n_embed, int E, long long n, bf16_t* out, hipStream_t st);
hipError_t launch_set_int(int* p, int v, hipStream_t st);

// skinny GEMM, M <= 128 rows (sgemm.hip): out[M, N(/2 for GEGLU)] = act(A W^T + bias) (+ res)
struct SgemmParams {
    const bf16_t* A; int lda; const bf16_t* W; int M, N, K;     // W [N][K]; GEGLU: N = 2 x outputs, rows interleaved in blocks of 32
    const float* bias; int act; const float* res_f32; const bf16_t* res_bf16; float* out_f32; bf16_t* out_bf16; int ldo;
    const float* ln_x; const float* ln_g; const float* ln_b; float ln_eps;      // A = LayerNorm(ln_x [M][K] fp32) formed in the kernel (A ignored)
    int fixed_split;          // deterministic mode (rdm_set_deterministic): a row's fp32 summation order must not follow the batch -- always the four-wave K split
                              // (K / 4 per wave, partial tiles added in wave order), never the eight-wave / folded / LayerNorm-in-tile forms that start at 384 rows
};
bool sgemm_supported(const SgemmParams& p);
hipError_t launch_sgemm(const SgemmParams& p, hipStream_t st);
// mid-size GEMM (mgemm.hip): one-row-per-sequence operands at 1024+ rows, LDS-staged BM x 64 tiles, fp32 / bf16 residual and output
bool mgemm_supported(const SgemmParams& p);
hipError_t launch_mgemm(const SgemmParams& p, hipStream_t st);
hipError_t launch_igemm(const IgemmParams& p, bool conv, int batch, hipStream_t st);
bool conv_halo_supported(const IgemmParams& p);
int conv_halo_ksplit(const IgemmParams& p);                // K-split factor worth using for this conv (1 = none); needs p.ws of ksplit*M*N floats
hipError_t launch_conv_halo(const IgemmParams& p, hipStream_t st);
// one-wave-per-SIMD halo kernel (conv_halo4.hip): needs p.Wfrag, the fragment-ordered weight copy
bool conv_halo4_supported(const IgemmParams& p);
bool conv_halo4_strip_supported(const IgemmParams& p);     // output wider than 64 pixels: 64-column strips (N % 128 == 0, no K-split); needs p.Wfrag
hipError_t launch_conv_halo4(const IgemmParams& p, hipStream_t st);
// one-wave-per-SIMD linear GEMM (lin4.hip): needs p.Wfrag = the fragment-ordered copy of W built by launch_lin_w_fragpack
bool lin4_supported(const IgemmParams& p, int batch);
hipError_t launch_lin4(const IgemmParams& p, hipStream_t st);
hipError_t launch_lin_w_fragpack(const bf16_t* W, bf16_t* dst, int N, int K, int ldw, int geglu, hipStream_t st, const float* gamma = nullptr);       // dst: N*K elements; geglu: W rows in packing.py's _geglu_perm order; gamma: dst = bf16(gamma[k] W[n][k]) (LayerNorm fold)
hipError_t launch_lin_ln_sb(const bf16_t* W, const float* gamma, const float* beta, const float* bias, float* sb, int N, int K, hipStream_t st);   // sb[n] = (sum_k bf16(gamma W), bias + sum_k beta W), n = stored row
hipError_t launch_conv_w_fragpack(const bf16_t* W, bf16_t* dst, int N, int Cin, hipStream_t st);   // dst: N*9*Cin elements
// 3x3 conv dispatcher: input-stationary halo kernels when the geometry allows, else the generic implicit GEMM
hipError_t launch_conv_phase_weights(const bf16_t* W, bf16_t* Wp, int N, int C, hipStream_t st);     // igemm.hip: [N][3][3][C] -> [4 phases][N][2][2][C]
inline hipError_t launch_conv3x3(const IgemmParams& p, hipStream_t st) {
    if (p.Wfrag && p.Wout > 64 && conv_halo4_strip_supported(p)) return launch_conv_halo4(p, st);
    return conv_halo_supported(p) ? launch_conv_halo(p, st) : launch_igemm(p, true, 1, st);
}
// ---- backward pieces (backward.hip)
hipError_t launch_conv_w_dgrad(const bf16_t* w, bf16_t* wd, int N, int C, hipStream_t st);             // [N][9][C] -> flipped [C][9][N]
size_t conv_wgrad_scratch_bytes(int B, int H, int W, int C, int N, int* WP, int* PR, int* Kc, int* Z, int* margin);
hipError_t launch_conv_wgrad(const bf16_t* x, const bf16_t* dy, float* dw, int B, int H, int W, int C, int N, char* scratch, const void* zero_page,
                             hipStream_t st);
hipError_t launch_reduce_planes(const float* parts, float* out, long long n, int Z, hipStream_t st);
hipError_t launch_add_bf16(const bf16_t* a, const bf16_t* b, bf16_t* out, long long n, hipStream_t st);
// GEGLU on an unpermuted pre-activation [M, 2F] ([x | gate]): dh null -> out [M, F] = x gelu(gate); else out [M, 2F] = [dx | dgate]
hipError_t launch_geglu(const bf16_t* pre, const bf16_t* dh, bf16_t* out, long long M, int F, hipStream_t st);
hipError_t launch_colsum(const bf16_t* x, float* out, long long M, int N, hipStream_t st, float* scratch = nullptr);     // scratch: colsum_scratch_bytes(M, N)
size_t colsum_scratch_bytes(long long M, int N);
// dW [N][K] fp32 = dy^T a for dy [M, N], a [M, K] bf16 (Linear weight gradient): K-split over the M rows, fp32 planes summed in fixed order
size_t linear_wgrad_scratch_bytes(long long M, int N, int K);
// wgrad.hip: dW = dY^T X read in the activation layout (no transposes), taps = 1 (linear) or 9 (conv3x3 over [B][H][W] images)
bool wgrad_tn_supported(long long M, int N, int K, int lda, int ldb);
bool conv_wgrad_tn_supported(int B, int H, int W, int C, int N);
size_t wgrad_tn_scratch_bytes(long long M, int N, int K, int taps);
size_t conv_wgrad_tn_scratch_bytes(int B, int H, int W, int C, int N);
hipError_t launch_wgrad_tn(const bf16_t* dy, int lda, const bf16_t* x, int ldb, float* dw, long long M, int N, int K, int taps, int H, int W, char* scratch,
                           const void* zero_page, hipStream_t st);
hipError_t launch_linear_wgrad(const bf16_t* dy, const bf16_t* a, float* dw, long long M, int N, int K, char* scratch, const void* zero_page, hipStream_t st);
size_t groupnorm_bwd_scratch_bytes(int B, int HW, int C, int groups);
hipError_t launch_groupnorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, const float* beta, int B, int HW, int C, int groups,
                                float eps, int silu, float* scratch, bf16_t* dx, float* dgamma, float* dbeta, hipStream_t st, const bf16_t* residual = nullptr);    // residual: dx = gradient + residual
hipError_t launch_layernorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, int M, int C, float eps, float* scratch, int* nb_out,
                                bf16_t* dx, float* dgamma, float* dbeta, hipStream_t st, const bf16_t* residual = nullptr);
hipError_t launch_groupnorm(GnParams p, hipStream_t st);
hipError_t launch_layernorm(const void* x, int in_is_f32, const float* gamma, const float* beta, void* out, int out_is_f32,
                            int M, int C, float eps, hipStream_t st, int Clog = -1);   // Clog: logical width of zero-padded rows (statistics over Clog)
hipError_t launch_flash_d32(const FlashParams& p, int heads, int batch, hipStream_t st);
hipError_t launch_small_attention(const SmallAttnParams& p, int D, int heads, int batch, hipStream_t st);
size_t small_attention_bwd_scratch_bytes(int B, int heads, int nq, int nkv);
hipError_t launch_small_attention_bwd(const bf16_t* q, int ldq, const bf16_t* k, const bf16_t* v, int ldkv, const bf16_t* dout, int ldo, int B, int nq, int nkv,
                                      int heads, float scale, bf16_t* dq, bf16_t* dk, bf16_t* dv, char* scratch, hipStream_t st);    // backward.hip: d_head 32, nkv <= 32
hipError_t launch_conv_in(const float* x, const float* w, const float* bias, bf16_t* out, int B, int Cin, int H, int W,
                          int Cout, hipStream_t st);
hipError_t launch_conv_out(const bf16_t* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cin,
                           int Cout, hipStream_t st);
hipError_t launch_timestep_embedding(const long long* t, bf16_t* out, int B, int dim, int ld, hipStream_t st);   // rows of ld >= dim, tail zeroed
hipError_t launch_cast_f32_bf16(const float* x, bf16_t* y, long long n, hipStream_t st);
hipError_t launch_transpose_bf16(const bf16_t* x, bf16_t* y, int rows, int cols, hipStream_t st, int batch = 1, int ldy = 0);       // y[c][r] = x[r][c] (batch contiguous matrices)
hipError_t launch_multi_tensor(int n, float* const* p, const float* const* g, float* const* m, float* const* v, void* const* pb, const long long* numel, int ema,
                               float lr, float b1, float b2, float eps, float wd, int step, float omd, hipStream_t st);      // AdamW / LitEma over a list of tensors, 48 per launch
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* pb, long long n, float lr, float b1, float b2, float eps, float wd, int step,
                        hipStream_t st);
hipError_t launch_ema(float* shadow, const float* p, long long n, float one_minus_decay, hipStream_t st);
hipError_t launch_silu(const float* x, const float* dy, bf16_t* ob, float* of, long long n, hipStream_t st);
// training-step glue (backward.hip, end): q_sample, squared-error loss + gradient, conditioning switch, scaling, per-sample column sums, 2x expansions
hipError_t launch_q_sample(const float* x0, const float* noise, const float* a, const float* b, float* out, bf16_t* out_nhwc, int B, int C, int HW, int cpad, hipStream_t st);
hipError_t launch_mse_loss(const bf16_t* eps, const float* target, const float* coef, float* se, bf16_t* deps, int B, int C, int HW, int ldc, hipStream_t st);
hipError_t launch_where_rows(const unsigned char* mask, const float* a, const float* x, float* out, long long rows, long long n, hipStream_t st);
hipError_t launch_scale_f32(float* x, long long n, float s, hipStream_t st);
size_t colsum_samples_scratch_bytes(int B, int HW, int N);
hipError_t launch_colsum_samples(const bf16_t* x, bf16_t* out, int B, int HW, int N, hipStream_t st, float* scratch = nullptr);
hipError_t launch_expand2(const bf16_t* x, bf16_t* out, int B, int H, int W, int C, int mode, hipStream_t st);
hipError_t launch_sumpool2(const bf16_t* x, bf16_t* out, int B, int H, int W, int C, hipStream_t st);
// fused attention backward, d_head = 32 (backward.hip): no score matrix in memory
size_t attn_bwd_scratch_bytes(int B, int H, int n, int m);
hipError_t launch_attention_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* o, const bf16_t* dout, int B, int n, int m, int H,
                                bf16_t* dq, bf16_t* dk, bf16_t* dv, char* scratch, hipStream_t st);
// attention backward helpers (backward.hip)
hipError_t launch_heads(const bf16_t* x, bf16_t* out, int B, int n, int H, int D, int ldx, int mode, hipStream_t st);
hipError_t launch_softmax_bwd(const bf16_t* P, const float* dP, bf16_t* dS, long long rows, int n, hipStream_t st);
hipError_t launch_expand_heads(const bf16_t* kv, int ld, int B, int k, int heads, int hd, int NP, float scale, bf16_t* out, hipStream_t st);
hipError_t launch_add_bias_rows(const bf16_t* x, const float* bias, bf16_t* out, long long rows, int C, hipStream_t st);
hipError_t launch_row_nonzero(const float* x, int rows, long long n, int* flag, hipStream_t st);   // flag[r] = row r has a non-zero element
hipError_t launch_ddim_step(const DdimStepParams& p, hipStream_t st);
hipError_t launch_ddpm_step(const DdpmStepParams& p, hipStream_t st);
hipError_t launch_vq_quantize(const float* z, const float* codebook, int n_embed, const float* pq_w, const float* pq_b,
                              float* out, int* idx_out, int B, int HW, int quantize, hipStream_t st);
hipError_t launch_softmax_rows(const float* s, bf16_t* p, long long rows, int n, hipStream_t st, int n_valid = 0);   // columns >= n_valid: probability 0
hipError_t launch_clip_embed(const long long* tokens, const float* tok_emb, const float* pos_emb, float* out, int B, int L,
                             int Wd, hipStream_t st);
hipError_t launch_clip_gather_eot(const long long* tokens, const float* x, float* out, int B, int L, int Wd, hipStream_t st);
hipError_t launch_clip_patchify(const float* img, bf16_t* out, int B, int R, int P, hipStream_t st);
// bicubic resize (align_corners) + (x+1)/2 + CLIP mean/std: into f32 [B,3,R,R] (out_patch null) or the bf16 patch matrix
hipError_t launch_clip_preprocess(const float* img, int B, int H, int W, int R, int P, float* out_f32, bf16_t* out_patch, hipStream_t st);
hipError_t launch_clip_vit_assemble(const float* patch, const float* cls, const float* pos, float* out, int B, int GG, int Wd,
                                    hipStream_t st);
hipError_t launch_gather_rows_f32(const float* x, float* out, int B, long long row_stride, int Wd, hipStream_t st);
hipError_t launch_to_uint8_hwc(const float* x, unsigned char* out, int B, int C, int H, int W, hipStream_t st);

// box calibration probes (calib.hip): a fixed MFMA stream on random bf16 operands and a fixed HBM copy, timed on `st`
hipError_t run_calib_probes(void* buf, double mfma_ms, size_t stream_bytes, int stream_reps, double* mfma_tflops, double* stream_gbps, hipStream_t st);

// fused feed-forward (ffn.hip, round 6: a measurement vehicle, C = 384 only): out = [x gelu(g) | t2] Wf^T + bf + xin, [x | g] = l3 W1^T + b1
size_t ffn_fused_scratch_bytes(int C);
bool ffn_fused_supported(int M, int C);
hipError_t launch_ffn_fused(const bf16_t* l3, const bf16_t* t2, const bf16_t* xin, const bf16_t* w1, const float* b1, const bf16_t* wf, const float* bf,
                            bf16_t* out, int M, int C, char* scratch, bool repack, hipStream_t st);

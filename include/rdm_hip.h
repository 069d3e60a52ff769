/* librdm_hip.so — C ABI of the MI355X-native retrieval-augmented diffusion sampling path.
 *
 * The reference (CompVis/retrieval-augmented-diffusion-models) is pure Python and has no FFI
 * boundary; each entry point below names the reference call it replaces (file:line relative to
 * the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - return 0 on success, negative on error; rdm_last_error(ctx) gives the message.
 *   - no exceptions cross the ABI; no torch types; plain pointers and sizes.
 *   - pointers marked [dev] are device pointers on the context's HIP device (e.g.
 *     torch.Tensor.data_ptr()); [host] are host pointers.  The caller owns every buffer; the
 *     library keeps no reference to caller memory after a call returns, except rdm_db_load with
 *     copy = 0, which is documented there.
 *   - work is enqueued on the context's stream (rdm_set_stream; default = the null stream, which is
 *     also torch's default current stream) and calls return after enqueue unless noted.
 *   - a context is bound to one device and is not thread-safe.
 *   - layouts at the boundary are the reference's: NCHW fp32 images/latents, [B,k,512] fp32
 *     conditioning, int64 timesteps/tokens.  Internal layout (NHWC bf16) is private.
 */
#ifndef RDM_HIP_H
#define RDM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rdm_ctx rdm_ctx;

#define RDM_MAX_LEVELS 8

/* UNetModel constructor arguments that shape the sampling graph
 * (rdm/modules/diffusionmodules/openaimodel.py:66-129; models/rdm/imagenet/config.yaml:36-59). */
typedef struct {
    int in_channels, out_channels, model_channels, num_res_blocks;
    int n_attention_resolutions, attention_resolutions[RDM_MAX_LEVELS];
    int n_channel_mult, channel_mult[RDM_MAX_LEVELS];
    int num_head_channels, context_dim;
} rdm_unet_cfg;

/* First-stage decoder: ldm VQModelInterface ddconfig (models/rdm/imagenet/config.yaml:60-80) and taming VQModel ddconfig of the
 * RARM models (models/rarm/imagenet/dogs/config.yaml:28-51: embed_dim = z_channels = 256, ch_mult 1,1,2,2,4, attn_resolutions [16]). */
typedef struct {
    int embed_dim, n_embed, z_channels, ch, n_ch_mult, ch_mult[RDM_MAX_LEVELS];
    int num_res_blocks, out_ch, resolution, mid_attn, kl; /* kl=1: AutoencoderKL decode (no quantiser) */
    int n_attn_resolutions, attn_resolutions[RDM_MAX_LEVELS];   /* AttnBlock after every ResnetBlock of the levels at these resolutions */
} rdm_vq_cfg;

/* RetrievalPatchTransformer constructor arguments of the RARM models (rdm/modules/attention.py:206-249;
 * models/rarm/imagenet/dogs/config.yaml:14-27: continuous = false, causal, cross_attend, positional_encodings). */
typedef struct {
    int vocab_in, vocab_out;      /* in_channels (token embedding rows, incl. mask / sos tokens), out_channels (logits) */
    int n_heads, d_head, depth, context_dim, sequence_length;
} rdm_rarm_cfg;

/* LatentImageRETRO.sample arguments (rdm/models/autoregression/transformer.py:224-294). */
typedef struct {
    int batch, k;                 /* sequences, neighbours per sequence in r [batch,k,context_dim] */
    int cond_len, steps;          /* conditioning tokens c [batch,cond_len] (the sos token), tokens to sample */
    float temperature; int top_k; /* top_k <= 0: no filter */
    float guidance_scale;         /* > 1: batch doubled with zero neighbours, logits_u + s (logits_c - logits_u) (:237-253) */
} rdm_rarm_sample_args;

/* CLIP constructor arguments (rdm/modules/custom_clip/model.py:238-252). */
typedef struct {
    int embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size;
    int context_length, vocab_size, transformer_width, transformer_heads, transformer_layers;
} rdm_clip_cfg;

/* DDIMSampler.sample arguments the native loop implements (rdm/models/diffusion/ddim.py:58-140). */
typedef struct {
    int S;                 /* number of DDIM steps */
    int batch;             /* B */
    int k;                 /* neighbours in the conditioning [B,k,context_dim] */
    int channels, height, width;      /* latent shape (3,64,64) */
    float eta, temperature;
    float unconditional_guidance_scale;   /* >= 1; > 1 enables CFG batch doubling (ddim.py:229-238) */
    int log_every_t;       /* intermediates rule, ddim.py:207 */
    /* schedule: the model's fp32 alphas_cumprod buffer [T] (ddim.py:30) */
    int T; const float* alphas_cumprod; /* [host] */
} rdm_ddim_args;

/* ldm LatentDiffusion.p_sample_loop arguments (reached from rdm/models/diffusion/ddpm.py:1008). */
typedef struct {
    int timesteps;         /* loop runs reversed(range(timesteps)) */
    int batch, k, channels, height, width;
    int clip_denoised; float temperature;
    int T;                 /* schedule length; arrays below are fp32 [T], [host] */
    const float* sqrt_recip_alphas_cumprod; const float* sqrt_recipm1_alphas_cumprod;
    const float* posterior_mean_coef1; const float* posterior_mean_coef2;
    const float* posterior_log_variance_clipped;
} rdm_ddpm_args;

/* ---- context -------------------------------------------------------------------------------- */
int rdm_ctx_create(int device_id, rdm_ctx** out);
void rdm_ctx_destroy(rdm_ctx* ctx);
const char* rdm_last_error(rdm_ctx* ctx);
/* Frees the grow-only work buffers of the context (backward scratch incl. up to 256 MB of fp32 weight-gradient planes per conv,
 * split-K planes, per-call weight re-packs, sampler scratch); they are re-created on demand.  Weights, caches and arenas stay. */
int rdm_release_scratch(rdm_ctx* ctx);
int rdm_set_stream(rdm_ctx* ctx, void* hip_stream);
const char* rdm_version(void);
/* Batch-invariant ("deterministic") execution: every kernel-selection decision of the library (skinny vs tiled GEMM, halo vs generic
 * 3x3 conv, conv split-K, the zero-context shortcut) becomes a function of the per-sample layer shape only, so a sample's result is
 * bitwise independent of the batch it is computed in and of the number of ranks the batch is sharded over (the reference gives no such
 * guarantee either way: cuDNN picks algorithms per shape).  Default off (env RDM_DETERMINISTIC=1 turns it on for new contexts); the
 * speed cost is stated in DESIGN.md. */
int rdm_set_deterministic(rdm_ctx* ctx, int on);
int rdm_get_deterministic(rdm_ctx* ctx);

/* ---- weights --------------------------------------------------------------------------------
 * The library defines the packed-blob layout; rdm_*_manifest writes it as text, one line per entry:
 *     <offset> <nbytes> <kind> <src>[,<src>...]
 * where <src> are the reference's own state_dict keys and <kind> is one of
 *   f32 | bf16 | bf16_t | conv3 | geglu_w | geglu_b   (see DESIGN.md section 2), or a DERIVED entry
 *   fuse_w  <ff.net.2.weight>,<proj_out.weight>                  -> bf16 [C][5C] = [W_out W_2 | W_out]   (multiplied out in fp64)
 *   fuse_b  <ff.net.2.bias>,<proj_out.weight>,<proj_out.bias>    -> f32  [C]     = W_out b_2 + b_out
 * (the SpatialTransformer's ff.net.2 and proj_out run as one GEMM, attention.py:88-96 + 190-195).
 * The caller fills a host blob accordingly and hands it to rdm_load_*, which copies it to HBM.
 * Returns the number of bytes needed for the text (excluding NUL) if buf is too small. */
long long rdm_unet_manifest(const rdm_unet_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes);
long long rdm_vq_manifest(const rdm_vq_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes);
long long rdm_clip_manifest(const rdm_clip_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes);
/* replaces torch.load + load_state_dict for model.diffusion_model / first_stage_model / CLIP
 * (scripts/rdm_sample.py:163-170, rdm/modules/retrievers.py:76). packed [host]. */
int rdm_load_unet(rdm_ctx* ctx, const rdm_unet_cfg* cfg, const void* packed, size_t nbytes);
int rdm_load_vq(rdm_ctx* ctx, const rdm_vq_cfg* cfg, const void* packed, size_t nbytes);
int rdm_load_clip(rdm_ctx* ctx, const rdm_clip_cfg* cfg, const void* packed, size_t nbytes);
/* RARM transformer (state_dict keys of rdm.modules.attention.RetrievalPatchTransformer, prefix `transformer.` stripped);
 * extra manifest kind  f32_t  = 2-D tensor stored transposed (positional_encoding [C,L] -> [L][C]). */
long long rdm_rarm_manifest(const rdm_rarm_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes);
int rdm_load_rarm(rdm_ctx* ctx, const rdm_rarm_cfg* cfg, const void* packed, size_t nbytes);

/* ---- UNet: UNetModel.forward(x, timesteps, context) (openaimodel.py:335-371) via
 *      MinimalRETRODiffusion.apply_model (rdm/models/diffusion/ddpm.py:445-458).
 * x [dev] f32 [b,Cin,H,W]; t [dev] int64 [b]; ctx [dev] f32 [b,k,context_dim]; eps_out [dev] f32 [b,Cout,H,W]. */
int rdm_unet_forward(rdm_ctx* ctx, const float* x, const int64_t* t, const float* context, int b, int k, int H, int W,
                     float* eps_out);

/* ---- samplers -------------------------------------------------------------------------------
 * DDIMSampler.sample / ddim_sampling / p_sample_ddim (rdm/models/diffusion/ddim.py:58-268).
 * x_T [dev] f32 [B,C,H,W]; cond, uncond [dev] f32 [B,k,ctx_dim] (uncond may be NULL when scale == 1);
 * noise [dev] f32 [S,B,C,H,W] or NULL (required when eta > 0; consumed in loop order);
 * z_out [dev] f32 [B,C,H,W]; x_inter / pred_x0_inter [dev] f32 [n_inter,B,C,H,W] or NULL, filled for the
 * steps ddim.py:207 logs (n_inter = rdm_ddim_num_intermediates). */
int rdm_ddim_num_intermediates(int S, int log_every_t);
int rdm_ddim_sample(rdm_ctx* ctx, const rdm_ddim_args* args, const float* x_T, const float* cond, const float* uncond,
                    const float* noise, float* z_out, float* x_inter, float* pred_x0_inter);
/* ldm LatentDiffusion.p_sample_loop / p_sample (no CFG on this path).
 * noise [dev] f32 [timesteps,B,C,H,W], consumed in loop order. */
int rdm_ddpm_sample(rdm_ctx* ctx, const rdm_ddpm_args* args, const float* x_T, const float* cond, const float* noise,
                    float* z_out);

/* ---- first stage: LatentDiffusion.decode_first_stage -> VQModelInterface.decode
 *      (called at rdm/models/diffusion/ddpm.py:840, 981). z [dev] f32 [b,3,h,w] -> img [dev] f32 [b,3,H,W];
 * indices_out [dev] int32 [b*h*w] or NULL. */
int rdm_vq_decode(rdm_ctx* ctx, const float* z, int b, int force_not_quantize, float* img_out, int32_t* indices_out);
/* first_stage_model.quantize(z) (taming VectorQuantizer2.forward: nearest codebook entry, first minimum on ties, straight-through form
 * z + (e - z)) WITHOUT post_quant_conv -- what DDIMSampler.p_sample_ddim applies to pred_x0 under quantize_x0 = True
 * (rdm/models/diffusion/ddim.py:260-261).  z [dev] f32 [b,3,h,w] -> zq_out [dev] f32 [b,3,h,w]; indices_out [dev] int32 [b*h*w] or NULL. */
int rdm_vq_quantize(rdm_ctx* ctx, const float* z, int b, float* zq_out, int32_t* indices_out);
/* taming Net2NetTransformer.decode_to_img (reached from rdm/models/autoregression/transformer.py:296-312): code indices
 * [dev] int64 [b, h*w] -> quantize.get_codebook_entry -> post_quant_conv -> Decoder -> img_out [dev] f32 [b,3,R,R].
 * For first stages with a wide latent (VQGAN-f16, z_channels % 64 == 0). */
int rdm_vq_decode_indices(rdm_ctx* ctx, const int64_t* indices, int b, float* img_out);
/* ---- first stage, encoder side (training input): LatentDiffusion.encode_first_stage -> VQModelInterface.encode = quant_conv(encoder(x))
 *      under torch.no_grad(), reached from MinimalRETRODiffusion.shared_step -> get_input (rdm/models/diffusion/ddpm.py:390-391).
 * Same rdm_vq_cfg as the decoder (the encoder mirrors it: ddconfig is shared); state_dict keys `encoder.*`, `quant_conv.*`.
 * img [dev] f32 [b,out_ch,R,R] -> z_out [dev] f32 [b,embed_dim,R/f,R/f] (no quantisation: VQModelInterface quantises in decode). */
long long rdm_vqenc_manifest(const rdm_vq_cfg* cfg, char* buf, size_t buflen, size_t* blob_bytes);
int rdm_load_vqenc(rdm_ctx* ctx, const rdm_vq_cfg* cfg, const void* packed, size_t nbytes);
int rdm_vq_encode(rdm_ctx* ctx, const float* img, int b, float* z_out);
/* scripts/rdm_sample.py:203-214 custom_to_np/custom_to_pil: f32 NCHW [-1,1] -> uint8 NHWC (truncating). */
int rdm_to_uint8(rdm_ctx* ctx, const float* img, int b, int c, int h, int w, uint8_t* out);

/* ---- RARM: RetrievalPatchTransformer.forward(x, context) (rdm/modules/attention.py:199-272) and LatentImageRETRO.sample
 *      (rdm/models/autoregression/transformer.py:224-294).  Both run token by token against a per-layer K/V cache (the
 *      reference re-runs the whole prefix for every new token, :241-248).
 * forward: tokens [dev] int64 [b,t]; context [dev] f32 [b,k,context_dim]; logits_out [dev] f32 [b,t,vocab_out].
 * sample : cond_tokens [dev] int64 [b,cond_len]; context as above; uniforms [dev] f32 [steps,b] in [0,1) — the multinomial draw
 *          of step s for sequence i is the inverse CDF (vocabulary order) at uniforms[s,i]; tokens_out [dev] int64 [b,steps]. */
int rdm_rarm_forward(rdm_ctx* ctx, const int64_t* tokens, int b, int t, const float* context, int k, float* logits_out);
int rdm_rarm_sample(rdm_ctx* ctx, const rdm_rarm_sample_args* args, const int64_t* cond_tokens, const float* context,
                    const float* uniforms, int64_t* tokens_out);

/* ---- CLIP: CLIP.encode_text / encode_image (rdm/modules/custom_clip/model.py:304-320), used by
 *      ClipImageRetriever / CLIPTextEmbedder (rdm/modules/retrievers.py:67-117).
 * tokens [dev] int64 [b,context_length]; image [dev] f32 [b,3,R,R] already CLIP-normalised; out [dev] f32 [b,embed]. */
int rdm_clip_encode_text(rdm_ctx* ctx, const int64_t* tokens, int b, float* out);
int rdm_clip_encode_image(rdm_ctx* ctx, const float* image, int b, float* out);
/* ClipImageRetriever.preprocess (rdm/modules/retrievers.py:83-91): kornia bicubic resize (align_corners=True, no antialias)
 * of image [dev] f32 [b,3,h,w] in [-1,1] to the tower resolution R, then (x+1)/2 and the CLIP mean/std -> out [dev] f32 [b,3,R,R]. */
int rdm_clip_preprocess(rdm_ctx* ctx, const float* image, int b, int h, int w, float* out);
/* ClipImageRetriever.forward (retrievers.py:93-95) = encode_image(preprocess(x)) with the preprocessing fused into the
 * patch-embedding gather (the resized image is never materialised): image [dev] f32 [b,3,h,w] in [-1,1] -> out [dev] f32 [b,embed]. */
int rdm_clip_encode_image_raw(rdm_ctx* ctx, const float* image, int b, int h, int w, float* out);

/* ---- retrieval: DatasetBuilder.train_searcher + searcher.search_batched
 *      (rdm/data/retrieval_dataset/dsetbuilder.py:534-619, 490; call sites ddpm.py:298,734,906).
 * rdm_db_load: emb [host or dev] fp16 or fp32 [n,dim] raw embeddings; the searcher's dataset is
 *   fp16(x/||x||) held in HBM (dsetbuilder.py:574).  dtype: 0 = fp16, 1 = fp32. is_device: 1 if emb is [dev].
 * rdm_knn: q [dev] f32 [b,dim] raw queries (normalised internally, dsetbuilder.py:487);
 *   idx_out [dev] uint32 [b,k] by descending score, ties -> lower index; score_out [dev] f32 [b,k] (or NULL). */
int rdm_db_load(rdm_ctx* ctx, const void* emb, long long n, int dim, int dtype, int is_device);
long long rdm_db_size(rdm_ctx* ctx);
int rdm_knn(rdm_ctx* ctx, const float* q, int b, int k, uint32_t* idx_out, float* score_out);
/* The same search with the scores as the fp64 values the ranking was made on (score_out [dev] f64 [b,k]).  For a database whose ROWS are
 * sharded over GPUs ("when memory matters"): every rank searches its rows, the per-rank lists are exchanged (one
 * all-gather of [b,k] (index, score) pairs) and merged under the same total order (score desc, global index asc) -- on the fp32
 * scores two rows whose fp64 scores differ could tie and swap. */
int rdm_knn_f64(rdm_ctx* ctx, const float* q, int b, int k, uint32_t* idx_out, double* score_out);
/* 1 if the last rdm_knn could not certify its MFMA-scored candidate set for some query (a cluster of near-duplicates within the
 * score error bound around the k-th neighbour) and answered it by the exact fp64 pass instead; 0 otherwise.  Diagnostic only:
 * the result is exact either way. */
int rdm_knn_last_fallback(rdm_ctx* ctx);
/* data_pool['embedding'][nns] gather (dsetbuilder.py:493): idx [dev] uint32 [n_idx] -> out [dev] f32 [n_idx,dim]
 * of the RAW (un-normalised) embeddings (rdm_db_load keeps a raw copy in HBM next to the normalised one). */
int rdm_db_gather(rdm_ctx* ctx, const uint32_t* idx, long long n_idx, float* out);

/* ---- multi-GPU: one process and one context per device; the path has ONE exchange, an all-gather -- of the
 * decoded images (replicated database) or of the per-shard (index, score) top-k pairs (row-sharded database).  Thin wrappers
 * around RCCL (loaded with dlopen at rdm_comm_init: single-GPU users never touch it) for callers that bind the C ABI directly;
 * the Python mirror reaches the same RCCL through torch.distributed (rdm_amd/parallel.py).  The reference has no multi-GPU
 * sampling (scripts/rdm_sample.py is single-process); the reference-side analogue is DDP's process group in main.py:783-785.
 *   rdm_comm_unique_id: rank 0 fills a 128-byte id and hands it to the other ranks by any out-of-band means.
 *   rdm_comm_init:      every rank, same id; binds the communicator to the context's device and stream.
 *   rdm_comm_all_gather: recv[dev] (world * nbytes) <- send[dev] (nbytes) of every rank, in rank order; enqueued on the stream. */
int rdm_comm_unique_id(rdm_ctx* ctx, void* id128 /*[host] 128 bytes*/);
int rdm_comm_init(rdm_ctx* ctx, const void* id128 /*[host]*/, int rank, int world);
int rdm_comm_all_gather(rdm_ctx* ctx, const void* send /*[dev]*/, void* recv /*[dev]*/, size_t nbytes);
/* gradient all-reduce of data-parallel training (the reference: DDP inside pytorch_lightning's Trainer, main.py:783-785): buf [dev] f32
 * [count] <- sum over ranks (average != 0: divided by the world size), in place, enqueued on the stream. */
int rdm_comm_all_reduce_f32(rdm_ctx* ctx, float* buf /*[dev]*/, size_t count, int average);
int rdm_comm_destroy(rdm_ctx* ctx);

/* ---- measurement: optional HIP-event brackets around launches on the context stream, by kernel class.
 * enable: bit mask of (1 << RDM_PROF_*) kinds to record (0 = off).  collect: sums elapsed ms and ALGORITHMIC work of the
 * launches of one kind recorded since the last reset -- FLOPs (2*M*N*K; 4*n^2*d per attention head) for the MFMA-bound
 * kinds, bytes (minimal tensor / database passes) for the HBM-bound kinds.  Used by bench.py for the roofline objects. */
enum { RDM_PROF_CONV3X3 = 0, RDM_PROF_LINEAR = 1, RDM_PROF_KNN = 2, RDM_PROF_ATTENTION = 3, RDM_PROF_GROUPNORM = 4,
       RDM_PROF_LAYERNORM = 5,
       RDM_PROF_UPSCONV = 6 /* Upsample's nearest-2x + conv3x3 run by output phase (2 x 2 taps at source resolution): EXECUTED FLOPs, 4/9 of the nine-tap count */ };
int rdm_prof_enable(rdm_ctx* ctx, int kind_mask);
int rdm_prof_collect(rdm_ctx* ctx, int kind, long long* launches, double* ms, double* flops);
int rdm_prof_reset(rdm_ctx* ctx);
/* one CSV row per recorded launch since the last reset: kind, the op's role in the executor's graph ("st.proj_in", "res.conv1", ...; the
 * reference modules behind the roles: rdm/modules/attention.py:122-196, ldm ResBlock), its shape (M, N, K | rows, channels | B, n, C),
 * elapsed ms and algorithmic work -- the per-op table behind DESIGN.md's level-by-level costs (tools/op_trace.py). */
int rdm_prof_dump(rdm_ctx* ctx, const char* path);
/* out = [ x gelu(g) | t2 ] wf^T + bf + xin with [x | g] = l3 w1^T + b1: the GEGLU feed-forward of BasicTransformerBlock and proj_out's residual
 * (rdm/modules/attention.py:77-96 `self.ff(self.norm3(x)) + x`, :194 `self.proj_out(x) + x_in`; ldm FeedForward / GEGLU) in ONE kernel, the hidden tensor never
 * in HBM -- round 6's measurement vehicle for that fusion (compiler-scheduled; csrc/ffn.hip), NOT used by the executors: it measured slower than the two
 * kernels it would replace.  bf16 [M, C] activations; w1 bf16 [8C, C] and b1 f32 [8C] in the packed GEGLU row order of rdm_op_linear
 * ([32 x | 32 gates] blocks); wf bf16 [C, 5C] = [W_out W_2 | W_out]; bf f32 [C].  C = 384 and M % 128 == 0 only (-5 otherwise). */
int rdm_op_ffn_fused(rdm_ctx* ctx, const void* l3_bf16, const void* t2_bf16, const void* xin_bf16, const void* w1_bf16, const float* b1,
                     const void* wf_bf16, const float* bf, void* out_bf16, int M, int C);
/* test / stress hook: library-wide debug counters.  which = 0: granules the four-blocks-per-sequence RARM decode cross-attention
 * (rarm.hip; the decode step of rdm/models/autoregression/transformer.py:241-248) had to RE-READ because their tag was an earlier launch's
 * -- the hand-over is self-validating, so a non-zero count is harmless, and it is the round-5 repeat mismatch caught in the act
 * (tools/rarm_stress.py).  Synchronises the context's stream. */
int rdm_debug_counter(rdm_ctx* ctx, int which, unsigned long long* value /*[host]*/);
/* Box calibration for bench.py's `calibration` object (no reference counterpart: the reference has no benchmark harness; the
 * boxes of the MI355X pool differ by +-4..5 % in sustained clock under the package power cap and the headline moves with them).  Two FIXED
 * instruction streams that do not change when a product kernel does: (1) back-to-back v_mfma_f32_32x32x16_bf16 on random bf16 operands, one wave
 * per SIMD on every CU, for ~mfma_ms of wall time -> sustained dense-bf16 TFLOP/s under this box's power cap; (2) stream_reps 16-byte-per-lane
 * copies of stream_bytes -> HBM read + write GB/s.  buf [dev]: scratch of at least max(1 MiB, 2 * stream_bytes) bytes; either result
 * pointer may be null (probe skipped).  Synchronous on the context's stream. */
int rdm_calib_probe(rdm_ctx* ctx, void* buf /*[dev]*/, size_t buf_bytes, double mfma_ms, size_t stream_bytes, int stream_reps,
                    double* mfma_tflops /*[host]*/, double* stream_gbps /*[host]*/);
/* test hook: the next UNet forwards copy ONE intermediate activation (bf16, row-major [rows, width] as the executor holds it: NHWC) into
 * buf [dev] (at most nbytes).  block: index into the top-level block table (state-dict order: input_blocks.0 .., middle_block,
 * output_blocks.0 ..; openaimodel.py:355-368's loop).  sub = 0: the block's output; sub = 16 * (layer inside the block) + stage:
 *   ResBlock stages 1 in_layers norm+SiLU, 2 in_layers conv + emb, 3 out_layers norm+SiLU, 4 skip_connection, 5 output;
 *   SpatialTransformer stages 1 norm, 2 proj_in, 3 norm1, 4 q|k|v, 5 attn1 heads, 6 x + attn1, 7 x + attn2, 8 norm3, 9 GEGLU, 10 output
 *   (rdm/modules/attention.py:92-96, 170-196).  buf null: off.  tests/test_gpu_emul.py compares the library with the CPU restatement of its
 * arithmetic stage by stage, TEACHER-FORCED, through this. */
int rdm_debug_tap(rdm_ctx* ctx, void* buf /*[dev]*/, size_t nbytes, int block, int sub);

/* ---- operator-level entry points (used by the parity tests; thin wrappers over the kernels) ---- */
int rdm_op_linear(rdm_ctx* ctx, const void* a_bf16, const void* w_bf16, const float* bias, const void* residual_bf16,
                  void* out_bf16, float* out_f32, int M, int N, int K, int act, float alpha);
/* out[m] = a[m] W^T + bias + rowvec[m / rows_per_group] (+ residual[m]): nn.Linear with a per-row-group additive vector (rowvec f32
 * [ceil(M / rows_per_group), N]).  The executor uses it for attn1.to_out over a guided batch [conditional | unconditional]
 * (rdm/modules/attention.py:237-238 with ddim.py:229-234's batch): the unconditional rows' cross-attention is exactly attn2.to_out's
 * bias, which rides in the projection's start values for those rows (two groups; any group count on the generic GEMM). */
int rdm_op_linear_rowvec(rdm_ctx* ctx, const void* a_bf16, const void* w_bf16, const float* bias, const float* rowvec, int rows_per_group,
                         const void* residual_bf16, void* out_bf16, int M, int N, int K);
/* out = act(LayerNorm(x; gamma, beta, eps) W^T + bias) in ONE kernel: nn.LayerNorm followed by nn.Linear as in BasicTransformerBlock
 * (rdm/modules/attention.py:147-168: `self.attn1(self.norm1(x))`, `self.ff(self.norm3(x))`), the LayerNorm folded into the GEMM
 * (lin4.hip).  x bf16 [M, K] RAW rows, w bf16 [N, K] (GEGLU: rows in the packed [32 x | 32 gates] order), out bf16 [M, N] (GEGLU: [M, N/2]).
 * Returns -5 when the shape is not one the folded kernel takes (the executors then run LayerNorm + Linear). */
int rdm_op_linear_ln(rdm_ctx* ctx, const void* x_bf16, const void* w_bf16, const float* bias, const float* gamma, const float* beta,
                     void* out_bf16, int M, int N, int K, int act, float eps);
int rdm_op_conv3x3(rdm_ctx* ctx, const void* x0_bf16, const void* x1_bf16, int C0, int C1, const void* w_bf16,
                   const float* bias, const float* rowvec, int rowvec_ld, const void* residual_bf16, void* out_bf16,
                   int B, int Hin, int Win, int N, int stride, int ups);
/* the RARM sampler kernel alone (taming top_k_logits + softmax + multinomial as an inverse CDF in vocabulary order, after the
 * classifier-free combine lu + scale (lc - lu) and the temperature; transformer.py:237-263): logits [dev] f32 [(cfg ? 2 : 1) * b, vocab]
 * (conditional rows first), uniforms [dev] f32 [b], tokens_out [dev] int64 [b].  top_k <= 0: no filter. */
int rdm_op_rarm_sampler(rdm_ctx* ctx, const float* logits, int b, int vocab, int cfg, float guidance_scale, float temperature,
                        int top_k, const float* uniforms, int64_t* tokens_out);
/* ---- backward building blocks of the training step (reference rdm/models/diffusion/ddpm.py:390-443 shared_step -> ldm
 * p_losses -> autograd through UNetModel): gradients of the ResBlock's ops.  bf16 activations / activation gradients, fp32 weight
 * gradients.  conv3x3: stride 1, pad 1, weights [N][3][3][C].  Tested against torch autograd (tests/test_gpu_backward.py). */
int rdm_op_conv3x3_dgrad(rdm_ctx* ctx, const void* dy_bf16 /*[B,H,W,N]*/, const void* w_bf16, void* dx_bf16 /*[B,H,W,C]*/, int B, int H, int W,
                         int C, int N);
int rdm_op_conv3x3_wgrad(rdm_ctx* ctx, const void* x_bf16 /*[B,H,W,C]*/, const void* dy_bf16 /*[B,H,W,N]*/, float* dw /*[N,3,3,C]*/, int B, int H,
                         int W, int C, int N);
int rdm_op_groupnorm_bwd(rdm_ctx* ctx, const void* x_bf16, const void* dy_bf16, const float* gamma, const float* beta, int B, int HW, int C,
                         float eps, int silu, void* dx_bf16, float* dgamma, float* dbeta);
int rdm_op_layernorm_bwd(rdm_ctx* ctx, const void* x_bf16, const void* dy_bf16, const float* gamma, int M, int C, float eps, void* dx_bf16,
                         float* dgamma, float* dbeta);
/* The same two with the gradient of a residual path joined in the kernel: dx = (norm gradient) + residual (bf16, dx's shape; one rounding).
 * Where autograd adds the branch's gradient to the skip connection's (x + f(norm(x)) in ResBlock / BasicTransformerBlock). */
int rdm_op_groupnorm_bwd_add(rdm_ctx* ctx, const void* x_bf16, const void* dy_bf16, const float* gamma, const float* beta, int B, int HW, int C,
                             float eps, int silu, const void* residual_bf16, void* dx_bf16, float* dgamma, float* dbeta);
int rdm_op_layernorm_bwd_add(rdm_ctx* ctx, const void* x_bf16, const void* dy_bf16, const float* gamma, int M, int C, float eps,
                             const void* residual_bf16, void* dx_bf16, float* dgamma, float* dbeta);
/* nn.Linear weight gradient dW [N, K] fp32 = dy^T a for dy [M, N], a [M, K] bf16 (autograd of F.linear in the training step): K-split
 * over the M rows, deterministic fixed-order sum of the fp32 partial planes. */
int rdm_op_linear_wgrad(rdm_ctx* ctx, const void* dy_bf16, const void* a_bf16, float* dw, long long M, int N, int K);
int rdm_op_colsum(rdm_ctx* ctx, const void* x_bf16 /*[M,N]*/, float* out /*[N]*/, long long M, int N);
int rdm_op_transpose(rdm_ctx* ctx, const void* x_bf16 /*[rows,cols]*/, void* y_bf16 /*[cols,rows]*/, int rows, int cols);
int rdm_op_add(rdm_ctx* ctx, const void* a_bf16, const void* b_bf16, void* out_bf16, long long n);
/* Elementwise pieces of the UNet's training graph: SiLU of the time-embedding MLP (`nn.SiLU()` in
 * openaimodel.py time_embed / emb_layers) -- dy null: out bf16 = silu(x), else out fp32 = dy * silu'(x) -- and the 2 x 2 sum pooling
 * that is the gradient of Upsample's nearest-neighbour F.interpolate: x bf16 [B, 2H, 2W, C] -> out bf16 [B, H, W, C]. */
int rdm_op_silu(rdm_ctx* ctx, const float* x, const float* dy_or_null, void* out_bf16_or_f32, long long n);
int rdm_op_sumpool2(rdm_ctx* ctx, const void* x_bf16, void* out_bf16, int B, int H, int W, int C);
/* The elementwise steps of MinimalRETRODiffusion.shared_step -> forward -> ldm p_losses around the UNet (rdm/models/diffusion/ddpm.py:390-443):
 *   rdm_op_q_sample   LatentDiffusion.q_sample: x_t = sqrt_ac[b] x0 + sqrt_1mac[b] noise; x0 / noise / out f32 NCHW [B,C,H,W] (out may be
 *                     NULL), out_nhwc bf16 [B,H,W,cpad] (may be NULL; channels >= C zero): the operand of the native training forward;
 *   rdm_op_mse_loss   p_losses (l2, eps parameterisation): se[b] = mean_{chw} (eps - target)^2 and, when deps is given,
 *                     deps = coef[b] (eps - target); eps / deps bf16 NHWC [B,H,W,ldc] (first C channels), target f32 NCHW;
 *   rdm_op_where_rows out[b,:] = mask[b] ? a[b,:] : x[b,:] (the Bernoulli(p_uncond) conditioning switch, :393-396); mask uint8 [rows];
 *   rdm_op_timestep_embedding  ldm timestep_embedding: t int64 [B] -> bf16 [B, ld] = [cos | sin | zero tail];
 *   rdm_op_colsum_samples      x bf16 [B,HW,N] -> bf16 [B,N], sum over a sample's pixels (gradient of the time-embedding row a ResBlock adds);
 *   rdm_op_expand2    x bf16 [B,H,W,C] -> [B,2H,2W,C]: mode 0 zero insertion (stride-2 conv gradient as a stride-1 one), mode 1 nearest copy. */
int rdm_op_q_sample(rdm_ctx* ctx, const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, float* out_or_null,
                    void* out_nhwc_bf16_or_null, int B, int C, int H, int W, int cpad);
int rdm_op_mse_loss(rdm_ctx* ctx, const void* eps_nhwc_bf16, const float* target, const float* coef_or_null, float* se, void* deps_nhwc_bf16_or_null,
                    int B, int C, int H, int W, int ldc);
int rdm_op_where_rows(rdm_ctx* ctx, const unsigned char* mask, const float* a, const float* x, float* out, long long rows, long long n);
int rdm_op_timestep_embedding(rdm_ctx* ctx, const int64_t* t, void* out_bf16, int B, int dim, int ld);
int rdm_op_colsum_samples(rdm_ctx* ctx, const void* x_bf16, void* out_bf16, int B, int HW, int N);
int rdm_op_expand2(rdm_ctx* ctx, const void* x_bf16, void* out_bf16, int B, int H, int W, int C, int mode);
/* LitEma.forward (ldm/modules/ema.py: `shadow.sub_(one_minus_decay * (shadow - param))`; the caller computes
 * decay = min(decay, (1 + num_updates) / (10 + num_updates)) like the reference) on fp32 tensors in place. */
int rdm_op_ema(rdm_ctx* ctx, float* shadow, const float* param, long long n, float one_minus_decay);
/* One AdamW step (torch.optim.AdamW: decoupled weight decay, bias-corrected moments; the reference's configure_optimizers,
 * rdm/models/diffusion/ddpm.py, hands the UNet parameters to it) on fp32 master parameters / moments in place; p_bf16 (optional) receives
 * the bf16 working copy the kernels read.  step counts from 1. */
int rdm_op_adamw(rdm_ctx* ctx, float* p, const float* grad, float* exp_avg, float* exp_avg_sq, void* p_bf16_or_null, long long n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int step);
/* The same two over LISTS of tensors (host arrays of n device pointers and element counts; p_bf16 may be null, or hold null entries): the
 * UNet has 688 parameter tensors, most of them tiny -- 48 tensors per launch instead of one launch each.  Same arithmetic per element. */
int rdm_op_adamw_multi(rdm_ctx* ctx, int n, float* const* p, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                       void* const* p_bf16_or_null, const long long* numel, float lr, float beta1, float beta2, float eps, float weight_decay, int step);
int rdm_op_ema_multi(rdm_ctx* ctx, int n, float* const* shadow, const float* const* param, const long long* numel, float one_minus_decay);
/* Attention backward, unfused first version (autograd through ldm CrossAttention.forward, attention.py:52-72:
 * sim = einsum(q, k) * scale; attn = sim.softmax(-1); out = einsum(attn, v)).  The scores are materialised per (sample, head):
 *   rdm_op_heads   x [B, n, ldx] (head h = columns [h D, (h+1) D)) <-> per-head matrices zero-padded to 64 columns:
 *                  mode 0 -> [B H][n][64], mode 1 -> transposed [B H][64][n], mode 2: [B H][n][64] -> [B, n, H D];
 *   rdm_op_bmm     out[z] = alpha * A[z] W[z]^T for batch contiguous bf16 matrices A [M, K], W [N, K] (K % 64 == 0), bf16 and / or fp32 out;
 *   rdm_op_transpose_batched, rdm_op_softmax (fp32 scores -> bf16 probabilities, row-wise), rdm_op_softmax_bwd
 *   (dS = P * (dP - rowsum(P * dP))).  rdm_amd/training.py attention_forward / attention_backward assemble them. */
/* Fused attention backward for d_head = 32 (no n x m score matrix): q, o, dout [B, n, heads * 32], k, v [B, m, heads * 32] bf16 (o = the forward
 * output) -> dq [B, n, C], dk, dv [B, m, C] bf16.  n, m multiples of 32 (pad the keys and mask them with rdm_op_softmax otherwise). */
int rdm_op_attention_bwd(rdm_ctx* ctx, const void* q, const void* k, const void* v, const void* o, const void* dout, int B, int n, int m, int heads,
                         void* dq, void* dk, void* dv);
int rdm_op_bmm(rdm_ctx* ctx, const void* a_bf16, const void* w_bf16, void* out_bf16_or_null, float* out_f32_or_null, int batch, int M, int N, int K,
               float alpha);
int rdm_op_heads(rdm_ctx* ctx, const void* x_bf16, void* out_bf16, int B, int n, int H, int D, int ldx, int mode);
int rdm_op_transpose_batched(rdm_ctx* ctx, const void* x_bf16, void* y_bf16, int batch, int rows, int cols);
int rdm_op_softmax(rdm_ctx* ctx, const float* scores, void* p_bf16, long long rows, int n, int n_valid /* columns >= n_valid are padding: probability 0 (0 = all valid) */);
int rdm_op_softmax_bwd(rdm_ctx* ctx, const void* p_bf16, const float* dp, void* ds_bf16, long long rows, int n);
/* GEGLU (ldm attention.py GEGLU.forward: `x, gate = proj(x).chunk(2, dim=-1); return x * F.gelu(gate)`) on an UNPERMUTED
 * pre-activation pre [M, 2F] = [x | gate], bf16.  dh null: forward, out [M, F].  dh [M, F] given: backward, out [M, 2F] = [dx | dgate]
 * (what autograd computes for that line in the training step). */
int rdm_op_geglu(rdm_ctx* ctx, const void* pre_bf16, const void* dh_bf16_or_null, void* out_bf16, long long M, int F);
int rdm_op_groupnorm(rdm_ctx* ctx, const void* x0_bf16, const void* x1_bf16, int C0, int C1, int B, int HW,
                     const float* gamma, const float* beta, float eps, int silu, void* out_bf16);
int rdm_op_layernorm(rdm_ctx* ctx, const void* x, int in_is_f32, const float* gamma, const float* beta, int M, int C,
                     float eps, void* out_bf16);
int rdm_op_self_attention(rdm_ctx* ctx, const void* qk_bf16, const void* vt_bf16, int B, int n, int heads,
                          void* out_bf16);
/* the same attention from ONE fused projection qkv [B, n, 3C] = [q | k | v] (the UNet sampling path's form: V stays token-major and is
 * transposed inside the kernel's LDS reads); n % 64 == 0 */
int rdm_op_self_attention_qkv(rdm_ctx* ctx, const void* qkv_bf16, int B, int n, int heads, void* out_bf16);
/* CrossAttention over k retrieved neighbours (rdm/modules/attention.py:52-72) in the re-associated per-sample form the UNet path uses:
 *   out[b] = softmax_groups(x[b] G[b]^T) U[b]^T + bias + res[b]
 * x / res / out bf16 [B, n, C], G bf16 [B, NP, C] (row h*k + j = key j restricted to head h, times W_q / sqrt(d)), U bf16 [B, C, NP]
 * (column h*k + j = W_o applied to value j restricted to head h); softmax over groups of `group` (= k: 1, 2, 4) adjacent columns of
 * the first ncols = heads * k columns.  n % 32 == 0, C % 64 == 0, NP % 32 == 0, ncols <= min(NP, 128).  bias / res may be null.
 * With ln_gamma / ln_beta given (res null) the LayerNorm in front and the residual are folded in as well:
 *   out[b] = softmax_groups(LayerNorm(x[b]) G[b]^T) U[b]^T + bias + x[b]      (norm2 + attn2 + residual of BasicTransformerBlock,
 * attention.py:238). */
int rdm_op_xattn_fused(rdm_ctx* ctx, const void* x_bf16, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* G_bf16,
                       const void* U_bf16, const float* bias, const void* res_bf16, int B, int n, int C, int NP, int ncols, int group,
                       void* out_bf16);
/* The LayerNorm-folded form IN PLACE with norm3 emitted as well (what the executor runs on a guided batch's conditional rows):
 *   x[b] <- softmax_groups(LayerNorm(x[b]; ln) G[b]^T) U[b]^T + bias + x[b];   ln3_out[b] = LayerNorm(x[b] (the bf16 rows just stored); ln3)
 * i.e. `x = attn2(norm2(x), context) + x` followed by `norm3(x)` of BasicTransformerBlock (attention.py:238-239) in one launch. */
int rdm_op_xattn_fused_ln3(rdm_ctx* ctx, void* x_bf16_inout, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* G_bf16,
                           const void* U_bf16, const float* bias, int B, int n, int C, int NP, int ncols, int group,
                           const float* ln3_gamma, const float* ln3_beta, void* ln3_out_bf16);
/* The UNet's `out` head (openaimodel.py:307-311: GroupNorm32 + SiLU + 3x3 conv to out_channels) and the VQ decoder's norm_out + swish +
 * conv_out as one statistics pass + one kernel: x bf16 NHWC [B, H, W, C] raw, 32 groups; gn_gamma / gn_beta null: no norm, x is convolved
 * as is.  w fp32 [Cout, C, 3, 3], bias fp32 [Cout] or null, out fp32 NCHW [B, Cout, H, W].  C % 32 == 0, C <= 240, W % 32 == 0, H even,
 * Cout <= 8. */
int rdm_op_head_conv(rdm_ctx* ctx, const void* x_bf16, const float* gn_gamma, const float* gn_beta, float gn_eps, const float* w,
                     const float* bias, int B, int H, int W, int C, int Cout, float* out_f32);
int rdm_op_small_attention(rdm_ctx* ctx, const void* q_bf16, int ldq, const void* k_bf16, const void* v_bf16, int ldkv,
                           int B, int nq, int nkv, int heads, int D, int causal, float scale, void* out_bf16, int ldo);
/* Gradient of rdm_op_small_attention for the UNet's cross-attention (ldm CrossAttention, rdm/modules/attention.py:52-72, with the few
 * retrieved-neighbour embeddings as keys): d_head = 32, 1..32 keys, not causal.  q [B, nq, ldq], k / v [B, nkv, ldkv], dout [B, nq, ldo]
 * (head h = columns [32 h, 32 h + 32)) -> dq [B, nq, 32 heads], dk / dv [B, nkv, 32 heads], all bf16; sums in a fixed order. */
int rdm_op_small_attention_bwd(rdm_ctx* ctx, const void* q_bf16, int ldq, const void* k_bf16, const void* v_bf16, int ldkv, const void* dout_bf16, int ldo,
                               int B, int nq, int nkv, int heads, float scale, void* dq_bf16, void* dk_bf16, void* dv_bf16);

#ifdef __cplusplus
}
#endif
#endif

// Shared device/host helpers for librdm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, same as torch's float->bfloat16 for finite values
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
// hardware RNE pack of two fp32 into one dword of two bf16 (lo in bits 0-15)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// exchange with the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float swap_adjacent_lane(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: these run once per activation element
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact-erf GELU. erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 output resolution):
// ~12 VALU instead of the ~30 of erff() -- the GEGLU epilogue evaluates this 201 M times per 32x32-level FF layer.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x + 0.5f * fabsf(x) * erf_abs;      // 0.5 x (1 + sign(x) erf|.|)
}
// two at a time on the packed-fp32 VALU forms (v_pk_mul_f32 / v_pk_fma_f32: one issue slot per pair): same formula, same error
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf_f2(f32x2_t x) {
    f32x2_t ax; ax.x = fabsf(x.x); ax.y = fabsf(x.y);
    const f32x2_t z = ax * 0.70710678118654752f;
    const f32x2_t d = z * 0.3275911f + 1.0f;
    f32x2_t t; t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
    const f32x2_t poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f32x2_t a = (z * z) * -1.4426950408889634f;
    f32x2_t e; e.x = __builtin_amdgcn_exp2f(a.x); e.y = __builtin_amdgcn_exp2f(a.y);
    const f32x2_t erf_abs = 1.0f - poly * e;
    return (x + ax * erf_abs) * 0.5f;
}
__device__ __forceinline__ float quickgelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }

// 16-byte async global->LDS copy: LDS address = wave-uniform `lds_wave_base` + lane*16.
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gptr, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// epilogue activation ids (igemm)
enum { ACT_NONE = 0, ACT_GEGLU = 1, ACT_QUICKGELU = 2, ACT_SILU = 3, ACT_SOFTMAXG = 4 };   // SOFTMAXG: softmax over groups of sm_group (1, 2, 4) adjacent columns

constexpr int RDM_EYE_OFFSET = 4096, RDM_EYE_N = 256;

// one-shot launch setup (hipFuncSetAttribute, CU counts, occupancy) is PER DEVICE: caches are indexed by the calling
// thread's current device (the C ABI binds it to the context's device at every entry point)
constexpr int RDM_MAX_DEVICES = 32;
inline int rdm_cur_device() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < RDM_MAX_DEVICES) ? d : 0; }

struct IgemmParams {
    // A operand: logical [M, K].  Two channel-concatenated sources (A1 may be null, C1 = 0).
    const bf16_t* A0; const bf16_t* A1;
    int C0, C1;                 // channels (row length) of each source; linear: K = C0 + C1
    const bf16_t* W;            // [N][K] bf16, K contiguous
    int M, N, K;
    // conv3x3 geometry (conv mode): input [B, Hin, Win, C0(+C1)], output [B, Hout, Wout, N]
    int Hin, Win, Hout, Wout, stride, ups;
    int phase2;                 // generic implicit GEMM only: conv3x3 on a nearest-2x upsampled input by output phase (igemm.hip CONV == 3): W = the phase weights
                                // [4][N][2][2][C0] (launch_conv_phase_weights), K = 4 C0, M = B Hin Win, launched with batch = 4
    int asym;                   // generic implicit GEMM only: the 3x3 window of output (oy, ox) starts AT input (stride oy, stride ox) -- F.pad(x, (0, 1, 0, 1)) + an unpadded conv (ldm autoencoder Downsample)
    // epilogue
    float alpha;                // acc scale
    const float* bias;          // [N] (permuted for GEGLU) or null
    const float* rowvec;        // [B, rowvec_ld] per-sample per-column add (time embedding) or null
    int rowvec_ld, rows_per_sample;
    const bf16_t* res_bf16;     // [M, ldo] residual or null
    const float* res_f32;       // [M, ldo] residual or null
    bf16_t* out_bf16; float* out_f32;   // either / both
    int ldo;                    // output row stride (elements)
    int act;
    int sm_group;               // ACT_SOFTMAXG group size
    // batching over blockIdx.z (element strides)
    long long sA, sW, sO;
    const void* zero_page;      // 4 KiB of zeros, followed by a 256 x 256 bf16 identity matrix (RDM_EYE_OFFSET)
    int ksplit; float* ws;      // halo conv split-K (conv_halo.hip): ksplit fp32 partial tiles [ksplit][M][N] in ws, summed by a finisher
    int res_k;                  // set by launch_igemm: the bf16 residual enters as BN extra K columns against that identity
    int lda, ldw;               // linear only: row strides of A0 and W in elements when they are views into wider matrices (0 = K); backward.hip
    int l4_any_tiles;           // lin4: take the GEMM whatever its tile count (deterministic mode: the choice must not follow the batch)
    int res_wrap_rows;          // lin4 only: > 0: the bf16 residual holds only that many rows and row m adds row m % res_wrap_rows (whole tiles, at most two copies)
    int a1_wrap_rows;           // lin4 only: > 0: A1 holds only that many rows and row m reads m % a1_wrap_rows (a multiple of the block tile's rows)
    const bf16_t* Wfrag;        // conv3x3: fragment-ordered copy of W (conv_halo4.hip, built by launch_conv_w_fragpack) or null
    // lin4 only: LayerNorm folded into the GEMM.  A0 holds the RAW rows, Wfrag the gamma-scaled fragment copy, ln_sb[n] = (s[n], b'[n]) per
    // stored weight row (launch_lin_ln_sb); the kernel takes the row statistics itself: out = rstd (A Wg^T - mean s) + b'
    const float* ln_sb; float ln_inv_c, ln_eps;      // ln_inv_c = 1 / (logical row width): zero padding beyond it adds nothing to the sums
    int dbg;                    // debug ablation bits (env RDM_IGEMM_DBG): 1 no MFMA, 2 no in-loop staging, 4 no stores
};

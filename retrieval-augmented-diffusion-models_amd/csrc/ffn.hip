// Fused feed-forward of BasicTransformerBlock (round 6, verdict item 2): GEGLU -> ff.net.2 x proj_out in ONE kernel, the hidden tensor never
// in HBM.  out = [ x gelu(g) | t2 ] Wf^T + bf + x_in  with  [x | g] = l3 W1^T + b1  (rdm/modules/attention.py:77-96, ldm FeedForward / GEGLU; Wf =
// [W_out W_2 | W_out] as packing.py's fuse_w).  A MEASUREMENT VEHICLE: compiler-scheduled (builtins, no hand-placed stream), C = 384 only (the 32 x 32
// level) -- a wave that keeps a 32-row strip of the output resident needs C / 32 accumulator fragments: 12 = 192 registers at C = 384, 18 / 30 at the
// 16 x 16 / 8 x 8 levels' 576 / 960 channels, which do not fit beside the 64 of the hidden chunk.  Reached only through rdm_op_ffn_fused.
//
// Block = 4 waves = 128 rows, wave w owns rows 32 w .. 32 w + 31 of BOTH GEMMs (D^T form as lin4.hip: A operand = weight fragment, B operand =
// activation fragment, a lane ends with 16 weight rows of ONE activation row):
//   * GEMM 1 per chunk of 64 hidden units = 4 GEGLU fragments (16 x rows + THEIR 16 gate rows, lin4's fragment order): 4 accumulators; the
//     B operand is the wave's own l3 strip, held in registers for the whole tile (24 k-steps x 4 registers);
//   * GELU in registers: a lane holds x and gate of the same 8 hidden units of its row;
//   * GEMM 2: those 8 values, rounded to bf16, ARE the B fragment of one k-step (16 hidden units) -- no LDS exchange, no transpose -- once the
//     K order of Wf inside every group of 16 hidden units is permuted to the order the D^T layout leaves them in (k 0..15 <-> hidden
//     0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15: the packer does it); 12 output accumulators stay resident; the t2 part of K follows at the end;
//   * every weight fragment is needed by all four waves: fragments travel L2 -> LDS once per block (global_load_lds, 1 KiB per wave
//     instruction) through two double-buffered rings (W1: 16 fragments per stage, Wf: 48), one barrier per stage.
#include <stdlib.h>
#include <type_traits>

#include "kernels.h"

struct FfnParams {
    const bf16_t* l3; const bf16_t* t2; const bf16_t* xin; bf16_t* out;       // [M, C] each
    const bf16_t* W1f;          // fragment-ordered GEGLU weights (launch_lin_w_fragpack, geglu = 1): [8C / 32][C / 16][512]
    const float* b1f;           // [8C] bias in FRAGMENT row order
    const bf16_t* Wff;          // fragment-ordered [C / 32][5C / 16][512], hidden K groups permuted (see above)
    const float* bf;            // [C]
    int M;
};

constexpr int FFN_C = 384, FFN_KQ1 = FFN_C / 16, FFN_KQ2 = 5 * FFN_C / 16, FFN_NO = FFN_C / 32, FFN_CHUNKS = 4 * FFN_C / 64;
constexpr int FFN_RING2 = 48 * 1024;                                    // bytes per Wf stage (4 k-steps x 12 fragments); W1 stage: SKS k-steps x 4 fragments

// fragment reads as asm statements (hipcc keeps their program order; left to itself it serialises read -> wait -> MFMA once ~450 registers are live) and
// counted waits TIED to the fragments they cover, so that no MFMA that reads one can be scheduled above its wait
#define FFN_LDSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define FFN_WAIT4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(n))
#define FFN_WAIT6(n, a, b, c, d, e, f) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "i"(n))

template <int SKS>      // k-steps of GEMM 1 per LDS stage (4, 6 or 8: one barrier per stage)
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnParams p) {
    constexpr int FFN_RING1 = SKS * 4 * 1024, NSUB = FFN_KQ1 / SKS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring1 = smem; char* const ring2 = smem + 2 * FFN_RING1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long long row = (long long)blockIdx.x * 128 + wave * 32 + r;

    // ---- stage loaders: 1 KiB fragments, one global_load_lds per wave and fragment
    auto issue_w1 = [&](int t) {            // stage t = (chunk, sub-step): fragments (nb = 4 chunk + f, kq = SKS sub + ks), slot order [f][ks]
        const int chunk = t / NSUB, sub = t - chunk * NSUB;
        char* dst = ring1 + (t & 1) * FFN_RING1;
#pragma unroll
        for (int q = 0; q < SKS; q++) {
            const int idx = wave * SKS + q, f = idx / SKS, ks = idx - f * SKS;
            const char* src = (const char*)p.W1f + ((long long)((chunk * 4 + f) * FFN_KQ1 + sub * SKS + ks) * 1024) + lane * 16;
            glds16(src, dst + idx * 1024);
        }
    };
    auto issue_wf = [&](int kq0, int slot) {   // 4 k-steps kq0 .. kq0 + 3 of all 12 output fragments, slot order [ks][j]
        char* dst = ring2 + slot * FFN_RING2;
#pragma unroll
        for (int q = 0; q < 12; q++) {
            const int idx = wave * 12 + q, ks = idx / 12, j = idx - ks * 12;
            const char* src = (const char*)p.Wff + ((long long)(j * FFN_KQ2 + kq0 + ks) * 1024) + lane * 16;
            glds16(src, dst + idx * 1024);
        }
    };

    // ---- the wave's l3 strip as B fragments (k-step kq: 8 channels 16 kq + 8 h .. of row r), for the whole tile
    bf16x8 xb[FFN_KQ1];
#pragma unroll
    for (int kq = 0; kq < FFN_KQ1; kq++) xb[kq] = *(const bf16x8*)(p.l3 + row * FFN_C + kq * 16 + h * 8);

    f32x16 oacc[FFN_NO];
#pragma unroll
    for (int j = 0; j < FFN_NO; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) oacc[j][e] = 0.f;

    issue_w1(0);
    issue_wf(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int chunk = 0; chunk < FFN_CHUNKS; chunk++) {
        f32x16 cacc[4];
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int e = 0; e < 16; e++) cacc[f][e] = 0.f;
        // ---- GEMM 1: NSUB stages of SKS k-steps
#pragma unroll
        for (int sub = 0; sub < NSUB; sub++) {
            const int t = chunk * NSUB + sub;
            if (t + 1 < FFN_CHUNKS * NSUB) issue_w1(t + 1);
            if (sub == 1 && chunk + 1 < FFN_CHUNKS) issue_wf((chunk + 1) * 4, (chunk + 1) & 1);      // (all waves are past GEMM 2 of chunk - 1: its slot is free)
            const unsigned st = (unsigned)(2 * 0 + (t & 1) * FFN_RING1 + lane * 16);       // LDS byte address (ring 1 starts at 0)
            bf16x8 af[2][4];
#pragma unroll
            for (int f = 0; f < 4; f++) FFN_LDSR(af[0][f], st, (f * SKS + 0) * 1024);
#pragma unroll
            for (int ks = 0; ks < SKS; ks++) {
                if (ks + 1 < SKS) {
#pragma unroll
                    for (int f = 0; f < 4; f++) FFN_LDSR(af[(ks + 1) & 1][f], st, (f * SKS + ks + 1) * 1024);
                    FFN_WAIT4(4, af[ks & 1][0], af[ks & 1][1], af[ks & 1][2], af[ks & 1][3]);          // the four reads just issued may stay in flight
                } else FFN_WAIT4(0, af[ks & 1][0], af[ks & 1][1], af[ks & 1][2], af[ks & 1][3]);
#pragma unroll
                for (int f = 0; f < 4; f++) cacc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][f], xb[sub * SKS + ks], cacc[f], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // ---- bias + GELU: register 4 j + i of fragment f = weight row 8 j + 4 h + i: rows 0..15 x, 16..31 the gates of the same hidden units
        bf16x8 hb[4];
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const float* bp = p.b1f + (chunk * 4 + f) * 32 + 4 * h;
            const float4 bx0 = *(const float4*)(bp), bx1 = *(const float4*)(bp + 8), bg0 = *(const float4*)(bp + 16), bg1 = *(const float4*)(bp + 24);
            const float bx[8] = {bx0.x, bx0.y, bx0.z, bx0.w, bx1.x, bx1.y, bx1.z, bx1.w}, bg[8] = {bg0.x, bg0.y, bg0.z, bg0.w, bg1.x, bg1.y, bg1.z, bg1.w};
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (cacc[f][e] + bx[e]) * gelu_erf_f(cacc[f][8 + e] + bg[e]);
            union { uint4 u; bf16x8 b; } t;
            t.u = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
            hb[f] = t.b;
        }
        // ---- GEMM 2: k-step ks = hidden group 4 chunk + ks against all 12 output fragments
        {
            // 8 half-steps of 6 fragments, read one half-step ahead
            const unsigned st = (unsigned)(2 * FFN_RING1 + (chunk & 1) * FFN_RING2 + lane * 16);
            bf16x8 wf6[2][6];
#pragma unroll
            for (int j = 0; j < 6; j++) FFN_LDSR(wf6[0][j], st, j * 1024);
#pragma unroll
            for (int hs = 0; hs < 8; hs++) {
                if (hs + 1 < 8) {
#pragma unroll
                    for (int j = 0; j < 6; j++) FFN_LDSR(wf6[(hs + 1) & 1][j], st, ((hs + 1) * 6 + j) * 1024);
                    FFN_WAIT6(6, wf6[hs & 1][0], wf6[hs & 1][1], wf6[hs & 1][2], wf6[hs & 1][3], wf6[hs & 1][4], wf6[hs & 1][5]);
                } else FFN_WAIT6(0, wf6[hs & 1][0], wf6[hs & 1][1], wf6[hs & 1][2], wf6[hs & 1][3], wf6[hs & 1][4], wf6[hs & 1][5]);
#pragma unroll
                for (int j = 0; j < 6; j++) oacc[(hs & 1) * 6 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf6[hs & 1][j], hb[hs >> 1], oacc[(hs & 1) * 6 + j], 0, 0, 0);
            }
        }
    }
    // ---- the t2 part of K: k-steps 4C/16 .. 5C/16 in 6 stages of 4 through ring 2
    __syncthreads();                                                          // every wave is done with the last chunk's ring-2 slot
    constexpr int KQH = 4 * FFN_C / 16;
    issue_wf(KQH, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < 6; s++) {
        if (s + 1 < 6) issue_wf(KQH + (s + 1) * 4, (s + 1) & 1);
        const char* st = ring2 + (s & 1) * FFN_RING2 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const bf16x8 b = *(const bf16x8*)(p.t2 + row * FFN_C + (s * 4 + ks) * 16 + h * 8);
#pragma unroll
            for (int j = 0; j < FFN_NO; j++) {
                const bf16x8 a = *(const bf16x8*)(st + (ks * 12 + j) * 1024);
                oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, oacc[j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- read-out: register 4 jj + i of fragment j = output channel 32 j + 8 jj + 4 h + i of row `row`: + bias + residual, one rounding
#pragma unroll
    for (int j = 0; j < FFN_NO; j++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int ch = 32 * j + 8 * jj + 4 * h;
            const float4 b4 = *(const float4*)(p.bf + ch);
            const uint2 x2 = *(const uint2*)(p.xin + row * FFN_C + ch);
            const float o0 = oacc[j][4 * jj + 0] + b4.x + __uint_as_float(x2.x << 16), o1 = oacc[j][4 * jj + 1] + b4.y + __uint_as_float(x2.x & 0xffff0000u);
            const float o2 = oacc[j][4 * jj + 2] + b4.z + __uint_as_float(x2.y << 16), o3 = oacc[j][4 * jj + 3] + b4.w + __uint_as_float(x2.y & 0xffff0000u);
            *(uint2*)(p.out + row * FFN_C + ch) = make_uint2(cvt_pk_bf16(o0, o1), cvt_pk_bf16(o2, o3));
        }
}

// Wf [C][5C] -> a copy whose hidden K columns (first 4C) are permuted inside every group of 16 to the order GEMM 1's D^T layout leaves the hidden
// units in a lane (MFMA k -> hidden: 0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15); the t2 part is copied as is.  b1 [8C] (stored GEGLU row order) -> fragment row order.
__global__ __launch_bounds__(256) void ffn_prep_kernel(const bf16_t* __restrict__ Wf, bf16_t* __restrict__ Wp, const float* __restrict__ b1, float* __restrict__ b1f, int C) {
    const int K = 5 * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)C * K; i += (long long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long long)n * K);
        int src = k;
        if (k < 4 * C) { const int g = k & ~15, kk = k & 15; const int hid = (kk & 3) | ((kk & 4) << 1) | ((kk & 8) >> 1); src = g + hid; }
        Wp[i] = Wf[(long long)n * K + src];
    }
    for (int n = blockIdx.x * 256 + threadIdx.x; n < 8 * C; n += gridDim.x * 256) {
        const int nb = n >> 5, fr = n & 31;
        b1f[n] = b1[64 * (nb >> 1) + 16 * (nb & 1) + (fr < 16 ? fr : 32 + (fr - 16))];
    }
}

size_t ffn_fused_scratch_bytes(int C) { return ((size_t)8 * C * C + (size_t)2 * 5 * C * C) * 2 + (size_t)8 * C * 4 + 1024; }
bool ffn_fused_supported(int M, int C) { return C == FFN_C && M > 0 && M % 128 == 0; }
// scratch: ffn_fused_scratch_bytes(C); w1 [8C, C] in the stored GEGLU row order ([32 x | 32 gates] blocks), b1 [8C] likewise, wf [C, 5C], bf [C]
hipError_t launch_ffn_fused(const bf16_t* l3, const bf16_t* t2, const bf16_t* xin, const bf16_t* w1, const float* b1, const bf16_t* wf, const float* bf,
                            bf16_t* out, int M, int C, char* scratch, bool repack, hipStream_t st) {
    if (!ffn_fused_supported(M, C)) return hipErrorInvalidValue;
    bf16_t* W1f = (bf16_t*)scratch; bf16_t* Wff = W1f + (size_t)8 * C * C; bf16_t* Wp = Wff + (size_t)5 * C * C; float* b1f = (float*)(Wp + (size_t)5 * C * C);
    hipError_t e;
    if (repack) {
        if ((e = launch_lin_w_fragpack(w1, W1f, 8 * C, C, C, 1, st)) != hipSuccess) return e;
        ffn_prep_kernel<<<1024, 256, 0, st>>>(wf, Wp, b1, b1f, C);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = launch_lin_w_fragpack(Wp, Wff, C, 5 * C, 5 * C, 0, st)) != hipSuccess) return e;
    }
    FfnParams p{l3, t2, xin, out, W1f, b1f, Wff, bf, M};
    static const int sks = getenv("RDM_FFN_SKS") ? atoi(getenv("RDM_FFN_SKS")) : 6;     // k-steps of GEMM 1 per LDS stage
    auto go = [&](auto tag) -> hipError_t {
        constexpr int SKS = decltype(tag)::value, smem = 2 * SKS * 4 * 1024 + 2 * FFN_RING2;
        static bool attr[RDM_MAX_DEVICES] = {};
        const int dev = rdm_cur_device();
        if (!attr[dev]) {
            hipError_t e2 = hipFuncSetAttribute((const void*)ffn_fused_kernel<SKS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e2 != hipSuccess) return e2;
            attr[dev] = true;
        }
        ffn_fused_kernel<SKS><<<M / 128, 256, smem, st>>>(p);
        return hipGetLastError();
    };
    return sks == 8 ? go(std::integral_constant<int, 8>{}) : sks == 6 ? go(std::integral_constant<int, 6>{}) : go(std::integral_constant<int, 4>{});
}

// Skinny GEMM for decode-sized batches: out[M,N] = act(A[M,K] W[N,K]^T + bias) (+ residual) with M <= 128 rows.
//
// The RARM decode step (rarm.hip) and the UNet's time-embedding MLP are M = B' <= 128 row GEMMs against 768..6144-row weight
// matrices: pure weight streaming.  The tiled kernel (igemm.hip) gives such a launch N/192 blocks (12 of 256 CUs for the
// 2304-wide qkv projection) each walking K serially -- 35 us per launch, ~100 GB/s of weights.  Here a block owns only 32
// output columns (N/32 blocks: 72..512), its four waves split K and hold the whole M x 32 partial tile in MFMA accumulators
// (32x32x16 bf16, fp32), every lane issues all of its 16-byte operand loads for a K quarter up front, and the four partial tiles
// meet in LDS for a fused bias / activation / residual epilogue.  GEGLU: the packed weight rows interleave 32 x-rows with their
// 32 gate rows (packing._geglu_perm), so a block owns a 64-row strip and emits x * gelu(gate) for 32 outputs.
#include <stdlib.h>

#include "kernels.h"

// LN: the A operand is LayerNorm(ln_x) of an fp32 [M, K] matrix, formed in the kernel (nn.LayerNorm in front of every projection of
// the RARM block, rdm/modules/attention.py:84-86, 199-272): a block needs its rows' whole K anyway (each wave one quarter), so the row
// statistics cost one cross-wave exchange and the separate LayerNorm launch (~5 us of a ~8 us GEMM) disappears.  K / 4 = 32 U only
// (one batch of loads holds a wave's whole K quarter in registers).
template <int MF, int NF, int U, bool LN = false>   // MF: 32-row blocks of M (1..4); NF: 1 plain, 2 GEGLU (x strip + gate strip); U: k-steps of 32 per batch of loads
__global__ __launch_bounds__(256) void sgemm_kernel(SgemmParams p) {
    // 16x16x32 MFMAs, not 32x32x16: the operands come straight from global memory in fragment order, and what such a launch pays
    // for is the number of cache LINES its load instructions touch (phase clocks at M = 64, K = 768: 9.1 of the launch's ~12 us are
    // the load phase).  A 32x32x16 fragment is 32 rows x 32 bytes per instruction -- 32 lines for 1 KB --, a 16x16x32 fragment is
    // 16 rows x 64 bytes: half the lines for the same bytes and the same FLOPs per cycle.
    extern __shared__ float part[];                       // [4 waves][MF][NF][32 rows][32 cols]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int n0 = blockIdx.x * 32 * NF;                  // first weight row of this block's strip
    const int mb0 = blockIdx.y * 32 * MF;                 // first of this block's rows (grid.y row blocks: more, lighter blocks -- a launch is
                                                          // as long as ONE block's dependent operand fetch, and 24 blocks left 232 CUs idle)
    const int kq = p.K >> 2, k0 = wave * kq;              // this wave's K quarter
    constexpr int MA = 2 * MF, NB = 2 * NF;               // 16-row fragments of M, 16-column fragments of the strip(s)
    const bf16_t* wp[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) wp[j] = p.W + (long long)(n0 + j * 16 + r16) * p.K + k0 + q4 * 8;
    const bf16_t* ap[MA];
#pragma unroll
    for (int i = 0; i < MA; i++) { int m = mb0 + i * 16 + r16; if (m >= p.M) m = p.M - 1; ap[i] = LN ? nullptr : p.A + (long long)m * p.lda + k0 + q4 * 8; }
    f32x4 acc[MA][NB];
#pragma unroll
    for (int i = 0; i < MA; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LN) {
        float* const lnv = part + 4 * MF * NF * 1024;           // [gamma K][beta K][4 waves][MA * 16 rows][sum, sumsq]
        float* const stat = lnv + 2 * p.K;
        for (int k = tid * 4; k < p.K; k += 1024) {
            *(float4*)(lnv + k) = *(const float4*)(p.ln_g + k);
            *(float4*)(lnv + p.K + k) = *(const float4*)(p.ln_b + k);
        }
        bf16x8 fb[U][NB];
        float4 x0[U][MA], x1[U][MA];
#pragma unroll
        for (int s = 0; s < U; s++) {
#pragma unroll
            for (int j = 0; j < NB; j++) fb[s][j] = *(const bf16x8*)(wp[j] + s * 32);
#pragma unroll
            for (int i = 0; i < MA; i++) {
                int m = mb0 + i * 16 + r16; if (m >= p.M) m = p.M - 1;
                const float* xp = p.ln_x + (long long)m * p.K + k0 + q4 * 8 + s * 32;
                x0[s][i] = *(const float4*)xp; x1[s][i] = *(const float4*)(xp + 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // row statistics: this lane's 8 U elements -> the 4 lanes of a row (q4) -> the 4 waves (K quarters) through LDS
#pragma unroll
        for (int i = 0; i < MA; i++) {
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int s = 0; s < U; s++) {
                const float4 a = x0[s][i], b = x1[s][i];
                sm += (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
                sq += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
            }
            sm += __shfl_xor(sm, 16); sq += __shfl_xor(sq, 16);
            sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
            if (q4 == 0) { stat[((wave * MA + i) * 16 + r16) * 2] = sm; stat[((wave * MA + i) * 16 + r16) * 2 + 1] = sq; }
        }
        __syncthreads();
        const float invk = 1.0f / (float)p.K;
#pragma unroll
        for (int i = 0; i < MA; i++) {
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) { sm += stat[((w * MA + i) * 16 + r16) * 2]; sq += stat[((w * MA + i) * 16 + r16) * 2 + 1]; }
            const float mean = sm * invk;
            const float var = fmaxf(sq * invk - mean * mean, 0.f);
            const float rstd = rsqrtf(var + p.ln_eps);
#pragma unroll
            for (int s = 0; s < U; s++) {
                const int kk = k0 + q4 * 8 + s * 32;
                const float4 g0 = *(const float4*)(lnv + kk), g1 = *(const float4*)(lnv + kk + 4);
                const float4 b0 = *(const float4*)(lnv + p.K + kk), b1 = *(const float4*)(lnv + p.K + kk + 4);
                const float4 a = x0[s][i], b = x1[s][i];
                union { uint4 u; bf16x8 f; } t;
                t.u = make_uint4(cvt_pk_bf16((a.x - mean) * rstd * g0.x + b0.x, (a.y - mean) * rstd * g0.y + b0.y),
                                 cvt_pk_bf16((a.z - mean) * rstd * g0.z + b0.z, (a.w - mean) * rstd * g0.w + b0.w),
                                 cvt_pk_bf16((b.x - mean) * rstd * g1.x + b1.x, (b.y - mean) * rstd * g1.y + b1.y),
                                 cvt_pk_bf16((b.z - mean) * rstd * g1.z + b1.z, (b.w - mean) * rstd * g1.w + b1.w));
#pragma unroll
                for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t.f, fb[s][j], acc[i][j], 0, 0, 0);
            }
        }
    } else {
    // U k-steps per iteration: U * (MA + NB) 16-byte loads in flight per lane.  The launch is one or two DEPENDENT round trips to
    // HBM long (a 768-deep K quarter is 6 k-steps of 32: U = 6 fetches a wave's whole operand set at once), so depth = latency.
    for (int k = 0; k < kq; k += U * 32) {
        bf16x8 fa[U][MA], fb[U][NB];
#pragma unroll
        for (int s = 0; s < U; s++) {
#pragma unroll
            for (int j = 0; j < NB; j++) fb[s][j] = *(const bf16x8*)(wp[j] + k + s * 32);
#pragma unroll
            for (int i = 0; i < MA; i++) fa[s][i] = *(const bf16x8*)(ap[i] + k + s * 32);
        }
        // all requests first, then the MFMAs: left alone the scheduler interleaves them to save registers (7 loads, wait, MFMA, 2
        // loads, wait, ...), i.e. a chain of dependent round trips -- the 9 us "load phase" of this kernel
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < U; s++)
#pragma unroll
            for (int i = 0; i < MA; i++)
#pragma unroll
                for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
    }
    }
    // partial tiles -> LDS (D layout 16x16: column = lane & 15, rows (lane >> 4) * 4 + r)
#pragma unroll
    for (int i = 0; i < MA; i++)
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                part[((((wave * MF + (i >> 1)) * NF + (j >> 1)) * 32) + ((i & 1) * 16 + q4 * 4 + r)) * 32 + (j & 1) * 16 + r16] = acc[i][j][r];
    __syncthreads();
    const int ncol0 = blockIdx.x * 32;                    // first OUTPUT column
    // Two phases: every residual / bias value this thread needs is requested first, THEN the results are formed and stored.  The
    // decode step accumulates in place (out == residual: x += ...), so inside a single loop no load may move above the previous
    // iteration's store and the 4 * MF iterations cost one L2 round trip each (~5 of the launch's ~14 us).  A thread reads and
    // writes the same elements, so hoisting its own loads above its own stores is safe whatever aliases.
    constexpr int IT = 4 * MF;
    float acc_[IT], res_[IT]; long long oi_[IT]; bool ok_[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int e = tid + it * 256;
        const int i = e >> 10, row = (e >> 5) & 31, col = e & 31, m = mb0 + i * 32 + row;
        ok_[it] = m < p.M;
        oi_[it] = (long long)m * p.ldo + ncol0 + col;
        float r = 0.f;
        if (ok_[it]) {
            if (p.res_f32) r += p.res_f32[oi_[it]];
            if (p.res_bf16) r += bf2f(p.res_bf16[oi_[it]]);
        }
        res_[it] = r;
    }
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int e = tid + it * 256;
        const int i = e >> 10, row = (e >> 5) & 31, col = e & 31;
        float v[NF];
#pragma unroll
        for (int j = 0; j < NF; j++) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) s += part[((((w * MF + i) * NF + j) * 32) + row) * 32 + col];
            v[j] = s + (p.bias ? p.bias[n0 + j * 32 + col] : 0.f);
        }
        float o = v[0];
        if (NF == 2) o = v[0] * gelu_erf_f(v[1]);
        else if (p.act == ACT_SILU) o = silu_f(o);
        else if (p.act == ACT_QUICKGELU) o = quickgelu_f(o);
        acc_[it] = o + res_[it];
    }
#pragma unroll
    for (int it = 0; it < IT; it++) {
        if (!ok_[it]) continue;
        if (p.out_f32) p.out_f32[oi_[it]] = acc_[it];
        if (p.out_bf16) p.out_bf16[oi_[it]] = f2bf(acc_[it]);
    }
}

bool sgemm_supported(const SgemmParams& p) {
    static const int off = getenv("RDM_NO_SGEMM") ? atoi(getenv("RDM_NO_SGEMM")) : 0;
    if (off || p.M < 1 || p.K % 256 != 0 || (!p.ln_x && p.lda % 8 != 0)) return false;      // any M: rows beyond 128 run as 32-row blocks (grid.y)
    if (p.ln_x && (p.K != 768 || !p.ln_g || !p.ln_b)) return false;                          // LayerNorm-fused A: a K quarter = one batch of 6 k-steps
    if (p.act == ACT_GEGLU) return p.N % 64 == 0;
    return p.N % 32 == 0 && (p.act == ACT_NONE || p.act == ACT_SILU || p.act == ACT_QUICKGELU);
}

template <int MF, int NF, int U, bool LN = false>
static hipError_t launch_one(const SgemmParams& p, int grid, hipStream_t st) {
    const size_t sm = (size_t)4 * MF * NF * 1024 * sizeof(float) + (LN ? (size_t)(2 * p.K + 4 * 2 * MF * 16 * 2) * sizeof(float) : 0);
    static bool attr_dev[RDM_MAX_DEVICES] = {false};
    bool& attr = attr_dev[rdm_cur_device()];
    if (!attr && sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)sgemm_kernel<MF, NF, U, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
        attr = true;
    }
    sgemm_kernel<MF, NF, U, LN><<<dim3(grid, (p.M + 32 * MF - 1) / (32 * MF)), 256, sm, st>>>(p);
    return hipGetLastError();
}
template <int NF>
static hipError_t launch_nf(const SgemmParams& p, int grid, hipStream_t st) {
    // rows per block: all of them (<= 128) when the column strips alone fill the chip, else 32-row blocks (grid.y = ceil(M / 32))
    static const int rowsplit = getenv("RDM_SGEMM_ROWSPLIT") ? atoi(getenv("RDM_SGEMM_ROWSPLIT")) : 1;
    const int mf = ((rowsplit && grid < 256) || p.M > 128) ? 1 : (p.M + 31) / 32;
    const bool deep = ((p.K >> 2) % 192) == 0;            // K quarter is a multiple of 6 k-steps of 32 (K = 768, 1536, 3072 ...)
    if (p.ln_x) return (mf == 1 || p.M <= 32) ? launch_one<1, NF, 6, true>(p, grid, st) : launch_one<2, NF, 6, true>(p, grid, st);
    switch (mf) {
        case 1: return deep ? launch_one<1, NF, 6>(p, grid, st) : launch_one<1, NF, 2>(p, grid, st);
        case 2: return deep ? launch_one<2, NF, 6>(p, grid, st) : launch_one<2, NF, 2>(p, grid, st);
        case 3: return deep ? launch_one<3, NF, 3>(p, grid, st) : launch_one<3, NF, 2>(p, grid, st);
        default: return deep ? launch_one<4, NF, 3>(p, grid, st) : launch_one<4, NF, 2>(p, grid, st);
    }
}

hipError_t launch_sgemm(const SgemmParams& p, hipStream_t st) {
    if (!sgemm_supported(p)) return hipErrorInvalidValue;
    if (p.act == ACT_GEGLU) return launch_nf<2>(p, p.N / 64, st);
    return launch_nf<1>(p, p.N / 32, st);
}

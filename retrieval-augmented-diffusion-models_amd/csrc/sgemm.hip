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

template <int MF, int NF, int U>   // MF: 32-row blocks of M (1..4); NF: 1 plain, 2 GEGLU (x strip + gate strip); U: k-steps of 32 per batch of loads
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
    for (int i = 0; i < MA; i++) { int m = mb0 + i * 16 + r16; if (m >= p.M) m = p.M - 1; ap[i] = p.A + (long long)m * p.lda + k0 + q4 * 8; }
    f32x4 acc[MA][NB];
#pragma unroll
    for (int i = 0; i < MA; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    if (off || p.M < 1 || p.K % 256 != 0 || p.lda % 8 != 0) return false;      // any M: rows beyond 128 run as 32-row blocks (grid.y)
    if (p.act == ACT_GEGLU) return p.N % 64 == 0;
    return p.N % 32 == 0 && (p.act == ACT_NONE || p.act == ACT_SILU || p.act == ACT_QUICKGELU);
}

template <int MF, int NF, int U>
static hipError_t launch_one(const SgemmParams& p, int grid, hipStream_t st) {
    constexpr size_t sm = (size_t)4 * MF * NF * 1024 * sizeof(float);
    static bool attr_dev[RDM_MAX_DEVICES] = {false};
    bool& attr = attr_dev[rdm_cur_device()];
    if (!attr && sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)sgemm_kernel<MF, NF, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
        attr = true;
    }
    sgemm_kernel<MF, NF, U><<<dim3(grid, (p.M + 32 * MF - 1) / (32 * MF)), 256, sm, st>>>(p);
    return hipGetLastError();
}
template <int NF>
static hipError_t launch_nf(const SgemmParams& p, int grid, hipStream_t st) {
    // rows per block: all of them (<= 128) when the column strips alone fill the chip, else 32-row blocks (grid.y = ceil(M / 32))
    static const int rowsplit = getenv("RDM_SGEMM_ROWSPLIT") ? atoi(getenv("RDM_SGEMM_ROWSPLIT")) : 1;
    const int mf = ((rowsplit && grid < 256) || p.M > 128) ? 1 : (p.M + 31) / 32;
    const bool deep = ((p.K >> 2) % 192) == 0;            // K quarter is a multiple of 6 k-steps of 32 (K = 768, 1536, 3072 ...)
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

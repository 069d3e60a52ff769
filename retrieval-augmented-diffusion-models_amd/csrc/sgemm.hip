// Skinny GEMM for decode-sized batches: out[M,N] = act(A[M,K] W[N,K]^T + bias) (+ residual) with M <= 128 rows.
//
// The RARM decode step (rarm.hip) and the UNet's time-embedding MLP are M = B' <= 128 row GEMMs against 768..6144-row weight
// matrices: pure weight streaming.  The tiled kernel (igemm.hip) gives such a launch N/192 blocks (12 of 256 CUs for the
// 2304-wide qkv projection) each walking K serially -- 35 us per launch, ~100 GB/s of weights.  Here a block owns only 32
// output columns (N/32 blocks: 72..512), its four waves split K and hold the whole M x 32 partial tile in MFMA accumulators
// (32x32x16 bf16, fp32), every lane issues all of its 16-byte operand loads for a K quarter up front, and the four partial tiles
// meet in LDS for a fused bias / activation / residual epilogue.  GEGLU: the packed weight rows interleave 32 x-rows with their
// 32 gate rows (packing._geglu_perm), so a block owns a 64-row strip and emits x * gelu(gate) for 32 outputs.
#include "kernels.h"

template <int MF, int NF, int U>   // MF: 32-row fragments of M (1..4); NF: 1 plain, 2 GEGLU (x strip + gate strip); U: k-steps of 16 per batch of loads
__global__ __launch_bounds__(256) void sgemm_kernel(SgemmParams p) {
    extern __shared__ float part[];                       // [4 waves][MF][NF][32 rows][32 cols]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int n0 = blockIdx.x * 32 * NF;                  // first weight row of this block's strip
    const int kq = p.K >> 2, k0 = wave * kq;              // this wave's K quarter
    const bf16_t* wp[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) wp[j] = p.W + (long long)(n0 + j * 32 + frow) * p.K + k0 + fhalf * 8;
    const bf16_t* ap[MF];
#pragma unroll
    for (int i = 0; i < MF; i++) { int m = i * 32 + frow; if (m >= p.M) m = p.M - 1; ap[i] = p.A + (long long)m * p.lda + k0 + fhalf * 8; }
    f32x16 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; i++)
#pragma unroll
        for (int j = 0; j < NF; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // U k-steps per iteration: U * (MF + NF) 16-byte loads in flight per lane.  The launch is one or two DEPENDENT round trips to
    // HBM long (a 768-deep K quarter is 12 k-steps: U = 12 fetches a wave's whole operand set at once), so depth = latency.
    for (int k = 0; k < kq; k += U * 16) {
        bf16x8 fa[U][MF], fb[U][NF];
#pragma unroll
        for (int s = 0; s < U; s++) {
#pragma unroll
            for (int j = 0; j < NF; j++) fb[s][j] = *(const bf16x8*)(wp[j] + k + s * 16);
#pragma unroll
            for (int i = 0; i < MF; i++) fa[s][i] = *(const bf16x8*)(ap[i] + k + s * 16);
        }
#pragma unroll
        for (int s = 0; s < U; s++)
#pragma unroll
            for (int i = 0; i < MF; i++)
#pragma unroll
                for (int j = 0; j < NF; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
    }
    // partial tiles -> LDS (D layout: column = frow, rows (r&3) + 8 (r>>2) + 4 fhalf)
#pragma unroll
    for (int i = 0; i < MF; i++)
#pragma unroll
        for (int j = 0; j < NF; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                part[((((wave * MF + i) * NF + j) * 32) + ((r & 3) + 8 * (r >> 2) + 4 * fhalf)) * 32 + frow] = acc[i][j][r];
    __syncthreads();
    const int ncol0 = blockIdx.x * 32;                    // first OUTPUT column
    for (int e = tid; e < MF * 1024; e += 256) {
        const int i = e >> 10, row = (e >> 5) & 31, col = e & 31, m = i * 32 + row;
        if (m >= p.M) continue;
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
        const long long oi = (long long)m * p.ldo + ncol0 + col;
        if (p.res_f32) o += p.res_f32[oi];
        if (p.res_bf16) o += bf2f(p.res_bf16[oi]);
        if (p.out_f32) p.out_f32[oi] = o;
        if (p.out_bf16) p.out_bf16[oi] = f2bf(o);
    }
}

bool sgemm_supported(const SgemmParams& p) {
    static const int off = getenv("RDM_NO_SGEMM") ? atoi(getenv("RDM_NO_SGEMM")) : 0;
    if (off || p.M < 1 || p.M > 128 || p.K % 256 != 0 || p.lda % 8 != 0) return false;
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
    sgemm_kernel<MF, NF, U><<<grid, 256, sm, st>>>(p);
    return hipGetLastError();
}
template <int NF>
static hipError_t launch_nf(const SgemmParams& p, int grid, hipStream_t st) {
    const int mf = (p.M + 31) / 32;
    const bool deep = ((p.K >> 2) % 192) == 0;            // K quarter is a multiple of 12 k-steps (K = 768, 1536, 3072 ...)
    switch (mf) {
        case 1: return deep ? launch_one<1, NF, 12>(p, grid, st) : launch_one<1, NF, 4>(p, grid, st);
        case 2: return deep ? launch_one<2, NF, 12>(p, grid, st) : launch_one<2, NF, 4>(p, grid, st);
        case 3: return deep ? launch_one<3, NF, 6>(p, grid, st) : launch_one<3, NF, 4>(p, grid, st);
        default: return deep ? launch_one<4, NF, 6>(p, grid, st) : launch_one<4, NF, 4>(p, grid, st);
    }
}

hipError_t launch_sgemm(const SgemmParams& p, hipStream_t st) {
    if (!sgemm_supported(p)) return hipErrorInvalidValue;
    if (p.act == ACT_GEGLU) return launch_nf<2>(p, p.N / 64, st);
    return launch_nf<1>(p, p.N / 32, st);
}

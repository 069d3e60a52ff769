// Mid-size GEMM for one-row-per-sequence operands at 1024+ rows: out[M,N] = act(A[M,K] W[N,K]^T + bias) (+ residual), fp32 or bf16
// residual / output (the RARM decode step's plain projections update the fp32 residual stream in place: attention.py:199-272).
//
// Why a third linear kernel (round 5).  The skinny kernel (sgemm.hip) gives every WAVE its own operands straight from global memory: a
// 64 x 64 output tile asks the L2 for 196 KB at K = 768 -- 32 FLOP per byte -- and holds eight partial copies of its tile in 128 KB of
// LDS, one block per CU.  Up to ~512 rows that is what the launch wants (every CU busy, one dependent round trip); at 2048 rows it is
// 384 such blocks in two rounds, 75 MB through the L2 -> CU fabric for a 2.4 GFLOP product: 33.6 us per launch, four launches per layer
// = 24 % of the decode step at 2048 sequences (profiles/r05_rarm_b2048_kernel_stats.csv).  The big-M kernels (lin4 / igemm) have 128- to
// 256-row tiles of bf16 output: 16 x 4 tiles for [2048 x 768], 64 of 256 CUs (21 us).  Here: 64 x 64 output tiles, operands staged ONCE
// per block in LDS by the asynchronous global -> LDS path (a ring of three stages of 64 x 64 + 64 x 64 bf16 = 48 KB: three blocks per CU;
// ONE barrier per K step: the request for stage k + 2 goes out, stage k is multiplied, stage k + 1 is waited for), four waves
// of 16x16x32 MFMAs over 16 rows x 64 columns each, the residual requested before the K loop, the epilogue straight from the accumulators
// (bias, SiLU / QuickGELU, fp32 / bf16 residual, fp32 / bf16 output: 64-byte row segments per 16 lanes).
// Measured (rocprofv3, tools/lin_bench.py with RDM_MGEMM_ANY): [2048 x 768] x [768 x 768] 15.5 us (skinny 33.6, tiled 21.1), q | k | v
// [2048 x 2304] 23.8, K = 3072 39.6; at 4096 rows 17.8 / 37.5 / 45.7 us.  RARM decode: 2048 sequences 776 -> 850 img/s, 4096: 844 -> 960;
// at 1024 sequences the skinny kernel is still ahead (675 vs 660): used from 1536 rows on.  (Those figures: two stages, two barriers per
// step; the three-stage ring with one barrier: 2048 sequences 840 -> 865 img/s, 4096: 960 -> 978.)  128-row tiles, 128-column tiles with
// the waves 2 x 2 (fewer, bigger blocks) and a fourth stage (fewer blocks per CU) measured slower: what paces a block is its chain of
// barrier -> fragment reads -> MFMAs per K step (77 % of the wave cycles wait, MFMA busy 6.5 %), hidden only by the other blocks of the CU.
// Reading the fragments of stage k + 1 under the MFMAs of stage k (current stage in registers, three buffers, lgkmcnt(0) before the one
// barrier): built and measured, 876 -> 868 img/s at 2048 sequences (profiles/r05e_rarm_mgemm_fragment_prefetch_sweep.log) -- removed.
// LDS rows are 128 bytes (64 k); the 16-byte piece p of row r sits at piece p ^ ((r >> 1) & 7): the 16 lanes of a fragment read
// (consecutive rows, one piece index) land on 16 different 16-byte bank groups.
#include <stdlib.h>

#include "kernels.h"

// NS: stages of the LDS ring.  The K loop is a chain of global -> LDS round trips (a K step of 64 is ~0.3 us of MFMA work): with two
// stages every step waited a full trip for the one request in flight (2048 sequences: 783 -> 808 img/s only); NS - 1 requests in flight
// per block cover it.
// BN = 128: the four waves sit 2 x 2 over the tile (a wave owns BM / 2 rows x 64 columns): 0.75 fragment reads per MFMA instead of 1.25.
template <int BM, int NS, int BN = 64>
__global__ __launch_bounds__(256) void mgemm_kernel(SgemmParams p) {
    constexpr int BK = 64;
    constexpr int WN_W = BN / 64, WM_W = 4 / WN_W;         // waves across the tile's columns / rows
    constexpr int WR = BM / WM_W;                          // rows of a wave
    constexpr int MA = WR / 16;                            // 16-row fragments per wave
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024 / 4, W_INSTR = W_BYTES / 1024 / 4;      // 1 KB per wave instruction, four waves
    extern __shared__ __attribute__((aligned(16))) char smem[];                    // [NS stages][A tile | W tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    // this lane's pieces of a stage: instruction u of this wave fills LDS rows (u * 4 + wave) * 8 .. + 7; lane -> (row, stored piece)
    const int lrow = lane >> 3, sp = lane & 7;
    const bf16_t* ag[A_INSTR]; const bf16_t* wg[W_INSTR];
#pragma unroll
    for (int u = 0; u < A_INSTR; u++) {
        const int row = (u * 4 + wave) * 8 + lrow;
        int m = m0 + row; if (m >= p.M) m = p.M - 1;
        ag[u] = p.A + (long long)m * p.lda + ((sp ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int u = 0; u < W_INSTR; u++) {
        const int row = (u * 4 + wave) * 8 + lrow;
        wg[u] = p.W + (long long)(n0 + row) * p.K + ((sp ^ ((row >> 1) & 7)) << 3);
    }
    auto request = [&](int kt, int buf) {
        char* const s = smem + buf * STAGE;
#pragma unroll
        for (int u = 0; u < A_INSTR; u++) glds16(ag[u] + kt * BK, s + (u * 4 + wave) * 1024);
#pragma unroll
        for (int u = 0; u < W_INSTR; u++) glds16(wg[u] + kt * BK, s + A_BYTES + (u * 4 + wave) * 1024);
    };
    f32x4 acc[MA][4];
#pragma unroll
    for (int i = 0; i < MA; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment read offsets inside a stage (k-step ks adds 4 to the piece index before the swizzle)
    int arow[MA], wrow[4];
    const int wave_m = wave / WN_W, wave_n = wave % WN_W;
#pragma unroll
    for (int i = 0; i < MA; i++) arow[i] = wave_m * WR + i * 16 + r16;
#pragma unroll
    for (int j = 0; j < 4; j++) wrow[j] = wave_n * 64 + j * 16 + r16;
    // The residual values this lane adds in the epilogue are requested BEFORE the K loop (the decode step accumulates in place: out ==
    // residual; a lane reads and writes the same elements and nothing else in the launch touches them): read in the epilogue they are one
    // dependent round trip per output row group -- loads may not move above the previous group's stores -- ~6 us of a ~20 us launch.
    float resv[MA][4][4];
#pragma unroll
    for (int i = 0; i < MA; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + wave_m * WR + i * 16 + q4 * 4 + r;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const long long oi = (long long)(m < p.M ? m : p.M - 1) * p.ldo + n0 + wave_n * 64 + j * 16 + r16;
                float v = 0.f;
                if (p.res_f32) v += p.res_f32[oi];
                if (p.res_bf16) v += bf2f(p.res_bf16[oi]);
                resv[i][r][j] = v;
            }
        }
    const int nk = p.K / BK;
    constexpr int PER = A_INSTR + W_INSTR;                 // requests of one stage per wave
    auto compute = [&](const char* const st) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 fa[MA], fb[4];
#pragma unroll
            for (int i = 0; i < MA; i++) fa[i] = *(const bf16x8*)(st + arow[i] * 128 + (((ks * 4 + q4) ^ ((arow[i] >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < 4; j++) fb[j] = *(const bf16x8*)(st + A_BYTES + wrow[j] * 128 + (((ks * 4 + q4) ^ ((wrow[j] >> 1) & 7)) << 4));
#pragma unroll
            for (int i = 0; i < MA; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (NS == 3) {
        // ONE barrier per K step: stage kt is known to have landed when the step begins (waited for at the end of the step before), the
        // request for stage kt + 2 goes into the buffer of stage kt - 1, which every wave finished reading before that same barrier
        request(0, 0);
        if (nk > 1) { request(1, 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; kt++) {
            if (kt + 2 < nk) request(kt + 2, (kt + 2) % 3);
            compute(smem + (kt % 3) * STAGE);
            if (kt + 1 < nk) {
                if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");      // stage kt + 1 landed (this wave's share)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    } else {
#pragma unroll
    for (int c = 0; c < NS - 1; c++) if (c < nk) request(c, c);
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt % NS;
        __syncthreads();                                   // every wave is done reading stage kt - 1 (= the buffer the next request overwrites)
        if (kt + NS - 1 < nk) request(kt + NS - 1, (kt + NS - 1) % NS);
        // stage kt landed once at most (stages requested after it) x PER of this wave's requests remain in flight
        const int ahead = min(nk - 1 - kt, NS - 1);
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * PER) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // ... and everybody else's
        compute(smem + buf * STAGE);
    }
    }
    // epilogue from the accumulators: D layout 16x16 = column lane & 15, rows (lane >> 4) * 4 + r
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bv[j] = p.bias ? p.bias[n0 + wave_n * 64 + j * 16 + r16] : 0.f;
#pragma unroll
    for (int i = 0; i < MA; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + wave_m * WR + i * 16 + q4 * 4 + r;
            if (m >= p.M) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float o = acc[i][j][r] + bv[j];
                if (p.act == ACT_SILU) o = silu_f(o);
                else if (p.act == ACT_QUICKGELU) o = quickgelu_f(o);
                v[j] = o + resv[i][r][j];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const long long oi = (long long)m * p.ldo + n0 + wave_n * 64 + j * 16 + r16;
                if (p.out_f32) p.out_f32[oi] = v[j];
                if (p.out_bf16) p.out_bf16[oi] = f2bf(v[j]);
            }
        }
    }
}

bool mgemm_supported(const SgemmParams& p) {
    static const int off = getenv("RDM_NO_MGEMM") ? atoi(getenv("RDM_NO_MGEMM")) : 0;
    if (off || p.ln_x || p.M < 64 || p.N % 64 != 0 || p.K % 64 != 0 || p.K < 128 || p.lda % 8 != 0) return false;
    return p.act == ACT_NONE || p.act == ACT_SILU || p.act == ACT_QUICKGELU;
}

template <int BM, int NS, int BN = 64>
static hipError_t mgemm_launch_one(const SgemmParams& p, hipStream_t st) {
    constexpr int sm = NS * (BM * 64 * 2 + BN * 64 * 2);
    static bool attr[RDM_MAX_DEVICES] = {false};
    bool& done = attr[rdm_cur_device()];
    if (!done && sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mgemm_kernel<BM, NS, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        if (e != hipSuccess) return e;
        done = true;
    }
    mgemm_kernel<BM, NS, BN><<<dim3(p.N / BN, (p.M + BM - 1) / BM), 256, sm, st>>>(p);
    return hipGetLastError();
}

hipError_t launch_mgemm(const SgemmParams& p, hipStream_t st) {
    if (!mgemm_supported(p)) return hipErrorInvalidValue;
    static const int bm = getenv("RDM_MGEMM_BM") ? atoi(getenv("RDM_MGEMM_BM")) : 64;
    static const int ns = getenv("RDM_MGEMM_NS") ? atoi(getenv("RDM_MGEMM_NS")) : 3;          // (dev switches: tile rows 64 / 128, ring stages 2 .. 4, tile columns 64 / 128)
    static const int bn = getenv("RDM_MGEMM_BN") ? atoi(getenv("RDM_MGEMM_BN")) : 64;
    if (bn == 128 && p.N % 128 == 0) return bm == 128 ? mgemm_launch_one<128, 2, 128>(p, st) : mgemm_launch_one<64, 2, 128>(p, st);
    if (bm == 128) return ns >= 3 ? mgemm_launch_one<128, 3>(p, st) : mgemm_launch_one<128, 2>(p, st);
    return ns >= 4 ? mgemm_launch_one<64, 4>(p, st) : ns == 3 ? mgemm_launch_one<64, 3>(p, st) : mgemm_launch_one<64, 2>(p, st);
}

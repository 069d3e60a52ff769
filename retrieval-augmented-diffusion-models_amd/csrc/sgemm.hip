// Skinny GEMM for decode-sized batches: out[M,N] = act(A[M,K] W[N,K]^T + bias) (+ residual), M = one row per sequence / sample.
//
// The RARM decode step (rarm.hip) and the UNet's time-embedding MLP are M = B' row GEMMs against 768..16384-row weight matrices: pure
// weight streaming, and a launch lasts as long as ONE block's dependent operand fetch -- what a CU can ingest (~25-35 GB/s) times the
// bytes the block asks for.  The tiled kernel (igemm.hip) gives such a launch N/192 blocks walking K serially (35 us, ~100 GB/s).
// Here a block owns a SMALL output tile -- 16 MA rows x 16 NB columns, chosen per shape so that the launch has >= ~192 blocks and
// each asks for as few bytes as possible (round 4: the N = 768 projections of the decode step ran 48 blocks of 32 x 32 outputs on 48
// of 256 CUs, 393 KB each at K = 3072: 10.9 us; as 192 blocks of 16 x 16 they ask for half) --, its four waves split K and hold the
// partial tile in MFMA accumulators (16x16x32 bf16, fp32), every lane issues all of its 16-byte operand loads for a K quarter up front,
// and the four partial tiles meet in LDS for a fused bias / activation / residual epilogue.  GEGLU: the packed weight rows interleave
// 32 x-rows with their 32 gate rows (packing._geglu_perm); a block owns a whole 64-row strip (NB = 4: 32 outputs) or one half of it
// (NB = 2: x rows [16 h, 16 h + 16) and their gates, 16 outputs) and emits x * gelu(gate).
#include <stdlib.h>

#include "kernels.h"

// LN: the A operand is LayerNorm(ln_x) of an fp32 [M, K] matrix, formed in the kernel (nn.LayerNorm in front of every projection of
// the RARM block, rdm/modules/attention.py:84-86, 199-272): a block needs its rows' whole K anyway (each wave one quarter), so the row
// statistics cost one cross-wave exchange and the separate LayerNorm launch (~5 us of a ~8 us GEMM) disappears.  K / 4 = 32 U only
// (one batch of loads holds a wave's whole K quarter in registers).
// MA: 16-row fragments of M per block (1, 2, 4); NB: 16-column fragments of the weight strip (plain: 1, 2, 4; GEGLU: 2, 4, 8 = per 64-row
// strip its x fragments then their gate fragments; 8 = two strips); U: k-steps of 32 per batch of loads.
// Round 5: the 64 x 64 tiles (MA = 4 with NB = 4, GEGLU NB = 8) are for M = 256 .. 1024 rows -- the RARM decode step at 256 / 512
// sequences per GPU -- where the small tiles re-read the A rows once per 16 / 32 output columns (LN-fused q | k | v at M = 512: 1152
// blocks x 147 KB = 169 MB through an L2 that holds neither operand: 40 us for a 1.8 GFLOP product).
// NW: waves per block = K split (4; 8 for the 64 x 64 tiles: a wave's K share is then ONE batch of loads at K = 768 -- one dependent round trip
// instead of two -- and half as many at K = 3072)
template <int MA, int NB, bool GEGLU, int U, bool LN = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void sgemm_kernel(SgemmParams p) {
    // 16x16x32 MFMAs, not 32x32x16: the operands come straight from global memory in fragment order, and what such a launch pays
    // for is the number of cache LINES its load instructions touch (phase clocks at M = 64, K = 768: 9.1 of the launch's ~12 us are
    // the load phase).  A 32x32x16 fragment is 32 rows x 32 bytes per instruction -- 32 lines for 1 KB --, a 16x16x32 fragment is
    // 16 rows x 64 bytes: half the lines for the same bytes and the same FLOPs per cycle.
    constexpr int RB = 16 * MA, CB = 16 * NB;             // rows / weight rows of the block tile
    constexpr int OC = GEGLU ? CB / 2 : CB;               // output columns of the block
    extern __shared__ float part[];                       // [NW waves][RB rows][CB cols]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int mb0 = blockIdx.y * RB;                      // first of this block's rows
    // weight rows of fragment j; first OUTPUT column of the block
    int wrow[NB], ncol0;
    if constexpr (GEGLU && NB == 2) {                     // half a strip: x rows [16 h, +16) and their gates (32 rows further on)
        const int strip = blockIdx.x >> 1, h = blockIdx.x & 1;
        wrow[0] = strip * 64 + h * 16; wrow[1] = strip * 64 + 32 + h * 16;
        ncol0 = strip * 32 + h * 16;
    } else {
#pragma unroll
        for (int j = 0; j < NB; j++) wrow[j] = blockIdx.x * CB + j * 16;
        ncol0 = blockIdx.x * OC;
    }
    const int kq = p.K / NW, k0 = wave * kq;              // this wave's share of K
    const bf16_t* wp[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) wp[j] = p.W + (long long)(wrow[j] + r16) * p.K + k0 + q4 * 8;
    const bf16_t* ap[MA];
#pragma unroll
    for (int i = 0; i < MA; i++) { int m = mb0 + i * 16 + r16; if (m >= p.M) m = p.M - 1; ap[i] = LN ? nullptr : p.A + (long long)m * p.lda + k0 + q4 * 8; }
    f32x4 acc[MA][NB];
#pragma unroll
    for (int i = 0; i < MA; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The residual values this thread will add in the epilogue are requested FIRST, together with the operands (round 5): as the first
    // thing after the partial-tile exchange they were one more dependent L2 round trip at the end of a launch that is three round trips
    // long.  A thread reads and writes the same elements (the decode step accumulates in place: out == residual), and nothing else in
    // the launch writes them, so the early read sees what the late one saw.
    constexpr int NT = 64 * NW, NOUT = RB * OC, IT = (NOUT + NT - 1) / NT;
    constexpr bool RES_EARLY = NB < 6;                   // (the 64 x 96 tile has no registers to park them in: read in the epilogue there)
    float res_[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int e = tid + it * NT;
        const int row = e / OC, col = e - row * OC, m = mb0 + row;
        float r = 0.f;
        if (RES_EARLY && e < NOUT && m < p.M) {
            const long long oi = (long long)m * p.ldo + ncol0 + col;
            if (p.res_f32) r += p.res_f32[oi];
            if (p.res_bf16) r += bf2f(p.res_bf16[oi]);
        }
        res_[it] = r;
    }
    if constexpr (LN) {
        float* const lnv = part + NW * RB * CB;                 // [gamma K][beta K][NW waves][RB rows][sum, sumsq]
        float* const stat = lnv + 2 * p.K;
        for (int k = tid * 4; k < p.K; k += 256 * NW) {
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
        // row statistics: this lane's 8 U elements -> the 4 lanes of a row (q4) -> the NW waves (K shares) through LDS
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
            for (int w = 0; w < NW; w++) { sm += stat[((w * MA + i) * 16 + r16) * 2]; sq += stat[((w * MA + i) * 16 + r16) * 2 + 1]; }
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
    // FOLD (eight waves, tiles whose eight partial copies would not fit the LDS): waves 4..7 park their tiles, waves 0..3 add them to their
    // own registers, and four copies meet in the epilogue
    constexpr bool FOLD = NW == 8 && (size_t)NW * RB * CB * 4 > 128 * 1024;
    constexpr int NWP = FOLD ? 4 : NW;                     // partial copies the epilogue sums
    if constexpr (FOLD) {
        if (wave >= 4) {
#pragma unroll
            for (int i = 0; i < MA; i++)
#pragma unroll
                for (int j = 0; j < NB; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        part[((wave - 4) * RB + i * 16 + q4 * 4 + r) * CB + j * 16 + r16] = acc[i][j][r];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int i = 0; i < MA; i++)
#pragma unroll
                for (int j = 0; j < NB; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        acc[i][j][r] += part[(wave * RB + i * 16 + q4 * 4 + r) * CB + j * 16 + r16];
        }
        __syncthreads();
    }
    if (!FOLD || wave < 4) {
#pragma unroll
        for (int i = 0; i < MA; i++)
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    part[(wave * RB + i * 16 + q4 * 4 + r) * CB + j * 16 + r16] = acc[i][j][r];
    }
    __syncthreads();
    // Results are formed for ALL of this thread's outputs, THEN stored (the residual came in at the top of the kernel; bias values are
    // L1 / L2 hits).  The decode step accumulates in place (out == residual: x += ...): with loads and stores interleaved in one loop no
    // load may move above the previous iteration's store and the iterations cost one L2 round trip each (~5 of the launch's ~14 us).
    float acc_[IT]; long long oi_[IT]; bool ok_[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int e = tid + it * NT;
        const int row = e / OC, col = e - row * OC, m = mb0 + row;
        ok_[it] = e < NOUT && m < p.M;
        oi_[it] = (long long)m * p.ldo + ncol0 + col;
    }
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int e = tid + it * NT;
        const int row = (e / OC) % RB, col = e % OC;
        // column `col` of the block's outputs = weight-tile column col (x) and, GEGLU, col + OC (its gate)
        float v[GEGLU ? 2 : 1];
#pragma unroll
        for (int j = 0; j < (GEGLU ? 2 : 1); j++) {
            // tile column of output `col` (x) and of its gate: half a strip [16 x | 16 g]; whole strips [32 x | 32 g] (NB = 8: two of them)
            const int cc = !GEGLU ? col : (NB == 2 ? col + j * 16 : (col >> 5) * 64 + (col & 31) + j * 32);
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWP; w++) s += part[(w * RB + row) * CB + cc];
            // weight row behind tile column cc (= wrow[cc >> 4] + (cc & 15), spelled without a run-time array index)
            int wr;
            if constexpr (GEGLU && NB == 2) wr = wrow[0] + (cc < 16 ? cc : cc + 16);
            else wr = wrow[0] + cc;
            v[j] = s + (p.bias ? p.bias[wr] : 0.f);
        }
        float o = v[0];
        if (GEGLU) o = v[0] * gelu_erf_f(v[GEGLU ? 1 : 0]);
        else if (p.act == ACT_SILU) o = silu_f(o);
        else if (p.act == ACT_QUICKGELU) o = quickgelu_f(o);
        if constexpr (RES_EARLY) acc_[it] = o + res_[it];
        else {
            float r = 0.f;
            if (ok_[it]) { if (p.res_f32) r += p.res_f32[oi_[it]]; if (p.res_bf16) r += bf2f(p.res_bf16[oi_[it]]); }
            acc_[it] = o + r;
        }
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
    if (off || p.M < 1 || p.K % 256 != 0 || (!p.ln_x && p.lda % 8 != 0)) return false;      // any M: 16 MA-row blocks on grid.y
    if (p.ln_x && (p.K != 768 || !p.ln_g || !p.ln_b)) return false;                          // LayerNorm-fused A: a K quarter = one batch of 6 k-steps
    if (p.act == ACT_GEGLU) return p.N % 64 == 0;
    return p.N % 32 == 0 && (p.act == ACT_NONE || p.act == ACT_SILU || p.act == ACT_QUICKGELU);
}

template <int MA, int NB, bool GEGLU, int U, bool LN = false, int NW = 4>
static hipError_t launch_one(const SgemmParams& p, hipStream_t st) {
    constexpr int RB = 16 * MA, CB = 16 * NB;
    const bool fold = NW == 8 && (size_t)NW * RB * CB * 4 > 128 * 1024;
    const size_t sm = (size_t)(fold ? 4 : NW) * RB * CB * sizeof(float) + (LN ? (size_t)(2 * p.K + NW * RB * 2) * sizeof(float) : 0);
    static bool attr_dev[RDM_MAX_DEVICES] = {false};
    bool& attr = attr_dev[rdm_cur_device()];
    if (!attr && sm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)sgemm_kernel<MA, NB, GEGLU, U, LN, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int gx = GEGLU ? (NB == 2 ? p.N / 32 : p.N / CB) : p.N / CB;
    sgemm_kernel<MA, NB, GEGLU, U, LN, NW><<<dim3(gx, (p.M + RB - 1) / RB), 64 * NW, sm, st>>>(p);
    return hipGetLastError();
}

// Block tile for a shape: among the compiled (MA, NB) pairs the one with the smallest  rounds x bytes-per-block  (a block's operand
// fetch is what a launch lasts; rounds = how many blocks a CU gets).  Depends on (M, N, K, kind) only.
template <bool GEGLU, bool LN>
static void pick_tile(const SgemmParams& p, int& ma, int& nb) {
    static const int force_ma = getenv("RDM_SGEMM_MA") ? atoi(getenv("RDM_SGEMM_MA")) : 0, force_nb = getenv("RDM_SGEMM_NB") ? atoi(getenv("RDM_SGEMM_NB")) : 0;
    const int mas[3] = {1, 2, 4}, nbs[3] = {GEGLU ? 2 : 1, GEGLU ? 4 : 2, GEGLU ? 8 : 4};
    double best = 1e30; ma = 2; nb = nbs[1];
    for (int a : mas) for (int b : nbs) {
        if (LN && (a > 2 || b == nbs[2])) continue;                   // the LayerNorm-fused variant holds fp32 rows in registers: <= 32 rows
        if (b == nbs[2] && (GEGLU || a != 4 || p.N % (16 * b) != 0)) continue; // the 64-column tiles come with 64 rows only; GEGLU: two strips per
                                                                               // block need three dependent load batches (31.8 vs 28.7 us at M = 512): not used
        const long long blocks = (long long)((p.M + 16 * a - 1) / (16 * a)) * (GEGLU ? (b == 2 ? p.N / 32 : p.N / (16 * b)) : p.N / (16 * b));
        const double bytes = (double)p.K * (16.0 * a * (LN ? 4.0 : 2.0) + 16.0 * b * 2.0);
        const double rounds = (double)((blocks + 255) / 256);
        (void)rounds;
        const double per_cu = blocks > 256 ? (double)blocks / 256.0 : 1.0;                      // blocks a CU's ingest path is shared by
        const double cost = per_cu * (bytes + 8192.0);                                          // + a fixed per-block latency term
        if ((force_ma == 0 || force_ma == a) && (force_nb == 0 || force_nb == b) && cost < best) { best = cost; ma = a; nb = b; }
    }
}

template <bool GEGLU>
static hipError_t launch_nf(const SgemmParams& p, hipStream_t st) {
    const bool deep = ((p.K >> 2) % 192) == 0;            // K quarter is a multiple of 6 k-steps of 32 (K = 768, 1536, 3072 ...)
    constexpr int N1 = GEGLU ? 2 : 1, N2 = GEGLU ? 4 : 2, N3 = GEGLU ? 8 : 4;
    int ma, nb;
    if (p.ln_x) {
        // RDM_SGEMM_LN8_FROM = m (default 0 = off): from m rows on, plain projections take the LayerNorm inside the 64-row tiles on eight
        // waves (a wave's K / 8 = 96 fp32 columns of 64 rows in registers) -- the separate LayerNorm launch is 5.4 us of a decode layer's
        // 150 at 512 sequences, three times per layer.  Measured round 5 (profiles/r05e_rarm_ln8_sweep.log): 512 sequences 564 -> 524
        // img/s, 256 sequences 419 -> 395: every column tile re-reads its 64 rows as fp32 (196 KB instead of 98 KB per block, 64 x 32
        // outputs only: 64 x 64 is 41 registers over an eight-wave block's budget) and repeats the statistics; the two launches saved
        // (10.8 us per layer) come back as + 26 us of GEMM.  Off.
        static const int ln8_from = getenv("RDM_SGEMM_LN8_FROM") ? atoi(getenv("RDM_SGEMM_LN8_FROM")) : 0;
        if constexpr (!GEGLU) {
            if (!p.fixed_split && ln8_from > 0 && p.M >= ln8_from && p.K == 768)
                return launch_one<4, N2, false, 3, true, 8>(p, st);      // (64 x 64 outputs: 41 registers over the budget of an eight-wave block)
        }
        pick_tile<GEGLU, true>(p, ma, nb);
        if (ma == 1) return nb == N1 ? launch_one<1, N1, GEGLU, 6, true>(p, st) : launch_one<1, N2, GEGLU, 6, true>(p, st);
        return nb == N1 ? launch_one<2, N1, GEGLU, 6, true>(p, st) : launch_one<2, N2, GEGLU, 6, true>(p, st);
    }
    if constexpr (!GEGLU) {
        // 64 x 96 outputs for wide projections at 384+ rows (q | k | v, N = 2304, at 512 rows: 8 x 24 = 192 blocks in ONE round of the 256 CUs;
        // as 64 x 64 tiles it is 288 blocks of 128 KB LDS each, one per CU: two rounds).  RDM_SGEMM_N96=0: off
        static const int n96 = getenv("RDM_SGEMM_N96") ? atoi(getenv("RDM_SGEMM_N96")) : 1;
        static const int nw8_off2 = getenv("RDM_SGEMM_NW4") ? atoi(getenv("RDM_SGEMM_NW4")) : 0;
        if (!p.fixed_split && n96 && !nw8_off2 && p.M >= 384 && p.N % 96 == 0 && p.N >= 1536 && p.K == 768) {
            const long long b64 = (long long)((p.M + 63) / 64) * (p.N / 64), b96 = (long long)((p.M + 63) / 64) * (p.N / 96);
            if ((b64 + 255) / 256 > (b96 + 255) / 256) return launch_one<4, 6, false, 3, false, 8>(p, st);
        }
    }
    pick_tile<GEGLU, false>(p, ma, nb);
    if (nb == N3) {       // 64 x 64 outputs: two (GEGLU) / three k-steps per batch of loads keep the operand registers under the budget
        if constexpr (GEGLU) return launch_one<4, N3, true, 2>(p, st);
        else {
            static const int nw8_off = getenv("RDM_SGEMM_NW4") ? atoi(getenv("RDM_SGEMM_NW4")) : 0;
            if (!p.fixed_split && !nw8_off && p.M >= 384 && (p.K >> 3) % 96 == 0) return launch_one<4, N3, false, 3, false, 8>(p, st);      // eight waves: K / 8 in batches of three k-steps
            return ((p.K >> 2) % 96) == 0 ? launch_one<4, N3, false, 3>(p, st) : launch_one<4, N3, false, 2>(p, st);
        }
    }
    if (nb == N1) {
        switch (ma) {
            case 1: return deep ? launch_one<1, N1, GEGLU, 6>(p, st) : launch_one<1, N1, GEGLU, 2>(p, st);
            case 2: return deep ? launch_one<2, N1, GEGLU, 6>(p, st) : launch_one<2, N1, GEGLU, 2>(p, st);
            default: return deep ? launch_one<4, N1, GEGLU, 6>(p, st) : launch_one<4, N1, GEGLU, 2>(p, st);
        }
    }
    switch (ma) {
        case 1: return deep ? launch_one<1, N2, GEGLU, 6>(p, st) : launch_one<1, N2, GEGLU, 2>(p, st);
        case 2: return deep ? launch_one<2, N2, GEGLU, 6>(p, st) : launch_one<2, N2, GEGLU, 2>(p, st);
        default: {
            static const int nw8_off = getenv("RDM_SGEMM_NW4") ? atoi(getenv("RDM_SGEMM_NW4")) : 0;
            if constexpr (!GEGLU) {
                // deep K (3072: a wave's share is 12 k-steps): six per batch of loads = two dependent round trips instead of four (RDM_SGEMM_U6=0: three)
                static const int u6 = getenv("RDM_SGEMM_U6") ? atoi(getenv("RDM_SGEMM_U6")) : 1;
                if (!p.fixed_split && !nw8_off && u6 && p.M >= 384 && (p.K >> 3) % 192 == 0) return launch_one<4, N2, false, 6, false, 8>(p, st);
                if (!p.fixed_split && !nw8_off && p.M >= 384 && (p.K >> 3) % 96 == 0) return launch_one<4, N2, false, 3, false, 8>(p, st);
            }      // (same box: 512 rows 511.6 -> 519.5 img/s, 256 rows 407.8 -> 404.3: from 384 rows on)
            return deep ? launch_one<4, N2, GEGLU, 6>(p, st) : launch_one<4, N2, GEGLU, 2>(p, st);
        }
    }
}

hipError_t launch_sgemm(const SgemmParams& p, hipStream_t st) {
    if (!sgemm_supported(p)) return hipErrorInvalidValue;
    return p.act == ACT_GEGLU ? launch_nf<true>(p, st) : launch_nf<false>(p, st);
}

// Linear / 1x1-conv GEMM  out[M, N] = A[M, K] W[N, K]^T + bias (+ residual)  for bf16 activations on gfx950, ONE WAVE PER SIMD:
// the short-K projections of the UNet's SpatialTransformer (proj_in / to_q / to_k / to_out / proj_out, ldm attention.py:52-72,
// 147-183 as reached from rdm/modules/diffusionmodules/openaimodel.py) and the ResBlock skip_connection 1x1 convs.
//
// Why a second GEMM kernel beside igemm.hip: at K = 384..576 the 8-wave persistent kernel runs its K loop at 35-40 % of the
// matrix pipe (two waves interleave MFMAs on every SIMD, LDS-DMA issue parks them) and at 3 TB/s of the 5+ the same bytes can
// move at.  Here the loop is conv_halo4.hip's, with one tap instead of nine:
//   * 4 waves, one per SIMD, each owning a 128 x 96 tile of the block's 256 rows x 192 columns in 192 literal AGPRs;
//   * weights in MFMA fragment order (lin_w_fragpack_kernel), loaded L2 -> VGPR one whole K-slice (12 fragments) ahead;
//   * activations staged THROUGH REGISTERS: global_load (8 rows x 128 B per instruction, fully coalesced) three slices ahead,
//     ds_write_b128 two steps later into a padded (144-byte rows: conflict-free fragment reads) double-buffered LDS slice;
//     vmcnt retires in order, so the single counted wait of a step -- "my weights have landed" -- also retires the
//     activation pieces of the step before the previous one, and nothing else;
//   * SWAPPED OPERANDS: the MFMA computes D^T (rows = output channels, columns = rows of A), so a lane ends up with 4
//     consecutive channels of ONE row of the output; a v_permlane32_swap per register pair makes that 8 consecutive channels =
//     one 16-byte store, with no LDS transpose, no DPP and no LDS wait in the epilogue; the bias is the accumulators' initial
//     value (written when the previous tile's results are read out), so the epilogue adds nothing but the residual.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"
#include "h4_asm.h"

__device__ unsigned long long g_lin4_prof[4];

// fragment-ordered weight copy: dst[(nb * KQ + kq) * 512 + lane * 8 + e] = W[nb * 32 + (lane & 31)][kq * 16 + (lane >> 5) * 8 + e]
// GEGLU (W rows stored as [32 x | 32 gates] blocks, packing.py _geglu_perm): fragment nb = rows {16 x, then their 16 gates} of block
// nb / 2, half nb % 2 -- after the read-out's lane swap a lane holds 8 x values and the 8 gates of the SAME channels
__host__ __device__ inline int l4_geglu_row(int n) {              // fragment-ordered row -> stored row
    const int nb = n >> 5, fr = n & 31;
    return 64 * (nb >> 1) + 16 * (nb & 1) + (fr < 16 ? fr : 32 + (fr - 16));
}
// gamma given (LayerNorm folded into the GEMM, see lin4_kernel<.., LN>): the copy holds bf16(gamma[k] * W[n][k])
__global__ __launch_bounds__(256) void lin_w_fragpack_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ dst, int N, int K, int ldw, int geglu,
                                                             const float* __restrict__ gamma) {
    const int KQ = K >> 4;
    const long long nvec = (long long)N * K / 8;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const int lane = (int)(v & 63);
        const long long f = v >> 6;
        const int kq = (int)(f % KQ), nb = (int)(f / KQ);
        int n = nb * 32 + (lane & 31);
        const int c = kq * 16 + (lane >> 5) * 8;
        if (geglu) n = l4_geglu_row(n);
        uint4 w = *(const uint4*)(W + (long long)n * ldw + c);
        if (gamma) {
            uint32_t* u = (uint32_t*)&w;
#pragma unroll
            for (int e = 0; e < 4; e++)
                u[e] = pack2bf(gamma[c + 2 * e] * __uint_as_float(u[e] << 16), gamma[c + 2 * e + 1] * __uint_as_float(u[e] & 0xffff0000u));
        }
        *(uint4*)(dst + v * 8) = w;
    }
}
hipError_t launch_lin_w_fragpack(const bf16_t* W, bf16_t* dst, int N, int K, int ldw, int geglu, hipStream_t st, const float* gamma) {
    const long long nvec = (long long)N * K / 8;
    long long g = (nvec + 255) / 256; if (g > 8192) g = 8192;
    lin_w_fragpack_kernel<<<dim3((unsigned)g), 256, 0, st>>>(W, dst, N, K, ldw > 0 ? ldw : K, geglu, gamma);
    return hipGetLastError();
}
// LayerNorm folded into the consuming GEMM:  LN(x) W^T + b = rstd (x (gamma o W)^T - mu s) + b'   per row, with
//   s[n] = sum_k bf16(gamma[k] W[n][k])  (of the ROUNDED products the GEMM multiplies by: the mean term cancels exactly what the
//   accumulator holds),  b'[n] = b[n] + sum_k beta[k] W[n][k].   sb[n] = (s[n], b'[n]), n = STORED row.  One wave per row.
__global__ __launch_bounds__(256) void lin_ln_sb_kernel(const bf16_t* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ bias, float* __restrict__ sb, int N, int K) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float s = 0.f, b = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = bf2f(W[(long long)n * K + k]);
        s += bf2f(f2bf(gamma[k] * w));
        b += beta[k] * w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); b += __shfl_xor(b, o); }
    if (lane == 0) { sb[2 * n] = s; sb[2 * n + 1] = b + (bias ? bias[n] : 0.f); }
}
hipError_t launch_lin_ln_sb(const bf16_t* W, const float* gamma, const float* beta, const float* bias, float* sb, int N, int K, hipStream_t st) {
    lin_ln_sb_kernel<<<dim3((unsigned)((N + 3) / 4)), 256, 0, st>>>(W, gamma, beta, bias, sb, N, K);
    return hipGetLastError();
}


struct ASrc { const char* base; unsigned pstride; unsigned voff; };      // activation pieces of one K-slice: SGPR base, piece stride, lane offset
__device__ __forceinline__ unsigned long long l4_uni64(unsigned long long v) {    // uniform value -> SGPR pair (hipcc does 64-bit multiplies on the VALU)
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ ASrc l4_asrc(bool dead, int row0, int kc, int C0, const char* A0p, const char* A1p, unsigned ld0, unsigned ld1,
                                        const char* zero, unsigned voffA0, unsigned voffA1, unsigned lane16, int wrap1) {
    ASrc s;
    const bool second = kc >= C0;
    const unsigned ld = second ? ld1 : ld0;
    if (second && row0 >= wrap1) row0 -= wrap1;             // second source holds wrap1 rows only (IgemmParams::a1_wrap_rows; at most two copies)
    const char* b = (second ? A1p : A0p) + (unsigned long long)row0 * ld + (unsigned)((second ? kc - C0 : kc) * 2);
    s.base = (const char*)l4_uni64((unsigned long long)(dead ? zero : b));
    s.pstride = (unsigned)__builtin_amdgcn_readfirstlane((int)(dead ? 0u : 32u * ld));
    s.voff = dead ? lane16 : (second ? voffA1 : voffA0);
    return s;
}

// WM: wave arrangement.  2 = 2 x 2 waves, block tile 256 x 192; 1 = 1 x 4 waves, block tile 128 x 384 (N % 384 == 0): every wave
// reads the same 128 rows of A from LDS and no weight fragment is loaded by two waves -- 64 instead of 80 KiB per K-slice through the
// CU's vector-memory return path (which, not the matrix pipe, paces the loop), and A is read from HBM by ONE block when N = 384
// LN: LayerNorm folded in (p.ln_sb).  A holds the RAW rows; the weights are the gamma-scaled copy; every row's sum and sum of squares
// are taken from the activation pieces as they pass through registers on their way to LDS (v_dot2_f32_bf16 against ones / itself,
// 8 lanes per row reduced by DPP once per tile), meet in LDS as (mean, rstd) one step before the tile's read-out, and the read-out
// forms  rstd * (acc - mean * s[n]) + b'[n]  in fp32 (no bias in the accumulators, no residual).  The separate LayerNorm pass (one
// read + one write of the tensor, 2.3 % of a sampling step) is gone; the statistics cost VALU slots beside the MFMAs.
template <int VAR, bool GEGLU, int WM, bool LN>       // VAR: dev-only ablations (RDM_L4_VAR): 1 = activation pieces from the zero page, 2 = no stores
__global__ __launch_bounds__(256, 1) void lin4_kernel(IgemmParams p) {
    constexpr int BM = 128 * WM, BK = 64, FM = 4, FN = 3, WN = FN * 32, BN = (4 / WM) * WN;
    constexpr int L4_ABUF = BM * 144;          // one staged K-slice: BM rows x (128 bytes of channels + 16 of padding)
    constexpr int NPC = 4 * WM;                // activation pieces (8 rows x 128 B) per wave per slice
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [A slice buffer 0][A slice buffer 1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WM == 2 ? wave >> 1 : 0, wn = WM == 2 ? wave & 1 : wave;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int lp = lane >> 3, lc = lane & 7;

    const int K = p.K, nslice = K / BK, KQ = K >> 4;
    const int nbn = p.N / BN, nbm = p.M / BM;
    const int ntiles = nbm * nbn;
    const int G = gridDim.x, xcd = blockIdx.x & 7;
    const int gx = (G - xcd + 7) >> 3;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_end = t_begin + tq + (xcd < tr ? 1 : 0);
    const int tile0 = t_begin + (blockIdx.x >> 3);
    if (tile0 >= t_end) return;
    const int gxq = gx / nbn, gxr = gx - gxq * nbn;

    const char* const A0p = (const char*)p.A0; const char* const A1p = (const char*)p.A1;
    const int C0 = p.C0;
    const float* const biasp = p.bias;
    const int ldo = p.ldo;
    const char* const zero = (const char*)p.zero_page;
    const char* const Wf = (const char*)p.Wfrag;
    const unsigned ld0 = (unsigned)(p.lda > 0 ? p.lda : p.C0) * 2u, ld1 = (unsigned)p.C1 * 2u;     // row strides in bytes
    const int wrap1 = p.a1_wrap_rows > 0 ? p.a1_wrap_rows : 0x7fffffff;

    // ---- per-lane constants
    unsigned vbase[FM];                                         // LDS offset of row `frow` of A fragment i, k-step 0, buffer 0
#pragma unroll
    for (int i = 0; i < FM; i++) vbase[i] = (unsigned)((wm * 128 + i * 32 + frow) * 144 + fhalf * 16);
    unsigned voffj[FN];
#pragma unroll
    for (int j = 0; j < FN; j++) voffj[j] = (unsigned)(lane * 16) + (unsigned)j * (unsigned)(KQ * 1024);
    // activation pieces: piece q of wave w = rows (q * 4 + w) * 8 + lp of the tile, 16-byte chunk lc of the slice
    const unsigned voffA0 = (unsigned)(wave * 8 + lp) * ld0 + (unsigned)(lc * 16);
    const unsigned voffA1 = (unsigned)(wave * 8 + lp) * ld1 + (unsigned)(lc * 16);
    const unsigned lane16 = (unsigned)(lane * 16);
    const unsigned ldsw0 = (unsigned)((wave * 8 + lp) * 144 + lc * 16);           // + q * 4608 (+ L4_ABUF for buffer 1)
    const unsigned ldsw1 = ldsw0 + (unsigned)L4_ABUF;

    bf16_t* const ob = p.out_bf16;
    const bf16_t* const rb = p.res_bf16;

    // ---- cursors over the block's stream of K-slices (tile after tile): compute (C), weights (B: one step ahead), activations
    // (A: three steps ahead).  All uniform.
    struct Cur { int tile, bm, bn, sl; };
    auto cur_adv = [&](Cur& c) __attribute__((always_inline)) {
        c.sl++;
        if (c.sl == nslice) { c.sl = 0; c.tile += gx; c.bn += gxr; c.bm += gxq; if (c.bn >= nbn) { c.bn -= nbn; c.bm++; } }
    };
    auto w_base = [&](const Cur& c) __attribute__((always_inline)) -> const char* {             // fragments of this wave's FN column blocks, slice c.sl, k-step 0
        const int nb0 = c.bn * (BN / 32) + wn * FN;
        const long long off = ((long long)nb0 * KQ + c.sl * 4) * 1024;
        return (const char*)l4_uni64((unsigned long long)(Wf + off));
    };
    auto a_src = [&](const Cur& c) __attribute__((always_inline)) {
        return l4_asrc(VAR == 1 || c.tile >= t_end, c.bm * BM, c.sl * BK, C0, A0p, A1p, ld0, ld1, zero, voffA0, voffA1, lane16, wrap1);
    };

    asm volatile("" ::: H4_ACC_CLOBBERS);                    // the kernel descriptor must allocate the accumulator AGPRs
    bf16x8 fa[2][FM];                                       // activation fragments: k-step parity
    bf16x8 fb[2][4][FN];                                    // weight fragments: [step parity][k-step][column fragment]
    h4_u32x4 hreg[2][NPC];                                    // activation pieces in flight: [step parity][piece]
    // LN: per-lane partial (sum, sum of squares) of the 8 channels x all slices seen so far of piece q's row, current tile
    float lsum[NPC], lsq[NPC];
#pragma unroll
    for (int q = 0; q < NPC; q++) { lsum[q] = 0.f; lsq[q] = 0.f; }
    const unsigned ones_bf = (unsigned)__builtin_amdgcn_readfirstlane(0x3f803f80);
    auto stat_half = [&](int q, const h4_u32x4& h, int half) __attribute__((always_inline)) {
        if constexpr (LN) {
            if (half == 0)
                asm volatile("v_dot2_f32_bf16 %0, %2, %4, %0\n\tv_dot2_f32_bf16 %1, %2, %2, %1\n\tv_dot2_f32_bf16 %0, %3, %4, %0\n\tv_dot2_f32_bf16 %1, %3, %3, %1"
                             : "+v"(lsum[q]), "+v"(lsq[q]) : "v"(h.x), "v"(h.y), "s"(ones_bf));
            else
                asm volatile("v_dot2_f32_bf16 %0, %2, %4, %0\n\tv_dot2_f32_bf16 %1, %2, %2, %1\n\tv_dot2_f32_bf16 %0, %3, %4, %0\n\tv_dot2_f32_bf16 %1, %3, %3, %1"
                             : "+v"(lsum[q]), "+v"(lsq[q]) : "v"(h.z), "v"(h.w), "s"(ones_bf));
        }
    };

    Cur cc{tile0, tile0 / nbn, tile0 % nbn, 0};
    Cur cb = cc, ca = cc;

    // Accumulators start at the bias of their output channel -- written by the MATRIX PIPE: one MFMA with C = 0 per fragment,
    // A = the fragment's 32 channels as (hi, lo) bf16 pairs in k = 0, 1 (hi + lo = the fp32 bias to 2^-17), B = ones in k = 0, 1:
    // 12 issue slots per tile where v_accvgpr_write needed 192.  The packed (hi | lo << 16) words live in LDS (staged once per
    // block): the epilogue must not issue global loads of its own -- vmcnt retires in order, so waiting for one would drain the
    // activation pieces and weight fragments already in flight for the next steps.
    constexpr int L4_STG = 2 * L4_ABUF;                        // per-wave fp32 staging of one fragment row: 32 rows x 400 bytes
    constexpr int L4_BIAS = L4_STG + 4 * 32 * 400;
    const int L4_STAT = L4_BIAS + p.N * 8;                     // LN: (mean, rstd) of the tile's BM rows
    // rowvec (IgemmParams: a per-sample per-column add) in the restricted form lin4_supported admits: at most TWO row groups of
    // rows_per_sample rows (whole tiles each) -- the guided batch's conditional | unconditional halves, whose attn1.to_out bias differs
    // by attn2's output bias (model.hip: the unconditional rows' cross-attention is exactly that bias).  Packed (hi | lo << 16) like the
    // bias, [group][N] behind the bias table; it rides in k = 2, 3 of the start-value MFMA.
    const int L4_RV = L4_BIAS + p.N * 4;
    const bool has_rv = !LN && p.rowvec != nullptr;
    const int rv_split = has_rv ? p.rows_per_sample : 0x7fffffff;          // tiles starting at or beyond this row use group 1
    if constexpr (LN) {                                        // (s, b') per column in FRAGMENT order; the accumulators start at zero
        for (int n = tid; n < p.N; n += 256)
            *(float2*)(smem + L4_BIAS + n * 8) = *(const float2*)(p.ln_sb + 2 * (GEGLU ? l4_geglu_row(n) : n));
    } else {
        for (int n = tid; n < p.N; n += 256) {
            const float bv = biasp ? biasp[GEGLU ? l4_geglu_row(n) : n] : 0.f;
            const uint32_t hi = cvt_pk_bf16(bv, 0.f) & 0xffffu;
            const uint32_t lo = cvt_pk_bf16(bv - __uint_as_float(hi << 16), 0.f) & 0xffffu;
            *(uint32_t*)(smem + L4_BIAS + n * 4) = hi | (lo << 16);
        }
        if (has_rv) {
            for (int i = tid; i < 2 * p.N; i += 256) {
                const int g = i >= p.N ? 1 : 0, n = i - g * p.N;
                const float bv = ((long long)g * p.rows_per_sample < p.M) ? p.rowvec[(long long)g * p.rowvec_ld + (GEGLU ? l4_geglu_row(n) : n)] : 0.f;
                const uint32_t hi = cvt_pk_bf16(bv, 0.f) & 0xffffu;
                const uint32_t lo = cvt_pk_bf16(bv - __uint_as_float(hi << 16), 0.f) & 0xffffu;
                *(uint32_t*)(smem + L4_RV + i * 4) = hi | (lo << 16);
            }
        }
    }
    __syncthreads();
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    union FragU { u32x4_t u; bf16x8 f; };
    auto bias_frags = [&](int bn, int bm, int row, int half, bf16x8 (&bf)[FN], bf16x8& ones) __attribute__((always_inline)) {
        const int rvoff = L4_RV + ((bm * BM >= rv_split) ? p.N * 4 : 0);          // uniform
#pragma unroll
        for (int j = 0; j < FN; j++) {
            uint32_t w = 0u, w2 = 0u;
            if constexpr (!LN) {
                w = *(const uint32_t*)(smem + L4_BIAS + (bn * BN + wn * WN + j * 32 + row) * 4);
                if (has_rv) w2 = *(const uint32_t*)(smem + rvoff + (bn * BN + wn * WN + j * 32 + row) * 4);
            }
            FragU t; t.u = (u32x4_t){half ? 0u : w, half ? 0u : w2, 0u, 0u};
            bf[j] = t.f;
        }
        FragU o; o.u = (u32x4_t){half ? 0u : 0x3f803f80u, half ? 0u : 0x3f803f80u, 0u, 0u};      // ones in k = 0..3 (the rowvec pair is zero without one)
        ones = o.f;
    };
    auto acc_init = [&](int bn, int bm) __attribute__((always_inline)) {
        bf16x8 bf[FN], ones;
        bias_frags(bn, bm, frow, fhalf, bf, ones);
        // the fragments were just written by the VALU: hipcc pads VALU -> MFMA operand hazards for its own MFMAs, not around asm
        asm volatile("s_nop 7" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(ones));
#pragma unroll
        for (int i = 0; i < FM; i++)
#pragma unroll
            for (int j = 0; j < FN; j++) H4_MFMA0(i * FN + j, bf[j], ones);
    };

    // ---- block prologue: slice 0 into LDS buffer 0; slices 1 and 2 in flight in the two piece sets; weights of slice 0
    {
        ASrc s0 = a_src(ca); cur_adv(ca);
#pragma unroll
        for (int q = 0; q < NPC; q++) H4_GLOADB(hreg[0][q], s0.voff, s0.base + (unsigned)q * s0.pstride, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NPC; q++) { H4_LDSWO(ldsw0, hreg[0][q], q * 4608); stat_half(q, hreg[0][q], 0); stat_half(q, hreg[0][q], 1); }
        ASrc s1 = a_src(ca); cur_adv(ca);
#pragma unroll
        for (int q = 0; q < NPC; q++) H4_GLOADB(hreg[0][q], s1.voff, s1.base + (unsigned)q * s1.pstride, 0);
        const char* sb = w_base(cb); cur_adv(cb);
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int j = 0; j < FN; j++) { H4_GLOADB(fb[0][ks][j], voffj[j], sb, ks * 1024); fb[1][ks][j] = fb[0][ks][j]; }
        ASrc s2 = a_src(ca); cur_adv(ca);
#pragma unroll
        for (int q = 0; q < NPC; q++) H4_GLOADB(hreg[1][q], s2.voff, s2.base + (unsigned)q * s2.pstride, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < FM; i++) { H4_LDSR(fa[0][i], vbase[i], 0); fa[1][i] = fa[0][i]; }
    }
    acc_init(cc.bn, cc.bm);

    unsigned long long tprof[2] = {0, 0};
    unsigned long long tp0 = 0, tp1 = 0;
    if (p.dbg & 16) tp0 = __builtin_readcyclecounter();

    // ---- main stream: one step = one 64-deep K-slice = 4 k-steps x 12 MFMAs.  Outstanding vector-memory requests at a step's
    // start, oldest first: [NPC pieces of step X-2][12 weight fragments requested in step X-1][NPC pieces of step X-1]: vmcnt(NPC).
    int epi_stores = 0;
    const char* sbn = Wf; ASrc an{zero, 0u, lane16};
    auto describe = [&]() __attribute__((always_inline)) {
        sbn = (const char*)l4_uni64((unsigned long long)((cb.tile < t_end) ? w_base(cb) : Wf));
        cur_adv(cb);
        an = a_src(ca);
        cur_adv(ca);
    };
    auto step = [&](auto ptag) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        constexpr int RB = P * L4_ABUF;                      // buffer read by this step; the other one is being written
        auto mfma = [&](int ks, int m) {
            const int j = m / FM, i = m % FM;
            H4_MFMA(i * FN + j, fb[P][ks][j], fa[ks & 1][i]);                 // D^T: rows = channels, columns = rows of A
        };
        if (epi_stores == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(NPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(NPC + (GEGLU ? FM * FN : 2 * FM * FN)) : "memory");
        epi_stores = 0;
#pragma unroll
        for (int m = 0; m < FM * FN; m++) {
            mfma(0, m);
            H4_GLOADB(fb[P ^ 1][m / FN][m % FN], voffj[m % FN], sbn, (m / FN) * 1024);
            if (m == 3) {
#pragma unroll
                for (int i = 0; i < FM; i++) H4_LDSR(fa[1][i], vbase[i], RB + 32);
            }
        }
#pragma unroll
        for (int ks = 1; ks < 4; ks++) {
            const int q0 = NPC == 8 ? (ks - 1) * 3 : (ks == 1 ? 0 : ks), nq = NPC == 8 ? (ks < 3 ? 3 : 2) : (ks == 1 ? 2 : 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // pieces requested two steps ago (slice X + 1) -> the buffer this step does not read
#pragma unroll
            for (int q = q0; q < q0 + nq; q++) {
                if (P == 0) H4_LDSWO(ldsw1, hreg[P][q], q * 4608); else H4_LDSWO(ldsw0, hreg[P][q], q * 4608);
            }
            // LN: the pieces just written also feed the row statistics, four dot products behind each of the next MFMAs
            auto stat_at = [&](int m) __attribute__((always_inline)) { if (m < 2 * nq) stat_half(q0 + (m >> 1), hreg[P][q0 + (m >> 1)], m & 1); };
            mfma(ks, 0); stat_at(0); mfma(ks, 1); stat_at(1); mfma(ks, 2); stat_at(2); mfma(ks, 3); stat_at(3);
            if (ks < 3) {
#pragma unroll
                for (int i = 0; i < FM; i++) H4_LDSR(fa[(ks + 1) & 1][i], vbase[i], RB + (ks + 1) * 32);
            } else {
                // every wave has retired its reads of this step's buffer and its writes of the next one
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < FM; i++) H4_LDSR(fa[0][i], vbase[i], (P ^ 1) * L4_ABUF);
            }
            mfma(ks, 4); stat_at(4); mfma(ks, 5); stat_at(5);
            // ... and the pieces of slice X + 3 into the registers just written out
#pragma unroll
            for (int q = q0; q < q0 + nq; q++) {
                H4_GLOADB(hreg[P][q], an.voff, an.base + (unsigned)q * an.pstride, 0);
                mfma(ks, 6 + 2 * (q - q0)); mfma(ks, 7 + 2 * (q - q0));
            }
            if (nq <= 2) { mfma(ks, 10); mfma(ks, 11); }
            if (nq == 1) { mfma(ks, 8); mfma(ks, 9); }
        }
    };

    // after a step: the next slice, or the tile's epilogue.  Returns true when the block has no work left.
    auto advance = [&]() __attribute__((always_inline)) -> bool {
        const int sl_done = cc.sl;
        if (sl_done + 1 < nslice) {
            cc.sl++;
            if constexpr (LN) {
                // the step just done wrote (and summed) the pieces of the tile's LAST slice: statistics complete.  The barrier inside
                // the coming step orders these LDS writes before the read-out; the previous tile's read-out finished at least one
                // barrier ago (nslice >= 2)
                if (cc.sl == nslice - 1) {
#pragma unroll
                    for (int q = 0; q < NPC; q++) {
                        float sv = lsum[q], tv = lsq[q];
                        sv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv), 0xB1, 0xf, 0xf, true));
                        tv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tv), 0xB1, 0xf, 0xf, true));
                        sv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv), 0x4E, 0xf, 0xf, true));
                        tv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tv), 0x4E, 0xf, 0xf, true));
                        sv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv), 0x141, 0xf, 0xf, true));   // row_half_mirror: lane i <-> 7 - i
                        tv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tv), 0x141, 0xf, 0xf, true));
                        const float mean = sv * p.ln_inv_c;
                        const float var = fmaxf(tv * p.ln_inv_c - mean * mean, 0.f);
                        const float rstd = __builtin_amdgcn_rsqf(var + p.ln_eps);
                        if (lc == 0) *(float2*)(smem + L4_STAT + ((q * 4 + wave) * 8 + lp) * 8) = make_float2(mean, rstd);
                        lsum[q] = 0.f; lsq[q] = 0.f;
                    }
                }
            }
            return false;
        }
        const int em0 = cc.bm * BM, en0 = cc.bn * BN;
        Cur nx = cc; cur_adv(nx);
        const bool has_next = nx.tile < t_end;
        // the requests made for the (non-existent) steps after the last one are still in flight towards registers hipcc now
        // considers dead: drain them before anything else may be allocated there
        if (!has_next) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // last MFMA -> v_accvgpr_read: 18 wait states (hipcc does not pad around asm)
        if (p.dbg & 16) { tp1 = __builtin_readcyclecounter(); tprof[0] += tp1 - tp0; }

        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int erow = lane_e & 31, ehalf = lane_e >> 5;
        // Read-out.  Fragment (i, j) register group g (4 registers) = channels j 32 + 8 g + 4 ehalf .. + 3 of row erow, fp32: the
        // groups go from the AGPRs STRAIGHT to a wave-private LDS tile (ds_write_b128 takes accumulator registers: no
        // v_accvgpr_read, no lane exchange), one fragment row (32 rows x 96 channels) at a time, and come back row-contiguous --
        // 8 consecutive channels per lane, 12 lanes per 192-byte row segment -- to be rounded, (+ residual,) packed and stored.
        // (A 32-bytes-per-row store pattern straight from the fragments cost 20 % of the whole GEMM.)
        const unsigned stgw = (unsigned)(L4_STG + wave * (32 * 400) + erow * 400 + ehalf * 16);
        const char* const stgr = smem + L4_STG + wave * (32 * 400);
        bf16x8 nbf[FN], nones;                                 // the next tile's bias fragments
        if constexpr (LN) {                                     // start values are zero: one pinned zero fragment as both operands
            FragU z; z.u = (u32x4_t){0u, 0u, 0u, 0u};
            nones = z.f;
        } else {
            bias_frags(has_next ? nx.bn : cc.bn, has_next ? nx.bm : cc.bm, erow, ehalf, nbf, nones);
        }
        // pin them in registers HERE: left to itself hipcc (re)materialises a fragment's dwords -- v_mov / v_cndmask -- right in front of
        // the asm MFMA that reads them, and a VALU write one or two instructions ahead of a matrix-pipe read is a hazard it does not pad
        // around asm statements: the MFMA multiplies the STALE register (round 5: garbage columns as soon as a second dword of the
        // ones-fragment became non-zero).  The first use is a dozen LDS writes and a wait away; check_mfma_hazard.py audits the assembly.
        if constexpr (LN) { asm volatile("" : "+v"(nones)); nbf[0] = nones; nbf[1] = nones; nbf[2] = nones; }
        else asm volatile("" : "+v"(nbf[0]), "+v"(nbf[1]), "+v"(nbf[2]), "+v"(nones));
        auto stage_row = [&](int i) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < FN; j++) {
                H4_LDSW_ACC(stgw, i * FN + j, 0, (j * 8 + 0) * 16); H4_LDSW_ACC(stgw, i * FN + j, 1, (j * 8 + 2) * 16);
                H4_LDSW_ACC(stgw, i * FN + j, 2, (j * 8 + 4) * 16); H4_LDSW_ACC(stgw, i * FN + j, 3, (j * 8 + 6) * 16);
            }
            // the LDS writes read their accumulator registers when they are EXECUTED, not when they issue (measured: re-initialising
            // the fragments right behind them gave intermittently corrupted tiles): retire them first
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < FN; j++) H4_MFMA0(i * FN + j, nbf[j], nones);
        };
        if constexpr (GEGLU) {
            // fragment channels 0..15 = x, 16..31 = the gates of the same 16 channels; 16 outputs per fragment, 48 per wave:
            // output channel 96 bn + 48 wn + 16 j + k.  Task idx = it 64 + lane: (row, j, half h) -> 8 outputs = one 16-byte store
            const int eno_o = (en0 >> 1) + wn * (WN / 2);
            unsigned voffs[3], lrd[3], sbo[3], srow[3];
#pragma unroll
            for (int it = 0; it < 3; it++) {
                const int idx = it * 64 + lane_e, row = idx / 6, c6 = idx - row * 6;
                voffs[it] = (unsigned)(row * ldo + c6 * 8) * 2u;
                lrd[it] = (unsigned)(row * 400 + ((c6 >> 1) * 32 + (c6 & 1) * 8) * 4);
                sbo[it] = (unsigned)(L4_BIAS + (en0 + wn * WN + (c6 >> 1) * 32 + (c6 & 1) * 8) * 8);      // (s, b') of the 8 x columns; their gates: + 16 columns
                srow[it] = (unsigned)(L4_STAT + (wm * 128 + row) * 8);
            }
            const char* const obase = (const char*)(ob + (long long)(em0 + wm * 128) * ldo + eno_o);
            const unsigned long long rowstep = (unsigned long long)(32 * ldo) * 2ull;
            auto store_row = [&](int i) __attribute__((always_inline)) {
                const char* const op = (const char*)l4_uni64((unsigned long long)(obase + i * rowstep));
#pragma unroll
                for (int it = 0; it < 3; it++) {
                    float4 x0 = *(const float4*)(stgr + lrd[it]), x1 = *(const float4*)(stgr + lrd[it] + 16);
                    float4 g0 = *(const float4*)(stgr + lrd[it] + 64), g1 = *(const float4*)(stgr + lrd[it] + 80);
                    if constexpr (LN) {
                        const float2 st = *(const float2*)(smem + srow[it] + i * 256);
                        const float r = st.y, m2 = -st.x * st.y;
                        const float4* const qx = (const float4*)(smem + sbo[it]); const float4* const qg = (const float4*)(smem + sbo[it] + 128);
                        const float4 a = qx[0], b = qx[1], c = qx[2], d = qx[3];
                        x0.x = fmaf(r, x0.x, fmaf(m2, a.x, a.y)); x0.y = fmaf(r, x0.y, fmaf(m2, a.z, a.w));
                        x0.z = fmaf(r, x0.z, fmaf(m2, b.x, b.y)); x0.w = fmaf(r, x0.w, fmaf(m2, b.z, b.w));
                        x1.x = fmaf(r, x1.x, fmaf(m2, c.x, c.y)); x1.y = fmaf(r, x1.y, fmaf(m2, c.z, c.w));
                        x1.z = fmaf(r, x1.z, fmaf(m2, d.x, d.y)); x1.w = fmaf(r, x1.w, fmaf(m2, d.z, d.w));
                        const float4 e = qg[0], f = qg[1], g = qg[2], h = qg[3];
                        g0.x = fmaf(r, g0.x, fmaf(m2, e.x, e.y)); g0.y = fmaf(r, g0.y, fmaf(m2, e.z, e.w));
                        g0.z = fmaf(r, g0.z, fmaf(m2, f.x, f.y)); g0.w = fmaf(r, g0.w, fmaf(m2, f.z, f.w));
                        g1.x = fmaf(r, g1.x, fmaf(m2, g.x, g.y)); g1.y = fmaf(r, g1.y, fmaf(m2, g.z, g.w));
                        g1.z = fmaf(r, g1.z, fmaf(m2, h.x, h.y)); g1.w = fmaf(r, g1.w, fmaf(m2, h.z, h.w));
                    }
                    const f32x2_t r0 = (f32x2_t){x0.x, x0.y} * gelu_erf_f2((f32x2_t){g0.x, g0.y});
                    const f32x2_t r1 = (f32x2_t){x0.z, x0.w} * gelu_erf_f2((f32x2_t){g0.z, g0.w});
                    const f32x2_t r2 = (f32x2_t){x1.x, x1.y} * gelu_erf_f2((f32x2_t){g1.x, g1.y});
                    const f32x2_t r3 = (f32x2_t){x1.z, x1.w} * gelu_erf_f2((f32x2_t){g1.z, g1.w});
                    const h4_u32x4 dv = {cvt_pk_bf16(r0.x, r0.y), cvt_pk_bf16(r1.x, r1.y), cvt_pk_bf16(r2.x, r2.y), cvt_pk_bf16(r3.x, r3.y)};
                    // NON-TEMPORAL stores (round 5): the hidden tensor is a write-once 400 MB stream per launch at the 32 x 32 level -- as plain
                    // stores it write-allocates through the L2 and evicts the A panels its eight column tiles share (FETCH_SIZE 4-9 x the
                    // algorithmic bytes, verdict round 4 item 7).  Same box: GEGLU 415.6 -> 400.8 us, its consumer ff.net.2 x proj_out 210.4 -> 203.0,
                    // headline + 0.45 .. 0.6 % (profiles/r05c_nt_stores_ab.log).  The same modifier on the other read-outs (plain lin4,
                    // conv_halo4) and on the GroupNorm outputs measured neutral to slightly negative: their consumers re-read them at once.
                    if constexpr (VAR == 2) { asm volatile("" :: "v"(dv), "v"(voffs[it])); } else H4_GSTORES_NT(voffs[it], dv, op);
                }
            };
            stage_row(0); store_row(0);
            stage_row(1); store_row(1);
            stage_row(2); store_row(2);
            stage_row(3); store_row(3);
        } else {
            const int eno = en0 + wn * WN;
            unsigned voffs[6], lrd[6], sbo[3], srow[3];          // LN: tasks it and it + 3 share their 8 columns, 16 rows apart
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int idx = it * 64 + lane_e, row = idx / 12, ch = idx - row * 12;
                voffs[it] = (unsigned)(row * ldo + ch * 8) * 2u;
                lrd[it] = (unsigned)(row * 400 + ch * 32);
                if (it < 3) { sbo[it] = (unsigned)(L4_BIAS + (en0 + wn * WN + ch * 8) * 8); srow[it] = (unsigned)(L4_STAT + (wm * 128 + row) * 8); }
            }
            const char* const obase = (const char*)(ob + (long long)(em0 + wm * 128) * ldo + eno);
            const int er0 = (p.res_wrap_rows > 0 && em0 >= p.res_wrap_rows) ? em0 - p.res_wrap_rows : em0;       // a residual that holds the first half of the rows only
            const char* const rbase = (const char*)(rb + (long long)(er0 + wm * 128) * ldo + eno);
            const unsigned long long rowstep = (unsigned long long)(32 * ldo) * 2ull;
            // residual rows: asm loads (saddr form) with counted waits -- as compiler-visible loads they became flat_load + vmcnt(0).
            // Program order of the vector-memory requests: R0 R1 | S0 (6 stores) R2 | S1 R3 | S2 | S3; the wait in front of row i's
            // adds leaves exactly the younger requests in flight
            h4_u32x4 rr4[2][6];
            auto res_request = [&](int i, h4_u32x4 (&dst)[6]) __attribute__((always_inline)) {
                const char* const rp = (const char*)l4_uni64((unsigned long long)(rbase + i * rowstep));
#pragma unroll
                for (int it = 0; it < 6; it++) H4_GLOADB(dst[it], voffs[it], rp, 0);
            };
            auto store_row = [&](int i) __attribute__((always_inline)) {
                const char* const op = (const char*)l4_uni64((unsigned long long)(obase + i * rowstep));
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    float4 a0 = *(const float4*)(stgr + lrd[it]), a1 = *(const float4*)(stgr + lrd[it] + 16);
                    if constexpr (LN) {
                        const float2 st = *(const float2*)(smem + srow[it % 3] + i * 256 + (it / 3) * 128);
                        const float r = st.y, m2 = -st.x * st.y;
                        const float4* const qs = (const float4*)(smem + sbo[it % 3]);
                        const float4 a = qs[0], b = qs[1], c = qs[2], d = qs[3];
                        a0.x = fmaf(r, a0.x, fmaf(m2, a.x, a.y)); a0.y = fmaf(r, a0.y, fmaf(m2, a.z, a.w));
                        a0.z = fmaf(r, a0.z, fmaf(m2, b.x, b.y)); a0.w = fmaf(r, a0.w, fmaf(m2, b.z, b.w));
                        a1.x = fmaf(r, a1.x, fmaf(m2, c.x, c.y)); a1.y = fmaf(r, a1.y, fmaf(m2, c.z, c.w));
                        a1.z = fmaf(r, a1.z, fmaf(m2, d.x, d.y)); a1.w = fmaf(r, a1.w, fmaf(m2, d.z, d.w));
                    }
                    if (!LN && rb) {                           // added in fp32: one rounding
                        const h4_u32x4 r4 = rr4[i & 1][it];
                        a0.x += __uint_as_float(r4.x << 16); a0.y += __uint_as_float(r4.x & 0xffff0000u);
                        a0.z += __uint_as_float(r4.y << 16); a0.w += __uint_as_float(r4.y & 0xffff0000u);
                        a1.x += __uint_as_float(r4.z << 16); a1.y += __uint_as_float(r4.z & 0xffff0000u);
                        a1.z += __uint_as_float(r4.w << 16); a1.w += __uint_as_float(r4.w & 0xffff0000u);
                    }
                    const h4_u32x4 dv = {cvt_pk_bf16(a0.x, a0.y), cvt_pk_bf16(a0.z, a0.w), cvt_pk_bf16(a1.x, a1.y), cvt_pk_bf16(a1.z, a1.w)};
                    if constexpr (VAR == 2) { asm volatile("" :: "v"(dv), "v"(voffs[it])); } else H4_GSTORES(voffs[it], dv, op);
                }
            };
            constexpr int NST = VAR == 2 ? 0 : 6;              // stores per fragment row
            if (!LN && rb) {
                res_request(0, rr4[0]); res_request(1, rr4[1]);
                stage_row(0);
                asm volatile("s_waitcnt vmcnt(6)" : "+v"(rr4[0][0]), "+v"(rr4[0][1]), "+v"(rr4[0][2]), "+v"(rr4[0][3]), "+v"(rr4[0][4]), "+v"(rr4[0][5]) :: "memory");
                store_row(0);
                res_request(2, rr4[0]);
                stage_row(1);
                asm volatile("s_waitcnt vmcnt(%6)" : "+v"(rr4[1][0]), "+v"(rr4[1][1]), "+v"(rr4[1][2]), "+v"(rr4[1][3]), "+v"(rr4[1][4]), "+v"(rr4[1][5]) : "i"(6 + NST) : "memory");
                store_row(1);
                res_request(3, rr4[1]);
                stage_row(2);
                asm volatile("s_waitcnt vmcnt(%6)" : "+v"(rr4[0][0]), "+v"(rr4[0][1]), "+v"(rr4[0][2]), "+v"(rr4[0][3]), "+v"(rr4[0][4]), "+v"(rr4[0][5]) : "i"(6 + NST) : "memory");
                store_row(2);
                stage_row(3);
                asm volatile("s_waitcnt vmcnt(%6)" : "+v"(rr4[1][0]), "+v"(rr4[1][1]), "+v"(rr4[1][2]), "+v"(rr4[1][3]), "+v"(rr4[1][4]), "+v"(rr4[1][5]) : "i"(NST) : "memory");
                store_row(3);
            } else {
                stage_row(0); store_row(0);
                stage_row(1); store_row(1);
                stage_row(2); store_row(2);
                stage_row(3); store_row(3);
            }
        }
        epi_stores = (VAR == 2) ? 0 : (GEGLU ? FM * FN : 2 * FM * FN);
        if (p.dbg & 16) tprof[1] += __builtin_readcyclecounter() - tp1;
        if (!has_next) return true;
        cc = nx;
        if (p.dbg & 16) tp0 = __builtin_readcyclecounter();
        return false;
    };
    while (true) {
        describe(); step(std::integral_constant<int, 0>{}); if (advance()) break;
        describe(); step(std::integral_constant<int, 1>{}); if (advance()) break;
    }
    if ((p.dbg & 16) && tid == 0) {
        atomicAdd(&g_lin4_prof[0], tprof[0]); atomicAdd(&g_lin4_prof[1], tprof[1]); atomicAdd(&g_lin4_prof[3], 1ull);
    }
}

static int lin4_wm(const IgemmParams& p) {        // wave arrangement: 1 x 4 waves (128 x 384 tiles) whenever N allows, else 2 x 2 (256 x 192)
    static const int force = getenv("RDM_L4_WM") ? atoi(getenv("RDM_L4_WM")) : 0;
    if (p.N % 384 == 0 && p.M % 128 == 0 && force != 2) return 1;
    if (p.N % 192 == 0 && p.M % 256 == 0) return 2;
    return 0;
}
static int lin4_smem_bytes(const IgemmParams& p, int wm) {
    return 2 * (128 * wm * 144) + 4 * 32 * 400 + (p.ln_sb ? p.N * 8 + 128 * wm * 8 : p.N * 4 + (p.rowvec ? 2 * p.N * 4 : 0));
}
bool lin4_supported(const IgemmParams& p, int batch) {
    static const int off = getenv("RDM_NO_LIN4") ? atoi(getenv("RDM_NO_LIN4")) : 0;
    if ((off & 1) || !p.Wfrag || batch != 1) return false;
    const int wm = lin4_wm(p);
    if (!wm || p.K % 64 || p.C0 % 64 || p.C1 % 64 || p.K != p.C0 + p.C1) return false;
    if (p.alpha != 1.0f || p.res_f32 || p.out_f32 || !p.out_bf16) return false;
    // rowvec: at most two row groups of whole tiles (see the kernel), never with the folded LayerNorm
    if (p.rowvec && (p.ln_sb || p.rows_per_sample <= 0 || p.rows_per_sample % (128 * wm) != 0 || 2LL * p.rows_per_sample < p.M || p.rowvec_ld < p.N)) return false;
    const bool geglu = p.act == ACT_GEGLU;
    if (p.act != ACT_NONE && !geglu) return false;
    if (geglu && (p.res_bf16 || (off & 2))) return false;
    const int No = geglu ? p.N / 2 : p.N;
    if (p.N > 8192 || p.ldo % 8 || p.ldo < No || (p.ldw > 0 && p.ldw != p.K)) return false;
    if (p.C1 > 0 && p.lda > 0) return false;
    if (p.res_wrap_rows > 0 && (!p.res_bf16 || p.res_wrap_rows % (128 * wm) != 0 || 2LL * p.res_wrap_rows < p.M)) return false;
    if (p.a1_wrap_rows > 0 && (p.C1 == 0 || p.a1_wrap_rows % (128 * wm) != 0 || 2LL * p.a1_wrap_rows < p.M)) return false;     // whole tiles, at most two copies
    // LayerNorm folded in: one source of >= 2 slices, no residual (the bias rides in ln_sb), everything in 160 KiB of LDS
    if (p.ln_sb && (p.C1 > 0 || p.K < 128 || p.res_bf16 || p.bias || lin4_smem_bytes(p, wm) > 160 * 1024 || !(p.ln_inv_c > 0.f))) return false;
    static const int min_tiles = getenv("RDM_L4_MIN_TILES") ? atoi(getenv("RDM_L4_MIN_TILES")) : 128;
    if (!p.l4_any_tiles && (long long)(p.M / (128 * wm)) * (p.N / (wm == 1 ? 384 : 192)) < min_tiles) return false;     // far fewer tiles than CUs: the 128-row tiles of igemm.hip (160 tiles -- the 8x8-level projections -- still win here: 28 vs 32 us)
    return true;
}

template <int VAR, bool GEGLU, int WM, bool LN>
static hipError_t launch_lin4_cfg(const IgemmParams& p, hipStream_t st) {
    const int smem = lin4_smem_bytes(p, WM);
    static int ncu_dev[RDM_MAX_DEVICES] = {0};
    const int dev = rdm_cur_device();
    if (!ncu_dev[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)lin4_kernel<VAR, GEGLU, WM, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        hipDeviceGetAttribute(&ncu_dev[dev], hipDeviceAttributeMultiprocessorCount, dev);
    }
    const long long ntiles = (long long)(p.M / (128 * WM)) * (p.N / (WM == 1 ? 384 : 192));
    long long g = (ncu_dev[dev] + 7) & ~7;
    if (g > ntiles) g = ntiles;
    static const int prof = getenv("RDM_LIN4_PROF") ? atoi(getenv("RDM_LIN4_PROF")) : 0;
    if (prof) {
        IgemmParams q = p; q.dbg |= 16;
        unsigned long long z[4] = {0, 0, 0, 0}, r[4];
        hipMemcpyToSymbol(HIP_SYMBOL(g_lin4_prof), z, sizeof(z));
        lin4_kernel<VAR, GEGLU, WM, LN><<<dim3((unsigned)g), 256, smem, st>>>(q);
        hipStreamSynchronize(st);
        hipMemcpyFromSymbol(r, HIP_SYMBOL(g_lin4_prof), sizeof(r));
        fprintf(stderr, "[lin4<%d,%d,%d> M=%d N=%d K=%d] blocks=%llu per-block cycles: main %.0f epilogue %.0f (tiles/block %.2f)\n", (int)GEGLU, WM, (int)LN, p.M, p.N, p.K,
                r[3], (double)r[0] / r[3], (double)r[1] / r[3], (double)ntiles / g);
        return hipGetLastError();
    }
    lin4_kernel<VAR, GEGLU, WM, LN><<<dim3((unsigned)g), 256, smem, st>>>(p);
    return hipGetLastError();
}
template <int VAR>
static hipError_t launch_lin4_var(const IgemmParams& p, hipStream_t st) {
    const int wm = lin4_wm(p);
    if (p.act == ACT_GEGLU) return wm == 1 ? launch_lin4_cfg<VAR, true, 1, false>(p, st) : launch_lin4_cfg<VAR, true, 2, false>(p, st);
    return wm == 1 ? launch_lin4_cfg<VAR, false, 1, false>(p, st) : launch_lin4_cfg<VAR, false, 2, false>(p, st);
}
hipError_t launch_lin4(const IgemmParams& p, hipStream_t st) {
    static const int var = getenv("RDM_L4_VAR") ? atoi(getenv("RDM_L4_VAR")) : 0;
    if (p.ln_sb) {                                   // LayerNorm folded in (no dev ablations of these)
        const int wm = lin4_wm(p);
        if (p.act == ACT_GEGLU) return wm == 1 ? launch_lin4_cfg<0, true, 1, true>(p, st) : launch_lin4_cfg<0, true, 2, true>(p, st);
        return wm == 1 ? launch_lin4_cfg<0, false, 1, true>(p, st) : launch_lin4_cfg<0, false, 2, true>(p, st);
    }
    return var == 2 ? launch_lin4_var<2>(p, st) : var == 1 ? launch_lin4_var<1>(p, st) : launch_lin4_var<0>(p, st);
}

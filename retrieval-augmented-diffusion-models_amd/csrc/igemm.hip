// Implicit-GEMM bf16 MFMA kernel for gfx950: one kernel family serves
//   * nn.Linear / 1x1 conv        out[M,N] = A[M,K] . W[N,K]^T
//   * 3x3 conv (pad 1, stride 1|2, optional fused nearest-2x upsample of the input), NHWC,
//     im2col-free: the A-tile loader gathers the (tap, channel-slice) directly from the activation.
// Replaces (reference call sites): conv_nd(2,..,3,padding=1) in ResBlock/Upsample/Downsample
// (rdm/modules/diffusionmodules/openaimodel.py:149,208-210,301), proj_in/proj_out 1x1 convs
// (rdm/modules/attention.py:149-168), to_q/k/v/out Linear (attention.py:30-37), GEGLU FF,
// time_embed / emb_layers linears (openaimodel.py:137-141).
//
// Structure: 128 x BN x 64 block tile, 4 waves (2x2), 32x32x16 bf16 MFMA, fp32 accumulate.
// A and B tiles are staged with 16-byte global_load_lds (LDS-DMA) into a double-buffered LDS ring;
// the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE
// address and again on the ds_read (chunk ^= (row>>1)&7 -> conflict-free ds_read_b128).
// Zero padding of the 3x3 halo (and M/N tails) is done by pointing the lane at a zero page.
// The skip-concat of the UNet decoder (th.cat([h, hs.pop()]), openaimodel.py:365) is never
// materialised: the loader switches source tensor per K-slice (dual-source A).
#include <stdlib.h>

#include "kernels.h"

template <int BM, int BN, bool CONV, bool GEGLU>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmParams p) {
    constexpr int BK = 64;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 32, FN = WN / 32;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int AP = BM / 32, BP = BN / 32;   // loader passes (256 threads cover 32 rows x 8 chunks)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> tile (XCD-aware: consecutive tiles land on the same XCD/L2; N tiles fastest so
    //      the blocks that share an A row-panel run back to back on one L2)
    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const int m0 = bm * BM, n0 = bn * BN;

    const long long zb = blockIdx.z;
    const bf16_t* A0 = p.A0 + zb * p.sA;
    const bf16_t* A1 = p.A1;
    const bf16_t* W = p.W + zb * p.sW;
    const char* zero = (const char*)p.zero_page;

    // ---- loader state
    const int lrow = tid >> 3;          // 0..31
    const int pchunk = tid & 7;         // physical 16B chunk in the LDS row
    int a_m[AP];                        // linear: row index m (or -1); conv: pixel base of sample
    int a_yx[AP];                       // conv: (oy << 16) | ox
    int a_src_chunk[AP];
#pragma unroll
    for (int i = 0; i < AP; i++) {
        const int r = i * 32 + lrow;
        const int m = m0 + r;
        a_src_chunk[i] = pchunk ^ ((r >> 1) & 7);
        if (CONV) {
            if (m < p.M) {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                a_m[i] = b * p.Hin * p.Win;
                a_yx[i] = (oy << 16) | ox;
            } else { a_m[i] = -1; a_yx[i] = 0; }
        } else {
            a_m[i] = (m < p.M) ? m : -1; a_yx[i] = 0;
        }
    }
    long long b_off[BP];
#pragma unroll
    for (int i = 0; i < BP; i++) {
        const int r = i * 32 + lrow;
        const int n = n0 + r;
        const int c = pchunk ^ ((r >> 1) & 7);
        b_off[i] = (n < p.N) ? ((long long)n * p.K + c * 8) : -1;
    }
    const int Cin = p.C0 + p.C1;

    auto stage = [&](int kt, int buf) {
        char* As = smem + buf * STAGE;
        char* Bs = As + A_BYTES;
        // which source / channel offset / tap does this K-slice belong to?
        int kc = kt * BK, dy = 0, dx = 0;
        if (CONV) { const int tap = kc / Cin; kc -= tap * Cin; dy = tap / 3; dx = tap - dy * 3; }
        const bf16_t* src; int ld, cofs;
        if (kc < p.C0) { src = A0; ld = p.C0; cofs = kc; } else { src = A1; ld = p.C1; cofs = kc - p.C0; }
#pragma unroll
        for (int i = 0; i < AP; i++) {
            const void* g = zero;
            if (a_m[i] >= 0) {
                if (CONV) {
                    const int oy = a_yx[i] >> 16, ox = a_yx[i] & 0xffff;
                    int iy, ix; bool ok;
                    if (p.ups) {
                        const int uy = oy + dy - 1, ux = ox + dx - 1;
                        ok = (uy >= 0) & (uy < p.Hout) & (ux >= 0) & (ux < p.Wout);
                        iy = uy >> 1; ix = ux >> 1;
                    } else {
                        iy = oy * p.stride + dy - 1; ix = ox * p.stride + dx - 1;
                        ok = (iy >= 0) & (iy < p.Hin) & (ix >= 0) & (ix < p.Win);
                    }
                    if (ok) g = src + ((long long)(a_m[i] + iy * p.Win + ix) * ld + cofs + a_src_chunk[i] * 8);
                } else {
                    g = src + ((long long)a_m[i] * ld + cofs + a_src_chunk[i] * 8);
                }
            }
            glds16(g, As + (i * 32 + wave * 8) * 128);
        }
#pragma unroll
        for (int i = 0; i < BP; i++) {
            const void* g = (b_off[i] >= 0) ? (const void*)(W + b_off[i] + (long long)kt * BK) : (const void*)zero;
            glds16(g, Bs + (i * 32 + wave * 8) * 128);
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // tile kt landed; everyone is done reading buf cur^1
        if (kt + 1 < nk && !(p.dbg & 2)) stage(kt + 1, cur ^ 1);
        const char* As = smem + ((p.dbg & 2) ? 0 : cur) * STAGE;
        const char* Bs = As + A_BYTES;
        if (p.dbg & 1) continue;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            bf16x8 af[FM], bfr[FN];
            const int chunk = kk * 2 + fhalf;
#pragma unroll
            for (int i = 0; i < FM; i++) {
                const int row = wm * WM + i * 32 + frow;
                af[i] = *(const bf16x8*)(As + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; j++) {
                const int row = wn * WN + j * 32 + frow;
                bfr[j] = *(const bf16x8*)(Bs + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FM; i++)
#pragma unroll
                for (int j = 0; j < FN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue. D layout (32x32): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // (all accumulator indices are compile-time constants: a runtime-indexed acc[][] would be demoted to
    //  scratch memory and re-spilled every K iteration)
    if (p.dbg & 4) { if (acc[0][0][0] == 12345.678f && p.out_bf16) p.out_bf16[0] = 0; return; }
    bf16_t* ob = p.out_bf16 ? p.out_bf16 + zb * p.sO : nullptr;
    float* of = p.out_f32 ? p.out_f32 + zb * p.sO : nullptr;
    const bf16_t* rb = p.res_bf16 ? p.res_bf16 + zb * p.sO : nullptr;
    const float* rf = p.res_f32 ? p.res_f32 + zb * p.sO : nullptr;
    const bool uniform_sample = (p.rows_per_sample % 32) == 0;
#pragma unroll
    for (int i = 0; i < FM; i++) {
        const int mf = m0 + wm * WM + i * 32;                 // first row of this fragment
        const float* rv = nullptr;
        if (p.rowvec && uniform_sample) rv = p.rowvec + (long long)(mf / p.rows_per_sample) * p.rowvec_ld;
#pragma unroll
        for (int j = 0; j < FN; j++) {
            if constexpr (GEGLU) { if (j & 1) continue; }
            const int ncol = n0 + wn * WN + j * 32 + frow;       // column in (permuted) weight space
            const bool col_ok = ncol < p.N;
            const int ocol = GEGLU ? ((n0 + wn * WN + j * 32) >> 1) + frow : ncol;
            float bias = 0.f, gbias = 0.f, rvv = 0.f;
            if (col_ok) {
                if (p.bias) { bias = p.bias[ncol]; if constexpr (GEGLU) gbias = p.bias[ncol + 32]; }
                if (rv) rvv = rv[ncol];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mf + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (col_ok && m < p.M) {
                    float v = acc[i][j][r] * p.alpha + bias + rvv;
                    if (p.rowvec && !uniform_sample) v += p.rowvec[(long long)(m / p.rows_per_sample) * p.rowvec_ld + ncol];
                    if constexpr (GEGLU) {
                        const float g = acc[i][(j + 1) < FN ? (j + 1) : j][r] * p.alpha + gbias;
                        v = v * gelu_erf_f(g);
                    } else {
                        if (p.act == ACT_QUICKGELU) v = quickgelu_f(v);
                        else if (p.act == ACT_SILU) v = silu_f(v);
                    }
                    const long long o = (long long)m * p.ldo + ocol;
                    if (rb) v += bf2f(rb[o]);
                    if (rf) v += rf[o];
                    if (ob) ob[o] = f2bf(v);
                    if (of) of[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, bool CONV, bool GEGLU>
static hipError_t launch_cfg(const IgemmParams& p, int batch, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, CONV, GEGLU>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    dim3 grid(nbm * nbn, 1, batch);
    igemm_kernel<BM, BN, CONV, GEGLU><<<grid, 256, smem, st>>>(p);
    return hipGetLastError();
}

// Host entry: picks the tile. K must be a multiple of 64 (and C0, C1 multiples of 64 for conv / dual).
hipError_t launch_igemm(const IgemmParams& p_in, bool conv, int batch, hipStream_t st) {
    static const int dbg = getenv("RDM_IGEMM_DBG") ? atoi(getenv("RDM_IGEMM_DBG")) : 0;
    IgemmParams p = p_in; p.dbg = dbg;
    if (p.K % 64 != 0 || p.C0 % 64 != 0 || p.C1 % 64 != 0) return hipErrorInvalidValue;
    if (p.act == ACT_GEGLU && (p.N % 64 != 0)) return hipErrorInvalidValue;
    if (p.act == ACT_GEGLU) return conv ? hipErrorInvalidValue : launch_cfg<128, 128, false, true>(p, batch, st);
    const bool wide = (p.N % 192 == 0);
    if (conv) return wide ? launch_cfg<128, 192, true, false>(p, batch, st) : launch_cfg<128, 128, true, false>(p, batch, st);
    return wide ? launch_cfg<128, 192, false, false>(p, batch, st) : launch_cfg<128, 128, false, false>(p, batch, st);
}

// Input-stationary 3x3 convolution (stride 1, pad 1) for NHWC bf16 on gfx950 — the UNet's dominant kernel.
//
// The generic implicit GEMM (igemm.hip) re-stages the A tile from L2 for each of the 9 taps; profiling shows it is
// bound by the L2->LDS fabric (~12 TB/s; operand staging alone takes as long as the MFMA loop).  Here a block owns
// 256 consecutive output pixels (whole image rows) and, per 64-channel slice, stages the (rows+2) x (W+2) HALO of the
// input in LDS ONCE; the 9 taps are then 9 K-slices whose A fragments are read from the halo at a tap-shifted
// position (a wave-uniform LDS offset), so only the 24 KB weight slice streams per K-slice: L2->LDS traffic per
// K-slice drops from 56 KB to ~30 KB (A/9 + B).  Zero padding = halo positions outside the image point at a zero page.
// Same persistent XCD-aware tile walk, hand-pipelined ds_read/MFMA loop and DPP-paired epilogue as igemm.hip.
// Replaces conv_nd(2, C, C', 3, padding=1) inside ResBlock in_layers/out_layers (ldm, via
// rdm/modules/diffusionmodules/openaimodel.py:144-305) — 44 of the 49 3x3 convs per UNet forward.
#include <stdio.h>
#include <stdlib.h>

#include "kernels.h"

// dev-only phase clock (env RDM_HALO_PROF=1): shader cycles spent by wave 0 of every block in [main loop, epilogue]
__device__ unsigned long long g_halo_prof[4];

template <int BN>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_kernel(IgemmParams p) {
    constexpr int BM = 256, BK = 64, NT = 512;
    constexpr int WM = 64, WN = BN / 2, FM = 2, FN = WN / 32;
    constexpr int HALO_BYTES = 400 * 128;                 // <= 400 halo positions x 64 channels bf16
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int BP = BN / 64;                           // weight loader passes (512 threads = 64 rows x 8 chunks)
    constexpr int HPASS = 7;                              // halo loader passes (64 positions each)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo0][halo1][B0][B1]
    char* const halo_base = smem;
    char* const b_base = smem + 2 * HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = wave >> 2;                            // ping-pong group: waves w and w+4 share a SIMD
    const int frow = lane & 31, fhalf = lane >> 5;

    // ---- geometry (uniform)
    const int H = p.Hout, W = p.Wout, HW = H * W;          // the conv's (virtual) input: for ups the 2x nearest-upsampled image
    const int RS = (HW >= BM) ? BM / W : H;               // image rows per sample-part of a tile
    const int NS = BM / (RS * W);                         // samples per tile (1, or 4 at 8x8)
    const int HPW = W + 2, HPS = (RS + 2) * HPW, HP = NS * HPS;
    const int Cin = p.C0 + p.C1, nslice = Cin / BK;

    const int nbn = p.N / BN, nbm = p.M / BM;
    const int ntiles_mn = nbm * nbn;
    // split-K: work item t = part * ntiles_mn + tile; a part covers a contiguous range of 64-channel slices (all 9 taps each)
    const int S = p.ksplit > 1 ? p.ksplit : 1;
    const int ntiles = ntiles_mn * S;
    const int G = gridDim.x, xcd = blockIdx.x & 7;
    const int gx = (G - xcd + 7) >> 3;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_end = t_begin + tq + (xcd < tr ? 1 : 0);
    int tile = t_begin + (blockIdx.x >> 3);
    if (tile >= t_end) return;

    const char* zero = (const char*)p.zero_page;
    const int lrow = tid >> 3, pchunk = tid & 7;
    const int sc8 = (pchunk ^ ((lrow >> 1) & 7)) * 8;     // source chunk for LDS row (pos) == lrow (mod 16): 64 | pass stride

    // ---- per-tile loader state
    int hpix[HPASS];                                      // pixel index of my halo position in each pass, or -1
    long long b_off[BP];                                  // element offset of my weight row + swizzled chunk (64-bit, once per tile:
                                                          // the per-piece address is then one 64-bit add of a wave-uniform term)
    int m0, n0;
    auto setup_mn = [&](int t) {
        t = t % ntiles_mn;
        const int bm = t / nbn, bn = t - bm * nbn;
        m0 = bm * BM; n0 = bn * BN;
#pragma unroll
        for (int i = 0; i < BP; i++) b_off[i] = (long long)(n0 + i * 64 + lrow) * p.K + sc8;
    };
    auto setup_halo = [&](int t) {
        t = t % ntiles_mn;
        const int tm0 = (t / nbn) * BM;
        const int b0 = tm0 / HW, y0 = (tm0 - b0 * HW) / W;
#pragma unroll
        for (int ps = 0; ps < HPASS; ps++) {
            const int hp = ps * 64 + lrow;
            int pix = -1;
            if (hp < HP) {
                const int s = hp / HPS, r = hp - s * HPS;
                const int hy = r / HPW, hx = r - hy * HPW;
                const int y = y0 + hy - 1, x = hx - 1;
                if (y >= 0 && y < H && x >= 0 && x < W) pix = p.ups ? ((b0 + s) * p.Hin + (y >> 1)) * p.Win + (x >> 1) : ((b0 + s) * H + y) * W + x;
            }
            hpix[ps] = pix;
        }
    };
    // one pass of the halo of channel slice `sl` -> halo buffer hb
    auto stage_halo_pass = [&](int ps, int sl, int hb) {
        if (ps * 64 >= HP) return;
        const int kc = sl * BK;
        const bool second = kc >= p.C0;
        const bf16_t* src = second ? p.A1 : p.A0;
        const int ld = second ? p.C1 : p.C0;
        const bf16_t* lane_src = src + ((second ? kc - p.C0 : kc) + sc8);
        const void* g = (hpix[ps] >= 0) ? (const void*)(lane_src + (long long)hpix[ps] * ld) : (const void*)zero;
        // lanes past the last halo position stay masked off: their LDS slot would lie beyond this halo buffer
        if (ps * 64 + lrow < HP) glds16(g, halo_base + hb * HALO_BYTES + (ps * 64 + wave * 8) * 128);
    };
    auto stage_b = [&](int sl, int tap, int bb) {
        const bf16_t* slice_w = p.W + ((long long)tap * Cin + sl * BK);       // wave-uniform
#pragma unroll
        for (int i = 0; i < BP; i++)
            glds16(slice_w + b_off[i], b_base + bb * B_BYTES + (i * 64 + wave * 8) * 128);
    };

    // ---- fragment addressing: pixel row -> halo position of tap (0,0)
    int hp0[FM];
    {
#pragma unroll
        for (int i = 0; i < FM; i++) {
            const int pl = wm * WM + i * 32 + frow;                    // pixel within the tile
            const int s = pl / (RS * W), r = pl - s * RS * W;
            const int ly = r / W, x = r - ly * W;
            hp0[i] = s * HPS + ly * HPW + x;
        }
    }
    const unsigned vb0 = (unsigned)(2 * HALO_BYTES) + (unsigned)(wn * WN * 128) + (unsigned)(frow * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4));

    bf16_t* ob = p.out_bf16;
    const bf16_t* rb = p.res_bf16;
    const bool uniform_sample = (p.rows_per_sample % 32) == 0;
    const int odd = lane & 1;

    // ---- pipeline prologue: halo of slice 0 and weights of (slice 0, tap 0)
    setup_mn(tile); setup_halo(tile);
    {
        const int s0 = ((tile / ntiles_mn) * nslice) / S;
#pragma unroll
        for (int ps = 0; ps < HPASS; ps++) stage_halo_pass(ps, s0, 0);
        stage_b(s0, 0, 0);
    }
    int hcur = 0, bcur = 0;
    int pending_stores = -1;

    unsigned long long tprof[2] = {0, 0};
    while (true) {
        unsigned long long tp0 = 0, tp1 = 0;
        if (p.dbg & 16) tp0 = __builtin_readcyclecounter();
        const int em0 = m0, en0 = n0;
        const int next = tile + gx;
        const bool has_next = next < t_end;
        const int part = tile / ntiles_mn;
        const int s_begin = (part * nslice) / S, s_end = ((part + 1) * nslice) / S;
        const int ns_begin = has_next ? ((next / ntiles_mn) * nslice) / S : 0;
        float pbias[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; i++) {
            const int mf = em0 + wm * WM + i * 32;
            const float* rv = (p.rowvec && uniform_sample) ? p.rowvec + (long long)(mf / p.rows_per_sample) * p.rowvec_ld : nullptr;
#pragma unroll
            for (int j = 0; j < FN; j++) {
                const int ncol = en0 + wn * WN + j * 32 + frow;
                float bv = p.bias ? p.bias[ncol] : 0.f;
                if (rv) bv += rv[ncol];
                pbias[i][j] = bv;
            }
        }
        f32x16 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; i++)
#pragma unroll
            for (int j = 0; j < FN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

        // ---- main loop: two wave groups in ping-pong.  Waves w and w+4 share a SIMD; group 1 (waves 4..7) runs one barrier
        // interval behind group 0, so in every interval one wave of each SIMD is in its MFMA section (12 MFMAs, raised
        // priority) while its partner is in its load section (the 10 ds_read_b128 of its next 12 MFMAs, the LDS-DMA issue for
        // the next tap, the waits).  A phase = half a tap (2 k-steps of 16); intervals per tap: g0 [L0|M0|L1|M1], g1 the same
        // shifted by one.  Waits are counted: weights of tap t+1 are issued in L0 and must have landed before the barrier that
        // ends interval 4t+3 (g0: after M1, g1: after L1); a halo pass (due at the next SLICE) stays in flight across taps.
        // Buffers are re-staged only after a barrier that follows the lgkmcnt(0) retiring their last reads.
#define RDM_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
        if (pending_stores == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (pending_stores == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (pending_stores == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (pending_stores == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // first slice's halo + weights of tap 0 landed (all waves)
        if (grp) __builtin_amdgcn_s_barrier();    // stagger: group 1 starts one interval late
        for (int sl = s_begin; sl < s_end; sl++) {
            const bool last_slice = sl + 1 == s_end;
#pragma unroll 1
            for (int tap = 0; tap < 9; tap++) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const int tapoff = dy * HPW + dx;
                unsigned va[FM];
#pragma unroll
                for (int i = 0; i < FM; i++) {
                    const int hp = hp0[i] + tapoff;
                    va[i] = (unsigned)(hcur * HALO_BYTES) + (unsigned)(hp * 128 + ((fhalf ^ ((hp >> 1) & 7)) << 4));
                }
                const unsigned vb = vb0 + (unsigned)(bcur * B_BYTES);
                const bool tile_end = last_slice && tap == 8;
                // one pass of the next slice's halo this tap -- true only if THIS wave has a lane in it (else nothing is issued
                // and the counted wait below must not leave a weight piece in flight instead)
                // (in the LAST slice of a tile the passes are those of the next tile's first slice: eight taps of lead instead
                // of one, the HBM latency of a tile's first halo no longer shows at the tile start)
                const bool halo_now = (!last_slice || has_next) && tap < HPASS && (tap * 64 + wave * 8 < HP);
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    bf16x8 fa[2][FM], fb[2][FN];
                    // ---- load section
#pragma unroll
                    for (int k2 = 0; k2 < 2; k2++) {
                        const unsigned x = (unsigned)((half * 2 + k2) << 5);
#pragma unroll
                        for (int j = 0; j < FN; j++) { const unsigned b = vb ^ x; RDM_LDS_READ(fb[k2][j], b, j * 4096); }
#pragma unroll
                        for (int i = 0; i < FM; i++) { const unsigned a = va[i] ^ x; RDM_LDS_READ(fa[k2][i], a, 0); }
                    }
                    if (half == 0) {
                        if (last_slice && tap == 0 && has_next) setup_halo(next);        // this tile's halo is all requested
                        if (tap < 8) stage_b(sl, tap + 1, bcur ^ 1);
                        else if (!last_slice) stage_b(sl + 1, 0, bcur ^ 1);
                        else if (has_next) { setup_mn(next); stage_b(ns_begin, 0, bcur ^ 1); }     // cross-tile prefetch
                    } else {
                        if (halo_now) stage_halo_pass(tap, last_slice ? ns_begin : sl + 1, hcur ^ 1);
                        if (grp && !tile_end) {
                            if (halo_now) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    // ---- MFMA section
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                        for (int i = 0; i < FM; i++)
#pragma unroll
                            for (int j = 0; j < FN; j++)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k2][i], fb[k2][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (half == 1 && !grp && !tile_end) {
                        if (halo_now) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                }
                bcur ^= 1;
            }
            hcur ^= 1;
        }
        if (!grp) __builtin_amdgcn_s_barrier();   // re-align the groups
#undef RDM_LDS_READ

        if (p.dbg & 16) { tp1 = __builtin_readcyclecounter(); tprof[0] += tp1 - tp0; }
        // ---- epilogue (same scheme as igemm.hip): per-column bias / time-embedding add in registers, DPP lane-pair swap
        // to packed column pairs, wave-private LDS transpose (staged in the halo buffer just consumed: its successor
        // was prefetched into the other buffer), whole rows leave as 16-byte stores, the residual arrives as 16-byte
        // loads.  Vector-memory instruction count, not bytes, is what an epilogue pays for.
        if (S > 1) {
            // split-K: this part's raw fp32 accumulators go to its plane of the workspace (bias, time-embedding row, residual and
            // the bf16 rounding happen once, in splitk_finish_kernel).  One 32x32 fragment at a time through a wave-private 4 KB
            // LDS tile: 16 ds_write_b32 (rows of 32 floats), back as 4 float4 per lane, out as 16-byte stores.
            __syncthreads();
            float* stg = (float*)(halo_base + (hcur ^ 1) * HALO_BYTES + wave * 4096);
            float* wsp = p.ws + (long long)part * p.M * p.N;
#pragma unroll
            for (int i = 0; i < FM; i++)
#pragma unroll
                for (int j = 0; j < FN; j++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) stg[((r & 3) + 8 * (r >> 2) + 4 * fhalf) * 32 + frow] = acc[i][j][r];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const int idx = it * 64 + lane, row = idx >> 3, ch = idx & 7;
                        const float4 u = *(const float4*)(stg + row * 32 + ch * 4);
                        *(float4*)(wsp + (long long)(em0 + wm * WM + i * 32 + row) * p.N + en0 + wn * WN + j * 32 + ch * 4) = u;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
        } else {
            constexpr int ROWB = WN * 2, CPR = WN / 8, NIT = (32 * CPR) / 64;
            static_assert(8 * 32 * ROWB <= HALO_BYTES && (32 * CPR) % 64 == 0, "epilogue staging geometry");
            __syncthreads();                                            // every wave is done with the last K-slice
            char* stg = halo_base + (hcur ^ 1) * HALO_BYTES + wave * (32 * ROWB);   // hcur already toggled: ^1 = consumed buffer
            const int eno = en0 + wn * WN;
#pragma unroll
            for (int i = 0; i < FM; i++) {
                const int mf = em0 + wm * WM + i * 32;
#pragma unroll
                for (int j = 0; j < FN; j++) {
                    const int ncol = en0 + wn * WN + j * 32 + frow;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        v[r] = acc[i][j][r] + pbias[i][j];    // (a time-embedding row never changes inside a 32-row fragment here:
                                                              //  conv_halo_supported; run-time tests per ELEMENT are expensive, igemm.hip)
                    }
                    char* wp = stg + (4 * fhalf + odd) * ROWB + (j * 32 + frow - odd) * 2;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const int roff = ((2 * t) & 3) + 8 * ((2 * t) >> 2);
                        const float give = odd ? v[2 * t] : v[2 * t + 1];
                        const float got = swap_adjacent_lane(give);
                        const float lo = odd ? got : v[2 * t], hi = odd ? v[2 * t + 1] : got;
                        *(uint32_t*)(wp + roff * ROWB) = cvt_pk_bf16(lo, hi);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int idx = it * 64 + lane, row = idx / CPR, ch = idx - row * CPR;
                    uint4 u = *(const uint4*)(stg + row * ROWB + ch * 16);
                    const long long o = (long long)(mf + row) * p.ldo + eno + ch * 8;
                    if (rb) {
                        const uint4 r4 = *(const uint4*)(rb + o);
                        const uint32_t uu[4] = {u.x, u.y, u.z, u.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
                        uint32_t oo[4];
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            oo[e] = cvt_pk_bf16(__uint_as_float(uu[e] << 16) + __uint_as_float(rr[e] << 16),
                                                __uint_as_float(uu[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u));
                        u = make_uint4(oo[0], oo[1], oo[2], oo[3]);
                    }
                    *(uint4*)(ob + o) = u;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        if (p.dbg & 16) tprof[1] += __builtin_readcyclecounter() - tp1;
        if (!has_next) break;
        tile = next;
        pending_stores = (S > 1) ? FM * FN * 4 : FM * ((32 * (WN / 8)) / 64);      // every tile is full: 16-byte stores per lane
    }
    if ((p.dbg & 16) && tid == 0) {
        atomicAdd(&g_halo_prof[0], tprof[0]); atomicAdd(&g_halo_prof[1], tprof[1]); atomicAdd(&g_halo_prof[3], 1ull);
    }
}

// out = bf16(sum of the ksplit fp32 partial planes + bias + time-embedding row + residual) -- one rounding --, 8 columns per thread
__global__ __launch_bounds__(256) void splitk_finish_kernel(IgemmParams p) {
    const long long nvec = (long long)p.M * (p.N >> 3);
    const long long plane = (long long)p.M * p.N;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const long long m = v / (p.N >> 3); const int n = (int)(v - m * (p.N >> 3)) * 8;
        float a[8];
        { const float4 b0 = p.bias ? *(const float4*)(p.bias + n) : make_float4(0, 0, 0, 0), b1 = p.bias ? *(const float4*)(p.bias + n + 4) : make_float4(0, 0, 0, 0);
          a[0] = b0.x; a[1] = b0.y; a[2] = b0.z; a[3] = b0.w; a[4] = b1.x; a[5] = b1.y; a[6] = b1.z; a[7] = b1.w; }
        for (int s = 0; s < p.ksplit; s++) {                  // fixed order: deterministic
            const float* w = p.ws + s * plane + m * p.N + n;
            const float4 w0 = *(const float4*)w, w1 = *(const float4*)(w + 4);
            a[0] += w0.x; a[1] += w0.y; a[2] += w0.z; a[3] += w0.w; a[4] += w1.x; a[5] += w1.y; a[6] += w1.z; a[7] += w1.w;
        }
        if (p.rowvec) {
            const float* rv = p.rowvec + (m / p.rows_per_sample) * p.rowvec_ld + n;
#pragma unroll
            for (int e = 0; e < 8; e++) a[e] += rv[e];
        }
        if (p.res_bf16) {     // the residual joins in fp32: ONE rounding, like conv_halo4's fused read-out (round 5: the finisher used to round
                              // the conv first -- which of the two a layer got depended on the batch through the K-split decision)
            const uint4 r4 = *(const uint4*)(p.res_bf16 + m * p.ldo + n);
            const uint32_t rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int e = 0; e < 4; e++) { a[2 * e] += __uint_as_float(rr[e] << 16); a[2 * e + 1] += __uint_as_float(rr[e] & 0xffff0000u); }
        }
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = cvt_pk_bf16(a[2 * e], a[2 * e + 1]);
        *(uint4*)(p.out_bf16 + m * p.ldo + n) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

template <int BN>
static hipError_t launch_halo(const IgemmParams& p, hipStream_t st) {
    constexpr int smem = 2 * 400 * 128 + 2 * BN * 128;
    static int ncu_dev[RDM_MAX_DEVICES] = {0};
    const int dev = rdm_cur_device();
    if (!ncu_dev[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_halo_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipDeviceGetAttribute(&ncu_dev[dev], hipDeviceAttributeMultiprocessorCount, dev);
    }
    const int ncu = ncu_dev[dev];
    const long long ntiles = (long long)(p.M / 256) * (p.N / BN) * (p.ksplit > 1 ? p.ksplit : 1);
    long long g = (ncu + 7) & ~7;
    if (g > ntiles) g = ntiles;
    static const int prof = getenv("RDM_HALO_PROF") ? atoi(getenv("RDM_HALO_PROF")) : 0;
    if (prof) {      // dev-only: synchronous launch, prints wave-0 shader cycles per block
        IgemmParams q = p; q.dbg |= 16 | (prof & ~1);
        unsigned long long z[4] = {0, 0, 0, 0}, r[4];
        hipMemcpyToSymbol(HIP_SYMBOL(g_halo_prof), z, sizeof(z));
        conv3x3_halo_kernel<BN><<<dim3((unsigned)g), 512, smem, st>>>(q);
        hipStreamSynchronize(st);
        hipMemcpyFromSymbol(r, HIP_SYMBOL(g_halo_prof), sizeof(r));
        fprintf(stderr, "[halo<%d> M=%d N=%d K=%d] blocks=%llu per-block cycles: main %.0f epilogue %.0f (tiles/block %.2f)\n", BN, p.M, p.N, p.K,
                r[3], (double)r[0] / r[3], (double)r[1] / r[3], (double)ntiles / g);
        return hipGetLastError();
    }
    conv3x3_halo_kernel<BN><<<dim3((unsigned)g), 512, smem, st>>>(p);
    return hipGetLastError();
}

// true if the halo kernel can take this conv (else the caller uses the generic implicit GEMM)
bool conv_halo_supported(const IgemmParams& p) {
    static const int off = getenv("RDM_NO_HALO") ? atoi(getenv("RDM_NO_HALO")) : 0;
    if (off) return false;
    static const int no_ups = getenv("RDM_NO_HALO_UPS") ? atoi(getenv("RDM_NO_HALO_UPS")) : 0;
    const int W = p.Wout, H = p.Hout;
    if (p.stride != 1) return false;
    if (p.ups ? (no_ups || p.Hout != 2 * p.Hin || p.Wout != 2 * p.Win) : (p.Hout != p.Hin || p.Wout != p.Win)) return false;
    if (W < 4 || W > 64 || 256 % W != 0) return false;
    const int HW = H * W;
    if (HW >= 256) { if (HW % 256 != 0 || H % (256 / W) != 0) return false; }
    else if (256 % HW != 0) return false;
    if (p.M % 256 != 0 || (p.N % 192 != 0 && p.N % 128 != 0)) return false;
    if (p.C0 % 64 || p.C1 % 64 || p.alpha != 1.0f || p.act != ACT_NONE || !p.out_bf16 || p.out_f32 || p.res_f32) return false;
    if (p.ldo % 8 || p.K != 9 * (p.C0 + p.C1)) return false;
    if (p.rowvec && p.rows_per_sample % 32 != 0) return false;       // the epilogue folds the per-sample row per 32-row fragment
    const int RS = (HW >= 256) ? 256 / W : H, NS = 256 / (RS * W);
    if (NS * (RS + 2) * (W + 2) > 400) return false;
    if ((long long)p.M * (p.C0 > p.C1 ? p.C0 : p.C1) >= 0x7fffffffLL * 1LL) return false;
    return true;
}

// K-split only pays when the MxN tiles leave most of the chip idle (the 8x8 level: 160 tiles on 256 CUs): S parts per tile
// turn one 62 %-occupied round into ceil(160 S / 256) rounds of 1/S the length
int conv_halo_ksplit(const IgemmParams& p) {
    static const int off = getenv("RDM_NO_SPLITK") ? atoi(getenv("RDM_NO_SPLITK")) : 0;
    if (off || !conv_halo_supported(p)) return 1;
    int dev = 0, ncu = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int bn = (p.N % 192 == 0) ? 192 : 128;
    const long long tiles = (long long)(p.M / 256) * (p.N / bn);
    const int nslice = (p.C0 + p.C1) / 64;
    if (tiles * 4 > (long long)ncu * 3 || p.N % 8 || p.ldo % 8) return 1;
    int best = 1; double bestc = 1.0;                       // cost = rounds / S (+6 % per extra plane for the fp32 round trip)
    for (int S = 2; S <= 3; S++) {
        if (nslice < 2 * S) continue;
        const double c = (double)((tiles * S + ncu - 1) / ncu) / S * (1.0 + 0.06 * (S - 1));
        if (c < bestc - 0.08) { bestc = c; best = S; }
    }
    return best;
}

hipError_t launch_conv_halo(const IgemmParams& p, hipStream_t st) {
    hipError_t e;
    if (conv_halo4_supported(p)) e = launch_conv_halo4(p, st);            // one wave per SIMD (conv_halo4.hip)
    else e = (p.N % 192 == 0) ? launch_halo<192>(p, st) : launch_halo<128>(p, st);
    if (e != hipSuccess || p.ksplit <= 1) return e;
    const long long nvec = (long long)p.M * (p.N >> 3);
    long long g = (nvec + 255) / 256; if (g > 4096) g = 4096;
    splitk_finish_kernel<<<dim3((unsigned)g), 256, 0, st>>>(p);
    return hipGetLastError();
}

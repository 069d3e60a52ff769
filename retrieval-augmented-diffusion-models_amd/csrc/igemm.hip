// Implicit-GEMM bf16 MFMA kernel for gfx950: one kernel family serves
//   * nn.Linear / 1x1 conv        out[M,N] = A[M,K] . W[N,K]^T
//   * 3x3 conv (pad 1, stride 1|2, optional fused nearest-2x upsample of the input), NHWC,
//     im2col-free: the A-tile loader gathers the (tap, channel-slice) directly from the activation.
// Replaces (reference call sites): conv_nd(2,..,3,padding=1) in ResBlock/Upsample/Downsample
// (rdm/modules/diffusionmodules/openaimodel.py:149,208-210,301), proj_in/proj_out 1x1 convs
// (rdm/modules/attention.py:149-168), to_q/k/v/out Linear (attention.py:30-37), GEGLU FF,
// time_embed / emb_layers linears (openaimodel.py:137-141).
//
// Structure: persistent blocks walk 128 x BN output tiles (XCD-aware order); per tile a 64-deep K loop,
// 4 waves (2x2), 32x32x16 bf16 MFMA, fp32 accumulate; the first K-slice of the NEXT tile is prefetched
// before the (register-only) epilogue so its HBM latency is hidden behind the stores.
// A and B tiles are staged with 16-byte global_load_lds (LDS-DMA) into a double-buffered LDS ring;
// the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE
// address and again on the ds_read (chunk ^= (row>>1)&7 -> conflict-free ds_read_b128).
// Zero padding of the 3x3 halo (and M/N tails) is done by pointing the lane at a zero page.
// The skip-concat of the UNet decoder (th.cat([h, hs.pop()]), openaimodel.py:365) is never
// materialised: the loader switches source tensor per K-slice (dual-source A).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"

// softmax over groups of g (1, 2, 4) ADJACENT COLUMNS of an accumulator fragment: adjacent columns are adjacent lanes (D layout:
// col = lane & 31), every r is one row.  Used by the skinny cross-attention scores GEMM (model.hip).
__device__ __forceinline__ void softmax_group16(float (&v)[16], int g) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float mx = v[r];
        if (g >= 2) mx = fmaxf(mx, __shfl_xor(mx, 1));
        if (g >= 4) mx = fmaxf(mx, __shfl_xor(mx, 2));
        const float e = __expf(v[r] - mx);
        float sum = e;
        if (g >= 2) sum += __shfl_xor(sum, 1);
        if (g >= 4) sum += __shfl_xor(sum, 2);
        v[r] = e / sum;
    }
}

// dev-only phase clock (env RDM_IGEMM_PROF=1): shader cycles spent by wave 0 of every block in [K loop, epilogue, wait at tile start]
__device__ unsigned long long g_igemm_prof[5];

// CONV: 0 linear, 1 conv3x3, 2 conv3x3 on a 2x nearest-upsampled input, 3 the same conv by OUTPUT PHASE: output pixel (2y + a, 2x + b) of
// conv3x3(nearest2x(src)) reads only the 2 x 2 source window rows {y - 1 + a, y + a} x columns {x - 1 + b, x + b} -- the taps that land
// on one source pixel are pre-summed (launch_conv_phase_weights: [phase][N][2][2][C]) -- so a phase is a 4-tap conv at SOURCE resolution:
// 16 tap-pixels per source pixel instead of 36 (2.25 x fewer FLOPs).  blockIdx.z = phase 2 a + b; rows m run over the B Hin Win source
// pixels; the read-out scatters row (b, y, x) to output pixel (2y + a, 2x + b).
template <int BM, int BN, int WAVES_M, int CONV, bool GEGLU>
__global__ __launch_bounds__(WAVES_M * 128, 2) void igemm_kernel(IgemmParams p) {
    constexpr int BK = 64;
    constexpr int WAVES_N = 2, NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int FM = WM / 32, FN = WN / 32;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int RPP = NT / 8;                  // rows per loader pass (NT threads cover RPP rows x 8 chunks)
    constexpr int AP = BM / RPP, BP = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0 && WM % 64 == 0 && WN % 32 == 0 && RPP % 16 == 0, "tile/wave geometry");
    static_assert((WN * 128) % 2048 == 0 && (FM + FN == 6 || FM + FN == 5 || FM + FN == 4), "fragment addressing / counted waits");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform (LDS-DMA base, wave tile)
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int frow = lane & 31, fhalf = lane >> 5;

    // ---- persistent tile walk. Blocks b = x (mod 8) run on XCD x (observed dispatch order; speed only):
    //      XCD x owns a contiguous range of tiles (N tiles fastest), its blocks take them round-robin, so
    //      the blocks resident on one L2 at any time share A row-panels and walk W in step.
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int ntiles = nbm * nbn;
    const int G = gridDim.x, xcd = blockIdx.x & 7;
    const int gx = (G - xcd + 7) >> 3;                       // blocks on my XCD
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_end = t_begin + tq + (xcd < tr ? 1 : 0);
    int tile = t_begin + (blockIdx.x >> 3);
    if (tile >= t_end) return;

    const long long zb = blockIdx.z;
    const int ph_a = CONV == 3 ? (int)(zb >> 1) : 0, ph_b = CONV == 3 ? (int)(zb & 1) : 0;
    const bf16_t* A0 = p.A0 + zb * p.sA;
    const bf16_t* A1 = p.A1;
    const bf16_t* W = p.W + zb * p.sW;
    const char* zero = (const char*)p.zero_page;
    bf16_t* ob = p.out_bf16 ? p.out_bf16 + (CONV == 3 ? 0 : zb * p.sO) : nullptr;
    float* of = p.out_f32 ? p.out_f32 + zb * p.sO : nullptr;
    const bf16_t* rb = (p.res_bf16 && !p.res_k) ? p.res_bf16 + zb * p.sO : nullptr;
    const bf16_t* const rk = p.res_k ? p.res_bf16 + zb * p.sO : nullptr;
    const float* rf = p.res_f32 ? p.res_f32 + zb * p.sO : nullptr;
    const int Cin = p.C0 + p.C1;
    // residual as K columns (linear GEMMs, set by launch_igemm): out = [A | R_tile] . [W | I]^T -- the residual tile streams
    // through the deep-prefetched A pipeline instead of being fetched (latency exposed, behind the next tile's operands in
    // the in-order memory queue) by the epilogue; BN/BK extra K-slices of MFMA work per tile, exact in the fp32 accumulator
    const int nkw = p.K / BK;                            // K-slices of the weight matrix proper
    const int nk = nkw + ((CONV == 0 && !GEGLU && p.res_k) ? BN / BK : 0);
    const bf16_t* const eye = (const bf16_t*)((const char*)p.zero_page + RDM_EYE_OFFSET);
    const bool uniform_sample = (p.rows_per_sample % 32) == 0;

    // ---- operand streams.  K-slices are consumed in one continuous stream that runs across tile boundaries
    //      (slice index g = 0,1,2,... of this block).  A (activations: HBM latency) is requested TWO slices ahead
    //      into a 3-slot LDS ring, B (weights: L2-resident) ONE slice ahead into a 2-slot ring; each stream carries
    //      its own tile state, so the prefetch runs through the epilogue of the tile being finished.  vmcnt retires in
    //      order (and counts stores), so every wait is COUNTED: it leaves the A slice requested after the needed B
    //      slice, and the previous tile's epilogue stores, in flight.
    const int lrow = tid >> 3;          // 0..RPP-1
    const int pchunk = tid & 7;         // physical 16B chunk in the LDS row
    const int sc8 = (pchunk ^ ((lrow >> 1) & 7)) * 8;   // source chunk (swizzle on the SOURCE side); RPP % 16 == 0
    constexpr int A_SLOTS = (BM >= 256) ? 3 : 2;        // 128-row tiles keep 2 slots (80 KB -> two blocks per CU)
    char* const a_ring = smem;                          // [A_SLOTS][A_BYTES]
    char* const b_ring = smem + A_SLOTS * A_BYTES;      // [2][B_BYTES]

    // A stream
    int a_tile = tile, a_kt = 0, a_g = 0;
    bool a_live = true;
    int a_pix[AP];                      // linear: row m; conv: centre pixel index (fused-upsample: sample base pixel)
    int a_mask[AP];                     // bit t: tap t readable (linear: 0x1ff or 0); upsample: | oy << 9 | ox << 20
    int k_ci = 0, k_tap = 0;            // channel offset inside the current tap, tap index (wave-uniform)
    auto setup_a = [&](int t) {
        const int m0 = (t / nbn) * BM;
#pragma unroll
        for (int i = 0; i < AP; i++) {
            const int m = m0 + i * RPP + lrow;
            a_pix[i] = 0; a_mask[i] = 0;
            if (m < p.M) {
                if (CONV == 3) {
                    const int hw = p.Hin * p.Win;
                    const int b = m / hw, rem = m - b * hw;
                    const int y = rem / p.Win, x = rem - y * p.Win;
                    a_pix[i] = (b * p.Hin + y) * p.Win + x;
                    int mask = 0;
#pragma unroll
                    for (int tp = 0; tp < 4; tp++) {
                        const int iy = y - 1 + ph_a + (tp >> 1), ix = x - 1 + ph_b + (tp & 1);
                        if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mask |= 1 << tp;
                    }
                    a_mask[i] = mask;
                } else if (CONV) {
                    const int hw = p.Hout * p.Wout;
                    const int b = m / hw, rem = m - b * hw;
                    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                    int mask = 0;
                    if (CONV == 2) {
                        a_pix[i] = b * p.Hin * p.Win;
#pragma unroll
                        for (int tp = 0; tp < 9; tp++) {
                            const int uy = oy + tp / 3 - 1, ux = ox + tp % 3 - 1;
                            if (uy >= 0 && uy < p.Hout && ux >= 0 && ux < p.Wout) mask |= 1 << tp;
                        }
                        mask |= (oy << 9) | (ox << 20);
                    } else {
                        const int cy = oy * p.stride + p.asym, cx = ox * p.stride + p.asym;     // asym: zero padding (0, 1, 0, 1) instead of 1 all round
                        a_pix[i] = (b * p.Hin + cy) * p.Win + cx;
#pragma unroll
                        for (int tp = 0; tp < 9; tp++) {
                            const int iy = cy + tp / 3 - 1, ix = cx + tp % 3 - 1;
                            if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mask |= 1 << tp;
                        }
                    }
                    a_mask[i] = mask;
                } else {
                    a_pix[i] = m; a_mask[i] = 0x1ff;
                }
            }
        }
    };
    // request the A slice at the head of the A stream, then advance the stream
    auto issue_a = [&]() {
        char* As = a_ring + (a_g % A_SLOTS) * A_BYTES;
        if (a_kt == 0) { k_ci = 0; k_tap = 0; }
        const int dy = k_tap / 3, dx = k_tap - dy * 3;
        const bool second = k_ci >= p.C0;
        const bf16_t* src = second ? A1 : A0;
        int ld = second ? p.C1 : p.C0;
        if (CONV == 0 && p.lda) ld = p.lda;
        const bf16_t* lane_src = src + ((second ? k_ci - p.C0 : k_ci) + sc8);
        if (CONV == 0 && !GEGLU && a_kt >= nkw) {       // residual slice: columns of this tile's own output block
            ld = p.ldo;
            lane_src = rk + ((a_tile % nbn) * BN + (a_kt - nkw) * BK + sc8);
        }
        if (CONV == 2) {
#pragma unroll
            for (int i = 0; i < AP; i++) {
                const int oy = (a_mask[i] >> 9) & 0x7ff, ox = (a_mask[i] >> 20) & 0x7ff;
                const int pix = a_pix[i] + ((oy + dy - 1) >> 1) * p.Win + ((ox + dx - 1) >> 1);
                const bool ok = (a_mask[i] >> k_tap) & 1;
                const void* g = ok ? (const void*)(lane_src + (long long)pix * ld) : (const void*)zero;
                glds16(g, As + (i * RPP + wave * 8) * 128);
            }
        } else {
            const int dpix = CONV == 3 ? (ph_a - 1 + (k_tap >> 1)) * p.Win + (ph_b - 1 + (k_tap & 1)) : CONV ? (dy - 1) * p.Win + (dx - 1) : 0;
#pragma unroll
            for (int i = 0; i < AP; i++) {
                const bool ok = (a_mask[i] >> k_tap) & 1;
                const void* g = ok ? (const void*)(lane_src + (long long)(a_pix[i] + dpix) * ld) : (const void*)zero;
                glds16(g, As + (i * RPP + wave * 8) * 128);
            }
        }
        k_ci += BK;
        if (CONV && k_ci >= Cin) { k_ci = 0; k_tap++; }
        a_g++;
        if (++a_kt == nk) {
            a_kt = 0; a_tile += gx;
            a_live = a_tile < t_end;
            if (a_live) setup_a(a_tile);
        }
    };
    // B stream
    int b_tile = tile, b_kt = 0, b_g = 0;
    bool b_live = true;
    int b_n[BP];                        // weight row (or -1)
    auto setup_b = [&](int t) {
        const int n0 = (t % nbn) * BN;
#pragma unroll
        for (int i = 0; i < BP; i++) {
            const int n = n0 + i * RPP + lrow;
            b_n[i] = (n < p.N) ? n : -1;
        }
    };
    auto issue_b = [&]() {
        char* Bs = b_ring + (b_g & 1) * B_BYTES;
        if (CONV == 0 && !GEGLU && b_kt >= nkw) {       // identity slice (rows are tile-local)
            const bf16_t* lane_e = eye + ((b_kt - nkw) * BK + sc8);
#pragma unroll
            for (int i = 0; i < BP; i++) glds16(lane_e + (i * RPP + lrow) * RDM_EYE_N, Bs + (i * RPP + wave * 8) * 128);
        } else {
            const bf16_t* lane_w = W + ((long long)b_kt * BK + sc8);
            const long long ldw = (CONV == 0 && p.ldw) ? p.ldw : p.K;
#pragma unroll
            for (int i = 0; i < BP; i++) {
                const void* g = (b_n[i] >= 0) ? (const void*)(lane_w + (long long)b_n[i] * ldw) : (const void*)zero;
                glds16(g, Bs + (i * RPP + wave * 8) * 128);
            }
        }
        b_g++;
        if (++b_kt == nk) {
            b_kt = 0; b_tile += gx;
            b_live = b_tile < t_end;
            if (b_live) setup_b(b_tile);
        }
    };

    // counted wait: at most n VMEM operations (all younger than the slices needed now) may remain in flight
    auto wait_vm = [&](int n) {
        switch (n) {
            case 52: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
            case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
            case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
            case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
            case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 8:  asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 4:  asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    static_assert(AP == 4, "counted waits assume 4 A requests per thread per slice");

    // ---- prologue: A(0), B(0), A(1)
    setup_a(tile); setup_b(tile);
    issue_a();
    issue_b();
    bool a_ahead = false;               // was an A slice requested AFTER the most recent B slice (3-slot ring only)?
    if (A_SLOTS == 3) { a_ahead = a_live; if (a_live) issue_a(); }
    int c_g = 0;                        // compute stream position
    int pending_stores = 0;             // stores issued after the most recent B request (previous tile's epilogue)

    unsigned long long tprof[4] = {0, 0, 0, 0};
    while (true) {
        unsigned long long tp0 = 0, tp1 = 0;
        if (p.dbg & 16) tp0 = __builtin_readcyclecounter();
        const int em0 = (tile / nbn) * BM, en0 = (tile % nbn) * BN;
        // bias (+ time-embedding row) into registers now; consumed in the epilogue, latency hides under the K loop
        const bool full = (em0 + BM <= p.M) && (en0 + BN <= p.N);
        const int odd = lane & 1;
        float pbias[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; i++) {
            const int mf = em0 + wm * WM + i * 32;
            const float* rv = nullptr;
            if (p.rowvec && uniform_sample && mf < p.M) rv = p.rowvec + (long long)(mf / p.rows_per_sample) * p.rowvec_ld;
#pragma unroll
            for (int j = 0; j < FN; j++) {
                const int ncol = en0 + wn * WN + j * 32 + frow;
                float bv = 0.f;
                if (full || ncol < p.N) {
                    if (p.bias) bv = p.bias[ncol];          // GEGLU: x and gate fragments each get their own (permuted) bias
                    if (rv) bv += rv[ncol];
                }
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

        const int next = tile + gx;
        const bool has_next = next < t_end;

        for (int kt = 0; kt < nk; kt++, c_g++) {
            // needed now: A(c_g) [requested two slices ago] and B(c_g).  Younger than B(c_g): the A slice requested right
            // after it (if any) and, at a tile start, the previous epilogue's stores.
            unsigned long long tw0 = 0;
            if ((p.dbg & 16) && kt == 0) tw0 = __builtin_readcyclecounter();
            wait_vm((a_ahead ? AP : 0) + pending_stores);
            pending_stores = 0;
            __syncthreads();                       // slices landed for every wave; ring slots of slice c_g-1 are free
            if ((p.dbg & 16) && kt == 0) tprof[2] += __builtin_readcyclecounter() - tw0;
            if (!(p.dbg & 2)) {
                if (b_live) issue_b();             // B(c_g+1)
                a_ahead = (A_SLOTS == 3) && a_live;
                if (a_live) issue_a();             // A(c_g+2), or A(c_g+1) with the 2-slot ring
            } else a_ahead = false;
            const char* As = a_ring + (c_g % A_SLOTS) * A_BYTES;
            const char* Bs = b_ring + (c_g & 1) * B_BYTES;
            if (p.dbg & 1) continue;
            // Hand-pipelined LDS->MFMA loop: the fragments of k-step kk+1 are requested (inline-asm ds_read_b128,
            // invisible to hipcc's waitcnt bookkeeping) before the MFMAs of k-step kk; a COUNTED lgkmcnt leaves
            // them in flight behind the matrix pipe.  All fragments of a lane share one swizzle term, so one
            // address VGPR per operand + immediate offsets serve every fragment; k-steps differ by an XOR.
            {
                const unsigned lofs = (unsigned)(frow * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4));
                const unsigned va = (unsigned)(size_t)(As - smem) + (unsigned)(wm * WM * 128) + lofs;
                const unsigned vb = (unsigned)(size_t)(Bs - smem) + (unsigned)(wn * WN * 128) + lofs;
                bf16x8 fa[2][FM], fb[2][FN];
#define RDM_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#pragma unroll
                for (int i = 0; i < FM; i++) RDM_LDS_READ(fa[0][i], va, i * 4096);
#pragma unroll
                for (int j = 0; j < FN; j++) RDM_LDS_READ(fb[0][j], vb, j * 4096);
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int cs = kk & 1, ns = cs ^ 1;
                    if (kk < 3) {
                        const unsigned van = va ^ (unsigned)((kk + 1) << 5), vbn = vb ^ (unsigned)((kk + 1) << 5);
#pragma unroll
                        for (int i = 0; i < FM; i++) RDM_LDS_READ(fa[ns][i], van, i * 4096);
#pragma unroll
                        for (int j = 0; j < FN; j++) RDM_LDS_READ(fb[ns][j], vbn, j * 4096);
                        if constexpr (FM + FN == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                        else if constexpr (FM + FN == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < FM; i++)
#pragma unroll
                        for (int j = 0; j < FN; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cs][i], fb[cs][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef RDM_LDS_READ
            }
        }

        // ---- epilogue (registers only, overlaps the prefetch of the next tile).
        // D layout (32x32): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5). All accumulator indices are
        // compile-time constants (a runtime-indexed acc[][] would be demoted to scratch and re-spilled per K step).
        // VALU diet: neighbouring lanes (adjacent columns) swap one value per register pair through DPP so that
        // every lane owns two adjacent columns of one row -> one v_cvt_pk_bf16_f32 + one 4-byte store per pair;
        // addresses are a per-fragment base pointer plus compile-time multiples of ldo; bounds checks only on
        // tail tiles.
        // Vector-memory STORE INSTRUCTIONS, not bytes, are what the epilogue pays for (~70 cycles per wave-store per CU
        // whatever the width): bf16 outputs therefore go through a wave-private LDS transpose -- packed column pairs
        // are written with conflict-free ds_write_b32, read back as whole rows, and leave as 16-byte-per-lane stores
        // (4x fewer store instructions; the residual arrives as 16-byte loads of the same rows).  The staging area is
        // the ring slots of the K-slice just consumed (free until the next request), so it costs one barrier per tile.
        if (p.dbg & 16) { tp1 = __builtin_readcyclecounter(); tprof[0] += tp1 - tp0; }
        constexpr int WNO = GEGLU ? WN / 2 : WN;                        // output columns per wave
        const int No = GEGLU ? p.N / 2 : p.N;
        // (a plain bf16 residual -- not the K-column form -- takes the direct epilogue, which adds it in fp32 BEFORE the one rounding: the LDS
        //  transpose below packs to bf16 first, and the rest of the library (conv_halo4 / lin4 read-outs, the split-K finisher, the
        //  residual-as-K-columns GEMMs) rounds once.  Reached by the generic-path convs of small batches / deterministic mode only.)
        const bool lds_epi = ob && !of && !rf && !rb && (No % 8 == 0) && (p.ldo % 8 == 0) && !(p.dbg & 4);
        // PLAIN = no activation, alpha 1, no per-row time-embedding lookup: every UNet projection except the GEGLU one.  The flag is
        // a compile-time parameter of the body: as run-time tests inside the unrolled element loops hipcc kept a compare + branch
        // (+ hazard nops) per ELEMENT, and the epilogue took twice as long as the halo kernel's for the same tile.
        auto lds_epilogue = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;            // 0 general, 1 plain, 2 plain + softmax over column groups
            constexpr bool PLAIN = MODE != 0;
            constexpr int ROWB = WNO * 2, CPR = WNO / 8, NIT = (32 * CPR) / 64;
            static_assert((32 * CPR) % 64 == 0 && 32 * ROWB * 4 <= B_BYTES, "epilogue staging geometry");
            __syncthreads();                                            // every wave is done reading slice c_g-1
            if (p.dbg & 16) tprof[3] += __builtin_readcyclecounter() - tp1;
            char* stg = (wave < 4) ? (char*)b_ring + ((c_g - 1) & 1) * B_BYTES + wave * (32 * ROWB)
                                   : (char*)a_ring + ((c_g - 1) % A_SLOTS) * A_BYTES + (wave - 4) * (32 * ROWB);
            const int eno = (GEGLU ? en0 / 2 : en0) + wn * WNO;         // first output column of this wave
#pragma unroll
            for (int i = 0; i < FM; i++) {
                const int mf = em0 + wm * WM + i * 32;
#pragma unroll
                for (int j = 0; j < FN; j++) {
                    if constexpr (GEGLU) { if (j & 1) continue; }
                    const float bias = pbias[i][j];
                    float gbias = 0.f;
                    if constexpr (GEGLU) gbias = pbias[i][(j + 1) < FN ? (j + 1) : j];
                    const int ncol = en0 + wn * WN + j * 32 + frow;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float x;
                        if constexpr (PLAIN) x = acc[i][j][r] + bias;
                        else {
                            x = acc[i][j][r] * p.alpha + bias;
                            if (p.rowvec && !uniform_sample) {
                                const int m = mf + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                                if (ncol < p.N && m < p.M) x += p.rowvec[(long long)(m / p.rows_per_sample) * p.rowvec_ld + ncol];
                            }
                        }
                        if constexpr (GEGLU) {
                            float g = acc[i][(j + 1) < FN ? (j + 1) : j][r];
                            if constexpr (PLAIN) g += gbias; else g = g * p.alpha + gbias;
                            x = x * gelu_erf_f(g);
                        } else if constexpr (!PLAIN) {
                            if (p.act == ACT_QUICKGELU) x = quickgelu_f(x);
                            else if (p.act == ACT_SILU) x = silu_f(x);
                        }
                        v[r] = x;
                    }
                    if constexpr (MODE == 2) softmax_group16(v, p.sm_group);
                    // even lane: row R(2t), cols (c, c+1); odd lane: row R(2t)+1, cols (c-1, c)
                    const int lc = (GEGLU ? (j >> 1) : j) * 32 + frow - odd;          // local column of the pair
                    char* wp = stg + (4 * fhalf + odd) * ROWB + lc * 2;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const int roff = ((2 * t) & 3) + 8 * ((2 * t) >> 2);
                        const float give = odd ? v[2 * t] : v[2 * t + 1];
                        const float got = swap_adjacent_lane(give);
                        const float lo = odd ? got : v[2 * t], hi = odd ? v[2 * t + 1] : got;
                        *(uint32_t*)(wp + roff * ROWB) = cvt_pk_bf16(lo, hi);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's tile is in LDS (LDS ops retire in order)
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int idx = it * 64 + lane, row = idx / CPR, ch = idx - row * CPR;
                    const int m = mf + row, col = eno + ch * 8;
                    uint4 u = *(const uint4*)(stg + row * ROWB + ch * 16);
                    if (full || (m < p.M && col < No)) {
                        long long o = (long long)m * p.ldo + col;
                        if constexpr (CONV == 3) {                      // source pixel (b, y, x) -> output pixel (2y + a, 2x + b)
                            const int hw = p.Hin * p.Win, bb = m / hw, rem = m - bb * hw, yy = rem / p.Win, xx = rem - yy * p.Win;
                            o = ((long long)(bb * p.Hout + 2 * yy + ph_a) * p.Wout + 2 * xx + ph_b) * p.ldo + col;
                        }
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
                        if (p.dbg & 8) o &= 0xfff8;
                        *(uint4*)(ob + o) = u;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next fragment row overwrites
            }
        };
        if (lds_epi) {
            const bool simple = p.alpha == 1.0f && !(p.rowvec && !uniform_sample);
            const bool plain = simple && (GEGLU || p.act == ACT_NONE);
            bool done = false;
            if constexpr (BN == 128 && CONV == 0 && !GEGLU) {          // only the 128-wide linear tiles carry the softmax variant
                if (simple && p.act == ACT_SOFTMAXG) { lds_epilogue(std::integral_constant<int, 2>{}); done = true; }
            }
            if (!done) { if (plain) lds_epilogue(std::integral_constant<int, 1>{}); else lds_epilogue(std::integral_constant<int, 0>{}); }
        } else if (!(p.dbg & 4)) {
#pragma unroll
            for (int i = 0; i < FM; i++) {
                const int mf = em0 + wm * WM + i * 32;                 // first row of this fragment
                // bf16 residual words are requested two fragments at a time, ahead of the stores: loads must not trail
                // the previous fragment's stores (out may alias the residual, so the compiler will not hoist them),
                // and 16 words in flight per lane is what the register budget of the 192-wide tiles allows
                constexpr bool BATCH_RES = !GEGLU && (CONV == 0 || BN == 128);   // the 192-wide conv tiles have no registers to spare
                uint32_t rw[2][8];
                auto load_res = [&](int j0) {
                    if constexpr (BATCH_RES) {
                        if (rb) {
#pragma unroll
                            for (int jj = 0; jj < 2; jj++) {
                                const int j = j0 + jj;
                                if (j >= FN) continue;
                                const int pc = en0 + wn * WN + j * 32 + frow - odd;
                                const long long rbase = (long long)(mf + 4 * fhalf + odd) * p.ldo + pc;
                                const bool pok = full || (pc + 1 < p.N);
#pragma unroll
                                for (int t = 0; t < 8; t++) {
                                    const int roff = ((2 * t) & 3) + 8 * ((2 * t) >> 2);
                                    rw[jj][t] = 0;
                                    if (full || (pok && mf + 4 * fhalf + odd + roff < p.M)) rw[jj][t] = *(const uint32_t*)(rb + rbase + (long long)roff * p.ldo);
                                }
                            }
                        }
                    }
                };
#pragma unroll
                for (int j = 0; j < FN; j++) {
                    if constexpr (GEGLU) { if (j & 1) continue; }
                    if ((j & 1) == 0) load_res(j);
                    const int ncol = en0 + wn * WN + j * 32 + frow;      // column in (permuted) weight space
                    const bool col_ok = full || ncol < p.N;
                    const int ocol = GEGLU ? ((en0 + wn * WN + j * 32) >> 1) + frow : ncol;
                    const float bias = pbias[i][j];
                    float gbias = 0.f;
                    if constexpr (GEGLU) gbias = pbias[i][(j + 1) < FN ? (j + 1) : j];
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float x = acc[i][j][r] * p.alpha + bias;
                        if (p.rowvec && !uniform_sample) {
                            const int m = mf + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                            if (col_ok && m < p.M) x += p.rowvec[(long long)(m / p.rows_per_sample) * p.rowvec_ld + ncol];
                        }
                        if constexpr (GEGLU) {
                            const float g = acc[i][(j + 1) < FN ? (j + 1) : j][r] * p.alpha + gbias;
                            x = x * gelu_erf_f(g);
                        } else {
                            if (p.act == ACT_QUICKGELU) x = quickgelu_f(x);
                            else if (p.act == ACT_SILU) x = silu_f(x);
                        }
                        v[r] = x;
                    }
                    // pair (r = 2t, 2t+1): even lane keeps row R(2t) cols (c, c+1); odd lane keeps row R(2t)+1 cols (c-1, c)
                    const int mrow = mf + 4 * fhalf + odd;
                    const int pcol = ocol - odd;
                    const bool pair_ok = full || (pcol + 1 < (GEGLU ? p.N / 2 : p.N));
                    const long long base = (long long)mrow * p.ldo + pcol;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const int roff = ((2 * t) & 3) + 8 * ((2 * t) >> 2);      // R(2t): 0,2,8,10,16,18,24,26
                        const float give = odd ? v[2 * t] : v[2 * t + 1];
                        const float got = swap_adjacent_lane(give);
                        float lo = odd ? got : v[2 * t];
                        float hi = odd ? v[2 * t + 1] : got;
                        if (full || (pair_ok && mrow + roff < p.M)) {
                            long long o = base + (long long)roff * p.ldo;
                            if (p.dbg & 8) o &= 0xfffe;            // ablation: all stores land in one 64 KB window (no HBM write stream)
                            if constexpr (BATCH_RES) { if (rb) { const uint32_t u = rw[j & 1][t]; lo += __uint_as_float(u << 16); hi += __uint_as_float(u & 0xffff0000u); } }
                            else { if (rb) { const uint32_t u = *(const uint32_t*)(rb + o); lo += __uint_as_float(u << 16); hi += __uint_as_float(u & 0xffff0000u); } }
                            if (rf) { const float2 f = *(const float2*)(rf + o); lo += f.x; hi += f.y; }
                            if (ob) *(uint32_t*)(ob + o) = cvt_pk_bf16(lo, hi);
                            if (of) *(float2*)(of + o) = make_float2(lo, hi);
                        }
                    }
                }
            }
        }
        if (p.dbg & 16) tprof[1] += __builtin_readcyclecounter() - tp1;
        if (!has_next) break;
        tile = next;
        {
            constexpr int NFRAG = GEGLU ? FM * FN / 2 : FM * FN;
            const int nout = (ob ? 1 : 0) + (of ? 1 : 0);
            if (lds_epi) pending_stores = full ? FM * ((32 * (WNO / 8)) / 64) : 0;     // 16-byte row stores per lane
            else pending_stores = (full && nout == 1 && !(p.dbg & 4)) ? NFRAG * 8 : 0;      // exact only for full tiles
            if (pending_stores == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // unknown count: drain now
        }
    }
    if ((p.dbg & 16) && tid == 0) {
        atomicAdd(&g_igemm_prof[0], tprof[0]); atomicAdd(&g_igemm_prof[1], tprof[1]); atomicAdd(&g_igemm_prof[2], tprof[2]);
        atomicAdd(&g_igemm_prof[4], tprof[3]);
        atomicAdd(&g_igemm_prof[3], 1ull);
    }
}

template <int BM, int BN, int WAVES_M, int CONV, bool GEGLU>
static hipError_t launch_cfg(const IgemmParams& p, int batch, hipStream_t st) {
    constexpr int smem = ((BM >= 256 ? 3 : 2) * BM + 2 * BN) * 128;     // A ring (3 slots for 256-row tiles) + B ring of 2
    constexpr int NT = WAVES_M * 128;
    static int bpc_dev[RDM_MAX_DEVICES] = {0}, ncu_dev[RDM_MAX_DEVICES] = {0};
    const int dev = rdm_cur_device();
    if (!bpc_dev[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, WAVES_M, CONV, GEGLU>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipDeviceGetAttribute(&ncu_dev[dev], hipDeviceAttributeMultiprocessorCount, dev);
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)igemm_kernel<BM, BN, WAVES_M, CONV, GEGLU>, NT, smem);
        if (e != hipSuccess) return e;
        bpc_dev[dev] = occ < 1 ? 1 : occ;
    }
    const int blocks_per_cu = bpc_dev[dev], ncu = ncu_dev[dev];
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const long long ntiles = (long long)nbm * nbn;
    long long g = (long long)ncu * blocks_per_cu;            // persistent: one resident wave of blocks
    if (batch > 1) g = (g + batch - 1) / batch;
    g = (g + 7) & ~7LL;                                      // multiple of the XCD count
    if (g > ntiles) g = ntiles;
    if (g < 1) g = 1;
    dim3 grid((unsigned)g, 1, batch);
    static const int prof = getenv("RDM_IGEMM_PROF") ? atoi(getenv("RDM_IGEMM_PROF")) : 0;
    if (prof) {      // dev-only: synchronous launch, prints wave-0 shader cycles per block
        IgemmParams q = p; q.dbg |= 16;
        unsigned long long z[5] = {0, 0, 0, 0, 0}, r[5];
        hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_prof), z, sizeof(z));
        igemm_kernel<BM, BN, WAVES_M, CONV, GEGLU><<<grid, NT, smem, st>>>(q);
        hipStreamSynchronize(st);
        hipMemcpyFromSymbol(r, HIP_SYMBOL(g_igemm_prof), sizeof(r));
        fprintf(stderr, "[igemm<%d,%d,%d,%d> M=%d N=%d K=%d] blocks=%llu per-block cycles: kloop %.0f (of which tile-start wait %.0f) epilogue %.0f (of which waiting for the other waves %.0f) (tiles/block %.2f)\n",
                BM, BN, CONV, (int)GEGLU, p.M, p.N, p.K, r[3], (double)r[0] / r[3], (double)r[2] / r[3], (double)r[1] / r[3], (double)r[4] / r[3], (double)ntiles / g);
        return hipGetLastError();
    }
    igemm_kernel<BM, BN, WAVES_M, CONV, GEGLU><<<grid, NT, smem, st>>>(p);
    return hipGetLastError();
}

// Host entry: picks the tile. K must be a multiple of 64 (and C0, C1 multiples of 64 for conv / dual).
// phase weights of conv3x3(nearest2x(.)):  Wp[2a + b][n][ty][tx][c] = sum of W[n][ky][kx][c] over the taps (ky, kx) that land on source
// offset (a - 1 + ty, b - 1 + tx): rows a = 0: {0} | {1, 2}, a = 1: {0, 1} | {2}; columns likewise.  Summed in fp32, rounded once to bf16.
__global__ __launch_bounds__(256) void conv_phase_weights_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ Wp, int N, int C) {
    const long long total = (long long)4 * N * 4 * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C); long long r = i / C;
        const int tx = (int)(r & 1), ty = (int)((r >> 1) & 1); r >>= 2;
        const int n = (int)(r % N), ph = (int)(r / N), a = ph >> 1, b = ph & 1;
        const int ky0 = a == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = a == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
        const int kx0 = b == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = b == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
        float s = 0.f;
        for (int ky = ky0; ky <= ky1; ky++)
            for (int kx = kx0; kx <= kx1; kx++) s += bf2f(W[((long long)n * 9 + ky * 3 + kx) * C + c]);
        Wp[i] = f2bf(s);
    }
}
hipError_t launch_conv_phase_weights(const bf16_t* W, bf16_t* Wp, int N, int C, hipStream_t st) {
    long long g = ((long long)16 * N * C + 255) / 256; if (g > 8192) g = 8192;
    conv_phase_weights_kernel<<<dim3((unsigned)g), 256, 0, st>>>(W, Wp, N, C);
    return hipGetLastError();
}

hipError_t launch_igemm(const IgemmParams& p_in, bool conv, int batch, hipStream_t st) {
    static const int dbg = getenv("RDM_IGEMM_DBG") ? atoi(getenv("RDM_IGEMM_DBG")) : 0;
    IgemmParams p = p_in; p.dbg = dbg;
    static const int no_resk = getenv("RDM_NO_RESK") ? atoi(getenv("RDM_NO_RESK")) : 0;
    p.res_k = 0;
    if (p.K % 64 != 0 || p.C0 % 64 != 0 || p.C1 % 64 != 0) return hipErrorInvalidValue;
    if (p.N % 2 != 0 || p.ldo % 2 != 0 || p.sO % 2 != 0) return hipErrorInvalidValue;   // paired-column epilogue
    if (p.act == ACT_GEGLU && (p.N % 64 != 0)) return hipErrorInvalidValue;
    if (p.act == ACT_SOFTMAXG && (conv || p.N % 192 == 0 || p.alpha != 1.0f || !p.out_bf16 || p.out_f32 || p.res_f32 || p.ldo % 8 ||
                                  !(p.sm_group == 1 || p.sm_group == 2 || p.sm_group == 4)))
        return hipErrorInvalidValue;                        // the column-group softmax lives in the 128-wide bf16 epilogue only
    // tall 256-row tiles (8 waves, 1 block/CU) cut the L2->LDS operand traffic per FLOP by 1.44x; use them
    // whenever there are enough row tiles to fill the chip, else the 128-row tile (4 waves, 2 blocks/CU).
    static const int force_bm = getenv("RDM_IGEMM_BM") ? atoi(getenv("RDM_IGEMM_BM")) : 0;
    const bool wide = (p.N % 192 == 0);
    const long long tiles256 = (long long)((p.M + 255) / 256) * ((p.N + (wide ? 191 : 127)) / (wide ? 192 : 128)) * batch;
    bool tall = tiles256 >= 256 && p.M > 128;      // 8 waves, 3-deep A ring: the HBM latency of the activation stream is covered
                                                   // (M <= 128 rows per batch item: a 256-row tile would be mostly padding)
    if (force_bm == 128) tall = false;
    if (force_bm == 256) tall = true;
    if (!conv && p.Wfrag && lin4_supported(p, batch)) return launch_lin4(p, st);       // lin4.hip
    if (p.a1_wrap_rows > 0 || p.res_wrap_rows > 0) return hipErrorInvalidValue;                               // a wrapped second source is read by lin4 only
    if (p.act == ACT_GEGLU) {
        if (conv) return hipErrorInvalidValue;
        // 256-wide tile (x and gate interleaved: 128 outputs): the A tile is re-read once per column tile, and the
        // L2->LDS path (1 KiB per ~20 cycles per CU), not the MFMA pipe, bounds the 128-wide tile's K loop
        static const int geglu_bn = getenv("RDM_GEGLU_BN") ? atoi(getenv("RDM_GEGLU_BN")) : 256;
        if (tall && geglu_bn == 256 && p.N % 256 == 0 && (long long)((p.M + 255) / 256) * (p.N / 256) * batch >= 256)
            return launch_cfg<256, 256, 4, 0, true>(p, batch, st);
        return tall ? launch_cfg<256, 128, 4, 0, true>(p, batch, st) : launch_cfg<128, 128, 2, 0, true>(p, batch, st);
    }
    if (conv && p.phase2) {
        // bf16 output through the LDS read-out only (it owns the row scatter); no residual / time row / activation at an Upsample conv
        if (batch != 4 || !p.out_bf16 || p.out_f32 || p.res_bf16 || p.res_f32 || p.rowvec || p.act != ACT_NONE || p.alpha != 1.0f || p.ldo % 8 || p.N % 8 ||
            p.C1 != 0 || p.K != 4 * p.C0 || p.M != (long long)(p.M / (p.Hin * p.Win)) * p.Hin * p.Win || p.Hout != 2 * p.Hin || p.Wout != 2 * p.Win)
            return hipErrorInvalidValue;
        if (wide) return tall ? launch_cfg<256, 192, 4, 3, false>(p, batch, st) : launch_cfg<128, 192, 2, 3, false>(p, batch, st);
        return tall ? launch_cfg<256, 128, 4, 3, false>(p, batch, st) : launch_cfg<128, 128, 2, 3, false>(p, batch, st);
    }
    if (conv && p.ups) {
        if (wide) return tall ? launch_cfg<256, 192, 4, 2, false>(p, batch, st) : launch_cfg<128, 192, 2, 2, false>(p, batch, st);
        return tall ? launch_cfg<256, 128, 4, 2, false>(p, batch, st) : launch_cfg<128, 128, 2, 2, false>(p, batch, st);
    }
    if (conv) {
        if (wide) return tall ? launch_cfg<256, 192, 4, 1, false>(p, batch, st) : launch_cfg<128, 192, 2, 1, false>(p, batch, st);
        return tall ? launch_cfg<256, 128, 4, 1, false>(p, batch, st) : launch_cfg<128, 128, 2, 1, false>(p, batch, st);
    }
    // linear with a bf16 residual: feed the residual tile through the A stream against an identity block (see the kernel)
    if (!no_resk && p.res_bf16 && !p.res_f32 && p.out_bf16 && !p.out_f32 && p.alpha == 1.0f && p.act == ACT_NONE &&
        p.N % (wide ? 192 : 128) == 0 && p.ldo % 8 == 0 && p.ldo >= p.N)
        p.res_k = 1;
    if (wide) return tall ? launch_cfg<256, 192, 4, 0, false>(p, batch, st) : launch_cfg<128, 192, 2, 0, false>(p, batch, st);
    return tall ? launch_cfg<256, 128, 4, 0, false>(p, batch, st) : launch_cfg<128, 128, 2, 0, false>(p, batch, st);
}

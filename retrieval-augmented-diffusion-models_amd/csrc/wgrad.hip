// Weight gradients of the training path (ldm/models/diffusion/ddpm.py p_losses -> backward of the UNet's convs and linears).
//
//   dW[n][k] = sum_m dY[m][n] X[m][k]          (m: pixel / token rows -- the reduction runs over the LONG axis, the output is small)
//
// Both operands arrive the way the forward and backward kernels leave them: activation layout, reduction index m OUTERMOST, so the
// MFMA fragments (8 consecutive m of one column per lane) are strided in memory on BOTH sides.  The round-3 path transposed both
// tensors through HBM first (13 % of a training step in transpose kernels) and ran the generic K-major GEMM.  This kernel reads the
// natural layout: 16-byte LDS-DMA copies [32 rows][32 columns] sub-images (64-byte rows) into a 3-deep ring, and the fragments
// are gathered with the hardware transpose read ds_read_b64_tr_b16 (each 16-lane group reads a [4 rows][16 columns] block and
// receives it column-major: the attention kernel's V recipe; 64-byte rows keep the 4 rows of a group on distinct banks).
//
// Block: 192 x 192 outputs (every channel count of the shipped UNet is a multiple of 192; others are masked per 32-column group),
// 4 waves as 2 x 2, a wave holds 3 x 3 MFMA 32x32x16 tiles (144 accumulators).  The M rows are split over Z blocks writing fp32 planes
// that launch_reduce_planes adds in plane order (deterministic: Z depends on the shape only).
// Conv3x3 mode (taps = 9): tap (ky, kx) is the same GEMM against X shifted by (ky-1, kx-1) pixels with out-of-image rows read from the
// zero page; the nine taps of one (tile, z) are queued on ONE XCD so that its L2 serves the operands they share.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int WG_MC = 32, WG_DEPTH = 3;
constexpr int WG_SUB = WG_MC * 64;                  // bytes of one [MC rows][32 columns] sub-image
constexpr int WG_STAGE = 12 * WG_SUB;               // 6 column groups of dY, 6 of X
constexpr int WG_LDS = WG_DEPTH * WG_STAGE;         // 73728 B: two blocks per CU

struct WgradParams {
    const bf16_t* a; int lda;                       // dY [M][N]
    const bf16_t* b; int ldb;                       // X  [M][K]
    float* out; long long plane; int ldo, tap_stride;       // plane z: out + z * plane + n * ldo + tap * tap_stride + k
    int M, N, K, mz, taps, tiles_n, tiles;
    int lw, lh;                                     // taps == 9: log2 of the image width / height
    const void* zero_page;
    int remap;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

}  // namespace

__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1, hf = lane >> 5;
    int tap, grp;
    {
        const int lin = blockIdx.x;
        if (p.remap) { const int xcd = lin & 7, slot = lin >> 3; grp = (slot / p.taps) * 8 + xcd; tap = slot % p.taps; }
        else { tap = lin % p.taps; grp = lin / p.taps; }
    }
    const int z = grp / p.tiles, t = grp % p.tiles;
    const int n0 = (t % p.tiles_n) * 192, k0 = (t / p.tiles_n) * 192;
    const int sy = p.taps == 9 ? tap / 3 - 1 : 0, sx = p.taps == 9 ? tap % 3 - 1 : 0;
    const int W = 1 << p.lw, H = 1 << p.lh;
    const int m_begin = z * p.mz, m_end = min(p.M, m_begin + p.mz);
    const int nchunk = m_end > m_begin ? (m_end - m_begin + WG_MC - 1) / WG_MC : 0;

    // loader: wave-instruction q = wave + 4 j (j = 0..5) fills column group q >> 1 (0..5 dY, 6..11 X), rows 16 (q & 1) + (lane >> 2) of the chunk,
    // this lane the 16-byte piece lane & 3 of the 64-byte row
    const bf16_t* src[6]; bool cval[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int q = wave + 4 * j, g = (q >> 1) - (j >= 3 ? 6 : 0), row = 16 * (q & 1) + (lane >> 2);
        if (j < 3) {
            const int col = n0 + g * 32 + (lane & 3) * 8;
            cval[j] = col < p.N;
            src[j] = p.a + (long long)(m_begin + row) * p.lda + col;
        } else {
            const int col = k0 + g * 32 + (lane & 3) * 8;
            cval[j] = col < p.K;
            src[j] = p.b + ((long long)(m_begin + row) + sy * W + sx) * p.ldb + col;
        }
    }
    auto request = [&](int c) {
        char* stage = lds + (c % WG_DEPTH) * WG_STAGE;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int q = wave + 4 * j;
            const int m = m_begin + c * WG_MC + 16 * (q & 1) + (lane >> 2);
            bool ok = cval[j] && m < m_end;
            if (j >= 3 && p.taps == 9) {
                const int x = (m & (W - 1)) + sx, y = ((m >> p.lw) & (H - 1)) + sy;
                ok = ok && (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
            }
            const void* g = ok ? (const void*)(src[j] + (long long)c * WG_MC * (j < 3 ? p.lda : p.ldb)) : p.zero_page;
            glds16(g, stage + (q >> 1) * WG_SUB + (q & 1) * 1024);
        }
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // transpose-read address inside a sub-image: rows 8 hf + (i >> 2) (+4 for the second read), columns 16 g + 4 (i & 3)   (i = lane & 15, g = (lane >> 4) & 1)
    const uint32_t tr = (8 * hf + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    for (int c = 0; c < WG_DEPTH - 1 && c < nchunk; c++) request(c);
    for (int c = 0; c < nchunk; c++) {
        if (c + 1 < nchunk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // chunk c visible; the slot of chunk c-1 is free
        if (c + WG_DEPTH - 1 < nchunk) request(c + WG_DEPTH - 1);
        const LDS_AS char* S = (const LDS_AS char*)(lds + (c % WG_DEPTH) * WG_STAGE) + tr;
#pragma unroll
        for (int ks = 0; ks < WG_MC / 16; ks++) {
            bf16x8 af[3], bfr[3];
#pragma unroll
            for (int f = 0; f < 3; f++) {
                union { bf16x8 v; s16x4 h[2]; } ua, ub;
                const LDS_AS char* pa = S + (wn * 3 + f) * WG_SUB + ks * 1024;
                const LDS_AS char* pb = S + (6 + wk * 3 + f) * WG_SUB + ks * 1024;
                ua.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(pa));
                ua.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(pa + 256));
                ub.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(pb));
                ub.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(pb + 256));
                af[f] = ua.v; bfr[f] = ub.v;
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    float* out = p.out + (long long)z * p.plane + (long long)tap * p.tap_stride;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int nf = n0 + wn * 96 + i * 32;
        if (nf >= p.N) continue;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int k = k0 + wk * 96 + j * 32 + (lane & 31);
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = nf + (r & 3) + 8 * (r >> 2) + 4 * hf;
                out[(long long)n * p.ldo + k] = acc[i][j][r];
            }
        }
    }
}


// ---- conv3x3 weight gradient with the nine taps in ONE block (image widths 16 / 32 / 64).  The per-tap GEMM above moves (192 + 192)
// columns per 18 MFMAs of a wave and measured ingest-bound (rocprofv3: 42 % of the wave cycles waiting on vmcnt, MFMA busy 26 %,
// ~30 GB/s per CU of LDS-DMA -- the same per-CU ceiling the decode-size GEMMs sit on).  Here a block owns 64 output channels x 64 input
// channels x 9 taps: the dY chunk and each X image row enter LDS once and serve all nine taps (3 x fewer bytes per MFMA, the same 144
// accumulators per wave).  X rows live in a ring of zero-bordered row slots ([2 column groups][W + 2 pixels][64 B]: the pad pixels are
// zeroed once, the DMA only ever writes the interior), so tap (ky, kx) is the SAME transpose read shifted by (ky - 1) slots and
// (kx - 1) pixels; rows above / below the image read a slot of zeros.  Rows stream through the ring in global row order (RY rows per
// chunk, two chunks ahead), samples back to back -- the row of the neighbouring sample that the ring holds at an image edge is simply
// not addressed.
template <int LW>
__global__ __launch_bounds__(256, 2) void wgrad_conv9_kernel(WgradParams p) {
    constexpr int W = 1 << LW, MCW = W > 32 ? W : 32, RY = MCW / W, NR = 3 * RY + 2, KS = MCW / 16, RUNS = W / 16;
    constexpr int XG = (W + 2) * 64, XSLOT = 2 * XG, AG = MCW * 64, ASLOT = 2 * AG;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const Ar = lds;                                  // [3][ASLOT]
    char* const Xr = lds + 3 * ASLOT;                      // [NR][XSLOT]
    char* const Zs = Xr + NR * XSLOT;                      // one slot of zeros
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1, hf = lane >> 5;
    int z, t;
    {
        const int lin = blockIdx.x;
        // p.remap = Z rounded down to a multiple of 8: those Z-chunks are dealt to the XCDs whole (all tiles of a chunk on one L2); the rest in launch order
        if (lin < p.remap * p.tiles) { const int xcd = lin & 7, slot = lin >> 3; z = (slot / p.tiles) * 8 + xcd; t = slot % p.tiles; }
        else { const int rem = lin - p.remap * p.tiles; t = rem % p.tiles; z = p.remap + rem / p.tiles; }
    }
    const int n0 = (t % p.tiles_n) * 64, c0 = (t / p.tiles_n) * 64;
    const int H = 1 << p.lh;
    const int total_rows = p.M >> LW;                       // image rows over all samples
    const int ch_begin = z * p.mz, ch_end = min(p.mz * (z + 1), total_rows / RY);      // p.mz: chunks per block
    const int nchunk = max(ch_end - ch_begin, 0);
    const int gs = ch_begin * RY;                          // first global image row of this block

    // zero borders and the zero slot
    for (int i = tid; i < XSLOT / 16; i += 256) *(uint4*)(Zs + i * 16) = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < NR * 2 * 2 * 4; i += 256) {      // (slot, group, side, 16-byte piece)
        const int piece = i & 3, side = (i >> 2) & 1, g = (i >> 3) & 1, sl = i >> 4;
        *(uint4*)(Xr + sl * XSLOT + g * XG + (side ? (W + 1) * 64 : 0) + piece * 16) = make_uint4(0, 0, 0, 0);
    }

    const int lrow = lane >> 2, lcol = (lane & 3) * 8;
    // X image row g (global row index) -> its ring slot, RUNS runs of 16 pixels x 2 column groups; unit u = run * 2 + group
    auto load_x_unit = [&](int g, int u) {
        const int grp = u & 1, run = u >> 1;
        const long long pix = (long long)g * W + run * 16 + lrow;
        const bool ok = g >= 0 && g < total_rows;
        const void* src = ok ? (const void*)(p.b + pix * p.ldb + c0 + grp * 32 + lcol) : p.zero_page;
        const int sl = ((g % NR) + NR) % NR;
        glds16(src, Xr + sl * XSLOT + grp * XG + (run * 16 + 1) * 64);
    };
    auto request = [&](int c) {
        // dY chunk c: KS runs x 2 groups (units 0 .. 2 KS - 1), then the RY image rows below the chunk's first row + ... (see the ring invariant)
#pragma unroll
        for (int j = 0; j < KS; j++) {
            const int q = wave + 4 * j;
            if (j < KS / 2) {
                const int grp = q & 1, run = q >> 1;
                const long long pix = (long long)(gs + c * RY) * W + run * 16 + lrow;
                const bool ok = pix < p.M;
                const void* src = ok ? (const void*)(p.a + pix * p.lda + n0 + grp * 32 + lcol) : p.zero_page;
                glds16(src, Ar + (c % 3) * ASLOT + grp * AG + run * 1024);
            } else {
                const int u = q - 2 * KS;                   // 0 .. 2 KS - 1 = RY rows x RUNS runs x 2 groups
                const int r = u / (2 * RUNS), uu = u - r * 2 * RUNS;
                load_x_unit(gs + c * RY + 1 + r, uu);
            }
        }
    };
    // prologue: rows gs - 1 and gs (2 x RUNS x 2 units over 4 waves), then chunks 0 and 1
#pragma unroll
    for (int j = 0; j < RUNS; j++) {
        const int q = wave + 4 * j, r = q / (2 * RUNS), uu = q - r * 2 * RUNS;
        load_x_unit(gs - 1 + r, uu);
    }
    __syncthreads();                                       // the zero fills above are ordinary stores: visible before any read below
    if (nchunk > 0) request(0);
    if (nchunk > 1) request(1);

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const uint32_t tr = (8 * hf + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    for (int c = 0; c < nchunk; c++) {
        if (c + 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 2 < nchunk) request(c + 2);
        const LDS_AS char* A = (const LDS_AS char*)(Ar + (c % 3) * ASLOT + wn * AG) + tr;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int g = gs + c * RY + ((16 * s) >> LW), x0 = (16 * s) & (W - 1), y = g & (H - 1);
            union { bf16x8 v; s16x4 h[2]; } ua;
            ua.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(A + s * 1024));
            ua.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(A + s * 1024 + 256));
#pragma unroll
            for (int ty = 0; ty < 3; ty++) {
                const int gy = g + ty - 1;
                const bool in = ty == 1 || (ty == 0 ? y > 0 : y < H - 1);
                const char* base = in ? Xr + (gy % NR) * XSLOT : Zs;
                const LDS_AS char* X = (const LDS_AS char*)(base + wk * XG + x0 * 64) + tr;
#pragma unroll
                for (int tx = 0; tx < 3; tx++) {
                    union { bf16x8 v; s16x4 h[2]; } ub;
                    ub.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(X + tx * 64));
                    ub.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(X + tx * 64 + 256));
                    acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[ty * 3 + tx], 0, 0, 0);
                }
            }
        }
    }
    float* out = p.out + (long long)z * p.plane;
    const int k = c0 + wk * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            out[(long long)n * p.ldo + tap * p.tap_stride + k] = acc[tap][r];
        }
}
template <int LW> constexpr int conv9_lds_bytes() {
    constexpr int W = 1 << LW, MCW = W > 32 ? W : 32, RY = MCW / W, NR = 3 * RY + 2;
    return 3 * 2 * MCW * 64 + (NR + 1) * 2 * (W + 2) * 64;
}

namespace {

void wgrad_geom(long long M, int N, int K, int taps, int* pZ, int* pmz) {
    const long long tiles = (long long)((N + 191) / 192) * ((K + 191) / 192);
    long long Z = 512 / (tiles * taps);                    // one round of blocks at two per CU
    if (Z > M / 256) Z = M / 256;
    const long long plane = (long long)N * taps * K * 4;
    if (Z > (256LL << 20) / plane) Z = (256LL << 20) / plane;
    if (Z >= 8) Z &= ~7LL;
    if (Z < 1) Z = 1;
    long long mz = (M + Z - 1) / Z; mz = (mz + WG_MC - 1) / WG_MC * WG_MC;
    Z = (M + mz - 1) / mz;
    *pZ = (int)Z; *pmz = (int)mz;
}

}  // namespace

bool wgrad_tn_supported(long long M, int N, int K, int lda, int ldb) {
    static const bool off = getenv("RDM_NO_WGRAD_TN") != nullptr;
    return !off && M >= 256 && M < (1LL << 30) && N % 32 == 0 && K % 32 == 0 && lda % 8 == 0 && ldb % 8 == 0;
}
bool conv_wgrad_tn_supported(int B, int H, int W, int C, int N) {
    return (H & (H - 1)) == 0 && (W & (W - 1)) == 0 && wgrad_tn_supported((long long)B * H * W, N, C, N, C);
}
size_t wgrad_tn_scratch_bytes(long long M, int N, int K, int taps) {
    int Z, mz; wgrad_geom(M, N, K, taps, &Z, &mz);
    return Z > 1 ? (size_t)Z * N * taps * K * 4 : 0;      // (3x3 convs: conv_wgrad_tn_scratch_bytes below -- the nine-tap kernel picks its own Z)
}

// nine-tap kernel: W in {16, 32, 64}, H a power of two, channel counts multiples of 64
static bool conv9_ok(int B, int H, int W, int C, int N) {
    static const bool off = getenv("RDM_NO_WGRAD_CONV9") != nullptr;
    return !off && (W == 16 || W == 32 || W == 64) && (H & (H - 1)) == 0 && H >= 2 && C % 64 == 0 && N % 64 == 0 && (long long)B * H * W < (1LL << 30) &&
           ((long long)B * H * W) % (W > 32 ? W : 32) == 0;
}
static void conv9_geom(int B, int H, int W, int C, int N, int* pZ, int* pcz) {
    const int MCW = W > 32 ? W : 32;
    const long long chunks = (long long)B * H * W / MCW, tiles = (long long)(N / 64) * (C / 64);
    // one round of blocks: 2 per CU at width 64 (75 KB of LDS each), 3 per CU below (168 VGPRs, 38 KB)
    const long long slots = W == 64 ? 512 : 768;
    long long Z = slots / tiles;
    if (Z > chunks / 8) Z = chunks / 8;
    const long long plane = (long long)N * 9 * C * 4;
    if (Z > (256LL << 20) / plane) Z = (256LL << 20) / plane;
    if (Z < 1) Z = 1;
    const long long cz = (chunks + Z - 1) / Z;
    Z = (chunks + cz - 1) / cz;
    *pZ = (int)Z; *pcz = (int)cz;
}
// scratch of launch_wgrad_tn(taps = 9) for THIS geometry: the planes of the kernel that will run (nine-tap kernel: its own Z, at most
// 256 MB; per-tap fallback: wgrad_geom's Z), nothing when one plane goes straight into dw
size_t conv_wgrad_tn_scratch_bytes(int B, int H, int W, int C, int N) {
    int Z, t;
    if (conv9_ok(B, H, W, C, N)) conv9_geom(B, H, W, C, N, &Z, &t);
    else wgrad_geom((long long)B * H * W, N, C, 9, &Z, &t);
    return Z > 1 ? (size_t)Z * N * 9 * C * 4 : 0;
}
template <int LW>
static hipError_t launch_conv9(WgradParams& p, int Z, hipStream_t st) {
    static bool attr[RDM_MAX_DEVICES] = {false};
    const int dev = rdm_cur_device();
    if (!attr[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_conv9_kernel<LW>, hipFuncAttributeMaxDynamicSharedMemorySize, conv9_lds_bytes<LW>());
        if (e != hipSuccess) return e;
        attr[dev] = true;
    }
    wgrad_conv9_kernel<LW><<<dim3((unsigned)(p.tiles * Z)), 256, conv9_lds_bytes<LW>(), st>>>(p);
    return hipGetLastError();
}
// dw [N][taps][K] fp32.  taps = 1: a = dY [M][N] (row stride lda), b = X [M][K] (row stride ldb).  taps = 9: M = B H W pixels of [B][H][W] images.
hipError_t launch_wgrad_tn(const bf16_t* dy, int lda, const bf16_t* x, int ldb, float* dw, long long M, int N, int K, int taps, int H, int W, char* scratch,
                           const void* zero_page, hipStream_t st) {
    static bool attr[RDM_MAX_DEVICES] = {false};
    const int dev = rdm_cur_device();
    if (!attr[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS);
        if (e != hipSuccess) return e;
        attr[dev] = true;
    }
    int Z, mz; wgrad_geom(M, N, K, taps, &Z, &mz);
    WgradParams p{};
    if (taps == 9 && lda == N && ldb == K && conv9_ok((int)(M / ((long long)H * W)), H, W, K, N)) {
        int cz; conv9_geom((int)(M / ((long long)H * W)), H, W, K, N, &Z, &cz);
        p.a = dy; p.lda = lda; p.b = x; p.ldb = ldb; p.plane = (long long)N * 9 * K; p.ldo = 9 * K; p.tap_stride = K;
        p.out = Z > 1 ? (float*)scratch : dw;
        p.M = (int)M; p.N = N; p.K = K; p.mz = cz; p.taps = 9; p.tiles_n = N / 64; p.tiles = p.tiles_n * (K / 64);
        while ((1 << p.lw) < W) p.lw++; while ((1 << p.lh) < H) p.lh++;
        p.zero_page = zero_page; p.remap = Z & ~7;
        hipError_t e = W == 64 ? launch_conv9<6>(p, Z, st) : W == 32 ? launch_conv9<5>(p, Z, st) : launch_conv9<4>(p, Z, st);
        if (e != hipSuccess) return e;
        return Z > 1 ? launch_reduce_planes((const float*)scratch, dw, (long long)N * 9 * K, Z, st) : hipSuccess;
    }
    p.a = dy; p.lda = lda; p.b = x; p.ldb = ldb;
    p.plane = (long long)N * taps * K; p.ldo = taps * K; p.tap_stride = K;
    p.out = Z > 1 ? (float*)scratch : dw;
    p.M = (int)M; p.N = N; p.K = K; p.mz = mz; p.taps = taps;
    p.tiles_n = (N + 191) / 192; p.tiles = p.tiles_n * ((K + 191) / 192);
    p.lw = 0; p.lh = 0;
    if (taps == 9) { while ((1 << p.lw) < W) p.lw++; while ((1 << p.lh) < H) p.lh++; }
    p.zero_page = zero_page;
    p.remap = ((long long)p.tiles * Z) % 8 == 0;
    wgrad_tn_kernel<<<dim3((unsigned)(p.tiles * Z * taps)), 256, WG_LDS, st>>>(p);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    return Z > 1 ? launch_reduce_planes((const float*)scratch, dw, (long long)N * taps * K, Z, st) : hipSuccess;
}

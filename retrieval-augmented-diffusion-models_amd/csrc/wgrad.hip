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

namespace {

void wgrad_geom(long long M, int N, int K, int taps, int* pZ, int* pmz) {
    const long long tiles = (long long)((N + 191) / 192) * ((K + 191) / 192);
    long long Z = (512 + tiles * taps - 1) / (tiles * taps);
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
    return Z > 1 ? (size_t)Z * N * taps * K * 4 : 0;
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

// Backward pieces of the training step (MinimalRETRODiffusion.shared_step / ldm p_losses, reference
// rdm/models/diffusion/ddpm.py:390-443), round 3: the two FLOP carriers of the UNet's ResBlocks and the normalisations around them.
//
//   * 3x3 conv dgrad  dX = conv3x3(dY, W~),  W~[c][ky][kx][n] = W[n][2-ky][2-kx][c]: one weight permutation (conv_w_dgrad_kernel) and
//     the FORWARD kernel (conv_halo4.hip / igemm.hip) -- the gradient w.r.t. the input of a stride-1, pad-1 correlation is the
//     correlation of the output gradient with the flipped, transposed filter.
//   * 3x3 conv wgrad  dW[n][tap][c] = sum_pixels dY[p][n] X[p + tap][c]: a pixel-reduction GEMM (M' = N_out, N' = C_in, K' = pixels).
//     Both operands are copied once into K-major, spatially ZERO-PADDED form (pad_transpose_kernel: [C][B (H+2) WP] with the row
//     pitch WP a multiple of 8 and the column shift kx baked into three copies of X), so that a tap is a pure, 16-byte-aligned
//     offset along K and the padding of dY zeroes every product that would wrap around an image edge; the reduction runs on the
//     MFMA implicit-GEMM kernel (igemm.hip) as 9 taps x Z K-chunks (blockIdx.z) of fp32 partials, summed in a fixed order.
//   * GroupNorm(+SiLU) and LayerNorm backward (two passes: group / row sums, then the elementwise gradient), bias / affine gradients
//     as deterministic column sums.
// Everything is fp32-accumulated from bf16 operands like the forward path; gradients w.r.t. weights are fp32.
#include <stdio.h>

#include "kernels.h"

// ---- dgrad weight permutation: w [N][9][C] -> wd [C][9][N] with the taps reversed
__global__ __launch_bounds__(256) void conv_w_dgrad_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wd, int N, int C) {
    const long long total = (long long)N * 9 * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i % N); const long long r = i / N; const int tap = (int)(r % 9); const int c = (int)(r / 9);
        wd[i] = w[((long long)n * 9 + (8 - tap)) * C + c];
    }
}
hipError_t launch_conv_w_dgrad(const bf16_t* w, bf16_t* wd, int N, int C, hipStream_t st) {
    const long long total = (long long)N * 9 * C;
    long long g = (total + 255) / 256; if (g > 8192) g = 8192;
    conv_w_dgrad_kernel<<<dim3((unsigned)g), 256, 0, st>>>(w, wd, N, C);
    return hipGetLastError();
}

// ---- K-major zero-padded copy: x [B, H, W, C] bf16 -> out [C][PR], PR = margin + B (H+2) WP + margin + tail, element (c, b, y, x)
// at  margin + ((b (H+2) + y + 1) WP + x + 1 + shift);  everything else zero.  32 x 32 tiles through LDS (coalesced both ways).
__global__ __launch_bounds__(256) void pad_transpose_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int C,
                                                            int WP, int PR, int margin, int shift) {
    __shared__ bf16_t tile[32][33];
    const int HW = H * W;
    const long long M = (long long)B * HW;
    const long long m0 = (long long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;             // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const long long m = m0 + r; const int c = c0 + tx;
        tile[r][tx] = (m < M && c < C) ? x[m * C + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r; const long long m = m0 + tx;
        if (c < C && m < M) {
            const int b = (int)(m / HW); const int rem = (int)(m - (long long)b * HW); const int y = rem / W, xx = rem - y * W;
            out[(long long)c * PR + margin + ((long long)(b * (H + 2) + y + 1) * WP + xx + 1 + shift)] = tile[tx][r];
        }
    }
}

// ---- fixed-order sum of Z fp32 partial planes
__global__ __launch_bounds__(256) void reduce_planes_kernel(const float* __restrict__ parts, float* __restrict__ out, long long n, int Z) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < Z; z++) s += parts[(long long)z * n + i];
        out[i] = s;
    }
}
hipError_t launch_reduce_planes(const float* parts, float* out, long long n, int Z, hipStream_t st) {
    long long g = (n + 255) / 256; if (g > 8192) g = 8192;
    reduce_planes_kernel<<<dim3((unsigned)g), 256, 0, st>>>(parts, out, n, Z);
    return hipGetLastError();
}

// ---- column sums of a bf16 [M, N] matrix (bias gradient), deterministic: one block per 64 columns, tree over rows
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long long M, int N) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float s = 0.f;
    if (col < N) for (long long m = part; m < M; m += 4) s += bf2f(x[m * N + col]);
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && col < N) out[col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// Two deterministic stages when a scratch buffer is given and N is a multiple of 8 (one block per 64 columns walking all M rows with
// 2-byte loads took 0.46 ms per call -- two thirds of a whole-UNet training step): (1) row chunks of 2048 x column blocks of 256, a
// thread sums 8 columns (16-byte loads) of every 8th row of its chunk, the block's 8 row lanes meet in LDS; (2) the chunk partials are
// added in chunk order.
// Column block = VCB 16-byte vectors (the N / 8 vectors of a row dealt evenly to ceil(N / 256) blocks), RL = 256 / VCB row lanes; the
// row chunk is sized for >= 1024 blocks whatever the shape (round 3 used 2048-row chunks x 256-column blocks: 128 blocks for a
// [262144, 192] gradient, half the chip idle, 0.5 TB/s).
__global__ __launch_bounds__(256) void colsum_part_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, long long M, int N, int CS, int VCB) {
    __shared__ float red[2048];                          // [8 elements][RL][VCB]
    const int RL = 256 / VCB;
    const int cv = threadIdx.x % VCB, rl = threadIdx.x / VCB;
    const int v = blockIdx.y * VCB + cv;                 // vector index in the row
    const long long r0 = (long long)blockIdx.x * CS, r1 = r0 + CS < M ? r0 + CS : M;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool act = rl < RL && v * 8 < N;
    if (act) {
        long long m = r0 + rl;
        for (; m + 3 * RL < r1; m += 4 * RL) {           // four 16-byte loads in flight
            uint4 d[4];
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = *(const uint4*)(x + (m + u * RL) * N + v * 8);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t w[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
#pragma unroll
                for (int e = 0; e < 4; e++) { s[2 * e] += __uint_as_float(w[e] << 16); s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
            }
        }
        for (; m < r1; m += RL) {
            const uint4 d = *(const uint4*)(x + m * N + v * 8);
            const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int e = 0; e < 4; e++) { s[2 * e] += __uint_as_float(w[e] << 16); s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) red[(e * RL + rl) * VCB + cv] = s[e];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < VCB * 8; idx += 256) {
        const int e = idx / VCB, vv = idx - e * VCB;
        const int col = (blockIdx.y * VCB + vv) * 8 + e;
        if (col < N) {
            float t = 0.f;
            for (int r = 0; r < RL; r++) t += red[(e * RL + r) * VCB + vv];
            part[(long long)blockIdx.x * N + col] = t;
        }
    }
}
// chunk partials added in sixteen fixed runs per column (16 columns x 16 runs per block), the runs in a fixed tree: a run is a chain of
// dependent adds behind L2 round trips, so the chain length sets the kernel's time
__device__ __forceinline__ float tree16(const float* r, int stride) {
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = r[i * stride];
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
        for (int i = 0; i < w; i++) t[i] += t[i + w];
    return t[0];
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int nchunk, int N) {
    __shared__ float red[16][16];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4, col = blockIdx.x * 16 + cl;
    const int per = (nchunk + 15) / 16, c0 = q * per, c1 = min(nchunk, c0 + per);
    float t = 0.f;
    if (col < N) for (int c = c0; c < c1; c++) t += part[(long long)c * N + col];
    red[q][cl] = t;
    __syncthreads();
    if (q == 0 && col < N) out[col] = tree16(&red[0][cl], 16);
}
static void colsum_geom(long long M, int N, int* pCS, int* pVCB, int* pcolblocks) {
    const int nv = N / 8, colblocks = (nv + 31) / 32, VCB = (nv + colblocks - 1) / colblocks;
    long long CS = (M * colblocks + 1023) / 1024; CS = (CS + 7) & ~7LL;
    if (CS < 64) CS = 64;
    if (CS > 4096) CS = 4096;
    *pCS = (int)CS; *pVCB = VCB; *pcolblocks = colblocks;
}
size_t colsum_scratch_bytes(long long M, int N) { return (size_t)((M + 63) / 64) * N * sizeof(float); }
hipError_t launch_colsum(const bf16_t* x, float* out, long long M, int N, hipStream_t st, float* scratch) {
    if (!scratch || N % 8 != 0 || M < 256) {
        colsum_kernel<<<dim3((N + 63) / 64), 256, 0, st>>>(x, out, M, N);
        return hipGetLastError();
    }
    int CS, VCB, colblocks; colsum_geom(M, N, &CS, &VCB, &colblocks);
    const int nchunk = (int)((M + CS - 1) / CS);
    colsum_part_kernel<<<dim3(nchunk, colblocks), 256, 0, st>>>(x, scratch, M, N, CS, VCB);
    colsum_finish_kernel<<<dim3((N + 15) / 16), 256, 0, st>>>(scratch, out, nchunk, N);
    return hipGetLastError();
}

// ---- wgrad driver
// scratch layout (bf16 elements): dyT [N][PR], xT0 / xT1 / xT2 [C][PR] (column shifts -1, 0, +1); then fp32 partials [Z][N][9][C]
size_t conv_wgrad_scratch_bytes(int B, int H, int W, int C, int N, int* pWP, int* pPR, int* pK, int* pZ, int* pmargin) {
    const int WP = (W + 2 + 7) & ~7;
    const int margin = WP + 8;
    const long long P = (long long)B * (H + 2) * WP;
    // K' = P rounded up so that Z chunks of a multiple of 64 cover it
    // K-chunks: enough of them that one tap's batched GEMM (ceil(N / 128) x ceil(C / 192) tiles per chunk) covers the chip twice, at
    // least 1 k positions per chunk, at most 256 chunks and 256 MB of fp32 partial planes
    const long long tiles = (long long)((N + 127) / 128) * ((C + 191) / 192);
    long long Zl = (P + 8191) / 8192;
    if (Zl < (512 + tiles - 1) / tiles) Zl = (512 + tiles - 1) / tiles;
    if (Zl > P / 1024) Zl = P / 1024;
    const long long plane = (long long)N * 9 * C * 4;
    if (Zl > (256LL << 20) / plane) Zl = (256LL << 20) / plane;
    if (Zl > 256) Zl = 256;
    if (Zl < 1) Zl = 1;
    int Z = (int)Zl;
    long long Kc = (P + Z - 1) / Z; Kc = (Kc + 63) & ~63LL;
    const long long K = Kc * Z;
    const long long PR = margin + K + margin;
    if (pWP) *pWP = WP; if (pPR) *pPR = (int)PR; if (pK) *pK = (int)Kc; if (pZ) *pZ = Z; if (pmargin) *pmargin = margin;
    return (size_t)(N + 3LL * C) * PR * 2 + (size_t)Z * N * 9 * C * 4 + 256;
}

hipError_t launch_conv_wgrad(const bf16_t* x, const bf16_t* dy, float* dw, int B, int H, int W, int C, int N, char* scratch, const void* zero_page,
                             hipStream_t st) {
    int WP, PR, Kc, Z, margin;
    conv_wgrad_scratch_bytes(B, H, W, C, N, &WP, &PR, &Kc, &Z, &margin);
    bf16_t* dyT = (bf16_t*)scratch;
    bf16_t* xT[3] = {dyT + (size_t)N * PR, dyT + (size_t)(N + C) * PR, dyT + (size_t)(N + 2 * C) * PR};
    float* parts = (float*)(scratch + (((size_t)(N + 3LL * C) * PR * 2 + 255) & ~(size_t)255));
    hipError_t e = hipMemsetAsync(scratch, 0, (size_t)(N + 3LL * C) * PR * 2, st);
    if (e != hipSuccess) return e;
    const long long M = (long long)B * H * W;
    pad_transpose_kernel<<<dim3((unsigned)((M + 31) / 32), (N + 31) / 32), 256, 0, st>>>(dy, dyT, B, H, W, N, WP, PR, margin, 0);
    for (int s = 0; s < 3; s++)            // X shifted so that tap column kx = s reads element p of the dY image: X(p + kx - 1) sits at p
        pad_transpose_kernel<<<dim3((unsigned)((M + 31) / 32), (C + 31) / 32), 256, 0, st>>>(x, xT[s], B, H, W, C, WP, PR, margin, 1 - s);
    e = hipGetLastError(); if (e != hipSuccess) return e;
    for (int tap = 0; tap < 9; tap++) {
        const int ky = tap / 3, kx = tap % 3;
        IgemmParams p{}; p.M = N; p.N = C; p.K = Kc; p.alpha = 1.f; p.zero_page = zero_page;
        p.Hin = p.Win = p.Hout = p.Wout = 1; p.stride = 1; p.rows_per_sample = 1;
        p.A0 = dyT + margin; p.C0 = Kc; p.lda = PR; p.sA = Kc;
        p.W = xT[kx] + margin + (ky - 1) * WP; p.ldw = PR; p.sW = Kc;       // row offset: a multiple of 8 elements (16-byte aligned)
        p.out_f32 = parts + (size_t)tap * C; p.ldo = 9 * C; p.sO = (long long)N * 9 * C;
        e = launch_igemm(p, false, Z, st);
        if (e != hipSuccess) return e;
    }
    return launch_reduce_planes(parts, dw, (long long)N * 9 * C, Z, st);
}

// ---- Linear weight gradient dW [N][K] = dy^T a: the reduction runs over the M rows, the output is tiny -- as ONE GEMM it is 6 tiles on
// 256 CUs walking K' = M serially (0.2 ms each, a third of the training step).  Like the conv wgrad: row-major transposes dy^T [N][Mp],
// a^T [K][Mp] (zero padded), Z K-chunks as a batched launch of fp32 planes, fixed-order sum.
static void linear_wgrad_geom(long long M, int* pZ, long long* pKc) {
    int Z = (int)((M + 1023) / 1024); if (Z < 1) Z = 1; if (Z > 128) Z = 128;
    long long Kc = (M + Z - 1) / Z; Kc = (Kc + 63) & ~63LL;
    *pZ = Z; *pKc = Kc;
}
size_t linear_wgrad_scratch_bytes(long long M, int N, int K) {
    int Z; long long Kc; linear_wgrad_geom(M, &Z, &Kc);
    return (size_t)(N + K) * Kc * Z * 2 + 256 + (size_t)Z * N * K * 4;
}
hipError_t launch_linear_wgrad(const bf16_t* dy, const bf16_t* a, float* dw, long long M, int N, int K, char* scratch, const void* zero_page, hipStream_t st) {
    int Z; long long Kc; linear_wgrad_geom(M, &Z, &Kc);
    const long long Mp = Kc * Z;
    bf16_t* dyT = (bf16_t*)scratch; bf16_t* aT = dyT + (size_t)N * Mp;
    float* parts = (float*)(scratch + (((size_t)(N + K) * Mp * 2 + 255) & ~(size_t)255));
    hipError_t e = hipSuccess;
    if (Mp != M) { e = hipMemsetAsync(scratch, 0, (size_t)(N + K) * Mp * 2, st); if (e != hipSuccess) return e; }
    e = launch_transpose_bf16(dy, dyT, (int)M, N, st, 1, (int)Mp); if (e != hipSuccess) return e;
    e = launch_transpose_bf16(a, aT, (int)M, K, st, 1, (int)Mp); if (e != hipSuccess) return e;
    IgemmParams p{}; p.M = N; p.N = K; p.K = (int)Kc; p.alpha = 1.f; p.zero_page = zero_page;
    p.Hin = p.Win = p.Hout = p.Wout = 1; p.stride = 1; p.rows_per_sample = 1;
    p.A0 = dyT; p.C0 = (int)Kc; p.lda = (int)Mp; p.sA = Kc;
    p.W = aT; p.ldw = (int)Mp; p.sW = Kc;
    p.out_f32 = parts; p.ldo = K; p.sO = (long long)N * K;
    e = launch_igemm(p, false, Z, st); if (e != hipSuccess) return e;
    return launch_reduce_planes(parts, dw, (long long)N * K, Z, st);
}

// ---- GroupNorm (+SiLU) backward.  x [B, HW, C] bf16 (the forward INPUT), dy [B, HW, C] bf16 (gradient w.r.t. the OUTPUT).
// pass 1 (one block per (sample, group)): mean / rstd from x (recomputed: two reads instead of keeping forward state), then
//   S1 = sum dxh, S2 = sum dxh * xh over the group, with dz = dy * silu'(z) (z = xh * gamma + beta) and dxh = dz * gamma;
// pass 2 (elementwise): dx = rstd * (dxh - S1 / n - xh * S2 / n), and per-(sample, channel) partials of dgamma = sum dz * xh and
//   dbeta = sum dz, summed over samples by gn_bwd_affine_kernel in a fixed order.
struct GnBwdParams {
    const bf16_t* x; const bf16_t* dy; const float* gamma; const float* beta;
    int B, HW, C, groups, silu; float eps;
    float* stats;                   // [B, groups, 4]: mean, rstd, S1 / n, S2 / n
    bf16_t* dx; float* part_g; float* part_b;       // partials [B, C]
    float* dgamma; float* dbeta;
};
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(GnBwdParams p) {
    __shared__ float red[4];
    const int b = blockIdx.x / p.groups, g = blockIdx.x % p.groups;
    const int cg = p.C / p.groups; const long long n = (long long)p.HW * cg;
    const bf16_t* xb = p.x + (long long)b * p.HW * p.C + g * cg;
    const bf16_t* db = p.dy + (long long)b * p.HW * p.C + g * cg;
    float s = 0.f, ss = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) { const float v = bf2f(xb[(i / cg) * p.C + (i % cg)]); s += v; ss += v * v; }
    s = block_sum_256(s, red); ss = block_sum_256(ss, red);
    const float mean = s / (float)n, var = fmaxf(ss / (float)n - mean * mean, 0.f), rstd = rsqrtf(var + p.eps);
    float s1 = 0.f, s2 = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const int c = (int)(i % cg); const long long o = (i / cg) * p.C + c;
        const float xh = (bf2f(xb[o]) - mean) * rstd;
        const float ga = p.gamma[g * cg + c];
        float dz = bf2f(db[o]);
        if (p.silu) { const float z = xh * ga + p.beta[g * cg + c]; const float sg = 1.f / (1.f + __expf(-z)); dz *= sg * (1.f + z * (1.f - sg)); }
        const float dxh = dz * ga;
        s1 += dxh; s2 += dxh * xh;
    }
    s1 = block_sum_256(s1, red); s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) { float* st = p.stats + (long long)blockIdx.x * 4; st[0] = mean; st[1] = rstd; st[2] = s1 / (float)n; st[3] = s2 / (float)n; }
}
// one block per (sample, 64-channel chunk): walks the pixels, writes dx, accumulates the channel's dgamma / dbeta partials
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnBwdParams p) {
    __shared__ float rg[4][64], rb_[4][64];
    const int nch = (p.C + 63) / 64;
    const int b = blockIdx.x / nch, c = (blockIdx.x % nch) * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int cg = p.C / p.groups;
    float ag = 0.f, ab = 0.f;
    if (c < p.C) {
        const int g = c / cg;
        const float* st = p.stats + ((long long)b * p.groups + g) * 4;
        const float mean = st[0], rstd = st[1], m1 = st[2], m2 = st[3];
        const float ga = p.gamma[c], be = p.beta[c];
        for (int i = part; i < p.HW; i += 4) {
            const long long o = ((long long)b * p.HW + i) * p.C + c;
            const float xh = (bf2f(p.x[o]) - mean) * rstd;
            float dz = bf2f(p.dy[o]);
            if (p.silu) { const float z = xh * ga + be; const float sg = 1.f / (1.f + __expf(-z)); dz *= sg * (1.f + z * (1.f - sg)); }
            ag += dz * xh; ab += dz;
            p.dx[o] = f2bf(rstd * (dz * ga - m1 - xh * m2));
        }
    }
    rg[part][threadIdx.x & 63] = ag; rb_[part][threadIdx.x & 63] = ab;
    __syncthreads();
    if (part == 0 && c < p.C) {
        const int l = threadIdx.x;
        p.part_g[(long long)b * p.C + c] = (rg[0][l] + rg[1][l]) + (rg[2][l] + rg[3][l]);
        p.part_b[(long long)b * p.C + c] = (rb_[0][l] + rb_[1][l]) + (rb_[2][l] + rb_[3][l]);
    }
}
__global__ __launch_bounds__(256) void gn_bwd_affine_kernel(GnBwdParams p) {          // 64 channels x 4 sample runs per block
    __shared__ float rg[4][64], rb[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int per = (p.B + 3) / 4, b0 = q * per, b1 = min(p.B, b0 + per);
    float g = 0.f, bb = 0.f;
    if (c < p.C) for (int b = b0; b < b1; b++) { g += p.part_g[(long long)b * p.C + c]; bb += p.part_b[(long long)b * p.C + c]; }
    rg[q][threadIdx.x & 63] = g; rb[q][threadIdx.x & 63] = bb;
    __syncthreads();
    if (q == 0 && c < p.C) {
        const int l = threadIdx.x;
        p.dgamma[c] = (rg[0][l] + rg[1][l]) + (rg[2][l] + rg[3][l]); p.dbeta[c] = (rb[0][l] + rb[1][l]) + (rb[2][l] + rb[3][l]);
    }
}
// ---- vectorised GroupNorm backward (C % 8 == 0, C <= 2048): the round-3 kernels above walk a (sample, group) with 2-byte loads and
// a division per element (1.2 TB/s over the step's 61 layers).  Here every pass is the forward's 16-byte stream (thread = one
// 8-channel vector for the whole launch, VC x R threads):
//   1. launch_gn_stats (the forward's statistics kernel)                                   -> (sum, sumsq) per (sample, chunk, group)
//   2. gn_bwd_chan_kernel: per-(sample, chunk, CHANNEL) sums  a_c = sum dz, b_c = sum dz xh -> cpart
//   3. gn_bwd_group_kernel (one block per sample): chunk order sums; dbeta / dgamma partials ARE a_c / b_c; S1 = sum_c gamma_c a_c, S2 likewise
//   4. gn_bwd_apply2_kernel: dx = rstd (dz gamma - S1 / n - xh S2 / n)
//   5. gn_bwd_affine_kernel (as before): dgamma / dbeta = sample-order sums
struct GnBwd2 {
    const bf16_t* x; const bf16_t* dy; const float* gamma; const float* beta;
    int B, HW, C, groups, silu, nchunk; float eps;
    const float* fpart; float* cpart; float* stats; float* part_g; float* part_b; bf16_t* dx;
    const bf16_t* res;              // optional: dx = gradient + res (the residual path's gradient joins here instead of in a separate add pass)
};
__device__ __forceinline__ float silu_grad_f(float z) { const float sg = 1.f / (1.f + __expf(-z)); return sg * (1.f + z * (1.f - sg)); }
__device__ __forceinline__ void unpack8(const uint4 d, float* f) {
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int e = 0; e < 4; e++) { f[2 * e] = __uint_as_float(w[e] << 16); f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
__global__ void gn_bwd_chan_kernel(GnBwd2 p) {
    extern __shared__ float sm[];                        // [8][R][VC] float2
    __shared__ float gstat[64][2];
    const int C = p.C, VC = C >> 3, cg = C / p.groups;
    const int R = blockDim.x / VC;
    const int v = threadIdx.x % VC, rr = threadIdx.x / VC;
    const int b = blockIdx.y, chunk = blockIdx.x;
    if (threadIdx.x < p.groups) {
        double a = 0.0, q = 0.0;
        const float* pp = p.fpart + ((long long)b * p.nchunk * p.groups + threadIdx.x) * 2;
        for (int k = 0; k < p.nchunk; k++) { const float2 t = *(const float2*)(pp + (long long)k * p.groups * 2); a += t.x; q += t.y; }
        const double n = (double)cg * p.HW, mean = a / n;
        double var = q / n - mean * mean; if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        gstat[threadIdx.x][0] = (float)mean; gstat[threadIdx.x][1] = rstd;
        if (chunk == 0) { float* st = p.stats + ((long long)b * p.groups + threadIdx.x) * 4; st[0] = (float)mean; st[1] = rstd; }
    }
    __syncthreads();
    if (rr < R) {
        float fa[8], fb[8], ga[8], be[8], sa[8], sb[8];
        {
            int g = (v * 8) / cg, r = v * 8 - g * cg;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                fa[e] = gstat[g][1]; fb[e] = -gstat[g][0] * gstat[g][1];
                ga[e] = p.gamma[v * 8 + e]; be[e] = p.beta[v * 8 + e]; sa[e] = 0.f; sb[e] = 0.f;
                if (++r == cg) { r = 0; g++; }
            }
        }
        const int rows_per = (p.HW + p.nchunk - 1) / p.nchunk;
        const int r0 = chunk * rows_per, r1 = min(p.HW, r0 + rows_per);
        const bf16_t* xb = p.x + (long long)b * p.HW * C + v * 8; const bf16_t* db = p.dy + (long long)b * p.HW * C + v * 8;
        int row = r0 + rr;
        for (; row + R < r1; row += 2 * R) {
            const uint4 x0 = *(const uint4*)(xb + (long long)row * C), d0 = *(const uint4*)(db + (long long)row * C);
            const uint4 x1 = *(const uint4*)(xb + (long long)(row + R) * C), d1 = *(const uint4*)(db + (long long)(row + R) * C);
            float xf[8], df[8];
            unpack8(x0, xf); unpack8(d0, df);
#pragma unroll
            for (int e = 0; e < 8; e++) { const float xh = xf[e] * fa[e] + fb[e]; float dz = df[e]; if (p.silu) dz *= silu_grad_f(xh * ga[e] + be[e]); sa[e] += dz; sb[e] += dz * xh; }
            unpack8(x1, xf); unpack8(d1, df);
#pragma unroll
            for (int e = 0; e < 8; e++) { const float xh = xf[e] * fa[e] + fb[e]; float dz = df[e]; if (p.silu) dz *= silu_grad_f(xh * ga[e] + be[e]); sa[e] += dz; sb[e] += dz * xh; }
        }
        for (; row < r1; row += R) {
            float xf[8], df[8];
            unpack8(*(const uint4*)(xb + (long long)row * C), xf); unpack8(*(const uint4*)(db + (long long)row * C), df);
#pragma unroll
            for (int e = 0; e < 8; e++) { const float xh = xf[e] * fa[e] + fb[e]; float dz = df[e]; if (p.silu) dz *= silu_grad_f(xh * ga[e] + be[e]); sa[e] += dz; sb[e] += dz * xh; }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) *(float2*)(sm + ((e * R + rr) * VC + v) * 2) = make_float2(sa[e], sb[e]);
    }
    __syncthreads();
    float* out = p.cpart + ((long long)b * p.nchunk + chunk) * C * 2;
    for (int idx = threadIdx.x; idx < C; idx += blockDim.x) {
        const int e = idx / VC, vv = idx - e * VC;
        float a = 0.f, q = 0.f;
        for (int r = 0; r < R; r++) { const float2 t = *(const float2*)(sm + ((e * R + r) * VC + vv) * 2); a += t.x; q += t.y; }
        *(float2*)(out + (vv * 8 + e) * 2) = make_float2(a, q);
    }
}
__global__ __launch_bounds__(256) void gn_bwd_group_kernel(GnBwd2 p) {
    __shared__ float ch[2048 * 2];
    const int b = blockIdx.x, C = p.C, cg = C / p.groups;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, q = 0.f;
        const float* pp = p.cpart + ((long long)b * p.nchunk * C + c) * 2;
        for (int k = 0; k < p.nchunk; k++) { const float2 t = *(const float2*)(pp + (long long)k * C * 2); a += t.x; q += t.y; }
        const float ga = p.gamma[c];
        ch[c * 2] = a * ga; ch[c * 2 + 1] = q * ga;
        p.part_b[(long long)b * C + c] = a; p.part_g[(long long)b * C + c] = q;
    }
    __syncthreads();
    if (threadIdx.x < p.groups) {
        float s1 = 0.f, s2 = 0.f;
        for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; c++) { s1 += ch[c * 2]; s2 += ch[c * 2 + 1]; }
        const float n = (float)cg * (float)p.HW;
        float* st = p.stats + ((long long)b * p.groups + threadIdx.x) * 4;
        st[2] = s1 / n; st[3] = s2 / n;
    }
}
__global__ void gn_bwd_apply2_kernel(GnBwd2 p) {
    __shared__ float gs[64][4];
    const int C = p.C, VC = C >> 3, cg = C / p.groups;
    const int R = blockDim.x / VC;
    const int v = threadIdx.x % VC, rr = threadIdx.x / VC;
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < p.groups * 4; i += blockDim.x) gs[i >> 2][i & 3] = p.stats[(long long)b * p.groups * 4 + i];      // (a block may have fewer than groups * 4 threads: 192 at C = 1536)
    __syncthreads();
    if (rr >= R) return;
    float fa[8], fb[8], ga[8], be[8], c1[8], c2[8], c3[8];
    {
        int g = (v * 8) / cg, r = v * 8 - g * cg;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float mean = gs[g][0], rstd = gs[g][1];
            fa[e] = rstd; fb[e] = -mean * rstd; ga[e] = p.gamma[v * 8 + e]; be[e] = p.beta[v * 8 + e];
            c1[e] = rstd * ga[e]; c2[e] = rstd * gs[g][2]; c3[e] = rstd * gs[g][3];
            if (++r == cg) { r = 0; g++; }
        }
    }
    const int rows_per = (p.HW + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per, r1 = min(p.HW, r0 + rows_per);
    const long long base = (long long)b * p.HW * C + v * 8;
    for (int row = r0 + rr; row < r1; row += R) {
        float xf[8], df[8], o[8];
        unpack8(*(const uint4*)(p.x + base + (long long)row * C), xf); unpack8(*(const uint4*)(p.dy + base + (long long)row * C), df);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float xh = xf[e] * fa[e] + fb[e]; float dz = df[e];
            if (p.silu) dz *= silu_grad_f(xh * ga[e] + be[e]);
            o[e] = dz * c1[e] - c2[e] - xh * c3[e];
        }
        if (p.res) {
            float rf[8]; unpack8(*(const uint4*)(p.res + base + (long long)row * C), rf);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] += rf[e];
        }
        *(uint4*)(p.dx + base + (long long)row * C) = make_uint4(cvt_pk_bf16(o[0], o[1]), cvt_pk_bf16(o[2], o[3]), cvt_pk_bf16(o[4], o[5]), cvt_pk_bf16(o[6], o[7]));
    }
}
static int gn_bwd_nchunk(int B, int HW) { int n = 2048 / (B > 0 ? B : 1); if (n > HW / 32) n = HW / 32; if (n < 1) n = 1; if (n > 64) n = 64; return n; }
static bool gn_bwd_vec_ok(int C, int groups) {
    static const bool off = getenv("RDM_NO_GN_BWD_VEC") != nullptr;
    return !off && C % 8 == 0 && C <= 2048 && groups <= 64 && C % groups == 0;
}
size_t groupnorm_bwd_scratch_bytes(int B, int HW, int C, int groups) {
    const size_t nch = gn_bwd_nchunk(B, HW);
    return ((size_t)B * groups * 4 + 2 * (size_t)B * C + (size_t)B * nch * groups * 2 + (size_t)B * nch * C * 2) * sizeof(float) + 256;
}
hipError_t launch_groupnorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, const float* beta, int B, int HW, int C, int groups,
                                float eps, int silu, float* scratch /* groupnorm_bwd_scratch_bytes */, bf16_t* dx, float* dgamma, float* dbeta,
                                hipStream_t st, const bf16_t* residual) {
    GnBwdParams p{}; p.x = x; p.dy = dy; p.gamma = gamma; p.beta = beta; p.B = B; p.HW = HW; p.C = C; p.groups = groups; p.silu = silu; p.eps = eps;
    p.stats = scratch; p.part_g = scratch + (size_t)B * groups * 4; p.part_b = p.part_g + (size_t)B * C; p.dx = dx; p.dgamma = dgamma; p.dbeta = dbeta;
    if (gn_bwd_vec_ok(C, groups)) {
        GnBwd2 q{}; q.x = x; q.dy = dy; q.gamma = gamma; q.beta = beta; q.B = B; q.HW = HW; q.C = C; q.groups = groups; q.silu = silu; q.eps = eps;
        q.nchunk = gn_bwd_nchunk(B, HW);
        q.stats = p.stats; q.part_g = p.part_g; q.part_b = p.part_b; q.dx = dx; q.res = residual;
        float* fpart = p.part_b + (size_t)B * C;
        q.fpart = fpart; q.cpart = fpart + (size_t)B * q.nchunk * groups * 2;
        GnParams f{}; f.x0 = x; f.C0 = C; f.HW = HW; f.B = B; f.groups = groups; f.nchunk = q.nchunk; f.partial = fpart; f.eps = eps;
        hipError_t e = launch_gn_stats(f, st); if (e != hipSuccess) return e;
        const int VC = C / 8; int R = 256 / VC; if (R < 1) R = 1;
        const int threads = ((VC * R + 63) / 64) * 64;
        gn_bwd_chan_kernel<<<dim3(q.nchunk, B), threads, (size_t)R * C * 2 * sizeof(float), st>>>(q);
        gn_bwd_group_kernel<<<B, 256, 0, st>>>(q);
        int nblk = (2048 + B - 1) / B; if (nblk > HW / 8) nblk = HW / 8; if (nblk < 1) nblk = 1;
        gn_bwd_apply2_kernel<<<dim3(nblk, B), threads, 0, st>>>(q);
        gn_bwd_affine_kernel<<<(C + 63) / 64, 256, 0, st>>>(p);
        return hipGetLastError();
    }
    gn_bwd_stats_kernel<<<B * groups, 256, 0, st>>>(p);
    gn_bwd_apply_kernel<<<B * ((C + 63) / 64), 256, 0, st>>>(p);
    gn_bwd_affine_kernel<<<(C + 63) / 64, 256, 0, st>>>(p);
    if (residual) return launch_add_bf16(dx, residual, dx, (long long)B * HW * C, st);
    return hipGetLastError();
}

// ---- LayerNorm backward: x, dy [M, C] bf16; dx = rstd (dxh - mean(dxh) - xh mean(dxh xh)), dxh = dy * gamma.  One wave per row;
// dgamma / dbeta through per-block partials [nblocks, C] summed in a fixed order.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ gamma,
                                                     int M, int C, float eps, bf16_t* __restrict__ dx, float* __restrict__ part_g,
                                                     float* __restrict__ part_b, int rows_per_block) {
    extern __shared__ float acc[];                      // [4 waves][2][C]: a wave owns its slice -- no atomics, fixed summation order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = acc + (size_t)wave * 2 * C;
    for (int c = lane; c < 2 * C; c += 64) mine[c] = 0.f;
    const int r0 = blockIdx.x * rows_per_block;
    for (int r = r0 + wave; r < r0 + rows_per_block && r < M; r += 4) {
        const bf16_t* xr = x + (long long)r * C; const bf16_t* dr = dy + (long long)r * C;
        float s = 0.f, ss = 0.f;
        for (int c = lane; c < C; c += 64) { const float v = bf2f(xr[c]); s += v; ss += v * v; }
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
        const float mean = s / C, rstd = rsqrtf(fmaxf(ss / C - mean * mean, 0.f) + eps);
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) { const float xh = (bf2f(xr[c]) - mean) * rstd, dxh = bf2f(dr[c]) * gamma[c]; s1 += dxh; s2 += dxh * xh; }
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        s1 /= C; s2 /= C;
        for (int c = lane; c < C; c += 64) {
            const float xh = (bf2f(xr[c]) - mean) * rstd, d = bf2f(dr[c]);
            dx[(long long)r * C + c] = f2bf(rstd * (d * gamma[c] - s1 - xh * s2));
            mine[c] += d * xh; mine[C + c] += d;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        part_g[(long long)blockIdx.x * C + c] = (acc[c] + acc[2 * C + c]) + (acc[4 * C + c] + acc[6 * C + c]);
        part_b[(long long)blockIdx.x * C + c] = (acc[C + c] + acc[3 * C + c]) + (acc[5 * C + c] + acc[7 * C + c]);
    }
}
__global__ __launch_bounds__(256) void ln_bwd_affine_kernel(const float* part_g, const float* part_b, int nb, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float g = 0.f, b = 0.f;
    for (int i = 0; i < nb; i++) { g += part_g[(long long)i * C + c]; b += part_b[(long long)i * C + c]; }
    dgamma[c] = g; dbeta[c] = b;
}
// vectorised LayerNorm backward (C % 8 == 0, C <= 1024): one wave per row, the row lives in registers as NV 16-byte vectors per lane, so
// x and dy are read ONCE (the scalar kernel above walks each row three times with 2-byte loads: 0.6 TB/s); the dgamma / dbeta partials
// of a wave stay in registers across its rows and meet in LDS once per block.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ gamma,
                                                         int M, int C, float eps, bf16_t* __restrict__ dx, float* __restrict__ part_g,
                                                         float* __restrict__ part_b, int rows_per_block, const bf16_t* __restrict__ res) {
    extern __shared__ float acc[];                      // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, VC = C >> 3;
    float ga[NV][8], ag[NV][8], ab[NV][8];
    bool on[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        on[j] = lane + 64 * j < VC;
#pragma unroll
        for (int e = 0; e < 8; e++) { ga[j][e] = on[j] ? gamma[(lane + 64 * j) * 8 + e] : 0.f; ag[j][e] = 0.f; ab[j][e] = 0.f; }
    }
    const float invC = 1.f / (float)C;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int r = r0 + wave; r < r1; r += 4) {
        float xf[NV][8], df[NV][8];
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            if (on[j]) {
                unpack8(*(const uint4*)(x + (long long)r * C + (lane + 64 * j) * 8), xf[j]);
                unpack8(*(const uint4*)(dy + (long long)r * C + (lane + 64 * j) * 8), df[j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) { xf[j][e] = 0.f; df[j][e] = 0.f; }
            }
#pragma unroll
            for (int e = 0; e < 8; e++) { s += xf[j][e]; ss += xf[j][e] * xf[j][e]; }
        }
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
        const float mean = s * invC, rstd = rsqrtf(fmaxf(ss * invC - mean * mean, 0.f) + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; j++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xh = on[j] ? (xf[j][e] - mean) * rstd : 0.f, dxh = df[j][e] * ga[j][e];
                xf[j][e] = xh; s1 += dxh; s2 += dxh * xh;
            }
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        s1 *= invC; s2 *= invC;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                o[e] = rstd * (df[j][e] * ga[j][e] - s1 - xf[j][e] * s2);
                ag[j][e] += df[j][e] * xf[j][e]; ab[j][e] += df[j][e];
            }
            if (on[j] && res) {
                float rf[8]; unpack8(*(const uint4*)(res + (long long)r * C + (lane + 64 * j) * 8), rf);
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] += rf[e];
            }
            if (on[j])
                *(uint4*)(dx + (long long)r * C + (lane + 64 * j) * 8) = make_uint4(cvt_pk_bf16(o[0], o[1]), cvt_pk_bf16(o[2], o[3]), cvt_pk_bf16(o[4], o[5]), cvt_pk_bf16(o[6], o[7]));
        }
    }
    float* mine = acc + (size_t)wave * 2 * C;
#pragma unroll
    for (int j = 0; j < NV; j++)
        if (on[j])
#pragma unroll
            for (int e = 0; e < 8; e++) { mine[(lane + 64 * j) * 8 + e] = ag[j][e]; mine[C + (lane + 64 * j) * 8 + e] = ab[j][e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        part_g[(long long)blockIdx.x * C + c] = (acc[c] + acc[2 * C + c]) + (acc[4 * C + c] + acc[6 * C + c]);
        part_b[(long long)blockIdx.x * C + c] = (acc[C + c] + acc[3 * C + c]) + (acc[5 * C + c] + acc[7 * C + c]);
    }
}
// block partials added in sixteen fixed runs per channel (16 channels x 16 runs per block), the runs in a fixed tree (the
// one-thread-per-channel loop above is a serial walk over up to 1024 partials: 0.13 ms per call at the UNet's widths)
__global__ __launch_bounds__(256) void ln_bwd_affine4_kernel(const float* part_g, const float* part_b, int nb, int C, float* dgamma, float* dbeta) {
    __shared__ float rg[16][16], rb[16][16];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    const int per = (nb + 15) / 16, i0 = q * per, i1 = min(nb, i0 + per);
    float g = 0.f, b = 0.f;
    if (c < C) for (int i = i0; i < i1; i++) { g += part_g[(long long)i * C + c]; b += part_b[(long long)i * C + c]; }
    rg[q][cl] = g; rb[q][cl] = b;
    __syncthreads();
    if (q == 0 && c < C) { dgamma[c] = tree16(&rg[0][cl], 16); dbeta[c] = tree16(&rb[0][cl], 16); }
}
hipError_t launch_layernorm_bwd(const bf16_t* x, const bf16_t* dy, const float* gamma, int M, int C, float eps, float* scratch /* 2*nb*C */,
                                int* nb_out, bf16_t* dx, float* dgamma, float* dbeta, hipStream_t st, const bf16_t* residual) {
    const int rows_per_block = M >= 16384 ? 64 : 16;      // nb <= M / 16: the scratch contract of the callers (2 * ceil(M / 16) * C floats)
    const int nb = (M + rows_per_block - 1) / rows_per_block;
    if (nb_out) *nb_out = nb;
    float* pg = scratch; float* pb = scratch + (size_t)nb * C;
    static const bool novec = getenv("RDM_NO_LN_BWD_VEC") != nullptr;
    if (!novec && C % 8 == 0 && C <= 1024) {
        if (C <= 512) ln_bwd_vec_kernel<1><<<nb, 256, (size_t)8 * C * sizeof(float), st>>>(x, dy, gamma, M, C, eps, dx, pg, pb, rows_per_block, residual);
        else ln_bwd_vec_kernel<2><<<nb, 256, (size_t)8 * C * sizeof(float), st>>>(x, dy, gamma, M, C, eps, dx, pg, pb, rows_per_block, residual);
        ln_bwd_affine4_kernel<<<(C + 15) / 16, 256, 0, st>>>(pg, pb, nb, C, dgamma, dbeta);
        return hipGetLastError();
    }
    ln_bwd_kernel<<<nb, 256, (size_t)8 * C * sizeof(float), st>>>(x, dy, gamma, M, C, eps, dx, pg, pb, rows_per_block);
    ln_bwd_affine_kernel<<<(C + 255) / 256, 256, 0, st>>>(pg, pb, nb, C, dgamma, dbeta);
    if (residual) return launch_add_bf16(dx, residual, dx, (long long)M * C, st);
    return hipGetLastError();
}

// ---- out = a + b (bf16, 8 elements per thread): gradient accumulation where two paths meet (residual / skip connections)
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, long long n) {
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long long)gridDim.x * 256 * 8) {
        if (i + 8 <= n) {
            const uint4 x = *(const uint4*)(a + i), y = *(const uint4*)(b + i);
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w}; uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = cvt_pk_bf16(__uint_as_float(xs[e] << 16) + __uint_as_float(ys[e] << 16), __uint_as_float(xs[e] & 0xffff0000u) + __uint_as_float(ys[e] & 0xffff0000u));
            *(uint4*)(out + i) = make_uint4(o[0], o[1], o[2], o[3]);
        } else for (long long j = i; j < n; j++) out[j] = f2bf(bf2f(a[j]) + bf2f(b[j]));
    }
}
hipError_t launch_add_bf16(const bf16_t* a, const bf16_t* b, bf16_t* out, long long n, hipStream_t st) {
    long long g = (n / 8 + 255) / 256; if (g < 1) g = 1; if (g > 8192) g = 8192;
    add_bf16_kernel<<<dim3((unsigned)g), 256, 0, st>>>(a, b, out, n);
    return hipGetLastError();
}

// ---- GEGLU (ldm attention.py GEGLU: x, gate = proj(x).chunk(2, -1); return x * gelu(gate)) on a pre-activation matrix [M, 2F] bf16:
// forward h = a * gelu(g); backward da = dh * gelu(g), dg = dh * a * (Phi(g) + g phi(g)).  One thread = 8 outputs (16-byte accesses).
__device__ __forceinline__ float gelu_grad_f(float g) {       // d/dg [g Phi(g)] = Phi(g) + g phi(g)
    const float phi = 0.3989422804014327f * __expf(-0.5f * g * g);
    const float Phi = (g == 0.f) ? 0.5f : gelu_erf_f(g) / g;
    return Phi + g * phi;
}
template <bool BWD>
__global__ __launch_bounds__(256) void geglu_kernel(const bf16_t* __restrict__ pre, const bf16_t* __restrict__ dh, bf16_t* __restrict__ out, long long M, int F) {
    const long long nvec = M * (F / 8);
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const long long m = v / (F / 8); const int f0 = (int)(v - m * (F / 8)) * 8;
        const uint4 av = *(const uint4*)(pre + m * 2 * F + f0), gv = *(const uint4*)(pre + m * 2 * F + F + f0);
        const uint32_t as[4] = {av.x, av.y, av.z, av.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
        float a[8], g[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            a[2 * e] = __uint_as_float(as[e] << 16); a[2 * e + 1] = __uint_as_float(as[e] & 0xffff0000u);
            g[2 * e] = __uint_as_float(gs[e] << 16); g[2 * e + 1] = __uint_as_float(gs[e] & 0xffff0000u);
        }
        if (!BWD) {
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = cvt_pk_bf16(a[2 * e] * gelu_erf_f(g[2 * e]), a[2 * e + 1] * gelu_erf_f(g[2 * e + 1]));
            *(uint4*)(out + m * F + f0) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            const uint4 dv = *(const uint4*)(dh + m * F + f0);
            const uint32_t ds[4] = {dv.x, dv.y, dv.z, dv.w};
            uint32_t oa[4], og[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float d0 = __uint_as_float(ds[e] << 16), d1 = __uint_as_float(ds[e] & 0xffff0000u);
                oa[e] = cvt_pk_bf16(d0 * gelu_erf_f(g[2 * e]), d1 * gelu_erf_f(g[2 * e + 1]));
                og[e] = cvt_pk_bf16(d0 * a[2 * e] * gelu_grad_f(g[2 * e]), d1 * a[2 * e + 1] * gelu_grad_f(g[2 * e + 1]));
            }
            *(uint4*)(out + m * 2 * F + f0) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
            *(uint4*)(out + m * 2 * F + F + f0) = make_uint4(og[0], og[1], og[2], og[3]);
        }
    }
}
hipError_t launch_geglu(const bf16_t* pre, const bf16_t* dh, bf16_t* out, long long M, int F, hipStream_t st) {
    if (F % 8 != 0) return hipErrorInvalidValue;
    long long g = (M * (F / 8) + 255) / 256; if (g < 1) g = 1; if (g > 16384) g = 16384;
    if (dh) geglu_kernel<true><<<dim3((unsigned)g), 256, 0, st>>>(pre, dh, out, M, F);
    else geglu_kernel<false><<<dim3((unsigned)g), 256, 0, st>>>(pre, nullptr, out, M, F);
    return hipGetLastError();
}

// ---- attention backward helpers (unfused first version of the training step's attention gradient, SURVEY 8 f-4: the scores are
// materialised per (sample, head); the GEMMs run on the MFMA implicit-GEMM kernel as batches).
// heads: x [B, n, ldx] with head h in columns [h D, (h + 1) D)  <->  per-head matrices zero-padded to 64 columns (the GEMM's K unit):
//   mode 0: out [B H][n][64] = head slices;  mode 1: out [B H][64][n] = their transposes;  mode 2: x-layout [B, n, H D] <- in [B H][n][64]
__global__ __launch_bounds__(256) void heads_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int n, int H, int D, int ldx, int mode) {
    const long long total = (long long)B * H * n * 64;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (mode == 1) {        // i over [bh][d][row]
            const int row = (int)(i % n); const long long r = i / n; const int d = (int)(r % 64); const long long bh = r / 64;
            const int b = (int)(bh / H), h = (int)(bh % H);
            out[i] = d < D ? x[((long long)b * n + row) * ldx + h * D + d] : (bf16_t)0;
        } else {                // i over [bh][row][d]
            const int d = (int)(i % 64); const long long r = i / 64; const int row = (int)(r % n); const long long bh = r / n;
            const int b = (int)(bh / H), h = (int)(bh % H);
            if (mode == 0) out[i] = d < D ? x[((long long)b * n + row) * ldx + h * D + d] : (bf16_t)0;
            else if (d < D) out[((long long)b * n + row) * (H * D) + h * D + d] = x[i];
        }
    }
}
hipError_t launch_heads(const bf16_t* x, bf16_t* out, int B, int n, int H, int D, int ldx, int mode, hipStream_t st) {
    if (D < 1 || D > 64 || mode < 0 || mode > 2) return hipErrorInvalidValue;
    const long long total = (long long)B * H * n * 64;
    long long g = (total + 255) / 256; if (g > 16384) g = 16384;
    heads_kernel<<<dim3((unsigned)g), 256, 0, st>>>(x, out, B, n, H, D, ldx, mode);
    return hipGetLastError();
}
// softmax backward per row: dS = P (dP - sum_j P_j dP_j).  One wave per row (n a multiple of 4), the row's P and dP read twice.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ P, const float* __restrict__ dP, bf16_t* __restrict__ dS, long long rows, int n) {
    const int lane = threadIdx.x & 63;
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
        const bf16_t* pr = P + row * n; const float* dr = dP + row * n;
        float s = 0.f;
        for (int i = lane * 4; i < n; i += 256) {
            const float4 d = *(const float4*)(dr + i); const uint2 pv = *(const uint2*)(pr + i);
            s += __uint_as_float(pv.x << 16) * d.x + __uint_as_float(pv.x & 0xffff0000u) * d.y + __uint_as_float(pv.y << 16) * d.z + __uint_as_float(pv.y & 0xffff0000u) * d.w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        for (int i = lane * 4; i < n; i += 256) {
            const float4 d = *(const float4*)(dr + i); const uint2 pv = *(const uint2*)(pr + i);
            uint2 w;
            w.x = cvt_pk_bf16(__uint_as_float(pv.x << 16) * (d.x - s), __uint_as_float(pv.x & 0xffff0000u) * (d.y - s));
            w.y = cvt_pk_bf16(__uint_as_float(pv.y << 16) * (d.z - s), __uint_as_float(pv.y & 0xffff0000u) * (d.w - s));
            *(uint2*)(dS + row * n + i) = w;
        }
    }
}
hipError_t launch_softmax_bwd(const bf16_t* P, const float* dP, bf16_t* dS, long long rows, int n, hipStream_t st) {
    if (n % 4) return hipErrorInvalidValue;
    long long g = (rows + 3) / 4; if (g > 8192) g = 8192; if (g < 1) g = 1;
    softmax_bwd_kernel<<<dim3((unsigned)g), 256, 0, st>>>(P, dP, dS, rows, n);
    return hipGetLastError();
}

// ---- AdamW step (torch.optim.AdamW semantics, decoupled weight decay, bias correction; ldm configure_optimizers in
// rdm/models/diffusion/ddpm.py uses torch.optim.AdamW(params, lr)): fp32 master parameters and moments updated in place, optional
// bf16 working copy of the new parameters for the next forward.
__device__ __forceinline__ float adamw_elem(float pi, float gi, float& mi, float& vi, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    pi = pi * (1.f - lr * wd);
    mi = __fmaf_rn(b1, mi, (1.f - b1) * gi);
    vi = __fmaf_rn(b2, vi, (1.f - b2) * gi * gi);
    return pi - (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ pb, long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float mi = m[i], vi = v[i];
        const float pi = adamw_elem(p[i], g[i], mi, vi, lr, b1, b2, eps, wd, bc1, bc2);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (pb) pb[i] = f2bf(pi);
    }
}
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* pb, long long n, float lr, float b1, float b2, float eps, float wd, int step,
                        hipStream_t st) {
    if (step < 1) return hipErrorInvalidValue;
    long long grid = (n + 255) / 256; if (grid > 16384) grid = 16384; if (grid < 1) grid = 1;
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    adamw_kernel<<<dim3((unsigned)grid), 256, 0, st>>>(p, g, m, v, pb, n, lr, b1, b2, eps, wd, bc1, bc2);
    return hipGetLastError();
}

// Multi-tensor form: the UNet has 688 parameter tensors, 450 of them a few hundred elements (biases, norm affines) -- one launch per tensor is
// launch-bound (6 us each, 4.4 ms per step for 3 ms of memory traffic).  Up to 48 tensors per launch; a block owns 2048 consecutive elements
// of one tensor and finds it by scanning the block-offset table in the kernel arguments.  ema = 1: shadow (p) <- shadow - omd (shadow - param (g)).
struct MultiTensorArgs {
    int n, ema; int blk[49];
    float* p[48]; const float* g[48]; float* m[48]; float* v[48]; bf16_t* pb[48]; long long numel[48];
    float lr, b1, b2, eps, wd, bc1, bc2, omd;
};
__global__ __launch_bounds__(256) void multi_tensor_kernel(MultiTensorArgs a) {
    int t = 0;
    while (t + 1 < a.n && (int)blockIdx.x >= a.blk[t + 1]) t++;
    const long long base = (long long)((int)blockIdx.x - a.blk[t]) * 2048, n = a.numel[t];
    float* __restrict__ p = a.p[t]; const float* __restrict__ g = a.g[t];
    if (a.ema) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const long long i = base + u * 256 + threadIdx.x; if (i < n) { const float s = p[i]; p[i] = s - a.omd * (s - g[i]); } }
        return;
    }
    float* __restrict__ m = a.m[t]; float* __restrict__ v = a.v[t]; bf16_t* __restrict__ pb = a.pb[t];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i < n) {
            float mi = m[i], vi = v[i];
            const float pi = adamw_elem(p[i], g[i], mi, vi, a.lr, a.b1, a.b2, a.eps, a.wd, a.bc1, a.bc2);
            m[i] = mi; v[i] = vi; p[i] = pi;
            if (pb) pb[i] = f2bf(pi);
        }
    }
}
// host arrays of n pointers / element counts; ema: p = shadows, g = parameters, m / v / pb unused (may be null)
hipError_t launch_multi_tensor(int n, float* const* p, const float* const* g, float* const* m, float* const* v, void* const* pb, const long long* numel, int ema,
                               float lr, float b1, float b2, float eps, float wd, int step, float omd, hipStream_t st) {
    if (!ema && step < 1) return hipErrorInvalidValue;
    MultiTensorArgs a{};
    a.ema = ema; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd; a.omd = omd;
    a.bc1 = ema ? 1.f : 1.f - powf(b1, (float)step); a.bc2 = ema ? 1.f : 1.f - powf(b2, (float)step);
    int i = 0;
    while (i < n) {
        int k = 0; long long blocks = 0;
        while (i < n && k < 48 && blocks < (1 << 20)) {
            if (numel[i] < 1) return hipErrorInvalidValue;
            a.p[k] = p[i]; a.g[k] = g[i]; a.m[k] = m ? m[i] : nullptr; a.v[k] = v ? v[i] : nullptr; a.pb[k] = pb ? (bf16_t*)pb[i] : nullptr; a.numel[k] = numel[i];
            a.blk[k] = (int)blocks; blocks += (numel[i] + 2047) / 2048; k++; i++;
        }
        a.n = k; a.blk[k] = (int)blocks;
        multi_tensor_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(a);
        hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- SiLU on an fp32 vector (the time-embedding MLP: ldm TimestepEmbedSequential `nn.SiLU()` between / after the two Linear layers):
// dy null: out_bf16 = silu(x) (the next Linear's operand); else out_f32 = dy * silu'(x)
__global__ __launch_bounds__(256) void silu_kernel(const float* __restrict__ x, const float* __restrict__ dy, bf16_t* __restrict__ ob, float* __restrict__ of, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i], s = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        if (dy) of[i] = dy[i] * (s * (1.0f + v * (1.0f - s)));
        else ob[i] = f2bf(v * s);
    }
}
hipError_t launch_silu(const float* x, const float* dy, bf16_t* ob, float* of, long long n, hipStream_t st) {
    long long g = (n + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
    silu_kernel<<<dim3((unsigned)g), 256, 0, st>>>(x, dy, ob, of, n);
    return hipGetLastError();
}
// ---- 2 x 2 sum pooling of an NHWC bf16 tensor [B, 2H, 2W, C] -> [B, H, W, C]: the gradient of the nearest-neighbour 2x upsample in front
// of Upsample's conv (ldm openaimodel.py Upsample.forward: F.interpolate(scale_factor=2, mode="nearest") then conv).  8 channels per thread.
__global__ __launch_bounds__(256) void sumpool2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int C) {
    const int CV = C / 8;
    const long long total = (long long)B * H * W * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV); long long r = i / CV; const int xw = (int)(r % W); r /= W; const int y = (int)(r % H); const int b = (int)(r / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                const uint4 v = *(const uint4*)(x + (((long long)b * 2 * H + 2 * y + dy) * 2 * W + 2 * xw + dx) * C + cv * 8);
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; e++) { acc[2 * e] += __uint_as_float(w4[e] << 16); acc[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u); }
            }
        *(uint4*)(out + (((long long)b * H + y) * W + xw) * C + cv * 8) =
            make_uint4(cvt_pk_bf16(acc[0], acc[1]), cvt_pk_bf16(acc[2], acc[3]), cvt_pk_bf16(acc[4], acc[5]), cvt_pk_bf16(acc[6], acc[7]));
    }
}
hipError_t launch_sumpool2(const bf16_t* x, bf16_t* out, int B, int H, int W, int C, hipStream_t st) {
    if (C % 8) return hipErrorInvalidValue;
    long long g = ((long long)B * H * W * (C / 8) + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1;
    sumpool2_kernel<<<dim3((unsigned)g), 256, 0, st>>>(x, out, B, H, W, C);
    return hipGetLastError();
}

// ---- LitEma update (ldm/modules/ema.py forward(): shadow.sub_(one_minus_decay * (shadow - param))) on fp32 tensors
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ shadow, const float* __restrict__ p, long long n, float omd) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const float s = shadow[i]; shadow[i] = s - omd * (s - p[i]); }
}
hipError_t launch_ema(float* shadow, const float* p, long long n, float one_minus_decay, hipStream_t st) {
    long long g = (n + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1;
    ema_kernel<<<dim3((unsigned)g), 256, 0, st>>>(shadow, p, n, one_minus_decay);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------
// Fused attention backward for d_head = 32 (the UNet's self-attention; ldm CrossAttention, attention.py:52-72): no n x m score
// matrix in memory.  Three kernels over (sample, head):
//   attn_bwd_prep:  L[q] = log2-sum-exp of the scaled scores of row q (running max / sum like the forward), D[q] = dO[q] . O[q];
//   attn_bwd_dkv:   a wave owns 32 keys and walks the query tiles: S = Q K^T (rows = queries), P = exp2(S c - L), dP = dO V^T,
//                   dS = P (dP - D); dV^T += dO^T P, dK^T += Q^T dS -- P and dS stay in registers: the MFMA result layout (a lane = one
//                   key column, 16 query rows) IS the B-operand layout of the next MFMA once the query rows of a tile are loaded in the
//                   order pi (bits 2 and 3 of the position swapped), which makes a lane's 8 k-slots 8 CONSECUTIVE queries;
//   attn_bwd_dq:    the mirror image: a wave owns 32 queries and walks the key tiles with S^T = K Q^T, dQ^T += K^T dS^T.
// The operands read along the contraction index come from per-head transposed copies (Q^T, K^T, dO^T: [B H][32][n], made by
// heads_kernel mode 3), exactly like the forward's V^T.  All bf16 in, fp32 accumulation, bf16 out.
struct AttnBwdParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* dout;     // [B, n|m, C], head h = columns [32 h, 32 h + 32)
    const bf16_t* qT; const bf16_t* kT; const bf16_t* doT;                                       // [B H][32][n|m|n]
    float* L; float* D;                                                                          // [B H][n]
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    int B, H, n, m, C; float scale, scale_log2e;
};
__device__ __forceinline__ int attn_pi(int p) { return (p & ~0xc) | ((p & 4) << 1) | ((p & 8) >> 1); }
typedef __attribute__((ext_vector_type(16))) float ab_f32x16;

__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(AttnBwdParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wave) * 32, h = blockIdx.y, b = blockIdx.z;
    if (q0 >= p.n) return;
    const long long qrow = (long long)b * p.n + q0 + l31;
    bf16x8 qf[2];
    qf[0] = *(const bf16x8*)(p.q + qrow * p.C + h * 32 + hf * 8); qf[1] = *(const bf16x8*)(p.q + qrow * p.C + h * 32 + 16 + hf * 8);
    float mx = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < p.m; k0 += 32) {
        const long long krow = (long long)b * p.m + k0 + l31;
        const bf16x8 k0f = *(const bf16x8*)(p.k + krow * p.C + h * 32 + hf * 8), k1f = *(const bf16x8*)(p.k + krow * p.C + h * 32 + 16 + hf * 8);
        ab_f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.f;
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0f, qf[0], s, 0, 0, 0);         // S^T[key][query]: lane = query column
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1f, qf[1], s, 0, 0, 0);
        float t = s[0];
#pragma unroll
        for (int r = 1; r < 16; r++) t = fmaxf(t, s[r]);
        const float mn = fmaxf(mx, t * p.scale_log2e);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) ps += __builtin_amdgcn_exp2f(s[r] * p.scale_log2e - mn);
        l = l * __builtin_amdgcn_exp2f(mx - mn) + ps;
        mx = mn;
    }
    // the two half-waves hold the two 16-key halves of every tile: merge
    const float mo = __shfl_xor(mx, 32), lo = __shfl_xor(l, 32);
    const float mm = fmaxf(mx, mo);
    const float lt = l * __builtin_amdgcn_exp2f(mx - mm) + lo * __builtin_amdgcn_exp2f(mo - mm);
    // D = dO . O over the head's 32 channels (each half-wave 16 of them)
    const bf16x8 o0 = *(const bf16x8*)(p.o + qrow * p.C + h * 32 + hf * 16), o1 = *(const bf16x8*)(p.o + qrow * p.C + h * 32 + hf * 16 + 8);
    const bf16x8 d0 = *(const bf16x8*)(p.dout + qrow * p.C + h * 32 + hf * 16), d1 = *(const bf16x8*)(p.dout + qrow * p.C + h * 32 + hf * 16 + 8);
    float dd = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) dd += bf2f((bf16_t)o0[e]) * bf2f((bf16_t)d0[e]) + bf2f((bf16_t)o1[e]) * bf2f((bf16_t)d1[e]);
    dd += __shfl_xor(dd, 32);
    if (hf == 0) {
        const long long i = ((long long)b * p.H + h) * p.n + q0 + l31;
        p.L[i] = mm + __log2f(lt); p.D[i] = dd;
    }
}

// common: scores of a 32 x 32 tile -> P and dS as packed bf16 operand pairs (regs 0..7 -> first k-step, 8..15 -> second)
__device__ __forceinline__ void attn_bwd_pds(const ab_f32x16& s, const ab_f32x16& dp, const float (&Lr)[16], const float (&Dr)[16], float c,
                                             bf16x8 (&pf)[2], bf16x8 (&dsf)[2]) {
    union { bf16x8 v; uint32_t u[4]; } a[2], g[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = t * 8 + 2 * i;
            const float p0 = __builtin_amdgcn_exp2f(s[r] * c - Lr[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1] * c - Lr[r + 1]);
            a[t].u[i] = cvt_pk_bf16(p0, p1);
            g[t].u[i] = cvt_pk_bf16(p0 * (dp[r] - Dr[r]), p1 * (dp[r + 1] - Dr[r + 1]));
        }
    pf[0] = a[0].v; pf[1] = a[1].v; dsf[0] = g[0].v; dsf[1] = g[1].v;
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
    const int k0 = (blockIdx.x * 4 + wave) * 32, h = blockIdx.y, b = blockIdx.z;
    if (k0 >= p.m) return;
    const long long krow = (long long)b * p.m + k0 + l31;
    bf16x8 kf[2], vf[2];                                     // B operands: column = key l31
    kf[0] = *(const bf16x8*)(p.k + krow * p.C + h * 32 + hf * 8); kf[1] = *(const bf16x8*)(p.k + krow * p.C + h * 32 + 16 + hf * 8);
    vf[0] = *(const bf16x8*)(p.v + krow * p.C + h * 32 + hf * 8); vf[1] = *(const bf16x8*)(p.v + krow * p.C + h * 32 + 16 + hf * 8);
    ab_f32x16 dvT, dkT;                                      // [d rows][key column]
#pragma unroll
    for (int r = 0; r < 16; r++) { dvT[r] = 0.f; dkT[r] = 0.f; }
    const long long bh = (long long)b * p.H + h;
    const bf16_t* qT = p.qT + (bh * 32 + l31) * p.n; const bf16_t* doT = p.doT + (bh * 32 + l31) * p.n;     // A operands: row = channel l31
    const int pr = attn_pi(l31);
    for (int q0 = 0; q0 < p.n; q0 += 32) {
        const long long qrow = (long long)b * p.n + q0 + pr;                                  // A operands of S / dP: row position l31 holds query pi(l31)
        const bf16x8 qa0 = *(const bf16x8*)(p.q + qrow * p.C + h * 32 + hf * 8), qa1 = *(const bf16x8*)(p.q + qrow * p.C + h * 32 + 16 + hf * 8);
        const bf16x8 da0 = *(const bf16x8*)(p.dout + qrow * p.C + h * 32 + hf * 8), da1 = *(const bf16x8*)(p.dout + qrow * p.C + h * 32 + 16 + hf * 8);
        const bf16x8 qt0 = *(const bf16x8*)(qT + q0 + hf * 8), qt1 = *(const bf16x8*)(qT + q0 + 16 + hf * 8);
        const bf16x8 dt0 = *(const bf16x8*)(doT + q0 + hf * 8), dt1 = *(const bf16x8*)(doT + q0 + 16 + hf * 8);
        float Lr[16], Dr[16];                                // regs 0..7 <-> queries q0 + 8 hf + 0..7, regs 8..15 <-> q0 + 16 + 8 hf + 0..7
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                const float4 lv = *(const float4*)(p.L + bh * p.n + q0 + t * 16 + hf * 8 + e), dv4 = *(const float4*)(p.D + bh * p.n + q0 + t * 16 + hf * 8 + e);
                Lr[t * 8 + e] = lv.x; Lr[t * 8 + e + 1] = lv.y; Lr[t * 8 + e + 2] = lv.z; Lr[t * 8 + e + 3] = lv.w;
                Dr[t * 8 + e] = dv4.x; Dr[t * 8 + e + 1] = dv4.y; Dr[t * 8 + e + 2] = dv4.z; Dr[t * 8 + e + 3] = dv4.w;
            }
        ab_f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa0, kf[0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa1, kf[1], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da0, vf[0], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da1, vf[1], dp, 0, 0, 0);
        bf16x8 pf[2], dsf[2];
        attn_bwd_pds(s, dp, Lr, Dr, p.scale_log2e, pf, dsf);
        dvT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dt0, pf[0], dvT, 0, 0, 0);
        dvT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dt1, pf[1], dvT, 0, 0, 0);
        dkT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt0, dsf[0], dkT, 0, 0, 0);
        dkT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt1, dsf[1], dkT, 0, 0, 0);
    }
    // lane = key column l31; reg r = channel 8 (r >> 2) + 4 hf + (r & 3)
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint2 w;
        w.x = cvt_pk_bf16(dvT[g * 4 + 0], dvT[g * 4 + 1]); w.y = cvt_pk_bf16(dvT[g * 4 + 2], dvT[g * 4 + 3]);
        *(uint2*)(p.dv + krow * p.C + h * 32 + 8 * g + 4 * hf) = w;
        w.x = cvt_pk_bf16(dkT[g * 4 + 0] * p.scale, dkT[g * 4 + 1] * p.scale); w.y = cvt_pk_bf16(dkT[g * 4 + 2] * p.scale, dkT[g * 4 + 3] * p.scale);
        *(uint2*)(p.dk + krow * p.C + h * 32 + 8 * g + 4 * hf) = w;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wave) * 32, h = blockIdx.y, b = blockIdx.z;
    if (q0 >= p.n) return;
    const long long qrow = (long long)b * p.n + q0 + l31;
    bf16x8 qf[2], df[2];                                     // B operands: column = query l31
    qf[0] = *(const bf16x8*)(p.q + qrow * p.C + h * 32 + hf * 8); qf[1] = *(const bf16x8*)(p.q + qrow * p.C + h * 32 + 16 + hf * 8);
    df[0] = *(const bf16x8*)(p.dout + qrow * p.C + h * 32 + hf * 8); df[1] = *(const bf16x8*)(p.dout + qrow * p.C + h * 32 + 16 + hf * 8);
    const long long bh = (long long)b * p.H + h;
    const float Lq = p.L[bh * p.n + q0 + l31], Dq = p.D[bh * p.n + q0 + l31];
    float Lr[16], Dr[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { Lr[r] = Lq; Dr[r] = Dq; }
    ab_f32x16 dqT;
#pragma unroll
    for (int r = 0; r < 16; r++) dqT[r] = 0.f;
    const bf16_t* kT = p.kT + (bh * 32 + l31) * p.m;
    const int pr = attn_pi(l31);
    for (int k0 = 0; k0 < p.m; k0 += 32) {
        const long long krow = (long long)b * p.m + k0 + pr;
        const bf16x8 ka0 = *(const bf16x8*)(p.k + krow * p.C + h * 32 + hf * 8), ka1 = *(const bf16x8*)(p.k + krow * p.C + h * 32 + 16 + hf * 8);
        const bf16x8 va0 = *(const bf16x8*)(p.v + krow * p.C + h * 32 + hf * 8), va1 = *(const bf16x8*)(p.v + krow * p.C + h * 32 + 16 + hf * 8);
        const bf16x8 kt0 = *(const bf16x8*)(kT + k0 + hf * 8), kt1 = *(const bf16x8*)(kT + k0 + 16 + hf * 8);
        ab_f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qf[0], s, 0, 0, 0);          // S^T[key position][query]
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, qf[1], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0, df[0], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1, df[1], dp, 0, 0, 0);
        bf16x8 pf[2], dsf[2];
        attn_bwd_pds(s, dp, Lr, Dr, p.scale_log2e, pf, dsf);
        dqT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt0, dsf[0], dqT, 0, 0, 0);
        dqT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt1, dsf[1], dqT, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint2 w;
        w.x = cvt_pk_bf16(dqT[g * 4 + 0] * p.scale, dqT[g * 4 + 1] * p.scale); w.y = cvt_pk_bf16(dqT[g * 4 + 2] * p.scale, dqT[g * 4 + 3] * p.scale);
        *(uint2*)(p.dq + qrow * p.C + h * 32 + 8 * g + 4 * hf) = w;
    }
}

// per-head transposes without padding: x [B, n, C] -> out [B H][32][n]
__global__ __launch_bounds__(256) void head_transpose32_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int n, int H) {
    __shared__ bf16_t tile[32][33];
    const int h = blockIdx.y, b = blockIdx.z, r0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = (r0 + r < n) ? x[((long long)b * n + r0 + r) * (H * 32) + h * 32 + tx] : (bf16_t)0;
    __syncthreads();
    for (int d = ty; d < 32; d += 8) if (r0 + tx < n) out[(((long long)b * H + h) * 32 + d) * n + r0 + tx] = tile[tx][d];
}
size_t attn_bwd_scratch_bytes(int B, int H, int n, int m) { return ((size_t)B * H * 32 * (2 * (size_t)n + m)) * 2 + (size_t)B * H * n * 8 + 512; }
hipError_t launch_attention_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* o, const bf16_t* dout, int B, int n, int m, int H,
                                bf16_t* dq, bf16_t* dk, bf16_t* dv, char* scratch, hipStream_t st) {
    if (n % 32 || m % 32) return hipErrorInvalidValue;
    AttnBwdParams p{}; p.q = q; p.k = k; p.v = v; p.o = o; p.dout = dout; p.B = B; p.H = H; p.n = n; p.m = m; p.C = H * 32;
    p.scale = 1.0f / sqrtf(32.f); p.scale_log2e = p.scale * 1.4426950408889634f;
    bf16_t* qT = (bf16_t*)scratch; bf16_t* doT = qT + (size_t)B * H * 32 * n; bf16_t* kT = doT + (size_t)B * H * 32 * n;
    float* L = (float*)(((uintptr_t)(kT + (size_t)B * H * 32 * m) + 255) & ~(uintptr_t)255); float* D = L + (size_t)B * H * n;
    p.qT = qT; p.doT = doT; p.kT = kT; p.L = L; p.D = D; p.dq = dq; p.dk = dk; p.dv = dv;
    head_transpose32_kernel<<<dim3((n + 31) / 32, H, B), 256, 0, st>>>(q, qT, B, n, H);
    head_transpose32_kernel<<<dim3((n + 31) / 32, H, B), 256, 0, st>>>(dout, doT, B, n, H);
    head_transpose32_kernel<<<dim3((m + 31) / 32, H, B), 256, 0, st>>>(k, kT, B, m, H);
    attn_bwd_prep_kernel<<<dim3((n / 32 + 3) / 4, H, B), 256, 0, st>>>(p);
    attn_bwd_dkv_kernel<<<dim3((m / 32 + 3) / 4, H, B), 256, 0, st>>>(p);
    attn_bwd_dq_kernel<<<dim3((n / 32 + 3) / 4, H, B), 256, 0, st>>>(p);
    return hipGetLastError();
}


// ==================================================================================== training-step glue
// The elementwise pieces around the UNet in MinimalRETRODiffusion.shared_step -> forward -> ldm p_losses
// (rdm/models/diffusion/ddpm.py:390-443; ldm LatentDiffusion.q_sample / p_losses): noising, the squared-error loss and its gradient,
// the Bernoulli(p_uncond) conditioning switch, the per-sample bias gradient of the time-embedding rows, the data movement of the
// Downsample / Upsample gradients, and the scaling that turns an all-reduced sum into a mean.

// q_sample: x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps per sample, fp32 NCHW in; fp32 NCHW out (what the reference hands to the
// UNet) and / or the bf16 NHWC [B, H, W, cpad] operand of the native training forward (channels beyond C zero)
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const float* __restrict__ a,
                                                       const float* __restrict__ b, float* __restrict__ out, bf16_t* __restrict__ out_nhwc, int B, int C,
                                                       int HW, int cpad) {
    const long long n = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int bi = (int)(i / HW), r = (int)(i % HW);
        const float av = a[bi], bv = b[bi];
        for (int c = 0; c < cpad; c++) {
            float v = 0.f;
            if (c < C) {
                const long long j = ((long long)bi * C + c) * HW + r;
                v = av * x0[j] + bv * noise[j];
                if (out) out[j] = v;
            }
            if (out_nhwc) out_nhwc[i * cpad + c] = f2bf(v);
        }
    }
}
hipError_t launch_q_sample(const float* x0, const float* noise, const float* a, const float* b, float* out, bf16_t* out_nhwc, int B, int C, int HW, int cpad,
                           hipStream_t st) {
    long long g = ((long long)B * HW + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
    q_sample_kernel<<<dim3((unsigned)g), 256, 0, st>>>(x0, noise, a, b, out, out_nhwc, B, C, HW, cpad < C ? C : cpad);
    return hipGetLastError();
}

// ldm p_losses, l2 / eps parameterisation: se[b] = mean over (C, H, W) of (eps_theta - target)^2 (the per-sample `loss_simple` before
// the batch mean) and deps = coef[b] (eps_theta - target): the caller folds 2 / (C H W B) and the l_simple / elbo weights into coef.
// eps bf16 NHWC [B, H, W, ldc] (the first C channels count), target fp32 NCHW.  One block per sample, fixed-order reduction.
__global__ __launch_bounds__(256) void mse_loss_kernel(const bf16_t* __restrict__ eps, const float* __restrict__ target, const float* __restrict__ coef,
                                                       float* __restrict__ se, bf16_t* __restrict__ deps, int C, int HW, int ldc) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < HW; r += 256) {
        for (int c = 0; c < ldc; c++) {
            float d = 0.f;
            if (c < C) {
                d = bf2f(eps[((long long)b * HW + r) * ldc + c]) - target[((long long)b * C + c) * HW + r];
                s += d * d;
            }
            if (deps) deps[((long long)b * HW + r) * ldc + c] = f2bf(coef[b] * d);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) se[b] = red[0] / (float)((long long)C * HW);
}
hipError_t launch_mse_loss(const bf16_t* eps, const float* target, const float* coef, float* se, bf16_t* deps, int B, int C, int HW, int ldc, hipStream_t st) {
    mse_loss_kernel<<<dim3((unsigned)B), 256, 0, st>>>(eps, target, coef, se, deps, C, HW, ldc);
    return hipGetLastError();
}

// out[b, :] = mask[b] ? a[b, :] : x[b, :]   (torch.where(repeat(mask, 'b -> b 1 1'), uncond_signal, r), ddpm.py:393-396)
__global__ __launch_bounds__(256) void where_rows_kernel(const unsigned char* __restrict__ mask, const float* __restrict__ a, const float* __restrict__ x,
                                                         float* __restrict__ out, long long rows, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * n; i += (long long)gridDim.x * 256) out[i] = mask[i / n] ? a[i] : x[i];
}
hipError_t launch_where_rows(const unsigned char* mask, const float* a, const float* x, float* out, long long rows, long long n, hipStream_t st) {
    long long g = (rows * n + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
    where_rows_kernel<<<dim3((unsigned)g), 256, 0, st>>>(mask, a, x, out, rows, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ x, long long n, float s) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= s;
}
hipError_t launch_scale_f32(float* x, long long n, float s, hipStream_t st) {
    long long g = (n + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1;
    scale_f32_kernel<<<dim3((unsigned)g), 256, 0, st>>>(x, n, s);
    return hipGetLastError();
}

// per-sample column sums of a bf16 [B, HW, N] tensor -> bf16 [B, N] (the gradient of the time-embedding row a ResBlock adds to every
// pixel of a sample: h = conv(.) + emb_out[b]): one block per (sample, 64 columns), fixed-order tree over the pixels
// two stages like launch_colsum, per sample: 16-byte loads over (pixel chunk, column block, sample), then the chunk partials in chunk order
__global__ __launch_bounds__(256) void colsum_samples_part_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, int HW, int N, int CS, int VCB) {
    __shared__ float red[2048];
    const int RL = 256 / VCB;
    const int cv = threadIdx.x % VCB, rl = threadIdx.x / VCB, b = blockIdx.z;
    const int v = blockIdx.y * VCB + cv;
    const int r0 = blockIdx.x * CS, r1 = min(HW, r0 + CS);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < RL && v * 8 < N) {
        const bf16_t* xb = x + (long long)b * HW * N + v * 8;
        for (int m = r0 + rl; m < r1; m += RL) {
            float f[8]; unpack8(*(const uint4*)(xb + (long long)m * N), f);
#pragma unroll
            for (int e = 0; e < 8; e++) s[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) red[(e * RL + rl) * VCB + cv] = s[e];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < VCB * 8; idx += 256) {
        const int e = idx / VCB, vv = idx - e * VCB, col = (blockIdx.y * VCB + vv) * 8 + e;
        if (col < N) {
            float t = 0.f;
            for (int r = 0; r < RL; r++) t += red[(e * RL + r) * VCB + vv];
            part[((long long)b * gridDim.x + blockIdx.x) * N + col] = t;
        }
    }
}
__global__ __launch_bounds__(256) void colsum_samples_finish_kernel(const float* __restrict__ part, bf16_t* __restrict__ out, int nchunk, int N) {
    const int b = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
    if (col >= N) return;
    float t = 0.f;
    for (int c = 0; c < nchunk; c++) t += part[((long long)b * nchunk + c) * N + col];
    out[(long long)b * N + col] = f2bf(t);
}
__global__ __launch_bounds__(256) void colsum_samples_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int HW, int N) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float s = 0.f;
    if (col < N) for (int m = part; m < HW; m += 4) s += bf2f(x[((long long)b * HW + m) * N + col]);
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && col < N) out[(long long)b * N + col] = f2bf((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}
static int colsum_samples_nchunk(int B, int HW) { int n = 1024 / (B > 0 ? B : 1); if (n > HW / 64) n = HW / 64; if (n > 32) n = 32; if (n < 1) n = 1; return n; }
size_t colsum_samples_scratch_bytes(int B, int HW, int N) { return (size_t)B * colsum_samples_nchunk(B, HW) * N * sizeof(float) + 256; }
hipError_t launch_colsum_samples(const bf16_t* x, bf16_t* out, int B, int HW, int N, hipStream_t st, float* scratch) {
    if (!scratch || N % 8 || HW < 64) {
        colsum_samples_kernel<<<dim3((N + 63) / 64, B), 256, 0, st>>>(x, out, HW, N);
        return hipGetLastError();
    }
    const int nchunk = colsum_samples_nchunk(B, HW), CS = (HW + nchunk - 1) / nchunk;
    const int nv = N / 8, colblocks = (nv + 31) / 32, VCB = (nv + colblocks - 1) / colblocks;
    colsum_samples_part_kernel<<<dim3(nchunk, colblocks, B), 256, 0, st>>>(x, scratch, HW, N, CS, VCB);
    colsum_samples_finish_kernel<<<dim3((N + 255) / 256, B), 256, 0, st>>>(scratch, out, nchunk, N);
    return hipGetLastError();
}

// mode 0: zero insertion  out[b, 2y, 2x, :] = x[b, y, x, :], zeros elsewhere (a stride-2 conv's output gradient seen as a stride-1 one);
// mode 1: nearest-neighbour 2x copy out[b, Y, X, :] = x[b, Y / 2, X / 2, :] (what Upsample's fused conv read).  x [B, H, W, C], out [B, 2H, 2W, C], C % 8 == 0
__global__ __launch_bounds__(256) void expand2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int C, int mode) {
    const int cv = C / 8;
    const long long n = (long long)B * 2 * H * 2 * W * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cv); long long r = i / cv;
        const int X = (int)(r % (2 * W)); r /= 2 * W;
        const int Y = (int)(r % (2 * H)); const int b = (int)(r / (2 * H));
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (mode == 1 || ((X & 1) == 0 && (Y & 1) == 0)) v = *(const uint4*)(x + (((long long)b * H + (Y >> 1)) * W + (X >> 1)) * C + c * 8);
        *(uint4*)(out + i * 8) = v;
    }
}
hipError_t launch_expand2(const bf16_t* x, bf16_t* out, int B, int H, int W, int C, int mode, hipStream_t st) {
    if (C % 8) return hipErrorInvalidValue;
    long long g = ((long long)B * 4 * H * W * (C / 8) + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1;
    expand2_kernel<<<dim3((unsigned)g), 256, 0, st>>>(x, out, B, H, W, C, mode);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------
// Cross-attention backward for FEW keys (the UNet's attn2: the conditioning is a handful of retrieved-neighbour embeddings, d_head =
// 32, at most 32 keys): the partner of small_attention_kernel (attention.cpp forward).  Round 3 sent this through the generic unfused
// path -- per-head padded copies of q / dO (D 32 -> 64), four batched GEMMs against 64 padded keys, a score matrix, two transposes:
// 24 ms of a 125 ms step for 4 keys.  Here one thread owns a query row: it recomputes its m probabilities from K (LDS, fp32), forms
// dP = dO V^T, dS = P (dP - sum P dP), dQ = scale dS K, and leaves P / dS in LDS; the block then contracts them with its 256 rows of
// dO / q (dV = P^T dO, dK = scale dS^T q), one (which, key, channel) output per thread, and writes a partial per (sample, head, block).
// small_attention_bwd_finish_kernel adds the block partials in block order.
struct SmallAttnBwdParams {
    const bf16_t* q; int ldq; const bf16_t* k; const bf16_t* v; int ldkv; const bf16_t* dout; int ldo;
    bf16_t* dq; bf16_t* dk; bf16_t* dv; float* part;
    int nq, nkv, H, nblk; float scale;
};
__global__ __launch_bounds__(256) void small_attention_bwd_kernel(SmallAttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    const int m = p.nkv, t = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    float* Ks = (float*)smem_; float* Vs = Ks + m * 32;
    float* Ps = Vs + m * 32; float* Ds = Ps + m * 256;                    // [m][256]: probabilities, then dS
    bf16_t* qs = (bf16_t*)(Ds + m * 256); bf16_t* dos = qs + 256 * 32;     // [256][32] bf16 rows of q and dO
    for (int i = t; i < m * 4; i += 256) {
        const int j = i >> 2, c = (i & 3) * 8;
        float kf[8], vf[8];
        unpack8(*(const uint4*)(p.k + ((long long)b * m + j) * p.ldkv + h * 32 + c), kf);
        unpack8(*(const uint4*)(p.v + ((long long)b * m + j) * p.ldkv + h * 32 + c), vf);
#pragma unroll
        for (int e = 0; e < 8; e++) { Ks[j * 32 + c + e] = kf[e]; Vs[j * 32 + c + e] = vf[e]; }
    }
    const int qi = blockIdx.x * 256 + t;
    const bool on = qi < p.nq;
    float qf[32], df[32];
#pragma unroll
    for (int c = 0; c < 32; c += 8) {
        uint4 a = make_uint4(0, 0, 0, 0), d = a;
        if (on) {
            a = *(const uint4*)(p.q + ((long long)b * p.nq + qi) * p.ldq + h * 32 + c);
            d = *(const uint4*)(p.dout + ((long long)b * p.nq + qi) * p.ldo + h * 32 + c);
        }
        *(uint4*)(qs + t * 32 + c) = a; *(uint4*)(dos + t * 32 + c) = d;
        unpack8(a, qf + c); unpack8(d, df + c);
    }
    __syncthreads();
    // probabilities of this row
    float mx = -INFINITY;
    for (int j = 0; j < m; j++) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 32; d++) s += qf[d] * Ks[j * 32 + d];
        s *= p.scale;
        Ps[j * 256 + t] = s; mx = fmaxf(mx, s);
    }
    float l = 0.f;
    for (int j = 0; j < m; j++) { const float e = __expf(Ps[j * 256 + t] - mx); Ps[j * 256 + t] = e; l += e; }
    const float inv = on ? 1.f / l : 0.f;
    float delta = 0.f;
    for (int j = 0; j < m; j++) {
        const float pj = Ps[j * 256 + t] * inv;
        float dp = 0.f;
#pragma unroll
        for (int d = 0; d < 32; d++) dp += df[d] * Vs[j * 32 + d];
        Ps[j * 256 + t] = pj; Ds[j * 256 + t] = dp; delta += pj * dp;
    }
    float dq[32];
#pragma unroll
    for (int d = 0; d < 32; d++) dq[d] = 0.f;
    for (int j = 0; j < m; j++) {
        const float ds = Ps[j * 256 + t] * (Ds[j * 256 + t] - delta);
        Ds[j * 256 + t] = ds;
#pragma unroll
        for (int d = 0; d < 32; d++) dq[d] += ds * Ks[j * 32 + d];
    }
    if (on) {
        bf16_t* op = p.dq + ((long long)b * p.nq + qi) * (p.H * 32) + h * 32;
#pragma unroll
        for (int c = 0; c < 32; c += 8)
            *(uint4*)(op + c) = make_uint4(cvt_pk_bf16(dq[c] * p.scale, dq[c + 1] * p.scale), cvt_pk_bf16(dq[c + 2] * p.scale, dq[c + 3] * p.scale),
                                           cvt_pk_bf16(dq[c + 4] * p.scale, dq[c + 5] * p.scale), cvt_pk_bf16(dq[c + 6] * p.scale, dq[c + 7] * p.scale));
    }
    __syncthreads();
    // dK[j][d] = scale sum_t dS[j][t] q[t][d],  dV[j][d] = sum_t P[j][t] dO[t][d]: one output per thread and pass, rows in thread order
    float* part = p.part + (((long long)(b * p.H + h) * p.nblk + blockIdx.x) * 2) * m * 32;
    for (int o = t; o < 2 * m * 32; o += 256) {
        const int which = o / (m * 32), jd = o - which * m * 32, j = jd >> 5, d = jd & 31;
        const float* A = (which ? Ps : Ds) + j * 256;
        const bf16_t* X = (which ? dos : qs) + d;
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < 256; r++) acc += A[r] * bf2f(X[r * 32]);
        part[o] = which ? acc : acc * p.scale;
    }
}
__global__ __launch_bounds__(256) void small_attention_bwd_finish_kernel(SmallAttnBwdParams p) {
    const int m = p.nkv, h = blockIdx.x, b = blockIdx.y;
    const float* part = p.part + (long long)(b * p.H + h) * p.nblk * 2 * m * 32;
    for (int o = threadIdx.x; o < 2 * m * 32; o += 256) {
        float acc = 0.f;
        for (int i = 0; i < p.nblk; i++) acc += part[(long long)i * 2 * m * 32 + o];
        const int which = o / (m * 32), jd = o - which * m * 32, j = jd >> 5, d = jd & 31;
        (which ? p.dv : p.dk)[((long long)b * m + j) * (p.H * 32) + h * 32 + d] = f2bf(acc);
    }
}
size_t small_attention_bwd_scratch_bytes(int B, int heads, int nq, int nkv) { return (size_t)B * heads * ((nq + 255) / 256) * 2 * nkv * 32 * sizeof(float) + 256; }
// q [B, nq, ldq], k / v [B, nkv, ldkv], dout [B, nq, ldo] (head h = columns [32 h, 32 h + 32)) -> dq [B, nq, 32 heads], dk / dv [B, nkv, 32 heads] bf16
hipError_t launch_small_attention_bwd(const bf16_t* q, int ldq, const bf16_t* k, const bf16_t* v, int ldkv, const bf16_t* dout, int ldo, int B, int nq, int nkv,
                                      int heads, float scale, bf16_t* dq, bf16_t* dk, bf16_t* dv, char* scratch, hipStream_t st) {
    if (nkv < 1 || nkv > 32 || ldq % 8 || ldkv % 8 || ldo % 8) return hipErrorInvalidValue;
    SmallAttnBwdParams p{}; p.q = q; p.ldq = ldq; p.k = k; p.v = v; p.ldkv = ldkv; p.dout = dout; p.ldo = ldo; p.dq = dq; p.dk = dk; p.dv = dv;
    p.part = (float*)scratch; p.nq = nq; p.nkv = nkv; p.H = heads; p.nblk = (nq + 255) / 256; p.scale = scale;
    const size_t smem = (size_t)nkv * 64 * 4 + (size_t)nkv * 512 * 4 + 2 * 256 * 32 * 2;
    static bool attr[RDM_MAX_DEVICES] = {};
    const int dev = rdm_cur_device();
    if (!attr[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)small_attention_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
        if (e != hipSuccess) return e;
        attr[dev] = true;
    }
    small_attention_bwd_kernel<<<dim3(p.nblk, heads, B), 256, smem, st>>>(p);
    small_attention_bwd_finish_kernel<<<dim3(heads, B), 256, 0, st>>>(p);
    return hipGetLastError();
}

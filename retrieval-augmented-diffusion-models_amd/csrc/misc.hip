// Small / bandwidth-bound kernels around the GEMM core (gfx950).
#include "kernels.h"

// ------------------------------------------------------------------ stem conv: few input channels
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
// acc.xy += p[HI] * w.xy: the scalar factor is broadcast out of one half of a register pair through op_sel (hipcc materialises a
// duplicated pair per factor instead: twice the registers for the 54 input taps a lane holds)
template <int HI> __device__ __forceinline__ void pk_fma_bcast(f2_t& acc, const f2_t p, const f2_t w) {
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(p), "v"(w));
    else    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(p), "v"(w));
}
// 3x3 pad-1 conv, input NCHW f32 [B,Cin,H,W] (Cin <= 4), output NHWC bf16 [B,H,W,Cout].
// Replaces input_blocks.0 conv (openaimodel.py:149) and the VQ decoder conv_in.  K = 9*Cin = 27 is
// far too small for MFMA: VALU, weights in LDS as [tap*Cin][Cout].
// OCT > 0 (Cout % (8 OCT) == 0): a lane's 16-byte result of one channel octet lands 2*Cout bytes from its neighbour's, so a direct store
// touches 64 lines for 1 KB; instead every OCT octets go through a per-wave LDS tile [64 pixels][16 OCT + 16 B] and leave as contiguous
// 16 OCT-byte runs (OCT lanes per pixel).  Same values, same rounding.  Measured (B = 64, 64 x 64, 3 -> 192): direct 123 us, OCT = 4 105 us,
// OCT = 8 162 us (its 95 KB of LDS leave one block per CU): the stores were never the bound -- the tap loop was (a runtime trip count put a
// branch between every tap's LDS reads and its FMAs, nothing could be hoisted): CIN is a template parameter now.
template <int OCT, int CIN>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* x, const float* w /*[Cout][Cin][3][3]*/,
                                                      const float* bias, bf16_t* out, int B, int Cin, int H, int W,
                                                      int Cout) {
    // thread = one output pixel: its 9*Cin input taps live in registers, every thread of the block reads the same
    // weight vector at the same time (LDS broadcast), 16-byte stores of 8 output channels.
    extern __shared__ __attribute__((aligned(16))) float ws[];   // [CIN*9][Cout] (rows of channels >= Cin zero) + bias[Cout] (+ OCT: 2 x 4 waves x 64 x (16 OCT + 16) B)
    constexpr int K = CIN * 9;
    const int Kw = Cin * 9;
    const int lane = threadIdx.x & 63;
    constexpr int TS = 16 * OCT + 16, CM = 8 * OCT - 8;      // tile row stride (bytes); channel mask of an octet inside its group
    char* tile = (char*)(ws + (K + 1) * Cout) + (threadIdx.x >> 6) * (64 * TS);
    for (int i = threadIdx.x; i < K * Cout; i += 256) {
        const int co = i / K, k = i % K;              // w index = co*Kw + (ci*9 + tap)
        ws[k * Cout + co] = k < Kw ? w[co * Kw + k] : 0.f;
    }
    for (int i = threadIdx.x; i < Cout; i += 256) ws[K * Cout + i] = bias[i];
    __syncthreads();
    const int b = blockIdx.y, npix = H * W;
    // two pixels per thread: every broadcast weight read from LDS feeds 16 FMAs instead of 8 (the LDS reads, not the FMAs, bound
    // the one-pixel version)
    for (int lp0 = blockIdx.x * 512 + threadIdx.x; lp0 < npix; lp0 += gridDim.x * 512) {
        f2_t patch[2][(K + 1) / 2];                   // tap k of pixel q: half k & 1 of patch[q][k / 2]
        const float* xb = x + (long long)b * Cin * npix;       // uniform base + 32-bit lane offsets: one address register per load
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int lp = lp0 + q * 256;
            const int xw = lp % W, yh = lp / W;
            const bool inb = lp < npix;
            const bool rv[3] = {inb && yh > 0, inb, inb && yh < H - 1}, cv[3] = {xw > 0, true, xw < W - 1};
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const bool ok = rv[t / 3] && cv[t % 3] && ci < Cin;
                    const int idx = ok ? lp + ci * npix + (t / 3 - 1) * W + (t % 3 - 1) : 0;
                    const float v = xb[idx];
                    if ((ci * 9 + t) & 1) patch[q][(ci * 9 + t) / 2].y = ok ? v : 0.f; else patch[q][(ci * 9 + t) / 2].x = ok ? v : 0.f;
                }
            }
        }
#pragma unroll 1
        for (int v8 = 0; v8 < Cout; v8 += 8) {
            // packed fp32 FMAs (two channels per instruction), the next tap's weights in flight while this tap's FMAs issue
            f2_t acc2[2][4];
            {
                const f4_t b0 = *(const f4_t*)(ws + K * Cout + v8), b1 = *(const f4_t*)(ws + K * Cout + v8 + 4);
#pragma unroll
                for (int q = 0; q < 2; q++) { acc2[q][0] = b0.xy; acc2[q][1] = b0.zw; acc2[q][2] = b1.xy; acc2[q][3] = b1.zw; }
            }
            f4_t w0 = *(const f4_t*)(ws + v8), w1 = *(const f4_t*)(ws + v8 + 4);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int kn = k + 1 < K ? k + 1 : k;
                const f4_t n0 = *(const f4_t*)(ws + kn * Cout + v8), n1 = *(const f4_t*)(ws + kn * Cout + v8 + 4);
                __builtin_amdgcn_sched_barrier(0);       // keep ONE tap of weights in flight: left alone, the scheduler hoists all 27 taps' reads
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const f2_t pp = patch[q][k / 2];
                    if (k & 1) { pk_fma_bcast<1>(acc2[q][0], pp, w0.xy); pk_fma_bcast<1>(acc2[q][1], pp, w0.zw); pk_fma_bcast<1>(acc2[q][2], pp, w1.xy); pk_fma_bcast<1>(acc2[q][3], pp, w1.zw); }
                    else       { pk_fma_bcast<0>(acc2[q][0], pp, w0.xy); pk_fma_bcast<0>(acc2[q][1], pp, w0.zw); pk_fma_bcast<0>(acc2[q][2], pp, w1.xy); pk_fma_bcast<0>(acc2[q][3], pp, w1.zw); }
                }
                __builtin_amdgcn_sched_barrier(0);
                w0 = n0; w1 = n1;
            }
            float acc[2][8];
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) { acc[q][2 * e] = acc2[q][e].x; acc[q][2 * e + 1] = acc2[q][e].y; }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int lp = lp0 + q * 256;
                const uint4 r = make_uint4(cvt_pk_bf16(acc[q][0], acc[q][1]), cvt_pk_bf16(acc[q][2], acc[q][3]), cvt_pk_bf16(acc[q][4], acc[q][5]), cvt_pk_bf16(acc[q][6], acc[q][7]));
                if (OCT == 0) {
                    if (lp < npix) *(uint4*)(out + ((long long)b * npix + lp) * Cout + v8) = r;
                } else {
                    *(uint4*)(tile + q * (64 * TS * 4) + lane * TS + (v8 & CM) * 2) = r;
                }
            }
            if (OCT > 0 && (v8 & CM) == CM) {                 // OCT octets staged: 64 pixels x 16 OCT bytes per q
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int wave_lp0 = lp0 - lane + q * 256;
#pragma unroll
                    for (int j = 0; j < (OCT > 0 ? OCT : 1); j++) {
                        constexpr int PPI = OCT > 0 ? 64 / OCT : 1;             // pixels per store instruction
                        const int pl = j * PPI + lane / (OCT > 0 ? OCT : 1), oc = lane % (OCT > 0 ? OCT : 1);
                        const uint4 r = *(const uint4*)(tile + q * (64 * TS * 4) + pl * TS + oc * 16);
                        if (wave_lp0 + pl < npix) *(uint4*)(out + ((long long)b * npix + wave_lp0 + pl) * Cout + (v8 - CM) + oc * 8) = r;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

hipError_t launch_conv_in(const float* x, const float* w, const float* bias, bf16_t* out, int B, int Cin, int H, int W,
                          int Cout, hipStream_t st) {
    if (Cout % 8) return hipErrorInvalidValue;
    if (Cin > 4 || Cin < 1 || Cout % 8) return hipErrorInvalidValue;
    const int cin_t = Cin <= 3 ? 3 : 4;
    const size_t sm = (size_t)(cin_t * 9 + 1) * Cout * sizeof(float);
    if (sm > 96 * 1024) return hipErrorInvalidValue;
    int grid = (H * W + 511) / 512; const int cap_in = (2048 + B - 1) / B; if (grid > cap_in) grid = cap_in; if (grid < 1) grid = 1;
    static const int oct_env = getenv("RDM_CONVIN_OCT") ? atoi(getenv("RDM_CONVIN_OCT")) : 4;     // 0: direct 16-byte stores
    static bool attr_dev[RDM_MAX_DEVICES] = {false};           // per device, like every other launcher (one process may hold contexts on several GPUs)
    bool& attr_set = attr_dev[rdm_cur_device()];
    if (!attr_set) {
        const void* fns[] = {(const void*)conv_in_kernel<4, 3>, (const void*)conv_in_kernel<4, 4>, (const void*)conv_in_kernel<0, 3>, (const void*)conv_in_kernel<0, 4>};
        for (const void* f : fns) { hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (e != hipSuccess) return e; }
        attr_set = true;
    }
    const dim3 g(grid, B);
    const size_t sm4 = sm + 2 * 4 * 64 * 80;
    if (oct_env >= 4 && Cout % 32 == 0) {
        if (cin_t == 3) conv_in_kernel<4, 3><<<g, 256, sm4, st>>>(x, w, bias, out, B, Cin, H, W, Cout);
        else            conv_in_kernel<4, 4><<<g, 256, sm4, st>>>(x, w, bias, out, B, Cin, H, W, Cout);
    } else {
        if (cin_t == 3) conv_in_kernel<0, 3><<<g, 256, sm, st>>>(x, w, bias, out, B, Cin, H, W, Cout);
        else            conv_in_kernel<0, 4><<<g, 256, sm, st>>>(x, w, bias, out, B, Cin, H, W, Cout);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ head conv: few output channels
// 3x3 pad-1 conv, input NHWC bf16 [B,H,W,Cin], output NCHW f32 [B,Cout,H,W], Cout <= 4.
// Replaces UNet `out` conv (openaimodel.py:310) and the VQ decoder conv_out. One wave per pixel:
// lanes split the (tap, 8-channel vector) items, butterfly-reduce, lane 0 stores.
__global__ __launch_bounds__(256) void conv_out_kernel(const bf16_t* x, const float* w /*[Cout][Cin][3][3]*/,
                                                       const float* bias, float* out, int B, int H, int W, int Cin,
                                                       int Cout) {
    // block = 32 pixels x 8 channel groups (lane & 7): a pixel's channel row is read by 8 neighbouring lanes in
    // 16-byte pieces (coalesced), partial sums are combined with three xor-shuffles inside the 8-lane group.
    extern __shared__ float ws[];   // [Cout][9][Cin]
    for (int i = threadIdx.x; i < Cout * Cin * 9; i += 256) {
        const int co = i / (Cin * 9), r = i % (Cin * 9), ci = r / 9, t = r % 9;
        ws[(co * 9 + t) * Cin + ci] = w[i];
    }
    __syncthreads();
    const int g = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int VPG = Cin / 64;                            // 16-byte vectors per channel group per tap
    const int b = blockIdx.y, npix = H * W;
    for (int p0 = blockIdx.x * 32; p0 < npix; p0 += gridDim.x * 32) {
        const int pix = p0 + pl;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (pix < npix) {
            const int xw = pix % W, yh = pix / W;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int iy = yh + t / 3 - 1, ix = xw + t % 3 - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const bf16_t* xp = x + (((long long)b * H + iy) * W + ix) * Cin;
                for (int v = 0; v < VPG; v++) {
                    const int c = (v * 8 + g) * 8;        // consecutive lanes -> consecutive 16-byte pieces
                    const bf16x8 d = *(const bf16x8*)(xp + c);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) f[e] = bf2f((bf16_t)d[e]);
#pragma unroll
                    for (int co = 0; co < 4; co++) {
                        if (co < Cout) {
                            const float* wr = ws + (co * 9 + t) * Cin + c;
                            const float4 w0 = *(const float4*)wr, w1 = *(const float4*)(wr + 4);
                            acc[co] += f[0] * w0.x + f[1] * w0.y + f[2] * w0.z + f[3] * w0.w + f[4] * w1.x + f[5] * w1.y + f[6] * w1.z + f[7] * w1.w;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < 4; co++) {
            acc[co] += __shfl_xor(acc[co], 1); acc[co] += __shfl_xor(acc[co], 2); acc[co] += __shfl_xor(acc[co], 4);
        }
        if (g == 0 && pix < npix) {
            const int xw = pix % W, yh = pix / W;
#pragma unroll
            for (int co = 0; co < 4; co++)
                if (co < Cout) out[((long long)(b * Cout + co) * H + yh) * W + xw] = acc[co] + bias[co];
        }
    }
}

hipError_t launch_conv_out(const bf16_t* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cin,
                           int Cout, hipStream_t st) {
    if (Cin % 64 || Cout > 4) return hipErrorInvalidValue;
    const size_t sm = (size_t)Cout * 9 * Cin * sizeof(float);
    if (sm > 64 * 1024) return hipErrorInvalidValue;
    const int npix = H * W;
    int grid = (npix + 31) / 32; const int cap_out = (4096 + B - 1) / B; if (grid > cap_out) grid = cap_out; if (grid < 1) grid = 1;
    conv_out_kernel<<<dim3(grid, B), 256, sm, st>>>(x, w, bias, out, B, H, W, Cin, Cout);
    return hipGetLastError();
}

// ------------------------------------------------------------------ head conv on the matrix pipe, GroupNorm-apply + SiLU folded in
// The VALU head conv above sits on the 9x tap re-read through L2 -> L1 (round 2: 26 B/clk/CU), and the GroupNorm-apply in front of it
// writes and re-reads the whole tensor.  Here a block owns a 32-pixel-wide strip of one image and walks down its rows, two at a time:
// raw rows arrive once (16-byte pieces, coalesced), are normalised (the per-channel a x + b of gn_apply_kernel, + SiLU, rounded to bf16
// as the standalone pass rounds) on their way into a 6-row LDS ring (34 pixels x C channels per row, pixel stride 2C + 16 bytes:
// conflict-free fragment reads), the rows of the NEXT pair being requested before the current pair's MFMAs.  The 3x3 conv is 9 C/32
// MFMA 16x16x32 steps per 16-pixel tile (a wave: one tile of one of the two rows): A = 16 pixels x 32 channels of a tap from the ring,
// B = the tap's weights from LDS, fragment-ordered, N padded to 16 -- and the padding is used: columns [0, Cout) hold the bf16 high
// part of the fp32 weights, [Cout, 2 Cout) the low part, so one MFMA carries ~16 weight bits.  Output NCHW fp32.
size_t head_conv_wp_bytes(int C) { return (size_t)9 * (C / 32) * 1024; }
// wp[(tap * C/32 + cb)][lane][8]: k = 32 cb + 8 (lane >> 4) + e, n = lane & 15
__global__ void head_conv_pack_kernel(const float* w, bf16_t* wp, int C, int Cout) {
    const int total = 9 * (C / 32) * 64 * 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 7, lane = (i >> 3) & 63, s = i >> 9;
        const int cb = s % (C / 32), tap = s / (C / 32);
        const int k = 32 * cb + 8 * (lane >> 4) + e, n = lane & 15;
        float v = 0.f;
        if (n < 2 * Cout) {
            const float f = w[((n % Cout) * C + k) * 9 + tap];
            const float hi = bf2f(f2bf(f));
            v = n < Cout ? hi : f - hi;
        }
        wp[i] = f2bf(v);
    }
}

constexpr int HC_RING = 6, HC_PX = 34;
__global__ __launch_bounds__(512) void head_conv_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    const int C = p.C, PS = 2 * C + 16, ROWB = HC_PX * PS, NCB = C >> 5;
    char* ring = hsm;                                            // [6][34][PS]
    char* wl = hsm + HC_RING * ROWB;                             // [9 NCB][1024]
    float* fab = (float*)(wl + 9 * NCB * 1024);                  // [C][2]: a, b of y = a x + b
    float* gstat = fab + 2 * C;                                  // [groups][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, x0 = blockIdx.x * 32;
    const int band = (p.H + gridDim.z - 1) / gridDim.z, y0 = blockIdx.z * band, y1 = min(p.H, y0 + band);
    // ---- one-time: weights, normalisation table
    for (int i = tid; i < 9 * NCB * 64; i += 512) *(uint4*)(wl + i * 16) = *(const uint4*)((const char*)p.wp + (size_t)i * 16);
    if (p.partial) {
        const int cg = C / p.groups;
        if (tid < p.groups) {
            double a = 0.0, q = 0.0;
            const float* pp = p.partial + ((long long)b * p.nchunk * p.groups + tid) * 2;
            for (int k = 0; k < p.nchunk; k++) { a += pp[(long long)k * p.groups * 2]; q += pp[(long long)k * p.groups * 2 + 1]; }
            const double n = (double)cg * p.H * p.W, mean = a / n;
            double var = q / n - mean * mean;
            if (var < 0) var = 0;
            gstat[tid * 2] = (float)mean; gstat[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
        }
        __syncthreads();
        for (int c = tid; c < C; c += 512) {
            const int g = c / cg;
            const float fa = gstat[g * 2 + 1] * p.gamma[c];
            fab[2 * c] = fa; fab[2 * c + 1] = p.beta[c] - gstat[g * 2] * fa;
        }
    }
    __syncthreads();
    // ---- row staging: rows come in pairs; a thread's pieces of a pair are requested together and committed later.  A thread's NPC
    // pieces sit at the same (row of the pair, pixel, channel piece) in every pair: located once.
    const int ppx = C >> 3, per_pair = 2 * HC_PX * ppx;          // 16-byte pieces per pixel / per row pair
    constexpr int NPC = 8;                                       // pieces per thread and pair (C <= 240)
    int pr[NPC], ploff[NPC], ppc[NPC]; long long pgoff[NPC];     // row in the pair (or -1: no such piece / outside the image in x), LDS and global offsets
    // waves 4-7 stage rows (loads, normalisation, LDS writes), waves 0-3 run the MFMAs: the two phases overlap instead of alternating
    const bool stager = w >= 4;
    const int stid = tid & 255;
#pragma unroll
    for (int u = 0; u < NPC; u++) {
        const int i = u * 256 + stid;
        const int ic = min(i, per_pair - 1);
        const int r = ic / (HC_PX * ppx), rem = ic - r * (HC_PX * ppx), px = rem / ppx, pc = rem - px * ppx;
        const int xx = x0 - 1 + px;
        pr[u] = i < per_pair ? (xx >= 0 && xx < p.W ? r : 2 + r) : -1;          // 2 + r: a column outside the image (stored as zero)
        ploff[u] = px * PS + pc * 16; ppc[u] = pc * 8;
        pgoff[u] = (long long)xx * C + pc * 8;
    }
    struct Pair { uint4 v[NPC]; };
    const bf16_t* xb = p.x + (long long)b * p.H * p.W * C;
    auto request = [&](Pair& t, int ya) {                        // image rows ya, ya + 1 (anything outside the image is zero)
#pragma unroll
        for (int u = 0; u < NPC; u++) {
            const int y = ya + (pr[u] & 1);
            const bool in = pr[u] >= 0 && pr[u] < 2 && y >= 0 && y < p.H;
            t.v[u] = in ? *(const uint4*)(xb + (long long)y * p.W * C + pgoff[u]) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto commit = [&](const Pair& t, int ya) {
#pragma unroll
        for (int u = 0; u < NPC; u++) {
            if (pr[u] < 0) continue;
            const int y = ya + (pr[u] & 1);
            uint4 o = t.v[u];
            if (p.partial && pr[u] < 2 && y >= 0 && y < p.H) {
                const uint32_t in[4] = {o.x, o.y, o.z, o.w};
                uint32_t ov[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const f32x4 ab = *(const f32x4*)(fab + 2 * (ppc[u] + 2 * e));            // a0 b0 a1 b1
                    const float v0 = silu_f(__uint_as_float(in[e] << 16) * ab[0] + ab[1]);
                    const float v1 = silu_f(__uint_as_float(in[e] & 0xffff0000u) * ab[2] + ab[3]);
                    ov[e] = cvt_pk_bf16(v0, v1);
                }
                o = make_uint4(ov[0], ov[1], ov[2], ov[3]);
            }
            const int slot = (y + HC_RING) % HC_RING;
            *(uint4*)(ring + slot * ROWB + ploff[u]) = o;
        }
    };
    Pair pa, pb;
    if (stager) {
        request(pa, y0 - 1); request(pb, y0 + 1);
        commit(pa, y0 - 1); commit(pb, y0 + 1);
    }
    __syncthreads();
    // ---- two output rows per round: wave w -> row ya + (w >> 1), 16-pixel tile w & 1.  Rows are requested TWO rounds ahead (one round
    // of MFMAs does not cover an HBM round trip with four waves on the CU) and committed one round ahead.
    const int m = lane & 15, kg = lane >> 4, tx = w & 1;
    const int aoff = (1 + tx * 16 + m) * PS + kg * 16;           // this lane's pixel (centre tap) inside a ring row
    const char* bl = wl + lane * 16;
    auto compute = [&](int ya) {
        const int yo = ya + (w >> 1);
        if (yo >= y1) return;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // two chains: consecutive MFMAs do not wait on each other
        const char* rowp[3];
#pragma unroll
        for (int d = 0; d < 3; d++) rowp[d] = ring + ((yo + d - 1 + HC_RING) % HC_RING) * ROWB + aoff;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const char* ar = rowp[tap / 3] + (tap % 3 - 1) * PS;
            const char* br = bl + tap * NCB * 1024;
            int cb = 0;
            for (; cb + 2 <= NCB; cb += 2) {
                const bf16x8 a0 = *(const bf16x8*)(ar + cb * 64), a1 = *(const bf16x8*)(ar + cb * 64 + 64);
                const bf16x8 b0 = *(const bf16x8*)(br + cb * 1024), b1 = *(const bf16x8*)(br + cb * 1024 + 1024);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc1, 0, 0, 0);
            }
            if (cb < NCB) {
                const bf16x8 a0 = *(const bf16x8*)(ar + cb * 64);
                const bf16x8 b0 = *(const bf16x8*)(br + cb * 1024);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc0, 0, 0, 0);
            }
        }
        const f32x4 acc = acc0 + acc1;
        // lane (n = lane & 15, rows 4 kg + i): channel n's high-part sum; the low-part sum sits in lane + Cout
        f32x4 lo;
#pragma unroll
        for (int i = 0; i < 4; i++) lo[i] = __shfl(acc[i], lane + p.Cout);
        if (m < p.Cout) {
            const float bs = p.bias ? p.bias[m] : 0.f;
            f32x4 o = {acc[0] + lo[0] + bs, acc[1] + lo[1] + bs, acc[2] + lo[2] + bs, acc[3] + lo[3] + bs};
            *(f32x4*)(p.out + (((long long)b * p.Cout + m) * p.H + yo) * p.W + x0 + tx * 16 + 4 * kg) = o;
        }
    };
    if (stager && y0 + 2 < y1) request(pa, y0 + 3);
    for (int ya = y0; ya < y1; ya += 4) {
        if (stager) {
            if (ya + 4 < y1) request(pb, ya + 5);
            if (ya + 2 < y1) commit(pa, ya + 3);
        } else compute(ya);
        __syncthreads();
        if (ya + 2 < y1) {
            if (stager) {
                if (ya + 6 < y1) request(pa, ya + 7);
                if (ya + 4 < y1) commit(pb, ya + 5);
            } else compute(ya + 2);
            __syncthreads();
        }
    }
}

bool head_conv_supported(const HeadParams& p) {
    return p.C % 32 == 0 && p.C >= 32 && p.C <= 240 && p.W % 32 == 0 && p.H % 2 == 0 && p.Cout >= 1 && p.Cout <= 8 &&
           (!p.partial || (p.groups > 0 && p.groups <= 64 && p.C % p.groups == 0));
}
hipError_t launch_head_conv(const HeadParams& p, hipStream_t st) {
    if (!head_conv_supported(p) || !p.x || !p.w || !p.wp || !p.out) return hipErrorInvalidValue;
    const size_t smem = (size_t)HC_RING * HC_PX * (2 * p.C + 16) + head_conv_wp_bytes(p.C) + (size_t)p.C * 8 + 64 * 8;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    static bool attr[RDM_MAX_DEVICES] = {};
    const int dev = rdm_cur_device();
    if (!attr[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)head_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr[dev] = true;
    }
    head_conv_pack_kernel<<<(9 * (p.C / 32) * 512 + 255) / 256, 256, 0, st>>>(p.w, p.wp, p.C, p.Cout);
    // bands: enough blocks for the chip when the batch is small (a band re-reads one halo row above and below)
    const int pairs = (p.W / 32) * p.B;
    int bands = (512 + pairs - 1) / pairs; if (bands > p.H / 8) bands = p.H / 8; if (bands < 1) bands = 1;
    int band = (p.H + bands - 1) / bands; band += band & 1;          // even band heights: rows are walked in pairs
    bands = (p.H + band - 1) / band;
    head_conv_kernel<<<dim3(p.W / 32, p.B, bands), 512, smem, st>>>(p);
    return hipGetLastError();
}

// ------------------------------------------------------------------ timestep embedding
// ldm timestep_embedding: [cos(t*f_i) | sin(t*f_i)], f_i = exp(-ln(1e4) * i / half). bf16 out.
__global__ void timestep_embedding_kernel(const long long* t, bf16_t* out, int B, int dim, int ld) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (ld / 2)) return;
    const int b = i / (ld / 2), j = i % (ld / 2);
    if (j >= half) {                        // zero tail of a padded row [dim, ld) (two columns per thread)
        const int c = dim + (j - half) * 2;
        if (c < ld) out[(long long)b * ld + c] = 0;
        if (c + 1 < ld) out[(long long)b * ld + c + 1] = 0;
        return;
    }
    const float freq = expf(-9.210340371976184f * (float)j / (float)half);
    const float arg = (float)t[b] * freq;
    out[(long long)b * ld + j] = f2bf(cosf(arg));
    out[(long long)b * ld + half + j] = f2bf(sinf(arg));
}
hipError_t launch_timestep_embedding(const long long* t, bf16_t* out, int B, int dim, int ld, hipStream_t st) {
    if (dim % 2 || ld % 2 || ld < dim) return hipErrorInvalidValue;
    const int n = B * (ld / 2);
    timestep_embedding_kernel<<<(n + 255) / 256, 256, 0, st>>>(t, out, B, dim, ld);
    return hipGetLastError();
}

// ------------------------------------------------------------------ casts
__global__ void cast_f32_bf16_kernel(const float* x, bf16_t* y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = f2bf(x[i]);
}
hipError_t launch_cast_f32_bf16(const float* x, bf16_t* y, long long n, hipStream_t st) {
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    cast_f32_bf16_kernel<<<grid, 256, 0, st>>>(x, y, n);
    return hipGetLastError();
}

// y[c][r] = x[r][c] (small weight matrices, once per sampling call)
__global__ void transpose_bf16_kernel(const bf16_t* x, bf16_t* y, int rows, int cols, int ldy) {
    __shared__ bf16_t tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;      // row tiles on grid.x: a [millions of rows, N] operand (Linear weight gradients) has more than 65 535 of them
    x += (long long)blockIdx.z * rows * cols; y += (long long)blockIdx.z * cols * ldy;        // batch of matrices; ldy: output row pitch (>= rows)
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? x[(long long)r * cols + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < rows && c < cols) y[(long long)c * ldy + r] = tile[threadIdx.x][i];
    }
}
hipError_t launch_transpose_bf16(const bf16_t* x, bf16_t* y, int rows, int cols, hipStream_t st, int batch, int ldy) {
    if ((cols + 31) / 32 > 65535) return hipErrorInvalidValue;
    transpose_bf16_kernel<<<dim3((rows + 31) / 32, (cols + 31) / 32, batch < 1 ? 1 : batch), dim3(32, 8), 0, st>>>(x, y, rows, cols, ldy > 0 ? ldy : rows);
    return hipGetLastError();
}

// Block-diagonal head expansion of the cross-attention keys / values: out[b][h*k + j][n] = scale * kv[(b*k + j)*ld + n] if
// n / hd == h else 0, rows >= heads*k zero.  out [B][NP][C], C = heads * hd.
__global__ void expand_heads_kernel(const bf16_t* kv, int ld, int B, int k, int heads, int hd, int NP, float scale, bf16_t* out) {
    const int C = heads * hd;
    const long long total = (long long)B * NP * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % C); const long long rr = i / C; const int hj = (int)(rr % NP); const int b = (int)(rr / NP);
        const int h = hj / k, j = hj - h * k;
        float v = 0.f;
        if (hj < heads * k && n / hd == h) v = scale * bf2f(kv[((long long)b * k + j) * ld + n]);
        out[i] = f2bf(v);
    }
}
hipError_t launch_expand_heads(const bf16_t* kv, int ld, int B, int k, int heads, int hd, int NP, float scale, bf16_t* out, hipStream_t st) {
    const long long total = (long long)B * NP * heads * hd;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
    expand_heads_kernel<<<grid, 256, 0, st>>>(kv, ld, B, k, heads, hd, NP, scale, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ DDIM / DDPM updates (fp32)
// rdm/models/diffusion/ddim.py:229-238 (CFG combine) + :253-267 (update), fused into one pass.
// eps holds [B] rows (scale == 1) or [2B] rows (cond | uncond). Scalars are the fp32 values the
// reference materialises with torch.full_like; sqrt taken in fp32 on device like a_t.sqrt().
__global__ void ddim_step_kernel(DdimStepParams p) {
    const float sa = sqrtf(p.a_t), sap = sqrtf(p.a_prev);
    const float dirc = sqrtf(1.0f - p.a_prev - p.sigma_t * p.sigma_t);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n_per_batch;
         i += (long long)gridDim.x * blockDim.x) {
        float e = p.eps[i];
        if (p.cfg) { const float eu = p.eps[p.n_per_batch + i]; e = eu + p.scale * (e - eu); }
        const float x = p.x[i];
        const float x0 = (x - p.sqrt_one_minus_at * e) / sa;
        const float dir = dirc * e;
        const float nz = p.noise ? p.sigma_t * p.noise[i] * p.temperature : 0.f;
        const float xp = sap * x0 + dir + nz;
        p.x_prev[i] = xp;
        if (p.x_dup) p.x_dup[i] = xp;
        if (p.pred_x0) p.pred_x0[i] = x0;
    }
}
hipError_t launch_ddim_step(const DdimStepParams& p, hipStream_t st) {
    int grid = (int)((p.n_per_batch + 255) / 256); if (grid > 2048) grid = 2048;
    ddim_step_kernel<<<grid, 256, 0, st>>>(p);
    return hipGetLastError();
}

// out[r, :] = x[r, :] + bias[:]   (bf16 rows, 8 columns per thread).  Cross-attention of a sample whose neighbours are all-zero
// vectors -- the unconditional half of a guided batch, rdm/models/diffusion/ddpm.py:673-680 -- is exactly to_out.bias: K = V = 0
// gives uniform attention over zero values (rdm/modules/attention.py:52-72), so t2 = t1 + b_o without any GEMM.
__global__ __launch_bounds__(256) void add_bias_rows_kernel(const bf16_t* x, const float* bias, bf16_t* out, long long rows, int C) {
    const int cv = C >> 3;
    const long long nvec = rows * cv;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const int c = (int)(v % cv) * 8;
        const uint4 u = *(const uint4*)(x + v * 8);
        const float4 b0 = *(const float4*)(bias + c), b1 = *(const float4*)(bias + c + 4);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
            o[e] = cvt_pk_bf16(__uint_as_float(w[e] << 16) + bb[2 * e], __uint_as_float(w[e] & 0xffff0000u) + bb[2 * e + 1]);
        *(uint4*)(out + v * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
hipError_t launch_add_bias_rows(const bf16_t* x, const float* bias, bf16_t* out, long long rows, int C, hipStream_t st) {
    if (C % 8) return hipErrorInvalidValue;
    const long long nvec = rows * (C >> 3);
    if (nvec <= 0) return hipSuccess;
    int grid = (int)((nvec + 255) / 256); if (grid > 8192) grid = 8192;
    add_bias_rows_kernel<<<grid, 256, 0, st>>>(x, bias, out, rows, C);
    return hipGetLastError();
}
// flag[r] = 1 if row r of x [rows, n] has a non-zero element (one block per row)
__global__ __launch_bounds__(256) void row_nonzero_kernel(const float* x, long long n, int* flag) {
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const float* r = x + (long long)blockIdx.x * n;
    int mine = 0;
    for (long long i = threadIdx.x; i < n; i += 256) mine |= (r[i] != 0.0f);
    if (mine) any = 1;
    __syncthreads();
    if (threadIdx.x == 0) flag[blockIdx.x] = any;
}
hipError_t launch_row_nonzero(const float* x, int rows, long long n, int* flag, hipStream_t st) {
    row_nonzero_kernel<<<rows, 256, 0, st>>>(x, n, flag);
    return hipGetLastError();
}

// ldm LatentDiffusion.p_sample: x0 = c_recip*x - c_recipm1*eps; clamp; mean; + sigma*z
__global__ void ddpm_step_kernel(DdpmStepParams p) {
    const float sd = p.nonzero ? expf(0.5f * p.log_var) : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x) {
        const float x = p.x[i];
        float x0 = p.sqrt_recip * x - p.sqrt_recipm1 * p.eps[i];
        if (p.clip) x0 = fminf(1.f, fmaxf(-1.f, x0));
        const float mean = p.coef1 * x0 + p.coef2 * x;
        p.x_prev[i] = mean + (p.nonzero ? sd * p.noise[i] * p.temperature : 0.f);
    }
}
hipError_t launch_ddpm_step(const DdpmStepParams& p, hipStream_t st) {
    int grid = (int)((p.n + 255) / 256); if (grid > 2048) grid = 2048;
    ddpm_step_kernel<<<grid, 256, 0, st>>>(p);
    return hipGetLastError();
}

// ------------------------------------------------------------------ VQ quantise + post_quant_conv
// taming VectorQuantizer2: argmin_j |z|^2 + |e_j|^2 - 2 z.e_j (first minimum), z_q = e[idx],
// then ldm post_quant_conv (1x1, embed_dim -> z_channels). embed_dim == z_channels == 3 in every shipped
// config. z NCHW f32 -> out NCHW f32 [B,3,H,W]; idx int32 [B*H*W] optional. Codebook (+|e|^2) lives in
// LDS as float4; every lane scans all codes (LDS broadcast reads).
__global__ __launch_bounds__(256) void vq_quantize_kernel(const float* z, const float* codebook, int n_embed,
                                                          const float* pq_w /*[3][3]*/, const float* pq_b,
                                                          float* out, int* idx_out, int B, int HW, int quantize) {
    extern __shared__ float4 cb[];
    for (int j = threadIdx.x; j < n_embed; j += 256) {
        const float e0 = codebook[j * 3], e1 = codebook[j * 3 + 1], e2 = codebook[j * 3 + 2];
        cb[j] = make_float4(e0, e1, e2, (e0 * e0 + e1 * e1) + e2 * e2);
    }
    __syncthreads();
    const long long npix = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), r = (int)(i % HW);
        const float* zp = z + (long long)b * 3 * HW + r;
        float z0 = zp[0], z1 = zp[HW], z2 = zp[2 * HW];
        int best = 0;
        if (quantize) {
            const float zz = (z0 * z0 + z1 * z1) + z2 * z2;
            float bd = INFINITY;
            for (int j = 0; j < n_embed; j++) {
                const float4 e = cb[j];
                const float dot = (z0 * e.x + z1 * e.y) + z2 * e.z;
                const float d = (zz + e.w) - 2.f * dot;
                if (d < bd) { bd = d; best = j; }
            }
            const float4 e = cb[best];
            // straight-through form reproduced literally: z + (e - z)
            z0 = z0 + (e.x - z0); z1 = z1 + (e.y - z1); z2 = z2 + (e.z - z2);
        }
        if (idx_out) idx_out[i] = best;
        float* op = out + (long long)b * 3 * HW + r;
#pragma unroll
        for (int co = 0; co < 3; co++)
            op[co * HW] = pq_b[co] + pq_w[co * 3] * z0 + pq_w[co * 3 + 1] * z1 + pq_w[co * 3 + 2] * z2;
    }
}
hipError_t launch_vq_quantize(const float* z, const float* codebook, int n_embed, const float* pq_w, const float* pq_b,
                              float* out, int* idx_out, int B, int HW, int quantize, hipStream_t st) {
    const size_t sm = (size_t)n_embed * sizeof(float4);
    if (sm > 160 * 1024 - 256) return hipErrorInvalidValue;
    static bool attr_dev[RDM_MAX_DEVICES] = {false};
    bool& attr = attr_dev[rdm_cur_device()];
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)vq_quantize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 256);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const long long npix = (long long)B * HW;
    int grid = (int)((npix + 255) / 256); if (grid > 1024) grid = 1024;
    vq_quantize_kernel<<<grid, 256, sm, st>>>(z, codebook, n_embed, pq_w, pq_b, out, idx_out, B, HW, quantize);
    return hipGetLastError();
}

// ------------------------------------------------------------------ row softmax f32 -> bf16
// VQ decoder AttnBlock (single head, 4096 tokens): softmax over the key axis of the f32 score matrix.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, bf16_t* p, long long rows, int n, int nv) {
    const int lane = threadIdx.x & 63;                    // nv: valid columns (the rest are padding: probability 0)
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
        const float* sr = s + row * n;
        float mx = -INFINITY;
        for (int i = lane * 4; i < n; i += 256) {
            const float4 v = *(const float4*)(sr + i);
            mx = fmaxf(fmaxf(mx, fmaxf(i < nv ? v.x : -INFINITY, i + 1 < nv ? v.y : -INFINITY)), fmaxf(i + 2 < nv ? v.z : -INFINITY, i + 3 < nv ? v.w : -INFINITY));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int i = lane * 4; i < n; i += 256) {
            const float4 v = *(const float4*)(sr + i);
            sum += ((i < nv ? __expf(v.x - mx) : 0.f) + (i + 1 < nv ? __expf(v.y - mx) : 0.f)) + ((i + 2 < nv ? __expf(v.z - mx) : 0.f) + (i + 3 < nv ? __expf(v.w - mx) : 0.f));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.f / sum;
        for (int i = lane * 4; i < n; i += 256) {
            const float4 v = *(const float4*)(sr + i);
            uint2 w;
            w.x = pack2bf(i < nv ? __expf(v.x - mx) * inv : 0.f, i + 1 < nv ? __expf(v.y - mx) * inv : 0.f);
            w.y = pack2bf(i + 2 < nv ? __expf(v.z - mx) * inv : 0.f, i + 3 < nv ? __expf(v.w - mx) * inv : 0.f);
            *(uint2*)(p + row * n + i) = w;
        }
    }
}
hipError_t launch_softmax_rows(const float* s, bf16_t* p, long long rows, int n, hipStream_t st, int n_valid) {
    if (n % 4) return hipErrorInvalidValue;
    int grid = (int)((rows + 3) / 4); if (grid > 8192) grid = 8192;
    softmax_rows_kernel<<<grid, 256, 0, st>>>(s, p, rows, n, (n_valid > 0 && n_valid < n) ? n_valid : n);
    return hipGetLastError();
}

// ------------------------------------------------------------------ CLIP helpers
// token + positional embedding -> f32 residual stream [B, L, W] (custom_clip/model.py:308-310)
__global__ void clip_embed_kernel(const long long* tokens, const float* tok_emb, const float* pos_emb, float* out, int B,
                                  int L, int Wd) {
    const long long n = (long long)B * L * Wd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Wd); const long long r = i / Wd; const int l = (int)(r % L);
        out[i] = tok_emb[tokens[r] * Wd + c] + pos_emb[l * Wd + c];
    }
}
hipError_t launch_clip_embed(const long long* tokens, const float* tok_emb, const float* pos_emb, float* out, int B, int L,
                             int Wd, hipStream_t st) {
    const long long n = (long long)B * L * Wd;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    clip_embed_kernel<<<grid, 256, 0, st>>>(tokens, tok_emb, pos_emb, out, B, L, Wd);
    return hipGetLastError();
}
// gather the EOT row (argmax of token ids, first max) of each sequence: [B,L,W] f32 -> [B,W] f32 (model.py:318)
__global__ void clip_gather_eot_kernel(const long long* tokens, const float* x, float* out, int L, int Wd) {
    const int b = blockIdx.x;
    __shared__ int pos;
    if (threadIdx.x == 0) {
        long long best = tokens[(long long)b * L]; int bi = 0;
        for (int l = 1; l < L; l++) { const long long t = tokens[(long long)b * L + l]; if (t > best) { best = t; bi = l; } }
        pos = bi;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Wd; c += blockDim.x) out[(long long)b * Wd + c] = x[((long long)b * L + pos) * Wd + c];
}
hipError_t launch_clip_gather_eot(const long long* tokens, const float* x, float* out, int B, int L, int Wd, hipStream_t st) {
    clip_gather_eot_kernel<<<B, 256, 0, st>>>(tokens, x, out, L, Wd);
    return hipGetLastError();
}
// ViT patchify: image NCHW f32 [B,3,R,R] -> bf16 [B*G*G, 3*P*P] rows in (c, py, px) order == conv1 weight flattening
__global__ void clip_patchify_kernel(const float* img, bf16_t* out, int B, int R, int P) {
    const int G = R / P, K = 3 * P * P;
    const long long n = (long long)B * G * G * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K); const long long r = i / K;
        const int gx = (int)(r % G), gy = (int)((r / G) % G), b = (int)(r / (G * G));
        const int c = k / (P * P), py = (k / P) % P, px = k % P;
        out[i] = f2bf(img[(((long long)b * 3 + c) * R + gy * P + py) * R + gx * P + px]);
    }
}
hipError_t launch_clip_patchify(const float* img, bf16_t* out, int B, int R, int P, hipStream_t st) {
    const long long n = (long long)B * (R / P) * (R / P) * 3 * P * P;
    int grid = (int)((n + 255) / 256); if (grid > 8192) grid = 8192;
    clip_patchify_kernel<<<grid, 256, 0, st>>>(img, out, B, R, P);
    return hipGetLastError();
}
// ---------------------------------------------------------------- ClipImageRetriever.preprocess (rdm/modules/retrievers.py:83-91)
// kornia.geometry.resize(x, (R,R), 'bicubic', align_corners=True, antialias=False) == torch upsample_bicubic2d: cubic
// convolution with A = -0.75 over the 4x4 neighbourhood of the source position oy*(H-1)/(R-1), indices clamped to the image;
// then (x+1)/2 and the CLIP mean / std.  One thread per output pixel and channel; PATCH = true writes straight into the bf16
// patch matrix [B*G*G, 3*P*P] the patch-embedding GEMM reads (the f32 [B,3,R,R] intermediate never exists).
__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}
struct PreprocParams { const float* img; int B, H, W, R, P; float sy, sx; float mean[3], istd[3]; };
template <bool PATCH>
__global__ __launch_bounds__(256) void clip_preprocess_kernel(PreprocParams p, float* out_f32, bf16_t* out_patch) {
    const long long n = (long long)p.B * 3 * p.R * p.R;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.R), oy = (int)((i / p.R) % p.R), c = (int)((i / ((long long)p.R * p.R)) % 3), b = (int)(i / ((long long)3 * p.R * p.R));
        // rounded products (no fma contraction into the subtraction below): the fractional position must be the one
        // torch's area_pixel_compute_source_index produces, or a 1200-px source shifts the taps by ~1e-4 px
        const float ry = __fmul_rn(p.sy, (float)oy), rx = __fmul_rn(p.sx, (float)ox);
        const float fy = floorf(ry), fx = floorf(rx);
        const int iy = (int)fy, ix = (int)fx;
        float wy[4], wx[4];
        cubic_coeffs(ry - fy, wy); cubic_coeffs(rx - fx, wx);
        const float* src = p.img + ((long long)b * 3 + c) * p.H * p.W;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int yy = min(max(iy - 1 + a, 0), p.H - 1);
            const float* row = src + (long long)yy * p.W;
            float r = 0.f;
#pragma unroll
            for (int d = 0; d < 4; d++) r += wx[d] * row[min(max(ix - 1 + d, 0), p.W - 1)];
            acc += wy[a] * r;
        }
        const float v = ((acc + 1.0f) * 0.5f - p.mean[c]) * p.istd[c];
        if (PATCH) {
            const int G = p.R / p.P, gy = oy / p.P, py = oy % p.P, gx = ox / p.P, px = ox % p.P;
            out_patch[(((long long)b * G + gy) * G + gx) * (3 * p.P * p.P) + (c * p.P + py) * p.P + px] = f2bf(v);
        } else out_f32[i] = v;
    }
}
hipError_t launch_clip_preprocess(const float* img, int B, int H, int W, int R, int P, float* out_f32, bf16_t* out_patch, hipStream_t st) {
    PreprocParams p{}; p.img = img; p.B = B; p.H = H; p.W = W; p.R = R; p.P = P;
    p.sy = R > 1 ? (float)(H - 1) / (float)(R - 1) : 0.f; p.sx = R > 1 ? (float)(W - 1) / (float)(R - 1) : 0.f;   // area_pixel_compute_scale, align_corners
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};     // retrievers.py:80-81
    for (int c = 0; c < 3; c++) { p.mean[c] = mean[c]; p.istd[c] = 1.0f / sd[c]; }
    const long long n = (long long)B * 3 * R * R;
    int grid = (int)((n + 255) / 256); if (grid > 16384) grid = 16384; if (grid < 1) grid = 1;
    if (out_patch) clip_preprocess_kernel<true><<<grid, 256, 0, st>>>(p, nullptr, out_patch);
    else clip_preprocess_kernel<false><<<grid, 256, 0, st>>>(p, out_f32, nullptr);
    return hipGetLastError();
}
// x[b, 0, :] = cls + pos[0]; x[b, 1+i, :] = patch[b*GG+i, :] + pos[1+i]  (model.py:221-222), f32
__global__ void clip_vit_assemble_kernel(const float* patch, const float* cls, const float* pos, float* out, int B, int GG,
                                         int Wd) {
    const long long n = (long long)B * (GG + 1) * Wd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Wd); const long long r = i / Wd; const int l = (int)(r % (GG + 1)); const int b = (int)(r / (GG + 1));
        const float v = (l == 0) ? cls[c] : patch[((long long)b * GG + (l - 1)) * Wd + c];
        out[i] = v + pos[l * Wd + c];
    }
}
hipError_t launch_clip_vit_assemble(const float* patch, const float* cls, const float* pos, float* out, int B, int GG, int Wd,
                                    hipStream_t st) {
    const long long n = (long long)B * (GG + 1) * Wd;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    clip_vit_assemble_kernel<<<grid, 256, 0, st>>>(patch, cls, pos, out, B, GG, Wd);
    return hipGetLastError();
}
// strided row gather f32: out[b,:] = x[(b*stride_rows), :]
__global__ void gather_rows_f32_kernel(const float* x, float* out, int B, long long row_stride, int Wd) {
    const long long n = (long long)B * Wd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Wd); const int b = (int)(i / Wd);
        out[i] = x[(long long)b * row_stride * Wd + c];
    }
}
hipError_t launch_gather_rows_f32(const float* x, float* out, int B, long long row_stride, int Wd, hipStream_t st) {
    const long long n = (long long)B * Wd;
    int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
    gather_rows_f32_kernel<<<grid, 256, 0, st>>>(x, out, B, row_stride, Wd);
    return hipGetLastError();
}

// ------------------------------------------------------------------ image conversion
// scripts/rdm_sample.py:203-214: clamp(-1,1) -> (x+1)/2 -> CHW->HWC -> *255 -> uint8 (truncation)
__global__ void to_uint8_hwc_kernel(const float* x, unsigned char* out, int B, int C, int H, int W) {
    const long long n = (long long)B * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long long r = i / C;
        const int xw = (int)(r % W), yh = (int)((r / W) % H), b = (int)(r / ((long long)W * H));
        float v = x[(((long long)b * C + c) * H + yh) * W + xw];
        v = fminf(1.f, fmaxf(-1.f, v));
        v = (v + 1.0f) / 2.0f;
        out[i] = (unsigned char)(255.0f * v);
    }
}
hipError_t launch_to_uint8_hwc(const float* x, unsigned char* out, int B, int C, int H, int W, hipStream_t st) {
    const long long n = (long long)B * H * W * C;
    int grid = (int)((n + 255) / 256); if (grid > 8192) grid = 8192;
    to_uint8_hwc_kernel<<<grid, 256, 0, st>>>(x, out, B, C, H, W);
    return hipGetLastError();
}

// GroupNorm(32) [+SiLU] and LayerNorm for NHWC / token-major bf16 activations (gfx950).
// Replaces: ldm `normalization` (GroupNorm32, eps 1e-5) + nn.SiLU in ResBlock in/out_layers and the
// UNet `out` head (rdm/modules/diffusionmodules/openaimodel.py:307-309), `Normalize` (eps 1e-6) in
// SpatialTransformer (rdm/modules/attention.py:16-17,147,183) and the VQ decoder, nn.LayerNorm x3 in
// BasicTransformerBlock (attention.py:84-86) and CLIP's LayerNorm (custom_clip/model.py:152-158).
//
// HBM-bound.  GroupNorm is two streaming passes: (1) per-(sample, pixel-chunk) partial sums with
// 16-byte coalesced row reads, reduced deterministically (fixed order, no atomics: results do not
// depend on scheduling, so batch sharding over GPUs is bit-reproducible); (2) an elementwise
// normalise+affine(+SiLU) pass that folds (mean, rstd, gamma, beta) into one per-channel FMA held in
// LDS.  Both passes read the decoder's skip-concat as two source tensors (never materialised).
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"


// Zero-padded sources (model.hip build_unet): x0 holds L0 <= C0 real channels, x1 L1 <= C1; groups partition the L0 + L1 logical
// channels.  gn_logical: physical concat index -> logical index (or -1 for a padding channel); gn_physical: the inverse.
__device__ __forceinline__ int gn_logical(const GnParams& p, int c) {
    if (c < p.C0) return c < p.L0 ? c : -1;
    return (c - p.C0) < p.L1 ? p.L0 + (c - p.C0) : -1;
}
__device__ __forceinline__ int gn_physical(const GnParams& p, int l) { return l < p.L0 ? l : p.C0 + (l - p.L0); }

__device__ __forceinline__ const bf16_t* gn_src(const GnParams& p, int b, int row, int c) {
    if (c < p.C0) return p.x0 + ((long long)((p.x0_bmod > 0 ? b % p.x0_bmod : b) * p.HW + row) * p.C0 + c);
    const int b1 = p.x1_bmod > 0 ? b % p.x1_bmod : b;
    return p.x1 + ((long long)(b1 * p.HW + row) * p.C1 + (c - p.C0));
}

// grid (nchunk, B); block = VC * R threads where VC = C/8 (16-byte vectors per row)
__global__ void gn_stats_kernel(GnParams p) {
    extern __shared__ float sm[];          // [R][C][2] then [C][2]
    const int C = p.C0 + p.C1, VC = C >> 3;
    const int R = blockDim.x / VC;
    const int v = threadIdx.x % VC, rr = threadIdx.x / VC;
    const int b = blockIdx.y + p.b0, chunk = blockIdx.x;
    const int rows_per = (p.HW + p.nchunk - 1) / p.nchunk;
    const int r0 = chunk * rows_per, r1 = min(p.HW, r0 + rows_per);
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { s[i] = 0.f; q[i] = 0.f; }
    if (rr < R) {
        // four independent 16-byte loads in flight per thread (a single dependent load per iteration leaves the HBM idle)
        int row = r0 + rr;
        for (; row + 3 * R < r1; row += 4 * R) {
            bf16x8 d[4];
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = *(const bf16x8*)gn_src(p, b, row + u * R, v * 8);
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 8; i++) { const float f = bf2f((bf16_t)d[u][i]); s[i] += f; q[i] += f * f; }
        }
        for (; row < r1; row += R) {
            const bf16x8 d = *(const bf16x8*)gn_src(p, b, row, v * 8);
#pragma unroll
            for (int i = 0; i < 8; i++) { const float f = bf2f((bf16_t)d[i]); s[i] += f; q[i] += f * f; }
        }
        // partials as [element i of the vector][row lane rr][vector v]: consecutive threads (v) write consecutive float2 -- the
        // [rr][channel] order put a thread's 8 channels 64 bytes apart from its neighbour's: a 16-way bank conflict on every one of the
        // 16 writes (round-3 counters: LDS bank-conflict / LDS-active = 0.87 in this kernel)
#pragma unroll
        for (int i = 0; i < 8; i++) *(float2*)(sm + ((i * R + rr) * VC + v) * 2) = make_float2(s[i], q[i]);
    }
    __syncthreads();
    // per-channel reduce over R (fixed order): thread -> (i, v), consecutive threads read consecutive float2
    float* ch = sm + R * C * 2;
    for (int idx = threadIdx.x; idx < C; idx += blockDim.x) {
        const int i = idx / VC, vv = idx - i * VC;
        float a = 0.f, bq = 0.f;
        for (int r = 0; r < R; r++) { const float2 t = *(const float2*)(sm + ((i * R + r) * VC + vv) * 2); a += t.x; bq += t.y; }
        ch[(vv * 8 + i) * 2] = a; ch[(vv * 8 + i) * 2 + 1] = bq;
    }
    __syncthreads();
    const int cg = (p.L0 + p.L1) / p.groups;
    if (threadIdx.x < p.groups) {
        float a = 0.f, bq = 0.f;
        for (int l = threadIdx.x * cg; l < (threadIdx.x + 1) * cg; l++) { const int c = gn_physical(p, l); a += ch[c * 2]; bq += ch[c * 2 + 1]; }
        float* o = p.partial + (((long long)b * p.nchunk + chunk) * p.groups + threadIdx.x) * 2;
        o[0] = a; o[1] = bq;
    }
}

// grid (nblk, B); block = VC * R threads (same shape as the stats kernel). Every thread owns ONE 8-channel vector
// for the whole launch, so (mean, rstd, gamma, beta) fold into 16 registers and the body is a pure 16-byte stream:
// no LDS table (a per-channel table read with a 64-byte lane stride is a 16-way bank conflict).
__global__ void gn_apply_kernel(GnParams p) {
    const int C = p.C0 + p.C1, VC = C >> 3, cg = (p.L0 + p.L1) / p.groups;
    const int R = blockDim.x / VC;
    const int v = threadIdx.x % VC, rr = threadIdx.x / VC;
    const int b = blockIdx.y + p.b0;
    __shared__ float gstat[64][2];
    if (threadIdx.x < p.groups) {
        double a = 0.0, q = 0.0;
        const float* pp = p.partial + ((long long)b * p.nchunk * p.groups + threadIdx.x) * 2;
        // eight (sum, sumsq) pairs in flight per round trip, summed in chunk order (left to a rolled loop this is up to 32 dependent
        // L2 round trips in front of the whole block: ~10 us per launch)
        for (int k0 = 0; k0 < p.nchunk; k0 += 8) {
            float2 t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = (k0 + u < p.nchunk) ? *(const float2*)(pp + (long long)(k0 + u) * p.groups * 2) : make_float2(0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; u++) { a += t[u].x; q += t[u].y; }
        }
        const double n = (double)cg * p.HW;
        const double mean = a / n;
        double var = q / n - mean * mean;
        if (var < 0) var = 0;
        gstat[threadIdx.x][0] = (float)mean;
        gstat[threadIdx.x][1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    if (rr >= R) return;
    float fa[8], fb[8];
    {   // The thread's 8 channels are all logical or all padding (sources and logical counts are multiples of 8): ONE division finds
        // the first channel's group, the rest follow by counting.  gamma / beta are read as two 16-byte vectors each, unconditionally
        // -- the packer zeroes them on padding channels, which therefore come out as exactly 0 whatever group is used for them
        // (guarding the loads per channel instead serialised 16 scalar loads: +10 us on every launch).
        const int l0 = gn_logical(p, v * 8);
        int g = l0 < 0 ? 0 : l0 / cg, r = l0 < 0 ? 0 : l0 - g * cg;
        const float4 g0 = *(const float4*)(p.gamma + v * 8), g1 = *(const float4*)(p.gamma + v * 8 + 4);
        const float4 b0 = *(const float4*)(p.beta + v * 8), b1 = *(const float4*)(p.beta + v * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
            fa[e] = gstat[g][1] * gg[e];
            fb[e] = bb[e] - gstat[g][0] * fa[e];
            if (++r == cg) { r = 0; g++; }
        }
    }
    const int rows_per = (p.HW + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per, r1 = min(p.HW, r0 + rows_per);
    for (int row = r0 + rr; row < r1; row += R) {
        const bf16x8 d = *(const bf16x8*)gn_src(p, b, row, v * 8);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            y[e] = bf2f((bf16_t)d[e]) * fa[e] + fb[e];
            if (p.silu) y[e] = silu_f(y[e]);
        }
        *(uint4*)(p.out + ((long long)(b * p.HW + row) * C + v * 8)) =
            make_uint4(cvt_pk_bf16(y[0], y[1]), cvt_pk_bf16(y[2], y[3]), cvt_pk_bf16(y[4], y[5]), cvt_pk_bf16(y[6], y[7]));
    }
}

// ------------------------------------------------------------------------------ one-pass GroupNorm (round 5)
// Groups are independent, so a block that owns WHOLE groups of one sample needs nobody else: it reads its [HW pixels] x [channels of gpb
// consecutive groups] slice ONCE into registers (a thread keeps one 8-channel vector position of NV pixels: <= 32 x 16 bytes), forms the
// group statistics (per-thread sums -> LDS per channel over the pixel lanes -> per group, all in a fixed order: results do not depend on
// scheduling), normalises from the registers and writes once: 2 tensor passes instead of the 3 of gn_stats + gn_apply, one launch instead
// of two.  Applies where a slice fits: every GroupNorm of the 32 x 32, 16 x 16 and 8 x 8 levels (the 64 x 64 level's groups are 6
// channels = 12 bytes of a pixel row: slices would be read at a quarter of a cache line's efficiency; it keeps the two-pass form).
// Slice bytes per pixel are gpb cg 2 (96 .. 480 B: whole cache lines or nearly); the blocks of one sample sit on ONE XCD (workgroups go
// round-robin over the XCDs by id) so that lines shared by neighbouring slices cross the fabric once.
// grid: B * nslice blocks of T threads; T / VS pixel lanes (VS = slice vectors per pixel).
template <int NV, int T>
__global__ __launch_bounds__(T) void gn_onepass_kernel(GnParams p, int gpb, int nslice) {
    extern __shared__ float sm[];                          // [R][SC][2] per-lane channel partials, then [SC][2], then [gpb][2]
    const int C = p.C0 + p.C1, cg = C / p.groups;
    const int SC = gpb * cg, VS = SC >> 3;                 // slice channels / vectors per pixel
    const int R = blockDim.x / VS;
    int lin = blockIdx.x, b, slice;
    if ((p.B & 7) == 0) { const int xcd = lin & 7, slot = lin >> 3; slice = slot % nslice; b = (slot / nslice) * 8 + xcd; }
    else { slice = lin % nslice; b = lin / nslice; }
    const int c0 = slice * SC;
    const int v = threadIdx.x % VS, rr = threadIdx.x / VS;
    const bool live = rr < R;
    uint4 d[NV];
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { s[i] = 0.f; q[i] = 0.f; }
    if (live) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int row = rr + i * R;
            d[i] = row < p.HW ? *(const uint4*)gn_src(p, b, row, c0 + v * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const uint32_t u[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float lo = __uint_as_float(u[e] << 16), hi = __uint_as_float(u[e] & 0xffff0000u);
                s[2 * e] += lo; q[2 * e] += lo * lo; s[2 * e + 1] += hi; q[2 * e + 1] += hi * hi;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) *(float2*)(sm + ((i * R + rr) * VS + v) * 2) = make_float2(s[i], q[i]);      // conflict-free: consecutive threads, consecutive float2
    }
    __syncthreads();
    float* ch = sm + R * SC * 2;
    for (int idx = threadIdx.x; idx < SC; idx += blockDim.x) {
        const int i = idx / VS, vv = idx - i * VS;
        float a = 0.f, bq = 0.f;
        for (int r = 0; r < R; r++) { const float2 t = *(const float2*)(sm + ((i * R + r) * VS + vv) * 2); a += t.x; bq += t.y; }
        ch[(vv * 8 + i) * 2] = a; ch[(vv * 8 + i) * 2 + 1] = bq;
    }
    __syncthreads();
    float* gst = ch + SC * 2;
    if (threadIdx.x < gpb) {
        double a = 0.0, bq = 0.0;
        for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; c++) { a += ch[c * 2]; bq += ch[c * 2 + 1]; }
        const double n = (double)cg * p.HW, mean = a / n;
        double var = bq / n - mean * mean; if (var < 0) var = 0;
        gst[threadIdx.x * 2] = (float)mean; gst[threadIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    if (!live) return;
    float fa[8], fb[8];
    {
        int g = (v * 8) / cg, r = v * 8 - g * cg;
        const float4 g0 = *(const float4*)(p.gamma + c0 + v * 8), g1 = *(const float4*)(p.gamma + c0 + v * 8 + 4);
        const float4 b0 = *(const float4*)(p.beta + c0 + v * 8), b1 = *(const float4*)(p.beta + c0 + v * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
            fa[e] = gst[g * 2 + 1] * gg[e];
            fb[e] = bb[e] - gst[g * 2] * fa[e];
            if (++r == cg) { r = 0; g++; }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int row = rr + i * R;
        if (row >= p.HW) break;
        const uint32_t u[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            y[2 * e] = __uint_as_float(u[e] << 16) * fa[2 * e] + fb[2 * e];
            y[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u) * fa[2 * e + 1] + fb[2 * e + 1];
        }
        if (p.silu) {
#pragma unroll
            for (int e = 0; e < 8; e++) y[e] = silu_f(y[e]);
        }
        *(uint4*)(p.out + ((long long)(b * p.HW + row) * C + c0 + v * 8)) =
            make_uint4(cvt_pk_bf16(y[0], y[1]), cvt_pk_bf16(y[2], y[3]), cvt_pk_bf16(y[4], y[5]), cvt_pk_bf16(y[6], y[7]));
    }
}

// slice geometry of the one-pass form for this shape, or false (the two-pass kernels run)
static bool gn_onepass_plan(const GnParams& p, int& gpb, int& T, int& nv) {
    static const int off = getenv("RDM_NO_GN1PASS") ? atoi(getenv("RDM_NO_GN1PASS")) : 0;
    static const int max_hw = getenv("RDM_GN1PASS_MAXHW") ? atoi(getenv("RDM_GN1PASS_MAXHW")) : 1024;
    static const int min_seg = getenv("RDM_GN1PASS_MINSEG") ? atoi(getenv("RDM_GN1PASS_MINSEG")) : 96;      // bytes of a pixel row per slice
    const int C = p.C0 + p.C1;
    if (off || p.L0 != p.C0 || p.L1 != p.C1 || C % p.groups || p.HW > max_hw || p.HW < 1) return false;      // (padded widths: two-pass)
    const int cg = C / p.groups;
    // groups per block: slices of whole 16-byte vectors (C0 % 8 == 0, so no vector straddles the x0 | x1 boundary), at least min_seg bytes
    // of every pixel row, and registers for it: T / VS pixel lanes x NV <= 32 (16 at 1024 threads) vectors each.  Among the feasible
    // slicings the one closest to a quarter of a sample per block.  A function of the PER-SAMPLE shape only: the slicing -- hence the
    // summation order -- must not follow the batch (deterministic mode).
    const long long want = (long long)p.HW * C / 4;
    long long best_d = -1;
    for (int g = 1; g <= p.groups; g++) {
        if (p.groups % g || (g * cg) % 8 || g * cg * 2 < min_seg) continue;
        const int VS = g * cg / 8;
        int t_ok = 0, n_ok = 0;
        for (int t : {256, 512, 1024}) {
            if (VS > t) continue;
            const int R = t / VS, n = (p.HW + R - 1) / R;
            // (1024 threads: 128 registers each; 512 threads: 256 each -- RDM_GN1PASS_NV512=52 admits 52 vectors per thread there: the 64 x 64
            //  level's 8-group slices of 192 / 384-channel tensors, 96-byte row segments, with RDM_GN1PASS_MAXHW=4096; round 6 experiment)
            static const int nv512 = getenv("RDM_GN1PASS_NV512") ? atoi(getenv("RDM_GN1PASS_NV512")) : 32;
            if (n <= (t == 1024 ? 16 : t == 512 ? nv512 : 32)) { t_ok = t; n_ok = n; break; }
        }
        if (!t_ok) continue;
        const long long el = (long long)g * cg * p.HW, dist = el > want ? el - want : want - el;
        if (best_d < 0 || dist < best_d) { best_d = dist; gpb = g; T = t_ok; nv = n_ok <= 4 ? 4 : n_ok <= 8 ? 8 : n_ok <= 16 ? 16 : n_ok <= 32 ? 32 : 52; }
    }
    return best_d >= 0;
}
static hipError_t launch_gn_onepass(const GnParams& p, int gpb, int T, int nv, hipStream_t st) {
    const int C = p.C0 + p.C1, cg = C / p.groups, SC = gpb * cg, VS = SC / 8, R = T / VS, nslice = p.groups / gpb;
    const size_t smb = ((size_t)(R + 1) * SC * 2 + (size_t)gpb * 2) * sizeof(float);
    const dim3 grid((unsigned)(p.B * nslice));
    auto go = [&](auto nvtag, auto ttag) -> hipError_t {
        constexpr int NV = decltype(nvtag)::value, TT = decltype(ttag)::value;
        static bool attr[RDM_MAX_DEVICES] = {};
        const int dev = rdm_cur_device();
        if (!attr[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)gn_onepass_kernel<NV, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return e;
            attr[dev] = true;
        }
        gn_onepass_kernel<NV, TT><<<grid, TT, smb, st>>>(p, gpb, nslice);
        return hipGetLastError();
    };
    if (smb > 96 * 1024) return hipErrorInvalidValue;
    auto by_t = [&](auto nvtag) -> hipError_t {
        if (T == 256) return go(nvtag, std::integral_constant<int, 256>{});
        if (T == 512) return go(nvtag, std::integral_constant<int, 512>{});
        if constexpr (decltype(nvtag)::value <= 16) return go(nvtag, std::integral_constant<int, 1024>{});
        return hipErrorInvalidValue;
    };
    switch (nv) {
        case 4: return by_t(std::integral_constant<int, 4>{});
        case 8: return by_t(std::integral_constant<int, 8>{});
        case 16: return by_t(std::integral_constant<int, 16>{});
        case 32: return by_t(std::integral_constant<int, 32>{});
        default: return T == 512 ? go(std::integral_constant<int, 52>{}, std::integral_constant<int, 512>{}) : hipErrorInvalidValue;
    }
}

static hipError_t gn_geometry(GnParams& p, int& R, int& threads) {
    const int C = p.C0 + p.C1;
    if (p.L0 <= 0) p.L0 = p.C0;
    if (p.L1 <= 0) p.L1 = p.C1;
    if (p.L0 > p.C0 || p.L1 > p.C1 || (p.L0 + p.L1) % p.groups || p.L0 % 8 || p.L1 % 8) return hipErrorInvalidValue;
    if (C % 8 || p.groups > 64 || (p.C0 % 8) || (p.C1 % 8)) return hipErrorInvalidValue;
    const int VC = C / 8;
    if (VC > 1024) return hipErrorInvalidValue;
    R = 256 / VC; if (R < 1) R = 1;
    threads = ((VC * R + 63) / 64) * 64;
    return hipSuccess;
}
hipError_t launch_gn_stats(GnParams p, hipStream_t st) {
    int R, threads;
    hipError_t e = gn_geometry(p, R, threads);
    if (e != hipSuccess) return e;
    const size_t sm1 = (size_t)(R + 1) * (p.C0 + p.C1) * 2 * sizeof(float);
    gn_stats_kernel<<<dim3(p.nchunk, p.B), threads, sm1, st>>>(p);
    return hipGetLastError();
}
hipError_t launch_groupnorm(GnParams p, hipStream_t st) {
    int R, threads;
    hipError_t e = gn_geometry(p, R, threads);
    if (e != hipSuccess) return e;
    {
        int gpb, T, nv;
        if (p.out && gn_onepass_plan(p, gpb, T, nv)) return launch_gn_onepass(p, gpb, T, nv, st);
    }
    // RDM_GN_RANGE_MB = n (default 0 = off): walk a tensor larger than 2n MB in sample ranges of n MB, statistics then apply per range, so
    // that the apply pass could re-read from the 256 MB memory-side cache what the statistics pass of the SAME range pulled in (64 x 64
    // level at B' = 128: 200 .. 600 MB per GroupNorm).  Measured round 5 (profiles/r05e_gn_sample_ranges_ab.log): 24 / 48 / 96 MB ranges
    // = -7.6 % / -2.9 % / -1.0 % on the headline -- the under-filled launches and their tails cost more than any reuse returns.  Off.
    static const long long range_mb = getenv("RDM_GN_RANGE_MB") ? atoll(getenv("RDM_GN_RANGE_MB")) : 0;
    const long long per_sample = (long long)p.HW * (p.C0 + p.C1) * 2;
    int nb = p.B;
    if (range_mb > 0 && per_sample * p.B > 2 * range_mb * (1 << 20)) { nb = (int)(range_mb * (1 << 20) / per_sample); if (nb < 1) nb = 1; }
    const size_t sm1 = (size_t)(R + 1) * (p.C0 + p.C1) * 2 * sizeof(float);
    for (int b0 = 0; b0 < p.B; b0 += nb) {
        GnParams q = p; q.b0 = p.b0 + b0;
        const int n = p.B - b0 < nb ? p.B - b0 : nb;
        gn_stats_kernel<<<dim3(q.nchunk, n), threads, sm1, st>>>(q);
        // apply: ~64 rows per thread-row, at least ~2k blocks across the range
        int nblk = (p.HW + 64 * R - 1) / (64 * R);
        const int min_blocks = (2048 + n - 1) / n;
        if (nblk < min_blocks) nblk = min_blocks;
        if (nblk > p.HW) nblk = p.HW;
        if (nblk < 1) nblk = 1;
        gn_apply_kernel<<<dim3(nblk, n), threads, 0, st>>>(q);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ LayerNorm
// one wave per row; input bf16 or f32, output bf16. C even, C <= 128*NP. Row cached in registers
// (compile-time indexed so nothing goes to scratch), exact two-pass variance.
template <typename TIN, typename TOUT, int NP>
__global__ __launch_bounds__(256) void layernorm_kernel(const TIN* x, const float* gamma, const float* beta,
                                                        TOUT* out, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const TIN* xr = x + row * C;
    float v[NP * 2];
    const int npair = C >> 1;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int pi = lane + j * 64;
        float a = 0.f, b = 0.f;
        if (pi < npair) {
            if constexpr (sizeof(TIN) == 2) {
                const uint32_t u = *(const uint32_t*)(xr + pi * 2);
                a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
            } else {
                const float2 f = *(const float2*)(xr + pi * 2);
                a = f.x; b = f.y;
            }
        }
        v[j * 2] = a; v[j * 2 + 1] = b; s += a + b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        if (lane + j * 64 < npair) {
            const float d0 = v[j * 2] - mean, d1 = v[j * 2 + 1] - mean;
            q += d0 * d0 + d1 * d1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / C + eps);
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int pi = lane + j * 64;
        if (pi < npair) {
            const int c = pi * 2;
            const float y0 = (v[j * 2] - mean) * rstd * gamma[c] + beta[c];
            const float y1 = (v[j * 2 + 1] - mean) * rstd * gamma[c + 1] + beta[c + 1];
            if constexpr (sizeof(TOUT) == 2) *(uint32_t*)(out + row * C + c) = pack2bf(y0, y1);
            else *(float2*)(out + row * C + c) = make_float2(y0, y1);
        }
    }
}

// bf16 -> bf16 rows with C % 8 == 0 (every LayerNorm of the UNet): 16-byte loads and stores, 8 channels per lane.  A store
// costs ~70 cycles per wave-instruction whatever its width, so the 4-byte-per-lane kernel above is bound by its store COUNT
// (256 B per instruction ~ 1.9 TB/s chip-wide); this one moves 1 KiB per instruction.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bf16x8_kernel(const bf16_t* x, const float* gamma, const float* beta,
                                                               bf16_t* out, int M, int C, float eps, int Clog) {
    // Clog <= C: the row's tail [Clog, C) is zero padding (Clog % 8 == 0): statistics over Clog values; the tail's gamma / beta
    // are zero, so it is written back as zero.
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = x + row * C;
    const int nvec = C >> 3, nlog = Clog >> 3;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int vi = lane + j * 64;
        if (vi < nvec) {
            const bf16x8 d = *(const bf16x8*)(xr + vi * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) { v[j][e] = bf2f((bf16_t)d[e]); s += v[j][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[j][e] = 0.f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / Clog;             // the padded tail contributed zeros to s
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++)
        if (lane + j * 64 < nlog) {
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = v[j][e] - mean; q += d * d; }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / Clog + eps);
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int vi = lane + j * 64;
        if (vi < nvec) {
            const float4 g0 = *(const float4*)(gamma + vi * 8), g1 = *(const float4*)(gamma + vi * 8 + 4);
            const float4 b0 = *(const float4*)(beta + vi * 8), b1 = *(const float4*)(beta + vi * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = pack2bf((v[j][2 * e] - mean) * rstd * gg[2 * e] + bb[2 * e], (v[j][2 * e + 1] - mean) * rstd * gg[2 * e + 1] + bb[2 * e + 1]);
            *(uint4*)(out + row * C + vi * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <int NP>
static void ln_dispatch(const void* x, int in_is_f32, const float* g, const float* b, void* out, int out_is_f32, int M, int C,
                        float eps, hipStream_t st) {
    const int grid = (M + 3) / 4;
    if (in_is_f32) {
        if (out_is_f32) layernorm_kernel<float, float, NP><<<grid, 256, 0, st>>>((const float*)x, g, b, (float*)out, M, C, eps);
        else layernorm_kernel<float, bf16_t, NP><<<grid, 256, 0, st>>>((const float*)x, g, b, (bf16_t*)out, M, C, eps);
    } else {
        if (out_is_f32) layernorm_kernel<bf16_t, float, NP><<<grid, 256, 0, st>>>((const bf16_t*)x, g, b, (float*)out, M, C, eps);
        else layernorm_kernel<bf16_t, bf16_t, NP><<<grid, 256, 0, st>>>((const bf16_t*)x, g, b, (bf16_t*)out, M, C, eps);
    }
}

hipError_t launch_layernorm(const void* x, int in_is_f32, const float* gamma, const float* beta, void* out, int out_is_f32,
                            int M, int C, float eps, hipStream_t st, int Clog) {
    if (C % 2 || C > 4096) return hipErrorInvalidValue;
    if (Clog < 0) Clog = C;
    if (Clog > C || Clog % 8 || (Clog != C && (in_is_f32 || out_is_f32 || C % 8 || C > 1024))) return hipErrorInvalidValue;   // padded rows: the bf16 vector kernel only
    static const int no_vec = getenv("RDM_LN_NOVEC") ? atoi(getenv("RDM_LN_NOVEC")) : 0;
    if (Clog != C && (no_vec || ((size_t)x % 16) || ((size_t)out % 16) || ((size_t)gamma % 16) || ((size_t)beta % 16))) return hipErrorInvalidValue;
    if (!no_vec && !in_is_f32 && !out_is_f32 && C % 8 == 0 && C <= 1024 && ((size_t)x % 16 == 0) && ((size_t)out % 16 == 0) &&
        ((size_t)gamma % 16 == 0) && ((size_t)beta % 16 == 0)) {
        const int grid = (M + 3) / 4;
        if (C <= 512) layernorm_bf16x8_kernel<1><<<grid, 256, 0, st>>>((const bf16_t*)x, gamma, beta, (bf16_t*)out, M, C, eps, Clog);
        else layernorm_bf16x8_kernel<2><<<grid, 256, 0, st>>>((const bf16_t*)x, gamma, beta, (bf16_t*)out, M, C, eps, Clog);
        return hipGetLastError();
    }
    if (C <= 1024) ln_dispatch<8>(x, in_is_f32, gamma, beta, out, out_is_f32, M, C, eps, st);
    else ln_dispatch<32>(x, in_is_f32, gamma, beta, out, out_is_f32, M, C, eps, st);
    return hipGetLastError();
}

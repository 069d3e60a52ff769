// Attention kernels (gfx950).
// Replaces rdm/modules/attention.py:42-74 (CrossAttention.forward: einsum QK^T * scale -> softmax ->
// einsum AV, d_head = 32) for both uses in BasicTransformerBlock (attention.py:92-96), and
// nn.MultiheadAttention inside CLIP's ResidualAttentionBlock (custom_clip/model.py:166-187).
//
// 1) flash_d32: self-attention, d_head = 32, n % 32 == 0.  Never materialises the n x n scores
//    (the reference moves 475 M score elements per sample per forward).  One wave owns 32 query rows.
//    Both contractions run on 32x32x16 bf16 MFMA in the *swapped* form:
//        S^T[kv][q] = K . Q^T      (A = K rows, B = Q rows)       -> a lane holds 16 scores of ONE query
//        O^T[d][q]  = V^T . P^T    (A = V^T rows, B = P^T = the lane's own exp'd scores)
//    so softmax statistics are lane-local (one cross-half shuffle per tile), the rescale of O is a
//    per-lane scalar, and P never leaves registers.  The K rows of a tile are loaded in a permuted
//    order (bits 2,3 of the in-tile index swapped) so that the 8 scores a lane holds per MFMA k-step
//    are 8 CONSECUTIVE keys: the V^T operand is then one 16-byte load per lane and no LDS, no
//    transpose and no cross-lane permute is needed.  V^T ([B, C, n]) is produced directly by the
//    projection GEMM (swapped operands), K/V tiles are L2-resident (<= 64 KB per head).
// 2) small_attention<D>: generic VALU kernel for short key sequences (cross-attention over the k
//    retrieved neighbours, n_kv = k <= 16; CLIP n = 77/50 with optional causal mask; odd sizes).
#include "kernels.h"


__global__ __launch_bounds__(256) void flash_d32_kernel(FlashParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const int q0 = (blockIdx.x * nw + wave) * 32;
    if (q0 >= p.n) return;
    const int h = blockIdx.y, b = blockIdx.z;
    const int l31 = lane & 31, hf = lane >> 5;
    const long long tok0 = (long long)b * p.n;

    // Q as the B operand of S^T = K.Q^T : lane (q = l31) holds d = ks*16 + hf*8 .. +8
    bf16x8 qf[2];
    {
        const bf16_t* qp = p.q + (tok0 + q0 + l31) * p.ldq + h * 32 + hf * 8;
        qf[0] = *(const bf16x8*)(qp);
        qf[1] = *(const bf16x8*)(qp + 16);
    }
    // in-tile key permutation: position p holds key pi(p) = p with bits 2 and 3 swapped
    const int pos = l31;
    const int pi = (pos & ~0xc) | ((pos & 4) << 1) | ((pos & 8) >> 1);
    const bf16_t* kbase = p.k + (tok0 + pi) * p.ldk + h * 32 + hf * 8;
    const bf16_t* vbase = p.vt + ((long long)b * p.C + h * 32 + l31) * p.n + hf * 8;

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int kv0 = 0; kv0 < p.n; kv0 += 32) {
        const bf16_t* kp = kbase + (long long)kv0 * p.ldk;
        const bf16x8 k0 = *(const bf16x8*)(kp);
        const bf16x8 k1 = *(const bf16x8*)(kp + 16);
        const bf16x8 v0 = *(const bf16x8*)(vbase + kv0);
        const bf16x8 v1 = *(const bf16x8*)(vbase + kv0 + 16);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.f;
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[1], s, 0, 0, 0);
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; r++) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(m, mx * p.scale_log2e);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        float ps = 0.f;
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { pr[r] = __builtin_amdgcn_exp2f(s[r] * p.scale_log2e - mnew); ps += pr[r]; }
        l = l * alpha + ps;
        m = mnew;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] *= alpha;
        union { bf16x8 v; uint32_t u[4]; } pb0, pb1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            pb0.u[i] = pack2bf(pr[2 * i], pr[2 * i + 1]);
            pb1.u[i] = pack2bf(pr[8 + 2 * i], pr[8 + 2 * i + 1]);
        }
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb0.v, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb1.v, o, 0, 0, 0);
    }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    bf16_t* op = p.out + (tok0 + q0 + l31) * p.ldo + h * 32 + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint2 w;
        w.x = pack2bf(o[g * 4 + 0] * inv, o[g * 4 + 1] * inv);
        w.y = pack2bf(o[g * 4 + 2] * inv, o[g * 4 + 3] * inv);
        *(uint2*)(op + 8 * g) = w;
    }
}

// ---- LDS-shared variant (n % 64 == 0): the 4 waves of a block (128 query rows) share every 64-key chunk of K and V^T
// through LDS instead of each fetching it from L2 (4x less L2 traffic: at n = 1024 the per-wave version moves 6.3 GB
// per layer through L2).  Chunks arrive by 16-byte LDS-DMA into a 4-deep ring, requested THREE chunks ahead with
// counted vmcnt waits (the K/V tensors of a layer do not fit L2, so a chunk pays a full HBM/MALL round trip: one chunk
// of look-ahead left the kernel latency-bound).  The ring image is lane-linear, so the bank-conflict swizzle sits on
// the source address and again on the ds_read_b128 fragment reads.
// VROW: V arrives token-major (v / ldv: a column block of the fused q|k|v projection) and the slot's second half holds V [64 keys][32]
// instead of V^T; the PV operand (8 consecutive keys of one channel per lane) is then gathered by the hardware transpose read
// ds_read_b64_tr_b16: each 16-lane group reads a [4 keys][16 channels] block, lane i the 4 channels 4(i&3).. of key i>>2, and lane j
// receives channel j of the 4 keys (probed in tools/ubench/tr_probe.hip).  No separate V^T GEMM per layer.
template <bool VROW>
__global__ __launch_bounds__(256, 2) void flash_d32_lds_kernel(FlashParams p) {
    constexpr int DEPTH = 4, CH = 8192;                               // per slot: K [64][32] bf16 | V^T [32][64] bf16 (VROW: V [64][32])
    __shared__ __attribute__((aligned(16))) char lds[DEPTH * CH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware work mapping: workgroups go round-robin over the 8 XCDs in linear id order (x fastest), so the query blocks of one
    // (sample, head) -- which all stream the same K / V -- would land on 8 different L2s and fetch K / V from HBM 8 times (measured
    // 708 MB per launch against 400 MB of tensors: the kernel is HBM bound).  Remapped: the blocks with linear id = 8 s + x form XCD
    // x's queue; consecutive queue slots are the query blocks of ONE (sample, head) group, groups dealt to the XCDs round-robin.
    int xb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    {
        const int gx = gridDim.x, G = gridDim.y * gridDim.z;
        if ((G & 7) == 0 && p.xcd_remap) {
            const int lin = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
            const int xcd = lin & 7, slot = lin >> 3;
            const int grp = (slot / gx) * 8 + xcd;
            xb = slot - (slot / gx) * gx; h = grp % gridDim.y; b = grp / gridDim.y;
        }
    }
    const int q0 = (xb * 4 + wave) * 32;
    const bool active = q0 < p.n;
    const int l31 = lane & 31, hf = lane >> 5;
    const long long tok0 = (long long)b * p.n;

    bf16x8 qf[2];
    if (active) {
        const bf16_t* qp = p.q + (tok0 + q0 + l31) * p.ldq + h * 32 + hf * 8;
        qf[0] = *(const bf16x8*)(qp); qf[1] = *(const bf16x8*)(qp + 16);
    }
    // loader roles (LDS image is lane-linear: thread t fills bytes [16t, 16t+16) of the K block and of the V^T block):
    //   K block: row kr = t>>2 (64 B rows), physical piece t&3 holds source piece (t&3) ^ ((kr>>2)&3)
    //   V block: row vr = t>>3 (128 B rows), physical piece t&7 holds source piece (t&7) ^ ((vr>>1)&7)
    const int kr = tid >> 2, vr = tid >> 3;
    const int ksp = (tid & 3) ^ ((kr >> 2) & 3), vsp = (tid & 7) ^ ((vr >> 1) & 7);
    const bf16_t* kg = p.k + (tok0 + kr) * p.ldk + h * 32 + ksp * 8;
    const bf16_t* vg = VROW ? p.v + (tok0 + kr) * p.ldv + h * 32 + (tid & 3) * 8          // V rows: same shape as K's, unswizzled
                            : p.vt + ((long long)b * p.C + h * 32 + vr) * p.n + vsp * 8;
    const int nchunk = p.n >> 6;
    auto request = [&](int c) {
        char* slot = lds + (c & (DEPTH - 1)) * CH;
        glds16(kg + (long long)c * 64 * p.ldk, slot + wave * 1024);
        if (VROW) glds16(vg + (long long)c * 64 * p.ldv, slot + 4096 + wave * 1024);
        else glds16(vg + c * 64, slot + 4096 + wave * 1024);
    };
    // transpose-read address of this lane inside a slot's V image: key 8 hf + (i >> 2), channels 16 g + 4 (i & 3)  (i = lane & 15, g = (lane >> 4) & 1)
    const uint32_t vtr = 4096 + (8 * hf + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int pi = (l31 & ~0xc) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // key permutation inside a 32-key sub-tile

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int c = 0; c < DEPTH - 1 && c < nchunk; c++) request(c);
    for (int c = 0; c < nchunk; c++) {
        // chunk c landed once at most 2 * (requests issued after it) loads remain in flight
        const int ahead = min(nchunk - 1 - c, DEPTH - 2);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                     // chunk c visible to all waves; slot of chunk c-1 is free again
        if (c + DEPTH - 1 < nchunk) request(c + DEPTH - 1);
        const char* L = lds + (c & (DEPTH - 1)) * CH;
        if (active) {
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const int krow = sub * 32 + pi;
                const bf16x8 k0 = *(const bf16x8*)(L + krow * 64 + (((0 + hf) ^ ((krow >> 2) & 3)) << 4));
                const bf16x8 k1 = *(const bf16x8*)(L + krow * 64 + (((2 + hf) ^ ((krow >> 2) & 3)) << 4));
                bf16x8 v0, v1;
                if (VROW) {       // keys sub*32 + 8 hf + 0..7 (v0) and + 16 (v1) of channel l31: two 4-key transpose reads each
                    typedef __attribute__((ext_vector_type(4))) short s16x4;
                    const LDS_AS char* vb = (const LDS_AS char*)(L + vtr + sub * 2048);
                    union { bf16x8 v; s16x4 h[2]; } a0, a1;
                    a0.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(vb));
                    a0.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(vb + 256));
                    a1.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(vb + 1024));
                    a1.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(vb + 1280));
                    v0 = a0.v; v1 = a1.v;
                } else {
                    const int vch = sub * 4 + hf;                  // 16-byte piece index of keys sub*32 + j*16 + hf*8
                    v0 = *(const bf16x8*)(L + 4096 + l31 * 128 + (((vch) ^ ((l31 >> 1) & 7)) << 4));
                    v1 = *(const bf16x8*)(L + 4096 + l31 * 128 + (((vch + 2) ^ ((l31 >> 1) & 7)) << 4));
                }
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; r++) s[r] = 0.f;
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[1], s, 0, 0, 0);
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; r++) mx = fmaxf(mx, s[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                // raw v_exp_f32 (exp2f() expands to ~6 instructions of denormal range handling per call; inputs here are
                // <= 0 and underflow to 0 is exactly what softmax wants).  The O rescale is skipped while the running
                // max of every row in the wave is unchanged (alpha == 1 exactly).
                const float mnew = fmaxf(m, mx * p.scale_log2e);
                float ps = 0.f, pr[16];
#pragma unroll
                for (int r = 0; r < 16; r++) { pr[r] = __builtin_amdgcn_exp2f(s[r] * p.scale_log2e - mnew); ps += pr[r]; }
                if (__builtin_amdgcn_ballot_w64(mnew != m) != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(m - mnew);      // m = -inf on the first tile -> alpha = 0
                    l *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; r++) o[r] *= alpha;
                    m = mnew;
                }
                l += ps;
                union { bf16x8 v; uint32_t u[4]; } pb0, pb1;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    pb0.u[i] = cvt_pk_bf16(pr[2 * i], pr[2 * i + 1]);
                    pb1.u[i] = cvt_pk_bf16(pr[8 + 2 * i], pr[8 + 2 * i + 1]);
                }
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb0.v, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb1.v, o, 0, 0, 0);
            }
        }
    }
    if (!active) return;
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    bf16_t* op = p.out + (tok0 + q0 + l31) * p.ldo + h * 32 + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint2 w;
        w.x = cvt_pk_bf16(o[g * 4 + 0] * inv, o[g * 4 + 1] * inv);
        w.y = cvt_pk_bf16(o[g * 4 + 2] * inv, o[g * 4 + 3] * inv);
        *(uint2*)(op + 8 * g) = w;
    }
}

hipError_t launch_flash_d32(const FlashParams& p, int heads, int batch, hipStream_t st) {
    if (p.n % 32 != 0 || p.C != heads * 32) return hipErrorInvalidValue;
    if (!p.vt && (!p.v || p.n % 64 != 0)) return hipErrorInvalidValue;          // token-major V: LDS-shared kernel only
    if (p.n % 64 == 0) {
        dim3 grid((p.n / 32 + 3) / 4, heads, batch);
        static const int old = getenv("RDM_FLASH_OLD") ? atoi(getenv("RDM_FLASH_OLD")) : 0;
        FlashParams q = p; q.xcd_remap = old ? 0 : 1;          // RDM_FLASH_OLD=1: plain blockIdx mapping (A/B)
        if (q.v && !q.vt) flash_d32_lds_kernel<true><<<grid, 256, 0, st>>>(q);
        else flash_d32_lds_kernel<false><<<grid, 256, 0, st>>>(q);
        return hipGetLastError();
    }
    int nw = p.n / 32; if (nw > 4) nw = 4;
    dim3 grid((p.n / 32 + nw - 1) / nw, heads, batch);
    flash_d32_kernel<<<grid, nw * 64, 0, st>>>(p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ fused skinny cross-attention
// scores GEMM (K = C, N <= 128) + group softmax + output GEMM (K <= 128, N = C) + bias + residual in one launch: the probabilities
// never leave registers.  As two batched GEMMs both halves were latency-bound (2 - 15 k-steps per tile behind a cold pipeline:
// 78 / 54 / 56 us per layer at the 32x32 / 16x16 / 8x8 levels for 184 / 43 / 17 MB of traffic).
// A block = 32 rows of one sample, 4 waves.  Phase 1 splits K: wave w forms the partial scores of channels [w C/4, (w+1) C/4); the
// x rows are read straight from global memory in MFMA layout (16 bytes per lane and 16-k step), G comes FRAGMENT-ORDERED (xattn_pack_g:
// one step's operand of a 32-row block = 1 KiB contiguous; row-major it was 64 cache lines per load instruction and the CU's address
// path, not latency, set the time); the four partial tiles meet in LDS and every wave sums them (swapped product: lane = x row,
// registers = score columns, so a softmax group is 1 / 2 / 4 adjacent registers of one lane).  Phase 2 splits N: wave w owns the
// 32-channel blocks w, w+4, ..; the probabilities (bf16, the rounding the two-GEMM form had) are the B operand as they sit, U comes
// fragment-ordered with the matching column order and with its rows permuted so that a lane ends up with 16 CONSECUTIVE channels of
// its row (two 16-byte residual loads / stores instead of four 8-byte ones); the next block's operands are requested before the
// current block's MFMAs.
__device__ __forceinline__ int xattn_chan(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }   // MFMA row i -> channel inside a 32-block
// Gp[b][ks][jb][lane][8] = G[b][32 jb + (lane & 31)][16 ks + 8 (lane >> 5) + e]
__global__ void xattn_pack_g_kernel(const bf16_t* G, bf16_t* Gp, int B, int NP, int C) {
    const long long total = (long long)B * NP * C / 8;
    const int nj = NP / 32, nk = C / 16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63); long long r = i >> 6;
        const int jb = (int)(r % nj); r /= nj;
        const int ks = (int)(r % nk); const int b = (int)(r / nk);
        *(uint4*)(Gp + i * 8) = *(const uint4*)(G + ((long long)b * NP + 32 * jb + (lane & 31)) * C + 16 * ks + 8 * (lane >> 5));
    }
}
// Up[b][cb][st][lane][8] = U[b][32 cb + chan(lane & 31)][16 st + 4 (lane >> 5) + {0..3, 8..11}]
__global__ void xattn_pack_u_kernel(const bf16_t* U, bf16_t* Up, int B, int NP, int C) {
    const long long total = (long long)B * NP * C / 8;
    const int ns = NP / 16, nc = C / 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63); long long r = i >> 6;
        const int st = (int)(r % ns); r /= ns;
        const int cb = (int)(r % nc); const int b = (int)(r / nc);
        const bf16_t* src = U + ((long long)b * C + 32 * cb + xattn_chan(lane & 31)) * NP + 16 * st + 4 * (lane >> 5);
        uint4 v; const uint2 lo = *(const uint2*)src, hi = *(const uint2*)(src + 8);
        v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
        *(uint4*)(Up + i * 8) = v;
    }
}
hipError_t launch_xattn_pack(const bf16_t* G, const bf16_t* U, bf16_t* Gp, bf16_t* Up, int B, int NP, int C, hipStream_t st) {
    if (NP % 32 != 0 || C % 32 != 0) return hipErrorInvalidValue;
    const long long total = (long long)B * NP * C / 8;
    int grid = (int)((total + 255) / 256); if (grid > 16384) grid = 16384; if (grid < 1) grid = 1;
    xattn_pack_g_kernel<<<grid, 256, 0, st>>>(G, Gp, B, NP, C);
    xattn_pack_u_kernel<<<grid, 256, 0, st>>>(U, Up, B, NP, C);
    return hipGetLastError();
}

// XCD-aware block order: workgroups go round-robin over the 8 XCDs in id order, so the n / 32 blocks of ONE sample -- which all stream
// that sample's G and U images -- would pull them through 8 different L2s (at the 16x16 level: 150 MB of fabric traffic for 38 MB of
// activations).  Remapped so that a sample's blocks are consecutive slots of one XCD's queue.
__device__ __forceinline__ int xattn_block(int n) {
    const int bps = n >> 5, lin = blockIdx.x;
    if (gridDim.x % (8 * bps) != 0) return lin;
    const int xcd = lin & 7, slot = lin >> 3;
    return (slot / bps) * (8 * bps) + xcd * bps + (slot % bps);
}

// softmax over groups of `group` (1, 2, 4) adjacent entries of the 16 scores a lane holds for one 32-column block
__device__ __forceinline__ void xattn_group_softmax(float* s, int group) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float* g = s + 4 * q;
        if (group == 4) {
            const float m = fmaxf(fmaxf(g[0], g[1]), fmaxf(g[2], g[3]));
            float e[4], sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) { e[i] = __expf(g[i] - m); sum += e[i]; }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int i = 0; i < 4; i++) g[i] = e[i] * inv;
        } else if (group == 2) {
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                const float m = fmaxf(g[2 * h2], g[2 * h2 + 1]);
                const float e0 = __expf(g[2 * h2] - m), e1 = __expf(g[2 * h2 + 1] - m);
                const float inv = 1.0f / (e0 + e1);
                g[2 * h2] = e0 * inv; g[2 * h2 + 1] = e1 * inv;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) g[i] = 1.0f;
        }
    }
}

template <int NB>        // 32-column score blocks in use: ceil(ncols / 32)
__global__ __launch_bounds__(256) void xattn_fused_kernel(XattnParams p) {
    __shared__ __attribute__((aligned(16))) float part[4 * NB * 4 * 64 * 4];       // [wave][jb*4 + q][lane][4]
    constexpr int SB = NB == 4 ? 5 : 3;                                             // k-steps requested together
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, NJ = p.NP >> 5;
    const long long row0 = (long long)xattn_block(p.n) * 32;
    const int b = (int)(row0 / p.n);
    // ---- phase 1: partial scores over this wave's K quarter
    const int nsteps = C >> 6;                               // 16-k steps in a quarter
    const bf16_t* xr = p.x + (row0 + l31) * C + w * (C >> 2) + 8 * hf;
    const bf16_t* gr = p.G + (((long long)b * (C >> 4) + (long long)w * nsteps) * NJ * 64 + lane) * 8;      // + (s * NJ + jb) * 512
    f32x16 acc[NB];
#pragma unroll
    for (int jb = 0; jb < NB; jb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[jb][r] = 0.f;
    int s0 = 0;
    for (; s0 + SB <= nsteps; s0 += SB) {
        bf16x8 xb[SB], gb[SB][NB];
#pragma unroll
        for (int u = 0; u < SB; u++) {
            xb[u] = *(const bf16x8*)(xr + (s0 + u) * 16);
#pragma unroll
            for (int jb = 0; jb < NB; jb++) gb[u][jb] = *(const bf16x8*)(gr + ((long long)(s0 + u) * NJ + jb) * 512);
        }
#pragma unroll
        for (int u = 0; u < SB; u++)
#pragma unroll
            for (int jb = 0; jb < NB; jb++) acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb[u][jb], xb[u], acc[jb], 0, 0, 0);
    }
    for (; s0 < nsteps; s0++) {
        const bf16x8 xb = *(const bf16x8*)(xr + s0 * 16);
#pragma unroll
        for (int jb = 0; jb < NB; jb++) {
            const bf16x8 gb = *(const bf16x8*)(gr + ((long long)s0 * NJ + jb) * 512);
            acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb, xb, acc[jb], 0, 0, 0);
        }
    }
    // first output block's operands: requested before the exchange so that their latency hides behind it
    const int ncb = C >> 5;
    const bf16_t* ub = p.U + ((long long)b * ncb * (2 * NJ) * 64 + lane) * 8;        // + (cb * 2 NJ + st) * 512
    struct Blk { bf16x8 uf[2 * NB]; uint4 rv[2]; f32x4 bv[4]; };
    auto request = [&](Blk& k, int cb) {
#pragma unroll
        for (int st = 0; st < 2 * NB; st++) k.uf[st] = *(const bf16x8*)(ub + ((long long)cb * (2 * NJ) + st) * 512);
        const long long o = (row0 + l31) * C + cb * 32 + 16 * hf;
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) k.rv[h2] = p.res ? *(const uint4*)(p.res + o + 8 * h2) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int q = 0; q < 4; q++) k.bv[q] = p.bias ? *(const f32x4*)(p.bias + cb * 32 + 16 * hf + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    Blk k0, k1;
    if (w < ncb) request(k0, w);
#pragma unroll
    for (int jb = 0; jb < NB; jb++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 v = {acc[jb][4 * q], acc[jb][4 * q + 1], acc[jb][4 * q + 2], acc[jb][4 * q + 3]};
            *(f32x4*)(part + (((w * NB + jb) * 4 + q) * 64 + lane) * 4) = v;
        }
    __syncthreads();
    // ---- sum of the four partials, group softmax, probabilities as the B operand of phase 2
    bf16x8 pf[2 * NB];
#pragma unroll
    for (int jb = 0; jb < NB; jb++) {
        float s[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 t = *(const f32x4*)(part + (((0 * NB + jb) * 4 + q) * 64 + lane) * 4);
#pragma unroll
            for (int ww = 1; ww < 4; ww++) t += *(const f32x4*)(part + (((ww * NB + jb) * 4 + q) * 64 + lane) * 4);
            s[4 * q] = t[0]; s[4 * q + 1] = t[1]; s[4 * q + 2] = t[2]; s[4 * q + 3] = t[3];
        }
        xattn_group_softmax(s, p.group);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
            for (int i = 0; i < 4; i++) pk.u[i] = cvt_pk_bf16(s[8 * half + 2 * i], s[8 * half + 2 * i + 1]);
            pf[2 * jb + half] = pk.v;
        }
    }
    // ---- phase 2: this wave's 32-channel output blocks, operands one block ahead
    auto finish = [&](const Blk& k, int cb) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 2 * NB; st++) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k.uf[st], pf[st], o, 0, 0, 0);
        const long long ob = (row0 + l31) * C + cb * 32 + 16 * hf;          // registers r = 0..15 <-> channels 16 hf + r (xattn_chan)
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            const uint32_t rr[4] = {k.rv[h2].x, k.rv[h2].y, k.rv[h2].z, k.rv[h2].w};
            uint32_t wv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 8 * h2 + 2 * i;
                const float a0 = o[r] + k.bv[r >> 2][r & 3] + __uint_as_float(rr[i] << 16);
                const float a1 = o[r + 1] + k.bv[(r + 1) >> 2][(r + 1) & 3] + __uint_as_float(rr[i] & 0xffff0000u);
                wv[i] = cvt_pk_bf16(a0, a1);
            }
            *(uint4*)(p.out + ob + 8 * h2) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
    };
    for (int cb = w; cb < ncb; cb += 8) {
        if (cb + 4 < ncb) request(k1, cb + 4);
        finish(k0, cb);
        if (cb + 4 < ncb) {
            if (cb + 8 < ncb) request(k0, cb + 8);
            finish(k1, cb + 4);
        }
    }
}

// ---- the same with the LayerNorm in front folded in (out = softmax_groups(LN(x) G^T) U^T + bias + x: norm2 + attn2 + the residual of
// BasicTransformerBlock, rdm/modules/attention.py:238): the block's 32 raw rows arrive ONCE, coalesced, in an LDS tile (padded rows:
// conflict-free 16-byte reads at one row per lane); the row statistics are two exchanges of per-wave partial sums (mean, then the
// centred squares, as layernorm_bf16x8_kernel forms them); the normalised operand is formed in registers on the way to the MFMA; the
// residual is read from the tile, the result written back into it, and the tile leaves coalesced.  Every global access of the
// activations is now whole rows (the MFMA-layout accesses above touch 32 - 64 cache lines per instruction).  The score exchange goes
// block by block through one 16 KiB buffer.
template <int NB>
__global__ __launch_bounds__(256) void xattn_ln_fused_kernel(XattnParams p) {
    extern __shared__ __attribute__((aligned(16))) char xsm[];
    constexpr int SB = 3;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, NJ = p.NP >> 5, RS = 2 * C + 16;
    char* T = xsm;                                           // [32 rows][RS bytes]
    float* part = (float*)(xsm + 32 * RS);                   // [wave][q][lane][4]
    float* stat = part + 4 * 4 * 64 * 4;                     // [wave][32 rows]
    const long long row0 = (long long)xattn_block(p.n) * 32;
    const int b = (int)(row0 / p.n);
    // the first score operands are on their way while the tile arrives
    const int nsteps = C >> 6;
    const bf16_t* gr = p.G + (((long long)b * (C >> 4) + (long long)w * nsteps) * NJ * 64 + lane) * 8;
    struct GB { bf16x8 g[SB][NB]; };
    auto gload = [&](GB& t, int sb) {
#pragma unroll
        for (int u = 0; u < SB; u++)
#pragma unroll
            for (int jb = 0; jb < NB; jb++) t.g[u][jb] = *(const bf16x8*)(gr + ((long long)(sb + u) * NJ + jb) * 512);
    };
    GB gA, gB;
    if (SB <= nsteps) gload(gA, 0);
    // ---- the tile in: 16-byte pieces, consecutive threads along a row
    const int ppr = C >> 3, npc = (32 * ppr) >> 8;
    struct Piece { long long g; int l; };
    auto piece = [&](int it) {       // piece `it` of this thread (clamped: a repeated piece is harmless)
        const int id = min(it, npc - 1) * 256 + tid, row = id / ppr, pc = id - row * ppr;
        return Piece{(row0 + row) * C + pc * 8, row * RS + pc * 16};
    };
    for (int it0 = 0; it0 < npc; it0 += 6) {                      // six requests in flight per thread
        const Piece q0 = piece(it0), q1 = piece(it0 + 1), q2 = piece(it0 + 2), q3 = piece(it0 + 3), q4 = piece(it0 + 4), q5 = piece(it0 + 5);
        const uint4 v0 = *(const uint4*)(p.x + q0.g), v1 = *(const uint4*)(p.x + q1.g), v2 = *(const uint4*)(p.x + q2.g);
        const uint4 v3 = *(const uint4*)(p.x + q3.g), v4 = *(const uint4*)(p.x + q4.g), v5 = *(const uint4*)(p.x + q5.g);
        *(uint4*)(T + q0.l) = v0; *(uint4*)(T + q1.l) = v1; *(uint4*)(T + q2.l) = v2;
        *(uint4*)(T + q3.l) = v3; *(uint4*)(T + q4.l) = v4; *(uint4*)(T + q5.l) = v5;
    }
    __syncthreads();
    // ---- row statistics over this wave's K quarter (lane = row, half-wave = 8-channel piece of every 16)
    const char* xl = T + l31 * RS + (w * (C >> 2) + 8 * hf) * 2;
    float sm = 0.f;
    for (int s0 = 0; s0 < nsteps; s0++) {
        const bf16x8 x8 = *(const bf16x8*)(xl + s0 * 32);
#pragma unroll
        for (int e = 0; e < 8; e++) sm += bf2f((bf16_t)x8[e]);
    }
    sm += __shfl_xor(sm, 32);
    if (hf == 0) stat[w * 32 + l31] = sm;
    __syncthreads();
    const float mean = (stat[l31] + stat[32 + l31] + stat[64 + l31] + stat[96 + l31]) / (float)C;
    __syncthreads();
    float sq = 0.f;
    for (int s0 = 0; s0 < nsteps; s0++) {
        const bf16x8 x8 = *(const bf16x8*)(xl + s0 * 32);
#pragma unroll
        for (int e = 0; e < 8; e++) { const float d = bf2f((bf16_t)x8[e]) - mean; sq += d * d; }
    }
    sq += __shfl_xor(sq, 32);
    if (hf == 0) stat[w * 32 + l31] = sq;
    __syncthreads();
    const float rstd = rsqrtf((stat[l31] + stat[32 + l31] + stat[64 + l31] + stat[96 + l31]) / (float)C + p.ln_eps);
    // ---- phase 1: partial scores of LayerNorm(x) over the quarter
    const float* gk = p.ln_g + w * (C >> 2) + 8 * hf;
    const float* bk = p.ln_b + w * (C >> 2) + 8 * hf;
    f32x16 acc[NB];
#pragma unroll
    for (int jb = 0; jb < NB; jb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[jb][r] = 0.f;
    auto normed = [&](int st) {
        const bf16x8 x8 = *(const bf16x8*)(xl + st * 32);
        const f32x4 g0 = *(const f32x4*)(gk + st * 16), g1 = *(const f32x4*)(gk + st * 16 + 4);
        const f32x4 b0 = *(const f32x4*)(bk + st * 16), b1 = *(const f32x4*)(bk + st * 16 + 4);
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            y[e] = (bf2f((bf16_t)x8[e]) - mean) * rstd * g0[e] + b0[e];
            y[4 + e] = (bf2f((bf16_t)x8[4 + e]) - mean) * rstd * g1[e] + b1[e];
        }
        union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
        for (int i = 0; i < 4; i++) pk.u[i] = cvt_pk_bf16(y[2 * i], y[2 * i + 1]);
        return pk.v;
    };
    auto batch = [&](const GB& t, int sb) {
        bf16x8 xb[SB];
#pragma unroll
        for (int u = 0; u < SB; u++) xb[u] = normed(sb + u);
#pragma unroll
        for (int u = 0; u < SB; u++)
#pragma unroll
            for (int jb = 0; jb < NB; jb++) acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.g[u][jb], xb[u], acc[jb], 0, 0, 0);
    };
    int s0 = 0;
    for (; s0 + SB <= nsteps; s0 += 2 * SB) {
        const bool two = s0 + 2 * SB <= nsteps;
        if (two) gload(gB, s0 + SB);
        batch(gA, s0);
        if (two) {
            if (s0 + 3 * SB <= nsteps) gload(gA, s0 + 2 * SB);
            batch(gB, s0 + SB);
        }
    }
    s0 = (nsteps / SB) * SB;
    for (; s0 < nsteps; s0++) {
        const bf16x8 xb = normed(s0);
#pragma unroll
        for (int jb = 0; jb < NB; jb++) {
            const bf16x8 gb = *(const bf16x8*)(gr + ((long long)s0 * NJ + jb) * 512);
            acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb, xb, acc[jb], 0, 0, 0);
        }
    }
    const int ncb = C >> 5;
    const bf16_t* ub = p.U + ((long long)b * ncb * (2 * NJ) * 64 + lane) * 8;
    struct Blk { bf16x8 uf[2 * NB]; f32x4 bv[4]; };
    auto request = [&](Blk& k, int cb) {
#pragma unroll
        for (int st = 0; st < 2 * NB; st++) k.uf[st] = *(const bf16x8*)(ub + ((long long)cb * (2 * NJ) + st) * 512);
#pragma unroll
        for (int q = 0; q < 4; q++) k.bv[q] = p.bias ? *(const f32x4*)(p.bias + cb * 32 + 16 * hf + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    Blk k0, k1;
    if (w < ncb) request(k0, w);
    // ---- exchange + softmax, one 32-column block at a time
    bf16x8 pf[2 * NB];
#pragma unroll
    for (int jb = 0; jb < NB; jb++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 v = {acc[jb][4 * q], acc[jb][4 * q + 1], acc[jb][4 * q + 2], acc[jb][4 * q + 3]};
            *(f32x4*)(part + ((w * 4 + q) * 64 + lane) * 4) = v;
        }
        __syncthreads();
        float s[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 t = *(const f32x4*)(part + ((0 * 4 + q) * 64 + lane) * 4);
#pragma unroll
            for (int ww = 1; ww < 4; ww++) t += *(const f32x4*)(part + ((ww * 4 + q) * 64 + lane) * 4);
            s[4 * q] = t[0]; s[4 * q + 1] = t[1]; s[4 * q + 2] = t[2]; s[4 * q + 3] = t[3];
        }
        __syncthreads();
        xattn_group_softmax(s, p.group);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
            for (int i = 0; i < 4; i++) pk.u[i] = cvt_pk_bf16(s[8 * half + 2 * i], s[8 * half + 2 * i + 1]);
            pf[2 * jb + half] = pk.v;
        }
    }
    // ---- phase 2: residual from the tile, result into the tile
    auto finish = [&](const Blk& k, int cb) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 2 * NB; st++) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k.uf[st], pf[st], o, 0, 0, 0);
        char* tp = T + l31 * RS + (cb * 32 + 16 * hf) * 2;
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            const uint4 rq = *(const uint4*)(tp + 16 * h2);
            const uint32_t rr[4] = {rq.x, rq.y, rq.z, rq.w};
            uint32_t wv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 8 * h2 + 2 * i;
                const float a0 = o[r] + k.bv[r >> 2][r & 3] + __uint_as_float(rr[i] << 16);
                const float a1 = o[r + 1] + k.bv[(r + 1) >> 2][(r + 1) & 3] + __uint_as_float(rr[i] & 0xffff0000u);
                wv[i] = cvt_pk_bf16(a0, a1);
            }
            *(uint4*)(tp + 16 * h2) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
    };
    for (int cb = w; cb < ncb; cb += 8) {
        if (cb + 4 < ncb) request(k1, cb + 4);
        finish(k0, cb);
        if (cb + 4 < ncb) {
            if (cb + 8 < ncb) request(k0, cb + 8);
            finish(k1, cb + 4);
        }
    }
    __syncthreads();
    // ---- the tile out
    for (int it0 = 0; it0 < npc; it0 += 3) {
        const Piece q0 = piece(it0), q1 = piece(it0 + 1), q2 = piece(it0 + 2);
        const uint4 v0 = *(const uint4*)(T + q0.l), v1 = *(const uint4*)(T + q1.l), v2 = *(const uint4*)(T + q2.l);
        *(uint4*)(p.out + q0.g) = v0; *(uint4*)(p.out + q1.g) = v1; *(uint4*)(p.out + q2.g) = v2;
    }
    if (!p.ln3_out) return;
    // ---- norm3 of the finished rows (the bf16 values just stored, as layernorm_bf16x8_kernel would read them back: mean, then the
    // centred squares): statistics as above (lane = row, this wave's K quarter), then every thread normalises its pieces of the tile
    {
        float s3 = 0.f;
        for (int st = 0; st < nsteps; st++) {
            const bf16x8 x8 = *(const bf16x8*)(xl + st * 32);
#pragma unroll
            for (int e = 0; e < 8; e++) s3 += bf2f((bf16_t)x8[e]);
        }
        s3 += __shfl_xor(s3, 32);
        if (hf == 0) stat[w * 32 + l31] = s3;
        __syncthreads();
        const float mean3 = (stat[l31] + stat[32 + l31] + stat[64 + l31] + stat[96 + l31]) / (float)C;
        __syncthreads();
        float q3 = 0.f;
        for (int st = 0; st < nsteps; st++) {
            const bf16x8 x8 = *(const bf16x8*)(xl + st * 32);
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = bf2f((bf16_t)x8[e]) - mean3; q3 += d * d; }
        }
        q3 += __shfl_xor(q3, 32);
        if (hf == 0) stat[w * 32 + l31] = q3;
        __syncthreads();
        const float rstd3 = rsqrtf((stat[l31] + stat[32 + l31] + stat[64 + l31] + stat[96 + l31]) / (float)C + p.ln_eps);
        __syncthreads();
        if (w == 0 && hf == 0) { stat[l31] = mean3; stat[32 + l31] = rstd3; }
        __syncthreads();
        for (int it = 0; it < npc; it++) {
            const int id = it * 256 + tid, row = id / ppr, pc = id - row * ppr;
            const uint4 v = *(const uint4*)(T + row * RS + pc * 16);
            const float m = stat[row], r = stat[32 + row];
            const f32x4 g0 = *(const f32x4*)(p.ln3_g + pc * 8), g1 = *(const f32x4*)(p.ln3_g + pc * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(p.ln3_b + pc * 8), b1 = *(const f32x4*)(p.ln3_b + pc * 8 + 4);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float x0 = __uint_as_float(u[i] << 16), x1 = __uint_as_float(u[i] & 0xffff0000u);
                const float ga = i < 2 ? g0[2 * i] : g1[2 * i - 4], gb = i < 2 ? g0[2 * i + 1] : g1[2 * i - 3];
                const float ba = i < 2 ? b0[2 * i] : b1[2 * i - 4], bb = i < 2 ? b0[2 * i + 1] : b1[2 * i - 3];
                o[i] = cvt_pk_bf16((x0 - m) * r * ga + ba, (x1 - m) * r * gb + bb);
            }
            *(uint4*)(p.ln3_out + (row0 + row) * C + pc * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

bool xattn_fused_supported(const XattnParams& p) {
    return p.n > 0 && p.n % 32 == 0 && p.rows % p.n == 0 && p.C % 64 == 0 && p.NP % 32 == 0 && p.ncols >= 1 && p.ncols <= p.NP && p.ncols <= 128 &&
           (p.group == 1 || p.group == 2 || p.group == 4);
}
// p.G / p.U: the fragment-ordered images written by launch_xattn_pack
static size_t xattn_ln_smem(int C) { return (size_t)32 * (2 * C + 16) + 4 * 4 * 64 * 4 * 4 + 4 * 32 * 4; }
hipError_t launch_xattn_fused(const XattnParams& p, hipStream_t st) {
    if (!xattn_fused_supported(p) || !p.x || !p.G || !p.U || !p.out) return hipErrorInvalidValue;
    const int nb = (p.ncols + 31) / 32;
    dim3 grid((unsigned)(p.rows / 32));
    if (p.ln_g) {
        if (!p.ln_b || xattn_ln_smem(p.C) > 160 * 1024) return hipErrorInvalidValue;
        static bool attr[RDM_MAX_DEVICES] = {};
        const int dev = rdm_cur_device();
        if (!attr[dev]) {
            hipError_t e = hipSuccess;
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xattn_ln_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xattn_ln_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xattn_ln_fused_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xattn_ln_fused_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr[dev] = true;
        }
        const size_t smem = xattn_ln_smem(p.C);
        switch (nb) {
            case 1: xattn_ln_fused_kernel<1><<<grid, 256, smem, st>>>(p); break;
            case 2: xattn_ln_fused_kernel<2><<<grid, 256, smem, st>>>(p); break;
            case 3: xattn_ln_fused_kernel<3><<<grid, 256, smem, st>>>(p); break;
            default: xattn_ln_fused_kernel<4><<<grid, 256, smem, st>>>(p); break;
        }
        return hipGetLastError();
    }
    switch (nb) {
        case 1: xattn_fused_kernel<1><<<grid, 256, 0, st>>>(p); break;
        case 2: xattn_fused_kernel<2><<<grid, 256, 0, st>>>(p); break;
        case 3: xattn_fused_kernel<3><<<grid, 256, 0, st>>>(p); break;
        default: xattn_fused_kernel<4><<<grid, 256, 0, st>>>(p); break;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ small attention

template <int D>
__global__ __launch_bounds__(256) void small_attention_kernel(SmallAttnParams p) {
    extern __shared__ float kv[];            // K [nkv][D] then V [nkv][D]  (f32)
    const int h = blockIdx.y, b = blockIdx.z;
    float* Ks = kv; float* Vs = kv + p.nkv * D;
    for (int i = threadIdx.x; i < p.nkv * (D / 8); i += blockDim.x) {
        const int j = i / (D / 8), c = (i % (D / 8)) * 8;
        const bf16x8 kk = *(const bf16x8*)(p.k + ((long long)b * p.nkv + j) * p.ldk + h * D + c);
        const bf16x8 vv = *(const bf16x8*)(p.v + ((long long)b * p.nkv + j) * p.ldv + h * D + c);
#pragma unroll
        for (int e = 0; e < 8; e++) { Ks[j * D + c + e] = bf2f((bf16_t)kk[e]); Vs[j * D + c + e] = bf2f((bf16_t)vv[e]); }
    }
    __syncthreads();
    for (int qi = blockIdx.x * blockDim.x + threadIdx.x; qi < p.nq; qi += gridDim.x * blockDim.x) {
        float qr[D], acc[D];
        const bf16_t* qp = p.q + ((long long)b * p.nq + qi) * p.ldq + h * D;
#pragma unroll
        for (int c = 0; c < D; c += 8) {
            const bf16x8 t = *(const bf16x8*)(qp + c);
#pragma unroll
            for (int e = 0; e < 8; e++) { qr[c + e] = bf2f((bf16_t)t[e]) * p.scale; acc[c + e] = 0.f; }
        }
        float m = -INFINITY, l = 0.f;
        const int jend = p.causal ? min(p.nkv, qi + 1) : p.nkv;
        for (int j = 0; j < jend; j++) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; d++) s += qr[d] * Ks[j * D + d];
            const float mn = fmaxf(m, s);
            const float a = __expf(m - mn), pj = __expf(s - mn);
            l = l * a + pj; m = mn;
#pragma unroll
            for (int d = 0; d < D; d++) acc[d] = acc[d] * a + pj * Vs[j * D + d];
        }
        const float inv = 1.f / l;
        bf16_t* op = p.out + ((long long)b * p.nq + qi) * p.ldo + h * D;
#pragma unroll
        for (int c = 0; c < D; c += 8) {
            uint4 w;
            w.x = pack2bf(acc[c] * inv, acc[c + 1] * inv); w.y = pack2bf(acc[c + 2] * inv, acc[c + 3] * inv);
            w.z = pack2bf(acc[c + 4] * inv, acc[c + 5] * inv); w.w = pack2bf(acc[c + 6] * inv, acc[c + 7] * inv);
            *(uint4*)(op + c) = w;
        }
    }
}

hipError_t launch_small_attention(const SmallAttnParams& p, int D, int heads, int batch, hipStream_t st) {
    const size_t sm = (size_t)p.nkv * D * 2 * sizeof(float);
    if (sm > 64 * 1024) return hipErrorInvalidValue;
    int threads = p.nq >= 256 ? 256 : ((p.nq + 63) / 64) * 64;
    int gx = (p.nq + threads - 1) / threads;
    dim3 grid(gx, heads, batch);
    if (D == 32) small_attention_kernel<32><<<grid, threads, sm, st>>>(p);
    else if (D == 64) small_attention_kernel<64><<<grid, threads, sm, st>>>(p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// Input-stationary 3x3 convolution (stride 1, pad 1, optional fused nearest-2x upsample) for NHWC bf16 on gfx950,
// ONE WAVE PER SIMD: the round-3 successor of conv_halo.hip's 8-wave ping-pong kernel for the UNet's dominant op
// (conv_nd(2, C, C', 3, padding=1) inside ResBlock in_layers / out_layers and Upsample, ldm, reached from
// rdm/modules/diffusionmodules/openaimodel.py:144-305).
//
// What bounded the 8-wave kernel (profiles/r02d_pmc_sq.csv: 48 % MFMA busy): every 384-cycle MFMA section had a ~600-cycle load
// section beside it -- 10 ds_read_b128 + up to 3 LDS-DMA issues per wave, the DMA issue parks the wave 60-185 cycles a piece, and
// the 96-accumulator wave tile needs 0.83 KB of LDS reads per MFMA.  Here:
//   * 4 waves, one per SIMD, each owning a 128 x 96 (FN = 3) output tile of the block's 256 pixels x 192 channels: 192 fp32
//     accumulators in the AGPR half of the 512-entry register file, 4 A + 3 B fragments per 12 MFMAs;
//   * the WEIGHT operand never touches LDS: the library keeps a fragment-ordered copy of every 3x3 weight
//     (conv_w_fragpack_kernel: [N/32][tap][Cin/16][lane][8], one contiguous KiB per 32x16 MFMA B fragment), so a fragment is ONE
//     fully coalesced global_load_dwordx4 from L2 straight into the registers the MFMA reads; it is requested one whole tap
//     (48 MFMAs, ~1.5 k cycles) ahead into the registers its predecessor has just been consumed from;
//   * LDS holds only the halo (2 buffers, double-buffered per 64-channel slice), staged THROUGH REGISTERS (global_load ->
//     ds_write_b128 three k-steps later) instead of LDS-DMA: no parked issue, and the LDS image need not be lane-linear: it is
//     an ADDITIVE layout (halo4_geom below: 144-byte positions, padded rows) in which a tap is a constant offset and every
//     A-fragment ds_read_b128 is conflict-free at all four resolutions -- no per-tap swizzle arithmetic at all;
//   * the instruction stream is hand-placed: MFMAs, fragment loads and waits are asm volatile statements in program order
//     (hipcc keeps their order and only allocates registers), 4 ds_read_b128 + <= 4 global loads + <= 1 ds_write per 12 MFMAs,
//     every wait counted (vmcnt retires in order); one s_barrier per 64-channel slice (9 taps, 432 MFMAs per wave).
// Same persistent XCD-aware tile walk, split-K planes and epilogue scheme (DPP pair swap -> wave-private LDS transpose -> 16-byte
// row stores) as conv_halo.hip.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"

__device__ unsigned long long g_halo4_prof[4];

// fragment-ordered weight copy: dst[((nb * 9 + tap) * KQ + kq) * 512 + lane * 8 + e] = W[nb * 32 + (lane & 31)][tap][kq * 16 + (lane >> 5) * 8 + e]
__global__ __launch_bounds__(256) void conv_w_fragpack_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ dst, int N, int Cin) {
    const int KQ = Cin >> 4;
    const long long nvec = (long long)N * 9 * Cin / 8;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const int lane = (int)(v & 63);
        long long f = v >> 6;
        const int kq = (int)(f % KQ); f /= KQ;
        const int tap = (int)(f % 9); const int nb = (int)(f / 9);
        const int n = nb * 32 + (lane & 31), c = kq * 16 + (lane >> 5) * 8;
        *(uint4*)(dst + v * 8) = *(const uint4*)(W + ((long long)n * 9 + tap) * Cin + c);
    }
}
hipError_t launch_conv_w_fragpack(const bf16_t* W, bf16_t* dst, int N, int Cin, hipStream_t st) {
    const long long nvec = (long long)N * 9 * Cin / 8;
    long long g = (nvec + 255) / 256; if (g > 8192) g = 8192;
    conv_w_fragpack_kernel<<<dim3((unsigned)g), 256, 0, st>>>(W, dst, N, Cin);
    return hipGetLastError();
}

#include "h4_asm.h"

// host + device: halo geometry of a conv (output H x W) in the one-wave-per-SIMD kernel.  A tile is 256 consecutive output pixels =
// NS sample parts of RS whole image rows; its halo is NROW = NS (RS + 2) rows of HPW = W + 2 positions.  LDS image (per 64-channel
// slice): row R at R * RSTR, position hx at + 144 hx (128 bytes of channels + 16 of padding), 16-byte channel chunk c at + 16 c.
// The layout is ADDITIVE -- a tap (dy, dx) is the constant offset dy RSTR + 144 dx, a k-step 32 bytes -- so the A-fragment
// addresses of a tap cost one add each, and the 144-byte position stride (9 x 16: odd) together with the row padding
// (RSTR = 144 HPW + 224 at W <= 16, making consecutive rows differ by 0 resp. 8 sixteen-byte slots mod 16) keeps every
// ds_read_b128 of a 32-pixel fragment conflict-free at all four resolutions (brute-forced over the instruction's lane groups;
// the XOR-swizzled 128-byte layout of the 8-wave kernel was 3-way conflicted at 16x16 and 7-way at 8x8).
struct Halo4Geom { int RS, NS, HPW, NROW, RSTR, NPR, NPT, HBYTES; };
__host__ __device__ inline Halo4Geom halo4_geom(int H, int W) {
    Halo4Geom g;
    const int HW = H * W;
    g.RS = (HW >= 256) ? 256 / W : H;
    g.NS = 256 / (g.RS * W);
    g.HPW = W + 2;
    g.NROW = g.NS * (g.RS + 2);
    g.RSTR = g.HPW * 144 + (W <= 16 ? 224 : 0);
    g.NPR = (g.HPW + 7) >> 3;                     // 8-position pieces per halo row
    g.NPT = g.NROW * g.NPR;
    g.HBYTES = (g.NROW * g.RSTR + 255) & ~255;
    return g;
}
constexpr int H4_HALO_MAX = 66560;                 // largest HBYTES admitted (W = 8: 40 rows x 1664 bytes)

// STRIP (round 4): images wider than 64 pixels (the first-stage decoder's 128- and 256-pixel levels).  A tile is still 256 output pixels
// with the W = 64 halo geometry -- 4 rows of a 64-COLUMN STRIP -- but the strip's left / right halo columns are the neighbouring
// strips' real pixels (zero only at the image border), rows of a tile are WI pixels apart in memory, and tile index -> (sample, row
// group, strip).  Costs three VALU per halo piece (column = strip origin + halo column, its validity, the source column), so it is a
// template variant: the UNet's kernels (W <= 64) are unchanged.
template <int FN, int VAR, bool STRIP = false>      // VAR: dev-only ablations (RDM_H4_VAR): 1 = no halo-piece address work (wrong results)
__global__ __launch_bounds__(256, 1) void conv3x3_halo4_kernel(IgemmParams p) {
    constexpr int BM = 256, BK = 64, FM = 4, WN = FN * 32, BN = 2 * WN;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo0][halo1][dump: 1 KB per wave]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int lp = lane >> 3, lc = lane & 7;

    // ---- geometry (uniform)
    const int H = p.Hout, WI = p.Wout, HW = H * WI;        // WI: image width; W: width of the halo geometry (a 64-column strip when STRIP)
    const int W = STRIP ? 64 : WI;
    const Halo4Geom gm = halo4_geom(H, W);
    const int RS = gm.RS, HPW = gm.HPW, RSTR = gm.RSTR, NPR = gm.NPR, NPT = gm.NPT, HBYTES = gm.HBYTES;
    const int mNPR = 65536 / NPR + 1, mRS2 = 65536 / (RS + 2) + 1;     // x / d = (x * m) >> 16 for the small x met here (checked on the host)
    const int Cin = p.C0 + p.C1, nslice = Cin / BK, KQ = Cin >> 4;
    const int npw = (NPT - wave + 3) >> 2;                  // halo pieces this wave stages: wave, wave + 4, ...

    const int nbn = p.N / BN, nbm = p.M / BM;
    const int ntiles_mn = nbm * nbn;
    const int S = p.ksplit > 1 ? p.ksplit : 1;
    const int ntiles = ntiles_mn * S;
    const int G = gridDim.x, xcd = blockIdx.x & 7;
    const int gx = (G - xcd + 7) >> 3;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_end = t_begin + tq + (xcd < tr ? 1 : 0);
    int tile = t_begin + (blockIdx.x >> 3);
    if (tile >= t_end) return;

    const char* const zero = (const char*)p.zero_page;
    const char* const Wf = (const char*)p.Wfrag;

    // ---- per-lane constants: LDS offset of fragment row `frow` of A fragment i at tap (0, 0), k-step 0, buffer 0
    unsigned vbase[FM];
#pragma unroll
    for (int i = 0; i < FM; i++) {
        const int pl = wm * 128 + i * 32 + frow;
        const int s = pl / (RS * W), r = pl - s * RS * W;
        const int fy = r / W, fx = r - fy * W;
        vbase[i] = (unsigned)((s * (RS + 2) + fy) * RSTR + fx * 144 + fhalf * 16);
    }
    unsigned voffj[FN];
#pragma unroll
    for (int j = 0; j < FN; j++) voffj[j] = (unsigned)(lane * 16) + (unsigned)j * (unsigned)(9 * KQ * 1024);
    const unsigned lane_pc = (unsigned)(lp * 144 + lc * 16);             // this lane's 16 bytes inside an 8-position piece
    const unsigned dump = (unsigned)(2 * HBYTES) + (unsigned)(wave * 1024 + lane * 16);

    // fragment-ordered weights of this wave's FN column fragments for work item t: SGPR base of (tap 0, slice 0, j = 0, ks = 0);
    // a tap further on is tap_stride bytes away, a slice 4 KiB
    const long long tap_stride = (long long)KQ * 1024;
    auto w_base = [&](int t) -> const char* {
        const int bn = (t % ntiles_mn) % nbn;
        const int nb0 = bn * (BN / 32) + wn * FN;
        const long long off = (long long)nb0 * 9 * tap_stride;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(off & 0xffffffffLL));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(off >> 32));
        return Wf + (((unsigned long long)hi << 32) | lo);
    };
    auto slice_begin = [&](int t) { return ((t / ntiles_mn) * nslice) / S; };
    auto slice_end = [&](int t) { return ((t / ntiles_mn + 1) * nslice) / S; };

    // ---- halo staging.  Piece g (8 consecutive positions of one halo row, 64 channels: one KiB, 16 bytes per lane) belongs to wave
    // g % 4.  Everything about a piece but the lane's column is wave-uniform (scalar unit): halo row R, first column, source row,
    // LDS row address.  Rows outside the image and pieces beyond the halo read the zero page; columns outside the image are
    // redirected there per lane; positions past a row's end, and whole pieces beyond the halo, land in a wave-private dump area --
    // every k-step issues the same requests whether its piece is real or not, so the wait counts are compile-time constants.
    // y0 | x0 << 16 in ONE register when STRIP (x0: first column of the tile's strip): the scalar file is full (hipcc already parks ~90
    // uniform values in VGPR lanes; three more live scalars pushed those VGPRs into scratch, whose traffic would count in vmcnt)
    struct HaloTile { int b0, y0; };
    auto ht_y0 = [](const HaloTile& h) { return STRIP ? (h.y0 & 0xffff) : h.y0; };
    auto ht_x0 = [](const HaloTile& h) { return STRIP ? (int)((unsigned)h.y0 >> 16) : 0; };
    struct HaloSrc { const char* src; unsigned ldb; };                   // src: channel slice of pixel 0 in the slice's source tensor
    const int ups = p.ups ? 1 : 0;
    auto halo_tile = [&](int t) {
        const int tm0 = ((t % ntiles_mn) / nbn) * BM;
        HaloTile h;
        if constexpr (STRIP) {              // tiles of an image: row groups of RS = 4 rows x strips of 64 columns, strips fastest
            const int spr = WI >> 6, tpi = (H >> 2) * spr, tmi = tm0 >> 8;
            h.b0 = tmi / tpi;
            const int r = tmi - h.b0 * tpi, rg = r / spr;
            h.y0 = (rg << 2) | (((r - rg * spr) << 6) << 16);
        } else { h.b0 = tm0 / HW; h.y0 = (tm0 - h.b0 * HW) / W; }
        return h;
    };
    auto halo_src = [&](int sl) {
        const int kc = sl * BK;
        const bool second = kc >= p.C0;
        HaloSrc h;
        h.src = (const char*)(second ? p.A1 : p.A0) + (second ? kc - p.C0 : kc) * 2;
        h.ldb = (unsigned)(second ? p.C1 : p.C0) * 2u;
        return h;
    };
    // What a piece needs that depends neither on the tile nor on the slice -- its halo row (sample part s, row hy), this lane's image
    // column, validity of that column, its LDS destination -- is tabulated ONCE per kernel, per wave, in LDS (21 pieces x 64 lanes
    // x 4 bytes): bits 0-12 LDS offset / 16 inside a halo buffer, 13-18 source column (x >> ups), 19 column inside the image,
    // 20-25 hy, 26-27 s, 28 position exists (else the 16 bytes go to the dump area).  Per piece and slice that leaves ~18 VALU and
    // no scalar work: measured, the scalar row arithmetic of a piece (~40 SALU + 2 branches in one clump) cost ~340 cycles of a
    // starved matrix pipe -- every instruction beside the MFMAs is paid for unless it sits in a 32-cycle MFMA shadow.
    const unsigned tbl = (unsigned)(2 * HBYTES + 4096) + (unsigned)(wave * (21 * 256) + lane * 4);
    for (int q = 0; q < 21; q++) {
        const int g = q * 4 + wave;
        const int R = (g * mNPR) >> 16, xc = g - R * NPR;
        const int sp = (R * mRS2) >> 16, hy = R - sp * (RS + 2);
        const int hx = xc * 8 + lp, x = hx - 1;
        const unsigned rel = (unsigned)(R * RSTR + xc * (8 * 144)) + lane_pc;
        const unsigned xok = ((unsigned)x < (unsigned)W) ? 1u : 0u;
        const unsigned wr = (g < NPT && hx < HPW) ? 1u : 0u;
        const unsigned xs = xok ? (unsigned)(x >> ups) : 0u;
        // STRIP: bits 13-19 hold the halo column hx (0 .. 65) itself: image column, validity and source column follow from the tile's strip
        const unsigned e = STRIP ? (((rel >> 4) & 0x1fffu) | ((unsigned)(hx & 127) << 13) | ((unsigned)(hy & 63) << 20) | ((unsigned)(sp & 3) << 26) | (wr << 28))
                                 : (((rel >> 4) & 0x1fffu) | (xs << 13) | (xok << 19) | ((unsigned)(hy & 63) << 20) | ((unsigned)(sp & 3) << 26) | (wr << 28));
        *(unsigned*)(smem + tbl + q * 256) = e;
    }
    const char* const zl = zero + lane * 16;                             // this lane's 16 bytes of the zero page
    // table entry e of a piece -> source address of this lane's 16 bytes (tile ht, slice source hs) and LDS destination.
    // `live` (uniform): the piece is one of this wave's (else: zero page -> dump area)
    auto halo_piece = [&](unsigned e, bool live, const HaloTile& ht, const HaloSrc& hs, unsigned hbase, const char*& gaddr, unsigned& ldst) {
        const int hy = (int)((e >> 20) & 63u), sp = (int)((e >> 26) & 3u);
        int xs = (int)((e >> 13) & 63u);
        bool xin = ((e >> 19) & 1u) != 0;
        if constexpr (STRIP) { const int x = ht_x0(ht) + (int)((e >> 13) & 127u) - 1; xin = (unsigned)x < (unsigned)WI; xs = x >> ups; }
        const int y = ht_y0(ht) + hy - 1;
        const bool ok = live & xin & ((unsigned)y < (unsigned)H);
        const unsigned pix = (unsigned)__mul24(__mul24(ht.b0 + sp, p.Hin) + (y >> ups), p.Win) + (unsigned)xs;
        const char* const ga = hs.src + (unsigned long long)pix * hs.ldb + (unsigned)(lc * 16);
        const unsigned long long sel = ok ? (unsigned long long)ga : (unsigned long long)zl;
        gaddr = (const char*)sel;
        const unsigned d = hbase + ((e & 0x1fffu) << 4);
        ldst = (live & (((e >> 28) & 1u) != 0)) ? d : dump;
    };

    bf16_t* const ob = p.out_bf16;
    const bf16_t* const rb = p.res_bf16;

    // ---- tile / slice state (uniform unless noted)
    asm volatile("" ::: H4_ACC_CLOBBERS);                    // the kernel descriptor must allocate the accumulator AGPRs
    bf16x8 fa[2][FM];                                       // A fragments: k-step parity
    bf16x8 fb[2][4][FN];                                    // B fragments: [tap-step parity][k-step][column fragment]
    bf16x8 hreg[2][3];                                      // halo pieces in flight: [tap-step parity][k-step - 1]
    unsigned hdst[2][3] = {{dump, dump, dump}, {dump, dump, dump}};
    unsigned va[FM];                                        // A fragment rows of the current tap (k-step 0)
    int hb = 0;                                             // halo buffer of the slice being computed
    int em0 = 0, en0 = 0, nem0 = 0, nen0 = 0, part = 0, next = 0, s_begin = 0, s_end = 0, ns_begin = 0;
    bool has_next = false;
    const char* wb_tile = Wf; const char* wb_next = Wf;
    HaloTile ht_tile{0, 0}, ht_next{0, 0};
    auto tile_setup = [&]() {
        const int tmn = tile % ntiles_mn;
        em0 = (tmn / nbn) * BM; en0 = (tmn % nbn) * BN;
        part = tile / ntiles_mn;
        next = tile + gx;
        has_next = next < t_end;
        s_begin = slice_begin(tile); s_end = slice_end(tile);
        ns_begin = has_next ? slice_begin(next) : 0;
        wb_tile = w_base(tile);
        wb_next = has_next ? w_base(next) + (long long)ns_begin * 4096 : wb_tile;       // first step of the next work item
        ht_next = has_next ? halo_tile(next) : HaloTile{0, 0};
        { const int nmn = (has_next ? next : tile) % ntiles_mn; nem0 = (nmn / nbn) * BM; nen0 = (nmn % nbn) * BN; }
        ht_tile = halo_tile(tile);
    };
    int sl = 0, tap = 0;
    bool last_slice = false;
    int hnp = 0;
    HaloTile ht{0, 0}; HaloSrc hs{nullptr, 0};
    unsigned hbase_cur = 0, hbase_nxt = 0;
    const char* sb_run = Wf; const char* sb_after = Wf;
    auto slice_setup = [&]() {
        last_slice = sl + 1 == s_end;
        const bool hload = !last_slice || has_next;          // is there a slice whose halo is staged while this one is computed?
        ht = last_slice ? ht_next : ht_tile;
        hs = halo_src(last_slice ? ns_begin : sl + 1);
        hnp = hload ? npw : 0;
        hbase_cur = (unsigned)(hb * HBYTES); hbase_nxt = (unsigned)((hb ^ 1) * HBYTES);
        sb_run = wb_tile + (long long)sl * 4096;             // weights of this slice's tap 0 ...
        sb_after = last_slice ? wb_next : sb_run + 4096;     // ... and of the step that follows its last tap
    };

    // ---- block prologue: whole halo of the first slice, weights of the first tap, A fragments of k-step 0
    tile_setup();
    sl = s_begin; tap = 0;
    {
        const HaloSrc hs0 = halo_src(sl);
        for (int q = 0; q < npw; q++) {
            const char* g; unsigned d;
            halo_piece(*(const unsigned*)(smem + tbl + q * 256), true, ht_tile, hs0, 0u, g, d);
            H4_GLOADH(hreg[0][0], g);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            H4_LDSW(d, hreg[0][0]);
        }
        const char* sb = wb_tile + (long long)sl * 4096;
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int j = 0; j < FN; j++) { H4_GLOADB(fb[0][ks][j], voffj[j], sb, ks * 1024); fb[1][ks][j] = fb[0][ks][j]; }
#pragma unroll
        for (int i = 0; i < FM; i++) va[i] = vbase[i];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < FM; i++) { H4_LDSR(fa[0][i], va[i], 0); fa[1][i] = fa[0][i]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { hreg[0][k] = hreg[0][0]; hreg[1][k] = hreg[0][0]; }
    }
    slice_setup();
    // The accumulators of a work item start at bias (+ the sample's time-embedding row), written by the MATRIX PIPE: one MFMA with
    // C = 0 per fragment, A = the fragment's 32 output channels as (hi, lo) bf16 pairs in k = 0, 1 (hi + lo = the fp32 value to
    // 2^-17), B = ones in k = 0, 1 -- 4 FN issue slots where v_accvgpr_write needed 64 FN.  K-split parts start at zero (their
    // finisher adds the bias).  The epilogue of a tile writes the next tile's start values as it empties the accumulators.
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    union FragU { u32x4_t u; bf16x8 f; };
    auto start_values = [&](int tm0, int tn0, int row, float (&pb)[FM][FN]) {      // fp32 start value of channel `row` of fragment (i, j), tile at (tm0, tn0)
#pragma unroll
        for (int i = 0; i < FM; i++) {
            const int mf = tm0 + wm * 128 + i * 32;
            const float* rv = p.rowvec ? p.rowvec + (long long)(mf / p.rows_per_sample) * p.rowvec_ld : nullptr;
#pragma unroll
            for (int j = 0; j < FN; j++) {
                const int ncol = tn0 + wn * WN + j * 32 + row;
                float bv = (S == 1 && p.bias) ? p.bias[ncol] : 0.f;
                if (S == 1 && rv) bv += rv[ncol];
                pb[i][j] = bv;
            }
        }
    };
    auto start_frag = [&](float bv, int half) -> bf16x8 {
        const uint32_t hi = cvt_pk_bf16(bv, 0.f) & 0xffffu;
        const uint32_t lo = cvt_pk_bf16(bv - __uint_as_float(hi << 16), 0.f) & 0xffffu;
        FragU t; t.u = (u32x4_t){half ? 0u : (hi | (lo << 16)), 0u, 0u, 0u};
        return t.f;
    };
    {
        float pb[FM][FN];
        start_values(em0, en0, frow, pb);
        FragU o; o.u = (u32x4_t){fhalf ? 0u : 0x3f803f80u, 0u, 0u, 0u};
        bf16x8 ones = o.f;
#pragma unroll
        for (int i = 0; i < FM; i++) {
            bf16x8 sf[FN];
#pragma unroll
            for (int j = 0; j < FN; j++) sf[j] = start_frag(pb[i][j], fhalf);
            // (VALU-written MFMA operands: hipcc pads that hazard for its own MFMAs, not around asm)
            if constexpr (FN == 3) asm volatile("s_nop 7" : "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(ones));
            else asm volatile("s_nop 7" : "+v"(sf[0]), "+v"(sf[1]), "+v"(ones));
#pragma unroll
            for (int j = 0; j < FN; j++) H4_MFMA0(i * FN + j, sf[j], ones);
        }
    }

    // ---- main stream: one iteration = one tap-step (a tap of a 64-channel slice: 4 k-steps of 16 channels, 16 FN MFMAs).
    // Weight fragments travel in two register sets by step parity: a step consumes set P and, during its first k-step, requests the
    // WHOLE next step's fragments into set P ^ 1 (4 FN loads, one behind each MFMA).  The three halo pieces a step requests (k-steps
    // 1..3) are therefore younger than the weight batch of their own step and older than the next one's: the single counted wait of
    // a step (vmcnt(3) at its first k-step: "my weights have landed") retires the pieces of the step BEFORE the previous one, which
    // are written to LDS during this step -- a piece has two whole steps (~3 k cycles) to arrive from HBM without ever stalling the
    // weight stream (vmcnt retires in order: with per-k-step weight requests a piece that was late stalled every wait behind it).
    unsigned long long tprof[2] = {0, 0};
    unsigned long long tp0 = 0, tp1 = 0;
    if (p.dbg & 16) tp0 = __builtin_readcyclecounter();
    // descriptors of the NEXT tap-step (uniform), refreshed before every step
    bool slice_last_tap = false;
    int epi_stores = 0;                                     // stores of the epilogue just finished that may still be in flight
    const char* sbn = Wf; unsigned noff = 0; int q0 = 0;
    auto describe = [&]() {
        slice_last_tap = tap == 8;
        const int n_tap = slice_last_tap ? 0 : tap + 1;
        sbn = slice_last_tap ? sb_after : sb_run + tap_stride;
        sb_run = sbn;
        const int ndy = (n_tap * 11) >> 5, ndx = n_tap - ndy * 3;
        noff = (slice_last_tap ? hbase_nxt : hbase_cur) + (unsigned)(ndy * RSTR + ndx * 144);
        q0 = tap * 3;
    };
    auto step = [&](auto ptag) {
        constexpr int P = decltype(ptag)::value;
        auto mfma = [&](int ks, int m) {
            const int j = m / FM, i = m % FM;
            H4_MFMA(i * FN + j, fb[P][ks][j], fa[ks & 1][i]);          // D^T: rows = output channels, columns = pixels
        };
        // ---- k-step 0: this step's weights (requested during the previous step's first k-step; its 3 halo requests are younger)
        // (right after an epilogue its stores are younger still: FM * NIT of them, or 4 FM FN split-K plane stores)
        if (epi_stores == 0) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if (epi_stores == FM * ((32 * (WN / 8)) / 64)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(3 + FM * ((32 * (WN / 8)) / 64)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(3 + 4 * FM * FN) : "memory");
        epi_stores = 0;
        H4_LDSW(hdst[P][0], hreg[P][0]);
#pragma unroll
        for (int m = 0; m < FM * FN; m++) {
            mfma(0, m);
            if constexpr (VAR == 3) H4_GLOADB(fb[P ^ 1][m / FN][m % FN], voffj[0], Wf, 0);      // ablation: every weight request hits the same KiB (wrong results)
            else H4_GLOADB(fb[P ^ 1][m / FN][m % FN], voffj[m % FN], sbn, (m / FN) * 1024);
            if (m == 3) {
#pragma unroll
                for (int i = 0; i < FM; i++) H4_LDSR(fa[1][i], va[i], 32);
            }
        }
        // ---- k-steps 1..3: A fragments only (lgkmcnt), one halo piece written, one requested.  The piece's address arithmetic is
        // branch-free and cut into stages of 2-4 VALU, each pinned (empty asm on its results) behind one MFMA
#pragma unroll
        for (int ks = 1; ks < 4; ks++) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ks < 3) H4_LDSW(hdst[P][ks], hreg[P][ks]);
            const int q = q0 + ks - 1;
            const bool live = (VAR != 1) && q < hnp;                       // uniform
            unsigned pe;                                                  // the piece's table entry (pieces past the wave's share: entry 20, unused)
            asm volatile("ds_read_b32 %0, %1" : "=v"(pe) : "v"(tbl + (unsigned)((q < 20 ? q : 20) * 256)));
            mfma(ks, 0); mfma(ks, 1); mfma(ks, 2); mfma(ks, 3);
            if (ks < 3) {
#pragma unroll
                for (int i = 0; i < FM; i++) H4_LDSR(fa[(ks + 1) & 1][i], va[i], (ks + 1) * 32);
            } else {
                // every wave has retired its last read of this slice's halo (lgkmcnt above) and its last write of the next one:
                // after the barrier the next slice may be read and this buffer re-staged
                if (slice_last_tap) __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < FM; i++) { va[i] = vbase[i] + noff; H4_LDSR(fa[0][i], va[i], 0); }
            }
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(pe) :: "memory");   // the table entry (requested before the four A fragments)
            mfma(ks, 4);
            int hy = (int)((pe >> 20) & 63u), sp = (int)((pe >> 26) & 3u), xs = (int)((pe >> 13) & (STRIP ? 127u : 63u));
            if constexpr (STRIP) xs += ht_x0(ht) - 1;                  // image column of this lane's halo position
            H4_PIN3(hy, sp, xs);
            mfma(ks, 5);
            int y = ht_y0(ht) + hy - 1;
            const bool xin = STRIP ? ((unsigned)xs < (unsigned)WI) : (((pe >> 19) & 1u) != 0);
            unsigned okm = (live & xin & ((unsigned)y < (unsigned)H)) ? 0xffffffffu : 0u;
            if constexpr (STRIP) xs >>= ups;                           // (an invalid column's address is discarded through okm)
            H4_PIN2(y, okm);
            mfma(ks, 6);
            int t = __mul24(ht.b0 + sp, p.Hin) + (y >> ups);
            H4_PIN1(t);
            mfma(ks, 7);
            unsigned pix = (unsigned)__mul24(t, p.Win) + (unsigned)xs;
            unsigned long long ga = (unsigned long long)hs.src + (unsigned long long)pix * hs.ldb + (unsigned)(lc * 16);
            unsigned glo = (unsigned)ga, ghi = (unsigned)(ga >> 32);
            H4_PIN2(glo, ghi);
            if constexpr (FN >= 3) mfma(ks, 8);
            const unsigned long long zb = (unsigned long long)zl;
            glo = (glo & okm) | ((unsigned)zb & ~okm); ghi = (ghi & okm) | ((unsigned)(zb >> 32) & ~okm);
            H4_PIN2(glo, ghi);
            if constexpr (FN >= 3) mfma(ks, 9);
            unsigned hd = hbase_nxt + ((pe & 0x1fffu) << 4);
            hd = (live & (((pe >> 28) & 1u) != 0)) ? hd : dump;
            H4_PIN1(hd);
            const char* hg = (const char*)(((unsigned long long)ghi << 32) | glo);
            if constexpr (VAR == 2) hg = zl;                              // all the address work, no HBM request
            if constexpr (FN >= 3) mfma(ks, 10);
            H4_GLOADH(hreg[P][ks - 1], hg); hdst[P][ks - 1] = hd;
            if constexpr (FN >= 3) mfma(ks, 11);
        }
    };
    // after a step: next tap / slice; at the end of a work item its epilogue.  Returns true when the block has no work left.
    auto advance = [&]() -> bool {
        if (!slice_last_tap) { tap++; return false; }
        tap = 0; hb ^= 1; sl++;
        if (sl < s_end) { slice_setup(); return false; }

        // MFMA results -> VALU readers: the last MFMA needs 18 wait states before v_accvgpr_read (hipcc does not pad around asm).
        // After the last tile the requests of the (non-existent) next step are still in flight towards registers hipcc now
        // considers dead: drain them before anything else may be allocated there.
        if (!has_next) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

        if (p.dbg & 16) { tp1 = __builtin_readcyclecounter(); tprof[0] += tp1 - tp0; }
        // ---- epilogue.  The halo buffer of the slice just finished (hb ^ 1 after the toggle) is free for every wave: all of them
        // passed the slice-end barrier after their last read of it.  Staging is wave-private.
        char* const stg_base = smem + (hb ^ 1) * HBYTES;
        // lane-derived epilogue addressing is recomputed per tile from an opaque copy of the lane id: hoisted out of the tile loop
        // it would sit in registers across the main loop, which has none to spare
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int erow = lane_e & 31, ehalf = lane_e >> 5;
        // Read-out.  Fragment (i, j) register group g (4 registers) = channels j 32 + 8 g + 4 ehalf .. + 3 of pixel erow, fp32: the
        // groups go from the AGPRs STRAIGHT to a wave-private LDS tile (ds_write_b128 takes accumulator registers: no
        // v_accvgpr_read, no lane exchange, no conversion), one fragment row (32 pixels x WN channels) at a time, and come back
        // pixel-contiguous -- 8 consecutive channels per lane -- to be (+ residual) rounded once, packed and stored 16 bytes a lane.
        constexpr int CPR = WN / 8, NIT = (32 * CPR) / 64;
        static_assert((32 * CPR) % 64 == 0, "epilogue staging geometry");
        // LDS tile rows: WN fp32 + 16 bytes of padding where the free halo buffer has room (conflict-free as is); else unpadded with
        // the 16-byte unit index XOR-ed by (row >> 1) & 7 inside its group of 8 (an unpadded 384-byte row stride alone puts the 16
        // lanes of a ds_write_b128 phase on 8 banks)
        const bool swz = HBYTES < 4 * 32 * (WN * 4 + 16);                     // uniform
        const int SROW = swz ? WN * 4 : WN * 4 + 16;
        const unsigned stg0 = (unsigned)((hb ^ 1) * HBYTES + wave * 32 * SROW);
        const unsigned fw = swz ? (unsigned)((erow >> 1) & 7) : 0u;
        unsigned stgw[4];                                      // this lane's unit (2 g + ehalf) of a fragment's 8: + j * 128
#pragma unroll
        for (int g = 0; g < 4; g++) stgw[g] = stg0 + (unsigned)(erow * SROW) + ((((unsigned)(2 * g + ehalf)) ^ fw) << 4);
        const int eno = en0 + wn * WN;
        // the next work item's start values (bias + time-embedding row; zero for K-split parts): requested before the read-out
        float pb[FM][FN];
        start_values(nem0, nen0, erow, pb);
        FragU o1; o1.u = (u32x4_t){ehalf ? 0u : 0x3f803f80u, 0u, 0u, 0u};
        bf16x8 ones = o1.f;
        auto stage_row = [&](int i) {
#pragma unroll
            for (int j = 0; j < FN; j++) {
                H4_LDSW_ACC(stgw[0], i * FN + j, 0, j * 128); H4_LDSW_ACC(stgw[1], i * FN + j, 1, j * 128);
                H4_LDSW_ACC(stgw[2], i * FN + j, 2, j * 128); H4_LDSW_ACC(stgw[3], i * FN + j, 3, j * 128);
            }
            bf16x8 sf[FN];
#pragma unroll
            for (int j = 0; j < FN; j++) sf[j] = start_frag(pb[i][j], ehalf);
            // the LDS writes read their accumulator registers when they execute: retire them before the matrix pipe overwrites those
            if constexpr (FN == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(ones) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(sf[0]), "+v"(sf[1]), "+v"(ones) :: "memory");
#pragma unroll
            for (int j = 0; j < FN; j++) H4_MFMA0(i * FN + j, sf[j], ones);
        };
        if (S > 1) {
            // K-split part: fp32 planes [part][M][N]; 16-byte units (4 channels), WN / 4 lanes per row segment
            constexpr int CPR4 = WN / 4, NIT4 = (32 * CPR4) / 64;
            unsigned voffs[NIT4], lrd4[NIT4];
#pragma unroll
            for (int it = 0; it < NIT4; it++) {
                const int idx = it * 64 + lane_e, row = idx / CPR4, c4 = idx - row * CPR4;
                voffs[it] = (unsigned)(row * p.N + c4 * 4) * 4u;
                const unsigned fr = swz ? (unsigned)((row >> 1) & 7) : 0u;
                lrd4[it] = stg0 + (unsigned)(row * SROW) + ((((unsigned)c4 & ~7u) | (((unsigned)c4 ^ fr) & 7u)) << 4);
            }
            const char* const wbase = (const char*)(p.ws + (long long)part * p.M * p.N + (long long)(em0 + wm * 128) * p.N + eno);
            const unsigned long long rowstep = (unsigned long long)(32 * p.N) * 4ull;
#pragma unroll
            for (int i = 0; i < FM; i++) {
                stage_row(i);
                const char* const op = (const char*)h4_uni64((unsigned long long)(wbase + i * rowstep));
#pragma unroll
                for (int it = 0; it < NIT4; it++) {
                    const float4 a0 = *(const float4*)(smem + lrd4[it]);
                    const h4_u32x4 d0 = {__float_as_uint(a0.x), __float_as_uint(a0.y), __float_as_uint(a0.z), __float_as_uint(a0.w)};
                    H4_GSTORES(voffs[it], d0, op);
                }
            }
        } else {
            unsigned voffs[NIT], lrd[NIT];                         // chunk = 8 channels = two 16-byte LDS units (2 ch, 2 ch + 1)
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int idx = it * 64 + lane_e, row = idx / CPR, ch = idx - row * CPR;
                voffs[it] = (unsigned)(row * p.ldo + ch * 8) * 2u;
                const unsigned fr = swz ? (unsigned)((row >> 1) & 7) : 0u, c4 = (unsigned)(2 * ch);
                lrd[it] = stg0 + (unsigned)(row * SROW) + (((c4 & ~7u) | ((c4 ^ fr) & 7u)) << 4);
            }
            const unsigned lx = swz ? 16u : 0u, la = swz ? 0u : 16u;   // second unit: XOR 16 (swizzled) / + 16 (padded)
            // first output row (pixel index) of fragment row i of this wave: consecutive pixels of the tile, or -- STRIP -- half a strip row
            auto frag_row0 = [&](int i) -> long long {
                if constexpr (STRIP) {
                    const int tp = wm * 128 + i * 32;
                    return ((long long)(ht_tile.b0 * H + ht_y0(ht_tile) + (tp >> 6)) * WI + ht_x0(ht_tile) + (tp & 63));
                } else return (long long)(em0 + wm * 128 + i * 32);
            };
            const char* const obase = (const char*)(ob + eno);
            const char* const rbase = (const char*)(rb + eno);
            const unsigned long long rowbytes = (unsigned long long)p.ldo * 2ull;
            // residual rows: asm loads (saddr form) with counted waits.  Program order of the vector-memory requests:
            // R0 R1 | S0 (NIT stores) R2 | S1 R3 | S2 | S3 -- the wait in front of row i's adds leaves exactly the younger ones in flight
            h4_u32x4 rr4[2][NIT];
            auto res_request = [&](int i, h4_u32x4 (&dst)[NIT]) {
                const char* const rp = (const char*)h4_uni64((unsigned long long)(rbase + (unsigned long long)frag_row0(i) * rowbytes));
#pragma unroll
                for (int it = 0; it < NIT; it++) H4_GLOADB(dst[it], voffs[it], rp, 0);
            };
            auto res_wait = [&](h4_u32x4 (&r)[NIT], auto ntag) {
                constexpr int N = decltype(ntag)::value;
#pragma unroll
                for (int it = 0; it < NIT; it++) asm volatile("" : "+v"(r[it]));
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
#pragma unroll
                for (int it = 0; it < NIT; it++) asm volatile("" : "+v"(r[it]));
            };
            auto store_row = [&](int i) {
                const char* const op = (const char*)h4_uni64((unsigned long long)(obase + (unsigned long long)frag_row0(i) * rowbytes));
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    float4 a0 = *(const float4*)(smem + lrd[it]), a1 = *(const float4*)(smem + ((lrd[it] ^ lx) + la));
                    if (rb) {                                  // added in fp32: one rounding
                        const h4_u32x4 r4 = rr4[i & 1][it];
                        a0.x += __uint_as_float(r4.x << 16); a0.y += __uint_as_float(r4.x & 0xffff0000u);
                        a0.z += __uint_as_float(r4.y << 16); a0.w += __uint_as_float(r4.y & 0xffff0000u);
                        a1.x += __uint_as_float(r4.z << 16); a1.y += __uint_as_float(r4.z & 0xffff0000u);
                        a1.z += __uint_as_float(r4.w << 16); a1.w += __uint_as_float(r4.w & 0xffff0000u);
                    }
                    const h4_u32x4 dv = {cvt_pk_bf16(a0.x, a0.y), cvt_pk_bf16(a0.z, a0.w), cvt_pk_bf16(a1.x, a1.y), cvt_pk_bf16(a1.z, a1.w)};
                    H4_GSTORES(voffs[it], dv, op);
                }
            };
            if (rb) {
                res_request(0, rr4[0]); res_request(1, rr4[1]);
                stage_row(0); res_wait(rr4[0], std::integral_constant<int, NIT>{}); store_row(0);
                res_request(2, rr4[0]);
                stage_row(1); res_wait(rr4[1], std::integral_constant<int, 2 * NIT>{}); store_row(1);
                res_request(3, rr4[1]);
                stage_row(2); res_wait(rr4[0], std::integral_constant<int, 2 * NIT>{}); store_row(2);
                stage_row(3); res_wait(rr4[1], std::integral_constant<int, NIT>{}); store_row(3);
            } else {
                stage_row(0); store_row(0);
                stage_row(1); store_row(1);
                stage_row(2); store_row(2);
                stage_row(3); store_row(3);
            }
        }
        // (the epilogue's own loads -- bias, residual -- were consumed above, i.e. waited for; its stores are asm)
        epi_stores = (S > 1) ? 2 * FM * ((32 * (WN / 8)) / 64) : FM * ((32 * (WN / 8)) / 64);
        if (p.dbg & 16) tprof[1] += __builtin_readcyclecounter() - tp1;
        if (!has_next) return true;
        // the staging area is the buffer the next tile's SECOND slice is staged into during its first steps: every wave must have
        // left its epilogue first
        __builtin_amdgcn_s_barrier();
        tile = next;
        tile_setup();
        sl = s_begin;
        slice_setup();
        if (p.dbg & 16) tp0 = __builtin_readcyclecounter();
        return false;
    };
    // straight-line pairs of steps (parity 0, parity 1): no control-flow merge ever sits between the request of a fragment and its use
    while (true) {
        describe(); step(std::integral_constant<int, 0>{}); if (advance()) break;
        describe(); step(std::integral_constant<int, 1>{}); if (advance()) break;
    }
    if ((p.dbg & 16) && tid == 0) {
        atomicAdd(&g_halo4_prof[0], tprof[0]); atomicAdd(&g_halo4_prof[1], tprof[1]); atomicAdd(&g_halo4_prof[3], 1ull);
    }
}

template <int FN, int VAR, bool STRIP = false>
static hipError_t launch_halo4_cfg(const IgemmParams& p, hipStream_t st) {
    constexpr int smem = 2 * H4_HALO_MAX + 4096 + 4 * 21 * 256;       // halo x 2, dump, piece tables
    constexpr int BN = FN * 64;
    static int ncu_dev[RDM_MAX_DEVICES] = {0};
    const int dev = rdm_cur_device();
    if (!ncu_dev[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_halo4_kernel<FN, VAR, STRIP>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipDeviceGetAttribute(&ncu_dev[dev], hipDeviceAttributeMultiprocessorCount, dev);
    }
    const int ncu = ncu_dev[dev];
    const long long ntiles = (long long)(p.M / 256) * (p.N / BN) * (p.ksplit > 1 ? p.ksplit : 1);
    long long g = (ncu + 7) & ~7;
    if (g > ntiles) g = ntiles;
    static const int prof = getenv("RDM_HALO_PROF") ? atoi(getenv("RDM_HALO_PROF")) : 0;
    if (prof) {
        IgemmParams q = p; q.dbg |= 16;
        unsigned long long z[4] = {0, 0, 0, 0}, r[4];
        hipMemcpyToSymbol(HIP_SYMBOL(g_halo4_prof), z, sizeof(z));
        conv3x3_halo4_kernel<FN, VAR, STRIP><<<dim3((unsigned)g), 256, smem, st>>>(q);
        hipStreamSynchronize(st);
        hipMemcpyFromSymbol(r, HIP_SYMBOL(g_halo4_prof), sizeof(r));
        fprintf(stderr, "[halo4<%d> M=%d N=%d K=%d] blocks=%llu per-block cycles: main %.0f epilogue %.0f (tiles/block %.2f)\n", BN, p.M, p.N, p.K,
                r[3], (double)r[0] / r[3], (double)r[1] / r[3], (double)ntiles / g);
        return hipGetLastError();
    }
    conv3x3_halo4_kernel<FN, VAR, STRIP><<<dim3((unsigned)g), 256, smem, st>>>(p);
    return hipGetLastError();
}

// images wider than 64 pixels as 64-column strips (conv3x3_halo4_kernel<.., STRIP>): the first-stage decoder's 128- / 256-pixel levels
bool conv_halo4_strip_supported(const IgemmParams& p) {
    static const int off = getenv("RDM_NO_HALO4_STRIP") ? atoi(getenv("RDM_NO_HALO4_STRIP")) : 0;
    if (off || getenv("RDM_NO_HALO4") || getenv("RDM_NO_HALO")) return false;
    const int W = p.Wout, H = p.Hout;
    if (p.stride != 1 || W <= 64 || W % 64 || H % 4 || W > 4096 || H > 4096) return false;
    if (p.ups ? (p.Hout != 2 * p.Hin || p.Wout != 2 * p.Win) : (p.Hout != p.Hin || p.Wout != p.Win)) return false;
    if (p.M % 256 != 0 || p.N % 128 != 0 || p.ksplit > 1) return false;
    if (p.C0 % 64 || p.C1 % 64 || p.alpha != 1.0f || p.act != ACT_NONE || !p.out_bf16 || p.out_f32 || p.res_f32) return false;
    if (p.ldo % 8 || p.K != 9 * (p.C0 + p.C1)) return false;
    if (p.rowvec && p.rows_per_sample % 32 != 0) return false;
    if ((long long)p.M * (p.C0 > p.C1 ? p.C0 : p.C1) >= 0x7fffffffLL || (long long)p.M * p.ldo >= 0x7fffffffLL) return false;
    return true;
}

// the one-wave-per-SIMD kernel takes every conv the halo geometry admits once a fragment-ordered weight copy exists
bool conv_halo4_supported(const IgemmParams& p) {
    static const int off = getenv("RDM_NO_HALO4") ? atoi(getenv("RDM_NO_HALO4")) : 0;
    if (off || !p.Wfrag) return false;
    if (!conv_halo_supported(p)) return false;
    if (p.rowvec && p.rows_per_sample % 32 != 0) return false;
    if ((p.C0 + p.C1) % 64 != 0 || p.N % 32 != 0) return false;
    const Halo4Geom g = halo4_geom(p.Hout, p.Wout);
    if (g.HBYTES > H4_HALO_MAX || g.NPT > 4 * 21) return false;          // three pieces per wave per tap-step, staged during taps 0..6
    // the kernel's multiply-shift divisions by NPR and RS + 2 must be exact over the ranges met
    const int mNPR = 65536 / g.NPR + 1, mRS2 = 65536 / (g.RS + 2) + 1;
    for (int x = 0; x < 4 * 36 + 4; x++) if (((x * mNPR) >> 16) != x / g.NPR) return false;
    for (int x = 0; x <= g.NROW + 36; x++) if (((x * mRS2) >> 16) != x / (g.RS + 2)) return false;
    return true;
}

hipError_t launch_conv_halo4(const IgemmParams& p, hipStream_t st) {
    static const int var = getenv("RDM_H4_VAR") ? atoi(getenv("RDM_H4_VAR")) : 0;
    if (p.Wout > 64) {
        if (!p.Wfrag || !conv_halo4_strip_supported(p)) return hipErrorInvalidValue;
        return launch_halo4_cfg<2, 0, true>(p, st);
    }
    if (p.N % 192 == 0) {
        if (var == 1) return launch_halo4_cfg<3, 1>(p, st);
        if (var == 3) return launch_halo4_cfg<3, 3>(p, st);
        if (var == 2) return launch_halo4_cfg<3, 2>(p, st);
        return launch_halo4_cfg<3, 0>(p, st);
    }
    return launch_halo4_cfg<2, 0>(p, st);
}

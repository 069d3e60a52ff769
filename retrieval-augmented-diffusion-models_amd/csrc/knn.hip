// Exact brute-force cosine top-k over the CLIP-embedding database resident in HBM (gfx950).
// Replaces scann's searcher.search_batched (rdm/data/retrieval_dataset/dsetbuilder.py:490) and the
// numpy normalisation around it (:487, :574); the gather replaces data_pool['embedding'][nns] (:493).
//
// HBM-bound for <= 64 queries: the fp16 database is streamed once per group of 64 queries.
//   scan kernel   persistent blocks (one per CU) walk 256-row tiles; the tile's K-slices and the
//                 matching query slices are staged by 16-byte LDS-DMA into a double-buffered LDS ring
//                 (same source-swizzled layout as the GEMM), scored with 32x32x16 f16 MFMA.  The
//                 query is split q^ = q_hi + q_lo (two fp16 words) so the approximate score is
//                 fp32-accurate.  A lane holds 16 rows' scores of ONE query (MFMA D layout), so it
//                 keeps a private sorted top-KSEL list in registers (threshold test + unrolled
//                 insertion under a strict total order (score desc, index asc)).
//   merge kernel  per query: thread-local top-KSEL over all lane lists -> LDS bitonic sort ->
//                 the best 2*KSEL candidates are re-scored EXACTLY (fp64 accumulation of exact
//                 products, the oracle's definition) and sorted by (score desc, index asc).
// Exactness is a GUARANTEE, not a likelihood.  The approximate score differs from the exact one by at most KNN_EPS
// (fp32 accumulation of 1024 exact f16 x f16 products with partial sums <= 1: <= 1024 * 2^-24 = 6.1e-5 when each step rounds
// to nearest, doubled for an unknown rounding mode inside the MFMA, plus <= 7e-7 for the subnormal tail of the lo word).  Every
// row that is dropped anywhere (lane list, merge-thread list, the cut after the best 2*KSEL) has an approximate score <= theta,
// the maximum of the full lane lists' tails, the scores the merge threads rejected and the first score after the cut.  If
// theta + KNN_EPS < tau (the k-th EXACT score among the re-scored candidates) no dropped row can belong to the top k and the result
// is certified.  Otherwise (a cluster of near-duplicates around the k-th score -- real in OpenImages patches) the query is
// flagged and an exact pass streams the database once more in fp64, collecting every row that beats the current k-th candidate
// under the total order (score desc, index asc); the top k of those is the answer.  Unflagged batches pay three empty launches.
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include "kernels.h"
#include "knn.h"

#define KNN_ROWS 256           // rows per tile
#define KNN_Q 64               // queries per pass (online search)
#define KNN_QMAX 512           // queries per launch of the bulk scan (4 groups of 128): size of the per-query scratch arrays
#define KNN_BK 64              // K slice per stage

constexpr float KNN_EPS = 2.0e-4f;   // bound on |approximate - exact| score (derivation above)
constexpr float KNN_EPS_BULK = 5.0e-4f;   // bulk scan: hi word of the query only (+ 2^-12 sum |q_j d_j| <= 2.4e-4)
constexpr int KNN_FB_CAP = 8192;      // exact-fallback candidates kept per query
struct Cand { float s; uint32_t i; };
__device__ __forceinline__ bool better(float s, uint32_t i, float ts, uint32_t ti) { return s > ts || (s == ts && i < ti); }

template <int KSEL>
__device__ __forceinline__ void list_insert(float (&ls)[KSEL], uint32_t (&li)[KSEL], float s, uint32_t i) {
    if (!better(s, i, ls[KSEL - 1], li[KSEL - 1])) return;
#pragma unroll
    for (int j = 0; j < KSEL; j++) {
        if (better(s, i, ls[j], li[j])) { const float ts = ls[j]; const uint32_t ti = li[j]; ls[j] = s; li[j] = i; s = ts; i = ti; }
    }
}

// Scan-side insertion: a lane visits its rows in ASCENDING index order, so a new candidate loses every score tie against what the
// list already holds -- better() collapses to one strict compare, and the insert is a one-pass shift (2 compares + 4 selects per
// entry) instead of the compare-and-swap bubble.  Same resulting list as list_insert for such a stream.
template <int KSEL>
__device__ __forceinline__ void list_insert_ascending(float (&ls)[KSEL], uint32_t (&li)[KSEL], float s, uint32_t i) {
    if (!(s > ls[KSEL - 1])) return;
#pragma unroll
    for (int j = KSEL - 1; j >= 1; j--) {
        const bool up = s > ls[j - 1];                  // entry j-1 moves down into j
        const bool here = !up && s > ls[j];
        ls[j] = up ? ls[j - 1] : (here ? s : ls[j]);
        li[j] = up ? li[j - 1] : (here ? i : li[j]);
    }
    if (s > ls[0]) { ls[0] = s; li[0] = i; }
}

// ---------------------------------------------------------------- database / query preparation
// one wave per row: n = fp32(sqrt(sum x^2 in fp64)); out = fp16(x / n)
template <typename TIN>
__global__ __launch_bounds__(256) void knn_normalize_rows_kernel(const TIN* x, _Float16* out, long long n, int dim) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    double acc = 0.0;
    for (int c = lane; c < dim; c += 64) { const double v = (double)(float)x[row * dim + c]; acc += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const float nrm = (float)sqrt(acc);
    for (int c = lane; c < dim; c += 64) {
        const float v = (float)x[row * dim + c];
        out[row * dim + c] = (_Float16)(nrm > 0.f ? v / nrm : 0.f);
    }
}

// q f32 [b, dim] -> qn f32 [64, dim] (normalised, zero padded), qh / ql fp16 [64, dim]
__global__ void knn_prep_queries_kernel(const float* q, int b, int dim, float* qn, _Float16* qh, _Float16* ql) {
    const int row = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    if (row < b) for (int c = lane; c < dim; c += 64) { const double v = (double)q[(long long)row * dim + c]; acc += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const float nrm = (float)sqrt(acc);
    for (int c = lane; c < dim; c += 64) {
        float v = 0.f;
        if (row < b) v = q[(long long)row * dim + c] / nrm;
        const _Float16 h = (_Float16)v;
        qn[(long long)row * dim + c] = v;
        qh[(long long)row * dim + c] = h;
        ql[(long long)row * dim + c] = (_Float16)(v - (float)h);
    }
}

// ---------------------------------------------------------------- scan
struct ScanParams {
    const _Float16* dbn; long long n; int dim; long long ntiles;
    const _Float16* qh; const _Float16* ql;
    float* cand_s; uint32_t* cand_i;      // [64][nlists][KSEL]
    int nlists; const void* zero_page;
    int hi_only;                          // score with the hi word of the query only (certificate bound KNN_EPS_BULK)
};

template <int KSEL>
__global__ __launch_bounds__(256) void knn_scan_kernel(ScanParams p) {
    constexpr int DB_BYTES = KNN_ROWS * 128, Q_BYTES = KNN_Q * 128, STAGE = DB_BYTES + 2 * Q_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = tid >> 3, pchunk = tid & 7;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int nkc = p.dim / KNN_BK;
    const char* zero = (const char*)p.zero_page;

    float ls[2][KSEL]; uint32_t li[2][KSEL];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int j = 0; j < KSEL; j++) { ls[t][j] = -INFINITY; li[t][j] = 0xffffffffu; }

    // my tiles: blockIdx.x, +gridDim.x, ...
    const long long my_tiles = (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long iters = my_tiles * nkc;

    auto stage = [&](long long it, int buf) {
        const long long tl = it / nkc; const int kc = (int)(it - tl * nkc);
        const long long tile = blockIdx.x + tl * gridDim.x;
        char* Ds = smem + buf * STAGE; char* Qh = Ds + DB_BYTES; char* Ql = Qh + Q_BYTES;
#pragma unroll
        for (int i = 0; i < KNN_ROWS / 32; i++) {
            const int r = i * 32 + lrow;
            const long long row = tile * KNN_ROWS + r;
            const int c = pchunk ^ ((r >> 1) & 7);
            const void* g = (row < p.n) ? (const void*)(p.dbn + row * p.dim + kc * KNN_BK + c * 8) : (const void*)zero;
            glds16(g, Ds + (i * 32 + wave * 8) * 128);
        }
#pragma unroll
        for (int i = 0; i < KNN_Q / 32; i++) {
            const int r = i * 32 + lrow;
            const int c = pchunk ^ ((r >> 1) & 7);
            glds16(p.qh + (long long)r * p.dim + kc * KNN_BK + c * 8, Qh + (i * 32 + wave * 8) * 128);
            glds16(p.ql + (long long)r * p.dim + kc * KNN_BK + c * 8, Ql + (i * 32 + wave * 8) * 128);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    if (iters > 0) stage(0, 0);
    for (long long it = 0; it < iters; it++) {
        const int cur = (int)(it & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 1 < iters) stage(it + 1, cur ^ 1);
        const char* Ds = smem + cur * STAGE; const char* Qh = Ds + DB_BYTES; const char* Ql = Qh + Q_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int chunk = kk * 2 + fhalf;
            f16x8 a[2], bh[2], bl[2];
#pragma unroll
            for (int rf = 0; rf < 2; rf++) {
                const int row = wave * 64 + rf * 32 + frow;
                a[rf] = *(const f16x8*)(Ds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                const int row = qt * 32 + frow;
                const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
                bh[qt] = *(const f16x8*)(Qh + off);
                bl[qt] = *(const f16x8*)(Ql + off);
            }
#pragma unroll
            for (int rf = 0; rf < 2; rf++)
#pragma unroll
                for (int qt = 0; qt < 2; qt++) {
                    acc[rf][qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rf], bl[qt], acc[rf][qt], 0, 0, 0);
                    acc[rf][qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rf], bh[qt], acc[rf][qt], 0, 0, 0);
                }
        }
        const long long tl = it / nkc; const int kc = (int)(it - tl * nkc);
        if (kc == nkc - 1) {
            const long long tile = blockIdx.x + tl * gridDim.x;
            const long long rbase = tile * KNN_ROWS + wave * 64 + 4 * fhalf;
#pragma unroll
            for (int rf = 0; rf < 2; rf++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const long long row = rbase + rf * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < p.n) {
                        list_insert<KSEL>(ls[0], li[0], acc[rf][0][r], (uint32_t)row);
                        list_insert<KSEL>(ls[1], li[1], acc[rf][1][r], (uint32_t)row);
                    }
                    acc[rf][0][r] = 0.f; acc[rf][1][r] = 0.f;
                }
        }
    }
    const int list_id = blockIdx.x * 8 + wave * 2 + fhalf;
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        const long long base = ((long long)(qt * 32 + frow) * p.nlists + list_id) * KSEL;
#pragma unroll
        for (int j = 0; j < KSEL; j++) { p.cand_s[base + j] = ls[qt][j]; p.cand_i[base + j] = li[qt][j]; }
    }
}

// ---------------------------------------------------------------- scan, dim == 512 (CLIP ViT-B/32: the reference's only database)
// The queries never touch LDS: wave w owns 32 queries (w & 1) and keeps their hi / lo fp16 words for ALL 512 dimensions as
// MFMA B fragments in registers (2 x 32 fragments = 256 registers; one wave per SIMD has 512), against 128 of the tile's 256
// rows (w >> 1).  LDS holds only the database ring (3 stages of 256 rows x 64 dims, requested two stages ahead with counted
// vmcnt across raw barriers), so a stage costs 32 LDS-DMA pieces instead of 48 and 4 instead of 6 LDS reads per k-step; the
// K loop of a tile is fully unrolled (the register-resident fragments need compile-time indices).
// Measured (20.9 M x 512, 64 queries): 5.2 ms = 4.1 TB/s of database streamed, vs 6.6 ms for the LDS-staged-query kernel; the
// bare LDS-DMA stream of the same walk runs at 6.3 TB/s (tools/ubench/hbm_pattern.hip), the gap is LDS-DMA issue time that a
// single wave per SIMD cannot overlap with its own MFMAs.
template <int KSEL>
__global__ __launch_bounds__(256, 1) void knn_scan512_kernel(ScanParams p) {
    constexpr int DB_BYTES = KNN_ROWS * 128, NKC = 8, DIM = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = tid >> 3, pchunk = tid & 7;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int rh = wave >> 1, qt = wave & 1;
    const char* zero = (const char*)p.zero_page;

    f16x8 qh[32], ql[32];
    {
        const _Float16* qhp = p.qh + (long long)(qt * 32 + frow) * DIM + fhalf * 8;
        const _Float16* qlp = p.ql + (long long)(qt * 32 + frow) * DIM + fhalf * 8;
#pragma unroll
        for (int kk = 0; kk < 32; kk++) { qh[kk] = *(const f16x8*)(qhp + kk * 16); ql[kk] = *(const f16x8*)(qlp + kk * 16); }
    }
    float ls[KSEL]; uint32_t li[KSEL];
#pragma unroll
    for (int j = 0; j < KSEL; j++) { ls[j] = -INFINITY; li[j] = 0xffffffffu; }

    const long long my_tiles = (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long iters = my_tiles * NKC;
    // Address arithmetic is the cost of an LDS-DMA request here (the 64-bit row * dim products of a naive loader took ~900 cycles
    // per stage, more than the 32 requests' time on the memory path): one 64-bit lane pointer per TILE, pieces and K chunks are
    // constant byte offsets from it; the swizzle term does not depend on the piece (32 | piece stride).
    const int csw = (pchunk ^ ((lrow >> 1) & 7)) * 16;    // byte offset of my (swizzled) source chunk inside a 128-byte segment
    auto tile_ptr = [&](long long tile) { return (const char*)p.dbn + (tile * KNN_ROWS + lrow) * (long long)(DIM * 2) + csw; };
    auto tile_rows = [&](long long tile) { const long long left = p.n - tile * KNN_ROWS; return (int)(left > KNN_ROWS ? KNN_ROWS : left); };
    // stage st_kc of the tile whose lane pointer is st_ptr (st_rows valid rows) -> ring slot st_slot; then advance the stream
    const char* st_ptr = tile_ptr(blockIdx.x); int st_rows = tile_rows(blockIdx.x); int st_kc = 0, st_slot = 0; long long st_tile = blockIdx.x;
    auto stage_next = [&]() {
        char* Ds = smem + st_slot * DB_BYTES;
        const char* src = st_ptr + st_kc * (KNN_BK * 2);
#pragma unroll
        for (int i = 0; i < KNN_ROWS / 32; i++) {
            const void* g = (i * 32 + lrow < st_rows) ? (const void*)(src + (long long)i * 32 * DIM * 2) : (const void*)zero;
            glds16(g, Ds + (i * 32 + wave * 8) * 128);
        }
        st_slot = st_slot == 2 ? 0 : st_slot + 1;
        if (++st_kc == NKC) { st_kc = 0; st_tile += gridDim.x; st_ptr = tile_ptr(st_tile); st_rows = tile_rows(st_tile); }
    };
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    if (iters > 0) stage_next();
    if (iters > 1) stage_next();
    long long it = 0;
    for (long long tl = 0; tl < my_tiles; tl++) {
#pragma unroll
        for (int kc = 0; kc < NKC; kc++, it++) {
            // needed now: stage `it`; the 8 pieces of stage it+1 (requested after it) may stay in flight
            if (it + 1 < iters) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // raw barrier (a __syncthreads would drain vmcnt): stage landed for all
            if (it + 2 < iters) stage_next();             // waves; the slot of stage it-1 is free (its reads fed MFMAs already)
            const char* Ds = smem + (int)(it % 3) * DB_BYTES;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int chunk = k * 2 + fhalf, kk = kc * 4 + k;
                f16x8 a[4];
#pragma unroll
                for (int rf = 0; rf < 4; rf++) {
                    const int row = rh * 128 + rf * 32 + frow;
                    a[rf] = *(const f16x8*)(Ds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int rf = 0; rf < 4; rf++) {
                    acc[rf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rf], ql[kk], acc[rf], 0, 0, 0);
                    acc[rf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rf], qh[kk], acc[rf], 0, 0, 0);
                }
            }
        }
        const long long tile = blockIdx.x + tl * gridDim.x;
        const long long rbase = tile * KNN_ROWS + rh * 128 + 4 * fhalf;
#pragma unroll
        for (int rf = 0; rf < 4; rf++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long row = rbase + rf * 32 + (r & 3) + 8 * (r >> 2);
                if (row < p.n) list_insert<KSEL>(ls, li, acc[rf][r], (uint32_t)row);
                acc[rf][r] = 0.f;
            }
    }
    const int list_id = blockIdx.x * 4 + rh * 2 + fhalf;
    const long long base = ((long long)(qt * 32 + frow) * p.nlists + list_id) * KSEL;
#pragma unroll
    for (int j = 0; j < KSEL; j++) { p.cand_s[base + j] = ls[j]; p.cand_i[base + j] = li[j]; }
}

// ---------------------------------------------------------------- scan, dim == 512, 8 waves (two per SIMD), KSEL == 8
// Same idea as knn_scan512_kernel (queries resident in registers, database-only LDS ring) with 16x16x32 MFMAs: a wave owns 16
// queries (128 registers of hi / lo fragments) x 128 rows, so 8 waves = 2 row halves x 4 query groups fit two per SIMD (256
// registers each) and one wave's LDS-DMA issue, LDS latency and list insertion hide behind its SIMD partner's MFMAs.
// D layout 16x16: col (query) = lane & 15, row = (lane >> 4) * 4 + reg -> still one query, one private top-k list per lane.
template <int KSEL, bool HI_ONLY>    // HI_ONLY: score with the hi fp16 word of the query (certificate bound KNN_EPS_BULK); halves the MFMA work, and the
                                      // lighter loop holds a higher clock: 4.99 -> 4.48 ms per 64 queries over 20.9 M rows
__global__ __launch_bounds__(512, 2) void knn_scan512w8_kernel(ScanParams p) {
    constexpr int DB_BYTES = KNN_ROWS * 128, NKC = 8, DIM = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = tid >> 3, pchunk = tid & 7;             // loader: 64 rows x 8 chunks per piece
    const int l15 = lane & 15, kq = lane >> 4;
    const int rh = wave >> 2, qg = wave & 3;
    const char* zero = (const char*)p.zero_page;

    f16x8 qh[16], ql[HI_ONLY ? 1 : 16];                       // B fragments of my 16 queries, all 16 k-steps of 32
    {
        const _Float16* qhp = p.qh + (long long)(qg * 16 + l15) * DIM + kq * 8;
        const _Float16* qlp = p.ql + (long long)(qg * 16 + l15) * DIM + kq * 8;
#pragma unroll
        for (int s = 0; s < 16; s++) { qh[s] = *(const f16x8*)(qhp + s * 32); if constexpr (!HI_ONLY) ql[s] = *(const f16x8*)(qlp + s * 32); }
    }
    float ls[KSEL]; uint32_t li[KSEL];
#pragma unroll
    for (int j = 0; j < KSEL; j++) { ls[j] = -INFINITY; li[j] = 0xffffffffu; }

    const long long my_tiles = (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long iters = my_tiles * NKC;
    const int csw = (pchunk ^ ((lrow >> 1) & 7)) * 16;
    auto tile_ptr = [&](long long tile) { return (const char*)p.dbn + (tile * KNN_ROWS + lrow) * (long long)(DIM * 2) + csw; };
    auto tile_rows = [&](long long tile) { const long long left = p.n - tile * KNN_ROWS; return (int)(left > KNN_ROWS ? KNN_ROWS : left); };
    const char* st_ptr = tile_ptr(blockIdx.x); int st_rows = tile_rows(blockIdx.x); int st_kc = 0, st_slot = 0; long long st_tile = blockIdx.x;
    auto stage_next = [&]() {                                 // 4 pieces of 64 rows per thread
        char* Ds = smem + st_slot * DB_BYTES;
        const char* src = st_ptr + st_kc * (KNN_BK * 2);
#pragma unroll
        for (int i = 0; i < KNN_ROWS / 64; i++) {
            const void* g = (i * 64 + lrow < st_rows) ? (const void*)(src + (long long)i * 64 * DIM * 2) : (const void*)zero;
            glds16(g, Ds + (i * 64 + wave * 8) * 128);
        }
        st_slot = st_slot == 2 ? 0 : st_slot + 1;
        if (++st_kc == NKC) { st_kc = 0; st_tile += gridDim.x; st_ptr = tile_ptr(st_tile); st_rows = tile_rows(st_tile); }
    };
    f32x4 acc[8];
#pragma unroll
    for (int a = 0; a < 8; a++) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (iters > 0) stage_next();
    if (iters > 1) stage_next();
    long long it = 0;
    for (long long tl = 0; tl < my_tiles; tl++) {
#pragma unroll
        for (int kc = 0; kc < NKC; kc++, it++) {
            if (it + 1 < iters) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // stage it landed; stage it+1 (4 pieces) may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (it + 2 < iters) stage_next();
            const char* Ds = smem + (int)(it % 3) * DB_BYTES;
#pragma unroll
            for (int h = 0; h < 2; h++) {                     // two k-steps of 32 per 64-dim stage
                const int chunk = h * 4 + kq, s = kc * 2 + h;
                f16x8 a[8];
#pragma unroll
                for (int rf = 0; rf < 8; rf++) {
                    const int row = rh * 128 + rf * 16 + l15;
                    a[rf] = *(const f16x8*)(Ds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
                }
                if constexpr (!HI_ONLY) {
#pragma unroll
                    for (int rf = 0; rf < 8; rf++) acc[rf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rf], ql[s], acc[rf], 0, 0, 0);
                }
#pragma unroll
                for (int rf = 0; rf < 8; rf++) acc[rf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rf], qh[s], acc[rf], 0, 0, 0);
            }
        }
        const long long tile = blockIdx.x + tl * gridDim.x;
        const long long rbase = tile * KNN_ROWS + rh * 128 + kq * 4;
        if ((tile + 1) * KNN_ROWS <= p.n) {                   // full tile (all but the last): no per-row bound check
            const uint32_t rb = (uint32_t)rbase;
#pragma unroll
            for (int rf = 0; rf < 8; rf++)
#pragma unroll
                for (int r = 0; r < 4; r++) { list_insert_ascending<KSEL>(ls, li, acc[rf][r], rb + rf * 16 + r); acc[rf][r] = 0.f; }
        } else {
#pragma unroll
            for (int rf = 0; rf < 8; rf++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const long long row = rbase + rf * 16 + r;
                    if (row < p.n) list_insert_ascending<KSEL>(ls, li, acc[rf][r], (uint32_t)row);
                    acc[rf][r] = 0.f;
                }
        }
    }
    const int list_id = blockIdx.x * 8 + rh * 4 + kq;
    const long long base = ((long long)(qg * 16 + l15) * p.nlists + list_id) * KSEL;
#pragma unroll
    for (int j = 0; j < KSEL; j++) { p.cand_s[base + j] = ls[j]; p.cand_i[base + j] = li[j]; }
}


// ---------------------------------------------------------------- bulk scan: 128 queries per walker, four query groups per database pass
// Offline neighbour pre-computation (scripts/search_neighbors.py:380-450: 1.28 M ImageNet queries against the 20.9 M-row database) is
// the same top-k at dataset scale.  One database pass per 64 queries (above) is HBM-bound there; this kernel scores 128 queries per
// block against each 256-row tile, and a launch carries up to FOUR query groups whose walkers sit on the same XCD and walk the same
// tiles, so three of them read the tile from that XCD's L2: HBM streams the database once per 512 queries.  Scores come from the hi
// word of the query only: |q - fp16(q)| <= 2^-12 |q| per element moves a score by at most 2^-12 sum |q_j d_j| <= 2.4e-4 (unit
// vectors) -- the certificate in the merge runs with KNN_EPS_BULK and the exact fallback absorbs the (rare) failures.  A stage is 256
// database rows + 128 query rows x 64 dims = 48 KB, three stages in an LDS ring requested two ahead.
//
// The block is cut into producer and consumer waves.  The first version (eight waves of 128 rows x 32 queries that also issued the
// LDS-DMA requests) was measured with ablation bits and counters at 4096 x 20.9 M: requests alone 54 ms, LDS reads + MFMA alone 54 ms,
// both together 108 ms -- nothing overlapped: every wave issued its requests right after the stage barrier (a wave whose request
// meets a full queue is parked), and the eight waves read 160 KB of LDS per stage for 16 MFMAs each.  Here
//   * waves 0-3 score: one per SIMD, 128 rows x 64 queries each (2 row halves x 2 query halves), 32 MFMAs per stage from 24 LDS
//     reads (96 KB per stage for the block); they never issue a memory request;
//   * waves 4-7 load: one per SIMD, each owns a quarter of every 64-row piece (12 requests per stage, counted vmcnt), waits for
//     its share of stage `it` and meets the scoring waves at the barrier.
// A lane owns TWO queries (one per query fragment), i.e. two private lists.  294 ms -> 110 ms for 4096 x 20.9 M, k = 20 (with the
// 8-entry lists, the ascending insertion and the 16-candidate pre-compare below); an idle sleep in place of the MFMAs hides completely
// behind the requests (62 ms), the real MFMA + LDS-read stream does not (96 ms without insertion): what is left is contention between
// the LDS-DMA stream (48 KB per stage and block at ~26 B/clk) and the scoring waves, not the ring depth.
struct BulkParams { ScanParams s; int groups; int dbg; };      // s.qh: [groups*128][dim]; s.cand_*: [groups*128][nlists][KSEL]; dbg: ablation bits (RDM_KNN_BULK_DBG)

template <int KSEL>
__global__ __launch_bounds__(512, 1) void knn_scan_bulk_kernel(BulkParams bp) {
    const ScanParams& p = bp.s;
    constexpr int QB = 128, DB_BYTES = KNN_ROWS * 128, Q_BYTES = QB * 128, STAGE = DB_BYTES + Q_BYTES, NSLOT = 3;
    constexpr int NP = KNN_ROWS / 64 + QB / 64;               // 64-row pieces per stage (4 database + 2 query)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int rg = (wave >> 1) & 1, qh2 = wave & 1;            // scoring wave: row half, query half
    const int nkc = p.dim / KNN_BK;
    const char* zero = (const char*)p.zero_page;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = slot % bp.groups;
    const int walkers = (int)(gridDim.x / bp.groups);
    const int wk = ((slot / bp.groups) << 3) | xcd;
    if (wk >= walkers) return;
    const _Float16* qh = p.qh + (long long)grp * QB * p.dim;
    const long long my_tiles = (p.ntiles - wk + walkers - 1) / walkers;
    const long long iters = my_tiles * nkc;

    if (loader) {
        // loader wave lw requests sub-pieces 2 lw and 2 lw + 1 (8 rows x 128 B each) of every 64-row piece
        const int lw = wave - 4;
        const int r0 = lw * 16 + (lane >> 3), r1 = r0 + 8;    // my two rows inside a piece
        const int pchunk = lane & 7;
        const long long row_bytes = (long long)p.dim * 2;
        const int csw0 = (pchunk ^ ((r0 >> 1) & 7)) * 16, csw1 = (pchunk ^ ((r1 >> 1) & 7)) * 16;
        auto tile_rows = [&](long long tile) { const long long left = p.n - tile * KNN_ROWS; return (int)(left > KNN_ROWS ? KNN_ROWS : (left < 0 ? 0 : left)); };
        const char* dbase = (const char*)p.dbn + ((long long)wk * KNN_ROWS + r0) * row_bytes;   // row r0 of my current tile
        const long long tile_step = (long long)walkers * KNN_ROWS * row_bytes;
        const char* qbase = (const char*)qh + (long long)r0 * row_bytes;
        int st_rows = tile_rows(wk), st_kc = 0, st_slot = 0; long long st_tile = wk;
        auto stage_next = [&]() {
            char* Ds = smem + st_slot * STAGE; char* Qs = Ds + DB_BYTES;
            const char* src = dbase + st_kc * (KNN_BK * 2);
#pragma unroll
            for (int i = 0; i < KNN_ROWS / 64; i++) {
                const char* s0 = src + (long long)i * 64 * row_bytes;
                const void* g0 = (i * 64 + r0 < st_rows) ? (const void*)(s0 + csw0) : (const void*)zero;
                const void* g1 = (i * 64 + r1 < st_rows) ? (const void*)(s0 + 8 * row_bytes + csw1) : (const void*)zero;
                glds16(g0, Ds + (i * 64 + lw * 16) * 128);
                glds16(g1, Ds + (i * 64 + lw * 16 + 8) * 128);
            }
            const char* qsrc = qbase + st_kc * (KNN_BK * 2);
#pragma unroll
            for (int i = 0; i < QB / 64; i++) {
                const char* s0 = qsrc + (long long)i * 64 * row_bytes;
                glds16(s0 + csw0, Qs + (i * 64 + lw * 16) * 128);
                glds16(s0 + 8 * row_bytes + csw1, Qs + (i * 64 + lw * 16 + 8) * 128);
            }
            st_slot = st_slot == NSLOT - 1 ? 0 : st_slot + 1;
            if (++st_kc == nkc) { st_kc = 0; st_tile += walkers; dbase += tile_step; st_rows = tile_rows(st_tile); }
        };
        static_assert(NP * 2 == 12, "counted vmcnt below");
        if (iters > 0) stage_next();
        if (iters > 1) stage_next();
        for (long long it = 0; it < iters; it++) {
            if (it + 1 < iters) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // stage `it` landed; the 12 requests of stage it+1 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // everyone is done reading the slot of stage it-1
            if (it + 2 < iters && !(bp.dbg & 1)) stage_next();
        }
        return;
    }

    float ls[2][KSEL]; uint32_t li[2][KSEL];
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
        for (int j = 0; j < KSEL; j++) { ls[f][j] = -INFINITY; li[f][j] = 0xffffffffu; }
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][f][r] = 0.f;
    int cslot = 0, ckc = 0; long long ctile = wk;
    // LDS byte addresses of my fragment rows for the four k-steps of a stage (slot base added per stage).  (row >> 1) & 7 of every
    // fragment row equals (frow >> 1) & 7: the row / query offsets are multiples of 32 rows, applied as immediates.
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    unsigned la[4], lb[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const unsigned o = (unsigned)(frow * 128 + (((kk * 2 + fhalf) ^ ((frow >> 1) & 7)) << 4));
        la[kk] = lds0 + o + (unsigned)(rg * 128 * 128);
        lb[kk] = lds0 + o + (unsigned)(DB_BYTES + qh2 * 64 * 128);
    }
#define KNN_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
    for (long long it = 0; it < iters; it++) {
        __builtin_amdgcn_s_barrier();                     // stage `it` landed (the loader waves waited for it before arriving here)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned sb = (unsigned)(cslot * STAGE);
        cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
        if (bp.dbg & 16) __builtin_amdgcn_s_sleep(16);    // ~1024 idle cycles per stage in place of the MFMAs (with bit 2)
        if (!(bp.dbg & 2)) {
            // One scoring wave per SIMD: nobody else hides its LDS latency, so the fragments of k-step kk+1 are requested before the
            // MFMAs of k-step kk are issued.  Left to the scheduler the loop came out as read-6 / wait / 8 MFMAs (2.1 k cycles per
            // stage for 1 k cycles of MFMA), hence explicit reads with counted lgkmcnt waits.
            f16x8 fa[2][4], fb[2][2];
            {
                const unsigned xa = la[0] + sb, xb = lb[0] + sb;
                KNN_LDS_READ(fa[0][0], xa, 0); KNN_LDS_READ(fa[0][1], xa, 4096); KNN_LDS_READ(fa[0][2], xa, 8192); KNN_LDS_READ(fa[0][3], xa, 12288);
                KNN_LDS_READ(fb[0][0], xb, 0); KNN_LDS_READ(fb[0][1], xb, 4096);
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
                    const unsigned xa = la[kk + 1] + sb, xb = lb[kk + 1] + sb;
                    KNN_LDS_READ(fa[nxt][0], xa, 0); KNN_LDS_READ(fa[nxt][1], xa, 4096); KNN_LDS_READ(fa[nxt][2], xa, 8192); KNN_LDS_READ(fa[nxt][3], xa, 12288);
                    KNN_LDS_READ(fb[nxt][0], xb, 0); KNN_LDS_READ(fb[nxt][1], xb, 4096);
                    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rf = 0; rf < 4; rf++)
#pragma unroll
                    for (int f = 0; f < 2; f++) acc[rf][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][rf], fb[cur][f], acc[rf][f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++ckc == nkc) {
            ckc = 0;
            const long long rbase = ctile * KNN_ROWS + rg * 128 + 4 * fhalf;
            const bool full = (ctile + 1) * KNN_ROWS <= p.n;      // all but the last tile: no per-row bound check
            ctile += walkers;
            if (!full) {                                   // last tile: rows past the end (zero page) must lose
#pragma unroll
                for (int rf = 0; rf < 4; rf++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const bool dead = rbase + rf * 32 + (r & 3) + 8 * (r >> 2) >= p.n;
                        acc[rf][0][r] = dead ? -INFINITY : acc[rf][0][r];
                        acc[rf][1][r] = dead ? -INFINITY : acc[rf][1][r];
                    }
            }
            const uint32_t rb = (uint32_t)rbase;
            const bool ins = !(bp.dbg & 4);
            auto insert_half = [&](auto F) {               // compile-time query fragment: the lists must stay in registers
                constexpr int f = decltype(F)::value;
#pragma unroll
                for (int rf = 0; rf < 4; rf++) {
                    // one compare per 16 candidates: late in the scan hardly any 32 x 32 block holds a row that enters a list
                    float m = acc[rf][f][0];
#pragma unroll
                    for (int r = 1; r < 16; r++) m = fmaxf(m, acc[rf][f][r]);
                    if (ins && m > ls[f][KSEL - 1]) {
#pragma unroll
                        for (int r = 0; r < 16; r++) list_insert_ascending<KSEL>(ls[f], li[f], acc[rf][f][r], rb + rf * 32 + (r & 3) + 8 * (r >> 2));
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[rf][f][r] = 0.f;
                }
            };
            insert_half(std::integral_constant<int, 0>{});
            insert_half(std::integral_constant<int, 1>{});
        }
    }
    const int list_id = wk * 4 + rg * 2 + fhalf;
#pragma unroll
    for (int f = 0; f < 2; f++) {
        const long long base = ((long long)(grp * QB + qh2 * 64 + f * 32 + frow) * p.nlists + list_id) * KSEL;
#pragma unroll
        for (int j = 0; j < KSEL; j++) { p.cand_s[base + j] = ls[f][j]; p.cand_i[base + j] = li[f][j]; }
    }
}

// ---------------------------------------------------------------- merge + exact re-score
struct Certify {
    int* flag;                          // [64] 1 = candidate set not provably sufficient -> exact fallback for this query
    double* tau; uint32_t* tau_idx;     // [64] the current k-th (exact score, index)
    int* fb_count;                      // [64] rows collected by the fallback
    double* fb_s; uint32_t* fb_i;       // [64][KNN_FB_CAP]
    int* status;                        // sticky bits for the host: 1 = a fallback ran, 2 = a fallback overflowed KNN_FB_CAP
};
struct MergeParams {
    const float* cand_s; const uint32_t* cand_i; int nlists;
    const _Float16* dbn; const float* qn; int dim; long long n;
    int k; uint32_t* idx_out; float* score_out; double* score64_out; int qbase;   // output row = qbase + blockIdx.x; score64_out: the fp64 scores themselves (shard merges)
    Certify cert; float eps;
};

// the oracle's score: exact products, fp64 accumulation.  ONE summation order (lane-strided, xor-butterfly) shared by the merge
// and the fallback so that a row scores bit-identically wherever it is evaluated.  All 64 lanes get the result.
__device__ __forceinline__ double exact_score(const float* qn_row, const _Float16* db_row, int dim, int lane) {
    double acc = 0.0;
    for (int c = lane; c < dim; c += 64) acc += (double)qn_row[c] * (double)(float)db_row[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}

template <int KSEL, int R>       // R: re-scored prefix of the merged candidates, >= k (16 for k <= 4, 64 up to k = 28)
__global__ __launch_bounds__(256) void knn_merge_kernel(MergeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 256 * KSEL;
    float* ss = (float*)smem; uint32_t* si = (uint32_t*)(smem + NS * 4);
    const int q = blockIdx.x, tid = threadIdx.x;
    float ls[KSEL]; uint32_t li[KSEL];
#pragma unroll
    for (int j = 0; j < KSEL; j++) { ls[j] = -INFINITY; li[j] = 0xffffffffu; }
    const long long total = (long long)p.nlists * KSEL, base = (long long)q * total;
    float theta = -INFINITY;            // upper bound of the approximate score of every row dropped so far
    for (long long c = tid; c < total; c += 256) {
        const float s = p.cand_s[base + c]; const uint32_t i = p.cand_i[base + c];
        if ((int)(c % KSEL) == KSEL - 1 && i != 0xffffffffu) theta = fmaxf(theta, s);   // tail of a FULL lane list bounds what that lane dropped
        if (!better(s, i, ls[KSEL - 1], li[KSEL - 1])) { if (i != 0xffffffffu) theta = fmaxf(theta, s); continue; }
        if (li[KSEL - 1] != 0xffffffffu) theta = fmaxf(theta, ls[KSEL - 1]);             // the entry this insertion evicts
        list_insert<KSEL>(ls, li, s, i);
    }
#pragma unroll
    for (int j = 0; j < KSEL; j++) { ss[tid * KSEL + j] = ls[j]; si[tid * KSEL + j] = li[j]; }
    __shared__ float s_theta[256];
    s_theta[tid] = theta;
    __syncthreads();
    // bitonic sort, best first
    for (int size = 2; size <= NS; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < NS / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = ((lo & size) == 0);       // ascending block => best first
                const float a = ss[lo], b = ss[hi]; const uint32_t ai = si[lo], bi = si[hi];
                const bool b_better = better(b, bi, a, ai);
                if (b_better == up) { ss[lo] = b; si[lo] = bi; ss[hi] = a; si[hi] = ai; }
            }
            __syncthreads();
        }
    }
    // exact re-score of the best R candidates: fp64 accumulation of exact products
    __shared__ double ex[R]; __shared__ uint32_t exi[R];
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < R; r += 4) {
        const uint32_t id = si[r];
        const bool valid = id != 0xffffffffu && (long long)id < p.n;
        const double acc = valid ? exact_score(p.qn + (long long)q * p.dim, p.dbn + (long long)id * p.dim, p.dim, lane) : 0.0;
        if (lane == 0) { ex[r] = valid ? acc : -INFINITY; exi[r] = id; }
    }
    __syncthreads();
    if (tid == 0) {
        for (int a = 1; a < R; a++) {       // insertion sort by (score desc, index asc)
            const double s = ex[a]; const uint32_t id = exi[a]; int b = a - 1;
            while (b >= 0 && (ex[b] < s || (ex[b] == s && exi[b] > id))) { ex[b + 1] = ex[b]; exi[b + 1] = exi[b]; b--; }
            ex[b + 1] = s; exi[b + 1] = id;
        }
        for (int j = 0; j < p.k; j++) {
            p.idx_out[(long long)(p.qbase + q) * p.k + j] = exi[j];
            if (p.score_out) p.score_out[(long long)(p.qbase + q) * p.k + j] = (float)ex[j];
            if (p.score64_out) p.score64_out[(long long)(p.qbase + q) * p.k + j] = ex[j];
        }
        // certificate: nothing that was dropped can reach the k-th exact score
        float th = (si[R] != 0xffffffffu) ? ss[R] : -INFINITY;          // first candidate after the re-scored prefix
        for (int t = 0; t < 256; t++) th = fmaxf(th, s_theta[t]);
        const double tau = ex[p.k - 1];
        const bool certified = (double)th + (double)p.eps < tau;
        p.cert.flag[q] = certified ? 0 : 1;
        p.cert.tau[q] = tau; p.cert.tau_idx[q] = exi[p.k - 1]; p.cert.fb_count[q] = 0;
        if (!certified) atomicOr(p.cert.status, 1);
    }
}

// ---------------------------------------------------------------- exact fallback (flagged queries only; normally none)
// One wave per row: exact fp64 score against every flagged query; rows that beat-or-tie the query's current k-th candidate
// under (score desc, index asc) are appended.  The true top k all do (the current k-th is the k-th best of a SUBSET).
__global__ __launch_bounds__(256) void knn_exact_collect_kernel(Certify c, const _Float16* dbn, const float* qn, long long n, int dim, int nq) {
    __shared__ int fq[KNN_QMAX]; __shared__ int nf;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) { int m = 0; for (int q = 0; q < nq; q++) if (c.flag[q]) fq[m++] = q; nf = m; }
    __syncthreads();
    if (nf == 0) return;
    const long long wave0 = (long long)blockIdx.x * 4 + (tid >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long row = wave0; row < n; row += nwaves) {
        for (int f = 0; f < nf; f++) {
            const int q = fq[f];
            const double s = exact_score(qn + (long long)q * dim, dbn + row * dim, dim, lane);
            if (lane == 0) {
                const double tau = c.tau[q]; const uint32_t ti = c.tau_idx[q];
                if (s > tau || (s == tau && (uint32_t)row <= ti)) {
                    const int slot = atomicAdd(&c.fb_count[q], 1);
                    if (slot < KNN_FB_CAP) { c.fb_s[(long long)q * KNN_FB_CAP + slot] = s; c.fb_i[(long long)q * KNN_FB_CAP + slot] = (uint32_t)row; }
                }
            }
        }
    }
}
// top k of the collected rows by k rounds of block-wide arg-best (k <= 28, <= KNN_FB_CAP entries)
__global__ __launch_bounds__(256) void knn_exact_finish_kernel(Certify c, int k, uint32_t* idx_out, float* score_out, double* score64_out, int qbase) {
    const int q = blockIdx.x, tid = threadIdx.x;
    if (!c.flag[q]) return;
    int cnt = c.fb_count[q];
    if (cnt > KNN_FB_CAP) { if (tid == 0) atomicOr(c.status, 2); cnt = KNN_FB_CAP; }
    double* fs = c.fb_s + (long long)q * KNN_FB_CAP; uint32_t* fi = c.fb_i + (long long)q * KNN_FB_CAP;
    __shared__ double bs[256]; __shared__ uint32_t bi[256]; __shared__ int bp[256];
    for (int j = 0; j < k; j++) {
        double s = -INFINITY; uint32_t id = 0xffffffffu; int pos = -1;
        for (int e = tid; e < cnt; e += 256) {
            const double es = fs[e]; const uint32_t ei = fi[e];
            if (ei != 0xffffffffu && (es > s || (es == s && ei < id))) { s = es; id = ei; pos = e; }
        }
        bs[tid] = s; bi[tid] = id; bp[tid] = pos;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o && (bs[tid + o] > bs[tid] || (bs[tid + o] == bs[tid] && bi[tid + o] < bi[tid]))) { bs[tid] = bs[tid + o]; bi[tid] = bi[tid + o]; bp[tid] = bp[tid + o]; }
            __syncthreads();
        }
        if (tid == 0 && bp[0] >= 0) {
            idx_out[(long long)(qbase + q) * k + j] = bi[0];
            if (score_out) score_out[(long long)(qbase + q) * k + j] = (float)bs[0];
            if (score64_out) score64_out[(long long)(qbase + q) * k + j] = bs[0];
            fi[bp[0]] = 0xffffffffu;                      // taken
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- gather
template <typename TIN>
__global__ void knn_gather_kernel(const TIN* raw, const uint32_t* idx, long long n_idx, int dim, float* out) {
    const long long total = n_idx * dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / dim; const int c = (int)(i - r * dim);
        out[i] = (float)raw[(long long)idx[r] * dim + c];
    }
}

// ---------------------------------------------------------------- host
static const char* hiperr(hipError_t e) { return e == hipSuccess ? nullptr : hipGetErrorString(e); }
#define KNN_TRY(x) do { const char* _m = hiperr(x); if (_m) return _m; } while (0)

void knn_free(KnnDb& db) {
    if (db.dbn) hipFree(db.dbn);
    if (db.raw) hipFree(db.raw);
    if (db.scratch) hipFree(db.scratch);
    if (db.zero_page) hipFree(db.zero_page);
    db = KnnDb();
}

static const char* knn_load_impl(KnnDb& db, const void* emb, long long n, int dim, int dtype, int is_device, hipStream_t st) {
    if (n <= 0 || dim <= 0 || dim % 64 != 0 || dim > 4096) return "dim must be a positive multiple of 64";
    if (n >= 0xffffffffLL) return "database too large for uint32 indices";
    if (dtype != 0 && dtype != 1) return "dtype must be 0 (fp16) or 1 (fp32)";
    knn_free(db);
    const size_t es = dtype == 0 ? 2 : 4;
    const long long n_pad = (n + KNN_ROWS - 1) / KNN_ROWS * KNN_ROWS;
    KNN_TRY(hipMalloc(&db.dbn, (size_t)n_pad * dim * 2));
    KNN_TRY(hipMemsetAsync(db.dbn, 0, (size_t)n_pad * dim * 2, st));
    KNN_TRY(hipMalloc(&db.raw, (size_t)n * dim * es));
    // chunked upload keeps pinned/pageable staging bounded
    const long long chunk = 1 << 20;
    for (long long r0 = 0; r0 < n; r0 += chunk) {
        const long long rows = (n - r0 < chunk) ? n - r0 : chunk;
        char* dst = (char*)db.raw + (size_t)r0 * dim * es;
        KNN_TRY(hipMemcpyAsync(dst, (const char*)emb + (size_t)r0 * dim * es, (size_t)rows * dim * es,
                               is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        const int grid = (int)((rows + 3) / 4);
        _Float16* out = (_Float16*)db.dbn + (size_t)r0 * dim;
        if (dtype == 0) knn_normalize_rows_kernel<_Float16><<<grid, 256, 0, st>>>((const _Float16*)dst, out, rows, dim);
        else knn_normalize_rows_kernel<float><<<grid, 256, 0, st>>>((const float*)dst, out, rows, dim);
        KNN_TRY(hipGetLastError());
    }
    KNN_TRY(hipStreamSynchronize(st));
    db.n = n; db.dim = dim; db.raw_dtype = dtype;
    return nullptr;
}

const char* knn_load(KnnDb& db, const void* emb, long long n, int dim, int dtype, int is_device, hipStream_t st) {
    const char* msg = knn_load_impl(db, emb, n, dim, dtype, is_device, st);
    if (msg) knn_free(db);       // never leave a half-loaded database behind (a failed second hipMalloc used to keep db.dbn with n = 0)
    return msg;
}

template <int KSEL, int R>
static const char* search_impl(KnnDb& db, const float* q, int b, int k, uint32_t* idx_out, float* score_out, double* score64_out, hipStream_t st) {
    int dev = 0, ncu = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const long long ntiles = (db.n + KNN_ROWS - 1) / KNN_ROWS;
    int grid = (int)(ntiles < ncu ? ntiles : ncu);
    const bool d512 = db.dim == 512;                 // register-resident queries (4 lists per block and query instead of 8)
    const int nlists = grid * (d512 ? 4 : 8);
    const int nlists_cap = grid * 8;
    const bool bulk = b >= 128 && db.dim % KNN_BK == 0 && ntiles >= 16;       // dataset-scale query batches: 128-query walkers, up to 4 groups per database pass
    const int QS = bulk ? KNN_QMAX : KNN_Q;                    // queries the scratch is sized for
    const size_t qn_b = (size_t)QS * db.dim * 4, qh_b = (size_t)QS * db.dim * 2;
    const size_t cand_b = (size_t)QS * nlists_cap * KSEL * 4;
    const size_t cert_off = (qn_b + 2 * qh_b + 2 * cand_b + 255) & ~(size_t)255;
    const size_t cert_b = 256 /*status*/ + KNN_QMAX * (4 + 8 + 4 + 4) + 256 + (size_t)KNN_QMAX * KNN_FB_CAP * 12;
    const size_t need = cert_off + cert_b + 1024;
    if (db.scratch_bytes < need) {
        if (db.scratch) { KNN_TRY(hipStreamSynchronize(st)); KNN_TRY(hipFree(db.scratch)); db.scratch = nullptr; db.scratch_bytes = 0; }
        KNN_TRY(hipMalloc(&db.scratch, need)); db.scratch_bytes = need;
    }
    char* s = (char*)db.scratch;
    float* qn = (float*)s; _Float16* qh = (_Float16*)(s + qn_b); _Float16* ql = (_Float16*)(s + qn_b + qh_b);
    float* cs = (float*)(s + qn_b + 2 * qh_b); uint32_t* ci = (uint32_t*)(s + qn_b + 2 * qh_b + cand_b);
    Certify cert{};
    {
        char* cb = s + cert_off;
        cert.status = (int*)cb; cb += 256;
        cert.tau = (double*)cb; cb += KNN_QMAX * 8;
        cert.fb_s = (double*)cb; cb += (size_t)KNN_QMAX * KNN_FB_CAP * 8;
        cert.fb_i = (uint32_t*)cb; cb += (size_t)KNN_QMAX * KNN_FB_CAP * 4;
        cert.tau_idx = (uint32_t*)cb; cb += KNN_QMAX * 4;
        cert.flag = (int*)cb; cb += KNN_QMAX * 4;
        cert.fb_count = (int*)cb;
    }
    KNN_TRY(hipMemsetAsync(cert.status, 0, 4, st));
    if (!db.zero_page) { KNN_TRY(hipMalloc(&db.zero_page, 256)); KNN_TRY(hipMemset(db.zero_page, 0, 256)); }
    void* zero_page = db.zero_page;
    constexpr int scan_smem = 2 * (KNN_ROWS * 128 + 2 * KNN_Q * 128);
    constexpr int scan512_smem = 3 * KNN_ROWS * 128;
    constexpr int merge_smem = 256 * KSEL * 8;
    static bool attr_dev[RDM_MAX_DEVICES] = {false};
    bool& attr = attr_dev[rdm_cur_device()];
    if (!attr) {
        KNN_TRY(hipFuncSetAttribute((const void*)knn_scan_kernel<KSEL>, hipFuncAttributeMaxDynamicSharedMemorySize, scan_smem));
        KNN_TRY(hipFuncSetAttribute((const void*)knn_scan512_kernel<KSEL>, hipFuncAttributeMaxDynamicSharedMemorySize, scan512_smem));
        KNN_TRY(hipFuncSetAttribute((const void*)knn_merge_kernel<KSEL, R>, hipFuncAttributeMaxDynamicSharedMemorySize, merge_smem));
        attr = true;
    }
    if (bulk) {
        constexpr int bulk_smem = 3 * (KNN_ROWS * 128 + 128 * 128);
        static bool battr_dev[RDM_MAX_DEVICES] = {false};
        bool& battr = battr_dev[rdm_cur_device()];
        if (!battr) { KNN_TRY(hipFuncSetAttribute((const void*)knn_scan_bulk_kernel<KSEL>, hipFuncAttributeMaxDynamicSharedMemorySize, bulk_smem)); battr = true; }
        const int bgrid = (ncu / 32) * 32;                                  // a multiple of 8 XCDs x 4 groups
        for (int q0 = 0; q0 < b; q0 += KNN_QMAX) {
            const int bq = (b - q0 < KNN_QMAX) ? b - q0 : KNN_QMAX;
            const int groups = (bq + 127) / 128;                        // 1 .. 4 query groups share every database pass
            knn_prep_queries_kernel<<<groups * 128, 64, 0, st>>>(q + (size_t)q0 * db.dim, bq, db.dim, qn, qh, ql);
            KNN_TRY(hipGetLastError());
            BulkParams bp{}; bp.groups = groups;
            // scan-only ablation (tools/bulk_pmc.sh): compiled in only with -DRDM_DEBUG_ABLATION -- a leaked environment variable must
            // never make the production search return success with uninitialised neighbours
#ifdef RDM_DEBUG_ABLATION
            static const int bdbg = getenv("RDM_KNN_BULK_DBG") ? atoi(getenv("RDM_KNN_BULK_DBG")) : 0;
#else
            constexpr int bdbg = 0;
#endif
            bp.dbg = bdbg;
            bp.s.dbn = (const _Float16*)db.dbn; bp.s.n = db.n; bp.s.dim = db.dim; bp.s.ntiles = ntiles; bp.s.qh = qh; bp.s.ql = ql;
            bp.s.cand_s = cs; bp.s.cand_i = ci; bp.s.zero_page = zero_page;
            int walkers = bgrid / groups; if ((long long)walkers > ntiles) walkers = (int)ntiles;
            walkers = (walkers / 8) * 8;                                    // >= 8 (ntiles >= 16): every walker owns a tile and writes its lists
            const int g2 = walkers * groups;
            bp.s.nlists = walkers * 4;
            knn_scan_bulk_kernel<KSEL><<<g2, 512, bulk_smem, st>>>(bp);
            KNN_TRY(hipGetLastError());
            MergeParams mp{}; mp.cand_s = cs; mp.cand_i = ci; mp.nlists = bp.s.nlists; mp.dbn = (const _Float16*)db.dbn; mp.qn = qn;
            mp.dim = db.dim; mp.n = db.n; mp.k = k; mp.idx_out = idx_out; mp.score_out = score_out; mp.score64_out = score64_out; mp.qbase = q0; mp.cert = cert; mp.eps = KNN_EPS_BULK;
            if (bdbg) continue;                                            // ablation timing of the scan alone: results are garbage
            knn_merge_kernel<KSEL, R><<<bq, 256, merge_smem, st>>>(mp);
            KNN_TRY(hipGetLastError());
            knn_exact_collect_kernel<<<ncu * 4, 256, 0, st>>>(cert, (const _Float16*)db.dbn, qn, db.n, db.dim, bq);
            KNN_TRY(hipGetLastError());
            knn_exact_finish_kernel<<<bq, 256, 0, st>>>(cert, k, idx_out, score_out, score64_out, q0);
            KNN_TRY(hipGetLastError());
        }
    } else
    for (int q0 = 0; q0 < b; q0 += KNN_Q) {
        const int bq = (b - q0 < KNN_Q) ? b - q0 : KNN_Q;
        knn_prep_queries_kernel<<<KNN_Q, 64, 0, st>>>(q + (size_t)q0 * db.dim, bq, db.dim, qn, qh, ql);
        KNN_TRY(hipGetLastError());
        ScanParams sp{}; sp.dbn = (const _Float16*)db.dbn; sp.n = db.n; sp.dim = db.dim; sp.ntiles = ntiles; sp.qh = qh; sp.ql = ql;
        sp.cand_s = cs; sp.cand_i = ci; sp.nlists = nlists; sp.zero_page = zero_page;
        static const int w8 = getenv("RDM_KNN_W8") ? atoi(getenv("RDM_KNN_W8")) : 1;
        static const int hi_only = getenv("RDM_KNN_HI_ONLY") ? atoi(getenv("RDM_KNN_HI_ONLY")) : 1;
        sp.hi_only = (d512 && KSEL == 8 && w8) ? hi_only : 0;
        if (d512 && KSEL == 8 && w8) {
            static bool attr8_dev[RDM_MAX_DEVICES] = {false};
            bool& attr8 = attr8_dev[rdm_cur_device()];
            if (!attr8) {
                KNN_TRY(hipFuncSetAttribute((const void*)knn_scan512w8_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, scan512_smem));
                KNN_TRY(hipFuncSetAttribute((const void*)knn_scan512w8_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, scan512_smem));
                attr8 = true;
            }
            sp.nlists = grid * 8;
            if (sp.hi_only) knn_scan512w8_kernel<8, true><<<grid, 512, scan512_smem, st>>>(sp);
            else knn_scan512w8_kernel<8, false><<<grid, 512, scan512_smem, st>>>(sp);
        } else if (d512) knn_scan512_kernel<KSEL><<<grid, 256, scan512_smem, st>>>(sp);
        else knn_scan_kernel<KSEL><<<grid, 256, scan_smem, st>>>(sp);
        KNN_TRY(hipGetLastError());
        MergeParams mp{}; mp.cand_s = cs; mp.cand_i = ci; mp.nlists = sp.nlists; mp.dbn = (const _Float16*)db.dbn; mp.qn = qn;
        mp.dim = db.dim; mp.n = db.n; mp.k = k; mp.idx_out = idx_out; mp.score_out = score_out; mp.score64_out = score64_out; mp.qbase = q0; mp.cert = cert; mp.eps = sp.hi_only ? KNN_EPS_BULK : KNN_EPS;
        knn_merge_kernel<KSEL, R><<<bq, 256, merge_smem, st>>>(mp);
        KNN_TRY(hipGetLastError());
        // exact fallback for the flagged queries of this group (both kernels return at once when nothing is flagged)
        knn_exact_collect_kernel<<<ncu * 4, 256, 0, st>>>(cert, (const _Float16*)db.dbn, qn, db.n, db.dim, bq);
        KNN_TRY(hipGetLastError());
        knn_exact_finish_kernel<<<bq, 256, 0, st>>>(cert, k, idx_out, score_out, score64_out, q0);
        KNN_TRY(hipGetLastError());
    }
    int status = 0;
    KNN_TRY(hipMemcpyAsync(&status, cert.status, 4, hipMemcpyDeviceToHost, st));
    KNN_TRY(hipStreamSynchronize(st));
    db.last_fallbacks = status & 1;
    if (status & 2) return "exact fallback overflow: more than 8192 rows tie with the k-th neighbour within the score error bound";
    return nullptr;
}

const char* knn_search(KnnDb& db, const float* q, int b, int k, uint32_t* idx_out, float* score_out, double* score64_out, hipStream_t st) {
    if (!db.dbn) return "no database loaded (rdm_db_load)";
    if (b < 1 || k < 1) return "b and k must be positive";
    if (k > db.n) return "k exceeds database size";
    // List length per lane: 8 for every k; the merge re-scores the best 16 (k <= 4) or 64 (k <= 28) of the merged candidates.  The
    // length is a SPEED choice, not a correctness margin: whatever a list drops is covered by the certificate (and the exact
    // fallback) in the merge.  A lane sees 1/2048 of the rows (1/256 in the bulk scan), so a list overflows only when more than 8
    // of a query's best ~64 rows fall into one lane's rows.  (The insertion chain runs for a candidate when ANY of the wave's 64
    // lanes accepts it, so its cost grows with the list length times the acceptance rate: 16-entry lists made k = 16 take 8.8 ms
    // per 64 queries against 4.4 ms for k = 4, 32-entry lists 8x and the 8-wave bulk kernel spilled.)
    if (k <= 4) return search_impl<8, 16>(db, q, b, k, idx_out, score_out, score64_out, st);
    if (k <= 28) return search_impl<8, 64>(db, q, b, k, idx_out, score_out, score64_out, st);
    return "k > 28 is not supported (the merge re-scores the best 64 candidates)";
}

const char* knn_gather(KnnDb& db, const uint32_t* idx, long long n_idx, float* out, hipStream_t st) {
    if (!db.raw) return "no database loaded (rdm_db_load)";
    const long long total = n_idx * db.dim;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    if (db.raw_dtype == 0) knn_gather_kernel<_Float16><<<grid, 256, 0, st>>>((const _Float16*)db.raw, idx, n_idx, db.dim, out);
    else knn_gather_kernel<float><<<grid, 256, 0, st>>>((const float*)db.raw, idx, n_idx, db.dim, out);
    return hiperr(hipGetLastError());
}

// Micro-benchmark (round 6, verdict item 1b): the K loop of a FUSED Winograd F(2x2, 3x3) conv on gfx950, with every operand already in LDS
// -- i.e. with the L2 -> CU ingest (which bounds the direct conv's loop) taken to be FREE.  It answers one question: what share of the
// matrix pipe can the loop hold when the 16 transformed input fragments V = B^T d B of each k-step have to be formed by the VALU (gfx950 has no
// packed bf16 arithmetic: v_pk_add_bf16 does not assemble for this target -- every bf16 is unpacked to fp32, added, re-packed)?
//
// One wave per SIMD, 4 waves per block (2 tile groups x 2 channel groups), a wave owns 16 Winograd positions x [32 tiles x 32 output channels]
// = 16 accumulators of 32x32 fp32 = 256 registers (the whole accumulator budget of a 512-register wave: the tile cannot be bigger).
// Per k-step (16 input channels) and wave: 16 ds_read_b128 of raw halo pixels (lane = tile x k-half: its 4x4 input window, 8 channels), the input
// transform, 16 ds_read_b128 of U fragments (G g G^T, fragment-ordered), 16 v_mfma_f32_32x32x16_bf16.
//   mode 0: the whole loop;  mode 1: no transform (raw pixels as A fragments: LDS reads + MFMA only);  mode 2: transform, no MFMA;
//   mode 3: MFMA only (operands loaded once).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/wino_loop.hip -o tools/ubench/wino_loop (the binary travels to the GPU box with the snapshot).
// Prints cycles per k-step and the DIRECT-CONV-EQUIVALENT rate: 2.25 x (MFMA FLOPs executed) / time, the number to put beside
// conv3x3_halo4_kernel's 1.25 PFLOP/s in isolation (0.50 of peak) / 1.05-1.10 in situ.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

constexpr int HW = 34, HROWS = 10, PXB = 80;           // halo of 8 x 32 output pixels, 32 channels per pixel (64 B) padded to 80 B: conflict-free b128 reads
constexpr int HALO_BYTES = HROWS * HW * PXB;           // 27 200
constexpr int U_BYTES = 2 * 2 * 16 * 1024;             // [k-step parity][channel group][position][1 KiB fragment]

template <int MODE>
__global__ __launch_bounds__(256, 1) void wino_loop_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int ksteps, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tg = wave >> 1, ng = wave & 1;
    for (int i = tid; i < (HALO_BYTES + U_BYTES) / 16; i += 256) ((uint4*)smem)[i] = src[i];
    __syncthreads();
    const int tile = lane & 31, kh = lane >> 5;
    const int ty = tg * 2 + (tile >> 4), tx = tile & 15;                    // tile (ty, tx): output rows 2 ty .. 2 ty + 1, halo rows 2 ty .. 2 ty + 3
    const char* const hbase = smem + ((2 * ty) * HW + 2 * tx) * PXB + kh * 16;
    const char* const ubase = smem + HALO_BYTES + ng * (16 * 1024) + lane * 16;
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[p][e] = 0.f;
    bf16x8 a0[16], b0[16];
    if (MODE == 3) {
#pragma unroll
        for (int p = 0; p < 16; p++) { a0[p] = *(const bf16x8*)(hbase + ((p >> 2) * HW + (p & 3)) * PXB); b0[p] = *(const bf16x8*)(ubase + p * 1024); }
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int ks = 0; ks < ksteps; ks++) {
        const int par = ks & 1;
        if (MODE == 3) {
#pragma unroll
            for (int p = 0; p < 16; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[p], b0[p], acc[p], 0, 0, 0);
            continue;
        }
        // raw 4 x 4 window, 8 channels (16 B) per pixel: channels par * 16 + kh * 8 ..
        u32x4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) d[i][j] = *(const u32x4*)(hbase + (i * HW + j) * PXB + par * 32);
        const char* const up = ubase + par * (2 * 16 * 1024);
#pragma unroll
        for (int xi = 0; xi < 4; xi++) {                                     // row xi of B^T d: rows (0) d0 - d2, (1) d1 + d2, (2) d2 - d1, (3) d1 - d3
            bf16x8 v[4];
            if (MODE == 1) {
#pragma unroll
                for (int nu = 0; nu < 4; nu++) { union { u32x4 u; bf16x8 f; } t; t.u = d[xi][nu]; v[nu] = t.f; }
            } else {
                unsigned vw[4][4];
#pragma unroll
                for (int w = 0; w < 4; w++) {                                // dword w = channels 2 w, 2 w + 1
                    float r[4][2];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const unsigned pa = xi == 0 ? d[0][j][w] : xi == 3 ? d[1][j][w] : d[1][j][w];
                        const unsigned pb = xi == 0 ? d[2][j][w] : xi == 3 ? d[3][j][w] : d[2][j][w];
                        const float alo = __uint_as_float(pa << 16), ahi = __uint_as_float(pa & 0xffff0000u);
                        const float blo = __uint_as_float(pb << 16), bhi = __uint_as_float(pb & 0xffff0000u);
                        if (xi == 1) { r[j][0] = alo + blo; r[j][1] = ahi + bhi; }
                        else if (xi == 2) { r[j][0] = blo - alo; r[j][1] = bhi - ahi; }
                        else { r[j][0] = alo - blo; r[j][1] = ahi - bhi; }
                    }
                    vw[0][w] = cvt_pk_bf16(r[0][0] - r[2][0], r[0][1] - r[2][1]);
                    vw[1][w] = cvt_pk_bf16(r[1][0] + r[2][0], r[1][1] + r[2][1]);
                    vw[2][w] = cvt_pk_bf16(r[2][0] - r[1][0], r[2][1] - r[1][1]);
                    vw[3][w] = cvt_pk_bf16(r[1][0] - r[3][0], r[1][1] - r[3][1]);
                }
#pragma unroll
                for (int nu = 0; nu < 4; nu++) { union { u32x4 u; bf16x8 f; } t; t.u = (u32x4){vw[nu][0], vw[nu][1], vw[nu][2], vw[nu][3]}; v[nu] = t.f; }
            }
#pragma unroll
            for (int nu = 0; nu < 4; nu++) {
                const int p = xi * 4 + nu;
                const bf16x8 u = *(const bf16x8*)(up + p * 1024);
                if (MODE == 2) { acc[p][0] += (float)v[nu][0] + (float)u[0]; }
                else acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[nu], u, acc[p], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 16; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) s += acc[p][e];
    if (s == 12345.678f) sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const uint4* src, float* sink, unsigned long long* cyc, int ncu, int ksteps) {
    const int sm = HALO_BYTES + U_BYTES;
    CK(hipFuncSetAttribute((const void*)wino_loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, sm));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    wino_loop_kernel<MODE><<<ncu, 256, sm>>>(src, sink, 200, cyc);
    CK(hipDeviceSynchronize());
    float best = 1e30f; std::vector<unsigned long long> h(ncu); double cyc_avg = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        wino_loop_kernel<MODE><<<ncu, 256, sm>>>(src, sink, ksteps, cyc);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) {
            best = ms;
            CK(hipMemcpy(h.data(), cyc, ncu * 8, hipMemcpyDeviceToHost));
            cyc_avg = 0; for (auto v : h) cyc_avg += (double)v; cyc_avg /= ncu;
        }
    }
    const double mfma_flop = (double)ncu * 4 * ksteps * 16 * 32768.0;
    const char* names[] = {"full loop (LDS reads + input transform + MFMA)", "no transform (LDS reads + MFMA)", "transform, no MFMA", "MFMA only"};
    printf("mode %d %-48s: %8.3f ms, %7.0f shader cycles per k-step (16 MFMAs = 512 matrix-pipe cycles) -> MFMA busy %.3f, executed %.0f TFLOP/s, direct-conv-equivalent %.0f TFLOP/s\n",
           MODE, names[MODE], best, cyc_avg / ksteps, MODE == 2 ? 0.0 : 512.0 * ksteps / cyc_avg, MODE == 2 ? 0.0 : mfma_flop / (best * 1e-3) / 1e12,
           MODE == 2 ? 0.0 : 2.25 * mfma_flop / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int ksteps = argc > 1 ? atoi(argv[1]) : 20000;
    const size_t nbytes = HALO_BYTES + U_BYTES;
    std::vector<uint16_t> h(nbytes / 2);
    uint32_t s = 12345u;
    for (auto& v : h) {            // random bf16 in [-1, 1): full operand toggling (the power cap is part of the answer)
        s = s * 1664525u + 1013904223u;
        const float f = ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f;
        uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16);
    }
    uint4* src; CK(hipMalloc(&src, nbytes)); CK(hipMemcpy(src, h.data(), nbytes, hipMemcpyHostToDevice));
    float* sink; CK(hipMalloc(&sink, (size_t)ncu * 256 * 4));
    unsigned long long* cyc; CK(hipMalloc(&cyc, ncu * 8));
    printf("wino_loop: %d CUs, %d k-steps per block, one block (4 waves, one per SIMD) per CU\n", ncu, ksteps);
    run<3>(src, sink, cyc, ncu, ksteps);
    run<1>(src, sink, cyc, ncu, ksteps);
    run<2>(src, sink, cyc, ncu, ksteps);
    run<0>(src, sink, cyc, ncu, ksteps);
    return 0;
}

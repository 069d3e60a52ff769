// Box calibration probes (bench.py's "calibration" object, round 6): two FIXED instruction streams that never change with the product
// kernels, so that a headline measured on one box of the pool can be put beside one measured on another (boxes differ by +-4..5 %
// in sustained clock under the package power cap):
//   * calib_mfma_kernel: one wave per SIMD, every CU, back-to-back v_mfma_f32_32x32x16_bf16 on eight independent accumulators with
//     RANDOM bf16 operands held in registers (random operands draw the matrix pipe's full switching power: the same binary on zeros
//     clocks ~20 % higher) -- sustained dense-bf16 TFLOP/s of THIS box under its power cap;
//   * calib_stream_kernel: a 16-byte-per-lane grid-stride copy -- HBM read + write GB/s.
// Neither touches any product state; both run on the context's stream and are timed there with HIP events.
#include "kernels.h"

__global__ __launch_bounds__(256, 1) void calib_mfma_kernel(const bf16_t* __restrict__ src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = *(const bf16x8*)(src + ((size_t)((wave * 8 + i) * 64 + lane)) * 8);
        b[i] = *(const bf16x8*)(src + ((size_t)((wave * 8 + 4 + i) * 64 + lane)) * 8);
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) s += acc[i][e];
    if (s == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = s;      // keeps the accumulators live; practically never taken
}
__global__ __launch_bounds__(256) void calib_fill_kernel(bf16_t* dst, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // uniform in [-0.25, 0.25): every mantissa bit toggles, sums over ~1e6 MFMAs stay far inside fp32 range
        dst[i] = f2bf(((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.5f);
    }
}
__global__ __launch_bounds__(256) void calib_stream_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// mfma_ms: wall time the MFMA probe should run for (after a short sizing pass); stream_bytes: bytes READ by one copy pass (as many are
// written); buf: device scratch of at least max(2 * stream_bytes, 64 KiB).  Results: dense bf16 TFLOP/s and GB/s (read + write).
hipError_t run_calib_probes(void* buf, double mfma_ms, size_t stream_bytes, int stream_reps, double* mfma_tflops, double* stream_gbps, hipStream_t st) {
    int dev = rdm_cur_device(), ncu = 0;
    hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    hipEvent_t ea, eb;
    if ((e = hipEventCreate(&ea)) != hipSuccess) return e;
    if ((e = hipEventCreate(&eb)) != hipSuccess) { (void)hipEventDestroy(ea); return e; }
    auto done = [&](hipError_t r) { (void)hipEventDestroy(ea); (void)hipEventDestroy(eb); return r; };
    bf16_t* src = (bf16_t*)buf; float* sink = (float*)((char*)buf + 32768);
    calib_fill_kernel<<<16, 256, 0, st>>>(src, 4 * 8 * 64 * 8, 0x1234567u);
    auto timed = [&](int iters, float* ms) -> hipError_t {
        hipError_t r;
        if ((r = hipEventRecord(ea, st)) != hipSuccess) return r;
        calib_mfma_kernel<<<ncu, 256, 0, st>>>(src, sink, iters);
        if ((r = hipGetLastError()) != hipSuccess) return r;
        if ((r = hipEventRecord(eb, st)) != hipSuccess) return r;
        if ((r = hipEventSynchronize(eb)) != hipSuccess) return r;
        return hipEventElapsedTime(ms, ea, eb);
    };
    if (mfma_tflops) {
        float ms = 0.f;
        if ((e = timed(2000, &ms)) != hipSuccess) return done(e);           // warm-up + sizing (2000 iterations = 64 k MFMAs per wave, ~1 ms)
        if ((e = timed(20000, &ms)) != hipSuccess) return done(e);
        double per_iter = ms / 20000.0;
        long long iters = (long long)(mfma_ms / (per_iter > 0 ? per_iter : 1e-3));
        if (iters < 20000) iters = 20000;
        if (iters > 2000000000LL) iters = 2000000000LL;
        if ((e = timed((int)iters, &ms)) != hipSuccess) return done(e);
        *mfma_tflops = (double)ncu * 4.0 * (double)iters * 32.0 * 32768.0 / (ms * 1e-3) / 1e12;
    }
    if (stream_gbps && stream_bytes) {
        const size_t nvec = stream_bytes / 16;
        uint4* s4 = (uint4*)buf; uint4* d4 = (uint4*)((char*)buf + nvec * 16);
        calib_fill_kernel<<<4096, 256, 0, st>>>((bf16_t*)buf, nvec * 8, 0x9e3779b9u);
        calib_stream_kernel<<<ncu * 8, 256, 0, st>>>(s4, d4, nvec);           // warm-up
        if ((e = hipEventRecord(ea, st)) != hipSuccess) return done(e);
        for (int r = 0; r < stream_reps; r++) calib_stream_kernel<<<ncu * 8, 256, 0, st>>>(s4, d4, nvec);
        if ((e = hipGetLastError()) != hipSuccess) return done(e);
        if ((e = hipEventRecord(eb, st)) != hipSuccess) return done(e);
        if ((e = hipEventSynchronize(eb)) != hipSuccess) return done(e);
        float ms = 0.f;
        if ((e = hipEventElapsedTime(&ms, ea, eb)) != hipSuccess) return done(e);
        *stream_gbps = 2.0 * (double)(nvec * 16) * stream_reps / (ms * 1e-3) / 1e9;
    }
    return done(hipSuccess);
}

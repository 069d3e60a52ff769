// Micro-benchmark: L2 -> LDS staging bandwidth per CU on gfx950 (dev tool, not part of the library).
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), 1: global_load_dwordx4 -> registers (discarded), 2: global_load_dwordx4 -> ds_write_b128
// Every block streams `rows` x 128 B rows with a row stride of `ld` bytes (weights-like: all blocks read the SAME rows
// when shared=1, activation-like: distinct rows per block when shared=0), 1 KiB per wave instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void stream_kernel(const char* src, long long ld, int rows, int iters, int shared, long long blk_stride,
                                                       unsigned long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = tid >> 3, chunk = tid & 7;                 // 64 rows x 8 chunks of 16 B per pass
    const char* base = src + (shared ? 0 : blockIdx.x * blk_stride);
    const int passes = rows / 64;
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        const char* b = base + (long long)(it & 7) * rows * ld;  // 8 different tiles in rotation (stay L2 resident)
        for (int ps = 0; ps < passes; ps++) {
            const char* g = b + (long long)(ps * 64 + lrow) * ld + chunk * 16;
            char* l = smem + ((ps & 3) * 64 + wave * 8) * 128;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
            } else if (MODE == 1) {
                uint4 v = *(const uint4*)g;
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            } else {
                uint4 v = *(const uint4*)g;
                *(uint4*)(l + lane * 16) = v;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (MODE == 1 && acc.x == 0x12345678u) sink[tid] = (float)acc.y;
    if (MODE != 1 && smem[tid * 4] == 77 && iters < 0) sink[tid] = 1.f;
}

int main(int argc, char** argv) {
    int dev = 0; CK(hipSetDevice(dev));
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const long long ld = argc > 1 ? atoll(argv[1]) : 6912;   // weight row stride (6912 = 384-channel 3x3 conv, K = 3456 bf16; 128 = packed tile)
    const int rows = 192, iters = 2000;
    const size_t tile = (size_t)rows * ld;     // 1.3 MB per tile, 8 tiles in rotation
    const size_t bytes = (size_t)ncu * 8 * tile + (1 << 20);
    char* src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
    unsigned long long* cyc; CK(hipMalloc(&cyc, ncu * 8)); float* sink; CK(hipMalloc(&sink, 4096));
    std::vector<unsigned long long> h(ncu);
    for (int shared = 1; shared >= 0; shared--)
        for (int mode = 0; mode < 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                const long long bs = 8 * (long long)tile;
                if (mode == 0) stream_kernel<0><<<ncu, 512, 32768>>>(src, ld, rows, iters, shared, bs, cyc, sink);
                if (mode == 1) stream_kernel<1><<<ncu, 512, 32768>>>(src, ld, rows, iters, shared, bs, cyc, sink);
                if (mode == 2) stream_kernel<2><<<ncu, 512, 32768>>>(src, ld, rows, iters, shared, bs, cyc, sink);
                hipEventRecord(e1); CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1);
                CK(hipMemcpy(h.data(), cyc, ncu * 8, hipMemcpyDeviceToHost));
                double avg = 0; for (auto c : h) avg += (double)c; avg /= ncu;
                const double bytes_blk = (double)iters * rows * 128;
                if (rep) printf("shared=%d mode=%d: %.3f ms  %.1f B/cycle/CU  (%.2f TB/s aggregate, clock %.2f GHz)\n", shared, mode, ms,
                                bytes_blk / avg, bytes_blk * ncu / (ms * 1e-3) / 1e12, avg / (ms * 1e-3) / 1e9);
            }
        }
    return 0;
}

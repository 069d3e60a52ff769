// Micro-benchmark: HBM -> LDS streaming rate of two database tile walks (dev tool).
//   pattern 0: a stage = 256 rows x 128 B, row stride 1024 B, the 8 column chunks of a 256-row tile in 8 consecutive stages
//              (the K-chunked walk of the first kNN scan kernel)
//   pattern 1: a stage = 32 full rows (1024 B contiguous each)
// Every block walks its own tiles (blockIdx.x, +gridDim.x, ...), 2 stages in flight, 256 threads, 1 block per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int PATTERN, int DEPTH>
__global__ __launch_bounds__(256) void walk(const char* db, long long ntiles, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long long iters = my_tiles * 8;
    auto stage = [&](long long it) {
        const long long tile = blockIdx.x + (it / 8) * gridDim.x; const int kc = (int)(it & 7);
        char* l = smem + (it % DEPTH) * 32768;
        const char* t = db + tile * 262144;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const void* g;
            if (PATTERN == 0) g = t + (long long)(i * 32 + (tid >> 3)) * 1024 + kc * 128 + (tid & 7) * 16;
            else if (PATTERN == 2) g = t + (long long)(i * 32 + (tid >> 3)) * 1024 + kc * 128 + (((tid & 7) ^ (((tid >> 3) >> 1) & 7)) * 16);
            else g = t + (long long)(kc * 32 + i * 4 + wave) * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(l + (i * 4 + wave) * 1024), 16, 0, 0);
        }
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < DEPTH - 1; s++) if (s < iters) stage(s);
    for (long long it = 0; it < iters; it++) {
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + DEPTH - 1 < iters) stage(it + DEPTH - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) cycles[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const long long ntiles = argc > 1 ? atoll(argv[1]) : 40000;   // x 256 KiB (40000 = 10.5 GB)
    char* db; CK(hipMalloc(&db, ntiles * 262144)); CK(hipMemset(db, 1, ntiles * 262144));
    unsigned long long* cyc; CK(hipMalloc(&cyc, ncu * 8));
    for (int pat = 0; pat < 3; pat++)
        for (int depth = 2; depth <= 4; depth++)
            for (int rep = 0; rep < 2; rep++) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
#define LAUNCH(P, D) walk<P, D><<<ncu, 256, D * 32768>>>(db, ntiles, cyc)
                if (pat == 0) { if (depth == 2) LAUNCH(0, 2); else if (depth == 3) LAUNCH(0, 3); else LAUNCH(0, 4); }
                else if (pat == 1) { if (depth == 2) LAUNCH(1, 2); else if (depth == 3) LAUNCH(1, 3); else LAUNCH(1, 4); }
                else { if (depth == 2) LAUNCH(2, 2); else if (depth == 3) LAUNCH(2, 3); else LAUNCH(2, 4); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("pattern %d depth %d: %.3f ms  %.2f TB/s\n", pat, depth, ms, ntiles * 262144.0 / (ms * 1e-3) / 1e12);
            }
    return 0;
}

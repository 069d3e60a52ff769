// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds u16 element index; lane l reads at byte address addr[l]; prints what each lane gets.
//   hipcc --offload-arch=gfx950 -O2 -o tr_probe tr_probe.hip && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    int h[64]; uint16_t o[256]; int* d; uint16_t* od;
    hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
    for (int pat = 0; pat < 2; pat++) {
        // pat 0: lane l reads 8 bytes at 8*l (contiguous);  pat 1: rows of 64 B: lane i of a 16-group reads row i/4 (+8 rows for lanes >= 32), cols 16*(g&1) + 4*(i%4)
        for (int l = 0; l < 64; l++) {
            const int g = l >> 4, i = l & 15;
            h[l] = pat == 0 ? 8 * l : ((i >> 2) + 8 * (g >> 1)) * 64 + (16 * (g & 1) + 4 * (i & 3)) * 2;
        }
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d, od);
        hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; l++) printf("lane %2d addr %4d (elem %4d): %4d %4d %4d %4d\n", l, h[l], h[l] / 2, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    }
    return 0;
}

// Does a 16-bit LDS store at an odd address write both bytes where they belong (gfx950)?  And what does it cost against two byte stores?
//   hipcc -O3 --offload-arch=gfx950 lds_u16.hip -o lds_u16 && ./lds_u16
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
__global__ void k_check(uint8_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[64 * 8];
    for (int i = threadIdx.x; i < 64 * 8; i += 64) buf[i] = 0xEE;
    __syncthreads();
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)buf));
    const uint32_t at = base + threadIdx.x * 8 + (threadIdx.x & 3);  // offsets 0, 1, 2, 3 in turn: odd ones too, and 3 = across a dword
    const uint32_t v = 0xA1B2u | (threadIdx.x << 24);
    asm volatile("ds_write_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(at), "v"(v) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 64) out[i] = buf[i];
}
template <int TWO>
__global__ void k_time(uint32_t *sink, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[256 * 72];
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)buf)) + threadIdx.x * 68;
    uint32_t x = threadIdx.x * 2654435761u;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t at = base + (x >> 27);  // 0..31: random alignment
        if (TWO) {
            asm volatile("ds_write_b8 %0, %1\n\tds_write_b8_d16_hi %0, %1 offset:1" ::"v"(at), "v"(x) : "memory");
        } else {
            asm volatile("ds_write_b16 %0, %1" ::"v"(at), "v"(x) : "memory");
        }
    }
    __syncthreads();
    sink[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x] + x;
}
int main() {
    uint8_t *d, h[512];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const int o = t & 3;
        for (int j = 0; j < 8; ++j) {
            const uint8_t want = j == o ? 0xB2 : (j == o + 1 ? 0xA1 : 0xEE);
            if (h[t * 8 + j] != want) ++bad;
        }
    }
    printf("16-bit LDS stores at offsets 0..3: %s (%d bytes differ); lane 1: %02x %02x %02x %02x, lane 3: %02x %02x %02x %02x %02x\n", bad ? "WRONG" : "right", bad, h[8], h[9], h[10], h[11],
           h[24], h[25], h[26], h[27], h[28]);
    uint32_t *sink;
    hipMalloc(&sink, 4096 * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int two = 0; two < 2; ++two) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (two) hipLaunchKernelGGL(k_time<1>, dim3(4096), dim3(256), 0, 0, sink, 2000);
            else hipLaunchKernelGGL(k_time<0>, dim3(4096), dim3(256), 0, 0, sink, 2000);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%s: %.3f ms\n", two ? "two byte stores" : "one 16-bit store", ms);
        }
    }
    return bad != 0;
}

// What the LDS pipe of gfx950 charges for the decode kernels' access patterns (cycles per wavefront instruction, all 64 lanes active, one
// workgroup of 256 threads per CU slot, addresses from an LCG so that every instruction depends on nothing but its address register):
//   hipcc -O3 --offload-arch=gfx950 lds_costs.hip -o lds_costs && ./lds_costs
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) uint8_t lds_u8;
enum { RD_U16_RANDOM, RD_B32_RANDOM, RD_B64_RANDOM, RD_B32X2_RANDOM, RD_B128_RANDOM, RD_U16_OWN_BANK, WR_B8_RANDOM, WR_B8_PAIR, WR_B32_RANDOM, WR_B8_OWN_STRIP, N_KINDS };
static const char *names[N_KINDS] = {"ds_read_u16, random in 32 KiB (D1's lookups)", "ds_read_b32, random in 16 KiB", "ds_read_b64, random 8-byte entries in 16 KiB (D3's lookups)",
                                     "2 x ds_read_b32, the same index in two 8 KiB arrays", "ds_read_b128, random 16-byte entries in 32 KiB", "ds_read_u16, every lane in a bank of its own",
                                     "ds_write_b8, random in 4 KiB (one symbol into a shared stage)", "2 x ds_write_b8, consecutive bytes, random in 4 KiB (D3's fast step)",
                                     "ds_write_b32, random in 4 KiB", "2 x ds_write_b8, consecutive bytes in the lane's own 68-byte strip"};
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t buf[];
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)buf));
    // eight offsets per lane, fixed; every iteration XORs one wavefront-uniform random value into all of them (a bijection of the
    // address space: the lanes' addresses stay as random against each other as they were) -- one VALU instruction per LDS instruction
    constexpr uint32_t RANGE = KIND == RD_U16_RANDOM ? 32768u : (KIND == RD_B128_RANDOM ? 32768u : (KIND >= WR_B8_RANDOM ? 4096u : 16384u));
    constexpr uint32_t ALIGN = KIND == RD_U16_RANDOM || KIND == RD_U16_OWN_BANK ? 2u : (KIND == RD_B64_RANDOM ? 8u : (KIND == RD_B128_RANDOM ? 16u : (KIND == RD_B32_RANDOM || KIND == RD_B32X2_RANDOM || KIND == WR_B32_RANDOM ? 4u : 1u)));
    uint32_t off[8];
    uint32_t h = (threadIdx.x + blockIdx.x * 256) * 2654435761u + 12345u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        h = h * 1664525u + 1013904223u;
        if (KIND == RD_U16_OWN_BANK) off[u] = (((h >> 20) & 0x7fu) << 7) + ((threadIdx.x & 31u) << 2);  // row random, bank = lane
        else if (KIND == WR_B8_OWN_STRIP) off[u] = (threadIdx.x & 63u) * 68u + (threadIdx.x >> 6) * 4352u + ((h >> 26) & 63u);
        else if (KIND == RD_B32X2_RANDOM) off[u] = (h >> 8) % 8192u & ~3u;
        else off[u] = ((h >> 8) % RANGE) & ~(ALIGN - 1u);
    }
    uint32_t acc = 0, s = blockIdx.x * 7919u + 1u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;  // (wavefront-uniform: scalar)
        const uint32_t m = KIND == RD_U16_OWN_BANK ? ((s >> 12) & 0x3f80u) : (KIND == WR_B8_OWN_STRIP ? 0u : (KIND == RD_B32X2_RANDOM ? ((s >> 12) & 8191u & ~3u) : ((s >> 12) & (RANGE - 1u) & ~(ALIGN - 1u))));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t at = base + (off[u] ^ m);
            uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (KIND == RD_U16_RANDOM || KIND == RD_U16_OWN_BANK) asm volatile("ds_read_u16 %0, %1" : "=v"(v0) : "v"(at) : "memory");
            if (KIND == RD_B32_RANDOM) asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(at) : "memory");
            if (KIND == RD_B64_RANDOM) { uint2 r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(at) : "memory"); v0 = r.x; v1 = r.y; }
            if (KIND == RD_B32X2_RANDOM) asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:8192" : "=v"(v0), "=v"(v1) : "v"(at) : "memory");
            if (KIND == RD_B128_RANDOM) { uint4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(at) : "memory"); v0 = r.x; v1 = r.y; v2 = r.z; v3 = r.w; }
            if (KIND == WR_B8_RANDOM) asm volatile("ds_write_b8 %0, %1" ::"v"(at), "v"(h) : "memory");
            if (KIND == WR_B8_PAIR || KIND == WR_B8_OWN_STRIP) asm volatile("ds_write_b8 %0, %1\n\tds_write_b8_d16_hi %0, %1 offset:1" ::"v"(at), "v"(h) : "memory");
            if (KIND == WR_B32_RANDOM) asm volatile("ds_write_b32 %0, %1" ::"v"(at), "v"(h) : "memory");
            acc += v0 ^ v1 ^ v2 ^ v3;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc + s;
}
template <int KIND>
static void run(uint32_t *sink, int cus) {
    const int iters = 2000, grid = cus * 2;  // two workgroups = 8 wavefronts per CU: the pipe is what is measured, not its latency
    const size_t smem = 49152;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), smem, 0, sink, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    // per CU: 8 wavefronts x iters instructions (x 2 for the pairs) in best ms at ~2.4 GHz
    const double instr_per_cu = 8.0 * 8 * iters * ((KIND == RD_B32X2_RANDOM || KIND == WR_B8_PAIR || KIND == WR_B8_OWN_STRIP) ? 2 : 1);
    printf("%-78s %7.3f ms  %5.1f cycles per instruction at 2.4 GHz\n", names[KIND], best, best * 1e-3 * 2.4e9 / instr_per_cu);
}
int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t *sink;
    (void)hipMalloc(&sink, static_cast<size_t>(cus) * 2 * 256 * 4);
    run<RD_U16_RANDOM>(sink, cus);
    run<RD_B32_RANDOM>(sink, cus);
    run<RD_B64_RANDOM>(sink, cus);
    run<RD_B32X2_RANDOM>(sink, cus);
    run<RD_B128_RANDOM>(sink, cus);
    run<RD_U16_OWN_BANK>(sink, cus);
    run<WR_B8_RANDOM>(sink, cus);
    run<WR_B8_PAIR>(sink, cus);
    run<WR_B32_RANDOM>(sink, cus);
    run<WR_B8_OWN_STRIP>(sink, cus);
    return 0;
}

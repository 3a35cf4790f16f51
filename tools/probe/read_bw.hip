// Streaming-read bandwidth of an MI355X for the access patterns K1 could use: what is the
// floor of a kernel that must read 1 GiB once?  hipcc --offload-arch=gfx950 -O3 -o build/read_bw tools/probe/read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ p, size_t n_vec, uint32_t *out) {
    uint32_t acc = 0;
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    for (; i + (INFLIGHT - 1) * stride < n_vec; i += INFLIGHT * stride) {
        uint4 v[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n_vec; i += stride) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}
// tile pattern of K1: a workgroup owns 64 KiB tiles, reads them as 16 rounds of 4 KiB, 4 rounds in flight
__global__ __launch_bounds__(256) void k_read_tiles(const uint4 *__restrict__ p, uint32_t n_tiles, uint32_t *out) {
    uint32_t acc = 0;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint4 *base = p + static_cast<size_t>(t) * 4096 + threadIdx.x;
        for (int r0 = 0; r0 < 16; r0 += 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = base[(r0 + u) * 256];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// K4's traffic without its arithmetic: a workgroup reads 64 KiB tiles round by round (4 KiB,
// next round prefetched) and after every round stores 600 dwords (0.586 of what it read).
__global__ __launch_bounds__(256) void k_rw_tiles(const uint4 *__restrict__ p, uint32_t n_tiles, uint32_t *__restrict__ q, uint32_t *out) {
    uint32_t acc = 0;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint4 *base = p + static_cast<size_t>(t) * 4096 + threadIdx.x;
        uint32_t *dst = q + static_cast<size_t>(t) * 9600;
        uint4 cur = base[0];
        for (int r = 0; r < 16; ++r) {
            uint4 nxt = cur;
            if (r + 1 < 16) nxt = base[(r + 1) * 256];
            acc += cur.x ^ cur.y ^ cur.z ^ cur.w;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < 600; i += 256) dst[r * 600 + i] = acc + i;
            cur = nxt;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <typename F>
static float timeit(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 10;
}
int main() {
    const size_t n = 1ull << 30;
    uint4 *d; uint32_t *o;
    hipMalloc(&d, n); hipMalloc(&o, 64);
    hipMemset(d, 1, n);
    uint32_t *q;
    hipMalloc(&q, n);
    for (int grid : {1024, 2048, 4096}) {
        float a = timeit([&] { hipLaunchKernelGGL(k_rw_tiles, dim3(grid), dim3(256), 0, 0, d, static_cast<uint32_t>(n / 65536), q, o); });
        printf("grid %5d: K4 traffic pattern (read 1 GiB, write 0.586 GiB) %.3f ms (%.2f TB/s of traffic)\n", grid, a, n * 1.586 / a / 1e9);
    }
    for (int grid : {1024, 1280, 2048, 4096, 8192, 16384}) {
        float a = timeit([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, d, n / 16, o); });
        float b = timeit([&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, d, n / 16, o); });
        float c = timeit([&] { hipLaunchKernelGGL(k_read_tiles, dim3(grid), dim3(256), 0, 0, d, static_cast<uint32_t>(n / 65536), o); });
        printf("grid %5d: grid-stride 4 in flight %.3f ms (%.2f TB/s), 8 in flight %.3f ms (%.2f TB/s), K1 tile pattern %.3f ms (%.2f TB/s)\n", grid, a, n / a / 1e9, b,
               n / b / 1e9, c, n / c / 1e9);
    }
    return 0;
}

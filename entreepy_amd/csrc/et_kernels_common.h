// et_kernels_common.h -- what et_kernels.hip (the kernels on the common path) and et_kernels_fallback.hip (the round-1
// kernels that remain as the decoder of what lies outside the tree walk's and the row walk's domains) share: wavefront
// scans, guarded loads, which blocks are "special", the launch helpers.  Device code; included by .hip files only.
#pragma once

#include "et_kernels.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace et {

// --------------------------------------------------------------------------------
// wavefront / workgroup scans (DPP, no LDS traffic inside a wavefront)
// --------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t x) {
    return x + static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xf, false));
}

// Inclusive prefix sum over the 64 lanes of a wavefront.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
    x = dpp_add<0x111, 0xf>(x);  // row_shr:1
    x = dpp_add<0x112, 0xf>(x);  // row_shr:2
    x = dpp_add<0x114, 0xf>(x);  // row_shr:4
    x = dpp_add<0x118, 0xf>(x);  // row_shr:8  -> each row of 16 scanned
    x = dpp_add<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
    x = dpp_add<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
    return x;
}

__device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// Exclusive prefix sum over the 256 threads of a workgroup; *total = sum of all.
// `scratch` is 4 LDS words.  Contains ONE barrier; the caller must separate two
// calls that reuse `scratch` by another barrier.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t x, uint32_t *scratch, uint32_t *total) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (scalar: what lies before a wavefront is added up on the scalar unit)
    const uint32_t inc = wave_inclusive_scan(x);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(scratch[0]), w1 = __builtin_amdgcn_readfirstlane(scratch[1]),
                   w2 = __builtin_amdgcn_readfirstlane(scratch[2]), w3 = __builtin_amdgcn_readfirstlane(scratch[3]);
    uint32_t before = 0;
    if (wave > 0) before += w0;
    if (wave > 1) before += w1;
    if (wave > 2) before += w2;
    *total = w0 + w1 + w2 + w3;
    return before + (inc - x);
}

extern __shared__ __attribute__((aligned(16))) uint8_t dec_smem_raw[];  // ALL dynamic LDS of a decode kernel

__device__ __forceinline__ uint32_t load_be32_guarded(const uint32_t *__restrict__ words, uint64_t idx, uint64_t n_bytes) {
    // big-endian numeric value of stream bytes [4*idx, 4*idx+4), zero beyond n_bytes
    const uint64_t b0 = idx * 4;
    if (b0 + 4 <= n_bytes) return __builtin_bswap32(words[idx]);
    uint32_t v = 0;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(words);
    for (int k = 0; k < 4; ++k)
        if (b0 + k < n_bytes) v |= static_cast<uint32_t>(bytes[b0 + k]) << (24 - 8 * k);
    return v;
}
__device__ __forceinline__ uint32_t block_limit(uint64_t n_bytes, uint64_t block) {
    const uint64_t rel = n_bytes * 8 - block * DEC_BLOCK_WORDS * 32 + DEC_WARMUP_BITS;  // same origin as walk_subsequence
    // UINT32_MAX unless the stream ends inside (or just after) the staged words of this block
    return rel < DEC_STAGED_WORDS * 32 + 64 ? static_cast<uint32_t>(rel) : 0xffffffffu;
}

// Special blocks keep the LDS-window kernels: the stream's first block and the one or two
// whose staged words reach the stream's end; everything else is "interior".
__device__ __forceinline__ bool special_block(uint64_t b, uint64_t n_bytes) { return b == 0 || block_limit(n_bytes, b) != 0xffffffffu; }
// workgroup i of a special-only launch (grid 3) looks at block 0, n-2, n-1
__device__ __forceinline__ uint64_t special_candidate(uint32_t i, uint32_t n_blocks) {
    if (i == 0) return 0;
    const uint64_t c = static_cast<uint64_t>(n_blocks) + i;
    return c >= 4 ? c - 3 : ~0ull;  // i = 1 -> n-2, i = 2 -> n-1; never block 0 again
}

// k_dec_sync_reg2 works on superblocks of two blocks (512-bit lanes) and takes those whose
// two blocks are both interior; the LDS-window kernel then gets the rest: blocks 0, 1 and
// up to six at the end (workgroup i of a grid of 8).
__device__ __forceinline__ bool super_interior(uint64_t s, uint64_t n_bytes, uint32_t n_blocks) {
    return 2 * s + 1 < n_blocks && !special_block(2 * s, n_bytes) && !special_block(2 * s + 1, n_bytes);
}
__device__ __forceinline__ uint64_t special_candidate2(uint32_t i, uint32_t n_blocks) {
    if (i < 2) return i;
    const uint64_t c = static_cast<uint64_t>(n_blocks) + i;
    return c >= 10 ? c - 8 : ~0ull;  // i = 2..7 -> n-6..n-1, never 0 or 1 again
}

constexpr int RW_WORDS = 13;  // W[j] = stream word 8 * sub - 4 + j (host order): 4 run-in words, 8 own, 1 beyond

typedef __attribute__((address_space(3))) uint8_t lds_u8;

// ---- launch helpers (host) -------------------------------------------------------------------------------
// A launch that carries its own timing events (hipExtLaunchKernelGGL: the dispatch's completion
// signal records begin and end, no marker packets in the stream -- ten hipEventRecord markers per
// encode+decode cost ~70 us at 1 GiB), or a plain launch when no events are asked for.
#define ET_LAUNCH_TIMED(kernel_, grid_, block_, smem_, stream_, evs_, ...)                                                        \
    do {                                                                                                                          \
        if ((evs_).start || (evs_).stop) hipExtLaunchKernelGGL(kernel_, grid_, block_, smem_, stream_, (evs_).start, (evs_).stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel_, grid_, block_, smem_, stream_, __VA_ARGS__);                                             \
    } while (0)

// Workgroups of `kernel` a CU holds at once (occupancy query), remembered per (kernel, device, LDS size):
// the query sits on the launch path, and kernels that share a signature (the k_encode_tiles<RING> variants,
// k_dec_sync<first/later>, the k_dec_sync_reg variants) are different entries.
static int resident_per_cu(const void *kernel, size_t smem, int *cus_out, int threads = BLOCK) {
    struct Entry {
        const void *kernel;
        size_t smem;
        int dev, cus, per_cu;
    };
    static thread_local Entry cache[24];
    static thread_local int n_cached = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < n_cached; ++i)
        if (cache[i].kernel == kernel && cache[i].smem == smem && cache[i].dev == dev) {
            *cus_out = cache[i].cus;
            return cache[i].per_cu;
        }
    int cus = 256, per_cu = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, smem) != hipSuccess) per_cu = 0;
    if (n_cached < 24) cache[n_cached++] = Entry{kernel, smem, dev, cus, per_cu};
    *cus_out = cus;
    return per_cu;
}

// Grid of the tile-striding encode kernels: the workgroups the device holds at once (both kernels use
// < 64 SGPRs, where the query is exact), so that every workgroup gets within one tile of the same share.// Grid of a chunked decode kernel: one workgroup per chunk, or -- ticketed -- as many
// workgroups as the occupancy API reports resident (an over-estimate is harmless).
template <typename K>
static uint32_t decode_grid(K kernel, size_t smem, uint32_t n_chunks, bool ticketed, int threads = BLOCK) {
    if (!ticketed) return n_chunks;
    int cus = 256;
    int per_cu = resident_per_cu(reinterpret_cast<const void *>(kernel), smem, &cus, threads);
    if (per_cu < 1) per_cu = 1;
    const uint32_t g = static_cast<uint32_t>(cus) * static_cast<uint32_t>(per_cu);
    return n_chunks < g ? (n_chunks ? n_chunks : 1) : g;
}

// k_dec_write_wave's fallback pair (et_kernels_fallback.hip): k_dec_write_reg for the interior blocks, k_dec_write for the first / last ones
void launch_dec_write_fallback(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint64_t n_subs, const DecodeTables &tb, const uint32_t *sub_state,
                               const unsigned long long *blk_off, uint64_t n_symbols, uint8_t *out, uint32_t *ticket, const SideLane *side, bool ticket_is_zero,
                               const uint32_t *void_flags, KernelEvents ev);

}  // namespace et

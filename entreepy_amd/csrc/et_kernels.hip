// et_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the entreepy
// Huffman path.  Pure integer/bit work: no MFMA; the bounds are HBM bandwidth, the
// LDS pipe (atomics + table reads) and VALU issue.
//
//   encode.zig:43-47   -> k_hist_tiles (+ k_hist_reduce)          "K1"
//   encode.zig:308-313 -> k_tile_bits, k_scan_fused               "K2" (the serial
//                         bits_written counter turned into a scan over tiles)
//   encode.zig:303-315 -> k_encode_tiles / k_encode_tiles_long    "K4"
//   decode.zig:143-203 -> "D1..D3": k_dec_sync_reg2 (first sweep, 512-bit lanes in registers),
//                         k_dec_check + k_dec_sync_reg<false> (repair on a worklist),
//                         k_scan_* (+ verification), k_dec_write_reg; the LDS-window
//                         kernels k_dec_sync / k_dec_write for the stream's first and last
//                         blocks; k_dec_maps[_reg], k_dec_compose, k_dec_chain,
//                         k_dec_resolve[_reg]: the bounded fallback for codes that do
//                         not self-synchronise
//
// Geometry: workgroups of 256 threads (4 wavefronts of 64); K1 uses 512 on one set of counters.
// Encode side: a "round" is 4 KiB of input, one 16-byte load per lane, fully
// coalesced; a "tile" is 1..16 consecutive rounds and is the unit for which K1
// leaves a 256-bin histogram and K4 gets a start bit offset.
#include "et_kernels.h"
#include "et_treewalk.h"

#include <hip/hip_ext.h>

#include <hip/hip_runtime.h>

namespace et {

// Non-temporal accesses, per kernel (measured r03, text-1G, ms): K1's loads of the text 0.218 -> 0.173 (the histogram pass was
// held up by the cache lines it left behind, not by its ds_add rate); D3's stores of the output 0.495 -> 0.490, and the text is
// not pushed out of the caches by the output in front of the next K1.  NOT: K4's stores (its rounds end mid-line: 0.37 -> 0.40,
// also with flushes cut at 128-byte lines), K4's loads (K4 0.37 -> 0.365 but D1 behind it 0.187 -> 0.20), D1's and D3's loads
// (their lanes share lines: 0.187 -> 0.29, 0.49 -> 0.53).
#ifndef ET_NT_LOAD_K1
#define ET_NT_LOAD_K1 1
#endif
#ifndef ET_NT_LOAD_K4
#define ET_NT_LOAD_K4 0
#endif
#ifndef ET_NT_STORE_K4
#define ET_NT_STORE_K4 0
#endif

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

// --------------------------------------------------------------------------------
// input addressing
// --------------------------------------------------------------------------------
// The text is addressed relative to a 16-byte aligned base: the stream occupies
// bytes [lo, hi) of it (lo < 16).  A lane's 16-byte chunk is loaded with one
// dwordx4 when it lies fully inside [lo, hi); the (at most two) partial chunks of a
// stream are assembled from guarded byte loads so nothing outside it is touched.
struct Chunk {
    uint32_t w[4];
    uint32_t valid;  // bit k set <=> byte k belongs to the stream
};

__device__ __forceinline__ Chunk load_chunk(const uint8_t *__restrict__ base, uint64_t off, uint64_t lo, uint64_t hi) {
    Chunk c;
    if (off >= lo && off + 16 <= hi) {
        const uint4 v = *reinterpret_cast<const uint4 *>(base + off);
        c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
        c.valid = 0xffffu;
    } else {
        c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
        c.valid = 0;
        if (off < hi && off + 16 > lo) {
            for (int k = 0; k < 16; ++k) {
                const uint64_t p = off + k;
                if (p >= lo && p < hi) {
                    c.w[k >> 2] |= static_cast<uint32_t>(base[p]) << (8 * (k & 3));
                    c.valid |= 1u << k;
                }
            }
        }
    }
    return c;
}

// Tile-level version: `interior` (workgroup-uniform, so a scalar branch) says the whole
// tile lies inside the stream and every chunk of it is a plain 16-byte load.
// NT: a non-temporal load (the text is read once per pass: nothing of it is worth a cache line)
template <bool NT = false>
__device__ __forceinline__ Chunk load_chunk_in_tile(const uint8_t *__restrict__ base, uint64_t off, uint64_t lo, uint64_t hi, bool interior) {
    if (interior) {
        Chunk c;
        typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
        u32x4_nt v;
        if (NT) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt *>(base + off));
        else v = *reinterpret_cast<const u32x4_nt *>(base + off);
        c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
        c.valid = 0xffffu;
        return c;
    }
    return load_chunk(base, off, lo, hi);
}

// --------------------------------------------------------------------------------
// K1: byte histogram, LDS-privatised per lane bank
// --------------------------------------------------------------------------------
// LDS layout [bin][32] u32, shared by the workgroup's 4 wavefronts: lane l adds to
// replica l & 31 of its symbol's bin, i.e. to LDS bank l & 31 whatever the symbol.
// A ds_add_u32 wave-instruction is served in two half-waves of 32 lanes, so within
// a half every lane hits its own bank: no bank conflicts and no same-address
// serialisation, however skewed the text.  (Per-wavefront private copies would only
// multiply the LDS footprint: LDS atomics from different wavefronts never overlap in
// time on the one LDS pipe of a CU.)  32 KiB per workgroup -> 5 workgroups per CU.
// Measured alternative: two 16-bit counters per word (16 KiB, 8 workgroups per CU) is
// SLOWER (0.30 vs 0.27 ms per GiB): the ceiling is the ds_add rate itself (~8 LDS
// cycles per wave-instruction, ~4.9 TB/s chip-wide), not occupancy.
// Counters are u32 (a tile is at most 512 KiB = MAX_ROUNDS_PER_TILE rounds of 4 KiB: 2^19 symbols, far inside a u32);
// tile totals go out as u32, workgroup totals as u64.
// HIST_BLOCK threads share the workgroup's one set of counters: the 32 KiB of LDS allow only
// 4 workgroups per CU (a fifth does not fit beside the others' 160 KiB exactly), so 256
// threads meant 4 wavefronts per SIMD and a latency-bound kernel; 512 threads double the
// loads in flight on the same LDS: 0.283 -> 0.227 ms at 1 GiB (1024 threads: 0.235).
constexpr int HIST_BLOCK = 512;
constexpr uint32_t HIST_ROWS = 1024;  // k_hist_tiles' grid at most: block_hist is [256][HIST_ROWS]
__global__ __launch_bounds__(HIST_BLOCK) void k_hist_tiles(const uint8_t *__restrict__ base, uint64_t lo, uint64_t hi,
                                                           uint32_t rounds_per_tile, uint32_t n_tiles,
                                                           uint32_t *__restrict__ tile_hist,
                                                           unsigned long long *__restrict__ block_hist, unsigned long long *__restrict__ hist) {
    __shared__ __attribute__((aligned(16))) uint32_t sh[256 * 32];
    const int tid = threadIdx.x;
    for (int i = tid; i < 256 * 32; i += HIST_BLOCK) sh[i] = 0;
    __syncthreads();

    uint32_t *mine = sh + (tid & 31);
    unsigned long long acc = 0;  // thread `tid` < 256 owns bin `tid` of the workgroup total
    const uint64_t tile_bytes = static_cast<uint64_t>(rounds_per_tile) * ROUND_BYTES;
    const uint32_t tile_chunks = rounds_per_tile * (ROUND_BYTES / 16);  // 16-byte chunks per tile

    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t t0 = static_cast<uint64_t>(t) * tile_bytes;
        const bool interior = t0 >= lo && t0 + tile_bytes <= hi;
        // Four 16-byte loads in flight per lane, then their 64 (conflict-free) LDS atomics.
        for (uint32_t c0 = tid; c0 < tile_chunks; c0 += 4 * HIST_BLOCK) {
            Chunk c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c[u].valid = 0;
                if (c0 + u * HIST_BLOCK < tile_chunks) c[u] = load_chunk_in_tile<ET_NT_LOAD_K1>(base, t0 + static_cast<uint64_t>(c0 + u * HIST_BLOCK) * 16, lo, hi, interior);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c[u].valid == 0xffffu) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const uint32_t sym = (c[u].w[d] >> (8 * b)) & 0xffu;
                            atomicAdd(mine + sym * 32, 1u);  // ds_add_u32, no return
                        }
                    }
                } else if (c[u].valid) {
                    for (int k = 0; k < 16; ++k)
                        if (c[u].valid & (1u << k)) atomicAdd(mine + ((c[u].w[k >> 2] >> (8 * (k & 3))) & 0xffu) * 32, 1u);
                }
            }
        }
        __syncthreads();
        // Tile flush: thread = bin; sum (and clear) its 32 replicas, 16 bytes at a time.
        // Rotating the start by the bin keeps the 16-lane groups of ds_read_b128 on
        // different bank quads.
        if (tid < 256) {
            uint32_t total = 0;
            uint4 *row = reinterpret_cast<uint4 *>(sh + tid * 32);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = (q + tid) & 7;
                const uint4 v = row[j];
                total += v.x + v.y + v.z + v.w;
                row[j] = make_uint4(0, 0, 0, 0);
            }
            tile_hist[static_cast<uint64_t>(t) * 256 + tid] = total;
            acc += total;
        }
        __syncthreads();
    }
    // (transposed: column `tid`, row blockIdx.x -- 256 scattered 8-byte stores per workgroup, once, so that the
    // reducing workgroups read their columns as contiguous rows)
    if (tid < 256) block_hist[static_cast<uint64_t>(tid) * HIST_ROWS + blockIdx.x] = acc;
}

// Sums of block_hist[256][HIST_ROWS] (one row of n_rows partial counts per byte value) into hist[256]: workgroup w
// OWNS values 2w and 2w + 1, so the totals are plain stores -- no zeroing beforehand, no atomics -- and can go to two
// places: the device's copy and, for the host's code construction that waits behind this kernel, pinned host memory
// (no copy command in between).
__global__ __launch_bounds__(BLOCK) void k_hist_reduce(const unsigned long long *__restrict__ block_hist, uint32_t n_rows,
                                                       unsigned long long *__restrict__ hist, unsigned long long *__restrict__ host_hist,
                                                       unsigned long long epoch, unsigned long long *__restrict__ hist_also) {
    __shared__ unsigned long long part[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long *row0 = block_hist + static_cast<uint64_t>(2 * blockIdx.x) * HIST_ROWS, *row1 = row0 + HIST_ROWS;
    unsigned long long s0 = 0, s1 = 0;
#pragma unroll
    for (uint32_t k = 0; k < HIST_ROWS / BLOCK; ++k) {
        const uint32_t r = k * BLOCK + tid;
        if (r < n_rows) {
            s0 += row0[r];
            s1 += row1[r];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
    }
    if (lane == 0) {
        part[0][wave] = s0;
        part[1][wave] = s1;
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t0 = part[0][0] + part[0][1] + part[0][2] + part[0][3], t1 = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        hist[2 * blockIdx.x] = t0;
        hist[2 * blockIdx.x + 1] = t1;
        if (hist_also) {  // (a second device copy where a caller wants one: the row a group's exchange sends)
            hist_also[2 * blockIdx.x] = t0;
            hist_also[2 * blockIdx.x + 1] = t1;
        }
        if (host_hist) {
            // host_hist[256 ..]: one "these two are there" word per workgroup -- the host polls them instead of waiting
            // for the stream (no completion signal, no wake-up in between)
            host_hist[2 * blockIdx.x] = t0;
            host_hist[2 * blockIdx.x + 1] = t1;
            __threadfence_system();
            __hip_atomic_store(host_hist + 256 + blockIdx.x, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The header and dictionary at src (any alignment; at most n bytes) into pinned host memory, then `epoch` into
// *host_done: what the host of a decode parses.  It polls the word instead of waiting for a copy command.  The first
// byte says how many dictionary entries follow the 5 header bytes (decode.zig:34), an entry is at most 8 + 8 + 32 bits:
// ET_HEADER_BOUND bytes are enough, whatever the codes are.
__global__ __launch_bounds__(1024) void k_header_to_host(const uint8_t *__restrict__ src, uint32_t n, uint32_t *__restrict__ host_dst,
                                                         unsigned long long *__restrict__ host_done, unsigned long long epoch) {
    // (the most a dictionary can take, not what THIS one takes: reading src[0] first to know is a memory round trip in front of
    // the copy -- 1544 bytes at most either way)
    const uint32_t bound = header_bound(255);
    if (bound < n) n = bound;
    for (uint32_t w = threadIdx.x; w * 4 < n; w += 1024) {
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4 && w * 4 + k < n; ++k) v |= static_cast<uint32_t>(src[w * 4 + k]) << (8 * k);
        host_dst[w] = v;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// n_words dwords of device memory into pinned host memory, then `epoch` into *host_done (the gathered histograms of
// a group encode: the host polls instead of waiting behind a copy command).
__global__ __launch_bounds__(1024) void k_words_to_host(const uint32_t *__restrict__ src, uint32_t n_words, uint32_t *__restrict__ host_dst,
                                                        unsigned long long *__restrict__ host_done, unsigned long long epoch) {
    for (uint32_t w = threadIdx.x; w < n_words; w += 1024) host_dst[w] = src[w];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// --------------------------------------------------------------------------------
// K2: tile bit totals and their exclusive scan
// --------------------------------------------------------------------------------
// One wavefront per tile: lane l holds the code lengths of bins 4l..4l+3 and reads
// the tile's counts for them with one 16-byte load.  The lengths arrive as a kernel ARGUMENT (256 bytes), so nothing
// has to be uploaded in front of this kernel; what the kernels BEHIND it need from the host -- K4's code table, the
// file header for the scan -- workgroup 0 copies from the pinned block into device memory on the side.
struct CodeLengths {
    uint32_t packed[64];  // four lengths per word, symbol 4l in the low byte of word l
};
__global__ __launch_bounds__(BLOCK) void k_tile_bits(const uint32_t *__restrict__ tile_hist, uint32_t n_tiles, const CodeLengths lengths,
                                                     unsigned long long *__restrict__ tile_bits, const uint32_t *__restrict__ host_src,
                                                     uint32_t *__restrict__ dev_dst, uint32_t copy_words, unsigned long long *__restrict__ host_taken,
                                                     unsigned long long epoch) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0) {
        // (all of a thread's loads first, then its stores: the source is HOST memory, ~2 us a round trip, and a plain copy loop
        // waits for each word before it asks for the next -- up to 8 round trips in a row for the ~1900 words of a block)
        for (uint32_t i0 = threadIdx.x; i0 < copy_words; i0 += 8 * BLOCK) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = i0 + k * BLOCK < copy_words ? host_src[i0 + k * BLOCK] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + k * BLOCK < copy_words) dev_dst[i0 + k * BLOCK] = v[k];
        }
        __syncthreads();  // (every thread's loads have returned: the host may fill the block again once it sees `epoch`)
        if (threadIdx.x == 0) __hip_atomic_store(host_taken, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t wave_global = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const uint32_t l4 = lengths.packed[lane];
    const uint32_t len_x = l4 & 0xffu, len_y = (l4 >> 8) & 0xffu, len_z = (l4 >> 16) & 0xffu, len_w = l4 >> 24;
    for (uint32_t t = wave_global; t < n_tiles; t += n_waves) {
        const uint4 c = reinterpret_cast<const uint4 *>(tile_hist + static_cast<uint64_t>(t) * 256)[lane];
        unsigned long long s = static_cast<unsigned long long>(c.x) * len_x + static_cast<unsigned long long>(c.y) * len_y +
                               static_cast<unsigned long long>(c.z) * len_z + static_cast<unsigned long long>(c.w) * len_w;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) tile_bits[t] = s;
    }
}

// Exclusive scan of in[0..n) into out[0..n] (+ base), ONE launch: every group of 1024 scans its own elements, publishes
// its total and adds up the totals of the groups before it as they appear -- nobody waits for anybody who waits (a
// group's total needs nothing from outside), groups start in index order, and the words travel as relaxed device-scope
// atomics.  pub[g] = epoch << 48 | mismatch << 47 | total: the caller hands a fresh epoch (1 .. 65535 within one
// lifetime of the zeroed buffer) instead of zeroing the words before every launch.
// K2's extras: the words that hold a tile boundary are zeroed, the header is copied in.  D2's extras: every block must
// have started where the block before it ended (verify_*), and the last group reports flags and total to the host.
constexpr unsigned long long SCAN_TOTAL_MASK = (1ull << 47) - 1ull;
constexpr uint32_t SCAN_POLLS = 1u << 22;  // (seconds: a group below that never publishes means a dead device; the launch still ends)
template <typename T>
__global__ __launch_bounds__(1024) void k_scan_fused(const T *__restrict__ in, uint32_t n, unsigned long long *__restrict__ out,
                                                     unsigned long long *__restrict__ pub, uint32_t epoch, unsigned long long base,
                                                     uint32_t *__restrict__ zero_words, const uint32_t *__restrict__ header_src, uint32_t header_words,
                                                     unsigned long long *__restrict__ total_copy, const uint32_t *__restrict__ verify_state,
                                                     const uint32_t *__restrict__ verify_exit, uint32_t *__restrict__ verify_flag, uint32_t verify_first,
                                                     uint32_t verify_stride, uint32_t verify_mask, const uint32_t *__restrict__ report_src,
                                                     uint32_t *__restrict__ report_dst, uint32_t report_epoch) {
    __shared__ unsigned long long wsum[16], psum[16];
    __shared__ uint32_t bad_any;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = blockIdx.x, i = g * 1024 + tid;
    if (tid == 0) bad_any = 0;
    __syncthreads();
    // (block 0 must have started at verify_first, the stream's known first bit; 0xffffffff: not checked)
    if (verify_state && i < n) {
        const uint32_t want = i > 0 ? verify_exit[i - 1] : verify_first;
        // (verify_state: the start each block's first lane used -- sub_state, every BLOCK-th entry's low byte, a bit
        // offset; or the tree walk's blk_start, a row -- against what the block before ended on)
        if (want != 0xffffffffu && (verify_state[static_cast<uint64_t>(i) * verify_stride] & verify_mask) != want) {
            *verify_flag = 1;  // (for the kernels behind this one)
            bad_any = 1;
        }
    }
    const unsigned long long x = (i < n) ? static_cast<unsigned long long>(in[i]) : 0ull;
    const unsigned long long inc = wave_inclusive_scan64(x);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    const unsigned long long mine = before + inc - x;
    unsigned long long group_total = 0;
    for (int w = 0; w < 16; ++w) group_total += wsum[w];
    const unsigned long long tag = static_cast<unsigned long long>(epoch) << 48;
    if (tid == 0) __hip_atomic_store(pub + g, tag | (bad_any ? 1ull << 47 : 0ull) | (group_total & SCAN_TOTAL_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the groups before this one
    unsigned long long part = 0, bad = 0;
    for (uint32_t j = tid; j < g; j += 1024) {
        unsigned long long v = 0;
        for (uint32_t poll = 0; poll < SCAN_POLLS; ++poll) {
            v = __hip_atomic_load(pub + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 48) == epoch) break;
            __builtin_amdgcn_s_sleep(2);
        }
        part += v & SCAN_TOTAL_MASK;
        bad |= (v >> 47) & 1ull;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        part += __shfl_xor(part, d, 64);
        bad |= __shfl_xor(bad, d, 64);
    }
    if (lane == 0) psum[wave] = part | (bad << 63);
    __syncthreads();
    unsigned long long prefix = base;
    bool bad_before = false;
    for (int w = 0; w < 16; ++w) {
        prefix += psum[w] & ~(1ull << 63);
        bad_before = bad_before || (psum[w] >> 63);
    }
    // The word the first tile starts in (the header/body seam of a head shard) is zeroed by ONE
    // thread, element 0's, so that the header can be copied over it below without a race: tiles
    // that begin in the same word (zero-bit tiles: text of nothing but the symbol the reference
    // drops, quirk Q1) and the stream's end leave it alone.
    const unsigned long long seam = base >> 5;
    if (i < n) {
        const unsigned long long v = mine + prefix;
        out[i] = v;
        // (K4 merges with atomicOr into the word a tile ENDS in when that end is not word-aligned: the next
        // tile's first word, zeroed here by that tile's thread, or the stream's last word, below; a tile that
        // starts on a word boundary owes nobody a zeroed word -- and for trailing zero-bit tiles at a
        // word-aligned end that word lies past the shard.  The shard's very first word is always zeroed: a
        // short shard may end in it.)
        if (zero_words && (i == 0 || ((v & 31) && (v >> 5) != seam))) zero_words[v >> 5] = 0;
    }
    if (g == gridDim.x - 1 && tid == 0) {
        const unsigned long long total = prefix + group_total;
        out[n] = total;
        if (total_copy) *total_copy = total;  // next to the sweep flags
        if (zero_words && (total >> 5) != seam && (total & 31)) zero_words[total >> 5] = 0;  // (no open word at a word-aligned end)
        if (report_dst) {
            // the decode's report to the host, stored straight into pinned host memory (no copy
            // command between this kernel and the write kernel behind it): words 0..11 = the
            // sweeps' flags, final since the kernels before this one -- but for word 2, "the verification
            // failed", which is this kernel's own: what the groups before reported with their totals, and this one
            for (int k = 0; k < 12; ++k) report_dst[k] = report_src[k];
            if (verify_state) report_dst[2] = (bad_before || bad_any) ? 1u : 0u;
            report_dst[12] = static_cast<uint32_t>(total);
            report_dst[13] = static_cast<uint32_t>(total >> 32);
            __threadfence_system();
            // word 14: "the report is there" -- the host polls it (no event behind this kernel, no wake-up)
            __hip_atomic_store(report_dst + 14, report_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (g == 0 && header_words) {
        // the file header (header_src: behind the code table in its upload) goes into the output image once
        // the seam word is zeroed (this group's thread 0, above; nobody else writes it in this kernel)
        __syncthreads();
        for (uint32_t k = tid; k < header_words; k += 1024) zero_words[k] = header_src[k];
    }
}

// --------------------------------------------------------------------------------
// K4: variable-length code scatter
// --------------------------------------------------------------------------------
// Per round every lane looks its 16 symbols up in an LDS copy of the code table
// ({left-aligned code, length}), the workgroup scans the per-lane bit totals, and
// each lane then ORs its bits into an LDS ring of 32-bit big-endian words
// (ds_or_b32).  Completed words leave the ring as coalesced dword stores; the word a
// tile shares with its neighbour is merged with a global atomicOr.
struct RingFlush {
    uint32_t *ring;
    uint32_t *out32;             // global word 0 of this tile
    bool first_word_shared;      // tile does not start on a word boundary
};

template <uint32_t RING_WORDS>
__device__ __forceinline__ void flush_words(const RingFlush &f, uint32_t from, uint32_t to) {
    for (uint32_t i = from + threadIdx.x; i < to; i += BLOCK) {
        const uint32_t slot = i & (RING_WORDS - 1);
        const uint32_t v = __builtin_bswap32(f.ring[slot]);
        f.ring[slot] = 0;
        if (i == 0 && f.first_word_shared) atomicOr(f.out32, v);
#if ET_NT_STORE_K4
        else __builtin_nontemporal_store(v, f.out32 + i);
#else
        else f.out32[i] = v;
#endif
    }
}

template <uint32_t RING_WORDS>
__global__ __launch_bounds__(BLOCK) void k_encode_tiles(const uint8_t *__restrict__ base, uint64_t lo, uint64_t hi,
                                                        uint32_t rounds_per_tile, uint32_t n_tiles,
                                                        const unsigned long long *__restrict__ tile_off,
                                                        const uint2 *__restrict__ enc_table, uint32_t *__restrict__ out32) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[RING_WORDS];
    __shared__ __attribute__((aligned(16))) uint2 tab[256];
    __shared__ uint32_t scratch[2][4];
    const int tid = threadIdx.x;
    tab[tid] = enc_table[tid];
    for (uint32_t i = tid; i < RING_WORDS; i += BLOCK) ring[i] = 0;
    __syncthreads();

    const uint64_t tile_bytes = static_cast<uint64_t>(rounds_per_tile) * ROUND_BYTES;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const unsigned long long bit0 = tile_off[t];
        RingFlush f;
        f.ring = ring;
        f.out32 = out32 + (bit0 >> 5);
        f.first_word_shared = (bit0 & 31) != 0;
        uint32_t run = static_cast<uint32_t>(bit0 & 31);  // bit cursor relative to the tile's first word
        uint32_t flushed = 0;                             // ring words already stored
        const uint64_t t0 = static_cast<uint64_t>(t) * tile_bytes + static_cast<uint64_t>(tid) * 16;
        const bool interior = static_cast<uint64_t>(t) * tile_bytes >= lo && static_cast<uint64_t>(t + 1) * tile_bytes <= hi;

        Chunk cur = load_chunk_in_tile<ET_NT_LOAD_K4>(base, t0, lo, hi, interior);
        // (`cur` is THERE when the loop is entered, and `nxt` is taken at a point of its own further down: left to the compiler,
        // the round's first use of `cur` waited with vmcnt(0) -- loads come back in order, the first round's `cur` may still be
        // on its way at the loop's head, so every round waited for the `nxt` it had only just asked for: no prefetch at all, a
        // full memory latency per round)
#pragma unroll
        for (int k = 0; k < 4; ++k) ET_PIN(cur.w[k]);
        for (uint32_t r = 0; r < rounds_per_tile; ++r) {
            Chunk nxt;
            nxt.valid = 0;
            nxt.w[0] = nxt.w[1] = nxt.w[2] = nxt.w[3] = 0;
            if (r + 1 < rounds_per_tile) nxt = load_chunk_in_tile<ET_NT_LOAD_K4>(base, t0 + static_cast<uint64_t>(r + 1) * ROUND_BYTES, lo, hi, interior);

            // Symbols are merged before the append step: neighbours into pairs, pairs into
            // quads of la + lb + lc + ld bits.  A group only fails to fit 32 bits when long
            // codes meet (for Huffman codes of text ~1e-4 per quad); the wavefront then
            // falls back to pairs, then to single symbols, recomputing the table lookups
            // (rare) instead of keeping 16 codes + 16 lengths live: 8 wavefronts per SIMD.
            // Lanes holding the stream's first/last partial chunk take the last path.
#define ET_ENTRY(k_) tab[(cur.w[(k_) >> 2] >> (8 * ((k_) & 3))) & 0xffu]
            uint32_t qcode[4], qlen[4];
            uint32_t tot = 0;
            bool wide4 = false;  // some quad of this lane exceeds 32 bits (a pair that does makes its quad do so too: pairs are looked at only then)
            if (cur.valid == 0xffffu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint2 e0 = ET_ENTRY(4 * q), e1 = ET_ENTRY(4 * q + 1), e2 = ET_ENTRY(4 * q + 2), e3 = ET_ENTRY(4 * q + 3);
                    const uint32_t l01 = e0.y + e1.y, l23 = e2.y + e3.y;
                    const uint32_t c01 = e0.x | (e1.x >> (e0.y & 31u)), c23 = e2.x | (e3.x >> (e2.y & 31u));  // (garbage for a 32-bit first code with a non-empty second: that quad is wide)
                    qlen[q] = l01 + l23;
                    qcode[q] = c01 | (c23 >> (l01 & 31u));
                }
                // one maximum and one compare for the four quads, two three-operand adds for their total
                wide4 = max(max(qlen[0], qlen[1]), max(qlen[2], qlen[3])) > 32u;
                tot = (qlen[0] + qlen[1] + qlen[2]) + qlen[3];
            } else {  // bytes outside the stream carry no bits
                wide4 = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) qcode[q] = 0, qlen[q] = 64;  // (every quad "wide": the single-symbol path below)
                for (int k = 0; k < 16; ++k)
                    if ((cur.valid >> k) & 1u) tot += ET_ENTRY(k).y;
            }
            uint32_t round_total;
            const uint32_t excl = block_exclusive_scan(tot, scratch[r & 1], &round_total);

            // Append: `part` is the open output word (bits filled from the top, `fill` of
            // them); a piece goes in with one shift+or, and when the word completes it is
            // OR-ed into the ring and the bits that did not fit start the next one.
            const uint32_t pos = run + excl;
            uint32_t fill = pos & 31;
            uint32_t wbyte = ((pos >> 5) & (RING_WORDS - 1)) * 4;  // byte offset of the open word in the ring
            uint32_t part = 0;
            uint8_t *ring_bytes = reinterpret_cast<uint8_t *>(ring);
#define ET_APPEND(piece_, len_)                                                                   \
    do {                                                                                          \
        part |= (piece_) >> fill;                                                                 \
        const uint32_t nf_ = fill + (len_);                                                       \
        if (nf_ >= 32) {                                                                          \
            atomicOr(reinterpret_cast<uint32_t *>(ring_bytes + wbyte), part);                     \
            part = __builtin_amdgcn_alignbit((piece_), 0u, fill); /* what did not fit: piece << (32 - fill), 0 when fill == 0 -- one instruction */ \
            wbyte = (wbyte + 4) & (RING_WORDS * 4 - 1);                                           \
        }                                                                                         \
        fill = nf_ & 31;                                                                          \
    } while (0)
            if (__builtin_amdgcn_ballot_w64(wide4) == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ET_APPEND(qcode[q], qlen[q]);
            } else {
                // some lane's quad is longer than 32 bits (a long-tailed alphabet: most wavefront rounds have one).  Does any of
                // those lanes hold a PAIR longer than 32 bits, or a partial chunk?  Looked up again here, where it is rare,
                // rather than tested in every round.
                bool wide2 = cur.valid != 0xffffu;
                if (!wide2 && wide4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (qlen[q] > 32) {
                            const uint32_t l01 = ET_ENTRY(4 * q).y + ET_ENTRY(4 * q + 1).y;
                            wide2 = wide2 || l01 > 32 || qlen[q] - l01 > 32;
                        }
                    }
                }
                if (!__any(wide2)) {
                    // a quad POSITION in which some lane's quad is wide goes in as its two pairs, looked up again, for every lane of the
                    // wavefront (the same bits either way); the other positions go in whole.  (Decided per lane, a wavefront with one
                    // wide quad ran BOTH branches at all four positions: 12 appends and 16 lookups a round on a long-tailed alphabet.)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (__builtin_amdgcn_ballot_w64(qlen[q] > 32) == 0) {
                            ET_APPEND(qcode[q], qlen[q]);
                        } else {
                            const uint2 e0 = ET_ENTRY(4 * q), e1 = ET_ENTRY(4 * q + 1), e2 = ET_ENTRY(4 * q + 2), e3 = ET_ENTRY(4 * q + 3);
                            ET_APPEND(e0.x | (e1.x >> (e0.y & 31u)), e0.y + e1.y);
                            ET_APPEND(e2.x | (e3.x >> (e2.y & 31u)), e2.y + e3.y);
                        }
                    }
                } else {
                    for (int k = 0; k < 16; ++k) {
                        uint2 e = ET_ENTRY(k);
                        if (!((cur.valid >> k) & 1u)) e = make_uint2(0u, 0u);
                        ET_APPEND(e.x, e.y);
                    }
                }
            }
#undef ET_APPEND
#undef ET_ENTRY
            if (fill) atomicOr(reinterpret_cast<uint32_t *>(ring_bytes + wbyte), part);
            run += round_total;
            __syncthreads();
            // the next round's chunk is taken HERE, in front of this round's stores: loads and stores share the in-order vmcnt,
            // and behind the stores the wait for the chunk would be a wait for them as well
#pragma unroll
            for (int k = 0; k < 4; ++k) ET_PIN(nxt.w[k]);
            flush_words<RING_WORDS>(f, flushed, run >> 5);
            flushed = run >> 5;
            cur = nxt;
        }
        // The open word at the tile's end is shared with the next tile (or is the
        // stream's zero-padded last word).
        if ((run & 31) && tid == 0) {
            const uint32_t slot = flushed & (RING_WORDS - 1);
            atomicOr(f.out32 + flushed, __builtin_bswap32(ring[slot]));
            ring[slot] = 0;
        }
        __syncthreads();
    }
}

// Same job for code tables with a length above 32 bits.  The reference emits bit
// (data >> ((j-1) & 31)) & 1 for j = len..1 (encode.zig:311), i.e. the low
// ((len-1)&31)+1 bits of data followed by whole copies of data: a code is a sequence
// of pieces of at most 32 bits.  One symbol per lane per step; enc_table holds
// {data, len}.  Rare path (needs a Fibonacci-like histogram), kept simple.
__global__ __launch_bounds__(BLOCK) void k_encode_tiles_long(const uint8_t *__restrict__ base, uint64_t lo, uint64_t hi,
                                                             uint32_t rounds_per_tile, uint32_t n_tiles,
                                                             const unsigned long long *__restrict__ tile_off,
                                                             const uint2 *__restrict__ enc_table, uint32_t *__restrict__ out32) {
    constexpr uint32_t RING_WORDS = 4096;  // one step emits at most 256 * 255 bits = 2040 words
    __shared__ uint32_t ring[RING_WORDS];
    __shared__ uint2 tab[256];
    __shared__ uint32_t scratch[2][4];
    const int tid = threadIdx.x;
    tab[tid] = enc_table[tid];
    for (uint32_t i = tid; i < RING_WORDS; i += BLOCK) ring[i] = 0;
    __syncthreads();

    const uint64_t tile_bytes = static_cast<uint64_t>(rounds_per_tile) * ROUND_BYTES;
    const uint32_t steps = rounds_per_tile * (ROUND_BYTES / BLOCK);
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const unsigned long long bit0 = tile_off[t];
        RingFlush f;
        f.ring = ring;
        f.out32 = out32 + (bit0 >> 5);
        f.first_word_shared = (bit0 & 31) != 0;
        uint32_t run = static_cast<uint32_t>(bit0 & 31), flushed = 0;
        for (uint32_t s = 0; s < steps; ++s) {
            const uint64_t p = static_cast<uint64_t>(t) * tile_bytes + static_cast<uint64_t>(s) * BLOCK + tid;
            uint32_t data = 0, len = 0;
            if (p >= lo && p < hi) {
                const uint2 e = tab[base[p]];
                data = e.x;
                len = e.y;
            }
            uint32_t step_total;
            const uint32_t excl = block_exclusive_scan(len, scratch[s & 1], &step_total);
            uint32_t pos = run + excl;
            uint32_t remaining = len;
            while (remaining) {
                const uint32_t piece = ((remaining - 1) & 31u) + 1;  // bits (remaining-1)&31 .. 0 of data
                const uint32_t bits = (piece == 32) ? data : (data & ((1u << piece) - 1u));
                const unsigned long long placed = (static_cast<unsigned long long>(bits) << (32 - piece)) << (32 - (pos & 31));
                atomicOr(&ring[(pos >> 5) & (RING_WORDS - 1)], static_cast<uint32_t>(placed >> 32));
                if (static_cast<uint32_t>(placed)) atomicOr(&ring[((pos >> 5) + 1) & (RING_WORDS - 1)], static_cast<uint32_t>(placed));
                pos += piece;
                remaining -= piece;
            }
            run += step_total;
            __syncthreads();
            flush_words<RING_WORDS>(f, flushed, run >> 5);
            flushed = run >> 5;
        }
        if ((run & 31) && tid == 0) {
            const uint32_t slot = flushed & (RING_WORDS - 1);
            atomicOr(f.out32 + flushed, __builtin_bswap32(ring[slot]));
            ring[slot] = 0;
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------
// Decode
// --------------------------------------------------------------------------------
// The .et body carries no block index, so workgroups cannot know where codewords
// start.  The bitstream (addressed from a 4-byte aligned base) is cut into
// subsequences of SUB_BITS bits, one per lane.  State per subsequence, packed in a
// u32: start (bits past the subsequence's first bit at which its first codeword
// begins), exit (same for the following subsequence, as implied by `start`) and the
// number of codewords that begin inside it.  k_dec_sync iterates start[i+1] =
// exit[i] to a fixed point; Huffman codes self-synchronise, so a wrong guess heals
// within a few codewords and the fixed point is reached after two or three sweeps.
// The unique fixed point with start[0] = the true first bit is the true parse.
//
// Inner loop (walk_subsequence): each lane keeps the next 32..64 stream bits in a
// 64-bit register and refills it with ONE LDS word per 32 bits consumed; a lookup of
// the next lut_bits bits in an LDS table yields up to TWO symbols per step.  The
// staged bitstream is padded by one word per 32 so that lanes, which read at a stride
// of SUB_BITS / 32 = 8 words, fall on different LDS banks.
__device__ __forceinline__ uint32_t phys(uint32_t logical_word) { return logical_word + (logical_word >> 5); }

// Dynamic LDS carve (all offsets multiples of 16 bytes).
struct DecodeSmem {
    uint32_t *sdata;   // DEC_SDATA_WORDS
    uint32_t *lut;     // 1 << lut_bits
    uint16_t *sub;     // n_sub << sub_bits
    uint8_t *sym_len;  // 256
    uint32_t *exits;   // BLOCK
    uint32_t *scratch; // 8 (scan scratch [0..3], flag [4])
    uint8_t *stage;    // DEC_STAGE_BYTES (write kernel only)
};

extern __shared__ __attribute__((aligned(16))) uint8_t dec_smem_raw[];

__device__ __forceinline__ uint32_t sub_words(const DecodeTables &tb) { return (((tb.n_sub << tb.sub_bits) + 7u) & ~7u) / 2; }

// WITH_EXITS: the sync kernels exchange exits through LDS; the write kernel does not and
// must stay under 32 KiB (5 workgroups per CU).
template <bool WITH_EXITS = true>
__device__ __forceinline__ DecodeSmem carve_decode_smem(const DecodeTables &tb) {
    DecodeSmem m;
    m.sdata = reinterpret_cast<uint32_t *>(dec_smem_raw);
    m.lut = m.sdata + DEC_SDATA_WORDS;
    m.sub = reinterpret_cast<uint16_t *>(m.lut + (1u << tb.lut_bits));
    m.sym_len = reinterpret_cast<uint8_t *>(m.lut + (1u << tb.lut_bits) + sub_words(tb));
    m.exits = m.lut + (1u << tb.lut_bits) + sub_words(tb) + 64;
    m.scratch = m.exits + (WITH_EXITS ? BLOCK : 0);
    m.stage = reinterpret_cast<uint8_t *>(m.scratch + 8);
    return m;
}

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

// Staging.  A decode workgroup handles chunks of SYNC_CHUNK / WRITE_CHUNK consecutive
// 8 KiB blocks (et_kernels.h): it copies the lookup tables into LDS once and then walks
// the blocks; while it works on one block, the next block's words are already in flight
// into registers
// (prefetch_block) and are written to LDS (commit_block) only when the current block
// is done with the staging area.  LDS holds host-order words whose numeric MSB is the
// first stream bit; logical word i of the stage = stream word
// first_word - DEC_FRONT_WORDS + i (zero before and after the stream).
constexpr int DEC_WORDS_PER_THREAD = (DEC_STAGED_WORDS + BLOCK - 1) / BLOCK;

struct Prefetch {
    uint32_t w[DEC_WORDS_PER_THREAD];
};

__device__ __forceinline__ void stage_tables(const DecodeSmem &m, const DecodeTables &tb) {
    const uint32_t n_lut = 1u << tb.lut_bits;
    for (uint32_t i = threadIdx.x; i < n_lut; i += BLOCK) m.lut[i] = tb.lut[i];
    const uint32_t n_sub_words = sub_words(tb);
    for (uint32_t i = threadIdx.x; i < n_sub_words; i += BLOCK) reinterpret_cast<uint32_t *>(m.sub)[i] = reinterpret_cast<const uint32_t *>(tb.sub)[i];
    if (threadIdx.x < 64) reinterpret_cast<uint32_t *>(m.sym_len)[threadIdx.x] = reinterpret_cast<const uint32_t *>(tb.sym_len)[threadIdx.x];
}

// front_ok: the DEC_FRONT_WORDS words BEFORE `words` are readable stream bytes (a rank's
// range of a stream decoded on several GPUs); otherwise they read as zero.
__device__ __forceinline__ void prefetch_block(Prefetch &p, const uint32_t *__restrict__ words, uint64_t block, uint64_t n_bytes,
                                               bool front_ok = false) {
    const uint64_t first_word = block * DEC_BLOCK_WORDS;
    // workgroup-uniform: every staged word lies wholly inside the stream
    const bool interior = (first_word >= DEC_FRONT_WORDS || front_ok) && (first_word + DEC_STAGED_WORDS - DEC_FRONT_WORDS) * 4 <= n_bytes;
#pragma unroll
    for (int j = 0; j < DEC_WORDS_PER_THREAD; ++j) {
        const uint32_t i = j * BLOCK + threadIdx.x;
        if (i < DEC_STAGED_WORDS) {
            const long long w = static_cast<long long>(first_word + i) - DEC_FRONT_WORDS;  // negative: before `words`
            if (interior) p.w[j] = __builtin_bswap32(words[w]);
            else if (w < 0) p.w[j] = front_ok ? __builtin_bswap32(words[w]) : 0u;
            else p.w[j] = load_be32_guarded(words, static_cast<uint64_t>(w), n_bytes);
        }
    }
}

__device__ __forceinline__ void commit_block(const DecodeSmem &m, const Prefetch &p) {
#pragma unroll
    for (int j = 0; j < DEC_WORDS_PER_THREAD; ++j) {
        const uint32_t i = j * BLOCK + threadIdx.x;
        if (i < DEC_STAGED_WORDS) m.sdata[phys(i)] = p.w[j];
    }
}

struct SubResult {
    uint32_t start_rel;
    uint32_t exit_rel;
    uint32_t count;
};

// Escape of a table step: the first code at the top of `window` is longer than the
// first-level table (or nothing starts here).  Returns (len << 8) | sym, 0 = no code.
__device__ __forceinline__ uint32_t long_code(const DecodeSmem &m, const DecodeTables &tb, uint32_t e, uint32_t window) {
    uint32_t hit = 0;
    if ((e >> LUT_SUB_SHIFT) & 1u)
        hit = m.sub[((e & 0xffu) << tb.sub_bits) | ((window << tb.lut_bits) >> (32 - tb.sub_bits))];
    if (hit == 0) {  // deeper than both tables (or no table slot left): search the list in global memory
        for (uint32_t i = 0; i < tb.n_long; ++i) {
            const uint32_t meta = tb.longc[2 * i + 1], l = meta >> 8;
            if (((window ^ tb.longc[2 * i]) >> (32 - l)) == 0) {
                hit = meta;
                break;
            }
        }
    }
    return hit;
}

// Walk the codewords that begin inside subsequence `sub` of the staged block.
// Positions are bits from the first STAGED bit (DEC_WARMUP_BITS before the block).
//   WARM: start DEC_WARMUP_BITS before the subsequence and run in (nothing counted);
//         the first codeword boundary at or after the subsequence's first bit becomes
//         start_rel.  Otherwise start at the given start_rel.
//   `lim` = stream end (same origin), tested only when CHECK_LIM (the block(s) the
//         stream ends in): a test with a `break` in the hot loop costs ~25 % everywhere.
// A symbol belongs to the subsequence in which it BEGINS.
//
// Every stretch [pos, limit) is walked in two phases: MULTI steps while the whole
// lut_bits window lies before `limit` -- one lookup yields up to three symbols, all of
// which therefore begin before `limit` -- then SINGLE steps (first symbol of the entry,
// its length from sym_len[]) for the last < lut_bits bits.  The loop bodies have no
// divergent branch except the escape for codes longer than the table (~0.1 % of
// symbols): one scalar unit serves the four SIMDs of a CU, and exec-mask bookkeeping
// was the first bottleneck.  The stream window is the 64-bit pair {r0, r1} read at bit
// `sh` in [1, 32] with one v_alignbit_b32; r2 holds the word after it.
//
// WRITE: 0 = count only, 1 = store every symbol at stage[stage_pos + index] (the
// caller guarantees the whole range is inside the stage), 2 = store only indices in
// [stage_lo, stage_hi).
template <int WRITE, bool CHECK_LIM, bool WARM>
__device__ __forceinline__ SubResult walk_subsequence(const DecodeSmem &m, const DecodeTables &tb, uint32_t sub,
                                                      uint32_t start_rel, uint32_t lim, uint32_t stage_pos, uint32_t stage_lo,
                                                      uint32_t stage_hi) {
    const uint32_t begin = DEC_WARMUP_BITS + sub * SUB_BITS;
    const uint32_t end = begin + SUB_BITS;
    const uint32_t lut_bits = tb.lut_bits;
    const uint32_t idx_shift = 32 - lut_bits;
    uint32_t pos = WARM ? begin - DEC_WARMUP_BITS : begin + start_rel;
    uint32_t count = 0;
    bool off_stream = false;
    SubResult res;
    res.start_rel = start_rel;

    const uint32_t k0 = pos >> 5;
    uint32_t sh = pos & 31;
    const uint32_t k1 = k0 + (sh != 0);
    uint32_t r0 = m.sdata[phys(k0)];
    uint32_t r1 = m.sdata[phys(k1)];
    uint32_t next_word = k1 + 1;
    uint32_t r2 = m.sdata[phys(next_word)];
    sh = sh ? sh : 32;  // sh == 32: the window is exactly r1

#define ET_ADVANCE(len_)                          \
    do {                                          \
        pos += (len_);                            \
        sh += (len_);                             \
        const bool rotate_ = sh > 32;             \
        r0 = rotate_ ? r1 : r0;                   \
        r1 = rotate_ ? r2 : r1;                   \
        sh = rotate_ ? sh - 32 : sh;              \
        next_word += rotate_;                     \
        r2 = m.sdata[phys(next_word)];            \
    } while (0)

// MULTI: whole-window steps while pos + lut_bits <= limit_ (never past the stream end:
// the caller clamps limit_).  SINGLE: one symbol per step while pos < limit_.
#define ET_WALK(limit_, COUNTING)                                                                         \
    do {                                                                                                  \
        const uint32_t multi_until_ = (CHECK_LIM && lim < (limit_)) ? lim : (limit_);                     \
        while (pos + lut_bits <= multi_until_) {                                                          \
            const uint32_t window_ = __builtin_amdgcn_alignbit(r0, r1, 32 - sh);                          \
            const uint32_t e_ = m.lut[window_ >> idx_shift];                                              \
            uint32_t n_ = (e_ >> LUT_N_SHIFT) & 3u, len_ = (e_ >> LUT_LEN_SHIFT) & 15u, syms_ = e_;       \
            if (n_ == 0) {                                                                                \
                const uint32_t hit_ = long_code(m, tb, e_, window_);                                      \
                len_ = hit_ ? (hit_ >> 8) : 1u; /* no code: resynchronise bit by bit */                  \
                syms_ = hit_ & 0xffu;                                                                     \
                n_ = hit_ ? 1u : 0u;                                                                      \
                if (CHECK_LIM && pos + len_ > lim) { off_stream = true; break; }                          \
            }                                                                                             \
            if (COUNTING) {                                                                               \
                if (WRITE == 1) {                                                                         \
                    const uint32_t o_ = stage_pos + count;                                                \
                    /* second byte first, at o + (n == 2); then the first symbol at o: a one-symbol */   \
                    /* step stores twice to the same byte and the later store (the symbol) wins     */   \
                    m.stage[o_ + (n_ >> 1)] = static_cast<uint8_t>(syms_ >> 8);                           \
                    m.stage[o_] = static_cast<uint8_t>(syms_);                                            \
                    /* the write kernel's table holds at most two symbols per entry (DEC_WRITE_SYMS) */  \
                } else if (WRITE == 2) {                                                                  \
                    for (uint32_t j_ = 0; j_ < n_; ++j_) {                                                \
                        const uint32_t o_ = stage_pos + count + j_;                                       \
                        if (o_ >= stage_lo && o_ < stage_hi) m.stage[o_ - stage_lo] = static_cast<uint8_t>(syms_ >> (8 * j_)); \
                    }                                                                                     \
                }                                                                                         \
                count += n_;                                                                              \
            }                                                                                             \
            ET_ADVANCE(len_);                                                                             \
        }                                                                                                 \
        while (!off_stream && pos < (limit_)) {                                                           \
            const uint32_t window_ = __builtin_amdgcn_alignbit(r0, r1, 32 - sh);                          \
            const uint32_t e_ = m.lut[window_ >> idx_shift];                                              \
            uint32_t n_ = (e_ >> LUT_N_SHIFT) & 3u, syms_ = e_ & 0xffu;                                   \
            uint32_t len_ = m.sym_len[syms_];                                                             \
            if (n_ == 0) {                                                                                \
                const uint32_t hit_ = long_code(m, tb, e_, window_);                                      \
                len_ = hit_ ? (hit_ >> 8) : 1u;                                                           \
                syms_ = hit_ & 0xffu;                                                                     \
                n_ = hit_ ? 1u : 0u;                                                                      \
            } else {                                                                                      \
                n_ = 1;                                                                                   \
            }                                                                                             \
            if (CHECK_LIM && pos + len_ > lim) { off_stream = true; break; }                              \
            if (COUNTING) {                                                                               \
                if (WRITE == 1) {                                                                         \
                    if (n_) m.stage[stage_pos + count] = static_cast<uint8_t>(syms_);                     \
                } else if (WRITE == 2) {                                                                  \
                    const uint32_t o_ = stage_pos + count;                                                \
                    if (n_ && o_ >= stage_lo && o_ < stage_hi) m.stage[o_ - stage_lo] = static_cast<uint8_t>(syms_); \
                }                                                                                         \
                count += n_;                                                                              \
            }                                                                                             \
            ET_ADVANCE(len_);                                                                             \
        }                                                                                                 \
    } while (0)

    if (WARM) {
        ET_WALK(begin, false);
        res.start_rel = off_stream ? 0u : pos - begin;  // the stream may end before this subsequence
    }
    if (!off_stream) ET_WALK(end, true);
#undef ET_WALK
#undef ET_ADVANCE
    // Ran off the stream (the code that begins at `pos` is cut by the stream's end): nothing further
    // begins here OR in the few bits the stream may still have in the next subsequence -- the exit
    // points at the stream's end, where the next walk stops at once.  (An exit of 0 let the next lane
    // decode the cut code's tail as if a codeword began there; `lim - end` < 32, the cut code's length.)
    res.exit_rel = off_stream ? (lim > end ? lim - end : 0u) : pos - end;
    res.count = count;
    return res;
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

// D1.  FIRST sweep: every subsequence runs in over the DEC_WARMUP_BITS before it (the
// stream's very first one starts at first_bit, which is exact); lanes whose run-in
// disagrees with their predecessor's exit are re-walked until the workgroup is
// consistent.  Later sweeps: a block whose predecessor's exit still equals the start
// its lane 0 used is skipped; otherwise its local fixed point is redone from the
// stored state and *changed is raised.  blk_exit[b] may be read by the workgroup
// handling block b+1 in the SAME launch without ordering: either value is a legal
// intermediate state, and a launch that ends with *changed == 0 has seen every block
// consistent with its predecessor.
template <bool FIRST>
__global__ __launch_bounds__(BLOCK) void k_dec_sync(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t first_bit,
                                                    uint64_t n_subs, uint32_t n_blocks, DecodeTables tb,
                                                    uint32_t *__restrict__ sub_state, uint32_t *__restrict__ blk_exit,
                                                    uint32_t *__restrict__ blk_count, uint32_t *__restrict__ changed,
                                                    uint32_t *__restrict__ ticket, uint32_t max_trips, uint32_t flags) {
    // flags: DEC_SPECIAL_ONLY: handle only the special blocks (first, last one or two; one
    // per workgroup), the others belong to k_dec_sync_reg.  (DEC_HAVE_START: first_bit is the exact start of subsequence 0; DEC_FRONT_OK:
    // the words before `words` belong to the stream) differ from {1, 0} only for a rank's
    // range of a stream decoded on several GPUs.
    const bool have_start = flags & DEC_HAVE_START, front_ok = flags & DEC_FRONT_OK;
    const DecodeSmem m = carve_decode_smem(tb);
    const int tid = threadIdx.x;
    bool tables_staged = FIRST;  // repair sweeps copy the tables only if a block needs repair
    if (FIRST) stage_tables(m, tb);
    Prefetch pf;
    uint32_t round = 0;
    // Blocks are handed out in chunks of SYNC_CHUNK consecutive blocks (tables staged
    // once per workgroup, next block prefetched inside a chunk): either one chunk per
    // workgroup, dispatched by the hardware, or (SYNC_TICKET) through a ticket counter
    // to a grid sized to the device, so that a workgroup that becomes resident late --
    // or never -- costs nothing.
    for (bool first_trip = true;; first_trip = false) {
        uint64_t b0;
        if (SYNC_TICKET && !(flags & DEC_SPECIAL_ONLY)) {
            __syncthreads();  // tables staged (first trip); everybody is done with scratch[7]
            if (tid == 0) m.scratch[7] = atomicAdd(ticket, SYNC_CHUNK);
            __syncthreads();
            b0 = m.scratch[7];
        } else {
            if (!first_trip) break;
            b0 = static_cast<uint64_t>(blockIdx.x) * SYNC_CHUNK;
            __syncthreads();
        }
        if (flags & DEC_SPECIAL_ONLY) {
            if (!first_trip) break;
            if (flags & DEC_SPECIAL_SUPER) {
                b0 = special_candidate2(blockIdx.x, n_blocks);
                if (b0 >= n_blocks || super_interior(b0 >> 1, n_bytes, n_blocks)) break;
            } else {
                b0 = special_candidate(blockIdx.x, n_blocks);
                if (b0 >= n_blocks || !special_block(b0, n_bytes)) break;
            }
        }
        if (b0 >= n_blocks) break;
        const uint64_t b1 = (flags & DEC_SPECIAL_ONLY) ? b0 + 1 : (b0 + SYNC_CHUNK < n_blocks ? b0 + SYNC_CHUNK : n_blocks);
        if (FIRST) prefetch_block(pf, words, b0, n_bytes, front_ok);
    for (uint64_t b = b0; b < b1; ++b, ++round) {
        const uint64_t sub_g = b * BLOCK + tid;
        const bool live = sub_g < n_subs;
        uint32_t start, exit_rel = 0, count = 0, first_cand = 0;
        bool need, warm = false;
        if (FIRST) {
            start = first_bit;  // exact for the stream's first subsequence; every other one runs in
            warm = sub_g != 0 || !have_start;
            need = live;
            commit_block(m, pf);
            if (b + 1 < b1) prefetch_block(pf, words, b + 1, n_bytes, front_ok);
        } else {
            const uint32_t st = live ? sub_state[sub_g] : 0u;
            start = st & 0xffu;
            exit_rel = (st >> 8) & 0xffu;
            count = st >> 16;
            need = false;
            uint32_t *flag = m.scratch + 4 + (round & 1);
            if (tid == 0) {
                const uint32_t in = (b == 0) ? (have_start ? first_bit : start) : blk_exit[b - 1];
                // a range whose start is still unknown and whose first block gave up: run in again
                warm = b == 0 && !have_start && start == 0xffu;
                need = warm || in != start;
                first_cand = in;
                *flag = need;
            }
            __syncthreads();
            if (!*flag) continue;
            if (tid == 0) *changed = 1;
            if (!tables_staged) {
                stage_tables(m, tb);
                tables_staged = true;
            }
            prefetch_block(pf, words, b, n_bytes, front_ok);
            commit_block(m, pf);
        }
        __syncthreads();

        const uint32_t lim = block_limit(n_bytes, b);
        // `start` is always the start that (exit_rel, count) belong to; `cand` is the start
        // the predecessor's exit asks for.  Only a walk moves cand into start, so whatever
        // is stored -- also after giving up -- is self-consistent per lane.
        uint32_t cand = (!FIRST && tid == 0) ? first_cand : start;
        for (uint32_t trip = 0;; ++trip) {
            if (trip == max_trips) {
                // Codes that do not self-synchronise (near-fixed-length ones) would crawl
                // one lane per trip: give up on this block for now.  The first sweep counts
                // such blocks so that the host can pick the exhaustive path (k_dec_maps
                // ...); a start of 0xff makes any later sweep redo the block.
                if (tid == 0) {
                    if (FIRST) atomicAdd(changed + 1, 1u);
                    else *changed = 1;
                    start = 0xffu;
                }
                break;
            }
            if (need) {
                SubResult r;
                if (lim != 0xffffffffu) {  // workgroup-uniform: the stream ends in this block
                    r = warm ? walk_subsequence<0, true, true>(m, tb, tid, 0, lim, 0, 0, 0) : walk_subsequence<0, true, false>(m, tb, tid, cand, lim, 0, 0, 0);
                } else if (warm) {
                    r = walk_subsequence<0, false, true>(m, tb, tid, 0, lim, 0, 0, 0);
                } else {
                    r = walk_subsequence<0, false, false>(m, tb, tid, cand, lim, 0, 0, 0);
                }
                start = r.start_rel;
                exit_rel = r.exit_rel;
                count = r.count;
                warm = false;
            }
            m.exits[tid] = exit_rel;
            __syncthreads();
            need = false;
            if (tid > 0 && live) {
                cand = m.exits[tid - 1];
                need = cand != start;
            }
            if (!__syncthreads_or(need)) break;
        }
        if (live) sub_state[sub_g] = start | (exit_rel << 8) | (count << 16);
        uint32_t total;
        block_exclusive_scan(live ? count : 0u, m.scratch, &total);
        if (tid == 0) blk_count[b] = total;
        // exit of the last live subsequence of this block
        const uint64_t last_live = (n_subs - b * BLOCK >= BLOCK) ? BLOCK - 1 : (n_subs - b * BLOCK - 1);
        if (tid == static_cast<int>(last_live)) blk_exit[b] = exit_rel;
    }
    }
}

// ---- exhaustive synchronisation -----------------------------------------------------
// For codes that barely self-synchronise the fixed point above degenerates to one
// subsequence per trip.  The bounded alternative: every subsequence computes its exit
// for EVERY possible start offset (n_starts = longest code length of them), which turns
// "start of i+1 = exit of i" into a composition of small maps; maps compose
// associatively, so blocks, then groups of 256 blocks, are resolved by short
// sequential chains over LDS-resident maps instead of sweeps over the stream.
// Cost: n_starts + 1 walks per subsequence, independent of the data.

// X1: lane maps (stride map_stride bytes per subsequence) and the block's composed map.
__global__ __launch_bounds__(BLOCK) void k_dec_maps(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t first_bit,
                                                    uint64_t n_subs, DecodeTables tb, uint32_t n_starts, uint32_t map_stride,
                                                    uint8_t *__restrict__ lane_maps, uint8_t *__restrict__ blk_maps, uint32_t special_only,
                                                    uint32_t have_start) {
    // have_start: the first subsequence starts exactly at first_bit (a stream's beginning, or a
    // range whose start is known): its map is constant.  Otherwise (a range of a stream split
    // over GPUs, et_decode_range_maps) it is a subsequence like any other.
    const DecodeSmem m = carve_decode_smem(tb);
    const int tid = threadIdx.x;
    const uint32_t n_blocks_all = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    // special_only: grid 3, the stream's first/last blocks (k_dec_maps_reg has the rest)
    const uint64_t b = special_only ? special_candidate(blockIdx.x, n_blocks_all) : blockIdx.x;
    if (special_only && (b >= n_blocks_all || !special_block(b, n_bytes))) return;
    stage_tables(m, tb);
    Prefetch pf;
    prefetch_block(pf, words, b, n_bytes);
    commit_block(m, pf);
    __syncthreads();

    const uint64_t sub_g = b * BLOCK + tid;
    const bool live = sub_g < n_subs;
    const uint32_t lim = block_limit(n_bytes, b);
    uint8_t *maps = m.stage;  // [BLOCK][32]
    for (uint32_t p = 0; p < 32; ++p) {
        uint32_t e = 0;
        if (live && (p < n_starts || (sub_g == 0 && have_start))) {
            const uint32_t st = (sub_g == 0 && have_start) ? first_bit : p;  // the stream's first subsequence has one start, whatever comes in
            const SubResult r = lim != 0xffffffffu ? walk_subsequence<0, true, false>(m, tb, tid, st, lim, 0, 0, 0)
                                                   : walk_subsequence<0, false, false>(m, tb, tid, st, lim, 0, 0, 0);
            e = r.exit_rel;
        }
        maps[tid * 32 + p] = static_cast<uint8_t>(e);
        if (p + 1 >= n_starts && !(b == 0 && have_start)) break;  // block 0 fills all 32 entries for its first lane
    }
    __syncthreads();
    if (live) {
        for (uint32_t k = 0; k < map_stride; k += 8)
            *reinterpret_cast<uint2 *>(lane_maps + sub_g * map_stride + k) = *reinterpret_cast<const uint2 *>(maps + tid * 32 + k);
    }
    const uint32_t n_live = static_cast<uint32_t>(n_subs - b * BLOCK >= BLOCK ? BLOCK : n_subs - b * BLOCK);
    if (tid < 32) {
        uint32_t sidx = tid;
        if (static_cast<uint32_t>(tid) < n_starts || (b == 0 && have_start))
            for (uint32_t i = 0; i < n_live; ++i) sidx = maps[i * 32 + sidx];
        blk_maps[b * 32 + tid] = static_cast<uint8_t>(sidx);
    }
}

// X2: chains over maps.  k_dec_compose composes `count` consecutive 32-byte maps per
// workgroup (level up); k_dec_chain walks them with a known input and writes the input
// of every map (level down).  Both stage up to 256 maps in LDS.
__global__ __launch_bounds__(BLOCK) void k_dec_compose(const uint8_t *__restrict__ maps_in, uint32_t n_maps, uint8_t *__restrict__ maps_out) {
    __shared__ __attribute__((aligned(16))) uint8_t sm[256 * 32];
    const uint32_t g = blockIdx.x, first = g * 256;
    const uint32_t count = n_maps - first < 256 ? n_maps - first : 256;
    for (uint32_t i = threadIdx.x; i < count * 2; i += BLOCK)
        reinterpret_cast<uint4 *>(sm)[i] = reinterpret_cast<const uint4 *>(maps_in + static_cast<uint64_t>(first) * 32)[i];
    __syncthreads();
    if (threadIdx.x < 32) {
        uint32_t sidx = threadIdx.x;
        for (uint32_t i = 0; i < count; ++i) sidx = sm[i * 32 + sidx];
        maps_out[g * 32 + threadIdx.x] = static_cast<uint8_t>(sidx);
    }
}

// inputs[i] = input of map i, for the maps [g*256, g*256+256) given the group's input
// (group_in[g], or first_in when group_in is null: the single top-level workgroup then
// loops over all groups).
__global__ __launch_bounds__(BLOCK) void k_dec_chain(const uint8_t *__restrict__ maps, uint32_t n_maps, const uint8_t *__restrict__ group_in,
                                                     uint32_t first_in, uint8_t *__restrict__ inputs) {
    __shared__ __attribute__((aligned(16))) uint8_t sm[256 * 32];
    __shared__ uint8_t s_in[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = group_in ? group_in[blockIdx.x] : first_in;
    const uint32_t n_groups_here = group_in ? 1 : (n_maps + 255) / 256;
    for (uint32_t gg = 0; gg < n_groups_here; ++gg) {
        const uint32_t g = group_in ? blockIdx.x : gg, first = g * 256;
        const uint32_t count = n_maps - first < 256 ? n_maps - first : 256;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < count * 2; i += BLOCK)
            reinterpret_cast<uint4 *>(sm)[i] = reinterpret_cast<const uint4 *>(maps + static_cast<uint64_t>(first) * 32)[i];
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sidx = carry;
            for (uint32_t i = 0; i < count; ++i) {
                s_in[i] = static_cast<uint8_t>(sidx);
                sidx = sm[i * 32 + sidx];
            }
            carry = sidx;
        }
        __syncthreads();
        if (threadIdx.x < count) inputs[first + threadIdx.x] = s_in[threadIdx.x];
    }
}

// X3: with every block's input start known, resolve the lanes from the stored lane
// maps, then one counting walk per lane.
__global__ __launch_bounds__(BLOCK) void k_dec_resolve(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t first_bit,
                                                       uint64_t n_subs, DecodeTables tb, uint32_t map_stride,
                                                       const uint8_t *__restrict__ lane_maps, const uint8_t *__restrict__ blk_in,
                                                       uint32_t *__restrict__ sub_state, uint32_t *__restrict__ blk_exit,
                                                       uint32_t *__restrict__ blk_count, uint32_t special_only, uint32_t const_first) {
    const DecodeSmem m = carve_decode_smem(tb);
    const int tid = threadIdx.x;
    const uint32_t n_blocks_all = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    const uint64_t b = special_only ? special_candidate(blockIdx.x, n_blocks_all) : blockIdx.x;  // as k_dec_maps
    if (special_only && (b >= n_blocks_all || !special_block(b, n_bytes))) return;
    stage_tables(m, tb);
    Prefetch pf;
    prefetch_block(pf, words, b, n_bytes);
    commit_block(m, pf);
    const uint64_t sub_g = b * BLOCK + tid;
    const bool live = sub_g < n_subs;
    uint8_t *maps = m.stage;  // [BLOCK][32]
    if (live)
        for (uint32_t k = 0; k < map_stride; k += 8)
            *reinterpret_cast<uint2 *>(maps + tid * 32 + k) = *reinterpret_cast<const uint2 *>(lane_maps + sub_g * map_stride + k);
    __syncthreads();
    const uint32_t n_live = static_cast<uint32_t>(n_subs - b * BLOCK >= BLOCK ? BLOCK : n_subs - b * BLOCK);
    if (tid == 0) {
        uint32_t sidx = b == 0 ? first_bit : blk_in[b];
        for (uint32_t i = 0; i < n_live; ++i) {
            m.exits[i] = sidx;
            // the stream's first subsequence has a constant map, of which only the first
            // map_stride entries were stored; first_bit may lie beyond them
            sidx = maps[i * 32 + ((b == 0 && i == 0 && const_first) ? 0u : sidx)];
        }
    }
    __syncthreads();
    const uint32_t lim = block_limit(n_bytes, b);
    uint32_t start = 0, exit_rel = 0, count = 0;
    if (live) {
        start = m.exits[tid];
        const SubResult r = lim != 0xffffffffu ? walk_subsequence<0, true, false>(m, tb, tid, start, lim, 0, 0, 0)
                                               : walk_subsequence<0, false, false>(m, tb, tid, start, lim, 0, 0, 0);
        exit_rel = r.exit_rel;
        count = r.count;
        sub_state[sub_g] = start | (exit_rel << 8) | (count << 16);
    }
    uint32_t total;
    block_exclusive_scan(live ? count : 0u, m.scratch, &total);
    if (tid == 0) blk_count[b] = total;
    if (tid == static_cast<int>(n_live - 1)) blk_exit[b] = exit_rel;
}

// ---- register-window walk (interior blocks) ----------------------------------------------
// The kernels above keep the block's bitstream in LDS and move a three-register window
// over it (rotation selects, padded-address arithmetic, one LDS read per step).  For
// blocks that lie wholly inside the stream -- all but the first and the last one or two
// -- a lane instead loads the 9 (13 with the run-in) words of its subsequence straight
// into registers and the walk is unrolled PER WORD: iteration w reads the window with one
// v_alignbit_b32 from the fixed register pair (word w-1, word w) while the bit offset
// `sh` stays in [1, 32], then sh -= 32.  No rotation, no window addressing, no bitstream
// in LDS.  Only the last word of a stretch needs the multi/single phase split.  The walk
// state is one packed register and table entries are added to it (walk_steps, walk_write).
// The slow path's search for a code longer than the first-level table (as long_code, with
// every table in global memory).
__device__ __forceinline__ uint32_t long_code_flat(const uint16_t *sub, const uint32_t *longc, uint32_t n_long, uint32_t bits, uint32_t e,
                                                  uint32_t window) {
    const uint32_t lut_bits = bits & 0xffu, sub_bits = bits >> 8;
    uint32_t hit = 0;
    if ((e >> LUT_SUB_SHIFT) & 1u) hit = sub[((e & 0xffu) << sub_bits) | ((window << lut_bits) >> (32 - sub_bits))];
    if (hit == 0) {
        for (uint32_t i = 0; i < n_long; ++i) {
            const uint32_t meta = longc[2 * i + 1], l = meta >> 8;
            if (((window ^ longc[2 * i]) >> (32 - l)) == 0) {
                hit = meta;
                break;
            }
        }
    }
    return hit;
}

constexpr int RW_WORDS = 13;  // W[j] = stream word 8 * sub - 4 + j (host order): 4 run-in words, 8 own, 1 beyond

__host__ __device__ __forceinline__ uint32_t step_table_words(const DecodeTables &tb) {
    return ((1u << tb.step_bits) + (tb.n_step_sub << tb.step_sub_bits) + 3u) & ~3u;
}

typedef __attribute__((address_space(3))) uint8_t lds_u8;

// Kernel-argument form of a step table (DecodeTables::steps ...): the table in global
// memory, its size in words (both levels, multiple of 4), and the device copy of the
// DecodeTables for the slow path.
struct StepTableArgs {
    const uint32_t *table;
    const DecodeTables *slow;
    uint32_t words, step_bits, sub_bits;
};
static inline StepTableArgs step_table_args(const DecodeTables &tb) {
    return StepTableArgs{tb.steps, tb.dev_copy, step_table_words(tb), tb.step_bits, tb.step_sub_bits};
}

// What the step walks need besides their LDS tables: by value only what a step touches;
// the tables of the slow path stay behind a pointer to a device copy of the DecodeTables
// (fewer SGPRs live across the walk: the kernels are SGPR-limited to 7 wavefronts per SIMD
// otherwise, and 8 is worth 9 % in k_dec_sync_reg).
struct StepWalk {
    const uint32_t *steps;      // LDS: first level, second level behind it
    const DecodeTables *slow;   // global memory
    uint32_t idx_shift, step_bits, sub_bits, multi_floor;
};

__device__ __attribute__((noinline)) uint32_t decode_one_slow_p(const DecodeTables *tb, uint32_t window) {
    const uint32_t bits = tb->lut_bits | (tb->sub_bits << 8);
    const uint32_t e = tb->lut[window >> (32 - tb->lut_bits)];
    if ((e >> LUT_N_SHIFT) & 3u) return (static_cast<uint32_t>(tb->sym_len[e & 0xffu]) << 8) | (e & 0xffu);
    return long_code_flat(tb->sub, tb->longc, tb->n_long, bits, e, window);
}

// The synchronisation walk over a lane's registers: counts, keeps no symbols (step table:
// et_kernels.h STEP_*).  Word iteration j works on the register pair (W[j-1], W[j]); the
// walk's position is kept RELATIVE TO THAT PAIR: the low half of X is G = 96 - sh, sh =
// bits from the first bit of W[j-1].  In the word <=> sh <= 32 <=> G >= 64; after the word
// G += 32; a lane thrown out by the escape pseudo-step (64 bits) has G < 32, a regular
// exit 32 <= G < 64: every word compares against the same two inline constants, and
// v_alignbit_b32's shift (-sh mod 32) is G's low five bits as they are.  A step is:
// alignbit, shift, address, LDS read, add, and, compare.
template <bool WARM>
__device__ __forceinline__ SubResult walk_steps(const StepWalk &sw, const uint32_t (&W)[RW_WORDS], uint32_t start_rel, uint32_t (&ck)[8]) {
    const uint32_t *steps = sw.steps;
    const uint32_t idx_shift = sw.idx_shift;
    const uint32_t steps_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)steps));  // the table's LDS address
    (void)steps_lds;
    uint32_t X, e = 0;
    SubResult res;
    res.start_rel = start_rel;
#define ET_F static_cast<uint16_t>(X)
#define ET_SW_STEP(hi_, lo_) X += (e = steps[__builtin_amdgcn_alignbit(hi_, lo_, X) >> idx_shift]);
// The whole-index step loop of one word, hand-written: the compiler's version spends a
// v_and + v_cmp on the 16-bit position field and three scalar instructions on the loop mask;
// here v_cmp_le_u16 reads the low half directly and lanes that leave the word are dropped
// from exec.  5 VALU + 1 LDS + 2 SALU per step.  floor_ = lowest G still in the word (an
// inline constant or an SGPR).
#define ET_SW_LOOP(hi_, lo_, floor_)                                                   \
    {                                                                                  \
        uint32_t t_;                                                                   \
        uint64_t saved_;                                                               \
        asm volatile(                                                                  \
            "s_mov_b64 %[sv], exec\n\t"                                                \
            "v_cmp_le_u16 vcc, %[fl], %[x]\n\t"                                        \
            "s_and_b64 exec, exec, vcc\n\t"                                            \
            "s_cbranch_execz 2f\n"                                                     \
            "1:\n\t"                                                                   \
            "v_alignbit_b32 %[t], %[hi], %[lo], %[x]\n\t"                              \
            "v_lshrrev_b32 %[t], %[sh], %[t]\n\t"                                      \
            "v_lshl_add_u32 %[t], %[t], 2, %[base]\n\t"                                \
            "ds_read_b32 %[e], %[t]\n\t"                                               \
            "s_waitcnt lgkmcnt(0)\n\t"                                                 \
            "v_add_u32 %[x], %[x], %[e]\n\t"                                           \
            "v_cmp_le_u16 vcc, %[fl], %[x]\n\t"                                        \
            "s_and_b64 exec, exec, vcc\n\t"                                            \
            "s_cbranch_execnz 1b\n"                                                    \
            "2:\n\t"                                                                   \
            "s_mov_b64 exec, %[sv]"                                                    \
            : [x] "+v"(X), [e] "+v"(e), [t] "=&v"(t_), [sv] "=&s"(saved_)              \
            : [hi] "v"(hi_), [lo] "v"(lo_), [sh] "s"(idx_shift), [base] "v"(steps_lds), [fl] "s"(floor_) \
            : "vcc", "scc");                                                           \
    }
// the code at X is longer than the index (`e` is its escape entry): second-level table, else the slow way
#define ET_SW_SLOW(hi_, lo_)                                                                                           \
    {                                                                                                                  \
        const uint32_t w_ = __builtin_amdgcn_alignbit(hi_, lo_, X), t_ = e >> 28;                                      \
        uint32_t add_ = 0;                                                                                             \
        if (t_) add_ = steps[(1u << sw.step_bits) + (((t_ - 1) << sw.sub_bits) | ((w_ << sw.step_bits) >> (32 - sw.sub_bits)))]; \
        if (add_ == 0) {                                                                                               \
            const uint32_t hit_ = decode_one_slow_p(sw.slow, w_);                                                      \
            add_ = hit_ ? (1u << 16) - (hit_ >> 8) : ~0u; /* no code: one bit on, no symbol */                         \
        }                                                                                                              \
        X += add_;                                                                                                     \
    }
// a word all of whose step_bits windows end before the stretch's limit
#define ET_SW_WORD(hi_, lo_)                                                          \
    for (;;) {                                                                        \
        ET_SW_LOOP(hi_, lo_, 64u)                                                     \
        if (ET_F >= 32) break;                                                        \
        X -= STEP_ESCAPE;                                                             \
        ET_SW_SLOW(hi_, lo_)                                                          \
    }                                                                                 \
    X += 32;
// the word at whose END the stretch ends: whole-index steps while step_bits bits are left
// before the limit, then single codewords.  Leaves G alone: 64 - G is how far the last
// codeword reached past the limit.
#define ET_SW_LAST_WORD(hi_, lo_)                                                     \
    for (;;) {                                                                        \
        ET_SW_LOOP(hi_, lo_, sw.multi_floor)                                          \
        if (ET_F >= 32) break;                                                        \
        X -= STEP_ESCAPE;                                                             \
        ET_SW_SLOW(hi_, lo_)                                                          \
    }                                                                                 \
    while (ET_F > 64) {                                                               \
        e = steps[__builtin_amdgcn_alignbit(hi_, lo_, X) >> idx_shift];               \
        if (static_cast<uint16_t>(e) != static_cast<uint16_t>(STEP_ESCAPE)) X += (1u << 16) - (e >> 28); \
        else ET_SW_SLOW(hi_, lo_)                                                     \
    }

    if (WARM) {
        X = 64;  // first bit of the run-in's first word (128 bits: 96 or 64 mean more re-walks, 0.61 / 0.64 vs 0.59 ms)
        ET_SW_WORD(0u, W[0])
        ET_SW_WORD(W[0], W[1])
        ET_SW_WORD(W[1], W[2])
        ET_SW_WORD(W[2], W[3])
        ET_SW_LAST_WORD(W[3], W[4])
        X &= 0xffffu;  // nothing counted so far
        res.start_rel = 64 - X;  // in [0, 31]; G is already what the next word wants
    } else {
        X = 64 - start_rel;
    }
    // ck[]: the state after each of the subsequence's first eight words (rewalk_steps)
    ET_SW_WORD(W[3], W[4])  // only lanes that start at bit 0
    ck[0] = X;
    ET_SW_WORD(W[4], W[5])
    ck[1] = X;
    ET_SW_WORD(W[5], W[6])
    ck[2] = X;
    ET_SW_WORD(W[6], W[7])
    ck[3] = X;
    ET_SW_WORD(W[7], W[8])
    ck[4] = X;
    ET_SW_WORD(W[8], W[9])
    ck[5] = X;
    ET_SW_WORD(W[9], W[10])
    ck[6] = X;
    ET_SW_WORD(W[10], W[11])
    ck[7] = X;
    ET_SW_LAST_WORD(W[11], W[12])
    res.exit_rel = 64 - (X & 0xffffu);
    res.count = (X >> 16) & 0xfffu;
    return res;
}

// Walk again from another start, given the checkpoints, exit and count of the walk before:
// codes re-synchronise within a few codewords, so after a word or two the new walk stands
// where the old one stood at the same word boundary -- from there on they are the same
// walk, and only the symbol count has to be carried over.  (The wavefront skips the words
// in which none of its lanes is still walking.)
__device__ __forceinline__ SubResult rewalk_steps(const StepWalk &sw, const uint32_t (&W)[RW_WORDS], uint32_t start_rel, uint32_t (&ck)[8],
                                                  uint32_t old_exit, uint32_t old_count) {
    const uint32_t *steps = sw.steps;
    const uint32_t idx_shift = sw.idx_shift;
    const uint32_t steps_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)steps));
    (void)steps_lds;
    uint32_t X = 64 - start_rel, e = 0;
    SubResult res;
    res.start_rel = start_rel;
    res.exit_rel = old_exit;
    res.count = 0;
    bool merged = false;
    uint32_t shift = 0;
#define ET_RW_CHECK(c_, hi_, lo_)                                                            \
    if (!merged) {                                                                           \
        ET_SW_WORD(hi_, lo_)                                                                 \
        if (static_cast<uint16_t>(X) == static_cast<uint16_t>(ck[c_])) {                     \
            merged = true;                                                                   \
            res.count = (old_count + (X >> 16) - (ck[c_] >> 16)) & 0xfffu;                   \
            shift = (res.count - old_count) << 16;                                           \
        }                                                                                    \
        ck[c_] = X;                                                                          \
    } else {                                                                                 \
        ck[c_] += shift; /* same walk from here on, other count before it */                 \
    }
    ET_RW_CHECK(0, W[3], W[4])
    ET_RW_CHECK(1, W[4], W[5])
    ET_RW_CHECK(2, W[5], W[6])
    ET_RW_CHECK(3, W[6], W[7])
    ET_RW_CHECK(4, W[7], W[8])
    ET_RW_CHECK(5, W[8], W[9])
    ET_RW_CHECK(6, W[9], W[10])
    ET_RW_CHECK(7, W[10], W[11])
#undef ET_RW_CHECK
    if (!merged) {
        ET_SW_LAST_WORD(W[11], W[12])
        res.exit_rel = 64 - (X & 0xffffu);
        res.count = (X >> 16) & 0xfffu;
    }
    return res;
}

// ---- 512-bit lanes for the first sweep ------------------------------------------------
// A lane that owns TWO consecutive subsequences pays the 128-bit run-in once per 512 bits
// (22 word iterations per 512 bits instead of 26): W[] holds 4 run-in words, 16 own, 1
// beyond.  The state arrays keep their 256-bit granularity (the lane produces both
// entries), so the repair sweeps, the scan and D3 are unchanged.
// (checkpointed re-walks as in rewalk_steps cost this kernel a wavefront of occupancy for
// the eight extra registers: 0.40 vs 0.39 ms -> none here; a re-walked lane walks its first
// subsequence again in full)
constexpr int RW2_WORDS = 21;
struct Sub2Result {
    uint32_t start_rel, exit1, count1, exit2, count2;
};
// one 256-bit half: the eight words from register pair (W[B+3], W[B+4]) on, then the last word
#define ET_SW_HALF_PLAIN(B_)                \
    ET_SW_WORD(W[(B_) + 3], W[(B_) + 4])    \
    ET_SW_WORD(W[(B_) + 4], W[(B_) + 5])    \
    ET_SW_WORD(W[(B_) + 5], W[(B_) + 6])    \
    ET_SW_WORD(W[(B_) + 6], W[(B_) + 7])    \
    ET_SW_WORD(W[(B_) + 7], W[(B_) + 8])    \
    ET_SW_WORD(W[(B_) + 8], W[(B_) + 9])    \
    ET_SW_WORD(W[(B_) + 9], W[(B_) + 10])   \
    ET_SW_WORD(W[(B_) + 10], W[(B_) + 11])  \
    ET_SW_LAST_WORD(W[(B_) + 11], W[(B_) + 12])
template <bool WARM>
__device__ __forceinline__ Sub2Result walk_steps2(const StepWalk &sw, const uint32_t (&W)[RW2_WORDS], uint32_t start_rel) {
    const uint32_t *steps = sw.steps;
    const uint32_t idx_shift = sw.idx_shift;
    const uint32_t steps_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)steps));
    (void)steps_lds;
    uint32_t X, e = 0;
    Sub2Result res;
    res.start_rel = start_rel;
    if (WARM) {
        X = 64;
        ET_SW_WORD(0u, W[0])
        ET_SW_WORD(W[0], W[1])
        ET_SW_WORD(W[1], W[2])
        ET_SW_WORD(W[2], W[3])
        ET_SW_LAST_WORD(W[3], W[4])
        X &= 0xffffu;
        res.start_rel = 64 - X;
    } else {
        X = 64 - start_rel;
    }
    ET_SW_HALF_PLAIN(0)
    res.exit1 = 64 - (X & 0xffffu);
    res.count1 = (X >> 16) & 0xfffu;
    X &= 0xffffu;  // the second subsequence starts where the first one's last codeword ended, and counts from zero
    ET_SW_HALF_PLAIN(8)  // (no checkpoints: a re-walk that reaches the second half is rare, it then walks all of it)
    res.exit2 = 64 - (X & 0xffffu);
    res.count2 = (X >> 16) & 0xfffu;
    return res;
}

// The re-walk of a 512-bit lane from a corrected start: `r` holds the previous walk's results and is
// updated.  If the first subsequence ends where it ended before, the second one stays as it is.
__device__ __forceinline__ void rewalk_steps2(const StepWalk &sw, const uint32_t (&W)[RW2_WORDS], uint32_t start_rel, Sub2Result &r) {
    const uint32_t *steps = sw.steps;
    const uint32_t idx_shift = sw.idx_shift;
    const uint32_t steps_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)steps));
    (void)steps_lds;
    uint32_t X = 64 - start_rel, e = 0;
    r.start_rel = start_rel;
    ET_SW_HALF_PLAIN(0)
    const uint32_t new_exit1 = 64 - (X & 0xffffu);
    r.count1 = (X >> 16) & 0xfffu;
    if (new_exit1 == r.exit1) return;  // the second subsequence starts where it started before
    r.exit1 = new_exit1;
    X &= 0xffffu;
    ET_SW_HALF_PLAIN(8)
    r.exit2 = 64 - (X & 0xffffu);
    r.count2 = (X >> 16) & 0xfffu;
}
#undef ET_SW_HALF_PLAIN
#undef ET_SW_LAST_WORD
#undef ET_SW_WORD
#undef ET_SW_SLOW
#undef ET_SW_STEP
#undef ET_F

// The words of lane `sub_g`'s subsequence (interior block: every index is inside the stream).
template <bool WITH_RUN_IN>
__device__ __forceinline__ void load_window(uint32_t (&W)[RW_WORDS], const uint32_t *__restrict__ words, uint64_t sub_g) {
    const uint32_t *src = words + sub_g * (SUB_BITS / 32) - 4;
#pragma unroll
    for (int j = WITH_RUN_IN ? 0 : 4; j < RW_WORDS; ++j) W[j] = __builtin_bswap32(src[j]);
    if (!WITH_RUN_IN) W[0] = W[1] = W[2] = W[3] = 0;
}

// D1 for interior blocks; same protocol and state as k_dec_sync (which keeps the special
// blocks: the stream's first block and the one or two it ends in).
// Occupancy targets handed to the compiler (amdgpu_waves_per_eu).  k_dec_sync_reg wants ~100
// SGPRs, which caps it at 7 wavefronts per SIMD; asked for 8 the compiler parks ~24 of them in
// VGPR lanes and the kernel is 11 % faster (0.52 -> 0.46 ms); 9 is out of reach.  k_dec_write_reg
// is held at 6 workgroups per CU by its LDS, so the same request changes nothing there.
#define ET_SYNC_ATTR __attribute__((amdgpu_waves_per_eu(8, 10)))
template <bool FIRST, bool TICKET>
__global__ __launch_bounds__(BLOCK) ET_SYNC_ATTR void k_dec_sync_reg(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t n_blocks,
                                                        StepTableArgs ta, uint32_t *__restrict__ sub_state,
                                                        uint32_t *__restrict__ blk_exit, uint32_t *__restrict__ blk_count,
                                                        uint32_t *__restrict__ changed, uint32_t *__restrict__ ticket, uint32_t max_trips, uint32_t chunk,
                                                        const uint32_t *__restrict__ worklist, const uint32_t *__restrict__ n_work) {
    // LDS: step table, its second-level tables | exits | scratch
    uint32_t *steps = reinterpret_cast<uint32_t *>(dec_smem_raw);
    const uint32_t step_words = ta.words;
    const StepWalk sw = {steps, ta.slow, 32 - ta.step_bits, ta.step_bits, ta.sub_bits, 64 + ta.step_bits};
    DecodeSmem m;
    m.exits = steps + step_words;
    m.scratch = m.exits + BLOCK;
    const int tid = threadIdx.x;
    bool staged = false;
    // TICKET: resident workgroups draw blocks from a counter and stage the table once;
    // worklist (repair sweeps): the blocks k_dec_check found, strided over the grid;
    // otherwise one block per workgroup.
    const uint32_t n_wl = worklist ? *n_work : 0;
    uint32_t wi = blockIdx.x;
    for (uint64_t b = blockIdx.x, b_end = 0;; ++b) {
        if (!TICKET && worklist) {
            if (wi >= n_wl) break;
            __syncthreads();  // everybody is done with scratch and exits of the previous block
            b = worklist[wi];
            wi += gridDim.x;
        }
        if (TICKET) {
            __syncthreads();  // everybody is done with scratch and exits of the previous block
            if (b >= b_end) {  // next chunk of consecutive blocks
                if (tid == 0) m.scratch[7] = atomicAdd(ticket, chunk);
                __syncthreads();
                b = m.scratch[7];
                b_end = b + chunk;
            }
        }
        if (b >= n_blocks) break;
        if (!special_block(b, n_bytes)) {
            const uint64_t sub_g = b * BLOCK + tid;
            uint32_t start = 0, exit_rel = 0, count = 0, cand = 0;
            bool need = FIRST, warm = FIRST, skip = false, have_ck = false;
            uint32_t ck[8];
            if (!FIRST) {
                const uint32_t st = sub_state[sub_g];
                start = cand = st & 0xffu;
                exit_rel = (st >> 8) & 0xffu;
                count = st >> 16;
                if (tid == 0) {
                    cand = blk_exit[b - 1];
                    need = cand != start;
                    m.scratch[4] = need;
                }
                __syncthreads();
                skip = !m.scratch[4];
                if (!skip && tid == 0) *changed = 1;
            }
            if (!skip) {
                uint32_t W[RW_WORDS];
                load_window<true>(W, words, sub_g);
                if (!staged) {
                    for (uint32_t i = tid * 4; i < step_words; i += BLOCK * 4)
                        *reinterpret_cast<uint4 *>(steps + i) = *reinterpret_cast<const uint4 *>(ta.table + i);
                    staged = true;
                    __syncthreads();
                }
                for (uint32_t trip = 0;; ++trip) {
                    if (trip == max_trips) {  // see k_dec_sync
                        if (tid == 0) {
                            if (FIRST) atomicAdd(changed + 1, 1u);
                            else *changed = 1;
                            start = 0xffu;
                        }
                        break;
                    }
                    if (need) {
                        const SubResult r = warm      ? walk_steps<true>(sw, W, 0, ck)
                                            : have_ck ? rewalk_steps(sw, W, cand, ck, exit_rel, count)
                                                      : walk_steps<false>(sw, W, cand, ck);
                        have_ck = true;
                        start = r.start_rel;
                        exit_rel = r.exit_rel;
                        count = r.count;
                        warm = false;
                    }
                    m.exits[tid] = exit_rel;
                    __syncthreads();
                    need = false;
                    if (tid > 0) {
                        cand = m.exits[tid - 1];
                        need = cand != start;
                    }
                    if (!__syncthreads_or(need)) break;
                }
                sub_state[sub_g] = start | (exit_rel << 8) | (count << 16);
                uint32_t total;
                block_exclusive_scan(count, m.scratch, &total);
                if (tid == 0) blk_count[b] = total;
                if (tid == BLOCK - 1) blk_exit[b] = exit_rel;
            }
        }
        if (!TICKET && !worklist) break;
    }
}

// D1, first sweep, 512-bit lanes: a workgroup takes a superblock of two blocks (16 KiB);
// protocol and outputs as k_dec_sync_reg<true, true>.
__global__ __launch_bounds__(BLOCK) ET_SYNC_ATTR void k_dec_sync_reg2(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t n_blocks,
                                                                     StepTableArgs ta, uint32_t *__restrict__ sub_state,
                                                                     uint32_t *__restrict__ blk_exit, uint32_t *__restrict__ blk_count,
                                                                     uint32_t *__restrict__ changed, uint32_t *__restrict__ ticket,
                                                                     uint32_t max_trips, uint32_t chunk) {
    uint32_t *steps = reinterpret_cast<uint32_t *>(dec_smem_raw);
    const StepWalk sw = {steps, ta.slow, 32 - ta.step_bits, ta.step_bits, ta.sub_bits, 64 + ta.step_bits};
    uint32_t *exits = steps + ta.words;
    uint32_t *scratch = exits + BLOCK;
    const int tid = threadIdx.x;
    const uint32_t n_super = n_blocks / 2;
    for (uint32_t i = tid * 4; i < ta.words; i += BLOCK * 4) *reinterpret_cast<uint4 *>(steps + i) = *reinterpret_cast<const uint4 *>(ta.table + i);
    for (uint64_t sb = 0, sb_end = 0;; ++sb) {
        __syncthreads();  // table staged (first trip); everybody is done with scratch and exits
        if (sb >= sb_end) {
            if (tid == 0) scratch[7] = atomicAdd(ticket, chunk);
            __syncthreads();
            sb = scratch[7];
            sb_end = sb + chunk;
        }
        if (sb >= n_super) break;
        if (!super_interior(sb, n_bytes, n_blocks)) continue;
        const uint64_t q = sb * BLOCK + tid;  // 512-bit lane index = subsequences 2q, 2q + 1
        uint32_t W[RW2_WORDS];
        {
            const uint32_t *src = words + q * (2 * SUB_BITS / 32) - 4;
#pragma unroll
            for (int j = 0; j < RW2_WORDS; ++j) W[j] = __builtin_bswap32(src[j]);
        }
        Sub2Result r = walk_steps2<true>(sw, W, 0);
        uint32_t start = r.start_rel;
        for (uint32_t trip = 1;; ++trip) {
            exits[tid] = r.exit2;
            __syncthreads();
            uint32_t cand = start;
            if (tid > 0) cand = exits[tid - 1];
            const bool need = cand != start;
            if (!__syncthreads_or(need)) break;
            if (trip == max_trips) {  // see k_dec_sync; BOTH blocks of the pair are marked for a redo
                if (tid == 0) atomicAdd(changed + 1, 2u);
                if (tid == 0 || tid == BLOCK / 2) start = 0xffu;
                break;
            }
            if (need) {
                rewalk_steps2(sw, W, cand, r);
                start = cand;
            }
        }
        sub_state[2 * q] = start | (r.exit1 << 8) | (r.count1 << 16);
        sub_state[2 * q + 1] = r.exit1 | (r.exit2 << 8) | (r.count2 << 16);
        uint32_t total;
        const uint32_t before = block_exclusive_scan(r.count1 + r.count2, scratch, &total);
        if (tid == BLOCK / 2) {  // symbols of lanes 0..127 = first block of the pair
            blk_count[2 * sb] = before;
            blk_count[2 * sb + 1] = total - before;
        }
        if (tid == BLOCK / 2 - 1) blk_exit[2 * sb] = r.exit2;
        if (tid == BLOCK - 1) blk_exit[2 * sb + 1] = r.exit2;
    }
}

// Repair sweeps, step 1: one thread per block compares the start its first subsequence used
// with the exit its predecessor ended on; the blocks that disagree (or gave up: start 0xff)
// go on the worklist of k_dec_sync_reg<false> (special blocks look after themselves).
__global__ __launch_bounds__(BLOCK) void k_dec_check(const uint32_t *__restrict__ sub_state, const uint32_t *__restrict__ blk_exit, uint32_t n_blocks,
                                                     uint32_t *__restrict__ worklist, uint32_t *__restrict__ n_work) {
    const uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b == 0 || b >= n_blocks) return;
    if ((sub_state[static_cast<uint64_t>(b) * BLOCK] & 0xffu) != blk_exit[b - 1]) worklist[atomicAdd(n_work, 1u)] = b;
}

// X1 / X3 for interior blocks: the exhaustive path's walks over registers (walk_steps; the
// step table of k_dec_sync_reg).  Same maps, same outputs as k_dec_maps / k_dec_resolve,
// which keep the stream's first and last blocks.  LDS: step table | maps[BLOCK][32] | exits | scratch.
__global__ __launch_bounds__(BLOCK) void k_dec_maps_reg(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t n_blocks, StepTableArgs ta,
                                                        uint32_t n_starts, uint32_t map_stride, uint8_t *__restrict__ lane_maps,
                                                        uint8_t *__restrict__ blk_maps) {
    const uint64_t b = blockIdx.x;
    if (b >= n_blocks || special_block(b, n_bytes)) return;
    uint32_t *steps = reinterpret_cast<uint32_t *>(dec_smem_raw);
    uint8_t *maps = reinterpret_cast<uint8_t *>(steps + ta.words);
    const StepWalk sw = {steps, ta.slow, 32 - ta.step_bits, ta.step_bits, ta.sub_bits, 64 + ta.step_bits};
    const int tid = threadIdx.x;
    const uint64_t sub_g = b * BLOCK + tid;
    for (uint32_t i = tid * 4; i < ta.words; i += BLOCK * 4) *reinterpret_cast<uint4 *>(steps + i) = *reinterpret_cast<const uint4 *>(ta.table + i);
    uint32_t W[RW_WORDS], ck[8];
    load_window<false>(W, words, sub_g);
    __syncthreads();
    for (uint32_t p = 0; p < 32; ++p) {
        uint32_t e = 0;
        if (p < n_starts) e = walk_steps<false>(sw, W, p, ck).exit_rel;
        maps[tid * 32 + p] = static_cast<uint8_t>(e);
        if (p + 1 >= n_starts && p + 1 >= map_stride) break;
    }
    __syncthreads();
    for (uint32_t k = 0; k < map_stride; k += 8)
        *reinterpret_cast<uint2 *>(lane_maps + sub_g * map_stride + k) = *reinterpret_cast<const uint2 *>(maps + tid * 32 + k);
    if (tid < 32) {
        uint32_t sidx = tid;
        if (static_cast<uint32_t>(tid) < n_starts)
            for (uint32_t i = 0; i < BLOCK; ++i) sidx = maps[i * 32 + sidx];
        blk_maps[b * 32 + tid] = static_cast<uint8_t>(sidx);
    }
}

__global__ __launch_bounds__(BLOCK) void k_dec_resolve_reg(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t n_blocks, StepTableArgs ta,
                                                           uint32_t map_stride, const uint8_t *__restrict__ lane_maps,
                                                           const uint8_t *__restrict__ blk_in, uint32_t *__restrict__ sub_state,
                                                           uint32_t *__restrict__ blk_exit, uint32_t *__restrict__ blk_count) {
    const uint64_t b = blockIdx.x;
    if (b >= n_blocks || special_block(b, n_bytes)) return;
    uint32_t *steps = reinterpret_cast<uint32_t *>(dec_smem_raw);
    uint8_t *maps = reinterpret_cast<uint8_t *>(steps + ta.words);
    uint32_t *exits = steps + ta.words + BLOCK * 32 / 4;
    uint32_t *scratch = exits + BLOCK;
    const StepWalk sw = {steps, ta.slow, 32 - ta.step_bits, ta.step_bits, ta.sub_bits, 64 + ta.step_bits};
    const int tid = threadIdx.x;
    const uint64_t sub_g = b * BLOCK + tid;
    for (uint32_t i = tid * 4; i < ta.words; i += BLOCK * 4) *reinterpret_cast<uint4 *>(steps + i) = *reinterpret_cast<const uint4 *>(ta.table + i);
    for (uint32_t k = 0; k < map_stride; k += 8)
        *reinterpret_cast<uint2 *>(maps + tid * 32 + k) = *reinterpret_cast<const uint2 *>(lane_maps + sub_g * map_stride + k);
    uint32_t W[RW_WORDS], ck[8];
    load_window<false>(W, words, sub_g);
    __syncthreads();
    if (tid == 0) {
        uint32_t sidx = blk_in[b];
        for (uint32_t i = 0; i < BLOCK; ++i) {
            exits[i] = sidx;
            sidx = maps[i * 32 + sidx];
        }
    }
    __syncthreads();
    const uint32_t start = exits[tid];
    const SubResult r = walk_steps<false>(sw, W, start, ck);
    sub_state[sub_g] = start | (r.exit_rel << 8) | (r.count << 16);
    uint32_t total;
    block_exclusive_scan(r.count, scratch, &total);
    if (tid == 0) blk_count[b] = total;
    if (tid == BLOCK - 1) blk_exit[b] = r.exit_rel;
}

// The write walk over a lane's registers (et_kernels.h WSTEP_*): as walk_steps, with the
// stage position riding in the state's upper bits.  A step stores two bytes: the second
// symbol first, at (position after the step) - 1 -- for a one-symbol entry that is the
// first symbol's own slot, which the first symbol then overwrites -- so no store is
// conditional and none leaves the lane's own slots.
//   MODE 1: X's upper bits are LDS addresses - 1 (the whole block fits the stage).
//   MODE 2: they are positions in the block's output; bytes in [lo, hi) go to stage[pos - lo].
template <int MODE>
__device__ __forceinline__ void walk_write(const StepWalk &sw, const uint8_t *sym_len, uint8_t *smem8, const uint32_t (&W)[RW_WORDS], uint32_t start_rel,
                                           uint32_t pos0, uint32_t lo, uint32_t hi, uint32_t stage_off) {
    const uint32_t *wsteps = sw.steps;
    const uint32_t idx_shift = sw.idx_shift;
    uint32_t X = (pos0 << 10) | (64 - start_rel), e = 0;  // low 10 bits: G as in walk_steps
#define ET_F (X & 1023u)
// MODE 1 positions are absolute LDS addresses minus one, used as integers (nothing to add,
// and both of a step's stores get their -1 / +0 folded into the instruction's offset)
#define ET_PUT(p_, v_)                                                                          \
    {                                                                                           \
        if (MODE == 1) *reinterpret_cast<lds_u8 *>(static_cast<uintptr_t>((p_) + 1u)) = static_cast<uint8_t>(v_); \
        else if ((p_) - lo < hi - lo) smem8[stage_off + ((p_) - lo)] = static_cast<uint8_t>(v_); \
    }
#define ET_WW_STEP(hi_, lo_)                                                    \
    {                                                                           \
        e = wsteps[__builtin_amdgcn_alignbit(hi_, lo_, X) >> idx_shift];        \
        const uint32_t p0_ = X >> 10;                                           \
        X += e & 0xffffu;                                                       \
        const uint32_t p1_ = X >> 10;                                           \
        ET_PUT(p1_ - 1, e >> 24)                                                \
        ET_PUT(p0_, e >> 16)                                                    \
    }
#define ET_WW_SLOW(hi_, lo_)                                                                                           \
    {                                                                                                                  \
        const uint32_t w_ = __builtin_amdgcn_alignbit(hi_, lo_, X), t_ = e >> 24;                                      \
        uint32_t ent_ = 0;                                                                                             \
        if (t_) ent_ = wsteps[(1u << sw.step_bits) + (((t_ - 1) << sw.sub_bits) | ((w_ << sw.step_bits) >> (32 - sw.sub_bits)))]; \
        if (ent_ == 0) {                                                                                               \
            const uint32_t hit_ = decode_one_slow_p(sw.slow, w_);                                                      \
            if (hit_) ent_ = ((hit_ & 0xffu) << 16) | ((1u << 10) - (hit_ >> 8));                                      \
        }                                                                                                              \
        if (ent_) {                                                                                                    \
            ET_PUT(X >> 10, ent_ >> 16)                                                                                \
            X += ent_ & 0xffffu;                                                                                       \
        } else {                                                                                                       \
            X -= 1; /* no code: one bit on, no symbol */                                                               \
        }                                                                                                              \
    }
// (A hand-written version of this loop like ET_SW_LOOP, with an SDWA byte compare on the
// position field, was 8 VALU + 2 SALU per step instead of 9 + 3 -- and hung one test in one
// ordering of the suite: lanes left the loop at the wrong time now and then, most likely the
// SDWA compare's VCC reaching the s_and a cycle late.  Not worth it for a kernel bound by the
// LDS pipe; the compiler's loop stays.)
#define ET_WW_LOOP(hi_, lo_, floor_) while (ET_F >= (floor_)) ET_WW_STEP(hi_, lo_)
#define ET_WW_WORD(hi_, lo_)                                                          \
    for (;;) {                                                                        \
        ET_WW_LOOP(hi_, lo_, 64u)                                                     \
        if (ET_F >= 32) break;                                                        \
        X -= WSTEP_ESCAPE;                                                            \
        ET_WW_SLOW(hi_, lo_)                                                          \
    }                                                                                 \
    X += 32;
    ET_WW_WORD(W[3], W[4])  // only lanes that start at bit 0
    ET_WW_WORD(W[4], W[5])
    ET_WW_WORD(W[5], W[6])
    ET_WW_WORD(W[6], W[7])
    ET_WW_WORD(W[7], W[8])
    ET_WW_WORD(W[8], W[9])
    ET_WW_WORD(W[9], W[10])
    ET_WW_WORD(W[10], W[11])
    for (;;) {  // the last word: two-symbol steps while step_bits bits are left, then one codeword at a time
        ET_WW_LOOP(W[11], W[12], sw.multi_floor)
        if (ET_F >= 32) break;
        X -= WSTEP_ESCAPE;
        ET_WW_SLOW(W[11], W[12])
    }
    while (ET_F > 64) {
        e = wsteps[__builtin_amdgcn_alignbit(W[11], W[12], X) >> idx_shift];
        if ((e & 0xffffu) != WSTEP_ESCAPE) {
            const uint32_t s1 = (e >> 16) & 0xffu;
            ET_PUT(X >> 10, s1)
            X += (1u << 10) - sym_len[s1];
        } else {
            ET_WW_SLOW(W[11], W[12])
        }
    }
#undef ET_WW_WORD
#undef ET_WW_LOOP
#undef ET_WW_SLOW
#undef ET_WW_STEP
#undef ET_PUT
#undef ET_F
}

// D3 for interior blocks (tickets of WRITE_CHUNK blocks, as k_dec_write).
__global__ __launch_bounds__(BLOCK) void k_dec_write_reg(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t n_blocks,
                                                         StepTableArgs ta, const uint8_t *__restrict__ sym_len_g, const uint32_t *__restrict__ sub_state,
                                                         const unsigned long long *__restrict__ blk_off, uint64_t n_symbols,
                                                         uint8_t *__restrict__ out, uint32_t *__restrict__ ticket,
                                                         const uint32_t *__restrict__ void_flags) {
    // A speculative launch (enqueued before the host has seen the sweeps' flags) does nothing
    // when the synchronisation did not settle (dec_state_final, the host's own rule):
    // void_flags[1] = blocks that gave up in the first sweep (their first subsequence carries the
    // start marker 0xff until a repair sweep replaces it; one that is left fails the
    // verification), void_flags[2] = the verification failed.  The host discards this launch's
    // output in exactly these cases and writes again once the state is final.  (Walking from the marker would send the packed walk state's
    // address field through the LDS tables: a stream with long runs of one code hung this kernel.)
    if (void_flags && !dec_state_final(void_flags[1], void_flags[2], n_blocks)) return;
    // LDS: step table, its second-level tables | code lengths | scratch | stage
    uint32_t *wsteps = reinterpret_cast<uint32_t *>(dec_smem_raw);
    const uint32_t step_words = ta.words;
    const StepWalk sw = {wsteps, ta.slow, 32 - ta.step_bits, ta.step_bits, ta.sub_bits, 64 + ta.step_bits};
    uint8_t *sym_len = reinterpret_cast<uint8_t *>(wsteps + step_words);
    uint32_t *scratch = wsteps + step_words + 64;
    const uint32_t stage_off = (step_words + 64 + 8) * sizeof(uint32_t);
    uint8_t *smem8 = reinterpret_cast<uint8_t *>(dec_smem_raw);
    uint8_t *stage = smem8 + stage_off;
    const uint32_t lds_stage = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)stage));  // the stage's LDS address
    const int tid = threadIdx.x;
    for (uint32_t i = tid * 4; i < step_words; i += BLOCK * 4)
        *reinterpret_cast<uint4 *>(wsteps + i) = *reinterpret_cast<const uint4 *>(ta.table + i);
    sym_len[tid] = sym_len_g[tid];
    for (;;) {
        __syncthreads();  // tables staged (first trip); everybody is done with scratch[7] and the stage
        if (tid == 0) scratch[7] = atomicAdd(ticket, WRITE_CHUNK);
        __syncthreads();
        const uint64_t b0 = scratch[7];
        if (b0 >= n_blocks) break;
        const uint64_t b1 = b0 + WRITE_CHUNK < n_blocks ? b0 + WRITE_CHUNK : n_blocks;
        for (uint64_t b = b0; b < b1; ++b) {
            if (special_block(b, n_bytes)) continue;  // k_dec_write
            const uint64_t o0 = blk_off[b];
            if (o0 >= n_symbols) break;  // pad bits decoded past the declared length; offsets only grow from here
            const uint64_t sub_g = b * BLOCK + tid;
            const uint32_t st = sub_state[sub_g];
            const uint32_t start = st & 31u, count = st >> 16;  // (a start is < 32 in a settled state; masked so that nothing else can reach the walk)
            uint32_t W[RW_WORDS];
            load_window<false>(W, words, sub_g);
            uint32_t block_total;
            const uint32_t my_off = block_exclusive_scan(count, scratch, &block_total);  // its barrier also separates the blocks' use of the stage

            uint64_t o1 = o0 + block_total;
            if (o1 > n_symbols) o1 = n_symbols;
            const uint32_t n_out = static_cast<uint32_t>(o1 - o0);
            const uint32_t phase = static_cast<uint32_t>(o0 & 15);  // stage offset of the first symbol
            uint8_t *out_base = out + (o0 - phase);
            const bool one_window = phase + block_total <= DEC_STAGE_BYTES && n_out == block_total;
            for (uint32_t win = 0; win < phase + n_out; win += DEC_STAGE_BYTES) {
                const uint32_t win_hi = min(win + DEC_STAGE_BYTES, phase + n_out);
                const uint32_t my_lo = phase + my_off, my_hi = my_lo + count;
                if (one_window) {
                    if (count) walk_write<1>(sw, sym_len, smem8, W, start, lds_stage + my_lo - 1u, 0, 0, 0);
                } else if (my_lo < win_hi && my_hi > win) {
                    walk_write<2>(sw, sym_len, smem8, W, start, my_lo, win, win_hi, stage_off);
                }
                __syncthreads();
                const uint32_t lo_valid = max(win, phase);  // first stage position holding a symbol in this window
                for (uint32_t g = win + tid * 16; g < win_hi; g += BLOCK * 16) {
                    if (g >= lo_valid && g + 16 <= win_hi) {
                        *reinterpret_cast<uint4 *>(out_base + g) = *reinterpret_cast<const uint4 *>(stage + (g - win));
                    } else {
                        for (uint32_t k = max(g, lo_valid); k < min(g + 16, win_hi); ++k) out_base[k] = stage[k - win];
                    }
                }
                __syncthreads();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// D3 over chained lookup tables (et_treewalk.h): the same greedy register-window walk, but the table entry names the
// table of the next lookup, so a code longer than the index is one more lockstep step of ITS lane instead of an
// escape that stops the wavefront.  State: X = (stage position << 10) | G as in walk_write, H = the hi dword of
// the last entry (next table's LDS address, second symbol, next shift).
struct ChainWalk {
    uint32_t root_h;  // H at a codeword boundary: the root table's LDS address | (32 - CH_ROOT_BITS) << 24
    uint32_t root_t;  // its address alone
    bool has_len32;   // the dictionary has a 32-bit code
};
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

// MODE 1: positions are LDS addresses - 1 (the whole block fits the stage).  Words whose positions lie 64 or more
// bits before the subsequence's end (32 or more when no code is 32 bits long) use the FAST step: both symbol bytes stored behind one another, whatever the
// entry holds.  After a one-symbol entry the second slot holds a stray byte, after a no-symbol entry both do -- the
// lane's own slots: a codeword that BEGINS there is at most 32 bits long, so it ends before the subsequence does,
// another codeword of this lane begins behind it, and its store (stores of one wavefront reach the LDS in program
// order) lands on the stray byte.  From there on the SAFE step, which touches nothing but the slots of the symbols
// the entry completes (the slot behind a lane's LAST symbol is the next lane's first, written long before).
// MODE 2: positions are indices into the block's output; bytes in [lo, hi) go to stage[pos - lo] (blocks that
// overflow the stage or the declared symbol count: conditional stores throughout).
template <int MODE>
__device__ __forceinline__ void walk_write_chain(const ChainWalk cw, uint8_t *smem8, const uint32_t (&W)[RW_WORDS], uint32_t start_rel, uint32_t pos0,
                                                 uint32_t lo, uint32_t hi, uint32_t stage_off) {
    uint32_t X = (pos0 << 10) | (64 - start_rel), H = cw.root_h;
    uint2 e;
#define CH_G (X & 1023u)
#define CH_MID ((H & 0xffffu) != cw.root_t)
#define CH_READ(hi_, lo_)                                                                                        \
    {                                                                                                            \
        const uint32_t w_ = __builtin_amdgcn_alignbit(hi_, lo_, X);                                              \
        const unsigned long long v_ = *reinterpret_cast<const lds_u64 *>(static_cast<uintptr_t>(((w_ >> (H >> 24)) << 3) + (H & 0xffffu))); \
        e.x = static_cast<uint32_t>(v_);                                                                         \
        e.y = static_cast<uint32_t>(v_ >> 32);                                                                   \
    }
#define CH_ADV X += static_cast<uint32_t>(static_cast<int32_t>(static_cast<int16_t>(e.x)))
#define CH_PUT1(slot_, v_) *reinterpret_cast<lds_u8 *>(static_cast<uintptr_t>((slot_) + 1u)) = static_cast<uint8_t>(v_)
#define CH_PUT2(slot_, v_)                                                                   \
    {                                                                                        \
        const uint32_t q_ = (slot_);                                                         \
        if (q_ - lo < hi - lo) smem8[stage_off + (q_ - lo)] = static_cast<uint8_t>(v_);       \
    }
#define CH_STEP_FAST(hi_, lo_)           \
    {                                    \
        CH_READ(hi_, lo_)                \
        H = e.y;                         \
        const uint32_t p0_ = X >> 10;    \
        CH_ADV;                          \
        CH_PUT1(p0_, e.x >> 16);         \
        CH_PUT1(p0_ + 1u, e.y >> 16);    \
    }
#define CH_STEP_SAFE(hi_, lo_)                                                              \
    {                                                                                       \
        CH_READ(hi_, lo_)                                                                   \
        H = e.y;                                                                            \
        const uint32_t p0_ = X >> 10;                                                       \
        CH_ADV;                                                                             \
        const uint32_t p1_ = X >> 10;                                                       \
        if (MODE == 1) {                                                                    \
            /* second symbol first: at p0 + 1 for two symbols, else at p0, where the first (or, for none, a later one) overwrites it */ \
            CH_PUT1(p1_ - 1u + (p1_ == p0_ ? 1u : 0u), e.y >> 16);                          \
            CH_PUT1(p0_, e.x >> 16);                                                        \
        } else {                                                                            \
            if (p1_ != p0_) CH_PUT2(p0_, e.x >> 16)                                         \
            if (p1_ - p0_ == 2u) CH_PUT2(p0_ + 1u, e.y >> 16)                               \
        }                                                                                   \
    }
// one codeword (or one more table of it) at a time
#define CH_STEP_ONE(hi_, lo_)                        \
    {                                                \
        CH_READ(hi_, lo_)                            \
        const uint32_t lf_ = e.x >> 24;              \
        if (lf_) {                                   \
            if (MODE == 1) CH_PUT1(X >> 10, e.x >> 16); \
            else CH_PUT2(X >> 10, e.x >> 16)         \
            X += (1u << 10) - lf_;                   \
            H = cw.root_h;                           \
        } else {                                     \
            CH_ADV;                                  \
            H = e.y;                                 \
        }                                            \
    }
#define CH_WORD_FAST(hi_, lo_)                        \
    while (CH_G >= 64u) {                             \
        if (MODE == 1) CH_STEP_FAST(hi_, lo_)         \
        else CH_STEP_SAFE(hi_, lo_)                   \
    }                                                 \
    X += 32;
    CH_WORD_FAST(W[3], W[4])  // only lanes that start at bit 0
    CH_WORD_FAST(W[4], W[5])
    CH_WORD_FAST(W[5], W[6])
    CH_WORD_FAST(W[6], W[7])
    CH_WORD_FAST(W[7], W[8])
    CH_WORD_FAST(W[8], W[9])
    CH_WORD_FAST(W[9], W[10])
    // positions 193..224: a 32-bit code that begins at 224 is the lane's last, so with such codes about this word is SAFE
    if (MODE == 1 && !cw.has_len32) {
        while (CH_G >= 64u) CH_STEP_FAST(W[10], W[11])
    } else {
        while (CH_G >= 64u) CH_STEP_SAFE(W[10], W[11])
    }
    X += 32;
    // the last word: whole-index steps while the lookup's index bits all lie inside the subsequence, then one codeword at a time
    while (CH_G + (H >> 24) >= 96u) CH_STEP_SAFE(W[11], W[12])
    for (uint32_t k = 0; k < 40 && (CH_G > 64u || (CH_G == 64u && CH_MID)); ++k) CH_STEP_ONE(W[11], W[12])
    // a codeword that began inside the subsequence and is still open behind it: its remaining tables
    if (CH_MID) {
        X += 32;
        for (uint32_t k = 0; k < 40 && CH_MID; ++k) CH_STEP_ONE(W[12], 0u)
    }
#undef CH_WORD_FAST
#undef CH_STEP_ONE
#undef CH_STEP_SAFE
#undef CH_STEP_FAST
#undef CH_PUT2
#undef CH_PUT1
#undef CH_ADV
#undef CH_READ
#undef CH_MID
#undef CH_G
}

// ---------------------------------------------------------------------------------------------------------
// D3 with wavefronts that own their work end to end (as D1's do): a wavefront takes a QUARTER of an 8 KiB block -- 64
// subsequences, ~3.5 K symbols of text -- works out where its output begins from the four quarters' counts (four loads and
// a DPP reduction instead of an LDS exchange behind a barrier), walks into a stage of its own and stores its own bytes of
// the output: whole 16-byte chunks as such, the two chunks it shares with its neighbours one byte per lane.  Nothing is
// shared after the tables are staged: no barrier, no ticket -- quarters are dealt out by wavefront number.
// (Rounds 2-3 had a workgroup of 512 threads per two blocks here, k_dec_write_chain: a trip was a serial chain -- ticket
// atomic, loads, scan, barrier, walk, barrier, stores, barrier -- that three workgroups per CU could not cover; without its
// walk that kernel still took 0.33 of its 0.49 ms.  24 wavefronts per CU are 24 chains: 0.49 -> 0.445 ms, of which the
// output stores and loads alone, in this pattern, are 0.36 -- the mixed-traffic floor of the HBM, ~5 TB/s -- and the walk
// alone 0.42.  With tickets of 8 blocks per workgroup, one barrier pair per ticket: 0.442-0.468, no better than dealt out.)
constexpr uint32_t WV_STAGE = DEC_STAGE_BYTES / 4;  // bytes of output a wavefront stages at once (a block's stage, quartered)
constexpr uint32_t WV_STAGE_ALLOC = WV_STAGE + 32;

__device__ __forceinline__ uint32_t wave_sum(uint32_t x) { return __builtin_amdgcn_readlane(wave_inclusive_scan(x), 63); }

// (round-4 probe, timing only -- wrong offsets: what D3 would load if D1 handed it per-quarter prefixes: its own state word, not the four quarters')
// (round-4 probe, timing only -- wrong output: the write pass with its stream words served from the L2 (the stream's first 64 KiB over and
// over): the floor of the write PHASE of a kernel that has the block in registers already and loads nothing)
#ifdef ET_PROBE_D3_L2_LOADS
#define ET_PROBE_D3_SRC(sub_) (((sub_) & 2047u) + 16u)
#else
#define ET_PROBE_D3_SRC(sub_) (sub_)
#endif
#ifdef ET_PROBE_D3_ONE_STATE
#define ET_PROBE_ONE_STATE_COND &&q == quarter_
#else
#define ET_PROBE_ONE_STATE_COND
#endif
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_dec_write_wave(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t n_blocks,
                                                              const uint2 *__restrict__ chain, uint32_t n_entries,
                                                              const uint32_t *__restrict__ sub_state, const unsigned long long *__restrict__ blk_off,
                                                              uint64_t n_symbols, uint8_t *__restrict__ out,
                                                              const uint32_t *__restrict__ void_flags, uint64_t n_subs, uint32_t max_len) {
    if (void_flags && !dec_state_final(void_flags[1], void_flags[2], n_blocks)) return;  // see k_dec_write_reg
    // LDS: tables | WAVES stages
    uint2 *tab = reinterpret_cast<uint2 *>(dec_smem_raw);
    const uint32_t tab_bytes = (n_entries * 8u + 15u) & ~15u;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t stage_off = tab_bytes + wv * WV_STAGE_ALLOC;
    uint8_t *smem8 = reinterpret_cast<uint8_t *>(dec_smem_raw);
    uint8_t *stage = smem8 + stage_off;
    const uint32_t lds_tab = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)dec_smem_raw));
    const uint32_t lds_stage = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8 *)stage));
    for (uint32_t i = tid; i < n_entries; i += 64 * WAVES) {
        uint2 v = chain[i];
        v.y += lds_tab;  // next-table offsets -> LDS addresses
        tab[i] = v;
    }
    const ChainWalk cw = {lds_tab | ((32u - CH_ROOT_BITS) << 24), lds_tab, max_len >= 32};
    __syncthreads();  // tables staged
    {
        // What a quarter needs from memory -- its block's output offset, the four quarters' state words, its nine stream words
        // -- is asked for one quarter AHEAD and taken in front of the quarter's own output stores (loads and stores share the
        // in-order vmcnt: behind the stores, a wait for the words is a wait for the stores as well).  The asm statements pin
        // where the loads have to be back; without them the first use waits with vmcnt(0) for whatever was asked for last.
        const uint32_t stride = gridDim.x * WAVES, n_units = n_blocks * 4;
        uint64_t n_o0 = 0;
        uint32_t n_stq[4] = {0, 0, 0, 0}, n_W[RW_WORDS];
        bool n_edge = false;
#define WV_FETCH(u_)                                                                                              \
    {                                                                                                             \
        const uint64_t b_ = (u_) >> 2;                                                                            \
        const uint32_t quarter_ = (u_) & 3u;                                                                      \
        n_o0 = blk_off[b_];                                                                                       \
        n_edge = block_limit(n_bytes, b_) != 0xffffffffu;                                                         \
        _Pragma("unroll") for (uint32_t q = 0; q < 4; ++q) {                                                      \
            const uint64_t sg = b_ * BLOCK + q * 64 + lane;                                                       \
            n_stq[q] = (sg < n_subs ET_PROBE_ONE_STATE_COND) ? sub_state[sg] : 0u;                                \
        }                                                                                                         \
        const uint64_t sub_g_ = b_ * BLOCK + quarter_ * 64 + lane;                                                \
        if (sub_g_ < n_subs) {                                                                                    \
            if (n_edge) {                                                                                         \
                _Pragma("unroll") for (int j = 0; j < RW_WORDS; ++j)                                              \
                    n_W[j] = j < 4 ? 0u : __builtin_bswap32(load_be32_guarded(words, sub_g_ * (SUB_BITS / 32) - 4 + j, n_bytes)); \
            } else { /* (as they lie in memory: swapped when they are taken) */                                   \
                const uint32_t *src_ = words + ET_PROBE_D3_SRC(sub_g_) * (SUB_BITS / 32) - 4;                     \
                _Pragma("unroll") for (int j = 0; j < RW_WORDS; ++j) n_W[j] = j < 4 ? 0u : src_[j];               \
            }                                                                                                     \
        } else {                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < RW_WORDS; ++j) n_W[j] = 0;                                      \
        }                                                                                                         \
    }
        uint64_t o0 = 0;
        uint32_t stq[4], W[RW_WORDS];
        bool edge = false;
#define WV_TAKE()                                                                     \
    {                                                                                 \
        o0 = n_o0;                                                                    \
        edge = n_edge;                                                                \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                               \
            stq[q] = n_stq[q];                                                        \
            ET_PIN(stq[q]);                                                           \
        }                                                                             \
        _Pragma("unroll") for (int j = 0; j < RW_WORDS; ++j) {                        \
            uint32_t w_ = n_W[j];                                                     \
            ET_PIN(w_);                                                               \
            W[j] = __builtin_bswap32(w_);                                             \
        }                                                                             \
    }
        uint32_t u = blockIdx.x * WAVES + wv;
        if (u < n_units) {
            WV_FETCH(u)
            WV_TAKE()
        }
        for (; u < n_units; u += stride) {
            const bool more = u + stride < n_units;
            if (more) WV_FETCH(u + stride)
            const uint32_t quarter = u & 3u;
            // the four quarters' counts (the same lane of each): where this one's output begins
            uint32_t before = 0, st = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                if (q == quarter) st = stq[q];
                if (q < quarter) before += stq[q] >> 16;
            }
            before = wave_sum(before);
            const uint32_t start = st & 31u;  // (a start is < 32 in a settled state; masked so that nothing else can reach the walk)
            const uint32_t count = st >> 16;
            const uint32_t inc = wave_inclusive_scan(count);
            const uint32_t wave_total = __builtin_amdgcn_readlane(inc, 63), my_off = inc - count;
            const uint64_t ow = o0 + before;  // the wavefront's first symbol
            const bool nothing = o0 >= n_symbols || ow >= n_symbols || wave_total == 0;  // (pad bits decoded past the declared length)
            uint64_t o1 = ow + wave_total;
            if (o1 > n_symbols) o1 = n_symbols;
            const uint32_t n_out = nothing ? 0u : static_cast<uint32_t>(o1 - ow);
            const uint32_t phase = static_cast<uint32_t>(ow & 15);  // stage offset of the first symbol
            const uint32_t span = nothing ? 0u : phase + n_out;
            uint8_t *out_base = out + (ow - phase);
            const bool this_edge = edge;
            const bool one_window = phase + wave_total <= WV_STAGE && n_out == wave_total && !this_edge;
            const uint32_t my_lo = phase + my_off, my_hi = my_lo + count;
            // A window: walk into the stage, then store the stage.  The NEXT unit's words are taken exactly once per unit, at one
            // place in program order -- behind the unit's last walk (W is free) and in front of that window's stores (loads and
            // stores share the in-order vmcnt).  (Round 3 took them inside the window loop, under `if (last window)`: the
            // compiler's wait-count pass does not correlate that condition across the loop, saw loads that MIGHT still be
            // pending at the unit loop's head and put `s_waitcnt vmcnt(0)` there -- in front of the next fetch, i.e. every unit
            // waited for its own output stores to land before it asked for anything.)
#define WV_WINDOW_WALK(win_, win_hi_)                                                                      \
    if (one_window) {                                                                                      \
        if (count) walk_write_chain<1>(cw, smem8, W, start, lds_stage + my_lo - 1u, 0, 0, 0);              \
    } else if (my_lo < (win_hi_) && my_hi > (win_)) {                                                      \
        walk_write_chain<2>(cw, smem8, W, start, my_lo, (win_), (win_hi_), stage_off);                     \
    }
// (the wavefront's own LDS stores, then its own loads: in order, no barrier)
// (WV_LDS_ORDER: the walk stores bytes through integer-made LDS addresses, the copy-out reads them through `stage`, and lanes read
// what OTHER lanes of the wavefront stored: a wavefront-scope release / acquire pair around a wave barrier says so to the compiler --
// no instruction comes of it, the LDS itself keeps a wavefront's accesses in order -- in front of the reads and again in front of
// the next walk's stores)
#define WV_LDS_ORDER()                                          \
    {                                                           \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    }
#define WV_WINDOW_STORE(win_, win_hi_)                                                                     \
    {                                                                                                      \
        WV_LDS_ORDER()                                                                                     \
        const uint32_t lo_valid = max((win_), phase); /* first stage position holding a symbol in this window */ \
        for (uint32_t g = (win_) + lane * 16; g < (win_hi_); g += 64 * 16) {                               \
            if (g >= lo_valid && g + 16 <= (win_hi_)) {                                                    \
                typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));                             \
                /* (non-temporal: plain stores cost this kernel nothing, but the NEXT encode's K4 6 % -- 0.392 against 0.370) */ \
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4_nt *>(stage + (g - (win_))), reinterpret_cast<u32x4_nt *>(out_base + g)); \
            }                                                                                              \
        }                                                                                                  \
        /* the two chunks the wavefront shares with its neighbours (or the window's ends): its own bytes only, one per lane */ \
        const uint32_t head = lo_valid & ~15u, tail = (win_hi_) & ~15u;                                    \
        const uint32_t pos = (lane < 16 ? head : tail) + (lane & 15u);                                     \
        const bool partial = lane < 16 ? (lo_valid & 15u) != 0 : ((win_hi_) & 15u) != 0; /* (one chunk for both: its bytes are stored twice) */ \
        if (lane < 32 && partial && pos >= lo_valid && pos < (win_hi_)) out_base[pos] = stage[pos - (win_)]; \
        WV_LDS_ORDER()                                                                                     \
    }
            uint32_t win = 0;
            if (span) {
                for (;; win += WV_STAGE) {
                    const uint32_t win_hi = min(win + WV_STAGE, span);
                    WV_WINDOW_WALK(win, win_hi)
                    if (win_hi == span) break;  // the last window: its stores come behind the take
                    WV_WINDOW_STORE(win, win_hi)
                }
            }
            WV_TAKE()
            if (span) WV_WINDOW_STORE(win, span)
#undef WV_WINDOW_STORE
#undef WV_LDS_ORDER
#undef WV_WINDOW_WALK
        }
#undef WV_TAKE
#undef WV_FETCH
    }
}

// D2 (scan of the workgroup symbol counts) is k_scan_fused above.

// D3: decode every subsequence from its synchronised start and write the symbols.
// Symbols are staged in LDS so that the workgroup's contiguous output range leaves
// as 16-byte stores; stage byte j maps to out byte (o0 & ~15) + j.
__global__ __launch_bounds__(BLOCK) void k_dec_write(const uint32_t *__restrict__ words, uint64_t n_bytes, uint64_t n_subs,
                                                     uint32_t n_blocks, DecodeTables tb, const uint32_t *__restrict__ sub_state,
                                                     const unsigned long long *__restrict__ blk_off, uint64_t n_symbols,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ ticket, uint32_t special_only,
                                                     const uint32_t *__restrict__ void_flags) {
    if (void_flags && !dec_state_final(void_flags[1], void_flags[2], n_blocks)) return;  // see k_dec_write_reg
    const DecodeSmem m = carve_decode_smem<false>(tb);
    const int tid = threadIdx.x;
    stage_tables(m, tb);
    Prefetch pf;
    for (bool first_trip = true;; first_trip = false) {  // chunks: see k_dec_sync
        uint64_t b0;
        if (WRITE_TICKET && !special_only) {  // (a special-only launch must not eat tickets of k_dec_write_reg)
            __syncthreads();  // tables staged (first trip); everybody is done with scratch[7] and the stage
            if (tid == 0) m.scratch[7] = atomicAdd(ticket, WRITE_CHUNK);
            __syncthreads();
            b0 = m.scratch[7];
        } else {
            if (!first_trip) break;
            b0 = static_cast<uint64_t>(blockIdx.x) * WRITE_CHUNK;
            __syncthreads();
        }
        if (special_only) {  // one special block per workgroup (grid 3); the interior ones belong to k_dec_write_reg
            if (!first_trip) break;
            b0 = special_candidate(blockIdx.x, n_blocks);
            if (b0 >= n_blocks || !special_block(b0, n_bytes)) break;
        }
        if (b0 >= n_blocks) break;
        const uint64_t b1 = special_only ? b0 + 1 : (b0 + WRITE_CHUNK < n_blocks ? b0 + WRITE_CHUNK : n_blocks);
        prefetch_block(pf, words, b0, n_bytes);
    for (uint64_t b = b0; b < b1; ++b) {
        const uint64_t o0 = blk_off[b];
        if (o0 >= n_symbols) break;  // pad bits decoded past the declared length; offsets only grow from here
        const uint64_t sub_g = b * BLOCK + tid;
        const bool live = sub_g < n_subs;
        const uint32_t st = live ? sub_state[sub_g] : 0u;
        const uint32_t start = st & 0xffu, count = live ? (st >> 16) : 0u;

        commit_block(m, pf);
        if (b + 1 < b1) prefetch_block(pf, words, b + 1, n_bytes);
        uint32_t block_total;
        const uint32_t my_off = block_exclusive_scan(count, m.scratch, &block_total);  // its barrier also covers the staging

        uint64_t o1 = o0 + block_total;
        if (o1 > n_symbols) o1 = n_symbols;
        const uint32_t n_out = static_cast<uint32_t>(o1 - o0);
        const uint32_t phase = static_cast<uint32_t>(o0 & 15);  // stage offset of the first symbol
        const uint32_t lim = block_limit(n_bytes, b);
        uint8_t *out_base = out + (o0 - phase);

        // stage positions are `phase + symbol index`; windows of DEC_STAGE_BYTES of them.
        // Usual case: the block's symbols fit one window and none is clamped away.
        const bool one_window = phase + block_total <= DEC_STAGE_BYTES && n_out == block_total;
        for (uint32_t win = 0; win < phase + n_out; win += DEC_STAGE_BYTES) {
            const uint32_t win_hi = min(win + DEC_STAGE_BYTES, phase + n_out);
            const uint32_t my_lo = phase + my_off, my_hi = my_lo + count;
            if (one_window && lim == 0xffffffffu) {
                if (live && count) walk_subsequence<1, false, false>(m, tb, tid, start, lim, my_lo, 0, 0);
            } else if (live && my_lo < win_hi && my_hi > win) {
                walk_subsequence<2, true, false>(m, tb, tid, start, lim, my_lo, win, win_hi);
            }
            __syncthreads();
            const uint32_t lo_valid = max(win, phase);  // first stage position holding a symbol in this window
            for (uint32_t g = win + tid * 16; g < win_hi; g += BLOCK * 16) {
                if (g >= lo_valid && g + 16 <= win_hi) {
                    *reinterpret_cast<uint4 *>(out_base + g) = *reinterpret_cast<const uint4 *>(m.stage + (g - win));
                } else {
                    for (uint32_t k = max(g, lo_valid); k < min(g + 16, win_hi); ++k) out_base[k] = m.stage[k - win];
                }
            }
            __syncthreads();
        }
    }
    }
}

// --------------------------------------------------------------------------------
// launch wrappers (plain C++ callable; everything is enqueued on `stream`)
// --------------------------------------------------------------------------------
static inline size_t decode_smem_bytes(const DecodeTables &tb, bool with_stage, bool with_exits = true, bool with_stream = true) {
    const uint32_t sub_w = (((tb.n_sub << tb.sub_bits) + 7u) & ~7u) / 2;
    return ((with_stream ? DEC_SDATA_WORDS : 0) + (1u << tb.lut_bits) + sub_w + 64 + (with_exits ? BLOCK : 0) + 8) * sizeof(uint32_t) + (with_stage ? DEC_STAGE_BYTES + 16 : 0);
}

// Interior blocks take the register-window kernels; the LDS-window kernels keep the stream's first
// and last blocks (and streams of nothing else).
static bool use_reg_kernels(uint32_t n_blocks) { return n_blocks > 3; }

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
// < 64 SGPRs, where the query is exact), so that every workgroup gets within one tile of the same share.
template <typename K>
static uint32_t tile_grid(K kernel, uint32_t n_tiles) {
    int cus = 256;
    const int per_cu = resident_per_cu(reinterpret_cast<const void *>(kernel), 0, &cus);
    uint32_t g = per_cu >= 1 ? static_cast<uint32_t>(cus) * static_cast<uint32_t>(per_cu) : MAX_GRID;
    if (g > MAX_GRID) g = MAX_GRID;
    return n_tiles < g ? n_tiles : g;
}

void launch_hist(hipStream_t stream, const uint8_t *base, uint64_t lo, uint64_t hi, uint32_t rounds_per_tile, uint32_t n_tiles,
                 uint32_t *tile_hist, unsigned long long *block_hist, unsigned long long *hist, unsigned long long *host_hist, unsigned long long epoch, KernelEvents ev,
                 unsigned long long *hist_also) {
    // 4 workgroups per CU are resident (LDS): 1024 = one full batch (0.227 ms at 1 GiB; 2048 = two
    // batches 0.231; 1280 or 1536 = a full and a partial batch, 0.32-0.36)
    const uint32_t grid = n_tiles < 1024u ? n_tiles : 1024u;
    ET_LAUNCH_TIMED(k_hist_tiles, dim3(grid), dim3(HIST_BLOCK), 0, stream, ev, base, lo, hi, rounds_per_tile, n_tiles, tile_hist, block_hist, hist);
    hipLaunchKernelGGL(k_hist_reduce, dim3(128), dim3(BLOCK), 0, stream, block_hist, grid, hist, host_hist, epoch, hist_also);
}


void launch_words_to_host(hipStream_t stream, const void *d_src, uint32_t n_words, void *host_dst, unsigned long long *host_done, unsigned long long epoch) {
    hipLaunchKernelGGL(k_words_to_host, dim3(1), dim3(1024), 0, stream, static_cast<const uint32_t *>(d_src), n_words, static_cast<uint32_t *>(host_dst), host_done, epoch);
}

void launch_header_to_host(hipStream_t stream, const void *d_src, uint32_t n, void *host_dst, unsigned long long *host_done, unsigned long long epoch) {
    hipLaunchKernelGGL(k_header_to_host, dim3(1), dim3(1024), 0, stream, static_cast<const uint8_t *>(d_src), n, static_cast<uint32_t *>(host_dst), host_done, epoch);
}

void launch_tile_scan(hipStream_t stream, const uint32_t *tile_hist, uint32_t n_tiles, const uint8_t *lengths, const uint32_t *host_src, uint32_t *dev_dst,
                      uint32_t copy_words, unsigned long long *host_taken, unsigned long long taken_epoch, unsigned long long *tile_bits, unsigned long long *group_sum, uint32_t epoch, unsigned long long base_bit,
                      unsigned long long *tile_off, uint32_t *out32, const uint32_t *header_src, uint32_t header_words) {
    uint32_t grid = (n_tiles + 3) / 4;
    if (grid > MAX_GRID) grid = MAX_GRID;  // (one tile per wavefront, 8192 workgroups: no faster, r03)
    CodeLengths cl;
    for (int l = 0; l < 64; ++l) cl.packed[l] = lengths[4 * l] | (lengths[4 * l + 1] << 8) | (lengths[4 * l + 2] << 16) | (static_cast<uint32_t>(lengths[4 * l + 3]) << 24);
    hipLaunchKernelGGL(k_tile_bits, dim3(grid), dim3(BLOCK), 0, stream, tile_hist, n_tiles, cl, tile_bits, host_src, dev_dst, copy_words, host_taken, taken_epoch);
    const uint32_t groups = (n_tiles + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_fused<unsigned long long>, dim3(groups), dim3(1024), 0, stream, tile_bits, n_tiles, tile_off, group_sum, epoch, base_bit, out32, header_src,
                       header_words, static_cast<unsigned long long *>(nullptr), static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr),
                       static_cast<uint32_t *>(nullptr), 0xffffffffu, 0u, 0u, static_cast<const uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr), 0u);
}

void launch_encode(hipStream_t stream, const uint8_t *base, uint64_t lo, uint64_t hi, uint32_t rounds_per_tile, uint32_t n_tiles,
                   const unsigned long long *tile_off, const uint2 *enc_table, uint32_t max_len, uint32_t *out32, KernelEvents ev) {
    if (max_len > 32)
        ET_LAUNCH_TIMED(k_encode_tiles_long, dim3(tile_grid(k_encode_tiles_long, n_tiles)), dim3(BLOCK), 0, stream, ev, base, lo, hi, rounds_per_tile, n_tiles, tile_off, enc_table, out32);
    else if (max_len <= 31)  // a round emits at most 4096 * 31 / 32 + 2 words: fits a 4096-word ring
        ET_LAUNCH_TIMED(k_encode_tiles<4096>, dim3(tile_grid(k_encode_tiles<4096>, n_tiles)), dim3(BLOCK), 0, stream, ev, base, lo, hi, rounds_per_tile, n_tiles, tile_off, enc_table, out32);
    else
        ET_LAUNCH_TIMED(k_encode_tiles<8192>, dim3(tile_grid(k_encode_tiles<8192>, n_tiles)), dim3(BLOCK), 0, stream, ev, base, lo, hi, rounds_per_tile, n_tiles, tile_off, enc_table, out32);
}

// ---------------------------------------------------------------------------------
// The decode tables, filled on the device from the host's plan (et_tables.h TablePlan): one
// workgroup of 1024.  The host builders (et_tables.cpp) cost ~48 us per decode call, which a
// device-resident .et pays in full with the GPU idle (the header has to come to the host first);
// this kernel is ~5 us.  Entry for entry what build_decode_tables / build_step_table /
// build_write_step_table produce -- tests/test_gpu_parity.py compares the two.
//   phase 1  a wavefront per symbol: its span of the "which code prefixes this index" arrays
//            (LDS), or -- a code longer than the index -- its long-list entry, the first-level
//            escape of its prefix and its span of the second-level table
//   phase 2  a thread per first-level entry: the greedy walk over whole codes inside the index
// ---------------------------------------------------------------------------------
constexpr uint32_t BUILD_THREADS = 1024;
__global__ __launch_bounds__(BUILD_THREADS) void k_build_dec_tables(const TablePlan *plan, uint32_t *__restrict__ lut,
                                                                   uint32_t *__restrict__ longc, uint16_t *__restrict__ sub,
                                                                   uint8_t *__restrict__ sym_len, uint32_t *__restrict__ steps,
                                                                   uint32_t *__restrict__ wsteps, uint32_t *__restrict__ zero16) {
    __shared__ TablePlan plan_lds;  // (read per symbol below: from global memory each of those reads is a microsecond)
    static_assert(sizeof(TablePlan) % 4 == 0, "copied by words");
    for (uint32_t i = threadIdx.x; i < sizeof(TablePlan) / 4; i += BUILD_THREADS)
        reinterpret_cast<uint32_t *>(&plan_lds)[i] = reinterpret_cast<const uint32_t *>(plan)[i];
    if (zero16 && threadIdx.x < 16) zero16[threadIdx.x] = 0;  // the decode's flag words (saves the caller a memset launch)
    __syncthreads();
    plan = &plan_lds;
    __shared__ uint16_t single[1u << DEC_LUT_BITS_MAX];      // (len << 8) | sym of the code that prefixes a lut_bits index
    __shared__ uint8_t first_step[1u << DEC_STEP_BITS_MAX];  // its length, for a step_bits index
    __shared__ uint8_t of_lut[1u << DEC_LUT_BITS_MAX], of_step[1u << DEC_STEP_BITS_MAX], of_w[1u << DEC_LUT_BITS_MAX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t k = plan->lut_bits, ks = plan->step_bits, kw = plan->wstep_bits;  // kw == k (one index width for both write-side tables)
    const uint32_t sub_bits = plan->sub_bits, ssub_bits = plan->step_sub_bits, wsub_bits = plan->wstep_sub_bits;
    const uint32_t n = 1u << k, ns = 1u << ks, nw = 1u << kw;
    uint32_t *ssub = steps + ns, *wsub = wsteps + nw;
    for (uint32_t i = tid; i < n; i += BUILD_THREADS) {
        single[i] = 0;
        of_lut[i] = 0;
        of_w[i] = 0;
    }
    for (uint32_t i = tid; i < ns; i += BUILD_THREADS) {
        first_step[i] = 0;
        of_step[i] = 0;
    }
    for (uint32_t i = tid; i < (plan->n_sub << sub_bits); i += BUILD_THREADS) sub[i] = 0;
    for (uint32_t i = tid; i < (plan->n_step_sub << ssub_bits); i += BUILD_THREADS) ssub[i] = 0;
    for (uint32_t i = tid; i < (plan->n_wstep_sub << wsub_bits); i += BUILD_THREADS) wsub[i] = 0;
    if (tid < 256) sym_len[tid] = plan->length[tid];
    __syncthreads();
    for (uint32_t s = wave; s < 256; s += BUILD_THREADS / 64) {
        const uint32_t len = plan->length[s];
        if (!len) continue;
        const uint32_t code = plan->data[s], meta = (len << 8) | s;
        if (len <= k) {  // (kw == k: the write-step table shares `single`)
            const uint32_t lo = code << (k - len), span = 1u << (k - len);
            for (uint32_t i = lane; i < span; i += 64) single[lo + i] = static_cast<uint16_t>(meta);
        } else {
            const uint32_t prefix = code >> (len - k), rest_bits = len - k;
            if (lane == 0) {
                longc[2 * plan->long_idx[s]] = code << (32 - len);
                longc[2 * plan->long_idx[s] + 1] = meta;
            }
            const uint32_t t_lut = plan->lut_sub[s], t_w = plan->wstep_sub[s];
            if (t_lut) {
                if (lane == 0) of_lut[prefix] = static_cast<uint8_t>(t_lut);
                if (rest_bits <= sub_bits) {
                    const uint32_t lo = ((code & ((1u << rest_bits) - 1u)) << (sub_bits - rest_bits)) + ((t_lut - 1) << sub_bits);
                    for (uint32_t i = lane; i < (1u << (sub_bits - rest_bits)); i += 64) sub[lo + i] = static_cast<uint16_t>(meta);
                }
            }
            if (t_w) {
                if (lane == 0) of_w[prefix] = static_cast<uint8_t>(t_w);
                if (rest_bits <= wsub_bits) {
                    const uint32_t lo = ((code & ((1u << rest_bits) - 1u)) << (wsub_bits - rest_bits)) + ((t_w - 1) << wsub_bits);
                    for (uint32_t i = lane; i < (1u << (wsub_bits - rest_bits)); i += 64) wsub[lo + i] = (s << 16) | ((1u << 10) - len);
                }
            }
        }
        if (len <= ks) {
            const uint32_t lo = code << (ks - len), span = 1u << (ks - len);
            for (uint32_t i = lane; i < span; i += 64) first_step[lo + i] = static_cast<uint8_t>(len);
        } else {
            const uint32_t prefix = code >> (len - ks), rest_bits = len - ks, t_s = plan->step_sub[s];
            if (t_s) {
                if (lane == 0) of_step[prefix] = static_cast<uint8_t>(t_s);
                if (rest_bits <= ssub_bits) {
                    const uint32_t lo = ((code & ((1u << rest_bits) - 1u)) << (ssub_bits - rest_bits)) + ((t_s - 1) << ssub_bits);
                    for (uint32_t i = lane; i < (1u << (ssub_bits - rest_bits)); i += 64) ssub[lo + i] = (1u << 16) - len;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t max_syms = plan->max_syms;
    for (uint32_t v = tid; v < n; v += BUILD_THREADS) {
        {  // older format: up to max_syms whole codes
            uint32_t entry = 0, used = 0, cnt = 0;
            while (cnt < max_syms) {
                const uint32_t e = single[(v << used) & (n - 1)], len = e >> 8;
                if (!len || used + len > k) break;
                entry |= (e & 0xffu) << (8 * cnt);
                used += len;
                ++cnt;
            }
            if (cnt) entry |= (used << LUT_LEN_SHIFT) | (cnt << LUT_N_SHIFT);
            else if (of_lut[v]) entry = static_cast<uint32_t>(of_lut[v] - 1) | (1u << LUT_SUB_SHIFT);
            lut[v] = entry;
        }
        {  // write-step table: two symbols and what the step adds to the walk state
            uint32_t used = 0, cnt = 0, syms = 0;
            while (cnt < 2) {
                const uint32_t f = single[(v << used) & (n - 1)], len = f >> 8;
                if (!len || used + len > k) break;
                syms |= (f & 0xffu) << (16 + 8 * cnt);
                used += len;
                ++cnt;
            }
            wsteps[v] = cnt ? syms | (((cnt << 10) - used) & 0xffffu) : (static_cast<uint32_t>(of_w[v]) << 24) | WSTEP_ESCAPE;
        }
    }
    for (uint32_t v = tid; v < ns; v += BUILD_THREADS) {
        uint32_t used = 0, cnt = 0;
        for (;;) {
            const uint32_t len = first_step[(v << used) & (ns - 1)];
            if (!len || used + len > ks) break;
            used += len;
            ++cnt;
        }
        steps[v] = cnt ? (static_cast<uint32_t>(first_step[v]) << 28) + (cnt << 16) - used : STEP_ESCAPE + (static_cast<uint32_t>(of_step[v]) << 28);
    }
}

void launch_build_dec_tables(hipStream_t stream, const TablePlan *d_plan, uint32_t *lut, uint32_t *longc, uint16_t *sub, uint8_t *sym_len,
                             uint32_t *steps, uint32_t *wsteps, uint32_t *zero16) {
    hipLaunchKernelGGL(k_build_dec_tables, dim3(1), dim3(BUILD_THREADS), 0, stream, d_plan, lut, longc, sub, sym_len, steps, wsteps, zero16);
}

// Grid of a chunked decode kernel: one workgroup per chunk, or -- ticketed -- as many
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

// `special` = the stream the three-workgroup launch goes to: the side lane's (made to wait
// for everything enqueued on `stream` so far) or `stream` itself; join_special makes
// `stream` wait for it again.
// (fork_mark first, then the big launch on `stream`, then fork_special: the big kernel is
// handed to the GPU two API calls earlier and the side lane still waits only for what was
// enqueued before the mark.)
static void fork_mark(const SideLane *side, hipStream_t stream) {
    if (side) (void)hipEventRecord(side->fork, stream);
}
static hipStream_t fork_special(const SideLane *side, hipStream_t stream) {
    if (!side) return stream;
    (void)hipStreamWaitEvent(side->stream, side->fork, 0);
    return side->stream;
}
static void join_special(const SideLane *side, hipStream_t stream) {
    if (!side) return;
    (void)hipEventRecord(side->join, side->stream);
    (void)hipStreamWaitEvent(stream, side->join, 0);
}

void launch_dec_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs,
                     const DecodeTables &tb, uint32_t iter, uint32_t max_trips,
                     uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_count, uint32_t *changed, uint32_t *ticket, uint32_t flags,
                     uint32_t *worklist, uint32_t *n_work, const SideLane *side, bool ticket_is_zero, KernelEvents ev) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    const uint32_t n_chunks = (n_blocks + SYNC_CHUNK - 1) / SYNC_CHUNK;
    const size_t smem = decode_smem_bytes(tb, false);
    if (use_reg_kernels(n_blocks)) {
        const size_t smem_reg = (step_table_words(tb) + BLOCK + 8) * sizeof(uint32_t);
        constexpr uint32_t chunk = 4;  // blocks per ticket; measured 1 / 4 / 8 / 16
        if (iter == 0 && n_blocks >= 16) {
            if (!ticket_is_zero) (void)hipMemsetAsync(ticket, 0, sizeof(uint32_t), stream);
            constexpr uint32_t chunk2 = 4;  // superblocks per ticket; measured 1 / 2 / 4 / 8 / 16: 0.54 / 0.37 / 0.35 / 0.37 / 0.44 ms
            fork_mark(side, stream);
            ET_LAUNCH_TIMED(k_dec_sync_reg2, dim3(decode_grid(k_dec_sync_reg2, smem_reg, (n_blocks / 2 + chunk2 - 1) / chunk2, true)), dim3(BLOCK), smem_reg, stream, ev, words, n_bytes, n_blocks, step_table_args(tb), sub_state, blk_exit, blk_count, changed, ticket, max_trips, chunk2);
            const hipStream_t special = fork_special(side, stream);
            hipLaunchKernelGGL(k_dec_sync<true>, dim3(8), dim3(BLOCK), smem, special, words, n_bytes, first_bit, n_subs, n_blocks, tb, sub_state, blk_exit, blk_count, changed, ticket, max_trips, flags | DEC_SPECIAL_ONLY | DEC_SPECIAL_SUPER);
            join_special(side, stream);
        } else if (iter == 0) {
            if (!ticket_is_zero) (void)hipMemsetAsync(ticket, 0, sizeof(uint32_t), stream);
            fork_mark(side, stream);
            ET_LAUNCH_TIMED((k_dec_sync_reg<true, true>), dim3(decode_grid(k_dec_sync_reg<true, true>, smem_reg, (n_blocks + chunk - 1) / chunk, true)), dim3(BLOCK), smem_reg, stream, ev, words, n_bytes, n_blocks, step_table_args(tb), sub_state, blk_exit, blk_count, changed, ticket, max_trips, chunk, static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr));
            const hipStream_t special = fork_special(side, stream);
            hipLaunchKernelGGL(k_dec_sync<true>, dim3(3), dim3(BLOCK), smem, special, words, n_bytes, first_bit, n_subs, n_blocks, tb, sub_state, blk_exit, blk_count, changed, ticket, max_trips, flags | DEC_SPECIAL_ONLY);
            join_special(side, stream);
        } else {
            if (worklist) {  // n_work zeroed by the caller
                hipLaunchKernelGGL(k_dec_check, dim3((n_blocks + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, sub_state, blk_exit, n_blocks, worklist, n_work);
                hipLaunchKernelGGL((k_dec_sync_reg<false, false>), dim3(n_blocks < 512 ? n_blocks : 512), dim3(BLOCK), smem_reg, stream, words, n_bytes, n_blocks, step_table_args(tb), sub_state, blk_exit, blk_count, changed, ticket, max_trips, 1u, static_cast<const uint32_t *>(worklist), static_cast<const uint32_t *>(n_work));
            } else {
                hipLaunchKernelGGL((k_dec_sync_reg<false, false>), dim3(n_blocks), dim3(BLOCK), smem_reg, stream, words, n_bytes, n_blocks, step_table_args(tb), sub_state, blk_exit, blk_count, changed, ticket, max_trips, 1u, static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr));
            }
            // (not on the side lane: the fork/join events cost more than these ~5 us)
            hipLaunchKernelGGL(k_dec_sync<false>, dim3(3), dim3(BLOCK), smem, stream, words, n_bytes, first_bit, n_subs, n_blocks, tb, sub_state, blk_exit, blk_count, changed, ticket, max_trips, flags | DEC_SPECIAL_ONLY);
        }
        return;
    }
    if (SYNC_TICKET) (void)hipMemsetAsync(ticket, 0, sizeof(uint32_t), stream);
    if (iter == 0)
        ET_LAUNCH_TIMED(k_dec_sync<true>, dim3(decode_grid(k_dec_sync<true>, smem, n_chunks, SYNC_TICKET)), dim3(BLOCK), smem, stream, ev, words, n_bytes, first_bit, n_subs, n_blocks, tb, sub_state, blk_exit, blk_count, changed, ticket, max_trips, flags);
    else
        hipLaunchKernelGGL(k_dec_sync<false>, dim3(decode_grid(k_dec_sync<false>, smem, n_chunks, SYNC_TICKET)), dim3(BLOCK), smem, stream, words, n_bytes, first_bit, n_subs, n_blocks, tb, sub_state, blk_exit, blk_count, changed, ticket, max_trips, flags);
}

// Exhaustive synchronisation (see k_dec_maps).  Workspaces: lane_maps n_subs * stride,
// blk_maps / blk_in per block, grp_maps / grp_in per 256 blocks.  Two halves: the maps up to
// one per 256 blocks (launch_dec_maps), and, once the input start is known, the way back down
// and the counting walk (launch_dec_resolve).  A single GPU runs them back to back; ranges of
// a stream split over GPUs exchange their composed maps in between.
void launch_dec_maps(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, bool have_start, uint64_t n_subs,
                     const DecodeTables &tb, uint32_t n_starts, uint32_t map_stride, uint8_t *lane_maps, uint8_t *blk_maps, uint8_t *grp_maps) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    const uint32_t n_groups = (n_blocks + 255) / 256;
    const size_t smem = decode_smem_bytes(tb, true);
    const bool reg = use_reg_kernels(n_blocks) && tb.steps != nullptr;
    const size_t smem_reg = (step_table_words(tb) + BLOCK * 32 / 4 + BLOCK + 8) * sizeof(uint32_t);
    if (reg) hipLaunchKernelGGL(k_dec_maps_reg, dim3(n_blocks), dim3(BLOCK), smem_reg, stream, words, n_bytes, n_blocks, step_table_args(tb), n_starts, map_stride, lane_maps, blk_maps);
    hipLaunchKernelGGL(k_dec_maps, dim3(reg ? 3 : n_blocks), dim3(BLOCK), smem, stream, words, n_bytes, first_bit, n_subs, tb, n_starts, map_stride, lane_maps, blk_maps, reg ? 1u : 0u, have_start ? 1u : 0u);
    hipLaunchKernelGGL(k_dec_compose, dim3(n_groups), dim3(BLOCK), 0, stream, blk_maps, n_blocks, grp_maps);
}

void launch_dec_resolve(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, bool const_first, uint64_t n_subs,
                        const DecodeTables &tb, uint32_t map_stride, const uint8_t *lane_maps, const uint8_t *blk_maps, const uint8_t *grp_maps,
                        uint8_t *blk_in, uint8_t *grp_in, uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_count) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    const uint32_t n_groups = (n_blocks + 255) / 256;
    const size_t smem = decode_smem_bytes(tb, true);
    const bool reg = use_reg_kernels(n_blocks) && tb.steps != nullptr;
    const size_t smem_reg = (step_table_words(tb) + BLOCK * 32 / 4 + BLOCK + 8) * sizeof(uint32_t);
    // one workgroup walks all group maps (256 per LDS refill), then every group resolves its blocks
    hipLaunchKernelGGL(k_dec_chain, dim3(1), dim3(BLOCK), 0, stream, grp_maps, n_groups, static_cast<const uint8_t *>(nullptr), first_bit, grp_in);
    hipLaunchKernelGGL(k_dec_chain, dim3(n_groups), dim3(BLOCK), 0, stream, blk_maps, n_blocks, grp_in, 0u, blk_in);
    if (reg) hipLaunchKernelGGL(k_dec_resolve_reg, dim3(n_blocks), dim3(BLOCK), smem_reg, stream, words, n_bytes, n_blocks, step_table_args(tb), map_stride, lane_maps, blk_in, sub_state, blk_exit, blk_count);
    hipLaunchKernelGGL(k_dec_resolve, dim3(reg ? 3 : n_blocks), dim3(BLOCK), smem, stream, words, n_bytes, first_bit, n_subs, tb, map_stride, lane_maps, blk_in, sub_state, blk_exit, blk_count, reg ? 1u : 0u, const_first ? 1u : 0u);
}

void launch_dec_exhaustive(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs,
                           const DecodeTables &tb, uint32_t n_starts, uint32_t map_stride, uint8_t *lane_maps, uint8_t *blk_maps,
                           uint8_t *grp_maps, uint8_t *blk_in, uint8_t *grp_in, uint32_t *sub_state, uint32_t *blk_exit,
                           uint32_t *blk_count) {
    launch_dec_maps(stream, words, n_bytes, first_bit, true, n_subs, tb, n_starts, map_stride, lane_maps, blk_maps, grp_maps);
    launch_dec_resolve(stream, words, n_bytes, first_bit, true, n_subs, tb, map_stride, lane_maps, blk_maps, grp_maps, blk_in, grp_in, sub_state, blk_exit,
                       blk_count);
}

void launch_dec_scan(hipStream_t stream, const uint32_t *blk_count, uint32_t n_blocks, unsigned long long *group_sum, uint32_t epoch,
                     unsigned long long *blk_off, unsigned long long *total_copy, const uint32_t *verify_state, const uint32_t *verify_exit,
                     uint32_t *verify_flag, uint32_t verify_first, const uint32_t *report_src, uint32_t *report_dst, bool verify_rows, uint32_t report_epoch) {
    const uint32_t groups = (n_blocks + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_fused<uint32_t>, dim3(groups), dim3(1024), 0, stream, blk_count, n_blocks, blk_off, group_sum, epoch, 0ull, static_cast<uint32_t *>(nullptr),
                       static_cast<const uint32_t *>(nullptr), 0u, total_copy, verify_state, verify_exit, verify_flag, verify_first,
                       verify_rows ? 1u : static_cast<uint32_t>(BLOCK), verify_rows ? 0xffffffffu : 0xffu, report_src, report_dst, report_epoch);
}

void launch_dec_write(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint64_t n_subs, const DecodeTables &tb,
                      const uint32_t *sub_state,
                      const unsigned long long *blk_off, uint64_t n_symbols, uint8_t *out, uint32_t *ticket, const SideLane *side, bool ticket_is_zero,
                      const uint32_t *void_flags, KernelEvents ev, const uint64_t *chain, uint32_t n_chain, uint32_t chain_max_len) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    if (chain) {  // every block, one launch, no side lane, no ticket; `tb` is not looked at
        // 8 wavefronts per workgroup share the tables (17 KiB) beside their 4 KiB stages: 3 workgroups = 24 wavefronts per CU
#ifdef ET_PROBE_FUSED_OCC  // (round-4 probe, DESIGN section 4: the write pass at the occupancy a kernel that also holds D1's tree table would have -- one workgroup of ET_PROBE_FUSED_OCC wavefronts per CU)
        constexpr int WAVES = ET_PROBE_FUSED_OCC;
#else
        constexpr int WAVES = 8;  // (12 x 2 per CU the same; 16 x 2 with 3.8 KiB stages, 32 wavefronts per CU, the same too: 0.441-0.445 ms; 4 x 4 or 16 x 1: 0.56)
#endif
        size_t smem_wave = ((static_cast<size_t>(n_chain) * 8 + 15) & ~static_cast<size_t>(15)) + WAVES * WV_STAGE_ALLOC;
#ifdef ET_PROBE_FUSED_OCC
        if (smem_wave < 84u * 1024u) smem_wave = 84u * 1024u;  // more than half the LDS: one workgroup per CU, as beside a 47 KiB tree table
#endif
        const uint32_t n_units = (n_blocks * 4 + WAVES - 1) / WAVES;
        ET_LAUNCH_TIMED(k_dec_write_wave<WAVES>, dim3(decode_grid(k_dec_write_wave<WAVES>, smem_wave, n_units, true, 64 * WAVES)), dim3(64 * WAVES), smem_wave, stream, ev, words, n_bytes, n_blocks, reinterpret_cast<const uint2 *>(chain), n_chain, sub_state, blk_off, n_symbols, out, void_flags, n_subs, chain_max_len);
        return;
    }
    if (!ticket_is_zero) (void)hipMemsetAsync(ticket, 0, sizeof(uint32_t), stream);
    const uint32_t n_chunks = (n_blocks + WRITE_CHUNK - 1) / WRITE_CHUNK;
    const size_t smem = decode_smem_bytes(tb, true, false);
    if (use_reg_kernels(n_blocks)) {
        const size_t smem_reg = (step_table_words(tb) + 64 + 8) * sizeof(uint32_t) + DEC_STAGE_BYTES + 16;
        fork_mark(side, stream);
        ET_LAUNCH_TIMED(k_dec_write_reg, dim3(decode_grid(k_dec_write_reg, smem_reg, n_chunks, true)), dim3(BLOCK), smem_reg, stream, ev, words, n_bytes, n_blocks, step_table_args(tb), tb.sym_len, sub_state, blk_off, n_symbols, out, ticket, void_flags);
        const hipStream_t special = fork_special(side, stream);
        hipLaunchKernelGGL(k_dec_write, dim3(3), dim3(BLOCK), smem, special, words, n_bytes, n_subs, n_blocks, tb, sub_state, blk_off, n_symbols, out, ticket, 1u, void_flags);
        join_special(side, stream);
        return;
    }
    ET_LAUNCH_TIMED(k_dec_write, dim3(decode_grid(k_dec_write, smem, n_chunks, WRITE_TICKET)), dim3(BLOCK), smem, stream, ev, words, n_bytes, n_subs, n_blocks, tb, sub_state, blk_off, n_symbols, out, ticket, 0u, void_flags);
}

}  // namespace et

// et_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the entreepy
// Huffman path.  Pure integer/bit work: no MFMA; the bounds are HBM bandwidth, the
// LDS pipe (atomics + table reads) and VALU issue.
//
//   encode.zig:43-47   -> k_hist_tiles (+ k_hist_reduce)          "K1"
//   encode.zig:308-313 -> k_tile_bits, k_scan_fused               "K2" (the serial
//                         bits_written counter turned into a scan over tiles)
//   encode.zig:303-315 -> k_encode_tiles / k_encode_tiles_long    "K4"
//   decode.zig:143-203 -> "D1" k_tw_sync (et_treewalk.hip), "D2" k_scan_fused (+ verification, report to the host),
//   decode.zig:186     -> "D3" k_dec_write_wave (chained lookup tables, wavefront-owned quarters)
//   uniform-like bytes -> k_row_sync / k_row_write (et_rowsync.hip); everything else of the decode --
//                         the round-1 sweeps, the exit maps, their write kernels and table builder -- et_kernels_fallback.hip
//
// Geometry: workgroups of 256 threads (4 wavefronts of 64); K1 uses 512 on one set of counters.
// Encode side: a "round" is 4 KiB of input, one 16-byte load per lane, fully
// coalesced; a "tile" is 1..16 consecutive rounds and is the unit for which K1
// leaves a 256-bin histogram and K4 gets a start bit offset.
#include "et_kernels_common.h"
#include "et_treewalk.h"

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

// MODE 3 of walk_write_chain: a lane's strip -- c <= 66 bytes at LDS address `at` (4-byte aligned) -- to dst, wherever that lies
// (gfx950 takes unaligned global stores; the pieces are 16, 8, 4, 2, 1 bytes and never reach past dst + c: behind it lies the next lane's output).
__device__ __forceinline__ void strip_flush(uint32_t at, uint32_t c, uint8_t *dst) {
#ifdef ET_PROBE_STRIPS_NO_FLUSH  // (timing probe, wrong output: what the strips' way out costs -- 0.27 -> 0.13 ms at 90 % zeros, 0.20 -> 0.13 at 97 %, 256 MiB)
    (void)at, (void)c, (void)dst;
    return;
#endif
    typedef __attribute__((address_space(3))) uint32_t lds_u32_;
    typedef __attribute__((address_space(3))) uint16_t lds_u16_;
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    uint32_t o = 0;
    for (; o + 16 <= c; o += 16) {
        u32x4_ v;
        v.x = *reinterpret_cast<const lds_u32_ *>(static_cast<uintptr_t>(at + o));
        v.y = *reinterpret_cast<const lds_u32_ *>(static_cast<uintptr_t>(at + o + 4));
        v.z = *reinterpret_cast<const lds_u32_ *>(static_cast<uintptr_t>(at + o + 8));
        v.w = *reinterpret_cast<const lds_u32_ *>(static_cast<uintptr_t>(at + o + 12));
        __builtin_memcpy(dst + o, &v, 16);
    }
    if (c & 8u) {
        uint2 v;
        v.x = *reinterpret_cast<const lds_u32_ *>(static_cast<uintptr_t>(at + o));
        v.y = *reinterpret_cast<const lds_u32_ *>(static_cast<uintptr_t>(at + o + 4));
        __builtin_memcpy(dst + o, &v, 8);
        o += 8;
    }
    if (c & 4u) {
        const uint32_t v = *reinterpret_cast<const lds_u32_ *>(static_cast<uintptr_t>(at + o));
        __builtin_memcpy(dst + o, &v, 4);
        o += 4;
    }
    if (c & 2u) {
        const uint16_t v = *reinterpret_cast<const lds_u16_ *>(static_cast<uintptr_t>(at + o));
        __builtin_memcpy(dst + o, &v, 2);
        o += 2;
    }
    if (c & 1u) dst[o] = *reinterpret_cast<const lds_u8 *>(static_cast<uintptr_t>(at + o));
}

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
                                                 uint32_t lo, uint32_t hi, uint32_t stage_off, uint8_t *gdst = nullptr) {
    constexpr bool M1 = MODE == 1 || MODE == 3;  // MODE 3: as MODE 1, into the lane's own strip, which leaves for gdst every two words (below)
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
        if (M1) {                                                                           \
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
            if (M1) CH_PUT1(X >> 10, e.x >> 16);        \
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
        if (M1) CH_STEP_FAST(hi_, lo_)                \
        else CH_STEP_SAFE(hi_, lo_)                   \
    }                                                 \
    X += 32;
// (timing probe, wrong output: the strips' pieces side by side and 16-byte aligned -- lane l's at 64 l of a 4 KiB row per flush -- instead of where they belong)
#ifdef ET_PROBE_STRIPS_DENSE_DST
#define ET_PROBE_STRIPS_STEP(c_) 4096u
#else
#define ET_PROBE_STRIPS_STEP(c_) (c_)
#endif
// MODE 3: two words hold at most 64 codewords; what they left in the strip goes to its place in the output and the strip begins anew
#define CH_FLUSH()                                                              \
    if (MODE == 3) {                                                            \
        const uint32_t c_ = (X >> 10) - pos0;                                   \
        strip_flush(pos0 + 1u, c_, gdst);                                       \
        gdst += ET_PROBE_STRIPS_STEP(c_);                                       \
        X = (X & 1023u) | (pos0 << 10);                                         \
    }
    CH_WORD_FAST(W[3], W[4])  // only lanes that start at bit 0
    CH_WORD_FAST(W[4], W[5])
    CH_FLUSH()
    CH_WORD_FAST(W[5], W[6])
    CH_WORD_FAST(W[6], W[7])
    CH_FLUSH()
    CH_WORD_FAST(W[7], W[8])
    CH_WORD_FAST(W[8], W[9])
    CH_FLUSH()
    CH_WORD_FAST(W[9], W[10])
    // positions 193..224: a 32-bit code that begins at 224 is the lane's last, so with such codes about this word is SAFE
    if (M1 && !cw.has_len32) {
        while (CH_G >= 64u) CH_STEP_FAST(W[10], W[11])
    } else {
        while (CH_G >= 64u) CH_STEP_SAFE(W[10], W[11])
    }
    X += 32;
    CH_FLUSH()
    // the last word: whole-index steps while the lookup's index bits all lie inside the subsequence, then one codeword at a time
    while (CH_G + (H >> 24) >= 96u) CH_STEP_SAFE(W[11], W[12])
    for (uint32_t k = 0; k < 40 && (CH_G > 64u || (CH_G == 64u && CH_MID)); ++k) CH_STEP_ONE(W[11], W[12])
    // a codeword that began inside the subsequence and is still open behind it: its remaining tables
    if (CH_MID) {
        X += 32;
        for (uint32_t k = 0; k < 40 && CH_MID; ++k) CH_STEP_ONE(W[12], 0u)
    }
    CH_FLUSH()
#undef CH_FLUSH
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
// STRIPS (a second instantiation, for streams with more than ~70 symbols per 256-bit subsequence -- a dominant symbol with a 1- or 2-bit
// codeword, alphabets of a few symbols): a quarter whose output does not fit the stage is not walked once per 4 KiB window of its
// output (3-4 times on such streams) but ONCE, every lane into a strip of its own (WS_STRIDE bytes of what is the stage otherwise),
// and the strips leave for their places in the output every two stream words (walk_write_chain<3>): 64-byte pieces instead of
// 16-byte chunks of a contiguous stage, but a third or a quarter of the lookups.  The text kernel is the instantiation without.
constexpr uint32_t WS_STRIDE = 68;                 // a strip: the <= 64 codewords of two words + the two stray bytes of a fast step, a multiple of 4
constexpr uint32_t WS_ALLOC = 64 * WS_STRIDE;      // (>= WV_STAGE_ALLOC: one-window quarters stage as ever)
template <int WAVES, bool STRIPS = false>
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
    const uint32_t stage_off = tab_bytes + wv * (STRIPS ? WS_ALLOC : WV_STAGE_ALLOC);
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
            bool strips_unit = false;
            if (STRIPS) strips_unit = span && !one_window && !this_edge && n_out == wave_total;  // (wavefront-uniform)
            if (strips_unit) {
                if (count) walk_write_chain<3>(cw, smem8, W, start, lds_stage + static_cast<uint32_t>(lane) * WS_STRIDE - 1u, 0, 0, 0,
#ifdef ET_PROBE_STRIPS_DENSE_DST
                                                     out + (ow & ~static_cast<uint64_t>(15)) + static_cast<uint32_t>(lane) * 64u);
#else
                                                     out + ow + my_off);
#endif
                WV_LDS_ORDER()
            } else if (span) {
                for (;; win += WV_STAGE) {
                    const uint32_t win_hi = min(win + WV_STAGE, span);
                    WV_WINDOW_WALK(win, win_hi)
                    if (win_hi == span) break;  // the last window: its stores come behind the take
                    WV_WINDOW_STORE(win, win_hi)
                }
            }
            WV_TAKE()
            if (span && !strips_unit) WV_WINDOW_STORE(win, span)
#undef WV_WINDOW_STORE
#undef WV_LDS_ORDER
#undef WV_WINDOW_WALK
        }
#undef WV_TAKE
#undef WV_FETCH
    }
}

// --------------------------------------------------------------------------------
// launch wrappers (plain C++ callable; everything is enqueued on `stream`)
// --------------------------------------------------------------------------------
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
                      const uint32_t *void_flags, KernelEvents ev, const uint64_t *chain, uint32_t n_chain, uint32_t chain_max_len, bool strips) {
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
        if (strips) {  // (the caller's estimate from the header: many symbols per subsequence)
            const size_t smem_strips = ((static_cast<size_t>(n_chain) * 8 + 15) & ~static_cast<size_t>(15)) + WAVES * WS_ALLOC;
            ET_LAUNCH_TIMED((k_dec_write_wave<WAVES, true>), dim3(decode_grid(k_dec_write_wave<WAVES, true>, smem_strips, n_units, true, 64 * WAVES)), dim3(64 * WAVES), smem_strips, stream, ev, words, n_bytes, n_blocks, reinterpret_cast<const uint2 *>(chain), n_chain, sub_state, blk_off, n_symbols, out, void_flags, n_subs, chain_max_len);
            return;
        }
        ET_LAUNCH_TIMED(k_dec_write_wave<WAVES>, dim3(decode_grid(k_dec_write_wave<WAVES>, smem_wave, n_units, true, 64 * WAVES)), dim3(64 * WAVES), smem_wave, stream, ev, words, n_bytes, n_blocks, reinterpret_cast<const uint2 *>(chain), n_chain, sub_state, blk_off, n_symbols, out, void_flags, n_subs, chain_max_len);
        return;
    }
    launch_dec_write_fallback(stream, words, n_bytes, n_subs, tb, sub_state, blk_off, n_symbols, out, ticket, side, ticket_is_zero, void_flags, ev);
}


}  // namespace et

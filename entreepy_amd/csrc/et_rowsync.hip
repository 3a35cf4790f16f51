// et_rowsync.hip -- one-pass synchronisation of 8-bit near-fixed-length codes (et_rowsync.h), gfx950 / wave64.
//
//   decode.zig:143-203 -> k_row_sync: where does the first codeword of every 256-bit subsequence begin, and how many begin in it?
//
// A lane owns a subsequence = 32 rows (bytes) x 8 columns (bit offsets).  It
//   1. transposes its 32 bytes so that four rows sit in one register (24 v_perm),
//   2. turns them into 15 bit PLANES (bit j of plane c = bit c of row j, or of the row below for c >= 8: 2 instructions per four
//      rows and plane) and works out S[r], r = 0..7: bit j of S[r] says "the 7 bits at row j, column r are a 7-bit codeword" --
//      a comparator of planes r .. r + 6 against t from the top bit down, 32 rows per instruction, no lookups (a first version
//      compared byte-parallel, four rows per instruction: 425 of the lane's 760 VALU instructions; this one takes ~250),
//   3. follows, for each of the 8 columns a walk can enter the subsequence in, the path to the subsequence's end: a path stays
//      in its column until the next set bit of S[column] (v_ffbl), then moves one column to the left.  ALL lanes' paths visit the
//      columns in the same order, so iteration t of the walk from column c works on the register S[(c - t) & 7] in every lane:
//      no indexing.  Result: the lane's MAP (entry column -> exit column, a byte each: 8 bytes) and the codeword count per entry,
//   4. composes maps: a map is 8 bytes of values 0..7, and v_perm_b32 IS the composition of two such maps -- a wavefront's
//      inclusive scan of its 64 maps is 6 x (2 shuffles + 2 v_perm).
// A workgroup takes chunks of ROW_CHUNK_BLOCKS 8 KiB blocks by ticket, keeps the lanes' prefix maps in LDS, publishes the
// chunk's map (one 8-byte word: the map's bytes, the kind in two spare bits), looks back over the chunks before it --
// 64 at a time, composing what they published -- until it meets one whose entry column is known, publishes its own, and then
// every lane reads its start off its prefix map.  The stream is read once; nothing but the results leaves the chip.
// (A ticket, not the block index, orders the chunks: a chunk only ever waits for chunks with smaller tickets, and those were
// taken by workgroups that are running.)
// At the file's end: fixed-length codes (k_fixed_sync, k_fixed_write) -- the same family's other extreme, where nothing has to be found.
#include "et_rowsync.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace et {

namespace {

constexpr int RS_THREADS = 256;
constexpr uint32_t RS_CH = ROW_CHUNK_BLOCKS;
constexpr uint32_t RS_WAVES = RS_CH * 4;        // wavefront-blocks per chunk
constexpr uint32_t ID_LO = 0x03020100u, ID_HI = 0x07060504u;  // the identity map
constexpr uint32_t RS_SPINS = 1u << 24;         // looks at a word that never comes before a chunk gives up (seconds)
constexpr unsigned long long KIND_MAP = 1ull << 3, KIND_ENTRY = 2ull << 3;  // in byte 0 of a published word, above the column's 3 bits

__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// a & ~b (left to the compiler: it inverts each plane once and ands -- an explicit v_bfi_b32 per use measured 3 % SLOWER, 2.52 against 2.46 ms)
__device__ __forceinline__ uint32_t rs_and_not(uint32_t a, uint32_t b) { return a & ~b; }
// byte `idx` (0..7) of the 8-byte map {lo, hi}
__device__ __forceinline__ uint32_t map_at(uint32_t lo, uint32_t hi, uint32_t idx) { return perm(hi, lo, idx | 0x0c0c0c00u); }

// stream word idx as it lies in memory, zero outside the stream
__device__ __attribute__((noinline)) uint32_t rs_load_guarded(const uint32_t *__restrict__ words, uint64_t idx, uint64_t n_bytes) {
    const uint64_t b0 = idx * 4;
    if (b0 + 4 <= n_bytes) return words[idx];
    uint32_t v = 0;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(words);
    for (int k = 0; k < 4; ++k)
        if (b0 + k < n_bytes) v |= static_cast<uint32_t>(bytes[b0 + k]) << (8 * k);
    return v;
}

// One codeword at a time from bit `pos` to the subsequence's end, never past the stream's (the few lanes the stream begins
// and ends in).  A codeword cut by the stream's end is nobody's; the exit then points at the stream's end, where the next
// subsequence's walk stops at once (as walk_subsequence's, et_kernels_fallback.hip).
// -> exit column | codewords << 8
__device__ __attribute__((noinline)) uint32_t rs_slow_walk(const uint8_t *__restrict__ bytes, uint64_t n_bytes, uint64_t pos, uint64_t sub_end, uint32_t t) {
    const uint64_t n_bits = n_bytes * 8;
    uint32_t n = 0;
    bool cut = false;
    while (pos < sub_end) {
        if (pos + 7 > n_bits) {
            cut = true;
            break;
        }
        const uint64_t at = pos >> 3;
        const uint32_t sh = static_cast<uint32_t>(pos & 7);
        const uint32_t b0 = bytes[at], b1 = at + 1 < n_bytes ? bytes[at + 1] : 0u;
        const uint32_t w8 = (((b0 << 8) | b1) >> (8 - sh)) & 0xffu;
        const uint32_t len = (w8 >> 1) < t ? 7u : 8u;
        if (pos + len > n_bits) {
            cut = true;
            break;
        }
        ++n;
        pos += len;
    }
    const uint32_t exit_col = cut ? (n_bits > sub_end ? static_cast<uint32_t>(n_bits - sub_end) : 0u) : static_cast<uint32_t>(pos - sub_end);
    return exit_col | (n << 8);
}

__device__ __forceinline__ uint32_t rs_wave_sum_scan(uint32_t x) {  // inclusive prefix sum over the wavefront (DPP)
#define RS_DPP_ADD(ctrl_, mask_) x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), ctrl_, mask_, 0xf, false))
    RS_DPP_ADD(0x111, 0xf);
    RS_DPP_ADD(0x112, 0xf);
    RS_DPP_ADD(0x114, 0xf);
    RS_DPP_ADD(0x118, 0xf);
    RS_DPP_ADD(0x142, 0xa);
    RS_DPP_ADD(0x143, 0xc);
#undef RS_DPP_ADD
    return x;
}

struct RsShared {
    unsigned long long lane_pre[RS_CH][RS_THREADS];  // a lane's INCLUSIVE prefix map inside its wavefront
    unsigned long long lane_cnt[RS_CH][RS_THREADS];  // codewords that begin in the lane's subsequence, per entry column (a byte each)
    unsigned long long wave_map[RS_WAVES];           // a wavefront's map ...
    unsigned long long wave_pre[RS_WAVES];           // ... and the map from the chunk's entry to the wavefront's
    uint32_t wave_count[RS_WAVES];
    uint32_t chunk, entry;
};

}  // namespace

#ifndef ET_ROW_SYNC_ATTR
#define ET_ROW_SYNC_ATTR
#endif
__global__ __launch_bounds__(RS_THREADS) ET_ROW_SYNC_ATTR void k_row_sync(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, uint32_t n_blocks,
                                                         uint32_t n_chunks, uint32_t code_t, unsigned long long *__restrict__ pub, uint32_t *__restrict__ ticket,
                                                         uint32_t *__restrict__ fault, uint32_t *__restrict__ sub_state, uint32_t *__restrict__ blk_exit,
                                                         uint32_t *__restrict__ blk_count, uint32_t flags, unsigned long long *__restrict__ map_out) {
    // flags (a RANGE of a stream split over GPUs): ROW_MAP_ONLY -- leave the range's MAP (entry column -> exit column) in *map_out and
    // nothing else: every chunk publishes its map, the last one composes them all; ROW_START_UNKNOWN -- the first subsequence is a
    // subsequence like any other (its first codeword may begin in any column).
    const bool map_only = (flags & ROW_MAP_ONLY) != 0, start_known = !(flags & ROW_START_UNKNOWN);
    __shared__ RsShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t n_bits = n_bytes * 8, n_words_full = n_bytes / 4;
    // the first subsequence whose last row's codeword may be cut by the stream's end ((s + 1) * 256 + 8 > n_bits): the lanes from there on
    // walk one codeword at a time (a scalar, worked out once: the lanes' own test is one compare)
    const uint64_t first_tail_sub = n_bits >= 264 ? (n_bits - 8) / 256 : 0;
    for (;;) {
        __syncthreads();  // everybody is done with the chunk before
        if (tid == 0) sh.chunk = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t c = sh.chunk;
        if (c >= n_chunks) break;
        const uint32_t b_first = c * RS_CH, n_here = n_blocks - b_first < RS_CH ? n_blocks - b_first : RS_CH;

        // ---- the lanes' maps, the wavefronts' prefix maps ------------------------------------------------------------------
        for (uint32_t q = 0; q < n_here; ++q) {
            const uint32_t b = b_first + q;
            const uint64_t sub_g = static_cast<uint64_t>(b) * RS_THREADS + tid;
            const bool live = sub_g < n_subs;
            const uint64_t sub_end = (sub_g + 1) * 256;
            uint32_t e_lo = ID_LO, e_hi = ID_HI, c_lo = 0, c_hi = 0;  // (a subsequence past the stream's end: nothing happens in it)
            const bool slow = live && ((sub_g == 0 && start_known) || sub_g >= first_tail_sub);
            if (live && !slow) {
                uint32_t w[9];
                const bool interior = static_cast<uint64_t>(b + 1) * 2048 + 1 <= n_words_full;  // wavefront-uniform
                if (interior) {
                    const uint32_t *src = words + sub_g * 8;  // (4-byte aligned; the compiler makes two 16-byte loads and one of 4 of it)
#pragma unroll
                    for (int j = 0; j < 9; ++j) w[j] = src[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 9; ++j) w[j] = rs_load_guarded(words, sub_g * 8 + j, n_bytes);
                }
                // 1. four rows per register: t[k] = bytes k, 8 + k, 16 + k, 24 + k of the subsequence
                uint32_t t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int o = k >> 2;  // bytes k < 4 lie in the even words, the others in the odd ones
                    const uint32_t sel = static_cast<uint32_t>(k & 3) | (static_cast<uint32_t>(4 + (k & 3)) << 8) | 0x0c0c0000u;
                    const uint32_t p01 = perm(w[2 + o], w[0 + o], sel), p23 = perm(w[6 + o], w[4 + o], sel);
                    t[k] = perm(p23, p01, 0x05040100u);
                }
                // 2. bit planes: bit j of B[c] = bit c (from the top) of row j, c = 0..7; c = 8..14: the same bits of row j + 1 (the
                //    seven bits behind the row's own) -- 2 instructions per (four rows, bit)
                uint32_t B[15];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {  // bit 7 - c of byte lane i of t[k] -> bit 8 i + k
                        const int up = k - (7 - c);
                        const uint32_t moved = up >= 0 ? t[k] << up : t[k] >> -up;
                        acc |= moved & (0x01010101u << k);
                    }
                    B[c] = acc;
                }
#pragma unroll
                for (int c = 0; c < 7; ++c) B[8 + c] = __builtin_amdgcn_alignbit(w[8] >> (7 - c), B[c], 1);  // rows 1..31 of plane c, then byte 32's bit
                // 3. S[r] bit j: the 7 bits at (row j, column r) -- planes r .. r + 6 -- are a 7-bit codeword: their value is < t.
                //    A comparator over the planes from the top bit down, 32 rows per instruction; t's bits are the same for everybody.
                uint32_t S[8], eq[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) S[r] = 0, eq[r] = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    if ((code_t >> (6 - i)) & 1u) {  // t has a 1 here: rows with a 0 (and equal so far) are below t; the others stay equal
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            S[r] |= rs_and_not(eq[r], B[r + i]);
                            eq[r] &= B[r + i];
                        }
                    } else {  // t has a 0 here: rows with a 1 are above it
#pragma unroll
                        for (int r = 0; r < 8; ++r) eq[r] = rs_and_not(eq[r], B[r + i]);
                    }
                }
                if (code_t >= 128u) {  // (128 codewords of 7 bits and no other: t does not fit the comparator's 7 bits)
#pragma unroll
                    for (int r = 0; r < 8; ++r) S[r] = 0xffffffffu;
                }
                // 4. the path from every entry column: in column (c0 - it) & 7 at iteration `it`, down to the next 7-bit code in that
                //    column (then one column to the left, from the row below) or to the subsequence's end.  `rows` = the rows still
                //    ahead of the path in the column it is about to look at.  Straight-line code, nothing indexed, no divergence.
                bool lane_stuck = false;
#pragma unroll
                for (int c0 = 0; c0 < 8; ++c0) {
                    // `rows` = the rows still ahead of the path in the column it looks at next (0: the path has left the subsequence);
                    // `hits` = the 7-bit codes it has met: it leaves in column (c0 - hits) & 7.  Per iteration: and, compare, v_ffbl, shift,
                    // select, add-with-carry -- and the compare's lane mask IS "somebody is still walking": no instruction for the test.
                    uint32_t rows = 0xffffffffu, hits = 0, wraps = 0;
                    bool more = true;
                    for (uint32_t lap = 0; lap < 6 && more; ++lap) {  // (<= 32 + 5 steps: all but the last are 7-bit codes, each in a row of its own but for <= 5)
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int col = (c0 - it) & 7;
                            const uint32_t m = S[col] & rows;
                            const bool some = m != 0;
                            uint32_t j;
                            asm("v_ffbl_b32 %0, %1" : "=v"(j) : "v"(m));
                            // a 7-bit code at (j, col) ends at (j + 1, col - 1): the rows below j -- from column 0 it ends at (j, 7): row j too.
                            // (from row 31 nothing is left below: the next iteration finds nothing, and the hit before it has moved the column)
                            rows = some ? (col == 0 ? 0xffffffffu : 0xfffffffeu) << j : 0u;
                            hits += some ? 1u : 0u;
                            if (col == 0) wraps += some ? 1u : 0u;  // two codewords begin in row j
                            more = __builtin_amdgcn_ballot_w64(some) != 0;
                            if (!more) break;
                        }
                    }
                    lane_stuck = lane_stuck || more;
                    const uint32_t ex = (static_cast<uint32_t>(c0) - hits) & 7u, cnt = 32u + wraps;
                    if (c0 == 0) e_lo = ex, c_lo = cnt;
                    else if (c0 < 4) e_lo |= ex << (8 * c0), c_lo |= cnt << (8 * c0);
                    else if (c0 == 4) e_hi = ex, c_hi = cnt;
                    else e_hi |= ex << (8 * (c0 - 4)), c_hi |= cnt << (8 * (c0 - 4));
                }
                if (lane_stuck) atomicOr(fault, 2u);  // (cannot happen: a path has at most 37 steps)
            } else if (slow) {
                // the stream's first subsequence (its first codeword begins at bit first_bit, whatever comes in) and the one or two
                // the stream ends in: one codeword at a time
                const uint8_t *bytes = reinterpret_cast<const uint8_t *>(words);
                e_lo = e_hi = c_lo = c_hi = 0;
                for (uint32_t c0 = 0; c0 < 8; ++c0) {
                    uint32_t ex = 0, cnt = 0;
                    if (sub_g == 0 && start_known && c0 > 0) {  // (a constant map)
                        ex = e_lo & 0xffu;
                        cnt = c_lo & 0xffu;
                    } else {
                        const uint32_t r = rs_slow_walk(bytes, n_bytes, (sub_g == 0 && start_known) ? first_bit : sub_g * 256 + c0, sub_end, code_t);
                        ex = r & 0xffu;
                        cnt = r >> 8;
                    }
                    if (c0 < 4) e_lo |= ex << (8 * c0), c_lo |= cnt << (8 * c0);
                    else e_hi |= ex << (8 * (c0 - 4)), c_hi |= cnt << (8 * (c0 - 4));
                }
            }
            // 5. inclusive scan of the wavefront's maps: pre_i = f_i o ... o f_0 (first f_0).  The scan's moves are DPP (row_shr 1, 2, 4, 8
            //    inside the rows of 16 lanes, then row_bcast:15 / :31 across them -- as a prefix sum's); a lane nothing moves into keeps the
            //    identity map, and composing with the identity is no change: no lane test.  4 VALU instructions per step, 24 in all.
            uint32_t p_lo = e_lo, p_hi = e_hi;
#define RS_SCAN_STEP(ctrl_, row_mask_)                                                                                                             \
    {                                                                                                                                              \
        const uint32_t o_lo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(ID_LO), static_cast<int>(p_lo), ctrl_, row_mask_, 0xf, false)); \
        const uint32_t o_hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(ID_HI), static_cast<int>(p_hi), ctrl_, row_mask_, 0xf, false)); \
        const uint32_t n_lo = perm(p_hi, p_lo, o_lo), n_hi = perm(p_hi, p_lo, o_hi); /* first theirs, then ours */                                 \
        p_lo = n_lo, p_hi = n_hi;                                                                                                                  \
    }
            RS_SCAN_STEP(0x111, 0xf)  // row_shr:1
            RS_SCAN_STEP(0x112, 0xf)  // row_shr:2
            RS_SCAN_STEP(0x114, 0xf)  // row_shr:4
            RS_SCAN_STEP(0x118, 0xf)  // row_shr:8   -> every row of 16 scanned
            RS_SCAN_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
            RS_SCAN_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef RS_SCAN_STEP
            sh.lane_pre[q][tid] = static_cast<unsigned long long>(p_lo) | (static_cast<unsigned long long>(p_hi) << 32);
            sh.lane_cnt[q][tid] = static_cast<unsigned long long>(c_lo) | (static_cast<unsigned long long>(c_hi) << 32);
            if (lane == 63) sh.wave_map[q * 4 + wv] = static_cast<unsigned long long>(p_lo) | (static_cast<unsigned long long>(p_hi) << 32);
        }
        __syncthreads();

        // ---- the chunk's map, its entry column by look-back ---------------------------------------------------------------------
        if (wv == 0) {
            uint32_t f_lo = ID_LO, f_hi = ID_HI;
            for (uint32_t k = 0; k < n_here * 4; ++k) {  // (every lane the same: 16 steps)
                if (lane == 0) sh.wave_pre[k] = static_cast<unsigned long long>(f_lo) | (static_cast<unsigned long long>(f_hi) << 32);
                const unsigned long long m = sh.wave_map[k];
                const uint32_t m_lo = static_cast<uint32_t>(m), m_hi = static_cast<uint32_t>(m >> 32);
                const uint32_t n_lo = perm(m_hi, m_lo, f_lo), n_hi = perm(m_hi, m_lo, f_hi);
                f_lo = n_lo, f_hi = n_hi;
            }
            uint32_t entry = 0;  // (the stream's first chunk: its first lane's map is constant)
            const bool look_back = map_only ? c + 1 == n_chunks : c > 0;  // (a range's map: only the last chunk looks, and all the way)
            if (c > 0 || map_only) {
                const unsigned long long mine = (static_cast<unsigned long long>(f_lo) | (static_cast<unsigned long long>(f_hi) << 32)) | KIND_MAP;
                if (lane == 0) __hip_atomic_store(pub + c, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // g: from the entry column of chunk i + 1 to ours; 64 chunks per look, nearest first
            uint32_t g_lo = ID_LO, g_hi = ID_HI;
            if (look_back) {
                long long i = static_cast<long long>(c) - 1;
                for (uint32_t spins = 0; i >= 0;) {
                    const long long idx = i - static_cast<long long>(lane);
                    unsigned long long v = 0;
                    if (idx >= 0) v = __hip_atomic_load(pub + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t kind = static_cast<uint32_t>(v >> 3) & 3u;
                    const unsigned long long ready = __ballot(kind != 0), known = __ballot(kind == 2);
                    const uint32_t n_ready = ~ready ? static_cast<uint32_t>(__builtin_ctzll(~ready)) : 64u;
                    if (n_ready == 0) {
                        if (++spins > RS_SPINS) {  // (a dead device; the launch still ends)
                            if (lane == 0) atomicOr(fault, 1u);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                        continue;
                    }
                    const uint32_t first_known = known ? static_cast<uint32_t>(__builtin_ctzll(known)) : 64u;
                    const uint32_t use = n_ready < first_known + 1 ? n_ready : first_known + 1;
                    const uint32_t v_lo = static_cast<uint32_t>(v) & 0x07070707u, v_hi = static_cast<uint32_t>(v >> 32) & 0x07070707u;
                    for (uint32_t l = 0; l < use; ++l) {  // g = g o (map of chunk i - l)
                        const uint32_t m_lo = __builtin_amdgcn_readlane(v_lo, l), m_hi = __builtin_amdgcn_readlane(v_hi, l);
                        const uint32_t n_lo = perm(g_hi, g_lo, m_lo), n_hi = perm(g_hi, g_lo, m_hi);
                        g_lo = n_lo, g_hi = n_hi;
                    }
                    if (first_known < use) break;  // (g is constant now: the entry column)
                    i -= use;  // (a range's map: on until the range's first chunk is in)
                }
                entry = g_lo & 7u;
            }
            if (map_only) {
                if (look_back && lane == 0) {  // the whole range: first g (everything before this chunk), then this chunk's own
                    const uint32_t t_lo = perm(f_hi, f_lo, g_lo), t_hi = perm(f_hi, f_lo, g_hi);
                    *map_out = static_cast<unsigned long long>(t_lo) | (static_cast<unsigned long long>(t_hi) << 32);
                }
            } else {
                const uint32_t out_col = map_at(f_lo, f_hi, entry);
                if (lane == 0) {
                    __hip_atomic_store(pub + c, static_cast<unsigned long long>(out_col) * 0x0101010101010101ull | KIND_ENTRY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh.entry = entry;
                }
            }
        }
        if (map_only) continue;  // (nothing per lane to leave; the loop's first barrier keeps the LDS in step)
        __syncthreads();

        // ---- every lane's start, exit and count ----------------------------------------------------------------------------------
        const uint32_t entry = sh.entry;
        for (uint32_t q = 0; q < n_here; ++q) {
            const uint32_t b = b_first + q;
            const uint64_t sub_g = static_cast<uint64_t>(b) * RS_THREADS + tid;
            const bool live = sub_g < n_subs;
            const unsigned long long wp = sh.wave_pre[q * 4 + wv], lp = sh.lane_pre[q][tid], lc = sh.lane_cnt[q][tid];
            const uint32_t wave_in = map_at(static_cast<uint32_t>(wp), static_cast<uint32_t>(wp >> 32), entry);
            const uint32_t out_col = map_at(static_cast<uint32_t>(lp), static_cast<uint32_t>(lp >> 32), wave_in);  // behind this lane
            // the column this lane is entered in = the one behind the lane before it (DPP wave_shr:1; lane 0 keeps the wavefront's)
            const uint32_t in_col = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(wave_in), static_cast<int>(out_col), 0x138, 0xf, 0xf, false));
            const uint32_t cnt = live ? map_at(static_cast<uint32_t>(lc), static_cast<uint32_t>(lc >> 32), in_col) : 0u;
            if (live) sub_state[sub_g] = ((sub_g == 0 && start_known) ? first_bit : in_col) | (out_col << 8) | (cnt << 16);
            const uint32_t sum = rs_wave_sum_scan(cnt);  // (inclusive prefix by DPP: lane 63 holds the wavefront's total)
            if (lane == 63) sh.wave_count[q * 4 + wv] = sum;
            if (tid == RS_THREADS - 1) blk_exit[b] = out_col;
        }
        __syncthreads();
        if (tid < n_here) blk_count[b_first + tid] = sh.wave_count[tid * 4] + sh.wave_count[tid * 4 + 1] + sh.wave_count[tid * 4 + 2] + sh.wave_count[tid * 4 + 3];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
//   decode.zig:186 -> k_row_write: the symbols of a row code's stream, written by rows.
//
// The general write pass (k_dec_write_wave, et_kernels.hip) looks codewords up in chained tables, one dependent LDS round trip
// per step, and stores a byte per symbol into its wavefront's stage.  On these streams every subsequence yields 32..37 symbols, so
// the 64 lanes' regions of the stage begin 32 bytes apart: lanes l and l + 4 store to the same LDS bank at every step -- 8-way
// conflicts on every byte store, 82 % of that kernel's LDS cycles (r04 PMC), 4.5 ms per 4 GiB.  Here a lane steps down the 32 rows
// of its subsequence: the 8 bits at (row, column) come out of registers (v_bfe_u32 at 8 - column), "is it a 7-bit code" is a compare
// with 2t that only the column depends on (the symbol lookup -- a 256-byte table in LDS -- is off that chain), and the stage is
// PADDED by 4 bytes per 128: position p lies at p + 4 (p >> 7), which puts the lanes' stores on different banks and keeps every
// 16-byte chunk of the output in one piece.
namespace {

constexpr uint32_t RWR_STAGE_LOGICAL = 64 * 37 + 16;                                 // symbols a wavefront can yield, + the phase of its first
constexpr uint32_t RWR_STAGE = ((RWR_STAGE_LOGICAL + 4 * (RWR_STAGE_LOGICAL >> 7) + 4) + 127) & ~127u;  // with the pads, rounded

struct RowLut {
    uint8_t sym[256];  // the symbol whose codeword the 8 bits begin with
};

__device__ __forceinline__ uint32_t rwr_scan(uint32_t x) {  // inclusive prefix sum over the wavefront (DPP)
#define RWR_DPP(ctrl_, mask_) x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), ctrl_, mask_, 0xf, false))
    RWR_DPP(0x111, 0xf);
    RWR_DPP(0x112, 0xf);
    RWR_DPP(0x114, 0xf);
    RWR_DPP(0x118, 0xf);
    RWR_DPP(0x142, 0xa);
    RWR_DPP(0x143, 0xc);
#undef RWR_DPP
    return x;
}

typedef __attribute__((address_space(3))) uint8_t rwr_lds_u8;
typedef __attribute__((address_space(3))) uint32_t rwr_lds_u32;

}  // namespace

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_row_write(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t n_blocks, uint64_t n_subs, uint32_t first_bit,
                                                         uint32_t code_t, const RowLut lut, const uint32_t *__restrict__ sub_state,
                                                         const unsigned long long *__restrict__ blk_off, uint64_t n_symbols, uint8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(128))) uint8_t smem[256 + WAVES * RWR_STAGE];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t i = tid; i < 64; i += 64 * WAVES) reinterpret_cast<uint32_t *>(smem)[i] = reinterpret_cast<const uint32_t *>(lut.sym)[i];
    __syncthreads();
    const uint32_t lds_lut = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((rwr_lds_u8 *)smem));
    const uint32_t lds_stage = lds_lut + 256u + wv * RWR_STAGE;  // this wavefront's stage: LDS address of position 0
    const uint64_t n_bits = n_bytes * 8, n_words_full = n_bytes / 4;
    const uint32_t two_t = 2 * code_t;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(words);
    // symbol at stage position p lies at LDS address lds_stage + p + 4 (p >> 7)
#define RWR_ADDR(p_) (lds_stage + (p_) + (((p_) >> 7) << 2))
    const uint32_t stride = gridDim.x * WAVES, n_units = n_blocks * 4;
    for (uint32_t u = blockIdx.x * WAVES + wv; u < n_units; u += stride) {
        const uint64_t b = u >> 2;
        const uint32_t quarter = u & 3u;
        const uint64_t sub_g = b * 256 + quarter * 64 + lane;
        const bool live = sub_g < n_subs;
        // where the wavefront's output begins: the block's offset + the quarters before this one (the same lane of each)
        uint32_t before = 0, st = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint64_t sg = b * 256 + q * 64 + lane;
            const uint32_t v = (q <= quarter && sg < n_subs) ? sub_state[sg] : 0u;
            if (q == quarter) st = v;
            if (q < quarter) before += v >> 16;
        }
        const unsigned long long o0 = blk_off[b];
        before = __builtin_amdgcn_readlane(rwr_scan(before), 63);
        const uint32_t count = live ? st >> 16 : 0u;
        const uint32_t inc = rwr_scan(count);
        const uint32_t wave_total = __builtin_amdgcn_readlane(inc, 63), my_off = inc - count;
        const uint64_t ow = o0 + before;
        const bool nothing = o0 >= n_symbols || ow >= n_symbols || wave_total == 0;  // (pad bits decoded past the declared length)
        uint64_t o1 = ow + wave_total;
        if (o1 > n_symbols) o1 = n_symbols;
        const uint32_t n_out = nothing ? 0u : static_cast<uint32_t>(o1 - ow);
        const uint32_t phase = static_cast<uint32_t>(ow & 15);
        const uint32_t span = nothing ? 0u : phase + n_out;
        uint8_t *out_base = out + (ow - phase);
        if (nothing) continue;  // (wavefront-uniform)
        const uint32_t my_lo = phase + my_off;
        const uint64_t sub_end = (sub_g + 1) * 256;
        const bool slow = live && count && (sub_g == 0 || sub_end + 8 > n_bits);
        const bool fast = live && count && !slow;
        if (fast) {
            const bool interior = (b + 1) * 2048 + 1 <= n_words_full;  // wavefront-uniform
            uint32_t X[9];
            if (interior) {
                const uint32_t *src = words + sub_g * 8;
#pragma unroll
                for (int j = 0; j < 9; ++j) X[j] = __builtin_bswap32(src[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 9; ++j) X[j] = __builtin_bswap32(rs_load_guarded(words, sub_g * 8 + j, n_bytes));
            }
            uint32_t sh = 8u - (st & 7u);  // 8 - column: how far the codeword's 8 bits lie above the NEXT row's first bit
            uint32_t pos = my_lo;          // stage position of the current row's first symbol, less the row's number
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int k = j >> 2, i = j & 3;
                // the 16 bits of rows j, j + 1 are bits [16 - 8 i, 32 - 8 i) of X[k] (i < 3) or the low 16 of (X[k] : X[k + 1]) >> 24
                uint32_t w8;
                if (i < 3) w8 = __builtin_amdgcn_ubfe(X[k], sh + (16 - 8 * i), 8);
                else w8 = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(X[k], X[k + 1], 24), sh, 8);
                const uint32_t sym = *reinterpret_cast<const rwr_lds_u8 *>(static_cast<uintptr_t>(lds_lut + w8));
                const uint32_t p = pos + j;
                *reinterpret_cast<rwr_lds_u8 *>(static_cast<uintptr_t>(RWR_ADDR(p))) = static_cast<uint8_t>(sym);
                sh += w8 < two_t ? 1u : 0u;  // a 7-bit code: the next one begins a column further left
                if (__any(sh == 9)) {        // ... from column 0 that is column 7 of the SAME row: one more codeword in it
                    if (sh == 9) {
                        uint32_t h;
                        if (i < 3) h = X[k] >> (16 - 8 * i);
                        else h = __builtin_amdgcn_alignbit(X[k], X[k + 1], 24);
                        const uint32_t w7 = (h >> 1) & 0xffu;
                        const uint32_t sym2 = *reinterpret_cast<const rwr_lds_u8 *>(static_cast<uintptr_t>(lds_lut + w7));
                        ++pos;
                        const uint32_t p2 = pos + j;
                        *reinterpret_cast<rwr_lds_u8 *>(static_cast<uintptr_t>(RWR_ADDR(p2))) = static_cast<uint8_t>(sym2);
                        sh = w7 < two_t ? 2u : 1u;
                    }
                }
            }
        } else if (slow) {
            // the stream's first subsequence and the one or two it ends in: one codeword at a time, as many as were counted
            uint64_t at = sub_g == 0 ? first_bit : sub_g * 256 + (st & 31u);
            for (uint32_t k = 0; k < count; ++k) {
                const uint64_t byte = at >> 3;
                const uint32_t s8 = static_cast<uint32_t>(at & 7);
                const uint32_t b0 = bytes[byte], b1 = byte + 1 < n_bytes ? bytes[byte + 1] : 0u;
                const uint32_t w8 = (((b0 << 8) | b1) >> (8 - s8)) & 0xffu;
                const uint32_t p = my_lo + k;
                *reinterpret_cast<rwr_lds_u8 *>(static_cast<uintptr_t>(RWR_ADDR(p))) = smem[w8];
                at += w8 < two_t ? 7 : 8;
            }
        }
        // (the wavefront's own LDS stores, then its own loads: in order; the fences say so to the compiler, no instruction comes of them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the stage leaves as 16-byte chunks (a chunk never straddles a pad), the two chunks shared with the neighbours byte by byte
        const uint32_t lo_valid = phase;
        for (uint32_t g = lane * 16; g < span; g += 64 * 16) {
            if (g >= lo_valid && g + 16 <= span) {
                const uint32_t a = RWR_ADDR(g);
                typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
                u32x4_nt v;
                v.x = *reinterpret_cast<const rwr_lds_u32 *>(static_cast<uintptr_t>(a));
                v.y = *reinterpret_cast<const rwr_lds_u32 *>(static_cast<uintptr_t>(a + 4));
                v.z = *reinterpret_cast<const rwr_lds_u32 *>(static_cast<uintptr_t>(a + 8));
                v.w = *reinterpret_cast<const rwr_lds_u32 *>(static_cast<uintptr_t>(a + 12));
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4_nt *>(out_base + g));
            }
        }
        {
            const uint32_t head = lo_valid & ~15u, tail = span & ~15u;
            const uint32_t p = (lane < 16 ? head : tail) + (lane & 15u);
            const bool partial = lane < 16 ? (lo_valid & 15u) != 0 : (span & 15u) != 0;
            if (lane < 32 && partial && p >= lo_valid && p < span) out_base[p] = *reinterpret_cast<const rwr_lds_u8 *>(static_cast<uintptr_t>(RWR_ADDR(p)));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#undef RWR_ADDR
}

// ---------------------------------------------------------------------------------------------------------------------------------
//   decode.zig:143-203 on a fixed-length code -> k_fixed_sync: where the codewords begin is arithmetic.
namespace {
struct FixedSpan {
    uint32_t start, exit, count;
};
// The codewords that begin in bits [lo, hi) of a stream of n_bits whose k-th codeword begins at first_bit + k L: the offset of the
// first of them, how many are whole (begin + L <= n_bits), and where the first one behind them begins, relative to hi -- or, if the
// stream ends first, the stream's end (the rule of rs_slow_walk above and of walk_subsequence, et_kernels_fallback.hip).
__device__ __forceinline__ FixedSpan fixed_span(uint64_t lo, uint64_t hi, uint64_t n_bits, uint32_t first_bit, uint32_t L) {
    const uint64_t pos0 = lo <= first_bit ? first_bit : first_bit + (lo - first_bit + L - 1) / L * L;
    uint64_t count = 0, next = pos0;
    if (n_bits >= L && pos0 < hi) {
        const uint64_t lim = hi - 1 < n_bits - L ? hi - 1 : n_bits - L;  // the last bit a whole codeword of this span can begin at
        if (pos0 <= lim) {
            count = (lim - pos0) / L + 1;
            next = pos0 + count * L;
        }
    }
    FixedSpan r;
    r.start = static_cast<uint32_t>(pos0 - lo);
    r.exit = next >= hi ? static_cast<uint32_t>(next - hi) : (n_bits > hi ? static_cast<uint32_t>(n_bits - hi) : 0u);
    r.count = static_cast<uint32_t>(count);
    return r;
}
}  // namespace

__global__ __launch_bounds__(256) void k_fixed_sync(uint64_t n_bits, uint32_t first_bit, uint64_t n_subs, uint32_t n_blocks, uint32_t L, uint32_t *__restrict__ sub_state,
                                                    uint32_t *__restrict__ blk_exit, uint32_t *__restrict__ blk_count) {
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const uint64_t s = static_cast<uint64_t>(b) * 256 + threadIdx.x;
        const FixedSpan sp = fixed_span(s * 256, s * 256 + 256, n_bits, first_bit, L);
        if (s < n_subs) sub_state[s] = (s == 0 ? first_bit : sp.start) | (sp.exit << 8) | (sp.count << 16);
        if (threadIdx.x == 255) blk_exit[b] = sp.exit;
        if (threadIdx.x == 0) blk_count[b] = fixed_span(s * 256, s * 256 + 65536, n_bits, first_bit, L).count;
    }
}

void launch_fixed_sync(hipStream_t stream, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, uint32_t code_bits, uint32_t *sub_state, uint32_t *blk_exit,
                       uint32_t *blk_count) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + 255) / 256);
    if (!n_blocks) return;
    const uint32_t grid = n_blocks < 16384u ? n_blocks : 16384u;
    hipLaunchKernelGGL(k_fixed_sync, dim3(grid), dim3(256), 0, stream, n_bytes * 8, first_bit, n_subs, n_blocks, code_bits, sub_state, blk_exit, blk_count);
}

size_t row_sync_scratch_bytes(uint32_t n_blocks) {
    const size_t n_chunks = (static_cast<size_t>(n_blocks) + RS_CH - 1) / RS_CH;
    return n_chunks * sizeof(unsigned long long) + 64;  // the chunks' words, then: ticket (4 bytes), pad, the range's map (8 bytes at + 8)
}

void launch_row_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, RowCode rc, void *scratch, uint32_t *fault,
                     uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_count, uint32_t flags, const unsigned long long **d_map) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + RS_THREADS - 1) / RS_THREADS);
    const uint32_t n_chunks = (n_blocks + RS_CH - 1) / RS_CH;
    if (!n_chunks) return;
    (void)hipMemsetAsync(scratch, 0, row_sync_scratch_bytes(n_blocks), stream);  // "nothing published", ticket 0
    unsigned long long *pub = static_cast<unsigned long long *>(scratch);
    uint32_t *ticket = reinterpret_cast<uint32_t *>(pub + n_chunks);
    unsigned long long *map_out = pub + n_chunks + 1;
    if (d_map) *d_map = map_out;
    static thread_local int seen_dev = -1, cus = 256;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != seen_dev) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        seen_dev = dev;
    }
    uint32_t grid = static_cast<uint32_t>(cus) * 8u;  // (workgroups that find no room wait their turn and take later tickets: nobody waits for them)
    if (grid > n_chunks) grid = n_chunks;
    hipLaunchKernelGGL(k_row_sync, dim3(grid), dim3(RS_THREADS), 0, stream, words, n_bytes, first_bit, n_subs, n_blocks, n_chunks, rc.t, pub, ticket, fault, sub_state, blk_exit,
                       blk_count, flags, map_out);
}

void launch_row_write(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, RowCode rc, const et_codebook *cb,
                      const uint32_t *sub_state, const unsigned long long *blk_off, uint64_t n_symbols, uint8_t *out, hipEvent_t ev_start, hipEvent_t ev_stop) {
    constexpr int WAVES = 8;
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + RS_THREADS - 1) / RS_THREADS);
    if (!n_blocks) return;
    RowLut lut = {};
    for (int s = 0; s < 256; ++s) {  // every 8-bit pattern begins with exactly one codeword (the code is complete)
        const uint32_t len = cb->length[s];
        if (len == 7) lut.sym[(cb->data[s] & 0x7fu) << 1] = lut.sym[((cb->data[s] & 0x7fu) << 1) | 1u] = static_cast<uint8_t>(s);
        else if (len == 8) lut.sym[cb->data[s] & 0xffu] = static_cast<uint8_t>(s);
    }
    static thread_local int seen_dev = -1, cus = 256, per_cu = 4;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != seen_dev) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_row_write<WAVES>, 64 * WAVES, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        seen_dev = dev;
    }
    const uint32_t n_wg = (n_blocks * 4 + WAVES - 1) / WAVES;
    uint32_t grid = static_cast<uint32_t>(cus) * static_cast<uint32_t>(per_cu);
    if (grid > n_wg) grid = n_wg;
    if (ev_start || ev_stop)
        hipExtLaunchKernelGGL(k_row_write<WAVES>, dim3(grid), dim3(64 * WAVES), 0, stream, ev_start, ev_stop, 0, words, n_bytes, n_blocks, n_subs, first_bit, rc.t, lut, sub_state, blk_off,
                              n_symbols, out);
    else
        hipLaunchKernelGGL(k_row_write<WAVES>, dim3(grid), dim3(64 * WAVES), 0, stream, words, n_bytes, n_blocks, n_subs, first_bit, rc.t, lut, sub_state, blk_off, n_symbols, out);
}

// ---------------------------------------------------------------------------------------------------------------------------------
//   decode.zig:186 on a fixed-length code -> k_fixed_write: symbol i is the L bits at first_bit + i L, whoever decodes it.
//
// No walk, no state, no stage: a thread takes 16 symbols -- 16 L <= 128 bits, five words from wherever they begin, shifted so that
// the first of them begins at bit 0 of the first, then 16 fields at offsets the compiler knows -- looks them up in the code's 2^L
// byte table in LDS and stores 16 bytes.  The chained-table write is at its worst on these streams: every lane of a wavefront
// yields the same number of symbols, so with L = 4 the lanes' regions of the stage begin 64 bytes apart and the byte stores of
// half the wavefront fall on one bank, and with L = 2 a subsequence's 128 symbols do not fit the stage and it is walked twice.
template <int L>
__global__ __launch_bounds__(256) void k_fixed_write(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_out, const RowLut lut,
                                                     uint8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[256];
    if (threadIdx.x < 64) reinterpret_cast<uint32_t *>(smem)[threadIdx.x] = reinterpret_cast<const uint32_t *>(lut.sym)[threadIdx.x];
    __syncthreads();
    const uint32_t lds_lut = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((rwr_lds_u8 *)smem));
    const uint64_t n_groups = (n_out + 15) / 16;
    for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; g < n_groups; g += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t bit = first_bit + g * (16 * L);
        const uint64_t w0 = bit >> 5;
        const uint32_t sh = static_cast<uint32_t>(bit & 31);
        uint32_t W[5];
        if ((w0 + 5) * 4 <= n_bytes) {
#pragma unroll
            for (int j = 0; j < 5; ++j) W[j] = __builtin_bswap32(words[w0 + j]);
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) W[j] = __builtin_bswap32(rs_load_guarded(words, w0 + j, n_bytes));
        }
        uint32_t V[4];  // the 128 bits from `bit` on
#pragma unroll
        for (int j = 0; j < 4; ++j) V[j] = sh ? __builtin_amdgcn_alignbit(W[j], W[j + 1], 32u - sh) : W[j];
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                constexpr uint32_t mask = (1u << L) - 1u;
                const int at = (4 * q + i) * L, k = at >> 5, r = at & 31;  // the field's L bits begin at bit r (from the top) of V[k]
                uint32_t f;
                if (r + L <= 32) f = (V[k] >> (32 - r - L)) & mask;
                else f = __builtin_amdgcn_alignbit(V[k], V[k < 3 ? k + 1 : 3], 64 - r - L) & mask;
                const uint32_t sym = *reinterpret_cast<const rwr_lds_u8 *>(static_cast<uintptr_t>(lds_lut + f));
                packed |= sym << (8 * i);
            }
            o[q] = packed;
        }
        uint8_t *dst = out + g * 16;
        if (g * 16 + 16 <= n_out) {
            typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
            u32x4_nt v;
            v.x = o[0], v.y = o[1], v.z = o[2], v.w = o[3];
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4_nt *>(dst));
        } else {  // the declared count's last few
            const uint32_t left = static_cast<uint32_t>(n_out - g * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (static_cast<uint32_t>(4 * q + i) < left) dst[4 * q + i] = static_cast<uint8_t>(o[q] >> (8 * i));
        }
    }
}

void launch_fixed_write(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, const et_codebook *cb, uint64_t n_out, uint8_t *out,
                        hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (!n_out) return;
    const uint32_t L = cb->max_length;
    RowLut lut = {};
    for (int s = 0; s < 256; ++s)
        if (cb->length[s] == L) lut.sym[cb->data[s] & ((1u << L) - 1u)] = static_cast<uint8_t>(s);
    static thread_local int seen_dev = -1, cus = 256;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != seen_dev) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        seen_dev = dev;
    }
    const uint64_t n_wg = ((n_out + 15) / 16 + 255) / 256;
    const uint32_t grid = static_cast<uint32_t>(n_wg < static_cast<uint64_t>(cus) * 32 ? n_wg : static_cast<uint64_t>(cus) * 32);
#define ET_FIXED_CASE(l_)                                                                                                                                  \
    case l_:                                                                                                                                               \
        if (ev_start || ev_stop) hipExtLaunchKernelGGL(k_fixed_write<l_>, dim3(grid), dim3(256), 0, stream, ev_start, ev_stop, 0, words, n_bytes, first_bit, n_out, lut, out); \
        else hipLaunchKernelGGL(k_fixed_write<l_>, dim3(grid), dim3(256), 0, stream, words, n_bytes, first_bit, n_out, lut, out);                         \
        break;
    switch (L) {
        ET_FIXED_CASE(1)
        ET_FIXED_CASE(2)
        ET_FIXED_CASE(3)
        ET_FIXED_CASE(4)
        ET_FIXED_CASE(5)
        ET_FIXED_CASE(6)
        ET_FIXED_CASE(7)
        ET_FIXED_CASE(8)
        default: break;  // (a complete fixed-length code over bytes has at most 256 codewords: callers check)
    }
#undef ET_FIXED_CASE
}

}  // namespace et

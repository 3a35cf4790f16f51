// et_kernels_fallback.hip -- the round-1 decode kernels, gfx950 / wave64: what decodes the streams outside the tree walk's
// (et_treewalk.hip) and the row walk's (et_rowsync.hip) domains.  Moved out of et_kernels.hip in round 4 so that the file people
// read holds the kernels a decode normally runs; same namespace, same launch wrappers (et_kernels.h), same tests
// (tests/test_gpu_parity.py::test_fallback_*, test_uniform_alphabets_fast_and_exhaustive_sync, the cold-range tests assert which path ran).
//
//   decode.zig:143-203 -> k_dec_sync (LDS window: a stream's first / last blocks, ranges), k_dec_sync_reg2 / k_dec_sync_reg (register
//                         window, greedy step table, escapes for long codes), k_dec_check (worklist), k_dec_maps[_reg] / k_dec_compose /
//                         k_dec_chain / k_dec_resolve[_reg] (exit maps for every start offset: codes that do not self-synchronise)
//   decode.zig:186     -> k_dec_write_reg, k_dec_write
//   decode.zig:123-125 -> k_build_dec_tables (their lookup tables, filled on the device from the host's plan)
// Used when: a hand-made dictionary's completed tree has more than 255 internal nodes or is not prefix-free; a near-fixed-length code
// that is not a row code; a range of a stream split over GPUs whose code does not self-synchronise; blocks that give up in the first sweep.
#include "et_kernels_common.h"
#include "et_treewalk.h"

namespace et {

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

__host__ __device__ __forceinline__ uint32_t step_table_words(const DecodeTables &tb) {
    return ((1u << tb.step_bits) + (tb.n_step_sub << tb.step_sub_bits) + 3u) & ~3u;
}

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
// launch wrappers
// --------------------------------------------------------------------------------
static inline size_t decode_smem_bytes(const DecodeTables &tb, bool with_stage, bool with_exits = true, bool with_stream = true) {
    const uint32_t sub_w = (((tb.n_sub << tb.sub_bits) + 7u) & ~7u) / 2;
    return ((with_stream ? DEC_SDATA_WORDS : 0) + (1u << tb.lut_bits) + sub_w + 64 + (with_exits ? BLOCK : 0) + 8) * sizeof(uint32_t) + (with_stage ? DEC_STAGE_BYTES + 16 : 0);
}

// Interior blocks take the register-window kernels; the LDS-window kernels keep the stream's first
// and last blocks (and streams of nothing else).
static bool use_reg_kernels(uint32_t n_blocks) { return n_blocks > 3; }

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

void launch_dec_write_fallback(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint64_t n_subs, const DecodeTables &tb, const uint32_t *sub_state,
                               const unsigned long long *blk_off, uint64_t n_symbols, uint8_t *out, uint32_t *ticket, const SideLane *side, bool ticket_is_zero,
                               const uint32_t *void_flags, KernelEvents ev) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
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

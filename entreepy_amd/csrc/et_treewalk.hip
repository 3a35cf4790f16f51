// et_treewalk.hip -- D1 by fixed-rate tree walk (format and rationale: et_treewalk.h), gfx950 / wave64.
//
//   decode.zig:143-203 -> k_tw_sync: where do codewords begin, and how many begin in each 256 bits?
//
// A wavefront owns one 8 KiB block: 128 lanes of 512 bits held in registers, thread t walks lanes t and
// 64 + t (two dependent chains in flight per thread: a step is an LDS round trip, ~200 cycles under the
// ~3.5-way bank conflicts of random lookups, and nothing else hides it).  Every lane runs in over the 128
// bits before its own, then walks its 64 bytes; lanes of a block agree on the tree node at their seams by a
// fixed point over shuffles.  After the table is staged nothing is shared between wavefronts: no barrier, no
// LDS traffic besides the lookups, no ticket.  The chain per step is table read -> shift -> and-or -> table
// read; the byte positions are compile-time constants.
#include "et_treewalk.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace et {

// ALL LDS of k_tw_sync is the dynamic block (cdna_hip_programming.md Guideline 17): the table, at LDS address 0.
extern __shared__ __attribute__((aligned(16))) uint8_t tw_smem[];
typedef __attribute__((address_space(3))) uint8_t tw_lds_u8;
typedef __attribute__((address_space(3))) uint16_t tw_lds_u16;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t tw_dpp_add(uint32_t x) {
    return x + static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ uint32_t tw_wave_inclusive_scan(uint32_t x) {
    x = tw_dpp_add<0x111, 0xf>(x);  // row_shr:1
    x = tw_dpp_add<0x112, 0xf>(x);  // row_shr:2
    x = tw_dpp_add<0x114, 0xf>(x);  // row_shr:4
    x = tw_dpp_add<0x118, 0xf>(x);  // row_shr:8
    x = tw_dpp_add<0x142, 0xa>(x);  // row_bcast:15
    x = tw_dpp_add<0x143, 0xc>(x);  // row_bcast:31
    return x;
}

// Word `idx` (may be negative: before `words`) of the stream AS IT LIES IN MEMORY (stream byte k of the word is its byte k: the
// walk picks bytes by SDWA selects, so nothing is swapped), zero outside the stream.  front_ok: the four words in front of `words`
// are stream bytes too (a range of a stream that began earlier).
__device__ __attribute__((noinline)) uint32_t tw_load_guarded(const uint32_t *__restrict__ words, long long idx, uint64_t n_bytes, bool front_ok) {
    if (idx < 0) return front_ok && idx >= -4 ? words[idx] : 0u;
    const uint64_t b0 = static_cast<uint64_t>(idx) * 4;
    if (b0 + 4 <= n_bytes) return words[idx];
    uint32_t v = 0;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(words);
    for (int k = 0; k < 4; ++k)
        if (b0 + k < n_bytes) v |= static_cast<uint32_t>(bytes[b0 + k]) << (8 * k);
    return v;
}

// ---- the table, filled on the device from the tree -------------------------------------------------
// One launch fills both tables of a code: the synchronisation walk's (table, may be null) and the write walk's chained
// lookup tables (chain, may be null; tw_chain_entry is the definition, shared with the host fill).
__global__ __launch_bounds__(1024) void k_tw_build(const TwUpload *__restrict__ up, uint32_t up_bytes, uint32_t n_int, uint16_t *__restrict__ table, uint32_t n_chain,
                                                   uint64_t *__restrict__ chain, uint32_t *__restrict__ zero16, uint32_t *__restrict__ zero_words, uint32_t n_zero) {
    // `up` may be pinned HOST memory (no upload in front of this kernel): every workgroup takes its own copy, once
    __shared__ TwUpload local;
    if (zero16 && blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0;  // the decode's flag words
    for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < n_zero; i += gridDim.x * 1024) zero_words[i] = 0;  // the sweep's "published" words
    for (uint32_t i = threadIdx.x; i * 4 < up_bytes; i += 1024) reinterpret_cast<uint32_t *>(&local)[i] = reinterpret_cast<const uint32_t *>(up)[i];
    __syncthreads();
    const TwTree &tree = local.tree;
    const uint32_t entries = table ? tw_table_entries(n_int) : 0;
    const int16_t *child = tree.child;
    for (uint32_t idx = blockIdx.x * 1024 + threadIdx.x; idx < entries + n_chain; idx += gridDim.x * 1024) {
        if (idx >= entries) {
            const uint32_t i = idx - entries;
            uint32_t t = 0;
            while (t + 1 < local.plan.n_tables && local.plan.tab[t + 1].first <= i) ++t;
            chain[i] = tw_chain_entry(&tree, &local.plan, t, i - local.plan.tab[t].first);
            continue;
        }
        const uint32_t r = idx >> 8, f = idx & 255u;
        uint32_t node = r < n_int ? r : 0, n = 0, first = 0;
        const uint32_t skip = r < n_int ? 0 : r - n_int + 1;
        for (uint32_t i = skip; i < 8; ++i) {
            const int c = child[2 * node + ((f >> (7 - i)) & 1u)];
            if (c >= 0) {
                node = static_cast<uint32_t>(c);
            } else {
                if (n++ == 0) first = i;
                node = 0;
            }
        }
        table[idx] = static_cast<uint16_t>(node | (n << TW_N_SHIFT) | (first << TW_OFF_SHIFT));
    }
}

// ---- the walk -------------------------------------------------------------------------------------
#ifndef ET_TW_RUN_WORDS
#define ET_TW_RUN_WORDS 4
#endif
constexpr int TW_RUN = ET_TW_RUN_WORDS;       // words a lane runs in over before its own
constexpr int TW_WORDS = TW_RUN + 17;         // W[j] = stream word 16 * lane - TW_RUN + j: the run-in words, 16 own, 1 beyond (a block's tail at the stream's end)
constexpr int TW_OWN = 32 * TW_RUN;           // bit of W at which the lane's own 512 begin
constexpr int TW_LANES = 2;   // 512-bit lanes walked by one thread
constexpr uint32_t TW_PUB_POLLS = 256;  // looks at the word the block before publishes (a short sleep in between: ~100 us in all) before a block gives up on it

struct TwTrack {  // where the subsequence's first codeword begins: the bit after the first completion
    bool found[TW_LANES];
    uint32_t start[TW_LANES];
};

// N_STEPS bytes from bit BIT0 of both lanes' words: R = row byte offset (row << 9), C += codewords completed.
// EDGE: only the steps below `limit` count (the stream ends inside the lane's bytes); SKIP: steps below
// `skip` do not count either (a walk that begins at a bit offset); TRACK: the first four steps also look for
// the first completion.
template <int BIT0, int N_STEPS, bool EDGE, bool SKIP, bool TRACK, bool COUNT = true>
__device__ __forceinline__ void tw_walk(const uint32_t (&W)[TW_LANES][TW_WORDS], uint32_t (&R)[TW_LANES], uint32_t (&C)[TW_LANES],
                                        const uint32_t (&skip)[TW_LANES], const uint32_t (&limit)[TW_LANES], int step0, TwTrack &t) {
    const uint32_t tab = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((tw_lds_u8 *)tw_smem));
#pragma unroll
    for (int j = 0; j < N_STEPS; ++j) {
        // both lanes' lookups first, then what is done with them: the two reads are in flight together (with one loop over the
        // lanes the compiler waited for lane 0's entry before it issued lane 1's read -- one LDS round trip at a time per wavefront)
        uint32_t ee[TW_LANES];
#pragma unroll
        for (int u = 0; u < TW_LANES; ++u) {
            const int bit = BIT0 + 8 * j;
            const uint32_t f2 = ((W[u][bit >> 5] >> (bit & 31)) & 0xffu) << 1;  // stream byte (bit / 8) % 4 of the word = its byte of that index, as loaded
            ee[u] = *reinterpret_cast<const tw_lds_u16 *>(static_cast<uintptr_t>(tab + R[u] + f2));  // (R = row << 9, f2 < 512: a sum, so that base, row and byte make one v_add3)
        }
#pragma unroll
        for (int u = 0; u < TW_LANES; ++u) {
            const uint32_t e = ee[u];
            bool on = true;
            if (EDGE) on = on && static_cast<uint32_t>(step0 + j) < limit[u];
            if (SKIP && j < 4) on = on && static_cast<uint32_t>(j) >= skip[u];
            uint32_t next;  // (e & TW_ROW_MASK) << 9 -- byte 0 of the entry, one SDWA shift (the compiler makes a shift and a mask of it)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(next) : "s"(9u), "v"(e));
            uint32_t n = (e >> TW_N_SHIFT) & 15u;
            const bool plain = !(EDGE || (SKIP && j < 4)) && !(TRACK && j < 4);  // nobody looks at n: it only joins the count
            if (EDGE || (SKIP && j < 4)) {
                R[u] = on ? next : R[u];
                n = on ? n : 0u;
            } else {
                R[u] = next;
            }
            if (TRACK && j < 4) {
                const bool hit = !t.found[u] && n > 0;
                t.start[u] = hit ? 8u * j + (e >> TW_OFF_SHIFT) + 1u : t.start[u];
                t.found[u] = t.found[u] || n > 0;
            }
            // (added here and now: left to itself the compiler keeps every entry of the walk alive and sums them at the end, out of scratch memory)
            if (!COUNT) {  // (a run-in, a look for the first completion: nobody wants the count)
            } else if (plain) asm volatile("v_dot8_u32_u4 %0, %1, %2, %0" : "+v"(C[u]) : "v"(e), "s"(1u << TW_N_SHIFT));  // C += nibble 2 of the entry: the count, one instruction
            else asm volatile("v_add_u32 %0, %0, %1" : "+v"(C[u]) : "v"(n));
        }
        // (nothing moves across a step: the compiler otherwise computes every byte offset of the walk up front and spills)
        __builtin_amdgcn_sched_barrier(0);
        if (TRACK && j + 1 < N_STEPS) {  // looking for first completions only: done once every lane of the wavefront has seen one
            bool open = false;
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) open = open || !t.found[u];
            if (!__any(open)) break;
        }
    }
}

struct TwLane {  // (two registers per lane, not four: the sweep lives on its wavefronts per SIMD)
    uint32_t s;  // rows (tree nodes, < 256) after 256 bits | after 512 << 16
    uint32_t c;  // codewords completed in the first 256 bits (<= 256) | in the second << 16
    __device__ __forceinline__ uint32_t s_mid() const { return s & 0xffffu; }
    __device__ __forceinline__ uint32_t s_out() const { return s >> 16; }
    __device__ __forceinline__ uint32_t c1() const { return c & 0xffffu; }
    __device__ __forceinline__ uint32_t c2() const { return c >> 16; }
    __device__ __forceinline__ void first_half(uint32_t count, uint32_t row) {
        c = (c & 0xffff0000u) | count;
        s = (s & 0xffff0000u) | row;
    }
    __device__ __forceinline__ void second_half(uint32_t count, uint32_t row) {
        c = (c & 0xffffu) | (count << 16);
        s = (s & 0xffffu) | (row << 16);
    }
};

// Both lanes of a thread from row offsets R0.  `take[u]`: lane u's results are wanted; REWALK: a lane that stands where its
// old walk stood keeps the rest -- after TW_CK steps (ck[u] = the old walk's row there | the codewords it had completed by
// then: a row offset's low 9 bits are free), which is where a walk from the wrong node has all but always met the right one (a
// re-walk is 8 steps instead of 32: they were a quarter of the sweep), and again after 256 bits.
#ifndef ET_TW_CK_STEPS
#define ET_TW_CK_STEPS 8
#endif
constexpr int TW_CK = ET_TW_CK_STEPS;  // 4 <= TW_CK < 32 (the steps that may skip lie in front of it; a count of <= 8 * TW_CK fits 9 bits)
static_assert(TW_CK >= 4 && TW_CK < 32 && 8 * TW_CK < 512, "checkpoint step");
template <bool EDGE, bool REWALK>
__device__ __forceinline__ void tw_lanes(uint32_t (&W)[TW_LANES][TW_WORDS], const uint32_t (&R0)[TW_LANES], const uint32_t (&skip)[TW_LANES],
                                         const uint32_t (&limit)[TW_LANES], const bool (&take)[TW_LANES], TwLane (&r)[TW_LANES], uint32_t (&ck)[TW_LANES]) {
    // (the words are "new" to every walk: the compiler otherwise keeps the first walk's byte offsets for the re-walk, in scratch memory)
#pragma unroll
    for (int u = 0; u < TW_LANES; ++u)
#pragma unroll
        for (int j = TW_RUN; j < TW_WORDS; ++j) asm volatile("" : "+v"(W[u][j]));
    uint32_t R[TW_LANES], C[TW_LANES];
    bool redo[TW_LANES];
    bool any_redo = false;
    TwTrack t = {};
#pragma unroll
    for (int u = 0; u < TW_LANES; ++u) {
        R[u] = R0[u];
        C[u] = 0;
    }
    tw_walk<TW_OWN, TW_CK, EDGE, true, false>(W, R, C, skip, limit, 0, t);
    {
        bool off = false;
#pragma unroll
        for (int u = 0; u < TW_LANES; ++u) {
            const uint32_t here = R[u] | C[u];
            if (REWALK) off = off || (take[u] && ((here ^ ck[u]) >> 9) != 0);
            if (REWALK && take[u]) r[u].c += C[u] - (ck[u] & 511u);  // (right if the wavefront stops here; recounted in full if it does not)
            if (take[u]) ck[u] = here;
        }
        if (REWALK && !__any(off)) return;
    }
    tw_walk<TW_OWN + 8 * TW_CK, 32 - TW_CK, EDGE, false, false>(W, R, C, skip, limit, TW_CK, t);
#pragma unroll
    for (int u = 0; u < TW_LANES; ++u) {
        if (EDGE && limit[u] < 32) R[u] = 0;  // the stream ended: nothing is pending
        const uint32_t mid = R[u] >> 9;
        redo[u] = take[u] && !(REWALK && mid == r[u].s_mid());
        if (take[u]) r[u].first_half(C[u], mid);
        C[u] = 0;
        any_redo = any_redo || redo[u];
    }
    if (__any(any_redo)) {
        tw_walk<TW_OWN + 256, 32, EDGE, false, false>(W, R, C, skip, limit, 32, t);
#pragma unroll
        for (int u = 0; u < TW_LANES; ++u) {
            if (EDGE && limit[u] < 64) R[u] = 0;
            if (redo[u]) r[u].second_half(C[u], R[u] >> 9);
        }
    }
}

// What the block before block b published -- bit 31 | the node it ends in -- or 0 when that is `guess`, the node b's first lane
// started from (0 also when nothing shows up within TW_PUB_POLLS looks).  Wavefront-uniform.
__device__ __forceinline__ uint32_t tw_seen_before(const uint32_t *__restrict__ blk_pub, uint32_t b, uint32_t guess, uint32_t lane_id, uint32_t seen = 0) {
    // (`seen`: what an earlier look found -- asked for at the top of the trip, a walk ago, so that nobody waits for it here)
    if (lane_id == 0) {
        for (uint32_t poll = 0; poll < TW_PUB_POLLS && !(seen & 0x80000000u); ++poll) {
            seen = __hip_atomic_load(blk_pub + (b - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(seen & 0x80000000u)) __builtin_amdgcn_s_sleep(8);
        }
        if (!(seen & 0x80000000u) || (seen & 0x7fffffffu) == guess) seen = 0;
    }
    return __builtin_amdgcn_readfirstlane(seen);
}

#ifndef ET_TW_WAVES_PER_EU
#define ET_TW_WAVES_PER_EU 6
#endif
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ET_TW_WAVES_PER_EU, 8))) void k_tw_sync(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs,
                                                    uint32_t n_blocks, const uint16_t *__restrict__ table, uint32_t table_entries, uint32_t n_int,
                                                    uint32_t *__restrict__ sub_state, uint32_t *__restrict__ blk_exit, uint32_t *__restrict__ blk_start,
                                                    uint32_t *__restrict__ blk_count, uint32_t *__restrict__ changed, uint32_t max_trips,
                                                    const uint32_t *__restrict__ worklist, const uint32_t *__restrict__ n_work, uint32_t *__restrict__ blk_pub,
                                                    uint32_t mode, uint32_t *__restrict__ exit_bits) {
    uint16_t *tab = reinterpret_cast<uint16_t *>(tw_smem);
    const uint32_t T = blockDim.x, tid = threadIdx.x, lane_id = tid & 63;
    const uint32_t n_todo = worklist ? *n_work : n_blocks;
    if (blockIdx.x * (T >> 6) >= n_todo) return;  // (a repair sweep usually has a handful of blocks)
    for (uint32_t i = tid * 8; i < table_entries; i += T * 8) *reinterpret_cast<uint4 *>(tab + i) = *reinterpret_cast<const uint4 *>(table + i);
    __syncthreads();
    const uint64_t n_lanes = (n_subs + 1) / 2;
    const uint64_t n_words_full = n_bytes / 4;
    const uint32_t waves_per_group = T >> 6;
    // Across blocks (first sweep, blk_pub given): a block that agrees with itself PUBLISHES the node it ends in and looks
    // at what the block before it published; if that is not the node its first lane started from (~0.4 % of the blocks
    // of a text) the wavefront's next trip is the same block again, from that node (`forced`).  Nobody waits for
    // anybody who waits: what is published is a first attempt's result -- one word, stored and polled with relaxed
    // device-scope atomics (nothing else travels with it; release/acquire here would write back and invalidate the L2
    // of every XCD per block: the sweep took 3.6 ms instead of 0.27).  Blocks stay statically assigned: handing them out
    // by ticket, so that a wavefront with a block to do twice takes one fewer, costs 77 K atomics on one address: 0.98 ms.
    // (A block whose end moves in its second attempt -- it did not re-synchronise within 8 KiB -- leaves a stale word
    // behind; the verification scan sees the mismatch and the host sweeps again.)
    // A block is held against the block before it ONE TRIP LATE (`pend_b`): looked at right behind its own walk, the word of
    // the block before is often not there yet (that block had a seam to settle, a slower lookup), the wavefront sleeps, starts
    // its next block late, publishes late, and the lateness travels from neighbour to neighbour -- the sweep ran at the pace
    // of the slowest of any two neighbours, trip after trip (0.227 ms; without the look 0.18).  A trip later the word has been
    // there for ~20 us.  What a late look costs: a block that turns out to begin elsewhere has already stored its results;
    // its second attempt stores them again (same wavefront, same addresses: in order).
    uint32_t it = blockIdx.x * waves_per_group + (tid >> 6), again_b = 0, again_row = 0xffffffffu, pend_b = 0xffffffffu, pend_start = 0;
    for (;;) {
        uint32_t b, forced = 0xffffffffu;
        if (again_row != 0xffffffffu) {
            b = again_b;
            forced = again_row;
            again_row = 0xffffffffu;
        } else if (it < n_todo) {
            b = worklist ? worklist[it] : it;
            it += gridDim.x * waves_per_group;
        } else if (pend_b != 0xffffffffu) {  // the wavefront's last block has not been looked at yet
            const uint32_t seen = tw_seen_before(blk_pub, pend_b, pend_start, lane_id);
            again_b = pend_b;
            pend_b = 0xffffffffu;
            if (seen & 0x80000000u) again_row = seen & 0x7fffffffu;
            continue;
        } else {
            break;
        }
        {
        uint32_t early = 0;  // the word the pending block's predecessor published: asked for now, looked at behind this block's walk
        if (blk_pub && !worklist && forced == 0xffffffffu && pend_b != 0xffffffffu && lane_id == 0)
            early = __hip_atomic_load(blk_pub + (pend_b - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // wavefront-uniform: every word of the block, its run-in and the word after it is a whole word of the stream
        const long long bw0 = static_cast<long long>(b) * 2048 - TW_RUN;
        const bool edge = bw0 < 0 || static_cast<uint64_t>(bw0 + 2048 + TW_RUN + 1) > n_words_full;
        uint32_t W[TW_LANES][TW_WORDS], limit[TW_LANES];
        uint64_t q[TW_LANES];
        bool live[TW_LANES];
#pragma unroll
        for (int u = 0; u < TW_LANES; ++u) {
            q[u] = static_cast<uint64_t>(b) * 128 + u * 64 + lane_id;  // 512-bit lane = subsequences 2q, 2q + 1
            live[u] = q[u] < n_lanes;
            limit[u] = 0xffffffffu;
            const long long w0 = static_cast<long long>(q[u]) * 16 - TW_RUN;
            if (!edge) {
#pragma unroll
                for (int j = 0; j < TW_WORDS; ++j) W[u][j] = words[w0 + j];  // (as they lie in memory: no byte swap, see tw_walk)
            } else {
#pragma unroll
                for (int j = 0; j < TW_WORDS; ++j) W[u][j] = tw_load_guarded(words, w0 + j, n_bytes, (mode & TW_FRONT_OK) != 0);
                // whole bytes of the stream inside the lane's own 64 -> steps that count
                const uint64_t lane_byte0 = q[u] * 64;
                limit[u] = static_cast<uint32_t>(n_bytes > lane_byte0 ? (n_bytes - lane_byte0 < 64 ? n_bytes - lane_byte0 : 64) : 0);
            }
        }
        // run-in: from the root, 128 bits before the lane's own
        uint32_t R0[TW_LANES] = {}, skip[TW_LANES] = {}, C0[TW_LANES] = {};
        const uint32_t no_limit[TW_LANES] = {0xffffffffu, 0xffffffffu};
        {
            TwTrack none = {};
            tw_walk<0, 4 * TW_RUN, false, false, false, false>(W, R0, C0, skip, no_limit, 0, none);
        }
        // The block's first lane may KNOW where it begins: the stream's first lane (the root at bit first_bit),
        // or, in a repair sweep, the node the block before ends in.
        const bool first_lane = lane_id == 0;
        bool known_bit = false;
        if (first_lane && b == 0 && !(mode & TW_START_UNKNOWN)) {
            known_bit = true;
            skip[0] = first_bit / 8;
            const uint32_t rem = first_bit % 8;
            R0[0] = rem ? (n_int + rem - 1) << 9 : 0u;
        } else if (first_lane && worklist) {
            R0[0] = blk_exit[b - 1] << 9;
        } else if (first_lane && forced != 0xffffffffu) {
            R0[0] = forced << 9;
        }
        uint32_t start[TW_LANES];  // the row each lane's walk begins in (0 also for a known bit offset: it is a codeword boundary)
        TwLane r[TW_LANES] = {};
        bool take[TW_LANES];
#pragma unroll
        for (int u = 0; u < TW_LANES; ++u) {
            start[u] = (u == 0 && known_bit) ? 0u : R0[u] >> 9;
            take[u] = true;
        }
        uint32_t ck[TW_LANES] = {};
        if (edge) tw_lanes<true, false>(W, R0, skip, limit, take, r, ck);
        else tw_lanes<false, false>(W, R0, skip, limit, take, r, ck);
        skip[0] = 0;
        bool gave_up = false;
        for (uint32_t trip = 1;; ++trip) {
            // lane li's start must be lane li - 1's exit (li = u * 64 + lane); the block's first lane keeps its own
            bool need[TW_LANES], any_need = false;
            uint32_t cand[TW_LANES];
            // the lane before's exit by DPP (wave_shr:1; lane 0 keeps what is handed in as `old`: its own start / the exit of lane 63
            // of the other half, a scalar) -- the shuffles were ds_bpermute round trips on the seam loop's dependent path
            const uint32_t out0 = r[0].s_out(), out1 = r[1].s_out();
            const uint32_t last0 = __builtin_amdgcn_readlane(out0, 63);
            cand[0] = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(start[0]), static_cast<int>(out0), 0x138, 0xf, 0xf, false));
            cand[1] = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(last0), static_cast<int>(out1), 0x138, 0xf, 0xf, false));
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) {
                need[u] = live[u] && cand[u] != start[u];
                any_need = any_need || need[u];
            }
            if (!__any(any_need)) break;
            if (trip == max_trips) {  // a code that does not self-synchronise: the block is marked for a redo
                gave_up = true;
                break;
            }
            uint32_t Rn[TW_LANES];
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) Rn[u] = need[u] ? cand[u] << 9 : 0u;
            if (edge) {
                // (the limits are "new" to every trip: left loop-invariant, the edge walk's 128 step-limit compares are hoisted in
                // front of the trip loop, where EVERY block pays for them -- 337 of a block's ~1360 VALU instructions)
                ET_PIN(limit[0]);
                ET_PIN(limit[1]);
                tw_lanes<true, true>(W, Rn, skip, limit, need, r, ck);
            } else {
                tw_lanes<false, true>(W, Rn, skip, limit, need, r, ck);
            }
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) start[u] = need[u] ? cand[u] : start[u];
        }
        if (blk_pub && !worklist && forced == 0xffffffffu) {
            if (lane_id == 63) __hip_atomic_store(blk_pub + b, 0x80000000u | r[1].s_out(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pend_b != 0xffffffffu) {  // the block of the trip before: does it begin where the block before IT ends?
                const uint32_t seen = tw_seen_before(blk_pub, pend_b, pend_start, lane_id, early);
                if (seen & 0x80000000u) {  // (never seen: the block keeps its guess, the verification decides)
                    again_b = pend_b;
                    again_row = seen & 0x7fffffffu;
                }
            }
            // (a block that gave up is done again whatever the block before it says: no row is 0x7fffffff)
            pend_b = b ? __builtin_amdgcn_readfirstlane(b) : 0xffffffffu;
            pend_start = __builtin_amdgcn_readfirstlane(gave_up ? 0x7fffffffu : start[0]);
        }
        // Where each subsequence's first codeword begins: one past the first completion seen from the node at
        // its first bit -- four more steps per half, once the nodes are settled (tracked inside the walks it
        // cost 25 registers, i.e. a third of the wavefronts).  found = false: the code that straddles the
        // boundary is cut by the stream's end.
        uint32_t st1[TW_LANES], st2[TW_LANES];
        bool f1[TW_LANES], f2[TW_LANES];
        {
            uint32_t Ra[TW_LANES], Ca[TW_LANES] = {};
            TwTrack ta;
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) {
                Ra[u] = start[u] << 9;
                ta.found[u] = start[u] == 0;  // (also the lane with a known bit offset: it begins there)
                ta.start[u] = (u == 0 && known_bit) ? first_bit : 0u;
            }
            if (edge) tw_walk<TW_OWN, 4, true, false, true, false>(W, Ra, Ca, skip, limit, 0, ta);
            else tw_walk<TW_OWN, 4, false, false, true, false>(W, Ra, Ca, skip, limit, 0, ta);
            TwTrack tb;
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) {
                st1[u] = ta.found[u] ? ta.start[u] : 0u;
                f1[u] = ta.found[u];
                Ra[u] = r[u].s_mid() << 9;
                tb.found[u] = r[u].s_mid() == 0;
                tb.start[u] = 0;
            }
            if (edge) tw_walk<TW_OWN + 256, 4, true, false, true, false>(W, Ra, Ca, skip, limit, 32, tb);
            else tw_walk<TW_OWN + 256, 4, false, false, true, false>(W, Ra, Ca, skip, limit, 32, tb);
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) {
                st2[u] = tb.found[u] ? tb.start[u] : 0u;
                f2[u] = tb.found[u];
            }
        }
        // Codewords that BEGIN in a subsequence = those that end in it, less the one that came in over its
        // first bit, plus the one that leaves over its last.  The one leaving the lane's last bit: at the
        // stream's end it may be cut (then it is nobody's): an edge block looks at the bytes after the lane.
        bool out_ok[TW_LANES] = {true, true};
        if (edge) {
            uint32_t Rt[TW_LANES], Ct[TW_LANES] = {}, lim_t[TW_LANES];
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) {
                Rt[u] = r[u].s_out() << 9;
                const uint64_t after = (q[u] + 1) * 64;
                lim_t[u] = static_cast<uint32_t>(n_bytes > after ? (n_bytes - after < 4 ? n_bytes - after : 4) : 0);
            }
            TwTrack tt = {};
            tw_walk<TW_OWN + 512, 4, true, false, false>(W, Rt, Ct, skip, lim_t, 0, tt);
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) out_ok[u] = Ct[u] > 0;
        }
        if (exit_bits && b == n_blocks - 1) {
            // a range of a stream split over GPUs: where, behind the range's last bit, the next codeword begins -- one past
            // the first completion seen from the node the range ends in (the word behind the lane is in W; 0 at the root)
            uint32_t Rx[TW_LANES], Cx[TW_LANES] = {};
            TwTrack tx;
#pragma unroll
            for (int u = 0; u < TW_LANES; ++u) {
                Rx[u] = r[u].s_out() << 9;
                tx.found[u] = r[u].s_out() == 0;
                tx.start[u] = 0;
            }
            tw_walk<TW_OWN + 512, 4, false, false, true, false>(W, Rx, Cx, skip, no_limit, 0, tx);
            if (lane_id == 63) *exit_bits = tx.found[1] ? tx.start[1] : 0u;
        }
        // (the lanes' numbers worked out again rather than kept in four registers across the walks: the kernel is at its register limit)
        uint32_t b_again = b;
        asm volatile("" : "+v"(b_again));
        uint32_t sum = 0;
        // (wave_shl:1: the lane behind's; lane 63 keeps `old`, which the selects below never look at)
        const uint32_t next_st0 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(st1[0]), 0x130, 0xf, 0xf, false));
        const uint32_t next_st1 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(st1[1]), 0x130, 0xf, 0xf, false));
        const uint32_t first1 = __builtin_amdgcn_readlane(st1[1], 0);
#pragma unroll
        for (int u = 0; u < TW_LANES; ++u) {
            const uint32_t begun1 = r[u].c1() - ((start[u] != 0 && f1[u]) ? 1u : 0u) + ((r[u].s_mid() != 0 && f2[u]) ? 1u : 0u);
            const uint32_t begun2 = r[u].c2() - ((r[u].s_mid() != 0 && f2[u]) ? 1u : 0u) + ((r[u].s_out() != 0 && out_ok[u]) ? 1u : 0u);
            // the start of the subsequence after this lane: the next lane's (the block's last lane: not known here, 0)
            const uint32_t after = u == 0 ? (lane_id == 63 ? first1 : next_st0) : (lane_id == 63 ? 0u : next_st1);
            const uint64_t qu = static_cast<uint64_t>(b_again) * 128 + u * 64 + lane_id;
            if (live[u]) {
                uint32_t first = st1[u];
                if (gave_up && u == 0 && lane_id == 0) first = 0xffu;  // the marker the write kernels' launch rule knows
                sub_state[2 * qu] = first | (st2[u] << 8) | (begun1 << 16);
                if (2 * qu + 1 < n_subs) sub_state[2 * qu + 1] = st2[u] | (after << 8) | (begun2 << 16);
                sum += begun1 + begun2;
            }
        }
        sum = tw_wave_inclusive_scan(sum);
        if (lane_id == 63) {
            blk_count[b] = sum;
            blk_exit[b] = r[1].s_out();  // (lanes past the stream's end stand at the root)
        }
        if (lane_id == 0) {
            blk_start[b] = gave_up ? 0xffffffffu : start[0];
            if (gave_up) atomicAdd(changed + 1, 1u);
            if (worklist) *changed = 1;
        }
        }
    }
}

__global__ __launch_bounds__(256) void k_tw_check(const uint32_t *__restrict__ blk_start, const uint32_t *__restrict__ blk_exit, uint32_t n_blocks,
                                                  uint32_t *__restrict__ worklist, uint32_t *__restrict__ n_work, uint32_t first_known) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks || (b == 0 && !first_known)) return;  // (a range that does not know its first bit: nothing to hold block 0 against)
    const uint32_t want = b ? blk_exit[b - 1] : 0u;  // (the stream's first block starts at a codeword boundary)
    if (blk_start[b] != want) worklist[atomicAdd(n_work, 1u)] = b;
}

// ---- launch wrappers --------------------------------------------------------------------------------
void launch_tw_build(hipStream_t stream, const TwUpload *d_up, uint32_t up_bytes, uint32_t n_int, uint16_t *table, uint32_t n_chain, uint64_t *chain, uint32_t *zero16,
                     uint32_t *zero_words, uint32_t n_zero) {
    const uint32_t entries = (table ? tw_table_entries(n_int) : 0) + (chain ? n_chain : 0);
    if (!entries) return;
    hipLaunchKernelGGL(k_tw_build, dim3((entries + 1023) / 1024), dim3(1024), 0, stream, d_up, (up_bytes + 3u) & ~3u, n_int, table, chain ? n_chain : 0u, chain, zero16, zero_words,
                       zero_words ? n_zero : 0u);
}

void launch_tw_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, const uint16_t *table,
                    uint32_t n_int, uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_start, uint32_t *blk_count, uint32_t *changed,
                    uint32_t max_trips, const uint32_t *worklist, const uint32_t *n_work, KernelEvents ev, uint32_t *blk_pub, uint32_t mode, uint32_t *exit_bits) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    const uint32_t entries = tw_table_entries(n_int);
    size_t smem = static_cast<size_t>(entries) * 2;
#ifdef ET_PROBE_FUSED_OCC
    if (smem < 84u * 1024u) smem = 84u * 1024u;
#endif
    // workgroups of 8 wavefronts, as many per CU as the table leaves room for in the LDS, at most 3 (<= 80 VGPRs: 6
    // wavefronts per SIMD); a table that leaves room for one workgroup only gets one of 16 wavefronts
    static thread_local int seen_dev = -1, cus = 256;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != seen_dev) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        seen_dev = dev;
    }
    uint32_t per_cu = static_cast<uint32_t>((160u * 1024u) / smem);
#ifndef ET_TW_PER_CU_MAX
#define ET_TW_PER_CU_MAX 3
#endif
    per_cu = per_cu < 1 ? 1 : (per_cu > ET_TW_PER_CU_MAX ? ET_TW_PER_CU_MAX : per_cu);
    // wavefronts per workgroup: what ET_TW_WAVES_PER_EU per SIMD come to on a CU, shared out among its workgroups (<= 16)
    uint32_t waves = (4u * ET_TW_WAVES_PER_EU + per_cu - 1) / per_cu;
    if (waves * per_cu > 4u * ET_TW_WAVES_PER_EU) --waves;
    waves = waves > 16 ? 16 : waves;
#ifdef ET_PROBE_FUSED_OCC  // (round-4 probe: the sweep at the occupancy of a kernel that also holds the write pass's tables and stages)
    per_cu = 1;
    waves = ET_PROBE_FUSED_OCC;
#endif
    const uint32_t threads = waves * 64;
    uint32_t grid = static_cast<uint32_t>(cus) * per_cu;
    if (grid > (n_blocks + waves - 1) / waves) grid = (n_blocks + waves - 1) / waves;
    if (worklist && grid > 64) grid = 64;  // a repair sweep: a handful of blocks (workgroups beyond the list leave at once)
    if (ev.start || ev.stop) hipExtLaunchKernelGGL(k_tw_sync, dim3(grid), dim3(threads), smem, stream, ev.start, ev.stop, 0, words, n_bytes, first_bit, n_subs, n_blocks, table, entries, n_int, sub_state, blk_exit, blk_start, blk_count, changed, max_trips, worklist, n_work, blk_pub, mode, exit_bits);
    else hipLaunchKernelGGL(k_tw_sync, dim3(grid), dim3(threads), smem, stream, words, n_bytes, first_bit, n_subs, n_blocks, table, entries, n_int, sub_state, blk_exit, blk_start, blk_count, changed, max_trips, worklist, n_work, blk_pub, mode, exit_bits);
}

void launch_tw_check(hipStream_t stream, const uint32_t *blk_start, const uint32_t *blk_exit, uint32_t n_blocks, uint32_t *worklist, uint32_t *n_work, bool first_known) {
    hipLaunchKernelGGL(k_tw_check, dim3((n_blocks + 255) / 256), dim3(256), 0, stream, blk_start, blk_exit, n_blocks, worklist, n_work, first_known ? 1u : 0u);
}

}  // namespace et

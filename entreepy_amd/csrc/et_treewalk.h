// et_treewalk.h -- the fixed-rate synchronisation walk ("tree walk", et_treewalk.hip): its table format,
// the host's part (the code table as a tree), launch wrappers.
//
// decode.zig:143-203 matches one codeword at a time against a map keyed by code value.  The register-window
// walks (et_kernels_fallback.hip; the write walk of et_kernels.hip) look whole codewords up greedily, so lanes take different numbers of steps and codes
// longer than the lookup index leave through an escape path -- on a long-tailed alphabet (enwik: ~200
// symbols, codes past 20 bits) a third of a percent of the symbols escape, which stalls ~70 % of a
// wavefront's word iterations.  Here the code table is its binary tree (root = a codeword boundary, one
// table row per internal node) and every lane consumes exactly one byte per step:
//     entry[row][byte] (u16) = row after the byte (a tree node: < 256) | codewords completed in it << 8
//                              | bit (0..7) at which the first of them ends << 12
// (nibble-aligned: the walk takes the next row with one SDWA shift and adds the count with one v_dot8_u32_u4)
// No exit test, no escape, no divergence; a 32-bit code is a path through four rows.  What the walk yields
// per 256-bit subsequence is what the write kernels expect: the bit offset at which its first codeword
// begins and the number of codewords that begin inside it.
//
// No HIP in the tree part: et_treewalk_host.cpp also compiles with plain g++ (tests/test_sanitizers.py).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "entreepy_hip.h"

namespace et {

constexpr uint32_t TW_ROW_BITS = 8, TW_ROW_MASK = (1u << TW_ROW_BITS) - 1u;  // a byte leads to a tree NODE (< TW_MAX_NODES <= 256); the entry rows are only walked FROM
constexpr uint32_t TW_N_SHIFT = 8, TW_OFF_SHIFT = 12;
constexpr uint32_t TW_MAX_NODES = 256;  // internal nodes of a FULL tree over <= 256 leaves (<= 255), + slack
constexpr uint32_t TW_ENTRY_ROWS = 7;   // rows S_1 .. S_7 behind the nodes': "skip the first b bits of the byte, then from the root"
constexpr int16_t TW_LEAF0 = -2;        // child value of a leaf: TW_LEAF0 - symbol

// The code table as a tree, for the device: child[2 * node + bit] >= 0 an internal node, <= TW_LEAF0 a leaf.
struct TwTree {
    uint32_t n_int;  // internal nodes (node 0 = root)
    uint32_t pad_;
    int16_t child[2 * TW_MAX_NODES];
};

// ET_OK when the codes form a FULL binary tree of at most TW_MAX_NODES internal nodes with >= 2 leaves: what an
// encoder produces (also under the reference's quirk Q1, whose tree is built without the dropped symbol).  A
// hand-made dictionary may leave bit patterns without a symbol; complete = true gives every such pattern a leaf of
// its own that decodes as byte 0 (no stream of a dictionary's own encoder contains one), so that any prefix-free
// dictionary -- down to a single code -- walks like an encoder's; complete = false turns them away with
// ET_ERR_UNSUPPORTED, as both do a code that is a prefix of another or is longer than 32 bits.
int tw_build_tree(const et_codebook *cb, TwTree *tree, bool complete = false);
// Host fill of the table (rows = n_int + TW_ENTRY_ROWS, 256 entries each): the reference the device's
// k_tw_build is tested against.
void tw_fill_table(const TwTree *tree, uint16_t *table);
#ifdef __HIPCC__
#define ET_TW_HD __host__ __device__
#else
#define ET_TW_HD
#endif
ET_TW_HD inline uint32_t tw_rows(uint32_t n_int) { return n_int + TW_ENTRY_ROWS; }
ET_TW_HD inline uint32_t tw_table_entries(uint32_t n_int) { return tw_rows(n_int) << 8; }

// ---------------------------------------------------------------------------------------------------------
// Chained lookup tables for the WRITE walk (D3, k_dec_write_wave in et_kernels.hip).  decode.zig:143-203 again:
// the greedy walk looks up to two whole codewords up per step in a table indexed by the next CH_ROOT_BITS bits.
// A code longer than the index used to be an "escape" that threw its lane out of the wavefront's lockstep loop;
// on a long-tailed alphabet one wavefront step in three contains one.  Here such an index is an ordinary entry --
// no symbol, all index bits consumed, "continue in table T with an index of s bits" -- where T is the table of
// the tree node the bits lead to.  Every table entry names the table of the NEXT lookup (the root table after a
// completed codeword), so a step never branches on what it found.  Entry (u64, read with one ds_read_b64):
//   lo  [15:0]  signed 16-bit: (symbols << 10) - bits consumed, added to the walk state
//       [23:16] first symbol          [31:24] bits up to the end of the first symbol (0: none completes here)
//   hi  [15:0]  byte offset of the next lookup's table        [23:16] second symbol
//       [31:24] right shift that turns a 32-bit stream window into the next lookup's index (32 - its index bits)
constexpr uint32_t CH_ROOT_BITS = 11, CH_SUB_BITS_MAX = 9, CH_SUB_ENTRIES_MAX = 576, CH_MAX_TABLES = 256;
// (576 sub-table entries: with the 16 KiB root and two 16 KiB stages a 512-thread workgroup stays under a third of the LDS)
constexpr uint32_t CH_MAX_ENTRIES = (1u << CH_ROOT_BITS) + CH_SUB_ENTRIES_MAX;

struct ChainTable {
    uint16_t node;  // the tree node a lookup in this table starts from (table 0: the root)
    uint8_t bits;   // index width
    uint8_t pad_;
    uint32_t first;  // index of its first entry
};
struct ChainPlan {
    uint32_t n_tables, n_entries, sub_bits, pad_;
    int16_t table_of[TW_MAX_NODES];  // tree node -> its table, -1: has none
    ChainTable tab[CH_MAX_TABLES];   // (only the first n_tables travel to the device)
};
// What the device needs for both tables of a code: the tree, then the plan.
struct TwUpload {
    TwTree tree;
    ChainPlan plan;
};
inline size_t tw_upload_bytes(const TwUpload *u) { return offsetof(TwUpload, plan) + offsetof(ChainPlan, tab) + u->plan.n_tables * sizeof(ChainTable); }

// Which tables a tree needs: the root's, and one for every internal node that a lookup can end on without having
// completed a codeword, as wide as the subtree below it is deep but at most sub_bits -- the largest value (<=
// CH_SUB_BITS_MAX) for which all of them fit CH_SUB_ENTRIES_MAX entries.  Always succeeds for a tree of tw_build_tree.
void tw_chain_plan(const TwTree *tree, ChainPlan *plan);

ET_TW_HD inline uint64_t tw_chain_entry(const TwTree *tree, const ChainPlan *plan, uint32_t t, uint32_t idx) {
    const uint32_t s = plan->tab[t].bits;
    uint32_t node = plan->tab[t].node, used = 0, n = 0, sym1 = 0, sym2 = 0, len_first = 0, done = 0;
    while (used < s) {
        const int16_t c = tree->child[2 * node + ((idx >> (s - 1 - used)) & 1u)];
        ++used;
        if (c >= 0) {
            node = static_cast<uint32_t>(c);
            continue;
        }
        const uint32_t sym = static_cast<uint32_t>(TW_LEAF0 - c);
        if (n == 0) {
            sym1 = sym;
            len_first = used;
        } else {
            sym2 = sym;
        }
        done = used;
        node = 0;
        if (++n == 2) break;
    }
    uint32_t next = 0, adv;
    if (n == 0) {
        next = static_cast<uint32_t>(plan->table_of[node]);
        adv = (0u - s) & 0xffffu;
    } else {
        adv = (n << 10) - done;
    }
    const uint32_t lo = adv | (sym1 << 16) | (len_first << 24);
    const uint32_t hi = (plan->tab[next].first * 8u) | (sym2 << 16) | ((32u - plan->tab[next].bits) << 24);
    return lo | (static_cast<uint64_t>(hi) << 32);
}
// Host fill (n_entries u64): what k_tw_chain_build is tested against.
void tw_chain_fill(const TwTree *tree, const ChainPlan *plan, uint64_t *table);

}  // namespace et

#ifdef __HIPCC__
#include "et_kernels.h"

namespace et {

// table: tw_table_entries(n_int) u16 in device memory (or null), chain: n_chain u64 (or null); both filled by one
// launch from a TwUpload (up_bytes = tw_upload_bytes of it) in device memory or in pinned host memory.  zero16 (optional): 16 words the kernel also clears
// (the decode's flags); zero_words / n_zero (optional): more of them (launch_tw_sync's blk_pub).
void launch_tw_build(hipStream_t stream, const TwUpload *d_up, uint32_t up_bytes, uint32_t n_int, uint16_t *table, uint32_t n_chain, uint64_t *chain, uint32_t *zero16 = nullptr,
                     uint32_t *zero_words = nullptr, uint32_t n_zero = 0);

// D1 by tree walk.  Outputs as the register-window sweep's: sub_state[s] = start bit | (start bit of s + 1) << 8
// | codewords that begin in s << 16; blk_count[b]; and, as ROWS instead of bit offsets, blk_exit[b] = the tree
// node at the end of 8 KiB block b, blk_start[b] = the node its first lane started from (0xffffffff: the block
// gave up within max_trips; changed[1] counts those).  worklist != null: only the blocks listed there
// (n_work of them), each with its first lane started from blk_exit[b - 1] (the repair sweep; changed[0] is
// raised when there was anything to repair).
void launch_tw_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, const uint16_t *table,
                    uint32_t n_int, uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_start, uint32_t *blk_count, uint32_t *changed,
                    uint32_t max_trips, const uint32_t *worklist, const uint32_t *n_work, KernelEvents ev = {},
                    uint32_t *blk_pub = nullptr, uint32_t mode = 0, uint32_t *exit_bits = nullptr);  // mode / exit_bits: a range of a stream that began earlier (TW_* below).   blk_pub (first sweep; n_blocks words, zeroed): the blocks also settle their seams with the blocks before them (see k_tw_sync)
// k_tw_sync's mode bits for a RANGE of a stream split over GPUs: the four words in front of `words` are stream bytes;
// the range's first lane does not know its first bit (it runs in like any other lane, first_bit is ignored).
// exit_bits (optional): receives the bit offset behind the range's end at which the next codeword begins.
constexpr uint32_t TW_FRONT_OK = 1, TW_START_UNKNOWN = 2;
// Blocks whose first lane did not start where the block before ends -> worklist (n_work zeroed by the caller).
void launch_tw_check(hipStream_t stream, const uint32_t *blk_start, const uint32_t *blk_exit, uint32_t n_blocks, uint32_t *worklist, uint32_t *n_work,
                     bool first_known = true);  // first_known = false: block 0 began at an unknown bit (TW_START_UNKNOWN) and is not checked

}  // namespace et
#endif

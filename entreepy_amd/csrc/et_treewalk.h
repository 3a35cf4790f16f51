// et_treewalk.h -- the fixed-rate synchronisation walk ("tree walk", et_treewalk.hip): its table format,
// the host's part (the code table as a tree), launch wrappers.
//
// decode.zig:143-203 matches one codeword at a time against a map keyed by code value.  The register-window
// walks (et_kernels.hip) look whole codewords up greedily, so lanes take different numbers of steps and codes
// longer than the lookup index leave through an escape path -- on a long-tailed alphabet (enwik: ~200
// symbols, codes past 20 bits) a third of a percent of the symbols escape, which stalls ~70 % of a
// wavefront's word iterations.  Here the code table is its binary tree (root = a codeword boundary, one
// table row per internal node) and every lane consumes exactly one byte per step:
//     entry[row][byte] (u16) = row after the byte | codewords completed in it << 9
//                              | bit (0..7) at which the first of them ends << 13
// No exit test, no escape, no divergence; a 32-bit code is a path through four rows.  What the walk yields
// per 256-bit subsequence is what the write kernels expect: the bit offset at which its first codeword
// begins and the number of codewords that begin inside it.
//
// No HIP in the tree part: et_treewalk_host.cpp also compiles with plain g++ (tests/test_sanitizers.py).
#pragma once

#include <stdint.h>

#include "entreepy_hip.h"

namespace et {

constexpr uint32_t TW_ROW_BITS = 9, TW_ROW_MASK = (1u << TW_ROW_BITS) - 1u;  // <= 512 rows
constexpr uint32_t TW_N_SHIFT = 9, TW_OFF_SHIFT = 13;
constexpr uint32_t TW_MAX_NODES = 256;  // internal nodes of a FULL tree over <= 256 leaves (<= 255), + slack
constexpr uint32_t TW_ENTRY_ROWS = 7;   // rows S_1 .. S_7 behind the nodes': "skip the first b bits of the byte, then from the root"
constexpr int16_t TW_LEAF0 = -2;        // child value of a leaf: TW_LEAF0 - symbol

// The code table as a tree, for the device: child[2 * node + bit] >= 0 an internal node, <= TW_LEAF0 a leaf.
struct TwTree {
    uint32_t n_int;  // internal nodes (node 0 = root)
    uint32_t pad_;
    int16_t child[2 * TW_MAX_NODES];
};

// ET_OK when the walk applies: the codes form a FULL binary tree (what an encoder produces; a corrupted
// dictionary that is still prefix-free may not) of at most TW_MAX_NODES internal nodes with >= 2 leaves.
// ET_ERR_UNSUPPORTED otherwise: the caller keeps the register-window sweep.
int tw_build_tree(const et_codebook *cb, TwTree *tree);
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

}  // namespace et

#ifdef __HIPCC__
#include "et_kernels.h"

namespace et {

// table: tw_table_entries(n_int) u16 in device memory, filled by launch_tw_build from a TwTree in device memory.
void launch_tw_build(hipStream_t stream, const TwTree *d_tree, uint32_t n_int, uint16_t *table);

// D1 by tree walk.  Outputs as the register-window sweep's: sub_state[s] = start bit | (start bit of s + 1) << 8
// | codewords that begin in s << 16; blk_count[b]; and, as ROWS instead of bit offsets, blk_exit[b] = the tree
// node at the end of 8 KiB block b, blk_start[b] = the node its first lane started from (0xffffffff: the block
// gave up within max_trips; changed[1] counts those).  worklist != null: only the blocks listed there
// (n_work of them), each with its first lane started from blk_exit[b - 1] (the repair sweep; changed[0] is
// raised when there was anything to repair).
void launch_tw_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, const uint16_t *table,
                    uint32_t n_int, uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_start, uint32_t *blk_count, uint32_t *changed,
                    uint32_t max_trips, const uint32_t *worklist, const uint32_t *n_work, KernelEvents ev = {});
// Blocks whose first lane did not start where the block before ends -> worklist (n_work zeroed by the caller).
void launch_tw_check(hipStream_t stream, const uint32_t *blk_start, const uint32_t *blk_exit, uint32_t n_blocks, uint32_t *worklist, uint32_t *n_work);

}  // namespace et
#endif

// et_fsm.h -- fixed-rate decode tables ("FSM walk") for the decode kernels in et_fsm.hip.
//
// decode.zig:143-203 matches one codeword at a time against a hash map keyed by code value.
// Here the code table is its binary tree (root = a codeword boundary, one row per internal
// node) and a lane consumes a FIXED number of bits K per step: entry[row][next K bits] =
// {row after those bits, symbols completed inside them}.  Every lane of a wavefront makes
// the same number of steps for the same number of bits, so the walk has no per-step exit
// test, no escape for long codes (a 32-bit code is just a path through 32 / K rows) and no
// divergence; bit positions are compile-time constants.
//
//   sync table  (k_fsm_sync):  u16  bits 0..10 next row | bits 11..14 symbols completed (<= K)
//   write table (k_fsm_write): u32  next row << fsm_write_row_shift(K) (= its byte offset in the
//                                   table: `entry & mask` IS the next lookup's row address)
//                                   | bits 3..4 8 x symbols (0, 8, 16) | bits 16..23 first symbol
//                                   | bits 24..31 second symbol
// Rows: 0 = root, 1 .. n_int-1 the other internal nodes, then K-1 entry rows S_1 .. S_{K-1}
// ("skip the first b bits of the field, then decode from the root") for walks that begin at
// a bit offset instead of in a state (the stream's first bit; ranges with a known start).
// Incomplete (corrupted) prefix sets: a missing branch leads back to the root without a symbol.
//
// No HIP in here: et_fsm_tables.cpp also compiles with plain g++ (tests/test_sanitizers.py).
#pragma once

#include <stdint.h>

#include "entreepy_hip.h"

namespace et {

constexpr uint32_t FSM_ROW_BITS = 11, FSM_ROW_MASK = (1u << FSM_ROW_BITS) - 1u, FSM_MAX_ROWS = 1u << FSM_ROW_BITS;
constexpr uint32_t FSM_SYNC_N_SHIFT = 11;   // u16 sync entry: symbols completed
constexpr uint32_t FSM_WRITE_N8_MASK = 0x18; // u32 write entry: 8 x symbols completed, in place
// Write table: a row is 1 << shift bytes (4 << K of entries; K = 2 pads to 32 so that bits 3..4 stay free),
// and the next-row field ends at bit 15: at most 1 << (16 - shift) rows can be a step's target.
constexpr uint32_t fsm_write_row_shift(uint32_t k) { return k + 2 > 5 ? k + 2 : 5; }
constexpr uint32_t fsm_write_row_mask(uint32_t k) { return 0xffffu & ~((1u << fsm_write_row_shift(k)) - 1u); }
constexpr uint32_t fsm_write_max_targets(uint32_t k) { return 1u << (16 - fsm_write_row_shift(k)); }
constexpr uint32_t FSM_MAX_NODES = 256 * 32;  // internal nodes of any prefix-free set of <= 256 codes of <= 32 bits

// The code table as a tree.  child[2 * node + bit]: >= 0 internal node, FSM_NONE no branch,
// <= FSM_LEAF0 a leaf (symbol = FSM_LEAF0 - value).
constexpr int32_t FSM_NONE = -1, FSM_LEAF0 = -2;
struct FsmTree {
    uint32_t n_int;  // internal nodes, >= 1 (node 0 = root)
    int32_t child[2 * FSM_MAX_NODES];
};

// What the host decides (and uploads): the tree and the two step widths.  The device fills the
// tables (et_fsm.hip k_fsm_build); fsm_fill_* below are the host reference it is tested against.
struct FsmPlan {
    uint32_t n_int;
    uint32_t k_sync, k_write;        // bits per step (sync: 8, 4 or 2; write: 6, 4 or 2)
    uint32_t rows_sync, rows_write;  // n_int + K - 1
    uint32_t max_per_lane;           // most symbols that can begin (or end) inside 256 bits (stage sizing)
    uint32_t pad_[2];
};

// Returns ET_OK, or ET_ERR_UNSUPPORTED when the tree has more rows than the tables can name
// (only a crafted dictionary gets there: an encoder's tree has n_coded - 1 internal nodes).
int fsm_build_tree(const et_codebook *cb, FsmTree *tree);
// Step widths: the widest the LDS budget admits; the write table also needs <= 2 symbols per entry.
int fsm_plan(const et_codebook *cb, const FsmTree *tree, uint32_t lds_budget_sync, uint32_t lds_budget_write, FsmPlan *plan);
// Host fills (reference): sync[rows << k] u16, write[rows << (fsm_write_row_shift(k) - 2)] u32.  fsm_max_symbols_per_entry:
// the largest symbol count of any write entry at width k (the planner's test).
void fsm_fill_sync(const FsmTree *tree, uint32_t k, uint16_t *table);
void fsm_fill_write(const FsmTree *tree, uint32_t k, uint32_t *table);
uint32_t fsm_max_symbols_per_entry(const FsmTree *tree, uint32_t k);

inline uint32_t fsm_rows(uint32_t n_int, uint32_t k) { return n_int + k - 1; }

}  // namespace et

// et_tables.h -- formats of the decode lookup tables and their host-side builders.
// No HIP in here: et_tables.cpp also compiles with plain g++ (tests/test_sanitizers.py
// runs the builders under AddressSanitizer/UBSan against a brute-force decoder).
#pragma once

#include <stdint.h>

#include "entreepy_hip.h"

namespace et {

constexpr uint32_t DEC_LUT_BITS_MAX = 12;    // older-format first-level table: at most 4096 x u32
constexpr uint32_t DEC_LUT_BITS_WRITE = 11;  // index width of the write kernels' tables: measured 10 / 11 / 12 -> 0.65 / 0.60 / 0.68 ms at 1 GiB
// Older format (LDS-window kernels k_dec_sync / k_dec_write / k_dec_maps / k_dec_resolve, and
// the slow path of the step walks).  First-level entry (u32), indexed by the next lut_bits
// bits: bytes 0..2 = up to three symbols whose codes all fit in the index, bits 24..27 = total
// length of those codes, bits 28..29 = how many (0: the first code is longer than the table,
// or no code starts here).  Escape entries (count 0): bit 30 set -> byte 0 is the index of a
// second-level table of 1 << sub_bits u16 entries ((len << 8) | sym, 0 = not here) indexed
// by the sub_bits bits that follow the first lut_bits.  longc: {left-aligned code, (len << 8)
// | sym} of every code longer than lut_bits.
constexpr uint32_t LUT_LEN_SHIFT = 24, LUT_N_SHIFT = 28, LUT_SUB_SHIFT = 30;
constexpr uint32_t DEC_WRITE_SYMS = 2;  // symbols per entry: k_dec_write stores at most two per step
constexpr uint32_t DEC_SUB_BITS_MAX = 8, DEC_SUB_TABLES_MAX = 16;

// Step table of k_dec_sync_reg / k_dec_sync_reg2 (no symbols, only how far a lookup moves the
// walk).  The walk state X is one u32: bits 0..15 = G, the position relative to the current
// register pair (et_kernels_fallback.hip walk_steps), bits 16..27 = symbols begun, bits 28..31 junk;
// entry = (len_first << 28) + (n << 16) - len_total is simply ADDED to X (n = all the whole
// codes inside the index, len_first = length of the first one for single steps).  No code
// inside the index: STEP_ESCAPE = one "symbol" of STEP_ESCAPE_BITS bits, which throws the
// lane out of the word loop it is in; the loop's exit test sees how far it flew and resolves
// the long code: bits 28..31 of the escape entry = 1 + the second-level table (entries
// (1 << 16) - len, 0 = not here) indexed by the step_sub_bits bits after the index, 0 = no
// table (older-format tables in global memory, slow).
constexpr uint32_t DEC_STEP_BITS_MAX = 13, DEC_STEP_BITS_DEFAULT = 12;
constexpr uint32_t STEP_ESCAPE_BITS = 64, STEP_ESCAPE = (1u << 16) - STEP_ESCAPE_BITS;
// k_dec_write_reg's table has the same shape with symbols: state X = (stage address << 10) |
// G; entry = sym2 << 24 | sym1 << 16 | u16((n << 10) - len_total), n <= 2, whose low half
// is added to X; escape = WSTEP_ESCAPE in the low half, 1 + second-level table in bits
// 24..31; second-level entries sym << 16 | ((1 << 10) - len), 0 = not here.
constexpr uint32_t WSTEP_ESCAPE = (1u << 10) - STEP_ESCAPE_BITS;
constexpr uint32_t DEC_STEP_SUB_WORDS = 1024;  // second-level entries, all tables of one step table together

struct HostDecodeTables {
    uint32_t lut_bits, n_long, sub_bits, n_sub;
};

// What the HOST decides about a code table's decode tables -- index widths, which long codes get
// which second-level table, the order of the long list -- handed to the device, which fills the
// tables themselves (et_kernels_fallback.hip k_build_dec_tables: same entries as the builders below, which
// stay as the reference the device's output is tested against, and as ET_DEC_TABLES_HOST=1).
struct TablePlan {
    uint32_t data[256];
    uint8_t length[256];
    uint8_t lut_sub[256];    // symbol longer than lut_bits: 1 + its second-level table in the older format (0: none)
    uint8_t step_sub[256];   // ... in the step table
    uint8_t wstep_sub[256];  // ... in the write-step table
    uint8_t long_idx[256];   // ... its position in the long list
    uint32_t lut_bits, n_long, sub_bits, n_sub, max_syms;
    uint32_t step_bits, step_sub_bits, n_step_sub;
    uint32_t wstep_bits, wstep_sub_bits, n_wstep_sub;
    uint32_t pad_;
};
void plan_tables(const et_codebook *cb, uint32_t lut_bits_max, uint32_t max_syms, uint32_t step_bits_max, uint32_t wstep_bits_max, TablePlan *plan);

// Older-format tables: lut[1 << k], longc[2 * n_long], sub[n_sub << sub_bits]; k =
// min(longest code, lut_bits_max).
void build_decode_tables(const et_codebook *cb, uint32_t lut_bits_max, uint32_t max_syms, uint32_t *lut, uint32_t *longc, uint16_t *sub,
                         HostDecodeTables *out);
// Step tables: steps[1 << k] followed by *n_sub second-level tables of 1 << *sub_bits
// entries (at most DEC_STEP_SUB_WORDS in all).  Return k = min(longest code, bits_max).
uint32_t build_step_table(const et_codebook *cb, uint32_t bits_max, uint32_t *steps, uint32_t *sub_bits, uint32_t *n_sub);
uint32_t build_write_step_table(const et_codebook *cb, uint32_t bits_max, uint32_t *steps, uint32_t *sub_bits, uint32_t *n_sub);

}  // namespace et

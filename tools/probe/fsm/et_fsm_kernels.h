// et_fsm_kernels.h -- launch wrappers of et_fsm.hip (the fixed-rate decode walks).
#pragma once

#include "et_fsm.h"
#include "et_kernels.h"

namespace et {

// Device-resident tables of one code table, as the kernels take them.
struct FsmTables {
    const uint16_t *sync;   // [rows_sync << k_sync]
    const uint32_t *write;  // [rows_write << k_write]
    uint32_t n_int, k_sync, k_write, rows_sync, rows_write, max_per_lane;
    uint32_t sync_threads, write_threads;  // workgroup sizes; 0 = the launchers' choice (tuning / tests)
    uint32_t write_window;                 // symbols staged per pass of a wavefront of k_fsm_write; 0 = from the stream's mean
};

size_t fsm_sync_smem(uint32_t table_entries);
uint32_t fsm_write_wave_stage(uint32_t window, uint32_t max_per_lane);
uint32_t fsm_write_table_words(uint32_t rows, uint32_t k);  // u32 words of a write table (rows padded to their stride, total to 16 bytes)

// D1, first sweep.  sub_state[s] = start row | exit row << 11 | symbols that end in s << 22;
// blk_exit[b] = row at the end of 8 KiB block b, blk_count[b] = its symbols; changed[1] += blocks
// that did not settle within max_trips.
void launch_fsm_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, const FsmTables &ft,
                     uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_count, uint32_t *changed, uint32_t max_trips, uint32_t flags,
                     KernelEvents ev = {});
// D3.  offset_mode: sub_state in the older layout (start bit | exit << 8 | symbols that begin << 16).
void launch_fsm_write(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, const FsmTables &ft,
                      const uint32_t *sub_state, const unsigned long long *blk_off, uint64_t n_symbols, uint8_t *out, bool offset_mode,
                      bool have_start, const uint32_t *void_flags = nullptr, KernelEvents ev = {});

}  // namespace et

// et_rowsync.h -- synchronisation of 8-bit near-fixed-length codes in ONE pass ("row walk", et_rowsync.hip).
//
// decode.zig:143-203 on the streams BASELINE.json calls its worst case: (nearly) uniform bytes.  With 129..256 symbols of
// about equal weight the reference's tree (encode.zig:82-138) is complete with depth 8: T codes of 7 bits -- as numbers
// 0 .. T-1, the two-queue merge hands them out in that order -- and 256 - 2T codes of 8 bits.  Uniform random bytes give 255
// coded symbols (the reference drops one of 256: SURVEY Q1) and T = 1.  Such a code never re-synchronises, so the tree walk's
// run-in (et_treewalk.hip) does not apply, and the general answer -- exit maps for every start offset, et_kernels_fallback.hip's
// k_dec_maps_reg / k_dec_resolve_reg -- walks every subsequence L + 1 = 9 times: 11.3 of the 16 ms a 4 GiB decode took.
//
// Here the stream is a matrix of BYTES: a codeword that begins at bit r of byte j ends at bit r of byte j + 1 (8 bits) or at
// bit r - 1 (7 bits; from r = 0 it ends at bit 7 of the SAME byte).  A walk is a path down the rows that moves one column to
// the left at every 7-bit code, and "is the code at (j, r) a 7-bit one" is a comparison of the 7 bits there with T -- computed
// for all 256 positions of a 256-bit subsequence at once, four rows per instruction, no table and no dependent lookups.
#pragma once

#include <stdint.h>

#include "entreepy_hip.h"

namespace et {

struct RowCode {
    uint32_t t;  // the 7-bit values 0 .. t-1 are codewords; every other codeword has 8 bits
};

// true when cb is such a code: lengths 7 and 8 only (or 7 only: 128 codewords, t = 128), complete (2 * n7 + n8 = 256), prefix-free, the 7-bit codes are the
// values 0 .. n7-1.  (No HIP in this function: et_rowsync_host.cpp compiles with plain g++.)
bool row_code_of(const et_codebook *cb, RowCode *rc);
// true when cb is a code of L- and (L + 1)-bit codewords, L <= 7, that re-synchronises within the tree walk's reach all the same
// (most of them do: et_rowsync_host.cpp): the decode then tries the tree walk first instead of going to the exit maps at once.
bool quick_to_synchronise(const et_codebook *cb);

#ifndef ET_ROW_CHUNK_BLOCKS
#define ET_ROW_CHUNK_BLOCKS 4
#endif
constexpr uint32_t ROW_CHUNK_BLOCKS = ET_ROW_CHUNK_BLOCKS;  // 8 KiB blocks a workgroup takes per ticket (measured: 2 / 4 / 8)
// launch_row_sync flags, for a RANGE of a stream split over GPUs (et_decode_range_maps / _resolve):
constexpr uint32_t ROW_MAP_ONLY = 1;       // leave the range's map -- byte c = the column the stream behind the range is entered in when the range is entered in column c -- and nothing else
constexpr uint32_t ROW_START_UNKNOWN = 2;  // the range's first codeword may begin in any column (first_bit is ignored)

}  // namespace et

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

namespace et {

// Bytes of scratch launch_row_sync needs for a stream of n_blocks 8 KiB blocks (zeroed by the launch itself).
size_t row_sync_scratch_bytes(uint32_t n_blocks);

// words / n_bytes / first_bit / n_subs as for the other synchronisation kernels (et_kernels.h: the stream from its 4-byte
// aligned base, its first codeword at bit first_bit < 32).  Outputs as k_dec_resolve's: sub_state[s] = start bit | exit << 8 |
// codewords that begin in s << 16; blk_count[b]; blk_exit[b].
// fault: a device word (zeroed by the caller) that the kernel raises if a chunk never saw what the chunks before it publish (bit 0).
// flags: 0, or ROW_* above; d_map (optional): receives the device address (inside scratch) of the 8-byte map a ROW_MAP_ONLY launch leaves.
void launch_row_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, RowCode rc, void *scratch, uint32_t *fault,
                     uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_count, uint32_t flags = 0, const unsigned long long **d_map = nullptr);

// The write pass for such a stream, by rows (k_row_write): sub_state as launch_row_sync leaves it, blk_off from the scan of
// blk_count; at most n_symbols symbols to out (16-byte aligned).  ev_start / ev_stop: events the dispatch carries (may be null).
void launch_row_write(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, RowCode rc, const et_codebook *cb,
                      const uint32_t *sub_state, const unsigned long long *blk_off, uint64_t n_symbols, uint8_t *out, hipEvent_t ev_start = nullptr,
                      hipEvent_t ev_stop = nullptr);

// decode.zig:143-203 on a FIXED-length code -- 2^L codewords of L bits each (L <= 32): four symbols of about equal weight (a DNA
// sequence), 16 (a hex dump), 64 (base64 of random bytes).  Nothing to walk: the k-th codeword begins at bit first_bit + k L, so a
// subsequence's start, exit and count are three divisions (k_fixed_sync; outputs as launch_row_sync's).  A codeword cut by the
// stream's end is nobody's, as everywhere.
void launch_fixed_sync(hipStream_t stream, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, uint32_t code_bits, uint32_t *sub_state, uint32_t *blk_exit,
                       uint32_t *blk_count);

// The write pass for such a stream (k_fixed_write): symbol i is the code_bits bits at first_bit + i code_bits; n_out of them to out
// (16-byte aligned) -- n_out <= the whole codewords the stream holds (callers clamp).  code_bits <= 8.
void launch_fixed_write(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, const et_codebook *cb, uint64_t n_out, uint8_t *out,
                        hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

}  // namespace et
#endif

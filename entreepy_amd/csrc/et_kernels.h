// et_kernels.h -- geometry constants and launch wrappers of et_kernels.hip and et_kernels_fallback.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "et_tables.h"

// A value made "new" to the compiler at this point of the program (an empty asm statement with the value as an in/out
// operand).  Three of round 3's gains hang on such pins: K4's and D3's prefetches (where the wait for a load that was asked
// for a round / a unit ahead is placed) and D1's edge-block step limits (kept from being hoisted in front of every block).
// None changes a result, so no parity test sees one go missing -- tests/test_isa_guard.py compiles the kernels to ISA and
// checks what the pins hold in place, and builds them once more with -DET_GUARD_DROP_PINS to show that it would notice.
#ifdef ET_GUARD_DROP_PINS
#define ET_PIN(x_) ((void)0)
#else
#define ET_PIN(x_) asm volatile("" : "+v"(x_))
#endif

namespace et {

constexpr int BLOCK = 256;                    // threads per workgroup: 4 wavefronts of 64
constexpr uint32_t ROUND_BYTES = BLOCK * 16;  // one 16-byte load per lane
constexpr uint32_t MAX_ROUNDS_PER_TILE = 128;  // tile <= 512 KiB (u32 tile counters; u32 bit cursors: 2^19 symbols of <= 255 bits)
constexpr uint32_t HIST_REDUCE_GROUPS = 128;   // k_hist_reduce: two histogram columns each
constexpr uint32_t MAX_GRID = 2048;           // 256 CUs x 8 workgroups, grid-stride beyond

constexpr uint32_t SUB_BITS = 256;                             // decode: bits per lane subsequence
constexpr uint32_t DEC_BLOCK_WORDS = BLOCK * SUB_BITS / 32;    // 8 KiB of bitstream per workgroup
constexpr uint32_t DEC_GUARD_WORDS = 4;                        // words a lane may read past its workgroup's 8 KiB
// How the LDS-window decode kernels take their 8 KiB blocks (measured choices): k_dec_sync one block per
// workgroup, k_dec_write chunks of four by ticket.
constexpr uint32_t SYNC_CHUNK = 1, WRITE_CHUNK = 4;  // consecutive blocks per chunk
constexpr bool SYNC_TICKET = false, WRITE_TICKET = true;  // chunks by ticket counter vs one per workgroup
constexpr uint32_t DEC_FIRST_SWEEP_TRIPS = 6;                   // local fixed-point trips before a block is declared non-synchronising
constexpr uint32_t DEC_REPAIR_SWEEP_TRIPS = 8;                  // same cap for the two speculatively enqueued repair sweeps
// ONE rule for "the sweeps left a final synchronisation state", applied by the host to its copy of
// the flags and by a speculatively launched D3 to the flags themselves -- the two must agree, or the
// host keeps output that the kernel declined to write: the verification passed, and no more blocks
// gave up in the first sweep than the repair sweeps are meant for (beyond that the host goes the
// exhaustive way).  Blocks that gave up AND were repaired by the second sweep are fine.
__host__ __device__ inline bool dec_state_final(uint32_t gave_up, uint32_t verify_failed, uint32_t n_blocks) {
    return verify_failed == 0 && static_cast<uint64_t>(gave_up) * 64 <= n_blocks;
}
constexpr uint32_t DEC_HAVE_START = 1, DEC_FRONT_OK = 2;       // k_dec_sync flags (ranges of a stream split over GPUs)
constexpr uint32_t DEC_SPECIAL_SUPER = 8;                      // with DEC_SPECIAL_ONLY: the blocks of 16 KiB superblocks k_dec_sync_reg2 leaves out
constexpr uint32_t DEC_SPECIAL_ONLY = 4;                       // k_dec_sync flag: first/last blocks only (k_dec_sync_reg has the rest)
constexpr uint32_t DEC_FRONT_WORDS = 4;                        // words staged BEFORE the workgroup's 8 KiB (warm-up run-in)
constexpr uint32_t DEC_WARMUP_BITS = DEC_FRONT_WORDS * 32;     // run-in before each subsequence in the first sync sweep
constexpr uint32_t DEC_STAGED_WORDS = DEC_FRONT_WORDS + DEC_BLOCK_WORDS + DEC_GUARD_WORDS;
constexpr uint32_t DEC_SDATA_WORDS = (DEC_STAGED_WORDS + (DEC_STAGED_WORDS >> 5) + 4) & ~3u;  // 1 pad word per 32
constexpr uint32_t DEC_STAGE_BYTES = 16384;                    // LDS staging of decoded symbols

// Device-resident decode tables (built on the host, et_api.cpp build_decode_tables).
struct DecodeTables {
    const uint32_t *lut;     // [1 << lut_bits] first-level entries (LUT_* above)
    const uint32_t *longc;   // [n_long * 2]: {left-aligned code, (len << 8) | sym} of every code longer than lut_bits
    const uint16_t *sub;     // [n_sub << sub_bits] second-level tables
    const uint8_t *sym_len;  // [256] code length per symbol (single-symbol steps)
    uint32_t lut_bits;
    uint32_t n_long;
    uint32_t sub_bits;
    uint32_t n_sub;
    const uint32_t *steps;   // [1 << step_bits] packed walk increments for k_dec_sync_reg (STEP_* below), or null;
                             // followed by n_step_sub second-level tables of 1 << step_sub_bits entries
    uint32_t step_bits;
    uint32_t step_sub_bits;
    uint32_t n_step_sub;
    const DecodeTables *dev_copy;  // this struct in device memory (slow path of the step walks), or null
};

// The three-workgroup launches for the stream's first/last blocks take ~25 us each (one
// block's latency); given a side lane they run beside the big launch instead of after it.
struct SideLane {
    hipStream_t stream;
    hipEvent_t fork, join;
};

// Timing events carried by a launch itself (no marker packets): begin / end of that kernel.
struct KernelEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};

void launch_hist(hipStream_t stream, const uint8_t *base, uint64_t lo, uint64_t hi, uint32_t rounds_per_tile, uint32_t n_tiles,
                 uint32_t *tile_hist, unsigned long long *block_hist, unsigned long long *hist, unsigned long long *host_hist = nullptr, unsigned long long epoch = 0,  // host_hist: 256 + HIST_REDUCE_GROUPS words of pinned host memory: the totals, and per reducing workgroup `epoch` once its two are stored
                 KernelEvents ev = {}, unsigned long long *hist_also = nullptr);  // hist_also: a second device copy of the totals
// Bytes that hold the header and the dictionary of a stream whose first byte is d (decode.zig:34: d + 1 entries of at
// most 8 + 8 + 32 bits behind 5 header bytes), rounded up.
__host__ __device__ inline uint32_t header_bound(uint8_t d) { return 8u + 6u * (static_cast<uint32_t>(d) + 1u); }
// the first min(n, header_bound(d_src[0])) bytes of d_src -> host_dst (pinned host memory, 4-byte aligned, room for that
// rounded up to 4), then *host_done = epoch (pinned as well)
// d_src[0 .. n_words) dwords (4-byte aligned) -> host_dst (pinned), then *host_done = epoch (pinned)
void launch_words_to_host(hipStream_t stream, const void *d_src, uint32_t n_words, void *host_dst, unsigned long long *host_done, unsigned long long epoch);
void launch_header_to_host(hipStream_t stream, const void *d_src, uint32_t n, void *host_dst, unsigned long long *host_done, unsigned long long epoch);
// lengths: the 256 code lengths (host memory; they travel as a kernel argument).  host_src / dev_dst / copy_words: a
// pinned host block the first kernel copies into device memory for the kernels behind it (K4's code table, the header);
// once it has, it stores taken_epoch into *host_taken (pinned): the block may be filled again.
void launch_tile_scan(hipStream_t stream, const uint32_t *tile_hist, uint32_t n_tiles, const uint8_t *lengths, const uint32_t *host_src, uint32_t *dev_dst,
                      uint32_t copy_words, unsigned long long *host_taken, unsigned long long taken_epoch, unsigned long long *tile_bits, unsigned long long *group_sum, uint32_t epoch, unsigned long long base_bit,  // group_sum, epoch: k_scan_fused's pub words (zeroed once) and a value in 1 .. 65535 not used on them since
                      unsigned long long *tile_off, uint32_t *out32, const uint32_t *header_src = nullptr, uint32_t header_words = 0);  // header_src (device): the file header, copied to out32[0 .. header_words) behind the seam word's zeroing
void launch_encode(hipStream_t stream, const uint8_t *base, uint64_t lo, uint64_t hi, uint32_t rounds_per_tile, uint32_t n_tiles,
                   const unsigned long long *tile_off, const uint2 *enc_table, uint32_t max_len, uint32_t *out32, KernelEvents ev = {});
void launch_dec_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs,
                     const DecodeTables &tb, uint32_t iter, uint32_t max_trips,
                     uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_count, uint32_t *changed, uint32_t *ticket,
                     uint32_t flags = DEC_HAVE_START, uint32_t *worklist = nullptr, uint32_t *n_work = nullptr, const SideLane *side = nullptr,
                     bool ticket_is_zero = false, KernelEvents ev = {});  // ev: the first sweep's main kernel
void launch_dec_maps(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, bool have_start, uint64_t n_subs,
                     const DecodeTables &tb, uint32_t n_starts, uint32_t map_stride, uint8_t *lane_maps, uint8_t *blk_maps, uint8_t *grp_maps);
void launch_dec_resolve(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, bool const_first, uint64_t n_subs,
                        const DecodeTables &tb, uint32_t map_stride, const uint8_t *lane_maps, const uint8_t *blk_maps, const uint8_t *grp_maps,
                        uint8_t *blk_in, uint8_t *grp_in, uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_count);
void launch_dec_exhaustive(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs,
                           const DecodeTables &tb, uint32_t n_starts, uint32_t map_stride, uint8_t *lane_maps, uint8_t *blk_maps,
                           uint8_t *grp_maps, uint8_t *blk_in, uint8_t *grp_in, uint32_t *sub_state, uint32_t *blk_exit,
                           uint32_t *blk_count);
// Fill the decode tables on the device from the host's plan (d_plan in device memory).  Layout of
// the outputs as the host builders': lut[1 << lut_bits], longc[2 * n_long], sub[n_sub << sub_bits],
// sym_len[256], steps[(1 << step_bits) + second level], wsteps[(1 << wstep_bits) + second level].
// zero16 (optional): 16 words the kernel also clears (the decode's flags).
void launch_build_dec_tables(hipStream_t stream, const TablePlan *d_plan, uint32_t *lut, uint32_t *longc, uint16_t *sub, uint8_t *sym_len,
                             uint32_t *steps, uint32_t *wsteps, uint32_t *zero16 = nullptr);
void launch_dec_scan(hipStream_t stream, const uint32_t *blk_count, uint32_t n_blocks, unsigned long long *group_sum, uint32_t epoch,
                     unsigned long long *blk_off, unsigned long long *total_copy = nullptr, const uint32_t *verify_state = nullptr,
                     const uint32_t *verify_exit = nullptr, uint32_t *verify_flag = nullptr, uint32_t verify_first = 0xffffffffu,
                     const uint32_t *report_src = nullptr, uint32_t *report_dst = nullptr,  // report_dst: 15 words of pinned host memory (flags 0..11 of report_src, symbol total, then report_epoch)
                     bool verify_rows = false,  // verify_state is the tree walk's blk_start (one row per block) instead of sub_state
                     uint32_t report_epoch = 0);
void launch_dec_write(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint64_t n_subs, const DecodeTables &tb,
                      const uint32_t *sub_state,
                      const unsigned long long *blk_off, uint64_t n_symbols, uint8_t *out, uint32_t *ticket, const SideLane *side = nullptr,
                      bool ticket_is_zero = false, const uint32_t *void_flags = nullptr, KernelEvents ev = {},  // ev: the main write kernel
                      const uint64_t *chain = nullptr, uint32_t n_chain = 0, uint32_t chain_max_len = 32,  // chained lookup tables (et_treewalk.h) and the dictionary's longest code: every block by k_dec_write_wave
                      bool strips = false);  // ... by its instantiation for streams with many symbols per subsequence (quarters that overflow the stage walk once, into strips)

}  // namespace et

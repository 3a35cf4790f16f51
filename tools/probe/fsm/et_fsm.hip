// et_fsm.hip -- the fixed-rate decode walks (table formats: et_fsm.h), gfx950 / wave64.
//
//   decode.zig:143-203 -> k_fsm_sync   "D1": where do codewords begin?  Every lane owns 512 bits
//                                      of the stream in registers, runs in over the 128 bits
//                                      before them and then counts; lanes of a workgroup agree on
//                                      the state at their seams by a local fixed point.
//                         k_fsm_write  "D3": walk again from the agreed state, gather symbols four
//                                      to a dword in a register, dword stores into an LDS stage,
//                                      16-byte stores to the output.
//
// A step consumes K bits, the same K for every lane: lookups, adds and shifts only -- no loop,
// no exit test, no escape path; bit positions are compile-time constants, so a field is one
// v_bfe_u32 (or v_alignbit_b32 + shift across a word boundary).  What bounds a walk is the chain
// table read -> next address -> table read: an LDS round trip per step.  So the chain is kept to
// one VALU instruction (k_fsm_write: the entry carries the next row's address in place) or two
// (k_fsm_sync), the next read is issued before the current entry is put to use, and a thread of
// k_fsm_sync walks two lanes at once (two chains in flight per thread).
#include "et_fsm_kernels.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace et {

// ALL LDS of these kernels is the dynamic block (cdna_hip_programming.md Guideline 17), tables first:
// a table entry's row field is then an LDS address as it stands.
extern __shared__ __attribute__((aligned(16))) uint8_t fsm_smem[];
typedef __attribute__((address_space(3))) uint8_t fsm_lds_u8;
typedef __attribute__((address_space(3))) uint16_t fsm_lds_u16;
typedef __attribute__((address_space(3))) uint32_t fsm_lds_u32;
__device__ __forceinline__ uint32_t fsm_lds_addr(const void *p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((fsm_lds_u8 *)p)); }

// ---- scans / votes over a workgroup of up to 1024 threads --------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t fsm_dpp_add(uint32_t x) {
    return x + static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ uint32_t fsm_wave_inclusive_scan(uint32_t x) {
    x = fsm_dpp_add<0x111, 0xf>(x);  // row_shr:1
    x = fsm_dpp_add<0x112, 0xf>(x);  // row_shr:2
    x = fsm_dpp_add<0x114, 0xf>(x);  // row_shr:4
    x = fsm_dpp_add<0x118, 0xf>(x);  // row_shr:8
    x = fsm_dpp_add<0x142, 0xa>(x);  // row_bcast:15
    x = fsm_dpp_add<0x143, 0xc>(x);  // row_bcast:31
    return x;
}
// Exclusive prefix over the workgroup; scratch: 16 words.  ONE barrier inside; the caller separates
// two calls that reuse `scratch` by another barrier.
__device__ __forceinline__ uint32_t fsm_block_exclusive_scan(uint32_t x, uint32_t *scratch, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t inc = fsm_wave_inclusive_scan(x);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < n_waves; ++w) {
        const uint32_t v = scratch[w];
        before += w < wave ? v : 0u;
        all += v;
    }
    *total = all;
    return before + inc - x;
}

// Word `idx` (may be negative: before `words`) of the stream in host order, zero outside it.
__device__ __attribute__((noinline)) uint32_t fsm_load_guarded(const uint32_t *__restrict__ words, long long idx, uint64_t n_bytes, bool front_ok) {
    if (idx < 0) return front_ok ? __builtin_bswap32(words[idx]) : 0u;
    const uint64_t b0 = static_cast<uint64_t>(idx) * 4;
    if (b0 + 4 <= n_bytes) return __builtin_bswap32(words[idx]);
    uint32_t v = 0;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(words);
    for (int k = 0; k < 4; ++k)
        if (b0 + k < n_bytes) v |= static_cast<uint32_t>(bytes[b0 + k]) << (24 - 8 * k);
    return v;
}

// The K bits from bit `bit` (first bit = most significant) of a lane's words, times 1 << SCALE.
template <int K, int SCALE, int NW>
__device__ __forceinline__ uint32_t fsm_field(const uint32_t (&W)[NW], int bit) {
    const int a = bit >> 5, w = bit & 31;
    constexpr uint32_t mask = ((1u << K) - 1u) << SCALE;
    uint32_t v;
    if (w + K <= 32) {
        const int sh = 32 - w - K - SCALE;  // right shift that leaves the field at bit SCALE
        v = sh >= 0 ? W[a] >> sh : W[a] << -sh;
    } else {
        v = __builtin_amdgcn_alignbit(W[a], W[a + 1 < NW ? a + 1 : a], 32 - w) >> (32 - K - SCALE);
    }
    return v & mask;
}

// ================================================================================================
// D1
// ================================================================================================
constexpr int SYNC_WORDS = 20;  // W[j] = stream word 16 * lane - 4 + j: 4 run-in words, 16 own
constexpr int SYNC_LANES = 2;   // 512-bit lanes walked by one thread: independent chains in flight

// Walk N_BITS bits from bit BIT0 of both lanes' words: R = row byte offset (row << (K + 1)), C +=
// symbols completed.  EDGE: only the steps below `limit` count (the stream ends inside the lane's
// words); SKIP: steps below `skip` do not count either (a walk that begins at a bit offset).
template <int K, int BIT0, int N_BITS, bool EDGE, bool SKIP>
__device__ __forceinline__ void fsm_sync_walk(const uint32_t (&W)[SYNC_LANES][SYNC_WORDS], uint32_t (&R)[SYNC_LANES], uint32_t (&C)[SYNC_LANES],
                                              const uint32_t (&skip)[SYNC_LANES], const uint32_t (&limit)[SYNC_LANES], int step0) {
    const uint32_t tab = fsm_lds_addr(fsm_smem);
#pragma unroll
    for (int j = 0; j < N_BITS / K; ++j) {
#pragma unroll
        for (int u = 0; u < SYNC_LANES; ++u) {
            const uint32_t f2 = fsm_field<K, 1, SYNC_WORDS>(W[u], BIT0 + j * K);
            const uint32_t e = *reinterpret_cast<const fsm_lds_u16 *>(static_cast<uintptr_t>(tab + (R[u] | f2)));
            bool on = true;
            if (EDGE) on = on && static_cast<uint32_t>(step0 + j) < limit[u];
            if (SKIP && j * K < 32) on = on && static_cast<uint32_t>(j) >= skip[u];
            const uint32_t next = (e << (K + 1)) & (FSM_ROW_MASK << (K + 1));
            uint32_t n = e >> FSM_SYNC_N_SHIFT;
            if ((EDGE) || (SKIP && j * K < 32)) {
                R[u] = on ? next : R[u];
                n = on ? n : 0u;
            } else {
                R[u] = next;
            }
            // (added here and now: left to itself the compiler keeps every entry of the walk alive and sums them at the end, out of scratch memory)
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(C[u]) : "v"(n));
        }
        // (nothing moves across a step: the compiler otherwise computes every field of the walk up front and spills)
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct SyncLane {
    uint32_t s_mid, s_out;  // rows after 256 bits, after 512
    uint32_t c1, c2;        // symbols completed in the first / second 256 bits
};

// Both lanes of a thread from row offsets R0 (first `skip` steps left out).  `take[u]`: lane u's
// results are wanted; REWALK: a lane that stands after 256 bits where its old walk stood keeps the rest.
template <int K, bool EDGE, bool REWALK>
__device__ __forceinline__ void fsm_sync_lanes(uint32_t (&W)[SYNC_LANES][SYNC_WORDS], const uint32_t (&R0)[SYNC_LANES], const uint32_t (&skip)[SYNC_LANES],
                                               const uint32_t (&limit)[SYNC_LANES], const bool (&take)[SYNC_LANES], SyncLane (&r)[SYNC_LANES]) {
    // (the words are "new" to every walk: the compiler otherwise keeps the first walk's fields for the re-walk, in scratch memory)
#pragma unroll
    for (int u = 0; u < SYNC_LANES; ++u)
#pragma unroll
        for (int j = 4; j < SYNC_WORDS; ++j) asm volatile("" : "+v"(W[u][j]));
    uint32_t R[SYNC_LANES], C[SYNC_LANES];
    bool redo[SYNC_LANES];
    bool any_redo = false;
#pragma unroll
    for (int u = 0; u < SYNC_LANES; ++u) {
        R[u] = R0[u];
        C[u] = 0;
    }
    fsm_sync_walk<K, 128, 256, EDGE, true>(W, R, C, skip, limit, 0);
#pragma unroll
    for (int u = 0; u < SYNC_LANES; ++u) {
        if (EDGE && limit[u] < 256 / K) R[u] = 0;  // the stream ended: nothing is pending
        const uint32_t mid = R[u] >> (K + 1);
        redo[u] = take[u] && !(REWALK && mid == r[u].s_mid);
        if (take[u]) {
            r[u].c1 = C[u];
            r[u].s_mid = mid;
        }
        C[u] = 0;
        any_redo = any_redo || redo[u];
    }
    if (__any(any_redo)) {
        fsm_sync_walk<K, 384, 256, EDGE, false>(W, R, C, skip, limit, 256 / K);
#pragma unroll
        for (int u = 0; u < SYNC_LANES; ++u) {
            if (EDGE && limit[u] < 512 / K) R[u] = 0;
            if (redo[u]) {
                r[u].c2 = C[u];
                r[u].s_out = R[u] >> (K + 1);
            }
        }
    }
}

// A wavefront owns one 8 KiB block: 128 lanes of 512 bits, thread t walks lanes t and 64 + t of it
// (two chains in flight per thread).  After the table is staged nothing is shared between wavefronts:
// seams are agreed on by shuffles, so there is no barrier and no LDS traffic besides the lookups.
// sub_state[s] (s = 256-bit subsequence): start row | exit row << 11 | symbols that END in s << 22.
// blk_exit[b] = row at the end of block b, blk_count[b] = its symbols.  A block's FIRST lane keeps
// what its run-in gave it (the stream's first lane: the root at first_bit); the check / repair sweep
// after this kernel compares it with the block before.
// flags: DEC_HAVE_START (first_bit is the exact first bit of the range, at a codeword boundary),
// DEC_FRONT_OK (the 16 bytes before `words` are stream bytes).
template <int K>
__global__ __launch_bounds__(512, 4) void k_fsm_sync(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs,
                                                     uint32_t n_blocks, const uint16_t *__restrict__ table, uint32_t table_entries, uint32_t n_int,
                                                     uint32_t *__restrict__ sub_state, uint32_t *__restrict__ blk_exit, uint32_t *__restrict__ blk_count,
                                                     uint32_t *__restrict__ changed, uint32_t max_trips, uint32_t flags) {
    uint16_t *tab = reinterpret_cast<uint16_t *>(fsm_smem);
    const uint32_t T = blockDim.x, tid = threadIdx.x, lane_id = tid & 63;
    for (uint32_t i = tid * 8; i < table_entries; i += T * 8) *reinterpret_cast<uint4 *>(tab + i) = *reinterpret_cast<const uint4 *>(table + i);
    __syncthreads();
    const bool have_start = flags & DEC_HAVE_START, front_ok = flags & DEC_FRONT_OK;
    const uint64_t n_lanes = (n_subs + 1) / 2;
    const uint64_t n_words_full = n_bytes / 4;
    const uint32_t waves_per_group = T >> 6;
    for (uint32_t b = blockIdx.x * waves_per_group + (tid >> 6); b < n_blocks; b += gridDim.x * waves_per_group) {
        // wavefront-uniform: every word of the block (and its run-in) is a whole word of the stream
        const long long bw0 = static_cast<long long>(b) * 2048 - 4;
        const bool edge = (bw0 < 0 && !front_ok) || static_cast<uint64_t>(bw0 + 2048 + 4) > n_words_full;
        uint32_t W[SYNC_LANES][SYNC_WORDS], limit[SYNC_LANES];
        uint64_t q[SYNC_LANES];
        bool live[SYNC_LANES];
#pragma unroll
        for (int u = 0; u < SYNC_LANES; ++u) {
            q[u] = static_cast<uint64_t>(b) * 128 + u * 64 + lane_id;  // 512-bit lane = subsequences 2q, 2q + 1
            live[u] = q[u] < n_lanes;
            limit[u] = 0xffffffffu;
            const long long w0 = static_cast<long long>(q[u]) * 16 - 4;
            if (!edge) {
#pragma unroll
                for (int j = 0; j < SYNC_WORDS; ++j) W[u][j] = __builtin_bswap32(words[w0 + j]);
            } else {
#pragma unroll
                for (int j = 0; j < SYNC_WORDS; ++j) W[u][j] = fsm_load_guarded(words, w0 + j, n_bytes, front_ok);
                // whole bytes of the stream inside the lane's own 512 bits -> steps that count
                const uint64_t lane_byte0 = q[u] * 64;
                const uint64_t nb = n_bytes > lane_byte0 ? (n_bytes - lane_byte0 < 64 ? n_bytes - lane_byte0 : 64) : 0;
                limit[u] = static_cast<uint32_t>(nb) * (8 / K);
            }
        }
        // run-in: from the root, 128 bits before the lane's own
        uint32_t R0[SYNC_LANES] = {}, skip[SYNC_LANES] = {}, C0[SYNC_LANES] = {};
        const uint32_t no_limit[SYNC_LANES] = {0xffffffffu, 0xffffffffu};
        fsm_sync_walk<K, 0, 128, false, false>(W, R0, C0, skip, no_limit, 0);
        // the range's first lane with a known start begins at the root at bit first_bit instead
        const bool known = q[0] == 0 && have_start;
        if (known) {
            skip[0] = first_bit / K;
            const uint32_t rem = first_bit % K;
            R0[0] = rem ? (n_int + rem - 1) << (K + 1) : 0u;
        }
        uint32_t start[SYNC_LANES];
        SyncLane r[SYNC_LANES] = {};
        bool take[SYNC_LANES];
#pragma unroll
        for (int u = 0; u < SYNC_LANES; ++u) {
            start[u] = (u == 0 && known) ? 0u : R0[u] >> (K + 1);
            take[u] = true;
        }
        if (edge) fsm_sync_lanes<K, true, false>(W, R0, skip, limit, take, r);
        else fsm_sync_lanes<K, false, false>(W, R0, skip, limit, take, r);
        skip[0] = 0;
        bool gave_up = false;
        for (uint32_t trip = 1;; ++trip) {
            // lane li's start must be lane li - 1's exit (li = u * 64 + lane); the block's first lane keeps its own
            bool need[SYNC_LANES], any_need = false;
            uint32_t cand[SYNC_LANES];
            const uint32_t up0 = __shfl_up(r[0].s_out, 1), up1 = __shfl_up(r[1].s_out, 1), last0 = __shfl(r[0].s_out, 63);
            cand[0] = lane_id ? up0 : start[0];
            cand[1] = lane_id ? up1 : last0;
#pragma unroll
            for (int u = 0; u < SYNC_LANES; ++u) {
                need[u] = live[u] && cand[u] != start[u];
                any_need = any_need || need[u];
            }
            if (!__any(any_need)) break;
            if (trip == max_trips) {  // a code that does not self-synchronise: the block is marked for a redo
                gave_up = true;
                break;
            }
            uint32_t Rn[SYNC_LANES];
#pragma unroll
            for (int u = 0; u < SYNC_LANES; ++u) Rn[u] = need[u] ? cand[u] << (K + 1) : 0u;
            if (edge) fsm_sync_lanes<K, true, true>(W, Rn, skip, limit, need, r);
            else fsm_sync_lanes<K, false, true>(W, Rn, skip, limit, need, r);
#pragma unroll
            for (int u = 0; u < SYNC_LANES; ++u) start[u] = need[u] ? cand[u] : start[u];
        }
        if (gave_up) {
            if (lane_id == 0) {
                atomicAdd(changed + 1, 1u);
                start[0] = FSM_ROW_MASK;  // no row: any later sweep redoes the block
            }
        }
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < SYNC_LANES; ++u) {
            if (live[u]) {
                sub_state[2 * q[u]] = start[u] | (r[u].s_mid << 11) | (r[u].c1 << 22);
                if (2 * q[u] + 1 < n_subs) sub_state[2 * q[u] + 1] = r[u].s_mid | (r[u].s_out << 11) | (r[u].c2 << 22);
                sum += r[u].c1 + r[u].c2;
            }
        }
        sum = fsm_wave_inclusive_scan(sum);
        if (lane_id == 63) {
            blk_count[b] = sum;
            blk_exit[b] = r[1].s_out;  // (lanes past the stream's end stand at the root)
        }
    }
}

// ================================================================================================
// D3
// ================================================================================================
constexpr int WRITE_WORDS = 10;            // 256 own bits + 32 + the last field's overhang
constexpr uint32_t FSM_STAGE_SLACK = 768;  // 64 dump words, then room for the unclipped walk of lanes that sit a pass out (K = 2: <= 510 bytes)

struct WriteAcc {
    uint32_t lo;    // bytes gathered for the dword at `dw`
    uint32_t fill;  // bits of `lo` in use (8 per symbol; the first dword starts part-filled), < 32 between steps
    uint32_t dw;    // LDS address of the dword being gathered
    uint32_t dump;  // LDS address of a word of the lane's own where a step that completes no dword stores to
};

// The entry's symbols join the gathered bytes; a full dword leaves for the stage.  EVERY step stores
// (a step that fills no dword: to the lane's dump word): straight-line code, no branch, and the
// compiler knows that exactly one store follows the read that was issued ahead of it -- the wait for
// that read then does not wait for the store as well.  A lane's LAST dword is the next lane's
// first: it never fills by the lane's own symbols (the walk's last steps are clipped to the lane's
// count) and is patched in byte by byte after the walk.
__device__ __forceinline__ void fsm_write_put(uint32_t e, WriteAcc &a) {
    const unsigned long long put = static_cast<unsigned long long>(e >> 16) << a.fill;
    a.lo |= static_cast<uint32_t>(put);
    a.fill += e & FSM_WRITE_N8_MASK;
    const bool full = a.fill >= 32;
    *reinterpret_cast<fsm_lds_u32 *>(static_cast<uintptr_t>(full ? a.dw : a.dump)) = a.lo;
    a.lo = full ? static_cast<uint32_t>(put >> 32) : a.lo;
    a.dw += (a.fill >> 5) << 2;
    a.fill &= 31u;
}
// The same for a step near the walk's end: no more symbols than the lane still has to emit
// (`end_bits` = 8 x the byte position, relative to the lane's first dword, at which its symbols end).
__device__ __forceinline__ void fsm_write_put_clipped(uint32_t e, WriteAcc &a, uint32_t dw0, uint32_t end_bits) {
    const uint32_t at = (a.dw - dw0) * 8u + a.fill;       // bits emitted so far, same origin
    const uint32_t left = end_bits > at ? end_bits - at : 0u;  // 8 x symbols still to come
    const uint32_t n8 = e & FSM_WRITE_N8_MASK;
    uint32_t use = e;
    if (n8 > left) use = left ? (e & 0x00ff0000u) | 8u : 0u;  // two on offer, one wanted: the first; none wanted: nothing
    fsm_write_put(use, a);
}

// A wavefront owns half an 8 KiB block: 128 lanes of 256 bits, thread t walks lanes t and 64 + t of it
// (two chains in flight per thread: the LDS round trip of one hides behind the other's arithmetic);
// its symbols are staged in its OWN slice of the LDS and leave as 16-byte stores -- no barrier,
// nothing shared but the table.
// offset_mode 0: sub_state = start row | .. | symbols << 22, a lane emits the symbols that END in
//                its 256 bits (k_fsm_sync's state);
// offset_mode 1: sub_state = start offset | exit << 8 | symbols << 16, a lane begins at that bit
//                offset and emits the symbols that BEGIN in its 256 bits (the exhaustive path's state).
// window: symbols staged per pass (a multiple of 16); a unit with more takes several passes, each
// over the lanes that begin inside its window.
constexpr int WRITE_LANES = 2;
template <int K>
__global__ __launch_bounds__(1024) void k_fsm_write(const uint32_t *__restrict__ words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs,
                                                    uint32_t n_blocks, const uint32_t *__restrict__ table, uint32_t table_words, uint32_t n_int,
                                                    const uint32_t *__restrict__ sub_state, const unsigned long long *__restrict__ blk_off,
                                                    uint64_t n_symbols, uint8_t *__restrict__ out, uint32_t window, uint32_t wave_stage_bytes,
                                                    uint32_t offset_mode, uint32_t have_start, const uint32_t *__restrict__ void_flags) {
    if (void_flags && !dec_state_final(void_flags[1], void_flags[2], n_blocks)) return;  // a speculative launch on a state that is not final
    constexpr int FIELDS = (256 + K - 1) / K, SKIPS = (32 + K - 1) / K;
    constexpr int CLIP_FROM = FIELDS - 1;  // fields whose bits reach past the lane's 256: clipped to its count
    constexpr uint32_t ROW_SHIFT = fsm_write_row_shift(K), ROW_MASK = fsm_write_row_mask(K);
    // LDS: table | per wavefront: stage (16 + window + one lane's most + slack)
    uint32_t *tab = reinterpret_cast<uint32_t *>(fsm_smem);
    const uint32_t T = blockDim.x, tid = threadIdx.x, lane_id = tid & 63, wave = tid >> 6;
    uint8_t *stage = fsm_smem + table_words * 4 + wave * wave_stage_bytes;
    const uint32_t lds_tab = fsm_lds_addr(tab), lds_stage = fsm_lds_addr(stage);
    const uint32_t lds_dump = lds_stage + wave_stage_bytes - FSM_STAGE_SLACK;  // [lane] dump words, then where lanes that sit a pass out put their walk's bytes
    for (uint32_t i = tid * 4; i < table_words; i += T * 4) *reinterpret_cast<uint4 *>(tab + i) = *reinterpret_cast<const uint4 *>(table + i);
    __syncthreads();
    const uint64_t n_words_full = n_bytes / 4;
    const uint32_t n_units = 2 * n_blocks, waves_per_group = T >> 6;
    for (uint32_t unit = blockIdx.x * waves_per_group + wave; unit < n_units; unit += gridDim.x * waves_per_group) {
        uint64_t sub[WRITE_LANES];
        uint32_t st[WRITE_LANES], count[WRITE_LANES], my_off[WRITE_LANES];
#pragma unroll
        for (int u = 0; u < WRITE_LANES; ++u) {
            sub[u] = static_cast<uint64_t>(unit) * 128 + u * 64 + lane_id;
            st[u] = sub[u] < n_subs ? sub_state[sub[u]] : 0u;
            count[u] = offset_mode ? st[u] >> 16 : st[u] >> 22;
        }
        // where the unit's symbols go: the block's offset, plus (second half) what the first half emits
        uint64_t o0 = blk_off[unit >> 1];
        if (unit & 1) {
            uint32_t before = 0;
#pragma unroll
            for (int u = 0; u < WRITE_LANES; ++u) {
                const uint64_t sp = sub[u] - 128;
                const uint32_t a = sp < n_subs ? sub_state[sp] : 0u;
                before += offset_mode ? a >> 16 : a >> 22;
            }
            o0 += __shfl(fsm_wave_inclusive_scan(before), 63);
        }
        if (o0 >= n_symbols) continue;  // pad bits decoded past the declared length
        const uint32_t inc0 = fsm_wave_inclusive_scan(count[0]), total0 = __shfl(inc0, 63);
        const uint32_t inc1 = fsm_wave_inclusive_scan(count[1]);
        my_off[0] = inc0 - count[0];
        my_off[1] = total0 + inc1 - count[1];
        const uint32_t total = total0 + __shfl(inc1, 63);
        const bool edge = static_cast<uint64_t>(static_cast<long long>(unit + 1) * 1024 + 2) > n_words_full;
        uint32_t W[WRITE_LANES][WRITE_WORDS], R0[WRITE_LANES], skip[WRITE_LANES];
#pragma unroll
        for (int u = 0; u < WRITE_LANES; ++u) {
            const long long w0 = static_cast<long long>(sub[u]) * 8;
            if (!edge) {
#pragma unroll
                for (int j = 0; j < WRITE_WORDS; ++j) W[u][j] = __builtin_bswap32(words[w0 + j]);
            } else {
#pragma unroll
                for (int j = 0; j < WRITE_WORDS; ++j) W[u][j] = fsm_load_guarded(words, w0 + j, n_bytes, false);
            }
            // where the lane's walk begins: R = LDS address of its row
            R0[u] = lds_tab;
            skip[u] = 0;
            if (offset_mode || (sub[u] == 0 && have_start)) {
                const uint32_t bit = (sub[u] == 0 && have_start) ? first_bit : (st[u] & 31u);
                skip[u] = bit / K;
                const uint32_t rem = bit % K;
                if (rem) R0[u] += (n_int + rem - 1) << ROW_SHIFT;
            } else {
                const uint32_t row = st[u] & FSM_ROW_MASK;
                if (row < n_int) R0[u] += row << ROW_SHIFT;  // (nothing but a row of the table reaches the walk)
            }
        }
        uint64_t o1 = o0 + total;
        if (o1 > n_symbols) o1 = n_symbols;
        const uint32_t n_out = static_cast<uint32_t>(o1 - o0);
        for (uint32_t lo = 0; lo < n_out;) {
            // this pass: the lanes that begin in [lo, lo + window); their symbols are [lo, hi)
            const uint32_t phase = static_cast<uint32_t>((o0 + lo) & 15);  // stage offset of the pass's first symbol
            uint32_t hi = 0, a0[WRITE_LANES], a1[WRITE_LANES], dw0[WRITE_LANES], end_bits[WRITE_LANES], R[WRITE_LANES], e[WRITE_LANES];
            bool mine[WRITE_LANES];
            WriteAcc acc[WRITE_LANES];
#pragma unroll
            for (int u = 0; u < WRITE_LANES; ++u) {
                mine[u] = count[u] && my_off[u] >= lo && my_off[u] < lo + window;
                hi = max(hi, mine[u] ? my_off[u] + count[u] : 0u);
                a0[u] = mine[u] ? lds_stage + phase + (my_off[u] - lo) : lds_dump + 256;
                a1[u] = a0[u] + (mine[u] ? count[u] : 0u);
                dw0[u] = a0[u] & ~3u;
                end_bits[u] = (a1[u] - dw0[u]) * 8u;
                acc[u] = WriteAcc{0u, (a0[u] & 3u) * 8u, dw0[u], lds_dump + lane_id * 4};
                R[u] = R0[u];
                asm volatile("" : "+v"(R[u]));  // (per pass, per lane: nothing of one walk is kept for another)
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) hi = max(hi, static_cast<uint32_t>(__shfl_xor(hi, d)));
#ifndef FSM_PROBE_NO_WALK
            // e = the entry of field j, read one step ahead of its use
#pragma unroll
            for (int u = 0; u < WRITE_LANES; ++u) e[u] = *reinterpret_cast<const fsm_lds_u32 *>(static_cast<uintptr_t>(R[u] + fsm_field<K, 2, WRITE_WORDS>(W[u], 0)));
#pragma unroll
            for (int j = 0; j < FIELDS; ++j) {
#pragma unroll
                for (int u = 0; u < WRITE_LANES; ++u) {
                    uint32_t use = e[u];
                    if (j < SKIPS) {  // fields before the walk's first bit: stay, put nothing
                        const bool on = static_cast<uint32_t>(j) >= skip[u];
                        use = on ? e[u] : 0u;
                        R[u] = on ? lds_tab + (e[u] & ROW_MASK) : R[u];
                    } else {
                        R[u] = lds_tab + (e[u] & ROW_MASK);
                    }
                    if (j + 1 < FIELDS) e[u] = *reinterpret_cast<const fsm_lds_u32 *>(static_cast<uintptr_t>(R[u] + fsm_field<K, 2, WRITE_WORDS>(W[u], (j + 1) * K)));
                    if (j >= CLIP_FROM) fsm_write_put_clipped(use, acc[u], dw0[u], end_bits[u]);
                    else fsm_write_put(use, acc[u]);
                }
                __builtin_amdgcn_sched_barrier(0);  // (a step's fields are computed in the step, not up front in 2 x FIELDS registers)
            }
            if (offset_mode) {  // the symbol that begins in these 256 bits and ends after them
#pragma unroll
                for (int j = FIELDS; j < FIELDS + SKIPS; ++j) {
#pragma unroll
                    for (int u = 0; u < WRITE_LANES; ++u) {
                        e[u] = *reinterpret_cast<const fsm_lds_u32 *>(static_cast<uintptr_t>(R[u] + fsm_field<K, 2, WRITE_WORDS>(W[u], j * K)));
                        R[u] = lds_tab + (e[u] & ROW_MASK);
                        fsm_write_put_clipped(e[u], acc[u], dw0[u], end_bits[u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#endif
            // the bytes of a lane's last dword that are its own (its first dword too, if it is the same one);
            // a wavefront's LDS operations execute in order, so these follow every lane's dword stores
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < WRITE_LANES; ++u) {
                if (mine[u] && acc[u].dw == (a1[u] & ~3u)) {
                    const uint32_t lo_i = acc[u].dw == dw0[u] ? (a0[u] & 3u) : 0u, hi_i = a1[u] & 3u;
#pragma unroll
                    for (uint32_t i = 0; i < 3; ++i)
                        if (i >= lo_i && i < hi_i) *reinterpret_cast<fsm_lds_u8 *>(static_cast<uintptr_t>(acc[u].dw + i)) = static_cast<uint8_t>(acc[u].lo >> (8 * i));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const uint32_t hi_c = hi < n_out ? hi : n_out;  // (clamped to the declared length)
            const uint32_t end = phase + (hi_c - lo);
            uint8_t *out_base = out + (o0 + lo - phase);
#ifndef FSM_PROBE_NO_COPY
            for (uint32_t p = lane_id * 16; p < end; p += 64 * 16) {
                if (p >= phase && p + 16 <= end) {
                    *reinterpret_cast<uint4 *>(out_base + p) = *reinterpret_cast<const uint4 *>(stage + p);
                } else {
                    for (uint32_t k = p > phase ? p : phase; k < (p + 16 < end ? p + 16 : end); ++k) out_base[k] = stage[k];
                }
            }
#else
            if (end == 12345u) out_base[0] = stage[0];
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is read before the next pass writes it
            __builtin_amdgcn_wave_barrier();
            lo = hi > lo ? hi : n_out;  // (hi == lo cannot happen: the lane at `lo` has symbols)
        }
    }
}

// ================================================================================================
// launch wrappers
// ================================================================================================
size_t fsm_sync_smem(uint32_t table_entries) { return (static_cast<size_t>(table_entries) * 2 + 15) & ~static_cast<size_t>(15); }
uint32_t fsm_write_table_words(uint32_t rows, uint32_t k) { return ((rows << (fsm_write_row_shift(k) - 2)) + 3u) & ~3u; }
// One wavefront's slice of the write kernel's stage: 16 (phase) + window + one lane's most + slack.
uint32_t fsm_write_wave_stage(uint32_t window, uint32_t max_per_lane) { return (16 + window + max_per_lane + FSM_STAGE_SLACK + 15) & ~15u; }

#define ET_FSM_LAUNCH(kernel_, grid_, block_, smem_, stream_, evs_, ...)                                                          \
    do {                                                                                                                          \
        if ((evs_).start || (evs_).stop) hipExtLaunchKernelGGL(kernel_, grid_, block_, smem_, stream_, (evs_).start, (evs_).stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel_, grid_, block_, smem_, stream_, __VA_ARGS__);                                             \
    } while (0)

static uint32_t fsm_cus() {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    static thread_local int seen_dev = -1, seen_cus = 256;
    if (dev != seen_dev) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        seen_dev = dev;
        seen_cus = cus;
    }
    return static_cast<uint32_t>(seen_cus);
}

void launch_fsm_sync(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, const FsmTables &ft,
                     uint32_t *sub_state, uint32_t *blk_exit, uint32_t *blk_count, uint32_t *changed, uint32_t max_trips, uint32_t flags,
                     KernelEvents ev) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    const uint32_t entries = ((ft.rows_sync << ft.k_sync) + 7u) & ~7u;
    const size_t smem = fsm_sync_smem(entries);
    // workgroups resident per CU: by LDS (the table), by registers (4 wavefronts per SIMD = 16 per CU)
    uint32_t threads = ft.sync_threads ? ft.sync_threads : 512;
    uint32_t per_cu = static_cast<uint32_t>((160u * 1024u) / smem);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 1536 / threads) per_cu = 1536 / threads;  // (72 VGPRs: 6 wavefronts per SIMD)
    const uint32_t waves = threads / 64;
    uint32_t grid = fsm_cus() * per_cu;
    if (grid > (n_blocks + waves - 1) / waves) grid = (n_blocks + waves - 1) / waves;
#define ET_FSM_SYNC_ARGS words, n_bytes, first_bit, n_subs, n_blocks, ft.sync, entries, ft.n_int, sub_state, blk_exit, blk_count, changed, max_trips, flags
    if (ft.k_sync == 8) ET_FSM_LAUNCH(k_fsm_sync<8>, dim3(grid), dim3(threads), smem, stream, ev, ET_FSM_SYNC_ARGS);
    else if (ft.k_sync == 4) ET_FSM_LAUNCH(k_fsm_sync<4>, dim3(grid), dim3(threads), smem, stream, ev, ET_FSM_SYNC_ARGS);
    else ET_FSM_LAUNCH(k_fsm_sync<2>, dim3(grid), dim3(threads), smem, stream, ev, ET_FSM_SYNC_ARGS);
#undef ET_FSM_SYNC_ARGS
}

void launch_fsm_write(hipStream_t stream, const uint32_t *words, uint64_t n_bytes, uint32_t first_bit, uint64_t n_subs, const FsmTables &ft,
                      const uint32_t *sub_state, const unsigned long long *blk_off, uint64_t n_symbols, uint8_t *out, bool offset_mode,
                      bool have_start, const uint32_t *void_flags, KernelEvents ev) {
    const uint32_t n_blocks = static_cast<uint32_t>((n_subs + BLOCK - 1) / BLOCK);
    const uint32_t table_words = fsm_write_table_words(ft.rows_write, ft.k_write);
    // the window: a little over the stream's mean symbols per wavefront (128 lanes of 256 bits) -- units
    // above it take a second pass --, at most what every lane could emit at once
    uint64_t mean = n_subs ? (n_symbols * 128 + n_subs - 1) / n_subs : 0;
    uint32_t window = static_cast<uint32_t>(((mean + mean / 32 + 160) + 15) & ~static_cast<uint64_t>(15));
    const uint32_t most = (128 * ft.max_per_lane + 15) & ~15u;
    if (ft.write_window) window = ft.write_window;
    if (window > most) window = most;
    if (window < 256) window = 256;
    const uint32_t wave_stage = fsm_write_wave_stage(window, ft.max_per_lane);
    // wavefronts per CU: by LDS
    const uint32_t lds_left = 160u * 1024u - table_words * 4;
    uint32_t waves_cu = lds_left / wave_stage;
    if (waves_cu > 32) waves_cu = 32;
    if (waves_cu < 1) waves_cu = 1;
    // one workgroup per CU of up to 16 wavefronts, or two of half as many (so that one table copy serves them)
    uint32_t waves = waves_cu > 16 ? 16 : waves_cu, per_cu = 1;
    if (ft.write_threads) waves = ft.write_threads / 64;
    const size_t smem = static_cast<size_t>(table_words) * 4 + static_cast<size_t>(waves) * wave_stage;
    per_cu = static_cast<uint32_t>((160u * 1024u) / smem);
    if (per_cu < 1) per_cu = 1;
    if (per_cu * waves > 32) per_cu = 32 / waves;
    uint32_t grid = fsm_cus() * per_cu;
    const uint32_t n_units = 2 * n_blocks;
    if (grid > (n_units + waves - 1) / waves) grid = (n_units + waves - 1) / waves;
#define ET_FSM_WRITE_ARGS words, n_bytes, first_bit, n_subs, n_blocks, ft.write, table_words, ft.n_int, sub_state, blk_off, n_symbols, out, window, wave_stage, offset_mode ? 1u : 0u, have_start ? 1u : 0u, void_flags
    if (ft.k_write == 6) ET_FSM_LAUNCH(k_fsm_write<6>, dim3(grid), dim3(waves * 64), smem, stream, ev, ET_FSM_WRITE_ARGS);
    else if (ft.k_write == 4) ET_FSM_LAUNCH(k_fsm_write<4>, dim3(grid), dim3(waves * 64), smem, stream, ev, ET_FSM_WRITE_ARGS);
    else ET_FSM_LAUNCH(k_fsm_write<2>, dim3(grid), dim3(waves * 64), smem, stream, ev, ET_FSM_WRITE_ARGS);
#undef ET_FSM_WRITE_ARGS
}

}  // namespace et

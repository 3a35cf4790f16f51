// et_api.cpp -- the extern "C" boundary of libentreepy_hip.so (include/entreepy_hip.h):
// context, workspaces, and the orchestration of the kernels in et_kernels.hip, et_treewalk.hip, et_rowsync.hip and et_kernels_fallback.hip.
//
// Encode (replaces encode.zig:25-337):
//   K1 histogram (totals stored into pinned host memory, polled) -> host code construction (et_codebook.cpp) -> K2 tile bit totals
//   (its first workgroup takes the code table and the header out of the pinned block) + scan -> K4 code scatter.
// Decode (replaces decode.zig:13-220):
//   header to pinned host memory (polled) -> host parse, the code as a tree + the chained tables' plan -> k_tw_build -> D1
//   synchronisation by tree walk (one launch; repair sweeps only if the verification fails) -> D2 scan of the blocks' symbol counts
//   (+ verification, report to the host) -> D3 write over chained tables.  Complete codes of 7- and 8-bit codewords (uniform-like
//   bytes): k_row_sync -> D2 -> k_row_write (et_rowsync.h); fixed-length codes (2^L codewords of L bits): k_fixed_write alone.
//   Anything outside those domains: et_kernels_fallback.hip.
// There is no CPU fallback anywhere in this file: without a usable HIP device every
// entry point returns ET_ERR_HIP.
#include "entreepy_hip.h"
#include <sys/stat.h>

#include "et_io.h"
#include "et_kernels.h"
#include "et_rowsync.h"
#include "et_tables.h"
#include "et_treewalk.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

constexpr size_t HEADER_STAGE = 8192;  // >= 4631-byte worst-case header, padded
constexpr size_t SUB_TABLE_ONLY = (static_cast<size_t>(et::DEC_SUB_TABLES_MAX) << et::DEC_SUB_BITS_MAX) * sizeof(uint16_t) + 64;
constexpr size_t SUB_TABLE_BYTES = SUB_TABLE_ONLY + 256;
constexpr size_t DEC_STEPS_OFFSET = (sizeof(uint32_t) << et::DEC_LUT_BITS_MAX) * 2 + 1024 * sizeof(uint32_t) + 2 * SUB_TABLE_BYTES;  // multiple of 64
constexpr size_t DEC_TABLES_BYTES = DEC_STEPS_OFFSET + (sizeof(uint32_t) << et::DEC_STEP_BITS_MAX) + (sizeof(uint32_t) << et::DEC_LUT_BITS_MAX) +
                                    2 * (et::DEC_STEP_SUB_WORDS + 4) * sizeof(uint32_t) + 2 * sizeof(et::DecodeTables) + sizeof(et::TablePlan) + 64;
//  // the per-symbol code lengths ride behind the tables  // + slack for 16-byte rounded copies

}  // namespace

struct et_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    et::SideLane side = {};  // second lane for the first/last-block launches of a decode
    hipStream_t stream = nullptr;
    bool timing = false;       // every phase carries events (et_ctx_enable_timing(ctx, 1))
    bool timing_body = false;  // only the decode's write kernel does (et_ctx_enable_timing(ctx, ET_TIMING_DECODE_BODY))
    uint32_t force_rpt = 0;
    uint32_t lut_bits_write = et::DEC_LUT_BITS_WRITE;
    uint32_t step_bits = et::DEC_STEP_BITS_DEFAULT;
    std::string err;

    // encode workspaces
    DevBuf tile_hist, block_hist, hist, tile_bits, tile_off, enc_table, group_sum;
    // decode workspaces
    DevBuf sub_state, blk_exit, blk_count, blk_off, lut, flag, worklist;  // flag: [0..3] sweep flags, [4] ticket, [8..] worklist counts  // lut: all decode tables, DEC_TABLES_BYTES
    DevBuf lane_maps, blk_maps, grp_maps, blk_in, grp_in;  // exhaustive synchronisation only
    DevBuf row_scratch;                                    // the row walk's published words and ticket (et_rowsync.h)
    DevBuf tw_table, tw_tree, blk_start, blk_pub, chain_table;  // tree-walk synchronisation, chained write tables (et_treewalk.h)
    et::TwUpload *h_tw_tree[2] = {};                       // pinned, used in turn like h_lut_buf
    int tw_turn = 0;
    // staging for the host-pointer / file-descriptor entry points
    DevBuf io_in, io_out;
    et_io::Pipe *io = nullptr;  // pinned double buffer + copy threads, made on first use

    // pinned host staging
    uint64_t *h_hist = nullptr;     // 256
    uint32_t *h_enc = nullptr;      // 768 words: {code,len} x 256, then len x 256; HEADER_STAGE bytes: the file header on its way to the image
    uint8_t *h_header = nullptr;    // HEADER_STAGE
    uint32_t *h_lut = nullptr;      // the decode tables being built (one of h_lut_buf)
    uint32_t *h_lut_buf[2] = {};    // DEC_TABLES_BYTES each, used in turn: the host fills one while the other's upload may still be queued
    int lut_turn = 0;
    uint64_t *h_scalar = nullptr;   // 16: [1] a total, [2..3] flags (range decode), [4..11] the body decode's copy of flag[0..15], [12] / [14] "taken" / "done" words the device stores (enc_block_epoch, header_epoch)

    // link between et_histogram_device and et_encode_body_device
    const void *hist_text = nullptr;
    size_t hist_n = 0;
    uint32_t hist_rpt = 0, hist_tiles = 0;
    bool hist_on_host = false;  // h_hist holds the counts of hist_text
    bool hist_empty = false;    // the last et_histogram_device was of an empty text (zeros everywhere, no tiles)
    const void *scan_buf = nullptr;  // the group_sum buffer scan_epoch_n counts on
    size_t scan_cap = 0;
    uint32_t scan_epoch_n = 0;
    uint32_t report_epoch = 0;     // h_scalar word (4 * 2 + 14) == report_epoch: the current decode's flags and total are in h_flags
    uint64_t enc_block_epoch = 0;  // h_scalar[12] == enc_block_epoch: the device has taken its copy of h_enc
    uint64_t header_epoch = 0;  // h_scalar[14] == header_epoch: the header bytes of the current decode are in h_header
    uint64_t hist_epoch = 0;    // h_hist[256 + w] == hist_epoch: reducing workgroup w of the current histogram has stored its totals

    hipEvent_t ev[12] = {};  // 0..5: encode calls, EV_DEC + 0..5: decode calls
    et_timings tm_enc = {}, tm_dec = {};
    // A full encode / body decode with timing on leaves its event arithmetic for the first
    // et_last_timings[_of] call (which waits for the call's last event): the call itself
    // then returns as asynchronously as it does with timing off.
    bool pend_enc = false, pend_dec = false, pend_enc_bits = false, pend_enc_shard = false, pend_dec_first = false;
    int last_kind = 0;  // 0 encode, 1 decode
    et_codebook last_cb = {};
    bool have_cb = false;

    // et_decode_range_sync -> et_decode_range_write
    struct {
        bool valid = false;
        const uint32_t *words = nullptr;
        uint64_t n_bytes = 0, n_subs = 0, total = 0;
        uint32_t n_blocks = 0, flags = 0;
        et::DecodeTables tb = {}, tb_write = {};
        bool tw = false;  // synchronised by tree walk: the write goes over the chained tables (n_chain entries in ctx->chain_table)
        uint32_t n_chain = 0, max_len = 0;
        // et_decode_range_maps -> et_decode_range_resolve
        bool maps_valid = false, maps_const = false;
        uint32_t map_stride = 0;
        // a row code's range (et_rowsync.h): maps and resolve are two runs of k_row_sync, the write goes by rows
        bool row = false;
        et::RowCode row_code = {};
        et_codebook row_cb = {};
        uint32_t first_bit = 0;
    } range;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int fail(et_ctx *ctx, int status, const char *what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) {
            ctx->err += ": ";
            ctx->err += hipGetErrorString(e);
        }
    }
    return status;
}

#define ET_HIP(call)                                                     \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) return fail(ctx, ET_ERR_HIP, #call, e_);   \
    } while (0)

int ensure(et_ctx *ctx, DevBuf &b, size_t bytes) {
    if (b.cap >= bytes) return ET_OK;
    if (b.p) {
        ET_HIP(hipStreamSynchronize(ctx->stream));
        ET_HIP(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = (bytes + 4095) & ~static_cast<size_t>(4095);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, e == hipErrorOutOfMemory ? ET_ERR_NOMEM : ET_ERR_HIP, "hipMalloc", e);
    }
    b.cap = want;
    return ET_OK;
}

#define ET_TRY(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != ET_OK) return rc_; \
    } while (0)

// Tile geometry for a stream of `span` bytes measured from the aligned base.
struct Geometry {
    const uint8_t *base;
    uint64_t lo, hi;
    uint32_t rpt, n_tiles;
};

Geometry make_geometry(const et_ctx *ctx, const void *d_text, size_t n) {
    Geometry g;
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_text);
    g.base = reinterpret_cast<const uint8_t *>(a & ~static_cast<uintptr_t>(15));
    g.lo = a & 15;
    g.hi = g.lo + n;
    // Aim for >= 2048 tiles (one per resident K4 workgroup) before growing the tile towards 512 KiB: 1 GiB is 2048 tiles of
    // 512 KiB.  A tile costs K1 a flush of its 32 counter replicas between two barriers and K4 a shared seam word and a
    // restart of its ring; with 64 KiB tiles (round 2: 16384 of them for 1 GiB) that was ~2 % of a step (775-780 -> 792-797 GB/s),
    // and the tile scan reads 2 MiB of tile histograms instead of 16.
    uint32_t rpt = 1;
    while (rpt < et::MAX_ROUNDS_PER_TILE && g.hi / (static_cast<uint64_t>(rpt) * et::ROUND_BYTES) > 2048) rpt <<= 1;
    if (ctx && ctx->force_rpt) rpt = ctx->force_rpt;
    g.rpt = rpt;
    const uint64_t tile_bytes = static_cast<uint64_t>(rpt) * et::ROUND_BYTES;
    g.n_tiles = static_cast<uint32_t>((g.hi + tile_bytes - 1) / tile_bytes);
    return g;
}

int ensure_encode_ws(et_ctx *ctx, uint32_t n_tiles) {
    ET_TRY(ensure(ctx, ctx->tile_hist, static_cast<size_t>(n_tiles) * 256 * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->block_hist, static_cast<size_t>(et::MAX_GRID) * 256 * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->hist, 256 * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->tile_bits, (static_cast<size_t>(n_tiles) + 1) * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->tile_off, (static_cast<size_t>(n_tiles) + 1) * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->enc_table, 768 * sizeof(uint32_t) + HEADER_STAGE));  // {code,len} x 256, then len x 256, then the file header: one upload
    ET_TRY(ensure(ctx, ctx->group_sum, (static_cast<size_t>(n_tiles) / 1024 + 2) * sizeof(uint64_t)));
    return ET_OK;
}

// A fresh epoch for k_scan_fused's published words in ctx->group_sum (call after the buffer is ensured): 1 .. 65535
// within one lifetime of the zeroed buffer; a new buffer, or the counter running out, zeroes it.
uint32_t scan_epoch(et_ctx *ctx) {
    if (ctx->group_sum.p != ctx->scan_buf || ctx->group_sum.cap != ctx->scan_cap || ctx->scan_epoch_n >= 0xffffu) {
        (void)hipMemsetAsync(ctx->group_sum.p, 0, ctx->group_sum.cap, ctx->stream);
        ctx->scan_buf = ctx->group_sum.p;
        ctx->scan_cap = ctx->group_sum.cap;
        ctx->scan_epoch_n = 0;
    }
    return ++ctx->scan_epoch_n;
}

void record(et_ctx *ctx, int i) {
    if (ctx->timing) (void)hipEventRecord(ctx->ev[i], ctx->stream);
}

constexpr int EV_DEC = 6;

// Events a timed kernel launch carries itself (begin = ev[a], end = ev[b]); none when timing is off.
et::KernelEvents timed(et_ctx *ctx, int a, int b) {
    et::KernelEvents e;
    if (ctx->timing) {
        e.start = ctx->ev[a];
        e.stop = ctx->ev[b];
    }
    return e;
}

// The decode's write kernel: also when it alone is timed.
et::KernelEvents timed_body(et_ctx *ctx, int a, int b) {
    et::KernelEvents e;
    if (ctx->timing || ctx->timing_body) {
        e.start = ctx->ev[a];
        e.stop = ctx->ev[b];
    }
    return e;
}

float elapsed(et_ctx *ctx, int a, int b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) != hipSuccess) ms = 0.f;
    return ms;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Where a call needs an answer from the GPU before it can go on, a kernel stores the answer into pinned host memory and
// then `want` into *word, and the calling thread polls that word: no copy command, no completion signal, no wake-up
// (a stream wait returns ~10 us after the kernel).  After patience_ms without the word -- a stream blocked by somebody
// else's work, or a fault -- the stream wait takes over and reports.
template <typename T>
int wait_for_word(et_ctx *ctx, volatile const T *word, T want, double patience_ms, const char *what) {
    const double t0 = now_ms();
    for (uint32_t spin = 0; *word != want; ++spin)
        if ((spin & 1023u) == 1023u && now_ms() - t0 > patience_ms) {
            ET_HIP(hipStreamSynchronize(ctx->stream));
            if (*word != want) return fail(ctx, ET_ERR_HIP, what);
        }
    std::atomic_thread_fence(std::memory_order_acquire);
    return ET_OK;
}

int run_histogram(et_ctx *ctx, const void *d_text, size_t n, const Geometry &g, void *d_hist_also = nullptr) {
    ET_TRY(ensure_encode_ws(ctx, g.n_tiles));
    et::launch_hist(ctx->stream, g.base, g.lo, g.hi, g.rpt, g.n_tiles, static_cast<uint32_t *>(ctx->tile_hist.p),
                    static_cast<unsigned long long *>(ctx->block_hist.p), static_cast<unsigned long long *>(ctx->hist.p),
                    reinterpret_cast<unsigned long long *>(ctx->h_hist), ++ctx->hist_epoch, timed(ctx, 0, 1),  // (the totals land in h_hist too: fetch_histogram only waits)
                    static_cast<unsigned long long *>(d_hist_also));
    ET_HIP(hipGetLastError());
    ctx->hist_text = d_text;
    ctx->hist_empty = false;
    ctx->hist_n = n;
    ctx->hist_rpt = g.rpt;
    ctx->hist_tiles = g.n_tiles;
    ctx->hist_on_host = false;
    return ET_OK;
}

// Upload the code table and run K2 + K4 for the text whose tile histograms are in ctx.
int run_body(et_ctx *ctx, const et_codebook *cb, const Geometry &g, uint32_t *out32, uint64_t base_bit,
             const uint8_t *header, size_t header_len, int ev_scan, int ev_body) {
    const bool long_codes = cb->max_length > 32;
    // (the pinned block is read by the device itself, K2's first workgroup: not before that has happened for the call
    // before may it be filled again -- it says so in h_scalar[12]; normally long ago)
    ET_TRY(wait_for_word<uint64_t>(ctx, ctx->h_scalar + 12, ctx->enc_block_epoch, 100.0, "the code table block was never taken"));
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = cb->length[s];
        uint32_t code = cb->data[s];
        if (!long_codes) code = len ? (len == 32 ? code : (code & ((1u << len) - 1u)) << (32 - len)) : 0u;  // left-aligned
        ctx->h_enc[2 * s] = code;
        ctx->h_enc[2 * s + 1] = len;
        ctx->h_enc[512 + s] = len;
    }
    // The header rides behind the code table in ONE upload; the scan kernel copies it into the image
    // once the word holding the header/body seam is zeroed (no copy command between K2 and K4).
    const size_t padded = (header_len + 3) & ~static_cast<size_t>(3);
    if (header_len) {
        if (padded > HEADER_STAGE) return fail(ctx, ET_ERR_ARG, "header too long");
        uint8_t *stage = reinterpret_cast<uint8_t *>(ctx->h_enc + 768);
        std::memcpy(stage, header, header_len);
        std::memset(stage + header_len, 0, padded - header_len);
    }
    // (no upload: the code lengths ride in K2's kernel arguments, and its first workgroup copies the pinned block --
    // code table, lengths, header -- into enc_table for the kernels behind it)
    et::launch_tile_scan(ctx->stream, static_cast<const uint32_t *>(ctx->tile_hist.p), g.n_tiles, cb->length, ctx->h_enc, static_cast<uint32_t *>(ctx->enc_table.p),
                         static_cast<uint32_t>(768 + padded / 4), reinterpret_cast<unsigned long long *>(ctx->h_scalar + 12), ++ctx->enc_block_epoch,
                         static_cast<unsigned long long *>(ctx->tile_bits.p),
                         static_cast<unsigned long long *>(ctx->group_sum.p), scan_epoch(ctx), base_bit,
                         static_cast<unsigned long long *>(ctx->tile_off.p), out32, static_cast<const uint32_t *>(ctx->enc_table.p) + 768,
                         static_cast<uint32_t>(padded / 4));
    ET_HIP(hipGetLastError());
    et::launch_encode(ctx->stream, g.base, g.lo, g.hi, g.rpt, g.n_tiles, static_cast<const unsigned long long *>(ctx->tile_off.p),
                      static_cast<const uint2 *>(ctx->enc_table.p), cb->max_length, out32, timed(ctx, ev_scan, ev_body));  // K4 carries its two events
    ET_HIP(hipGetLastError());
    return ET_OK;
}

int fetch_histogram(et_ctx *ctx) {
    if (ctx->hist_on_host) return ET_OK;
    // k_hist_reduce stores the totals into h_hist and, behind them, one "done" word per workgroup (this sits between
    // the two halves of every encode)
    for (uint32_t w = 0; w < et::HIST_REDUCE_GROUPS; ++w)
        ET_TRY(wait_for_word<uint64_t>(ctx, ctx->h_hist + 256 + w, ctx->hist_epoch, 100.0, "the histogram never reached the host"));
    ctx->hist_on_host = true;
    return ET_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------
extern "C" const char *et_version(void) { return "entreepy-hip 0.1.0 (gfx950; .et format 0x01, reference v1.1.0)"; }

extern "C" const char *et_last_error(const et_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }


extern "C" int et_ctx_create(int device, et_ctx **out) {
    if (!out) return ET_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return ET_ERR_HIP;
    et_ctx *ctx = new (std::nothrow) et_ctx();
    if (!ctx) return ET_ERR_NOMEM;
    ctx->device = device;
    DeviceGuard guard(device);
    bool ok = guard.ok;
    ok = ok && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) == hipSuccess;
    ctx->stream = ctx->own_stream;
    ok = ok && hipStreamCreateWithFlags(&ctx->side.stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->side.fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->side.join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&ctx->h_hist), (256 + et::HIST_REDUCE_GROUPS) * sizeof(uint64_t)) == hipSuccess;
    if (ok) std::memset(ctx->h_hist, 0, (256 + et::HIST_REDUCE_GROUPS) * sizeof(uint64_t));  // (no workgroup's word reads as the first epoch)
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&ctx->h_enc), 768 * sizeof(uint32_t) + HEADER_STAGE) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&ctx->h_header), HEADER_STAGE) == hipSuccess;
    for (int i = 0; i < 2; ++i) {
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&ctx->h_lut_buf[i]), DEC_TABLES_BYTES) == hipSuccess;
    }
    ctx->h_lut = ctx->h_lut_buf[0];
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&ctx->h_scalar), 16 * sizeof(uint64_t)) == hipSuccess;
    if (ok) std::memset(ctx->h_scalar, 0, 16 * sizeof(uint64_t));
    for (int i = 0; i < 2; ++i) ok = ok && hipHostMalloc(reinterpret_cast<void **>(&ctx->h_tw_tree[i]), sizeof(et::TwUpload)) == hipSuccess;
    // timing-only events: no system-scope fence when they complete (hip_runtime_api.h: "for events that
    // are only being used to measure timing"); with the default flags the ten records of an
    // encode+decode cost ~65 us of cache write-backs and waits at 1 GiB
    for (auto &e : ctx->ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableSystemFence) == hipSuccess;
    if (!ok) {
        et_ctx_destroy(ctx);
        return ET_ERR_HIP;
    }
    *out = ctx;
    return ET_OK;
}

extern "C" void et_ctx_destroy(et_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->tile_hist, &ctx->block_hist, &ctx->hist, &ctx->tile_bits, &ctx->tile_off, &ctx->enc_table, &ctx->group_sum,
                      &ctx->sub_state, &ctx->blk_exit, &ctx->blk_count, &ctx->blk_off, &ctx->lut, &ctx->flag,
                      &ctx->worklist, &ctx->lane_maps, &ctx->blk_maps, &ctx->grp_maps, &ctx->blk_in, &ctx->grp_in, &ctx->row_scratch,
                      &ctx->tw_table, &ctx->tw_tree, &ctx->blk_start, &ctx->blk_pub, &ctx->chain_table, &ctx->io_in, &ctx->io_out};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    delete ctx->io;
    void *pinned[] = {ctx->h_hist, ctx->h_enc, ctx->h_header, ctx->h_lut_buf[0], ctx->h_lut_buf[1], ctx->h_scalar, ctx->h_tw_tree[0], ctx->h_tw_tree[1]};
    for (void *p : pinned)
        if (p) (void)hipHostFree(p);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->side.stream) (void)hipStreamDestroy(ctx->side.stream);
    if (ctx->side.fork) (void)hipEventDestroy(ctx->side.fork);
    if (ctx->side.join) (void)hipEventDestroy(ctx->side.join);
    delete ctx;
}

extern "C" int et_ctx_set_stream(et_ctx *ctx, void *hip_stream) {
    if (!ctx) return ET_ERR_ARG;
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return ET_OK;
}

extern "C" int et_ctx_use_own_stream(et_ctx *ctx) {
    if (!ctx) return ET_ERR_ARG;
    ctx->stream = ctx->own_stream;
    return ET_OK;
}

extern "C" void *et_ctx_stream(const et_ctx *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

extern "C" int et_ctx_device(const et_ctx *ctx) { return ctx ? ctx->device : -1; }

extern "C" int et_ctx_set_tile_rounds(et_ctx *ctx, uint32_t rounds) {
    if (!ctx) return ET_ERR_ARG;
    if (rounds > et::MAX_ROUNDS_PER_TILE || (rounds & (rounds - 1))) return ET_ERR_ARG;
    ctx->force_rpt = rounds;
    ctx->hist_text = nullptr;  // tile histograms of another geometry are stale
    return ET_OK;
}

extern "C" int et_ctx_enable_timing(et_ctx *ctx, int on) {
    if (!ctx) return ET_ERR_ARG;
    ctx->timing = on != 0 && on != ET_TIMING_DECODE_BODY;
    ctx->timing_body = on == ET_TIMING_DECODE_BODY;
    ctx->pend_enc = ctx->pend_dec = false;  // (what an earlier mode left to be worked out is gone with its events)
    return ET_OK;
}

extern "C" int et_last_timings_of(et_ctx *ctx, int which, et_timings *out) {
    if (!ctx || !out || which < 0 || which > 1) return ET_ERR_ARG;
    DeviceGuard guard(ctx->device);
    if (which == 0 && ctx->pend_enc) {
        ET_HIP(hipEventSynchronize(ctx->ev[3]));
        if (!ctx->pend_enc_shard) {  // (a shard encode's histogram was a call of its own, hist_ms is its)
            ctx->tm_enc.hist_ms = elapsed(ctx, 0, 1);
            ctx->tm_enc.total_ms = elapsed(ctx, 0, 3);
        }
        ctx->tm_enc.scan_ms = ctx->pend_enc_bits ? elapsed(ctx, 1, 2) : 0.f;  // everything between K1 and K4: histogram reduce, host code construction, tile scan, uploads
        ctx->tm_enc.body_ms = elapsed(ctx, 2, 3);
        ctx->pend_enc = ctx->pend_enc_shard = false;
    }
    if (which == 1 && ctx->pend_dec && ctx->timing_body) {  // the write kernel's own pair is all there is
        ET_HIP(hipEventSynchronize(ctx->ev[EV_DEC + 3]));
        ctx->tm_dec.body_ms = elapsed(ctx, EV_DEC + 2, EV_DEC + 3);
        ctx->pend_dec = false;
    }
    if (which == 1 && ctx->pend_dec) {
        ET_HIP(hipEventSynchronize(ctx->ev[EV_DEC + 3]));
        // events 0/5 = begin/end of the first sweep's main kernel, 2/3 = of the write kernel
        ctx->tm_dec.sync_ms = elapsed(ctx, EV_DEC + 0, EV_DEC + 2);  // everything before the write: sweeps, check, scan
        ctx->tm_dec.scan_ms = elapsed(ctx, EV_DEC + 5, EV_DEC + 2);  // ... of which after the first sweep (repair sweep, verification, scan)
        ctx->tm_dec.body_ms = elapsed(ctx, EV_DEC + 2, EV_DEC + 3);
        ctx->tm_dec.total_ms = elapsed(ctx, EV_DEC + 0, EV_DEC + 3);
        ctx->tm_dec.sync_first_ms = ctx->pend_dec_first ? elapsed(ctx, EV_DEC + 0, EV_DEC + 5) : 0.f;
        ctx->pend_dec = false;
    }
    *out = which == 0 ? ctx->tm_enc : ctx->tm_dec;
    return ET_OK;
}

extern "C" int et_last_timings(et_ctx *ctx, et_timings *out) {
    if (!ctx) return ET_ERR_ARG;
    return et_last_timings_of(ctx, ctx->last_kind, out);
}

extern "C" int et_last_codebook(const et_ctx *ctx, et_codebook *out) {
    if (!ctx || !out) return ET_ERR_ARG;
    if (!ctx->have_cb) return ET_ERR_ARG;
    *out = ctx->last_cb;
    return ET_OK;
}

extern "C" int et_ctx_reserve(et_ctx *ctx, size_t max_text_bytes) {
    if (!ctx) return ET_ERR_ARG;
    DeviceGuard guard(ctx->device);
    const Geometry g = make_geometry(nullptr, reinterpret_cast<const void *>(static_cast<uintptr_t>(15)), max_text_bytes);
    ET_TRY(ensure_encode_ws(ctx, g.n_tiles + 1));  // size-based geometry; a forced smaller tile grows on demand
    // decode: the body is at most ~max_text_bytes (+ header) bytes
    const uint64_t n_subs = (static_cast<uint64_t>(max_text_bytes) + 8192) * 8 / et::SUB_BITS + 2;
    const uint64_t n_blocks = n_subs / et::BLOCK + 2;
    ET_TRY(ensure(ctx, ctx->sub_state, n_subs * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_exit, n_blocks * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_count, n_blocks * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_off, (n_blocks + 1) * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->worklist, (n_blocks + 1) * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->group_sum, (n_blocks / 1024 + 2) * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->lut, DEC_TABLES_BYTES));
    ET_TRY(ensure(ctx, ctx->flag, 64));
    ET_TRY(ensure(ctx, ctx->tw_table, static_cast<size_t>(et::tw_table_entries(et::TW_MAX_NODES)) * sizeof(uint16_t) + 64));
    ET_TRY(ensure(ctx, ctx->tw_tree, sizeof(et::TwUpload)));
    ET_TRY(ensure(ctx, ctx->chain_table, static_cast<size_t>(et::CH_MAX_ENTRIES) * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->blk_start, n_blocks * sizeof(uint32_t)));
    return ET_OK;
}

// ---------------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------------
extern "C" int et_histogram_device(et_ctx *ctx, const void *d_text, size_t n, void *d_hist) {
    if (!ctx || (n && !d_text)) return ET_ERR_ARG;
    DeviceGuard guard(ctx->device);
    if (n == 0) {
        ET_TRY(ensure(ctx, ctx->hist, 256 * sizeof(uint64_t)));
        ET_HIP(hipMemsetAsync(ctx->hist.p, 0, 256 * sizeof(uint64_t), ctx->stream));
        if (d_hist) ET_HIP(hipMemsetAsync(d_hist, 0, 256 * sizeof(uint64_t), ctx->stream));
        std::memset(ctx->h_hist, 0, 256 * sizeof(uint64_t));
        ctx->hist_text = nullptr;
        ctx->hist_empty = true;
        return ET_OK;
    }
    ctx->hist_empty = false;
    const Geometry g = make_geometry(ctx, d_text, n);
    ET_TRY(run_histogram(ctx, d_text, n, g, d_hist));  // (K1 carries events 0 and 1; the reduction stores the totals into d_hist as well: no copy behind it)
    if (ctx->timing) {
        ET_HIP(hipStreamSynchronize(ctx->stream));
        ctx->tm_enc = et_timings{};
        ctx->tm_enc.hist_ms = elapsed(ctx, 0, 1);
        ctx->pend_enc = false;
        ctx->last_kind = 0;
    }
    return ET_OK;
}

namespace {

int encode_shard(et_ctx *ctx, const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap_bytes, uint64_t start_bit,
                 const uint8_t *header, size_t header_len, uint64_t *end_bit) {
    if (!ctx || !cb || !d_out || !end_bit || (n && !d_text)) return ET_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(d_out) & 3) return fail(ctx, ET_ERR_ARG, "d_out must be 4-byte aligned");
    if (header_len > HEADER_STAGE - 4) return fail(ctx, ET_ERR_ARG, "header too long");
    DeviceGuard guard(ctx->device);
    // A shard without text, or with nothing but zero-length symbols, still owns the word its start bit lies
    // in: that word (after the header, padded to a word, for the head shard) is written as zeros, so that a
    // concatenation which ORs pieces together never reads what an earlier call left in d_out.
    auto empty_shard = [&]() -> int {
        const size_t head_words = (header_len + 3) / 4;
        const size_t need = (header_len ? head_words : 1) * 4;
        if (need > cap_bytes) return fail(ctx, ET_ERR_CAP, "body does not fit d_out");
        if (header_len) {
            std::memset(ctx->h_header, 0, head_words * 4);
            std::memcpy(ctx->h_header, header, header_len);
            ET_HIP(hipMemcpyAsync(d_out, ctx->h_header, head_words * 4, hipMemcpyHostToDevice, ctx->stream));
        } else {
            ET_HIP(hipMemsetAsync(d_out, 0, 4, ctx->stream));
        }
        *end_bit = start_bit;
        return ET_OK;
    };
    if (n == 0) {
        ET_HIP(hipStreamSynchronize(ctx->stream));  // pinned staging may still feed an earlier call
        return empty_shard();
    }
    if (ctx->hist_text != d_text || ctx->hist_n != n)
        return fail(ctx, ET_ERR_ARG, "shard encode needs et_histogram_device on the same (d_text, n) first");
    ET_TRY(fetch_histogram(ctx));
    ET_HIP(hipStreamSynchronize(ctx->stream));  // pinned staging may still feed an earlier call
    uint64_t bits = 0;
    et_codebook_bits(cb, ctx->h_hist, &bits);
    const uint64_t end = start_bit + bits;
    if (((end + 31) / 32) * 4 > cap_bytes) return fail(ctx, ET_ERR_CAP, "body does not fit d_out");
    if (header_len) {
        std::memset(ctx->h_header, 0, HEADER_STAGE);
        std::memcpy(ctx->h_header, header, header_len);
    }
    Geometry g = make_geometry(ctx, d_text, n);
    g.rpt = ctx->hist_rpt;
    g.n_tiles = ctx->hist_tiles;
    if (bits == 0) return empty_shard();  // nothing but zero-length symbols
    ET_TRY(run_body(ctx, cb, g, static_cast<uint32_t *>(d_out), start_bit, header_len ? ctx->h_header : nullptr, header_len, 2, 3));
    *end_bit = end;
    if (ctx->timing) {  // hist_ms is already there (et_histogram_device); the rest when asked for
        ctx->pend_enc = true;
        ctx->pend_enc_bits = true;
        ctx->pend_enc_shard = true;
        ctx->last_kind = 0;
    }
    return ET_OK;
}

}  // namespace

extern "C" int et_histogram_host(et_ctx *ctx, uint64_t counts[256]) {
    if (!ctx || !counts) return ET_ERR_ARG;
    if (ctx->hist_empty) {  // an empty shard's: zeros
        std::memset(counts, 0, 256 * sizeof(uint64_t));
        return ET_OK;
    }
    if (!ctx->hist_text) return fail(ctx, ET_ERR_ARG, "no current histogram (et_histogram_device first)");
    DeviceGuard guard(ctx->device);
    ET_TRY(fetch_histogram(ctx));
    std::memcpy(counts, ctx->h_hist, 256 * sizeof(uint64_t));
    return ET_OK;
}

extern "C" int et_histogram_device_ptr(et_ctx *ctx, const void **d_hist) {
    if (!ctx || !d_hist) return ET_ERR_ARG;
    if (!ctx->hist_text && !ctx->hist_empty) return fail(ctx, ET_ERR_ARG, "no current histogram (et_histogram_device first)");
    *d_hist = ctx->hist.p;
    return ET_OK;
}

extern "C" int et_histogram_on_host(et_ctx *ctx, const uint64_t counts[256]) {
    if (!ctx || !counts) return ET_ERR_ARG;
    if (!ctx->hist_text) return fail(ctx, ET_ERR_ARG, "no current histogram (et_histogram_device first)");
    std::memcpy(ctx->h_hist, counts, 256 * sizeof(uint64_t));
    ctx->hist_on_host = true;
    return ET_OK;
}

extern "C" int et_encode_body_device(et_ctx *ctx, const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap_bytes,
                                     uint64_t start_bit, uint64_t *end_bit) {
    return encode_shard(ctx, cb, d_text, n, d_out, cap_bytes, start_bit, nullptr, 0, end_bit);
}

extern "C" int et_encode_head_shard_device(et_ctx *ctx, const et_codebook *cb, const void *d_text, size_t n, void *d_out, size_t cap_bytes,
                                           const uint8_t *header, size_t header_len, uint64_t *end_bit) {
    if (!header || !header_len) return ET_ERR_ARG;
    return encode_shard(ctx, cb, d_text, n, d_out, cap_bytes, static_cast<uint64_t>(header_len) * 8, header, header_len, end_bit);
}

extern "C" int et_encode_device(et_ctx *ctx, const void *d_text, size_t n, void *d_out, size_t cap, size_t *out_len) {
    if (!ctx || !d_out || !out_len || (n && !d_text)) return ET_ERR_ARG;
    if (n == 0) return fail(ctx, ET_ERR_EMPTY, "empty input");
    if (reinterpret_cast<uintptr_t>(d_out) & 15) return fail(ctx, ET_ERR_ARG, "d_out must be 16-byte aligned");
    if (cap < et_encode_bound(n)) return fail(ctx, ET_ERR_CAP, "cap < et_encode_bound(n)");
    DeviceGuard guard(ctx->device);
    const Geometry g = make_geometry(ctx, d_text, n);
    ET_TRY(run_histogram(ctx, d_text, n, g));  // (K1 carries events 0 and 1)
    ET_TRY(fetch_histogram(ctx));
    const double t1 = now_ms();

    et_codebook cb;
    int rc = et_build_codebook(ctx->h_hist, &cb);
    if (rc != ET_OK) return fail(ctx, rc, "et_build_codebook");
    ctx->last_cb = cb;
    ctx->have_cb = true;
    size_t header_len = 0;
    std::memset(ctx->h_header, 0, HEADER_STAGE);
    rc = et_write_header(&cb, n, ctx->h_header, HEADER_STAGE - 4, &header_len);
    if (rc != ET_OK) return fail(ctx, rc, "et_write_header");
    uint64_t bits = 0;
    et_codebook_bits(&cb, ctx->h_hist, &bits);
    const double t2 = now_ms();

    if (bits == 0) {
        // Single distinct symbol: the whole file is the 9-byte header (encode.zig:137-138, :270-275).
        const size_t padded = (header_len + 3) & ~static_cast<size_t>(3);
        ET_HIP(hipMemcpyAsync(d_out, ctx->h_header, padded, hipMemcpyHostToDevice, ctx->stream));
        record(ctx, 2);
        record(ctx, 3);
    } else {
        ET_TRY(run_body(ctx, &cb, g, static_cast<uint32_t *>(d_out), static_cast<uint64_t>(header_len) * 8, ctx->h_header, header_len, 2, 3));
    }
    *out_len = header_len + static_cast<size_t>((bits + 7) / 8);  // encode.zig:318,336
    if (ctx->timing) {
        ctx->tm_enc = et_timings{};
        ctx->tm_enc.host_ms = static_cast<float>(t2 - t1);
        ctx->pend_enc = true;
        ctx->pend_enc_bits = bits != 0;
        ctx->pend_enc_shard = false;
        ctx->last_kind = 0;
    }
    return ET_OK;
}

namespace {

// The staging pipeline of the host-pointer / fd entry points (et_io.h).
int ensure_io(et_ctx *ctx) {
    if (ctx->io) return ET_OK;
    size_t chunk = 32u << 20;
    int threads = static_cast<int>(std::thread::hardware_concurrency() / 2);
    if (threads > 8) threads = 8;
    if (const char *e = std::getenv("ET_IO_CHUNK_MB")) {
        const long v = std::strtol(e, nullptr, 10);
        if (v >= 1 && v <= 1024) chunk = static_cast<size_t>(v) << 20;
    }
    if (const char *e = std::getenv("ET_IO_THREADS")) {
        const long v = std::strtol(e, nullptr, 10);
        if (v >= 1 && v <= 64) threads = static_cast<int>(v);
    }
    ctx->io = new (std::nothrow) et_io::Pipe();
    if (!ctx->io || !ctx->io->init(chunk, threads)) {
        delete ctx->io;
        ctx->io = nullptr;
        return fail(ctx, ET_ERR_NOMEM, "pinned staging buffers");
    }
    return ET_OK;
}

int io_status(et_ctx *ctx, int rc, const char *what) {
    if (rc == 0) return ET_OK;
    if (rc == -1) return fail(ctx, ET_ERR_IO, what);
    return fail(ctx, ET_ERR_HIP, what, ctx->io->last_hip);
}

int file_size(int fd, uint64_t *size) {
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return -1;
    *size = static_cast<uint64_t>(st.st_size);
    return 0;
}

// encode: source -> io_in -> kernels -> io_out -> sink
int encode_through_pipe(et_ctx *ctx, const et_io::HostEnd &src, size_t n, const et_io::HostEnd *dst, size_t cap, size_t *out_len) {
    if (n == 0) return fail(ctx, ET_ERR_EMPTY, "empty input");
    const size_t bound = et_encode_bound(n);
    ET_TRY(ensure_io(ctx));
    ET_TRY(ensure(ctx, ctx->io_in, n + 16));
    ET_TRY(ensure(ctx, ctx->io_out, bound + 16));
    ET_TRY(io_status(ctx, ctx->io->upload(ctx->stream, ctx->io_in.p, src, n), "reading the input"));
    size_t len = 0;
    ET_TRY(et_encode_device(ctx, ctx->io_in.p, n, ctx->io_out.p, bound, &len));
    if (len > cap) return fail(ctx, ET_ERR_CAP, "output buffer too small");
    if (dst) ET_TRY(io_status(ctx, ctx->io->download(ctx->stream, *dst, ctx->io_out.p, len), "writing the output"));
    else ET_HIP(hipStreamSynchronize(ctx->stream));
    *out_len = len;
    return ET_OK;
}

}  // namespace

extern "C" int et_encode(et_ctx *ctx, const uint8_t *text, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (!ctx || !out || !out_len || (n && !text)) return ET_ERR_ARG;
    DeviceGuard guard(ctx->device);
    et_io::HostEnd src, dst;
    src.ptr = const_cast<uint8_t *>(text);
    dst.ptr = out;
    return encode_through_pipe(ctx, src, n, &dst, cap, out_len);
}

extern "C" int et_encode_fd(et_ctx *ctx, int in_fd, int out_fd, size_t *in_len, size_t *out_len) {
    if (!ctx || !in_len || !out_len || in_fd < 0) return ET_ERR_ARG;
    *in_len = *out_len = 0;
    uint64_t n = 0;
    if (file_size(in_fd, &n) != 0) return fail(ctx, ET_ERR_IO, "input is not a regular file");
    DeviceGuard guard(ctx->device);
    et_io::HostEnd src, dst;
    src.fd = in_fd;
    dst.fd = out_fd;
    *in_len = static_cast<size_t>(n);
    return encode_through_pipe(ctx, src, static_cast<size_t>(n), out_fd >= 0 ? &dst : nullptr, ~static_cast<size_t>(0), out_len);
}

extern "C" int et_fd_to_device(et_ctx *ctx, int fd, uint64_t file_offset, size_t len, void *d_dst) {
    if (!ctx || fd < 0 || (len && !d_dst)) return ET_ERR_ARG;
    if (len == 0) return ET_OK;
    DeviceGuard guard(ctx->device);
    ET_TRY(ensure_io(ctx));
    et_io::HostEnd src;
    src.fd = fd;
    src.offset = file_offset;
    return io_status(ctx, ctx->io->upload(ctx->stream, d_dst, src, len), "reading the input");
}

extern "C" int et_device_to_fd(et_ctx *ctx, const void *d_src, size_t len, int fd, uint64_t file_offset) {
    if (!ctx || fd < 0 || (len && !d_src)) return ET_ERR_ARG;
    if (len == 0) return ET_OK;
    DeviceGuard guard(ctx->device);
    ET_TRY(ensure_io(ctx));
    et_io::HostEnd dst;
    dst.fd = fd;
    dst.offset = file_offset;
    return io_status(ctx, ctx->io->download(ctx->stream, dst, d_src, len), "writing the output");
}

// ---------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------
namespace {

// Build the decode tables on the host and upload them (one pinned block, one device block,
// one copy): the step tables of the register-window kernels (k_dec_sync_reg: index
// step_bits, symbol-free; k_dec_write_reg: index lut_bits_write, two symbols) and ONE set of
// first/second-level tables + long list in the older format (index lut_bits_write, two
// symbols per entry) for the LDS-window kernels -- first/last blocks, ranges, the exhaustive
// path -- and the slow path of the step walks.  (A three-symbol set for the counting kernels
// used to be built as well: 12 us of host time per call for kernels that now see three
// blocks of a stream; near-fixed-length codes, the exhaustive path's domain, rarely fit two
// codes in an index anyway.)
using et::HostDecodeTables;
using et::build_decode_tables;
using et::build_step_table;
using et::build_write_step_table;

// zero16 / zeroed (optional): 16 device words the table-building kernel clears on its way, and
// whether it did (the host-built variant has no kernel: the caller clears them itself).
int prepare_decode_tables(et_ctx *ctx, const et_codebook *cb, et::DecodeTables *tb_out, et::DecodeTables *tb_write_out, uint32_t *zero16 = nullptr,
                          bool *zeroed = nullptr) {
    if (zeroed) *zeroed = false;
    // one pinned block, one device block, one upload: [first-level x 2 | long lists | second-level (+ lengths) x 2]
    ET_TRY(ensure(ctx, ctx->lut, DEC_TABLES_BYTES));
    ET_TRY(ensure(ctx, ctx->flag, 64));
    // Two pinned blocks used in turn, and no wait here: every caller waits for something enqueued
    // behind this upload before it returns (the decode for its flags, the range calls and the
    // self-test for the stream), so the upload from the block filled two calls ago is long done and
    // the host can fill this one while the stream is still busy with whatever precedes this decode.
    const int turn = ctx->lut_turn ^= 1;
    ctx->h_lut = ctx->h_lut_buf[turn];
    static const bool on_host = [] { const char *e = std::getenv("ET_DEC_TABLES_HOST"); return e && e[0] == '1'; }();  // the host builders instead of k_build_dec_tables (they are what the device's tables are tested against)
    HostDecodeTables ht, hw;
    uint32_t *h_lut_w = ctx->h_lut + (1u << et::DEC_LUT_BITS_MAX);
    uint32_t *h_long = ctx->h_lut + (2u << et::DEC_LUT_BITS_MAX), *h_long_w = h_long + 512;
    uint16_t *h_sub = reinterpret_cast<uint16_t *>(h_long + 1024);
    uint16_t *h_sub_w = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(h_sub) + SUB_TABLE_BYTES);
    uint32_t *h_steps = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ctx->h_lut) + DEC_STEPS_OFFSET);
    uint32_t step_bits = 0, step_sub_bits = 0, n_step_sub = 0, wstep_bits = 0, wstep_sub_bits = 0, n_wstep_sub = 0;
    et::TablePlan plan;
    if (on_host) {
        build_decode_tables(cb, ctx->lut_bits_write, et::DEC_WRITE_SYMS, h_lut_w, h_long_w, h_sub_w, &hw);
        std::memcpy(reinterpret_cast<uint8_t *>(h_sub_w) + SUB_TABLE_ONLY, cb->length, 256);
        step_bits = build_step_table(cb, ctx->step_bits, h_steps, &step_sub_bits, &n_step_sub);
    } else {  // the host only decides (widths, second-level tables, long-list order); k_build_dec_tables fills
        et::plan_tables(cb, ctx->lut_bits_write, et::DEC_WRITE_SYMS, ctx->step_bits, ctx->lut_bits_write, &plan);
        hw = HostDecodeTables{plan.lut_bits, plan.n_long, plan.sub_bits, plan.n_sub};
        step_bits = plan.step_bits;
        step_sub_bits = plan.step_sub_bits;
        n_step_sub = plan.n_step_sub;
        wstep_bits = plan.wstep_bits;
        wstep_sub_bits = plan.wstep_sub_bits;
        n_wstep_sub = plan.n_wstep_sub;
    }
    ht = hw;
    const size_t step_bytes = (((static_cast<size_t>(1) << step_bits) + (static_cast<size_t>(n_step_sub) << step_sub_bits) + 3) & ~static_cast<size_t>(3)) * sizeof(uint32_t);
    uint32_t *h_wsteps = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(h_steps) + step_bytes);  // right behind, one upload
    if (on_host) wstep_bits = build_write_step_table(cb, ctx->lut_bits_write, h_wsteps, &wstep_sub_bits, &n_wstep_sub);
    const size_t wstep_bytes = (((static_cast<size_t>(1) << wstep_bits) + (static_cast<size_t>(n_wstep_sub) << wstep_sub_bits) + 3) & ~static_cast<size_t>(3)) * sizeof(uint32_t);
    uint32_t *d_lut = static_cast<uint32_t *>(ctx->lut.p);
    uint32_t *d_long = d_lut + (2u << et::DEC_LUT_BITS_MAX);
    uint8_t *subt = reinterpret_cast<uint8_t *>(d_long + 1024);
    *tb_out = et::DecodeTables{d_lut + (1u << et::DEC_LUT_BITS_MAX), d_long + 512, reinterpret_cast<const uint16_t *>(subt + SUB_TABLE_BYTES),
                               subt + SUB_TABLE_BYTES + SUB_TABLE_ONLY, ht.lut_bits, ht.n_long, ht.sub_bits,
                               ht.n_sub, reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(d_lut) + DEC_STEPS_OFFSET), step_bits,
                               step_sub_bits, n_step_sub, nullptr};
    *tb_write_out = et::DecodeTables{d_lut + (1u << et::DEC_LUT_BITS_MAX), d_long + 512,
                                     reinterpret_cast<const uint16_t *>(subt + SUB_TABLE_BYTES), subt + SUB_TABLE_BYTES + SUB_TABLE_ONLY,
                                     hw.lut_bits, hw.n_long, hw.sub_bits, hw.n_sub,
                                     reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(d_lut) + DEC_STEPS_OFFSET + step_bytes), wstep_bits,
                                     wstep_sub_bits, n_wstep_sub, nullptr};
    // device copies of the two structs ride behind the tables (slow path of the step walks), the plan behind them
    const size_t structs_at = DEC_STEPS_OFFSET + step_bytes + wstep_bytes;
    const et::DecodeTables *d_structs = reinterpret_cast<const et::DecodeTables *>(reinterpret_cast<const uint8_t *>(d_lut) + structs_at);
    tb_out->dev_copy = d_structs;
    tb_write_out->dev_copy = d_structs + 1;
    et::DecodeTables *h_structs = reinterpret_cast<et::DecodeTables *>(reinterpret_cast<uint8_t *>(ctx->h_lut) + structs_at);
    h_structs[0] = *tb_out;
    h_structs[1] = *tb_write_out;
    if (on_host) {
        ET_HIP(hipMemcpyAsync(ctx->lut.p, ctx->h_lut, structs_at + 2 * sizeof(et::DecodeTables), hipMemcpyHostToDevice, ctx->stream));
    } else {
        std::memcpy(h_structs + 2, &plan, sizeof plan);
        uint8_t *d_block = reinterpret_cast<uint8_t *>(d_lut);
        ET_HIP(hipMemcpyAsync(d_block + structs_at, h_structs, 2 * sizeof(et::DecodeTables) + sizeof plan, hipMemcpyHostToDevice, ctx->stream));
        et::launch_build_dec_tables(ctx->stream, reinterpret_cast<const et::TablePlan *>(d_block + structs_at + 2 * sizeof(et::DecodeTables)),
                                    d_lut + (1u << et::DEC_LUT_BITS_MAX), d_long + 512, reinterpret_cast<uint16_t *>(subt + SUB_TABLE_BYTES),
                                    subt + SUB_TABLE_BYTES + SUB_TABLE_ONLY, reinterpret_cast<uint32_t *>(d_block + DEC_STEPS_OFFSET),
                                    reinterpret_cast<uint32_t *>(d_block + DEC_STEPS_OFFSET + step_bytes), zero16);
        ET_HIP(hipGetLastError());
        if (zeroed) *zeroed = zero16 != nullptr;
    }
    return ET_OK;
}

}  // namespace

extern "C" int et_selftest_decode_tables(et_ctx *ctx, const et_codebook *cb, int *where) {
    if (!ctx || !cb || !where) return ET_ERR_ARG;
    *where = 0;
    if (cb->max_length > 32 || cb->n_coded == 0) return fail(ctx, ET_ERR_UNSUPPORTED, "no decode tables for this code table");
    DeviceGuard guard(ctx->device);
    et::DecodeTables tb, tbw;
    ET_TRY(prepare_decode_tables(ctx, cb, &tb, &tbw));  // the device's (unless ET_DEC_TABLES_HOST=1: then this compares the host's with themselves)
    std::vector<uint8_t> dev(DEC_TABLES_BYTES);
    ET_HIP(hipMemcpyAsync(dev.data(), ctx->lut.p, DEC_TABLES_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> lut(1u << et::DEC_LUT_BITS_MAX), longc(512), steps((1u << et::DEC_STEP_BITS_MAX) + et::DEC_STEP_SUB_WORDS + 8),
        wsteps((1u << et::DEC_LUT_BITS_MAX) + et::DEC_STEP_SUB_WORDS + 8);
    std::vector<uint16_t> sub(SUB_TABLE_ONLY / 2);
    HostDecodeTables hw;
    build_decode_tables(cb, ctx->lut_bits_write, et::DEC_WRITE_SYMS, lut.data(), longc.data(), sub.data(), &hw);
    uint32_t ssb = 0, nss = 0, wsb = 0, nws = 0;
    const uint32_t sbits = build_step_table(cb, ctx->step_bits, steps.data(), &ssb, &nss);
    const uint32_t wbits = build_write_step_table(cb, ctx->lut_bits_write, wsteps.data(), &wsb, &nws);
    auto at = [&](const void *dptr) { return dev.data() + (static_cast<const uint8_t *>(dptr) - static_cast<const uint8_t *>(ctx->lut.p)); };
    const bool meta_ok = hw.lut_bits == tbw.lut_bits && hw.n_long == tbw.n_long && hw.sub_bits == tbw.sub_bits && hw.n_sub == tbw.n_sub &&
                         sbits == tb.step_bits && ssb == tb.step_sub_bits && nss == tb.n_step_sub && wbits == tbw.step_bits &&
                         wsb == tbw.step_sub_bits && nws == tbw.n_step_sub;
    if (!meta_ok) *where = 7;
    else if (std::memcmp(at(tbw.lut), lut.data(), sizeof(uint32_t) << hw.lut_bits)) *where = 1;
    else if (std::memcmp(at(tbw.longc), longc.data(), 2 * sizeof(uint32_t) * hw.n_long)) *where = 2;
    else if (std::memcmp(at(tbw.sub), sub.data(), (sizeof(uint16_t) * hw.n_sub) << hw.sub_bits)) *where = 3;
    else if (std::memcmp(at(tbw.sym_len), cb->length, 256)) *where = 4;
    else if (std::memcmp(at(tb.steps), steps.data(), sizeof(uint32_t) * ((1u << sbits) + (nss << ssb)))) *where = 5;
    else if (std::memcmp(at(tbw.steps), wsteps.data(), sizeof(uint32_t) * ((1u << wbits) + (nws << wsb)))) *where = 6;
    return *where ? fail(ctx, ET_ERR_FORMAT, "device-built decode tables differ from the host builders'") : ET_OK;
}

extern "C" int et_selftest_treewalk_table(et_ctx *ctx, const et_codebook *cb, uint32_t *first_diff) {
    if (!ctx || !cb || !first_diff) return ET_ERR_ARG;
    *first_diff = 0;
    DeviceGuard guard(ctx->device);
    et::TwUpload *up = ctx->h_tw_tree[0];
    et::TwTree *tree = &up->tree;
    ET_HIP(hipStreamSynchronize(ctx->stream));
    if (et::tw_build_tree(cb, tree) != ET_OK) return fail(ctx, ET_ERR_UNSUPPORTED, "not a full code tree: the tree walk does not apply");
    et::tw_chain_plan(tree, &up->plan);
    const uint32_t entries = et::tw_table_entries(tree->n_int), n_chain = up->plan.n_entries;
    ET_TRY(ensure(ctx, ctx->tw_table, static_cast<size_t>(et::tw_table_entries(et::TW_MAX_NODES)) * sizeof(uint16_t) + 64));
    ET_TRY(ensure(ctx, ctx->tw_tree, sizeof(et::TwUpload)));
    ET_TRY(ensure(ctx, ctx->chain_table, static_cast<size_t>(et::CH_MAX_ENTRIES) * sizeof(uint64_t)));
    ET_HIP(hipMemcpyAsync(ctx->tw_tree.p, up, et::tw_upload_bytes(up), hipMemcpyHostToDevice, ctx->stream));
    et::launch_tw_build(ctx->stream, static_cast<const et::TwUpload *>(ctx->tw_tree.p), static_cast<uint32_t>(et::tw_upload_bytes(up)), tree->n_int, static_cast<uint16_t *>(ctx->tw_table.p), n_chain,
                        static_cast<uint64_t *>(ctx->chain_table.p));
    ET_HIP(hipGetLastError());
    std::vector<uint16_t> dev(entries), host(entries);
    std::vector<uint64_t> dev_chain(n_chain), host_chain(n_chain);
    ET_HIP(hipMemcpyAsync(dev.data(), ctx->tw_table.p, entries * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipMemcpyAsync(dev_chain.data(), ctx->chain_table.p, n_chain * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipStreamSynchronize(ctx->stream));
    et::tw_fill_table(tree, host.data());
    et::tw_chain_fill(tree, &up->plan, host_chain.data());
    for (uint32_t i = 0; i < entries; ++i)
        if (dev[i] != host[i]) {
            *first_diff = i + 1;
            return fail(ctx, ET_ERR_FORMAT, "device-built tree-walk table differs from the host fill");
        }
    for (uint32_t i = 0; i < n_chain; ++i)
        if (dev_chain[i] != host_chain[i]) {
            *first_diff = entries + i + 1;
            return fail(ctx, ET_ERR_FORMAT, "device-built chained write tables differ from the host fill");
        }
    return ET_OK;
}

extern "C" int et_treewalk_table(const et_codebook *cb, uint16_t *table, size_t cap_entries, uint32_t *n_int) {
    if (!cb || !n_int) return ET_ERR_ARG;
    static thread_local et::TwTree tree;
    const int rc = et::tw_build_tree(cb, &tree);
    if (rc != ET_OK) return rc;
    *n_int = tree.n_int;
    if (table) {
        if (cap_entries < et::tw_table_entries(tree.n_int)) return ET_ERR_CAP;
        et::tw_fill_table(&tree, table);
    }
    return ET_OK;
}

extern "C" int et_chain_tables(const et_codebook *cb, uint64_t *table, size_t cap_entries, uint32_t *n_entries, uint32_t *table_first, uint8_t *table_bits,
                               size_t cap_tables, uint32_t *n_tables) {
    if (!cb || !n_entries || !n_tables) return ET_ERR_ARG;
    static thread_local et::TwUpload up;
    const int rc = et::tw_build_tree(cb, &up.tree);
    if (rc != ET_OK) return rc;
    et::tw_chain_plan(&up.tree, &up.plan);
    *n_entries = up.plan.n_entries;
    *n_tables = up.plan.n_tables;
    if (table) {
        if (cap_entries < up.plan.n_entries) return ET_ERR_CAP;
        et::tw_chain_fill(&up.tree, &up.plan, table);
    }
    if (table_first && table_bits) {
        if (cap_tables < up.plan.n_tables) return ET_ERR_CAP;
        for (uint32_t t = 0; t < up.plan.n_tables; ++t) {
            table_first[t] = up.plan.tab[t].first;
            table_bits[t] = up.plan.tab[t].bits;
        }
    }
    return ET_OK;
}

extern "C" int et_row_code(const et_codebook *cb, uint32_t *t) {
    if (!cb || !t) return ET_ERR_ARG;
    et::RowCode rc{};
    if (!et::row_code_of(cb, &rc)) return ET_ERR_UNSUPPORTED;
    *t = rc.t;
    return ET_OK;
}

// Which way a one-GPU decode of a whole stream goes for this code table, before it has seen the stream (et_decode_body_device makes
// the same choices in the same order -- tests/test_gpu_fixedsync.py holds the two against each other; the switches ET_NO_* aside).
extern "C" int et_decode_path(const et_codebook *cb, uint32_t *path) {
    if (!cb || !path) return ET_ERR_ARG;
    if (cb->n_coded == 0) return ET_ERR_ARG;
    if (cb->max_length > 32) return ET_ERR_UNSUPPORTED;
    et::TwTree tree;
    const bool have_tree = et::tw_build_tree(cb, &tree, true) == ET_OK;
    et::RowCode rc{};
    if (!have_tree) *path = ET_PATH_WINDOWS;
    else if (cb->n_coded >= 2 && cb->min_length == cb->max_length && tree.n_int + 1 == cb->n_coded) *path = ET_PATH_FIXED;
    else if (!(cb->max_length <= cb->min_length + 1 && cb->n_coded > 2) || et::quick_to_synchronise(cb)) *path = ET_PATH_TREE_WALK;
    else *path = et::row_code_of(cb, &rc) ? ET_PATH_ROWS : ET_PATH_EXIT_MAPS;
    return ET_OK;
}

extern "C" int et_decode_body_device(et_ctx *ctx, const et_codebook *cb, const void *d_body, size_t body_bytes, uint32_t start_bit,
                                     uint64_t n_symbols, void *d_out, size_t cap, size_t *out_len) {
    if (!ctx || !cb || !out_len) return ET_ERR_ARG;
    *out_len = 0;
    if (cb->max_length > 32) return fail(ctx, ET_ERR_UNSUPPORTED, "code length > 32");
    if (start_bit >= 8) return fail(ctx, ET_ERR_ARG, "start_bit must be < 8");
    if (n_symbols == 0 || body_bytes == 0 || cb->n_coded == 0 || static_cast<uint64_t>(body_bytes) * 8 <= start_bit) return ET_OK;
    if (!d_body || !d_out) return ET_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(d_out) & 15) return fail(ctx, ET_ERR_ARG, "d_out must be 16-byte aligned");
    DeviceGuard guard(ctx->device);

    const uintptr_t a = reinterpret_cast<uintptr_t>(d_body);
    const uint32_t *words = reinterpret_cast<const uint32_t *>(a & ~static_cast<uintptr_t>(3));
    const uint32_t first_bit = static_cast<uint32_t>(a & 3) * 8 + start_bit;
    const uint64_t n_bytes = (a & 3) + body_bytes;  // stream bytes measured from the aligned base
    const uint64_t n_subs = (n_bytes * 8 + et::SUB_BITS - 1) / et::SUB_BITS;
    const uint64_t n_blocks64 = (n_subs + et::BLOCK - 1) / et::BLOCK;
    if (n_blocks64 > 0x7fffffffull) return fail(ctx, ET_ERR_ARG, "body too large");
    const uint32_t n_blocks = static_cast<uint32_t>(n_blocks64);

    ET_TRY(ensure(ctx, ctx->sub_state, n_subs * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_exit, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_count, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_off, (static_cast<size_t>(n_blocks) + 1) * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->worklist, (static_cast<size_t>(n_blocks) + 1) * sizeof(uint32_t)));

    ctx->range.valid = false;  // shares the workspaces
    const double t0 = now_ms();
    // Which tables this decode needs.  A full code tree (an encoder's always is), whatever the stream's size: the
    // tree walk's table and the chained write tables (et_treewalk.h), both filled by ONE small launch
    // from the tree -- the lookup tables of the LDS-window / register-window kernels are then built only if the
    // stream turns out to need them (it does not synchronise: exhaustive path).  Otherwise those, up front.
    et::DecodeTables tb{}, tb_write{};
    ET_TRY(ensure(ctx, ctx->flag, 64));
    bool flags_zeroed = false, have_tables = false;
    auto need_tables = [&](bool zero_flags) -> int {
        if (have_tables) return ET_OK;
        have_tables = true;
        return prepare_decode_tables(ctx, cb, &tb, &tb_write, zero_flags ? static_cast<uint32_t *>(ctx->flag.p) : nullptr, zero_flags ? &flags_zeroed : nullptr);
    };
    // A (nearly) fixed-length code has little to re-synchronise on: unless its mix of L- and (L + 1)-bit codewords says otherwise
    // (et::quick_to_synchronise; the sweep's own verdict still decides: blocks that gave up -> the exit maps), do not even try.
    static const bool quick_off = [] { const char *e = std::getenv("ET_NO_QUICK_SYNC"); return e && e[0] == '1'; }();  // (A/B and the tests of the paths behind it)
    static const bool quick_always = [] { const char *e = std::getenv("ET_QUICK_SYNC_ALWAYS"); return e && e[0] == '1'; }();  // (tools/probe/ab_flat_rule.sh: where does the tree walk stop settling?)
    const bool near_fixed = cb->max_length <= cb->min_length + 1 && cb->n_coded > 2 && !quick_always && (quick_off || !et::quick_to_synchronise(cb));
    bool exhaustive = near_fixed;
    et::TwUpload *h_up = nullptr;
    {
        h_up = ctx->h_tw_tree[ctx->tw_turn ^= 1];  // two pinned blocks in turn, as prepare_decode_tables' (this call waits for its flags before it returns)
        if (et::tw_build_tree(cb, &h_up->tree, true) != ET_OK) h_up = nullptr;  // (bit patterns without a symbol become leaves that decode as byte 0)
    }
    // Fixed-length codes (2^L codewords of L bits: two, four, 16, 64 symbols of about equal weight): where the codewords begin is arithmetic.
    static const bool fixed_off = [] { const char *e = std::getenv("ET_NO_FIXED_SYNC"); return e && e[0] == '1'; }();  // (A/B and the fallback's tests)
    const bool fixed_sync = h_up && !fixed_off && cb->n_coded >= 2 && cb->min_length == cb->max_length && h_up->tree.n_int + 1 == cb->n_coded;
    if (fixed_sync) exhaustive = true;  // (two 1-bit codewords as well: nothing for the tree walk to find)
    const bool tw_sweeps = h_up && !exhaustive;
    // Uniform-like bytes (complete codes of 7 and 8 bits, BASELINE's worst case): one pass by rows and columns (et_rowsync.h)
    // instead of the exit maps for every start offset; the write then goes over the chained tables as for any full tree.
    et::RowCode row_code{};
    static const bool row_off = [] { const char *e = std::getenv("ET_NO_ROW_SYNC"); return e && e[0] == '1'; }();  // (A/B and the fallback's tests)
    const bool row_ok = h_up && !fixed_sync && !row_off && et::row_code_of(cb, &row_code);
    bool row_sync = exhaustive && row_ok;  // (also where a row code that tried the tree walk ends up if its blocks give up, below)
    // ... and so is where symbol i lies: no synchronisation, no scan, no tables -- the write alone (k_fixed_write).  ET_NO_FIXED_WRITE=1 keeps
    // k_fixed_sync's start / count words and the chained-table write behind them (A/B; what a range of such a stream on another GPU would take).
    static const bool fixed_write_off = [] { const char *e = std::getenv("ET_NO_FIXED_WRITE"); return e && e[0] == '1'; }();
    const bool fixed_direct = fixed_sync && !fixed_write_off && cb->max_length <= 8;
    if (!tw_sweeps && !row_sync && !fixed_sync) ET_TRY(need_tables(true));
    const double t1 = now_ms();

    uint32_t *sub_state = static_cast<uint32_t *>(ctx->sub_state.p);
    uint32_t *blk_exit = static_cast<uint32_t *>(ctx->blk_exit.p);
    uint32_t *blk_count = static_cast<uint32_t *>(ctx->blk_count.p);
    uint32_t *flag = static_cast<uint32_t *>(ctx->flag.p);
    unsigned long long *blk_off = static_cast<unsigned long long *>(ctx->blk_off.p);
    uint32_t *worklist = static_cast<uint32_t *>(ctx->worklist.p);
    const et::SideLane *side = &ctx->side;  // the first/last blocks' small launches run beside the large kernels (2.3 % at 1 GiB)

    // D1..D3.  Sweep 0 runs in and repairs inside each block; sweep 1 repairs across blocks
    // (on text ~0.4 % of the block boundaries); the scan that follows also verifies that
    // every block now starts where its predecessor ends (the "sweep that changes nothing").
    // Everything up to the write kernel is enqueued without waiting; the flags and the
    // symbol total reach the host with ONE synchronisation (stored into pinned memory by the scan, no copy), and only if they say so
    // (blocks that do not synchronise -> exhaustive path; verification failed -> more
    // sweeps) is the tail redone.  Device words: flag[0] sweep-1 changed, [1] blocks that
    // gave up, [2] verification failed, [4] / [5] tickets of D1 / D3, [8] worklist count,
    // [12..13] symbol total.
    uint32_t iters = 0;
    ET_TRY(ensure(ctx, ctx->group_sum, (static_cast<size_t>(n_blocks) / 1024 + 2) * sizeof(uint64_t)));
    unsigned long long *group_sum = static_cast<unsigned long long *>(ctx->group_sum.p);
    uint32_t *h_flags = reinterpret_cast<uint32_t *>(ctx->h_scalar + 4);  // host copy of flag[0..15]
    const bool can_speculate = cap >= n_symbols;
    bool wrote = false, write_ticket_zero = false;
    uint32_t *blk_start = nullptr;        // tree-walk sweeps (below)
    const uint16_t *tw_table = nullptr;
    const uint64_t *chain = nullptr;      // chained write tables (below)
    uint32_t tw_n_int = 0, n_chain = 0;
    // the scan's last group stores flags and total into the pinned h_flags and then the launch's epoch into word 14
    auto wait_report = [&]() -> int { return wait_for_word<uint32_t>(ctx, h_flags + 14, ctx->report_epoch, 200.0, "the decode's report never reached the host"); };
    auto scan_and_total = [&](bool verify) -> int {
        // (the scan's last thread stores the flags and the total straight into the pinned h_flags)
        et::launch_dec_scan(ctx->stream, blk_count, n_blocks, group_sum, scan_epoch(ctx), blk_off, reinterpret_cast<unsigned long long *>(flag + 12),
                            verify ? sub_state : nullptr, blk_exit, flag + 2, first_bit, flag, h_flags, false, ++ctx->report_epoch);
        ET_HIP(hipGetLastError());
        return ET_OK;
    };
    bool used_strips = false;
    auto write_symbols = [&](uint64_t clamp, bool speculative) -> int {
        // speculative: the kernel itself looks at the sweeps' flags and does nothing if the state is not final
        static const bool row_write_off = [] { const char *e = std::getenv("ET_NO_ROW_WRITE"); return e && e[0] == '1'; }();  // (A/B: the chained-table write on a row code's stream)
        if (row_sync && !row_write_off) {  // by rows (et_rowsync.h): no table chain, no bank conflicts between the lanes' regions
            const et::KernelEvents ev = timed_body(ctx, EV_DEC + 2, EV_DEC + 3);
            et::launch_row_write(ctx->stream, words, n_bytes, first_bit, n_subs, row_code, cb, sub_state, blk_off, clamp, static_cast<uint8_t *>(d_out), ev.start, ev.stop);
            ET_HIP(hipGetLastError());
            return ET_OK;
        }
        if (fixed_direct) {  // symbol i is the L bits at first_bit + i L (et_rowsync.h): no walk, no state
            const et::KernelEvents ev = timed_body(ctx, EV_DEC + 2, EV_DEC + 3);
            et::launch_fixed_write(ctx->stream, words, n_bytes, first_bit, cb, clamp, static_cast<uint8_t *>(d_out), ev.start, ev.stop);
            ET_HIP(hipGetLastError());
            return ET_OK;
        }
        // More than 128 symbols per 256-bit subsequence (the header says how many symbols the body's bits hold): a quarter's output is three or more
        // windows of the write's 4 KiB stage, i.e. it would be walked three or more times -- the instantiation that walks it once, into strips
        // (measured: +45 % at 140 symbols per subsequence, +75 % at 200; at 90-110, two windows, the strips' scattered stores cost what they save).
        static const bool strips_off = [] { const char *e = std::getenv("ET_NO_STRIPS"); return e && e[0] == '1'; }();  // (A/B and the windows' tests)
        const bool strips = !strips_off && chain && n_symbols / 128 > n_subs;
        used_strips = used_strips || strips;
        et::launch_dec_write(ctx->stream, words, n_bytes, n_subs, tb_write, sub_state, blk_off, clamp, static_cast<uint8_t *>(d_out), flag + 5, side,
                             write_ticket_zero, speculative ? flag : nullptr, timed_body(ctx, EV_DEC + 2, EV_DEC + 3), chain, n_chain, cb->max_length, strips);
        write_ticket_zero = false;
        ET_HIP(hipGetLastError());
        return ET_OK;
    };
    bool more_sweeps = false;
    // The synchronisation sweeps by tree walk and the write walk over chained lookup tables (no escapes).
    if (h_up && !fixed_direct) {
        et::tw_chain_plan(&h_up->tree, &h_up->plan);
        ET_TRY(ensure(ctx, ctx->tw_tree, sizeof(et::TwUpload)));
        ET_TRY(ensure(ctx, ctx->chain_table, static_cast<size_t>(et::CH_MAX_ENTRIES) * sizeof(uint64_t)));
        if (tw_sweeps) {
            ET_TRY(ensure(ctx, ctx->tw_table, static_cast<size_t>(et::tw_table_entries(et::TW_MAX_NODES)) * sizeof(uint16_t) + 64));
            ET_TRY(ensure(ctx, ctx->blk_start, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_pub, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
        }
        tw_n_int = h_up->tree.n_int;
        n_chain = h_up->plan.n_entries;
        const bool zero_here = !flags_zeroed && (!exhaustive || row_sync || fixed_sync);
        // (the kernel reads the tree and the plan from the pinned block itself: no upload in front of it)
        et::launch_tw_build(ctx->stream, h_up, static_cast<uint32_t>(et::tw_upload_bytes(h_up)), tw_n_int, tw_sweeps ? static_cast<uint16_t *>(ctx->tw_table.p) : nullptr, n_chain,
                            static_cast<uint64_t *>(ctx->chain_table.p), zero_here ? flag : nullptr, tw_sweeps ? static_cast<uint32_t *>(ctx->blk_pub.p) : nullptr, n_blocks);
        flags_zeroed = flags_zeroed || zero_here;
        chain = static_cast<const uint64_t *>(ctx->chain_table.p);
        if (tw_sweeps) {
            tw_table = static_cast<const uint16_t *>(ctx->tw_table.p);
            blk_start = static_cast<uint32_t *>(ctx->blk_start.p);
        }
    }
    auto scan_and_total_tw = [&]() -> int {
        et::launch_dec_scan(ctx->stream, blk_count, n_blocks, group_sum, scan_epoch(ctx), blk_off, reinterpret_cast<unsigned long long *>(flag + 12), blk_start, blk_exit, flag + 2, 0u,
                            flag, h_flags, true, ++ctx->report_epoch);
        ET_HIP(hipGetLastError());
        return ET_OK;
    };
    if (!exhaustive) {
        if (!flags_zeroed) ET_HIP(hipMemsetAsync(flag, 0, 16 * sizeof(uint32_t), ctx->stream));
        write_ticket_zero = true;
        if (tw_table) {
            // ONE sweep: the blocks run in, settle inside and then with the block before them (k_tw_sync's blk_pub); what
            // that leaves open -- a block that did not re-synchronise within its 8 KiB -- the verification finds
            et::launch_tw_sync(ctx->stream, words, n_bytes, first_bit, n_subs, tw_table, tw_n_int, sub_state, blk_exit, blk_start, blk_count, flag,
                               et::DEC_FIRST_SWEEP_TRIPS, nullptr, nullptr, timed(ctx, EV_DEC + 0, EV_DEC + 5), static_cast<uint32_t *>(ctx->blk_pub.p));
        } else {
            et::launch_dec_sync(ctx->stream, words, n_bytes, first_bit, n_subs, tb, 0, et::DEC_FIRST_SWEEP_TRIPS, sub_state, blk_exit, blk_count, flag, flag + 4,
                                et::DEC_HAVE_START, nullptr, nullptr, side, true, timed(ctx, EV_DEC + 0, EV_DEC + 5));
            et::launch_dec_sync(ctx->stream, words, n_bytes, first_bit, n_subs, tb, 1, et::DEC_REPAIR_SWEEP_TRIPS, sub_state, blk_exit, blk_count, flag, flag + 4,
                                et::DEC_HAVE_START, worklist, flag + 8, side);
        }
        ET_HIP(hipGetLastError());
        iters = tw_table ? 2 : 3;  // run-in sweep, (repair sweep,) verification
        if (tw_table) ET_TRY(scan_and_total_tw());
        else ET_TRY(scan_and_total(true));
        if (can_speculate) {
            ET_TRY(write_symbols(n_symbols, true));
            wrote = true;
        }
        ET_TRY(wait_report());  // not the stream: the write kernel keeps running while the caller moves on
        exhaustive = static_cast<uint64_t>(h_flags[1]) * 64 > n_blocks;
        row_sync = exhaustive && row_ok;
        more_sweeps = !exhaustive && h_flags[2] != 0;
        if (!et::dec_state_final(h_flags[1], h_flags[2], n_blocks)) wrote = false;  // the speculative launch declined by the same rule
    }
    if (exhaustive && row_sync) {
        if (iters == 0) {
            record(ctx, EV_DEC + 0);
            record(ctx, EV_DEC + 5);
        }
        ET_TRY(ensure(ctx, ctx->row_scratch, et::row_sync_scratch_bytes(n_blocks)));
        et::launch_row_sync(ctx->stream, words, n_bytes, first_bit, n_subs, row_code, ctx->row_scratch.p, flag + 3, sub_state, blk_exit, blk_count);
        ET_HIP(hipGetLastError());
        iters += 1;
    } else if (exhaustive && fixed_sync) {
        if (iters == 0) {
            record(ctx, EV_DEC + 0);
            record(ctx, EV_DEC + 5);
        }
        if (!fixed_direct) {
            et::launch_fixed_sync(ctx->stream, n_bytes, first_bit, n_subs, cb->max_length, sub_state, blk_exit, blk_count);
            ET_HIP(hipGetLastError());
            iters += 1;
        }
    } else if (exhaustive) {
        ET_TRY(need_tables(false));
        // The exhaustive kernels count with the older lookup tables, for which a bit pattern without a symbol is passed
        // over bit by bit; in the chained tables it is a leaf that decodes as byte 0.  The two agree on every stream of a
        // FULL tree (an encoder's) -- for a completed one the write has to count like the synchronisation did.
        if (h_up && h_up->tree.n_int + 1 != cb->n_coded) chain = nullptr;
        if (iters == 0) {  // no first sweep carried the events: plain markers in front of the exhaustive kernels
            record(ctx, EV_DEC + 0);
            record(ctx, EV_DEC + 5);
        }
        const uint32_t n_starts = cb->max_length;
        const uint32_t stride = n_starts <= 8 ? 8 : (n_starts <= 16 ? 16 : 32);
        const size_t n_groups = (static_cast<size_t>(n_blocks) + 255) / 256;
        ET_TRY(ensure(ctx, ctx->lane_maps, n_subs * stride + 64));
        ET_TRY(ensure(ctx, ctx->blk_maps, static_cast<size_t>(n_blocks) * 32 + 64));
        ET_TRY(ensure(ctx, ctx->grp_maps, n_groups * 32 + 64));
        ET_TRY(ensure(ctx, ctx->blk_in, static_cast<size_t>(n_blocks) + 64));
        ET_TRY(ensure(ctx, ctx->grp_in, n_groups + 64));
        et::launch_dec_exhaustive(ctx->stream, words, n_bytes, first_bit, n_subs, tb, n_starts, stride, static_cast<uint8_t *>(ctx->lane_maps.p),
                                  static_cast<uint8_t *>(ctx->blk_maps.p), static_cast<uint8_t *>(ctx->grp_maps.p),
                                  static_cast<uint8_t *>(ctx->blk_in.p), static_cast<uint8_t *>(ctx->grp_in.p), sub_state, blk_exit, blk_count);
        ET_HIP(hipGetLastError());
        iters += 5;
    }
    while (more_sweeps) {
        ET_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), ctx->stream));
        ET_HIP(hipMemsetAsync(flag + 8, 0, sizeof(uint32_t), ctx->stream));
        if (tw_table) {
            et::launch_tw_check(ctx->stream, blk_start, blk_exit, n_blocks, worklist, flag + 8);
            et::launch_tw_sync(ctx->stream, words, n_bytes, first_bit, n_subs, tw_table, tw_n_int, sub_state, blk_exit, blk_start, blk_count, flag, 0xffffffffu,
                               worklist, flag + 8);
        } else
        et::launch_dec_sync(ctx->stream, words, n_bytes, first_bit, n_subs, tb, iters, 0xffffffffu, sub_state, blk_exit, blk_count, flag, flag + 4,
                            et::DEC_HAVE_START, worklist, flag + 8);
        ET_HIP(hipGetLastError());
        ++iters;
        ET_HIP(hipMemcpyAsync(h_flags, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        ET_HIP(hipStreamSynchronize(ctx->stream));
        if (h_flags[0] == 0) break;
        if (iters > n_blocks + 4) return fail(ctx, ET_ERR_HIP, "decode synchronisation did not converge");
    }
    if ((exhaustive || more_sweeps) && !fixed_direct) {
        ET_TRY(scan_and_total(false));
        ET_TRY(wait_report());
        if (row_sync && h_flags[3] != 0) return fail(ctx, ET_ERR_HIP, "the row walk's chunks never saw the chunks before them");
    }
    const uint64_t decodable = fixed_direct ? (n_bytes * 8 >= first_bit ? (n_bytes * 8 - first_bit) / cb->max_length : 0)  // the whole codewords from first_bit on
                                            : static_cast<uint64_t>(h_flags[12]) | (static_cast<uint64_t>(h_flags[13]) << 32);
    const uint64_t n_out = decodable < n_symbols ? decodable : n_symbols;
    if (n_out > cap) return fail(ctx, ET_ERR_CAP, "output buffer too small");
    if (n_out && !wrote) ET_TRY(write_symbols(n_out, false));
    *out_len = static_cast<size_t>(n_out);
    if (ctx->timing || ctx->timing_body) {
        ctx->tm_dec = et_timings{};
        ctx->tm_dec.host_ms = static_cast<float>(t1 - t0);
        ctx->tm_dec.sync_iters = iters;
        ctx->tm_dec.reserved = (exhaustive ? 1u : 0u) | (tw_table ? 2u : 0u) | (chain ? 4u : 0u) | (row_sync ? 8u : 0u) | (fixed_sync ? 16u : 0u) | (used_strips ? 32u : 0u);  // (row_sync: k_row_sync, and k_row_write unless ET_NO_ROW_WRITE; fixed_sync: k_fixed_sync)
        ctx->pend_dec = true;
        ctx->pend_dec_first = iters > 0 && !near_fixed && !fixed_sync;
        ctx->last_kind = 1;
    }
    return ET_OK;
}

extern "C" int et_decode_range_sync(et_ctx *ctx, const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes,
                                    int has_front, int32_t in_start_bit, et_range_info *info) {
    if (!ctx || !cb || !d_range || !info || range_bytes == 0) return ET_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(d_range) & 3) return fail(ctx, ET_ERR_ARG, "d_range must be 4-byte aligned");
    if (tail_bytes && (range_bytes % (et::DEC_BLOCK_WORDS * 4) || tail_bytes < 16)) return fail(ctx, ET_ERR_ARG, "an inner range is a multiple of 8192 bytes with >= 16 bytes after it");
    if (in_start_bit >= 32) return fail(ctx, ET_ERR_ARG, "in_start_bit must be < 32");
    if (in_start_bit < 0 && !has_front) return fail(ctx, ET_ERR_ARG, "an unknown start needs the 16 bytes in front of the range");
    if (cb->max_length > 32) return fail(ctx, ET_ERR_UNSUPPORTED, "code length > 32");
    if (cb->n_coded == 0) return fail(ctx, ET_ERR_ARG, "empty code table");
    DeviceGuard guard(ctx->device);
    const uint32_t *words = static_cast<const uint32_t *>(d_range);
    const uint64_t n_bytes = static_cast<uint64_t>(range_bytes) + tail_bytes;
    const uint64_t n_subs = (static_cast<uint64_t>(range_bytes) * 8 + et::SUB_BITS - 1) / et::SUB_BITS;
    const uint64_t n_blocks64 = (n_subs + et::BLOCK - 1) / et::BLOCK;
    if (n_blocks64 > 0x7fffffffull) return fail(ctx, ET_ERR_ARG, "range too large");
    const uint32_t n_blocks = static_cast<uint32_t>(n_blocks64);
    auto &rs = ctx->range;
    // A full code tree (an encoder's always is): the sweeps of et_decode_body_device -- k_tw_sync with its seam step,
    // told that the words in front of the range are stream bytes and that the first lane runs in like any other unless
    // the caller knows its first bit -- then check + repair launches until no block disagrees with the one before it.
    // A second call for the same range with the predecessor's exit simply sweeps again from that bit.
    {
        et::TwUpload *h_up = ctx->h_tw_tree[ctx->tw_turn ^= 1];
        if (et::tw_build_tree(cb, &h_up->tree, true) == ET_OK) {
            rs.valid = rs.row = false;
            et::tw_chain_plan(&h_up->tree, &h_up->plan);
            ET_TRY(ensure(ctx, ctx->sub_state, n_subs * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_exit, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_count, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_start, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_pub, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_off, (static_cast<size_t>(n_blocks) + 1) * sizeof(uint64_t)));
            ET_TRY(ensure(ctx, ctx->worklist, (static_cast<size_t>(n_blocks) + 1) * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->group_sum, (static_cast<size_t>(n_blocks) / 1024 + 2) * sizeof(uint64_t)));
            ET_TRY(ensure(ctx, ctx->tw_table, static_cast<size_t>(et::tw_table_entries(et::TW_MAX_NODES)) * sizeof(uint16_t) + 64));
            ET_TRY(ensure(ctx, ctx->chain_table, static_cast<size_t>(et::CH_MAX_ENTRIES) * sizeof(uint64_t)));
            ET_TRY(ensure(ctx, ctx->flag, 64));
            uint32_t *sub_state = static_cast<uint32_t *>(ctx->sub_state.p), *blk_exit = static_cast<uint32_t *>(ctx->blk_exit.p);
            uint32_t *blk_count = static_cast<uint32_t *>(ctx->blk_count.p), *blk_start = static_cast<uint32_t *>(ctx->blk_start.p);
            uint32_t *flag = static_cast<uint32_t *>(ctx->flag.p), *worklist = static_cast<uint32_t *>(ctx->worklist.p);
            uint16_t *tw_table = static_cast<uint16_t *>(ctx->tw_table.p);
            unsigned long long *blk_off = static_cast<unsigned long long *>(ctx->blk_off.p);
            uint32_t *h_flags = reinterpret_cast<uint32_t *>(ctx->h_scalar + 2);
            const uint32_t n_int = h_up->tree.n_int, n_chain = h_up->plan.n_entries;
            const bool known = in_start_bit >= 0;
            const uint32_t mode = (has_front ? et::TW_FRONT_OK : 0u) | (known ? 0u : et::TW_START_UNKNOWN);
            const uint32_t first_bit = known ? static_cast<uint32_t>(in_start_bit) : 0u;
            et::launch_tw_build(ctx->stream, h_up, static_cast<uint32_t>(et::tw_upload_bytes(h_up)), n_int, tw_table, n_chain, static_cast<uint64_t *>(ctx->chain_table.p), flag,
                                static_cast<uint32_t *>(ctx->blk_pub.p), n_blocks);
            et::launch_tw_sync(ctx->stream, words, n_bytes, first_bit, n_subs, tw_table, n_int, sub_state, blk_exit, blk_start, blk_count, flag, et::DEC_FIRST_SWEEP_TRIPS,
                               nullptr, nullptr, {}, static_cast<uint32_t *>(ctx->blk_pub.p), mode, flag + 9);
            ET_HIP(hipGetLastError());
            uint32_t sweeps = 1;
            for (;;) {  // (normally one look: nothing on the list)
                ET_HIP(hipMemsetAsync(flag + 8, 0, sizeof(uint32_t), ctx->stream));
                et::launch_tw_check(ctx->stream, blk_start, blk_exit, n_blocks, worklist, flag + 8, known);
                ET_HIP(hipMemcpyAsync(h_flags, flag + 8, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                ET_HIP(hipStreamSynchronize(ctx->stream));
                if (h_flags[0] == 0) break;
                if (sweeps > n_blocks + 4) return fail(ctx, ET_ERR_HIP, "decode synchronisation did not converge");
                et::launch_tw_sync(ctx->stream, words, n_bytes, first_bit, n_subs, tw_table, n_int, sub_state, blk_exit, blk_start, blk_count, flag, 0xffffffffu, worklist,
                                   flag + 8, {}, nullptr, mode, flag + 9);
                ET_HIP(hipGetLastError());
                ++sweeps;
            }
            et::launch_dec_scan(ctx->stream, blk_count, n_blocks, static_cast<unsigned long long *>(ctx->group_sum.p), scan_epoch(ctx), blk_off);
            ET_HIP(hipGetLastError());
            ET_HIP(hipMemcpyAsync(ctx->h_scalar + 1, blk_off + n_blocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            ET_HIP(hipMemcpyAsync(h_flags, sub_state, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            ET_HIP(hipMemcpyAsync(h_flags + 1, flag + 9, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            ET_HIP(hipStreamSynchronize(ctx->stream));
            rs.words = words;
            rs.n_bytes = n_bytes;
            rs.n_subs = n_subs;
            rs.n_blocks = n_blocks;
            rs.total = ctx->h_scalar[1];
            rs.tw = true;
            rs.n_chain = n_chain;
            rs.max_len = cb->max_length;
            rs.valid = true;
            info->start_bit = h_flags[0] & 0xffu;
            info->exit_bit = h_flags[1];
            info->n_symbols = rs.total;
            info->sweeps = sweeps;
            info->reserved = 2;  // tree walk
            return ET_OK;
        }
    }
    rs.tw = rs.row = false;
    const bool repair = rs.valid && rs.words == words && rs.n_subs == n_subs && in_start_bit >= 0;
    uint32_t sweeps = 0;
    if (!repair) {
        rs.valid = false;
        ET_TRY(ensure(ctx, ctx->sub_state, n_subs * sizeof(uint32_t)));
        ET_TRY(ensure(ctx, ctx->blk_exit, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
        ET_TRY(ensure(ctx, ctx->blk_count, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
        ET_TRY(ensure(ctx, ctx->blk_off, (static_cast<size_t>(n_blocks) + 1) * sizeof(uint64_t)));
        ET_TRY(ensure(ctx, ctx->group_sum, (static_cast<size_t>(n_blocks) / 1024 + 2) * sizeof(uint64_t)));
        ET_TRY(prepare_decode_tables(ctx, cb, &rs.tb, &rs.tb_write));
        rs.words = words;
        rs.n_bytes = n_bytes;
        rs.n_subs = n_subs;
        rs.n_blocks = n_blocks;
    }
    rs.flags = (in_start_bit >= 0 ? et::DEC_HAVE_START : 0u) | (has_front ? et::DEC_FRONT_OK : 0u);
    const uint32_t first_bit = in_start_bit >= 0 ? static_cast<uint32_t>(in_start_bit) : 0u;
    uint32_t *sub_state = static_cast<uint32_t *>(ctx->sub_state.p);
    uint32_t *blk_exit = static_cast<uint32_t *>(ctx->blk_exit.p);
    uint32_t *blk_count = static_cast<uint32_t *>(ctx->blk_count.p);
    uint32_t *flag = static_cast<uint32_t *>(ctx->flag.p);
    unsigned long long *blk_off = static_cast<unsigned long long *>(ctx->blk_off.p);
    uint32_t *h_flags = reinterpret_cast<uint32_t *>(ctx->h_scalar + 2);
    if (!repair) {
        // Sweep 0 (run-in, local repair with a trip cap); codes that do not synchronise take
        // many capped sweeps here -- the exhaustive path is single-GPU only for now.
        ET_HIP(hipMemsetAsync(flag, 0, 4 * sizeof(uint32_t), ctx->stream));
        et::launch_dec_sync(ctx->stream, words, n_bytes, first_bit, n_subs, rs.tb, 0, et::DEC_FIRST_SWEEP_TRIPS, sub_state, blk_exit, blk_count, flag,
                            flag + 4, rs.flags);
        ET_HIP(hipGetLastError());
        ++sweeps;
    }
    for (;;) {
        ET_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), ctx->stream));
        et::launch_dec_sync(ctx->stream, words, n_bytes, first_bit, n_subs, rs.tb, 1 + sweeps, 0xffffffffu, sub_state, blk_exit, blk_count, flag, flag + 4,
                            rs.flags);
        ET_HIP(hipGetLastError());
        ++sweeps;
        ET_HIP(hipMemcpyAsync(h_flags, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        ET_HIP(hipStreamSynchronize(ctx->stream));
        if (h_flags[0] == 0) break;
        if (sweeps > n_blocks + 4) return fail(ctx, ET_ERR_HIP, "decode synchronisation did not converge");
    }
    et::launch_dec_scan(ctx->stream, blk_count, n_blocks, static_cast<unsigned long long *>(ctx->group_sum.p), scan_epoch(ctx), blk_off);
    ET_HIP(hipGetLastError());
    ET_HIP(hipMemcpyAsync(ctx->h_scalar + 1, blk_off + n_blocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipMemcpyAsync(h_flags, sub_state, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipMemcpyAsync(h_flags + 1, blk_exit + (n_blocks - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipStreamSynchronize(ctx->stream));
    rs.total = ctx->h_scalar[1];
    rs.valid = true;
    info->start_bit = h_flags[0] & 0xffu;
    info->exit_bit = h_flags[1];
    info->n_symbols = rs.total;
    info->sweeps = sweeps;
    info->reserved = 0;
    return ET_OK;
}

extern "C" int et_decode_range_maps(et_ctx *ctx, const et_codebook *cb, const void *d_range, size_t range_bytes, size_t tail_bytes,
                                    int32_t in_start_bit, uint8_t map[32], uint32_t *n_starts_out) {
    if (!ctx || !cb || !d_range || !map || !n_starts_out || range_bytes == 0) return ET_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(d_range) & 3) return fail(ctx, ET_ERR_ARG, "d_range must be 4-byte aligned");
    if (tail_bytes && (range_bytes % (et::DEC_BLOCK_WORDS * 4) || tail_bytes < 16)) return fail(ctx, ET_ERR_ARG, "an inner range is a multiple of 8192 bytes with >= 16 bytes after it");
    if (in_start_bit >= 32) return fail(ctx, ET_ERR_ARG, "in_start_bit must be < 32");
    if (cb->max_length > 32) return fail(ctx, ET_ERR_UNSUPPORTED, "code length > 32");
    if (cb->n_coded == 0) return fail(ctx, ET_ERR_ARG, "empty code table");
    DeviceGuard guard(ctx->device);
    const uint32_t *words = static_cast<const uint32_t *>(d_range);
    const uint64_t n_bytes = static_cast<uint64_t>(range_bytes) + tail_bytes;
    const uint64_t n_subs = (static_cast<uint64_t>(range_bytes) * 8 + et::SUB_BITS - 1) / et::SUB_BITS;
    const uint64_t n_blocks64 = (n_subs + et::BLOCK - 1) / et::BLOCK;
    if (n_blocks64 > 0x7fffffffull) return fail(ctx, ET_ERR_ARG, "range too large");
    const uint32_t n_blocks = static_cast<uint32_t>(n_blocks64);
    auto &rs = ctx->range;
    rs.valid = rs.maps_valid = rs.tw = rs.row = false;
    {
        // Uniform-like bytes (a complete code of 7- and 8-bit codewords): the range's map by rows and columns -- every chunk publishes
        // its map, the last one composes them (k_row_sync, ROW_MAP_ONLY); the resolve is a second run with the start known.
        et::RowCode row_code{};
        static const bool row_off = [] { const char *e = std::getenv("ET_NO_ROW_SYNC"); return e && e[0] == '1'; }();
        if (!row_off && et::row_code_of(cb, &row_code)) {
            ET_TRY(ensure(ctx, ctx->sub_state, n_subs * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_exit, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_count, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
            ET_TRY(ensure(ctx, ctx->blk_off, (static_cast<size_t>(n_blocks) + 1) * sizeof(uint64_t)));
            ET_TRY(ensure(ctx, ctx->group_sum, (static_cast<size_t>(n_blocks) / 1024 + 2) * sizeof(uint64_t)));
            ET_TRY(ensure(ctx, ctx->row_scratch, et::row_sync_scratch_bytes(n_blocks)));
            ET_TRY(ensure(ctx, ctx->flag, 64));
            uint32_t *flag = static_cast<uint32_t *>(ctx->flag.p);
            ET_HIP(hipMemsetAsync(flag, 0, 16 * sizeof(uint32_t), ctx->stream));
            const bool known = in_start_bit >= 0;
            const unsigned long long *d_map = nullptr;
            et::launch_row_sync(ctx->stream, words, n_bytes, known ? static_cast<uint32_t>(in_start_bit) : 0u, n_subs, row_code, ctx->row_scratch.p, flag + 3,
                                static_cast<uint32_t *>(ctx->sub_state.p), static_cast<uint32_t *>(ctx->blk_exit.p), static_cast<uint32_t *>(ctx->blk_count.p),
                                et::ROW_MAP_ONLY | (known ? 0u : et::ROW_START_UNKNOWN), &d_map);
            ET_HIP(hipGetLastError());
            ET_HIP(hipMemcpyAsync(ctx->h_scalar + 1, d_map, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            ET_HIP(hipMemcpyAsync(ctx->h_scalar + 2, flag + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            ET_HIP(hipStreamSynchronize(ctx->stream));
            if (*reinterpret_cast<const uint32_t *>(ctx->h_scalar + 2) != 0) return fail(ctx, ET_ERR_HIP, "the row walk's chunks never saw the chunks before them");
            const uint64_t m = ctx->h_scalar[1];
            for (uint32_t p = 0; p < 32; ++p) map[p] = static_cast<uint8_t>(p < 8 ? (m >> (8 * p)) & 0xffu : (known ? m & 0xffu : p));
            *n_starts_out = 8;
            rs.words = words;
            rs.n_bytes = n_bytes;
            rs.n_subs = n_subs;
            rs.n_blocks = n_blocks;
            rs.flags = 0;
            rs.row = true;
            rs.row_code = row_code;
            rs.row_cb = *cb;
            rs.maps_const = known;
            rs.maps_valid = true;
            return ET_OK;
        }
    }
    const uint32_t n_starts = cb->max_length;
    const uint32_t stride = n_starts <= 8 ? 8 : (n_starts <= 16 ? 16 : 32);
    const size_t n_groups = (static_cast<size_t>(n_blocks) + 255) / 256;
    ET_TRY(ensure(ctx, ctx->sub_state, n_subs * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_exit, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_count, static_cast<size_t>(n_blocks) * sizeof(uint32_t)));
    ET_TRY(ensure(ctx, ctx->blk_off, (static_cast<size_t>(n_blocks) + 1) * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->group_sum, (static_cast<size_t>(n_blocks) / 1024 + 2) * sizeof(uint64_t)));
    ET_TRY(ensure(ctx, ctx->lane_maps, n_subs * stride + 64));
    ET_TRY(ensure(ctx, ctx->blk_maps, static_cast<size_t>(n_blocks) * 32 + 64));
    ET_TRY(ensure(ctx, ctx->grp_maps, n_groups * 32 + 64));
    ET_TRY(ensure(ctx, ctx->blk_in, static_cast<size_t>(n_blocks) + 64));
    ET_TRY(ensure(ctx, ctx->grp_in, n_groups + 64));
    ET_TRY(prepare_decode_tables(ctx, cb, &rs.tb, &rs.tb_write));
    rs.words = words;
    rs.n_bytes = n_bytes;
    rs.n_subs = n_subs;
    rs.n_blocks = n_blocks;
    rs.flags = 0;
    rs.map_stride = stride;
    rs.maps_const = in_start_bit >= 0;
    et::launch_dec_maps(ctx->stream, words, n_bytes, in_start_bit >= 0 ? static_cast<uint32_t>(in_start_bit) : 0u, rs.maps_const, n_subs, rs.tb, n_starts, stride,
                        static_cast<uint8_t *>(ctx->lane_maps.p), static_cast<uint8_t *>(ctx->blk_maps.p), static_cast<uint8_t *>(ctx->grp_maps.p));
    ET_HIP(hipGetLastError());
    // last level on the host: compose the group maps (32 bytes per 2 MiB of stream)
    std::vector<uint8_t> grp(n_groups * 32);
    ET_HIP(hipMemcpyAsync(grp.data(), ctx->grp_maps.p, grp.size(), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipStreamSynchronize(ctx->stream));
    for (uint32_t p = 0; p < 32; ++p) {
        uint32_t sidx = p;
        if (p < n_starts || rs.maps_const)
            for (size_t g = 0; g < n_groups; ++g) sidx = grp[g * 32 + sidx];
        map[p] = static_cast<uint8_t>(sidx);
    }
    *n_starts_out = n_starts;
    rs.maps_valid = true;
    return ET_OK;
}

extern "C" int et_decode_range_resolve(et_ctx *ctx, uint32_t in_start_bit, et_range_info *info) {
    if (!ctx || !info) return ET_ERR_ARG;
    auto &rs = ctx->range;
    if (!rs.maps_valid) return fail(ctx, ET_ERR_ARG, "et_decode_range_resolve needs et_decode_range_maps first");
    if (in_start_bit >= 32) return fail(ctx, ET_ERR_ARG, "in_start_bit must be < 32");
    DeviceGuard guard(ctx->device);
    uint32_t *sub_state = static_cast<uint32_t *>(ctx->sub_state.p);
    uint32_t *blk_exit = static_cast<uint32_t *>(ctx->blk_exit.p);
    uint32_t *blk_count = static_cast<uint32_t *>(ctx->blk_count.p);
    unsigned long long *blk_off = static_cast<unsigned long long *>(ctx->blk_off.p);
    uint32_t *h_flags = reinterpret_cast<uint32_t *>(ctx->h_scalar + 2);
    if (rs.row) {  // the same walk again, the start known: every lane's start, exit and count
        uint32_t *flag = static_cast<uint32_t *>(ctx->flag.p);
        ET_HIP(hipMemsetAsync(flag, 0, 16 * sizeof(uint32_t), ctx->stream));
        et::launch_row_sync(ctx->stream, rs.words, rs.n_bytes, in_start_bit, rs.n_subs, rs.row_code, ctx->row_scratch.p, flag + 3, sub_state, blk_exit, blk_count);
        et::launch_dec_scan(ctx->stream, blk_count, rs.n_blocks, static_cast<unsigned long long *>(ctx->group_sum.p), scan_epoch(ctx), blk_off);
        ET_HIP(hipGetLastError());
        ET_HIP(hipMemcpyAsync(ctx->h_scalar + 1, blk_off + rs.n_blocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        ET_HIP(hipMemcpyAsync(h_flags, sub_state, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        ET_HIP(hipMemcpyAsync(h_flags + 1, blk_exit + (rs.n_blocks - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        ET_HIP(hipMemcpyAsync(h_flags + 2, flag + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        ET_HIP(hipStreamSynchronize(ctx->stream));
        if (h_flags[2] != 0) return fail(ctx, ET_ERR_HIP, "the row walk's chunks never saw the chunks before them");
        rs.total = ctx->h_scalar[1];
        rs.flags = et::DEC_HAVE_START;
        rs.first_bit = in_start_bit;
        rs.valid = true;
        info->start_bit = h_flags[0] & 0xffu;
        info->exit_bit = h_flags[1];
        info->n_symbols = rs.total;
        info->sweeps = 0;
        info->reserved = 3;  // row walk
        return ET_OK;
    }
    et::launch_dec_resolve(ctx->stream, rs.words, rs.n_bytes, in_start_bit, rs.maps_const, rs.n_subs, rs.tb, rs.map_stride,
                           static_cast<const uint8_t *>(ctx->lane_maps.p), static_cast<const uint8_t *>(ctx->blk_maps.p),
                           static_cast<const uint8_t *>(ctx->grp_maps.p), static_cast<uint8_t *>(ctx->blk_in.p), static_cast<uint8_t *>(ctx->grp_in.p),
                           sub_state, blk_exit, blk_count);
    et::launch_dec_scan(ctx->stream, blk_count, rs.n_blocks, static_cast<unsigned long long *>(ctx->group_sum.p), scan_epoch(ctx), blk_off);
    ET_HIP(hipGetLastError());
    ET_HIP(hipMemcpyAsync(ctx->h_scalar + 1, blk_off + rs.n_blocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipMemcpyAsync(h_flags, sub_state, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipMemcpyAsync(h_flags + 1, blk_exit + (rs.n_blocks - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    ET_HIP(hipStreamSynchronize(ctx->stream));
    rs.total = ctx->h_scalar[1];
    rs.flags = et::DEC_HAVE_START;
    rs.valid = true;
    info->start_bit = h_flags[0] & 0xffu;
    info->exit_bit = h_flags[1];
    info->n_symbols = rs.total;
    info->sweeps = 0;
    info->reserved = 1;  // exhaustive
    return ET_OK;
}

extern "C" int et_decode_range_write(et_ctx *ctx, uint64_t max_symbols, void *d_out, size_t cap, size_t *out_len) {
    if (!ctx || !out_len) return ET_ERR_ARG;
    *out_len = 0;
    auto &rs = ctx->range;
    if (!rs.valid) return fail(ctx, ET_ERR_ARG, "et_decode_range_write needs et_decode_range_sync first");
    const uint64_t n_out = rs.total < max_symbols ? rs.total : max_symbols;
    if (n_out == 0) return ET_OK;
    if (!d_out || (reinterpret_cast<uintptr_t>(d_out) & 15)) return fail(ctx, ET_ERR_ARG, "d_out must be 16-byte aligned");
    if (n_out > cap) return fail(ctx, ET_ERR_CAP, "output buffer too small");
    DeviceGuard guard(ctx->device);
    if (rs.row) {
        et::launch_row_write(ctx->stream, rs.words, rs.n_bytes, rs.first_bit, rs.n_subs, rs.row_code, &rs.row_cb, static_cast<const uint32_t *>(ctx->sub_state.p),
                             static_cast<const unsigned long long *>(ctx->blk_off.p), n_out, static_cast<uint8_t *>(d_out));
        ET_HIP(hipGetLastError());
        *out_len = static_cast<size_t>(n_out);
        return ET_OK;
    }
    et::launch_dec_write(ctx->stream, rs.words, rs.n_bytes, rs.n_subs, rs.tb_write, static_cast<const uint32_t *>(ctx->sub_state.p),
                         static_cast<const unsigned long long *>(ctx->blk_off.p), n_out, static_cast<uint8_t *>(d_out),
                         static_cast<uint32_t *>(ctx->flag.p) + 4, nullptr, false, nullptr, {},
                         rs.tw ? static_cast<const uint64_t *>(ctx->chain_table.p) : nullptr, rs.n_chain, rs.max_len);
    ET_HIP(hipGetLastError());
    *out_len = static_cast<size_t>(n_out);
    return ET_OK;
}

extern "C" int et_decode_device(et_ctx *ctx, const void *d_compressed, size_t len, void *d_out, size_t cap, size_t *out_len) {
    if (!ctx || !d_compressed || !out_len) return ET_ERR_ARG;
    *out_len = 0;
    if (len < 5) return fail(ctx, ET_ERR_FORMAT, "stream shorter than its header");
    DeviceGuard guard(ctx->device);
    // The header and dictionary (<= 4627 bytes after the 4 stripped ones) are parsed on the host.
    const size_t head = len < HEADER_STAGE ? len : HEADER_STAGE;
    // (no wait before the copy: an earlier encode's upload FROM the pinned header stage is
    // ahead of this copy INTO it on the same stream)
    // A one-workgroup kernel stores the bytes into the pinned stage and then a "done" word, which the host polls (a
    // copy command and a stream wait cost ~10 us more, between the two halves of an encode + decode pipeline).
    uint8_t *hdr_data = ctx->h_header;
    volatile uint64_t *done = ctx->h_scalar + 14;
    const uint64_t epoch = ++ctx->header_epoch;
    et::launch_header_to_host(ctx->stream, d_compressed, static_cast<uint32_t>(head), hdr_data, const_cast<unsigned long long *>(reinterpret_cast<volatile unsigned long long *>(done)), epoch);
    ET_HIP(hipGetLastError());
    ET_TRY(wait_for_word<uint64_t>(ctx, done, epoch, 100.0, "the header never reached the host"));
    et_codebook cb;
    uint64_t n_symbols = 0;
    size_t body_offset = 0;
    // Parsing only needs the dictionary (the kernel has sent as many bytes as one with that many entries can
    // have); give the parser the true length when the stream is short so that truncation is detected.
    const size_t sent = std::min<size_t>(head, et::header_bound(hdr_data[0]));
    int rc = et_parse_header(hdr_data, sent, &cb, &n_symbols, &body_offset);
    if (rc != ET_OK) return fail(ctx, rc, "et_parse_header");
    if (body_offset > len) return fail(ctx, ET_ERR_FORMAT, "dictionary runs past the end of the stream");
    return et_decode_body_device(ctx, &cb, static_cast<const uint8_t *>(d_compressed) + body_offset, len - body_offset, 0, n_symbols, d_out, cap, out_len);
}

namespace {

// decode: source -> io_in (header parsed from the bytes as they pass) -> kernels -> io_out -> sink
int decode_through_pipe(et_ctx *ctx, const et_io::HostEnd &src, size_t len, const et_io::HostEnd *dst, size_t cap, size_t *out_len) {
    *out_len = 0;
    if (len < 5) return fail(ctx, ET_ERR_FORMAT, "stream shorter than its header");
    ET_TRY(ensure_io(ctx));
    ET_TRY(ensure(ctx, ctx->io_in, len + 16));
    std::vector<uint8_t> head(len < HEADER_STAGE ? len : HEADER_STAGE);
    ET_TRY(io_status(ctx, ctx->io->upload(ctx->stream, ctx->io_in.p, src, len, head.data(), head.size()), "reading the input"));
    et_codebook cb;
    uint64_t n_symbols = 0;
    size_t body_offset = 0;
    const int rc = et_parse_header(head.data(), head.size(), &cb, &n_symbols, &body_offset);
    if (rc != ET_OK) return fail(ctx, rc, "et_parse_header");
    if (body_offset > len) return fail(ctx, ET_ERR_FORMAT, "dictionary runs past the end of the stream");
    ET_TRY(ensure(ctx, ctx->io_out, n_symbols + 64));
    size_t n_out = 0;
    ET_TRY(et_decode_body_device(ctx, &cb, static_cast<const uint8_t *>(ctx->io_in.p) + body_offset, len - body_offset, 0, n_symbols, ctx->io_out.p,
                                 n_symbols + 64, &n_out));
    if (n_out > cap) return fail(ctx, ET_ERR_CAP, "output buffer too small");
    if (dst && n_out) ET_TRY(io_status(ctx, ctx->io->download(ctx->stream, *dst, ctx->io_out.p, n_out), "writing the output"));
    else ET_HIP(hipStreamSynchronize(ctx->stream));
    *out_len = n_out;
    return ET_OK;
}

}  // namespace

extern "C" int et_decode(et_ctx *ctx, const uint8_t *compressed, size_t len, uint8_t *out, size_t cap, size_t *out_len) {
    if (!ctx || !compressed || !out_len) return ET_ERR_ARG;
    DeviceGuard guard(ctx->device);
    et_io::HostEnd src, dst;
    src.ptr = const_cast<uint8_t *>(compressed);
    dst.ptr = out;
    return decode_through_pipe(ctx, src, len, out ? &dst : nullptr, out ? cap : 0, out_len);
}

extern "C" int et_decode_fd(et_ctx *ctx, int in_fd, size_t in_skip, int out_fd, size_t *in_len, size_t *out_len) {
    if (!ctx || !in_len || !out_len || in_fd < 0) return ET_ERR_ARG;
    *in_len = *out_len = 0;
    uint64_t size = 0;
    if (file_size(in_fd, &size) != 0) return fail(ctx, ET_ERR_IO, "input is not a regular file");
    if (size < in_skip) return fail(ctx, ET_ERR_FORMAT, "file shorter than the bytes to skip");
    DeviceGuard guard(ctx->device);
    et_io::HostEnd src, dst;
    src.fd = in_fd;
    src.offset = in_skip;
    dst.fd = out_fd;
    *in_len = static_cast<size_t>(size - in_skip);
    return decode_through_pipe(ctx, src, *in_len, out_fd >= 0 ? &dst : nullptr, ~static_cast<size_t>(0), out_len);
}
